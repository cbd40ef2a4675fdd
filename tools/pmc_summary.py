"""Summarise a rocprofv3 counter_collection CSV: mean counter values per kernel (name filter optional).
usage: python tools/pmc_summary.py file.csv [substring]"""
import csv, sys, collections, re
csv.field_size_limit(1 << 30)
flt = sys.argv[2] if len(sys.argv) > 2 else 'gemm'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if flt not in n: continue
    n = re.sub(r'\(.*$', '', n.replace('dcf::', '').replace('void ', ''))[:60]
    acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[(n, r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, cs in acc.items():
    d = [v for (k, _), v in dur.items() if k == n]
    print(n, 'dispatches', len(d), 'mean us', sum(d) / len(d) / 1e3)
    for c, v in sorted(cs.items()):
        print(f'   {c:32s} {sum(v) / len(v):16.0f}')
