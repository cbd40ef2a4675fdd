"""per-dispatch durations of the kernels whose name contains argv[2] from a rocprofv3 --kernel-trace --output-format csv directory (argv[1]),
in dispatch order: python tools/kernel_seq.py <dir> <substring> [max rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = 0
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        print('%8.1f us  grid %s  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size', '?'), r['Kernel_Name'][:110]))
        n += 1
        if len(sys.argv) > 3 and n >= int(sys.argv[3]): break
