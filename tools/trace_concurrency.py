"""Kernel-trace CSV of a multi-stream bench run -> how busy the device is and which kernels stretch when they share it.
usage: python tools/trace_concurrency.py <kernel_trace.csv> <solo_kernel_trace.csv>
Prints, for the last 60 % of the trace (steady state): wall time, union of kernel intervals (time with >= 1 kernel resident),
average number of kernels in flight, and per kernel name the mean duration here vs in the one-video trace."""
import csv, sys, re, collections
csv.field_size_limit(1 << 30)
short = lambda n: re.sub(r'\(.*$', '', n.replace('dcf::', '').replace('void ', ''))[:64]


def load(path):
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in csv.DictReader(open(path))]
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + int(0.4 * (t1 - t0))
    return [r for r in rows if r[0] >= cut]


multi, solo = load(sys.argv[1]), load(sys.argv[2])
wall = multi[-1][1] - multi[0][0]
ev = sorted([(s, 1) for s, e, n in multi] + [(e, -1) for s, e, n in multi])
busy = area = 0
depth, last = 0, ev[0][0]
hist = collections.Counter()
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    hist[depth] += t - last
    depth += d
    last = t
print(f'wall {wall / 1e3:.0f} us, some kernel resident {100 * busy / wall:.1f} %, mean kernels in flight {area / wall:.2f}')
print('time share by number of kernels in flight:', {k: round(100 * v / wall, 1) for k, v in sorted(hist.items())})
dm, ds = collections.defaultdict(list), collections.defaultdict(list)
for s, e, n in multi: dm[n].append(e - s)
for s, e, n in solo: ds[n].append(e - s)
tot_m = sum(sum(v) for v in dm.values())
print(f'{"kernel":64s} {"solo us":>8s} {"multi us":>9s} {"stretch":>8s} {"share of kernel-time":>10s}')
for n, v in sorted(dm.items(), key=lambda kv: -sum(kv[1])):
    a = sum(v) / len(v) / 1e3
    b = sum(ds[n]) / len(ds[n]) / 1e3 if n in ds else float('nan')
    print(f'{n:64s} {b:8.1f} {a:9.1f} {a / b if b == b else float("nan"):8.2f} {100 * sum(v) / tot_m:9.1f}%')
