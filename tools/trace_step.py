"""Summarise one steady-state forward from a rocprofv3 --kernel-trace CSV: per-dispatch kernel, grid, duration and the
gap to the previous kernel.  usage: python tools/trace_step.py <kernel_trace.csv> [dispatches.json]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# one step starts at k_text_cls_norm (the first kernel of a forward)
starts = [i for i, n in enumerate(names) if 'k_text_cls_norm' in n]
if len(starts) < 3:
    print('not enough steps', len(starts)); sys.exit(1)
# steady-state = the step with the smallest wall time (graph replay inside the timed region)
best = None
for a, b in zip(starts[:-1], starts[1:]):
    w = int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])
    if best is None or w < best[0]: best = (w, a, b)
_, a, b = best
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
tot = 0
short = lambda n: re.sub(r'\(.*$', '', n.replace('dcf::', '').replace('void ', ''))[:70]
gaps = sum(max(0, int(n_['Start_Timestamp']) - int(p_['End_Timestamp'])) for p_, n_ in zip(step[:-1], step[1:]))
print(f'{len(step)} kernel dispatches of one steady-state forward (the fastest of {len(starts) - 1}), wall {(int(step[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us, '
      f'sum of inter-kernel gaps {gaps / 1e3:.1f} us')
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
    gz = r.get('Grid_Size_Z', '')
    wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {g}x{gz} wg {wg}  {short(r["Kernel_Name"])}')
    prev_end = e
    tot += e - s
print(f'sum of kernel durations {tot / 1e3:.1f} us')
if len(sys.argv) > 2:      # <out.json>: the dispatch count keyed on the kernel sources (bench.py reports it as dispatches_per_forward)
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({'one_video_dispatches': len(step), 'wall_us': (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'csrc_sha16': bench.csrc_hash(),
               'note': 'kernel dispatches of one steady-state one-video forward (rocprofv3 --kernel-trace of bench.py --videos 1 --batch 1, tools/run_trace.sh)'},
              open(sys.argv[2], 'w'), indent=1)
