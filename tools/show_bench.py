import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.0f clips/s  ms/step %.3f  event_ms %.3f' % (r['value'], r['ms_per_step'], r.get('event_ms_per_step', 0)))
print('cpu', r.get('cpu_baseline')); print('post', r.get('post')); print('roofline', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.get('roofline', {}).items() if k != 'note'})
for k, v in r.get('stages', {}).items():
    print(f"{k:32s} {v['ms_per_step']:.3f} ms  n={v['launches_per_step']:.0f} share={v['share']:.3f} tflops={v['tflops']:.1f} GB/s={v['alg_GBps']:.0f}")
