"""Print a bench.py JSON line as a table.  usage: python tools/show_bench.py gpurun_out/x.json [other.json ...]"""
import json, sys
for path in sys.argv[1:]:
    p = json.loads([l for l in open(path) if l.startswith('{')][-1])
    print(f"== {path}: value {p['value'] / 1e6:.3f} M clips/s, {p['ms_per_step']:.3f} ms/step, event_ms/step {p.get('event_ms_per_step', 0):.3f}")
    r = p.get('roofline', {})
    if r:
        print(f"   roofline: {r['achieved']:.1f} TF = {r['frac']:.3f} of {r['peak']:.0f}; stages/forward {r.get('stages_per_forward', r.get('launches_per_forward'))}, dispatches {r.get('dispatches_per_forward')}, non-GEMM share {r.get('non_gemm_share', 0):.3f}")
    for k in ('one_forward_in_flight', 'one_video_per_call', 'nq8', 'xattn_config2', 'parity', 'cpu_baseline'):
        if k in p:
            v = dict(p[k]); v.pop('note', None); v.pop('sample', None); v.pop('thread_sweep', None); v.pop('crosscheck', None)
            print(f'   {k}: {json.dumps(v)[:230]}')
    if 'post' in p:
        print('   post:', json.dumps(p['post'])[:400])
    for k, v in p.get('stages', {}).items():
        print(f"   {k:46s} {v['ms_per_step']:.4f} ms n={v['launches_per_step']:.0f} share={v['share']:.3f} tflops={v['tflops']:.1f} GBps={v['alg_GBps']:.0f}")
