#!/bin/bash
# Ablation builds + in-kernel stamps of the eight-wave FFN kernel (csrc/ffn_chain.hip k_ffn_pair), GPU box only:
#   bash tools/ffn_pair_probe.sh [rows]     -> gpurun_out/r5/ffn_pair_probe.txt
# Builds PRIVATE copies of the library under /tmp (the product .so is not touched): plain, -DDCF_FFN_STAMP, -DDCF_PAIR_NO_DMA
# (no weight-stream requests: wrong results, timing only), -DDCF_PAIR_NO_GELU (no GELU / split vector work), plus whatever
# extra -D flags are given in $PROBE_EXTRA (space separated, one build each).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-131072}
OUT=$R/gpurun_out/r5/ffn_pair_probe.txt
mkdir -p $R/gpurun_out/r5
: > $OUT
P=$R/cvpr2025-decafnet_amd
for v in plain STAMP:-DDCF_FFN_STAMP NODMA:-DDCF_PAIR_NO_DMA NOGELU:-DDCF_PAIR_NO_GELU $PROBE_EXTRA; do
  name=${v%%:*}; flag=${v#*:}; [ "$flag" = "$v" ] && flag=""
  D=/tmp/ffnprobe_$name
  rm -rf $D && mkdir -p $D && cp -r $P $D/ && cp -r $R/tools $D/
  (cd $D/cvpr2025-decafnet_amd && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize $flag -c csrc/ffn_chain.hip -o build/ffn_chain.o \
     && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libdecafnet_hip.so build/*.o)
  echo "== $name ($flag)" >> $OUT
  (cd $D && timeout -k 10 200 python3 - $M $name >> $OUT 2>&1 <<'PY'
import ctypes, importlib, math, sys, os, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module('cvpr2025-decafnet_amd')
lib = pkg._lib.lib(); P = pkg._lib.ptr
M = int(sys.argv[1]); name = sys.argv[2]
E = 256
g = torch.Generator().manual_seed(1)
W1 = (torch.randn(4 * E, E, generator=g) / 16).cuda(); b1 = torch.randn(4 * E, generator=g).cuda()
W2 = (torch.randn(E, 4 * E, generator=g) / 32).cuda(); b2 = torch.randn(E, generator=g).cuda()
X = torch.randn(M, E, generator=g).cuda(); C = torch.empty(M, E, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.dcf_profile_enable.argtypes = [ctypes.c_int]
def run(chain, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg._lib.check(lib.dcf_op_ffn(P(X), None, None, P(W1), P(b1), P(W2), P(b2), None, None, P(C), None, M, E, chain, st))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts[1:])
# the op = two weight splits + the kernel; the splits are the same for every chain value: report the difference to chain = 2 as well
t2, t3 = run(2), run(3)
print(f'M={M}: whole op, four-wave kernel {t2:.1f} us, eight-wave kernel {t3:.1f} us (delta {t3 - t2:+.1f})')
if name == 'STAMP':
    out = (ctypes.c_ulonglong * 16)()
    lib.dcf_debug_pair_stamps.restype = ctypes.c_int
    assert lib.dcf_debug_pair_stamps(out) == 0
    tiles = (M + 127) // 128; mine = (tiles + 255) // 256 if tiles >= 256 else 1
    iters = mine * 33 + 2
    for role, base, names in (('producer wave 0', 0, ['slot loop (A / A+G iterations)', 'handshake + exchange write', 'row split / wait vmcnt(0)', 'barrier', 'slot loop of the reload iteration']),
                              ('consumer wave 4', 8, ['product iterations', '-', 'wait vmcnt(0)', 'barrier', 'epilogue / idle iterations'])):
        tot = sum(out[base:base + 8])
        print(f'  {role}: total {tot} ticks over {iters} iterations ({mine} tiles)')
        for k, n in enumerate(names):
            v = out[base + k]
            if v: print(f'     {n:36s} {v:9d}  {100.0 * v / tot:5.1f} %   per iteration {v / iters:8.1f}')
PY
  ) || echo "   (run failed)" >> $OUT
  # kernel-only durations of the same build (rocprofv3 kernel trace of tools/ffn_time.py's rows)
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ffnprobe_prof_$name && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ffnprobe_prof_$name -o t -- python3 $D/tools/ffn_kernel_only.py $D $M > /dev/null 2>&1 \
     && python3 - /tmp/ffnprobe_prof_$name >> $OUT <<'PY2'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'k_ffn' in n:
        d[n.split('(')[0].replace('void dcf::', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = v[1:]
    print(f'   kernel only: {k:24s} n={len(v)} min {min(v):7.1f} us  median {sorted(v)[len(v) // 2]:7.1f} us')
PY2
  ) || echo "   (kernel trace failed)" >> $OUT
done
cat $OUT
