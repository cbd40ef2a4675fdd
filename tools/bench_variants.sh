#!/bin/bash
# Several variant builds against the product on the default bench workload inside ONE GPU call, round robin.
#   tools/bench_variants.sh "<variant> <variant> ..." [rounds] [stage substring ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
VS=$1; N=${2:-2}; shift; shift
cd "$R" || exit 1
mkdir -p gpurun_out
show() {
python3 - "$1" "${@:2}" <<'P'
import json, sys
d = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
if d is None:
    print('no JSON line'); sys.exit(0)
st = d.get('stages', {})
pick = {k: round(v['ms_per_step'], 4) for k, v in st.items() if any(s in k for s in sys.argv[2:])}
print(f"{d['value'] / 1e6:7.3f} M clips/s  event ms {d.get('event_ms_per_step', 0):.3f}  frac {d['roofline']['frac']:.4f}  {pick}")
P
}
for i in $(seq "$N"); do
  python3 bench.py --no-post --no-cpu-baseline > gpurun_out/ab_prod.json 2> gpurun_out/ab_prod.err; echo -n "product  "; show gpurun_out/ab_prod.json "$@"
  for V in $VS; do
    LIB="$R/cvpr2025-decafnet_amd/build/variants/$V/libdecafnet_hip.so"
    test -f "$LIB" || { echo "no variant $V"; continue; }
    DCF_LIB_PATH=$LIB python3 bench.py --no-post --no-cpu-baseline > gpurun_out/ab_$V.json 2> gpurun_out/ab_$V.err; echo -n "$V  "; show gpurun_out/ab_$V.json "$@"
  done
done
