#!/bin/bash
# per-kernel times of tools/ffn_time.py under rocprofv3 (run on the GPU box): tools/ffn_prof.sh <tag> [rows ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ffnprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ffnprof -- python3 $GRAFT_REPO_ROOT/tools/ffn_time.py "$@" > /tmp/ffnprof.log 2>&1
tail -3 /tmp/ffnprof.log
f=$(ls /tmp/ffnprof/*/*kernel_stats.csv | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp $f $GRAFT_REPO_ROOT/gpurun_out/ffn_${tag}_kernel_stats.csv
python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'ffn_chain' in n or 'gemm_bf16s' in n or 'k_ln' in n or 'row_stats' in n:
        print(f"{n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
P
