#!/bin/bash
# HBM-side traffic of the GEMM kernels of one bench run (GPU box, via gpurun): two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes -> gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --min-timed-s 0 --videos 1 --batch 8 --no-cpu-baseline --no-post > /tmp/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_traffic.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) 8 > gpurun_out/pmc_traffic.json
