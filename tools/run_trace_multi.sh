#!/bin/bash
# kernel traces of bench.py with 3 videos in flight and with 1 (GPU box) -> gpurun_out/concurrency.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --videos 3 --no-cpu-baseline --no-post > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --videos 1 --no-cpu-baseline --no-post > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_concurrency.py $(ls /tmp/tr3/*/*_kernel_trace.csv | head -1) $(ls /tmp/tr1/*/*_kernel_trace.csv | head -1) > gpurun_out/concurrency.txt
