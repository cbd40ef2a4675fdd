#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/run_trace.sh <tag> [bench args]   -> gpurun_out/step_<tag>.txt
# (a forward is delimited by its first kernel, k_text_cls_norm: pass --videos 1 --batch 1 so that every forward holds ONE video)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-post "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_step.py $(ls gpurun_out/trace_$tag/*/*_kernel_trace.csv | head -1) gpurun_out/${tag}_dispatches.json > gpurun_out/step_$tag.txt
rm -rf gpurun_out/trace_$tag
