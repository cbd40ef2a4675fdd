#!/bin/bash
# Single-stream kernel statistics of the bench forward (GPU box, via gpurun): bash tools/solo_stats.sh <tag>
#   -> gpurun_out/<tag>_bench_n1_solo.json            the JSON line of the same command (roofline from HIP events)
#   -> gpurun_out/<tag>_bench_n1_solo_kernel_stats.csv rocprofv3 --kernel-trace --stats of that command
#   -> gpurun_out/<tag>_solo_check.txt                 sum(2MNK) / sum(duration) of the GEMM rows against roofline.achieved
# One forward of eight videos at a time on ONE stream (--videos 1 --batch 8): kernels never overlap, so the per-kernel
# durations of the CSV are the undisturbed ones and reproduce the JSON's roofline (unlike the 3-stream default run).
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/solo_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/solo_$tag -- python3 $R/bench.py --videos 1 --batch 8 --steps 10 --warmup 3 --min-timed-s 0 --no-cpu-baseline --no-post > $R/gpurun_out/${tag}_bench_n1_solo.json 2> /tmp/solo_$tag.err || { tail -20 /tmp/solo_$tag.err; exit 1; }
cp $(ls /tmp/solo_$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_bench_n1_solo_kernel_stats.csv
cd $R && python3 tools/solo_check.py gpurun_out/${tag}_bench_n1_solo_kernel_stats.csv gpurun_out/${tag}_bench_n1_solo.json | tee gpurun_out/${tag}_solo_check.txt
