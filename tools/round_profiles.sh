#!/bin/bash
# everything profiles/ holds for a round, on the GPU box (via gpurun): bash tools/round_profiles.sh <tag>
# (writes gpurun_out/<tag>_*; copy what is to be kept into profiles/ afterwards)
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R
# traffic first: bench.py reports roofline.traffic / hbm_budget only from a PMC summary collected on the sources it runs (csrc hash)
bash tools/pmc_traffic.sh && cp gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json && cp gpurun_out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
echo "pmc traffic done"
# the one-video timeline next: bench.py reports its dispatch count (profiles/<tag>_dispatches.json, same csrc hash)
bash tools/run_trace.sh $tag --videos 1 --batch 1 --min-timed-s 0 && cp gpurun_out/step_$tag.txt gpurun_out/${tag}_step_timeline.txt
[ -f gpurun_out/${tag}_dispatches.json ] && cp gpurun_out/${tag}_dispatches.json profiles/${tag}_dispatches.json
echo "timeline done"
python3 bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
python3 bench.py --steps 10 --warmup 3 --nq 8 --no-cpu-baseline > gpurun_out/${tag}_bench_n1_nq8.json 2>/dev/null
python3 bench.py --steps 10 --warmup 3 --videos 1 --batch 1 --no-cpu-baseline > gpurun_out/${tag}_bench_n1_one_video.json 2>/dev/null
echo "bench done"
# single-stream kernel statistics: reproduce roofline.achieved from rocprofv3 (tools/solo_check.py)
bash tools/solo_stats.sh $tag
echo "solo stats done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python3 $R/bench.py --steps 10 --warmup 3 --min-timed-s 0 --no-cpu-baseline --no-post > /tmp/kstats.log 2>&1
cp $(ls /tmp/kstats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${tag}_bench_n1_kernel_stats.csv
echo "kernel stats (three streams) done"
cd $R
bash tools/pmc_xattn.sh && cp gpurun_out/pmc_xattn.json gpurun_out/${tag}_pmc_xattn_traffic.json
echo "xattn traffic done"
