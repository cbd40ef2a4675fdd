#!/bin/bash
# usage (GPU box): bash tools/run_pmc.sh <tag> "<counters>" <program args...>   -> gpurun_out/pmc_<tag>.csv (counter_collection)
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 "$@" > /tmp/pmc_$tag.log 2>&1
f=$(ls /tmp/pmc_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.csv; else tail -5 /tmp/pmc_$tag.log; fi
