#!/bin/bash
# Memory-side counters of the chain kernels in the two regimes of profiles/r06_notes.md section 7: one video per call (the workgroups of a launch
# start together and stay in lockstep) against eight videos per forward (four row tiles per CU: they drift apart).  One rocprofv3 --pmc pass per
# counter group and mode (program directly after --, counters alone with --kernel-trace) -> gpurun_out/pmc_ffn_modes.json.  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
mkdir -p "$R/gpurun_out"
GROUPS_=("GRBM_GUI_ACTIVE SQ_WAVES" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum")
i=0
for mode in "1 1" "1 8"; do
  set -- $mode
  for g in "${GROUPS_[@]}"; do
    rm -rf /tmp/pmcf_$i
    rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmcf_$i -- python3 $R/bench.py --steps 3 --warmup 2 --min-timed-s 0 --videos $1 --batch $2 --no-cpu-baseline --no-post > /tmp/pmcf_$i.log 2>&1 || echo "group $i refused: $g"
    echo "$2 $g" > /tmp/pmcf_$i.names
    i=$((i+1))
  done
done
cd $R && python3 - $i <<'PY' > gpurun_out/pmc_ffn_modes.json
import csv, glob, json, sys
csv.field_size_limit(1 << 30)
fam = ['k_ffn_pair', 'k_head_chain<288>', 'k_head_chain<256>', 'k_dec_chain']
out = {}
for i in range(int(sys.argv[1])):
    names = open(f'/tmp/pmcf_{i}.names').read().split()
    mode = 'one_video' if names[0] == '1' else 'eight_videos'
    fs = glob.glob(f'/tmp/pmcf_{i}/*/*counter_collection.csv')
    if not fs:
        out.setdefault('refused', []).append(names[1:])
        continue
    for r in csv.DictReader(open(fs[0])):
        k = next((f for f in fam if f in r['Kernel_Name']), None)
        if k is None:
            continue
        d = out.setdefault(mode, {}).setdefault(k, {}).setdefault(r['Counter_Name'], [0.0, 0])
        d[0] += float(r['Counter_Value']); d[1] += 1
res = {}
for mode, ks in out.items():
    if mode == 'refused':
        res['refused'] = ks
        continue
    res[mode] = {k: {c: v[0] / v[1] for c, v in d.items()} | {'dispatches': max(v[1] for v in d.values())} for k, d in ks.items()}
print(json.dumps(res, indent=1))
PY
