#!/bin/bash
# Which resource limits the cross-attention core at BASELINE config 2?  One rocprofv3 --pmc pass per counter group over
# tools/xattn_bench.py config2 (program directly after --, counters alone with --kernel-trace) -> gpurun_out/pmc_xattn_stall.json.
# A group the profiler refuses is skipped (its names are listed under "refused").  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
mkdir -p "$R/gpurun_out"
rocprofv3 -L > $R/gpurun_out/pmc_counters_available.txt 2>&1 || true
GROUPS_=("GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU"
         "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
         "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"
         "TCC_EA0_RDREQ_LEVEL_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
         "MemUnitBusy MemUnitStalled" "MeanOccupancyPerCU L2CacheHit" "LDSBankConflict WriteUnitStalled")
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/pmcs_$i
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmcs_$i -- python3 $R/tools/xattn_bench.py config2 > /tmp/pmcs_$i.log 2>&1 || echo "group $i refused: $g"
  echo "$g" > /tmp/pmcs_$i.names
  i=$((i+1))
done
cd $R && python3 - $i <<'PY' > gpurun_out/pmc_xattn_stall.json
import csv, glob, json, sys
csv.field_size_limit(1 << 30)
out, refused = {}, []
for i in range(int(sys.argv[1])):
    names = open(f'/tmp/pmcs_{i}.names').read().split()
    fs = glob.glob(f'/tmp/pmcs_{i}/*/*counter_collection.csv')
    if not fs:
        refused += names
        continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if 'k_xattn_mfma' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for n in names:
        if n in acc:
            out[n] = {'launches': len(acc[n]), 'per_launch': sum(acc[n]) / len(acc[n])}
        else:
            refused.append(n)
print(json.dumps({'kernel': 'k_xattn_mfma T=4096 E=1024 heads=16 Lk=33 B=8 (tools/xattn_bench.py config2: 40 + 100 cold, 40 + 100 warm launches)',
                  'counters': out, 'refused': refused}, indent=1))
PY
