#!/bin/bash
# SQ / LDS counters of the chain kernels on the bench forward (one forward of eight videos at a time): one rocprofv3 --pmc pass per
# counter group over the single-stream bench (program directly after --, counters alone with --kernel-trace) -> gpurun_out/pmc_chain.json
# (per kernel: every counter summed over the kernel's dispatches / its dispatch count).  GPU box only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
mkdir -p "$R/gpurun_out"
GROUPS_=("GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "MeanOccupancyPerCU LDSBankConflict")
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf /tmp/pmcc_$i
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmcc_$i -- python3 $R/bench.py --steps 3 --warmup 2 --min-timed-s 0 --videos 1 --batch 8 --no-cpu-baseline --no-post > /tmp/pmcc_$i.log 2>&1 || echo "group $i refused: $g"
  echo "$g" > /tmp/pmcc_$i.names
  i=$((i+1))
done
cd $R && python3 - $i <<'PY' > gpurun_out/pmc_chain.json
import csv, glob, json, re, sys
csv.field_size_limit(1 << 30)
fam = {'k_head_chain<288>': 'k_head_chain<288>', 'k_head_chain<256>': 'k_head_chain<256>', 'k_ffn_chain': 'k_ffn_chain', 'k_ffn_pair': 'k_ffn_pair', 'k_dec_chain': 'k_dec_chain',
       'k_enc_qkv': 'k_enc_qkv', 'k_enc_attn': 'k_enc_attn', 'gemm_bf16s_kernel<1, 4, 4, 2, 1': 'gemm 128x256 k3', 'gemm_bf16s_kernel<1, 4, 2, 2, 2': 'gemm 64x256 channel-major'}
out, refused = {}, []
for i in range(int(sys.argv[1])):
    names = open(f'/tmp/pmcc_{i}.names').read().split()
    fs = glob.glob(f'/tmp/pmcc_{i}/*/*counter_collection.csv')
    if not fs:
        refused += names
        continue
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = next((v for p, v in fam.items() if p in r['Kernel_Name']), None)
        if k is None:
            continue
        d = out.setdefault(k, {}).setdefault(r['Counter_Name'], [0.0, 0])
        d[0] += float(r['Counter_Value']); d[1] += 1
        seen.add(r['Counter_Name'])
    refused += [n for n in names if n not in seen]
res = {k: {c: v[0] / v[1] for c, v in d.items()} | {'dispatches': max(v[1] for v in d.values())} for k, d in out.items()}
print(json.dumps({'per_dispatch': res, 'refused': refused,
                  'note': 'averages per dispatch over the forwards of bench.py --videos 1 --batch 8 --steps 3 --warmup 2 (one 8-video forward at a time, one stream); '
                          'SQ counters are summed over the chip'}, indent=1))
PY
