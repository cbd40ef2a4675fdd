#!/bin/bash
# SQ counters of the f16x3 tile kernel on three level-0 shapes of the five-videos-per-forward bench (M = 81920) (two passes each, tools/run_pmc.sh)
C1="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
C2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
i=0
for shape in "81920 256 256 20 0" "81920 1024 256 20 1" "81920 256 1024 20 0"; do
  i=$((i+1))
  bash tools/run_pmc.sh r2_${i}a "$C1" $GRAFT_REPO_ROOT/tools/gemm_one.py f16 $shape
  bash tools/run_pmc.sh r2_${i}b "$C2" $GRAFT_REPO_ROOT/tools/gemm_one.py f16 $shape
  echo "== shape $shape"; python tools/pmc_summary.py gpurun_out/pmc_r2_${i}a.csv gemm_bf16s; python tools/pmc_summary.py gpurun_out/pmc_r2_${i}b.csv gemm_bf16s
done
