#!/bin/bash
set -e
mkdir -p gpurun_out/r2d
K=${1:-3}
timeout -k 10 60 python3 tools/pair_probe.py 128 $K 1 16 55168 bf16 20
bash tools/pmc_pass.sh gpurun_out/r2d/pmc_a_k$K "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" python3 tools/pair_probe.py 128 $K 1 16 55168 bf16 3 | grep -A12 resblock_pair
bash tools/pmc_pass.sh gpurun_out/r2d/pmc_b_k$K "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" python3 tools/pair_probe.py 128 $K 1 16 55168 bf16 3 | grep -A12 resblock_pair
bash tools/pmc_pass.sh gpurun_out/r2d/pmc_c_k$K "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE" python3 tools/pair_probe.py 128 $K 1 16 55168 bf16 3 | grep -A12 resblock_pair
