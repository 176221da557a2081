set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
rm -rf $O/pmc_c $O/pmc_d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc_c -o p --output-format csv -- python3 tools/conv_probe.py 128 3 1 16 55168 store fp32-split 5 > /dev/null 2>&1
python3 tools/pmc_kernel_summary.py $(find $O/pmc_c -name '*counter_collection.csv') | grep -A9 conv_split
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc_d -o p --output-format csv -- python3 tools/conv_probe.py 128 3 1 16 55168 store fp32-split 5 > /dev/null 2>&1
python3 tools/pmc_kernel_summary.py $(find $O/pmc_d -name '*counter_collection.csv') | grep -A9 conv_split
timeout -k 10 120 python3 tools/stamp_probe.py 128 3 1 55168 fp32-split 2>&1 | grep -v amdgpu.ids | head -8 || true
