set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
P="timeout -k 10 120 python3 tools/conv_probe.py"
{
for shape in "128 3 1 16 55168" "128 7 3 16 55168" "128 11 5 16 55168" "64 7 3 16 110336" "32 3 1 16 220672" "32 11 5 16 220672"; do
  $P $shape store fp32-split
  VK_DEBUG_SKIP=16 $P $shape store fp32-split
  VK_DEBUG_SKIP=2 $P $shape store fp32-split
  VK_DEBUG_SKIP=1 $P $shape store fp32-split
done
} > $O/split_probe.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc_a -o p --output-format csv -- python3 tools/conv_probe.py 128 7 3 16 55168 store fp32-split 5 > /dev/null 2>&1
python3 tools/pmc_kernel_summary.py $(find $O/pmc_a -name '*counter_collection.csv') > $O/pmc_a.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc_b -o p --output-format csv -- python3 tools/conv_probe.py 32 3 1 16 220672 store fp32-split 5 > /dev/null 2>&1
python3 tools/pmc_kernel_summary.py $(find $O/pmc_b -name '*counter_collection.csv') > $O/pmc_b.txt
cat $O/split_probe.txt $O/pmc_a.txt $O/pmc_b.txt
