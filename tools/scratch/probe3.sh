set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
for shape in "128 3 1 55168" "128 11 5 55168" "32 3 1 220672"; do
  timeout -k 10 120 python3 tools/stamp_probe.py $shape fp32-split 2>&1 | grep -v amdgpu.ids | head -12
done > $O/stamps_split.txt
cat $O/stamps_split.txt
