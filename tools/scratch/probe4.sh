set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
P="timeout -k 10 120 python3 tools/conv_probe.py"
{
for shape in "128 3 1 16 55168" "128 11 5 16 55168" "64 7 3 16 110336" "32 3 1 16 220672"; do
  $P $shape store fp32-split
  VK_DEBUG_SKIP=32 $P $shape store fp32-split
done
VK_DEBUG_SKIP=32 timeout -k 10 120 python3 tools/stamp_probe.py 128 3 1 55168 fp32-split | head -8
} 2>&1 | grep -v amdgpu.ids > $O/split_probe_prio.txt
cat $O/split_probe_prio.txt
