set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
P="timeout -k 10 120 python3 tools/conv_probe.py"
TAG=${1:-x}
{
for shape in "128 3 1 16 55168" "128 7 3 16 55168" "128 11 5 16 55168" "256 7 1 16 6896" "64 3 1 16 110336" "64 7 3 16 110336" "32 3 1 16 220672" "32 11 5 16 220672"; do
  $P $shape store fp32-split
done
$P 128 7 3 16 55168 resid fp32-split
$P 32 7 3 16 220672 resid fp32-split
} 2>&1 | grep -v amdgpu.ids > $O/split_probe_$TAG.txt
cat $O/split_probe_$TAG.txt
