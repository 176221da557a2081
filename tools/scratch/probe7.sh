cd $GRAFT_REPO_ROOT
P="timeout -k 10 120 python3 tools/conv_probe.py"
for shape in "128 3 1 16 55168" "128 11 5 16 55168" "32 3 1 16 220672"; do
  for dbg in 0 64 192 448 960 512 256; do
    echo -n "dbg=$dbg  "; VK_DEBUG_SKIP=$dbg $P $shape store fp32-split 2>&1 | grep -v amdgpu.ids
  done
done
