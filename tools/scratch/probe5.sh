set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_conv_kernel.py tests/test_hip_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k "split" 2>&1 | tail -15
bash tools/scratch/probe2.sh v2
