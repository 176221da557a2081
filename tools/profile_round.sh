#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace statistics of the headline bench step and the
# two HBM-traffic counter passes (each counter pass is a run of its own).  The corpus, the other
# configurations and the CPU baseline are left out of these runs (their launches would swamp the
# traces); raw traces are reduced to summaries on the box.  Usage:
#   tools/profile_round.sh <tag> [bench.py arguments, e.g. --precision bf16]
#   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json, <tag>_hbm_traffic.{txt,json},
#      <tag>_mfma_util.txt (matrix-pipe busy fraction and effective clock per kernel)
set -e
tag=${1:-rX}
shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
common="--corpus 0 --no-other-configs --no-cpu-baseline --no-other-precisions"
raw=/tmp/prof_$tag
rm -rf $raw; mkdir -p $raw gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $raw/stats -o bench --output-format csv -- \
    python3 bench.py --steps 5 --warmup 2 $common "$@" > gpurun_out/${tag}_bench_under_rocprof.json
cp $(find $raw/stats -name 'bench_kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $raw/fetch -o fetch --output-format csv -- \
    python3 bench.py --steps 1 --warmup 1 $common "$@" > /dev/null
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $raw/write -o write --output-format csv -- \
    python3 bench.py --steps 1 --warmup 1 $common "$@" > /dev/null
python3 tools/traffic_summary.py $(find $raw/fetch -name 'fetch_counter_collection.csv' | head -1) \
    $(find $raw/write -name 'write_counter_collection.csv' | head -1) gpurun_out/${tag}_hbm_traffic.json \
    > gpurun_out/${tag}_hbm_traffic.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $raw/util -o util --output-format csv -- \
    python3 bench.py --steps 1 --warmup 1 $common "$@" > /dev/null
python3 tools/mfma_util.py $(find $raw/util -name 'util_counter_collection.csv' | head -1) > gpurun_out/${tag}_mfma_util.txt
rm -rf $raw
