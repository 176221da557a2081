#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace statistics of the default bench run and the
# two HBM-traffic counter passes (each counter pass is a run of its own).  Usage:
#   tools/profile_round.sh <tag> [bench.py arguments, e.g. --precision bf16]   -> gpurun_out/<tag>_{stats,fetch,write}/
set -e
tag=${1:-rX}
shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o bench --output-format csv -- \
    python3 bench.py --steps 5 --warmup 2 --no-other-precisions "$@" > gpurun_out/${tag}_bench_under_rocprof.json
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch -o fetch --output-format csv -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions "$@" > /dev/null
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_write -o write --output-format csv -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions "$@" > /dev/null
