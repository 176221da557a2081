#!/bin/bash
# One rocprofv3 counter pass over a short program: pmc_pass.sh <outdir> "<counters>" <program> [args...]
# (each pass is a run of its own; at most 8 SQ_ counters per pass)
set -e
out=$1; shift
ctrs=$1; shift
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs -d "$out" -o pmc --output-format csv -- "$@" > "$out.log" 2>&1
python3 tools/pmc_kernel_summary.py $(ls "$out"/*/pmc_counter_collection.csv "$out"/pmc_counter_collection.csv 2>/dev/null | head -1)
