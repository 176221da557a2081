#!/bin/bash
# per-family table of one bf16 step (B=16, T=862) + the step time, and the bf16 kernel tests
set -e
BENCH_VERBOSE=1 timeout -k 10 200 python -u bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-other-precisions --no-other-configs --corpus 0 2> /tmp/fam.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'parity', d['parity']['max_abs'])"
grep "^#" /tmp/fam.err
