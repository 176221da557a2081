#!/bin/bash
# A/B on one box (devices differ by up to 12 %): the same probes through several builds of the
# library, interleaved and repeated.  ab_probe.sh <library> [<library> ...]
for rep in 1 2; do
  for k in 3 7 11; do
    for V in "$@"; do
      timeout -k 10 100 python tools/pair_probe.py 128 $k 1 16 55168 bf16 20 $V | sed "s|^|$(basename $V): |" || exit 1
    done
  done
done
