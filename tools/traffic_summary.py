#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, each
collected in a run of its own).  Both counters are in KiB.  On gfx950 FETCH_SIZE tallies a
128-byte request of a wide (16 bytes per lane) streaming read as 64 bytes, so fetched bytes are
doubled; WRITE_SIZE is exact for streaming stores.

Usage: traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    return tot, disp


fetch, fd = per_kernel(sys.argv[1], "FETCH_SIZE")
write, wd = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    n = len(fd[k])
    rd = 2.0 * fetch[k] * 1024 / n
    wr = write.get(k, 0.0) * 1024 / max(1, len(wd.get(k, ())))
    out[k] = dict(dispatches=n, fetch_bytes_per_dispatch=rd, write_bytes_per_dispatch=wr,
                  hbm_bytes_per_dispatch=rd + wr)
for k, v in list(out.items())[:25]:
    print(f"{v['dispatches']:5d}  rd {v['fetch_bytes_per_dispatch'] / 1e6:9.1f} MB  wr {v['write_bytes_per_dispatch'] / 1e6:9.1f} MB  {k[:100]}")

# The same figures keyed by the kernel family names bench.py reports
# (conv_mfma_kernel<A_REP, B_REP, WM, WN, CK, EPI, ACT, VEC> -> "conv_mfma[tile,epilogue,act=..]").
TILES = {("1", "4", "4", "1"): "m128", ("1", "4", "2", "2"): "m64", ("1", "4", "1", "4"): "m32", ("2", "2", "2", "2"): "pair",
         ("1", "2", "4", "1"): "m128n", ("1", "2", "2", "2"): "m64n", ("1", "2", "1", "4"): "m32n", ("2", "1", "2", "2"): "pairn"}
EPIS = ["store", "resid", "resid_acc", "resid_avg", "gate", "wn_mid", "wn_last", "couple_rev", "couple_fwd",
        "upsample", "sample"]
fams = {}
for k, v in out.items():
    if "<" not in k:
        continue
    args = [a.strip() for a in k[k.index("<") + 1:k.index(">")].split(",")]
    if k.startswith("void conv_mfma_kernel"):
        fam = f"conv_mfma[{TILES.get(tuple(args[:4]), '?')},{EPIS[int(args[5])]},act={int(args[6] == 'true')}]"
    elif k.startswith("void conv_bf16_kernel"):
        fam = f"conv_bf16[{TILES.get(tuple(args[:4]), '?')},{EPIS[int(args[5])]},act={int(args[6] == 'true')}]"
    elif k.startswith("void conv_split_kernel"):
        fam = f"conv_split[{TILES.get(tuple(args[:4]), '?')},{EPIS[int(args[5])]},act={int(args[6] == 'true')}]"
    elif k.startswith("void resblock_pair_kernel"):
        fam = f"resblock_pair[c{32 * int(args[1])},{EPIS[int(args[4])]}]"
    elif k.startswith("void resblock_pair_bf16_kernel"):
        fam = f"resblock_pair_bf16[c{32 * int(args[0]) * int(args[2])},{EPIS[int(args[4])]}]"
    elif k.startswith("void mrf_stage_bf16_kernel"):
        fam = f"mrf_stage_bf16[c{32 * int(args[0])}]"
    else:
        continue
    f = fams.setdefault(fam, dict(dispatches=0, fetch=0.0, write=0.0))
    f["dispatches"] += v["dispatches"]
    f["fetch"] += v["fetch_bytes_per_dispatch"] * v["dispatches"]
    f["write"] += v["write_bytes_per_dispatch"] * v["dispatches"]
by_family = {fam: dict(dispatches=f["dispatches"],
                       fetch_bytes_per_launch=f["fetch"] / f["dispatches"],
                       write_bytes_per_launch=f["write"] / f["dispatches"],
                       hbm_bytes_per_launch=(f["fetch"] + f["write"]) / f["dispatches"]) for fam, f in fams.items()}
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vits_amd.build import csrc_hash
    json.dump(dict(csrc_hash=csrc_hash(), note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps 1 --warmup 1`; "
                        "FETCH_SIZE doubled (gfx950 tallies 128-byte read requests as 64 bytes); bytes per launch",
                   by_family=by_family, by_symbol=out), open(sys.argv[3], "w"), indent=1)
