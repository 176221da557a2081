#!/usr/bin/env python3
"""Which calls show up as __amd_rocclr_copyBuffer in a kernel trace?  Prints a marker line per phase;
run under rocprofv3 --kernel-trace and count the copies per phase from the trace's timestamps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import native, packing
from vits_amd.engine import choose_tile
dev = "cuda:0"
x = torch.randn(1, 256, 2064, device=dev); out = torch.empty_like(x)
w = torch.randn(256, 256, 3) / 30
tile = choose_tile(256); ti = native.tile_info(tile)
wp = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block).to(dev)
torch.cuda.synchronize()
mode = sys.argv[1]
n = 50
if mode == "conv_split":
    for _ in range(n): native.conv1d(x, wp, out, c_in=256, c_out=256, taps=3, dilation=1, pad=1, tile=tile, epilogue=native.EPI_STORE)
elif mode == "conv_nosplit":
    native.SPLIT_SMALL_LAUNCHES = False
    for _ in range(n): native.conv1d(x, wp, out, c_in=256, c_out=256, taps=3, dilation=1, pad=1, tile=tile, epilogue=native.EPI_STORE)
elif mode == "empty":
    for _ in range(n): torch.empty_like(x)
elif mode == "event":
    for _ in range(n):
        e = torch.cuda.Event(); e.record()
torch.cuda.synchronize()
