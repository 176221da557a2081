#!/usr/bin/env python3
"""Diagnostic: time one conv shape (needs a GPU).  conv_probe.py C K DIL B L [epi]"""
import os
import sys
import torch
sys.path.insert(0, ".")
from vits_amd import native, packing

native.use_diagnostic_library()      # build it with `python -m vits_amd.build --diag`
from vits_amd.engine import choose_tile

C, K, DIL, B, L = (int(a) for a in sys.argv[1:6])
epi = sys.argv[6] if len(sys.argv) > 6 else "store"
dev = torch.device("cuda:0")
tile = choose_tile(C)
ti = native.tile_info(tile)
w = torch.randn(C, C, K) / (C * K) ** 0.5
wp = packing.pack_conv_weight(w, ti.ck, ti.rows_per_block).to(dev)
bias = torch.randn(C, device=dev)
x = torch.randn(B, C, L, device=dev)
res = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)
def run():
    native.conv1d(x, wp, out, c_in=C, c_out=C, taps=K, dilation=DIL, pad=(K - 1) * DIL // 2, tile=tile,
                  epilogue=native.EPI_STORE if epi == "store" else native.EPI_RESID, bias=bias,
                  res=res if epi != "store" else None, act_in=(os.environ.get("NOACT") is None), in_slope=0.1)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * L * C * C * K
print(f"C={C} k={K} d={DIL} B={B} L={L} {epi}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF  act bytes {(2 + (epi != 'store')) * 4 * B * C * L / 1e6:.0f} MB")
