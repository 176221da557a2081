#!/usr/bin/env python3
"""Diagnostic: time one conv shape (needs a GPU).
  conv_probe.py C K DIL B L [store|resid] [fp32|fp32-split|bf16] [iterations]
Loads the diagnostic library (python -m vits_amd.build --diag), so the VK_* switches apply."""
import os
import sys
import torch
sys.path.insert(0, ".")
from vits_amd import engine, native

native.use_diagnostic_library()

C, K, DIL, B, L = (int(a) for a in sys.argv[1:6])
epi = sys.argv[6] if len(sys.argv) > 6 else "store"
prec = sys.argv[7] if len(sys.argv) > 7 else "fp32"
n = int(sys.argv[8]) if len(sys.argv) > 8 else 20
engine.set_precision(prec)
dev = torch.device("cuda:0")
w = torch.randn(C, C, K) / (C * K) ** 0.5
if os.environ.get('ZERO'):
    w.zero_()
pc = engine.PackedConv(w, torch.randn(C), dev, dilation=DIL)
x = torch.randn(B, C, L, device=dev)
if os.environ.get('ZERO'):
    x.zero_()
res = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)


def run():
    native.conv1d(x, pc.w, out, c_in=C, c_out=C, taps=K, dilation=DIL, pad=pc.pad, tile=pc.tile,
                  epilogue=native.EPI_STORE if epi == "store" else native.EPI_RESID, bias=pc.bias,
                  res=res if epi != "store" else None, act_in=(os.environ.get("NOACT") is None), in_slope=0.1)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * L * C * C * K
print(f"{prec} C={C} k={K} d={DIL} B={B} L={L} {epi}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF  "
      f"act bytes {(2 + (epi != 'store')) * 4 * B * C * L / 1e6:.0f} MB")
