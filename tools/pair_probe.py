#!/usr/bin/env python3
"""Diagnostic: time the fused resblock pair of one shape (needs a GPU).
  pair_probe.py C K DIL B L [bf16|fp32] [iterations] [diagnostic library]"""
import sys
import torch
sys.path.insert(0, ".")
from vits_amd import native, packing

C, K, DIL, B, L = (int(a) for a in sys.argv[1:6])
prec = sys.argv[6] if len(sys.argv) > 6 else "bf16"
n = int(sys.argv[7]) if len(sys.argv) > 7 else 20
if len(sys.argv) > 8:                      # a diagnostic build: path of the library
    native.DIAG_LIB_PATH = sys.argv[8]
    native.use_diagnostic_library()
dev = torch.device("cuda:0")
P = native.PREC_BF16 if prec == "bf16" else native.PREC_F32
pi = native.pair_info(C, P)
pack = (lambda w: packing.pack_conv_weight_bf16(w, pi.ck_bf16, pi.rows_per_block)) if prec == "bf16" else \
       (lambda w: packing.pack_conv_weight(w, pi.ck, pi.rows_per_block))
w1 = pack(torch.randn(C, C, K) / (C * K) ** 0.5).to(dev)
w2 = pack(torch.randn(C, C, K) / (C * K) ** 0.5).to(dev)
b1, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
x = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)


def run():
    native.resblock_pair(x, w1, b1, w2, b2, out, taps=K, dilation=DIL, epilogue=native.EPI_RESID, slope=0.1)


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
fl = 4.0 * B * L * C * C * K
print(f"pair {prec} C={C} k={K} d={DIL} B={B} L={L}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF  "
      f"{12.0 * B * C * L / ms / 1e6:6.0f} GB/s algorithmic")
