#!/usr/bin/env python3
"""Diagnostic: time the fused resblock pair of one shape (needs a GPU).
  pair_probe.py C K DIL B L [bf16|fp32|split] [iterations] [diagnostic library]
With `split` also the two sliced convolution launches the pair replaces."""
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
P = {"bf16": native.PREC_BF16, "split": native.PREC_F32_SPLIT}.get(prec, native.PREC_F32)
pi = native.pair_info(C, P)
pack = (lambda w: packing.pack_conv_weight_bf16(w, pi.ck_bf16, pi.rows_per_block)) if prec == "bf16" else \
       (lambda w: packing.pack_conv_weight_split(w, pi.ck_split, pi.rows_per_block)) if prec == "split" else \
       (lambda w: packing.pack_conv_weight(w, pi.ck, pi.rows_per_block))
wa, wb = (torch.randn(C, C, K) / (C * K) ** 0.5).to(dev), (torch.randn(C, C, K) / (C * K) ** 0.5).to(dev)
w1, w2 = pack(wa), pack(wb)
b1, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
x = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)


def run():
    native.resblock_pair(x, w1, b1, w2, b2, out, taps=K, dilation=DIL, epilogue=native.EPI_RESID, slope=0.1)


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if prec == "split":
    from vits_amd.engine import choose_tile
    tile = choose_tile(C)
    ti = native.tile_info(tile)
    q1 = packing.pack_conv_weight_split(wa, ti.ck_split, ti.rows_per_block)
    q2 = packing.pack_conv_weight_split(wb, ti.ck_split, ti.rows_per_block)
    h, out2 = torch.empty_like(x), torch.empty_like(x)

    def two():
        native.conv1d(x, q1, h, c_in=C, c_out=C, taps=K, dilation=DIL, pad=(K - 1) * DIL // 2, tile=tile,
                      epilogue=native.EPI_STORE, bias=b1, act_in=True, in_slope=0.1)
        native.conv1d(h, q2, out2, c_in=C, c_out=C, taps=K, dilation=1, pad=(K - 1) // 2, tile=tile,
                      epilogue=native.EPI_RESID, bias=b2, res=x, act_in=True, in_slope=0.1)
    ms2 = timed(two)
    run()
    torch.cuda.synchronize()
    print(f"  two sliced convolution launches: {ms2 * 1e3:8.1f} us, max abs difference to the pair {(out - out2).abs().max().item()}")

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
