#!/usr/bin/env python3
"""Diagnostic: time the one-launch multi-receptive-field stage kernel (needs a GPU).
  mrf_probe.py C B L [iterations] [diag|-] [bf16|fp32]      kernel sizes 3/7/11, dilations 1/3/5
Prints the whole stage in one launch and the same stage as one launch per resblock (the sum
carried through `out`), with the difference between the two results."""
import sys
import torch
sys.path.insert(0, ".")
from vits_amd import native, packing

C, B, L = (int(a) for a in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
prec = native.PREC_F32 if (len(sys.argv) > 6 and sys.argv[6] == "fp32") else native.PREC_BF16
if len(sys.argv) > 5 and sys.argv[5] != "-":   # a diagnostic build (path of the library, or "diag" for the default one)
    if sys.argv[5] != "diag":
        native.DIAG_LIB_PATH = sys.argv[5]
    native.use_diagnostic_library()
dev = torch.device("cuda:0")
mi = native.mrf_info(C, prec)


def pack(w):
    if prec == native.PREC_BF16:
        return packing.pack_conv_weight_bf16(w, mi.ck_bf16, mi.rows_per_block).to(dev)
    return packing.pack_conv_weight(w, mi.ck, mi.rows_per_block).to(dev)



kernels = (3, 7, 11) if mi.max_blocks >= 3 else (3,)
blocks = []
for k in kernels:
    pairs = []
    for d in (1, 3, 5):
        w1 = pack(torch.randn(C, C, k) / (C * k) ** 0.5)
        w2 = pack(torch.randn(C, C, k) / (C * k) ** 0.5)
        pairs.append((w1, torch.randn(C, device=dev), w2, torch.randn(C, device=dev), d))
    blocks.append((k, pairs))
x = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)
for _ in range(3):
    native.mrf_stage(x, out, blocks, slope=0.1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    native.mrf_stage(x, out, blocks, slope=0.1)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 4.0 * B * L * C * C * sum(kernels) * 3
print(f"mrf stage C={C} B={B} L={L} kernels {kernels}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF algorithmic")


def per_block(dst):
    for r, blk in enumerate(blocks):
        native.mrf_stage(x, dst, [blk], slope=0.1, acc=dst if r else None,
                         navg=len(blocks) if r + 1 == len(blocks) else 1)


if len(blocks) > 1:
    out2 = torch.empty_like(x)
    for _ in range(3):
        per_block(out2)
    torch.cuda.synchronize()
    print("one launch per resblock: max abs difference", float((out2 - out).abs().max()))
    e0.record()
    for _ in range(n):
        per_block(out2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"  one launch per resblock: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF algorithmic")
    for r, blk in enumerate(blocks):
        e0.record()
        for _ in range(n):
            native.mrf_stage(x, out2, [blk], slope=0.1)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"    kernel size {blk[0]:2d} alone: {us:8.1f} us  {4.0 * B * L * C * C * blk[0] * 3 / us / 1e6:7.1f} TF")
