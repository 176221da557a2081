#!/usr/bin/env python3
"""Diagnostic for the persistent sliced-fp32 kernel: per workgroup, prologue and run time and the
time each of its waves spent waiting at the item barriers.  stamp_probe2.py C K DIL L  (B = 16)"""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from vits_amd import engine, native

native.use_diagnostic_library()
C, K, DIL, B, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 16, int(sys.argv[4])
engine.set_precision("fp32-split")
dev = torch.device("cuda:0")
lib = native.lib()
lib.vk_debug_set_stamps.argtypes = [ctypes.c_void_p]
pc = engine.PackedConv(torch.randn(C, C, K) / (C * K) ** 0.5, torch.randn(C), dev, dilation=DIL)
x = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)


def run():
    native.conv1d(x, pc.w, out, c_in=C, c_out=C, taps=K, dilation=DIL, pad=pc.pad, tile=pc.tile,
                  epilogue=native.EPI_STORE, bias=pc.bias, act_in=True, in_slope=0.1)


for _ in range(3):
    run()
torch.cuda.synchronize()
lib.vk_debug_set_stamps(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
lib.vk_debug_set_stamps(None)
s = stamps.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(s)} workgroups")
for name, col in (("start", 0), ("prologue end", 1), ("run end", 2)):
    v = (s[:, col] - t0) / 100.0
    print(f"{name:14s} mean {v.mean():8.2f}  min {v.min():8.2f}  max {v.max():8.2f} us")
for wv in range(4):
    v = s[:, 3 + wv] / 100.0
    print(f"wave {wv} at barriers: mean {v.mean():8.2f}  p10 {np.percentile(v, 10):8.2f}  p90 {np.percentile(v, 90):8.2f} us")
