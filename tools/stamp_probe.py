#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timing of one conv launch (needs a GPU).
Prints, per phase, the distribution over workgroups and a per-CU timeline summary."""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from vits_amd import native, packing

native.use_diagnostic_library()      # build it with `python -m vits_amd.build --diag`

C, K, DIL, B, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 16, int(sys.argv[4])
PREC = sys.argv[5] if len(sys.argv) > 5 else "fp32"
from vits_amd import engine
engine.set_precision(PREC)
dev = torch.device("cuda:0")
lib = native.lib()
lib.vk_debug_set_stamps.argtypes = [ctypes.c_void_p]
tile = native.TILE_M128 if C >= 128 else (native.TILE_M64 if C == 64 else native.TILE_M32)
ti = native.tile_info(tile)
w = torch.randn(C, C, K) / (C * K) ** 0.5
pc = engine.PackedConv(w, torch.randn(C), dev, dilation=DIL, tile=tile)
wp, bias = pc.w, pc.bias
x = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)
time_tile = {"fp32": ti.time_tile, "fp32-split": 256 if tile == native.TILE_M32 else ti.time_tile, "bf16": ti.time_tile}[PREC]
nblk = ((L + time_tile - 1) // time_tile) * ((C + ti.rows_per_block - 1) // ti.rows_per_block) * B
stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
def run():
    native.conv1d(x, wp, out, c_in=C, c_out=C, taps=K, dilation=DIL, pad=(K - 1) * DIL // 2, tile=tile,
                  epilogue=native.EPI_STORE, bias=bias, act_in=True, in_slope=0.1)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.vk_debug_set_stamps(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
lib.vk_debug_set_stamps(None)
s = stamps.cpu().numpy().reshape(nblk, 8)
t0 = s[:, 0].min()
s = s[s[:, 3] > 0]      # edge workgroups take the predicated epilogue and leave stamp 3 unset
nblk = len(s)
us = (s[:, :4] - t0) / 100.0
print(f"kernel {e0.elapsed_time(e1)*1e3:.1f} us, {nblk} workgroups, span by stamps {us[:,3].max():.1f} us")
for name, a, b in (("prologue", 0, 1), ("main loop", 1, 2), ("epilogue issue", 2, 3), ("whole", 0, 3)):
    dd = us[:, b] - us[:, a]
    print(f"{name:15s} mean {dd.mean():8.2f}  p10 {np.percentile(dd,10):8.2f}  p50 {np.percentile(dd,50):8.2f}  p90 {np.percentile(dd,90):8.2f}  max {dd.max():8.2f}")
bt = s[:, 4] / 100.0
print(f"chunk hand-over (chunks 1..n-1, wave 0) per workgroup: mean {bt.mean():.2f} us  p50 {np.percentile(bt,50):.2f}  p90 {np.percentile(bt,90):.2f}")
hw = s[:, 7] & 0xffffffff
xcc = s[:, 7] >> 32
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
order = np.argsort(us[:, 0])
print("start times (us) of first 24 / around 512 / last 8 workgroups in start order:")
print(np.round(us[order[:24], 0], 1))
print(np.round(us[order[500:530], 0], 1))
print(np.round(us[order[-8:], 0], 1))
# per-CU: how many workgroups, and gaps between consecutive workgroups in the same slot
ucu = np.unique(cu)
print("distinct CUs seen:", len(ucu), " workgroups per CU min/max:", min((cu == c).sum() for c in ucu), max((cu == c).sum() for c in ucu))
c = ucu[0]
idx = np.where(cu == c)[0]
idx = idx[np.argsort(us[idx, 0])]
print("timeline of one CU (start, first-barrier, loop-end, end) us:")
for i in idx:
    print("   ", np.round(us[i], 1), "slot", int(hw[i] & 0xf), "simd", int((hw[i] >> 4) & 3))
