#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of one fused bf16 resblock-pair launch (needs a GPU and the
diagnostic library, `python -m vits_amd.build --diag`).   pair_stamps.py C K DIL B L [library | -] [bf16|split]"""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from vits_amd import native, packing

C, K, DIL, B, L = (int(a) for a in sys.argv[1:6])
if len(sys.argv) > 6 and sys.argv[6] != "-":
    native.DIAG_LIB_PATH = sys.argv[6]
split = len(sys.argv) > 7 and sys.argv[7] == "split"
native.use_diagnostic_library()
dev = torch.device("cuda:0")
lib = native.lib()
lib.vk_debug_set_stamps.argtypes = [ctypes.c_void_p]
pi = native.pair_info(C, native.PREC_F32_SPLIT if split else native.PREC_BF16)
pack = (lambda w: packing.pack_conv_weight_split(w.to(dev), pi.ck_split, pi.rows_per_block)) if split else \
       (lambda w: packing.pack_conv_weight_bf16(w, pi.ck_bf16, pi.rows_per_block).to(dev))
w1, w2 = pack(torch.randn(C, C, K) / (C * K) ** 0.5), pack(torch.randn(C, C, K) / (C * K) ** 0.5)
b1, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
x = torch.randn(B, C, L, device=dev)
out = torch.empty_like(x)
tno = pi.time_tile
nblk = ((L + tno - 1) // tno) * B
stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)


def run():
    native.resblock_pair(x, w1, b1, w2, b2, out, taps=K, dilation=DIL, epilogue=native.EPI_RESID, slope=0.1)


for _ in range(3):
    run()
torch.cuda.synchronize()
lib.vk_debug_set_stamps(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
lib.vk_debug_set_stamps(None)
s = stamps.cpu().numpy().reshape(nblk, 8)
s = s[s[:, 6] > 0]
t0 = s[:, 0].min()
us = (s[:, :7] - t0) / 100.0
print(f"kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(s)} workgroups of {nblk}, span by stamps {us[:, 6].max():.1f} us")
for name, a, b in (("load + stage chunk 0", 0, 1), ("conv 1 (4 chunks)", 1, 2), ("h hand-over", 2, 3),
                   ("conv 2", 3, 5), ("store", 5, 6), ("whole", 0, 6)):
    dd = us[:, b] - us[:, a]
    print(f"{name:22s} mean {dd.mean():7.2f}  p10 {np.percentile(dd, 10):7.2f}  p50 {np.percentile(dd, 50):7.2f}  "
          f"p90 {np.percentile(dd, 90):7.2f}  max {dd.max():7.2f}")
f = s[:, 4].astype(np.uint64)
if f.any():     # a build with -DVK_PAIR_FINE: wave 0's time inside the first convolution's chunk loop
    parts = [((f >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(np.float64) / 100.0 for i in range(4)]
    for name, v in zip(("wait for the chunk's loads (chunks 1..)", "convert + LDS store (chunks 1..)",
                        "barrier (chunks 1..)", "tap loops (all chunks)"), parts):
        print(f"  {name:42s} mean {v.mean():6.2f} us  p90 {np.percentile(v, 90):6.2f}")
hw = s[:, 7] & 0xffffffff
xcc = s[:, 7] >> 32
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
ucu = np.unique(cu)
print("distinct CUs:", len(ucu), " workgroups per CU min/max:", min((cu == c).sum() for c in ucu), max((cu == c).sum() for c in ucu))
idx = np.where(cu == ucu[3])[0]
idx = idx[np.argsort(us[idx, 0])]
print("timeline of one CU (start, staged, conv1 end, handed over, conv2 end, end) us:")
for i in idx:
    print("   ", np.round(us[i, [0, 1, 2, 3, 5, 6]], 1))
