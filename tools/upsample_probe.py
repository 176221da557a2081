#!/usr/bin/env python3
"""Diagnostic: time the polyphase transposed convolution of one decoder stage (needs a GPU).
upsample_probe.py C_IN C_OUT STRIDE B L [fp32|bf16] [library]     (VK_DEBUG_SKIP=1 drops the epilogue: timing only)"""
import sys
import torch
sys.path.insert(0, ".")
from vits_amd import native, packing

if len(sys.argv) > 7:
    native.DIAG_LIB_PATH = sys.argv[7]
native.use_diagnostic_library()      # build it with `python -m vits_amd.build --diag`
from vits_amd.engine import choose_tile

cin, cout, s, B, L = (int(a) for a in sys.argv[1:6])
dev = torch.device("cuda:0")
wt = torch.randn(cin, cout, 2 * s) / (cin * 2) ** 0.5
poly = packing.transposed_to_polyphase(wt, s)
tile = choose_tile(poly.shape[0])
ti = native.tile_info(tile)
bf16 = len(sys.argv) > 6 and sys.argv[6] == "bf16"
order = packing.polyphase_row_order(cout, s)
wp = (packing.pack_conv_weight_bf16(poly, ti.ck_bf16, ti.rows_per_block, order) if bf16
      else packing.pack_conv_weight(poly, ti.ck, ti.rows_per_block, order)).to(dev)
x = torch.randn(B, cin, L, device=dev)
bias = torch.randn(cout, device=dev)
out = torch.empty(B, cout, L * s, device=dev)
def run():
    native.conv1d(x, wp, out, c_in=cin, c_out=cout, taps=2, dilation=1, pad=1, tile=tile,
                  epilogue=native.EPI_UPSAMPLE, bias=bias, act_in=True, in_slope=0.1, up_stride=s)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * L * cout * s * cin * 2
print(f"upsample {cin}->{cout} x{s} B={B} L={L}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF  out {4*B*cout*L*s/1e6:.0f} MB")
