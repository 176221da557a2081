import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.nn.functional as F
from vits_amd import native, packing
from test_conv_kernel import _rand, _mrf_reference, DEV
def run(ch, B, L, kernels, dils):
    mi = native.mrf_info(ch, native.PREC_BF16)
    x = _rand(B, ch, L, seed=201)
    ref_blocks, dev_blocks = [], []
    seed = 210
    for k in kernels:
        rp, dp = [], []
        for d in dils:
            w1 = _rand(ch, ch, k, seed=seed) / (ch * k) ** 0.5
            w2 = _rand(ch, ch, k, seed=seed + 1) / (ch * k) ** 0.5
            b1, b2 = _rand(ch, seed=seed + 2) * 0.3, _rand(ch, seed=seed + 3) * 0.3
            seed += 4
            rp.append((w1, b1, w2, b2, d))
            dp.append((packing.pack_conv_weight_bf16(w1, mi.ck_bf16, mi.rows_per_block).to(DEV), b1.to(DEV),
                       packing.pack_conv_weight_bf16(w2, mi.ck_bf16, mi.rows_per_block).to(DEV), b2.to(DEV), d))
        ref_blocks.append((k, rp)); dev_blocks.append((k, dp))
    ref = _mrf_reference(x, ref_blocks)
    out = torch.full((B, ch, L), float("nan"), device=DEV)
    native.mrf_stage(x.to(DEV), out, dev_blocks, slope=0.1)
    e = (out.cpu() - ref).abs()
    print(ch, B, L, kernels, dils, "max", e.max().item(), "mean", e.mean().item())
    percol = e.amax(dim=(0, 1))
    print(" per 16 columns max:", " ".join(f"{percol[i:i+16].max().item():.1e}" for i in range(0, min(L, 640), 16)))
    print(" per channel max:", " ".join(f"{v:.1e}" for v in e.amax(dim=(0, 2)).tolist()))
run(32, 1, 257, (3, 7, 11), (1, 3, 5))
run(32, 1, 257, (3,), (1,))
run(32, 1, 257, (3,), (1, 3, 5))
run(32, 1, 257, (11,), (1, 3, 5))
run(32, 1, 600, (3, 7, 11), (1, 3, 5))
run(64, 1, 600, (3, 7, 11), (1, 3, 5))
