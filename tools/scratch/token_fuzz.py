#!/usr/bin/env python3
"""One-off: random shapes and masks through the token-level kernels against the CPU layer-by-layer pass."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import synth
from vits_amd.module.vits_generator import VitsGenerator
torch.set_grad_enabled(False)
dev = "cuda:0"
cpu = VitsGenerator(40, 100)
sd = synth.full_state_dict(cpu.state_dict(), 4242)
cpu.load_state_dict(sd, strict=True); cpu.eval()
gpu = VitsGenerator(40, 100); gpu.load_state_dict(sd, strict=True); gpu = gpu.to(dev).eval()
rng = random.Random(7)
worst = 0.0
for it in range(40):
    B = rng.choice([1, 1, 2, 3, 5])
    T = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 257])
    lens = [T] + [rng.randint(1, T) for _ in range(B - 1)]
    g = torch.Generator().manual_seed(it)
    ids = torch.randint(0, 40, (B, T), generator=g); lt = torch.tensor(lens)
    spk = torch.randint(0, 100, (B,), generator=g)
    nw = torch.randn(B, 2, T, generator=g)
    w = cpu._token_stage(ids, lt, spk, nw, 1.3, 0.8)
    h = gpu._token_stage(ids.to(dev), lt.to(dev), spk.to(dev), nw.to(dev), 1.3, 0.8)
    errs = []
    for name, a, b in zip(("m_p", "logs_p", "mask", "w_ceil", "g"), h, w):
        e = (a.cpu() - b).abs().max().item()
        errs.append(e)
        if name == "w_ceil" and e != 0:
            print(f"  it {it} B{B} T{T}: a duration differs by {e}")
        elif name != "w_ceil" and e > 3e-5 * max(1.0, b.abs().max().item()):
            print(f"  it {it} B{B} T{T}: {name} off by {e}")
    worst = max(worst, errs[0], errs[1])
    fa = gpu.frame_prior(ids.to(dev), lt.to(dev), spk.to(dev), length_scale=1.3, noise_w=nw.to(dev))
    fb = cpu.frame_prior(ids, lt, spk, length_scale=1.3, noise_w=nw)
    if fa[0].shape != fb[0].shape:
        print(f"  it {it}: frame count {fa[0].shape} vs {fb[0].shape}")
    elif not torch.equal(fa[2].cpu(), fb[2]):
        print(f"  it {it}: frame mask differs")
print("worst abs error on the prior statistics", worst)
