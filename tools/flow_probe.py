#!/usr/bin/env python3
"""Diagnostic: time the reverse flow alone (needs a GPU).   flow_probe.py B T [fp32|bf16|fp32-split] [diag]
With `diag` the diagnostic library is loaded, so its environment switches apply (VK_WN_BREP=1/2/4
forces the one-launch WN layer's window to 32 / 64 / 128 columns)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vits_amd import engine, native, synth
from vits_amd.module.vits_generator import HotPathGenerator

B, T = int(sys.argv[1]), int(sys.argv[2])
engine.set_precision(sys.argv[3] if len(sys.argv) > 3 else "fp32")
if len(sys.argv) > 4:
    native.use_diagnostic_library()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
z = synth.latent(999, "flow", B, T).to(dev)
mask = synth.length_mask(torch.full((B,), T, dtype=torch.int64), T).to(dev)
g = gen.speaker_ref(synth.speaker_ids(999, "flow", B).to(dev))
for _ in range(5):
    gen.flow(z, mask, g, reverse=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    gen.flow(z, mask, g, reverse=True)
e1.record()
torch.cuda.synchronize()
print(f"reverse flow B={B} T={T} {sys.argv[3] if len(sys.argv) > 3 else 'fp32'}: {e0.elapsed_time(e1) / 20:.3f} ms")
