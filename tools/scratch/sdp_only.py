#!/usr/bin/env python3
"""Runs the duration predictor's reverse pass a few times (for a kernel trace)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import synth
from vits_amd.module.vits_generator import VitsGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = VitsGenerator(40, 100)
gen.load_state_dict(synth.full_state_dict(gen.state_dict(), 999), strict=True)
gen = gen.to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 43
enc = torch.randn(1, 192, T, device=dev); mask = torch.ones(1, 1, T, device=dev)
g = torch.randn(1, 256, 1, device=dev); nw = torch.randn(1, 2, T, device=dev)
for _ in range(20):
    out = gen.stochastic_duration_predictor(enc, mask, speaker_id_embedded=g, reverse=True, noise_scale=0.8, noise=nw)
torch.cuda.synchronize()
print(out.shape)
