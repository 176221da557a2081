#!/usr/bin/env python3
"""Runs the reverse flow of one short utterance a few times (for a kernel trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import synth
from vits_amd.module.vits_generator import HotPathGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 237
z = synth.latent(999, "b1", 1, T).to(dev)
mask = torch.ones(1, 1, T, device=dev)
g = gen.embed_speaker(torch.tensor([9], device=dev))
for _ in range(20):
    out = gen.flow(z, mask, speaker_id_embedded=g, reverse=True)
torch.cuda.synchronize()
print(out.shape)
