#!/usr/bin/env python3
"""Headline step (B=16, T=862) in the sliced-fp32 mode with the diagnostic library (threshold experiments)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import engine, native, synth
native.use_diagnostic_library()
from vits_amd.module.vits_generator import HotPathGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
engine.set_precision("fp32-split")
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
B, T = 16, 862
z = synth.latent(999, "bench.rank0", B, T).to(dev)
mask = torch.ones(B, 1, T, device=dev)
ids = synth.speaker_ids(999, "bench.rank0", B).to(dev)
def step():
    return gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("split step %.3f ms" % ((time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(20): gen.flow(z, mask, speaker_id_embedded=gen.embed_speaker(ids), reverse=True)
torch.cuda.synchronize()
print("flow %.3f ms" % ((time.perf_counter() - t0) * 50))
