#!/usr/bin/env python3
"""Runs text_to_speech of the one benchmark sentence a few times (for a kernel trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import synth
from vits_amd.module.vits_generator import VitsGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = VitsGenerator(40, 100)
gen.load_state_dict(synth.full_state_dict(gen.state_dict(), 999), strict=True)
gen = gen.to(dev).eval()
gen.use_front_graph = False              # eager, so that the trace names every kernel
ids = torch.tensor([synth.TTS_SENTENCE_IDS], device=dev); lens = torch.tensor([ids.shape[1]], device=dev)
spk = torch.tensor([9], device=dev)
torch.manual_seed(1)
nw = torch.randn(1, 2, ids.shape[1]).to(dev)
for _ in range(20):
    wav = gen.text_to_speech(ids, lens, spk, length_scale=3, noise_w=nw)
torch.cuda.synchronize()
print(wav.shape)
