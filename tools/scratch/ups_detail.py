#!/usr/bin/env python3
"""Per-launch device time of the upsampling convolutions in one headline step, per precision."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import engine, native, synth
from vits_amd.module.vits_generator import HotPathGenerator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
engine.set_precision(prec)
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
B, T = 16, 862
z = synth.latent(999, "bench.rank0", B, T).to(dev)
mask = torch.ones(B, 1, T, device=dev)
ids = synth.speaker_ids(999, "bench.rank0", B).to(dev)
for _ in range(3): gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))
torch.cuda.synchronize()
with native.LaunchTimer() as lt:
    for _ in range(3): gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))
torch.cuda.synchronize()
seen = {}
for fam, flops, nbytes, e0, e1, detail in lt.records:
    if "upsample" in fam or "post" in fam or ("store" in fam and "act=0" in fam):
        seen.setdefault((fam, detail), []).append((e0.elapsed_time(e1), flops, nbytes))
for (fam, detail), v in seen.items():
    ms = sum(t for t, _, _ in v) / len(v)
    print(f"{fam:34s} {detail:34s} {1e3*ms:8.1f} us  {v[0][1]/ms/1e9:7.1f} TF  {v[0][2]/ms/1e6:7.0f} GB/s")
