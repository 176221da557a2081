#!/usr/bin/env python3
"""Diagnostic (needs a GPU): per-kernel-family device time of the headline step (B=16, T=862),
every launch bracketed by events, averaged over several steps.
  family_table.py [fp32|fp32-split|bf16] [steps] [batch] [frames]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from vits_amd import engine, native, synth
from vits_amd.module.vits_generator import HotPathGenerator

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
T = int(sys.argv[4]) if len(sys.argv) > 4 else 862
dev = torch.device("cuda:0")
engine.set_precision(prec)
torch.set_grad_enabled(False)
sd = synth.hot_path_state_dict(999, with_posterior=False)
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(sd, strict=True)
gen = gen.to(dev).eval()
z = synth.latent(999, "bench.rank0", B, T).to(dev)
mask = synth.length_mask(torch.full((B,), T, dtype=torch.int64), T).to(dev)
ids = synth.speaker_ids(999, "bench.rank0", B).to(dev)


def step():
    return gen.flow_reverse_decode(z, mask, gen.embed_speaker(ids))


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / steps * 1e3
timer = native.LaunchTimer()
with timer:
    for _ in range(steps):
        step()
torch.cuda.synchronize()
fam = timer.summary()
tot = sum(d["ms"] for d in fam.values()) / steps
print(f"{prec} B={B} T={T}: {plain:.3f} ms/step un-instrumented, {tot:.3f} ms/step of bracketed device time")
for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
    n = d["launches"] // steps
    print(f"  {k:44s} n={n:4d} ms/step={d['ms'] / steps:8.3f} us/launch={1e3 * d['ms'] / d['launches']:8.1f} "
          f"TF={d['flops'] / max(d['ms'], 1e-9) / 1e9:7.1f} GB/s={d['bytes'] / max(d['ms'], 1e-9) / 1e6:7.0f}")
