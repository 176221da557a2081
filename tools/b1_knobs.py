#!/usr/bin/env python3
"""Diagnostic: single-utterance decoder latency (eager and replayed as a graph) under the small-launch
knobs of native.py.   b1_knobs.py T [precision]   (VITS_AMD_NARROW_BELOW, VITS_AMD_SPLIT_K in the environment)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vits_amd import engine, native, synth
from vits_amd.graphs import GraphedCall
from vits_amd.module.vits_generator import HotPathGenerator

T = int(sys.argv[1]) if len(sys.argv) > 1 else 258
engine.set_precision(sys.argv[2] if len(sys.argv) > 2 else "fp32")
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
z = synth.latent(999, "b1", 1, T).to(dev)
g = gen.speaker_ref(torch.tensor([9], device=dev))


def timed(f, n):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timed(lambda: gen.decoder(z, g), 30)
graphed = GraphedCall(lambda zz: gen.decoder(zz, g), z)
rep = timed(lambda: graphed(z), 50)
print(f"T={T} narrow<{native.NARROW_BELOW_BLOCKS} splitk={native.SPLIT_SMALL_LAUNCHES} "
      f"split_below={os.environ.get('VITS_AMD_SPLIT_BELOW', '512')}: eager {eager:.3f} ms, graph {rep:.3f} ms")
