#!/usr/bin/env python3
"""Host-side cost of an eager B=1 decoder forward: cProfile over 100 calls."""
import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vits_amd import synth
from vits_amd.module.vits_generator import HotPathGenerator
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
gen = HotPathGenerator(100, with_posterior=False)
gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
gen = gen.to(dev).eval()
z = synth.latent(999, "b1", 1, 237).to(dev)
g = gen.embed_speaker(torch.tensor([9], device=dev))
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "stream":                      # another stream used before the decoder makes its own
    s0 = torch.cuda.Stream()
    with torch.cuda.stream(s0):
        torch.zeros(10, device=dev).add_(1)
    torch.cuda.synchronize()
if mode == "warmgraph":
    gen.decoder(z, g); torch.cuda.synchronize()
if mode == "ws":                          # only the capture workspace, no graph
    from vits_amd import native
    native.ensure_capture_workspace(dev)
if mode == "rawgraph":                    # a torch graph without the workspace
    gr = torch.cuda.CUDAGraph()
    buf = torch.zeros(100, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        buf.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        out = buf * 2
    gr.replay(); torch.cuda.synchronize()
if mode in ("graph", "warmgraph"):                       # a graph captured and replayed before
    from vits_amd.graphs import GraphedCall
    gc = GraphedCall(lambda a: a * 2 + 1, torch.zeros(100, device=dev))
    gc(torch.ones(100, device=dev)); torch.cuda.synchronize()
if mode == "decgraph":
    from vits_amd.graphs import GraphedCall
    gc = GraphedCall(gen.decoder, z, g)
    gc(z, g); torch.cuda.synchronize()
for _ in range(5):
    gen.decoder(z, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    gen.decoder(z, g)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per forward %.3f ms, to completion %.3f ms" % ((t1 - t0) * 10, (t2 - t0) * 10))
if mode:
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    gen.decoder(z, g)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
