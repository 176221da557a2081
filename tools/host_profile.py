#!/usr/bin/env python3
"""Host issue time against device time of the hot path (reverse flow + decoder), per batch shape.

  python tools/host_profile.py [--precision fp32] [--profile] B,T [B,T ...]

For each shape: the time the Python thread needs to enqueue one pass (no synchronisation inside the
loop, the queue drained before it starts) and the time the device needs for it (events around a
pass that was enqueued while the device was idle, so nothing waits for the host).  A rank keeps its
GPU busy only while the first stays below the second.  --profile adds a cProfile table of the host
side for the first shape."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def measure(gen, dev, B, T, reps=10, seed=999):
    from vits_amd import synth
    lens = torch.full((B,), T, dtype=torch.int64)
    mask = synth.length_mask(lens, T).to(dev)
    z = synth.latent(seed, f"host.{B}.{T}", B, T).to(dev)
    ids = synth.speaker_ids(seed, "host.spk", B).to(dev)

    def step():
        return gen.flow_reverse_decode(z, mask, gen.speaker_ref(ids))
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    host = []
    for _ in range(reps):                       # host: enqueue only, the device queue is empty at the start
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()                    # device: back-to-back passes, the queue never runs dry
    for _ in range(reps):
        step()
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / reps
    host.sort()
    return {"batch": B, "frames": T, "host_ms": 1e3 * host[len(host) // 2], "wall_ms": 1e3 * wall, "step": step}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=["16,172", "16,430", "16,862", "80,172"])
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    from vits_amd import engine, synth
    from vits_amd.module.vits_generator import HotPathGenerator
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    engine.set_precision(args.precision)
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
    gen = gen.to(dev).eval()
    first = None
    for s in args.shapes:
        B, T = (int(v) for v in s.split(","))
        r = measure(gen, dev, B, T)
        first = first or r
        print(f"B={B:3d} T={T:4d}: host {r['host_ms']:7.2f} ms, back-to-back wall {r['wall_ms']:7.2f} ms per pass "
              f"({args.precision}); host/wall = {r['host_ms'] / r['wall_ms']:.2f}", flush=True)
    if args.profile and first is not None:
        pr = cProfile.Profile()
        torch.cuda.synchronize(dev)
        pr.enable()
        for _ in range(20):
            first["step"]()
        pr.disable()
        torch.cuda.synchronize(dev)
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
