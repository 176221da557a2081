"""Utterance sharding for a multi-GPU node: length bucketing and bucket-to-rank assignment.

Every utterance is independent on the generator hot path (no cross-item op, read-only
speaker table), so a node runs one process per GPU on disjoint buckets of utterances with
no collective in the hot path; RCCL is only used afterwards to gather a few metrics.
Because the decoder is unmasked (its output for a short item depends on the padded tail,
/root/reference/module/model_component/decoder.py:110-131), padding is pure waste: buckets
are built from length-sorted utterances so each is padded only to its own longest item.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synth

MIN_FRAMES, MAX_FRAMES = 172, 1292          # 2 s .. 15 s at 86.13 frames/s (jvs_preprocessor.py:30 drops < 2 s)


def synthetic_corpus(n: int, seed: int = 999, n_speakers: int = 100):
    """Lengths (frames) from a clipped log-normal with a 5 s median, and speaker ids."""
    z = synth.normal(seed, "corpus.len", (n,))
    frames = np.clip(np.round(430.0 * np.exp(0.5 * z)), MIN_FRAMES, MAX_FRAMES).astype(np.int64)
    spk = synth.speaker_ids(seed, "corpus.spk", n, n_speakers).numpy()
    return frames, spk


def make_buckets(frames, max_items: int = 16, frame_budget: int = 16 * 862):
    """Sort by length (longest first) and cut into buckets of at most ``max_items`` items whose
    padded size items * longest stays within ``frame_budget``.  Returns lists of utterance ids."""
    order = np.argsort(-np.asarray(frames), kind="stable")
    buckets, cur = [], []
    for i in order:
        longest = frames[cur[0]] if cur else frames[i]
        if cur and (len(cur) + 1 > max_items or (len(cur) + 1) * longest > frame_budget):
            buckets.append(cur)
            cur = []
        cur.append(int(i))
    if cur:
        buckets.append(cur)
    return buckets


def bucket_cost(frames, bucket) -> int:
    """Work of a bucket: every item is computed at the padded length."""
    return len(bucket) * int(max(frames[i] for i in bucket))


def assign_buckets(frames, buckets, world: int):
    """Greedy longest-processing-time assignment: heaviest bucket to the least loaded rank."""
    loads = [0] * world
    parts = [[] for _ in range(world)]
    for b in sorted(range(len(buckets)), key=lambda j: -bucket_cost(frames, buckets[j])):
        r = min(range(world), key=lambda k: loads[k])
        parts[r].append(b)
        loads[r] += bucket_cost(frames, buckets[b])
    return parts, loads


def gather_metrics(values, device=None):
    """All-gather a short list of floats from every rank (RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return t.unsqueeze(0).cpu()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu()
