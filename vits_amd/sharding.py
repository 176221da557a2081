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


MAX_ITEMS, FRAME_BUDGET = 96, 16 * 862


def make_buckets(frames, max_items: int = MAX_ITEMS, frame_budget: int = FRAME_BUDGET):
    """Sort by length (longest first) and cut into buckets of at most ``max_items`` items whose
    padded size items * longest stays within ``frame_budget``.  Returns lists of utterance ids.
    The budget, not the item count, is what binds (16 utterances of 10 s; 80 of 2 s): a bucket is
    one launch sequence, and the GPU runs a bucket of 2,752 frames (16 x 2 s) at 81 % of the frames
    per second it reaches on 13,792 (measured: 15.96 ms against 67.1 ms), so every bucket is filled
    to the same number of padded frames whatever the length of its utterances."""
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


def split_for_ranks(frames, buckets, world: int):
    """Full buckets are all about as heavy as each other, so a greedy assignment over ``world``
    ranks can end up to one bucket apart (2.9 % at 8 ranks and 148 buckets).  Halving the ``world``
    lightest buckets gives it small pieces to even the ranks out with (0.3 %), at the price of
    ``world`` launch sequences of half the size."""
    if world <= 1:
        return [list(b) for b in buckets]
    light = set(sorted(range(len(buckets)), key=lambda j: bucket_cost(frames, buckets[j]))[:world])
    out = []
    for j, b in enumerate(buckets):
        if j in light and len(b) >= 2:
            out += [list(b[:len(b) // 2]), list(b[len(b) // 2:])]
        else:
            out.append(list(b))
    return out


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


NODE_RANKS = 8          # GPUs of a node: the buckets are cut for this many ranks whatever the world size


def rank_shard(frames, spk, world: int, rank: int, max_items: int = MAX_ITEMS, frame_budget: int = FRAME_BUDGET,
               balance_ranks: int = NODE_RANKS):
    """The buckets rank ``rank`` of ``world`` processes: a list of
    (bucket id, utterance ids, lengths [n] int64, speaker ids [n] int64), heaviest first.
    The buckets themselves do not depend on ``world`` (an utterance's waveform depends on the
    padded length of its bucket, the decoder being unmasked, so a corpus must be cut the same way
    on 1, 2, 4 or 8 ranks to give the same samples): they are always cut for ``balance_ranks``."""
    buckets = split_for_ranks(frames, make_buckets(frames, max_items, frame_budget), max(balance_ranks, world))
    parts, _ = assign_buckets(frames, buckets, world)
    shard = []
    for bi in parts[rank]:
        idx = buckets[bi]
        shard.append((bi, idx, torch.tensor([int(frames[i]) for i in idx], dtype=torch.int64),
                      torch.tensor([int(spk[i]) for i in idx], dtype=torch.int64)))
    return shard


# columns of the per-rank metrics row every rank contributes after the timed region
M_ELAPSED, M_VALID_FRAMES, M_PADDED_FRAMES, M_UTTERANCES, M_PARITY, M_CHECKSUM, M_HOST = range(7)
METRIC_COLUMNS = 7      # M_HOST: seconds the rank's Python thread needed to enqueue the timed work (0: not measured)


def summarize(table, steps: int, hop: int = 256, sample_rate: int = 22050):
    """Whole-job figures from the gathered [world, METRIC_COLUMNS] table: valid audio seconds of
    all ranks over the slowest rank's time (the job is done when the last rank is)."""
    t = torch.as_tensor(table, dtype=torch.float64)
    elapsed = t[:, M_ELAPSED]
    slowest = float(elapsed.max())
    audio_s = float(t[:, M_VALID_FRAMES].sum()) * hop / sample_rate * steps
    return {
        "value": audio_s / slowest,
        "ms_per_step": 1e3 * slowest / steps,
        "ranks_seen": int(t.shape[0]),
        "per_rank_ms_per_step": [1e3 * float(e) / steps for e in elapsed],
        "per_rank_host_issue_ms_per_step": [1e3 * float(h) / steps for h in t[:, M_HOST]],
        "per_rank_value": [float(f) * hop / sample_rate * steps / float(e)
                           for f, e in zip(t[:, M_VALID_FRAMES], elapsed)],
        "load_imbalance": slowest / float(elapsed.mean()),
        "padding_overhead": float(t[:, M_PADDED_FRAMES].sum()) / max(float(t[:, M_VALID_FRAMES].sum()), 1.0),
        "utterances_per_step": int(t[:, M_UTTERANCES].sum()),
        "parity_max_abs": float(t[:, M_PARITY].max()),
        "checksum": float(t[:, M_CHECKSUM].sum()),
    }


class Bucket:
    """One padded batch of a rank's shard: utterance ids, lengths, latents, mask, speakers."""

    __slots__ = ("bucket_id", "utterances", "lengths", "z", "mask", "speakers")

    def __init__(self, bucket_id, utterances, lengths, z, mask, speakers):
        self.bucket_id, self.utterances, self.lengths = bucket_id, utterances, lengths
        self.z, self.mask, self.speakers = z, mask, speakers

    @property
    def valid_frames(self) -> int:
        return int(self.lengths.sum())

    @property
    def padded_frames(self) -> int:
        return int(self.lengths.numel() * self.lengths.max())


def shard_inputs(shard, seed: int = 999, device=None):
    """Synthetic masked latents for every bucket of a shard (the data a bucket holds depends
    only on the seed and the bucket id, never on which rank processes it)."""
    work = []
    for bi, idx, lens, spk in shard:
        tmax = int(lens.max())
        m = synth.length_mask(lens, tmax)
        z = synth.latent(seed, f"corpus.b{bi}", len(idx), tmax).mul_(m)
        work.append(Bucket(bi, idx, lens, z.to(device), m.to(device), spk.to(device)))
    return work


def run_shard(work, hot_path, keep=None):
    """One pass over a rank's buckets: ``hot_path(z, mask, speakers)`` -> padded waveforms
    [n, 1, 256 * longest].  ``keep`` is None (drop outputs), "all", or a set of bucket ids whose
    output is returned as {bucket id: waveform}."""
    kept = {}
    for b in work:
        wav = hot_path(b.z, b.mask, b.speakers)
        if keep == "all" or (keep is not None and b.bucket_id in keep):
            kept[b.bucket_id] = wav
    return kept


def host_device_split(work, hot_path, device, reps: int = 3):
    """Per bucket of ``work``: the time this thread needs to enqueue one pass of ``hot_path`` (device
    idle at the start, nothing synchronised inside) and the time the device needs for it (back to
    back passes, the queue never dry).  A rank keeps its GPU busy while the first stays below the
    second.  Returns a list of dicts (bucket id, batch, frames, host_ms, device_ms)."""
    import time
    rows = []
    for b in work:
        hot_path(b.z, b.mask, b.speakers)
        host = []
        for _ in range(reps):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            hot_path(b.z, b.mask, b.speakers)
            host.append(time.perf_counter() - t0)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            hot_path(b.z, b.mask, b.speakers)
        torch.cuda.synchronize(device)
        dev_s = (time.perf_counter() - t0) / reps
        rows.append({"bucket": b.bucket_id, "batch": int(b.lengths.numel()), "frames": int(b.lengths.max()),
                     "host_ms": 1e3 * sorted(host)[len(host) // 2], "device_ms": 1e3 * dev_s})
    return rows


def valid_checksum(work, outputs, hop: int = 256) -> float:
    """Sum of the valid samples of every kept output (float64), a per-rank fingerprint that
    does not depend on how buckets were dealt to ranks."""
    total = 0.0
    for b in work:
        wav = outputs.get(b.bucket_id)
        if wav is None:
            continue
        t = torch.arange(wav.shape[-1], device=wav.device).unsqueeze(0)
        valid = t < (b.lengths.to(wav.device) * hop).unsqueeze(1)
        total += float((wav[:, 0].double() * valid).sum())
    return total


def metrics_row(elapsed, work, parity=0.0, checksum=0.0, host=0.0):
    row = [0.0] * METRIC_COLUMNS
    row[M_ELAPSED] = float(elapsed)
    row[M_HOST] = float(host)
    row[M_VALID_FRAMES] = float(sum(b.valid_frames for b in work))
    row[M_PADDED_FRAMES] = float(sum(b.padded_frames for b in work))
    row[M_UTTERANCES] = float(sum(len(b.utterances) for b in work))
    row[M_PARITY] = float(parity)
    row[M_CHECKSUM] = float(checksum)
    return row
