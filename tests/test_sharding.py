"""Multi-GPU path on CPU: bucketing / assignment invariants, and a world_size-2 gloo run that
shards a corpus, processes the shards independently and gathers the metrics."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vits_amd import sharding


def test_buckets_cover_corpus_and_respect_limits():
    frames, spk = sharding.synthetic_corpus(4096)
    assert frames.min() >= sharding.MIN_FRAMES and frames.max() <= sharding.MAX_FRAMES
    assert 380 <= np.median(frames) <= 480 and spk.min() >= 0 and spk.max() < 100
    buckets = sharding.make_buckets(frames)
    flat = sorted(i for b in buckets for i in b)
    assert flat == list(range(4096))
    for b in buckets:
        assert 1 <= len(b) <= 16
        assert len(b) * max(frames[i] for i in b) <= 16 * 862
    padded = sum(sharding.bucket_cost(frames, b) for b in buckets)
    assert padded / frames.sum() < 1.03          # length-sorted buckets waste < 3 % on padding


def test_assignment_balances_ranks():
    frames, _ = sharding.synthetic_corpus(4096)
    buckets = sharding.make_buckets(frames)
    for world in (1, 2, 4, 8):
        parts, loads = sharding.assign_buckets(frames, buckets, world)
        assert sorted(b for p in parts for b in p) == list(range(len(buckets)))
        assert max(loads) / (sum(loads) / world) < 1.02


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, _ = sharding.synthetic_corpus(512)
    buckets = sharding.make_buckets(frames)
    parts, _ = sharding.assign_buckets(frames, buckets, world)
    mine = [i for b in parts[rank] for i in buckets[b]]
    audio_frames = float(sum(frames[i] for i in mine))       # stand-in for the per-rank hot path
    checksum = float(sum(int(i) * int(frames[i]) for i in mine))
    dist.barrier()
    table = sharding.gather_metrics([audio_frames, checksum, float(len(mine))])
    if rank == 0:
        np.save(os.path.join(out_dir, "table.npy"), table.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    table = np.load(tmp_path / "table.npy")
    frames, _ = sharding.synthetic_corpus(512)
    assert table.shape == (2, 3)
    assert table[:, 2].sum() == 512                              # every utterance on exactly one rank
    assert table[:, 0].sum() == frames.sum()
    assert table[:, 1].sum() == float(sum(i * int(f) for i, f in enumerate(frames)))
    assert abs(table[0, 0] - table[1, 0]) / frames.sum() < 0.02  # balanced
