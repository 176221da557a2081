"""Multi-GPU path on CPU: bucketing / assignment invariants, the launcher that starts one rank
per GPU, and a world_size-2 gloo run that drives the same shard -> run -> metrics -> gather ->
summarize code as bench.py, with the CPU oracle standing in for the HIP hot path."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vits_amd import launcher, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_buckets_cover_corpus_and_respect_limits():
    frames, spk = sharding.synthetic_corpus(4096)
    assert frames.min() >= sharding.MIN_FRAMES and frames.max() <= sharding.MAX_FRAMES
    assert 380 <= np.median(frames) <= 480 and spk.min() >= 0 and spk.max() < 100
    buckets = sharding.make_buckets(frames)
    flat = sorted(i for b in buckets for i in b)
    assert flat == list(range(4096))
    for b in buckets:
        assert 1 <= len(b) <= sharding.MAX_ITEMS
        assert len(b) * max(frames[i] for i in b) <= sharding.FRAME_BUDGET
    padded = sum(sharding.bucket_cost(frames, b) for b in buckets)
    assert padded / frames.sum() < 1.03          # length-sorted buckets waste < 3 % on padding
    # the frame budget binds, not the item count: all but the last bucket are filled to within one
    # utterance of it, so every launch sequence is as large as the headline batch
    full = [sharding.bucket_cost(frames, b) for b in buckets[:-1]]
    assert min(full) > sharding.FRAME_BUDGET - sharding.MAX_FRAMES
    assert max(len(b) for b in buckets) > 16
    old_style = sharding.make_buckets(frames, max_items=16)
    assert all(len(b) <= 16 for b in old_style) and len(old_style) > len(buckets)


def test_assignment_balances_ranks():
    frames, _ = sharding.synthetic_corpus(4096)
    buckets = sharding.make_buckets(frames)
    for world in (1, 2, 4, 8):
        parts, loads = sharding.assign_buckets(frames, buckets, world)
        assert sorted(b for p in parts for b in p) == list(range(len(buckets)))
        assert max(loads) / (sum(loads) / world) < 1.04
        parts, loads = sharding.assign_buckets(frames, sharding.split_for_ranks(frames, buckets, world), world)
        assert max(loads) / (sum(loads) / world) < 1.005


def test_rank_shards_partition_the_corpus():
    frames, spk = sharding.synthetic_corpus(300)
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            for bi, idx, lens, ids in sharding.rank_shard(frames, spk, world, r):
                assert lens.tolist() == [int(frames[i]) for i in idx]
                assert ids.tolist() == [int(spk[i]) for i in idx]
                seen += idx
        assert sorted(seen) == list(range(300))


def test_summarize_takes_the_slowest_rank():
    rows = [[2.0, 1000, 1100, 4, 1e-6, 1.5, 0.2], [4.0, 3000, 3300, 12, 3e-6, 2.5, 0.3]]
    s = sharding.summarize(rows, steps=2)
    audio = 4000 * 256 / 22050 * 2
    assert abs(s["value"] - audio / 4.0) < 1e-9 and s["ranks_seen"] == 2
    assert s["per_rank_ms_per_step"] == [1000.0, 2000.0]
    assert abs(s["load_imbalance"] - 4.0 / 3.0) < 1e-12 and s["utterances_per_step"] == 16
    assert s["parity_max_abs"] == 3e-6 and s["checksum"] == 4.0 and abs(s["padding_overhead"] - 1.1) < 1e-12


# ---------------------------------------------------------------------------- launcher

def test_needs_spawn_only_for_a_parent_asked_for_several_gpus():
    assert not launcher.needs_spawn(1, {})
    assert launcher.needs_spawn(2, {})
    assert not launcher.needs_spawn(2, {"RANK": "0", "WORLD_SIZE": "2"})     # already a rank


def test_rank_env():
    env = launcher.rank_env(3, 8, 29511, base={"PATH": "/bin"})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["LOCAL_WORLD_SIZE"]) == ("3", "3", "8", "8")
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511" and env["PATH"] == "/bin"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_run_ranks_starts_every_rank_and_relays_rank0(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        rec = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(sys.argv[1], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        print("line from rank", rec["RANK"])
    """))
    import io
    out = io.StringIO()
    rc = launcher.run_ranks([str(script), str(tmp_path), "--gpus", "2"], 2, stdout=out, timeout=60)
    assert rc == 0
    assert out.getvalue() == "line from rank 0\n"                  # only rank 0's standard output is relayed
    recs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(2)]
    assert [r["RANK"] for r in recs] == ["0", "1"] and all(r["WORLD_SIZE"] == "2" for r in recs)
    assert recs[0]["MASTER_PORT"] == recs[1]["MASTER_PORT"] and recs[0]["MASTER_ADDR"] == "127.0.0.1"
    assert recs[1]["argv"] == [str(tmp_path), "--gpus", "2"]


def test_run_ranks_fails_when_a_rank_fails(tmp_path):
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1': sys.exit(7)\n"
                      "time.sleep(30)\n")
    import io
    rc = launcher.run_ranks([str(script)], 2, stdout=io.StringIO(), timeout=60)
    assert rc == 7                                                 # and rank 0 was terminated, not waited for


def test_bench_parent_spawns_ranks_before_touching_the_gpu(tmp_path):
    """`bench.py --gpus 2` from a plain shell: the parent starts two children with the rank
    environment.  There is no GPU here, so each child stops at its own GPU check - which shows
    the children, not the parent, are the ones that reach it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    if torch.cuda.is_available():
        return                                                      # on a GPU box the run itself is the test
    assert r.returncode != 0
    # (the launcher stops the other rank as soon as one has failed, so one or both get to say it)
    import re
    said = re.findall(r"bench\.py \(rank (\d) of 2\) needs an MI355X", r.stderr)
    assert 1 <= len(said) <= 2 and set(said) <= {"0", "1"}


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE is 1" in r.stderr


# ---------------------------------------------------------------------------- gloo, world size 2

N_UTT = 12


def _tiny_corpus():
    """A corpus small enough for the CPU oracle: 12 utterances of 6..20 frames."""
    frames, spk = sharding.synthetic_corpus(N_UTT)
    frames = 6 + (frames - sharding.MIN_FRAMES) * 14 // (sharding.MAX_FRAMES - sharding.MIN_FRAMES)
    return frames, spk


def _oracle_hot_path(sd):
    from oracle import vits_oracle as O

    def hot_path(z, mask, speakers):
        return O.flow_reverse_decode(z, mask, O.speaker_embed(speakers, sd), sd)
    return hot_path


def _run(world, rank):
    """What a bench rank does with its shard, on the CPU."""
    import time
    from vits_amd import synth
    torch.set_grad_enabled(False)
    torch.set_num_threads(2)
    sd = synth.hot_path_state_dict(999, with_posterior=False)
    frames, spk = _tiny_corpus()
    work = sharding.shard_inputs(sharding.rank_shard(frames, spk, world, rank, max_items=3, frame_budget=60), 999)
    t0 = time.perf_counter()
    kept = sharding.run_shard(work, _oracle_hot_path(sd), keep="all")
    el = time.perf_counter() - t0
    row = sharding.metrics_row(el, work, parity=0.0, checksum=sharding.valid_checksum(work, kept))
    per_utt = {}
    for b in work:
        for j, u in enumerate(b.utterances):
            per_utt[int(u)] = float(kept[b.bucket_id][j, 0, :int(b.lengths[j]) * 256].double().sum())
    return row, per_utt


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    row, per_utt = _run(world, rank)
    dist.barrier()
    table = sharding.gather_metrics(row)
    json.dump(per_utt, open(os.path.join(out_dir, f"utt{rank}.json"), "w"))
    if rank == 0:
        np.save(os.path.join(out_dir, "table.npy"), table.numpy())
    dist.destroy_process_group()


def test_two_rank_gloo_shard_run_gather(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, launcher.free_port(), str(tmp_path)), nprocs=world, join=True)
    table = np.load(tmp_path / "table.npy")
    frames, _ = _tiny_corpus()
    assert table.shape == (2, sharding.METRIC_COLUMNS)
    s = sharding.summarize(table, steps=1)
    assert s["ranks_seen"] == 2 and s["utterances_per_step"] == N_UTT
    assert table[:, sharding.M_VALID_FRAMES].sum() == frames.sum()
    assert abs(s["value"] - frames.sum() * 256 / 22050 / table[:, sharding.M_ELAPSED].max()) < 1e-6
    # every utterance was synthesised on exactly one rank, and sharding does not change what it
    # sounds like: the same utterance in a single-process run gives the same samples
    sharded = {}
    for r in range(world):
        part = json.load(open(tmp_path / f"utt{r}.json"))
        assert not set(part) & set(sharded)
        sharded.update(part)
    assert sorted(int(k) for k in sharded) == list(range(N_UTT))
    row1, single = _run(1, 0)
    for u, v in single.items():
        assert abs(sharded[str(u)] - v) <= 1e-4 * max(1.0, abs(v)), (u, sharded[str(u)], v)
    assert abs(s["checksum"] - row1[sharding.M_CHECKSUM]) <= 1e-3 * max(1.0, abs(row1[sharding.M_CHECKSUM]))


@pytest.mark.gpu
def test_host_keeps_ahead_of_the_device_on_the_lightest_bucket():
    """A rank keeps its GPU busy only while enqueueing a bucket takes the Python thread less time
    than the device needs to run it.  The lightest bucket the old 16-item cap produced (16
    utterances of 2 s) is the hardest case; in the fastest precision mode the device needs the
    least time.  Also: a bucket filled to the frame budget with 2 s utterances costs the device no
    more than the headline batch of the same number of frames (within 5 %)."""
    from vits_amd import engine, synth
    from vits_amd.module.vits_generator import HotPathGenerator
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(synth.hot_path_state_dict(999, with_posterior=False), strict=True)
    gen = gen.to(dev).eval()

    def hot_path(z, mask, speakers):
        return gen.flow_reverse_decode(z, mask, gen.speaker_ref(speakers))

    def bucket(n, t, bid):
        lens = torch.full((n,), t, dtype=torch.int64)
        m = synth.length_mask(lens, t)
        return sharding.Bucket(bid, list(range(n)), lens, synth.latent(999, f"host.b{bid}", n, t).mul_(m).to(dev),
                               m.to(dev), synth.speaker_ids(999, f"host.s{bid}", n).to(dev))
    for prec in ("fp32", "bf16"):
        gen.set_precision(prec)
        light, = sharding.host_device_split([bucket(16, sharding.MIN_FRAMES, 0)], hot_path, dev, reps=5)
        assert light["host_ms"] < light["device_ms"], (prec, light)
    gen.set_precision("fp32")
    rows = sharding.host_device_split([bucket(80, sharding.MIN_FRAMES, 1), bucket(16, 860, 2)], hot_path, dev, reps=3)
    assert rows[0]["device_ms"] < 1.05 * rows[1]["device_ms"], rows
