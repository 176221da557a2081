#!/usr/bin/env python3
"""Benchmark of the VITS generator hot path on MI355X: reverse flow + HiFi-GAN decoder.

  python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path (flow reverse -> decoder) over one batch of
synthetic latents already resident in HBM: 16 utterances x 862 frames (10.008 s each at
22050 Hz), 16 seeded speakers out of 100, random-init weights from vits_amd.synth.

For N > 1 this process starts N ranks itself (one per GPU, fresh child processes, before
anything here touches the GPU) unless it already is a rank of torch.distributed.run.  Every
rank runs its own per-GPU batch (weak scaling, utterance-sharded, no collective in the hot
path); RCCL is used for the barriers and for one all-gather of a seven-number metrics row per
rank after each timed region.

Rank 0 prints ONE JSON line: audio seconds generated per wall second (whole job), a roofline
object for the kernel family that took the most device time, the largest difference between
the timed output and the CPU oracle ("parity"; the run fails above 1e-4), the sharded
4096-utterance corpus of the same hot path ("corpus", strong scaling) and - at N = 1 - the
other precision modes, the other benchmark configurations and the CPU oracle timed on this
host's cores.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE = 22050
HOP = 256
BF16_MATRIX_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0
FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
SEED = 999
PARITY_TOL = 1e-4                    # fp32 contract: waveform max-abs against the reference arithmetic
BF16_TOL = 1e-2                      # separately stated tolerance of the bf16 mode

DTYPE_NAMES = {"fp32": "f32", "bf16": "bf16 matrix inputs, f32 accumulate",
               "fp32-split": "f32 operands as 3 exact bf16 slices, 6 slice products, f32 accumulate"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=862, help="latent frames per utterance (862 = 10 s)")
    ap.add_argument("--workload", choices=["batch", "corpus"], default="batch",
                    help="batch: fixed batch per GPU (weak scaling, the headline; the corpus is reported beside "
                         "it); corpus: the length-bucketed corpus sharded over the ranks is the timed step "
                         "(strong scaling)")
    ap.add_argument("--corpus", type=int, default=4096, help="utterances in the corpus workload (0: skip it)")
    ap.add_argument("--corpus-steps", type=int, default=1, help="timed passes over the corpus beside the headline")
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp32-split"], default="fp32",
                    help="matrix-core input precision of the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip the short runs of the other precision modes after the timed region")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the text_to_speech / single-utterance decoder / voice-conversion lines")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL over xGMI, the real thing) or gloo (rehearsals on a box with fewer GPUs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks map onto the visible GPUs modulo their number")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (all threads)")
    return ap.parse_args(argv)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def csrc_hash() -> str:
    """Fingerprint of the kernel sources; the stored HBM-traffic figures carry the one they were
    measured with."""
    from vits_amd.build import csrc_hash as h
    return h()


def cpu_baseline(sd, frames: int, budget_s: float):
    """The CPU oracle (stock PyTorch CPU ops, the kernels the reference dispatches to) on one
    utterance of the same length: repeated for about ``budget_s`` seconds on all threads, then
    once on a single thread."""
    import torch
    from oracle import vits_oracle as O
    from vits_amd import synth
    z = synth.latent(SEED, "bench.cpu.z", 1, frames)
    mask = synth.length_mask([frames], frames)
    g = O.speaker_embed(torch.tensor([9]), sd)
    utt_s = frames * HOP / SAMPLE_RATE
    with torch.no_grad():
        O.flow_reverse_decode(z, mask, g, sd)                     # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            O.flow_reverse_decode(z, mask, g, sd)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 100:
                break
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            t1 = time.perf_counter()
            O.flow_reverse_decode(z, mask, g, sd)
            el1 = time.perf_counter() - t1
        finally:
            torch.set_num_threads(threads)
    return {"value": n * utt_s / el, "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": f"{n} x (1 utterance x {frames} frames = {utt_s:.2f} s): "
                      f"flow-reverse + decoder, fp32, {el:.1f} s of CPU time",
            "one_thread_value": utt_s / el1, "one_thread_sample": f"1 x the same utterance, {el1:.1f} s",
            "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count()}


def oracle_items(sd, z_p, mask, ids, items):
    """Waveforms of the chosen batch items from the CPU oracle.  Items are independent on this
    path (no cross-item op), so each goes through alone at the batch's padded length."""
    import torch
    from oracle import vits_oracle as O
    out = {}
    with torch.no_grad():
        for i in items:
            g = O.speaker_embed(ids[i:i + 1].cpu(), sd)
            out[i] = O.flow_reverse_decode(z_p[i:i + 1].cpu(), mask[i:i + 1].cpu(), g, sd)
    return out


def main():
    args = parse()
    from vits_amd import launcher
    if launcher.needs_spawn(args.gpus):
        # the parent never touches the GPU: it starts the ranks, relays rank 0's line and leaves
        sys.exit(launcher.run_ranks([os.path.abspath(__file__), *sys.argv[1:]], args.gpus))

    import torch
    from vits_amd import engine, native, sharding, synth
    from vits_amd.module.vits_generator import HotPathGenerator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py (rank {rank} of {world}) needs an MI355X (no CPU fallback for the HIP path)")
    if args.share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    engine.set_precision(args.precision)
    dist = None
    if world > 1 or "RANK" in os.environ:           # a rank of a multi-process job, any world size
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    mdev = dev if (dist is not None and args.dist_backend == "nccl") else None     # where metric rows are gathered

    torch.set_grad_enabled(False)
    sd = synth.hot_path_state_dict(SEED, with_posterior=False)
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(dev).eval()
    tol = BF16_TOL if args.precision == "bf16" else PARITY_TOL

    def hot_path(z, mask, speakers):
        return gen.flow_reverse_decode(z, mask, gen.speaker_ref(speakers))     # speaker conditioning: table rows

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ corpus (4096 length-bucketed utterances, sharded)
    def corpus_pass(steps: int, warmup: int):
        """The 4096-utterance length-bucketed corpus, sharded over the ranks by bucket (strong
        scaling): ``steps`` timed passes over this rank's shard, three of its utterances checked
        against the CPU oracle, metrics row gathered from every rank."""
        frames, spk = sharding.synthetic_corpus(args.corpus, SEED)
        shard = sharding.rank_shard(frames, spk, world, rank)
        work = sharding.shard_inputs(shard, SEED, dev)
        picks = [work[0], work[len(work) // 2], work[-1]] if work else []
        for _ in range(warmup):
            sharding.run_shard(work, hot_path)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            kept = sharding.run_shard(work, hot_path, keep="all")
        issued = time.perf_counter() - t0                  # the Python thread is done enqueueing
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        worst, checked = 0.0, []
        for b in picks:
            i = len(b.utterances) - 1                      # the shortest item of the bucket: most padding
            ref = oracle_items(sd, b.z, b.mask, b.speakers, [i])[i]
            n = int(b.lengths[i]) * HOP                    # compare the whole padded item, report the valid part too
            diff = (kept[b.bucket_id][i:i + 1].cpu() - ref).abs()
            worst = max(worst, float(diff.max()))
            checked.append({"bucket": b.bucket_id, "utterance": int(b.utterances[i]), "frames": int(b.lengths[i]),
                            "max_abs": float(diff.max()), "max_abs_valid": float(diff[..., :n].max())})
        checksum = sharding.valid_checksum(work, kept)
        del kept
        # Host against device on this rank's lightest and heaviest bucket.  (`issued` is not the
        # host's own cost here: with more than a few buckets in flight the enqueueing thread waits for
        # room in the device queue.)  The rank's host time per pass is estimated from the two.
        ends = [min(work, key=lambda b: b.padded_frames), max(work, key=lambda b: b.padded_frames)] if work else []
        split = sharding.host_device_split(ends, hot_path, dev)
        host_est = steps * len(work) * sum(r["host_ms"] for r in split) / max(len(split), 1) / 1e3
        row = sharding.metrics_row(el, work, parity=worst, checksum=checksum, host=host_est)
        table = sharding.gather_metrics(row, device=mdev)
        out = sharding.summarize(table, steps)
        out.update({"utterances": args.corpus, "steps": steps, "scaling": "strong", "checked_rank0": checked,
                    "tol": tol, "buckets_rank0": len(work),
                    "bucket_policy": f"length-sorted, <= {sharding.MAX_ITEMS} items and <= {sharding.FRAME_BUDGET} "
                                     f"padded frames per bucket, the {sharding.NODE_RANKS} lightest halved",
                    "enqueue_returned_after_ms_rank0": 1e3 * issued / steps,
                    "host_note": "per_rank_host_issue_ms_per_step: buckets x mean enqueue time of the rank's lightest "
                                 "and heaviest bucket (device idle, no synchronisation inside); the device needs "
                                 "per_rank_ms_per_step",
                    "host_vs_device_rank0": split})
        return out, worst

    # ------------------------------------------------------------------ headline step
    B, T = args.batch, args.frames
    key = f"bench.rank{rank}"
    z_p = synth.latent(SEED, key, B, T).to(dev)
    lengths = torch.full((B,), T, dtype=torch.int64)
    mask = synth.length_mask(lengths, T).to(dev)
    ids = synth.speaker_ids(SEED, key, B).to(dev)

    if args.workload == "batch":
        def step():
            return hot_path(z_p, mask, ids)
        valid_frames_local, padded_local, utts_local = float(B * T), float(B * T), float(B)
    else:
        frames_c, spk_c = sharding.synthetic_corpus(args.corpus, SEED)
        work_c = sharding.shard_inputs(sharding.rank_shard(frames_c, spk_c, world, rank), SEED, dev)

        def step():
            return sharding.run_shard(work_c, hot_path, keep={work_c[0].bucket_id})[work_c[0].bucket_id]
        valid_frames_local = float(sum(b.valid_frames for b in work_c))
        padded_local = float(sum(b.padded_frames for b in work_c))
        utts_local = float(sum(len(b.utterances) for b in work_c))

    for _ in range(args.warmup):
        wav = step()
    # One untimed step with every launch bracketed by events: finds the kernel family with the
    # most device time and gives the per-family table.  In the timed region only that family's
    # launches carry events (bracketing all ~420 launches of a step costs about 0.7 ms per step).
    survey = native.LaunchTimer()
    with survey:
        wav = step()
    torch.cuda.synchronize(dev)
    fam_all = survey.summary()
    dominant = max(fam_all.items(), key=lambda kv: kv[1]["ms"])[0]
    fence()
    timer = native.LaunchTimer(only={dominant})
    t0 = time.perf_counter()
    with timer:
        for _ in range(args.steps):
            wav = step()
    issued = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    assert bool(torch.isfinite(wav).all())

    # ---- the timed output against the CPU oracle: first and last item of this rank's batch
    if args.workload == "batch":
        items = sorted({0, B - 1})
        ref = oracle_items(sd, z_p, mask, ids, items)
        parity_items = {i: float((wav[i:i + 1].cpu() - ref[i]).abs().max()) for i in items}
    else:
        b0 = work_c[0]
        items = sorted({0, len(b0.utterances) - 1})
        ref = oracle_items(sd, b0.z, b0.mask, b0.speakers, items)
        parity_items = {i: float((wav[i:i + 1].cpu() - ref[i]).abs().max()) for i in items}
    parity_local = max(parity_items.values())

    row = [0.0] * sharding.METRIC_COLUMNS
    row[sharding.M_ELAPSED] = elapsed
    row[sharding.M_HOST] = issued
    row[sharding.M_VALID_FRAMES] = valid_frames_local
    row[sharding.M_PADDED_FRAMES] = padded_local
    row[sharding.M_UTTERANCES] = utts_local
    row[sharding.M_PARITY] = parity_local
    row[sharding.M_CHECKSUM] = float(wav.double().sum())
    table = sharding.gather_metrics(row, device=mdev)
    job = sharding.summarize(table, args.steps)        # whole-job value = all ranks' audio / slowest rank's time
    failures = []
    if job["parity_max_abs"] > tol:
        failures.append(f"timed output differs from the CPU oracle by {job['parity_max_abs']:.3e} (> {tol})")

    corpus = None
    if args.workload == "batch" and args.corpus > 0 and args.corpus_steps > 0:
        corpus, _ = corpus_pass(args.corpus_steps, warmup=1)
        if corpus["parity_max_abs"] > tol:
            failures.append(f"corpus output differs from the CPU oracle by {corpus['parity_max_abs']:.3e} (> {tol})")

    if rank == 0:
        fam = fam_all                                   # per-family table of the surveyed (untimed) step
        name, dom = dominant, timer.summary()[dominant] # the dominant family over the timed region
        def make_roofline(prec, name, dom, fam):
            """The roofline object of kernel family ``name``: ``dom`` holds its launches' summed FLOPs,
            bytes and device time, ``fam`` the table of every family of one step."""
            traffic, traffic_stale = None, None     # HBM bytes per launch from the committed counter passes
            tname = "hbm_traffic.json" if prec == "fp32" else f"hbm_traffic_{prec}.json"
            tpath = os.path.join(ROOT, "profiles", tname)
            if args.workload == "batch" and (B, T) == (16, 862) and os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic = tj["by_family"].get(name, {}).get("hbm_bytes_per_launch")
                traffic_stale = tj.get("csrc_hash") != csrc_hash()
            achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            total_ms = sum(d["ms"] for d in fam.values())
            total_flops = sum(d["flops"] for d in fam.values())
            common = {"traffic": traffic, "traffic_stale": traffic_stale,
                      "traffic_note": ("HBM bytes per launch, rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes "
                                       f"stored in profiles/{tname}") if traffic else None,
                      "kernel": name, "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                      "gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
                      "mb_per_launch": dom["bytes"] / dom["launches"] / 1e6,
                      "share_of_device_time": fam[name]["ms"] / total_ms,
                      "all_kernels_tflops": total_flops / (total_ms * 1e-3) / 1e12}
            if prec == "fp32":
                r = {"bound": "mfma", "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_MATRIX_PEAK_TFLOPS}
            elif prec == "fp32-split":
                # six bf16 slice products per fp32 product: the matrix-core ceiling in algorithmic FLOPs
                peak = BF16_MATRIX_PEAK_TFLOPS / 6.0
                r = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "peak_note": "dense bf16 MFMA peak / 6 slice products per fp32 product"}
            else:
                # bf16 matrix inputs: whichever of the two roofs the family's launches sit under
                t_hbm = dom["bytes"] / (HBM_PEAK_GBS * 1e9)
                t_mfma = dom["flops"] / (BF16_MATRIX_PEAK_TFLOPS * 1e12)
                if t_hbm >= t_mfma:
                    gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                    r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS}
                else:
                    r = {"bound": "mfma", "achieved": achieved, "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / BF16_MATRIX_PEAK_TFLOPS}
            r.update(common)
            return r

        roof = make_roofline(args.precision, name, dom, fam)
        line = {
            "metric": "audio seconds generated/sec (RTF^-1), 10 s utterances",
            "value": job["value"], "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": job["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if args.workload == "batch" else "strong",
            "vs_baseline": None,
            "dtype": DTYPE_NAMES[args.precision],
            "data": "synthetic",
            "config": {"workload": ("reverse flow + HiFi-GAN decoder (TTS hot path), "
                                    f"batch {B} x {T} frames ({T * HOP / SAMPLE_RATE:.2f} s) per GPU, 100 speakers, "
                                    "random-init weights") if args.workload == "batch" else
                                   ("reverse flow + HiFi-GAN decoder over a length-bucketed synthetic corpus of "
                                    f"{args.corpus} utterances (2-15 s), sharded over the ranks, 100 speakers, "
                                    "random-init weights"),
                       "batch_per_gpu": B if args.workload == "batch" else None,
                       "frames": T if args.workload == "batch" else None, "sample_rate": SAMPLE_RATE,
                       "parallelism": f"utterance-sharded x{world}, no hot-path collective"},
            "roofline": roof,
            "per_gpu_value": job["value"] / world,
            "ranks_seen": job["ranks_seen"],
            "per_rank_ms_per_step": job["per_rank_ms_per_step"],
            "per_rank_host_issue_ms_per_step": job["per_rank_host_issue_ms_per_step"],
            "load_imbalance": job["load_imbalance"],
            "parity": {"max_abs": job["parity_max_abs"], "items": items, "tol": tol,
                       "against": "CPU oracle on the same latents, every rank's first and last item",
                       "rank0_items": parity_items},
        }
        if corpus is not None:
            line["corpus"] = corpus
        if world == 1 and args.workload == "batch" and not args.no_other_precisions:
            # the same step in the other matrix-core input modes (3 timed steps each), each checked
            # against the oracle waveforms of the same two items
            others = {}
            for prec in ("fp32", "fp32-split", "bf16"):
                if prec == args.precision:
                    continue
                engine.set_precision(prec)
                w2 = step()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(3):
                    w2 = step()
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t1) / 3
                sv = native.LaunchTimer()           # one more step with every launch bracketed
                with sv:
                    step()
                torch.cuda.synchronize(dev)
                fam2 = sv.summary()
                dom2 = max(fam2.items(), key=lambda kv: kv[1]["ms"])[0]
                others[prec] = {"ms_per_step": 1e3 * dt, "value": B * T * HOP / SAMPLE_RATE / dt,
                                "dtype": DTYPE_NAMES[prec],
                                "parity_max_abs": max(float((w2[i:i + 1].cpu() - ref[i]).abs().max()) for i in items),
                                "tol": BF16_TOL if prec == "bf16" else PARITY_TOL,
                                "roofline": make_roofline(prec, dom2, fam2[dom2], fam2),
                                "roofline_note": "dominant family of one step with every launch bracketed by events"}
                if others[prec]["parity_max_abs"] > others[prec]["tol"]:
                    failures.append(f"{prec}: differs from the CPU oracle by {others[prec]['parity_max_abs']:.3e}")
            engine.set_precision(args.precision)
            line["other_precisions"] = others
        if world == 1 and args.workload == "batch" and not args.no_other_precisions:
            # The same step with the resblocks of a decoder stage on streams of their own (the small-batch
            # path, forced on): the launches of one branch fill the tails of another's.  Not the timed region:
            # launches that share the GPU cannot be bracketed one by one, so `roofline` needs the serial order.
            before_env = os.environ.get("VITS_AMD_BRANCH_STREAMS")
            os.environ["VITS_AMD_BRANCH_STREAMS"] = "on"
            try:
                for _ in range(2):
                    w3 = step()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    w3 = step()
                torch.cuda.synchronize(dev)
                dt = (time.perf_counter() - t1) / args.steps
            finally:
                if before_env is None:
                    del os.environ["VITS_AMD_BRANCH_STREAMS"]
                else:
                    os.environ["VITS_AMD_BRANCH_STREAMS"] = before_env
            line["overlapped_branches"] = {
                "ms_per_step": 1e3 * dt, "value": B * T * HOP / SAMPLE_RATE / dt, "steps": args.steps,
                "switch": "VITS_AMD_BRANCH_STREAMS=on",
                "same_bits_as_timed_output": bool(torch.equal(w3, wav)),
                "note": "resblocks of a stage on three streams; the default keeps them on one stream above "
                        "2048 frames in flight so that every launch's device time is its own"}
        if world == 1 and not args.no_other_configs:
            from tools import bench_configs
            line["other_configs"] = bench_configs.run(dev, SEED)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, T, args.cpu_seconds)
        if os.environ.get("BENCH_VERBOSE"):
            for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"# {k:44s} n={d['launches']:5d} ms={d['ms']:9.2f} "
                      f"TF={d['flops'] / max(d['ms'], 1e-9) / 1e9:7.2f}", file=sys.stderr)
            if os.environ.get("BENCH_VERBOSE") == "2":
                for fam_, fl, _, e0, e1, det in survey.records:
                    ms = e0.elapsed_time(e1)
                    print(f"## {fam_:40s} {det:32s} ms={ms:8.3f} TF={fl / max(ms, 1e-9) / 1e9:7.2f}", file=sys.stderr)
        if failures:
            line["failed"] = failures
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failures:
        raise SystemExit("bench.py: " + "; ".join(failures))


if __name__ == "__main__":
    main()
