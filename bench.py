#!/usr/bin/env python3
"""Benchmark of the VITS generator hot path on MI355X: reverse flow + HiFi-GAN decoder.

  python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path (flow reverse -> decoder) over one batch of
synthetic latents already resident in HBM: 16 utterances x 862 frames (10.008 s each at
22050 Hz), 16 seeded speakers out of 100, random-init weights from vits_amd.synth.
For N > 1 the driver starts one process per GPU (torch.distributed.run); every rank runs
the same per-GPU batch (weak scaling, utterance-sharded, no collective in the hot path)
and RCCL is used only for the barrier and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line: audio seconds generated per wall second (whole job), a
roofline object for the kernel family that took the most device time (algorithmic FLOPs
of its launches / their summed device time from per-launch events on the launch stream),
and - at N = 1 - the CPU oracle timed on this host's cores on one 10 s utterance.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vits_amd import native, synth                                   # noqa: E402
from vits_amd.module.vits_generator import HotPathGenerator          # noqa: E402

SAMPLE_RATE = 22050
HOP = 256
BF16_MATRIX_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0
FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
SEED = 999


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=862, help="latent frames per utterance (862 = 10 s)")
    ap.add_argument("--workload", choices=["batch", "corpus"], default="batch",
                    help="batch: fixed batch per GPU (weak scaling, the headline); corpus: a length-bucketed "
                         "synthetic corpus sharded over the ranks (strong scaling)")
    ap.add_argument("--corpus", type=int, default=4096, help="utterances in the corpus workload")
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp32-split"], default="fp32",
                    help="matrix-core input precision; fp32 is the <= 1e-4 parity path and the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip the short untimed-region runs of the other precision modes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    return ap.parse_args()


def cpu_baseline(sd, frames: int, budget_s: float):
    """The CPU oracle (stock PyTorch CPU ops, the kernels the reference dispatches to) on one
    utterance of the same length, repeated for about ``budget_s`` seconds."""
    from oracle import vits_oracle as O
    z = synth.latent(SEED, "bench.cpu.z", 1, frames)
    mask = synth.length_mask([frames], frames)
    g = O.speaker_embed(torch.tensor([9]), sd)
    with torch.no_grad():
        O.flow_reverse_decode(z, mask, g, sd)                     # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            O.flow_reverse_decode(z, mask, g, sd)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 100:
                break
    audio = n * frames * HOP / SAMPLE_RATE
    return {"value": audio / el, "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x (1 utterance x {frames} frames = {frames * HOP / SAMPLE_RATE:.2f} s): "
                      f"flow-reverse + decoder, fp32, {el:.1f} s of CPU time"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from vits_amd import engine
    engine.set_precision(args.precision)
    dist = None
    if world > 1 or "RANK" in os.environ:           # launched by torch.distributed.run, any world size
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    torch.set_grad_enabled(False)
    sd = synth.hot_path_state_dict(SEED, with_posterior=False)
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(dev).eval()

    B, T = args.batch, args.frames
    key = f"bench.rank{rank}"
    if args.workload == "batch":
        z_p = synth.latent(SEED, key, B, T).to(dev)
        lengths = torch.full((B,), T, dtype=torch.int64)
        mask = synth.length_mask(lengths, T).to(dev)
        ids = synth.speaker_ids(SEED, key, B).to(dev)
        out_shape = (B, 1, T * HOP)
        valid_frames_local = float(lengths.sum().item())

        def step():
            g = gen.embed_speaker(ids)
            return gen.flow_reverse_decode(z_p, mask, g)
    else:
        from vits_amd import sharding
        frames, spk = sharding.synthetic_corpus(args.corpus, SEED)
        buckets = sharding.make_buckets(frames)
        parts, _ = sharding.assign_buckets(frames, buckets, world)
        work = []
        for bi in parts[rank]:
            idx = buckets[bi]
            lens = torch.tensor([int(frames[i]) for i in idx])
            tmax = int(lens.max())
            m = synth.length_mask(lens, tmax)
            work.append((synth.latent(SEED, f"corpus.b{bi}", len(idx), tmax).mul_(m).to(dev), m.to(dev),
                         torch.tensor([int(spk[i]) for i in idx]).to(dev)))
        valid_frames_local = float(sum(int(frames[i]) for bi in parts[rank] for i in buckets[bi]))
        out_shape = None

        def step():
            wav_ = None
            for zb, mb, sb in work:
                wav_ = gen.flow_reverse_decode(zb, mb, gen.embed_speaker(sb))
            return wav_

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

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
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert (out_shape is None or wav.shape == out_shape) and bool(torch.isfinite(wav).all())

    total_frames = valid_frames_local
    if dist is not None:
        tf = torch.tensor([valid_frames_local], device=dev, dtype=torch.float64)
        dist.all_reduce(tf, op=dist.ReduceOp.SUM)
        total_frames = float(tf.item())
    value = total_frames * HOP / SAMPLE_RATE * args.steps / elapsed

    if rank == 0:
        fam = fam_all                                   # per-family table of the surveyed (untimed) step
        name, dom = dominant, timer.summary()[dominant] # the dominant family over the timed region
        traffic = None      # HBM bytes per launch from the committed counter passes of this same workload
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
        if args.workload == "batch" and (B, T) == (16, 862) and os.path.exists(tpath):
            traffic = json.load(open(tpath))["by_family"].get(name, {}).get("hbm_bytes_per_launch")
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        total_ms = sum(d["ms"] for d in fam.values())
        total_flops = sum(d["flops"] for d in fam.values())
        if args.precision == "bf16":
            # bf16 matrix inputs: the convolutions turn HBM-bound (fp32 tensors in memory)
            t_hbm = dom["bytes"] / (HBM_PEAK_GBS * 1e9)
            t_mfma = dom["flops"] / (BF16_MATRIX_PEAK_TFLOPS * 1e12)
            if t_hbm >= t_mfma:
                gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": name, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "traffic": None}
            else:
                roof = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": BF16_MATRIX_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": achieved / BF16_MATRIX_PEAK_TFLOPS, "traffic": None}
            roof.update({"launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                         "gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
                         "mb_per_launch": dom["bytes"] / dom["launches"] / 1e6,
                         "share_of_device_time": fam[name]["ms"] / total_ms,
                         "all_kernels_tflops": total_flops / (total_ms * 1e-3) / 1e12})
        roof_fp32 = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch, rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes "
                                         "stored in profiles/hbm_traffic.json" if traffic else None,
                         "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                         "gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
                         "share_of_device_time": fam[name]["ms"] / total_ms,
                         "all_kernels_tflops": total_flops / (total_ms * 1e-3) / 1e12}
        if args.precision == "fp32-split":
            # six bf16 slice products per fp32 product: the matrix-core ceiling in algorithmic FLOPs
            peak = BF16_MATRIX_PEAK_TFLOPS / 6.0
            roof = dict(roof_fp32, peak=peak, frac=achieved / peak, traffic=None, traffic_note=None,
                        peak_note="dense bf16 MFMA peak / 6 slice products per fp32 product")
        line = {
            "metric": "audio seconds generated/sec (RTF^-1), 10 s utterances",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak" if args.workload == "batch" else "strong",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 matrix inputs, f32 accumulate, f32 tensors",
                      "fp32-split": "f32 operands as 3 exact bf16 slices, 6 slice products, f32 accumulate"}[args.precision],
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
            "roofline": roof_fp32 if args.precision == "fp32" else roof,
            "per_gpu_value": value / world,
        }
        if world == 1 and args.workload == "batch" and not args.no_other_precisions:
            # the same step in the other matrix-core input modes (3 timed steps each), with the
            # largest waveform difference from the timed mode's output
            others = {}
            ref_wav = wav
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
                others[prec] = {"ms_per_step": 1e3 * dt, "value": total_frames * HOP / SAMPLE_RATE / dt,
                                "max_abs_diff_vs_timed_mode": float((w2 - ref_wav).abs().max().item())}
            engine.set_precision(args.precision)
            line["other_precisions"] = others
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth.hot_path_state_dict(SEED, with_posterior=False),
                                                T, args.cpu_seconds)
        if os.environ.get("BENCH_VERBOSE"):
            for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"# {k:44s} n={d['launches']:5d} ms={d['ms']:9.2f} "
                      f"TF={d['flops'] / max(d['ms'], 1e-9) / 1e9:7.2f}", file=sys.stderr)
            if os.environ.get("BENCH_VERBOSE") == "2":
                for fam_, fl, _, e0, e1, det in survey.records:
                    ms = e0.elapsed_time(e1)
                    print(f"## {fam_:40s} {det:32s} ms={ms:8.3f} TF={fl / max(ms, 1e-9) / 1e9:7.2f}", file=sys.stderr)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
