#!/usr/bin/env python3
"""Kernel trace of ONE steady-state step of the hot path (reverse flow + decoder), to show which
kernels a step launches once plans, tables and buffers exist.

Two modes:
  step_trace.py run [--precision fp32] [--batch 16] [--frames 862]
      (under `rocprofv3 --kernel-trace --output-format csv -- python3 tools/step_trace.py run ...`)
      warm steps, then a marker kernel, one step, a marker kernel.  The marker is a launch of
      torch's `erfinv` on a 1-element tensor, which nothing else in the run uses.
  step_trace.py cut <kernel_trace.csv> <out.csv>
      keeps the rows between the two markers (in start-time order) and prints per-kernel counts.
"""
import csv
import sys


def run(argv):
    import argparse
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from vits_amd import engine, synth
    from vits_amd.module.vits_generator import HotPathGenerator
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=862)
    a = ap.parse_args(argv)
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    engine.set_precision(a.precision)
    sd = synth.hot_path_state_dict(999, with_posterior=False)
    gen = HotPathGenerator(100, with_posterior=False)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(dev).eval()
    z_p = synth.latent(999, "trace", a.batch, a.frames).to(dev)
    mask = synth.length_mask(torch.full((a.batch,), a.frames, dtype=torch.int64), a.frames).to(dev)
    ids = synth.speaker_ids(999, "trace", a.batch).to(dev)
    marker = torch.full((1,), 0.5, device=dev)
    for _ in range(3):
        gen.flow_reverse_decode(z_p, mask, gen.speaker_ref(ids))
    torch.cuda.synchronize()
    torch.erfinv(marker)
    wav = gen.flow_reverse_decode(z_p, mask, gen.speaker_ref(ids))
    torch.erfinv(marker)
    torch.cuda.synchronize()
    print("step traced:", tuple(wav.shape))


def cut(src, dst):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"].lower()]
    if len(marks) != 2:
        raise SystemExit(f"expected 2 marker launches, found {len(marks)}")
    step = rows[marks[0] + 1:marks[1]]
    t0 = int(step[0]["Start_Timestamp"])
    counts = {}
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["index", "kernel", "start_us", "duration_us", "grid", "workgroup", "lds_bytes", "vgprs"])
        for i, r in enumerate(step):
            name = r["Kernel_Name"]
            counts[name] = counts.get(name, 0) + 1
            w.writerow([i, name, f"{(int(r['Start_Timestamp']) - t0) / 1e3:.2f}",
                        f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.2f}",
                        r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                        r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")])
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e6
    print(f"{len(step)} launches in {span:.3f} ms")
    for name, n in sorted(counts.items(), key=lambda kv: -kv[1]):
        print(f"{n:5d}  {name[:110]}")
    foreign = [n for n in counts if not any(k in n for k in
               ("conv_mfma", "conv_bf16", "conv_split", "resblock_pair", "mrf_stage", "wn_layer", "post_conv_tanh",
                "speaker_linear", "res3_chain"))]
    print("kernels that are not the library's:", foreign if foreign else "none")


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run(sys.argv[2:])
    elif len(sys.argv) == 4 and sys.argv[1] == "cut":
        cut(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
