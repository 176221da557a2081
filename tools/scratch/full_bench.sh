#!/bin/bash
mkdir -p gpurun_out/r2e
timeout -k 10 800 python -u bench.py "$@" > gpurun_out/r2e/bench_full.json 2> gpurun_out/r2e/bench_full.err; echo rc=$?
tail -n 3 gpurun_out/r2e/bench_full.err
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r2e/bench_full.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dtype", "parity")})
print(d["roofline"])
for k, v in d.get("other_precisions", {}).items():
    print(k, v["ms_per_step"], v["parity_max_abs"], v["roofline"]["kernel"], v["roofline"]["frac"], v["roofline"]["traffic"], v["roofline"]["traffic_stale"])
if d.get("corpus"): print("corpus", d["corpus"]["value"], d["corpus"]["ms_per_step"], d["corpus"]["parity_max_abs"])
print(d.get("cpu_baseline"))
print(list(d.get("other_configs", {}).keys()), d.get("failed"))
PY
