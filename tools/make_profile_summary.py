#!/usr/bin/env python3
"""Assemble profiles/<round>/bench_driver_summary.json from rocprofv3 runs of THE DRIVER'S bench command
(`python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline`: untimed pre-roll to the steady state, 5 warm-up steps, 20 timed
steps, then the sustained leg of 2000 steps; 3 sub-batches), taken on the GPU box by tools/gpu_profile_driver_cmd.sh:
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>_trace -o t -- python3 bench.py <args>
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/<tag>_w -o w -- python3 bench.py <args>
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/<tag>_f -o f -- python3 bench.py <args>
  python3 bench.py <args without --no-cpu-baseline> > gpurun_out/<tag>_bench.json        (unprofiled, same box)
The sustained leg of each run = its last sustained x streams dispatches of the step kernel, the timed region = the steps x streams
dispatches in front of them.
usage: make_profile_summary.py <tag> <kernel name | auto = roofline.kernel of the bench line> <out.json> [steps=20] [streams=3] [sustained=2000]"""
import csv
import glob
import json
import sys

tag, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
streams = int(sys.argv[5]) if len(sys.argv) > 5 else 3
sustained = int(sys.argv[6]) if len(sys.argv) > 6 else 2000
last = steps * streams
tail = sustained * streams


def timed(v):      # the dispatches of the timed region
    return v[len(v) - tail - last: len(v) - tail]


def sust(v):       # the dispatches of the sustained leg
    return v[len(v) - tail:] if tail else []
if kernel == "auto":
    kernel = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])["roofline"]["kernel"]


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    assert files, pattern
    return files[0]


durs = []
info = {}
for r in csv.DictReader(open(one(f"gpurun_out/{tag}_trace/**/*kernel_trace.csv"))):
    if r["Kernel_Name"] == kernel:
        durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        info = {"grid_threads": int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                "workgroup": int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                "scratch": int(r.get("Scratch_Size", 0) or 0), "vgpr": int(r.get("VGPR_Count", 0) or 0),
                "lds": int(r.get("LDS_Block_Size", 0) or 0)}
durs.sort()
d = [x[1] for x in durs]


def pmc_mean(path, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Kernel_Name"] == kernel and r["Counter_Name"] == name]
    vals = timed(vals) + sust(vals)
    return sum(vals) / len(vals), len(vals)


w, nw = pmc_mean(one(f"gpurun_out/{tag}_w/**/*counter_collection.csv"), "WRITE_SIZE")
f, nf = pmc_mean(one(f"gpurun_out/{tag}_f/**/*counter_collection.csv"), "FETCH_SIZE")
b = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
bt = json.loads(open(f"gpurun_out/{tag}_bench_under_trace.json").read().strip().splitlines()[-1])
total = w * 1024 + 2 * f * 1024
kernel_us = sum(timed(d)) / len(timed(d)) / 1e3
kernel_us_sustained = sum(sust(d)) / len(sust(d)) / 1e3 if tail else None
summary = {
    "command": f"python3 bench.py --steps {steps} --warmup 5 --no-cpu-baseline   (the driver's command; defaults --gpus 1 --envs 4096 "
               f"--streams {streams}; untimed pre-roll of {b['config']['preroll_steps']} steps)",
    "workload": "base", "envs_per_gpu": b["config"]["envs_per_gpu"], "obs_dtype": b["dtype"],
    "kernel": kernel,
    "concurrent_launches": streams,
    "mean_agents_per_env": bt["config"]["mean_agents_per_env"],
    "counted_bytes_per_launch": bt["roofline"]["counted_bytes_per_launch"],
    "kernel_trace": dict(info, dispatches=len(d), mean_us_all=sum(d) / len(d) / 1e3, mean_us_timed_region=kernel_us,
                         timed_dispatches=len(timed(d)), mean_us_sustained_leg=kernel_us_sustained, sustained_dispatches=len(sust(d))),
    "bench_py_under_trace": {"kernel_ms": bt["roofline"]["kernel_ms"], "kernel_ms_sustained": bt["roofline"].get("kernel_ms_sustained"),
                             "value": bt["value"], "frac": bt["roofline"]["frac"], "frac_sustained": bt["roofline"].get("frac_sustained")},
    "bench_py_same_box_unprofiled": {"value": b["value"], "ms_per_step": b["ms_per_step"], "kernel_ms": b["roofline"]["kernel_ms"],
                                     "kernel_ms_sustained": b["roofline"].get("kernel_ms_sustained"),
                                     "frac": b["roofline"]["frac"], "frac_sustained": b["roofline"].get("frac_sustained"),
                                     "mean_agents_per_env": b["config"]["mean_agents_per_env"],
                                     "fallback_spawn_envs": b["config"].get("fallback_spawn_envs"),
                                     "cpu_baseline": b.get("cpu_baseline")},
    "pmc_passes": f"separate runs of the same command: --pmc FETCH_SIZE | --pmc WRITE_SIZE; means over the {last} dispatches of the "
                  f"timed region and the {tail} of the sustained leg",
    "pmc_mean_per_launch": {"FETCH_SIZE": f, "WRITE_SIZE": w, "dispatches_used": [nf, nw]},
    "hbm_traffic_per_launch_bytes": {
        "write": w * 1024, "fetch_raw": f * 1024, "fetch_corrected_x2_gfx950": 2 * f * 1024, "total_corrected": total,
        "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reads exactly 1/2 of a wide coalesced read on gfx950 (doubled "
                "here); WRITE_SIZE is exact for 16-B-per-lane streaming stores. Units KB -> x1024."},
    "roofline_check": {
        "frac_from_pmc_bytes_and_rocprof_duration": streams * total / (kernel_us * 1e-6) / 8e12,
        "frac_sustained_from_pmc_bytes_and_rocprof_duration": None if not tail else streams * total / (kernel_us_sustained * 1e-6) / 8e12,
        "frac_bench_py_under_trace": bt["roofline"]["frac"], "frac_bench_py_unprofiled": b["roofline"]["frac"],
        "frac_sustained_bench_py_under_trace": bt["roofline"].get("frac_sustained"),
        "frac_sustained_bench_py_unprofiled": b["roofline"].get("frac_sustained")},
}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("kernel_trace", "hbm_traffic_per_launch_bytes", "roofline_check")}, indent=1))
