#!/usr/bin/env python3
"""Assemble profiles/<round>/<tag>_bench_default_summary.json from the rocprofv3 outputs of the default bench command.
Run on the GPU box after (see profiles/README.md):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_trace -o t -- python3 bench.py --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_w -o w -- python3 bench.py --no-cpu-baseline --steps 300 --warmup 300
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_f -o f -- python3 bench.py --no-cpu-baseline --steps 300 --warmup 300
  python3 bench.py > gpurun_out/bench_default.json ; python3 bench.py --no-cpu-baseline --streams 1 > gpurun_out/bench_s1.json
usage: make_profile_summary.py <kernel name> <out.json>"""
import csv
import json
import sys

kernel, out = sys.argv[1], sys.argv[2]
durs = []
grid = wg = scratch = vgpr = lds = None
for r in csv.DictReader(open("gpurun_out/p_trace/t_kernel_trace.csv")):
    if r["Kernel_Name"] == kernel:
        durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
        scratch = int(r.get("Scratch_Size", 0) or 0)
        vgpr, lds = int(r.get("VGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0)
durs.sort()
d = [x[1] for x in durs]


def pmc_mean(path, name, last):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Kernel_Name"] == kernel and r["Counter_Name"] == name]
    vals = vals[-last:]
    return sum(vals) / len(vals), len(vals)


w, nw = pmc_mean("gpurun_out/p_w/w_counter_collection.csv", "WRITE_SIZE", 900)
f, nf = pmc_mean("gpurun_out/p_f/f_counter_collection.csv", "FETCH_SIZE", 900)
b = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
b1 = json.loads(open("gpurun_out/bench_s1.json").read().strip().splitlines()[-1])
summary = {
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (defaults: --gpus 1 "
               "--steps 3000 --warmup 300 --envs 4096 --streams 3)",
    "kernel": kernel,
    "envs_per_launch": 4096 / 3,
    "concurrent_launches": 3,
    "kernel_trace": {"dispatches": len(d), "mean_us_all": sum(d) / len(d) / 1e3,
                     "mean_us_timed_region_last_9000": sum(d[-9000:]) / len(d[-9000:]) / 1e3,
                     "grid_threads": grid, "workgroup": wg, "scratch": scratch},
    "bench_py_same_box_unprofiled": {"value": b["value"], "ms_per_step": b["ms_per_step"],
                                     "kernel_ms": b["roofline"]["kernel_ms"], "frac": b["roofline"]["frac"],
                                     "cpu_baseline": b.get("cpu_baseline")},
    "bench_py_streams_1_same_box": {"value": b1["value"], "ms_per_step": b1["ms_per_step"], "frac": b1["roofline"]["frac"]},
    "pmc_passes": "separate runs (bench.py --steps 300 --warmup 300): --pmc FETCH_SIZE | --pmc WRITE_SIZE; means over the last 900 dispatches",
    "pmc_mean_per_launch": {"FETCH_SIZE": f, "WRITE_SIZE": w, "dispatches_used": [nf, nw]},
    "hbm_traffic_per_launch_bytes": {
        "write": w * 1024, "fetch_raw": f * 1024, "fetch_corrected_x2_gfx950": 2 * f * 1024,
        "total_corrected": w * 1024 + 2 * f * 1024,
        "note": "MI355X_MICROARCH.md HBM section: FETCH_SIZE reads exactly 1/2 of a wide coalesced read on gfx950 (doubled "
                "here); WRITE_SIZE is exact for 16-B-per-lane streaming stores. Units KB -> x1024."},
}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("kernel_trace", "hbm_traffic_per_launch_bytes")}, indent=1))
