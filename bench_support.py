"""bench_support.py -- what bench.py needs around its measurement and nothing of the measurement itself: the device-state sampler (clocks,
power, temperatures under load), the per-rank records of the N-GPU line, the self-launch of N ranks, the PMC child runs that measure the
step kernel's HBM traffic.  No test code, no oracle (bench.py's cpu_baseline leg is the only place outside tests/ that touches oracle/).
"""
from __future__ import annotations

import json
import os
import sys
import time  # noqa: F401  (the sampler's code string uses its own import)


def device_state():
    """Clocks, package power and temperatures as `rocm-smi` reports them right now (one subprocess, ~0.3 s): printed next to the numbers so
    that a slow run can be told from a hot or throttled GPU.  If the tool cannot be run or says nothing the record holds the REASON
    (`unavailable`) instead of silently being null (the driver's round-4 record was)."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if exe is None:
        return {"unavailable": "rocm-smi is neither on PATH nor under /opt/rocm/bin"}
    try:
        res = subprocess.run([exe, "--showtemp", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20)
        out = res.stdout
    except Exception as exc:
        return {"unavailable": f"{exe}: {type(exc).__name__}: {str(exc)[:160]}"}
    state = {}
    try:   # which GPU of the node this is: the placements a process draws were seen to differ from GPU to GPU
        import torch
        state["gpu_uuid"] = str(torch.cuda.get_device_properties(torch.cuda.current_device()).uuid)
    except Exception:
        pass
    for line in out.splitlines():
        m = re.search(r"GPU\[0\]\s*:\s*(.+?):\s*(.+)$", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key.startswith("Temperature"):
            state["temp_" + key.split("(Sensor ")[-1].split(")")[0].strip().replace(" ", "_") + "_C"] = val
        elif "clock level" in key and key.split()[0] in ("sclk", "mclk", "fclk", "socclk"):
            mm = re.search(r"\((\d+)Mhz\)", val)
            state[key.split()[0] + "_MHz"] = int(mm.group(1)) if mm else val
        elif "Power" in key:
            state["power_W"] = val
    if not any(k != "gpu_uuid" for k in state):
        state["unavailable"] = (f"{exe} exited with {res.returncode} and printed no GPU[0] lines; stderr: {res.stderr.strip()[:200]!r}; "
                                f"stdout starts {out.strip()[:120]!r}")
    return state


_SAMPLER_CODE = r"""
import subprocess, sys, time
exe, path = sys.argv[1], sys.argv[2]
t_end = time.time() + 600.0            # (never outlives a bench run by long, whatever happens to the parent)
while time.time() < t_end:
    t = time.time()
    try:
        out = subprocess.run([exe, "--showtemp", "--showclocks", "--showpower", "--showpids"], capture_output=True, text=True, timeout=20).stdout
    except Exception as exc:
        out = "ERR " + repr(exc)
    with open(path, "a") as f:
        f.write("@@ %.3f %.3f\n%s\n" % (t, time.time(), out))
    time.sleep(0.3)
"""


def start_state_sampler():
    """A side process that asks rocm-smi for clocks / power / temperatures / KFD processes a few times per second, started BEFORE this
    process touches the GPU and with the profiler hooks stripped from its environment: on the GPU boxes a process that has initialised
    the GPU may not exec another program (rocm-smi is a script), and under LD_PRELOAD every child would initialise it.  Returns
    (Popen, path) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if exe is None:
        return None, "rocm-smi is neither on PATH nor under /opt/rocm/bin"
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "HSA_TOOLS"))}
    try:
        fd, path = tempfile.mkstemp(prefix="ppg_bench_state_", suffix=".txt")
        os.close(fd)
        proc = subprocess.Popen([sys.executable, "-c", _SAMPLER_CODE, exe, path], env=env, stdin=subprocess.DEVNULL,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    except Exception as exc:
        return None, f"the sampler could not be started: {type(exc).__name__}: {str(exc)[:160]}"
    return proc, path


def sampled_state(path, t0, t1):
    """The last rocm-smi sample that was taken entirely inside [t0, t1] (wall clock), parsed like device_state(); None if there is none."""
    import re
    try:
        text = open(path).read()
    except OSError:
        return None
    best = None
    for block in text.split("@@ ")[1:]:
        head, _, body = block.partition("\n")
        try:
            ta, tb = (float(x) for x in head.split()[:2])
        except ValueError:
            continue
        if ta >= t0 and tb <= t1 and "GPU[0]" in body:
            best = (ta, tb, body)
    if best is None:
        return None
    state = {"sampled_by": "a side process started before this one touched the GPU (rocm-smi, %.2f s per call)" % (best[1] - best[0])}
    pids = 0
    in_pids = False
    for line in best[2].splitlines():
        if line.startswith("PID"):
            in_pids = True
            continue
        if in_pids and re.match(r"^\d+\s", line):
            pids += 1
        m = re.search(r"GPU\[0\]\s*:\s*(.+?):\s*(.+)$", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key.startswith("Temperature"):
            state["temp_" + key.split("(Sensor ")[-1].split(")")[0].strip().replace(" ", "_") + "_C"] = val
        elif "clock level" in key and key.split()[0] in ("sclk", "mclk", "fclk", "socclk"):
            mm = re.search(r"\((\d+)Mhz\)", val)
            state[key.split()[0] + "_MHz"] = int(mm.group(1)) if mm else val
        elif "Power" in key:
            state["power_W"] = val
    # KFD processes on the node's GPUs as rocm-smi lists them (this bench is one of them): more than a handful = the lease shares its GPU
    state["kfd_processes_listed"] = pids
    return state


def gpu_uuid(device, dry=False):
    if dry:
        return "cpu-dry-run"
    try:
        import torch
        return str(torch.cuda.get_device_properties(device).uuid)
    except Exception as exc:
        return f"unavailable ({type(exc).__name__})"


def gather_per_rank(dist, world, mine):
    """Every rank's record on every rank (one small object all-gather, outside every timed region)."""
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def per_rank_summary(per_rank, n_gpus, envs_per_gpu):
    """What the N-GPU line would be if every rank ran at the MEDIAN rank's pace: the distance between this and `value` is the
    slowest GPU's doing, not the code's."""
    if not per_rank:
        return {}
    ms = sorted(r["ms_per_step"] for r in per_rank)
    med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
    return {"per_rank": per_rank,
            "value_if_every_rank_were_median": round(n_gpus * envs_per_gpu / (med * 1e-3), 1),
            "slowest_over_median_rank": round(ms[-1] / med, 4)}


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N ranks (one per GPU) with torch.distributed.run as
    CHILD processes.  Nothing in this parent has touched the GPU (torch is not even imported yet), so no process that has
    initialised HIP is ever replaced; the parent only forwards the children's output and exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + list(argv)
    return subprocess.call(cmd, env=env)


def measure_traffic(argv, args, bench_path, timeout=240):
    """HBM bytes per launch of the step kernel ON THIS BOX, from the PMC counters: two CHILD runs of this very command under
    `rocprofv3 --pmc WRITE_SIZE` / `--pmc FETCH_SIZE` (separate passes, counters only -- MI355X_MICROARCH.md's HBM section), started
    before this process has touched the GPU.  Mean over the dispatches of the child's timed region; FETCH_SIZE doubled (gfx950
    tallies 128-byte requests at 64 bytes), WRITE_SIZE exact for 16-byte-per-lane stores; units KB.  Returns None on any problem
    (no rocprofv3, a refused profiler, a timeout): the line then falls back to the figure under profiles/."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    child_args = [a for a in argv if a not in ("--measure-traffic",)]
    child_args += ["--no-cpu-baseline", "--sustained-steps", "0", "--fused-steps", "0", "--traffic-child", "--device-warm-seconds", "0.5",
                   "--placement-candidates", "1", "--obs-spread", "0"]
    out = {}
    tmp = tempfile.mkdtemp(prefix="ppg_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "c", "--", sys.executable, os.path.abspath(bench_path)] + child_args
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")))
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not line or not files:
                return None
            child = json.loads(line[-1])
            kernel, steps, n_sub = child["roofline"]["kernel"], child["steps"], child["roofline"]["concurrent_launches"]
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r["Kernel_Name"] == kernel and r["Counter_Name"] == counter]
            vals = vals[-steps * n_sub:]
            if len(vals) < steps * n_sub:
                return None
            out[counter] = sum(vals) / len(vals) * 1024.0
            out["kernel"], out["counted"] = kernel, child["roofline"]["counted_bytes_per_launch"]
        out["total"] = out["WRITE_SIZE"] + 2.0 * out["FETCH_SIZE"]
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
