#!/usr/bin/env python3
"""Ablation builds of the step kernel (never the product): time per 4096-env step for -D variants."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from predpreygrass_amd import _abi
from predpreygrass_amd.config import config_env
CSRC = os.path.join(ROOT, "predpreygrass_amd", "csrc")
variants = {"baseline": [], "no_obs_reads": ["-DPPG_EXP_NO_OBS_READS"], "no_obs_stores": ["-DPPG_EXP_NO_OBS_STORES"]}
variants.update({k: v for k, v in [a.split("=", 1) for a in sys.argv[1:] if "=" in a] and {}} )
for name, flags in variants.items():
    lib_path = os.path.join(ROOT, "gpurun_out", f"libppg_exp_{name}.so")
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    import __graft_entry__ as graft   # the library is built from several translation units
    graft.build_hip(force=True, extra_flags=flags, out=lib_path)
    code = f"""
import ctypes, sys, time, torch
sys.path.insert(0, {ROOT!r})
from predpreygrass_amd import _abi
_abi._lib = _abi.bind(ctypes.CDLL({lib_path!r}))
from predpreygrass_amd.subbatch import SubBatchedPredPreyGrass
from predpreygrass_amd.config import config_env
for n_sub in (1, 3):
    g = SubBatchedPredPreyGrass(config_env, batch_size=4096, n_sub=n_sub, device='cuda:0', seed=0)
    g.reset(); g.synchronize()
    for _ in range(300): g.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 2000
    for _ in range(K): g.step(random_actions=True, auto_reset=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('{name:14s} streams=%d: %.1f us/step' % (n_sub, dt / K * 1e6), flush=True)
"""
    subprocess.run([sys.executable, "-c", code], check=True)
