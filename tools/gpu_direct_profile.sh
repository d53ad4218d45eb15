#!/bin/bash
# On the GPU box: where a wavefront of the direct-head policy kernels spends its cycles (diagnostic library built on the CPU:
#   python -c "import os, __graft_entry__ as g; g.build_hip_variant(os.path.abspath('predpreygrass_amd/csrc/libppg_hip_dprof.so'), ('-DPPG_DIRECT_PROFILE',))")
# usage: tools/gpu_direct_profile.sh TAG [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export PPG_HIP_LIB=$PWD/predpreygrass_amd/csrc/libppg_hip_dprof.so PPG_DIRECT_PROFILE_FILE=$PWD/gpurun_out/${tag}_dprof PPG_DIRECT_PROFILE_RUN=${PPG_DIRECT_PROFILE_RUN:-300}
python3 bench.py --workload policy_rollout --steps 20 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/${tag}_dprof.json 2> gpurun_out/${tag}_dprof.err
python3 - <<PY | tee gpurun_out/${tag}_direct_profile.txt
import numpy as np, json
names = ["tile set-up", "conv1", "barrier", "conv2", "barrier", "conv3+", "barrier", "red writes", "barrier", "logits/actions", "stage", "request", "head"]
print(json.loads(open("gpurun_out/${tag}_dprof.json").readlines()[-1])["roofline"])
for sp in ("prey", "pred"):
    a = np.fromfile("gpurun_out/${tag}_dprof." + sp, dtype=np.uint64).reshape(-1, 4, 16).astype(np.float64)
    a = a[a[:, 0, 15] > 0]
    tot = a[:, :, :13].sum(axis=2)
    print(f"== {sp}: {len(a)} workgroups, sub-groups per workgroup {a[:, 0, 15].mean():.1f}, cycles per wavefront {tot.mean():.0f} (100 MHz clock ticks x ?)")
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{names[i]} {a[:, w, i].sum() / tot[:, w].sum() * 100:.1f}%" for i in range(13)))
    per_sg = a[:, :, :13].sum(axis=(0, 1)) / (a[:, :, 15].sum())
    print("  cycles per sub-group and wavefront: " + "  ".join(f"{names[i]} {per_sg[i]:.0f}" for i in range(13)) + f"  total {per_sg.sum():.0f}")
PY
rm -f gpurun_out/${tag}_dprof.prey gpurun_out/${tag}_dprof.pred
