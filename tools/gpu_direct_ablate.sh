#!/bin/bash
# On the GPU box: the direct-head policy kernels with parts switched off (timing only, results meaningless): which phase costs what.
# Libraries built beforehand (CPU): python -c "import __graft_entry__ as g; [g.build_hip_variant(os.path.abspath(f"predpreygrass_amd/csrc/libppg_hip_dabl{b}.so"),
#   ('-DPPG_DIRECT_ABLATE=%d' % b,)) for b in (1, 3, 4, 8, 16, 31)]"        usage: tools/gpu_direct_ablate.sh TAG BITS [BITS ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
for bits in 0 "$@"; do
  if [ "$bits" = 0 ]; then unset PPG_HIP_LIB; else export PPG_HIP_LIB=$PWD/predpreygrass_amd/csrc/libppg_hip_dabl$bits.so; fi
  python3 bench.py --workload policy_rollout --steps 20 --warmup 10 --no-cpu-baseline --preroll-min 300 2> gpurun_out/${tag}_dabl_$bits.err |
    python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('PPG_DIRECT_ABLATE=$bits kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])" | tee -a gpurun_out/${tag}_direct_ablate.txt
done
