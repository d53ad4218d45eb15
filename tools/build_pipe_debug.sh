#!/bin/bash
# CPU: tools/_build/libppg_hip_pipedbg.so = the product library with the host unit (C ABI + policy kernels) compiled with -DPPG_PIPE_DEBUG:
# the pipeline kernels' private barrier gives up after PPG_PIPE_DEBUG_POLLS polls and reports who waited for what; ppg_policy_act checks
# the report after every launch (tests/test_policy.py::test_pipeline_barrier_status_stays_clean_over_a_rollout).  Never the product.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/_build
cd $root && python3 -c "
import __graft_entry__ as g
print(g.build_hip_variant('$root/tools/_build/libppg_hip_pipedbg.so', host_flags=['-DPPG_PIPE_DEBUG']))"
