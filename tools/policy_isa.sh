#!/bin/bash
# CPU only: device assembly of the policy kernels (host unit) -> /tmp/ppg_policy.s, register / scratch summary, and the memory waits of one
# convolution phase.   usage: tools/policy_isa.sh [OBS NCH]   (OBS 0 f64 / 1 f32 / 2 bf16; NCH 4 / 8 / 16)
cd "$(dirname "$0")/../predpreygrass_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -mllvm -pragma-unroll-threshold=1000000 --cuda-device-only -S -o /tmp/ppg_policy.s ppg_hip.hip 2>&1 | grep -v "hip-link"
grep -n "^ppg_policy_forward\|^_ZN6ppgpol\|; NumVgprs\|; ScratchSize" /tmp/ppg_policy.s | awk -F: '/NumVgprs|ScratchSize/{printf " %s", $3; next} {printf "\n%-60s", substr($2,1,58)}' ; echo
v="${1:-2}ELi${2:-4}E"
L=$(grep -n "^_ZN6ppgpol10phase_convILi$v" /tmp/ppg_policy.s | cut -d: -f1)
echo "== phase_conv<$1,$2>: vector-memory waits, barriers, MFMA groups in program order"
awk -v a=$L 'NR>=a' /tmp/ppg_policy.s | awk '/s_setpc_b64/{print; exit} {print}' | grep -n "s_waitcnt vm\|s_barrier\|v_mfma\|global_load\|global_store\|scratch_\|s_setpc" |
  awk '{ k=$2; if (k==last) { n++ } else { if (last!="") printf "%s x%d\n", first, n; first=$0; last=k; n=1 } } END { printf "%s x%d\n", first, n }' | cut -c1-100
