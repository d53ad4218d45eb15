#!/bin/bash
# CPU: predpreygrass_amd/csrc/libppg_hip_<name>.so = the product library with the HOST unit (C ABI + policy kernels) of git revision REV
# (for tools/gpu_policy_ab.sh).   usage: tools/build_policy_rev.sh REV NAME
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rev=$1; name=$2; tmp=$(mktemp -d)
mkdir -p $tmp/predpreygrass_amd/csrc $tmp/include
for f in ppg_hip.hip ppg_host.h ppg_kernel.h ppg_env_load.h ppg_env_move.h ppg_env_sort.h ppg_env_observe.h ppg_env_coop.h ppg_env_engage.h ppg_env_reproduce.h ppg_env_step.h ppg_pack.h wave.h ppg_kernel_list.h ppg_policy.h; do git -C $root show $rev:predpreygrass_amd/csrc/$f > $tmp/predpreygrass_amd/csrc/$f; done
git -C $root show $rev:include/ppg.h > $tmp/include/ppg.h
cd $tmp/predpreygrass_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -mllvm -pragma-unroll-threshold=1000000 -c -o $tmp/host.o ppg_hip.hip
cd $root/predpreygrass_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libppg_hip_$name.so $tmp/host.o _obj/kernels_*.o
rm -rf $tmp
echo built libppg_hip_$name.so
