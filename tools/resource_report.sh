#!/bin/bash
# Compile every kernel unit of libppg_hip.so with -Rpass-analysis=kernel-resource-usage and tabulate registers / scratch / spills
# per kernel (CPU only: hipcc cross-compiles gfx950).   usage: tools/resource_report.sh > profiles/rNN/kernel_resources.txt
cd "$(dirname "$0")/../predpreygrass_amd/csrc" || exit 1
tmp=$(mktemp -d)
units=""
for g in 1 2 3 4; do for q in 1 2 4; do units="$units $g,$q"; done; done
units="$units 5,1 5,2"
for u in $units; do
  g=${u%,*}; q=${u#*,}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -mllvm -pragma-unroll-threshold=1000000 -DPPG_TU_GEN=$g -DPPG_TU_NQ=$q \
      -Rpass-analysis=kernel-resource-usage -c -o $tmp/k_${g}_${q}.o ppg_kernels.hip > $tmp/r_${g}_${q}.txt 2>&1 ) &
  while [ "$(jobs -r | wc -l)" -ge 7 ]; do sleep 1; done
done
wait
echo "# kernel-resource-usage of every step / reset / observe kernel (hipcc -O3 --offload-arch=gfx950, $(date -u +%Y-%m-%d))"
echo "# unit(gen,nq) kernel VGPRs scratch_bytes_per_lane waves_per_SIMD SGPR_spills VGPR_spills"
for u in $units; do
  g=${u%,*}; q=${u#*,}
  grep -E "Function Name|VGPRs:|ScratchSize|SGPRs Spill|VGPRs Spill|Occupancy" $tmp/r_${g}_${q}.txt | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - - |
    awk -v u="g$g,q$q" '{print u, $2, $4, $7, $10, $13, $16}'
done
rm -rf $tmp
