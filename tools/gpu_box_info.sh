#!/bin/bash
# On the GPU box: what kind of box is this?  partition modes, memory info, and a quick placement probe.
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i "partition" | head -6
for f in /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/mem_info_vram_total /sys/class/drm/card*/device/mem_info_vram_used /sys/class/drm/card*/device/mem_info_vis_vram_total /sys/class/drm/card*/device/vbios_version /sys/class/drm/card*/device/unique_id; do [ -r $f ] && echo "$f: $(cat $f)"; done
echo "== rocminfo"; rocminfo 2>/dev/null | grep -i -E "Marketing Name|Compute Unit|Max Clock|Chip ID|ASIC Revision|Cacheline|Size:.*KB|Uuid" | head -24
echo "== uptime / kernel"; uptime; uname -r; cat /proc/driver/amdgpu/version 2>/dev/null | head -2
echo "== firmware"; rocm-smi --showfwinfo 2>/dev/null | grep -i -E "MEC|SMC|VBIOS|SDMA|RLC|PSP|TA|ASD" | head -14
echo "== probe"; python3 tools/exp_churn.py 32 2>&1 | grep "round 0"
