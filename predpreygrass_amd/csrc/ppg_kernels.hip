// ppg_kernels.hip -- one group of gfx950 kernels of libppg_hip.so per compilation:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c -DPPG_TU_GEN=<1|2|3> -DPPG_TU_NQ=<1|2|4> ppg_kernels.hip
// (1 = base family, 2 = second generation, 3 = second generation with walls, 4 = base family with drive channels, 5 = cooperative
// step kernels of the base family; NQ = prey row registers).  Host code: ppg_hip.hip.
#include <hip/hip_runtime.h>

#include "ppg_kernel.h"

#if !defined(PPG_TU_GEN) || !defined(PPG_TU_NQ)
#error "define PPG_TU_GEN and PPG_TU_NQ"
#endif

// Waves per SIMD the compiler is asked to make room for (the register cap): 4 = 128 VGPRs.  The walls kernels are bound by
// latency, not by registers: at 8 (64 VGPRs, a few spilled) the four-wave step kernel is 12 % faster than at 4 (84 VGPRs = 5).
#ifndef PPG_WPE
#define PPG_WPE 4
#endif
#ifndef PPG_WPE_WALLS
#define PPG_WPE_WALLS 8
#endif
#ifndef PPG_WPE_GEN2
#define PPG_WPE_GEN2 PPG_WPE
#endif
#ifndef PPG_WPE_DRIVE
#define PPG_WPE_DRIVE PPG_WPE
#endif
#define PPG_K(name, NQ, MODE, FAST)                                                         \
    PPG_KERNEL(name, (NQ <= 2 ? PPG_WPE : 2))(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, MODE, FAST>(P, lds); }
#define PPG_K2(name, NQ, MODE, FAST)                                                        \
    PPG_KERNEL(name, (NQ <= 2 ? PPG_WPE_GEN2 : 2))(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, MODE, FAST, true>(P, lds); }
#define PPG_K3(name, NQ, MODE)                                                               \
    PPG_KERNEL(name, (NQ <= 2 ? PPG_WPE_WALLS : 2))(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, MODE, false, true, true>(P, lds); }
#define PPG_K4(name, NQ, MODE)                                                               \
    PPG_KERNEL(name, (NQ <= 2 ? PPG_WPE_DRIVE : 2))(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, MODE, false, false, false, true>(P, lds); }
#define PPG_KW(name, NQ, FAST, NW)                                                           \
    PPG_KERNEL_NW(name, (NQ <= 2 ? PPG_WPE : 2), NW)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, ppg::MODE_STEP, FAST, false, false, false, NW>(P, lds); }
#define PPG_KW2(name, NQ, FAST, NW)                                                          \
    PPG_KERNEL_NW(name, (NQ <= 2 ? PPG_WPE_GEN2 : 2), NW)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, ppg::MODE_STEP, FAST, true, false, false, NW>(P, lds); }
#define PPG_KW3(name, NQ, NW)                                                                \
    PPG_KERNEL_NW(name, (NQ <= 2 ? PPG_WPE_WALLS : 2), NW)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, ppg::MODE_STEP, false, true, true, false, NW>(P, lds); }
#define PPG_KW4(name, NQ, NW)                                                                \
    PPG_KERNEL_NW(name, (NQ <= 2 ? PPG_WPE_DRIVE : 2), NW)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::env_main<NQ, ppg::MODE_STEP, false, false, false, true, NW>(P, lds); }
#ifndef PPG_WPE_COOP
#define PPG_WPE_COOP PPG_WPE
#endif
#define PPG_KC(name, NQ, GEN2, NW)                                                           \
    PPG_KERNEL_NW(name, PPG_WPE_COOP, NW)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::coop_main<NQ, GEN2, NW>(P, lds); }
// the four-wave cooperative kernels without a channel-0 cell map (three maps per env: KParams::ch0_map 0, large grids)
#define PPG_KCM(name, NQ, GEN2)                                                              \
    PPG_KERNEL_NW(name, PPG_WPE_COOP, 4)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::coop_main<NQ, GEN2, 4, false>(P, lds); }
// the same four-wave cooperative kernel at 64 registers (8 wavefronts per SIMD, all 16 envs of a CU resident): for bfloat16 rows, where
// the transitions and not the write streams decide (policy rollouts: 51 -> 44 us per 4096-env step; float64 rows: 62 -> 70 us)
#define PPG_KCH(name, NQ)                                                                    \
    PPG_KERNEL_NW(name, 8, 4)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::coop_main<NQ, false, 4>(P, lds); }
// the cooperative walls kernel (8-bit maps: up to two prey row registers; the four-register unit gets an empty kernel nothing launches)
#ifndef PPG_WPE_COOP_WALLS
#define PPG_WPE_COOP_WALLS PPG_WPE_COOP
#endif
#define PPG_KC3(name, NQ)                                                                    \
    PPG_KERNEL_NW(name, PPG_WPE_COOP_WALLS, 4)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::coop_walls_main<NQ>(P, lds); }
#define PPG_KCR(name, NQ, GEN2, NW)                                                          \
    PPG_KERNEL_NW(name, PPG_WPE, NW)(const ppg::KParams P) { PPG_DYNAMIC_LDS(lds); ppg::coop_main_fused<NQ, GEN2, NW>(P, lds); }
#include "ppg_kernel_list.h"

#define PPG_APPLY(M, NQ) M(NQ)  // expands PPG_TU_NQ before the list pastes it into the kernel names
#if PPG_TU_GEN == 1
PPG_APPLY(PPG_DEFINE_KERNELS, PPG_TU_NQ)
PPG_APPLY(PPG_DEFINE_KERNELSW, PPG_TU_NQ)
#elif PPG_TU_GEN == 2
PPG_APPLY(PPG_DEFINE_KERNELS2, PPG_TU_NQ)
PPG_APPLY(PPG_DEFINE_KERNELSW2, PPG_TU_NQ)
#elif PPG_TU_GEN == 3
PPG_APPLY(PPG_DEFINE_KERNELS3, PPG_TU_NQ)
#elif PPG_TU_GEN == 5
PPG_APPLY(PPG_DEFINE_KERNELSC, PPG_TU_NQ)
#else
PPG_APPLY(PPG_DEFINE_KERNELS4, PPG_TU_NQ)
#endif
