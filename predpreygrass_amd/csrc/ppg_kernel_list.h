// ppg_kernel_list.h -- the kernel instantiations of libppg_hip.so.  The including unit defines PPG_K / PPG_K2
// (a definition in ppg_kernels.hip, a declaration in ppg_hip.hip).
#pragma once

#define PPG_DEFINE_KERNELS(NQ)                                        \
    PPG_K(ppg_step_q##NQ, NQ, ppg::MODE_STEP, true)                   \
    PPG_K(ppg_reset_q##NQ, NQ, ppg::MODE_RESET, true)                 \
    PPG_K(ppg_observe_q##NQ, NQ, ppg::MODE_OBSERVE, true)             \
    PPG_K(ppg_grid_q##NQ, NQ, ppg::MODE_EXPORT_GRID, true)            \
    PPG_K(ppg_step_ord_q##NQ, NQ, ppg::MODE_STEP_ORDERED, true)       \
    PPG_K(ppg_rollout_q##NQ, NQ, ppg::MODE_ROLLOUT, true)             \
    PPG_K(ppg_step_kick_q##NQ, NQ, ppg::MODE_STEP_KICK, true)         \
    PPG_K(ppg_step_ord_kick_q##NQ, NQ, ppg::MODE_STEP_ORDERED_KICK, true) \
    PPG_K(ppg_step_q##NQ##g, NQ, ppg::MODE_STEP, false)               \
    PPG_K(ppg_reset_q##NQ##g, NQ, ppg::MODE_RESET, false)             \
    PPG_K(ppg_observe_q##NQ##g, NQ, ppg::MODE_OBSERVE, false)         \
    PPG_K(ppg_grid_q##NQ##g, NQ, ppg::MODE_EXPORT_GRID, false)        \
    PPG_K(ppg_step_ord_q##NQ##g, NQ, ppg::MODE_STEP_ORDERED, false)   \
    PPG_K(ppg_rollout_q##NQ##g, NQ, ppg::MODE_ROLLOUT, false)         \
    PPG_K(ppg_step_kick_q##NQ##g, NQ, ppg::MODE_STEP_KICK, false)     \
    PPG_K(ppg_step_ord_kick_q##NQ##g, NQ, ppg::MODE_STEP_ORDERED_KICK, false)

// multi-wave step kernels of the base family (4, 8 or 2 wavefronts per env; see Env's NW): ppgw_step_q<NQ>[g], ppgw8_step_*, ppgwp_step_* (a pair)
#define PPG_DEFINE_KERNELSW(NQ)                                       \
    PPG_KW(ppgw_step_q##NQ, NQ, true, 4)                              \
    PPG_KW(ppgw_step_q##NQ##g, NQ, false, 4)                          \
    PPG_KW(ppgw8_step_q##NQ, NQ, true, 8)                             \
    PPG_KW(ppgw8_step_q##NQ##g, NQ, false, 8)                         \
    PPG_KW(ppgwp_step_q##NQ, NQ, true, 2)                             \
    PPG_KW(ppgwp_step_q##NQ##g, NQ, false, 2)                         \
    PPG_KW(ppgw16_step_q##NQ, NQ, true, 16)

// cooperative step kernels (Env's COOP): coop_e envs per workgroup of 4 / 8 / 16 wavefronts; ppgc2_*: second generation; ppgcm_* / ppgcm2_*:
// the four-wave kernels without a channel-0 cell map (large grids)
#define PPG_DEFINE_KERNELSC(NQ)                                       \
    PPG_KC(ppgc_step_q##NQ, NQ, false, 4)                             \
    PPG_KCH(ppgch_step_q##NQ, NQ)                                     \
    PPG_KC(ppgc6_step_q##NQ, NQ, false, 6)                            \
    PPG_KC(ppgc8_step_q##NQ, NQ, false, 8)                            \
    PPG_KC(ppgc16_step_q##NQ, NQ, false, 16)                          \
    PPG_KC(ppgc2_step_q##NQ, NQ, true, 4)                             \
    PPG_KCM(ppgcm_step_q##NQ, NQ, false)                              \
    PPG_KCM(ppgcm2_step_q##NQ, NQ, true)                              \
    PPG_KCR(ppgc_rollout_q##NQ, NQ, false, 4)                         \
    PPG_KCR(ppgc2_rollout_q##NQ, NQ, true, 4)

#define PPG_DEFINE_KERNELSW2(NQ)                                      \
    PPG_KW2(ppgw2_step_q##NQ, NQ, true, 4)                            \
    PPG_KW2(ppgw2_step_q##NQ##g, NQ, false, 4)                        \
    PPG_KW2(ppgw28_step_q##NQ, NQ, true, 8)                           \
    PPG_KW2(ppgw28_step_q##NQ##g, NQ, false, 8)

// walls variant of the second generation: generic observation geometry only (ppg3_<mode>_q<NQ>; ppgw3_step: 4 waves per env, ppgwp3_step: 2;
// ppgc3_step: cooperative, two envs per four-wave workgroup -- up to 128 prey rows)
#define PPG_DEFINE_KERNELS3(NQ)                                       \
    PPG_KW3(ppgw3_step_q##NQ, NQ, 4)                                  \
    PPG_KC3(ppgc3_step_q##NQ, NQ)                                     \
    PPG_KW3(ppgwp3_step_q##NQ, NQ, 2)                                 \
    PPG_K3(ppg3_step_q##NQ, NQ, ppg::MODE_STEP)                       \
    PPG_K3(ppg3_reset_q##NQ, NQ, ppg::MODE_RESET)                     \
    PPG_K3(ppg3_observe_q##NQ, NQ, ppg::MODE_OBSERVE)                 \
    PPG_K3(ppg3_grid_q##NQ, NQ, ppg::MODE_EXPORT_GRID)                \
    PPG_K3(ppg3_step_ord_q##NQ, NQ, ppg::MODE_STEP_ORDERED)          \
    PPG_K3(ppg3_vis_q##NQ, NQ, ppg::MODE_VIS)

// drive-conditioned variant of the base family: generic observation geometry only (ppg4_<mode>_q<NQ>)
#define PPG_DEFINE_KERNELS4(NQ)                                       \
    PPG_KW4(ppgw4_step_q##NQ, NQ, 4)                                  \
    PPG_KW4(ppgwp4_step_q##NQ, NQ, 2)                                 \
    PPG_K4(ppg4_step_q##NQ, NQ, ppg::MODE_STEP)                       \
    PPG_K4(ppg4_reset_q##NQ, NQ, ppg::MODE_RESET)                     \
    PPG_K4(ppg4_observe_q##NQ, NQ, ppg::MODE_OBSERVE)                 \
    PPG_K4(ppg4_grid_q##NQ, NQ, ppg::MODE_EXPORT_GRID)                \
    PPG_K4(ppg4_step_ord_q##NQ, NQ, ppg::MODE_STEP_ORDERED)

#define PPG_DEFINE_KERNELS2(NQ)                                       \
    PPG_K2(ppg2_step_q##NQ, NQ, ppg::MODE_STEP, true)                 \
    PPG_K2(ppg2_reset_q##NQ, NQ, ppg::MODE_RESET, true)               \
    PPG_K2(ppg2_observe_q##NQ, NQ, ppg::MODE_OBSERVE, true)           \
    PPG_K2(ppg2_grid_q##NQ, NQ, ppg::MODE_EXPORT_GRID, true)          \
    PPG_K2(ppg2_step_ord_q##NQ, NQ, ppg::MODE_STEP_ORDERED, true)     \
    PPG_K2(ppg2_step_q##NQ##g, NQ, ppg::MODE_STEP, false)             \
    PPG_K2(ppg2_reset_q##NQ##g, NQ, ppg::MODE_RESET, false)           \
    PPG_K2(ppg2_observe_q##NQ##g, NQ, ppg::MODE_OBSERVE, false)       \
    PPG_K2(ppg2_grid_q##NQ##g, NQ, ppg::MODE_EXPORT_GRID, false)      \
    PPG_K2(ppg2_step_ord_q##NQ##g, NQ, ppg::MODE_STEP_ORDERED, false)
