// ppg_emu_part.cpp -- TEST-ONLY: the kernel instantiations of one (family, prey registers) pair for the wave emulator,
//   g++ -c -DPPG_EMU_FAMILY=<0 base | 1 second generation | 2 walls | 3 drive> -DPPG_EMU_NQ=<1|2|4> ppg_emu_part.cpp
// (the counterpart of predpreygrass_amd/csrc/ppg_kernels.hip; dispatch and the C ABI: ppg_emu.cpp).
#include "wave_emu.h"

#include "../../predpreygrass_amd/csrc/ppg_kernel.h"

#if !defined(PPG_EMU_FAMILY) || !defined(PPG_EMU_NQ)
#error "define PPG_EMU_FAMILY and PPG_EMU_NQ"
#endif

namespace {

constexpr int NQ = PPG_EMU_NQ;
constexpr bool GEN2 = PPG_EMU_FAMILY == 1 || PPG_EMU_FAMILY == 2, WALLS = PPG_EMU_FAMILY == 2, DRIVE = PPG_EMU_FAMILY == 3;

template <int MODE, bool FAST>
void run1(const ppg::KParams &P) {
    PPG_DYNAMIC_LDS(lds);
    ppg::env_main<NQ, MODE, FAST, GEN2, WALLS, DRIVE>(P, lds);
}
template <bool FAST, int NW>
void run_nw(const ppg::KParams &P) {
    PPG_DYNAMIC_LDS(lds);
    ppg::env_main<NQ, ppg::MODE_STEP, FAST, GEN2, WALLS, DRIVE, NW>(P, lds);
}

#if PPG_EMU_FAMILY <= 2 && PPG_EMU_NQ <= 2
template <int NW>
void run_coop(const ppg::KParams &P) {
    PPG_DYNAMIC_LDS(lds);
    if constexpr (WALLS) { ppg::coop_walls_main<NQ>(P, lds); return; }                      // ppgc3_step
    if (NW == 4 && !P.ch0_map) { ppg::coop_main<NQ, GEN2, 4, false>(P, lds); return; }   // three cell maps per env (ppgcm_*)
    ppg::coop_main<NQ, GEN2, NW>(P, lds);
}
#endif

template <bool FAST>
void run(const ppg::KParams &P, int mode, int nw) {
#if PPG_EMU_FAMILY == 0 && PPG_EMU_NQ <= 2
    if (P.coop_e > 0) {   // cooperative kernels (Env's COOP)
        if (mode == ppg::MODE_ROLLOUT) { PPG_DYNAMIC_LDS(lds); ppg::coop_main_fused<NQ, false, 4>(P, lds); return; }   // ppg_rollout
        if (nw == 16) run_coop<16>(P); else if (nw == 8) run_coop<8>(P); else if (nw == 6) run_coop<6>(P); else run_coop<4>(P);
        return;
    }
#elif PPG_EMU_FAMILY == 1 && PPG_EMU_NQ <= 2
    if (P.coop_e > 0) {
        if (mode == ppg::MODE_ROLLOUT) { PPG_DYNAMIC_LDS(lds); ppg::coop_main_fused<NQ, true, 4>(P, lds); return; }   // ppg_rollout
        run_coop<4>(P);
        return;
    }
#elif PPG_EMU_FAMILY == 2 && PPG_EMU_NQ <= 2
    if (P.coop_e > 0) { run_coop<4>(P); return; }   // ppgc3_step
#endif
    if (nw == 4) { run_nw<FAST, 4>(P); return; }
#if PPG_EMU_FAMILY != 1
    if (nw == 2) { run_nw<FAST, 2>(P); return; }
#endif
#if PPG_EMU_FAMILY < 2
    if (nw == 8) { run_nw<FAST, 8>(P); return; }
#endif
    switch (mode) {
        case ppg::MODE_STEP: run1<ppg::MODE_STEP, FAST>(P); break;
        case ppg::MODE_RESET: run1<ppg::MODE_RESET, FAST>(P); break;
        case ppg::MODE_OBSERVE: run1<ppg::MODE_OBSERVE, FAST>(P); break;
        case ppg::MODE_STEP_ORDERED: run1<ppg::MODE_STEP_ORDERED, FAST>(P); break;
#if PPG_EMU_FAMILY == 0
        case ppg::MODE_ROLLOUT: run1<ppg::MODE_ROLLOUT, FAST>(P); break;
        case ppg::MODE_STEP_KICK: run1<ppg::MODE_STEP_KICK, FAST>(P); break;
        case ppg::MODE_STEP_ORDERED_KICK: run1<ppg::MODE_STEP_ORDERED_KICK, FAST>(P); break;
#endif
#if PPG_EMU_FAMILY == 2
        case ppg::MODE_VIS: run1<ppg::MODE_VIS, FAST>(P); break;
#endif
        default: run1<ppg::MODE_EXPORT_GRID, FAST>(P); break;
    }
}

}  // namespace

#define PPG_EMU_NAME2(F, Q) ppg_emu_run_f##F##_q##Q
#define PPG_EMU_NAME(F, Q) PPG_EMU_NAME2(F, Q)
void PPG_EMU_NAME(PPG_EMU_FAMILY, PPG_EMU_NQ)(const ppg::KParams &P, int mode, int nw, bool fast) {
#if PPG_EMU_FAMILY < 2
    if (fast) { run<true>(P, mode, nw); return; }   // (walls / drive: generic observation geometry only)
#endif
    (void)fast;
    run<false>(P, mode, nw);
}
