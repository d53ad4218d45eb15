// ppg_emu.cpp -- TEST-ONLY build of the C ABI in include/ppg.h that runs the kernel source
// predpreygrass_amd/csrc/ppg_kernel.h on the CPU under tests/wave_emu/wave_emu.h.
// "Device" pointers are host pointers.  Never loaded by predpreygrass_amd.
#include "wave_emu.h"

#include "../../predpreygrass_amd/csrc/ppg_host.h"

#if defined(__x86_64__)
__asm__(
    ".text\n"
    ".globl ppg_emu_ctx_switch\n"
    ".type ppg_emu_ctx_switch,@function\n"
    "ppg_emu_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size ppg_emu_ctx_switch, .-ppg_emu_ctx_switch\n");
#else
#error "wave emulator context switch is written for x86-64"
#endif

struct EmuLaunch {
    const ppg::KParams *P;
    int nq, mode, gen2;
    int nw;  // wavefronts per workgroup: 1, or 4 / 8 for the multi-wave step kernels (MODE_STEP only)
};

// The kernel instantiations live in twelve separately compiled units (ppg_emu_part.cpp: family x prey registers), built in
// parallel by tests/emu_backend.py; family 0 base, 1 second generation, 2 walls, 3 drive.
#define PPG_EMU_PART_DECL(F, NQ) void ppg_emu_run_f##F##_q##NQ(const ppg::KParams &P, int mode, int nw, bool fast);
PPG_EMU_PART_DECL(0, 1) PPG_EMU_PART_DECL(0, 2) PPG_EMU_PART_DECL(0, 4)
PPG_EMU_PART_DECL(1, 1) PPG_EMU_PART_DECL(1, 2) PPG_EMU_PART_DECL(1, 4)
PPG_EMU_PART_DECL(2, 1) PPG_EMU_PART_DECL(2, 2) PPG_EMU_PART_DECL(2, 4)
PPG_EMU_PART_DECL(3, 1) PPG_EMU_PART_DECL(3, 2) PPG_EMU_PART_DECL(3, 4)

static void run_part(const EmuLaunch *L, bool fast) {
    typedef void (*part_fn)(const ppg::KParams &, int, int, bool);
    static const part_fn table[4][3] = {
        {ppg_emu_run_f0_q1, ppg_emu_run_f0_q2, ppg_emu_run_f0_q4}, {ppg_emu_run_f1_q1, ppg_emu_run_f1_q2, ppg_emu_run_f1_q4},
        {ppg_emu_run_f2_q1, ppg_emu_run_f2_q2, ppg_emu_run_f2_q4}, {ppg_emu_run_f3_q1, ppg_emu_run_f3_q2, ppg_emu_run_f3_q4}};
    table[L->gen2][L->nq == 1 ? 0 : L->nq == 2 ? 1 : 2](*L->P, L->mode, L->nw, fast);
}

static void lane_entry(void *arg) {
    const EmuLaunch *L = (const EmuLaunch *)arg;
    // same selection rule as the HIP backend; PPG_EMU_FORCE_GENERIC_OBS=1 exercises the LDS-descriptor path
    static const bool force_generic = getenv("PPG_EMU_FORCE_GENERIC_OBS") != nullptr;
    run_part(L, L->P->nch_p <= 2 && L->P->nch_q <= 3 && !force_generic && L->gen2 < 2);
}

static int backend_init(ppg_handle *h, int) {
    h->lut_dev = (uint32_t *)malloc(h->lut_host.size() * sizeof(uint32_t));
    memcpy(h->lut_dev, h->lut_host.data(), h->lut_host.size() * sizeof(uint32_t));
    if (h->coop_ok) {
        h->coop_tab_dev = (uint32_t *)malloc(h->coop_tab_host.size() * sizeof(uint32_t));
        memcpy(h->coop_tab_dev, h->coop_tab_host.data(), h->coop_tab_host.size() * sizeof(uint32_t));
    }
    return PPG_OK;
}
static int backend_alloc(ppg_handle *, void **out, size_t bytes) {
    *out = malloc(bytes);
    return *out ? PPG_OK : PPG_ENOMEM;
}
static void backend_release(ppg_handle *h) {
    free(h->vis_dev);
    h->vis_dev = nullptr;
    free(h->lut_dev);
    h->lut_dev = nullptr;
    free(h->coop_tab_dev);
    h->coop_tab_dev = nullptr;
    free(h->order_dev);
    h->order_dev = nullptr;
    free(h->fetch_dev);
    h->fetch_dev = nullptr;
    h->fetch_cap = 0;
}
static void backend_free(ppg_handle *, void *p) { free(p); }
static int backend_rebalance(ppg_handle *h, int wp, int wq, void *) {
    if (!h->order_dev) h->order_dev = (int32_t *)malloc((size_t)h->batch * sizeof(int32_t));
    const int32_t *es = h->bufs.env_state;
    for (int i = 0; i < h->batch; ++i) {
        const int ki = es[(size_t)i * PPG_ENV_WORDS + PPG_ENV_N_PRED_ROWS] * wp + es[(size_t)i * PPG_ENV_WORDS + PPG_ENV_N_PREY_ROWS] * wq;
        int rank = 0;
        for (int j = 0; j < h->batch; ++j) {
            const int kj = es[(size_t)j * PPG_ENV_WORDS + PPG_ENV_N_PRED_ROWS] * wp + es[(size_t)j * PPG_ENV_WORDS + PPG_ENV_N_PREY_ROWS] * wq;
            rank += (kj > ki) || (kj == ki && j < i);
        }
        h->order_dev[rank] = i;
    }
    return PPG_OK;
}
static int backend_launch(ppg_handle *h, int mode, const ppg::KParams &P, void *) {
    EmuLaunch L{&P, h->nq, mode, h->drive ? 3 : h->gen2 ? (h->cfg2.walls ? 2 : 1) : 0, 1};
    // The HIP backend picks the multi-wave step kernels from the batch size (ppg_use_multiwave); here the tests ask for
    // them explicitly: PPG_EMU_WAVES=2|4|8 (walls / drive have pair and four-wave kernels, as in the library).
    if ((mode == ppg::MODE_STEP || mode == ppg::MODE_ROLLOUT) && P.coop_e > 0) {   // cooperative kernels (ppg_set_wave_plan): coop_e envs per workgroup of plan.nw waves
        L.nw = h->plan.nw;
        const int groups = (h->batch + P.coop_e - 1) / P.coop_e;
        for (int g = 0; g < groups; ++g) wv::run_block(lane_entry, &L, g, (size_t)P.lds_bytes, L.nw);
        return PPG_OK;
    }
    if (mode == ppg::MODE_STEP) {
        const char *w = getenv("PPG_EMU_WAVES");
        const int nw = w ? atoi(w) : 1;
        if (nw == 4 || (nw == 8 && L.gen2 < 2) || (nw == 2 && L.gen2 != 1)) L.nw = nw;
        else if (nw == 8 || nw == 2) L.nw = 4;
        else if (nw != 1 && nw != 0) return ppg_fail(h, PPG_EINVAL, "PPG_EMU_WAVES=%d (1, 2, 4 or 8)", nw);
    }
    for (int b = 0; b < h->batch; ++b) wv::run_block(lane_entry, &L, b, (size_t)P.lds_bytes, L.nw);
    return PPG_OK;
}

static void pack_scan_entry(void *arg) { ppg::pack_scan_main(*(const ppg::PackParams *)arg, wv::emu().lds); }
static void pack_rows_entry(void *arg) { ppg::pack_rows_main(*(const ppg::PackParams *)arg); }
static int backend_pack(ppg_handle *, const ppg::PackParams &K, void *) {
    wv::run_block(pack_scan_entry, (void *)&K, 0, 512, 1);
    for (int e = 0; e < K.n_envs; ++e) wv::run_block(pack_rows_entry, (void *)&K, e, 16, 1);
    return PPG_OK;
}
static void fetch_entry(void *arg) { ppg::fetch_main(*(const ppg::FetchParams *)arg, wv::emu().lds); }
static int backend_fetch(ppg_handle *, const ppg::FetchParams &K, void *) {
    for (int e = 0; e < K.n_envs; ++e) wv::run_block(fetch_entry, (void *)&K, e, 1024, 1);
    return PPG_OK;
}
static int backend_copy(ppg_handle *, void *dst, const void *src, size_t bytes, bool, void *) {
    memcpy(dst, src, bytes);
    return PPG_OK;
}
static int backend_sync(ppg_handle *, void *) { return PPG_OK; }

extern "C" uint64_t ppg_emu_collectives(void) { return wv::emu().n_collectives; }
// wavefronts per workgroup of the most recent launch (the tests check that PPG_EMU_WAVES really took effect)
extern "C" int ppg_emu_last_waves(void) { return wv::emu().nwaves; }
