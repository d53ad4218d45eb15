// ppg_env_load.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): the load phase: every global load of a step issued up front (Pre / prefetch), env words, rows, LDS initialisation, the cooperative kernels' tables, the grass table.
    // ---- load ----------------------------------------------------------------------
    // Every global load that does not depend on another load is issued first, back to back, so the
    // wave pays ONE memory round trip: env words, seed, the first two row registers (speculatively:
    // rows beyond n_rows are valid memory holding stale data and are masked out), the first 128 grass
    // patches and the observation descriptor table.
    struct Pre {
        uint32_t w_env;
        uint64_t sd;
        uint32_t xy[T], key[T], fl[T];
        int32_t id[T], a[T];
        double e[T];
        double cum[T];
        uint32_t gxy[2];
        double ge[2];
        uint2 lutd[5];
    };

    PPG_MEMBER void prefetch(Pre &p, bool want_rows, bool want_actions) {
        const int32_t *es = C.env_state + (size_t)b * PPG_ENV_WORDS;
        p.w_env = ln < PPG_ENV_WORDS ? (uint32_t)es[ln] : 0u;
        p.sd = C.env_seed[b];
#pragma unroll
        for (int r = 0; r < T; ++r) {
            p.xy[r] = 0xFFFFu; p.key[r] = 0; p.fl[r] = 0; p.id[r] = 0; p.a[r] = -1; p.e[r] = 0.0; p.cum[r] = 0.0;
            if (r < 2 && want_rows) {
                const size_t s = (size_t)b * P.S + slot_of(r, ln);
                p.xy[r] = C.row_xy[s];
                p.e[r] = C.row_e[s];
                if (CARRY_CUM) p.cum[r] = C.row_cum[s];
                p.id[r] = C.row_id[s];
                p.key[r] = C.row_key[s];
                p.fl[r] = C.row_flags[s];
                if (want_actions) p.a[r] = C.actions[s];
            }
        }
        const size_t gb = (size_t)b * C.cap_grass;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pp = ln + 64 * q;
            p.gxy[q] = 0; p.ge[q] = 0.0;
            if (want_rows && pp < C.n_grass) { p.gxy[q] = C.grass_xy[gb + pp]; p.ge[q] = C.grass_e[gb + pp]; }
        }
        if (FASTOBS) {  // this lane's observation descriptors (row-independent), kept in registers
            const uint2 *L2 = (const uint2 *)C.obs_lut;
#pragma unroll
            for (int c = 0; c < 2; ++c) { p.lutd[c].x = 0; p.lutd[c].y = 0; if (c < P.nch_p) p.lutd[c] = L2[c * 64 + ln]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) { p.lutd[2 + c].x = 0; p.lutd[2 + c].y = 0; if (c < P.nch_q) p.lutd[2 + c] = L2[(P.nch_p + c) * 64 + ln]; }
        }
    }

    PPG_MEMBER void load_env_words(const Pre &p) {
        const uint32_t w = p.w_env;
        n_rows[0] = (int)wv::readlane(w, PPG_ENV_N_PRED_ROWS);
        n_rows[1] = (int)wv::readlane(w, PPG_ENV_N_PREY_ROWS);
        next_id[0] = (int)wv::readlane(w, PPG_ENV_NEXT_PRED_ID);
        next_id[1] = (int)wv::readlane(w, PPG_ENV_NEXT_PREY_ID);
        step = (int)wv::readlane(w, PPG_ENV_STEP);
        envflags = wv::readlane(w, PPG_ENV_FLAGS);
        status = wv::readlane(w, PPG_ENV_STATUS);
        episode = wv::readlane(w, PPG_ENV_EPISODE);
        fb_count = (int)wv::readlane(w, PPG_ENV_FALLBACK_SPAWNS);
        calls = (int)wv::readlane(w, PPG_ENV_CALLS);
        obs_count[0] = (int)wv::readlane(w, PPG_ENV_OBS_PRED);
        obs_count[1] = (int)wv::readlane(w, PPG_ENV_OBS_PREY);
        next_id2[0] = next_id2[1] = 0;
        draws = 0;
        if (GEN2) {
            next_id2[0] = (int)wv::readlane(w, PPG_ENV_NEXT_PRED_ID_T2);
            next_id2[1] = (int)wv::readlane(w, PPG_ENV_NEXT_PREY_ID_T2);
        }
        seed = ((uint64_t)wv::first((uint32_t)(p.sd >> 32)) << 32) | wv::first((uint32_t)p.sd);
    }

    PPG_MEMBER void load_rows(const Pre &p) {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            const bool valid = i < n_rows[type_of(r)];
            uint32_t fl = 0;
            xy[r] = 0xFFFFu; id[r] = 0; key[r] = 0; e[r] = 0.0; act[r] = -1; ev[r] = 0; cum[r] = 0.0;
            if (valid) {
                if (r < 2) {
                    xy[r] = p.xy[r]; e[r] = p.e[r]; id[r] = p.id[r]; key[r] = p.key[r];
                    fl = p.fl[r]; act[r] = p.a[r];
                    if (CARRY_CUM) cum[r] = p.cum[r];
                } else {  // rows 64.. of the prey table: rarely in use, loaded on demand
                    const size_t s = (size_t)b * P.S + slot_of(r, ln);
                    xy[r] = C.row_xy[s];
                    e[r] = C.row_e[s];
                    if (CARRY_CUM) cum[r] = C.row_cum[s];
                    id[r] = C.row_id[s];
                    key[r] = C.row_key[s];
                    fl = C.row_flags[s];
                    if (C.actions) act[r] = C.actions[s];
                }
            }
            keep[r] = (fl & (PPG_ROW_ATE | (GEN2 ? PPG_ROW_GRID_E0 : 0u))) | ((uint32_t)slot_of(r, ln) << 8);  // bits 8..: where this row's start-of-step energy lives
            lr[r] = 0;
            t2m[r] = GEN2 ? (wv::ballot(valid && ((id[r] >> 16) & 1)) ) : 0ull;
            rows[r] = wv::ballot(valid);
            alive[r] = rows[r] & ~wv::ballot(valid && (fl & PPG_ROW_DIED));
            owns[r] = wv::ballot(valid && (fl & PPG_ROW_OWNS)) & alive[r];
        }
        n_alive[0] = n_alive[1] = 0;
#pragma unroll
        for (int r = 0; r < T; ++r) n_alive[type_of(r)] += wv::popc(alive[r]);
    }

    // maps -> all zero, observation descriptors -> LDS
    PPG_MEMBER void init_lds(const Pre &p) {
        if (!COOP) init_maps();   // (COOP: coop_tab_store)
        if (COOP) {
            if (CH0MAP && ln == 0) val[ONE_IDX] = 1.0;
        } else if (FASTOBS) {
#pragma unroll
            for (int c = 0; c < 5; ++c) { lutr[2 * c] = p.lutd[c].x; lutr[2 * c + 1] = p.lutd[c].y; }
        } else {
            for (int i = ln; i < (P.nch_p + P.nch_q) * 128; i += 64) lut[i] = C.obs_lut[i];
        }
        if (ln == 0) val[0] = 0.0;
        if (MAP8 && ln < 2) val[ln ? SEC_G : SEC_Q] = 0.0;   // the zero entries leading the prey and grass sections
        gxyr[0] = p.gxy[0];
        gxyr[1] = p.gxy[1];
        if (WALLS)
            for (int i = ln; i < C.n_wall_words; i += 64) wallw[i] = C.wall_bits[(size_t)b * C.n_wall_words + i];
    }

    // all cell maps empty.  COOP with a channel-0 map: plus that map's halo -> the constant 1.0 of the value table ("outside the grid",
    // BASE:520-523); the halos of channels 1-3 stay 0 -> the zero entry of their section.
    static constexpr int ONE_IDX = 65;   // a free entry of the predator section (rows use 1..64)
    PPG_MEMBER void zero_maps(int first_word) {   // words first_word.. of the map area
        uint32_t *m32 = (uint32_t *)map;
        const int n32 = N_MAPS * P.map_n * (int)sizeof(map_t) / 4;
        // (16-byte stores where the range allows: 64x64 grids zero 14.7 KB per step)
        const int lo16 = (first_word + 3) >> 2, n128 = n32 >> 2;
        uint4 *m128 = (uint4 *)map;
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (int i = first_word + ln; i < 4 * lo16 && i < n32; i += 64) m32[i] = 0u;
        for (int i = lo16 + ln; i < n128; i += 64) m128[i] = zero;
        for (int i = (4 * n128 > first_word ? 4 * n128 : first_word) + ln; i < n32; i += 64) m32[i] = 0u;
    }
    PPG_MEMBER void init_maps() {
        if (COOP && CH0MAP) {   // (channel 0 from the template behind the descriptors in C.coop_tab; a step has it prefetched: TabPre)
            const uint32_t *tmpl = C.coop_tab + C.blk_p + C.blk_q;
            const int n0 = P.map_n / 4;
            uint32_t *m32 = (uint32_t *)map;
            for (int i = ln; i < n0; i += 64) m32[i] = tmpl[i];
            zero_maps(n0);
        } else {
            zero_maps(0);
        }
    }
    // COOP: the workgroup's descriptor table and (ch0_map) this env's channel-0 map come from C.coop_tab.  Their loads are issued in
    // front of everything else and held in registers (up to LUT_REGS / TMPL_REGS words per lane, enough for 7x7 / 9x9 windows on a
    // 25x25 grid; larger geometries finish with plain copy loops), so the tables cost no memory round trip of their own.
    static constexpr int LUT_REGS = 9, TMPL_REGS = 5;
    struct TabPre { uint32_t l[LUT_REGS], m[TMPL_REGS]; };
    PPG_MEMBER void coop_tab_issue(TabPre &t) const {   // (WALLS: no descriptors -- its rows go through obs_row_walls_in)
        const int nl = WALLS ? 0 : C.blk_p + C.blk_q, nm = CH0MAP ? P.map_n / 4 : 0;
#pragma unroll
        for (int u = 0; u < LUT_REGS; ++u) { t.l[u] = 0; if (u * 64 + ln < nl) t.l[u] = C.coop_tab[u * 64 + ln]; }
#pragma unroll
        for (int u = 0; u < TMPL_REGS; ++u) { t.m[u] = 0; if (u * 64 + ln < nm) t.m[u] = C.coop_tab[nl + u * 64 + ln]; }
    }
    PPG_MEMBER void coop_tab_store(const TabPre &t) {
        const int nl = WALLS ? 0 : C.blk_p + C.blk_q, nm = CH0MAP ? P.map_n / 4 : 0;
        uint32_t *m32 = (uint32_t *)map;
        zero_maps(nm);   // channels 1-3: empty
#pragma unroll
        for (int u = 0; u < LUT_REGS; ++u) if (u * 64 + ln < nl) lut2[u * 64 + ln] = t.l[u];
#pragma unroll
        for (int u = 0; u < TMPL_REGS; ++u) if (u * 64 + ln < nm) m32[u * 64 + ln] = t.m[u];
        for (int i = LUT_REGS * 64 + ln; i < nl; i += 64) lut2[i] = C.coop_tab[i];
        for (int i = TMPL_REGS * 64 + ln; i < nm; i += 64) m32[i] = C.coop_tab[nl + i];
    }

    // grass table -> LDS (value table + channel-3 map).  regrow: BASE:252-256.
    PPG_MEMBER void load_grass(bool regrow, const Pre &p) {
        const size_t gb = (size_t)b * C.cap_grass;
        // seasonal variant: square wave on current_step (base_environment_seasonal/...:224-234,268)
        double gain = C.gain_g;
        if (C.season_len > 0) gain = C.gain_g * (((step / C.season_len) & 1) ? C.season_lo : C.season_hi);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pp = ln + 64 * q;
            if (pp < C.n_grass) {
                double g = p.ge[q];
                if (regrow) {
                    double v = g + gain;
                    g = (C.cap_g < v) ? C.cap_g : v;  // Python min(v, cap)
                }
                val[grass_validx(pp)] = g;
                chmap(3)[cell_of(gxyr[q])] = to_map(3, grass_validx(pp));
            }
        }
        for (int pp = 128 + ln; pp < C.n_grass; pp += 64) {
            double g = C.grass_e[gb + pp];
            if (regrow) {
                double v = g + gain;
                g = (C.cap_g < v) ? C.cap_g : v;
            }
            val[grass_validx(pp)] = g;
            chmap(3)[cell_of(C.grass_xy[gb + pp])] = to_map(3, grass_validx(pp));
        }
    }

