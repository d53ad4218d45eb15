// ppg_env_step.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): rewards + table stores (BASE:288-411), the device reset (BASE:129-217), the transition's phase order (step_body) and the kernel modes' entry points.
    // ---- rewards, cumulative rewards (BASE:288,322-323,328-329,341-344,365-366,375-378,408-411) ----
    // COOP: the table stores are issued by coop_main (finish_stores()), not here.  Under a saturated store pipe the ~25 store
    // instructions of the tables take 9 k cycles to issue; in front of a workgroup barrier that was 9 k cycles in which the helper
    // waves could not start writing (66.4 -> 65.1 us per 4096-env step when they moved behind the writing).  Without the barrier
    // (Env::DYN) they go right behind the READY bit; the fused rollout kernels keep them behind the writing.
    bool pend = false, pend_grass = false, pend_transition = false, pend_done = false;
    PPG_MEMBER void finish_stores() {
        if (pend) { pend = false; rewards_and_store(pend_grass, pend_transition); }
    }
    PPG_MEMBER void rewards_and_store(bool write_grass, bool transition = true) {
        if (COOP && !pend_done) { pend = true; pend_done = true; pend_grass = write_grass; pend_transition = transition; return; }
        int n_new[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < T; ++r) n_new[type_of(r)] += wv::popc(wv::ballot(ev[r] & EV_BORN) & rows[r]);
        double rew_[T], cum_[T];
        uint32_t fl_[T];
        int32_t par_[T];
        const bool dense = !GEN2 && transition && C.reward_mode != 0;
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            const uint32_t v = ev[r];
            double rew = 0.0, c = 0.0;
            uint32_t fl = 0;
            if (i < n_rows[type_of(r)]) {
                // cumulative_rewards of a surviving agent still sits in HBM at the row's start-of-step slot
                // (keep[] bits 8..): read it here instead of carrying two registers per row through the step
                if (transition && !(v & EV_BORN)) c = CARRY_CUM ? cum[r] : C.row_cum[(size_t)b * P.S + (keep[r] >> 8)];
                if (!transition) {
                    rew = 0.0;  // reset returns observations only; cumulative_rewards = 0 (BASE:150)
                } else if (v & EV_BORN) {
                    c = 0.0;    // BASE:410
                } else if (v & EV_TRUNC) {
                    rew = 0.0;
                } else if (GEN2) {
                    // type-specific rewards (_get_type_specific, RQ:1099-1106); cumulative_rewards is credited where the
                    // reference credits it (RQ:596,618,643,663,691,721,769)
                    const bool t2 = (id[r] >> 16) & 1;
                    if (v & EV_STARVED) {
                        rew = 0.0;                                                       // RQ:554
                    } else if (v & EV_CAUGHT) {
                        if (v & EV_TURN) {  // it had its own turn before a predator of a later class caught it
                            const double x = (v & EV_ATE) ? (t2 ? C.r2_eat[1] : C.r2_eat[0]) : (t2 ? C.r2_qstep[1] : C.r2_qstep[0]);
                            c += x;
                            if (v & EV_ATE) c += x;
                        }
                        rew = t2 ? C.r2_caught[1] : C.r2_caught[0]; c += rew;           // RQ:616-618
                    } else {
                        if (v & EV_ATE) {
                            rew = r ? (t2 ? C.r2_eat[1] : C.r2_eat[0]) : (t2 ? C.r2_catch[1] : C.r2_catch[0]);
                            c += rew; c += rew;                                          // RQ:596+643 / 663+691
                        } else {
                            rew = r ? (t2 ? C.r2_qstep[1] : C.r2_qstep[0]) : (t2 ? C.r2_pstep[1] : C.r2_pstep[0]);
                            c += rew;                                                    // RQ:643 / 691
                        }
                        if (v & EV_PARENT) {                                             // RQ:719-721 / 767-769 overwrite
                            rew = r ? (t2 ? C.r2_repro_q[1] : C.r2_repro_q[0]) : (t2 ? C.r2_repro_p[1] : C.r2_repro_p[0]);
                            c += rew;
                        }
                    }
                } else if (dense) {
                    // dense variants: reward = energy now - energy at the start of the step (still in HBM at the
                    // row's old slot); a caught prey's account goes to zero (0.0 - before)
                    const double before = C.row_e[(size_t)b * P.S + (keep[r] >> 8)];
                    rew = (v & EV_CAUGHT) ? (0.0 - before) : (e[r] - before);
                    if (C.reward_mode == 2 && !(v & (EV_STARVED | EV_CAUGHT)))
                        rew = rew + ((v & EV_PARENT) ? (r ? C.r_repro_q : C.r_repro_p) : 0.0);
                    c += rew;
                } else if (v & EV_STARVED) {
                    rew = 0.0;
                } else if (v & EV_CAUGHT) {
                    rew = C.r_caught; c += rew;
                } else {
                    if (v & EV_ATE) { rew = r ? C.r_eat : C.r_catch; c += rew; c += rew; }
                    else { rew = r ? C.r_qstep : C.r_pstep; c += rew; }
                    if (KICK) {
                        const double kb = r ? C.kick_q : C.kick_p;
                        const uint32_t n_before = (v >> 8) & 15u, n_after = (v >> 12) & 15u;
                        for (uint32_t i = 0; i < n_before; ++i) { rew = rew + kb; c = c + kb; }   // KICK:446-447
                        if (v & EV_PARENT) { rew = r ? C.r_repro_q : C.r_repro_p; c += rew; }       // BASE:409 overwrites
                        for (uint32_t i = 0; i < n_after; ++i) { rew = rew + kb; c = c + kb; }
                    } else if (v & EV_PARENT) {
                        rew = r ? C.r_repro_q : C.r_repro_p; c += rew;
                    }
                }
                if (v & (EV_STARVED | EV_CAUGHT)) fl |= PPG_ROW_DIED;
                if ((owns[r] >> ln) & 1ull) fl |= PPG_ROW_OWNS;
                if (v & EV_BORN) fl |= PPG_ROW_NEWBORN;
                if (v & EV_ATE) fl |= PPG_ROW_ATE;
                if (v & EV_TRUNC) fl |= PPG_ROW_TRUNC | (GEN2 ? 0u : (keep[r] & PPG_ROW_ATE));  // RQ:200 clears agents_just_ate first
                if (GEN2) {
                    fl |= keep[r] & PPG_ROW_GRID_E0;
                    // agent_last_reproduction: -cooldown at registration (RQ:999), current_step at a birth (RQ:737;
                    // `step` has already been advanced when this runs)
                    if (!transition || (v & EV_BORN)) lr[r] = -C.cooldown;
                    else if (v & EV_REPRO) lr[r] = step - 1;
                }
            }
            rew_[r] = rew; cum_[r] = c; fl_[r] = fl;
            par_[r] = -1;
            if (KICK && transition && i < n_rows[type_of(r)] && !(v & (EV_STARVED | EV_CAUGHT))) {
                // agent_parent rides along with its row: newborns got it in reproduce() (LDS), survivors keep theirs
                if (v & EV_BORN) par_[r] = ((const int32_t *)scr)[slot_of(r, ln)];
                else par_[r] = C.row_parent[(size_t)b * P.S + (keep[r] >> 8)];
            }
        }
        // every lane has its start-of-step values before any row is overwritten (CARRY_CUM: nothing was read here unless the dense
        // reward modes looked up the start-of-step energies)
        if (transition && (!CARRY_CUM || dense)) wv::drain_loads();
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            if (i >= n_rows[type_of(r)]) continue;
            const size_t s = (size_t)b * P.S + slot_of(r, ln);
            C.row_xy[s] = (uint16_t)xy[r];
            C.row_e[s] = e[r];
            C.row_id[s] = id[r];
            C.row_key[s] = key[r];
            C.row_cum[s] = cum_[r];
            C.row_flags[s] = (uint8_t)fl_[r];
            C.row_reward[s] = rew_[r];
            if (KICK) C.row_parent[s] = par_[r];
            if (GEN2) C.row_lastrep[s] = lr[r];
            if (WALLS) C.row_info[s] = (uint8_t)((transition && !(ev[r] & (EV_TRUNC | EV_BORN))) ? (keep[r] & 7u) : 0u);
            keep[r] = (keep[r] & ~0xFFu) | (fl_[r] & PPG_ROW_ATE);
        }
        obs_count[0] += n_rows[0];       // every row in use got an observation
        obs_count[1] += n_rows[1];
        if (write_grass) {
            const size_t gb = (size_t)b * C.cap_grass;
            for (int p = ln; p < C.n_grass; p += 64) C.grass_e[gb + p] = val[grass_validx(p)];
        }
        int32_t *es = C.env_state + (size_t)b * PPG_ENV_WORDS;
        if (ln < PPG_ENV_WORDS) {
            int32_t w = 0;
            switch (ln) {
                case PPG_ENV_N_PRED_ROWS: w = n_rows[0]; break;
                case PPG_ENV_N_PREY_ROWS: w = n_rows[1]; break;
                case PPG_ENV_N_PRED_NEW: w = n_new[0]; break;
                case PPG_ENV_N_PREY_NEW: w = n_new[1]; break;
                case PPG_ENV_NEXT_PRED_ID: w = next_id[0]; break;
                case PPG_ENV_NEXT_PREY_ID: w = next_id[1]; break;
                case PPG_ENV_STEP: w = step; break;
                case PPG_ENV_N_PRED_ALIVE: w = n_alive[0]; break;
                case PPG_ENV_N_PREY_ALIVE: w = n_alive[1]; break;
                case PPG_ENV_FLAGS: w = (int32_t)envflags; break;
                case PPG_ENV_STATUS: w = (int32_t)status; break;
                case PPG_ENV_EPISODE: w = (int32_t)episode; break;
                case PPG_ENV_FALLBACK_SPAWNS: w = fb_count; break;
                case PPG_ENV_CALLS: w = calls; break;
                case PPG_ENV_OBS_PRED: w = obs_count[0]; break;
                case PPG_ENV_OBS_PREY: w = obs_count[1]; break;
                case PPG_ENV_NEXT_PRED_ID_T2: w = next_id2[0]; break;
                case PPG_ENV_NEXT_PREY_ID_T2: w = next_id2[1]; break;
                case PPG_ENV_DRAWS: w = draws; break;
                default: w = 0; break;
            }
            es[ln] = w;
        }
    }

    // ---- reset (BASE:129-217) with Philox Fisher-Yates placement ---------------------
    PPG_MEMBER void do_reset(uint32_t new_episode) {
        episode = new_episode;
        const int n = P.G * P.G;
        const int K = C.n_init_pred + C.n_init_prey + C.n_grass;
        // two arrays of 16-bit cell indices over the map area: the G*G cells, and the K placed entities (MAP8: the four 8-bit maps
        // together hold two arrays of map_n >= G*G entries; three maps hold G*G + K entries -- ppg_coop_layout admits only
        // configurations where they do)
        uint16_t *perm = MAP8 ? (uint16_t *)map : (uint16_t *)chmap(1);
        uint16_t *ent = THREE ? (uint16_t *)map + ((n + 7) & ~7) : MAP8 ? (uint16_t *)map + P.map_n : (uint16_t *)chmap(2);
        uint32_t *rnd = (uint32_t *)scr;  // 256 words per round
        wv::sync();
        int n_free = n;
        if (!WALLS) {
            for (int i = ln; i < n; i += 64) perm[i] = (uint16_t)i;
        } else {  // the cells that are not walls, in cell-index order (the walls stay; build contract, see oracle/rq_oracle.c)
            n_free = 0;
            for (int base = 0; base < n; base += 64) {
                const int c = base + ln;
                const bool fr = c < n && !((wallw[c >> 5] >> (c & 31)) & 1u);
                const uint64_t m = wv::ballot(fr);
                if (fr) perm[n_free + (int)wv::prefix(m)] = (uint16_t)c;
                n_free += wv::popc(m);
            }
        }
        // more entities than free cells (walls set after create; ppg_create rejects it for the open grid, BASE:167-168): flagged,
        // and only the entities that fit are placed -- the Fisher-Yates below must never index past the free cells
        const int Kp = K < n_free ? K : n_free;
        if (K > n_free) status |= PPG_STATUS_FAILED_SPAWN;
        for (int i = Kp + ln; i < K; i += 64) ent[i] = perm[0];  // (defined, never meaningful: the status bit is set)
        for (int base = 0; base < Kp; base += 256) {
            uint32_t w[4];
            philox4x32_10((uint32_t)(base >> 2) + (uint32_t)ln, 0u, 0u, episode, (uint32_t)seed,
                          (uint32_t)(seed >> 32) ^ TAG_RST, w);
            wv::sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) rnd[4 * ln + q] = w[q];
            wv::sync();
            const int hi = (Kp - base) < 256 ? (Kp - base) : 256;
            for (int kk = 0; kk < hi; ++kk) {
                const int k = base + kk;
                const uint32_t rr = wv::first(rnd[kk]);
                const int j = k + (int)wv::mulhi(rr, (uint32_t)(n_free - k));
                const uint32_t a = wv::first(perm[k]);
                const uint32_t bb = wv::first(perm[j]);
                if (ln == 0) { perm[j] = (uint16_t)a; perm[k] = (uint16_t)bb; ent[k] = (uint16_t)bb; }
            }
        }
        wv::sync();
        const int P0 = C.n_init_pred, Q0 = C.n_init_prey;
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int i = row_of(r, ln);
            const int cnt = r ? Q0 : P0;
            const bool valid = i < cnt;
            xy[r] = 0xFFFFu; id[r] = 0; key[r] = 0; e[r] = 0.0; act[r] = -1; ev[r] = 0; keep[r] = 0;
            if (valid) {
                const uint32_t c = ent[(r ? P0 : 0) + i];
                const uint32_t cx = wv::mulhi(c, C.g_magic);
                xy[r] = (cx << 8) | (c - cx * (uint32_t)P.G);
                id[r] = i;
                key[r] = lexkey((uint32_t)i);
                if (GEN2) {  // RQ:125-133: type 1 first, then type 2; creation number = position in self.agents
                    const int n1 = r ? C.ninit2[2] : C.ninit2[0];
                    const int t2 = i >= n1, idx = t2 ? i - n1 : i, seq = r ? P0 + i : i;
                    id[r] = (int32_t)(((uint32_t)seq << 17) | ((uint32_t)t2 << 16) | (uint32_t)idx);
                    key[r] = (t2 ? KEY_TYPE2 : 0u) + lexkey((uint32_t)idx);
                }
                e[r] = r ? C.e0_q : C.e0_p;
            }
            rows[r] = wv::ballot(valid);
            alive[r] = rows[r];
            owns[r] = rows[r];
        }
        const size_t gb = (size_t)b * C.cap_grass;
        for (int p = ln; p < C.n_grass; p += 64) {
            const uint32_t c = ent[P0 + Q0 + p];
            const uint32_t cx = wv::mulhi(c, C.g_magic);
            const uint32_t gxy = (cx << 8) | (c - cx * (uint32_t)P.G);
            C.grass_xy[gb + p] = (uint16_t)gxy;
            C.grass_e[gb + p] = C.e0_g;
            if (p == ln) gxyr[0] = gxy;
            if (p == ln + 64) gxyr[1] = gxy;
        }
        wv::sync();
        if (COOP) init_maps();   // (the arrays lay across the padded maps and their halos)
        else for (int i = ln; i < n; i += 64) { perm[i] = 0; ent[i] = 0; }
        wv::sync();
        for (int p = ln; p < C.n_grass; p += 64) {
            // re-read what this lane just wrote (same lane, same address)
            val[grass_validx(p)] = C.e0_g;
            chmap(3)[cell_of(C.grass_xy[gb + p])] = to_map(3, grass_validx(p));
        }
        n_rows[0] = P0; n_rows[1] = Q0;
        next_id[0] = P0; next_id[1] = Q0;            // BASE:153-154
        if (GEN2) {                                  // RQ:129
            next_id[0] = C.ninit2[0]; next_id2[0] = C.ninit2[1];
            next_id[1] = C.ninit2[2]; next_id2[1] = C.ninit2[3];
#pragma unroll
            for (int r = 0; r < T; ++r) t2m[r] = wv::ballot((id[r] >> 16) & 1) & rows[r];
        }
        n_alive[0] = P0; n_alive[1] = Q0;            // BASE:210-211
        step = 0;                                    // BASE:134
        fb_count = 0;
        envflags = PPG_ENVF_WAS_RESET | PPG_ENVF_LIST_IS_ROW_ORDER;
        build_maps();
        obs_all_alive(false);                        // BASE:215
        rewards_and_store(false, false);
        obs_finish();
    }

    // ---- the transition ----------------------------------------------------------------
    // One transition: the tables were prefetched from HBM into `pre`.  `it` = index into the action tape.
    PPG_MEMBER void step_body(const Pre &pre, int it) {
        calls += 1;
        if ((C.flags & PPG_STEP_AUTO_RESET) && (envflags & PPG_ENVF_DONE)) {
            wv::sync();
            do_reset(episode + 1u);
            return;
        }
        load_rows(pre);
        if (FUSED && it > 0 && C.actions && !(C.flags & PPG_STEP_RANDOM_ACTIONS)) {  // action tape [n_steps,B,S]
#pragma unroll
            for (int r = 0; r < T; ++r)
                if ((alive[r] >> ln) & 1ull) act[r] = C.actions[((size_t)it * P.batch + b) * P.S + slot_of(r, ln)];
        }
        const bool list_is_row_order = (envflags & PPG_ENVF_LIST_IS_ROW_ORDER) != 0;

        if (step >= C.max_steps) {  // truncation, BASE:228-238: no state change
            wv::sync();
            load_grass(false, pre);
            compact_and_sort(!list_is_row_order);
            if (GEN2) after_compact();
            build_maps();
            obs_all_alive(false);
#pragma unroll
            for (int r = 0; r < T; ++r) ev[r] = ((alive[r] >> ln) & 1ull) ? EV_TRUNC : 0u;
            envflags = (envflags & PPG_ENVF_LIST_IS_ROW_ORDER) | PPG_ENVF_TRUNC_ALL | PPG_ENVF_DONE;
            rewards_and_store(false);  // (agents_just_ate is untouched by a truncation call: keep[] rides along in the flags)
            obs_finish();
            return;
        }

        PPG_STAMP(1);
        uint64_t acted[T];
        load_actions(acted);
#pragma unroll
        for (int r = 0; r < T; ++r) keep[r] &= GEN2 ? ~(uint32_t)(PPG_ROW_ATE | 7u) : ~0xFFu;  // agents_just_ate.clear(), BASE:241
        wv::sync();                                // LDS zeros visible
        PPG_STAMP(2);
        decay(acted);                              // BASE:244-250
        PPG_STAMP(3);
        load_grass(true, pre);                     // BASE:252-256
        PPG_STAMP(4);
        move(acted);                               // BASE:259-276
        PPG_STAMP(5);
        compact_and_sort(!list_is_row_order);      // BASE:222-225 + the sort of BASE:468
        PPG_STAMP(6);
        if (GEN2) after_compact();
        build_maps();
        PPG_STAMP(7);
        if (!GEN2 || list_is_row_order) {
            engage_predators();                    // BASE:302-346 (+ starvation BASE:284-301)
            wv::sync();
            PPG_STAMP(8);
            engage_prey();                         // BASE:347-380
            wv::sync();
        } else {
            // RQ:225-233 walks the sorted self.agents: type_1_predator*, type_1_prey*, type_2_predator*, type_2_prey*
            engage_predators(~t2m[0]);
            wv::sync();
            engage_prey(1);
            wv::sync();
            PPG_STAMP(8);
            engage_predators(t2m[0]);
            wv::sync();
            engage_prey(2);
            wv::sync();
        }
        PPG_STAMP(9);
        if (GEN2) reproduce2(list_is_row_order);   // RQ:248-254
        else reproduce();                          // BASE:389-448
        PPG_STAMP(10);
        obs_all_alive(false);                      // BASE:451-453
        PPG_STAMP(11);
        step += 1;                                 // BASE:471
        envflags = 0;
        if (n_alive[0] <= 0 || n_alive[1] <= 0) envflags |= PPG_ENVF_TERM_ALL | PPG_ENVF_DONE;  // BASE:466
        rewards_and_store(true);
        PPG_STAMP(12);
        obs_finish();
    }

    PPG_MEMBER void run_step(int it = 0) {
        PPG_STAMP(0);
        Pre pre;
        TabPre tab;
        if (COOP) coop_tab_issue(tab);
        prefetch(pre, true, C.actions != nullptr && !(C.flags & PPG_STEP_RANDOM_ACTIONS));
        if (COOP) coop_tab_store(tab);   // (waits for the table words only: the row loads behind them stay in flight)
        init_lds(pre);
        load_env_words(pre);
        step_body(pre, it);
    }

    PPG_MEMBER void run_reset() {
        Pre pre;
        prefetch(pre, false, false);
        load_env_words(pre);
        init_lds(pre);
        if (C.seeds) {
            uint64_t sd = C.seeds[b];
            seed = ((uint64_t)wv::first((uint32_t)(sd >> 32)) << 32) | wv::first((uint32_t)sd);
            if (ln == 0) C.env_seed[b] = seed;
        }
        status = 0;
        calls = 0;
        wv::sync();
        do_reset(C.reset_episode);
    }

    PPG_MEMBER void run_observe() {
        Pre pre;
        prefetch(pre, true, false);
        load_env_words(pre);
        init_lds(pre);
        load_rows(pre);
        wv::sync();
        load_grass(false, pre);
        build_maps();
        obs_all_alive();
    }

    // MODE_VIS: the line-of-sight masks of every cell of this env from its wall bitmap (one word = 32 window offsets per item)
    PPG_MEMBER void run_vis() {
        for (int i = ln; i < C.n_wall_words; i += 64) wallw[i] = C.wall_bits[(size_t)b * C.n_wall_words + i];
        wv::sync();
        const int n = P.G * P.G, nw = C.vis_words, wm = C.vis_w, neg = C.vis_neg;
        for (int it = ln; it < n * nw; it += 64) {
            const int cell = it / nw, w = it - cell * nw;
            const int x = (int)wv::mulhi((uint32_t)cell, C.g_magic), y = cell - x * P.G;
            uint32_t word = 0;
            for (int k = 0; k < 32; ++k) {
                const int bi = 32 * w + k;
                if (bi >= wm * wm) break;
                const int ci = bi / wm, cj = bi - ci * wm;
                const int gx = x - neg + ci, gy = y - neg + cj;
                if ((unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G && los_clear(x, y, gx, gy)) word |= 1u << k;
            }
            C.vis_masks[((size_t)b * n + cell) * nw + w] = word;
        }
    }

    PPG_MEMBER void run_export_grid() {
        Pre pre;
        prefetch(pre, true, false);
        load_env_words(pre);
        init_lds(pre);
        load_rows(pre);
        wv::sync();
        load_grass(false, pre);
        build_maps();
        const int n = P.G * P.G;
        double *out = C.grid_out + (size_t)b * 4 * n;
        for (int i = ln; i < 4 * n; i += 64) {
            const int ch = i / n, c = i - ch * n;
            out[i] = ch ? val[from_map(ch, chmap(ch)[c])] : ((WALLS && ((wallw[c >> 5] >> (c & 31)) & 1u)) ? 1.0 : 0.0);  // WO:271-273
        }
    }
