// ppg_env_reproduce.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): step 5: reproduction and the spawn fallback (BASE:389-448,738-766); second generation: cooldown, chance gate, mutation (RQ:695-866).
    // ---- step 5: reproduction (BASE:389-448, _find_available_spawn_position BASE:738-766) ----
    PPG_MEMBER bool fallback_spawn(int type, int cid, uint32_t &child_xy) {
        // BASE:759-764.  The reference draws from the unseeded global np.random; the build's
        // contract (oracle/ppg_oracle.c:find_spawn) is the k-th free cell in x-major order.
        // the occupancy board: the map of channel 0 (all-zero inside the grid); the cooperative kernels without such a map borrow bit 7
        // of the predator map's entries (8-bit maps, predator entries are <= 65) for the length of this function
        static_assert(!COOP || MAP8, "the cooperative kernels run on 8-bit maps");
        constexpr bool borrow = THREE;
        map_t *occ = borrow ? chmap(1) : chmap(0);
        const uint32_t OCC = borrow ? 0x80u : 1u;
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) {   // (two agents on one cell write the same byte value)
                map_t *at = occ + cell_of(xy[r]);
                *at = (map_t)(borrow ? ((uint32_t)*at | OCC) : OCC);
            }
        wv::sync();
        const int n = P.G * P.G;
        int nfree = 0;
        for (int base = 0; base < n; base += 64) {
            const int c = base + ln;
            nfree += wv::popc(wv::ballot(c < n && ((uint32_t)occ[cell_index(c < n ? c : 0)] & OCC) == 0u));
        }
        bool ok = false;
        if (nfree > 0) {
            uint32_t w[4];
            philox4x32_10((uint32_t)step, (uint32_t)cid, (uint32_t)type, episode, (uint32_t)seed,
                          (uint32_t)(seed >> 32) ^ TAG_SPW, w);
            int kth = (int)wv::mulhi(wv::first(w[0]), (uint32_t)nfree);
            for (int base = 0; base < n; base += 64) {
                const int c = base + ln;
                uint64_t fm = wv::ballot(c < n && ((uint32_t)occ[cell_index(c < n ? c : 0)] & OCC) == 0u);
                const int cnt = wv::popc(fm);
                if (kth < cnt) {
                    for (int s = 0; s < kth; ++s) fm &= fm - 1;
                    const int cellidx = base + wv::ctz(fm);
                    const uint32_t cx = wv::mulhi((uint32_t)cellidx, C.g_magic);
                    child_xy = (cx << 8) | ((uint32_t)cellidx - cx * (uint32_t)P.G);
                    ok = true;
                    break;
                }
                kth -= cnt;
            }
        }
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) {
                map_t *at = occ + cell_of(xy[r]);
                *at = (map_t)(borrow ? ((uint32_t)*at & ~OCC) : 0u);
            }
        wv::sync();
        return ok;
    }

    PPG_MEMBER void reproduce() {
        uint64_t cand[T];
#pragma unroll
        for (int r = 0; r < T; ++r) cand[r] = alive[r] & wv::ballot(e[r] >= (r ? C.thr_q : C.thr_p));
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int type = type_of(r);
            const int npos = type ? C.npos_prey : C.npos_pred;
            const int cap = type ? P.cap_prey : P.cap_pred;
            const double e0 = type ? C.e0_q : C.e0_p;
            uint64_t m = cand[r];
            while (m) {
                const int k = wv::ctz(m);
                m &= m - 1;
                if (next_id[type] >= npos) continue;  // id pool exhausted: no child, no reward (E6)
                if (n_rows[type] >= cap) {
                    status |= type ? PPG_STATUS_PREY_OVERFLOW : PPG_STATUS_PRED_OVERFLOW;
                    continue;
                }
                const uint32_t s_xy = wv::readlane(xy[r], k);
                const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
                uint32_t child_xy = 0;
                bool found = false;
#pragma unroll
                for (int d = 0; d < 4; ++d) {  // (x-1,y),(x+1,y),(x,y-1),(x,y+1), BASE:749
                    const int cx = x + (d == 0 ? -1 : d == 1 ? 1 : 0), cy = y + (d == 2 ? -1 : d == 3 ? 1 : 0);
                    if (found || cx < 0 || cx >= P.G || cy < 0 || cy >= P.G) continue;
                    const uint32_t c_xy = ((uint32_t)cx << 8) | (uint32_t)cy;
                    if (!any_agent_at(c_xy)) { child_xy = c_xy; found = true; }
                }
                const int cid = next_id[type];
                if (!found) {
                    status |= PPG_STATUS_FALLBACK_SPAWN;
                    fb_count += 1;
                    if (!fallback_spawn(type, cid, child_xy)) { status |= PPG_STATUS_FAILED_SPAWN; continue; }
                }
                next_id[type] += 1;                        // BASE:397/426
                const int j = n_rows[type]++;              // appended to self.agents, BASE:398/427
                const int cr = type ? 1 + (j >> 6) : 0, ck = j & 63;
                const uint32_t ckey = lexkey((uint32_t)cid);
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    if (q != cr) continue;
                    xy[q] = wv::writelane(xy[q], ck, child_xy);
                    id[q] = (int32_t)wv::writelane((uint32_t)id[q], ck, (uint32_t)cid);
                    key[q] = wv::writelane(key[q], ck, ckey);
                    e[q] = writelane_f64(e[q], ck, e0);          // BASE:403
                    if (ln == ck) ev[q] = EV_BORN;
                }
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    rows[q] |= (q == cr) ? bit64(ck) : 0ull;
                    alive[q] |= (q == cr) ? bit64(ck) : 0ull;
                }
                n_alive[type] += 1;
                grid_set(cr, ck, child_xy, e0, true);        // BASE:405
                const double ne = readlane_f64(e[r], k) - e0;  // BASE:404
                e[r] = writelane_f64(e[r], k, ne);
                if (ln == k) ev[r] |= EV_PARENT;               // reward overwrite, BASE:409/438 (E4)
                grid_set(r, k, s_xy, ne, true);                // BASE:406
                if (KICK) {
                    // kickback variant: agent_parent[child] = parent (KICK:434); the parent's own parent, if still
                    // alive, gets a bonus (KICK:443-447).  Whether that bonus lands before or after the grandparent's
                    // own reproduction in this loop decides if its reward survives (BASE:409 overwrites), so the two
                    // cases are counted separately (ev bits 8-11 / 12-15) and replayed in rewards_and_store.
                    const int my_id = (int)wv::readlane((uint32_t)id[r], k);
                    if (ln == 0) ((int32_t *)scr)[slot_of(cr, ck)] = my_id;
                    const uint32_t k_keep = wv::readlane(keep[r], k);
                    const int gp = (int)wv::first((uint32_t)C.row_parent[(size_t)b * P.S + (k_keep >> 8)]);
                    if (gp >= 0) {
#pragma unroll
                        for (int q = 0; q < T; ++q) {
                            if (type_of(q) != type) continue;
                            const uint64_t gm = wv::ballot(id[q] == gp) & alive[q] & ~wv::ballot(ev[q] & EV_BORN);
                            if (gm) {
                                const int gk = wv::ctz(gm);
                                const uint32_t gev = wv::readlane(ev[q], gk);
                                const int sh = (gev & EV_PARENT) ? 12 : 8;
                                if (((gev >> sh) & 15u) == 15u) status |= PPG_STATUS_KICK_OVERFLOW;
                                else if (ln == gk) ev[q] += 1u << sh;
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- second generation: reproduction with cooldown, chance gate and mutation (RQ:695-866) ----------
    // self.rng.random() number d of this call: the caller's stream (ppg_step_uniforms) or Philox keyed by (step, d)
    PPG_MEMBER double uniform(int d) {
        if (C.uniforms) {
            if (d >= C.uniforms_per_env) { status |= PPG_STATUS_UNIFORMS_DRY; return 0.0; }
            return first_f64(C.uniforms[(size_t)b * C.uniforms_per_env + d]);
        }
        uint32_t w[4];
        philox4x32_10((uint32_t)step, (uint32_t)d, 0u, episode, (uint32_t)seed, (uint32_t)(seed >> 32) ^ TAG_REP, w);
        return first_f64(((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) * (1.0 / 9007199254740992.0));
    }

    // The parent in row (r,k) of species SP passed the gates with enough energy (RQ:704-778 / 789-866).
    template <int SP>
    PPG_MEMBER void spawn2(int r, int k, bool mutated) {
        const uint32_t pid = id_at(r, k);
        const int pty = (int)((pid >> 16) & 1u), nty = mutated ? (pty ^ 1) : pty;   // RQ:705-712
        const int cur = nty ? next_id2[SP] : next_id[SP];
        if (cur >= C.npos2[SP * 2 + nty]) {  // RQ:715-725: no id left in that pool -- the reward is granted anyway
#pragma unroll
            for (int q = 0; q < T; ++q) ev[q] |= (q == r && ln == k) ? (uint32_t)EV_PARENT : 0u;
            return;
        }
        const int cap = SP ? P.cap_prey : P.cap_pred;
        if (n_rows[SP] >= cap) { status |= SP ? PPG_STATUS_PREY_OVERFLOW : PPG_STATUS_PRED_OVERFLOW; return; }
        const uint32_t s_xy = xy_at(r, k);
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        uint32_t child_xy = 0;
        bool found = false;
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // (x-1,y),(x+1,y),(x,y-1),(x,y+1), RQ:384-394
            const int cx = x + (d == 0 ? -1 : d == 1 ? 1 : 0), cy = y + (d == 2 ? -1 : d == 3 ? 1 : 0);
            if (found || cx < 0 || cx >= P.G || cy < 0 || cy >= P.G) continue;
            const uint32_t c_xy = ((uint32_t)cx << 8) | (uint32_t)cy;
            if (!any_agent_at(c_xy)) { child_xy = c_xy; found = true; }
        }
        if (!found) {
            status |= PPG_STATUS_FALLBACK_SPAWN;
            fb_count += 1;
            if (!fallback_spawn(SP, cur, child_xy)) { status |= PPG_STATUS_FAILED_SPAWN; return; }
        }
        const int seq = next_id[0] + next_id2[0] + next_id[1] + next_id2[1];  // agents created so far this episode
        if (nty) next_id2[SP] += 1; else next_id[SP] += 1;    // RQ:728
        const int j = n_rows[SP]++;                           // appended to self.agents, RQ:729
        const int cr = SP ? 1 + (j >> 6) : 0, ck = j & 63;
        const uint32_t cidw = ((uint32_t)seq << 17) | ((uint32_t)nty << 16) | (uint32_t)cur;
        const uint32_t ckey = (nty ? KEY_TYPE2 : 0u) + lexkey((uint32_t)cur);
        const double e0 = SP ? C.e0_q : C.e0_p;
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (type_of(q) != SP || q != cr) continue;
            xy[q] = wv::writelane(xy[q], ck, child_xy);
            id[q] = (int32_t)wv::writelane((uint32_t)id[q], ck, cidw);
            key[q] = wv::writelane(key[q], ck, ckey);
            e[q] = writelane_f64(e[q], ck, e0 * C.eff_repro);   // RQ:754-756
            if (ln == ck) { ev[q] = EV_BORN; keep[q] = PPG_ROW_GRID_E0; }
        }
#pragma unroll
        for (int q = 0; q < T; ++q) {
            rows[q] |= (q == cr) ? bit64(ck) : 0ull;
            alive[q] |= (q == cr) ? bit64(ck) : 0ull;
            t2m[q] |= (q == cr && nty) ? bit64(ck) : 0ull;
        }
        n_alive[SP] += 1;                                     // RQ:763
        grid_set(cr, ck, child_xy, e0, true);                 // RQ:760: the grid shows the full initial energy
        const double ne = e_at(r, k) - e0;                    // RQ:757
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (type_of(q) != SP) continue;
            e[q] = (q == r) ? writelane_f64(e[q], k, ne) : e[q];
            if (q == r && ln == k) { ev[q] |= EV_PARENT | EV_REPRO; keep[q] &= ~(uint32_t)PPG_ROW_GRID_E0; }  // RQ:737,767
        }
        grid_set(r, k, s_xy, ne, true);                       // RQ:761
    }

    // row_order: self.agents is still in creation order (the call right after reset): predators then prey.  Otherwise
    // it is sorted: type_1_predator*, type_1_prey*, type_2_predator*, type_2_prey* (RQ:270).
    PPG_MEMBER void reproduce2(bool row_order) {
        uint64_t elig[T], cand[T];
#pragma unroll
        for (int r = 0; r < T; ++r) {
            elig[r] = alive[r] & wv::ballot(step - lr[r] >= C.cooldown);                 // RQ:697
            cand[r] = elig[r] & wv::ballot(e[r] >= (r ? C.thr_q : C.thr_p));             // RQ:704/789
        }
        // Every eligible agent draws once (chance gate); only those with enough energy matter afterwards.  Publish
        // the candidates in self.agents order, each with the number of eligible agents in front of it.
        uint32_t *lst = (uint32_t *)scr;
        int n_cand = 0, base = 0;
        wv::sync();
#pragma unroll
        for (int sgi = 0; sgi < 4; ++sgi) {
            if (row_order && sgi >= 2) continue;
            const int species = sgi & 1, ty = sgi >> 1;
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if (type_of(r) != species) continue;
                const uint64_t segm = row_order ? ~0ull : (ty ? t2m[r] : ~t2m[r]);
                const uint64_t el = elig[r] & segm, cm = cand[r] & segm;
                if ((cm >> ln) & 1ull)
                    lst[n_cand + (int)wv::prefix(cm)] = ((uint32_t)(base + (int)wv::prefix(el)) << 16) |
                                                        ((uint32_t)species << 15) | (uint32_t)row_of(r, ln);
                n_cand += wv::popc(cm);
                base += wv::popc(el);
            }
        }
        wv::sync();
        int n_second = 0;
        for (int i = 0; i < n_cand; ++i) {
            const uint32_t w = wv::first(lst[i]);
            const int species = (int)((w >> 15) & 1u), row = (int)(w & 0x7FFFu);
            const int d1 = (int)(w >> 16) + n_second;
            if (uniform(d1) > (species ? C.chance_q : C.chance_p)) continue;             // RQ:701-702
            const double u2 = uniform(d1 + 1);                                          // RQ:708/793
            n_second += 1;
            if (species) spawn2<1>(1 + (row >> 6), row & 63, u2 < C.mut_q);
            else spawn2<0>(0, row & 63, u2 < C.mut_p);
        }
        draws = base + n_second;
    }

