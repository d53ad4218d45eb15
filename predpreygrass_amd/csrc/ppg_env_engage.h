// ppg_env_engage.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): step 3: starvation and engagements in self.agents order (BASE:279-380).
    // ---- step 3: engagement in self.agents order (BASE:279-380) ----------------------
    PPG_MEMBER void starve(int r, int k, uint32_t s_xy) {  // BASE:284-301
        const int type = type_of(r);
        obs_row(type, row_of(r, k), s_xy, DRIVE ? e_at(r, k) : 0.0);
        if (ln == k) ev[r] |= EV_STARVED;
        n_alive[type] -= 1;
        grid_zero(type, s_xy, true);
#pragma unroll
        for (int q = 0; q < T; ++q) alive[q] &= ~((q == r) ? bit64(k) : 0ull);
    }

    PPG_MEMBER void engage_predators(uint64_t sel = ~0ull) {
        uint64_t m = alive[0] & sel;
        while (m) {
            const int k = wv::ctz(m);
            m &= m - 1;
            const double s_e = readlane_f64(e[0], k);
            const uint32_t s_xy = wv::readlane(xy[0], k);
            if (s_e <= 0.0) { starve(0, k, s_xy); continue; }
            uint64_t pm[T];
            match(1, s_xy, pm);
            int total = 0;
#pragma unroll
            for (int q = 1; q < T; ++q) total += wv::popc(pm[q]);
            if (total == 0) continue;  // reward_predator_step, BASE:341
            // first prey in agent_positions order == lowest id (ids are handed out in insertion order)
            // (GEN2: the creation number sits in the top bits of row_id, so the same comparison picks the first-inserted prey)
            int cr = 0, ck = 0;
            uint32_t best = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 1; q < T; ++q) {
                uint64_t mq = pm[q];
                while (mq) {
                    const int kk = wv::ctz(mq);
                    mq &= mq - 1;
                    const uint32_t cid = wv::readlane((uint32_t)id[q], kk);
                    if (cid < best) { best = cid; cr = q; ck = kk; }
                }
            }
            double pe = 0.0;
#pragma unroll
            for (int q = 1; q < T; ++q)
                if (q == cr) pe = readlane_f64(e[q], ck);
            double ne = s_e + pe;                           // BASE:324 (E1: pe may be <= 0)
            if (GEN2) {  // RQ:598-606: capped gain times the transfer efficiency, then the predator's energy cap
                const double raw = (C.cap_gain_prey < pe) ? C.cap_gain_prey : pe;
                ne = s_e + raw * C.eff_transfer;
                ne = (C.max_e_pred < ne) ? C.max_e_pred : ne;
                if (ln == k) keep[0] &= ~(uint32_t)PPG_ROW_GRID_E0;
            }
            e[0] = writelane_f64(e[0], k, ne);
            if (ln == k) ev[0] |= EV_ATE;                   // BASE:319
            grid_set(0, k, s_xy, ne, true);                 // BASE:325
            obs_row(1, row_of(cr, ck), s_xy, pe);           // BASE:327 (before the prey is erased)
            n_alive[1] -= 1;
#pragma unroll
            for (int q = 1; q < T; ++q) {
                alive[q] &= ~((q == cr) ? bit64(ck) : 0ull);
                ev[q] |= (q == cr && ln == ck) ? (uint32_t)EV_CAUGHT : 0u;
            }
            grid_zero(1, s_xy, true);                       // BASE:335
        }
    }

    // GEN2: the gain of a prey eating grass energy g (RQ:665-673)
    PPG_MEMBER double prey_after_eating(double s_e, double g) const {
        if (!GEN2) return s_e + g;                          // BASE:367
        const double raw = (C.cap_gain_grass < g) ? C.cap_gain_grass : g;
        const double ne = s_e + raw * C.eff_transfer;
        return (C.max_e_prey < ne) ? C.max_e_prey : ne;
    }

    // sel: 0 = every live prey; 1 / 2 = only type-1 / type-2 prey (GEN2 runs the engagement class by class)
    PPG_MEMBER void engage_prey(int sel = 0) {
        uint32_t pidx[T];
        uint64_t ong[T], stv[T];
        uint64_t anystv = 0;
#pragma unroll
        for (int r = 1; r < T; ++r) {
            const uint64_t mine = alive[r] & (sel == 0 ? ~0ull : (sel == 2 ? t2m[r] : ~t2m[r]));
            const uint32_t gm = ((mine >> ln) & 1ull) ? (uint32_t)chmap(3)[cell_of(xy[r])] : 0u;
            pidx[r] = gm ? (uint32_t)from_map(3, gm) : 0u;   // 0 = not standing on a patch
            ong[r] = wv::ballot(pidx[r] != 0u) & mine;
            stv[r] = wv::ballot(e[r] <= 0.0) & mine;
            anystv |= stv[r];
            if (GEN2 && ((mine >> ln) & 1ull)) ev[r] |= EV_TURN;
        }
        if (!anystv && !cooc[1]) {
            // no mid-step observation needed and one prey per cell: all eaters at once (BASE:359-372)
#pragma unroll
            for (int r = 1; r < T; ++r) {
                if ((ong[r] >> ln) & 1ull) {
                    e[r] = prey_after_eating(e[r], val[pidx[r]]);
                    if (GEN2) keep[r] &= ~(uint32_t)PPG_ROW_GRID_E0;
                    val[pidx[r]] = 0.0;
                    val[validx(r, ln)] = e[r];
                    chmap(2)[cell_of(xy[r])] = to_map(2, validx(r, ln));
                    ev[r] |= EV_ATE;
                }
                owns[r] |= ong[r];
            }
            return;
        }
#pragma unroll
        for (int r = 1; r < T; ++r) {
            uint64_t m = ong[r] | stv[r];
            while (m) {
                const int k = wv::ctz(m);
                m &= m - 1;
                const double s_e = readlane_f64(e[r], k);
                const uint32_t s_xy = wv::readlane(xy[r], k);
                if (s_e <= 0.0) { starve(r, k, s_xy); continue; }
                const uint32_t p = wv::readlane(pidx[r], k);
                wv::sync();
                const double g = first_f64(val[p]);
                const double ne = prey_after_eating(s_e, g);  // BASE:367
                e[r] = writelane_f64(e[r], k, ne);
                if (ln == k) { ev[r] |= EV_ATE; if (GEN2) keep[r] &= ~(uint32_t)PPG_ROW_GRID_E0; }  // BASE:362
                grid_set(r, k, s_xy, ne, true);             // BASE:368
                if (ln == 0) val[p] = 0.0;                  // BASE:371-372
            }
        }
    }

