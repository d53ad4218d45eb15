// ppg_env_move.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): actions (BASE:96-108), step 1 decay (BASE:244-250), step 2 movement in action order (BASE:259-276,495-509).
    // ---- actions -------------------------------------------------------------------
    // size of this lane's action space: 9 (BASE:108), or range^2 of the agent's type (RQ:974-985)
    PPG_MEMBER int n_actions(int r) const {
        if (!GEN2) return 9;
        const int a = ((id[r] >> 16) & 1) ? C.ar[1] : C.ar[0];
        return a * a;
    }
    // action -> (dx, dy): BASE:96-106 (a//3-1, a%3-1); RQ:141-146 with the range of the agent's type
    PPG_MEMBER void move_vector(int a, bool type2, int &dx, int &dy) const {
        if (!GEN2) {
            const int ax = (a * 11) >> 5;  // a / 3 for 0..8
            dx = ax - 1; dy = a - 3 * ax - 1;
        } else {
            const int side = type2 ? C.ar[1] : C.ar[0];
            const uint32_t inv = type2 ? C.ar_inv[1] : C.ar_inv[0];
            const int ax = (int)(((uint32_t)a * inv) >> 16), delta = (side - 1) >> 1;
            dx = ax - delta; dy = a - ax * side - delta;
        }
    }
    // the value grid[type, pos] shows for this lane's row: its energy, or the birth value (RQ:760)
    PPG_MEMBER double shown(int r) const {
        if (GEN2 && (keep[r] & PPG_ROW_GRID_E0)) return r ? C.e0_q : C.e0_p;
        return e[r];
    }
    PPG_MEMBER bool shown_positive(int r) const {
        if (GEN2) return (float)shown(r) > 0.0f;  // the reference's grid is float32 (RQ:138,339)
        return e[r] > 0.0;
    }

    PPG_MEMBER void load_actions(uint64_t (&acted)[T]) {
        bool bad = false;
        if (C.flags & PPG_STEP_RANDOM_ACTIONS) {
            uint32_t w[4];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if ((r & 3) == 0)
                    philox4x32_10((uint32_t)step, (uint32_t)ln + 64u * (uint32_t)(r >> 2), 0u, episode,
                                  (uint32_t)seed, (uint32_t)(seed >> 32) ^ TAG_ACT, w);
                act[r] = (int32_t)wv::mulhi(w[r & 3], (uint32_t)n_actions(r));
            }
        } else {
#pragma unroll
            for (int r = 0; r < T; ++r) {
                int a = ((alive[r] >> ln) & 1ull) ? act[r] : -1;  // fetched with the rows
                if (a < -1 || a >= n_actions(r)) { bad = true; a = -1; }
                act[r] = a;
            }
        }
        if (wv::ballot(bad)) status |= PPG_STATUS_BAD_ACTION;
#pragma unroll
        for (int r = 0; r < T; ++r) acted[r] = alive[r] & wv::ballot(act[r] >= 0);
#pragma unroll
        for (int r = 0; r < T; ++r) {
            rank[r] = 0;
            if (ORDERED && C.act_rank && ((acted[r] >> ln) & 1ull)) rank[r] = C.act_rank[(size_t)b * P.S + slot_of(r, ln)];
        }
    }

    // Explicit action order (a dict whose order differs from the previous observation dict): rows of
    // `type` that act, as (register, lane) pairs in action order, through the LDS scratch.
    PPG_MEMBER int publish_order(int type, const uint64_t (&acted)[T]) {
        uint16_t *ord = (uint16_t *)scr + (type ? 64 : 0);
        int n = 0;
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r) {
            if (type_of(r) != type) continue;
            if ((acted[r] >> ln) & 1ull) ord[rank[r]] = (uint16_t)row_of(r, ln);
            n += wv::popc(acted[r]);
        }
        wv::sync();
        return n;
    }
    PPG_MEMBER void ordered_row(int type, int i, int &r, int &k) const {
        const uint16_t *ord = (const uint16_t *)scr + (type ? 64 : 0);
        const int row = (int)wv::first((uint32_t)ord[i]);
        r = type ? 1 + (row >> 6) : 0;
        k = row & 63;
    }
    // xy of row (r,k) where r may be a run-time (wave-uniform) register index.  The lane is read from every
    // register first and the scalars are selected afterwards: selecting between the member arrays themselves makes
    // the compiler select between their ADDRESSES, which pins the whole Env object (and the parameters) in scratch.
    // With a compile-time r the unused reads fold away.
    PPG_MEMBER uint32_t xy_at(int r, int k) const {
        uint32_t v = wv::readlane(xy[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane(xy[q], k); v = (q == r) ? vq : v; }
        return v;
    }
    PPG_MEMBER int act_at(int r, int k) const {
        uint32_t v = wv::readlane((uint32_t)act[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane((uint32_t)act[q], k); v = (q == r) ? vq : v; }
        return (int)v;
    }
    PPG_MEMBER uint32_t id_at(int r, int k) const {
        uint32_t v = wv::readlane((uint32_t)id[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane((uint32_t)id[q], k); v = (q == r) ? vq : v; }
        return v;
    }
    PPG_MEMBER double e_at(int r, int k) const {
        double v = readlane_f64(e[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const double vq = readlane_f64(e[q], k); v = (q == r) ? vq : v; }
        return v;
    }

    // ---- step 1: decay (BASE:244-250) ----------------------------------------------
    PPG_MEMBER void decay(const uint64_t (&acted)[T]) {
        // Same-type co-occupancy check on the (still all-zero) channel maps used as claim boards.
        bool mism[T];
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) chmap(1 + type_of(r))[cell_of(xy[r])] = to_map(1 + type_of(r), validx(r, ln));
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r)
            mism[r] = ((alive[r] >> ln) & 1ull) && chmap(1 + type_of(r))[cell_of(xy[r])] != to_map(1 + type_of(r), validx(r, ln));
        uint64_t mm[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < T; ++r) mm[type_of(r)] |= wv::ballot(mism[r]);
#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((alive[r] >> ln) & 1ull) chmap(1 + type_of(r))[cell_of(xy[r])] = 0;
        cooc[0] = mm[0] != 0;
        cooc[1] = mm[1] != 0;

#pragma unroll
        for (int r = 0; r < T; ++r)
            if ((acted[r] >> ln) & 1ull) {
                e[r] -= (r ? C.loss_q : C.loss_p);
                if (GEN2) keep[r] &= ~(uint32_t)PPG_ROW_GRID_E0;  // RQ:489: the grid now shows the real energy
            }

        // grid[type, pos] = energy, in action order
#pragma unroll
        for (int type = 0; type < 2; ++type) {
            if (!cooc[type]) {
#pragma unroll
                for (int r = 0; r < T; ++r)
                    if (type_of(r) == type) owns[r] |= acted[r];  // one live agent per cell: each acting agent owns its cell
            } else if (ORDERED && C.act_rank) {
                const int n = publish_order(type, acted);
                for (int i = 0; i < n; ++i) {
                    int r, k;
                    ordered_row(type, i, r, k);
                    grid_set(r, k, xy_at(r, k), 0.0, false);
                }
            } else {
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    if (type_of(r) != type) continue;
                    uint64_t m = acted[r];
                    while (m) {
                        const int k = wv::ctz(m);
                        m &= m - 1;
                        grid_set(r, k, wv::readlane(xy[r], k), 0.0, false);
                    }
                }
            }
        }
    }

    // ---- step 2: movement in action order (BASE:259-276, _get_move BASE:495-509) ------
    // What an acting row wants, computed for all rows at once before anybody moves (nothing another agent does changes it:
    // _get_move reads the agent's own position and action only, BASE:495-505): bits 0-15 the clipped target cell, bits 16-20 the
    // squared displacement (second generation: the move's energy cost, RQ:301-313), bits 24-26 what the walls say (WO:466-488).
    // A target the walls refuse as a WALL cell is the agent's own cell (WO:469-471).
    PPG_MEMBER uint32_t move_wish(int r, bool acts) const {
        const int G1 = P.G - 1;
        int dx = 0, dy = 0;
        if (act[r] >= 0) move_vector(act[r], GEN2 && ((id[r] >> 16) & 1), dx, dy);
        const int x = (int)(xy[r] >> 8), y = (int)(xy[r] & 255u);
        int tx = x + dx, ty = y + dy;
        tx = tx < 0 ? 0 : (tx > G1 ? G1 : tx);  // np.clip, BASE:505
        ty = ty < 0 ? 0 : (ty > G1 ? G1 : ty);
        uint32_t verdict = MV_NONE;
        if (WALLS && acts) {
            verdict = (C.ar[0] <= 5 && C.ar[1] <= 5) ? wall_verdict_near(x, y, tx, ty) : wall_verdict(x, y, tx, ty);
            if (verdict == MV_WALL) { tx = x; ty = y; }
        }
        const int ddx = tx - x, ddy = ty - y;
        return ((uint32_t)tx << 8) | (uint32_t)ty | (((uint32_t)(ddx * ddx + ddy * ddy) & 31u) << 16) | (verdict << 24);
    }
    PPG_MEMBER uint32_t wish_at(const uint32_t (&wish)[T], int r, int k) const {
        uint32_t v = wv::readlane(wish[0], k);
#pragma unroll
        for (int q = 1; q < T; ++q) { const uint32_t vq = wv::readlane(wish[q], k); v = (q == r) ? vq : v; }
        return v;
    }

    // One agent at its turn: row (r,k); r may be a run-time register index (explicit-order path) -- with a
    // compile-time r every (q == r) below folds away.  moved[]: rows that changed cell (their move cost is charged by the caller,
    // lane-parallel); sp[]: rows whose energy after that cost still shows as positive on the grid.
    PPG_MEMBER void move_agent(int r, int k, uint64_t (&pos)[T], const uint32_t (&wish)[T], uint64_t (&moved)[T], const uint64_t (&sp)[T], bool costly) {
        const int type = type_of(r);
        const uint32_t s_xy = xy_at(r, k);
        const uint32_t w = wish_at(wish, r, k);
        const uint32_t t_xy = w & 0xFFFFu;
        const uint32_t verdict = w >> 24;
        uint64_t mt[T], mo[T];
        match(type, t_xy, mt);
        uint64_t occ = 0;  // grid[type, target] > 0 (BASE:506): an owner with positive energy sits there
#pragma unroll
        for (int q = 0; q < T; ++q) occ |= mt[q] & owns[q] & pos[q];
        if (WALLS) {  // WO:466-488: wall, then occupied, then corner cutting / line of sight
            const uint32_t reason = verdict == MV_WALL ? (uint32_t)MV_WALL : (occ ? (uint32_t)MV_OCCUPIED : verdict);
#pragma unroll
            for (int q = 0; q < T; ++q)
                if (q == r && ln == k) set_move_info(q, reason);
            if (verdict == MV_CORNER_CUT || verdict == MV_LOS) occ = 1;  // refused like an occupied target: stay
        }
        if (t_xy == s_xy) {
#pragma unroll
            for (int q = 0; q < T; ++q) mo[q] = mt[q];
        } else if (!cooc[type]) {   // no cell holds two live agents of this type: the agent is alone on its cell
#pragma unroll
            for (int q = 0; q < T; ++q) mo[q] = (q == r) ? bit64(k) : 0ull;
        } else {
            match(type, s_xy, mo);
        }
        // grid[old] = 0 (BASE:268/272)
#pragma unroll
        for (int q = 0; q < T; ++q) owns[q] &= ~mo[q];
        uint64_t others = 0;
        if (occ) {  // stay (BASE:506-507): grid[old] = energy
#pragma unroll
            for (int q = 0; q < T; ++q) others |= mo[q] & ~((q == r) ? bit64(k) : 0ull);
        } else {    // move: grid[new] = energy (BASE:269/273)
#pragma unroll
            for (int q = 0; q < T; ++q) {
                xy[q] = (q == r) ? wv::writelane(xy[q], k, t_xy) : xy[q];
                owns[q] &= ~mt[q];
                others |= mt[q] & ~((q == r) ? bit64(k) : 0ull);
            }
            if (GEN2 && costly && t_xy != s_xy) {
                // _get_movement_energy_cost (RQ:301-313) is paid before the grid write (RQ:526,538): the grid shows the energy after it
#pragma unroll
                for (int q = 0; q < T; ++q)
                    if (q == r) { moved[q] |= bit64(k); pos[q] = (pos[q] & ~bit64(k)) | (sp[q] & bit64(k)); }
            }
        }
#pragma unroll
        for (int q = 0; q < T; ++q) owns[q] |= (q == r) ? bit64(k) : 0ull;
        if (others) cooc[type] = true;
    }

    // one more agent touches the cell (its own, or the target of its move): counts on the -- at this point all-zero -- channel maps
    PPG_MEMBER void touch(int ch, uint32_t s_xy) { wv::lds_count(chmap(ch) + cell_of(s_xy)); }

    PPG_MEMBER void move(const uint64_t (&acted)[T]) {
        uint64_t pos[T];  // grid value > 0 requires the owner's energy > 0 (BASE:506)
        uint32_t wish[T];
        uint64_t moved[T], sp[T];
        const bool costly = GEN2 && C.move_factor != 0.0;
        // distance * factor per squared displacement (RQ:310-312), once per wavefront in the LDS scratch instead of a chain of selects
        // per row register.  BEHIND the explicit-order path's row lists (publish_order: predators at bytes 0..127, prey at
        // 128..128 + 2 * cap_prey <= 640): bytes 640..895 of a scratch that is at least 1024 bytes in every layout (ppg_host.h).
        double *cost = (double *)scr + 80;
        if (costly) {
            if (ln < 32) cost[ln] = ln < 19 ? move_distance(ln) * C.move_factor : 0.0;   // (rows not in use index anything below 32)
            wv::sync();
        }
#pragma unroll
        for (int r = 0; r < T; ++r) {
            pos[r] = wv::ballot(shown_positive(r)) & alive[r];
            wish[r] = move_wish(r, (acted[r] >> ln) & 1ull);
            moved[r] = 0; sp[r] = 0;
            if (costly) sp[r] = wv::ballot((float)(e[r] - cost[(wish[r] >> 16) & 31u] * e[r]) > 0.0f);
        }
        if (ORDERED && C.act_rank) {
#pragma unroll
            for (int type = 0; type < 2; ++type) {
                const int n = publish_order(type, acted);
                for (int i = 0; i < n; ++i) {
                    int r, k;
                    ordered_row(type, i, r, k);
                    move_agent(r, k, pos, wish, moved, sp, costly);
                }
            }
        } else {
            // An agent whose old cell and target cell are touched by no other agent of its type commutes
            // with all others: its move cannot be blocked (an empty target cell holds 0, see the header)
            // and nobody reads or writes its cells.  Those agents move lane-parallel; only agents that
            // share a cell with someone (contested target, target occupied, someone entering my cell)
            // go through the ordered loop.  Who touches a cell is COUNTED on the (all-zero) channel maps,
            // both species at once (they never meet on a channel): every live agent counts on its own cell,
            // every agent that wants to leave on its target; alone = both counts are 1.
#ifdef PPG_PROFILE_MOVE   // (diagnostic build: cycles of the lane-parallel part / of the ordered loop, agents in the ordered loop)
            const unsigned long long mv_t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            bool mover[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                mover[r] = ((acted[r] >> ln) & 1ull) && (wish[r] & 0xFFFFu) != xy[r];
                if ((alive[r] >> ln) & 1ull) touch(1 + type_of(r), xy[r]);
                if (mover[r]) touch(1 + type_of(r), wish[r] & 0xFFFFu);
            }
            wv::sync();
            uint64_t todo[T];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                map_t *A = chmap(1 + type_of(r));
                bool c = false;
                if ((alive[r] >> ln) & 1ull) c = A[cell_of(xy[r])] != 1 || (mover[r] && A[cell_of(wish[r] & 0xFFFFu)] != 1);
                // a move the walls refuse still has to see whether its target is occupied at its turn (the reported
                // reason depends on it, WO:472-488): ordered loop
                if (WALLS && ((wish[r] >> 24) == MV_CORNER_CUT || (wish[r] >> 24) == MV_LOS)) c = true;
                todo[r] = (cooc[type_of(r)] ? ~0ull : wv::ballot(c)) & acted[r];
            }
#pragma unroll
            for (int r = 0; r < T; ++r) {   // (in program order behind the reads above: one wavefront's LDS accesses do not overtake each other)
                map_t *A = chmap(1 + type_of(r));
                if ((alive[r] >> ln) & 1ull) {
                    A[cell_of(xy[r])] = 0;
                    if (mover[r]) A[cell_of(wish[r] & 0xFFFFu)] = 0;
                }
            }
            wv::sync();
#pragma unroll
            for (int r = 0; r < T; ++r) {
                const uint64_t simple = acted[r] & ~todo[r];
                if (WALLS && ((simple >> ln) & 1ull))  // nobody else touches its cells: the target is free, or its own cell
                    set_move_info(r, (wish[r] >> 24) == MV_WALL ? (uint32_t)MV_WALL
                                     : (!mover[r] && shown_positive(r)) ? (uint32_t)MV_OCCUPIED : (uint32_t)MV_NONE);
                if ((simple >> ln) & 1ull) xy[r] = wish[r] & 0xFFFFu;   // BASE:263
                if (costly) moved[r] = simple & wv::ballot(mover[r]);
                owns[r] |= simple;                            // grid[new] = energy, BASE:269/273
            }
#ifdef PPG_PROFILE_MOVE
            const unsigned long long mv_t1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            unsigned long long mv_n = 0, mv_all = 0;
#pragma unroll
            for (int r = 0; r < T; ++r) { mv_n += (unsigned long long)wv::popc(todo[r]); mv_all += (unsigned long long)wv::popc(acted[r]); }
#endif
#pragma unroll
            for (int r = 0; r < T; ++r) {
                uint64_t m = todo[r];
                while (m) {
                    const int k = wv::ctz(m);
                    m &= m - 1;
                    move_agent(r, k, pos, wish, moved, sp, costly);
                }
            }
#ifdef PPG_PROFILE_MOVE
            const unsigned long long mv_t2 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (C.prof && ln == 0) {
                C.prof[(size_t)b * 16 + 13] = mv_t1 - mv_t0; C.prof[(size_t)b * 16 + 14] = mv_t2 - mv_t1; C.prof[(size_t)b * 16 + 15] = (mv_all << 16) | mv_n;
            }
#endif
        }
        if (costly) {   // RQ:301-313,526: distance * factor * energy, for every agent that changed cell
#pragma unroll
            for (int r = 0; r < T; ++r)
                if ((moved[r] >> ln) & 1ull) e[r] = e[r] - cost[(wish[r] >> 16) & 31u] * e[r];
        }
    }

