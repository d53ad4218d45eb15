#ifndef PPG_COOP_QUAD32
#define PPG_COOP_QUAD32 1
#endif
// ppg_env_coop.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): the cooperative kernels' observation writing: whole 1 KB pieces of an env's run of live rows, both env-region layouts.
    // ---- COOP: observations as whole 1 KB pieces of an env's run of live rows ---------------------------------
    // The live rows of `type` of the env whose LDS region is `region` are listed in `list` (n_live words: row << 16 | the agent's
    // padded cell -- ch0_map 0: | x << 8 | y); concatenated they are a run of n_live * blk elements.  Piece p is elements 128 p ..
    // 128 p + 127 of the run: lane l produces elements 128 p + 2l and + 1 (blk is even: a pair never straddles two rows) --
    // BASE:511-526 per element: value = val[map[cell + offset of the element] + section of its channel]; the padded maps make the
    // window clipping of _obs_clip (BASE:528-539) implicit.  ch0_map 0: channel 0 is 1.0 iff the element's cell lies outside the grid
    // (BASE:520-523), from the agent's position and the element's window offsets alone.  This wavefront writes pieces first,
    // first + stride, ...
    // elements per piece: 128 (two per lane); 256 for the second generation's float32 rows on the four-map layout (four per lane)
    static constexpr bool QUAD32 = PPG_COOP_QUAD32 && GEN2 && CH0MAP && COOP && !WALLS;
    PPG_MEMBER int piece_shift() const { return (QUAD32 && P.obs_f32 == 1) ? 8 : 7; }
    // Who writes which piece: statically -- this wavefront writes pieces first, first + stride, ... (tk == nullptr: mid-step
    // observations, the fused kernels) -- or through an LDS ticket word (DYN): a ticket is U consecutive pieces, and the next ticket is
    // issued before the pieces in hand are produced, so its round trip hides behind them.
#ifndef PPG_COOP_TICKET_GROUPS
#define PPG_COOP_TICKET_GROUPS 4
#endif
    // a ticket of the piece writer is TG groups of U consecutive pieces (one LDS round trip per 1-2 K elements: second generation
    // +1.5-2 %, 64x64 grids +1.2 % against one group per ticket, headline equal; the walls variant's chunks of 256 window cells are
    // large enough as they are -- there larger tickets cost 2 % of balance: profiles/r06/y_*)
    static constexpr int TG = PPG_COOP_TICKET_GROUPS;
    struct Turn {
        uint32_t *tk; int stride, us, end, take; uint32_t pend;
    };
    PPG_MEMBER int turn_begin(Turn &t, uint32_t *tk, int first, int stride, int U, int groups = 1) const {
        t.tk = tk; t.stride = stride; t.us = stride; t.pend = 0; t.end = 0; t.take = U * groups;
        if (DYN && tk) {
            t.us = 1;
            const uint32_t mine = wv::lds_take_issue(tk, (uint32_t)t.take);
            t.pend = wv::lds_take_issue(tk, (uint32_t)t.take);
            const int p = (int)wv::lds_take_value(mine);
            t.end = p + t.take;
            return p;
        }
        return first;
    }
    PPG_MEMBER int turn_next(Turn &t, int p0, int U) const {
        if (DYN && t.tk) {
            if (p0 + U < t.end) return p0 + U;
            const int p = (int)wv::lds_take_value(t.pend);
            t.pend = wv::lds_take_issue(t.tk, (uint32_t)t.take);
            t.end = p + t.take;
            return p;
        }
        return p0 + U * t.stride;
    }
    PPG_MEMBER void coop_pieces(int type, const unsigned char *region, const uint32_t *list, int n_live, int eb, int first, int stride,
                                uint32_t *tk = nullptr) {
        const map_t *m = (const map_t *)(region + P.off_map);
        const double *vt = (const double *)(region + P.off_val);
        const int blk = C.blk_p + (type ? C.blk_q - C.blk_p : 0);   // (arithmetic, not a select of fields: see window_sum)
        const uint32_t magic = C.bp_magic + (type ? C.bq_magic - C.bp_magic : 0u);
        const uint32_t *L = lut2 + (type ? C.blk_p : 0);
        const int total = n_live * blk;
        const size_t obase = (size_t)eb * (size_t)(type ? P.cap_prey : P.cap_pred) * (size_t)blk;
        // pieces in flight per wavefront: the three dependent LDS lookups of one hide behind the other's.  (Round 6, measured and not
        // kept: four in flight, and the first lookup of the next group issued beside the map reads of the group in hand -- 53.8 / 55.4
        // against 53.0 us per 4096-env step on 64x64 grids, 63.5 / 67.4 against 62.3 on the headline: the write phase is bound by how
        // fast the memory system takes the stores, not by this chain.  profiles/r06/b_*)
        constexpr int U = 2;
        // The three-map kernels decide the dtype dispatch of the stores (float64 / float32 / bfloat16: wave-uniform branches per piece) ONCE
        // per call -- a loop of its own for float64 rows: 64x64 grids 54.6-55.6 -> 52.7-53.3 us per 4096-env step.  The four-map kernels
        // keep the one generic loop: there the second copy cost the headline 0.5-2 % (code size) -- profiles/r06/o_*.
        void *const obuf = type ? P.obs_prey : P.obs_pred;
        // The second generation's kernels do the same for float32 rows, the reference's dtype there (RQ:137-139): its branch chain is the
        // longest of the three -- 57.5 -> 53.5 us per 4096-env step, +7 % (profiles/r06/r_*).
        auto pieces = [&](auto f64_tag, auto f32_tag) {
        constexpr bool F64 = decltype(f64_tag)::value, F32 = decltype(f32_tag)::value;
        if (QUAD32 && F32) {
            // float32 rows, four maps: FOUR elements per lane -- a 16-byte store per lane like a float64 pair, a piece is 256 elements.
            // The write phase is bound by the store INSTRUCTIONS the memory pipeline takes, not by bytes or lookups (round 6: window cells
            // with 4-byte stores and 40 % fewer LDS reads ran no faster, profiles/r06/v_*): half the instructions for the same rows.
            const uint32_t safe_cell = (uint32_t)(P.pad * P.Gp + P.pad);
            Turn t;
            for (int p0 = turn_begin(t, tk, first, stride, U, TG); p0 * 256 < total; p0 = turn_next(t, p0, U)) {
                uint32_t o[U], ix[U][4];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s0 = (p0 + u * t.us) * 256 + 4 * ln;
                    on[u] = s0 < total;
                    const uint32_t sc = on[u] ? (uint32_t)s0 : 0u;
                    const uint32_t i = wv::mulhi(sc, magic), w = sc - wv::mul24(i, (uint32_t)blk);   // (blk = 4 R^2: a quad never straddles two rows)
                    const uint32_t ent = on[u] ? list[i] : safe_cell;
                    const uint4 d = *(const uint4 *)(L + w);
                    const int pc = (int)(ent & 0xFFFFu);
                    ix[u][0] = (uint32_t)m[pc + (int)(int16_t)(d.x & 0xFFFFu)] + (d.x >> 16);
                    ix[u][1] = (uint32_t)m[pc + (int)(int16_t)(d.y & 0xFFFFu)] + (d.y >> 16);
                    ix[u][2] = (uint32_t)m[pc + (int)(int16_t)(d.z & 0xFFFFu)] + (d.z >> 16);
                    ix[u][3] = (uint32_t)m[pc + (int)(int16_t)(d.w & 0xFFFFu)] + (d.w >> 16);
                    o[u] = wv::mul24(ent >> 16, (uint32_t)blk) + w;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float4 f;
                    f.x = (float)vt[ix[u][0]]; f.y = (float)vt[ix[u][1]]; f.z = (float)vt[ix[u][2]]; f.w = (float)vt[ix[u][3]];
                    if (on[u]) *(float4 *)((float *)obuf + obase + o[u]) = f;
                }
            }
            return;
        }
        if (CH0MAP) {   // four maps: every element is a map lookup
            const uint32_t safe_cell = (uint32_t)(P.pad * P.Gp + P.pad);   // lanes behind the end of the run look at cell (0,0): inside the maps
            Turn t;
            for (int p0 = turn_begin(t, tk, first, stride, U, TG); p0 * 128 < total; p0 = turn_next(t, p0, U)) {
                uint32_t o[U], i0[U], i1[U];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int s0 = (p0 + u * t.us) * 128 + 2 * ln;
                    on[u] = s0 < total;
                    const uint32_t sc = on[u] ? (uint32_t)s0 : 0u;
                    const uint32_t i = wv::mulhi(sc, magic), w = sc - wv::mul24(i, (uint32_t)blk);   // (rows x block elements < 2^24)
                    const uint32_t ent = on[u] ? list[i] : safe_cell;
                    const uint2 d = *(const uint2 *)(L + w);
                    const int pc = (int)(ent & 0xFFFFu);
                    i0[u] = (uint32_t)m[pc + (int)(int16_t)(d.x & 0xFFFFu)] + (d.x >> 16);
                    i1[u] = (uint32_t)m[pc + (int)(int16_t)(d.y & 0xFFFFu)] + (d.y >> 16);
                    o[u] = wv::mul24(ent >> 16, (uint32_t)blk) + w;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double v0 = vt[i0[u]], v1 = vt[i1[u]];
                    if (!on[u]) continue;
                    if (F64) { double2 g; g.x = v0; g.y = v1; *(double2 *)((double *)obuf + obase + o[u]) = g; }
                    else if (F32) { float2 f; f.x = (float)v0; f.y = (float)v1; *(float2 *)((float *)obuf + obase + o[u]) = f; }
                    else store_obs_pair(obuf, P.obs_f32, obase + o[u], v0, v1);
                }
            }
            return;
        }
        const uint32_t G = (uint32_t)P.G;
        Turn t;
        for (int p0 = turn_begin(t, tk, first, stride, U, TG); p0 * 128 < total; p0 = turn_next(t, p0, U)) {
            uint32_t o[U], i0[U], i1[U];
            bool on[U], out0[U], out1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s0 = (p0 + u * t.us) * 128 + 2 * ln;
                on[u] = s0 < total;
                const uint32_t sc = on[u] ? (uint32_t)s0 : 0u;
                const uint32_t i = wv::mulhi(sc, magic), w = sc - wv::mul24(i, (uint32_t)blk);
                const uint32_t ent = on[u] ? list[i] : 0u;   // (lanes behind the end of the run look at cell (0,0): inside the maps)
                const uint2 d = *(const uint2 *)(L + w);
                const uint32_t ax = (ent >> 8) & 255u, ay = ent & 255u;
                const int pc = (int)(wv::mul24(ax + (uint32_t)P.pad, (uint32_t)P.Gp) + ay + (uint32_t)P.pad);
                const bool z0 = (d.x >> 16) == 0xFFFFu, z1 = (d.y >> 16) == 0xFFFFu;   // channel 0: no map
                // (a channel-0 descriptor's low bits as a map offset stay inside the three maps: no lane reads outside LDS)
                const uint32_t m0 = (uint32_t)m[pc + (int)(int16_t)(d.x & 0xFFFFu)], m1 = (uint32_t)m[pc + (int)(int16_t)(d.y & 0xFFFFu)];
                i0[u] = z0 ? 0u : m0 + (d.x >> 16);
                i1[u] = z1 ? 0u : m1 + (d.y >> 16);
                // outside the grid: unsigned compares (a coordinate below 0 wraps far above G); computed for every lane, no branches
                const uint32_t tx0 = ax + ((d.x >> 4) & 15u) - 8u, ty0 = ay + (d.x & 15u) - 8u;
                const uint32_t tx1 = ax + ((d.y >> 4) & 15u) - 8u, ty1 = ay + (d.y & 15u) - 8u;
                out0[u] = z0 & ((tx0 > ty0 ? tx0 : ty0) >= G);
                out1[u] = z1 & ((tx1 > ty1 ? tx1 : ty1) >= G);
                o[u] = wv::mul24(ent >> 16, (uint32_t)blk) + w;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double t0 = vt[i0[u]], t1 = vt[i1[u]];   // (channel 0 inside the grid: entry 0 = 0.0)
                const double v0 = out0[u] ? 1.0 : t0, v1 = out1[u] ? 1.0 : t1;
                if (!on[u]) continue;
                if (F64) { double2 g; g.x = v0; g.y = v1; *(double2 *)((double *)obuf + obase + o[u]) = g; }
                else if (F32) { float2 f; f.x = (float)v0; f.y = (float)v1; *(float2 *)((float *)obuf + obase + o[u]) = f; }
                else store_obs_pair(obuf, P.obs_f32, obase + o[u], v0, v1);
            }
        }
        };
        if (THREE && P.obs_f32 == 0) pieces(TagTrue{}, TagFalse{});
        else if (GEN2 && P.obs_f32 == 1) pieces(TagFalse{}, TagTrue{});
        else pieces(TagFalse{}, TagFalse{});
    }
    // a mid-step observation (an agent that starves or is caught, BASE:287,327): its block alone, at this point of the sequence
    PPG_MEMBER void obs_row_coop(int type, int j, uint32_t s_xy) {
        wv::sync();   // LDS writes of the sequential phases -> visible
        uint32_t *mid = ctl + CTL_MID + wave_idx;
        if (ln == 0) mid[0] = ((uint32_t)j << 16) | (CH0MAP ? (uint32_t)cell_of(s_xy) : s_xy);
        wv::sync();
        coop_pieces(type, (const unsigned char *)map - P.off_map, mid, 1, b, 0, 1);
        wv::sync();   // reads done before the caller touches the maps again
    }
    // the rows to observe at the end of the call, per species, in the env's scratch: [0] predators, [64] prey
    PPG_MEMBER void coop_publish() {
        uint32_t *lst = (uint32_t *)scr;
        int n[2] = {0, 0};
        wv::sync();
#pragma unroll
        for (int r = 0; r < T; ++r) {
            const int type = type_of(r);
            if ((alive[r] >> ln) & 1ull)
                lst[(type ? 64 : 0) + n[type] + (int)wv::prefix(alive[r])] = ((uint32_t)row_of(r, ln) << 16) | (CH0MAP ? (uint32_t)cell_of(xy[r]) : xy[r]);
            n[type] += wv::popc(alive[r]);
        }
        if (ln == 0) {
            uint32_t *slot = ctl + CTL_SLOT + 4 * wave_idx;
            slot[0] = (uint32_t)n[0]; slot[1] = (uint32_t)n[1]; slot[2] = (uint32_t)b;
        }
        if (WALLS) walls_stage_masks();
        wv::sync();
    }
    // DYN, no workgroup barrier: the envs of the workgroup in the order in which their READY bits appear; the pieces (WALLS: chunks of
    // window cells) of a ready env's two runs go to whichever wavefront asks next (tickets CTL_TICKET + 2 k + species).
    PPG_MEMBER void coop_write_dynamic(unsigned char *wg_lds) {
        const uint32_t all = (1u << C.coop_e) - 1u;
        uint32_t done = 0;
        while (done != all) {
            const uint32_t rdy = wv::lds_poll(ctl + CTL_READY) & ~done;
            if (!rdy) { wv::poll_sleep(); continue; }
            const int k = wv::ctz((uint64_t)rdy);
            done |= 1u << k;
            const uint32_t *slot = ctl + CTL_SLOT + 4 * k;
            const int eb = (int)wv::first(slot[2]);
            if (eb < 0) continue;
            unsigned char *region = wg_lds + (size_t)k * C.lds_env_bytes;
            const uint32_t *lst = (const uint32_t *)(region + P.off_scr);
            const bool flat = WALLS && walls_flat();
#pragma unroll
            for (int type = 0; type < 2; ++type) {
                const int n_live = (int)wv::first(slot[type]);
                uint32_t *tk = ctl + CTL_TICKET + 2 * k + type;
                if (WALLS) {
                    const uint32_t *vm = (const uint32_t *)(region + C.off_vm);
                    if (flat) { obs_cells_walls(type, lst + (type ? 64 : 0), n_live, vm + (type ? 64 * C.vis_words : 0), region, eb, 0, 1, tk); continue; }
                    // (no precomputed masks: whole rows, each walks its lines -- a ticket is one row)
                    uint32_t pend = wv::lds_take_issue(tk, 1u);
                    for (int i = (int)wv::lds_take_value(pend); i < n_live; i = (int)wv::lds_take_value(pend)) {
                        pend = wv::lds_take_issue(tk, 1u);
                        const uint32_t en = wv::first(lst[(type ? 64 : 0) + i]);
                        obs_row_walls_in(type, (int)(en >> 16), en & 0xFFFFu, region, eb);
                    }
                    continue;
                }
                coop_pieces(type, region, lst + (type ? 64 : 0), n_live, eb, 0, 1, tk);
            }
        }
    }
    // after the workgroup barrier: all the workgroup's envs, piece p of the workgroup to wavefront p mod NW
    // WALLS (ppgc3_step): whole rows instead of pieces -- row i of the workgroup to wavefront i mod NW, each through obs_row_walls_in
    // (its per-cell work -- wall bit, line-of-sight bit, three lookups -- feeds all 4 / 5 channels of the cell)
    PPG_MEMBER void coop_write_all(unsigned char *wg_lds) {
        int at = 0;   // pieces (WALLS: rows) handed out so far, mod NW
        for (int k = 0; k < C.coop_e; ++k) {
            const uint32_t *slot = ctl + CTL_SLOT + 4 * k;
            const int eb = (int)wv::first(slot[2]);
            if (eb < 0) continue;
            unsigned char *region = wg_lds + (size_t)k * C.lds_env_bytes;
            const uint32_t *lst = (const uint32_t *)(region + P.off_scr);
            if (WALLS) {
                const bool flat = walls_flat();
                const uint32_t *vm = (const uint32_t *)(region + C.off_vm);
#pragma unroll
                for (int type = 0; type < 2; ++type) {
                    const int n_live = (int)wv::first(slot[type]);
                    int first = wave_idx - at;
                    if (first < 0) first += NW;
                    if (flat) {   // runs of window cells, 64 per pass (obs_cells_walls)
                        obs_cells_walls(type, lst + (type ? 64 : 0), n_live, vm + (type ? 64 * C.vis_words : 0), region, eb, first, NW);
                        at = (at + walls_chunks(type, n_live)) % NW;
                        continue;
                    }
                    for (int i = first; i < n_live; i += NW) {   // (no precomputed masks: whole rows, each walks its lines)
                        const uint32_t en = wv::first(lst[(type ? 64 : 0) + i]);
                        obs_row_walls_in(type, (int)(en >> 16), en & 0xFFFFu, region, eb);
                    }
                    at = (at + n_live) % NW;
                }
                continue;
            }
#pragma unroll
            for (int type = 0; type < 2; ++type) {
                const int n_live = (int)wv::first(slot[type]);
                const int blk = C.blk_p + (type ? C.blk_q - C.blk_p : 0);
                const int psh = piece_shift();
                const int pieces = (n_live * blk + (1 << psh) - 1) >> psh;
                int first = wave_idx - at;
                if (first < 0) first += NW;
                coop_pieces(type, region, lst + (type ? 64 : 0), n_live, eb, first, NW);
                at = (at + pieces) % NW;
            }
        }
    }

