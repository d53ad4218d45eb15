// ppg_env_observe.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): the observation writers of the one-wave and multi-wave kernels: cell maps, _get_observation per row (BASE:511-539), drive channels (DRV:551-616), walls / line of sight (WO:527-601), the shared row lists.
    // ---- LDS acceleration structure for observations --------------------------------
    PPG_MEMBER void build_maps() {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            if ((alive[r] >> ln) & 1ull) {
                val[validx(r, ln)] = shown(r);
                if ((owns[r] >> ln) & 1ull) chmap(1 + type_of(r))[cell_of(xy[r])] = to_map(1 + type_of(r), validx(r, ln));
            }
        }
        wv::sync();
    }

    // _get_observation (BASE:511-526) + _obs_clip (BASE:528-539) for the agent of `type` in
    // per-type row j standing on s_xy; coalesced 16-byte stores of the (4,R,R) block.
    //
    // Lane l of chunk ch produces elements e = 128*ch + 2l and e+1 of the block (C order: channel,
    // i, j).  Everything that depends only on (R, G, e) is precomputed on the host into one LDS word
    // per element:  bits 0-15  moff = c*map_n + (i-off)*G + (j-off)   (signed; map index relative to
    //               the observer's cell),  bits 16-19 (i-off)+8,  bits 20-23 (j-off)+8,  bits 24-25 c,
    //               bit 26 element exists (e < 4*R*R),  bit 27 inside the (2*off+1)^2 window.
    // A row whose window lies inside the grid takes the branch-uniform fast path: value =
    // val[map[moff + cell]], no bounds checks (channel 0 reads the all-zero map 0).
    // FASTOBS version: descriptors in registers, all map reads issued together, then all value reads,
    // then the stores -- two LDS latencies per row.
    template <int TYPE>
    PPG_MEMBER void obs_row_fast(int j, uint32_t s_xy) {
        constexpr int NCH = TYPE ? 3 : 2;
        constexpr int BASE = TYPE ? 4 : 0;
        wv::sync();
        const int R = TYPE ? P.Rq : P.Rp;
        const int blk = 4 * R * R;
        const int off = (R - 1) / 2;
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int s_cell = x * P.G + y;
        const bool interior = (R & 1) && x >= off && y >= off && x + off < P.G && y + off < P.G;
        const size_t obase = ((size_t)b * (TYPE ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)blk;
        uint32_t idx[NCH][2];
        bool one[NCH][2];
        double v[NCH][2];
        if (interior) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = lutr[BASE + 2 * c + h];
                    idx[c][h] = (uint32_t)map[(int)(int16_t)(w & 0xFFFFu) + s_cell] + (MAP8 ? (w >> 30) * 129u : 0u);  // bits 30-31: section
                    one[c][h] = false;
                }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = lutr[BASE + 2 * c + h];
                    const int gx = x + (int)((w >> 16) & 15u) - 8, gy = y + (int)((w >> 20) & 15u) - 8;
                    const bool inb = (w & 0x8000000u) && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                    idx[c][h] = (uint32_t)map[inb ? (int)(int16_t)(w & 0xFFFFu) + s_cell : 0] + ((MAP8 && inb) ? (w >> 30) * 129u : 0u);
                    one[c][h] = !inb && (w & 0x3000000u) == 0u;  // channel 0 outside the grid (BASE:522-523)
                }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[c][h] = val[idx[c][h]];
#ifdef PPG_EXP_NO_OBS_READS  // ablation build only (tools/exp_variants.py): stores without LDS lookups
#pragma unroll
        for (int c = 0; c < NCH; ++c) { v[c][0] = 0.0; v[c][1] = 0.0; }
#endif
#ifdef PPG_EXP_NO_OBS_STORES  // ablation build only: no observation stores at all
        if (P.batch > 0) { wv::sync(); return; }
#endif
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (lutr[BASE + 2 * c] & 0x4000000u) {
                const double v0 = one[c][0] ? 1.0 : v[c][0], v1 = one[c][1] ? 1.0 : v[c][1];
                const size_t o = obase + (size_t)c * 128 + 2 * (size_t)ln;
                store_obs_pair(TYPE ? P.obs_prey : P.obs_pred, P.obs_f32, o, v0, v1);
            }
        }
        wv::sync();
    }

    // ---- drive channels (DRV:551-616) ------------------------------------------------------------
    // np.sum over the n staged float64 values win[lo .. lo+n): numpy's pairwise summation (plain loop below 8 elements,
    // eight interleaved accumulators up to 128, two halves above) -- the order of the additions is part of the result.
    PPG_MEMBER double np_sum_block(const double *win, int lo, int n) const {
        if (n < 8) {
            double res = 0.0;
            for (int i = 0; i < n; ++i) res += first_f64(win[lo + i]);
            return res;
        }
        const int n8 = n - (n & 7);
        double acc = 0.0;
        if (ln < 8) {
            acc = win[lo + ln];
            for (int i = 8 + ln; i < n8; i += 8) acc += win[lo + i];
        }
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = readlane_f64(acc, q);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = n8; i < n; ++i) res += first_f64(win[lo + i]);
        return res;
    }
    // np.sum(observation[ch]) for the agent of `type` standing on s_xy (DRV:601-608)
    PPG_MEMBER double window_sum(int type, int ch, uint32_t s_xy) {
        // (not `type ? P.Rq : P.Rp`: with a run-time type hipcc selects the fields' ADDRESSES and spills both to scratch)
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0), n = R * R;
#ifdef PPG_EXP_DRIVE_NO_SUM  // ablation build only: what the window sums cost altogether
        if (P.batch > 0) return 0.0;
#endif
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int s_cell = x * P.G + y;
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;   // one staging area per wave of a multi-wave workgroup
        double *win = (double *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * 4 * rmax * rmax;
        const uint32_t *L = lut + (type ? P.nch_p * 128 : 0);
        const bool strided = (((4 + (type ? C.n_drive[1] : C.n_drive[0])) * n) & 1) != 0;   // see obs_row / ppg_build_lut
        wv::sync();
        for (int i = ln; i < n; i += 64) {
            const int el = ch * n + i, w7 = el & 127;
            const uint32_t w = L[strided ? (el & ~127) + (w7 & 63) * 2 + (w7 >> 6) : el];
            const int gx = x + (int)((w >> 16) & 15u) - 8, gy = y + (int)((w >> 20) & 15u) - 8;
            const bool inb = (w & 0x8000000u) && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
            win[i] = val[from_map(ch, map[inb ? (int)(int16_t)(w & 0xFFFFu) + s_cell : 0])];
        }
        wv::sync();
        double res;
#ifdef PPG_EXP_DRIVE_NO_REDUCE  // ablation build only: staging without the ordered reduction
        if (P.batch > 0) { res = first_f64(win[0]); wv::sync(); return res; }
#endif
        if (n <= 128) {
            res = np_sum_block(win, 0, n);
        } else {
            int n2 = n / 2;
            n2 -= n2 & 7;
            const double a = np_sum_block(win, 0, n2);
            res = a + np_sum_block(win, n2, n - n2);
        }
        wv::sync();
        return res;
    }
    // _safe_clip01 (DRV:612-615)
    static PPG_MEMBER double safe_clip01(double v) {
        if (!(v - v == 0.0)) return 0.0;
        return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    }
    // the drive features of one agent (DRV:577-610); s_e = its energy at this moment
    PPG_MEMBER void drive_features(int type, double s_e, uint32_t s_xy, double (&dv)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dv[k] = 0.0;
            if (k >= (type ? C.n_drive[1] : C.n_drive[0])) continue;
            const int kind = type ? C.drive_kind[1][k] : C.drive_kind[0][k];
            double v;
            if (kind == 0) v = 1.0 - s_e / (type ? C.hunger_safe[1] : C.hunger_safe[0]);
            else if (kind == 1) v = s_e / (type ? C.thr_q : C.thr_p);
            else if (kind == 2) v = window_sum(type, 2, s_xy) / C.norm_prey_opp;
            else if (kind == 3) v = window_sum(type, 1, s_xy) / C.norm_pred_danger;
            else v = window_sum(type, 3, s_xy) / C.norm_grass_opp;
            dv[k] = safe_clip01(v);
        }
    }

    // _get_observation of the walls env (WO:527-601), one window CELL per lane (64 cells per pass): in-grid test, wall bit,
    // line-of-sight bit and the three channel lookups are done once per cell and feed all 4 / 5 channels -- the per-element
    // formulation below does that work once per channel.  Channel 0 = walls inside the window (0 outside the grid); channels
    // 1-3 optionally multiplied, in float32 like the reference, by the mask; optional last channel = the mask itself, which is
    // computed for every cell of the R x R array that maps into the grid (also the last row / column of an even R, which the
    // window copy WO:543 leaves untouched).  Consecutive lanes write consecutive elements of a channel plane.
    // COOP (ppgc3_step: the cooperative walls kernel): `region` = the LDS region of the env the row belongs to -- after the workgroup's
    // barrier every wavefront writes rows of all the workgroup's envs -- and eb its index; every wavefront stages the mask in ITS area
    // of that region.
    PPG_MEMBER void obs_row_walls(int type, int j, uint32_t s_xy) { obs_row_walls_in(type, j, s_xy, (unsigned char *)map - P.off_map, b); }
    PPG_MEMBER void obs_row_walls_in(int type, int j, uint32_t s_xy, unsigned char *region, int eb) {
        wv::sync();  // LDS writes of the sequential phases -> visible
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);   // (arithmetic, not a select of fields: see window_sum)
        const uint32_t rmagic = C.rp_magic + (type ? C.rq_magic - C.rp_magic : 0u);
        const int n = R * R, off = (R - 1) / 2, Wc = 2 * off + 1;
        const int nchan = C.vis_channel ? 5 : 4;
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;   // (areas are strided by the larger window: waves work on both species)
        float *visb = (float *)(region + C.off_win) + wave_idx * rmax * rmax;
        const uint32_t *visw = (const uint32_t *)visb;
        const map_t *const m = (const map_t *)(region + P.off_map);
        const double *const vt = (const double *)(region + P.off_val);
        const uint32_t *const ww = (const uint32_t *)(region + C.off_wall);
        const bool want_vis = C.mask_obs || C.vis_channel;
        const bool have_masks = C.vis_masks != nullptr;
        if (want_vis && have_masks) {
            // walls are static: the mask of this agent's cell was computed when they were set (ppg_walls_changed) -- a few words
            // instead of one Bresenham walk per window cell
            if (ln < C.vis_words) ((uint32_t *)visb)[ln] = C.vis_masks[((size_t)eb * C.vis_env_stride + x * P.G + y) * C.vis_words + ln];
            wv::sync();
        } else if (want_vis) {
            for (int i = ln; i < n; i += 64) {
                const int ci = (int)wv::mulhi((uint32_t)i, rmagic), cj = i - ci * R;
                const int gx = x - off + ci, gy = y - off + cj;
                const bool in_grid = (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                visb[i] = (in_grid && los_clear_in(ww, x, y, gx, gy)) ? 1.0f : 0.0f;
            }
            wv::sync();
        }
        const size_t obase = ((size_t)eb * (type ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)(nchan * n);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int cell = c0 + ln;
            const bool valid = cell < n;
            const int ci = (int)wv::mulhi((uint32_t)cell, rmagic), cj = cell - ci * R;
            const int gx = x - off + ci, gy = y - off + cj;
            const bool in_grid = valid && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
            const bool inb = in_grid && ci < Wc && cj < Wc;
            const int a = inb ? gx * P.G + gy : 0;
            const int am = COOP ? (inb ? (gx + P.pad) * P.Gp + gy + P.pad : 0) : a;   // (COOP: padded maps)
            double v[5];
            v[0] = (inb && ((ww[a >> 5] >> (a & 31)) & 1u)) ? 1.0 : 0.0;
            float vis = 0.0f;
            if (want_vis && in_grid) {
                if (have_masks) {
                    const int bi = (ci - off + C.vis_neg) * C.vis_w + (cj - off + C.vis_neg);
                    vis = ((visw[bi >> 5] >> (bi & 31)) & 1u) ? 1.0f : 0.0f;
                } else {
                    vis = visb[cell];
                }
            }
#pragma unroll
            for (int ch = 1; ch < 4; ++ch) {
                double t = vt[from_map(ch, (m + (THREE ? ch - 1 : ch) * P.map_n)[am])];
                if (!inb) t = 0.0;
                if (C.mask_obs) t = (double)((float)t * (inb ? vis : 0.0f));
                v[ch] = t;
            }
            v[4] = (double)vis;
            if (valid) {
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) {
                    if (ch >= nchan) continue;
                    const size_t o = obase + (size_t)ch * n + cell;
                    if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[o] = (float)v[ch];
                    else ((double *)(type ? P.obs_prey : P.obs_pred))[o] = v[ch];
                }
            }
        }
        wv::sync();  // reads done before the caller touches the maps again
    }

    // The same for a LIST of rows of one species (round 6; multi-wave and cooperative kernels, masks precomputed or not needed): the
    // rows' window cells are ONE run of n_live * R*R cells, 64 per pass with every lane busy -- a 9x9 window alone fills 1.27 passes,
    // the per-row form above spends 2 -- and their line-of-sight masks were staged for all rows at once (walls_stage_masks: one
    // memory round trip per env instead of one per row).  list[i] = row << 16 | x << 8 | y (bit 31 ignored); vm = the staged masks of
    // this species' list; chunk c of the run (64 cells) is written by the wavefront with c = first (mod stride).
    PPG_MEMBER bool walls_flat() const { return !(C.mask_obs || C.vis_channel) || C.vis_masks != nullptr; }
    // float32 rows (the reference's dtype, WO:137-139): WQ ADJACENT window cells per lane -- per channel ONE 16-byte store per lane
    // (4-byte aligned: global_store_dwordx4 takes that) instead of four 4-byte stores; the write phase is bound by the store
    // instructions the memory pipeline takes (profiles/r06/v_*).  A row's R*R cells are cut into ceil(R*R / WQ) quads; the last quad
    // of a row starts at R*R - WQ, i.e. overlaps its neighbour and writes up to WQ - 1 cells a second time with the same values, so
    // every quad is whole and there is one store path.
#ifndef PPG_WALLS_QUAD
#define PPG_WALLS_QUAD 4
#endif
    static constexpr int WQ = PPG_WALLS_QUAD;
    PPG_MEMBER bool walls_quads(int type) const {
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);
        return WQ > 1 && P.obs_f32 == 1 && R * R >= WQ;
    }
    // 64-lane chunks of a species' run of n_live rows (what obs_cells_walls hands out to the wavefronts)
    PPG_MEMBER int walls_chunks(int type, int n_live) const {
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);
        const int per_row = walls_quads(type) ? (R * R + WQ - 1) / WQ : R * R;
        return (n_live * per_row + 63) >> 6;
    }
    PPG_MEMBER void obs_quads_walls(int type, const uint32_t *list, int n_live, const uint32_t *vm, const unsigned char *region, int eb,
                                    int first, int stride, uint32_t *tk = nullptr) {
        typedef float fq_t __attribute__((vector_size((WQ > 1 ? WQ : 2) * 4), aligned(4)));   // (4-byte aligned: rows of R*R cells start anywhere)
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);
        const uint32_t rmagic = C.rp_magic + (type ? C.rq_magic - C.rp_magic : 0u);
        const int n = R * R, off = (R - 1) / 2, Wc = 2 * off + 1;
        const int nchan = C.vis_channel ? 5 : 4;
        const bool want_vis = C.mask_obs || C.vis_channel;
        const map_t *const m = (const map_t *)(region + P.off_map);
        const double *const vt = (const double *)(region + P.off_val);
        const uint32_t *const ww = (const uint32_t *)(region + C.off_wall);
        const int QR = (n + WQ - 1) / WQ;                                  // quads per row
        const uint32_t qmagic = 0xFFFFFFFFu / (uint32_t)QR + 1u;           // ceil(2^32 / QR) (QR == 1: unused)
        const int total = n_live * QR, blk = nchan * n;
        float *const out = (float *)(type ? P.obs_prey : P.obs_pred) + (size_t)eb * (size_t)(type ? P.cap_prey : P.cap_pred) * (size_t)blk;
        Turn t;   // (chunks of 64 quads: handed out statically, or through the ticket word tk -- ppg_env_coop.h)
        for (int p0 = turn_begin(t, tk, first, stride, 1); p0 * 64 < total; p0 = turn_next(t, p0, 1)) {
            const int g = p0 * 64 + ln;
            const bool valid = g < total;
            const uint32_t gs = valid ? (uint32_t)g : 0u;
            const int i = QR == 1 ? (int)gs : (int)wv::mulhi(gs, qmagic);
            int cell0 = ((int)gs - i * QR) * WQ;
            cell0 = cell0 > n - WQ ? n - WQ : cell0;                       // the row's last quad overlaps its neighbour
            const uint32_t en = list[i];
            const int x = (int)((en >> 8) & 255u), y = (int)(en & 255u);
            int ci = (int)wv::mulhi((uint32_t)cell0, rmagic), cj = cell0 - ci * R;
            bool in_grid[WQ], inb[WQ];
            int a[WQ], bi[WQ];
            uint32_t wallw_[WQ], visw_[WQ], mb[WQ][3];
#pragma unroll
            for (int k = 0; k < WQ; ++k) {
                const int gx = x - off + ci, gy = y - off + cj;
                in_grid[k] = valid && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                inb[k] = in_grid[k] && ci < Wc && cj < Wc;
                a[k] = inb[k] ? gx * P.G + gy : 0;
                const int am = COOP ? (inb[k] ? (gx + P.pad) * P.Gp + gy + P.pad : 0) : a[k];   // (COOP: padded maps)
                wallw_[k] = ww[a[k] >> 5];
                bi[k] = (ci - off + C.vis_neg) * C.vis_w + (cj - off + C.vis_neg);
                visw_[k] = want_vis ? vm[i * C.vis_words + (bi[k] >> 5)] : 0u;
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) mb[k][ch - 1] = (uint32_t)(m + (THREE ? ch - 1 : ch) * P.map_n)[am];
                if (++cj == R) { cj = 0; ++ci; }
            }
            float f[5][WQ];
#pragma unroll
            for (int k = 0; k < WQ; ++k) {
                f[0][k] = (inb[k] && ((wallw_[k] >> (a[k] & 31)) & 1u)) ? 1.0f : 0.0f;
                const float vis = (want_vis && in_grid[k] && ((visw_[k] >> (bi[k] & 31)) & 1u)) ? 1.0f : 0.0f;
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) {
                    const double tt = vt[from_map(ch, mb[k][ch - 1])];
                    f[ch][k] = inb[k] ? (float)tt : 0.0f;
                    if (C.mask_obs) f[ch][k] = f[ch][k] * (inb[k] ? vis : 0.0f);   // (in float32, like the reference: WO:591-594)
                }
                f[4][k] = vis;
            }
            if (valid) {
                const uint32_t o0 = (uint32_t)((int)((en >> 16) & 0x7FFFu) * blk + cell0);
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) {
                    if (ch >= nchan) continue;
                    fq_t v;
#pragma unroll
                    for (int k = 0; k < WQ; ++k) v[k] = f[ch][k];
                    *(fq_t *)(out + o0 + (uint32_t)(ch * n)) = v;
                }
            }
        }
    }
    PPG_MEMBER void obs_cells_walls(int type, const uint32_t *list, int n_live, const uint32_t *vm, const unsigned char *region, int eb,
                                    int first, int stride, uint32_t *tk = nullptr) {
        if (walls_quads(type)) { obs_quads_walls(type, list, n_live, vm, region, eb, first, stride, tk); return; }
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);
        const uint32_t rmagic = C.rp_magic + (type ? C.rq_magic - C.rp_magic : 0u);
        const uint32_t nmagic = C.np_magic + (type ? C.nq_magic - C.np_magic : 0u);
        const int n = R * R, off = (R - 1) / 2, Wc = 2 * off + 1;
        const int nchan = C.vis_channel ? 5 : 4;
        const bool want_vis = C.mask_obs || C.vis_channel;
        const map_t *const m = (const map_t *)(region + P.off_map);
        const double *const vt = (const double *)(region + P.off_val);
        const uint32_t *const ww = (const uint32_t *)(region + C.off_wall);
        const int total = n_live * n, blk = nchan * n;
        const size_t obase = (size_t)eb * (size_t)(type ? P.cap_prey : P.cap_pred) * (size_t)blk;   // (wave-uniform; the lane's part is 32-bit)
        // float32 rows -- the reference's dtype (WO:137-139) -- have a loop of their own: no dtype branch per store, no float64 detour
        auto cells = [&](auto f32_tag) {
        constexpr bool F32 = decltype(f32_tag)::value;
        // chunks in flight per wavefront: the dependent LDS reads of one (list entry -> map bytes -> values) hide behind the other's
        constexpr int U = 2;
        Turn t;
        for (int p0 = turn_begin(t, tk, first, stride, U); p0 * 64 < total; p0 = turn_next(t, p0, U)) {
            bool valid[U], in_grid[U], inb[U];
            int cell[U], i[U], ci[U], cj[U];
            uint32_t en[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = (p0 + u * t.us) * 64 + ln;
                valid[u] = g < total;
                const uint32_t gs = valid[u] ? (uint32_t)g : 0u;
                i[u] = n == 1 ? (int)gs : (int)wv::mulhi(gs, nmagic);   // (a 1x1 window: ceil(2^32 / 1) does not fit the magic word)
                cell[u] = (int)gs - i[u] * n;
                en[u] = list[i[u]];
                ci[u] = (int)wv::mulhi((uint32_t)cell[u], rmagic); cj[u] = cell[u] - ci[u] * R;
            }
            uint32_t wallw_[U], visw_[U], mb[U][3];
            int a[U], bi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x = (int)((en[u] >> 8) & 255u), y = (int)(en[u] & 255u);
                const int gx = x - off + ci[u], gy = y - off + cj[u];
                in_grid[u] = valid[u] && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                inb[u] = in_grid[u] && ci[u] < Wc && cj[u] < Wc;
                a[u] = inb[u] ? gx * P.G + gy : 0;
                const int am = COOP ? (inb[u] ? (gx + P.pad) * P.Gp + gy + P.pad : 0) : a[u];   // (COOP: padded maps)
                wallw_[u] = ww[a[u] >> 5];
                bi[u] = (ci[u] - off + C.vis_neg) * C.vis_w + (cj[u] - off + C.vis_neg);
                visw_[u] = want_vis ? vm[i[u] * C.vis_words + (bi[u] >> 5)] : 0u;
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) mb[u][ch - 1] = (uint32_t)(m + (THREE ? ch - 1 : ch) * P.map_n)[am];
            }
            double t[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) t[u][ch - 1] = vt[from_map(ch, mb[u][ch - 1])];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double v[5];
                float f[5];
                v[0] = (inb[u] && ((wallw_[u] >> (a[u] & 31)) & 1u)) ? 1.0 : 0.0;
                f[0] = (float)v[0];
                const float vis = (want_vis && in_grid[u] && ((visw_[u] >> (bi[u] & 31)) & 1u)) ? 1.0f : 0.0f;
#pragma unroll
                for (int ch = 1; ch < 4; ++ch) {
                    double tt = inb[u] ? t[u][ch - 1] : 0.0;
                    f[ch] = (float)tt;
                    if (C.mask_obs) { f[ch] = f[ch] * (inb[u] ? vis : 0.0f); tt = (double)f[ch]; }   // (in float32, like the reference: WO:591-594)
                    v[ch] = tt;
                }
                v[4] = (double)vis; f[4] = vis;
                if (valid[u]) {
                    const uint32_t o0 = (uint32_t)((int)((en[u] >> 16) & 0x7FFFu) * blk + cell[u]);
#pragma unroll
                    for (int ch = 0; ch < 5; ++ch) {
                        if (ch >= nchan) continue;
                        const uint32_t o = o0 + (uint32_t)(ch * n);
                        if (F32) ((float *)(type ? P.obs_prey : P.obs_pred) + obase)[o] = f[ch];
                        else if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred) + obase)[o] = f[ch];
                        else ((double *)(type ? P.obs_prey : P.obs_pred) + obase)[o] = v[ch];
                    }
                }
            }
        }
        };
        if (P.obs_f32 == 1) cells(TagTrue{});
        else cells(TagFalse{});
    }
    // the line-of-sight masks of all live rows, straight from the row registers into their list positions (predator list entry i at
    // i, prey list entry i at 64 + i): every load is independent of every other
    PPG_MEMBER void walls_stage_masks() {
        if (!(C.mask_obs || C.vis_channel) || C.vis_masks == nullptr) return;
        uint32_t *vm = (uint32_t *)((unsigned char *)map - P.off_map + C.off_vm);
        constexpr int MW = 4;   // words fetched side by side (windows up to 11x11; the rest in a loop)
        const int vw = C.vis_words;
        uint32_t mw[T][MW];
        int pos[T], n[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < T; ++r) {   // all loads first: one memory round trip
            const int type = type_of(r);
            const bool on = (alive[r] >> ln) & 1ull;
            pos[r] = ((type ? 64 : 0) + n[type] + (int)wv::prefix(alive[r])) * vw;
            n[type] += wv::popc(alive[r]);
            const uint32_t *src = C.vis_masks + ((size_t)b * C.vis_env_stride + (on ? (xy[r] >> 8) * P.G + (xy[r] & 255u) : 0u)) * vw;
#pragma unroll
            for (int w = 0; w < MW; ++w) { mw[r][w] = 0; if (on && w < vw) mw[r][w] = src[w]; }
        }
#pragma unroll
        for (int r = 0; r < T; ++r) {
            if (!((alive[r] >> ln) & 1ull)) continue;
#pragma unroll
            for (int w = 0; w < MW; ++w) if (w < vw) vm[pos[r] + w] = mw[r][w];
            if (vw > MW) {
                const uint32_t *src = C.vis_masks + ((size_t)b * C.vis_env_stride + (xy[r] >> 8) * P.G + (xy[r] & 255u)) * vw;
                for (int w = MW; w < vw; ++w) vm[pos[r] + w] = src[w];
            }
        }
    }

    // _get_observation of the drive-conditioned env (DRV:551-616), one window CELL per lane: the three world channels of a cell
    // are looked up once, stored, and staged in LDS for the window sums -- np.sum(observation[c]) in numpy's order: eight
    // interleaved accumulators r_q = a[q] + a[q+8] + ... (lanes 8g .. 8g+7 of lane group g = channel g+1 run them side by side for
    // all three channels), combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by three xor-shuffles (IEEE addition commutes, so
    // every lane of a group ends with the same bits), then the tail elements one by one.  Needs 8 <= R*R <= 128 (numpy switches
    // to a plain loop below and to recursive halves above); other sizes take the per-element path.
    // Tried and slower (per 1365-env launch, against this version = 1.00): staging all four planes and writing the block in
    // element order with 16-byte stores 1.11 (the second pass over LDS costs more than the wider stores save); two adjacent cells
    // per lane with aligned pair stores 1.07 (shuffles for the odd planes, 41 of 64 lanes busy); element order through the
    // descriptor table with the staging folded into the same pass 1.11 (one lookup per ELEMENT instead of per cell, also for
    // plane 0).  The per-cell lookups are what this path is bound by, not the width of its stores.
    PPG_MEMBER void obs_row_drive(int type, int j, uint32_t s_xy, double s_e) {
        wv::sync();
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);
        const uint32_t rmagic = C.rp_magic + (type ? C.rq_magic - C.rp_magic : 0u);
        const int n = R * R, off = (R - 1) / 2, Wc = 2 * off + 1;
        const int nd = type ? C.n_drive[1] : C.n_drive[0];
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq, rmax2 = rmax * rmax;
        double *win = (double *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * 4 * rmax2;
        const size_t obase = ((size_t)b * (type ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)((4 + nd) * n);
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int cell = c0 + ln;
            const bool valid = cell < n;
            const int ci = (int)wv::mulhi((uint32_t)cell, rmagic), cj = cell - ci * R;
            const int gx = x - off + ci, gy = y - off + cj;
            const bool inb = valid && ci < Wc && cj < Wc && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
            const int a = inb ? gx * P.G + gy : 0;
            double v[4];
            v[0] = inb ? 0.0 : 1.0;                      // DRV:561-562: 1 everywhere except the in-grid part of the window
#pragma unroll
            for (int ch = 1; ch < 4; ++ch) {
                const double t = val[from_map(ch, chmap(ch)[a])];
                v[ch] = inb ? t : 0.0;
            }
            if (valid) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const size_t o = obase + (size_t)ch * n + cell;
                    if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[o] = (float)v[ch];
                    else ((double *)(type ? P.obs_prey : P.obs_pred))[o] = v[ch];
                    if (ch) win[(ch - 1) * rmax2 + cell] = v[ch];
                }
            }
        }
        wv::sync();
        // the three window sums, side by side
        const int grp = ln >> 3, q = ln & 7, n8 = n & ~7;
        const double *wc = win + (grp < 3 ? grp : 0) * rmax2;
        double acc = wc[q];
        for (int i = 8 + q; i < n8; i += 8) acc += wc[i];
        acc = acc + wv::shfl_xor_f64(acc, 1);
        acc = acc + wv::shfl_xor_f64(acc, 2);
        acc = acc + wv::shfl_xor_f64(acc, 4);
        for (int i = n8; i < n; ++i) acc += wc[i];
        const double sum1 = readlane_f64(acc, 0), sum2 = readlane_f64(acc, 8), sum3 = readlane_f64(acc, 16);
        double dv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= nd) continue;
            const int kind = type ? C.drive_kind[1][k] : C.drive_kind[0][k];
            double t;
            if (kind == 0) t = 1.0 - s_e / (type ? C.hunger_safe[1] : C.hunger_safe[0]);
            else if (kind == 1) t = s_e / (type ? C.thr_q : C.thr_p);
            else if (kind == 2) t = sum2 / C.norm_prey_opp;
            else if (kind == 3) t = sum1 / C.norm_pred_danger;
            else t = sum3 / C.norm_grass_opp;
            dv[k] = safe_clip01(t);
        }
        // the drive planes: one scalar per plane (DRV:566-569).  The nd planes are ONE contiguous run of nd * n elements: written as
        // element pairs (16-byte stores for float64), behind one leading single element when the run starts on an odd element
        {
            const size_t start = obase + (size_t)4 * n;
            const int len = nd * n, sh = (int)(start & 1);
            auto plane_value = [&](int e) { return e < n ? dv[0] : e < 2 * n ? dv[1] : e < 3 * n ? dv[2] : dv[3]; };
            if (sh && ln == 0) {
                if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[start] = (float)dv[0];
                else ((double *)(type ? P.obs_prey : P.obs_pred))[start] = dv[0];
            }
            for (int e = sh + 2 * ln; e < len; e += 128) {
                const double v0 = plane_value(e), v1 = plane_value(e + 1);
                if (e + 1 < len) {
                    if (P.obs_f32) {
                        float2 f; f.x = (float)v0; f.y = (float)v1;
                        *(float2 *)((float *)(type ? P.obs_prey : P.obs_pred) + start + e) = f;
                    } else {
                        double2 g; g.x = v0; g.y = v1;
                        *(double2 *)((double *)(type ? P.obs_prey : P.obs_pred) + start + e) = g;
                    }
                } else if (P.obs_f32) {
                    ((float *)(type ? P.obs_prey : P.obs_pred))[start + e] = (float)v0;
                } else {
                    ((double *)(type ? P.obs_prey : P.obs_pred))[start + e] = v0;
                }
            }
        }
        wv::sync();
    }

    PPG_MEMBER void obs_row(int type, int j, uint32_t s_xy, double s_e = 0.0) {
        if (COOP && !WALLS) { obs_row_coop(type, j, s_xy); return; }
        if (FASTOBS) {
            if (type) obs_row_fast<1>(j, s_xy);
            else obs_row_fast<0>(j, s_xy);
            return;
        }
        if (WALLS) { obs_row_walls(type, j, s_xy); return; }
        if (DRIVE) {
            const int Rn = P.Rp + (type ? P.Rq - P.Rp : 0);
            if (Rn * Rn >= 8 && Rn * Rn <= 128) { obs_row_drive(type, j, s_xy, s_e); return; }
        }
        wv::sync();  // LDS writes of the sequential phases -> visible
        const int R = P.Rp + (type ? P.Rq - P.Rp : 0);   // (arithmetic, not a select of fields: see window_sum)
        double dv[4] = {0.0, 0.0, 0.0, 0.0};
        if (DRIVE) drive_features(type, s_e, s_xy, dv);
        const int blk = C.blk_p + (type ? C.blk_q - C.blk_p : 0);   // channels x R x R
        const int off = (R - 1) / 2;
        const int x = (int)(s_xy >> 8), y = (int)(s_xy & 255u);
        const int s_cell = x * P.G + y;
        const bool interior = !WALLS && !DRIVE && (R & 1) && x >= off && y >= off && x + off < P.G && y + off < P.G;
        const uint2 *L = (const uint2 *)(lut + (type ? P.nch_p * 128 : 0));
        const int nch = type ? P.nch_q : P.nch_p;
        // walls variant: the line-of-sight mask of this agent, one value per cell of the R x R array (WO:577-589), staged in
        // LDS once and used by up to four channels (every wave of a multi-wave workgroup has its own staging area)
        const int rmax = P.Rp > P.Rq ? P.Rp : P.Rq;   // (areas are strided by the larger window: waves work on both species)
        float *visb = (float *)((unsigned char *)map + C.off_win - P.off_map) + wave_idx * rmax * rmax;
        const bool want_vis = WALLS && (C.mask_obs || C.vis_channel);
        const bool have_masks = WALLS && C.vis_masks != nullptr;
        const uint32_t *visw = (const uint32_t *)visb;
        if (want_vis && have_masks) {
            // walls are static: the mask of this agent's cell was computed when they were set (ppg_walls_changed) -- a few words
            // instead of one Bresenham walk per window cell
            if (ln < C.vis_words) ((uint32_t *)visb)[ln] = C.vis_masks[((size_t)b * C.vis_env_stride + s_cell) * C.vis_words + ln];
            wv::sync();
        } else if (want_vis) {
            for (int i = ln; i < R * R; i += 64) {
                const int ci = i / R, cj = i - ci * R;
                const int gx = x - off + ci, gy = y - off + cj;
                const bool in_grid = (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                visb[i] = (in_grid && los_clear(x, y, gx, gy)) ? 1.0f : 0.0f;
            }
            wv::sync();
        }
        const size_t obase = ((size_t)b * (type ? P.cap_prey : P.cap_pred) + (size_t)j) * (size_t)blk;
        for (int ch = 0; ch < nch; ++ch) {
            const uint2 d = L[ch * 64 + ln];
            double v[2];
            if (interior) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = h ? d.y : d.x;
                    const int a = (int)(int16_t)(w & 0xFFFFu) + s_cell;
                    v[h] = val[from_map((int)((w >> 24) & 3u), map[a])];  // a non-existent element has moff 0: reads the observer's own cell in the all-zero map, unused
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w = h ? d.y : d.x;
                    const int gx = x + (int)((w >> 16) & 15u) - 8, gy = y + (int)((w >> 20) & 15u) - 8;
                    const bool inb = (w & 0x8000000u) && (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                    const int a = (int)(int16_t)(w & 0xFFFFu) + s_cell;
                    double t = val[from_map((int)((w >> 24) & 3u), map[inb ? a : 0])];   // map[0] (channel 0, cell 0) is always 0 -> val[0] = 0.0
                                                                                          // (a drive element carries its plane index there: its map entry is 0 anyway)
                    if (DRIVE && (w & 0x20000000u)) {       // a drive channel: the whole (R,R) plane holds one scalar (DRV:566-569)
                        const uint32_t k = (w >> 24) & 3u;
                        t = k == 0 ? dv[0] : k == 1 ? dv[1] : k == 2 ? dv[2] : dv[3];
                    } else if (!WALLS) {
                        if (!inb && (w & 0x3000000u) == 0u) t = 1.0;  // channel 0: 1 outside the grid (BASE:522-523)
                    } else {
                        // _get_observation of the walls env (WO:527-601): channel 0 = walls inside the window (0 outside the
                        // grid); channels 1-3 optionally multiplied -- in float32, like the reference -- by the line-of-
                        // sight mask; optional last channel = the mask itself
                        const bool vis_elem = (w & 0x10000000u) != 0u;
                        // the mask is computed for every cell of the R x R array that maps into the grid (WO:577-589), also
                        // for the last row / column of an even R, which the window copy (WO:543) leaves untouched
                        const bool in_grid = (unsigned)gx < (unsigned)P.G && (unsigned)gy < (unsigned)P.G;
                        const bool need_vis = vis_elem ? in_grid : (inb && C.mask_obs && (w & 0x3000000u) != 0u);
                        const int vdx = (int)((w >> 16) & 15u) - 8, vdy = (int)((w >> 20) & 15u) - 8;
                        float vis = 0.0f;
                        if (need_vis && have_masks) {
                            const int bi = (vdx + C.vis_neg) * C.vis_w + (vdy + C.vis_neg);
                            vis = ((visw[bi >> 5] >> (bi & 31)) & 1u) ? 1.0f : 0.0f;
                        } else if (need_vis) {
                            vis = visb[(vdx + off) * R + (vdy + off)];
                        }
                        if (vis_elem) t = (double)vis;
                        else if ((w & 0x3000000u) == 0u) t = (inb && wall_at(gx, gy)) ? 1.0 : 0.0;
                        else if (C.mask_obs) t = (double)((float)t * vis);
                    }
                    v[h] = t;
                }
            }
            if ((WALLS || DRIVE) && (blk & 1)) {
                // an odd number of channels x an odd window: blocks start at odd element offsets, so element pairs cannot be
                // stored as aligned vectors.  For these geometries the host lays the descriptors out "strided": this lane's
                // two elements are ch*128 + ln and ch*128 + 64 + ln, i.e. each store instruction writes 64 consecutive
                // elements (ppg_build_lut)
                const size_t o = obase + (size_t)ch * 128 + (size_t)ln;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!((h ? d.y : d.x) & 0x4000000u)) continue;
                    if (P.obs_f32) ((float *)(type ? P.obs_prey : P.obs_pred))[o + 64 * h] = (float)v[h];
                    else ((double *)(type ? P.obs_prey : P.obs_pred))[o + 64 * h] = v[h];
                }
            } else if (d.x & 0x4000000u) {
                const size_t o = obase + (size_t)ch * 128 + 2 * (size_t)ln;
                store_obs_pair(type ? P.obs_prey : P.obs_pred, P.obs_f32, o, v[0], v[1]);
            }
        }
        wv::sync();  // reads done before the caller touches the maps again
    }

    // multi-wave variants: rows of the published list, every stride-th one starting at `w`
    PPG_MEMBER void obs_shared(int w, int stride = NW) {
        const uint32_t *lst = (const uint32_t *)scr;
        const uint32_t head = wv::first(lst[0]);   // rows in the list | predators among them (they come first) << 16
        const int n = (int)(head & 0xFFFFu);
        if (FASTOBS) {
            // predators come first in the list: two loops with a compile-time species each (one loop with a run-time species keeps
            // both species' unrolled observation code and all ten descriptor registers live together: +20 registers)
            const int n_pred = (int)(head >> 16);
            int i = w;
            for (; i < n_pred; i += stride) {
                const uint32_t en = wv::first(lst[1 + i]);
                obs_row_fast<0>((int)((en >> 16) & 0x7FFFu), en & 0xFFFFu);
            }
            for (; i < n; i += stride) {
                const uint32_t en = wv::first(lst[1 + i]);
                obs_row_fast<1>((int)((en >> 16) & 0x7FFFu), en & 0xFFFFu);
            }
            return;
        }
        if (WALLS && walls_flat()) {   // the listed rows as runs of window cells, chunk c to wavefront c mod stride
            const int n_pred = (int)(head >> 16);
            const unsigned char *region = (const unsigned char *)map - P.off_map;
            const uint32_t *vm = (const uint32_t *)(region + C.off_vm);
            wv::sync();
            obs_cells_walls(0, lst + 1, n_pred, vm, region, b, w, stride);
            const int chunks = walls_chunks(0, n_pred);
            int first = (w - chunks) % stride;
            if (first < 0) first += stride;
            obs_cells_walls(1, lst + 1 + n_pred, n - n_pred, vm + 64 * C.vis_words, region, b, first, stride);
            return;
        }
        for (int i = w; i < n; i += stride) {
            const uint32_t en = wv::first(lst[1 + i]);
            const int ty = (int)(en >> 31), row = (int)((en >> 16) & 0x7FFFu);
            // (drive variant: the agent's energy is its entry of the LDS value table -- row energies are kept current there)
            const double s_e = DRIVE ? first_f64(val[validx_row(ty, row)]) : 0.0;
            obs_row(ty, row, en & 0xFFFFu, s_e);
        }
    }
    // a helper wave of a multi-wave workgroup: wait until wave 0 has finished the transition, then write its share
    PPG_MEMBER void run_helper(int w) {
        if (FASTOBS) {
            const uint2 *L2 = (const uint2 *)C.obs_lut;
#pragma unroll
            for (int c = 0; c < 2; ++c) { uint2 d; d.x = 0; d.y = 0; if (c < P.nch_p) d = L2[c * 64 + ln]; lutr[2 * c] = d.x; lutr[2 * c + 1] = d.y; }
#pragma unroll
            for (int c = 0; c < 3; ++c) { uint2 d; d.x = 0; d.y = 0; if (c < P.nch_q) d = L2[(P.nch_p + c) * 64 + ln]; lutr[4 + 2 * c] = d.x; lutr[5 + 2 * c] = d.y; }
        }
        wv::wg_barrier();
        obs_shared(w);
    }

    // Multi-wave kernels: the shared writing of the published rows.  It comes AFTER rewards_and_store: the row registers are dead by
    // then, which is what keeps these kernels inside 128 registers (with the stores behind the observation loops they spilled).
    static constexpr bool DEFER_OBS = NW > 1 && !COOP;
    PPG_MEMBER void obs_finish() {
        if (!DEFER_OBS) return;
        if (!ADAPTIVE_HELPERS || helpers) { wv::wg_barrier(); obs_shared(0); }
        else { wv::sync(); obs_shared(0, 1); }
    }

    // write_now = false (step paths of the multi-wave kernels): publish only, obs_finish() follows the table stores
    PPG_MEMBER void obs_all_alive(bool write_now = true) {
        if (COOP) { coop_publish(); return; }   // written by the whole workgroup after its barrier (env_main)
        if (NW > 1) {  // publish (type, row, cell) of every live row, then all waves of the workgroup share the rows
            // (an env whose helper waves have left -- ADAPTIVE_HELPERS -- goes through the same list with stride 1: a second, register-
            // indexed copy of the observation code in one kernel is what pushed the multi-wave kernels over 128 registers)
            uint32_t *lst = (uint32_t *)scr;
            int n = 0;
            wv::sync();
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if ((alive[r] >> ln) & 1ull)
                    lst[1 + n + (int)wv::prefix(alive[r])] = ((uint32_t)type_of(r) << 31) | ((uint32_t)row_of(r, ln) << 16) | xy[r];
                n += wv::popc(alive[r]);
            }
            if (ln == 0) lst[0] = (uint32_t)n | ((uint32_t)wv::popc(alive[0]) << 16);
            if (WALLS) walls_stage_masks();
            if (write_now) obs_finish();
            return;
        }
#pragma unroll
        for (int r = 0; r < T; ++r) {
            uint64_t m = alive[r];
            while (m) {
                const int k = wv::ctz(m);
                m &= m - 1;
                obs_row(type_of(r), row_of(r, k), wv::readlane(xy[r], k), DRIVE ? readlane_f64(e[r], k) : 0.0);
            }
        }
    }

