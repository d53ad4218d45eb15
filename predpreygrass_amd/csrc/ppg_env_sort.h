// ppg_env_sort.h -- part of struct ppg::Env (ppg_kernel.h includes it INSIDE the struct's body: member functions, no include guard,
// not a header of its own): removal of last call's dead rows and the agents.sort() of BASE:468 (ballot-counted ranks, rows permuted through LDS).
    // ---- drop last call's dead rows and bring the rows into self.agents order ----------
    // (BASE:222-225 removal; BASE:468 sort).  Rows are [sorted prefix..., appended rows...];
    // appended rows are inserted by counting smaller keys with ballots.
    PPG_MEMBER void compact_and_sort(bool do_sort) {
#pragma unroll
        for (int type = 0; type < 2; ++type) {
            // sorted-prefix length m over this type's rows
            int m_sorted = n_rows[type];
            bool has_dead = false;
#pragma unroll
            for (int r = 0; r < T; ++r)
                if (type_of(r) == type) has_dead = has_dead || ((rows[r] & ~alive[r]) != 0);
            if (do_sort) {
#pragma unroll
                for (int r = T - 1; r >= 0; --r) {
                    if (type_of(r) != type) continue;
                    uint32_t prev = wv::shfl_up1(key[r]);
                    if (r >= 2) {
                        uint32_t carry = wv::readlane(key[r - 1], 63);
                        if (ln == 0) prev = carry;
                    }
                    const bool first_row = (row_of(r, ln) == 0);
                    uint64_t brk = wv::ballot(!first_row && key[r] < prev) & rows[r];
                    if (brk) m_sorted = row_of(r, wv::ctz(brk));
                }
            }
            if (!has_dead && m_sorted >= n_rows[type]) continue;  // nothing to do

            uint32_t rk[T];
            uint64_t sorted_alive[T], unsorted_alive[T];
            int before = 0;
#pragma unroll
            for (int r = 0; r < T; ++r) {
                rk[r] = 0; sorted_alive[r] = 0; unsorted_alive[r] = 0;
                if (type_of(r) != type) continue;
                const int lo = row_of(r, 0);
                uint64_t in_prefix = lowmask(m_sorted - lo);
                sorted_alive[r] = alive[r] & in_prefix;
                unsorted_alive[r] = alive[r] & ~in_prefix;
                rk[r] = (uint32_t)before + wv::prefix(sorted_alive[r]);
                before += wv::popc(sorted_alive[r]);
            }
#pragma unroll
            for (int ru = 0; ru < T; ++ru) {
                if (type_of(ru) != type) continue;
                uint64_t mu = unsorted_alive[ru];
                while (mu) {
                    const int ku = wv::ctz(mu);
                    mu &= mu - 1;
                    const uint32_t s_key = wv::readlane(key[ru], ku);
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < T; ++r) {
                        if (type_of(r) != type) continue;
                        cnt += wv::popc(wv::ballot(key[r] < s_key) & alive[r]);
                        if (((sorted_alive[r] >> ln) & 1ull) && key[r] > s_key) rk[r] += 1;
                    }
#pragma unroll
                    for (int r = 0; r < T; ++r)
                        if (r == ru) rk[r] = wv::writelane(rk[r], ku, (uint32_t)cnt);
                }
            }
            // scatter through LDS, one 8-byte field at a time
            const int sbase = 0;   // (one species at a time)
            int n_new = 0;
#pragma unroll
            for (int r = 0; r < T; ++r)
                if (type_of(r) == type) n_new += wv::popc(alive[r]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (f == 1 && !CARRY_CUM) continue;  // (the cumulative reward: only the cooperative kernels carry it in registers)
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    if (type_of(r) != type) continue;
                    if ((alive[r] >> ln) & 1ull) {
                        uint64_t v;
                        if (f == 0) v = (uint64_t)__double_as_longlong(e[r]);
                        else if (f == 1) v = (uint64_t)__double_as_longlong(cum[r]);
                        else if (f == 2) v = ((uint64_t)key[r] << 32) | (uint32_t)id[r];
                        else v = (uint64_t)xy[r] | ((uint64_t)((owns[r] >> ln) & 1ull) << 16) | ((uint64_t)keep[r] << 20);
                        scr[sbase + rk[r]] = v;
                    }
                }
                wv::sync();
#pragma unroll
                for (int r = 0; r < T; ++r) {
                    if (type_of(r) != type) continue;
                    const int i = row_of(r, ln);
                    if (i < n_new) {
                        uint64_t v = scr[sbase + i];
                        if (f == 0) e[r] = __longlong_as_double((long long)v);
                        else if (f == 1) cum[r] = __longlong_as_double((long long)v);
                        else if (f == 2) { key[r] = (uint32_t)(v >> 32); id[r] = (int32_t)(uint32_t)v; }
                        else { xy[r] = (uint32_t)(v & 0xFFFFu); ev[r] = (uint32_t)((v >> 16) & 1u); keep[r] = (uint32_t)(v >> 20) & 0x3FFFFu; }
                    } else if (f == 3) {
                        xy[r] = 0xFFFFu; ev[r] = 0; keep[r] = 0;
                    }
                }
                wv::sync();
            }
#pragma unroll
            for (int r = 0; r < T; ++r) {
                if (type_of(r) != type) continue;
                rows[r] = lowmask(n_new - row_of(r, 0));
                alive[r] = rows[r];
                owns[r] = wv::ballot(ev[r] & 1u) & rows[r];
                ev[r] = 0;
            }
            n_rows[type] = n_new;
        }
    }

    // GEN2, after the rows have their final order: type masks, and agent_last_reproduction of every surviving row,
    // read from HBM at the row's start-of-step slot like the cumulative rewards
    PPG_MEMBER void after_compact() {
#pragma unroll
        for (int r = 0; r < T; ++r) {
            t2m[r] = wv::ballot((id[r] >> 16) & 1) & rows[r];
            lr[r] = ((alive[r] >> ln) & 1ull) ? C.row_lastrep[(size_t)b * P.S + (keep[r] >> 8)] : 0;
        }
    }

