// sppm_deposit_wg.hpp -- the per-pass rule's deposit with the pair tests on the matrix pipe (round 3).
// AN EXPERIMENT, OFF BY DEFAULT (SPPM_DEP_WG=1 selects it): bit-identical deposits, but 4.56 ms per launch against the
// 3.03 ms of k_deposit_tiles at the driver's command.  What was learnt is in DESIGN.md section 10: the matrix-pipe form of
// the distance test itself is ~1.7x faster than the vector form (removing it saves 0.6 ms here, 1.03 ms there), but four
// wavefronts that stage behind a barrier per round wait for the slowest of their loads every round (staging alone:
// 2.46 ms here, ~1.4 ms there).
// Part of the single translation unit sppm_kernels.hip (included there after sppm_deposit.hpp, whose plan, tile,
// cluster and region helpers it uses; not a stand-alone header).
//
// Same work items, same staged sets and same accepted pairs as k_deposit_tiles (sppm_deposit.hpp); what changes is who
// does what.  The distance-only test of a staged record against a hitpoint is a contraction over K = 4,
//
//     |h - q|^2 - |h'|^2 = [q'x q'y q'z |q'|^2] . [-2h'x -2h'y -2h'z 1]      (h' = h - c, q' = q - c, c the cluster's centre),
//
// so ONE v_mfma_f64_16x16x4_f64 evaluates it for 16 records x 16 hitpoints: A[i][k] comes from the staged rows (one 8-byte
// LDS read per lane where the vector form broadcast 32 bytes per record to the whole wavefront), B[k][j] is a register,
// and lane l receives D[(l >> 4) + 4r][l & 15], r = 0..3 -- four records against ONE hitpoint.  A wavefront therefore
// owns 16 hitpoints, four lanes (l >> 4 = 0..3) per hitpoint, each lane summing the records it is dealt; the four
// wavefronts of a workgroup own the 64 hitpoints of a tile together, stage the cluster's candidates together (a
// quarter of the groups each, into one LDS buffer, two buffers in turn, one barrier per round) and each tests every
// staged row against its own 16 hitpoints.  The full test (normal, flatness and distance; ~20 % of the staged rows)
// runs in the same lane layout on the vector pipe: lane (j, k) tests hitpoint j against rows k, k + 4, ...
//
// Exactness is the vector form's: the matrix pipe's value is rounded differently from the reference's |h - q|^2, so it
// only feeds comparisons against the band E around the bound (sppm_deposit.hpp), and a pair inside the band is decided
// from the positions in the reference's operation order.  The flux sums are v_add_f64 of integer-valued doubles
// (k_grid_scatter), exact below 2^53, moved to the 64-bit atomics before they can reach it.
#pragma once

typedef double dep_v4d __attribute__((ext_vector_type(4)));
#define DWG_ROWS 272         // 256 candidates staged per round (distance-only rows from 0 up, full-test rows from the top down) + the last MFMA block's overhang
#define DWG_ROW 11           // doubles per staged row: 22 dwords spread the 16 rows of an A operand over all LDS banks
#define DWG_FLUSH_ROWS 8192  // a lane adds at most a quarter of the staged rows: 2048 x 2^41 = 2^52

// A wavefront's walk over the virtual list of a region's records: the region's cells, x fastest, then y, then z (the
// order k_deposit_plan counts in), i.e. one contiguous record range per (y, z) row of cells.  Rows are taken 64 at a
// time (one per lane), clipped to the item's chunk [v_lo, v_hi) of the list; the clipped entries of a row group are cut
// into groups of 64 and numbered across row groups.  All four wavefronts walk the same sequence and take the groups
// that are theirs (group number mod 4), so they agree on when the list ends.
struct DwgWalk {
    int rg;                 // current row group
    bool loaded, done;
    uint32_t r_s, r_pre;    // per lane: first record of the row (clipped), entries before it in this row group (clipped)
    unsigned int rg_tot;    // clipped entries of this row group
    unsigned int g_base;    // number of the row group's first group
    unsigned int voff;      // list offset of the row group's start
};

template <bool TAP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_deposit_wg(HpArrays hp, RecArrays rec, const uint32_t* __restrict__ cell_start, TileParams tp, const uint32_t* __restrict__ items,
             unsigned int* __restrict__ q, unsigned long long* __restrict__ counters, int32_t* __restrict__ err)
{
    static_assert(DEP_H == 1, "one hitpoint per lane of the tile");
    __shared__ double s_row[2][DWG_ROWS][DWG_ROW];
    __shared__ unsigned int s_cnt[3][2];  // [round mod 3][distance-only, full test]: rows staged
    __shared__ unsigned int s_item;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, kq = lane >> 4, jq = lane & 15;
    const int P = hp.P;
    const double thr = tp.gather_r2 * 0.01;
    const double band_rel = 1e-9 + 8e-15 / thr;
    const double thr2_lo = thr * thr * (1.0 - band_rel), thr2_hi = thr * thr * (1.0 + band_rel);
    const unsigned int n_items = q[0];
    for (bool first_item = true;; first_item = false) {
        __syncthreads();  // every wavefront is done with the item before (its counters, its item number)
        if (tid == 0) s_item = first_item ? blockIdx.x : gridDim.x + atomicAdd(&q[1], 1u);
        if (tid < 6) (&s_cnt[0][0])[tid] = 0u;
        __syncthreads();
        const unsigned int item_i = s_item;
        if (item_i >= n_items) break;
        const uint32_t item = items[item_i];
        const int k = (int)(item >> 12), ci = (int)((item >> 8) & 15u), cj = (int)((item >> 4) & 15u), cn = (int)(item & 15u) + 1;
        // ---- the cluster and its region: every wavefront for itself, lane = hitpoint of the tile (as k_deposit_tiles) ----
        DepTile t;
        if (!dep_load_tile(hp, tp, k, lane, t)) continue;
        unsigned long long remaining[1] = {ballot64(t.live[0])};
        bool inc[1] = {false};
        for (int c = 0; c <= ci && remaining[0] != 0ull; c++) {
            bool in_c[1];
            dep_next_cluster(remaining, c, t, lane, tp.cluster_d, in_c);
            inc[0] = c == ci && in_c[0];
        }
        const DepRegion R = dep_region(t, inc, tp);
        if (R.n_inc == 0) continue;
        const unsigned long long inc_mask = ballot64(inc[0]);
        const double rmax2 = pin_scalar_d(R.rmax * R.rmax);
        const int kx = R.cx1 - R.cx0 + 1, ky = R.cy1 - R.cy0 + 1, kz = R.cz1 - R.cz0 + 1;
        const int Krows = ky * kz;
        const d3 hn_t = inc[0] ? mk(hp.nrm[t.p[0]], hp.nrm[P + t.p[0]], hp.nrm[2 * P + t.p[0]]) : mk(0, 0, 0);
        const double bcx = pin_scalar_d(0.5 * (R.bx0 + R.bx1)), bcy = pin_scalar_d(0.5 * (R.by0 + R.by1)), bcz = pin_scalar_d(0.5 * (R.bz0 + R.bz1));
        const double bex = pin_scalar_d((0.5 * (R.bx1 - R.bx0)) * (1.0 + 1e-12) + 1e-12), bey = pin_scalar_d((0.5 * (R.by1 - R.by0)) * (1.0 + 1e-12) + 1e-12),
                     bez = pin_scalar_d((0.5 * (R.bz1 - R.bz0)) * (1.0 + 1e-12) + 1e-12);
        const int seed = __ffsll((long long)inc_mask) - 1;
        const d3 nbar = mk(rl_d(hn_t.x, seed), rl_d(hn_t.y, seed), rl_d(hn_t.z, seed));
        const double ndev = pin_scalar_d(sqrt(wave_max_d(inc[0] ? len2(hn_t - nbar) : 0.0)) * (1.0 + 1e-9) + 1e-15);
        double E;
        {
            const double tx = t.x[0] - bcx, ty = t.y[0] - bcy, tz = t.z[0] - bcz;
            const double hh_t = fma(tz, tz, fma(ty, ty, tx * tx));
            const double Hm = sqrt(wave_max_d(inc[0] ? hh_t : 0.0)), Qm = sqrt(bex * bex + bey * bey + bez * bez) + R.rmax;
            E = pin_scalar_d(1e-13 * (Hm + Qm) * (Hm + Qm));
        }
        // ---- this lane's hitpoint: number 16 wv + jq of the tile, shared by the four lanes kq = 0..3 ----
        const int src = 16 * wv + jq;
        const bool my_inc = ((inc_mask >> src) & 1ull) != 0ull;
        const double hx = __shfl(t.x[0], src), hy = __shfl(t.y[0], src), hz = __shfl(t.z[0], src), hr2 = __shfl(t.r2[0], src);
        const d3 hn = mk(__shfl(hn_t.x, src), __shfl(hn_t.y, src), __shfl(hn_t.z, src));
        const int my_p = __shfl(t.p[0], src);
        const double lim = fmin(hr2, tp.gather_r2);  // a record is accepted at d2 < gather_r2 && d2 <= r2
        const double lim_lo = my_inc ? lim * (1.0 - 1e-13) : -1.0, lim_hi = my_inc ? lim * (1.0 + 1e-13) : -1.0;
        double bop, lo_c = -INFINITY, hi_c = -INFINITY;  // a hitpoint outside the cluster: never inside, never too close to call
        {
            const double hcx = hx - bcx, hcy = hy - bcy, hcz = hz - bcz;
            const double hh = fma(hcz, hcz, fma(hcy, hcy, hcx * hcx));
            if (my_inc) { lo_c = (lim_lo - hh) - E; hi_c = (lim_hi - hh) + E; }
            bop = kq == 0 ? -2.0 * hcx : (kq == 1 ? -2.0 * hcy : (kq == 2 ? -2.0 * hcz : 1.0));
        }
        // ---- the item's chunk of the region's list ----
        unsigned int v_lo = 0u, v_hi = 0xFFFFFFFFu;
        if (cn > 1) {
            unsigned int T = 0;
            for (int rb = 0; rb < Krows; rb += 64) {
                const int rj = rb + lane;
                uint32_t cnt = 0;
                if (rj < Krows) {
                    const uint32_t b0 = (uint32_t)(((R.cz0 + rj / ky) * tp.g.ny + (R.cy0 + rj % ky)) * tp.g.nx + R.cx0);
                    cnt = cell_start[b0 + kx] - cell_start[b0];
                }
                for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
                T += (unsigned int)__builtin_amdgcn_readfirstlane((int)cnt);
            }
            v_lo = (unsigned int)((unsigned long long)T * (unsigned int)cj / (unsigned int)cn);
            v_hi = (unsigned int)((unsigned long long)T * (unsigned int)(cj + 1) / (unsigned int)cn);
        }
        DwgWalk w;
        w.rg = 0; w.loaded = false; w.done = false; w.r_s = 0; w.r_pre = 0; w.rg_tot = 0; w.g_base = 0; w.voff = 0;
        // a group whose loads are in flight
        bool p_has = false;
        double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0, p8 = 0;
        unsigned long long p_key = 0;
        bool pend = false;
        // find group number `target` (advancing the walk) and start its loads; pend = false if the list ends before it
#define DWG_ISSUE(target)                                                                                              \
        {                                                                                                              \
            pend = false;                                                                                              \
            while (!w.done) {                                                                                          \
                if (!w.loaded) {                                                                                       \
                    if (w.rg * 64 >= Krows || w.voff >= v_hi) { w.done = true; break; }                                \
                    const int rj = w.rg * 64 + lane;                                                                   \
                    uint32_t rs = 0, len = 0;                                                                          \
                    if (rj < Krows) {                                                                                  \
                        const uint32_t b0 = (uint32_t)(((R.cz0 + rj / ky) * tp.g.ny + (R.cy0 + rj % ky)) * tp.g.nx + R.cx0); \
                        rs = cell_start[b0];                                                                           \
                        len = cell_start[b0 + kx] - rs;                                                                \
                    }                                                                                                  \
                    uint32_t incl = len;                                                                               \
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t yy = __shfl_up(incl, off); if (lane >= off) incl += yy; } \
                    const unsigned int tot_u = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);                 \
                    /* clip the row to the chunk */                                                                    \
                    const unsigned int a0 = w.voff + (incl - len), a1 = a0 + len;                                      \
                    const unsigned int c0 = a0 > v_lo ? a0 : v_lo, c1 = a1 < v_hi ? a1 : v_hi;                         \
                    const uint32_t clen = c1 > c0 ? c1 - c0 : 0u;                                                      \
                    w.r_s = rs + (c0 - a0);                                                                            \
                    uint32_t cincl = clen;                                                                             \
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t yy = __shfl_up(cincl, off); if (lane >= off) cincl += yy; } \
                    w.r_pre = cincl - clen;                                                                            \
                    w.rg_tot = (unsigned int)__builtin_amdgcn_readlane((int)cincl, 63);                                \
                    w.voff += tot_u;                                                                                   \
                    w.loaded = true;                                                                                   \
                }                                                                                                      \
                const unsigned int rg_groups = (w.rg_tot + 63u) >> 6;                                                  \
                if ((target) < w.g_base + rg_groups) {                                                                 \
                    const unsigned int i = (((target) - w.g_base) << 6) + (unsigned int)lane;                          \
                    p_has = i < w.rg_tot;                                                                              \
                    int j = 0; /* the row that owns entry i: the largest j with pre_j <= i */                          \
                    for (int step = 32; step >= 1; step >>= 1) {                                                       \
                        const int cand = j + step;                                                                     \
                        const uint32_t pv = __shfl(w.r_pre, cand & 63);                                                \
                        if (cand < 64 && pv <= i) j = cand;                                                            \
                    }                                                                                                  \
                    const uint32_t p_r = __shfl(w.r_s, j) + (i - __shfl(w.r_pre, j));                                  \
                    if (p_has) {                                                                                       \
                        const double2* rp = (const double2*)(rec.f + (size_t)REC_ROW * p_r);  /* one row: sppm_records.hpp */ \
                        const double2 ra = rp[0], rb = rp[1], rc = rp[2], rd = rp[3], re = rp[4];                      \
                        p0 = ra.x; p1 = ra.y; p2 = rb.x; p3 = rb.y; p4 = rc.x; p5 = rc.y; p6 = rd.x; p7 = rd.y; p8 = re.x; \
                        if (TAP) p_key = (unsigned long long)__double_as_longlong(re.y);                               \
                    }                                                                                                  \
                    pend = true;                                                                                       \
                    break;                                                                                             \
                }                                                                                                      \
                w.g_base += rg_groups; w.rg++; w.loaded = false;                                                       \
            }                                                                                                          \
        }
        double ex = 0.0, ey = 0.0, ez = 0.0;
        int em = 0;
        unsigned int rows_unflushed = 0, c_acc = 0, c_sel = 0, c_round = 0;
        unsigned long long c_pairs = 0;
        // the four lanes of a hitpoint added up, the sums to the 64-bit atomics
#define DWG_FLUSH()                                                                                                    \
        {                                                                                                              \
            double vx = ex, vy = ey, vz = ez;                                                                          \
            int vm = em;                                                                                               \
            vx += __shfl_xor(vx, 16); vy += __shfl_xor(vy, 16); vz += __shfl_xor(vz, 16); vm += __shfl_xor(vm, 16);    \
            vx += __shfl_xor(vx, 32); vy += __shfl_xor(vy, 32); vz += __shfl_xor(vz, 32); vm += __shfl_xor(vm, 32);    \
            if (kq == 0 && vm > 0) {                                                                                   \
                atomicAdd((unsigned long long*)hp.dep + my_p, (unsigned long long)(long long)vx);                      \
                atomicAdd((unsigned long long*)hp.dep + (size_t)P + my_p, (unsigned long long)(long long)vy);          \
                atomicAdd((unsigned long long*)hp.dep + 2 * (size_t)P + my_p, (unsigned long long)(long long)vz);      \
                atomicAdd((unsigned long long*)hp.dep + 3 * (size_t)P + my_p, (unsigned long long)vm);                 \
            }                                                                                                          \
            c_acc += kq == 0 ? (unsigned int)vm : 0u;                                                                  \
            ex = ey = ez = 0.0; em = 0; rows_unflushed = 0;                                                            \
        }
#define DWG_ACCEPT(rowp)                                                                                               \
        {                                                                                                              \
            ex += fx; ey += fy; ez += fz; em++;                                                                        \
            if (TAP) {                                                                                                 \
                const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);                                               \
                if ((int64_t)ts < tp.tap_cap) { tp.tap_key[ts] = (unsigned long long)__double_as_longlong((rowp)[DWG_ROW - 1]); tp.tap_pix[ts] = my_p; } \
            }                                                                                                          \
        }
        DWG_ISSUE((unsigned int)wv)
        for (unsigned int n = 0;; n++) {
            // the round exists if its first group does: a wavefront without a group has walked to the end and counted them
            if (!pend && !(w.g_base > 4u * n)) break;
            double (*buf)[DWG_ROW] = s_row[n & 1];
            if (pend) {
                // classify the group (in reach of the cluster; dropped / distance test only / full test: sppm_deposit.hpp) and
                // append it to the round's rows
                bool in = false, easy = false;
                if (p_has) {
                    const double nx = fmax(fmax(R.bx0 - p0, p0 - R.bx1), 0.0), ny = fmax(fmax(R.by0 - p1, p1 - R.by1), 0.0), nz = fmax(fmax(R.bz0 - p2, p2 - R.bz1), 0.0);
                    const double dmin2 = nx * nx + ny * ny + nz * nz;
                    in = dmin2 <= rmax2;
                    if (in) {
                        const double fx_ = fmax(p0 - R.bx0, R.bx1 - p0), fy_ = fmax(p1 - R.by0, R.by1 - p1), fz_ = fmax(p2 - R.bz0, R.bz1 - p2);
                        const double dmax2 = (fx_ * fx_ + fy_ * fy_ + fz_ * fz_) * (1.0 + 1e-12);
                        const double nd = nbar.x * p3 + nbar.y * p4 + nbar.z * p5;                    // a hitpoint's n.g lies within ndev of it
                        const double sc = fabs(p3 * (bcx - p0) + p4 * (bcy - p1) + p5 * (bcz - p2));  // |g.(h - q)| lies within `spread` of it
                        const double spread = fabs(p3) * bex + fabs(p4) * bey + fabs(p5) * bez;
                        const double f_hi = (sc + spread) * (1.0 + 1e-9), f_lo = fmax(sc - spread, 0.0) * (1.0 - 1e-9);
                        const bool rej = nd + ndev < 1e-3 - 1e-9 || f_lo * f_lo > thr2_hi * dmax2 * (1.0 + 1e-9);
                        easy = !rej && nd - ndev > 1e-3 + 1e-9 && dmin2 > 0.0 && f_hi * f_hi < thr2_lo * dmin2 * (1.0 - 1e-9);
                        in = !rej;
                    }
                }
                const unsigned long long m_e = ballot64(in && easy), m_f = ballot64(in && !easy);
                unsigned int base_e = 0, base_f = 0;
                if (lane == 0) {
                    if (m_e) base_e = atomicAdd(&s_cnt[n % 3][0], (unsigned int)__popcll(m_e));
                    if (m_f) base_f = atomicAdd(&s_cnt[n % 3][1], (unsigned int)__popcll(m_f));
                }
                base_e = (unsigned int)__builtin_amdgcn_readfirstlane((int)base_e);
                base_f = (unsigned int)__builtin_amdgcn_readfirstlane((int)base_f);
                if (in) {
                    double* row = buf[easy ? (int)(base_e + mbcnt64(m_e)) : DWG_ROWS - 1 - (int)(base_f + mbcnt64(m_f))];
                    if (easy) {
                        const double qx = p0 - bcx, qy = p1 - bcy, qz = p2 - bcz;
                        row[0] = qx; row[1] = qy; row[2] = qz; row[3] = fma(qz, qz, fma(qy, qy, qx * qx));
                        row[4] = p6; row[5] = p7; row[6] = p8; row[7] = p0; row[8] = p1; row[9] = p2;
                    } else {
                        row[0] = p0; row[1] = p1; row[2] = p2; row[3] = p3; row[4] = p4; row[5] = p5;
                        row[6] = p6; row[7] = p7; row[8] = p8;
                    }
                    if (TAP) row[DWG_ROW - 1] = __longlong_as_double((long long)p_key);
                }
            }
            DWG_ISSUE(4u * (n + 1u) + (unsigned int)wv)
            // the round's rows are staged (and every wavefront is done with the rows of the round before last).  Only LDS is
            // waited for: __syncthreads() would also wait for the loads of the next group, which are meant to stay in flight
            // while this round is processed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int ne = (int)s_cnt[n % 3][0], nf = (int)s_cnt[n % 3][1];
            if (tid == 0) { s_cnt[(n + 2) % 3][0] = 0u; s_cnt[(n + 2) % 3][1] = 0u; }
            if (ne + nf == 0) continue;
            if (rows_unflushed + (unsigned int)(ne + nf) > DWG_FLUSH_ROWS) DWG_FLUSH()
            rows_unflushed += (unsigned int)(ne + nf);
            if (wv == 0) { c_sel += (unsigned int)(ne + nf); c_pairs += (unsigned long long)(ne + nf) * R.n_inc; c_round++; }
            // ---- distance-only rows: 16 at a time on the matrix pipe; the block after is issued before this one is compared ----
            unsigned long long und = 0ull;
            const dep_v4d zero4 = {0.0, 0.0, 0.0, 0.0};
            if (ne > 0) {
                dep_v4d dcur = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[jq][kq], bop, zero4, 0, 0, 0);
                for (int b0 = 0; b0 < ne; b0 += 16) {
                    dep_v4d dnext = zero4;
                    if (b0 + 16 < ne) dnext = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[b0 + 16 + jq][kq], bop, zero4, 0, 0, 0);
                    const bool whole = b0 + 16 <= ne;  // the last block may hang over the end: its rows beyond ne are not rows
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int ri = b0 + kq + 4 * r;  // D row (l >> 4) + 4r
                        const double* fr = buf[ri];
                        const double fx = fr[4], fy = fr[5], fz = fr[6];
                        const double cv = dcur[r];
                        const bool is_row = whole || ri < ne;
                        const bool ok = is_row && cv < lo_c;
                        und |= ballot64(is_row && !ok && !(cv > hi_c));
                        if (ok) DWG_ACCEPT(fr)
                    }
                    dcur = dnext;
                }
            }
            if (und != 0ull) {
                // (once in ~10^9 pairs) the pairs inside the band: decided from the positions themselves, in the reference's
                // operation order.  The same MFMA on the same operands gives the same value, so these are exactly the pairs
                // left out above.
                for (int b0 = 0; b0 < ne; b0 += 16) {
                    const dep_v4d dv = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[b0 + jq][kq], bop, zero4, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int ri = b0 + kq + 4 * r;
                        const double cv = dv[r];
                        if (ri < ne && !(cv < lo_c) && !(cv > hi_c)) {
                            const double* fr = buf[ri];
                            const double e = len2(mk(hx, hy, hz) - mk(fr[7], fr[8], fr[9]));
                            if (my_inc && (e < tp.gather_r2) && (e <= hr2)) {
                                const double fx = fr[4], fy = fr[5], fz = fr[6];
                                DWG_ACCEPT(fr)
                            }
                        }
                    }
                }
            }
            // ---- full-test rows: lane (jq, kq) tests its hitpoint against rows kq, kq + 4, ... (from the top down) ----
            for (int t0 = 0; t0 < nf; t0 += 4) {
                const int ti = t0 + kq;
                const bool is_row = ti < nf;
                const double* row = buf[DWG_ROWS - 1 - (is_row ? ti : 0)];
                const d3 g0 = mk(row[3], row[4], row[5]);
                const d3 qq = mk(row[0], row[1], row[2]);
                const d3 dir = mk(hx, hy, hz) - qq;
                bool ok, unsure;
                {   // fast path (fused), bounds 1e-13 tighter and looser than the real ones
                    const double d2f = fma(dir.z, dir.z, fma(dir.y, dir.y, dir.x * dir.x));
                    const double ndf = fma(hn.z, g0.z, fma(hn.y, g0.y, hn.x * g0.x));
                    const double s1f = fma(g0.z, dir.z, fma(g0.y, dir.y, g0.x * dir.x));
                    const double lhs = s1f * s1f;
                    const bool in_d = d2f < lim_lo, out_d = d2f > lim_hi;     // both false: too close to call
                    const bool in_n = ndf > 1e-3 + 1e-13, out_n = ndf < 1e-3 - 1e-13;
                    const bool in_f = lhs < thr2_lo * d2f, out_f = lhs > thr2_hi * d2f;
                    ok = is_row && in_d && in_n && in_f;
                    unsure = is_row && my_inc && !(out_d || out_n || out_f) && !ok;
                }
                if (ballot64(unsure) != 0ull) {
                    // the reference's operation order (every lane of the step: the result replaces the fast one)
                    const double d2 = len2(dir);
                    const bool o = is_row && my_inc && (d2 < tp.gather_r2) && (d2 <= hr2) && (dot(hn, g0) > 1e-3);
                    const double s1 = dot(g0, dir);
                    const double l2 = s1 * s1, rhs = thr * thr * d2;
                    const double band = rhs * band_rel;
                    bool flat = l2 < rhs - band;
                    if (o && !flat && !(l2 > rhs + band)) {
                        const double dt = dot(g0, dir / sqrt(d2));  // NaN at d2 == 0: rejected as in the reference
                        flat = fabs(dt) <= thr;
                    }
                    ok = o && flat;
                }
                if (ok) {
                    const double fx = row[6], fy = row[7], fz = row[8];
                    DWG_ACCEPT(row)
                }
            }
        }
#undef DWG_ISSUE
#undef DWG_ACCEPT
        DWG_FLUSH()
#undef DWG_FLUSH
        for (int off = 32; off > 0; off >>= 1) c_acc += __shfl_down(c_acc, off);
        if (lane == 0) {
            if (c_pairs) atomicAdd(&counters[3], c_pairs);
            if (c_acc) atomicAdd(&counters[4], (unsigned long long)c_acc);
            if (c_sel) atomicAdd(&counters[5], (unsigned long long)c_sel);
            if (c_round) atomicAdd(&counters[6], (unsigned long long)c_round);
            if (wv == 0) atomicAdd(&counters[7], 1ull);
        }
    }
}
