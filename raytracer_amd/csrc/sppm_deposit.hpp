// sppm_deposit.hpp -- the per-pass rule's own gather (update_rule 1, src/stochastic_ppm.h:439-456 + :721-741).
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
//
// Under the per-pass rule a hitpoint's radius is constant while a pass's photons are deposited, so which photon
// hits a hitpoint accepts does not depend on their order, and what it accumulates is a plain sum: the photon flux
// (the reference's multiply(weight, f) with f = colour / pi, which is the same for every deposit of a hitpoint and
// is applied once at the end of the pass) and the count M.  Both are kept as 64-bit INTEGERS (fixed point, SURVEY.md
// 8e): integer sums are exact, so the result does not depend on the order of the records, on how the work is cut,
// or on how many GPUs deposit -- one all-reduce of dep[4][P] per pass combines the ranks (sppm_export_deposits /
// sppm_import_deposits), and every rank then applies the same end-of-pass update (k_end_pass).
//
// So there is no key ranking, no slab of time and no serial walk here (the ordered k_gather_tiles needs all three
// for the per-photon rule): the records of a batch are binned by cell only; a work item is (tile of 64 hitpoints,
// cluster of its lanes, chunk of the records in the cluster's region); a wavefront stages the chunk's in-reach
// records through LDS, 64 at a time, and every lane tests its hitpoint against each staged record (an LDS
// broadcast) and adds the record's pre-converted flux.  Several items add to the same hitpoint with integer atomics.
#pragma once

// (fx_fix, the fixed point of the deposits, is defined in sppm_gather_simple.hpp, which also uses it)

// work item: tile << 12 | cluster << 8 | chunk << 4 | (chunks - 1)
#define DEP_MAX_CLUSTERS 16
#define DEP_MAX_CHUNKS 16
__device__ __forceinline__ uint32_t pack_dep_item(int tile, int cluster, int chunk, int chunks)
{
    return ((uint32_t)tile << 12) | ((uint32_t)cluster << 8) | ((uint32_t)chunk << 4) | (uint32_t)(chunks - 1);
}

// The lanes of cluster `c` of a tile, same rule in the plan and in the deposit: the first remaining lane is the
// seed, lanes within cluster_d (max norm) of it join; the last cluster takes whatever is left.
__device__ __forceinline__ bool dep_next_cluster(unsigned long long& remaining, int c, bool live, int lane, double x, double y, double z, double cluster_d)
{
    const int seed = __ffsll((long long)remaining) - 1;
    const double sx = rl_d(x, seed), sy = rl_d(y, seed), sz = rl_d(z, seed);
    const bool inc = live && ((remaining >> lane) & 1ull) &&
                     (c == DEP_MAX_CLUSTERS - 1 || (fabs(x - sx) <= cluster_d && fabs(y - sy) <= cluster_d && fabs(z - sz) <= cluster_d));
    remaining &= ~__ballot(inc);
    return inc;
}

// One wavefront per tile: the clusters, the records in each cluster's region (8 lookups in the summed-area
// table), and the cut of every region's records into at most DEP_MAX_CHUNKS chunks of ~chunk_records.  Tile k
// writes its pieces to raw[k*64 ..] as (cost estimate << 32 | item) and their number (<= 64) to n_pieces[k].
__global__ void __launch_bounds__(256) k_deposit_plan(HpArrays hp, const uint32_t* __restrict__ sat, TileParams tp, uint32_t chunk_records,
                                                      unsigned long long* __restrict__ raw, uint8_t* __restrict__ n_pieces, uint32_t* __restrict__ est)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wv;
    if (k >= tp.n_owned) return;
    const int slot = k * 64 + lane;
    const int P = hp.P;
    const int p = slot < tp.n_pix ? tp.order[slot] : 0;
    bool live = slot < tp.n_pix;
    if (live) live = hp.valid[p] != 0;
    if (__ballot(live) == 0) { if (lane == 0) { est[k] = 0; n_pieces[k] = 0; } return; }
    const double BIG = 1e300;
    const double x = live ? hp.pos[p] : 0.0, y = live ? hp.pos[P + p] : 0.0, z = live ? hp.pos[2 * P + p] : 0.0;
    const double rr = live ? sqrt(fmin(hp.r2[p], tp.gather_r2)) * (1.0 + 1e-9) + 1e-9 : 0.0;
    unsigned long long remaining = __ballot(live);
    int n_out = 0;
    unsigned int total = 0;
    for (int c = 0; remaining; c++) {
        const bool inc = dep_next_cluster(remaining, c, live, lane, x, y, z, tp.cluster_d);
        const double rm = wave_max_d(inc ? rr : 0.0);
        const int cx0 = grid_coord(wave_min_d(inc ? x : BIG) - rm, tp.g.ox, tp.g.inv_c, tp.g.nx), cx1 = grid_coord(wave_max_d(inc ? x : -BIG) + rm, tp.g.ox, tp.g.inv_c, tp.g.nx);
        const int cy0 = grid_coord(wave_min_d(inc ? y : BIG) - rm, tp.g.oy, tp.g.inv_c, tp.g.ny), cy1 = grid_coord(wave_max_d(inc ? y : -BIG) + rm, tp.g.oy, tp.g.inv_c, tp.g.ny);
        const int cz0 = grid_coord(wave_min_d(inc ? z : BIG) - rm, tp.g.oz, tp.g.inv_c, tp.g.nz), cz1 = grid_coord(wave_max_d(inc ? z : -BIG) + rm, tp.g.oz, tp.g.inv_c, tp.g.nz);
        const unsigned int T = sat_box(sat, tp.g.nx, tp.g.ny, cx0, cx1, cy0, cy1, cz0, cz1, lane);
        const unsigned int cells = (unsigned int)((cx1 - cx0 + 1) * (cy1 - cy0 + 1) * (cz1 - cz0 + 1));
        // every later cluster (at most DEP_MAX_CLUSTERS - 1 - c of them) keeps one of the tile's 64 slots
        const int avail = 64 - n_out - (DEP_MAX_CLUSTERS - 1 - c);
        int cn = (int)((T + chunk_records - 1) / chunk_records);
        cn = cn < 1 ? 1 : cn;
        cn = cn > DEP_MAX_CHUNKS ? DEP_MAX_CHUNKS : cn;
        cn = cn > avail ? avail : cn;
        // a piece stages its share of the records and walks all the region's cells (twice)
        const unsigned int cost = T / (unsigned int)cn + 16u + (cells >> 3);
        if (lane < cn) raw[(size_t)k * 64 + n_out + lane] = ((unsigned long long)cost << 32) | pack_dep_item(k, c, lane, cn);
        n_out += cn;
        total += T + 16u + (cells >> 3);
    }
    if (lane == 0) { est[k] = total; n_pieces[k] = (uint8_t)n_out; }
}

template <bool TAP>
__global__ void __launch_bounds__(256) k_deposit_tiles(HpArrays hp, RecArrays rec, const uint32_t* __restrict__ cell_start, TileParams tp, double fx_scale,
                                                       const uint32_t* __restrict__ items, unsigned int* __restrict__ q,
                                                       unsigned long long* __restrict__ counters, int32_t* __restrict__ err)
{
    __shared__ double s_row[4][64][10];  // a staged record: position, geometric normal, fixed-point flux (3 x int64 bits), key bits
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int P = hp.P;
    const double BIG = 1e300;
    const double thr = tp.gather_r2 * 0.01;
    const double band_rel = 1e-9 + 8e-15 / thr;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t cap = rec.cap;
    const unsigned int n_items = q[0];
    const unsigned int n_waves = gridDim.x * 4u;
    for (bool first_item = true;; first_item = false) {
    // persistent wavefronts over the list of items, heaviest first (as k_gather_tiles)
    unsigned int item_i = blockIdx.x * 4u + (unsigned int)wv;
    if (!first_item) {
        if (lane == 0) item_i = n_waves + atomicAdd(&q[1], 1u);
        item_i = (unsigned int)__builtin_amdgcn_readfirstlane((int)item_i);
    }
    if (item_i >= n_items) break;
    const uint32_t item = items[item_i];
    const int k = (int)(item >> 12), ci = (int)((item >> 8) & 15u), cj = (int)((item >> 4) & 15u), cn = (int)(item & 15u) + 1;
    const int slot = k * 64 + lane;
    const int p = slot < tp.n_pix ? tp.order[slot] : 0;
    bool live = slot < tp.n_pix;
    if (live) live = hp.valid[p] != 0;
    if (__ballot(live) == 0) continue;
    d3 hpos = mk(0, 0, 0), hn = mk(0, 0, 0);
    double r2 = 0.0;
    if (live) {
        hpos = mk(hp.pos[p], hp.pos[P + p], hp.pos[2 * P + p]);
        hn = mk(hp.nrm[p], hp.nrm[P + p], hp.nrm[2 * P + p]);
        r2 = hp.r2[p];
    }
    // the lanes of cluster ci
    unsigned long long remaining = __ballot(live);
    bool inc = false;
    for (int c = 0; c <= ci && remaining; c++) {
        const bool in_c = dep_next_cluster(remaining, c, live, lane, hpos.x, hpos.y, hpos.z, tp.cluster_d);
        inc = c == ci && in_c;
    }
    const unsigned int n_inc = (unsigned int)__popcll(__ballot(inc));
    if (n_inc == 0) continue;
    // region: the cluster's bounding box grown by its largest radius
    const double rr = inc ? sqrt(fmin(r2, tp.gather_r2)) * (1.0 + 1e-9) + 1e-9 : 0.0;
    const double rmax = wave_max_d(rr);
    const double rmax2 = rmax * rmax;
    const double bx0 = wave_min_d(inc ? hpos.x : BIG), bx1 = wave_max_d(inc ? hpos.x : -BIG);
    const double by0 = wave_min_d(inc ? hpos.y : BIG), by1 = wave_max_d(inc ? hpos.y : -BIG);
    const double bz0 = wave_min_d(inc ? hpos.z : BIG), bz1 = wave_max_d(inc ? hpos.z : -BIG);
    const int cx0 = __builtin_amdgcn_readfirstlane(grid_coord(bx0 - rmax, tp.g.ox, tp.g.inv_c, tp.g.nx));
    const int cx1 = __builtin_amdgcn_readfirstlane(grid_coord(bx1 + rmax, tp.g.ox, tp.g.inv_c, tp.g.nx));
    const int cy0 = __builtin_amdgcn_readfirstlane(grid_coord(by0 - rmax, tp.g.oy, tp.g.inv_c, tp.g.ny));
    const int cy1 = __builtin_amdgcn_readfirstlane(grid_coord(by1 + rmax, tp.g.oy, tp.g.inv_c, tp.g.ny));
    const int cz0 = __builtin_amdgcn_readfirstlane(grid_coord(bz0 - rmax, tp.g.oz, tp.g.inv_c, tp.g.nz));
    const int cz1 = __builtin_amdgcn_readfirstlane(grid_coord(bz1 + rmax, tp.g.oz, tp.g.inv_c, tp.g.nz));
    const int kx = cx1 - cx0 + 1, ky = cy1 - cy0 + 1, kz = cz1 - cz0 + 1;
    const int Ktot = kx * ky * kz;
    // the region's records form one virtual list (cells in q order): its length, then this item's share of it
    unsigned int T = 0;
    for (int qbase = 0; qbase < Ktot; qbase += 64) {
        const int qi = qbase + lane;
        uint32_t cnt = 0;
        if (qi < Ktot) {
            const int ix = cx0 + qi % kx, iy = cy0 + (qi / kx) % ky, iz = cz0 + qi / (kx * ky);
            const uint32_t b = (uint32_t)((iz * tp.g.ny + iy) * tp.g.nx + ix);
            cnt = cell_start[b + 1] - cell_start[b];
        }
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
        T += (unsigned int)__builtin_amdgcn_readfirstlane((int)cnt);
    }
    const unsigned int v_lo = (unsigned int)((unsigned long long)T * (unsigned int)cj / (unsigned int)cn);
    const unsigned int v_hi = (unsigned int)((unsigned long long)T * (unsigned int)(cj + 1) / (unsigned int)cn);
    long long ax = 0, ay = 0, az = 0;
    int am = 0;
    unsigned int c_sel = 0, c_round = 0;
    unsigned long long c_pairs = 0;
    bool clamped = false;
    int sel_n = 0;
    // every lane against the sel_n staged records: the search filter of src/kdtree.h:123-127 (its |cos| does not
    // depend on the orientation of the normal) and the acceptance test of src/stochastic_ppm.h:441-445
#define DEP_PROCESS()                                                                                                  \
    {                                                                                                                  \
        __builtin_amdgcn_wave_barrier();                                                                               \
        c_sel += (unsigned int)sel_n; c_pairs += (unsigned long long)sel_n * n_inc; c_round++;                         \
        const int sel_u = __builtin_amdgcn_readfirstlane(sel_n);                                                        \
        for (int t = 0; t < sel_u; t++) {                                                                              \
            const double* row = s_row[wv][t];                                                                          \
            const d3 g0 = mk(row[3], row[4], row[5]);                                                                  \
            const d3 dir = hpos - mk(row[0], row[1], row[2]);                                                          \
            const double d2 = len2(dir);                                                                               \
            bool ok = inc && (d2 < tp.gather_r2) && (d2 <= r2) && (dot(hn, g0) > 1e-3);                                \
            if (__ballot(ok) == 0) continue;                                                                           \
            {                                                                                                          \
                const double s1 = dot(g0, dir);                                                                        \
                const double lhs = s1 * s1, rhs = thr * thr * d2;                                                      \
                const double band = rhs * band_rel;                                                                    \
                const bool sure_pass = lhs < rhs - band, sure_fail = lhs > rhs + band;                                 \
                bool flat = sure_pass;                                                                                 \
                const bool unsure = ok && !sure_pass && !sure_fail;                                                    \
                if (__ballot(unsure) != 0) {                                                                           \
                    if (unsure) {                                                                                      \
                        const double dt = dot(g0, dir / sqrt(d2)); /* NaN at d2 == 0: rejected as in the reference */ \
                        flat = fabs(dt) <= thr;                                                                        \
                    }                                                                                                  \
                }                                                                                                      \
                ok = ok && flat;                                                                                       \
            }                                                                                                          \
            if (ok) {                                                                                                  \
                ax += __double_as_longlong(row[6]); ay += __double_as_longlong(row[7]); az += __double_as_longlong(row[8]); \
                am++;                                                                                                  \
                if (TAP) {                                                                                             \
                    const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);                                           \
                    if ((int64_t)ts < tp.tap_cap) { tp.tap_key[ts] = (unsigned long long)__double_as_longlong(row[9]); tp.tap_pix[ts] = p; } \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_wave_barrier();                                                                               \
        sel_n = 0;                                                                                                     \
    }
    unsigned int voff = 0;
    for (int qbase = 0; qbase < Ktot && voff < v_hi; qbase += 64) {
        const int qi = qbase + lane;
        uint32_t cs = 0, cnt = 0;
        if (qi < Ktot) {
            const int ix = cx0 + qi % kx, iy = cy0 + (qi / kx) % ky, iz = cz0 + qi / (kx * ky);
            const uint32_t b = (uint32_t)((iz * tp.g.ny + iy) * tp.g.nx + ix);
            cs = cell_start[b];
            cnt = cell_start[b + 1] - cs;
        }
        uint32_t incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t yy = __shfl_up(incl, off); if (lane >= off) incl += yy; }
        const uint32_t pre = incl - cnt;
        const unsigned int tot = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
        const unsigned int g_lo = voff, g_hi = voff + tot;
        voff = g_hi;
        if (g_hi <= v_lo || g_lo >= v_hi) continue;
        const unsigned int i_begin = (v_lo > g_lo ? v_lo : g_lo) - g_lo, i_end = (v_hi < g_hi ? v_hi : g_hi) - g_lo;
        for (unsigned int cbase = i_begin; cbase < i_end; cbase += 64) {
            const unsigned int i = cbase + (unsigned int)lane;
            const bool has = i < i_end;
            int j = 0;  // the cell lane that owns list entry i: the largest j with pre_j <= i
            for (int step = 32; step >= 1; step >>= 1) {
                const int cand = j + step;
                const uint32_t pv = __shfl(pre, cand & 63);
                if (cand < 64 && pv <= i) j = cand;
            }
            const uint32_t r = __shfl(cs, j) + (i - __shfl(pre, j));
            bool in = false;
            double qx = 0.0, qy = 0.0, qz = 0.0;
            if (has) {
                qx = rec.f[r]; qy = rec.f[cap + r]; qz = rec.f[2 * cap + r];
                // within rmax of the cluster's bounding box (a hitpoint of the cluster can only accept such records)
                const double ex = fmax(fmax(bx0 - qx, qx - bx1), 0.0), ey = fmax(fmax(by0 - qy, qy - by1), 0.0), ez = fmax(fmax(bz0 - qz, qz - bz1), 0.0);
                in = ex * ex + ey * ey + ez * ez <= rmax2;
            }
            const unsigned long long m = __ballot(in);
            const int cnt_in = __popcll(m);
            if (sel_n + cnt_in > 64) DEP_PROCESS()
            if (in) {
                double* row = s_row[wv][sel_n + __popcll(m & lt_mask)];
                row[0] = qx; row[1] = qy; row[2] = qz;
                row[3] = rec.f[3 * cap + r]; row[4] = rec.f[4 * cap + r]; row[5] = rec.f[5 * cap + r];
                row[6] = __longlong_as_double(fx_fix(rec.f[6 * cap + r], fx_scale, clamped));
                row[7] = __longlong_as_double(fx_fix(rec.f[7 * cap + r], fx_scale, clamped));
                row[8] = __longlong_as_double(fx_fix(rec.f[8 * cap + r], fx_scale, clamped));
                if (TAP) row[9] = __longlong_as_double((long long)rec.key[r]);
            }
            sel_n += cnt_in;
        }
    }
    if (sel_n > 0) DEP_PROCESS()
#undef DEP_PROCESS
    if (am > 0) {
        atomicAdd((unsigned long long*)hp.dep + p, (unsigned long long)ax);
        atomicAdd((unsigned long long*)hp.dep + (size_t)P + p, (unsigned long long)ay);
        atomicAdd((unsigned long long*)hp.dep + 2 * (size_t)P + p, (unsigned long long)az);
        atomicAdd((unsigned long long*)hp.dep + 3 * (size_t)P + p, (unsigned long long)am);
    }
    if (__ballot(clamped) != 0 && lane == 0) err[6] = 1;
    unsigned int c_acc = (unsigned int)am;
    for (int off = 32; off > 0; off >>= 1) c_acc += __shfl_down(c_acc, off);
    if (lane == 0) {
        if (c_pairs) atomicAdd(&counters[3], c_pairs);
        if (c_acc) atomicAdd(&counters[4], (unsigned long long)c_acc);
        if (c_sel) atomicAdd(&counters[5], (unsigned long long)c_sel);
        if (c_round) atomicAdd(&counters[6], (unsigned long long)c_round);
        atomicAdd(&counters[7], 1ull);
    }
    __builtin_amdgcn_wave_barrier();
    }  // next item
}

// End-of-pass update of the per-pass rule, src/stochastic_ppm.h:721-741, from the (summed) deposits; clears them.
__global__ void k_end_pass(HpArrays hp, DevScene sc, Own own, double alpha, double fx_inv, int32_t* __restrict__ err)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t P = (size_t)hp.P;
    if (p >= hp.P) return;
    if (!hp.valid[p] || !owned(own, p)) return;
    const long long madd = hp.dep[3 * P + p];
    if (madd > 0) {
        if (madd >= FX_MAX_DEPOSITS) err[6] = 1;
        const double N = hp.nd[p], r2 = hp.r2[p];
        const double Nnew = N + alpha * (double)madd;
        const double Rnew2 = r2 * Nnew / (N + (double)madd);
        const d3 tau = mk(hp.tau[p], hp.tau[P + p], hp.tau[2 * P + p]);
        // accumulated_flux = sum of multiply(weight, f), f = colour / pi (:449-453): the sum of the weights times f
        const d3 phi = had(mk((double)hp.dep[p] * fx_inv, (double)hp.dep[P + p] * fx_inv, (double)hp.dep[2 * P + p] * fx_inv), ld3(sc.sp[hp.obj[p]].color) / SPPM_PI);
        const d3 w = mk(hp.w[p], hp.w[P + p], hp.w[2 * P + p]);
        const d3 tn = (tau + had(phi, w)) * Rnew2 / r2;
        hp.tau[p] = tn.x; hp.tau[P + p] = tn.y; hp.tau[2 * P + p] = tn.z;
        hp.nd[p] = Nnew; hp.r2[p] = Rnew2;
        hp.dep[p] = 0; hp.dep[P + p] = 0; hp.dep[2 * P + p] = 0; hp.dep[3 * P + p] = 0;
    }
    hp.w[p] = 0.0; hp.w[P + p] = 0.0; hp.w[2 * P + p] = 0.0;
}
