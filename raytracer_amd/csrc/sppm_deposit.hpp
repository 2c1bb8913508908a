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

// (fx_fix, the fixed point of the deposits, is applied by k_grid_scatter, sppm_records.hpp: the flux planes this kernel reads hold its bits)

// A wavefront holds DEP_H hitpoints per lane: a tile of the deposit is 64 * DEP_H consecutive hitpoints of the
// Hilbert order (lane l holds entries l, 64 + l, ...).  Every staged record is read from LDS once per wavefront
// (a broadcast of 72 bytes, which costs the same LDS cycles whether one lane or 64 want it), so with one hitpoint
// per lane the kernel was bound by the LDS pipe that the four SIMDs of a CU share; two hitpoints per lane halve
// the LDS reads per (record, hitpoint) pair and leave it bound by the tests' arithmetic.
#ifndef DEP_H
#define DEP_H 1  // 2 was measured: 18 % more pair tests per second, but tiles of 128 hitpoints test a third more pairs per update
#endif
#define DEP_TILE (64 * DEP_H)
// work item: tile << 12 | cluster << 8 | chunk << 4 | (chunks - 1)
#define DEP_MAX_CLUSTERS 16
#define DEP_MAX_CHUNKS 16
__device__ __forceinline__ uint32_t pack_dep_item(int tile, int cluster, int chunk, int chunks)
{
    return ((uint32_t)tile << 12) | ((uint32_t)cluster << 8) | ((uint32_t)chunk << 4) | (uint32_t)(chunks - 1);
}
// ballot without the bool -> int -> compare round trip of __ballot
__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// The hitpoints of a deposit tile, DEP_H per lane, and what the plan and the deposit both derive from them.
struct DepTile {
    int p[DEP_H]; bool live[DEP_H];
    double x[DEP_H], y[DEP_H], z[DEP_H], r2[DEP_H];
};
struct DepRegion;
__device__ __forceinline__ double dep_box_dmin2(const DepRegion& R, double p0, double p1, double p2);
__device__ __forceinline__ bool dep_load_tile(const HpArrays& hp, const TileParams& tp, int k, int lane, DepTile& t)
{
    const int P = hp.P;
    bool any = false;
#pragma unroll
    for (int h = 0; h < DEP_H; h++) {
        const int slot = k * DEP_TILE + h * 64 + lane;
        t.p[h] = slot < tp.n_pix ? tp.order[slot] : 0;
        t.live[h] = slot < tp.n_pix;
        if (t.live[h]) t.live[h] = hp.valid[t.p[h]] != 0;
        t.x[h] = t.live[h] ? hp.pos[t.p[h]] : 0.0; t.y[h] = t.live[h] ? hp.pos[P + t.p[h]] : 0.0; t.z[h] = t.live[h] ? hp.pos[2 * P + t.p[h]] : 0.0;
        t.r2[h] = t.live[h] ? hp.r2[t.p[h]] : 0.0;
        any = any || t.live[h];
    }
    return ballot64(any) != 0ull;
}
// The hitpoints of cluster `c` of a tile, same rule in the plan and in the deposit: the first remaining hitpoint
// (lowest h, then lowest lane) is the seed, hitpoints within cluster_d (max norm) of it join; the last cluster
// takes whatever is left.  remaining[h] = ballot of the hitpoints not yet in a cluster.
// (Round 3: clusters also split by normal -- a tile at an edge of the scene holds hitpoints of two surfaces, and as one
// cluster none of its staged records can be classified for all hitpoints at once, 40 % of all full-test records at
// 1080p -- halved the full-test records and the kernel's time by 1 %, and gave it back in the plan, which then loads
// the normals too: 269M photons/s against 271M without.  Dropped.)
__device__ __forceinline__ void dep_next_cluster(unsigned long long (&remaining)[DEP_H], int c, const DepTile& t, int lane, double cluster_d, bool (&inc)[DEP_H])
{
    int sh = 0;
#pragma unroll
    for (int h = DEP_H - 1; h >= 0; h--) if (remaining[h]) sh = h;
    unsigned long long rs = remaining[0];
    double xs = t.x[0], ys = t.y[0], zs = t.z[0];
#pragma unroll
    for (int h = 1; h < DEP_H; h++) if (sh == h) { rs = remaining[h]; xs = t.x[h]; ys = t.y[h]; zs = t.z[h]; }
    const int seed = __ffsll((long long)rs) - 1;
    const double sx = rl_d(xs, seed), sy = rl_d(ys, seed), sz = rl_d(zs, seed);
#pragma unroll
    for (int h = 0; h < DEP_H; h++) {
        inc[h] = t.live[h] && ((remaining[h] >> lane) & 1ull) &&
                 (c == DEP_MAX_CLUSTERS - 1 || (fabs(t.x[h] - sx) <= cluster_d && fabs(t.y[h] - sy) <= cluster_d && fabs(t.z[h] - sz) <= cluster_d));
        remaining[h] &= ~ballot64(inc[h]);
    }
}
__device__ __forceinline__ bool dep_any(const unsigned long long (&m)[DEP_H])
{
    unsigned long long a = 0;
#pragma unroll
    for (int h = 0; h < DEP_H; h++) a |= m[h];
    return a != 0ull;
}
// region of a cluster: its bounding box and its largest radius (grown a little: the cell range must cover every
// record a hitpoint of the cluster can accept)
struct DepRegion { double bx0, bx1, by0, by1, bz0, bz1, rmax; int cx0, cx1, cy0, cy1, cz0, cz1; unsigned int n_inc; };
__device__ __forceinline__ double dep_box_dmin2(const DepRegion& R, double p0, double p1, double p2)
{
    const double nx = fmax(fmax(R.bx0 - p0, p0 - R.bx1), 0.0), ny = fmax(fmax(R.by0 - p1, p1 - R.by1), 0.0), nz = fmax(fmax(R.bz0 - p2, p2 - R.bz1), 0.0);
    return nx * nx + ny * ny + nz * nz;
}
__device__ __forceinline__ DepRegion dep_region(const DepTile& t, const bool (&inc)[DEP_H], const TileParams& tp)
{
    const double BIG = 1e300;
    double lx = BIG, hx = -BIG, ly = BIG, hy = -BIG, lz = BIG, hz = -BIG, rr = 0.0;
    unsigned int n = 0;
#pragma unroll
    for (int h = 0; h < DEP_H; h++) {
        if (inc[h]) {
            lx = fmin(lx, t.x[h]); hx = fmax(hx, t.x[h]); ly = fmin(ly, t.y[h]); hy = fmax(hy, t.y[h]); lz = fmin(lz, t.z[h]); hz = fmax(hz, t.z[h]);
            rr = fmax(rr, sqrt(fmin(t.r2[h], tp.gather_r2)) * (1.0 + 1e-9) + 1e-9);
        }
        n += (unsigned int)__popcll(ballot64(inc[h]));
    }
    DepRegion R;
    R.n_inc = n;
    R.rmax = wave_max_d(rr);
    R.bx0 = wave_min_d(lx); R.bx1 = wave_max_d(hx); R.by0 = wave_min_d(ly); R.by1 = wave_max_d(hy); R.bz0 = wave_min_d(lz); R.bz1 = wave_max_d(hz);
    R.cx0 = __builtin_amdgcn_readfirstlane(grid_coord(R.bx0 - R.rmax, tp.g.ox, tp.g.inv_c, tp.g.nx));
    R.cx1 = __builtin_amdgcn_readfirstlane(grid_coord(R.bx1 + R.rmax, tp.g.ox, tp.g.inv_c, tp.g.nx));
    R.cy0 = __builtin_amdgcn_readfirstlane(grid_coord(R.by0 - R.rmax, tp.g.oy, tp.g.inv_c, tp.g.ny));
    R.cy1 = __builtin_amdgcn_readfirstlane(grid_coord(R.by1 + R.rmax, tp.g.oy, tp.g.inv_c, tp.g.ny));
    R.cz0 = __builtin_amdgcn_readfirstlane(grid_coord(R.bz0 - R.rmax, tp.g.oz, tp.g.inv_c, tp.g.nz));
    R.cz1 = __builtin_amdgcn_readfirstlane(grid_coord(R.bz1 + R.rmax, tp.g.oz, tp.g.inv_c, tp.g.nz));
    return R;
}

// One wavefront per tile: the clusters, the records in each cluster's region, and the cut of every region's records into at most DEP_MAX_CHUNKS chunks of ~chunk_records.  Tile k
// writes its pieces to raw[k*64 ..] as (cost estimate << 32 | item) and their number (<= 64) to n_pieces[k].
// Round 3: it also does what three more launches between two deposits used to do -- the tile's extent bin for the grid
// statistics (k_tile_extent: the same hitpoints were loaded and reduced a second time there) and the clearing of the item
// queue and of the statistics that k_items_count / k_items_place (below) fill in.
__global__ void __launch_bounds__(256) k_deposit_plan(HpArrays hp, const uint32_t* __restrict__ cell_start, TileParams tp, int n_tiles, uint32_t chunk_records,
                                                      unsigned long long* __restrict__ raw, uint8_t* __restrict__ n_pieces, uint32_t* __restrict__ est,
                                                      unsigned int* __restrict__ q, uint8_t* __restrict__ ext_bin, TileStats* __restrict__ st)
{
    // (between two gathers the next gather waits for this kernel while the eye pass of the pass after runs beside it:
    // its wavefronts go first where they share a SIMD)
    __builtin_amdgcn_s_setprio(3);
    // (the item queue of the batch before is done with, and so are the statistics: cleared here, ahead of the kernels
    // that count into them)
    if (blockIdx.x == 0) {
        if (threadIdx.x < 2 + 2 * ITEM_CLASSES) q[threadIdx.x] = 0u;
        if (threadIdx.x < 64) st->hist[threadIdx.x] = 0u;
        if (threadIdx.x == 64) { st->live_tiles = 0u; st->pad = 0u; st->est_sum = 0ull; }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wv;
    if (k >= n_tiles) return;
    DepTile t;
    if (!dep_load_tile(hp, tp, k, lane, t)) { if (lane == 0) { est[k] = 0; n_pieces[k] = 0; ext_bin[k] = 0; } return; }
    unsigned long long remaining[DEP_H];
#pragma unroll
    for (int h = 0; h < DEP_H; h++) remaining[h] = ballot64(t.live[h]);
    int n_out = 0;
    unsigned int total = 0;
    double tile_ext = -1.0;
    for (int c = 0; dep_any(remaining); c++) {
        bool inc[DEP_H];
        dep_next_cluster(remaining, c, t, lane, tp.cluster_d, inc);
        const DepRegion R = dep_region(t, inc, tp);
        // a tile that is one cluster (nearly all are): its extent is this region's
        if (c == 0 && !dep_any(remaining)) tile_ext = fmax(R.bx1 - R.bx0, fmax(R.by1 - R.by0, R.bz1 - R.bz0)) + 2.0 * R.rmax;
        // the records in the region's cells, counted directly: one slab, and the cells of a row (x) are consecutive in the
        // table, so a row's count is the difference of two starts -- one lane per (y, z) row, usually one trip
        const int kx = R.cx1 - R.cx0 + 1, ky = R.cy1 - R.cy0 + 1, kz = R.cz1 - R.cz0 + 1;
        const unsigned int cells = (unsigned int)(kx * ky * kz);
        const int rows = ky * kz;
        unsigned int T = 0;
        for (int rb = 0; rb < rows; rb += 64) {
            const int rj = rb + lane;
            if (rj < rows) {
                const uint32_t b0 = (uint32_t)(((R.cz0 + rj / ky) * tp.g.ny + (R.cy0 + rj % ky)) * tp.g.nx + R.cx0);
                T += cell_start[b0 + kx] - cell_start[b0];
            }
        }
        for (int off = 32; off > 0; off >>= 1) T += __shfl_xor(T, off);
        T = (unsigned int)__builtin_amdgcn_readfirstlane((int)T);
        // every later cluster (at most DEP_MAX_CLUSTERS - 1 - c of them) keeps one of the tile's 64 slots
        const int avail = 64 - n_out - (DEP_MAX_CLUSTERS - 1 - c);
        int cn = (int)((T + chunk_records - 1) / chunk_records);
        cn = cn < 1 ? 1 : cn;
        cn = cn > DEP_MAX_CHUNKS ? DEP_MAX_CHUNKS : cn;
        cn = cn > avail ? avail : cn;
        // a piece stages its share of the records and walks the region's cells
        const unsigned int cost = T / (unsigned int)cn + 16u + (cells >> 3);
        if (lane < cn) raw[(size_t)k * 64 + n_out + lane] = ((unsigned long long)cost << 32) | pack_dep_item(k, c, lane, cn);
        n_out += cn;
        total += T + 16u + (cells >> 3);
    }
    if (tile_ext < 0.0) {  // several clusters: the whole tile's box and largest radius, as k_tile_extent takes them
        const double BIG = 1e300;
        double lx = BIG, hx = -BIG, ly = BIG, hy = -BIG, lz = BIG, hz = -BIG, rr = 0.0;
#pragma unroll
        for (int h = 0; h < DEP_H; h++)
            if (t.live[h]) {
                lx = fmin(lx, t.x[h]); hx = fmax(hx, t.x[h]); ly = fmin(ly, t.y[h]); hy = fmax(hy, t.y[h]); lz = fmin(lz, t.z[h]); hz = fmax(hz, t.z[h]);
                rr = fmax(rr, sqrt(fmin(t.r2[h], tp.gather_r2)));
            }
        tile_ext = fmax(wave_max_d(hx) - wave_min_d(lx), fmax(wave_max_d(hy) - wave_min_d(ly), wave_max_d(hz) - wave_min_d(lz))) + 2.0 * wave_max_d(rr);
    }
    if (lane == 0) {
        est[k] = total; n_pieces[k] = (uint8_t)n_out;
        int e;
        frexp(tile_ext, &e);  // in [2^(e-1), 2^e)
        e = e + 32;
        e = e < 0 ? 0 : (e > 62 ? 62 : e);
        ext_bin[k] = (uint8_t)(e + 1);
    }
}

// The pieces by cost class, heaviest first (as k_items_hist / k_items_bases / k_items_scatter of sppm_gather_tiles.hpp, in
// two launches instead of three, and with the grid statistics riding along).  q: [0] number of items, [1] the deposit
// kernel's queue head, [2 + c] pieces of class c, [2 + ITEM_CLASSES + c] write cursor of class c; all cleared by
// k_deposit_plan.
// First launch: count the pieces per class and the tiles per extent bin.
__global__ void __launch_bounds__(256) k_items_count(const unsigned long long* __restrict__ raw, const uint8_t* __restrict__ n_pieces, int n_tiles,
                                                     unsigned int* __restrict__ q, const uint8_t* __restrict__ ext_bin, TileStats* __restrict__ st)
{
    __shared__ unsigned int h[ITEM_CLASSES], eh[64], live;
    if (threadIdx.x < ITEM_CLASSES) h[threadIdx.x] = 0;
    if (threadIdx.x < 64) eh[threadIdx.x] = 0;
    if (threadIdx.x == 0) live = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = k < n_tiles ? n_pieces[k] : 0;
    for (int j = 0; j < n; j++) atomicAdd(&h[item_class((unsigned int)(raw[(size_t)k * 64 + j] >> 32))], 1u);
    const int b = k < n_tiles ? ext_bin[k] : 0;
    if (b) { atomicAdd(&eh[b - 1], 1u); atomicAdd(&live, 1u); }
    __syncthreads();
    if (threadIdx.x < ITEM_CLASSES && h[threadIdx.x]) { atomicAdd(&q[2 + threadIdx.x], h[threadIdx.x]); atomicAdd(&q[0], h[threadIdx.x]); }
    if (threadIdx.x < 64 && eh[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], eh[threadIdx.x]);
    if (threadIdx.x == 0 && live) atomicAdd(&st->live_tiles, live);
}
// Second launch: every block derives the classes' bases from the counts (heaviest class first) and places its pieces;
// block 0 also clears the record counter of the batch's buffer (its records are binned by now) and writes the host's
// snapshot of the statistics (k_stat_snapshot).
__global__ void __launch_bounds__(256) k_items_place(const unsigned long long* __restrict__ raw, const uint8_t* __restrict__ n_pieces, int n_tiles,
                                                     unsigned int* __restrict__ q, uint32_t* __restrict__ items, unsigned long long* __restrict__ rec_count,
                                                     const TileStats* __restrict__ st, const unsigned long long* __restrict__ bb,
                                                     const unsigned long long* __restrict__ counters, StatSnapshot* __restrict__ out)
{
    __shared__ unsigned int h[ITEM_CLASSES], base[ITEM_CLASSES];
    if (threadIdx.x < ITEM_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = k < n_tiles ? n_pieces[k] : 0;
    for (int j = 0; j < n; j++) atomicAdd(&h[item_class((unsigned int)(raw[(size_t)k * 64 + j] >> 32))], 1u);
    __syncthreads();
    if (threadIdx.x < ITEM_CLASSES) {
        unsigned int before = 0;  // pieces of the heavier classes
        for (int c = ITEM_CLASSES - 1; c > (int)threadIdx.x; c--) before += q[2 + c];
        base[threadIdx.x] = before + (h[threadIdx.x] ? atomicAdd(&q[2 + ITEM_CLASSES + threadIdx.x], h[threadIdx.x]) : 0u);
        h[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int j = 0; j < n; j++) {
        const unsigned long long w = raw[(size_t)k * 64 + j];
        const int c = item_class((unsigned int)(w >> 32));
        items[base[c] + atomicAdd(&h[c], 1u)] = (uint32_t)w;
    }
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t == 0) *rec_count = 0ull;
        if (t < 64) out->hs.hist[t] = st->hist[t];
        if (t == 0) { out->hs.live_tiles = st->live_tiles; out->hs.pad = 0; out->hs.est_sum = st->est_sum; }
        if (t < 6) out->bbq[t] = bb[t];
        if (t < 24) out->cnt[t] = counters[t];
    }
}

#ifndef DEP_WAVES
#define DEP_WAVES (DEP_H == 1 ? 4 : 3)  // register budget: 512 / DEP_WAVES per lane
#endif
#define DEP_ATTR __attribute__((amdgpu_waves_per_eu(DEP_WAVES, DEP_WAVES)))
#define DEP_FLUSH_ROWS 2048  // staged rows after which the double sums move to the 64-bit integers (2048 x 2^41 = 2^52)
#ifndef DEP_NEAR_MAX
// (timing experiments, tools/dep_ablation.sh: parts of the kernel compiled out; the default build defines none of these)
#ifdef DEP_EXP_NOLOAD
#define DEP_EXP_NOLOAD_V 1
#else
#define DEP_EXP_NOLOAD_V 0
#endif
#ifdef DEP_EXP_LOAD2
#define DEP_EXP_L2 0
#define DEP_EXP_L3 1
#define DEP_EXP_L4 0
#else
#define DEP_EXP_L2 2
#define DEP_EXP_L3 3
#define DEP_EXP_L4 4
#endif
#define DEP_NEAR_MAX 1e-4
#ifndef DEP_E32_SCALE
#define DEP_E32_SCALE 1.0  // (a build with 1e4 here sends every hundredth pair through the exact fallback: how that path was validated)
#endif
#endif
#define DEP_ROWS 64  // staged records per wavefront: one group of candidates at a time, processed while the next group's loads are in flight
// VETO: the reference's K cap is enforced (sppm_cap.hpp): a pair the capped kd search dropped is in tp.veto and is not
// deposited (src/stochastic_ppm.h:425-427 hand gahter_max_photon_num to the same SearchNearest as src/sppm2.h).
template <bool TAP, bool STAMPS = false, bool VETO = false>
__global__ void __launch_bounds__(256) DEP_ATTR k_deposit_tiles(HpArrays hp, RecArrays rec, const uint32_t* __restrict__ cell_start, TileParams tp,
                                                       const uint32_t* __restrict__ items, unsigned int* __restrict__ q,
                                                       unsigned long long* __restrict__ counters, int32_t* __restrict__ err)
{
    constexpr bool KEYS = TAP || VETO;  // the staged rows carry the record's key
    constexpr int ROW = KEYS ? 11 : 10;
    constexpr bool F32 = !VETO && DEP_H == 1;  // the distance-only rows in single precision (see the easy loop)
    // a staged record of the full test: position, geometric normal, fixed-point flux (integer-valued doubles); of the distance-only
    // test: position about the cluster's centre, its square, flux, position; [ROW - 1: key bits]
    __shared__ __align__(16) double s_row[4][DEP_ROWS][ROW];
    __shared__ __align__(16) float s_fp[4][DEP_ROWS / 2][12];  // (single-precision full test: two rows share (x0 x1 y0 y1)(z0 z1 gx0 gx1)(gy0 gy1 gz0 gz1))
    __shared__ uint32_t s_idx[4][DEP_ROWS];  // (single-precision rows: the record's place, for the pairs decided from the record itself)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int P = hp.P;
    const double thr = tp.gather_r2 * 0.01;
    const double band_rel = 1e-9 + 8e-15 / thr;
    const double thr2_lo = thr * thr * (1.0 - band_rel), thr2_hi = thr * thr * (1.0 + band_rel);
    // (the two constants of the classification, with their margins folded in; pinned: uniform values computed by vector
    // instructions otherwise sit in vector registers, and these two were spilled and re-read from scratch memory per group)
    const double c_rej = pin_scalar_d(thr2_hi * (1.0 + 1e-9) * (1.0 + 1e-12)), c_easy = pin_scalar_d(thr2_lo * (1.0 - 1e-9));
    const double inv_thr2_up = (1.0 / thr2_lo) * (1.0 + 4e-9);  // x this >= x / (thr2_lo (1 - 1e-9)), with room for the roundings
    const unsigned int n_items = q[0];
    const unsigned int n_waves = gridDim.x * 4u;
#define DSTAMP() (STAMPS ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
    unsigned long long st_setup = 0, st_count = 0, st_issue = 0, st_append = 0, st_process = 0, st_flush = 0, st_wait = 0;  // diagnostic build only
    unsigned long long w_pairs = 0;
    unsigned int w_acc = 0, w_sel = 0, w_round = 0, w_items = 0;  // (w_acc: per lane, summed over the wavefront at the end; a lane's share of a launch stays far below 2^32)
    for (bool first_item = true;; first_item = false) {
    const unsigned long long ts_item = DSTAMP();
    // persistent wavefronts over the list of items, heaviest first (as k_gather_tiles)
    unsigned int item_i = blockIdx.x * 4u + (unsigned int)wv;
    if (!first_item) {
        if (lane == 0) item_i = n_waves + atomicAdd(&q[1], 1u);
        item_i = (unsigned int)__builtin_amdgcn_readfirstlane((int)item_i);
    }
    if (item_i >= n_items) break;
    const uint32_t item = items[item_i];
    const int k = (int)(item >> 12), ci = (int)((item >> 8) & 15u), cj = (int)((item >> 4) & 15u), cn = (int)(item & 15u) + 1;
    DepTile t;
    if (!dep_load_tile(hp, tp, k, lane, t)) continue;
    // the hitpoints of cluster ci
    unsigned long long remaining[DEP_H];
#pragma unroll
    for (int h = 0; h < DEP_H; h++) remaining[h] = ballot64(t.live[h]);
    bool inc[DEP_H];
#pragma unroll
    for (int h = 0; h < DEP_H; h++) inc[h] = false;
    for (int c = 0; c <= ci && dep_any(remaining); c++) {
        bool in_c[DEP_H];
        dep_next_cluster(remaining, c, t, lane, tp.cluster_d, in_c);
#pragma unroll
        for (int h = 0; h < DEP_H; h++) inc[h] = c == ci && in_c[h];
    }
    const DepRegion R = dep_region(t, inc, tp);
    if (R.n_inc == 0) continue;
    const double rmax2 = pin_scalar_d(R.rmax * R.rmax);  // (pin_scalar_d: wave-uniform values kept in scalar registers, see the cluster bounds below)
    const int kx = R.cx1 - R.cx0 + 1, ky = R.cy1 - R.cy0 + 1, kz = R.cz1 - R.cz0 + 1;
    const int Ktot = kx * ky * kz;
    // (a row of kx cells along x is one run of consecutive records: the region is walked row by row, 64 rows at a time)
    const int Krows = ky * kz;
    d3 hn[DEP_H];
#pragma unroll
    for (int h = 0; h < DEP_H; h++) hn[h] = inc[h] ? mk(hp.nrm[t.p[h]], hp.nrm[P + t.p[h]], hp.nrm[2 * P + t.p[h]]) : mk(0, 0, 0);
    // the region's records form one virtual list (cells in q order); an item of a region cut into cn chunks takes
    // entries [T cj / cn, T (cj + 1) / cn) of it, T = the list's length
    unsigned int v_lo = 0u, v_hi = 0xFFFFFFFFu;
    const unsigned long long ts_setup = DSTAMP();
    st_setup += ts_setup - ts_item;
    if (cn > 1) {
        unsigned int T = 0;
        for (int qbase = 0; qbase < Krows; qbase += 64) {
            const int qi = qbase + lane;
            uint32_t cnt = 0;
            if (qi < Krows) {
                const uint32_t b = (uint32_t)(((R.cz0 + qi / ky) * tp.g.ny + R.cy0 + qi % ky) * tp.g.nx + R.cx0);
                cnt = cell_start[b + kx] - cell_start[b];
            }
            T += (unsigned int)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(cnt), 63);
        }
        v_lo = (unsigned int)((unsigned long long)T * (unsigned int)cj / (unsigned int)cn);
        v_hi = (unsigned int)((unsigned long long)T * (unsigned int)(cj + 1) / (unsigned int)cn);
    }
    // Per-hitpoint bounds of the fast test.  The fast path evaluates the squared distance and the two dot products with
    // fused multiply-adds (three instructions each instead of five); its value differs from the reference's unfused
    // expression by a few units in the last place of the largest term, so a comparison is only trusted when the
    // value is farther than 1e-13 (relative; the dot products: absolute) from its bound, and a record for which any
    // hitpoint is closer than that is evaluated again with the reference's exact operation order (once in ~10^10 pairs).
    double lim_lo[DEP_H], lim_hi[DEP_H];
    // sums of the records' fixed-point flux, which k_grid_scatter stores as integer-valued doubles below 2^41 + 1: v_add_f64
    // is exact while a sum stays below 2^53, so the sums move to the 64-bit atomics before DEP_FLUSH_ROWS rows are added
    double ax[DEP_H], ay[DEP_H], az[DEP_H];
    int am[DEP_H];
    unsigned int c_acc = 0, rows_unflushed = 0;
#pragma unroll
    for (int h = 0; h < DEP_H; h++) {
        const double lim = fmin(t.r2[h], tp.gather_r2);  // a record is accepted at d2 < gather_r2 && d2 <= r2
        lim_lo[h] = inc[h] ? lim * (1.0 - 1e-13) : -1.0; lim_hi[h] = inc[h] ? lim * (1.0 + 1e-13) : -1.0;  // -1: never inside, never too close to call
        ax[h] = ay[h] = az[h] = 0.0; am[h] = 0;
    }
#define DEP_FLUSH()                                                                                                    \
    {                                                                                                                  \
        _Pragma("unroll") for (int h = 0; h < DEP_H; h++) {                                                            \
            if (am[h] > 0) {                                                                                           \
                int pi = t.p[h];                                                                                       \
                asm volatile("" : "+v"(pi)); /* (the four addresses are formed here, not kept in registers through the loops) */ \
                atomicAdd((unsigned long long*)hp.dep + pi, (unsigned long long)(long long)ax[h]);                     \
                atomicAdd((unsigned long long*)hp.dep + (size_t)P + pi, (unsigned long long)(long long)ay[h]);         \
                atomicAdd((unsigned long long*)hp.dep + 2 * (size_t)P + pi, (unsigned long long)(long long)az[h]);     \
                atomicAdd((unsigned long long*)hp.dep + 3 * (size_t)P + pi, (unsigned long long)am[h]);                \
            }                                                                                                          \
            c_acc += (unsigned int)am[h];                                                                              \
            ax[h] = ay[h] = az[h] = 0.0; am[h] = 0;                                                                    \
        }                                                                                                              \
        rows_unflushed = 0;                                                                                            \
    }
    // Whole-cluster bounds for the two tests that do not involve the radius (DEP_H == 1).  The 64 hitpoints of a cluster
    // lie within a fraction of a unit of each other, nearly always on one surface: for most staged records the normal
    // test and the flatness test come out the same for EVERY hitpoint of the cluster.  One lane per staged record
    // decides that from the cluster's bounding box and the spread of its normals, with margins far above rounding --
    // where a bound holds, each lane's exact test is known to pass (or fail), so the outcome is the exact one:
    //   rejected by all (normal or flatness fails for the whole cluster): the record is skipped;
    //   "easy" (both pass for the whole cluster): only the distance test is left for the lanes -- half the arithmetic
    //   and two thirds of the LDS reads of the full test, which the rest of the records still take.
    double ndev = 0.0;
    d3 nbar = mk(0, 0, 0);
    // (wave-uniform, but computed by vector instructions: without the pin they stay in vector registers, of which the
    // kernel has none to spare -- three of them were spilled and re-read from scratch memory for every group staged)
    const double bcx = pin_scalar_d(0.5 * (R.bx0 + R.bx1)), bcy = pin_scalar_d(0.5 * (R.by0 + R.by1)), bcz = pin_scalar_d(0.5 * (R.bz0 + R.bz1));
    const double bex = pin_scalar_d((0.5 * (R.bx1 - R.bx0)) * (1.0 + 1e-12) + 1e-12), bey = pin_scalar_d((0.5 * (R.by1 - R.by0)) * (1.0 + 1e-12) + 1e-12),
                 bez = pin_scalar_d((0.5 * (R.bz1 - R.bz0)) * (1.0 + 1e-12) + 1e-12);
    if (DEP_H == 1) {
        const int seed = __ffsll((long long)ballot64(inc[0])) - 1;
        nbar = mk(rl_d(hn[0].x, seed), rl_d(hn[0].y, seed), rl_d(hn[0].z, seed));
        ndev = pin_scalar_d(sqrt(wave_max_d(inc[0] ? len2(hn[0] - nbar) : 0.0)) * (1.0 + 1e-9) + 1e-15);
    }
    // The distance test of the "easy" records, expanded about the cluster's centre c (h' = h - c, q' = q - c):
    // |h - q|^2 - |h'|^2 = |q'|^2 - 2 h'.q', three fused multiply-adds per pair from the record's (q', |q'|^2) against
    // per-lane constants, compared with lim - |h'|^2.  About the centre every term is of the size of the cluster plus
    // the search radius, so the value differs from the reference's |h - q|^2 by at most ~40 roundings of (Hm + Qm)^2
    // (Hm, Qm: the largest |h'|, |q'|); a comparison closer than E = 900 such roundings to its bound is not trusted and
    // the record is evaluated again in the reference's operation order from the positions themselves.
    const double hcx = t.x[0] - bcx, hcy = t.y[0] - bcy, hcz = t.z[0] - bcz;
    const double m2x = -2.0 * hcx, m2y = -2.0 * hcy, m2z = -2.0 * hcz;
    const double hh = fma(hcz, hcz, fma(hcy, hcy, hcx * hcx));
    double lo_c = -INFINITY, hi_c = -INFINITY;  // a lane outside the cluster: never inside, never too close to call
    float m2xf = 0.0f, m2yf = 0.0f, m2zf = 0.0f, lo32 = -INFINITY, hi32 = -INFINITY;
    float hnxf = 0.0f, hnyf = 0.0f, hnzf = 0.0f, dlo32 = -1.0f, dhi32 = -1.0f, ef32 = 0.0f, tlo2f = 0.0f, thi2f = 0.0f;  // (-1: never inside, never too close to call)
    double gl = 0.0;  // the value c of a pair is within E of d2 - hh: c - G > gl = E - hh guarantees d2 > G (third class, below)
    if (DEP_H == 1) {
        const double Hm = sqrt(wave_max_d(inc[0] ? hh : 0.0)), Qm = sqrt(bex * bex + bey * bey + bez * bez) + R.rmax;
        const double E = 1e-13 * (Hm + Qm) * (Hm + Qm);
        if (inc[0]) { lo_c = (lim_lo[0] - hh) - E; hi_c = (lim_hi[0] - hh) + E; }
        gl = E - hh;
        if (F32) {
            // The same test in single precision (the "easy" rows, below).  With h', q' and |q'|^2 rounded to float and three
            // float fused multiply-adds, the value differs from the exact |q'|^2 - 2 h'.q' by at most 2^-24 x (3 (Hm + Qm)^2
            // for the three roundings of partial sums, each below (Hm + Qm)^2, + 4 Hm Qm for the rounded factors + Qm^2
            // for the rounded square) <= 7 x 2^-24 (Hm + Qm)^2; the two bounds lose at most 2^-24 (Hm + Qm)^2 more when
            // they are rounded to float.  E32 = 10 x 2^-24 (Hm + Qm)^2 on top of E; a pair closer than that is decided
            // from the record itself in the reference's operation order (about one row in ten thousand).
            const double E32 = DEP_E32_SCALE * 10.0 * 5.9604644775390625e-8 * (Hm + Qm) * (Hm + Qm) + 1e-30;
            m2xf = -2.0f * (float)hcx; m2yf = -2.0f * (float)hcy; m2zf = -2.0f * (float)hcz;
            if (inc[0]) { lo32 = (float)(lo_c - E32); hi32 = (float)(hi_c + E32); }
            // The full test in single precision (below), about the same centre.  With u = 2^-24, D = Hm + Qm and e = q' - h':
            // the computed e is within 3.5 u D of its value (as a vector), so |e|^2 within 7 u D |e| + 3 u |e|^2 and g.e
            // (|g| = 1) within 4 u |e| + 3.5 u D; then F = thr^2 |e|^2 - (g.e)^2 is within
            //   u [(7 + 7 thr^2) D |e| + (10 + 5 thr^2) |e|^2] + O(u^2 D^2),   and with D |e| <= (D^2 / 32 + 32 |e|^2) / 2:
            //   A D^2 + B |e|^2,   A = u (7 + 7 thr^2) / 64,   B = u (122 + 117 thr^2)
            // (a band proportional to D^2 alone would send every pair much closer than the cluster's size to the fallback).
            // Both are taken half as wide again, B also covers the rounding of thr^2 -+ B to float and the relative 1e-9
            // band of the double-precision form.  flat for sure: (thr^2 - B) |e|^2 - (g.e)^2 > A D^2; not flat for sure:
            // (thr^2 + B) |e|^2 - (g.e)^2 < -A D^2.  The distance: within 1.5 (7 u D sqrt(lim) + 3 u lim) of a lane's bound
            // `lim` (plus E for the centring); n.g of unit vectors: within 3e-7, band 4e-6.  A pair inside a band sends
            // its row to the reference's operation order in double precision, as before.
            const double u_ = 5.9604644775390625e-8 * DEP_E32_SCALE, D2 = (Hm + Qm) * (Hm + Qm), t2_ = thr * thr;
            const double Bf = 1.5 * u_ * (122.0 + 117.0 * t2_) + 2.0 * u_ * t2_ + 2.0 * band_rel * t2_;
            ef32 = (float)((1.5 * u_ * (7.0 + 7.0 * t2_) / 64.0 + 1e-12) * D2 + 1e-30);
            tlo2f = (float)(t2_ - Bf); thi2f = (float)(t2_ + Bf);
            hnxf = (float)hn[0].x; hnyf = (float)hn[0].y; hnzf = (float)hn[0].z;
            const double lim_ = fmin(t.r2[0], tp.gather_r2);
            const double Ed = 1.5 * u_ * (7.0 * (Hm + Qm) * sqrt(lim_) + 3.0 * lim_) + E + 1e-30;
            if (inc[0]) { dlo32 = (float)(lim_lo[0] - Ed); dhi32 = (float)(lim_hi[0] + Ed); }
        }
    }
    // Third class of staged records ("near"): the normal test passes for the whole cluster, but the record lies inside
    // the cluster's box or so close to it that the flatness test cannot be decided from the box alone.  For the whole
    // cluster |g.(h - q)| <= f_hi, so the flatness test of a PAIR passes once its own squared distance exceeds
    // G = f_hi^2 / thr^2 (margins as for the other bounds) -- one subtraction and one comparison more than the
    // distance-only test; a pair at or below G (for a flat cluster G is ~1e-10: a record on top of a hitpoint) is
    // decided by the reference's expressions from the record itself.  Only records whose G is a negligible share of
    // the search area take this class.
    const double g_max = pin_scalar_d(DEP_NEAR_MAX * rmax2);
    st_count += DSTAMP() - ts_setup;
    unsigned int c_sel = 0, c_round = 0;
    unsigned long long c_pairs = 0;
    unsigned long long d_easy = 0, d_full = 0, d_near = 0, d_near_slow = 0, d_win = 0, d_grp = 0, d_load = 0;  // diagnostic build: staged records by class
    int sel_n = 0, n_easy = 0, n_near = 0, n_full = 0;
    // iterator over the list in groups of 64 entries
    int qbase = -64;
    unsigned int voff = 0, cbase = 0, i_end = 0;
    uint32_t g_d = 0, g_pre = 0;
    bool pend = false, p_has = false;          // a group whose loads are in flight
    uint32_t p_r = 0;
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0, p8 = 0;
    unsigned long long p_key = 0;
#define DEP_ISSUE()                                                                                                    \
    {                                                                                                                  \
        pend = true;                                                                                                   \
        while (cbase >= i_end) {                                                                                       \
            qbase += 64;                                                                                               \
            if (STAMPS) d_win++;                                                                                       \
            if (qbase >= Krows || voff >= v_hi) { pend = false; break; }                                               \
            const int qi = qbase + lane;                                                                               \
            uint32_t cnt = 0, cs = 0;                                                                                  \
            if (qi < Krows) {                                                                                          \
                const uint32_t b = (uint32_t)(((R.cz0 + qi / ky) * tp.g.ny + R.cy0 + qi % ky) * tp.g.nx + R.cx0);      \
                cs = cell_start[b];                                                                                    \
                cnt = cell_start[b + kx] - cs;                                                                         \
            }                                                                                                          \
            const uint32_t incl = wave_incl_scan_u32(cnt);                                                             \
            g_pre = incl - cnt;                                                                                        \
            g_d = cs - g_pre; /* list entry i of a row lies at g_d + i */                                              \
            const unsigned int tot = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);                           \
            const unsigned int g_lo = voff, g_hi = voff + tot;                                                         \
            voff = g_hi;                                                                                               \
            cbase = 0; i_end = 0;                                                                                      \
            if (g_hi <= v_lo || g_lo >= v_hi) continue;                                                                \
            cbase = (v_lo > g_lo ? v_lo : g_lo) - g_lo; i_end = (v_hi < g_hi ? v_hi : g_hi) - g_lo;                    \
        }                                                                                                              \
        if (pend) {                                                                                                    \
            const unsigned int i = cbase + (unsigned int)lane;                                                         \
            p_has = i < i_end;                                                                                         \
            if (STAMPS) { d_grp++; d_load += (unsigned long long)__popcll(ballot64(p_has)); }                          \
            {                                                                                                          \
                int j = 0; /* the row lane that owns list entry i: the largest j with pre_j <= i */                    \
                for (int step = 32; step >= 1; step >>= 1) {                                                           \
                    const int cand = j + step;                                                                         \
                    const uint32_t pv = __shfl(g_pre, cand & 63);                                                      \
                    if (cand < 64 && pv <= i) j = cand;                                                                \
                }                                                                                                      \
                p_r = __shfl(g_d, j) + i;                                                                              \
            }                                                                                                          \
            if (p_has && !DEP_EXP_NOLOAD_V) {                                                                          \
                const double2* rp = (const double2*)(rec.f + (size_t)REC_ROW * p_r);  /* one row: sppm_records.hpp */  \
                const double2 ra = rp[0], rb = rp[1], rc = rp[DEP_EXP_L2], rd = rp[DEP_EXP_L3], re = rp[DEP_EXP_L4];   \
                p0 = ra.x; p1 = ra.y; p2 = rb.x; p3 = rb.y; p4 = rc.x; p5 = rc.y; p6 = rd.x; p7 = rd.y; p8 = re.x;     \
                if (KEYS) p_key = (unsigned long long)__double_as_longlong(re.y);                                      \
            }                                                                                                          \
            cbase += 64;                                                                                               \
        }                                                                                                              \
    }
    { const unsigned long long t0 = DSTAMP(); DEP_ISSUE() st_issue += DSTAMP() - t0; }
    while (pend || sel_n > 0) {
        if (pend) {
            const unsigned long long t0 = DSTAMP();
            if (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_wait += DSTAMP() - t0; }  // (how long the group's loads are waited for)
            // the pending group: keep what lies within rmax of the cluster's bounding box (a hitpoint of the cluster
            // can only accept such records), sort it by class (see above: dropped / distance test only / full test)
            // while each lane still holds its record in registers, compact it into the staging rows -- "easy" records
            // from row 0 up, the others from row 63 down -- and start the loads of the next group
            bool in = false, easy = false, near = false;
            double G = 0.0;
            if (p_has) {
                // distance to the cluster's box from its centre and half extents (the half extents carry a margin upwards, so the
                // nearest point comes out no farther and the farthest no nearer than they are: every use below is on the safe side)
                const double ux = p0 - bcx, uy = p1 - bcy, uz = p2 - bcz;
                const double nx = fmax(fabs(ux) - bex, 0.0), ny = fmax(fabs(uy) - bey, 0.0), nz = fmax(fabs(uz) - bez, 0.0);
                const double dmin2 = DEP_H == 1 ? nx * nx + ny * ny + nz * nz : dep_box_dmin2(R, p0, p1, p2);  // (two hitpoints per lane: no centre is made)
                in = dmin2 <= rmax2;
#ifdef DEP_EXP_NOCLASS
                easy = in;
                if (false) {
#else
                if (DEP_H == 1 && in) {
#endif
                    const double fx_ = fabs(ux) + bex, fy_ = fabs(uy) + bey, fz_ = fabs(uz) + bez;
                    const double dmax2 = fx_ * fx_ + fy_ * fy_ + fz_ * fz_;  // (its margin upwards is in c_rej)
                    const double nd = nbar.x * p3 + nbar.y * p4 + nbar.z * p5;                    // a hitpoint's n.g lies within ndev of it
                    const double sc = fabs(p3 * ux + p4 * uy + p5 * uz);                          // |g.(h - q)| lies within `spread` of it
                    const double spread = fabs(p3) * bex + fabs(p4) * bey + fabs(p5) * bez;
                    const double f_hi = (sc + spread) * (1.0 + 1e-9), f_lo = fmax(sc - spread, 0.0) * (1.0 - 1e-9);
                    const bool rej = nd + ndev < 1e-3 - 1e-9 || f_lo * f_lo > c_rej * dmax2;
                    const bool n_all = nd - ndev > 1e-3 + 1e-9;
                    easy = !rej && n_all && dmin2 > 0.0 && f_hi * f_hi < c_easy * dmin2;
                    G = (f_hi * f_hi) * inv_thr2_up;
                    near = !F32 && !rej && !easy && n_all && G < g_max;  // (single precision: those records take the full test, which costs little there)
                    in = !rej;
                }
            }
            const unsigned long long m_e = ballot64(in && easy), m_n = ballot64(in && near), m_f = ballot64(in && !easy && !near);
#ifdef DEP_EXP_NOWRITE
            if (false) {
#else
            if (in) {
#endif
                double* row = s_row[wv][easy ? (int)mbcnt64(m_e) : (near ? __popcll(m_e) + (int)mbcnt64(m_n) : DEP_ROWS - 1 - (int)mbcnt64(m_f))];
                const double f6 = p6, f7 = p7, f8 = p8;  // fixed point already (k_grid_scatter): integer-valued doubles
                if (F32 && easy) {
                    // Two rows share 80 bytes: (x0 x1 y0 y1)(z0 z1 S0 S1) as floats -- q' and S = |q'|^2, laid out as the operand
                    // pairs of packed single-precision multiply-adds -- then the flux of both rows as doubles; pairs are
                    // packed from the start of the wavefront's staging area (pair k lies inside row k of the other
                    // classes' layout, and those begin at row n_easy).  The record's place goes to s_idx.
                    const double qx = p0 - bcx, qy = p1 - bcy, qz = p2 - bcz;
                    const int slot = (int)mbcnt64(m_e);
                    char* pb = (char*)&s_row[wv][0][0] + 80 * (slot >> 1) + 4 * (slot & 1);
                    *(float*)pb = (float)qx; *(float*)(pb + 8) = (float)qy; *(float*)(pb + 16) = (float)qz;
                    *(float*)(pb + 24) = (float)fma(qz, qz, fma(qy, qy, qx * qx));
                    double* fb = (double*)((char*)&s_row[wv][0][0] + 80 * (slot >> 1) + 32 + 24 * (slot & 1));
                    fb[0] = f6; fb[1] = f7; fb[2] = f8;
                    s_idx[wv][slot] = p_r;
                } else if (easy || near) {
                    const double qx = p0 - bcx, qy = p1 - bcy, qz = p2 - bcz;
                    row[0] = qx; row[1] = qy; row[2] = qz; row[3] = fma(qz, qz, fma(qy, qy, qx * qx));
                    row[4] = f6; row[5] = f7; row[6] = f8;
                    if (easy) { row[7] = p0; row[8] = p1; row[9] = p2; }
                    else { row[7] = G; row[8] = __longlong_as_double((long long)p_r); }  // (the record's place: a pair at or below G reads it again)
                } else {
                    row[0] = p0; row[1] = p1; row[2] = p2; row[3] = p3; row[4] = p4; row[5] = p5;
                    row[6] = f6; row[7] = f7; row[8] = f8;
                    if (F32) {
                        const int j = (int)mbcnt64(m_f);
                        float* pp = &s_fp[wv][j >> 1][j & 1];
                        pp[0] = (float)(p0 - bcx); pp[2] = (float)(p1 - bcy); pp[4] = (float)(p2 - bcz);
                        pp[6] = (float)p3; pp[8] = (float)p4; pp[10] = (float)p5;
                    }
                }
                if (KEYS && !(F32 && easy)) row[ROW - 1] = __longlong_as_double((long long)p_key);
            }
            n_easy = __popcll(m_e); n_near = __popcll(m_n); n_full = __popcll(m_f);
            if (F32 && (n_full & 1) && lane == 0) {  // an odd full-test row's partner: at infinity (outside every radius), normal zero
                float* pp = &s_fp[wv][n_full >> 1][1];
                pp[0] = INFINITY; pp[2] = 0.0f; pp[4] = 0.0f; pp[6] = 0.0f; pp[8] = 0.0f; pp[10] = 0.0f;
            }
            if (F32 && (n_easy & 1) && lane == 0) {  // an odd row's partner: S = +inf is rejected by every lane and is never too close to call
                char* pb = (char*)&s_row[wv][0][0] + 80 * (n_easy >> 1) + 4;
                *(float*)pb = 0.0f; *(float*)(pb + 8) = 0.0f; *(float*)(pb + 16) = 0.0f; *(float*)(pb + 24) = INFINITY;
            }
            sel_n = n_easy + n_near + n_full;
            if (STAMPS) { d_easy += (unsigned long long)n_easy; d_full += (unsigned long long)n_full; d_near += (unsigned long long)n_near; }
            const unsigned long long t1 = DSTAMP();
            DEP_ISSUE()
            if (STAMPS) { st_append += t1 - t0; st_issue += DSTAMP() - t1; }
            if (sel_n == 0) continue;
        }
        const unsigned long long tp0 = DSTAMP();
        // every hitpoint against the staged records: the search filter of src/kdtree.h:123-127 (its |cos| does not
        // depend on the orientation of the normal) and the acceptance test of src/stochastic_ppm.h:441-445
        __builtin_amdgcn_wave_barrier();
        c_sel += (unsigned int)sel_n; c_pairs += (unsigned long long)sel_n * R.n_inc; c_round++;
        const int ne_u = __builtin_amdgcn_readfirstlane(n_easy), nn_u = __builtin_amdgcn_readfirstlane(n_near), nf_u = __builtin_amdgcn_readfirstlane(n_full);
        if (rows_unflushed + (unsigned int)sel_n > DEP_FLUSH_ROWS) DEP_FLUSH()
        rows_unflushed += (unsigned int)sel_n;
        // rows [0, n_easy): the normal and flatness tests pass for the whole cluster, the distance test alone decides.
        // Two records per trip (independent chains of LDS read -> three fused multiply-adds -> compares); a pair too
        // close to call (see E above) is left out here and only noted in `und` -- the loop carries no branch for it.
#define DEP_TAKE(rowp)                                                                                                 \
        if (!VETO || !veto_has(tp.veto, (unsigned long long)__double_as_longlong((rowp)[ROW - 1]), t.p[0])) {          \
            ax[0] += (rowp)[4]; ay[0] += (rowp)[5]; az[0] += (rowp)[6]; am[0]++;                                      \
            if (TAP) {                                                                                                 \
                const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);                                               \
                if ((int64_t)ts < tp.tap_cap) { tp.tap_key[ts] = (unsigned long long)__double_as_longlong((rowp)[ROW - 1]); tp.tap_pix[ts] = t.p[0]; } \
            }                                                                                                          \
        }
        unsigned long long und = 0ull;
        if constexpr (F32) {
            // Single precision, two rows per trip as ONE packed operation each (v_pk_fma_f32): the loop is bound by the vector
            // instructions it issues (four cycles each per wavefront), 16 per trip in this form against 21 in double
            // precision.  All of a trip's LDS reads are requested at its top (the empty asm uses the flux ahead of the
            // tests: left to itself the compiler sinks those reads behind them, where each is waited for by itself).
#define DEP_TAKE32(fx_, fy_, fz_, slot_)                                                                               \
            {                                                                                                          \
                ax[0] += (fx_); ay[0] += (fy_); az[0] += (fz_); am[0]++;                                               \
                if (TAP) {                                                                                             \
                    const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);                                           \
                    if ((int64_t)ts < tp.tap_cap) {                                                                    \
                        tp.tap_key[ts] = (unsigned long long)__double_as_longlong(rec.f[(size_t)REC_ROW * s_idx[wv][slot_] + 9]); \
                        tp.tap_pix[ts] = t.p[0];                                                                       \
                    }                                                                                                  \
                }                                                                                                      \
            }
            const char* eb = (const char*)&s_row[wv][0][0];
            const v2f mx2 = {m2xf, m2xf}, my2 = {m2yf, m2yf}, mz2 = {m2zf, m2zf};
#ifdef DEP_EXP_SKIP_EASY
            if (false)
#endif
            int ti = 0;
            // (four rows per trip while they last: one wait for a trip's LDS reads however many rows it asked for; -3 % kernel time)
            for (; ti + 2 < ne_u; ti += 4) {
                const char* pb = eb + 40 * ti;
                const v4f A = *(const v4f*)pb, B = *(const v4f*)(pb + 16), A2 = *(const v4f*)(pb + 80), B2 = *(const v4f*)(pb + 96);
                v2d F0 = *(const v2d*)(pb + 32), F1 = *(const v2d*)(pb + 48), F2 = *(const v2d*)(pb + 64);
                v2d G0 = *(const v2d*)(pb + 112), G1 = *(const v2d*)(pb + 128), G2 = *(const v2d*)(pb + 144);
                const v2f c = __builtin_elementwise_fma(mz2, B.xy, __builtin_elementwise_fma(my2, A.zw, __builtin_elementwise_fma(mx2, A.xy, B.zw)));
                const v2f c2 = __builtin_elementwise_fma(mz2, B2.xy, __builtin_elementwise_fma(my2, A2.zw, __builtin_elementwise_fma(mx2, A2.xy, B2.zw)));
                asm volatile("" : "+v"(F0), "+v"(F1), "+v"(F2), "+v"(G0), "+v"(G1), "+v"(G2));
                const bool ok0 = c.x < lo32, ok1 = c.y < lo32, ok2 = c2.x < lo32, ok3 = c2.y < lo32;
                und |= ~(ballot64(ok0) | ballot64(c.x > hi32)) | ~(ballot64(ok1) | ballot64(c.y > hi32)) | ~(ballot64(ok2) | ballot64(c2.x > hi32)) | ~(ballot64(ok3) | ballot64(c2.y > hi32));
                if (ok0) DEP_TAKE32(F0.x, F0.y, F1.x, ti)
                if (ok1) DEP_TAKE32(F1.y, F2.x, F2.y, ti + 1)
                if (ok2) DEP_TAKE32(G0.x, G0.y, G1.x, ti + 2)
                if (ok3) DEP_TAKE32(G1.y, G2.x, G2.y, ti + 3)
            }
            for (; ti < ne_u; ti += 2) {
                const char* pb = eb + 40 * ti;
                const v4f A = *(const v4f*)pb, B = *(const v4f*)(pb + 16);
                v2d F0 = *(const v2d*)(pb + 32), F1 = *(const v2d*)(pb + 48), F2 = *(const v2d*)(pb + 64);
                const v2f c = __builtin_elementwise_fma(mz2, B.xy, __builtin_elementwise_fma(my2, A.zw, __builtin_elementwise_fma(mx2, A.xy, B.zw)));
                asm volatile("" : "+v"(F0), "+v"(F1), "+v"(F2));
                const bool ok0 = c.x < lo32, ok1 = c.y < lo32;
                und |= ~(ballot64(ok0) | ballot64(c.x > hi32)) | ~(ballot64(ok1) | ballot64(c.y > hi32));
                if (ok0) DEP_TAKE32(F0.x, F0.y, F1.x, ti)
                if (ok1) DEP_TAKE32(F1.y, F2.x, F2.y, ti + 1)
            }
            if ((und & ballot64(true)) != 0ull) {
                // the pairs left out: decided from the record's position itself, in the reference's operation order
                if (STAMPS) d_near_slow++;
                for (int tj = 0; tj < ne_u; tj++) {
                    const char* pb = eb + 80 * (tj >> 1) + 4 * (tj & 1);
                    const float c0 = __builtin_fmaf(m2zf, *(const float*)(pb + 16), __builtin_fmaf(m2yf, *(const float*)(pb + 8), __builtin_fmaf(m2xf, *(const float*)pb, *(const float*)(pb + 24))));
                    if (!(c0 < lo32) && !(c0 > hi32)) {
                        const double* rp = rec.f + (size_t)REC_ROW * s_idx[wv][tj];
                        int pr_ = t.p[0];  // (the hitpoint is read again: its double-precision state is not kept in registers through the loops)
                        asm volatile("" : "+v"(pr_));
                        const double e = len2(mk(hp.pos[pr_], hp.pos[P + pr_], hp.pos[2 * P + pr_]) - mk(rp[0], rp[1], rp[2]));
                        const double* fb = (const double*)(eb + 80 * (tj >> 1) + 32 + 24 * (tj & 1));
                        if (inc[0] && (e < tp.gather_r2) && (e <= hp.r2[pr_])) DEP_TAKE32(fb[0], fb[1], fb[2], tj)
                    }
                }
            }
#undef DEP_TAKE32
        } else {
        int ti = 0;
        for (; ti + 1 < ne_u; ti += 2) {
            const double* r0 = s_row[wv][ti];
            const double* r1 = s_row[wv][ti + 1];
            const double c0 = fma(m2z, r0[2], fma(m2y, r0[1], fma(m2x, r0[0], r0[3])));
            const double c1 = fma(m2z, r1[2], fma(m2y, r1[1], fma(m2x, r1[0], r1[3])));
            const bool ok0 = c0 < lo_c, ok1 = c1 < lo_c;
            und |= ~(ballot64(ok0) | ballot64(c0 > hi_c)) | ~(ballot64(ok1) | ballot64(c1 > hi_c));
            if (ok0) DEP_TAKE(r0)
            if (ok1) DEP_TAKE(r1)
        }
        if (ti < ne_u) {
            const double* r0 = s_row[wv][ti];
            const double c0 = fma(m2z, r0[2], fma(m2y, r0[1], fma(m2x, r0[0], r0[3])));
            const bool ok0 = c0 < lo_c;
            und |= ~(ballot64(ok0) | ballot64(c0 > hi_c));
            if (ok0) DEP_TAKE(r0)
        }
        if ((und & ballot64(true)) != 0ull) {
            // (once in ~10^9 pairs) the pairs left out: decided from the positions themselves, in the reference's operation order
            for (int tj = 0; tj < ne_u; tj++) {
                const double* row = s_row[wv][tj];
                const double c0 = fma(m2z, row[2], fma(m2y, row[1], fma(m2x, row[0], row[3])));
                if (!(c0 < lo_c) && !(c0 > hi_c)) {
                    const double e = len2(mk(t.x[0], t.y[0], t.z[0]) - mk(row[7], row[8], row[9]));
                    if (inc[0] && (e < tp.gather_r2) && (e <= t.r2[0])) DEP_TAKE(row)
                }
            }
        }
        }
        // rows [n_easy, n_easy + n_near): the distance test and the pair's own flatness bound (see g_max above)
#ifdef DEP_EXP_SKIP_NEAR
        if (false)
#endif
        if (!F32 && nn_u > 0) {
            unsigned long long und2 = 0ull;
            for (int tn = ne_u; tn < ne_u + nn_u; tn++) {
                const double* r0 = s_row[wv][tn];
                const double c0 = fma(m2z, r0[2], fma(m2y, r0[1], fma(m2x, r0[0], r0[3])));
                const bool acc = c0 < lo_c && c0 - r0[7] > gl;
                und2 |= ~(ballot64(acc) | ballot64(c0 > hi_c));
                if (acc) DEP_TAKE(r0)
            }
            if ((und2 & ballot64(true)) != 0ull) {
                if (STAMPS) d_near_slow++;
                for (int tn = ne_u; tn < ne_u + nn_u; tn++) {
                    const double* row = s_row[wv][tn];
                    const double c0 = fma(m2z, row[2], fma(m2y, row[1], fma(m2x, row[0], row[3])));
                    if (!(c0 < lo_c && c0 - row[7] > gl) && !(c0 > hi_c)) {
                        // the reference's operation order, from the record itself
                        const double* rp = rec.f + (size_t)REC_ROW * (size_t)__double_as_longlong(row[8]);
                        const d3 qq = mk(rp[0], rp[1], rp[2]), g0 = mk(rp[3], rp[4], rp[5]);
                        const d3 dir = mk(t.x[0], t.y[0], t.z[0]) - qq;
                        const double d2 = len2(dir);
                        const bool o = inc[0] && (d2 < tp.gather_r2) && (d2 <= t.r2[0]) && (dot(hn[0], g0) > 1e-3);
                        const double s1 = dot(g0, dir);
                        const double l2 = s1 * s1, rhs = thr * thr * d2;
                        const double band = rhs * band_rel;
                        bool flat = l2 < rhs - band;
                        if (o && !flat && !(l2 > rhs + band)) {
                            const double dt = dot(g0, dir / sqrt(d2));  // NaN at d2 == 0: rejected as in the reference
                            flat = fabs(dt) <= thr;
                        }
                        if (o && flat) DEP_TAKE(row)
                    }
                }
            }
        }
#undef DEP_TAKE
        // rows [64 - n_full, 64): the full test
#ifdef DEP_EXP_SKIP_FULL
        if (false)
#endif
        if constexpr (F32) {
            // Single precision first, two rows per packed instruction (the loop is bound by the vector instructions issued:
            // ~19 per row in this form against ~45); a row with a pair inside one of the bands is evaluated again in the
            // reference's operation order from its double-precision row.
            const v2f mx2 = {m2xf, m2xf}, my2 = {m2yf, m2yf}, mz2 = {m2zf, m2zf}, half2 = {0.5f, 0.5f};
            const v2f nx2 = {hnxf, hnxf}, ny2 = {hnyf, hnyf}, nz2 = {hnzf, hnzf}, tl2 = {tlo2f, tlo2f}, th2 = {thi2f, thi2f};
            for (int tj = 0; tj < nf_u; tj += 2) {
                const bool two = tj + 1 < nf_u;
                const double* rowA = s_row[wv][DEP_ROWS - 1 - tj];
                const double* rowB = s_row[wv][two ? DEP_ROWS - 2 - tj : DEP_ROWS - 1 - tj];
                const float* pp = &s_fp[wv][tj >> 1][0];
                const v4f PA = *(const v4f*)pp, PB = *(const v4f*)(pp + 4), PC = *(const v4f*)(pp + 8);
                double fAx = rowA[6], fAy = rowA[7], fAz = rowA[8], fBx = rowB[6], fBy = rowB[7], fBz = rowB[8];
                const v2f ex = __builtin_elementwise_fma(mx2, half2, PA.xy), ey = __builtin_elementwise_fma(my2, half2, PA.zw), ez = __builtin_elementwise_fma(mz2, half2, PB.xy);  // q' - h'
                const v2f gx = PB.zw, gy = PC.xy, gz = PC.zw;
                const v2f d2 = __builtin_elementwise_fma(ez, ez, __builtin_elementwise_fma(ey, ey, ex * ex));
                const v2f s1 = __builtin_elementwise_fma(gz, ez, __builtin_elementwise_fma(gy, ey, gx * ex));
                const v2f nd = __builtin_elementwise_fma(nz2, gz, __builtin_elementwise_fma(ny2, gy, nx2 * gx));
                const v2f l2 = s1 * s1;
                const v2f sp = __builtin_elementwise_fma(tl2, d2, -l2), sn = __builtin_elementwise_fma(th2, d2, -l2);  // (thr^2 -+ B) d2 - (g.e)^2
                asm volatile("" : "+v"(fAx), "+v"(fAy), "+v"(fAz), "+v"(fBx), "+v"(fBy), "+v"(fBz));
                bool okA = d2.x < dlo32 && nd.x > 1e-3f + 4e-6f && sp.x > ef32;
                bool okB = d2.y < dlo32 && nd.y > 1e-3f + 4e-6f && sp.y > ef32;
                const bool unA = inc[0] && !okA && !(d2.x > dhi32 || nd.x < 1e-3f - 4e-6f || sn.x < -ef32);
                const bool unB = inc[0] && !okB && !(d2.y > dhi32 || nd.y < 1e-3f - 4e-6f || sn.y < -ef32);
#define DEP_EXACT(rowp, okv)                                                                                           \
                {                                                                                                      \
                    /* (the hitpoint is read again: its double-precision state is not kept in registers through the loops) */ \
                    int pr_ = t.p[0];                                                                                  \
                    asm volatile("" : "+v"(pr_));                                                                      \
                    const d3 hp_ = mk(hp.pos[pr_], hp.pos[P + pr_], hp.pos[2 * P + pr_]), hn_ = mk(hp.nrm[pr_], hp.nrm[P + pr_], hp.nrm[2 * P + pr_]); \
                    const double hr2_ = hp.r2[pr_];  /* (all seven loads at once, not one behind each test) */         \
                    const d3 g0 = mk((rowp)[3], (rowp)[4], (rowp)[5]);                                                 \
                    const d3 dir = hp_ - mk((rowp)[0], (rowp)[1], (rowp)[2]);                                          \
                    const double d2e = len2(dir);                                                                      \
                    const bool o = inc[0] & (d2e < tp.gather_r2) & (d2e <= hr2_) & (dot(hn_, g0) > 1e-3);              \
                    const double s1e = dot(g0, dir);                                                                   \
                    const double l2 = s1e * s1e, rhs = thr * thr * d2e;                                                \
                    const double band = rhs * band_rel;                                                                \
                    bool flat = l2 < rhs - band;                                                                       \
                    if (o && !flat && !(l2 > rhs + band)) {                                                            \
                        const double dt = dot(g0, dir / sqrt(d2e)); /* NaN at d2 == 0: rejected as in the reference */ \
                        flat = fabs(dt) <= thr;                                                                        \
                    }                                                                                                  \
                    okv = o && flat;                                                                                   \
                }
                if (ballot64(unA) != 0ull) { if (STAMPS) d_near_slow++; DEP_EXACT(rowA, okA) }
                if (two && ballot64(unB) != 0ull) { if (STAMPS) d_near_slow++; DEP_EXACT(rowB, okB) }
#undef DEP_EXACT
                if (okA) {
                    ax[0] += fAx; ay[0] += fAy; az[0] += fAz; am[0]++;
                    if (TAP) {
                        const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);
                        if ((int64_t)ts < tp.tap_cap) { tp.tap_key[ts] = (unsigned long long)__double_as_longlong(rowA[ROW - 1]); tp.tap_pix[ts] = t.p[0]; }
                    }
                }
                if (two && okB) {
                    ax[0] += fBx; ay[0] += fBy; az[0] += fBz; am[0]++;
                    if (TAP) {
                        const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);
                        if ((int64_t)ts < tp.tap_cap) { tp.tap_key[ts] = (unsigned long long)__double_as_longlong(rowB[ROW - 1]); tp.tap_pix[ts] = t.p[0]; }
                    }
                }
            }
        } else
        for (int ti = DEP_ROWS - nf_u; ti < DEP_ROWS; ti++) {
            const double* row = s_row[wv][ti];
            const d3 g0 = mk(row[3], row[4], row[5]);
            const d3 qq = mk(row[0], row[1], row[2]);
            bool ok[DEP_H];
            bool unsure = false;
#pragma unroll
            for (int h = 0; h < DEP_H; h++) {  // fast path (fused)
                const d3 dir = mk(t.x[h], t.y[h], t.z[h]) - qq;
                const double d2f = fma(dir.z, dir.z, fma(dir.y, dir.y, dir.x * dir.x));
                const double ndf = fma(hn[h].z, g0.z, fma(hn[h].y, g0.y, hn[h].x * g0.x));
                const double s1f = fma(g0.z, dir.z, fma(g0.y, dir.y, g0.x * dir.x));
                const double lhs = s1f * s1f;
                const bool in_d = d2f < lim_lo[h], out_d = d2f > lim_hi[h];     // both false: too close to call
                const bool in_n = ndf > 1e-3 + 1e-13, out_n = ndf < 1e-3 - 1e-13;
                const bool in_f = lhs < thr2_lo * d2f, out_f = lhs > thr2_hi * d2f;
                ok[h] = in_d && in_n && in_f;
                unsure = unsure || (inc[h] && !(out_d || out_n || out_f) && !ok[h]);
            }
            if (ballot64(unsure) != 0ull) {
                // the reference's operation order for this record (every hitpoint: the result replaces the fast one)
#pragma unroll
                for (int h = 0; h < DEP_H; h++) {
                    const d3 dir = mk(t.x[h], t.y[h], t.z[h]) - qq;
                    const double d2 = len2(dir);
                    bool o = inc[h] && (d2 < tp.gather_r2) && (d2 <= t.r2[h]) && (dot(hn[h], g0) > 1e-3);
                    const double s1 = dot(g0, dir);
                    const double l2 = s1 * s1, rhs = thr * thr * d2;
                    const double band = rhs * band_rel;
                    bool flat = l2 < rhs - band;
                    if (o && !flat && !(l2 > rhs + band)) {
                        const double dt = dot(g0, dir / sqrt(d2));  // NaN at d2 == 0: rejected as in the reference
                        flat = fabs(dt) <= thr;
                    }
                    ok[h] = o && flat;
                }
            }
            bool any_ok = false;
#pragma unroll
            for (int h = 0; h < DEP_H; h++) any_ok = any_ok || ok[h];
            if (any_ok) {
                const double fx = row[6], fy = row[7], fz = row[8];
#pragma unroll
                for (int h = 0; h < DEP_H; h++) {
                    if (ok[h] && (!VETO || !veto_has(tp.veto, (unsigned long long)__double_as_longlong(row[ROW - 1]), t.p[h]))) {
                        ax[h] += fx; ay[h] += fy; az[h] += fz; am[h]++;
                        if (TAP) {
                            const unsigned long long ts = atomicAdd(tp.tap_n, 1ull);
                            if ((int64_t)ts < tp.tap_cap) { tp.tap_key[ts] = (unsigned long long)__double_as_longlong(row[ROW - 1]); tp.tap_pix[ts] = t.p[h]; }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        sel_n = 0;
        st_process += DSTAMP() - tp0;
    }
#undef DEP_ISSUE
    const unsigned long long tf0 = DSTAMP();
    DEP_FLUSH()
#undef DEP_FLUSH
    // (the tallies of an item are kept by the wavefront and added to the global counters once, when it runs out of items:
    // five atomics per item on one cache line, every wavefront waiting for each of its own, were a third of the kernel)
    w_pairs += c_pairs; w_acc += c_acc; w_sel += c_sel; w_round += c_round; w_items++;
    if (lane == 0) {
        if (STAMPS) { atomicAdd(&counters[22], d_easy); atomicAdd(&counters[15], d_full); atomicAdd(&counters[8], d_near); atomicAdd(&counters[9], d_near_slow); atomicAdd(&counters[10], d_win); atomicAdd(&counters[11], d_grp); atomicAdd(&counters[12], d_load); atomicAdd(&counters[13], (unsigned long long)Ktot); }
    }
    __builtin_amdgcn_wave_barrier();
    st_flush += DSTAMP() - tf0;
    }  // next item
    for (int off = 32; off > 0; off >>= 1) w_acc += __shfl_down(w_acc, off);
    if (lane == 0) {
        if (w_pairs) atomicAdd(&counters[3], w_pairs);
        if (w_acc) atomicAdd(&counters[4], (unsigned long long)w_acc);
        if (w_sel) atomicAdd(&counters[5], (unsigned long long)w_sel);
        if (w_round) atomicAdd(&counters[6], (unsigned long long)w_round);
        if (w_items) atomicAdd(&counters[7], (unsigned long long)w_items);
    }
    if (STAMPS && lane == 0) {
        atomicAdd(&counters[16], st_setup); atomicAdd(&counters[17], st_count); atomicAdd(&counters[18], st_issue);
        atomicAdd(&counters[19], st_append); atomicAdd(&counters[20], st_process); atomicAdd(&counters[21], st_flush); atomicAdd(&counters[23], st_wait);
    }
#undef DSTAMP
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

// k_end_pass and k_resolve (sppm_image.hpp) in one sweep over the hitpoints: the image reads what the update has just
// computed from registers instead of from memory in a second launch.  Same operations, same results.
// Hitpoints [first, first + n) by image index (all of them, or one slice of the sliced all-reduce); image == nullptr:
// the update alone.
__global__ void k_end_pass_resolve(HpArrays hp, DevScene sc, Own own, double alpha, double fx_inv, int32_t* __restrict__ err, int pass, int64_t photon_num,
                                   double* __restrict__ image, int first, int n)
{
    __builtin_amdgcn_s_setprio(3);  // (as k_deposit_plan: the next gather waits for it)
    const int p = first + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const size_t P = (size_t)hp.P;
    if (p >= first + n || p >= hp.P) return;
    d3 c = mk(hp.ld[p], hp.ld[P + p], hp.ld[2 * P + p]) / (double)(pass + 1);
    if (hp.valid[p]) {
        d3 tau = mk(hp.tau[p], hp.tau[P + p], hp.tau[2 * P + p]);
        double r2 = hp.r2[p];
        if (owned(own, p)) {
            const long long madd = hp.dep[3 * P + p];
            if (madd > 0) {
                if (madd >= FX_MAX_DEPOSITS) err[6] = 1;
                const double N = hp.nd[p];
                const double Nnew = N + alpha * (double)madd;
                const double Rnew2 = r2 * Nnew / (N + (double)madd);
                const d3 phi = had(mk((double)hp.dep[p] * fx_inv, (double)hp.dep[P + p] * fx_inv, (double)hp.dep[2 * P + p] * fx_inv), ld3(sc.sp[hp.obj[p]].color) / SPPM_PI);
                const d3 w = mk(hp.w[p], hp.w[P + p], hp.w[2 * P + p]);
                tau = (tau + had(phi, w)) * Rnew2 / r2;
                r2 = Rnew2;
                hp.tau[p] = tau.x; hp.tau[P + p] = tau.y; hp.tau[2 * P + p] = tau.z;
                hp.nd[p] = Nnew; hp.r2[p] = Rnew2;
                hp.dep[p] = 0; hp.dep[P + p] = 0; hp.dep[2 * P + p] = 0; hp.dep[3 * P + p] = 0;
            }
            hp.w[p] = 0.0; hp.w[P + p] = 0.0; hp.w[2 * P + p] = 0.0;
        }
        c = c + tau * (1.0 / (SPPM_PI * r2 * (double)((int64_t)(pass + 1) * photon_num)));
    }
    if (image) { image[3 * (size_t)p] = c.x; image[3 * (size_t)p + 1] = c.y; image[3 * (size_t)p + 2] = c.z; }
}
