// sppm_gather_tiles.hpp -- the tile gather: one wavefront replays, for a tile of 64 spatially
// neighbouring hitpoints, every photon-hit record near the tile in key (= photon, bounce) order.
// Tiles are 64 consecutive hitpoints of this rank's pixels sorted along a Hilbert curve of their
// positions (re-sorted after every eye pass): screen-space 8x8 tiles are compact on directly seen
// diffuse surfaces but scattered over the whole scene behind the mirror and glass spheres.
//
// Included by sppm_kernels.hip.  Semantics are exactly those of k_gather (the simple per-hitpoint
// kernel kept as a cross-check): the search filter of src/kdtree.h:123-127, the acceptance test and
// the per-photon update of src/sppm2.h:391-421 (rule 0) or the deposit of
// src/stochastic_ppm.h:439-456 (rule 1), applied per hitpoint in ascending key order.
//
// How the order is obtained without sorting records by time globally:
//   * records of a batch are counting-sorted by (slab, cell): slab = photon index / B (time),
//     cell = dense grid cell of size c (space).  Inside a bucket the order is arbitrary.
//   * a tile walks the slabs in order.  For a slab it looks up the cells that overlap its region
//     (tile bounding box grown by the largest radius of the tile), pulls the candidate records that
//     fall inside the region into an LDS staging list of at most 64, ranks them by key (counting
//     rank, one candidate per lane) and applies them in rank order: the record is broadcast from
//     its lane with v_readlane and each lane tests it against its own hitpoint.
//   * more than 64 candidates in a slab: the slab's key range is halved until it fits (keys are
//     unique, so this terminates); the next attempt takes the rest of the slab.
// Everything that steers control flow (cell ranges, counts, ranks, source lanes) is wave-uniform.
#pragma once

struct TileGrid {
    double ox, oy, oz, inv_c;
    int nx, ny, nz, ncells;
    int S;          // slabs in this batch
    int64_t B;      // photons per slab
    int64_t first;  // first photon index of the batch
};

struct TileParams {
    TileGrid g;
    double gather_r2, alpha;
    int rule;
    const int32_t* order;  // this rank's image indices, sorted along a Hilbert curve of the hitpoint positions
    int n_pix, n_owned;    // entries of order[]; virtual tiles = ceil(n_pix / 64)
    double cluster_d;      // lanes farther than this (max norm) from a tile's seed lane are replayed separately
    uint64_t* tap_key; int32_t* tap_pix; unsigned long long* tap_n; int64_t tap_cap;
    Veto veto;  // pairs dropped by the reference's K cap (sppm_cap.hpp); the ordered gather only
    unsigned long long* item_times;  // diagnostic (SPPM_ITEM_TIMES): start/end clock of every work item
};

// (rl_d, first_d, wave_min_perm_d, wave_max_perm_d: sppm_device.hpp)
// grid coordinate of x, clamped to [0, n-1]; monotone in x, so an interval maps to an index range
__device__ __forceinline__ int grid_coord(double x, double o, double inv_c, int n)
{
    // (clamped as an integer: a `(double)(n - 1)` per axis is a wave-uniform value the compiler keeps in vector registers)
    const double f = fmin(fmax(floor((x - o) * inv_c), 0.0), 2147483000.0);
    const int i = (int)f;
    return i > n - 1 ? n - 1 : i;
}

__device__ __forceinline__ int tile_of_pixel(int p, int W, int tiles_x) { return ((p / W) >> 3) * tiles_x + ((p % W) >> 3); }

// ---- spatial order of the hitpoints ------------------------------------------------------------------
// bounding box of the valid hitpoints among pix[0..n): bb[0..2] = min, bb[3..5] = max (ordered ints)
__global__ void k_hp_bbox(HpArrays hp, const int32_t* __restrict__ pix, int n, unsigned long long* __restrict__ bb)
{
    const double BIG = 1e300;
    double lo[3] = {BIG, BIG, BIG}, hi[3] = {-BIG, -BIG, -BIG};
    const int P = hp.P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = pix[i];
        if (!hp.valid[p]) continue;
        for (int a = 0; a < 3; a++) { const double v = hp.pos[a * P + p]; lo[a] = fmin(lo[a], v); hi[a] = fmax(hi[a], v); }
    }
    __shared__ double s_lo[4][3], s_hi[4][3];
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], off)); }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {  // one pair of atomics per block and axis (launched with 256 threads)
        const int a = threadIdx.x;
        const double l = fmin(fmin(s_lo[0][a], s_lo[1][a]), fmin(s_lo[2][a], s_lo[3][a]));
        const double h = fmax(fmax(s_hi[0][a], s_hi[1][a]), fmax(s_hi[2][a], s_hi[3][a]));
        if (l <= h) { atomicMin(&bb[a], ord_d(l)); atomicMax(&bb[3 + a], ord_d(h)); }
    }
}
__global__ void k_hp_sortkeys(HpArrays hp, const int32_t* __restrict__ pix, int n, const unsigned long long* __restrict__ bb,
                              hp_key_t* __restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pix[i], P = hp.P;
    if (!hp.valid[p]) { keys[i] = MORTON_INVALID; return; }
    const hp_key_t code = hp_order_key(hp.pos[p], hp.pos[P + p], hp.pos[2 * P + p], bb);
    keys[i] = code;
}

// (slab, cell) bucket of every record of the batch
// nm.cells != nullptr: records that cannot matter to this rank (near_rank, sppm_exchange.hpp) are dropped here.  Only
// used when the occupancy map is known to be current (binning on the context's stream).
__global__ void k_bin_count(RecArrays rec, const unsigned long long* __restrict__ rec_count, TileGrid g, uint32_t* __restrict__ cell_cnt,
                            uint32_t* __restrict__ rec_bucket, int32_t* __restrict__ err, NearMap nm)
{
    const int64_t n = min((int64_t)*rec_count, rec.cap);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (rec.key[i] == ~0ull) { rec_bucket[i] = 0xFFFFFFFFu; continue; }  // a hole left by the tracer's chunked slot reservation
        if (nm.cells && !near_rank(nm, rec.f[i], rec.f[rec.cap + i], rec.f[2 * rec.cap + i])) { rec_bucket[i] = 0xFFFFFFFFu; continue; }
        const int ix = grid_coord(rec.f[i], g.ox, g.inv_c, g.nx);
        const int iy = grid_coord(rec.f[rec.cap + i], g.oy, g.inv_c, g.ny);
        const int iz = grid_coord(rec.f[2 * rec.cap + i], g.oz, g.inv_c, g.nz);
        int64_t slab = ((int64_t)(rec.key[i] >> KEY_SHIFT) - g.first) / g.B;
        if (slab < 0 || slab >= g.S) { err[3] = 1; slab = slab < 0 ? 0 : g.S - 1; }  // a record outside the declared photon range
        const uint32_t b = (uint32_t)slab * (uint32_t)g.ncells + (uint32_t)((iz * g.ny + iy) * g.nx + ix);
        rec_bucket[i] = b;
        atomicAdd(&cell_cnt[b], 1u);
    }
}

// Global statistics of the owned tiles, to size the grid: bounding box of all hitpoints (ordered-int
// encoded doubles), a histogram of log2(tile region extent) and the number of live tiles.
struct TileStats { unsigned int hist[64]; unsigned int live_tiles; unsigned int pad; unsigned long long est_sum; };
// extent (largest bounding-box side + twice the largest radius) of every tile, as log2 bin + 1 (0 = no live lane)
__global__ void __launch_bounds__(256) k_tile_extent(HpArrays hp, TileParams tp, uint8_t* __restrict__ ext_bin)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wv;
    if (k >= tp.n_owned) return;
    const int slot = k * 64 + lane;
    const int P = hp.P;
    const int p = slot < tp.n_pix ? tp.order[slot] : 0;
    bool live = slot < tp.n_pix;
    if (live) live = hp.valid[p] != 0;
    if (__ballot(live) == 0) { if (lane == 0) ext_bin[k] = 0; return; }
    const double BIG = 1e300;
    const double x = live ? hp.pos[p] : 0.0, y = live ? hp.pos[P + p] : 0.0, z = live ? hp.pos[2 * P + p] : 0.0;
    const double r = live ? sqrt(fmin(hp.r2[p], tp.gather_r2)) : 0.0;
    const double x0 = wave_min_perm_d(live ? x : BIG), x1 = wave_max_perm_d(live ? x : -BIG);
    const double y0 = wave_min_perm_d(live ? y : BIG), y1 = wave_max_perm_d(live ? y : -BIG);
    const double z0 = wave_min_perm_d(live ? z : BIG), z1 = wave_max_perm_d(live ? z : -BIG);
    const double rm = wave_max_perm_d(r);
    if (lane == 0) {
        const double ext = fmax(x1 - x0, fmax(y1 - y0, z1 - z0)) + 2.0 * rm;
        int e;
        frexp(ext, &e);  // ext in [2^(e-1), 2^e)
        e = e + 32;
        e = e < 0 ? 0 : (e > 62 ? 62 : e);
        ext_bin[k] = (uint8_t)(e + 1);
    }
}
// one block: histogram of the tile extents, and the sum of the per-tile record estimates of the last plan
__global__ void __launch_bounds__(1024) k_tile_hist(const uint8_t* __restrict__ ext_bin, const uint32_t* __restrict__ est, int n, TileStats* __restrict__ st)
{
    __shared__ unsigned int h[64];
    __shared__ unsigned int live;
    __shared__ unsigned long long esum;
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) { live = 0; esum = 0; }
    __syncthreads();
    unsigned long long mysum = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int b = ext_bin[i];
        if (b) { atomicAdd(&h[b - 1], 1u); atomicAdd(&live, 1u); }
        if (est) mysum += est[i];
    }
    for (int off = 32; off > 0; off >>= 1) mysum += __shfl_xor(mysum, off);
    if ((threadIdx.x & 63) == 0 && mysum) atomicAdd(&esum, mysum);
    __syncthreads();
    if (threadIdx.x < 64) st->hist[threadIdx.x] = h[threadIdx.x];
    if (threadIdx.x == 0) { st->live_tiles = live; st->est_sum = esum; }
}

// what the host reads one batch late, written straight into its pinned slot (one launch instead of three copies)
struct StatSnapshot { TileStats hs; unsigned long long bbq[6]; unsigned long long cnt[24]; };
__global__ void k_stat_snapshot(const TileStats* __restrict__ st, const unsigned long long* __restrict__ bb, const unsigned long long* __restrict__ counters,
                                StatSnapshot* __restrict__ out)
{
    const int t = threadIdx.x;
    if (t < 64) out->hs.hist[t] = st->hist[t];
    if (t == 0) { out->hs.live_tiles = st->live_tiles; out->hs.pad = 0; out->hs.est_sum = st->est_sum; }
    if (t < 6) out->bbq[t] = bb[t];
    if (t < 24) out->cnt[t] = counters[t];
}

// records per cell summed over the slabs of the batch (what the plan's estimates read)
__global__ void k_cell_totals(const uint32_t* __restrict__ cell_start, int ncells, int S, uint32_t* __restrict__ cell_tot)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    uint32_t t = 0;
    for (int s = 0; s < S; s++) { const size_t b = (size_t)s * ncells + c; t += cell_start[b + 1] - cell_start[b]; }
    cell_tot[c] = t;
}

// Summed-area table over the per-cell totals, in place: after the three passes cell (x, y, z) holds the
// records of all cells (<= x, <= y, <= z), so the plan gets a region's record count from 8 corners.
__global__ void k_sat_axis(uint32_t* __restrict__ t, int nx, int ny, int nz, int axis)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_lines = axis == 0 ? ny * nz : (axis == 1 ? nx * nz : nx * ny);
    if (i >= n_lines) return;
    size_t base, stride; int len;
    if (axis == 0) { base = (size_t)i * nx; stride = 1; len = nx; }                                    // i = z*ny + y
    else if (axis == 1) { base = (size_t)(i / nx) * nx * ny + (size_t)(i % nx); stride = (size_t)nx; len = ny; }  // i = z*nx + x
    else { base = (size_t)i; stride = (size_t)nx * ny; len = nz; }                                     // i = y*nx + x
    // (a thread's line is a chain of dependent additions, not of dependent loads: 16 elements are in flight at a time)
    uint32_t run = 0;
    for (int k = 0; k < len; k += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = k + j < len ? t[base + (size_t)(k + j) * stride] : 0u;
#pragma unroll
        for (int j = 0; j < 16; j++) { run += v[j]; v[j] = run; }
#pragma unroll
        for (int j = 0; j < 16; j++) if (k + j < len) t[base + (size_t)(k + j) * stride] = v[j];
    }
}
// records in cells [x0..x1] x [y0..y1] x [z0..z1]; lanes 0..7 fetch one corner each (wave-uniform result)
__device__ __forceinline__ unsigned int sat_box(const uint32_t* __restrict__ sat, int nx, int ny, int x0, int x1, int y0, int y1, int z0, int z1, int lane)
{
    int v = 0;
    if (lane < 8) {
        const int x = (lane & 1) ? x1 : x0 - 1, y = (lane & 2) ? y1 : y0 - 1, z = (lane & 4) ? z1 : z0 - 1;
        if (x >= 0 && y >= 0 && z >= 0) {
            const int val = (int)sat[((size_t)z * ny + y) * nx + x];
            v = (__popc(lane & 7) & 1) ? val : -val;   // + for an odd number of upper corners, - otherwise (3 -> +, 2 -> -, 1 -> +, 0 -> -)
        }
    }
    for (int off = 4; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return (unsigned int)__builtin_amdgcn_readfirstlane(v);
}

// ---- work items: (tile, lane range) ---------------------------------------------------------------------
// Record density is very uneven (caustics, the gap between the light and the ceiling hold a few
// percent of all records in a few square units).  A tile there has ~100x the average work, and a tile
// is replayed by ONE wavefront, so it would set the kernel's duration.  k_tile_est counts the records
// in each tile's cells; k_tile_emit cuts tiles that exceed a threshold into 2..64 contiguous lane
// ranges (consecutive lanes are consecutive on the Hilbert curve, so a range is compact and its region
// smaller); each range becomes a work item replayed by its own wavefront.  Results do not depend on
// the cut: every hitpoint still sees every record near it, in key order.
__device__ __forceinline__ uint32_t pack_item(int tile, int lane0, int lanes) { return ((uint32_t)tile << 12) | ((uint32_t)lane0 << 6) | (uint32_t)(lanes - 1); }

// One wavefront per tile: count the records in the cells of the tile's region; if that exceeds the
// threshold, halve the lane range and look at each half's own (smaller) region, down to single lanes.
// Tile k writes its pieces to raw[k*64 ..] as (record estimate << 32 | item) and their number to
// n_pieces[k]: no shared counters here.  Cost class of a piece: c = floor(log2(estimate | 1)).
#define ITEM_CLASSES 32
__device__ __forceinline__ int item_class(unsigned int est) { return 31 - __clz((int)(est | 1u)); }
__global__ void __launch_bounds__(256) k_tile_plan(HpArrays hp, const uint32_t* __restrict__ cell_tot, TileParams tp, uint32_t threshold,
                                                   unsigned long long* __restrict__ raw, uint8_t* __restrict__ n_pieces,
                                                   uint32_t* __restrict__ est)
{
    __shared__ uint32_t s_stack[4][16];
    __builtin_amdgcn_s_setprio(3);  // (the gather waits for this kernel while the tracer and the eye pass run beside it: as k_deposit_plan)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wv;
    if (k >= tp.n_owned) return;
    const int slot = k * 64 + lane;
    const int P = hp.P;
    const int p = slot < tp.n_pix ? tp.order[slot] : 0;
    bool live = slot < tp.n_pix;
    if (live) live = hp.valid[p] != 0;
    if (__ballot(live) == 0) { if (lane == 0) { est[k] = 0; n_pieces[k] = 0; } return; }  // a tile without live hitpoints: no item
    const double BIG = 1e300;
    const double x = live ? hp.pos[p] : 0.0, y = live ? hp.pos[P + p] : 0.0, z = live ? hp.pos[2 * P + p] : 0.0;
    const double rr = live ? sqrt(fmin(hp.r2[p], tp.gather_r2)) * (1.0 + 1e-9) + 1e-9 : 0.0;
    int sp = 0;
    if (lane == 0) s_stack[wv][0] = (0u << 8) | 64u;  // (first lane, lanes)
    sp = 1;
    bool first = true;
    int n_out = 0;
    while (sp > 0) {
        __builtin_amdgcn_wave_barrier();
        const uint32_t top = s_stack[wv][sp - 1];
        __builtin_amdgcn_wave_barrier();
        sp--;
        const int a = (int)(top >> 8), len = (int)(top & 0xffu);
        const bool in = live && lane >= a && lane < a + len;
        if (__ballot(in) == 0) continue;
        // the gather replays the range cluster by cluster (same rule as k_gather_tiles): its cost is the
        // sum of the records in the clusters' regions, which for a scattered range is far more than the
        // records in the range's bounding box
        unsigned int sum = 0;
        unsigned long long remaining = __ballot(in);
        while (remaining) {
            const int seed = __ffsll((long long)remaining) - 1;
            const double sx = rl_d(x, seed), sy = rl_d(y, seed), sz = rl_d(z, seed);
            const bool inc = in && ((remaining >> lane) & 1ull) && fabs(x - sx) <= tp.cluster_d && fabs(y - sy) <= tp.cluster_d && fabs(z - sz) <= tp.cluster_d;
            remaining &= ~__ballot(inc);
            const double rm = wave_max_perm_d(inc ? rr : 0.0);
            const int cx0 = grid_coord(wave_min_perm_d(inc ? x : BIG) - rm, tp.g.ox, tp.g.inv_c, tp.g.nx), cx1 = grid_coord(wave_max_perm_d(inc ? x : -BIG) + rm, tp.g.ox, tp.g.inv_c, tp.g.nx);
            const int cy0 = grid_coord(wave_min_perm_d(inc ? y : BIG) - rm, tp.g.oy, tp.g.inv_c, tp.g.ny), cy1 = grid_coord(wave_max_perm_d(inc ? y : -BIG) + rm, tp.g.oy, tp.g.inv_c, tp.g.ny);
            const int cz0 = grid_coord(wave_min_perm_d(inc ? z : BIG) - rm, tp.g.oz, tp.g.inv_c, tp.g.nz), cz1 = grid_coord(wave_max_perm_d(inc ? z : -BIG) + rm, tp.g.oz, tp.g.inv_c, tp.g.nz);
            sum += sat_box(cell_tot, tp.g.nx, tp.g.ny, cx0, cx1, cy0, cy1, cz0, cz1, lane);
            // every slab of every cluster walks the region's cells whether they hold records or not: a fixed
            // part worth ~16 records, and 64 cells ~ 16 records of work
            sum += (unsigned int)((16 + (((long long)(cx1 - cx0 + 1) * (cy1 - cy0 + 1) * (cz1 - cz0 + 1)) >> 2)) * tp.g.S);
        }
        if (first) { if (lane == 0) est[k] = sum; first = false; }
        if (sum <= threshold || len == 1) {
            if (lane == 0) raw[(size_t)k * 64 + n_out] = ((unsigned long long)sum << 32) | pack_item(k, a, len);
            n_out++;
        } else {
            if (lane == 0) { s_stack[wv][sp] = ((uint32_t)(a + len / 2) << 8) | (uint32_t)(len / 2); s_stack[wv][sp + 1] = ((uint32_t)a << 8) | (uint32_t)(len / 2); }
            sp += 2;
        }
    }
    if (lane == 0) n_pieces[k] = (uint8_t)n_out;  // 1..64
}

// Order the pieces by cost class, heaviest first (a counting sort; the order inside a class is
// arbitrary).  A tile is replayed by ONE wavefront and the cost per tile spans three orders of magnitude, so
// the long items must start first for the kernel not to end in a tail of a few waves.
// q: [0] number of items, [1] the gather's queue head, [2 + c] count, then write cursor, of class c.
__global__ void __launch_bounds__(256) k_items_hist(const unsigned long long* __restrict__ raw, const uint8_t* __restrict__ n_pieces, int n_tiles,
                                                    unsigned int* __restrict__ q)
{
    __shared__ unsigned int h[ITEM_CLASSES];
    if (threadIdx.x < ITEM_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = k < n_tiles ? n_pieces[k] : 0;
    for (int j = 0; j < n; j++) atomicAdd(&h[item_class((unsigned int)(raw[(size_t)k * 64 + j] >> 32))], 1u);
    __syncthreads();
    if (threadIdx.x < ITEM_CLASSES && h[threadIdx.x]) { atomicAdd(&q[2 + threadIdx.x], h[threadIdx.x]); atomicAdd(&q[0], h[threadIdx.x]); }
}
// (also clears the record counter of the batch's buffer: its records are binned by now, and a one-word memset queued
// behind the gather would be one more launch between two gathers)
__global__ void k_items_bases(unsigned int* __restrict__ q, unsigned long long* __restrict__ rec_count)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *rec_count = 0ull;
    unsigned int run = 0;
    for (int c = ITEM_CLASSES - 1; c >= 0; c--) { const unsigned int n = q[2 + c]; q[2 + c] = run; run += n; }
    q[1] = 0;
}
__global__ void __launch_bounds__(256) k_items_scatter(const unsigned long long* __restrict__ raw, const uint8_t* __restrict__ n_pieces, int n_tiles,
                                                       unsigned int* __restrict__ q, uint32_t* __restrict__ items)
{
    __shared__ unsigned int h[ITEM_CLASSES], base[ITEM_CLASSES];
    if (threadIdx.x < ITEM_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = k < n_tiles ? n_pieces[k] : 0;
    for (int j = 0; j < n; j++) atomicAdd(&h[item_class((unsigned int)(raw[(size_t)k * 64 + j] >> 32))], 1u);
    __syncthreads();
    if (threadIdx.x < ITEM_CLASSES) { base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&q[2 + threadIdx.x], h[threadIdx.x]) : 0u; h[threadIdx.x] = 0; }
    __syncthreads();
    for (int j = 0; j < n; j++) {
        const unsigned long long w = raw[(size_t)k * 64 + j];
        const int c = item_class((unsigned int)(w >> 32));
        items[base[c] + atomicAdd(&h[c], 1u)] = (uint32_t)w;
    }
}

#ifndef GATHER_WAVES
#define GATHER_WAVES 4
#endif
#if GATHER_WAVES == 4
#define GATHER_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define GATHER_WAVES_ATTR
#endif
// UNIT: classic PPM -- the record's flux is applied as it is (no hitpoint weight or colour),
// src/progressive_photonmapping.h:385-387
#ifndef GT_E32_SCALE
#define GT_E32_SCALE 1.0  // (a build with 3e3 here sends a large share of the candidates through the exact fallback: how that path was validated)
#endif
template <bool STAMPS, bool UNIT>
__global__ void __launch_bounds__(256) GATHER_WAVES_ATTR k_gather_tiles(HpArrays hp, DevScene sc, RecArrays rec, const uint32_t* __restrict__ cell_start,
                                                      TileParams tp, const uint32_t* __restrict__ items, unsigned int* __restrict__ q,
                                                      unsigned long long* __restrict__ counters)
{
    __shared__ unsigned long long s_key[4][64];
    __shared__ __align__(16) double s_row[4][64][10];  // the round's candidates in key order: position, geometric normal, key, flux
    __shared__ __align__(16) float s_fp[4][32][12];  // the same in single precision about the cluster's centre, two candidates per entry: (x0 x1 y0 y1)(z0 z1 gx0 gx1)(gy0 gy1 gz0 gz1)
    const unsigned long long t_entry = STAMPS ? wall_clock64() : 0ull;  // 100 MHz, one clock for the whole device
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int P = hp.P;
    const double BIG = 1e300;
#ifdef GT_PIN
    const double thr = pin_scalar_d(tp.gather_r2 * 0.01);
    const double band_rel = pin_scalar_d(1e-9 + 8e-15 / thr);
#else
    const double thr = tp.gather_r2 * 0.01;
    const double band_rel = 1e-9 + 8e-15 / thr;
#endif
    // Persistent wavefronts: wave w starts with item w (thousands of waves asking one counter at the same
    // moment would queue up for a millisecond), then each takes the next item of the ordered list until the
    // list is empty (q[0] = number of items, q[1] = queue head beyond the first gridDim.x*4; every wave
    // reaches the break).
    const unsigned int n_items = q[0];
    const unsigned int n_waves = gridDim.x * 4u;
    const bool vetoes = veto_active(tp.veto);
    unsigned long long w_pairs = 0, w_cyc = 0, m_cyc = 0, m_selk = 0;
    unsigned int w_cand = 0, w_acc = 0, w_sel = 0, w_round = 0, w_items = 0, m_sel = 0, m_cand = 0, m_round = 0;  // (w_acc: per lane; a wavefront's share of a launch stays far below 2^32)
    for (bool first_item = true;; first_item = false) {
    unsigned int item_i = blockIdx.x * 4u + (unsigned int)wv;
    if (!first_item) {
        if (lane == 0) item_i = n_waves + atomicAdd(&q[1], 1u);
        item_i = (unsigned int)__builtin_amdgcn_readfirstlane((int)item_i);
    }
    if (item_i >= n_items) break;
    // the heaviest few percent run at raised priority: they share their SIMD with three other waves and set the kernel's end
    if (item_i < (n_items >> 5)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
    const uint32_t item = items[item_i];
    const int k = (int)(item >> 12), lane0 = (int)((item >> 6) & 63u), lanes = (int)(item & 63u) + 1;
    const int slot = k * 64 + lane;
    const int p = slot < tp.n_pix ? tp.order[slot] : 0;
    bool live = slot < tp.n_pix && lane >= lane0 && lane < lane0 + lanes;
    if (live) live = hp.valid[p] != 0;
    if (__ballot(live) == 0) continue;

    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const unsigned long long w_start = STAMPS ? wall_clock64() : 0ull;
    d3 hpos = mk(0, 0, 0), hn = mk(0, 0, 0), hw = mk(0, 0, 0), hcol = mk(0, 0, 0), acc = mk(0, 0, 0);
    double r2 = 0.0;
    int n = 0;
    if (live) {
        hpos = mk(hp.pos[p], hp.pos[P + p], hp.pos[2 * P + p]);
        hn = mk(hp.nrm[p], hp.nrm[P + p], hp.nrm[2 * P + p]);
        if constexpr (UNIT) { hw = mk(1, 1, 1); hcol = mk(1, 1, 1); }
        else {
            hw = mk(hp.w[p], hp.w[P + p], hp.w[2 * P + p]);
            hcol = ld3(sc.sp[hp.obj[p]].color);
        }
        r2 = hp.r2[p];
        n = hp.n[p]; acc = mk(hp.flux[p], hp.flux[P + p], hp.flux[2 * P + p]);
    }
    unsigned int c_cand = 0, c_sel = 0, c_acc = 0, c_round = 0;  // wave-uniform tallies except c_acc (per lane)
    unsigned long long c_pairs = 0;
    unsigned long long st_scan = 0, st_rank = 0, st_apply = 0, st_walk = 0, c_walk = 0;   // diagnostic build only (STAMPS)
#define TSTAMP() (STAMPS ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)

    // A tile is 64 consecutive hitpoints of the Hilbert order, which is compact except where the curve
    // jumps (or where a surface ends).  Lanes far from the seed lane are replayed as a cluster of
    // their own, so that a scattered tile costs the sum of its parts and not their bounding box.
    unsigned long long remaining = __ballot(live);
    unsigned int n_clusters = 0;
    while (remaining) {
    n_clusters++;
    const int seed = __ffsll((long long)remaining) - 1;
    const double sx = rl_d(hpos.x, seed), sy = rl_d(hpos.y, seed), sz = rl_d(hpos.z, seed);
    const bool inc = live && ((remaining >> lane) & 1ull) && fabs(hpos.x - sx) <= tp.cluster_d && fabs(hpos.y - sy) <= tp.cluster_d &&
                     fabs(hpos.z - sz) <= tp.cluster_d;
    remaining &= ~__ballot(inc);
    const unsigned int n_inc = (unsigned int)__popcll(__ballot(inc));
    const double bx0 = wave_min_perm_d(inc ? hpos.x : BIG), bx1 = wave_max_perm_d(inc ? hpos.x : -BIG);
    const double by0 = wave_min_perm_d(inc ? hpos.y : BIG), by1 = wave_max_perm_d(inc ? hpos.y : -BIG);
    const double bz0 = wave_min_perm_d(inc ? hpos.z : BIG), bz1 = wave_max_perm_d(inc ? hpos.z : -BIG);
    // Phase 1 in single precision (below): everything about the box's centre.  D bounds |h'| + |q'| for every hitpoint of the
    // cluster and every candidate its region can hold (the region is the box grown by the largest radius, which only shrinks).
    const double cbx = 0.5 * (bx0 + bx1), cby = 0.5 * (by0 + by1), cbz = 0.5 * (bz0 + bz1);
    const v2f nhx = {(float)(cbx - hpos.x), (float)(cbx - hpos.x)}, nhy = {(float)(cby - hpos.y), (float)(cby - hpos.y)}, nhz = {(float)(cbz - hpos.z), (float)(cbz - hpos.z)};  // -h'
    const v2f fnx = {(float)hn.x, (float)hn.x}, fny = {(float)hn.y, (float)hn.y}, fnz = {(float)hn.z, (float)hn.z};
    float g_lo = 0.0f, g_hi = 0.0f, e_d = 0.0f, e_a = 0.0f, t_lo = 0.0f, t_hi = 0.0f;
    {
        const double r0 = wave_max_perm_d(inc ? sqrt(fmin(r2, tp.gather_r2)) * (1.0 + 1e-9) + 1e-9 : 0.0);
        const double hx_ = 0.5 * (bx1 - bx0), hy_ = 0.5 * (by1 - by0), hz_ = 0.5 * (bz1 - bz0);
        const double Hm = sqrt(hx_ * hx_ + hy_ * hy_ + hz_ * hz_) * (1.0 + 1e-9) + 1e-9;
        const double Qm = sqrt((hx_ + r0) * (hx_ + r0) + (hy_ + r0) * (hy_ + r0) + (hz_ + r0) * (hz_ + r0)) * (1.0 + 1e-9) + 1e-9;
        const double D = Hm + Qm, u_ = 5.9604644775390625e-8 * GT_E32_SCALE, t2_ = thr * thr;
        // error bands: sppm_deposit.hpp (the full test in single precision) derives them; the distance test against the
        // search radius is within 1.5 (7 u D sqrt(R^2) + 3 u R^2) + the centring's roundings
        const double Ed = 1.5 * u_ * (7.0 * D * sqrt(tp.gather_r2) + 3.0 * tp.gather_r2) + 1e-13 * D * D + 1e-30;
        const double Bf = 1.5 * u_ * (122.0 + 117.0 * t2_) + 2.0 * u_ * t2_ + 2.0 * band_rel * t2_;
        g_lo = (float)(tp.gather_r2 - Ed); g_hi = (float)(tp.gather_r2 + Ed); e_d = (float)(Ed * (1.0 + 1e-6));
        e_a = (float)((1.5 * u_ * (7.0 + 7.0 * t2_) / 64.0 + 1e-12) * D * D + 1e-30);
        t_lo = (float)(t2_ - Bf); t_hi = (float)(t2_ + Bf);
    }
    for (int s = 0; s < tp.g.S; s++) {
        // region of this slab: bounding box grown by the largest radius any lane can still accept at
        const double rr = inc ? sqrt(fmin(r2, tp.gather_r2)) * (1.0 + 1e-9) + 1e-9 : 0.0;
        const double rmax = wave_max_perm_d(rr);
        const double lx = bx0 - rmax, hx = bx1 + rmax, ly = by0 - rmax, hy = by1 + rmax, lz = bz0 - rmax, hz = bz1 + rmax;
        const int cx0 = __builtin_amdgcn_readfirstlane(grid_coord(lx, tp.g.ox, tp.g.inv_c, tp.g.nx));
        const int cx1 = __builtin_amdgcn_readfirstlane(grid_coord(hx, tp.g.ox, tp.g.inv_c, tp.g.nx));
        const int cy0 = __builtin_amdgcn_readfirstlane(grid_coord(ly, tp.g.oy, tp.g.inv_c, tp.g.ny));
        const int cy1 = __builtin_amdgcn_readfirstlane(grid_coord(hy, tp.g.oy, tp.g.inv_c, tp.g.ny));
        const int cz0 = __builtin_amdgcn_readfirstlane(grid_coord(lz, tp.g.oz, tp.g.inv_c, tp.g.nz));
        const int cz1 = __builtin_amdgcn_readfirstlane(grid_coord(hz, tp.g.oz, tp.g.inv_c, tp.g.nz));
        const int kx = cx1 - cx0 + 1, ky = cy1 - cy0 + 1, kz = cz1 - cz0 + 1;
        const int Krows = ky * kz;
        const uint32_t slab_base = (uint32_t)s * (uint32_t)tp.g.ncells;
        const unsigned long long kend = (unsigned long long)(tp.g.first + (int64_t)(s + 1) * tp.g.B) << KEY_SHIFT;
        unsigned long long klo = (unsigned long long)(tp.g.first + (int64_t)s * tp.g.B) << KEY_SHIFT;
        unsigned long long khi = kend;
        while (klo < kend) {
            // ---- collect the candidates with key in [klo, khi) inside the region into LDS (at most 64)
            int sel_n = 0;
            bool overflow = false;
            const unsigned long long ts0 = TSTAMP();
            // (a row of kx cells along x is one run of consecutive records: the region is walked row by row, 64 rows at a time)
            for (int qbase = 0; qbase < Krows && !overflow; qbase += 64) {
                const int q = qbase + lane;
                uint32_t cs = 0, cn = 0;
                if (q < Krows) {
                    const uint32_t b = slab_base + (uint32_t)(((cz0 + q / ky) * tp.g.ny + cy0 + q % ky) * tp.g.nx + cx0);
                    cs = cell_start[b];
                    cn = cell_start[b + kx] - cs;
                }
                const uint32_t incl = wave_incl_scan_u32(cn);
                const uint32_t pre = incl - cn;
                const int tot = __builtin_amdgcn_readlane((int)incl, 63);
                for (int cbase = 0; cbase < tot && !overflow; cbase += 64) {
                    const int i = cbase + lane;
                    const bool has = i < tot;
                    // the cell lane j that owns candidate i: the largest j with pre_j <= i
                    int j = 0;
                    for (int step = 32; step >= 1; step >>= 1) {
                        const int cand = j + step;
                        const uint32_t pv = __shfl(pre, cand & 63);
                        if (cand < 64 && pv <= (uint32_t)i) j = cand;
                    }
                    const uint32_t r = __shfl(cs, j) + ((uint32_t)i - __shfl(pre, j));
                    bool in = false;
                    unsigned long long key = 0;
                    double2 ra = {0, 0}, rb = {0, 0}, rc = {0, 0}, rd = {0, 0}, re = {0, 0};
                    if (has) {
                        // (key and position are read together: a slab's cells hold that slab's records, so the key test
                        // nearly always passes, and a second, dependent round trip to memory per group costs more than the bytes)
                        // The whole row at once, too: what is selected goes straight into the staging rows (unsorted; the rank
                        // stage permutes them in LDS), so a round has two dependent round trips to memory -- the cell table,
                        // the rows -- instead of three; the loads are waited for, not counted by their bytes.
                        const double2* rp = (const double2*)(rec.f + (size_t)REC_ROW * r);  // one row: sppm_records.hpp
                        ra = rp[0]; rb = rp[1]; rc = rp[2]; rd = rp[3]; re = rp[4];
                        key = (unsigned long long)__double_as_longlong(re.y);
                        in = key >= klo && key < khi && ra.x >= lx && ra.x <= hx && ra.y >= ly && ra.y <= hy && rb.x >= lz && rb.x <= hz;
                    }
                    c_cand += (unsigned int)__popcll(__ballot(has));
                    const unsigned long long m = __ballot(in);
                    const int cnt = __popcll(m);
                    if (sel_n + cnt > 64) { overflow = true; break; }
                    if (in) {
                        const int slot = sel_n + (int)mbcnt64(m);
                        s_key[wv][slot] = key;
                        double2* row = (double2*)s_row[wv][slot];  // q, g0, key, flux
                        row[0] = ra; row[1] = rb; row[2] = rc; row[3] = make_double2(__longlong_as_double((long long)key), rd.x); row[4] = make_double2(rd.y, re.x);
                    }
                    sel_n += cnt;
                }
            }
            const unsigned long long ts1 = TSTAMP();
            st_scan += ts1 - ts0;
            if (overflow) { khi = klo + (khi - klo) / 2; continue; }
            if (sel_n > 0) {
                __builtin_amdgcn_wave_barrier();
                c_sel += (unsigned int)sel_n;
                c_pairs += (unsigned long long)sel_n * n_inc;  // (record, hitpoint) pairs whose distance test runs
                c_round++;
                // ---- one candidate per lane: load it, rank it by key, store it in LDS at its rank
                const bool own = lane < sel_n;
                unsigned long long mykey = ~0ull;
                d3 q = mk(0, 0, 0), gn = mk(0, 0, 0), pf = mk(0, 0, 0);
                if (own) {
                    mykey = s_key[wv][lane];
                    const double2* row = (const double2*)s_row[wv][lane];
                    const double2 ra = row[0], rb = row[1], rc = row[2], rd = row[3], re = row[4];
                    q = mk(ra.x, ra.y, rb.x); gn = mk(rb.y, rc.x, rc.y); pf = mk(rd.y, re.x, re.y);
                }
                int rank = 0;
                for (int jj = 0; jj < sel_n; jj++) rank += (s_key[wv][jj] < mykey) ? 1 : 0;
                __builtin_amdgcn_wave_barrier();  // (every lane has read its row before any lane writes one at its rank)
                if (own) {
                    double* row = s_row[wv][rank];  // q, g0, key, flux
                    row[0] = q.x; row[1] = q.y; row[2] = q.z; row[3] = gn.x; row[4] = gn.y; row[5] = gn.z;
                    row[6] = __longlong_as_double((long long)mykey); row[7] = pf.x; row[8] = pf.y; row[9] = pf.z;
                    float* pp = &s_fp[wv][rank >> 1][rank & 1];
                    pp[0] = (float)(q.x - cbx); pp[2] = (float)(q.y - cby); pp[4] = (float)(q.z - cbz);
                    pp[6] = (float)gn.x; pp[8] = (float)gn.y; pp[10] = (float)gn.z;
                }
                if ((sel_n & 1) && lane == 63) {  // an odd candidate's partner: at infinity (outside every radius), normal zero
                    float* pp = &s_fp[wv][sel_n >> 1][1];
                    pp[0] = INFINITY; pp[2] = 0.0f; pp[4] = 0.0f; pp[6] = 0.0f; pp[8] = 0.0f; pp[10] = 0.0f;
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned long long ts2 = TSTAMP();
                st_rank += ts2 - ts1;
                // ---- phase 1, all lanes against candidate t (an LDS broadcast): every test that does not
                // depend on the radius as it shrinks.  Bit t of `todo` = lane may accept candidate t.
                // The row of candidate t+1 is read while candidate t is tested.
                unsigned long long todo = 0;
                const int sel_u = __builtin_amdgcn_readfirstlane(sel_n);
                // In single precision first, two candidates per packed instruction (the loop is bound by the vector
                // instructions it issues); a candidate with a lane inside one of the error bands is tested again in the
                // reference's operation order from its double-precision row, for every lane.
                const float r2_hi = (float)(r2 * (1.0 + 1e-6)) + e_d;  // (r2 only shrinks: a superset here, the walk tests d2 <= r2 exactly)
                const v2f tl2 = {t_lo, t_lo}, th2 = {t_hi, t_hi};
#define GT_EXACT(tt, okv)                                                                                              \
                {                                                                                                      \
                    const double* row = s_row[wv][tt];                                                                 \
                    const d3 qq = mk(row[0], row[1], row[2]), g0 = mk(row[3], row[4], row[5]);                         \
                    const d3 dir = hpos - qq;                                                                          \
                    const double d2 = len2(dir);                                                                       \
                    bool o_ = inc && (d2 < tp.gather_r2) && (d2 <= r2) && (dot(hn, g0) > 1e-3);                        \
                    const double s1 = dot(g0, dir);  /* (its square: the orientation of the normal does not matter) */ \
                    const double lhs = s1 * s1, rhs = thr * thr * d2;                                                  \
                    const double band = rhs * band_rel;                                                                \
                    bool flat = lhs < rhs - band;                                                                      \
                    if (o_ && !flat && !(lhs > rhs + band)) {                                                          \
                        const unsigned int klow = (unsigned int)__double_as_longlong(row[6]);                          \
                        const d3 on = (klow & 1u) ? g0 * -1.0 : g0;                                                    \
                        const double dt = dot(on, dir / sqrt(d2));  /* NaN at d2 == 0: rejected as in the reference */ \
                        flat = fabs(dt) <= thr;                                                                        \
                    }                                                                                                  \
                    okv = o_ && flat;                                                                                  \
                }
                for (int t = 0; t < sel_u; t += 2) {
                    const float* pp = &s_fp[wv][t >> 1][0];
                    const v4f PA = *(const v4f*)pp, PB = *(const v4f*)(pp + 4), PC = *(const v4f*)(pp + 8);
                    const v2f ex = PA.xy + nhx, ey = PA.zw + nhy, ez = PB.xy + nhz;  // q' - h'
                    const v2f gx = PB.zw, gy = PC.xy, gz = PC.zw;
                    const v2f d2 = __builtin_elementwise_fma(ez, ez, __builtin_elementwise_fma(ey, ey, ex * ex));
                    const v2f s1 = __builtin_elementwise_fma(gz, ez, __builtin_elementwise_fma(gy, ey, gx * ex));
                    const v2f nd = __builtin_elementwise_fma(fnz, gz, __builtin_elementwise_fma(fny, gy, fnx * gx));
                    const v2f l2 = s1 * s1;
                    const v2f sp = __builtin_elementwise_fma(tl2, d2, -l2), sn = __builtin_elementwise_fma(th2, d2, -l2);
                    bool ok0 = d2.x <= r2_hi && d2.x < g_lo && nd.x > 1e-3f + 4e-6f && sp.x > e_a;
                    bool ok1 = d2.y <= r2_hi && d2.y < g_lo && nd.y > 1e-3f + 4e-6f && sp.y > e_a;
                    const bool un0 = inc && !ok0 && !(d2.x > r2_hi || d2.x > g_hi || nd.x < 1e-3f - 4e-6f || sn.x < -e_a);
                    const bool un1 = inc && !ok1 && !(d2.y > r2_hi || d2.y > g_hi || nd.y < 1e-3f - 4e-6f || sn.y < -e_a);
                    ok0 = ok0 && inc; ok1 = ok1 && inc;
                    if (__ballot(un0) != 0) GT_EXACT(t, ok0)
                    if (t + 1 < sel_u && __ballot(un1) != 0) GT_EXACT(t + 1, ok1)
                    if (ok0) todo |= 1ull << t;
                    if (ok1 && t + 1 < sel_u) todo |= 2ull << t;
                }
#undef GT_EXACT
                // ---- phase 2, every lane walks ITS OWN candidates in key order: only accepted pairs cost
                // an update, and the trip count is the longest list of the wave, not the candidate count.
                // A lane with nothing to do adds zero flux and scales by one, which changes nothing.
                const unsigned long long ts3 = TSTAMP();
                while (__ballot(todo != 0ull) != 0ull) {
                    if (STAMPS) c_walk++;
                    const bool has = todo != 0ull;
                    const int t = has ? __ffsll((long long)todo) - 1 : 0;
                    todo &= todo - 1ull;   // 0 stays 0
                    const double* row = s_row[wv][t];
                    const d3 dir = hpos - mk(row[0], row[1], row[2]);
                    const d3 ff = mk(row[7], row[8], row[9]);
                    const double d2 = len2(dir);
                    bool take = has && (d2 <= r2);
                    if (vetoes && take) take = !veto_has(tp.veto, (unsigned long long)__double_as_longlong(row[6]), p);
                    if (__ballot(take) != 0ull) {  // src/sppm2.h:397-419
                        const double na = n * tp.alpha;
                        double g = (na + tp.alpha) / (na + 1.0);
                        if (n == 0) g = 1;
                        const d3 num = had(hw, had(hcol, ff));
                        const d3 add = mk(dm_div_pi(num.x), dm_div_pi(num.y), dm_div_pi(num.z));  // == num / PI, bit for bit
                        if (take) {
                            r2 = r2 * g;
                            n++;
                            acc = (acc + add) * g;
                            c_acc++;
                        }
                    }
                    if (tp.tap_cap > 0 && take) {
                        const unsigned long long slot = atomicAdd(tp.tap_n, 1ull);
                        if ((int64_t)slot < tp.tap_cap) { tp.tap_key[slot] = (unsigned long long)__double_as_longlong(row[6]); tp.tap_pix[slot] = p; }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (STAMPS) { const unsigned long long ts4 = TSTAMP(); st_apply += ts4 - ts2; st_walk += ts4 - ts3; }
            }
            klo = khi;
            khi = kend;
        }
    }
    }  // clusters
    if (live) {
        int pi = p;
        asm volatile("" : "+v"(pi));  // (the five addresses are formed here, not kept in registers through the item's loops)
        hp.r2[pi] = r2; hp.n[pi] = n;
        hp.flux[pi] = acc.x; hp.flux[P + pi] = acc.y; hp.flux[2 * P + pi] = acc.z;
    }
    // (an item's tallies are kept by the wavefront and reach the global counters once, when it runs out of items: a dozen
    // atomics per item on one cache line, every wavefront waiting for each of its own, cost the deposit kernel a fifth
    // of its time -- DESIGN.md section 10, round 3)
    {
        const unsigned long long cyc = (unsigned long long)(__builtin_amdgcn_s_memtime() - t_start);
        w_pairs += c_pairs; w_cand += c_cand; w_acc += c_acc; w_sel += c_sel; w_round += c_round; w_items++;
        if (STAMPS) {  // (work distribution over tiles: the diagnostic build only)
            w_cyc += cyc;
            m_sel = m_sel > c_sel ? m_sel : c_sel; m_cand = m_cand > c_cand ? m_cand : c_cand; m_round = m_round > c_round ? m_round : c_round;
            m_cyc = m_cyc > cyc ? m_cyc : cyc;
            const unsigned long long sk = ((unsigned long long)c_sel << 24) | (unsigned long long)k;
            m_selk = m_selk > sk ? m_selk : sk;
        }
    }
    if (lane == 0) {
        if (STAMPS && tp.item_times) {  // XCC_ID (hardware register 20, bits 3:0): the clock is per XCD
            const unsigned long long xcc = (unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u);
            tp.item_times[4 * (size_t)item_i] = first_item ? t_entry : w_start; tp.item_times[4 * (size_t)item_i + 1] = (wall_clock64() << 4) | xcc;
            tp.item_times[4 * (size_t)item_i + 2] = ((unsigned long long)c_sel << 32) | item;
            tp.item_times[4 * (size_t)item_i + 3] = ((unsigned long long)c_round << 32) | ((unsigned long long)n_clusters << 16) | (unsigned long long)(c_cand >> 6 > 65535u ? 65535u : c_cand >> 6);
        }
        if (STAMPS) { atomicAdd(&counters[16], st_scan); atomicAdd(&counters[17], st_rank); atomicAdd(&counters[18], st_apply); atomicAdd(&counters[19], st_walk); atomicAdd(&counters[20], c_walk); }
    }
    __builtin_amdgcn_wave_barrier();
    }  // next item
    for (int off = 32; off > 0; off >>= 1) w_acc += __shfl_down(w_acc, off);
    if (lane == 0 && w_items) {
        if (w_pairs) atomicAdd(&counters[3], w_pairs);
        if (w_cand) atomicAdd(&counters[14], (unsigned long long)w_cand);
        if (w_acc) atomicAdd(&counters[4], (unsigned long long)w_acc);
        if (w_sel) atomicAdd(&counters[5], (unsigned long long)w_sel);
        if (w_round) atomicAdd(&counters[6], (unsigned long long)w_round);
        atomicAdd(&counters[7], (unsigned long long)w_items);  // live tiles
        if (STAMPS) {  // work distribution over tiles (read by the SPPM_DEBUG report of the diagnostic build)
            atomicMax(&counters[8], (unsigned long long)m_sel); atomicMax(&counters[9], (unsigned long long)m_cand);
            atomicMax(&counters[10], (unsigned long long)m_round);
            atomicMax(&counters[11], m_cyc); atomicAdd(&counters[12], w_cyc);
            atomicMax(&counters[13], m_selk);
        }
    }
#undef TSTAMP
}
