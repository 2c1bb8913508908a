// sppm_exchange.hpp -- record exchange between ranks (packed rows) and the occupancy map that drops far records.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
#pragma once

// ---- record exchange between ranks: SoA buffer <-> packed rows of 10 doubles (key bits, 9 fields) -------
// One wavefront per WAVE_ROWS buffer slots, in groups of 64: a first sweep counts the valid ones, ONE atomic per
// wave reserves a contiguous run of rows (thousands of waves asking one counter per 64 slots would queue up on
// it), a second sweep writes the rows, staged through LDS so that the row-major side is accessed with
// consecutive lanes on consecutive doubles.
#define WAVE_ROWS 1024
__global__ void __launch_bounds__(256) k_pack_records(RecArrays rec, const unsigned long long* __restrict__ rec_count, double* __restrict__ rows, int64_t cap_rows,
                               unsigned long long* __restrict__ n_out)
{
    __shared__ double s_row[4][64 * 10];
    const int64_t n = min((int64_t)*rec_count, rec.cap);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + wv) * WAVE_ROWS;
    if (w0 >= n) return;
    int total = 0;
    for (int g = 0; g < WAVE_ROWS / 64; g++) {
        const int64_t i = w0 + g * 64 + lane;
        total += __popcll(__ballot(i < n && rec.key[i] != REC_INVALID));
    }
    if (total == 0) return;
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(n_out, (unsigned long long)total);
    at = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) |
         (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)at);
    for (int g = 0; g < WAVE_ROWS / 64; g++) {
        const int64_t i = w0 + g * 64 + lane;
        const unsigned long long key = i < n ? rec.key[i] : REC_INVALID;
        const bool has = key != REC_INVALID;
        const unsigned long long m = __ballot(has);
        if (m == 0) continue;
        const int cnt = __popcll(m);
        if (has) {
            double* r = s_row[wv] + 10 * __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            r[0] = __longlong_as_double((long long)key);
#pragma unroll
            for (int c = 0; c < 9; c++) r[1 + c] = rec.f[c * rec.cap + i];
        }
        __builtin_amdgcn_wave_barrier();
        // rows at .. at+cnt-1, as far as they fit (the count is complete either way: the host reports the overflow)
        const int64_t fit = min((int64_t)cnt, max((int64_t)0, cap_rows - (int64_t)at));
        for (int64_t j = lane; j < fit * 10; j += 64) rows[(int64_t)at * 10 + j] = s_row[wv][j];
        __builtin_amdgcn_wave_barrier();
        at += (unsigned long long)cnt;
    }
}
// ---- which records can matter to this rank ---------------------------------------------------------------
// A coarse occupancy map over the bounding box of the rank's hitpoints (grown by the search radius), NEAR_DIM
// cells per axis.  k_mark_near sets every cell that the box [pos - r, pos + r] of a valid hitpoint of the rank
// overlaps, r = sqrt(min(r2, search r2)) at that moment.  Radii only shrink, so whenever the map is used later a
// hitpoint can only accept records inside the box that was marked for it: a record whose cell is not set is
// dropped when rows are appended or a batch is binned.  Coordinates clamp into the map (monotone, so "the cells a
// box overlaps" survives the clamp).
#define NEAR_DIM 64
struct NearMap { const unsigned long long* bb; const uint8_t* cells; double reach; };  // bb: ordered-int doubles, min[3] max[3]
__device__ __forceinline__ double near_undo(unsigned long long o)
{
    return __longlong_as_double((long long)((o >> 63) ? (o & 0x7fffffffffffffffull) : ~o));
}
__device__ __forceinline__ int near_coord(const unsigned long long* bb, double reach, int a, double v)
{
    const double lo = near_undo(bb[a]) - reach, ext = near_undo(bb[3 + a]) + reach - lo;
    const double f = floor((v - lo) / (ext / (double)NEAR_DIM));
    return f < 0.0 ? 0 : (f > (double)(NEAR_DIM - 1) ? NEAR_DIM - 1 : (int)f);
}
__device__ __forceinline__ bool near_rank(const NearMap& nm, double x, double y, double z)
{
    if (!(near_undo(nm.bb[0]) <= near_undo(nm.bb[3]))) return false;  // the rank has no valid hitpoint
    const int ix = near_coord(nm.bb, nm.reach, 0, x), iy = near_coord(nm.bb, nm.reach, 1, y), iz = near_coord(nm.bb, nm.reach, 2, z);
    return nm.cells[(iz * NEAR_DIM + iy) * NEAR_DIM + ix] != 0;
}
__global__ void k_mark_near(HpArrays hp, const int32_t* __restrict__ pix, int n, const unsigned long long* __restrict__ bb, double reach,
                            double gather_r2, uint8_t* __restrict__ cells)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pix[i], P = hp.P;
    if (!hp.valid[p]) return;
    const double r = sqrt(fmin(hp.r2[p], gather_r2)) * (1.0 + 1e-9) + 1e-9;
    const double x = hp.pos[p], y = hp.pos[P + p], z = hp.pos[2 * P + p];
    const int x0 = near_coord(bb, reach, 0, x - r), x1 = near_coord(bb, reach, 0, x + r);
    const int y0 = near_coord(bb, reach, 1, y - r), y1 = near_coord(bb, reach, 1, y + r);
    const int z0 = near_coord(bb, reach, 2, z - r), z1 = near_coord(bb, reach, 2, z + r);
    for (int iz = z0; iz <= z1; iz++)
        for (int iy = y0; iy <= y1; iy++)
            for (int ix = x0; ix <= x1; ix++) cells[(iz * NEAR_DIM + iy) * NEAR_DIM + ix] = 1;
}

// seeds of `n` bounding boxes: min slots all ones, max slots zero (ordered-int doubles)
__global__ void k_bb_seed(unsigned long long* __restrict__ bb, int n)
{
    for (int i = threadIdx.x; i < 6 * n; i += blockDim.x) bb[i] = (i % 6) < 3 ? ~0ull : 0ull;
}

// k_pack_records by destination: block d of `rows` ([world][cap_rows][10]) gets the records whose cell of rank d's map
// is set (cells_all == nullptr: every record goes to every rank).  The records are read once per destination.
__global__ void __launch_bounds__(256) k_pack_records_to(RecArrays rec, const unsigned long long* __restrict__ rec_count, double* __restrict__ rows, int64_t cap_rows,
                                                         unsigned long long* __restrict__ n_out, int world, const unsigned long long* __restrict__ bb_all,
                                                         const uint8_t* __restrict__ cells_all, double reach)
{
    __shared__ double s_row[4][64 * 10];
    const int64_t n = min((int64_t)*rec_count, rec.cap);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + wv) * WAVE_ROWS;
    if (w0 >= n) return;
    for (int d = 0; d < world; d++) {
        NearMap nm;
        nm.bb = bb_all ? bb_all + 6 * d : nullptr; nm.cells = cells_all ? cells_all + (size_t)d * NEAR_DIM * NEAR_DIM * NEAR_DIM : nullptr; nm.reach = reach;
        unsigned long long keep[WAVE_ROWS / 64];
        int total = 0;
#pragma unroll
        for (int g = 0; g < WAVE_ROWS / 64; g++) {
            const int64_t i = w0 + g * 64 + lane;
            bool has = i < n && rec.key[i] != REC_INVALID;
            if (has && nm.cells) has = near_rank(nm, rec.f[i], rec.f[rec.cap + i], rec.f[2 * rec.cap + i]);
            keep[g] = __ballot(has);
            total += __popcll(keep[g]);
        }
        if (total == 0) continue;
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(&n_out[d], (unsigned long long)total);
        at = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) |
             (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)at);
        double* out = rows + (int64_t)d * cap_rows * 10;
#pragma unroll
        for (int g = 0; g < WAVE_ROWS / 64; g++) {
            const unsigned long long m = keep[g];
            if (m == 0) continue;
            const int64_t i = w0 + g * 64 + lane;
            const int cnt = __popcll(m);
            if ((m >> lane) & 1ull) {
                double* r = s_row[wv] + 10 * __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                r[0] = __longlong_as_double((long long)rec.key[i]);
#pragma unroll
                for (int c = 0; c < 9; c++) r[1 + c] = rec.f[c * rec.cap + i];
            }
            __builtin_amdgcn_wave_barrier();
            const int64_t fit = min((int64_t)cnt, max((int64_t)0, cap_rows - (int64_t)at));  // the count is complete either way
            for (int64_t j = lane; j < fit * 10; j += 64) out[(int64_t)at * 10 + j] = s_row[wv][j];
            __builtin_amdgcn_wave_barrier();
            at += (unsigned long long)cnt;
        }
    }
}

// nm.cells != nullptr: rows that cannot matter to this rank (near_rank) are not stored.
__global__ void __launch_bounds__(256) k_append_records(RecArrays rec, unsigned long long* __restrict__ rec_count, const double* __restrict__ rows, int64_t n,
                                                        int32_t* __restrict__ err, NearMap nm)
{
    __shared__ double s_row[4][64 * 10];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + wv) * WAVE_ROWS;
    if (w0 >= n) return;
    // first sweep: which rows of this wave's WAVE_ROWS matter to the rank (one bit mask per group of 64)
    unsigned long long keep[WAVE_ROWS / 64];
    int total = 0;
#pragma unroll
    for (int g = 0; g < WAVE_ROWS / 64; g++) {
        const int64_t i = w0 + g * 64 + lane;
        bool has = i < n;
        if (has && nm.cells) has = near_rank(nm, rows[10 * i + 1], rows[10 * i + 2], rows[10 * i + 3]);
        keep[g] = __ballot(has);
        total += __popcll(keep[g]);
    }
    if (total == 0) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(rec_count, (unsigned long long)total);  // one reservation per wave
    base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)base);
    if ((int64_t)base + total > rec.cap) { if (lane == 0) err[0] = 1; return; }
    // second sweep: the kept rows, staged through LDS (consecutive lanes read consecutive doubles)
#pragma unroll
    for (int g = 0; g < WAVE_ROWS / 64; g++) {
        const unsigned long long m = keep[g];
        if (m == 0) continue;
        const int64_t g0 = w0 + g * 64;
        const int64_t here = min((int64_t)64, n - g0);
        for (int64_t j = lane; j < here * 10; j += 64) s_row[wv][j] = rows[g0 * 10 + j];
        __builtin_amdgcn_wave_barrier();
        if ((m >> lane) & 1ull) {
            const double* r = s_row[wv] + 10 * lane;
            const int64_t slot = (int64_t)base + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            rec.key[slot] = (unsigned long long)__double_as_longlong(r[0]);
#pragma unroll
            for (int c = 0; c < 9; c++) rec.f[c * rec.cap + slot] = r[1 + c];
        }
        __builtin_amdgcn_wave_barrier();
        base += (unsigned long long)__popcll(m);
    }
}
