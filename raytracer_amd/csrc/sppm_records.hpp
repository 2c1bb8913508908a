// sppm_records.hpp -- photon-hit records of a batch: hash grid, prefix scans, counting-sort scatter.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
#pragma once

// ---- hash grid over the records of a batch ----------------------------------------------------------
struct Grid { double inv_cell; uint32_t mask; };

__device__ __forceinline__ int cell_coord(double x, double inv_cell) { return (int)floor(x * inv_cell); }
__device__ __forceinline__ uint32_t cell_hash(int ix, int iy, int iz, uint32_t mask)
{
    return (((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)iz * 83492791u)) & mask;
}

__global__ void k_grid_count(RecArrays rec, const unsigned long long* __restrict__ rec_count, Grid g, uint32_t* __restrict__ cell_cnt,
                             uint32_t* __restrict__ rec_bucket)
{
    const int64_t n = min((int64_t)*rec_count, rec.cap);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (rec.key[i] == REC_INVALID) { rec_bucket[i] = 0xFFFFFFFFu; continue; }  // a hole left by the tracer's chunked slot reservation
        const uint32_t b = cell_hash(cell_coord(rec.f[i], g.inv_cell), cell_coord(rec.f[rec.cap + i], g.inv_cell),
                                     cell_coord(rec.f[2 * rec.cap + i], g.inv_cell), g.mask);
        rec_bucket[i] = b;
        atomicAdd(&cell_cnt[b], 1u);
    }
}

// exclusive scan of cell_cnt[0..n) -> cell_start[0..n], three small kernels (n is a power of two >= 1024)
#define SCAN_BLOCK 1024
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ block_sum)
{
    __shared__ uint32_t wave_tot[SCAN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t i = (size_t)blockIdx.x * SCAN_BLOCK + tid;
    const uint32_t v = in[i];
    uint32_t x = v;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    if (wv == 0) {
        uint32_t t = lane < SCAN_BLOCK / 64 ? wave_tot[lane] : 0;
        for (int off = 1; off < SCAN_BLOCK / 64; off <<= 1) { const uint32_t y = __shfl_up(t, off); if (lane >= off) t += y; }
        if (lane < SCAN_BLOCK / 64) wave_tot[lane] = t;
    }
    __syncthreads();
    const uint32_t base = wv ? wave_tot[wv - 1] : 0;
    out[i] = base + x - v;
    if (tid == SCAN_BLOCK - 1) block_sum[blockIdx.x] = base + x;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tops(uint32_t* __restrict__ block_sum, int nblocks, uint32_t* __restrict__ total_out,
                                                          unsigned long long* __restrict__ records_acc)
{
    // single block; nblocks <= 16384 handled by a serial carry over chunks of SCAN_BLOCK
    __shared__ uint32_t wave_tot[SCAN_BLOCK / 64];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base_i = 0; base_i < nblocks; base_i += SCAN_BLOCK) {
        const int i = base_i + tid;
        const uint32_t v = i < nblocks ? block_sum[i] : 0;
        uint32_t x = v;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (lane >= off) x += y; }
        if (lane == 63) wave_tot[wv] = x;
        __syncthreads();
        if (wv == 0) {
            uint32_t t = lane < SCAN_BLOCK / 64 ? wave_tot[lane] : 0;
            for (int off = 1; off < SCAN_BLOCK / 64; off <<= 1) { const uint32_t y = __shfl_up(t, off); if (lane >= off) t += y; }
            if (lane < SCAN_BLOCK / 64) wave_tot[lane] = t;
        }
        __syncthreads();
        const uint32_t b = carry + (wv ? wave_tot[wv - 1] : 0);
        if (i < nblocks) block_sum[i] = b + x - v;
        __syncthreads();
        if (tid == SCAN_BLOCK - 1) carry = b + x;
        __syncthreads();
    }
    if (tid == 0) { *total_out = carry; if (records_acc) atomicAdd(records_acc, (unsigned long long)carry); }
}
// ...and clears the counts the scan has consumed, so that the next batch finds them zero without a memset of the
// whole table (tens of MB per batch, on a GPU whose SIMDs the gather holds)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_sum, uint32_t* __restrict__ cursor,
                                                         size_t n, const uint32_t* __restrict__ total, uint32_t* __restrict__ counts)
{
    const size_t i = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const uint32_t v = out[i] + block_sum[blockIdx.x];
    out[i] = v;
    cursor[i] = v;
    counts[i] = 0u;
    if (i == n - 1) out[n] = *total;
}
// fixed point of the deposits: x * scale (scale a power of two chosen from the emitted flux by sppm_create: the
// largest emitted component lands in [2^27, 2^28)), rounded to nearest-even, clamped at 2^41 (reported: err[6])
#define FX_CLAMP 2199023255552.0
#define FX_MAX_DEPOSITS (1ll << 21)  // per hitpoint and pass: with the clamp a sum stays below 2^62
__device__ __forceinline__ long long fx_fix(double v, double scale, bool& clamped)
{
    double x = v * scale;
    if (!(x < FX_CLAMP)) { x = FX_CLAMP; clamped = true; }
    if (x < 0.0) x = 0.0;
    return __double2ll_rn(x);
}

// rows != 0 (the tile gathers read dst): dst is laid out record by record, REC_ROW doubles each: position, geometric
// normal, flux, key.  The gathers read every record of a cell region many times, a few consecutive records at a time: as
// rows those are one or two cache lines, as nine planes nine -- and this kernel writes 80 contiguous bytes per record.
// fx_scale > 0 (the per-pass rule's deposit kernel): the flux is written as the fixed point of the deposits (fx_fix: an
// integer of at most 42 bits, stored as the double of that value) -- once per record here instead of once per cluster
// that stages it.  Sums of fewer than 2^11 such values are exact in double arithmetic, in any order: the deposit kernel
// adds them with v_add_f64 (and the matrix pipe's result feeds only comparisons) and moves them to 64-bit integers
// before a sum can reach 2^53.
#define REC_ROW 10
__global__ void k_grid_scatter(RecArrays src, RecArrays dst, const unsigned long long* __restrict__ rec_count,
                               const uint32_t* __restrict__ rec_bucket, uint32_t* __restrict__ cursor, int rows, double fx_scale, int32_t* __restrict__ err)
{
    const int64_t n = min((int64_t)*rec_count, src.cap);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (rec_bucket[i] == 0xFFFFFFFFu) continue;
        const uint32_t p = atomicAdd(&cursor[rec_bucket[i]], 1u);
        if (rows) {
            bool clamped = false;
            double v[REC_ROW];
#pragma unroll
            for (int c = 0; c < 9; c++) v[c] = src.f[c * src.cap + i];
            if (fx_scale > 0.0) {
#pragma unroll
                for (int c = 6; c < 9; c++) v[c] = (double)fx_fix(v[c], fx_scale, clamped);  // an integer below 2^41 + 1: exact as a double
            }
            v[9] = __longlong_as_double((long long)src.key[i]);
            double2* row = (double2*)(dst.f + (size_t)REC_ROW * p);
#pragma unroll
            for (int c = 0; c < REC_ROW / 2; c++) row[c] = make_double2(v[2 * c], v[2 * c + 1]);
            if (clamped) err[6] = 1;
        } else {
            dst.key[p] = src.key[i];
#pragma unroll
            for (int c = 0; c < 9; c++) dst.f[c * dst.cap + p] = src.f[c * src.cap + i];
        }
    }
}
