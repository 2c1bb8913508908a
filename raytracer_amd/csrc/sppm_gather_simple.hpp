// sppm_gather_simple.hpp -- ownership mask, the simple per-hitpoint gather (cross-check of the tile gather), the K cap check.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
#pragma once

// (fx_fix, the fixed point of the deposits: sppm_records.hpp)

// Pairs (photon-hit record, hitpoint) that the reference's K-capped search dropped (sppm_cap.hpp): an open-addressing
// table of (record key, hitpoint) entries, filled before the gather starts and only read by it.  n == nullptr or
// *n == 0: no pair is dropped (the cap cannot act, or did not), and the gathers never look.
struct Veto { unsigned long long* tab; unsigned long long mask; const unsigned long long* n; };
__device__ __forceinline__ unsigned long long veto_hash(unsigned long long key, unsigned int pix)
{
    unsigned long long h = key * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)pix + 1ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}
__device__ __forceinline__ bool veto_active(const Veto& v) { return v.n != nullptr && *v.n != 0ull; }
__device__ __forceinline__ bool veto_has(const Veto& v, unsigned long long key, int pix)
{
    for (unsigned long long h = veto_hash(key, (unsigned int)pix) & v.mask;; h = (h + 1ull) & v.mask) {
        const unsigned long long k = v.tab[2 * h];
        if (k == 0ull) return false;  // record keys are > 0 (depth >= 1)
        if (k == key && v.tab[2 * h + 1] == (unsigned long long)pix) return true;
    }
}

// max over valid hitpoints of min(r2, gather_r2), as an ordered-int atomic max on the bit pattern
struct Own { const uint8_t* mask; };  // mask[p] != 0: this rank owns pixel p
__device__ __forceinline__ bool owned(const Own& o, int p) { return o.mask[p] != 0; }
__global__ void k_mask_from_list(const int32_t* __restrict__ pix, int n, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[pix[i]] = 1;
}
__global__ void k_r2max(HpArrays hp, Own own, double gather_r2, unsigned long long* __restrict__ out)
{
    double m = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hp.P; p += gridDim.x * blockDim.x) {
        if (hp.valid[p] && owned(own, p)) { const double r = fmin(hp.r2[p], gather_r2); m = fmax(m, r); }
    }
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// ---- gather: each hitpoint replays the records near it in key order ------------------------------------
struct GatherParams {
    Grid g; double gather_r2, alpha; int rule; Own own; double fx_scale; int32_t* err;
    int unit_weight;  // classic PPM: the record's flux is applied as it is (no hitpoint weight or colour)
    Veto veto;
    uint64_t* tap_key; int32_t* tap_pix; unsigned long long* tap_n; int64_t tap_cap;
};

#define SEL 8  // keys selected per scan of the neighbourhood

// Does record r pass the search filter of src/kdtree.h:123-127 and the acceptance test of
// src/sppm2.h:393-395 for a hitpoint at hpos with normal hn and squared radius r2?
__device__ __forceinline__ bool rec_accept(const RecArrays& rec, int64_t r, uint64_t key, d3 hpos, d3 hn, double r2, double gather_r2)
{
    const int64_t c = rec.cap;
    const d3 q = mk(rec.f[r], rec.f[c + r], rec.f[2 * c + r]);
    const d3 dir = hpos - q;  // node position - search position, kdtree.h:123
    const double d2 = len2(dir);
    if (!(d2 < gather_r2) || !(d2 <= r2)) return false;
    const d3 gn = mk(rec.f[3 * c + r], rec.f[4 * c + r], rec.f[5 * c + r]);
    if (!(dot(hn, gn) > 1e-3)) return false;
    const d3 on = (key & 1ull) ? gn * -1.0 : gn;
    const double dt = dot(on, dir / sqrt(d2));           // NaN when d2 == 0: rejected, as in the reference
    return fabs(dt) <= gather_r2 * 0.01;
}

__global__ void __launch_bounds__(256) k_gather(HpArrays hp, DevScene sc, RecArrays rec, const uint32_t* __restrict__ cell_start, GatherParams gp,
                                                unsigned long long* __restrict__ counters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_cand = 0, n_acc = 0;
    if (t < hp.P) {
        const int p = t;
        const int P = hp.P;
        const bool vetoes = veto_active(gp.veto);
        if (hp.valid[p] && owned(gp.own, p)) {
            const d3 hpos = mk(hp.pos[p], hp.pos[P + p], hp.pos[2 * P + p]);
            const d3 hn = mk(hp.nrm[p], hp.nrm[P + p], hp.nrm[2 * P + p]);
            const d3 hw = gp.unit_weight ? mk(1, 1, 1) : mk(hp.w[p], hp.w[P + p], hp.w[2 * P + p]);
            const d3 hcol = gp.unit_weight ? mk(1, 1, 1) : ld3(sc.sp[hp.obj[p]].color);
            double r2 = hp.r2[p];
            int n = 0;
            d3 acc = mk(0, 0, 0);
            long long fx = 0, fy = 0, fz = 0, madd = 0;  // per-pass rule: fixed-point deposits (sppm_deposit.hpp)
            bool clamped = false;
            if (gp.rule == 0) { n = hp.n[p]; acc = mk(hp.flux[p], hp.flux[P + p], hp.flux[2 * P + p]); }
            // cells overlapping the sphere of the larger of the two bounds the tests use
            const double rr = sqrt(fmin(r2, gp.gather_r2)) * (1.0 + 1e-9) + 1e-12;
            const int x0 = cell_coord(hpos.x - rr, gp.g.inv_cell), x1 = cell_coord(hpos.x + rr, gp.g.inv_cell);
            const int y0 = cell_coord(hpos.y - rr, gp.g.inv_cell), y1 = cell_coord(hpos.y + rr, gp.g.inv_cell);
            const int z0 = cell_coord(hpos.z - rr, gp.g.inv_cell), z1 = cell_coord(hpos.z + rr, gp.g.inv_cell);
            uint64_t last = 0;  // keys are > 0: depth >= 1
            for (;;) {
                // select the SEL smallest keys > last among accepted candidates (sorted ascending)
                uint64_t sk[SEL]; int64_t si[SEL];
#pragma unroll
                for (int j = 0; j < SEL; j++) { sk[j] = ~0ull; si[j] = -1; }
                for (int iz = z0; iz <= z1; iz++)
                    for (int iy = y0; iy <= y1; iy++)
                        for (int ix = x0; ix <= x1; ix++) {
                            const uint32_t b = cell_hash(ix, iy, iz, gp.g.mask);
                            const uint32_t e = cell_start[b + 1];
                            for (uint32_t r = cell_start[b]; r < e; r++) {
                                const uint64_t key = rec.key[r];
                                if (last == 0) n_cand++;
                                if (key <= last || key >= sk[SEL - 1]) continue;
                                if (!rec_accept(rec, r, key, hpos, hn, r2, gp.gather_r2)) continue;
                                // two cells of the range may share a bucket: skip a key already held
                                bool dup = false;
#pragma unroll
                                for (int j = 0; j < SEL; j++) dup |= (sk[j] == key);
                                if (dup) continue;
                                uint64_t ck = key; int64_t ci = r;
#pragma unroll
                                for (int j = 0; j < SEL; j++) {
                                    if (ck < sk[j]) { const uint64_t tk = sk[j]; const int64_t ti = si[j]; sk[j] = ck; si[j] = ci; ck = tk; ci = ti; }
                                }
                            }
                        }
                if (si[0] < 0) break;
#pragma unroll
                for (int j = 0; j < SEL; j++) {
                    if (si[j] < 0) continue;
                    const int64_t r = si[j];
                    last = sk[j];
                    // the radius may have shrunk since the scan: test again with the current one
                    if (!rec_accept(rec, r, sk[j], hpos, hn, r2, gp.gather_r2)) continue;
                    if (vetoes && veto_has(gp.veto, sk[j], p)) continue;  // dropped by the K cap of the reference's search
                    const int64_t c = rec.cap;
                    const d3 pf = mk(rec.f[6 * c + r], rec.f[7 * c + r], rec.f[8 * c + r]);
                    if (gp.rule == 0) {  // src/sppm2.h:397-419
                        double g;
                        if (n != 0) g = (n * gp.alpha + gp.alpha) / (n * gp.alpha + 1.0);
                        else g = 1;
                        r2 = r2 * g;
                        n++;
                        acc = (acc + had(hw, had(hcol, pf)) / SPPM_PI) * g;
                    } else {             // src/stochastic_ppm.h:449-454
                        fx += fx_fix(pf.x, gp.fx_scale, clamped); fy += fx_fix(pf.y, gp.fx_scale, clamped); fz += fx_fix(pf.z, gp.fx_scale, clamped);
                        madd++;
                    }
                    n_acc++;
                    if (gp.tap_cap > 0) {
                        const unsigned long long slot = atomicAdd(gp.tap_n, 1ull);
                        if ((int64_t)slot < gp.tap_cap) { gp.tap_key[slot] = sk[j]; gp.tap_pix[slot] = p; }
                    }
                }
                if (si[SEL - 1] < 0) break;  // the scan found fewer than SEL: nothing is left
            }
            if (gp.rule == 0) {
                hp.r2[p] = r2; hp.n[p] = n;
                hp.flux[p] = acc.x; hp.flux[P + p] = acc.y; hp.flux[2 * P + p] = acc.z;
            } else if (madd > 0) {
                hp.dep[p] += fx; hp.dep[(size_t)P + p] += fy; hp.dep[2 * (size_t)P + p] += fz; hp.dep[3 * (size_t)P + p] += madd;
                if (clamped) gp.err[6] = 1;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) { n_cand += __shfl_down(n_cand, off); n_acc += __shfl_down(n_acc, off); }
    if ((threadIdx.x & 63) == 0) {
        if (n_cand) atomicAdd(&counters[3], (unsigned long long)n_cand);
        if (n_acc) atomicAdd(&counters[4], (unsigned long long)n_acc);
    }
}

// ---- K cap check (cfg.check_gather_max) ------------------------------------------------------------------
// The reference's search keeps at most K hitpoints per photon hit (src/kdtree.h:128-133).  While no query
// holds more than K candidates the cap never acts and the uncapped set this build computes IS the
// reference's set.  This kernel counts, for every record of the batch, the hitpoints that pass the search
// filter of kdtree.h:123-127 (d2 < R^2 and |dot(on, dir/|dir|)| <= 0.01 R^2) -- ALL hitpoints, not only those
// whose own radius still reaches the record -- by brute force: a block of 256 records against the hitpoint
// positions staged through LDS.  A validation mode: exact, O(records x hitpoints).
__global__ void __launch_bounds__(256) k_count_queries(HpArrays hp, RecArrays rec, const unsigned long long* __restrict__ rec_count, double gather_r2,
                                                       long long K, int32_t* __restrict__ err, unsigned long long* __restrict__ qmax)
{
    __shared__ double s_p[3][256];
    const long long slots = min((long long)*rec_count, (long long)rec.cap);
    if ((long long)blockIdx.x * 256 >= slots) return;
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long cap = rec.cap, P = hp.P;
    bool have = false;
    unsigned long long key = 0;
    d3 q = mk(0, 0, 0), on = mk(0, 0, 0);
    if (r < slots) {
        key = rec.key[r];
        have = key != REC_INVALID;
        if (have) {
            q = mk(rec.f[r], rec.f[cap + r], rec.f[2 * cap + r]);
            const d3 gn = mk(rec.f[3 * cap + r], rec.f[4 * cap + r], rec.f[5 * cap + r]);
            on = (key & 1ull) ? gn * -1.0 : gn;
        }
    }
    const double thr = gather_r2 * 0.01;
    long long cnt = 0;
    for (long long base = 0; base < P; base += 256) {
        const long long h = base + threadIdx.x;
        const bool live = h < P && hp.valid[h] != 0;
        // a hitpoint that does not exist sits infinitely far away
        s_p[0][threadIdx.x] = live ? hp.pos[h] : 1e300; s_p[1][threadIdx.x] = live ? hp.pos[P + h] : 1e300; s_p[2][threadIdx.x] = live ? hp.pos[2 * P + h] : 1e300;
        __syncthreads();
        if (have) {
            for (int j = 0; j < 256; j++) {
                const d3 dir = mk(s_p[0][j], s_p[1][j], s_p[2][j]) - q;
                const double d2 = len2(dir);
                if (d2 < gather_r2) {
                    const double dt = dot(on, dir / sqrt(d2));  // NaN at d2 == 0: rejected, as in the reference
                    if (fabs(dt) <= thr) cnt++;
                }
            }
        }
        __syncthreads();
    }
    if (have) {
        if (cnt > K) err[2] = 1;
        atomicMax(qmax, (unsigned long long)cnt);
    }
}
