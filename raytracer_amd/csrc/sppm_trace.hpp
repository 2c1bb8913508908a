// sppm_trace.hpp -- the photon tracer: create_photon as persistent wavefronts that emit photon-hit records; reference-stream probing and chaining.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
#pragma once

struct TraceParams {
    int pass; int64_t first, count, photons_per_pass; uint32_t seed_photon; double flux_scale;
    int64_t rec_cap;
    // reference-stream mode only
    const double* u; int64_t u_len; const uint32_t* start; uint32_t* consumed;
    // classic PPM (src/progressive_photonmapping.h:326,339,344,385-387): depth starts at 0, so the first diffuse
    // hit deposits too, and the flux that reaches a hitpoint is multiply(obj.color, current_flux) of the PHOTON's
    // surface: the record stores that product
    int ppm;
    // reference-stream mode only: 1 = create_photon of src/stochastic_ppm.h:371-572 AS SHIPPED instead of the live
    // src/sppm2.h:317-525 -- no two discarded draws (:377), and a diffuse bounce samples about the LIGHT's emission
    // basis (:482-483 use u, v, w of :382-393).  Exists to reproduce the compiled file's runs (tests/golden/ref_pp_*)
    int miss;
};
struct RecArrays { uint64_t* key; double* f; int64_t cap; };  // f[9][cap]
// A record's key: photon << KEY_SHIFT | depth << 1 | orient -- key order is the reference's application order.  The depth
// field holds 524 287 bounces: a path caught in total reflection inside the glass sphere loses 1 % of its flux per bounce
// (colour 0.99) and the reference ends it when the flux underflows to zero (src/sppm2.h:355), after ~75 000 bounces, so
// the field cannot overflow on the reference's scene (16 bits did: once in ~5 G photons a path passed 32 767).
#define KEY_SHIFT 20
#define KEY_MAX_DEPTH ((1 << (KEY_SHIFT - 1)) - 1)

// create_photon, src/sppm2.h:317-525, with the kd search + update replaced by appending a photon-hit
// record (diffuse hit, depth > 0).  Photon paths have very different lengths (the longest of 64 is
// ~3x the mean), so lanes are not bound to photons: a lane whose photon has ended takes the next
// photon index of its wavefront's current block of 64 (blocks come from one global counter), emits it,
// and all lanes then advance one bounce together.  Which lane traces which photon does not matter:
// the stream is keyed by the photon index and records are ordered by key later.
//
// MODE 0: one XorShift stream per photon.
// Reference-stream mode (the photons of a pass share ONE sequential stream, src/sppm2.h:530):
// MODE 2: "photon" i is a path started at stream offset i; no records, consumed[i] = draws it used.
//         Following consumed[] from offset 0 (k_photon_chain) gives every real photon's start offset.
// MODE 1: photon i reads the stream from start[i]; records as in mode 0.
template <class RNG> struct PhotonState { RNG rng; d3 org, dir, flux; int depth; int64_t index; d3 lw; /* emission normal, read only when tp.miss */ };
struct Deposit { bool want; unsigned long long key; d3 pos, nrm, flux; };
#define STREAM_OVER 0xFFFFFFFFu

template <int MODE, class RNG>
__device__ __forceinline__ void photon_emit(const DevScene& sc, const TraceParams& tp, int64_t i, PhotonState<RNG>& ps)
{
    ps.index = i;
    if constexpr (MODE == 0) ps.rng = make_stream(tp.seed_photon, (uint64_t)tp.pass * (uint64_t)tp.photons_per_pass + (uint64_t)i);
    else { ps.rng.u = tp.u; ps.rng.end = tp.u_len; ps.rng.pos = MODE == 1 ? (int64_t)tp.start[i] : i; }
    bool miss = false;
    if constexpr (MODE != 0) miss = tp.miss != 0;
    if (!miss) { (void)ps.rng.next01(); (void)ps.rng.next01(); }  // two draws computed and discarded, src/sppm2.h:322-324
    const double r1 = 2 * SPPM_PI * ps.rng.next01();
    const double r2 = 1.0 - 2.0 * ps.rng.next01();
    const double r3 = 1.0 - r2 * r2;
    double sn, cs;
    dm_sincos(r1, &sn, &cs);
    const d3 lc = ld3(sc.sp[0].position);
    const d3 source_pos = lc + mk(sqrt(r3) * cs, sqrt(r3) * sn, r2) * (sc.sp[0].radius + SPPM_EPS);
    const d3 w = normalize(source_pos - lc);
    d3 u, v;
    basis(w, u, v);
    ps.org = source_pos;
    if constexpr (MODE != 0) ps.lw = w;
    ps.dir = cosine_sampling(w, u, v, ps.rng);
    ps.flux = ld3(sc.sp[0].emission) * 4.0 * tp.flux_scale;
    ps.depth = tp.ppm ? 0 : -1;
}

// one iteration of the while loop of create_photon; returns false when the path ends
template <class RNG>
__device__ __forceinline__ bool photon_shade(const DevScene& sc, PhotonState<RNG>& ps, const Hit& is, Deposit& dep, int32_t* __restrict__ err, const bool ppm, const bool miss);

template <class RNG, int GRID>
__device__ __forceinline__ bool photon_bounce(const DevScene& sc, PhotonState<RNG>& ps, Deposit& dep, int32_t* __restrict__ err, uint32_t& rays, const bool ppm, const bool miss)
{
    if (ps.rng.stop()) return false;
    if (max3(ps.flux) <= 0.0) return false;
    Hit is;
    rays++;
    if (!intersect_scene<GRID>(sc, ps.org, ps.dir, is)) return false;
    return photon_shade(sc, ps, is, dep, err, ppm, miss);
}

// what create_photon does with a hit (src/sppm2.h:339-525): the record of a diffuse hit, roulette, the next direction
template <class RNG>
__device__ __forceinline__ bool photon_shade(const DevScene& sc, PhotonState<RNG>& ps, const Hit& is, Deposit& dep, int32_t* __restrict__ err, const bool ppm, const bool miss)
{
    const sppm_sphere* obj = &sc.sp[is.object_id];
    const d3 color = ld3(obj->color);
    const int type = obj->reflection_type;
    const bool flip = !(dot(is.normal, ps.dir) < 0.0);
    const d3 on = flip ? is.normal * -1.0 : is.normal;
    ps.depth++;
    if (type == SPPM_DIFFUSE) {
        if (ps.depth > 0) {
            if (ps.depth > KEY_MAX_DEPTH) { err[1] = 1; return false; }
            // the record is stored by the caller, where the whole wavefront has reconverged
            dep.want = true;
            dep.key = ((uint64_t)ps.index << KEY_SHIFT) | ((uint64_t)ps.depth << 1) | (flip ? 1ull : 0ull);
            dep.pos = is.position; dep.nrm = is.normal; dep.flux = ppm ? had(color, ps.flux) : ps.flux;
        }
        const double prob = (color.x + color.y + color.z) / 3;
        if (!(prob > ps.rng.next01())) return false;
        d3 du, dv;
        bool about_light = false;
        if constexpr (std::is_same<RNG, StreamRng>::value) about_light = miss;
        if (about_light) {  // src/stochastic_ppm.h:468-483: diffuse_w/u/v are built and the light's w, u, v are used
            basis(ps.lw, du, dv);
            ps.dir = cosine_sampling(ps.lw, du, dv, ps.rng);
        } else {
            basis(on, du, dv);
            ps.dir = cosine_sampling(on, du, dv, ps.rng);
        }
        ps.org = is.position;
        ps.flux = had(ps.flux, color) / prob;
    } else if (type == SPPM_SPECULAR) {
        ps.dir = reflect_dir(ps.dir, is.normal);
        ps.org = is.position;
        ps.flux = had(ps.flux, color);
    } else {
        const d3 rdir = reflect_dir(ps.dir, is.normal);
        bool into; double nnt, Re; d3 tdir;
        ps.org = is.position;
        if (!glass(ps.dir, is.normal, on, false, into, nnt, tdir, Re)) ps.dir = rdir;
        else ps.dir = (ps.rng.next01() < Re) ? rdir : tdir;
        ps.flux = had(ps.flux, color);
    }
    return true;
}

template <int MODE> struct TraceRng { typedef StreamRng type; };
template <> struct TraceRng<0> { typedef XorShift type; };

#define TRACE_GRAB 64
#ifndef TRACE_COOP_MAX
#define TRACE_COOP_MAX 6  // lanes still active below which the wavefront shares each lane's closest-hit search
#endif
#define REC_CHUNK 128  // record slots a wavefront reserves at a time (one atomic per chunk instead of one per record)
#define REC_INVALID (~0ull)
#ifndef TRACE_WAVES
#define TRACE_WAVES 0  // 0: the compiler's choice (147 registers, three wavefronts per SIMD)
#endif
#if TRACE_WAVES > 0
#define TRACE_ATTR __attribute__((amdgpu_waves_per_eu(TRACE_WAVES, TRACE_WAVES)))
#else
#define TRACE_ATTR
#endif
template <int MODE, int GRID>
__global__ void __launch_bounds__(256) TRACE_ATTR k_photon_trace(DevScene sc, TraceParams tp, RecArrays rec, unsigned long long* __restrict__ rec_count,
                                                      unsigned long long* __restrict__ next_block, unsigned long long* __restrict__ counters,
                                                      int32_t* __restrict__ err)
{
    typedef typename TraceRng<MODE>::type RNG;
    STAGE_SCENE_T(sc, GRID);
    const int lane = threadIdx.x & 63;
    uint32_t rays = 0;
    PhotonState<RNG> ps;
    ps.depth = 0; ps.index = 0; ps.org = mk(0, 0, 0); ps.dir = mk(0, 0, 1); ps.flux = mk(0, 0, 0); ps.lw = mk(0, 0, 1);
    if constexpr (MODE == 0) ps.rng.seed(1u); else { ps.rng.u = tp.u; ps.rng.pos = 0; ps.rng.end = tp.u_len; }
    bool active = false, exhausted = false;
    int64_t blk_lo = 0, blk_hi = 0;  // wave-uniform: the unassigned rest of this wave's current block
    int64_t chunk_lo = 0, chunk_hi = 0;  // wave-uniform: the unused rest of this wave's reserved record slots
    bool rec_full = false;
    for (;;) {
        const unsigned long long need = __ballot(!active);
        if (need != 0 && !exhausted) {
            if (blk_lo >= blk_hi) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(next_block, (unsigned long long)TRACE_GRAB);
                base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                       (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)base);
                blk_lo = (int64_t)base;
                blk_hi = min((int64_t)base + TRACE_GRAB, tp.count);
                if (blk_lo >= tp.count) { exhausted = true; blk_lo = blk_hi = 0; }
            }
            const int64_t idx = blk_lo + (int64_t)mbcnt64(need);
            if (!active && idx < blk_hi) { photon_emit<MODE>(sc, tp, tp.first + idx, ps); active = true; }
            blk_lo = min(blk_hi, blk_lo + (int64_t)__popcll(need));
        }
        if (__ballot(active) == 0) { if (exhausted) break; else continue; }
        Deposit dep;
        dep.want = false;
        // The tail of a launch: the photons are handed out and a few lanes still follow paths of thousands of bounces
        // (total reflection inside the glass sphere), one lane's arithmetic per instruction.  Then the wavefront shares
        // a lane's closest-hit search: the ray is broadcast, lane i tests sphere i (the arithmetic of intersect_scene's
        // loop body), the smallest t wins and among equal t the lowest index -- the loop's strict `<` in index order.
        const unsigned long long am = __ballot(active);
        if (GRID == 0 && exhausted && __popcll(am) <= TRACE_COOP_MAX) {
            const bool go = active && !ps.rng.stop() && !(max3(ps.flux) <= 0.0);
            double my_best = SPPM_INF;
            int my_id = -1;
            for (unsigned long long mm = __ballot(go); mm != 0ull; mm &= mm - 1ull) {
                const int a = __ffsll((long long)mm) - 1;
                const d3 org = mk(rl_d(ps.org.x, a), rl_d(ps.org.y, a), rl_d(ps.org.z, a));
                const d3 dir = mk(rl_d(ps.dir.x, a), rl_d(ps.dir.y, a), rl_d(ps.dir.z, a));
                double best = SPPM_INF;
                int id = -1;
                for (int base = 0; base < sc.ns; base += 64) {
                    const int i = base + lane;
                    double t = SPPM_INF;
                    if (i < sc.ns) {
                        const d3 p_o = ld3(sc.sp[i].position) - org;
                        const double b = dot(p_o, dir);
                        const double D4 = b * b - dot(p_o, p_o) + sc.sp[i].radius * sc.sp[i].radius;
                        const double sq = sqrt(D4);
                        const double t1 = b - sq, t2 = b + sq;
                        const bool counts = !(D4 < 0.0) && !(t1 < SPPM_EPS && t2 < SPPM_EPS);
                        const double tt = (t1 > SPPM_EPS) ? t1 : t2;
                        if (counts && tt < SPPM_INF) t = tt;
                    }
                    const double m = wave_min_d(t);
                    if (m < best) { best = m; id = base + __ffsll((long long)__ballot(t == m)) - 1; }
                }
                if (lane == a) { my_best = best; my_id = id; }
            }
            // every lane then goes on with its own hit, in parallel
            if (active) {
                if (!go) active = false;
                else {
                    rays++;
                    if (my_id < 0) active = false;
                    else {
                        Hit is;
                        is.distance = my_best; is.object_id = my_id;
                        is.position = ps.org + ps.dir * my_best;
                        is.normal = normalize(is.position - ld3(sc.sp[my_id].position));
                        active = photon_shade(sc, ps, is, dep, err, tp.ppm != 0, MODE != 0 && tp.miss != 0);
                    }
                }
                if constexpr (MODE == 2)
                    if (!active) tp.consumed[ps.index] = ps.rng.stop() ? STREAM_OVER : (uint32_t)(ps.rng.pos - ps.index);
                if constexpr (MODE == 1)
                    if (!active && ps.rng.stop()) err[4] = 1;
            }
        } else
        if (active) {
            active = photon_bounce<RNG, GRID>(sc, ps, dep, err, rays, tp.ppm != 0, MODE != 0 && tp.miss != 0);
            if constexpr (MODE == 2)
                if (!active) tp.consumed[ps.index] = ps.rng.stop() ? STREAM_OVER : (uint32_t)(ps.rng.pos - ps.index);
            if constexpr (MODE == 1)
                if (!active && ps.rng.stop()) err[4] = 1;
        }
        if constexpr (MODE == 2) continue;
        // store the records of this step: slots come from the wave's chunk, a new chunk costs one atomic
        const unsigned long long dm = __ballot(dep.want);
        if (dm != 0 && !rec_full) {
            const int cnt = __popcll(dm);
            if (chunk_lo + cnt > chunk_hi) {
                if (lane < chunk_hi - chunk_lo) rec.key[chunk_lo + lane] = REC_INVALID;  // fewer than 64 left: mark them as holes
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(rec_count, (unsigned long long)REC_CHUNK);
                base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                       (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)base);
                chunk_lo = (int64_t)base;
                chunk_hi = (int64_t)base + REC_CHUNK;
                if (chunk_hi > rec.cap) { rec_full = true; chunk_lo = chunk_hi = 0; if (lane == 0) err[0] = 1; }
            }
            if (!rec_full) {
                if (dep.want) {
                    const int64_t slot = chunk_lo + (int64_t)mbcnt64(dm);
                    rec.key[slot] = dep.key;
                    double* f = rec.f + slot;
                    const int64_t c = rec.cap;
                    f[0] = dep.pos.x; f[c] = dep.pos.y; f[2 * c] = dep.pos.z;
                    f[3 * c] = dep.nrm.x; f[4 * c] = dep.nrm.y; f[5 * c] = dep.nrm.z;
                    f[6 * c] = dep.flux.x; f[7 * c] = dep.flux.y; f[8 * c] = dep.flux.z;
                }
                chunk_lo += cnt;
            }
        }
    }
    for (int64_t j = chunk_lo + lane; j < chunk_hi; j += 64) rec.key[j] = REC_INVALID;
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (MODE != 2 && lane == 0 && rays) atomicAdd(&counters[1], (unsigned long long)rays);  // [1]: rays cast by photon tracing
}

// Reference-stream mode: photon 0 starts at offset 0, photon i+1 where photon i stopped.  One
// wavefront follows consumed[] through an LDS window (the chain is sequential; each step is one LDS read).
#define CHAIN_WINDOW 2048
__global__ void __launch_bounds__(64) k_photon_chain(const uint32_t* __restrict__ consumed, int64_t u_len, int64_t count, uint32_t* __restrict__ start,
                                                     unsigned long long* __restrict__ total, int32_t* __restrict__ err)
{
    __shared__ uint32_t win[CHAIN_WINDOW];
    const int lane = threadIdx.x;
    int64_t cur = 0, wbase = -(int64_t)CHAIN_WINDOW;
    bool bad = false;
    for (int64_t i = 0; i < count; i++) {
        if (cur >= u_len) { bad = true; break; }
        if (cur >= wbase + CHAIN_WINDOW) {
            __syncthreads();
            wbase = cur;
            for (int k = lane; k < CHAIN_WINDOW; k += 64) win[k] = wbase + k < u_len ? consumed[wbase + k] : STREAM_OVER;
            __syncthreads();
        }
        const uint32_t c = win[cur - wbase];
        if (c == STREAM_OVER) { bad = true; break; }
        if (lane == 0) start[i] = (uint32_t)cur;
        cur += c;
    }
    if (lane == 0) { total[0] = (unsigned long long)cur; if (bad) err[4] = 1; }
}
