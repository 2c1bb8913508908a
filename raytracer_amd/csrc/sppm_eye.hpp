// sppm_eye.hpp -- the eye pass: camera rays and first hits, hitpoint arrays, create_point_loop per pixel (sppm2.h), the hitpoint list of classic PPM, and the reference-stream variants.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
#pragma once

__global__ void k_camera_rays(Camera cam, int W, int H, int sx, int sy, uint32_t seed_lens, const double* __restrict__ lens_u, double* __restrict__ rays)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int x = i % W, y = i / W;
    d3 o, d;
    if (cam.model == 1) {  // lens draws: the stream of image index (H-y-1)*W+x at pass 0, or draws 2i, 2i+1 of the sequential stream
        LensRng l;
        l.u = lens_u; l.pos = 2ll * i; l.mode = lens_u ? 2 : 1;
        l.xs = make_stream(seed_lens, (uint64_t)((H - y - 1) * W + x));
        camera_ray_lens(cam, x, y, sx, sy, l, o, d);
    } else
        camera_ray(cam, x, y, sx, sy, o, d);
    double* r = rays + 6 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
}

__global__ void k_intersect(DevScene sc, const double* __restrict__ rays, int64_t n, int32_t* __restrict__ id,
                            double* __restrict__ out)
{
    STAGE_SCENE(sc);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Hit h;
    const bool hit = intersect_scene(sc, ld3(rays + 6 * i), ld3(rays + 6 * i + 3), h);
    id[i] = hit ? h.object_id : -1;
    double* o = out + 7 * i;
    o[0] = h.distance;
    if (hit) { o[1] = h.position.x; o[2] = h.position.y; o[3] = h.position.z; o[4] = h.normal.x; o[5] = h.normal.y; o[6] = h.normal.z; }
    else { o[1] = o[2] = o[3] = o[4] = o[5] = o[6] = 0.0; }
}

struct HpArrays {
    double *pos, *nrm, *w, *flux, *ld, *tau;  // [3][P]
    double *r2, *dist, *nd;
    int32_t *n, *obj, *valid;
    long long* dep;  // per-pass rule only: [4][P] = the pass's deposits, fixed-point flux x y z and the count M (sppm_deposit.hpp)
    int P;
};

struct EyeParams {
    Camera cam; int W, H, samples, supersamples; int pass; uint32_t seed_eye; double initial_r2; int double_nee;
    // reference-stream mode only
    const double* u; int64_t u_len; uint32_t* cnt; unsigned long long* off; unsigned long long* total;
    // classic progressive photon mapping only: hitpoints per pixel (out) / first hitpoint slot per pixel (in),
    // both indexed by the pixel's place t in the render order
    uint32_t* hcnt; const long long* hoff;
    // thin-lens camera: seed of the per-(pass, pixel) lens streams; reference-stream mode: this pass's lens draws
    uint32_t seed_lens; const double* lens_u;
    // k_eye_pass only, all three or none: the bounding box of the hitpoints this pass leaves (bb_out, seeded by the caller)
    // and their sort keys (keys_out, by image index), quantised over the box of the pass BEFORE (bb_prev) -- what
    // k_hp_bbox and k_hp_sortkeys would do in two more launches and two more sweeps over the positions
    const unsigned long long* bb_prev; unsigned long long* bb_out; uint32_t* keys_out;
};

__device__ __forceinline__ LensRng make_lens(const EyeParams& ep, int t, int index)
{
    LensRng l;
    l.u = ep.lens_u; l.pos = 0; l.mode = 0; l.xs.seed(1u);
    if (ep.cam.model == 1) {
        if (ep.lens_u) { l.mode = 2; l.pos = 2ll * t * (ep.samples * ep.supersamples * ep.supersamples); }
        else { l.mode = 1; l.xs = make_stream(ep.seed_lens, (uint64_t)ep.pass * (uint64_t)(ep.W * ep.H) + (uint64_t)index); }
    }
    return l;
}

// One pixel of an eye pass: the sy/sx/s loops of src/sppm2.h:628-643 around create_point_loop (:156-315),
// in pieces so that a kernel can either run a pixel from start to end (eye_pixel) or keep one pixel per lane in
// flight and advance all lanes one ray at a time (k_eye_pass_refill).  t = y*W + x is the pixel's place in the
// reference's render order.  COMMIT = false runs the same paths without storing anything (used to count draws).
template <class RNG>
struct EyeState {
    RNG rng; LensRng lens;
    d3 org, dir, weight, ld;
    int x, y, index, sub, depth, valid;
    bool direct, wrote;  // wrote: a hitpoint was stored for this pixel in this pass
};

template <class RNG>
__device__ __forceinline__ void eye_pixel_begin(const EyeParams& ep, const HpArrays& hp, int t, EyeState<RNG>& e)
{
    const int P = ep.W * ep.H;
    e.x = t % ep.W; e.y = t / ep.W;
    e.index = (ep.H - e.y - 1) * ep.W + e.x;
    e.ld = mk(hp.ld[e.index], hp.ld[P + e.index], hp.ld[2 * P + e.index]);
    e.valid = hp.valid[e.index];
    e.wrote = false;
    e.lens = make_lens(ep, t, e.index);
    e.sub = 0;
}

// the next of the pixel's paths, in the order of the reference's loops (sy, sx, s); false when none is left
template <class RNG>
__device__ __forceinline__ bool eye_path_begin(const EyeParams& ep, EyeState<RNG>& e)
{
    const int ss = ep.supersamples, per = ep.samples;
    if (e.sub >= per * ss * ss) return false;
    const int sy = e.sub / (ss * per), sx = (e.sub / per) % ss;
    if (ep.cam.model == 1) camera_ray_lens(ep.cam, e.x, e.y, sx, sy, e.lens, e.org, e.dir);
    else camera_ray(ep.cam, e.x, e.y, sx, sy, e.org, e.dir);
    e.sub++;
    e.depth = -1;
    e.direct = true;
    e.weight = mk(1, 1, 1);
    return true;
}

// one trip of create_point_loop's loop; false when the path has ended
template <class RNG, bool COMMIT, int GRID>
__device__ __forceinline__ bool eye_path_step(const DevScene& sc, const EyeParams& ep, const HpArrays& hp, const HpArrays& out, EyeState<RNG>& e, uint32_t& rays)
{
    const int P = ep.W * ep.H;
    // (opaque to the compiler: with a loop-invariant index it computes the ~25 store addresses of a pixel once, ahead of
    // the path loop, and keeps them in 50 vector registers -- of which this kernel, at 128, then spills twice as many)
    int index = e.index;
    asm volatile("" : "+v"(index));
    e.depth++;
    Hit is;
    rays++;
    if (!intersect_scene<GRID>(sc, e.org, e.dir, is)) return false;
    const sppm_sphere* obj = &sc.sp[is.object_id];
    const d3 color = ld3(obj->color);
    const int type = obj->reflection_type;
    const d3 on = dot(is.normal, e.dir) < 0.0 ? is.normal : is.normal * -1.0;
    if (e.direct) e.ld = e.ld + had(e.weight, ld3(obj->emission));
    e.direct = type != SPPM_DIFFUSE;
    if (type == SPPM_DIFFUSE) {
        const d3 est = next_event_est<RNG, GRID>(sc, is, on, e.rng, rays);
        e.ld = e.ld + had(e.weight, est);
        // create or overwrite the pixel's hitpoint (src/sppm2.h:204-227)
        if constexpr (COMMIT) {
            out.pos[index] = is.position.x; out.pos[P + index] = is.position.y; out.pos[2 * P + index] = is.position.z;
            out.nrm[index] = is.normal.x; out.nrm[P + index] = is.normal.y; out.nrm[2 * P + index] = is.normal.z;
            out.w[index] = e.weight.x; out.w[P + index] = e.weight.y; out.w[2 * P + index] = e.weight.z;
            out.obj[index] = is.object_id; out.dist[index] = is.distance;
            e.wrote = true;
        }
        if (e.valid) return false;
        e.valid = 1;
        if constexpr (COMMIT) {
            hp.r2[index] = ep.initial_r2;
            hp.n[index] = 0;  // the reference leaves photon_count unset (src/sppm2.h:108-116); its call site passes 0
            hp.flux[index] = 0.0; hp.flux[P + index] = 0.0; hp.flux[2 * P + index] = 0.0;
            if (hp.nd) {  // per-pass rule (the deposits of a pixel that was not valid are zero already)
                hp.nd[index] = 0.0;
                hp.tau[index] = 0.0; hp.tau[P + index] = 0.0; hp.tau[2 * P + index] = 0.0;
            }
        }
        // the reference's `break` leaves the switch only: fall into the roulette block and go round again
        // with the same ray (second direct sample)
        if (!ep.double_nee) return false;
    } else if (type == SPPM_SPECULAR) {
        e.org = is.position;
        e.dir = reflect_dir(e.dir, is.normal);
        e.weight = had(e.weight, color);
    } else {
        const d3 rdir = reflect_dir(e.dir, is.normal);
        bool into; double nnt, Re; d3 tdir;
        e.org = is.position;
        if (!glass(e.dir, is.normal, on, true, into, nnt, tdir, Re)) {
            e.dir = rdir;
            e.weight = had(e.weight, color);
        } else {
            const double nnt2 = nnt * nnt;
            const double Tr = (1.0 - Re) * nnt2;
            const double prob = 0.25 + 0.5 * Re;
            e.rng.decision();
            if (e.rng.next01() < prob) {
                e.dir = rdir;
                e.weight = had(e.weight, color * Re / prob);
            } else {
                e.dir = tdir;
                e.weight = had(e.weight, color * Tr / (1.0 - prob));
            }
        }
    }
    {   // Russian roulette, src/sppm2.h:298-313
        double rr = max3(color);
        if (e.depth >= SPPM_KDEPTH) {
            rr *= ldexp(1.0, -(e.depth - SPPM_KDEPTH));  // std::pow(0.5, depth - MaxDepth)
            e.rng.decision();
            if (e.rng.next01() >= rr) return false;
        } else
            rr = 1.0;
        e.weight = e.weight / rr;
    }
    return true;
}

template <class RNG, bool COMMIT>
__device__ __forceinline__ void eye_pixel_end(const EyeParams& ep, const HpArrays& hp, const HpArrays& out, const EyeState<RNG>& e)
{
    if constexpr (COMMIT) {
        // `out` is the view the NEXT gather reads (the eye pass of pass i+1 may run while the gather of pass i
        // still reads `hp`): the direct term, the valid flag, and -- where this pass stored no hitpoint -- the
        // hitpoint the pixel already had (the reference overwrites a pixel's hitpoint only on a diffuse hit)
        const int P = ep.W * ep.H;
        int index = e.index;
        asm volatile("" : "+v"(index));  // (see eye_path_step)
        out.ld[index] = e.ld.x; out.ld[P + index] = e.ld.y; out.ld[2 * P + index] = e.ld.z;
        out.valid[index] = e.valid;
        if (!e.wrote && e.valid) {
            out.pos[index] = hp.pos[index]; out.pos[P + index] = hp.pos[P + index]; out.pos[2 * P + index] = hp.pos[2 * P + index];
            out.nrm[index] = hp.nrm[index]; out.nrm[P + index] = hp.nrm[P + index]; out.nrm[2 * P + index] = hp.nrm[2 * P + index];
            // per-pass rule: the end of every pass zeroes the weight (src/stochastic_ppm.h:740), so a hitpoint that this
            // pass did not renew carries none -- written here, so that this pass may run ahead of that zeroing
            const bool keep_w = hp.nd == nullptr;
            out.w[index] = keep_w ? hp.w[index] : 0.0; out.w[P + index] = keep_w ? hp.w[P + index] : 0.0; out.w[2 * P + index] = keep_w ? hp.w[2 * P + index] : 0.0;
            out.obj[index] = hp.obj[index]; out.dist[index] = hp.dist[index];
        }
    }
}

template <class RNG, bool COMMIT, int GRID = SCENE_ANY>
__device__ __forceinline__ void eye_pixel(const DevScene& sc, const EyeParams& ep, const HpArrays& hp, const HpArrays& out, int t, RNG& rng, uint32_t& rays)
{
    EyeState<RNG> e;
    e.rng = rng;
    eye_pixel_begin(ep, hp, t, e);
    while (eye_path_begin(ep, e))
        for (;;) {
            const bool go = eye_path_step<RNG, COMMIT, GRID>(sc, ep, hp, out, e, rays);
            if (e.rng.stop()) { rng = e.rng; return; }  // sequential-stream samplers only: the pixel is abandoned
            if (!go) break;
        }
    rng = e.rng;
    eye_pixel_end<RNG, COMMIT>(ep, hp, out, e);
}

// Register budget of the eye pass: 128 per lane (four wavefronts per SIMD), although the kernel would like 168.  The next
// pass's eye pass runs beside the gather, whose persistent wavefronts hold 3 x 128 registers of every SIMD: with 128 a
// wavefront of the eye pass fits into what is left and the pass finishes in the gather's shadow; with 168 it waited
// for the gather's tail (measured: +2 % on the pass).
#ifndef EYE_WAVES
#define EYE_WAVES 4
#endif
// One thread per pixel, one XorShift stream per (pass, image index).
template <int GRID>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EYE_WAVES, EYE_WAVES))) k_eye_pass(DevScene sc, EyeParams ep, HpArrays hp, HpArrays out, unsigned long long* __restrict__ counters)
{
    STAGE_SCENE_T(sc, GRID);
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = ep.W * ep.H;
    // a wavefront takes an 8x8 block of pixels rather than 64 of one row: neighbouring eye paths resemble each
    // other, and a wavefront waits for its longest one (which pixel a thread takes changes no result)
    if (((ep.W | ep.H) & 7) == 0) {
        const int tile = t >> 6, w = t & 63, tiles_x = ep.W >> 3;
        t = ((tile / tiles_x) * 8 + (w >> 3)) * ep.W + (tile % tiles_x) * 8 + (w & 7);
    }
    uint32_t rays = 0;
    if (t < P) {
        const int index = (ep.H - t / ep.W - 1) * ep.W + t % ep.W;
        XorShift rng = make_stream(ep.seed_eye, (uint64_t)ep.pass * (uint64_t)P + (uint64_t)index);
        eye_pixel<XorShift, true, GRID>(sc, ep, hp, out, t, rng, rays);
    }
    if (ep.keys_out) {  // (the same for every thread of the launch)
        const double BIG = 1e300;
        double lo[3] = {BIG, BIG, BIG}, hi[3] = {-BIG, -BIG, -BIG};
        if (t < P) {
            int index = (ep.H - t / ep.W - 1) * ep.W + t % ep.W;
            asm volatile("" : "+v"(index));  // (see eye_path_step)
            hp_key_t key = MORTON_INVALID;
            if (out.valid[index]) {  // what this thread has just stored, or copied from the view before
                const double x = out.pos[index], y = out.pos[P + index], z = out.pos[2 * P + index];
                lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
                key = hp_order_key(x, y, z, ep.bb_prev);
            }
            ep.keys_out[index] = key;
        }
        __shared__ double s_lo[4][3], s_hi[4][3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            for (int off = 32; off > 0; off >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], off)); }
            if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
        }
        __syncthreads();
        if (threadIdx.x < 3) {  // one pair of atomics per block and axis
            const int a = threadIdx.x;
            const double l = fmin(fmin(s_lo[0][a], s_lo[1][a]), fmin(s_lo[2][a], s_lo[3][a]));
            const double h = fmax(fmax(s_hi[0][a], s_hi[1][a]), fmax(s_hi[2][a], s_hi[3][a]));
            if (l <= h) { atomicMin(&ep.bb_out[a], ord_d(l)); atomicMax(&ep.bb_out[3 + a], ord_d(h)); }
        }
    }
    // rays counter: one atomic per block (32 400 wavefronts adding to one address were 2 % of a late pass)
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    __shared__ uint32_t s_rays[4];
    if ((threadIdx.x & 63) == 0) s_rays[threadIdx.x >> 6] = rays;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long r = (unsigned long long)s_rays[0] + s_rays[1] + s_rays[2] + s_rays[3];
        if (r) atomicAdd(&counters[0], r);
    }
}

// The same pass for scenes whose eye paths differ much in length (thousands of small mirrors and glass balls):
// persistent wavefronts keep one pixel per lane in flight, all lanes advance one ray together, and a lane whose
// pixel is done takes the next pixel of its wavefront's block of 64 (blocks come from one counter; 8x8 pixel
// blocks where the image allows).  Which lane runs which pixel changes no result.
template <int GRID>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_eye_pass_refill(DevScene sc, EyeParams ep, HpArrays hp, HpArrays out,
                                                                                                   unsigned long long* __restrict__ next_block,
                                                                                                   unsigned long long* __restrict__ counters)
{
    STAGE_SCENE_T(sc, GRID);
    const int lane = threadIdx.x & 63;
    const int P = ep.W * ep.H;
    const bool blocks8 = ((ep.W | ep.H) & 7) == 0;
    uint32_t rays = 0;
    EyeState<XorShift> e;
    e.rng.seed(1u); e.lens.u = nullptr; e.lens.pos = 0; e.lens.mode = 0; e.lens.xs.seed(1u);
    e.org = mk(0, 0, 0); e.dir = mk(0, 0, 1); e.weight = mk(0, 0, 0); e.ld = mk(0, 0, 0);
    e.x = e.y = e.index = e.sub = e.depth = e.valid = 0; e.direct = false; e.wrote = false;
    bool active = false, exhausted = false;
    long long blk_lo = 0, blk_hi = 0;  // wave-uniform: the unassigned rest of this wave's current block
    for (;;) {
        const unsigned long long need = __ballot(!active);
        if (need != 0 && !exhausted) {
            if (blk_lo >= blk_hi) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(next_block, 64ull);
                base = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                       (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)base);
                blk_lo = (long long)base;
                blk_hi = min((long long)base + 64, (long long)P);
                if (blk_lo >= P) { exhausted = true; blk_lo = blk_hi = 0; }
            }
            const long long idx = blk_lo + (long long)mbcnt64(need);
            if (!active && idx < blk_hi) {
                int t = (int)idx;
                if (blocks8) {
                    const int tile = t >> 6, w = t & 63, tiles_x = ep.W >> 3;
                    t = ((tile / tiles_x) * 8 + (w >> 3)) * ep.W + (tile % tiles_x) * 8 + (w & 7);
                }
                eye_pixel_begin(ep, hp, t, e);
                e.rng = make_stream(ep.seed_eye, (uint64_t)ep.pass * (uint64_t)P + (uint64_t)e.index);
                active = eye_path_begin(ep, e);
                if (!active) eye_pixel_end<XorShift, true>(ep, hp, out, e);
            }
            blk_lo = min(blk_hi, blk_lo + (long long)__popcll(need));
        }
        if (__ballot(active) == 0) { if (exhausted) break; else continue; }
        if (active) {
            const bool go = eye_path_step<XorShift, true, GRID>(sc, ep, hp, out, e, rays);
            if (!go && !eye_path_begin(ep, e)) {
                eye_pixel_end<XorShift, true>(ep, hp, out, e);
                active = false;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if (lane == 0 && rays) atomicAdd(&counters[0], (unsigned long long)rays);
}

// ---- classic progressive photon mapping: the eye pass ------------------------------------------------------
// create_point, src/progressive_photonmapping.h:496-615, for one pixel: the sy/sx/s loops of render()
// (:769-791) around a recursion that leaves a hitpoint at EVERY diffuse surface it reaches.  Glass follows
// both branches while depth <= 2 (reflection first, :606-609) and one branch chosen by a draw afterwards
// (:586-603); roulette from depth > kDepth (:519-524); the weight is a scalar.  The recursion is a loop with
// a stack: only the refracted branches of depths 0, 1 and 2 can be pending at the same time, and the draws
// are consumed in the reference's depth-first order.  Hitpoints go to slots base, base+1, ... in creation
// order (COMMIT); the function returns how many the pixel has.
#define PPM_DEPTH_LIMIT 64  // kDepthLimit, src/radiance.h:18
template <class RNG, bool COMMIT>
__device__ __forceinline__ int ppm_pixel(const DevScene& sc, const EyeParams& ep, const HpArrays& out, int t, long long base, RNG& rng, uint32_t& rays)
{
    const int P = out.P;
    const int x = t % ep.W, y = t / ep.W;
    int count = 0;
    for (int sy = 0; sy < ep.supersamples; sy++)
        for (int sx = 0; sx < ep.supersamples; sx++)
            for (int s = 0; s < ep.samples; s++) {
                d3 org, dir;
                camera_ray(ep.cam, x, y, sx, sy, org, dir);
                double weight = 1.0;
                int depth = 0;
                d3 st_org[3], st_dir[3]; double st_w[3]; int st_depth[3];
                int sp = 0;
                for (;;) {  // one create_point call per trip
                    bool ended = true;
                    Hit is;
                    rays++;
                    if (intersect_scene(sc, org, dir, is)) {
                        const sppm_sphere* obj = &sc.sp[is.object_id];
                        const d3 color = ld3(obj->color);
                        const int type = obj->reflection_type;
                        const d3 on = dot(is.normal, dir) < 0.0 ? is.normal : is.normal * -1.0;
                        double rr = max3(color);
                        if (depth > PPM_DEPTH_LIMIT) rr *= ldexp(1.0, -(depth - PPM_DEPTH_LIMIT));  // std::pow(0.5, depth - kDepthLimit)
                        bool alive = true;
                        if (depth > SPPM_KDEPTH) {
                            rng.decision();
                            if (rng.next01() >= rr) alive = false;
                        } else
                            rr = 1.0;
                        if (alive) {
                            if (type == SPPM_DIFFUSE) {  // :529-534
                                if constexpr (COMMIT) {
                                    const long long h = base + count;
                                    const double w = weight / rr;
                                    out.pos[h] = is.position.x; out.pos[P + h] = is.position.y; out.pos[2 * (long long)P + h] = is.position.z;
                                    out.nrm[h] = is.normal.x; out.nrm[P + h] = is.normal.y; out.nrm[2 * (long long)P + h] = is.normal.z;
                                    out.w[h] = w; out.w[P + h] = w; out.w[2 * (long long)P + h] = w;
                                    out.obj[h] = is.object_id; out.dist[h] = is.distance; out.valid[h] = 1;
                                    out.r2[h] = ep.initial_r2;
                                    out.n[h] = 0;  // the constructor drops photon_count (:46-55); its call site passes 0
                                    out.flux[h] = 0.0; out.flux[P + h] = 0.0; out.flux[2 * (long long)P + h] = 0.0;
                                }
                                count++;
                            } else if (type == SPPM_SPECULAR) {  // :537-544
                                org = is.position;
                                dir = reflect_dir(dir, is.normal);
                                weight = weight / rr;
                                depth++;
                                ended = false;
                            } else {
                                const d3 rdir = reflect_dir(dir, is.normal);
                                bool into; double nnt, Re; d3 tdir;
                                org = is.position;
                                if (!glass(dir, is.normal, on, false, into, nnt, tdir, Re)) {  // :561-566
                                    dir = rdir;
                                    weight = weight / rr;
                                } else {
                                    const double nnt2 = nnt * nnt;  // std::pow(nnt, 2.0)
                                    const double Tr = (1.0 - Re) * nnt2;
                                    const double prob = 0.25 + 0.5 * Re;
                                    if (depth > 2) {
                                        rng.decision();
                                        if (rng.next01() < prob) { dir = rdir; weight = weight * Re / (prob * rr); }
                                        else { dir = tdir; weight = weight * Tr / ((1.0 - prob) * rr); }
                                    } else {  // both: the reflected branch now, the refracted one once that subtree is done
                                        st_org[sp] = org; st_dir[sp] = tdir; st_w[sp] = weight * Tr / rr; st_depth[sp] = depth + 1;
                                        sp++;
                                        dir = rdir;
                                        weight = weight * Re / rr;
                                    }
                                }
                                depth++;
                                ended = false;
                            }
                        }
                    }
                    if (rng.stop()) return count;  // sequential-stream samplers only
                    if (ended) {
                        if (sp == 0) break;
                        sp--;
                        org = st_org[sp]; dir = st_dir[sp]; weight = st_w[sp]; depth = st_depth[sp];
                    }
                }
            }
    return count;
}

// One thread per pixel (t = place in the render order), one XorShift stream per image index.  COMMIT = false
// counts the pixel's hitpoints (hcnt[t]); COMMIT = true stores them from slot hoff[t] on.
template <bool COMMIT>
__global__ void __launch_bounds__(256) k_ppm_eye(DevScene sc, EyeParams ep, HpArrays out, unsigned long long* __restrict__ counters, int32_t* __restrict__ err)
{
    STAGE_SCENE(sc);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int npix = ep.W * ep.H;
    uint32_t rays = 0;
    if (t < npix) {
        const int index = (ep.H - t / ep.W - 1) * ep.W + t % ep.W;
        XorShift rng = make_stream(ep.seed_eye, (uint64_t)index);
        const int n = ppm_pixel<XorShift, COMMIT>(sc, ep, out, t, COMMIT ? ep.hoff[t] : 0ll, rng, rays);
        if constexpr (COMMIT) { if ((long long)n != ep.hoff[t + 1] - ep.hoff[t]) err[5] = 1; }
        else ep.hcnt[t] = (uint32_t)n;
    }
    if constexpr (COMMIT) {
        for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
        if ((threadIdx.x & 63) == 0 && rays) atomicAdd(&counters[0], (unsigned long long)rays);
    }
}

// image assembly of render(), src/progressive_photonmapping.h:807-813: one thread per pixel sums its
// hitpoints in creation order (the reference walks its kd-sorted vector; the two orders differ by the
// rounding of a sum of a few terms).
__global__ void k_ppm_resolve(HpArrays hp, DevScene sc, const long long* __restrict__ hoff, int W, int H, int64_t photon_num, double spp, double* __restrict__ image)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= W * H) return;
    const int index = (H - t / W - 1) * W + t % W;
    const long long P = hp.P;
    d3 c = mk(0, 0, 0);
    for (long long h = hoff[t]; h < hoff[t + 1]; h++) {
        const d3 em = ld3(sc.sp[hp.obj[h]].emission);
        const d3 flux = mk(hp.flux[h], hp.flux[P + h], hp.flux[2 * P + h]);
        c = c + (em + flux * hp.w[h] * (1.0 / (SPPM_PI * hp.r2[h] * (double)photon_num))) / spp;
    }
    image[3 * (size_t)index] = c.x; image[3 * (size_t)index + 1] = c.y; image[3 * (size_t)index + 2] = c.z;
}

// ---- reference-stream mode of the eye pass ----------------------------------------------------------
// The reference draws every pixel's samples from ONE sequential stream (src/sppm2.h:621), so pixel t
// starts where pixels 0..t-1 stopped.  Most pixels consume a number of draws that does not depend on
// the values drawn (a diffuse first hit: the light samples only); the rest (a glass vertex, roulette
// from depth 5 on) are "dynamic".  Three kernels: count/classify in parallel, one wavefront walks the
// pixels in render order and resolves the dynamic ones with the real stream, then the pass itself runs
// in parallel from the offsets found.  DYN_PIXEL marks a dynamic pixel in cnt[].
#define DYN_PIXEL 0x80000000u
template <int ALGO>  // 0: sppm2.h, 1: progressive_photonmapping.h
__global__ void __launch_bounds__(256) k_eye_classify(DevScene sc, EyeParams ep, HpArrays hp)
{
    STAGE_SCENE(sc);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ep.W * ep.H) return;
    ProbeRng rng;
    rng.count = 0; rng.dyn = false;
    uint32_t rays = 0;
    if constexpr (ALGO == 0) eye_pixel<ProbeRng, false>(sc, ep, hp, hp, t, rng, rays);
    else (void)ppm_pixel<ProbeRng, false>(sc, ep, hp, t, 0ll, rng, rays);
    ep.cnt[t] = rng.dyn ? DYN_PIXEL : (uint32_t)rng.count;
}

template <int ALGO>
__global__ void __launch_bounds__(64) k_eye_offsets(DevScene sc, EyeParams ep, HpArrays hp, int32_t* __restrict__ err)
{
    STAGE_SCENE(sc);
    const int lane = threadIdx.x;
    const int P = ep.W * ep.H;
    unsigned long long cur = 0;
    for (int base = 0; base < P; base += 64) {
        const int t = base + lane;
        const uint32_t c = t < P ? ep.cnt[t] : 0u;
        const unsigned long long dyn = __ballot(t < P && (c & DYN_PIXEL) != 0u);
        int pos = 0;
        while (pos < 64) {
            const unsigned long long rest = dyn & (pos == 0 ? ~0ull : (~0ull << pos));
            const int f = rest ? __ffsll((long long)rest) - 1 : 64;
            // lanes [pos, f): offsets from a prefix sum of their fixed counts
            const bool in = lane >= pos && lane < f;
            unsigned long long incl = in ? (unsigned long long)c : 0ull;
            for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            if (in && t < P) ep.off[t] = cur + incl - c;
            cur += __shfl(incl, 63);
            if (f < 64) {  // the dynamic pixel: run its paths on the real stream to see where it stops
                unsigned long long used = 0;
                if (lane == f) {
                    ep.off[t] = cur;
                    StreamRng rng; rng.u = ep.u; rng.pos = (int64_t)cur; rng.end = ep.u_len;
                    uint32_t rays = 0;
                    if constexpr (ALGO == 0) eye_pixel<StreamRng, false>(sc, ep, hp, hp, t, rng, rays);
                    else (void)ppm_pixel<StreamRng, false>(sc, ep, hp, t, 0ll, rng, rays);
                    used = (unsigned long long)rng.pos - cur;
                    if (rng.stop()) err[4] = 1;
                }
                cur += __shfl(used, f);
            }
            pos = f + 1;
        }
    }
    if (lane == 0) { ep.total[0] = cur; if ((int64_t)cur > ep.u_len) err[4] = 1; }
}

__global__ void __launch_bounds__(256) k_eye_pass_stream(DevScene sc, EyeParams ep, HpArrays hp, HpArrays out, unsigned long long* __restrict__ counters,
                                                        int32_t* __restrict__ err)
{
    STAGE_SCENE(sc);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = ep.W * ep.H;
    uint32_t rays = 0;
    if (t < P) {
        StreamRng rng; rng.u = ep.u; rng.pos = (int64_t)ep.off[t]; rng.end = ep.u_len;
        eye_pixel<StreamRng, true>(sc, ep, hp, out, t, rng, rays);
        // the pixel must stop exactly where the walk said the next one starts
        const unsigned long long next = t + 1 < P ? ep.off[t + 1] : ep.total[0];
        if ((unsigned long long)rng.pos != next) err[5] = 1;
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(&counters[0], (unsigned long long)rays);
}

// classic PPM on the reference's stream: pixel t reads from off[t]; COMMIT as in k_ppm_eye
template <bool COMMIT>
__global__ void __launch_bounds__(256) k_ppm_eye_stream(DevScene sc, EyeParams ep, HpArrays out, unsigned long long* __restrict__ counters, int32_t* __restrict__ err)
{
    STAGE_SCENE(sc);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int npix = ep.W * ep.H;
    uint32_t rays = 0;
    if (t < npix) {
        StreamRng rng; rng.u = ep.u; rng.pos = (int64_t)ep.off[t]; rng.end = ep.u_len;
        const int n = ppm_pixel<StreamRng, COMMIT>(sc, ep, out, t, COMMIT ? ep.hoff[t] : 0ll, rng, rays);
        const unsigned long long next = t + 1 < npix ? ep.off[t + 1] : ep.total[0];
        if ((unsigned long long)rng.pos != next) err[5] = 1;
        if constexpr (COMMIT) { if ((long long)n != ep.hoff[t + 1] - ep.hoff[t]) err[5] = 1; }
        else ep.hcnt[t] = (uint32_t)n;
    }
    if constexpr (COMMIT) {
        for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off);
        if ((threadIdx.x & 63) == 0 && rays) atomicAdd(&counters[0], (unsigned long long)rays);
    }
}
