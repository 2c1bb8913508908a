// sppm_device.hpp -- device-side building blocks of the SPPM path (gfx950, FP64).
//
// Every function keeps the association order of the reference expression it implements (cited
// per function, paths relative to the reference root) so that, built with -ffp-contract=off, the
// results are bit-identical to oracle/sppm_oracle.c (sampler = XorShift, math = detmath).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sppm_hip.h"
#include "detmath.h"

#define SPPM_PI 3.14159265358979323846
#define SPPM_EPS 1e-6   // src/constant.h:8
#define SPPM_INF 1e128  // src/constant.h:7
#define SPPM_IOR 1.5    // src/material.h:17
#define SPPM_KDEPTH 5   // src/radiance.h:17

struct d3 { double x, y, z; };

__device__ __forceinline__ d3 mk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
// src/vec.h:11-25
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator*(d3 a, double b) { return mk(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ d3 operator/(d3 a, double b) { return mk(a.x / b, a.y / b, a.z / b); }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double len2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ d3 normalize(d3 a) { return a * (1.0 / sqrt(len2(a))); }
__device__ __forceinline__ d3 had(d3 a, d3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ d3 cross(d3 a, d3 b)
{
    return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
__device__ __forceinline__ double max3(d3 c) { const double m = c.y > c.z ? c.y : c.z; return c.x > m ? c.x : m; }

// ---- edupt::XorShift, src/random.h:9-33, plus this build's stream derivation -------------------
struct XorShift {
    uint32_t s0, s1, s2, s3;
    __device__ __forceinline__ void seed(uint32_t initial_seed)
    {
        uint32_t s = initial_seed;
        s0 = s = 1812433253u * (s ^ (s >> 30)) + 1u;
        s1 = s = 1812433253u * (s ^ (s >> 30)) + 2u;
        s2 = s = 1812433253u * (s ^ (s >> 30)) + 3u;
        s3 = s = 1812433253u * (s ^ (s >> 30)) + 4u;
    }
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t t = s0 ^ (s0 << 11);
        s0 = s1; s1 = s2; s2 = s3;
        return s3 = (s3 ^ (s3 >> 19)) ^ (t ^ (t >> 8));
    }
    // (double)next() / UINT_MAX, src/random.h:23 (can return exactly 1.0, as the reference's can)
    __device__ __forceinline__ double next01() { return (double)next() / 4294967295.0; }
    // hooks of the sequential-stream samplers below (nothing to do for a per-path stream)
    __device__ __forceinline__ void decision() {}
    __device__ __forceinline__ bool stop() const { return false; }
};

// ---- the reference's own sampler: ONE sequential stream of uniform doubles ------------------------
// (src/sampler/value_sampler.h:8-15: std::mt19937 + uniform_real_distribution<double>(0,1), shared by
// every pixel of every eye pass / every photon of a photon pass).  The host generates the stream
// (std::mt19937, same library code as the reference); a path reads it from the offset where the paths
// before it stopped.  See "reference-stream mode" in sppm_eye.hpp / sppm_trace.hpp for how those offsets are found
// in parallel.
struct StreamRng {
    const double* u; int64_t pos, end;
    __device__ __forceinline__ double next01() { const double v = pos < end ? u[pos] : 0.5; pos++; return v; }
    __device__ __forceinline__ void decision() {}
    __device__ __forceinline__ bool stop() const { return pos > end; }  // ran past the uploaded part: abandon the path
};
// Counts the draws of a path whose control flow does not depend on them, and flags the ones where it does.
struct ProbeRng {
    int count; bool dyn;
    __device__ __forceinline__ double next01() { count++; return 0.5; }
    __device__ __forceinline__ void decision() { dyn = true; }  // called right before a draw that steers the path
    __device__ __forceinline__ bool stop() const { return dyn; }
};
__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// Low 32 key bits through a bijective mixer (distinct keys < 2^32 -> distinct seeds), expanded by
// the reference's constructor; high key bits perturb the second state word.
__device__ __forceinline__ XorShift make_stream(uint32_t base, uint64_t key)
{
    XorShift r;
    r.seed(fmix32((uint32_t)key ^ base));
    r.s1 ^= (uint32_t)(key >> 32) * 0x9E3779B9u;
    return r;
}

// ---- scene ----------------------------------------------------------------------------------
struct Hit { double distance; d3 position, normal; int object_id; };

// ---- wave-level helpers: a lane's double to all lanes, min / max over the wavefront
__device__ __forceinline__ double rl_d(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double first_d(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// the number of set bits of a wave mask below this lane (v_mbcnt: no per-lane mask register)
__device__ __forceinline__ unsigned int mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
// A wave-uniform double moved into scalar registers for good.  (first_d of a value the compiler can prove uniform is
// folded away and the value stays in vector registers; this one it cannot see through.)
__device__ __forceinline__ double pin_scalar_d(double v)
{
#ifdef PIN_RAW_ASM  // the first form, WRONG: kept only to reproduce DESIGN.md section 10 ("A pin that was wrong")
    int lo, hi;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(lo) : "v"(__double2loint(v)));
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(hi) : "v"(__double2hiint(v)));
    return __hiloint2double(hi, lo);
#else
    // (the value passes through an empty asm, so the compiler no longer knows it is uniform and cannot fold the
    // readfirstlane away; the instruction itself is the compiler's, with its hazard handling)
    int lo = __double2loint(v), hi = __double2hiint(v);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return __hiloint2double(__builtin_amdgcn_readfirstlane(hi), __builtin_amdgcn_readfirstlane(lo));
#endif
}
// min / max over the wavefront by data-parallel primitives (DPP) instead of LDS permutes: four shifts within each row of
// 16 lanes, then the rows' last lanes broadcast to the rows after them; lane 63 ends up with the result.  (A lane whose
// DPP source does not exist keeps its own value: min and max are idempotent.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_d(double v)
{
    v = fmax(v, dpp_d<0x111, 0xF>(v)); v = fmax(v, dpp_d<0x112, 0xF>(v)); v = fmax(v, dpp_d<0x114, 0xF>(v)); v = fmax(v, dpp_d<0x118, 0xF>(v));  // row_shr 1, 2, 4, 8
    v = fmax(v, dpp_d<0x142, 0xA>(v));  // row_bcast 15 into rows 1 and 3
    v = fmax(v, dpp_d<0x143, 0xC>(v));  // row_bcast 31 into rows 2 and 3
    return rl_d(v, 63);
}
__device__ __forceinline__ double wave_min_d(double v)
{
    v = fmin(v, dpp_d<0x111, 0xF>(v)); v = fmin(v, dpp_d<0x112, 0xF>(v)); v = fmin(v, dpp_d<0x114, 0xF>(v)); v = fmin(v, dpp_d<0x118, 0xF>(v));
    v = fmin(v, dpp_d<0x142, 0xA>(v));
    v = fmin(v, dpp_d<0x143, 0xC>(v));
    return rl_d(v, 63);
}

// native vectors: two floats are one operand of the packed single-precision instructions (v_pk_fma_f32 ...)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// inclusive prefix sum over the wavefront, the same way: Hillis-Steele inside each row of 16 lanes (a lane whose source
// lies outside its row adds 0), then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3.
// Every lane must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

// the same by LDS permutes (the ordered gather keeps these: with the DPP forms its register allocation came out worse
// and the kernel 7 % slower)
__device__ __forceinline__ double wave_max_perm_d(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return first_d(v);
}
__device__ __forceinline__ double wave_min_perm_d(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return first_d(v);
}

// edupt::intersect_scene, src/scene.h:53-76, over Sphere::intersect, src/sphere.h:31-56.
// The kernels stage the sphere table in LDS first (STAGE_SCENE), so the per-sphere fields are LDS
// broadcasts, and the body is branch-free so that the independent per-sphere chains (the square root
// above all) overlap.  Same decisions as the reference's early-outs: a sphere counts when
// !(D4 < 0) and !(t1 < eps && t2 < eps), and the first sphere with the smallest t wins.
__device__ __forceinline__ bool intersect_scene(const sppm_sphere* __restrict__ sp, int ns, d3 org, d3 dir, Hit& h)
{
    double best = SPPM_INF;
    int id = -1;
#pragma unroll 3
    for (int i = 0; i < ns; i++) {
        const d3 p_o = ld3(sp[i].position) - org;
        const double b = dot(p_o, dir);
        const double D4 = b * b - dot(p_o, p_o) + sp[i].radius * sp[i].radius;
        const double sq = sqrt(D4);  // NaN for D4 < 0: every comparison below is then false
        const double t1 = b - sq, t2 = b + sq;
        const bool counts = !(D4 < 0.0) && !(t1 < SPPM_EPS && t2 < SPPM_EPS);
        const double t = (t1 > SPPM_EPS) ? t1 : t2;
        if (counts && t < best) { best = t; id = i; }
    }
    h.distance = best;
    h.object_id = id;
    if (id < 0) return false;
    h.position = org + dir * best;                            // sphere.h:53
    h.normal = normalize(h.position - ld3(sp[id].position));  // sphere.h:54
    return true;
}

// ---- scenes too large for the brute-force loop: a uniform grid over the small spheres ---------------------
// The reference's closest hit is a minimum over ALL spheres (src/scene.h:53-76); which spheres are tested
// does not change it as long as every sphere the ray can hit is among them, and ties keep the reference's
// rule (the lowest index wins: strict `<` in index order).  Spheres too large for the cells ("big": walls,
// light) are tested for every ray; the others are listed in every cell their bounding box, grown by a
// margin far above the rounding of the cell walk, overlaps.  The walk (Amanatides & Woo) stops when the
// closest hit so far lies before the exit of the current cell.  The per-sphere arithmetic is the brute-force
// loop's, so the hit is bit-identical to it.
struct DevScene {
    const sppm_sphere* sp; int ns;
    // grid mode (n_cells > 0): the table stays in HBM/L2
    int n_cells, n_big, gx, gy, gz;
    double g_lo[3], g_inv, g_cell;
    const int32_t* big;         // ids of the big spheres, ascending
    const uint32_t* cell_start; // [n_cells + 1]
    const int32_t* cell_items;  // sphere ids per cell, ascending
    const double* cell_geom;    // [entries][4]: centre and radius of cell_items[k], so a test costs one level of loads less
    const uint32_t* cell_mask;  // one bit per cell: it lists something (at most 2^18 cells: the kernels keep this in LDS)
};

__device__ __forceinline__ void sphere_test(const sppm_sphere* __restrict__ sp, int i, d3 org, d3 dir, double& best, int& id)
{
    const d3 p_o = ld3(sp[i].position) - org;
    const double b = dot(p_o, dir);
    const double D4 = b * b - dot(p_o, p_o) + sp[i].radius * sp[i].radius;
    const double sq = sqrt(D4);
    const double t1 = b - sq, t2 = b + sq;
    const bool counts = !(D4 < 0.0) && !(t1 < SPPM_EPS && t2 < SPPM_EPS);
    const double t = (t1 > SPPM_EPS) ? t1 : t2;
    if (counts && (t < best || (t == best && i < id))) { best = t; id = i; }
}

// the same test from a cell's copy of the sphere's centre and radius
__device__ __forceinline__ void sphere_test_geom(const double* __restrict__ g, int i, d3 org, d3 dir, double& best, int& id)
{
    const d3 p_o = mk(g[0], g[1], g[2]) - org;
    const double b = dot(p_o, dir);
    const double D4 = b * b - dot(p_o, p_o) + g[3] * g[3];
    const double sq = sqrt(D4);
    const double t1 = b - sq, t2 = b + sq;
    const bool counts = !(D4 < 0.0) && !(t1 < SPPM_EPS && t2 < SPPM_EPS);
    const double t = (t1 > SPPM_EPS) ? t1 : t2;
    if (counts && (t < best || (t == best && i < id))) { best = t; id = i; }
}

__device__ __forceinline__ bool intersect_scene_grid(const DevScene& sc, d3 org, d3 dir, Hit& h)
{
    double best = SPPM_INF;
    int id = -1;
    for (int k = 0; k < sc.n_big; k++) sphere_test(sc.sp, sc.big[k], org, dir, best, id);
    // clip the ray against the grid box
    const double o[3] = {org.x, org.y, org.z}, d[3] = {dir.x, dir.y, dir.z};
    const int dims[3] = {sc.gx, sc.gy, sc.gz};
    double t0 = 0.0, t1 = best;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double lo = sc.g_lo[a], hi = sc.g_lo[a] + dims[a] * sc.g_cell;
        if (d[a] == 0.0) { if (o[a] < lo || o[a] > hi) miss = true; }
        else {
            const double inv = 1.0 / d[a];
            double ta = (lo - o[a]) * inv, tb = (hi - o[a]) * inv;
            if (ta > tb) { const double tt = ta; ta = tb; tb = tt; }
            t0 = fmax(t0, ta); t1 = fmin(t1, tb);
        }
    }
    if (!miss && t0 <= t1) {
        const double ts = t0 + 1e-9;  // just inside
        int cx, cy, cz, sx, sy, sz;
        double nx, ny, nz, dx, dy, dz;
#define SPPM_DDA_AXIS(A, ci, st, tn, td)                                                                              \
        {                                                                                                             \
            int k_ = (int)floor((o[A] + d[A] * ts - sc.g_lo[A]) * sc.g_inv);                                          \
            k_ = k_ < 0 ? 0 : (k_ >= dims[A] ? dims[A] - 1 : k_);                                                     \
            ci = k_;                                                                                                  \
            if (d[A] > 0.0) { st = 1; td = sc.g_cell / d[A]; tn = (sc.g_lo[A] + (k_ + 1) * sc.g_cell - o[A]) / d[A]; } \
            else if (d[A] < 0.0) { st = -1; td = -sc.g_cell / d[A]; tn = (sc.g_lo[A] + k_ * sc.g_cell - o[A]) / d[A]; } \
            else { st = 0; td = SPPM_INF; tn = SPPM_INF; }                                                            \
        }
        SPPM_DDA_AXIS(0, cx, sx, nx, dx)
        SPPM_DDA_AXIS(1, cy, sy, ny, dy)
        SPPM_DDA_AXIS(2, cz, sz, nz, dz)
#undef SPPM_DDA_AXIS
        for (;;) {
            const uint32_t cell = (uint32_t)((cz * sc.gy + cy) * sc.gx + cx);
            if ((sc.cell_mask[cell >> 5] >> (cell & 31u)) & 1u) {  // most cells are empty: no trip to memory for those
                const uint32_t e = sc.cell_start[cell + 1];
                for (uint32_t k = sc.cell_start[cell]; k < e; k++) sphere_test_geom(sc.cell_geom + 4 * (size_t)k, sc.cell_items[k], org, dir, best, id);
            }
            // step into the next cell along the ray; stop once the closest hit lies inside what has been walked
            if (nx <= ny && nx <= nz) { if (best < nx) break; cx += sx; if (cx < 0 || cx >= sc.gx) break; nx += dx; }
            else if (ny <= nz) { if (best < ny) break; cy += sy; if (cy < 0 || cy >= sc.gy) break; ny += dy; }
            else { if (best < nz) break; cz += sz; if (cz < 0 || cz >= sc.gz) break; nz += dz; }
        }
    }
    h.distance = best;
    h.object_id = id;
    if (id < 0) return false;
    h.position = org + dir * best;
    h.normal = normalize(h.position - ld3(sc.sp[id].position));
    return true;
}

// GRID 0: the brute-force loop (small scene, staged in LDS); 1: the grid; 2: whichever the scene has.  The
// kernels that set the frame time are compiled for 0 and 1 separately, so that the small-scene path carries
// none of the grid walk's registers.
#define SCENE_ANY 2
template <int GRID = SCENE_ANY>
__device__ __forceinline__ bool intersect_scene(const DevScene& sc, d3 org, d3 dir, Hit& h)
{
    if constexpr (GRID == 0) return intersect_scene(sc.sp, sc.ns, org, dir, h);
    else if constexpr (GRID == 1) return intersect_scene_grid(sc, org, dir, h);
    else {
        if (sc.n_cells > 0) return intersect_scene_grid(sc, org, dir, h);
        return intersect_scene(sc.sp, sc.ns, org, dir, h);
    }
}

// Copies the sphere table (88-byte rows = 11 doubles) into dynamic LDS and points `sc` at the copy (small
// scenes; a gridded scene stays where it is).  Every thread of the block must reach it; launch with
// scene_lds() bytes of dynamic LDS.
#define SPPM_MAX_SPHERES 512          // brute force from LDS up to here, the grid above it
#define SPPM_MAX_SPHERES_GRID (1 << 20)
#define STAGE_SCENE(sc)                                                                                              \
    extern __shared__ __align__(16) double s_scene_dyn[];                                                             \
    if ((sc).n_cells == 0) {                                                                                          \
        for (int i_ = threadIdx.x; i_ < (sc).ns * 11; i_ += blockDim.x) s_scene_dyn[i_] = ((const double*)(sc).sp)[i_]; \
        __syncthreads();                                                                                              \
        (sc).sp = (const sppm_sphere*)s_scene_dyn;                                                                    \
    }
// The same in a kernel compiled for one kind of scene: with GRID == 0 the pointer is known to be an LDS
// address from here on (LDS reads instead of flat ones in the sphere loop); with GRID == 1 the grid's occupancy
// bits are staged instead (scene_lds() bytes either way).
#define STAGE_SCENE_T(sc, GRID)                                                                                      \
    extern __shared__ __align__(16) double s_scene_dyn[];                                                             \
    if constexpr ((GRID) == 0) {                                                                                      \
        for (int i_ = threadIdx.x; i_ < (sc).ns * 11; i_ += blockDim.x) s_scene_dyn[i_] = ((const double*)(sc).sp)[i_]; \
        __syncthreads();                                                                                              \
        (sc).sp = (const sppm_sphere*)s_scene_dyn;                                                                    \
    } else {  /* the grid's occupancy bits */                                                                         \
        uint32_t* m_ = (uint32_t*)s_scene_dyn;                                                                        \
        for (int i_ = threadIdx.x; i_ < ((sc).n_cells + 31) / 32; i_ += blockDim.x) m_[i_] = (sc).cell_mask[i_];      \
        __syncthreads();                                                                                              \
        (sc).cell_mask = m_;                                                                                          \
    }

// edupt::PinholeCamera, src/cameras.h:34-64
struct Camera {
    d3 pos, sx, sy, center; double w, h, ss;
    // thin lens (DoFCamera, src/cameras.h:66-139): model 1
    int model; d3 dir, ux, uy; double lens_radius, focus_dist;
};
__device__ __host__ inline Camera make_camera_fields(const double* pos, const double* sx, const double* sy,
                                                     const double* center, double w, double h, double ss)
{
    Camera c;
    c.pos.x = pos[0]; c.pos.y = pos[1]; c.pos.z = pos[2];
    c.sx.x = sx[0]; c.sx.y = sx[1]; c.sx.z = sx[2];
    c.sy.x = sy[0]; c.sy.y = sy[1]; c.sy.z = sy[2];
    c.center.x = center[0]; c.center.y = center[1]; c.center.z = center[2];
    c.w = w; c.h = h; c.ss = ss;
    c.model = 0; c.dir = c.pos; c.ux = c.pos; c.uy = c.pos; c.lens_radius = 0.0; c.focus_dist = 0.0;
    return c;
}
__device__ __forceinline__ void camera_ray(const Camera& c, double x, double y, int sx, int sy, d3& org, d3& dir)
{
    const double rate = (1.0 / c.ss);
    const double r1 = sx * rate + rate / 2.0;
    const double r2 = sy * rate + rate / 2.0;
    const d3 sp = c.center + c.sx * ((r1 + x) / c.w - 0.5) + c.sy * ((r2 + y) / c.h - 0.5);
    org = c.pos;
    dir = normalize(sp - c.pos);
}

// The lens draws have a generator of their own (StochasticPpm::engine_, src/sppm2.h:70-71): mode 1 = one
// XorShift stream per (pass, image index); mode 2 = the reference's sequential stream, two draws per ray at
// a position known in advance; mode 0 = pinhole, no draws.
struct LensRng {
    XorShift xs; const double* u; long long pos; int mode;
    // std::uniform_real_distribution<double>(-1, 1): canonical * (b - a) + a
    __device__ __forceinline__ double next_pm1()
    {
        double v = 0.5;
        if (mode == 1) v = xs.next01();
        else if (mode == 2) v = u[pos++];
        return v * 2.0 + -1.0;
    }
};
// PointSamplerDisk::sample, src/sampler/point_sampler_disk.h:28-71 (concentric map; the reference takes cosf/sinf
// in FLOAT: here the detmath double value rounded to float)
__device__ __forceinline__ void disk_sample(LensRng& lens, double& lx, double& ly)
{
    double r, theta;
    const double sx = lens.next_pm1();
    const double sy = lens.next_pm1();
    if (sx == 0.0 && sy == 0.0) { lx = 0.0; ly = 0.0; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; if (sy > 0.0) theta = sy / r; else theta = 8.0 + sy / r; }
        else { r = sy; theta = 2.0 - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0 - sy / r; }
        else { r = -sy; theta = 6.0 + sx / r; }
    }
    theta *= SPPM_PI / 4.0;
    double sn, cs;
    dm_sincos((double)(float)theta, &sn, &cs);
    lx = r * (double)(float)cs; ly = r * (double)(float)sn;
}
// DoFCamera::get_ray, src/cameras.h:123-139
__device__ __forceinline__ void camera_ray_lens(const Camera& c, double x, double y, int sx, int sy, LensRng& lens, d3& org, d3& dir)
{
    const double rate = (1.0 / c.ss);
    const double r1 = sx * rate + rate / 2.0;
    const double r2 = sy * rate + rate / 2.0;
    const d3 sp = c.center + c.sx * ((r1 + x) / c.w - 0.5) + c.sy * ((r2 + y) / c.h - 0.5);
    double lx, ly;
    disk_sample(lens, lx, ly);
    lx = lx * 1.0 * c.lens_radius; ly = ly * 1.0 * c.lens_radius;
    const d3 d = normalize(sp - c.pos);
    const double ft = fabs(c.focus_dist / dot(d, c.dir));
    const d3 focus_point = c.pos + d * ft;
    org = sp + c.ux * lx + c.uy * ly;
    dir = normalize(focus_point - org);
}

__device__ __forceinline__ d3 reflect_dir(d3 d, d3 n) { return d - n * 2.0 * dot(n, d); }

// the (u, v) of an orthonormal basis around w: src/sppm2.h:334-342, :435-443
__device__ __forceinline__ void basis(d3 w, d3& u, d3& v)
{
    if (fabs(w.x) > 0.1) u = normalize(cross(mk(0.0, 1.0, 0.0), w));
    else u = normalize(cross(mk(1.0, 0.0, 0.0), w));
    v = cross(w, u);
}
// cosine_sampling, src/sppm2.h:138-144
template <class RNG>
__device__ __forceinline__ d3 cosine_sampling(d3 w, d3 u, d3 v, RNG& rng)
{
    const double u1 = 2 * SPPM_PI * rng.next01();
    const double u2 = rng.next01();
    const double u3 = sqrt(u2);
    double sn, cs;
    dm_sincos(u1, &sn, &cs);
    return normalize((u * cs * u3 + v * sn * u3 + w * sqrt(1.0 - u2)));
}

// next_event_est, src/radiance.h:29-70
template <class RNG, int GRID = SCENE_ANY>
__device__ __forceinline__ d3 next_event_est(const DevScene& sc, const Hit& is, d3 on,
                                             RNG& rng, uint32_t& rays)
{
    if (is.object_id == 0) return mk(0, 0, 0);
    const sppm_sphere* sp = sc.sp;
    const double r1 = 2 * SPPM_PI * rng.next01();
    const double r2 = 1.0 - 2.0 * rng.next01();
    const double r3 = sqrt(1.0 - r2 * r2);
    double sn, cs;
    dm_sincos(r1, &sn, &cs);
    const d3 lc = ld3(sp[0].position);
    const d3 lp = lc + normalize(mk(r3 * cs, r3 * sn, r2)) * (sp[0].radius + SPPM_EPS);
    d3 ldir = lp - is.position;
    const double dist2 = dot(ldir, ldir);
    ldir = normalize(ldir);
    const d3 ln = normalize(lp - lc);
    const double dot0 = dot(on, ldir);
    const double dot1 = dot(ln, ldir);
    const double radius2 = sp[0].radius * sp[0].radius;
    Hit sh;
    rays++;
    const bool hit = intersect_scene<GRID>(sc, is.position, ldir, sh);
    if (hit && sh.object_id == 0 && dot1 < 0 && dot0 > 0) {
        const double G = (fabs(dot0) * fabs(dot1)) / dist2;
        const double pdf = 1.0 / (4.0 * SPPM_PI * radius2);
        const d3 fs = ld3(sp[is.object_id].color) / SPPM_PI;
        return had(ld3(sp[0].emission), fs) * G / pdf;
    }
    return mk(0, 0, 0);
}

// Shared glass interface logic of src/sppm2.h:245-293 (eye) and :465-520 (photon): returns false
// on total internal reflection; otherwise the refracted direction and the Fresnel reflectance.
// `eye` selects the eye path's cosine term (dot(t, -on)) against the photon's (dot(t, n)).
__device__ __forceinline__ bool glass(d3 dir, d3 n, d3 on, bool eye, bool& into, double& nnt, d3& tdir, double& Re)
{
    into = dot(n, on) > 0.0;
    const double nc = 1.0, nt = SPPM_IOR;
    nnt = into ? nc / nt : nt / nc;
    const double ddn = dot(dir, on);
    const double cos2t = 1.0 - nnt * nnt * (1.0 - ddn * ddn);
    if (cos2t < 0.0) return false;
    tdir = normalize(dir * nnt - n * (into ? 1.0 : -1.0) * (ddn * nnt + sqrt(cos2t)));
    const double a = nt - nc, b = nt + nc;
    const double R0 = (a * a) / (b * b);
    const double c = 1.0 - (into ? -ddn : (eye ? dot(tdir, on * -1.0) : dot(tdir, n)));
    Re = R0 + (1.0 - R0) * dm_pow5(c);
    return true;
}

// ---- spatial order of the hitpoints (sort keys: k_hp_sortkeys in sppm_gather_tiles.hpp, and the eye pass itself) ----
__device__ __forceinline__ unsigned long long ord_d(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double undo_ord_d(unsigned long long o)
{
    const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ unsigned long long spread3(unsigned long long x)  // 21 bits -> every third bit
{
    x &= 0x1fffffull;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
// Space-filling-curve key, MORTON_BITS bits along the bounding box's longest side (ties keep pixel order: the sort is stable);
// invalid hitpoints sort last
typedef uint32_t hp_key_t;
#ifndef MORTON_BITS
#define MORTON_BITS 10
#endif
#ifdef MORTON_INVALID_MAX
#define MORTON_KEY_BITS (3 * MORTON_BITS)
#define MORTON_INVALID (((hp_key_t)1u << (3 * MORTON_BITS)) - 1u)
#else
#define MORTON_KEY_BITS (3 * MORTON_BITS + 1)  // (+1: the invalid hitpoints' key)
#define MORTON_INVALID ((hp_key_t)1u << (3 * MORTON_BITS))
#endif
#define MORTON_MAX ((double)((1 << MORTON_BITS) - 1))
// the key of a valid hitpoint at (x, y, z): bb[0..2] = min, bb[3..5] = max of a bounding box (ordered ints); positions
// outside the box clamp to its faces (any order of the hitpoints is a valid one: the box only has to be about right)
__device__ __forceinline__ hp_key_t hp_order_key(double x, double y, double z, const unsigned long long* __restrict__ bb)
{
    unsigned long long code = 0;
    const double v[3] = {x, y, z};
    unsigned int q[3];
    // one scale for the three axes (the box's longest side): cubic cells, so a run of the curve is as wide as it is long
    // (scaling each axis by its own side made clusters 1.7 times longer along the Cornell box's depth: +3 % kernel time)
    const double ext = fmax(fmax(undo_ord_d(bb[3]) - undo_ord_d(bb[0]), undo_ord_d(bb[4]) - undo_ord_d(bb[1])), undo_ord_d(bb[5]) - undo_ord_d(bb[2]));
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double lo = undo_ord_d(bb[a]);
        double f = ext > 0.0 ? (v[a] - lo) / ext * MORTON_MAX : 0.0;
        f = f < 0.0 ? 0.0 : (f > MORTON_MAX ? MORTON_MAX : f);
        q[a] = (unsigned int)f;
    }
#ifndef HP_ZCURVE
    // Hilbert curve (Skilling's axes-to-transpose form): consecutive keys are neighbours in space, so the 64 hitpoints
    // of a cluster never straddle one of the Z curve's jumps (deposit kernel 2.20 -> 2.05 ms; -DHP_ZCURVE restores the Z curve)
#pragma unroll
    for (unsigned int Q = 1u << (MORTON_BITS - 1); Q > 1; Q >>= 1) {
        const unsigned int Pm = Q - 1;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (q[a] & Q) q[0] ^= Pm;
            else { const unsigned int t = (q[0] ^ q[a]) & Pm; q[0] ^= t; q[a] ^= t; }
        }
    }
    q[1] ^= q[0]; q[2] ^= q[1];
    unsigned int t = 0;
#pragma unroll
    for (unsigned int Q = 1u << (MORTON_BITS - 1); Q > 1; Q >>= 1)
        if (q[2] & Q) t ^= Q - 1;
    q[0] ^= t; q[1] ^= t; q[2] ^= t;
    code = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
#else
#pragma unroll
    for (int a = 0; a < 3; a++) code |= spread3(q[a]) << a;
#endif
    return (hp_key_t)code;
}
