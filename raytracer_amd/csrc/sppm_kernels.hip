// sppm_kernels.hip -- the translation unit of libsppm_hip.so: the gfx950 kernels (included below, one header per
// stage) and the host side / C ABI (include/sppm_hip.h) of the SPPM path.
//
// Data layout in HBM (P = pixels = hitpoint slots, R = photon-hit records of one batch):
//   hitpoints, SoA by component, indexed by image index:
//     hp_pos[3][P] hp_nrm[3][P] hp_w[3][P] hp_flux[3][P] ld[3][P]  (f64)
//     hp_r2[P] hp_dist[P] (f64)  hp_n[P] hp_obj[P] hp_valid[P] (i32)
//     rule 1 only: hp_phi[3][P] hp_tau[3][P] hp_nd[P] (f64) hp_madd[P] (i32)
//   photon-hit records, SoA: key[R] (u64: photon<<20 | depth<<1 | orient), f[9][Rcap] (position,
//     geometric normal, flux).  Traced into the *unsorted* arrays by atomic append, then
//     counting-sorted by hash-grid bucket into the *sorted* arrays each batch.
//
// Why records are gridded and hitpoints gather (instead of hitpoints gridded and photons
// scattering with atomics): the reference's update (src/sppm2.h:391-421) is applied per photon,
// in photon order, and each accepted photon shrinks the hitpoint's radius, which decides whether
// later photons are accepted.  A hitpoint's final state is a sequential function of the
// time-ordered photons near it -- but photon PATHS do not depend on hitpoints.  So photons are
// traced in parallel, and each hitpoint replays "its" photons in key (= photon, bounce) order in
// registers: exact reference semantics, no atomics on hitpoint data, deterministic.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <chrono>
#include <type_traits>
#include <vector>

#include "sppm_device.hpp"

// ================================ kernels ==========================================================

#include "sppm_eye.hpp"
#include "sppm_trace.hpp"
#include "sppm_records.hpp"
#include "sppm_gather_simple.hpp"
#include "sppm_cap.hpp"
#include "sppm_image.hpp"
#include "sppm_exchange.hpp"
#include "sppm_gather_tiles.hpp"
#include "sppm_deposit.hpp"
#include "sppm_deposit_wg.hpp"

// ================================ host side / C ABI =================================================

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(SPPM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#ifndef NBUF
#define NBUF 4  // record buffers in the ring: the one being gathered and up to NBUF - 1 traced ahead
#endif
// Tracer streams.  The device gives a process FOUR hardware queues (GPU_MAX_HW_QUEUES); streams beyond that share
// one, and a kernel waits behind whatever runs on the stream it shares a queue with.  With a stream per record buffer
// (round 1: 3 + eye + context = 5) the context's stream shared its queue with one of the tracer streams, and every
// NBUF-th pass its kernels sat behind a 4-10 ms tracer launch (profiles/r02_pass_gaps_before.txt: one gap in three
// of 1.5-7 ms).  So: context, eye, and TWO tracer streams that the buffers of the ring take in turn (NBUF is even).
#define NTS 2
#define TSTREAM(c, b) ((c)->tstream[(b) % NTS])
struct sppm_ctx {
    sppm_config cfg;
    int device;
    hipStream_t stream;
    int P;     // hitpoint slots: one per pixel (sppm2.h); the length of the hitpoint list (classic PPM)
    int npix;  // width * height
    int algo;  // cfg.algorithm
    // classic PPM: hitpoints per pixel / first slot per pixel, by place in the render order; made by sppm_create
    uint32_t* d_hcnt; long long* d_hoff; std::vector<long long> h_hoff;
    int tiles_x, tiles_y, n_tiles, tile_first, tile_stride, n_owned;
    int gather_mode;          // 1 = tile gather (default), 0 = simple per-hitpoint gather (cross-check)
    int64_t table_cap;        // entries of the (slab, cell) table
    double cell_factor;
    int64_t slab_B;           // photons per slab, adapted from the previous batch
    int64_t rec_first, rec_end;  // photon index range of the records in the buffer
    uint64_t photons_applied; // photons gathered so far (sizes the next batch)
    TileStats* d_stats;
    // spatial order of this rank's hitpoints (tile gather): own pixel list, Morton keys, sorted order
    int32_t *d_own_pix, *d_order; int n_own_pix;
    uint8_t* d_own_mask;   // [P]
    int own_pending;       // spatial ownership: the pixel list is cut out of the first eye pass's Morton order
    hp_key_t *d_hp_keys, *d_hp_keys_out; unsigned long long* d_hp_bb;
    void* d_sort_tmp; size_t sort_tmp_bytes;
    uint32_t *d_est, *d_items; unsigned long long* d_items_raw; unsigned int* d_item_q; uint8_t* d_n_pieces; unsigned long long* d_est_sum; uint8_t* d_ext_bin; int est_valid;  // d_item_q: [0] items [1] queue head [2..33] per cost class
    double split_factor, est_per_photon; int64_t est_range;
    int64_t hist_est_range, stat_est_range[2];  // photons of the batch whose estimates the statistics (and each pinned slot) hold
    int64_t batch_min; double batch_growth;  // per-photon rule: first batch and growth of the batches with the photons applied (batch_size)
    double fx_scale;          // per-pass rule: fixed point of the deposits (sppm_deposit.hpp), a power of two
    double cell_factor_dep;   // per-pass rule: cell size of the record grid as a share of the median tile extent
    int64_t dep_chunk;        // per-pass rule: records per work item
    double dep_light, dep_heavy;  // per-pass rule: pairs per photon at which the deposit kernel runs on 576 / 768 workgroups (deposit_tiles)
    int order_valid;
    sppm_sphere* d_spheres; int ns;
    DevScene dscene;   // what the kernels get: the table, and for large scenes the grid over the small spheres
    Camera cam;
    double flux_scale;
    std::vector<void*> allocs;
    HpArrays hp;       // the current view (= view[cur_view])
    // What an eye pass writes (hitpoint geometry, valid flags, direct term, tile order, bounding box) exists
    // twice: the eye pass of pass i+1 fills the other view on its own stream while the gather of pass i still
    // reads this one.  Radii, counts and flux are shared (an eye pass only initialises them for new pixels).
    HpArrays view[2]; uint8_t* d_near_v[2]; uint8_t* d_near; int cur_view;
    // The tile order (and the bounding box made with it) of an eye pass, in a ring of three buffers (in use, made
    // last, being made); view_chain[v]: the buffer of the eye pass that wrote view v.  (Using the order of the pass
    // BEFORE, so that the sort chain need not finish between two gathers, was tried: any permutation is a valid order,
    // but the ~5 % of the pixels that look through the glass sphere land somewhere else every pass, their tiles of a
    // stale order are scattered over the scene, and the gather took four times as long.)
    int32_t* d_order_r[3]; unsigned long long* d_hp_bb_r[3]; hipEvent_t ev_order[3];
    int chain_n, view_chain[2];
    hipStream_t eye_stream; hipEvent_t ev_eye_done, ev_eye_go;
    struct { int valid, pass; } pre_eye;  // an eye pass computed ahead of its call
    int fuse_order;  // the eye pass makes the hitpoints' bounding box and sort keys itself where it can (eye_launch)
    int last_eye_pass;
    unsigned long long *d_rows_n, *h_rows_n; hipEvent_t ev_rows; int rows_pending; int64_t rows_cap;  // sppm_look_ahead_rows
    // Record exchange by destination (sppm_look_ahead_rows_to): with spatial ownership every rank knows every rank's
    // hitpoints (the eye pass is replicated), so it keeps, per view, the occupancy map of EVERY rank and sends a record
    // only to the ranks whose map holds its cell.  d_all_pix: the first eye pass's whole Morton order (rank r owns
    // positions [r P/N, (r+1) P/N)).
    int all_maps; int32_t* d_all_pix; uint8_t* d_near_all_v[2]; unsigned long long* d_bb_all_v[2]; int view_pass[2]; hipEvent_t ev_view[2];
    unsigned long long *d_rows_n_to, *h_rows_n_to;  // [tile_stride]
    RecArrays rec_u, rec_s;  // unsorted (traced / received), sorted by bucket
    unsigned long long* d_rec_count;
    unsigned long long* d_counters;  // 8: [0] rays [3] candidates [4] updates [5] selected [6] rounds [7] live tile launches
    unsigned long long* d_r2max;
    unsigned long long* d_eye_next;  // block counter of k_eye_pass_refill
    unsigned long long* d_qmax;   // check_gather_max: the largest number of candidates any query held
    unsigned long long* d_trace_next;
    // A ring of NBUF record buffers, each with a tracer stream of its own: the gather consumes the current one while
    // the next two are traced and binned.  A tracer launch lasts as long as its longest photon path (a few thousand
    // bounces inside the glass ball: 3-6 ms whatever the GPU could do meanwhile), so consecutive launches must be
    // able to overlap -- on one stream they alone would set the pass time.
    // rec_u / d_rec_count / d_trace_next above always alias buffer `cur`.
    RecArrays rec_buf[NBUF]; unsigned long long* d_rec_count_buf[NBUF]; unsigned long long* d_trace_next_buf[NBUF];
    int cur;
    hipStream_t tstream[NTS];
    hipEvent_t ev_traced[NBUF], ev_free[NBUF];
    struct Pre { int valid, buf, pass; int64_t first, count; } pre;  // a batch traced ahead through sppm_look_ahead
    std::deque<Pre> ahead;  // sppm_photon_pass: the batches traced ahead, next first, in ring buffers cur+1, cur+2, ...
    int ahead_depth;  // batches traced ahead of their photon_pass call: the next one, and (steady state) the one after
    int prefetch;
    int32_t* d_err;                  // 4
    uint32_t *d_cell_cnt, *d_cell_start, *d_cursor, *d_block_sum, *d_total, *d_rec_bucket;  // alias bins[cur]
    // One set of binning tables and sorted records per record buffer: a batch is binned on the tracer's
    // stream right after it was traced, while the gather of the batch before it still runs.
    struct BinSet { RecArrays rec_s; uint32_t *rec_bucket, *cell_cnt, *cell_start, *cursor, *block_sum, *total, *cell_tot; TileGrid g; int binned; } bins[NBUF];
    // Tile statistics travel to the host one batch late (pinned slots, no stream sync in steady state):
    // they only size the grid and balance the load, never change a result.
    typedef StatSnapshot StatSlot;
    StatSlot* h_stat; StatSlot* d_stat_map;  // pinned host memory and the address the device writes it through
    hipEvent_t ev_stat[2]; int stat_w; long stat_n;
    struct { double lo[3], cell, cluster_d; int64_t dims[3], ncells; int valid; } gs;  // spatial part of the grid, from the last statistics read
    unsigned long long snap_prev[24]; int snap_prev_valid; long gather_j; int64_t hist_S[4], hist_B[4], hist_range[4]; double pairs_per_photon;  // pairs_per_photon: what the deposit kernel examined two batches ago (0: not known)
    uint32_t n_buckets;
    int64_t batch_photons;
    // K cap of the reference's search (sppm_cap.hpp): the reference's hitpoint vector in the order its tree leaves it
    // (host), the same on the device, one heap per search lane, the table of dropped pairs the gathers consult
    std::vector<int32_t> kd_order; std::vector<uint8_t> kd_listed; int kd_pass, kd_fallback;
    double4* d_kd; int64_t kd_cap;
    // the K cap's filter (sppm_cap.hpp): hitpoint counts on a coarse grid, the photon hits whose bound exceeds K
    uint32_t* d_cap_cnt; int64_t cap_cnt_cap; CapGrid cap_grid; uint32_t* d_cap_surv; int64_t cap_surv_cap; unsigned long long* d_cap_nsurv;
    uint64_t cap_filtered;  // photon hits that passed the filter so far
    double2* d_heap; int64_t heap_cap; uint32_t *d_cap_cell, *d_cap_surv_s, *d_cap_cell_s; int64_t cap_cell_cap; void* d_cap_sort_tmp; size_t cap_sort_bytes;
    unsigned long long* d_veto_tab; int64_t veto_entries; unsigned long long *d_veto_n, *d_cap_stat;
    Veto veto_dev;  // what the gathers get: n == nullptr while the cap cannot act
    uint64_t cap_queries, cap_capped, cap_nodes, cap_vetoes, cap_max_fill;
    // accept tap
    uint64_t* d_tap_key; int32_t* d_tap_pix; unsigned long long* d_tap_n; int64_t tap_cap;
    // timing of the gather kernel
    int timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev, ev_trace;
    uint64_t gather_ns, gather_launches, gather_batches, trace_ns, trace_launches;
    uint64_t photons_traced, records_total;
    // reference-stream mode (cfg.sampler == 1): the reference's own sequential mt19937 streams
    int sampler;
    std::mt19937 eye_eng;               // one engine for all eye passes, src/sppm2.h:621
    std::mt19937 lens_eng;              // the lens draws' engine, src/sppm2.h:70-71
    double* d_u_lens; int64_t u_lens_cap;
    std::vector<double> eye_pending;    // drawn from eye_eng, not yet consumed by a pass
    double *d_u_eye, *d_u_ph; int64_t u_eye_cap, u_ph_cap, u_eye_len;
    uint32_t *d_eye_cnt, *d_ph_start, *d_ph_consumed; int64_t ph_start_cap, ph_consumed_cap;
    unsigned long long *d_eye_off, *d_stream_total;
    int64_t stream_len;                 // uploaded length of the photon stream of stream_pass
    int stream_pass;
};

// grow-only device buffer outside the allocs list (freed in sppm_destroy)
template <typename T>
static int grow(sppm_ctx* c, T** p, int64_t* cap, int64_t need)
{
    if (*cap >= need) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int b = 0; b < NTS; b++) if (c->tstream[b]) HIPCHK(hipStreamSynchronize(c->tstream[b]));
    if (*p) hipFree(*p);
    *p = nullptr; *cap = 0;
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, (size_t)need * sizeof(T)));
    *p = (T*)q; *cap = need;
    return 0;
}

template <typename T>
static int dev_alloc(sppm_ctx* c, T** p, size_t n, bool zero = true)
{
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    c->allocs.push_back(q);
    if (zero) HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), c->stream));
    *p = (T*)q;
    return 0;
}

// The sort of the hitpoints' Morton keys runs beside the gather, whose persistent wavefronts leave one wavefront slot of
// 128 registers per SIMD.  rocPRIM's tuned Onesweep kernel for these keys and values uses workgroups of 512 threads,
// two wavefronts per SIMD, which found no room until the gather's tail (one pass of it took 2.4 ms instead of 40 us).
// The same algorithm (8 bits per pass, match-based ranking) with workgroups of 256 threads fits.
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 32>, rocprim::kernel_config<256, 12>, 8,
                                                                       rocprim::block_radix_rank_algorithm::match>> HpSortConfig;

static void host_normalize(const double* v, double* o)
{
    const double inv = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}
static void host_cross(const double* a, const double* b, double* o)
{
    o[0] = (a[1] * b[2]) - (a[2] * b[1]); o[1] = (a[2] * b[0]) - (a[0] * b[2]); o[2] = (a[0] * b[1]) - (a[1] * b[0]);
}

// A context lives on the HIP device that was current in sppm_create (include/sppm_hip.h): its buffers, streams and
// events belong to that device, so a call made with another device current is refused rather than left to fault.
static int ctx_device(const sppm_ctx* c, const char* who)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return fail(SPPM_ERR_HIP, std::string(who) + ": hipGetDevice failed");
    if (d != c->device)
        return fail(SPPM_ERR_STATE, std::string(who) + ": the context belongs to HIP device " + std::to_string(c->device) + " but device " + std::to_string(d) + " is current");
    return 0;
}
#define CTX_DEVICE(c, who) do { int rc_dev_ = ctx_device((c), who); if (rc_dev_) return rc_dev_; } while (0)

extern "C" const char* sppm_last_error(void) { return g_err.c_str(); }

extern "C" int sppm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int64_t env_i64(const char* name, int64_t dflt)
{
    const char* s = getenv(name);
    return (s && *s) ? atoll(s) : dflt;
}


// The grid of intersect_scene_grid (sppm_device.hpp) over a scene of many spheres, built once on the host.
// Big spheres (radius > 16 x the lower-quartile radius: walls, lights) are tested for every ray; every other
// sphere is listed in each cell its bounding box, grown by a margin far above the rounding of the cell walk,
// overlaps.  Cubic cells sized for about one small sphere per two cells.
static int build_scene_grid(sppm_ctx* c, const sppm_sphere* sp, int n)
{
    std::vector<double> radii((size_t)n);
    for (int i = 0; i < n; i++) radii[(size_t)i] = sp[i].radius;
    std::vector<double> sorted = radii;
    std::sort(sorted.begin(), sorted.end());
    const double r_q1 = sorted[(size_t)n / 4];
    std::vector<int32_t> big, small;
    for (int i = 0; i < n; i++) (radii[(size_t)i] > 16.0 * r_q1 ? big : small).push_back(i);
    if (small.empty()) { small.swap(big); }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int32_t i : small)
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], sp[i].position[a] - sp[i].radius); hi[a] = std::max(hi[a], sp[i].position[a] + sp[i].radius); }
    const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    const double margin = 1e-7 * diag + 1e-9;
    for (int a = 0; a < 3; a++) { lo[a] -= 2.0 * margin; hi[a] += 2.0 * margin; }
    const double vol = std::max(1e-30, (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]));
    double cell = std::max(std::cbrt(vol / (2.0 * (double)small.size())), 2.0 * sorted[small.size() / 2 < (size_t)n ? small.size() / 2 : 0]);
    if (!(cell > 0.0)) cell = std::max(diag, 1.0);
    { const char* e = getenv("SPPM_GRID_CELL_FACTOR"); if (e && *e && atof(e) > 0.0) cell *= atof(e); }
    std::vector<uint32_t> start;
    std::vector<int32_t> items;
    int dims[3];
    for (;;) {
        int64_t ncells = 1;
        for (int a = 0; a < 3; a++) { dims[a] = (int)std::max<int64_t>(1, std::min<int64_t>(512, (int64_t)std::ceil((hi[a] - lo[a]) / cell))); ncells *= dims[a]; }
        bool too_fine = ncells > (1ll << 18);  // the occupancy bits of the cells live in LDS (32 KB)
        for (int a = 0; a < 3; a++) if (dims[a] * cell < (hi[a] - lo[a])) too_fine = true;  // clamped to 512: the box must stay covered
        if (!too_fine) {
            const double inv = 1.0 / cell;
            auto range = [&](int32_t i, int a, int& c0, int& c1) {
                c0 = (int)std::floor((sp[i].position[a] - sp[i].radius - margin - lo[a]) * inv);
                c1 = (int)std::floor((sp[i].position[a] + sp[i].radius + margin - lo[a]) * inv);
                c0 = std::max(0, std::min(dims[a] - 1, c0)); c1 = std::max(0, std::min(dims[a] - 1, c1));
            };
            std::vector<uint32_t> cnt((size_t)ncells + 1, 0);
            int64_t total = 0;
            for (int32_t i : small) {
                int x0, x1, y0, y1, z0, z1;
                range(i, 0, x0, x1); range(i, 1, y0, y1); range(i, 2, z0, z1);
                total += (int64_t)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1);
            }
            if (total <= 64 * (int64_t)small.size() + (1 << 20)) {
                for (int32_t i : small) {
                    int x0, x1, y0, y1, z0, z1;
                    range(i, 0, x0, x1); range(i, 1, y0, y1); range(i, 2, z0, z1);
                    for (int z = z0; z <= z1; z++) for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++) cnt[(size_t)((z * dims[1] + y) * dims[0] + x) + 1]++;
                }
                for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
                start = cnt;
                items.assign((size_t)total, 0);
                std::vector<uint32_t> cur(start.begin(), start.end() - 1);
                for (int32_t i : small) {  // ascending ids: every cell's list comes out ascending
                    int x0, x1, y0, y1, z0, z1;
                    range(i, 0, x0, x1); range(i, 1, y0, y1); range(i, 2, z0, z1);
                    for (int z = z0; z <= z1; z++) for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++) items[cur[(size_t)((z * dims[1] + y) * dims[0] + x)]++] = i;
                }
                break;
            }
        }
        cell *= 1.5;
    }
    const size_t ncells_final = (size_t)dims[0] * dims[1] * dims[2];
    std::vector<uint32_t> mask((ncells_final + 31) / 32, 0u);
    for (size_t cell = 0; cell < ncells_final; cell++) if (start[cell + 1] > start[cell]) mask[cell >> 5] |= 1u << (cell & 31);
    std::vector<double> geom(items.size() * 4);
    for (size_t k = 0; k < items.size(); k++) {
        const sppm_sphere& q = sp[items[k]];
        geom[4 * k] = q.position[0]; geom[4 * k + 1] = q.position[1]; geom[4 * k + 2] = q.position[2]; geom[4 * k + 3] = q.radius;
    }
    int32_t* d_big = nullptr; uint32_t* d_start = nullptr; int32_t* d_items = nullptr; uint32_t* d_mask = nullptr; double* d_geom = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_big, big.size(), false)) || (rc = dev_alloc(c, &d_start, start.size(), false)) || (rc = dev_alloc(c, &d_items, items.size(), false)) ||
        (rc = dev_alloc(c, &d_mask, mask.size(), false)) || (rc = dev_alloc(c, &d_geom, geom.size(), false))) return rc;
    HIPCHK(hipMemcpy(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    if (!geom.empty()) HIPCHK(hipMemcpy(d_geom, geom.data(), geom.size() * 8, hipMemcpyHostToDevice));
    if (!big.empty()) HIPCHK(hipMemcpy(d_big, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_start, start.data(), start.size() * 4, hipMemcpyHostToDevice));
    if (!items.empty()) HIPCHK(hipMemcpy(d_items, items.data(), items.size() * 4, hipMemcpyHostToDevice));
    DevScene& g = c->dscene;
    g.n_cells = dims[0] * dims[1] * dims[2]; g.n_big = (int)big.size(); g.gx = dims[0]; g.gy = dims[1]; g.gz = dims[2];
    for (int a = 0; a < 3; a++) g.g_lo[a] = lo[a];
    g.g_cell = cell; g.g_inv = 1.0 / cell;
    g.big = d_big; g.cell_start = d_start; g.cell_items = d_items; g.cell_mask = d_mask; g.cell_geom = d_geom;
    if (getenv("SPPM_DEBUG"))
        fprintf(stderr, "[sppm] scene grid: %d spheres, %zu big, %dx%dx%d cells of %.3f, %zu list entries\n", n, big.size(), dims[0], dims[1], dims[2], cell, items.size());
    return 0;
}

static void set_cur(sppm_ctx* c, int b);
static int ppm_prepare(sppm_ctx* c);

static int static_scratch(unsigned char** out);  // (sppm_intersect_static, at the end of this file)
extern "C" int sppm_create(const sppm_config* cfg, const sppm_sphere* h_spheres, int32_t n_spheres, void* stream, sppm_ctx** out)
{
    if (!cfg || !h_spheres || !out || n_spheres < 1) return fail(SPPM_ERR_INVALID, "sppm_create: null argument or empty scene");
    if (n_spheres > SPPM_MAX_SPHERES_GRID) return fail(SPPM_ERR_INVALID, "sppm_create: more than 2^20 spheres");
    if (cfg->width < 1 || cfg->height < 1 || cfg->samples < 1 || cfg->supersamples < 1)
        return fail(SPPM_ERR_INVALID, "sppm_create: width, height, samples, supersamples must be >= 1");
    // the kernels index the SoA arrays with 32-bit integers (component * P + pixel) and name a tile of 64 hitpoints in 20 bits
    if ((int64_t)cfg->width * cfg->height > (1ll << 26)) return fail(SPPM_ERR_INVALID, "sppm_create: more than 2^26 pixels");
    if (!(cfg->gather_radius2 > 0.0) || !(cfg->initial_radius2 > 0.0) || !(cfg->alpha > 0.0))
        return fail(SPPM_ERR_INVALID, "sppm_create: radii and alpha must be positive");
    if (cfg->update_rule != 0 && cfg->update_rule != 1) return fail(SPPM_ERR_INVALID, "sppm_create: update_rule must be 0 or 1");
    if (cfg->ownership != 0 && cfg->ownership != 1) return fail(SPPM_ERR_INVALID, "sppm_create: ownership must be 0 (screen tiles) or 1 (ranges of the hitpoints' Morton order)");
    if (cfg->sampler != 0 && cfg->sampler != 1) return fail(SPPM_ERR_INVALID, "sppm_create: sampler must be 0 (XorShift streams) or 1 (reference mt19937 stream)");
    // the per-pass rule shards PHOTONS over ranks (every rank keeps all hitpoints and the deposits are all-reduced); with a
    // share of the hitpoints per rank the end-of-pass kernels would leave the imported sums of the hitpoints a rank does
    // not own in place, to be exported and counted again in the next pass
    if (cfg->update_rule == 1 && cfg->tile_stride > 1)
        return fail(SPPM_ERR_INVALID, "sppm_create: update_rule 1 keeps all hitpoints on every rank (tile_stride 0 or 1); hitpoint ownership (tile_stride > 1) is for update_rule 0");
    if (cfg->tracer_variant != 0 && cfg->tracer_variant != 1) return fail(SPPM_ERR_INVALID, "sppm_create: tracer_variant must be 0 (src/sppm2.h) or 1 (src/stochastic_ppm.h as shipped)");
    if (cfg->tracer_variant == 1 && (cfg->sampler != 1 || cfg->algorithm != 0)) return fail(SPPM_ERR_INVALID, "sppm_create: tracer_variant 1 reproduces runs of the compiled src/stochastic_ppm.h and needs sampler 1, algorithm 0");
    if (cfg->algorithm != 0 && cfg->algorithm != 1) return fail(SPPM_ERR_INVALID, "sppm_create: algorithm must be 0 (sppm2.h) or 1 (progressive_photonmapping.h)");
    if (cfg->algorithm == 1 && (cfg->update_rule != 0 || (cfg->tile_stride > 1 && cfg->ownership != 1)))
        return fail(SPPM_ERR_INVALID, "sppm_create: classic progressive photon mapping runs with update_rule 0; over several ranks its hitpoint list is shared out by ownership 1 (ranges of the Morton order)");
    if (cfg->camera_model != 0 && cfg->camera_model != 1) return fail(SPPM_ERR_INVALID, "sppm_create: camera_model must be 0 (pinhole) or 1 (thin lens)");
    if (cfg->camera_model == 1 && (cfg->algorithm != 0 || !(cfg->aperture >= 0.0) || !(cfg->focus_dist > 0.0)))
        return fail(SPPM_ERR_INVALID, "sppm_create: the thin-lens camera belongs to algorithm 0 and needs aperture >= 0 and focus_dist > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(SPPM_ERR_HIP, "sppm_create: no HIP device (this library has no CPU path)");
    sppm_ctx* c = new sppm_ctx();
    c->cfg = *cfg;
    c->stream = (hipStream_t)stream;
    c->P = c->npix = cfg->width * cfg->height;
    c->algo = cfg->algorithm; c->d_hcnt = nullptr; c->d_hoff = nullptr;
    c->tiles_x = (cfg->width + 7) / 8; c->tiles_y = (cfg->height + 7) / 8; c->n_tiles = c->tiles_x * c->tiles_y;
    c->tile_stride = cfg->tile_stride > 0 ? cfg->tile_stride : 1;
    c->tile_first = cfg->tile_stride > 0 ? cfg->tile_first : 0;
    if (c->tile_first < 0 || c->tile_first >= c->tile_stride) { delete c; return fail(SPPM_ERR_INVALID, "sppm_create: need 0 <= tile_first < tile_stride"); }
    c->gather_mode = (int)env_i64("SPPM_GATHER_TILES", 1);
    // (slab, cell) entries: a rank of an N-GPU run sees N times the photons per batch over its part of the
    // space, so it needs N times the slabs for the same number of candidates per tile and slab
    c->table_cap = env_i64("SPPM_TABLE_CAP", (1ll << 24) * std::min(8, std::max(1, cfg->tile_stride)));
    if (c->table_cap < 4096) c->table_cap = 4096;
    if (c->table_cap > (1ll << 28)) c->table_cap = 1ll << 28;
    c->cell_factor = 0.35;
    { const char* e = getenv("SPPM_CELL_FACTOR"); if (e && *e) c->cell_factor = atof(e); }
    c->slab_B = env_i64("SPPM_SLAB_PHOTONS", 1024);
    c->cell_factor_dep = 0.15;
    { const char* e = getenv("SPPM_DEP_CELL_FACTOR"); if (e && *e && atof(e) > 0.0) c->cell_factor_dep = atof(e); }
    c->dep_chunk = std::max<int64_t>(64, env_i64("SPPM_DEP_CHUNK", 4096));
    c->dep_light = (double)env_i64("SPPM_DEP_LIGHT", 1700); c->dep_heavy = std::max(c->dep_light + 1.0, (double)env_i64("SPPM_DEP_HEAVY", 2400));
    c->rec_first = c->rec_end = 0;
    c->photons_applied = 0;
    c->ns = n_spheres;
    for (int b = 0; b < NTS; b++) c->tstream[b] = nullptr;
    for (int b = 0; b < NBUF; b++) { c->ev_traced[b] = nullptr; c->ev_free[b] = nullptr; }
    c->eye_stream = nullptr; c->ev_eye_done = c->ev_eye_go = nullptr; for (int k = 0; k < 3; k++) { c->d_order_r[k] = nullptr; c->d_hp_bb_r[k] = nullptr; c->ev_order[k] = nullptr; }
    c->chain_n = 0; c->view_chain[0] = c->view_chain[1] = -1;
    c->tap_cap = 0; c->d_tap_key = nullptr; c->d_tap_pix = nullptr; c->d_tap_n = nullptr;
    c->sampler = cfg->sampler; c->eye_eng.seed(cfg->seed_eye); c->lens_eng.seed(cfg->seed_lens); c->d_u_lens = nullptr; c->u_lens_cap = 0;
    c->d_u_eye = c->d_u_ph = nullptr; c->u_eye_cap = c->u_ph_cap = 0; c->u_eye_len = 0; c->d_eye_cnt = nullptr; c->d_ph_start = c->d_ph_consumed = nullptr;
    c->ph_start_cap = c->ph_consumed_cap = 0; c->d_eye_off = c->d_stream_total = nullptr; c->stream_len = 0; c->stream_pass = -1;
    c->timing = 0; c->gather_ns = c->gather_launches = c->gather_batches = 0; c->trace_ns = c->trace_launches = 0; c->photons_traced = c->records_total = 0;
    int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { sppm_destroy(c); return rc; } } while (0)
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return fail(SPPM_ERR_HIP, "hipGetDevice failed"); }
    TRY(dev_alloc(c, &c->d_spheres, (size_t)n_spheres, false));
    if (hipMemcpyAsync(c->d_spheres, h_spheres, sizeof(sppm_sphere) * (size_t)n_spheres, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        sppm_destroy(c); return fail(SPPM_ERR_HIP, "copying the scene failed");
    }
    memset(&c->dscene, 0, sizeof c->dscene);
    c->dscene.sp = c->d_spheres; c->dscene.ns = n_spheres;
    if (n_spheres > SPPM_MAX_SPHERES || env_i64("SPPM_SCENE_GRID", 0)) TRY(build_scene_grid(c, h_spheres, n_spheres));
    {   // PinholeCamera constructor, src/cameras.h:34-50
        double sxn[3], cx[3], sx[3], sy[3], cy[3], syn[3], center[3];
        const double screen_width = cfg->screen_height * (double)cfg->width / (double)cfg->height;
        host_cross(cfg->camera_dir, cfg->camera_up, cx); host_normalize(cx, sxn);
        for (int k = 0; k < 3; k++) sx[k] = sxn[k] * screen_width;
        host_cross(sx, cfg->camera_dir, cy); host_normalize(cy, syn);
        for (int k = 0; k < 3; k++) sy[k] = syn[k] * cfg->screen_height;
        for (int k = 0; k < 3; k++) center[k] = cfg->camera_position[k] + cfg->camera_dir[k] * cfg->screen_dist;
        c->cam = make_camera_fields(cfg->camera_position, sx, sy, center, (double)cfg->width, (double)cfg->height, (double)cfg->supersamples);
        if (cfg->camera_model == 1) {  // DoFCamera constructor, src/cameras.h:104-121
            double uy[3], uyn[3];
            host_cross(sxn, cfg->camera_dir, uy); host_normalize(uy, uyn);  // unit_y = normalize(cross(unit_x, camera_dir))
            c->cam.model = 1;
            c->cam.dir.x = cfg->camera_dir[0]; c->cam.dir.y = cfg->camera_dir[1]; c->cam.dir.z = cfg->camera_dir[2];
            c->cam.ux.x = sxn[0]; c->cam.ux.y = sxn[1]; c->cam.ux.z = sxn[2];
            c->cam.uy.x = uyn[0]; c->cam.uy.y = uyn[1]; c->cam.uy.z = uyn[2];
            c->cam.lens_radius = cfg->aperture * 0.5; c->cam.focus_dist = cfg->focus_dist;
        }
    }
    {   // std::pow(light.radius * M_PI, 2.0), src/sppm2.h:348
        const double rp = h_spheres[0].radius * SPPM_PI;
        c->flux_scale = rp * rp;
        // fixed point of the per-pass rule's deposits: the largest emitted flux component x scale lies in [2^27, 2^28)
        double m = 0.0;
        for (int k = 0; k < 3; k++) m = std::max(m, h_spheres[0].emission[k] * 4.0 * c->flux_scale);
        int e = 0;
        if (m > 0.0) std::frexp(m, &e);
        c->fx_scale = m > 0.0 ? std::ldexp(1.0, 28 - e) : 1.0;
    }
    if (c->sampler == 1) {
        TRY(dev_alloc(c, &c->d_eye_cnt, (size_t)c->npix)); TRY(dev_alloc(c, &c->d_eye_off, (size_t)c->npix));
        TRY(dev_alloc(c, &c->d_stream_total, 1));
    }
    TRY(dev_alloc(c, &c->d_err, 8));
    if (c->algo == 1) TRY(ppm_prepare(c));  // counts the hitpoint list: c->P becomes its length
    const size_t P = (size_t)c->P;
    c->hp.P = c->P;
    TRY(dev_alloc(c, &c->hp.pos, 3 * P)); TRY(dev_alloc(c, &c->hp.nrm, 3 * P)); TRY(dev_alloc(c, &c->hp.w, 3 * P));
    TRY(dev_alloc(c, &c->hp.flux, 3 * P)); TRY(dev_alloc(c, &c->hp.ld, 3 * P));
    TRY(dev_alloc(c, &c->hp.r2, P)); TRY(dev_alloc(c, &c->hp.dist, P));
    TRY(dev_alloc(c, &c->hp.n, P)); TRY(dev_alloc(c, &c->hp.obj, P)); TRY(dev_alloc(c, &c->hp.valid, P));
    if (cfg->update_rule == 1) {
        TRY(dev_alloc(c, &c->hp.tau, 3 * P)); TRY(dev_alloc(c, &c->hp.nd, P)); TRY(dev_alloc(c, &c->hp.dep, 4 * P));
    } else { c->hp.tau = c->hp.nd = nullptr; c->hp.dep = nullptr; }
    c->view[0] = c->hp; c->view[1] = c->hp; c->cur_view = 0; c->pre_eye.valid = 0; c->last_eye_pass = -1;
    TRY(dev_alloc(c, &c->view[1].pos, 3 * P)); TRY(dev_alloc(c, &c->view[1].nrm, 3 * P)); TRY(dev_alloc(c, &c->view[1].w, 3 * P));
    TRY(dev_alloc(c, &c->view[1].ld, 3 * P)); TRY(dev_alloc(c, &c->view[1].dist, P)); TRY(dev_alloc(c, &c->view[1].obj, P)); TRY(dev_alloc(c, &c->view[1].valid, P));
    // (stream priorities for the look-ahead streams were tried: the run hung until its time limit; not used)
    if (hipStreamCreateWithFlags(&c->eye_stream, hipStreamNonBlocking) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating the eye stream failed"); }
    if (hipEventCreateWithFlags(&c->ev_eye_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_eye_go, hipEventDisableTiming) != hipSuccess) {
        sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating events failed");
    }
    for (int k = 0; k < 3; k++)
        if (hipEventCreateWithFlags(&c->ev_order[k], hipEventDisableTiming) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating events failed"); }
    // record buffers: one batch of photons, ~2.6 records per photon on the Cornell scene; 8x head room
    c->batch_min = std::max<int64_t>(64, env_i64("SPPM_BATCH_MIN", 32768));
    { const char* e = getenv("SPPM_BATCH_GROWTH"); c->batch_growth = e && *e && atof(e) > 0.0 ? atof(e) : 0.5; }
    c->batch_photons = env_i64("SPPM_BATCH_PHOTONS", c->gather_mode ? (1 << 20) : (1 << 18));
    if (c->batch_photons < 1024) c->batch_photons = 1024;
    // a rank of a multi-GPU run keeps the records near its hitpoints out of tile_stride x as many photons
    const int64_t cap = env_i64("SPPM_RECORD_CAP", c->batch_photons * ((c->gather_mode && c->tile_stride == 1) ? 4 : 8));
    c->rec_u.cap = cap;
    for (int b = 0; b < NBUF; b++) {
        c->rec_buf[b].cap = cap;
        TRY(dev_alloc(c, &c->rec_buf[b].key, (size_t)cap, false)); TRY(dev_alloc(c, &c->rec_buf[b].f, 9 * (size_t)cap, false));
        TRY(dev_alloc(c, &c->d_rec_count_buf[b], 1)); TRY(dev_alloc(c, &c->d_trace_next_buf[b], 1));
    }
    c->cur = 0; c->rec_u = c->rec_buf[0]; c->d_rec_count = c->d_rec_count_buf[0]; c->d_trace_next = c->d_trace_next_buf[0];  // bins[] aliases: set_cur below, once they exist
    c->pre.valid = 0;
    c->ahead_depth = (int)std::max<int64_t>(1, std::min<int64_t>(NBUF - 1, env_i64("SPPM_AHEAD", NBUF - 1)));
    c->prefetch = (int)env_i64("SPPM_PREFETCH", 1);
    c->fuse_order = (int)env_i64("SPPM_FUSE_ORDER", 1);
    if (c->sampler == 1) {
        c->prefetch = 0;  // a pass's photon stream offsets exist only once its photon pass has begun
    }
    for (int b = 0; b < NTS; b++)
        if (hipStreamCreateWithFlags(&c->tstream[b], hipStreamNonBlocking) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating the trace streams failed"); }
    for (int b = 0; b < NBUF; b++)
        if (hipEventCreateWithFlags(&c->ev_traced[b], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_free[b], hipEventDisableTiming) != hipSuccess) {
            sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating events failed");
        }

    uint32_t nb = 1024;
    while ((int64_t)nb < cap / 2 && nb < (1u << 24)) nb <<= 1;
    c->n_buckets = nb;
    if (c->gather_mode) nb = (uint32_t)((c->table_cap + SCAN_BLOCK) / SCAN_BLOCK * SCAN_BLOCK);
    for (int b = 0; b < (c->gather_mode ? NBUF : 1); b++) {
        sppm_ctx::BinSet& B = c->bins[b];
        B.rec_s.cap = cap; B.binned = 0;
        TRY(dev_alloc(c, &B.rec_s.key, (size_t)cap, false)); TRY(dev_alloc(c, &B.rec_s.f, REC_ROW * (size_t)cap, false));  // (nine planes, or rows of REC_ROW: k_grid_scatter)
        TRY(dev_alloc(c, &B.rec_bucket, (size_t)cap, false));
        TRY(dev_alloc(c, &B.cell_cnt, (size_t)nb)); TRY(dev_alloc(c, &B.cell_start, (size_t)nb + 1)); TRY(dev_alloc(c, &B.cursor, (size_t)nb));
        TRY(dev_alloc(c, &B.block_sum, (size_t)nb / SCAN_BLOCK + 1)); TRY(dev_alloc(c, &B.total, 1));
        B.cell_tot = nullptr;
        if (c->gather_mode && cfg->update_rule != 1) TRY(dev_alloc(c, &B.cell_tot, (size_t)c->table_cap / 4 + 1));  // the plan's summed-area table (k_tile_plan)
    }
    if (!c->gather_mode) for (int b = 1; b < NBUF; b++) c->bins[b] = c->bins[0];
    c->h_rows_n = nullptr; c->ev_rows = nullptr; c->rows_pending = 0; c->rows_cap = 0;
    TRY(dev_alloc(c, &c->d_rows_n, 1));
    if (hipHostMalloc((void**)&c->h_rows_n, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming) != hipSuccess) {
        sppm_destroy(c); return fail(SPPM_ERR_HIP, "pinned memory / event for the packed rows");
    }
    c->h_stat = nullptr; c->ev_stat[0] = c->ev_stat[1] = nullptr; c->stat_w = 0; c->stat_n = 0; c->gs.valid = 0; c->snap_prev_valid = 0; c->gather_j = 0; c->pairs_per_photon = 0.0;
    c->d_stat_map = nullptr;
    if (hipHostMalloc((void**)&c->h_stat, 2 * sizeof(sppm_ctx::StatSlot), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_stat_map, c->h_stat, 0) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "pinned host memory for the tile statistics"); }
    for (int b = 0; b < 2; b++)
        if (hipEventCreateWithFlags(&c->ev_stat[b], hipEventDisableTiming) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating events failed"); }
    TRY(dev_alloc(c, &c->d_counters, 24)); TRY(dev_alloc(c, &c->d_r2max, 1)); TRY(dev_alloc(c, &c->d_qmax, 1)); TRY(dev_alloc(c, &c->d_eye_next, 1));
    TRY(dev_alloc(c, &c->d_stats, 1));
    {   // the image indices this rank owns (8x8 screen tiles dealt round-robin), in index order
        std::vector<int32_t> own;
        own.reserve(P / (size_t)c->tile_stride + 64);
        // ownership 1 (ranges of the Morton order of the first eye pass's hitpoints): every pixel until that
        // order exists, then the rank's range of it (adopt_range)
        c->own_pending = cfg->ownership == 1 && c->tile_stride > 1;
        for (int p = 0; p < c->P; p++) {
            const int t = ((p / cfg->width) >> 3) * c->tiles_x + ((p % cfg->width) >> 3) - c->tile_first;
            if (c->own_pending || (t >= 0 && t % c->tile_stride == 0)) own.push_back(p);
        }
        c->n_own_pix = (int)own.size();
        c->n_owned = (c->n_own_pix + 63) / 64;  // virtual tiles of 64 hitpoints
        TRY(dev_alloc(c, &c->d_own_pix, own.size(), false)); for (int k = 0; k < 3; k++) TRY(dev_alloc(c, &c->d_order_r[k], own.size(), false));
        c->d_order = c->d_order_r[0];
        TRY(dev_alloc(c, &c->d_hp_keys, own.size(), false)); TRY(dev_alloc(c, &c->d_hp_keys_out, own.size(), false));
        for (int k = 0; k < 3; k++) TRY(dev_alloc(c, &c->d_hp_bb_r[k], 6));
        c->d_hp_bb = c->d_hp_bb_r[0];
        TRY(dev_alloc(c, &c->d_near_v[0], (size_t)NEAR_DIM * NEAR_DIM * NEAR_DIM)); TRY(dev_alloc(c, &c->d_near_v[1], (size_t)NEAR_DIM * NEAR_DIM * NEAR_DIM)); c->d_near = c->d_near_v[0];
        TRY(dev_alloc(c, &c->d_own_mask, P, false));  // every byte is written by the upload below
        c->all_maps = c->own_pending && c->gather_mode;
        c->view_pass[0] = c->view_pass[1] = -1;
        for (int v = 0; v < 2; v++)
            if (hipEventCreateWithFlags(&c->ev_view[v], hipEventDisableTiming) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "creating events failed"); }
        if (c->all_maps) {
            const size_t N = (size_t)c->tile_stride;
            TRY(dev_alloc(c, &c->d_all_pix, P, false));
            for (int v = 0; v < 2; v++) { TRY(dev_alloc(c, &c->d_near_all_v[v], N * NEAR_DIM * NEAR_DIM * NEAR_DIM)); TRY(dev_alloc(c, &c->d_bb_all_v[v], N * 6)); }
            TRY(dev_alloc(c, &c->d_rows_n_to, N));
            if (hipHostMalloc((void**)&c->h_rows_n_to, N * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "pinned memory for the row counts"); }
        }
        std::vector<uint8_t> mask(P, 0);  // lives until the stream synchronisation at the end of sppm_create
        for (int32_t p : own) mask[(size_t)p] = 1;
        // both uploads on the context's stream, like everything else here (a caller's non-blocking stream is not ordered with the null stream)
        if (P > 0 && hipMemcpyAsync(c->d_own_mask, mask.data(), P, hipMemcpyHostToDevice, c->stream) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "copying the ownership mask failed"); }
        if (!own.empty() && hipMemcpyAsync(c->d_own_pix, own.data(), own.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            sppm_destroy(c); return fail(SPPM_ERR_HIP, "copying the pixel list failed");
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "sppm_create: stream sync failed"); }
        c->sort_tmp_bytes = 0; c->d_sort_tmp = nullptr; c->order_valid = 0;
        TRY(dev_alloc(c, &c->d_est, (size_t)c->n_owned + 1)); TRY(dev_alloc(c, &c->d_items, (size_t)c->n_owned * 64 + 64, false)); TRY(dev_alloc(c, &c->d_items_raw, (size_t)c->n_owned * 64 + 64, false));
        TRY(dev_alloc(c, &c->d_item_q, 2 + 2 * ITEM_CLASSES)); TRY(dev_alloc(c, &c->d_n_pieces, (size_t)c->n_owned + 1)); TRY(dev_alloc(c, &c->d_est_sum, 1)); TRY(dev_alloc(c, &c->d_ext_bin, (size_t)c->n_owned + 1)); c->est_valid = 0;
        c->split_factor = 4.0; c->est_per_photon = 0.0; c->est_range = 0; c->hist_est_range = 0; c->stat_est_range[0] = c->stat_est_range[1] = 0;
        { const char* e = getenv("SPPM_SPLIT_FACTOR"); if (e && *e) c->split_factor = atof(e); }
        if (c->n_own_pix > 0) {
            if (rocprim::radix_sort_pairs<HpSortConfig>(nullptr, c->sort_tmp_bytes, c->d_hp_keys, c->d_hp_keys_out, c->d_own_pix, c->d_order,
                                          (size_t)c->n_own_pix, 0, MORTON_KEY_BITS, c->stream) != hipSuccess) {
                sppm_destroy(c); return fail(SPPM_ERR_HIP, "sizing the hitpoint sort failed");
            }
            char* tmp = nullptr;
            TRY(dev_alloc(c, &tmp, c->sort_tmp_bytes, false));
            c->d_sort_tmp = tmp;
        }
    }
#undef TRY
    set_cur(c, 0);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { sppm_destroy(c); return fail(SPPM_ERR_HIP, "sppm_create: stream sync failed"); }
    // (the address of sppm_intersect_static's array on this device is looked up here, outside any stream capture, so that
    // a captured first call of that entry point has no symbol to resolve; a caller without a context makes one plain
    // call before capturing)
    { unsigned char* unused = nullptr; (void)static_scratch(&unused); }
    *out = c;
    return SPPM_OK;
}

extern "C" int sppm_destroy(sppm_ctx* c)
{
    if (!c) return SPPM_OK;
    for (int b = 0; b < NTS; b++) if (c->tstream[b]) hipStreamSynchronize(c->tstream[b]);
    if (c->eye_stream) { hipStreamSynchronize(c->eye_stream); hipStreamDestroy(c->eye_stream); }
    for (int k = 0; k < 3; k++) if (c->ev_order[k]) hipEventDestroy(c->ev_order[k]);
    if (c->ev_eye_done) hipEventDestroy(c->ev_eye_done);
    if (c->ev_eye_go) hipEventDestroy(c->ev_eye_go);
    hipStreamSynchronize(c->stream);
    for (int b = 0; b < NBUF; b++) { if (c->ev_traced[b]) hipEventDestroy(c->ev_traced[b]); if (c->ev_free[b]) hipEventDestroy(c->ev_free[b]); }
    for (int b = 0; b < NTS; b++) if (c->tstream[b]) hipStreamDestroy(c->tstream[b]);
    for (auto& e : c->ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : c->ev_trace) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (void* p : c->allocs) hipFree(p);
    if (c->h_stat) hipHostFree(c->h_stat);
    if (c->h_rows_n) hipHostFree(c->h_rows_n);
    if (c->h_rows_n_to) hipHostFree(c->h_rows_n_to);
    for (int v = 0; v < 2; v++) if (c->ev_view[v]) hipEventDestroy(c->ev_view[v]);
    if (c->ev_rows) hipEventDestroy(c->ev_rows);
    for (int b = 0; b < 2; b++) if (c->ev_stat[b]) hipEventDestroy(c->ev_stat[b]);
    if (c->d_kd) hipFree(c->d_kd);
    if (c->d_cap_cnt) hipFree(c->d_cap_cnt);
    if (c->d_cap_surv) hipFree(c->d_cap_surv);
    if (c->d_cap_nsurv) hipFree(c->d_cap_nsurv);
    if (c->d_heap) hipFree(c->d_heap);
    if (c->d_cap_cell) { hipFree(c->d_cap_cell); hipFree(c->d_cap_surv_s); hipFree(c->d_cap_cell_s); }
    if (c->d_cap_sort_tmp) hipFree(c->d_cap_sort_tmp);
    if (c->d_veto_tab) hipFree(c->d_veto_tab);
    if (c->d_veto_n) { hipFree(c->d_veto_n); hipFree(c->d_cap_stat); }
    if (c->d_u_eye) hipFree(c->d_u_eye);
    if (c->d_u_lens) hipFree(c->d_u_lens);
    if (c->d_u_ph) hipFree(c->d_u_ph);
    if (c->d_ph_start) hipFree(c->d_ph_start);
    if (c->d_ph_consumed) hipFree(c->d_ph_consumed);
    delete c;
    return SPPM_OK;
}

static inline DevScene scene_of(const sppm_ctx* c) { return c->dscene; }
static inline size_t scene_lds(const sppm_ctx* c) { return c->dscene.n_cells ? (size_t)((c->dscene.n_cells + 31) / 32) * 4 : (size_t)c->ns * sizeof(sppm_sphere); }  // dynamic LDS of STAGE_SCENE
static inline int nblk(int64_t n, int b) { return (int)((n + b - 1) / b); }
static inline double reach_of(const sppm_ctx* c) { return std::sqrt(c->cfg.gather_radius2) * (1.0 + 1e-9) + 1e-9; }
// the occupancy map of the CURRENT view (valid on the context's stream once the eye pass is ordered before it)
static inline NearMap near_map(const sppm_ctx* c, bool use)
{
    NearMap nm; nm.bb = c->d_hp_bb; nm.cells = use ? c->d_near : nullptr; nm.reach = reach_of(c);
    return nm;
}

extern "C" int sppm_camera_rays(sppm_ctx* c, int32_t sx, int32_t sy, double* d_rays)
{
    if (!c || !d_rays) return fail(SPPM_ERR_INVALID, "sppm_camera_rays: null argument");
    CTX_DEVICE(c, "sppm_camera_rays");
    const double* lens_u = nullptr;
    if (c->cam.model == 1 && c->sampler == 1) {  // a fresh lens engine: draws 2i, 2i+1 belong to ray i of the render order
        std::mt19937 eng(c->cfg.seed_lens);
        std::uniform_real_distribution<double> dist(0.0, 1.0);
        std::vector<double> lu((size_t)c->npix * 2);
        for (double& v : lu) v = dist(eng);
        int rc = grow(c, &c->d_u_lens, &c->u_lens_cap, (int64_t)lu.size());
        if (rc) return rc;
        HIPCHK(hipMemcpy(c->d_u_lens, lu.data(), lu.size() * 8, hipMemcpyHostToDevice));
        lens_u = c->d_u_lens;
    }
    hipLaunchKernelGGL(k_camera_rays, dim3(nblk(c->npix, 256)), dim3(256), 0, c->stream, c->cam, c->cfg.width, c->cfg.height, sx, sy, c->cfg.seed_lens, lens_u, d_rays);
    HIPCHK(hipGetLastError());
    if (lens_u) HIPCHK(hipStreamSynchronize(c->stream));  // the buffer is reused by the next eye pass
    return SPPM_OK;
}

extern "C" int sppm_intersect(sppm_ctx* c, const double* d_rays, int64_t n, int32_t* d_id, double* d_hit)
{
    if (!c || n < 0 || (n > 0 && (!d_rays || !d_id || !d_hit))) return fail(SPPM_ERR_INVALID, "sppm_intersect: bad argument");
    CTX_DEVICE(c, "sppm_intersect");
    if (n == 0) return SPPM_OK;
    hipLaunchKernelGGL(k_intersect, dim3(nblk(n, 256)), dim3(256), scene_lds(c), c->stream, scene_of(c), d_rays, n, d_id, d_hit);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// Eye pass on the reference's sequential stream (see k_eye_classify / k_eye_offsets / k_eye_pass_stream).

// ---- classic PPM: the hitpoint list ----------------------------------------------------------------------
static EyeParams eye_params(const sppm_ctx* c, int32_t pass)
{
    EyeParams ep;
    memset(&ep, 0, sizeof ep);
    ep.cam = c->cam; ep.W = c->cfg.width; ep.H = c->cfg.height; ep.samples = c->cfg.samples; ep.supersamples = c->cfg.supersamples;
    ep.pass = pass; ep.seed_eye = c->cfg.seed_eye; ep.initial_r2 = c->cfg.initial_radius2; ep.double_nee = c->cfg.double_nee;
    ep.cnt = c->d_eye_cnt; ep.off = c->d_eye_off; ep.total = c->d_stream_total; ep.hcnt = c->d_hcnt; ep.hoff = c->d_hoff;
    ep.u = c->d_u_eye; ep.u_len = c->u_eye_len;
    ep.seed_lens = c->cfg.seed_lens; ep.lens_u = nullptr; ep.bb_prev = nullptr; ep.bb_out = nullptr; ep.keys_out = nullptr;
    return ep;
}

// Part of sppm_create: trace every pixel's eye paths once without storing anything, to learn how many
// hitpoints each pixel leaves (create_point appends to a vector, src/progressive_photonmapping.h:531).  The
// prefix sum gives each pixel its slots in creation order; sppm_eye_pass(0) then stores the hitpoints.  In
// reference-stream mode the per-pixel stream offsets are found first (classify / walk, as for sppm2.h).
static int ppm_prepare(sppm_ctx* c)
{
    hipStream_t st = c->stream;
    const int npix = c->npix;
    int rc;
    if ((rc = dev_alloc(c, &c->d_hcnt, (size_t)npix))) return rc;
    if ((rc = dev_alloc(c, &c->d_hoff, (size_t)npix + 1))) return rc;
    HpArrays none;
    memset(&none, 0, sizeof none);
    EyeParams ep = eye_params(c, 0);
    if (c->sampler == 1) {
        std::uniform_real_distribution<double> dist(0.0, 1.0);
        const int64_t paths = (int64_t)npix * c->cfg.samples * c->cfg.supersamples * c->cfg.supersamples;
        int64_t need = 8 * paths + 4096;
        hipLaunchKernelGGL(k_eye_classify<1>, dim3(nblk(npix, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, none);
        HIPCHK(hipGetLastError());
        for (int attempt = 0;; attempt++) {
            while ((int64_t)c->eye_pending.size() < need) c->eye_pending.push_back(dist(c->eye_eng));
            if ((rc = grow(c, &c->d_u_eye, &c->u_eye_cap, need))) return rc;
            HIPCHK(hipMemcpyAsync(c->d_u_eye, c->eye_pending.data(), (size_t)need * 8, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(c->d_err + 4, 0, sizeof(int32_t), st));
            c->u_eye_len = need;
            ep = eye_params(c, 0);
            hipLaunchKernelGGL(k_eye_offsets<1>, dim3(1), dim3(64), scene_lds(c), st, scene_of(c), ep, none, c->d_err);
            HIPCHK(hipGetLastError());
            int32_t short_stream = 0;
            HIPCHK(hipMemcpyAsync(&short_stream, c->d_err + 4, sizeof short_stream, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (!short_stream) break;
            if (attempt >= 6) return fail(SPPM_ERR_OVERFLOW, "sppm_create: the eye paths consume more of the stream than 512 draws per path");
            need *= 2;
        }
        hipLaunchKernelGGL(k_ppm_eye_stream<false>, dim3(nblk(npix, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, none, (unsigned long long*)nullptr, c->d_err);
    } else
        hipLaunchKernelGGL(k_ppm_eye<false>, dim3(nblk(npix, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, none, (unsigned long long*)nullptr, c->d_err);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> cnt((size_t)npix);
    HIPCHK(hipMemcpyAsync(cnt.data(), c->d_hcnt, (size_t)npix * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    c->h_hoff.assign((size_t)npix + 1, 0);
    for (int t = 0; t < npix; t++) c->h_hoff[(size_t)t + 1] = c->h_hoff[(size_t)t] + (long long)cnt[(size_t)t];
    const long long nh = c->h_hoff[(size_t)npix];
    // a work item of the tile gather names its tile in 20 bits (64 hitpoints per tile)
    if (nh > (1ll << 26)) return fail(SPPM_ERR_INVALID, "sppm_create: more than 2^26 hitpoints (lower samples / supersamples)");
    HIPCHK(hipMemcpyAsync(c->d_hoff, c->h_hoff.data(), ((size_t)npix + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    c->P = (int)nh;
    return 0;
}

// the eye pass of classic PPM: store the hitpoint list into view nv (same paths as ppm_prepare counted)
static int ppm_eye_emit(sppm_ctx* c, int nv, hipStream_t st)
{
    if (c->P == 0) return 0;
    const EyeParams ep = eye_params(c, 0);
    HpArrays out = c->view[nv];
    if (c->sampler == 1)
        hipLaunchKernelGGL(k_ppm_eye_stream<true>, dim3(nblk(c->npix, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, out, c->d_counters, c->d_err);
    else
        hipLaunchKernelGGL(k_ppm_eye<true>, dim3(nblk(c->npix, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, out, c->d_counters, c->d_err);
    HIPCHK(hipGetLastError());
    return 0;
}

static int eye_pass_stream(sppm_ctx* c, EyeParams& ep)
{
    hipStream_t st = c->stream;
    std::uniform_real_distribution<double> dist(0.0, 1.0);  // the reference's ValueSampler, src/sampler/value_sampler.h:8-15
    const int64_t paths = (int64_t)c->P * c->cfg.samples * c->cfg.supersamples * c->cfg.supersamples;
    int64_t need = 8 * paths + 4096;
    ep.cnt = c->d_eye_cnt; ep.off = c->d_eye_off; ep.total = c->d_stream_total;
    if (c->cam.model == 1) {  // two lens draws per eye ray, in render order, from the lens engine (it runs on across passes)
        std::vector<double> lu((size_t)(2 * paths));
        for (double& v : lu) v = dist(c->lens_eng);
        int rc = grow(c, &c->d_u_lens, &c->u_lens_cap, 2 * paths);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->d_u_lens, lu.data(), lu.size() * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        ep.lens_u = c->d_u_lens;
    }
    hipLaunchKernelGGL(k_eye_classify<0>, dim3(nblk(c->P, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, c->hp);
    HIPCHK(hipGetLastError());
    unsigned long long total = 0;
    for (int attempt = 0;; attempt++) {
        while ((int64_t)c->eye_pending.size() < need) c->eye_pending.push_back(dist(c->eye_eng));
        int rc = grow(c, &c->d_u_eye, &c->u_eye_cap, need);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->d_u_eye, c->eye_pending.data(), (size_t)need * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(c->d_err + 4, 0, sizeof(int32_t), st));
        ep.u = c->d_u_eye; ep.u_len = need;
        hipLaunchKernelGGL(k_eye_offsets<0>, dim3(1), dim3(64), scene_lds(c), st, scene_of(c), ep, c->hp, c->d_err);
        HIPCHK(hipGetLastError());
        int32_t short_stream = 0;
        HIPCHK(hipMemcpyAsync(&short_stream, c->d_err + 4, sizeof short_stream, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&total, c->d_stream_total, sizeof total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!short_stream) break;
        if (attempt >= 6) return fail(SPPM_ERR_OVERFLOW, "sppm_eye_pass: the eye paths consume more of the stream than 512 draws per path");
        need *= 2;
    }
    hipLaunchKernelGGL(k_eye_pass_stream, dim3(nblk(c->P, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, c->hp, c->view[1 - c->cur_view], c->d_counters, c->d_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));  // the upload source shrinks below
    c->eye_pending.erase(c->eye_pending.begin(), c->eye_pending.begin() + (ptrdiff_t)total);
    return 0;
}

// Photon pass on the reference's sequential stream: a fresh mt19937 per pass (src/sppm2.h:530; the seed
// is seed_photon + pass where the reference asks std::random_device), one probe path per stream offset,
// then the chain of start offsets.  After this, k_photon_trace<1> can trace the pass's photons in parallel.
static int prepare_photon_stream(sppm_ctx* c, int32_t pass, int64_t count)
{
    hipStream_t st = c->stream;
    if (count > (1ll << 27)) return fail(SPPM_ERR_INVALID, "reference-stream mode supports at most 2^27 photons per pass");
    std::mt19937 eng(c->cfg.seed_photon + (uint32_t)pass);
    std::uniform_real_distribution<double> dist(0.0, 1.0);
    std::vector<double> u;
    int64_t len = 16 * count + 4096;
    int rc = grow(c, &c->d_ph_start, &c->ph_start_cap, count);
    if (rc) return rc;
    for (int attempt = 0;; attempt++) {
        u.reserve((size_t)len);
        while ((int64_t)u.size() < len) u.push_back(dist(eng));
        if ((rc = grow(c, &c->d_u_ph, &c->u_ph_cap, len))) return rc;
        if ((rc = grow(c, &c->d_ph_consumed, &c->ph_consumed_cap, len))) return rc;
        HIPCHK(hipMemcpyAsync(c->d_u_ph, u.data(), (size_t)len * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(c->d_err + 4, 0, sizeof(int32_t), st));
        HIPCHK(hipMemsetAsync(c->d_trace_next_buf[c->cur], 0, sizeof(unsigned long long), st));
        TraceParams tp;
        memset(&tp, 0, sizeof tp);
        tp.pass = pass; tp.first = 0; tp.count = len; tp.photons_per_pass = c->cfg.photons_per_pass; tp.seed_photon = c->cfg.seed_photon;
        tp.flux_scale = c->flux_scale; tp.rec_cap = 0; tp.u = c->d_u_ph; tp.u_len = len; tp.start = nullptr; tp.consumed = c->d_ph_consumed; tp.ppm = c->algo; tp.miss = c->cfg.tracer_variant;
        const int blocks = (int)std::min<int64_t>(nblk(len, 256), 256 * 4);
        if (c->dscene.n_cells) hipLaunchKernelGGL((k_photon_trace<2, 1>), dim3(blocks), dim3(256), scene_lds(c), st, scene_of(c), tp, c->rec_buf[c->cur], c->d_rec_count_buf[c->cur],
                           c->d_trace_next_buf[c->cur], c->d_counters, c->d_err);
        else hipLaunchKernelGGL((k_photon_trace<2, 0>), dim3(blocks), dim3(256), scene_lds(c), st, scene_of(c), tp, c->rec_buf[c->cur], c->d_rec_count_buf[c->cur],
                           c->d_trace_next_buf[c->cur], c->d_counters, c->d_err);
        hipLaunchKernelGGL(k_photon_chain, dim3(1), dim3(64), 0, st, c->d_ph_consumed, len, count, c->d_ph_start, c->d_stream_total, c->d_err);
        HIPCHK(hipGetLastError());
        int32_t short_stream = 0;
        HIPCHK(hipMemcpyAsync(&short_stream, c->d_err + 4, sizeof short_stream, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!short_stream) break;
        if (attempt >= 6) return fail(SPPM_ERR_OVERFLOW, "sppm_photon_pass: the photons consume more than 1024 draws each on average");
        len *= 2;
    }
    c->stream_len = len; c->stream_pass = pass;
    return 0;
}

// One eye pass on stream st: reads the current view, writes the other one (hitpoints, valid flags, direct
// term, bounding box, tile order).  The caller flips the views once the work is ordered before its readers.
static int eye_launch(sppm_ctx* c, int32_t pass, hipStream_t st)
{
    const int nv = 1 - c->cur_view;
    EyeParams ep;
    ep.cam = c->cam; ep.W = c->cfg.width; ep.H = c->cfg.height; ep.samples = c->cfg.samples; ep.supersamples = c->cfg.supersamples;
    ep.pass = pass; ep.seed_eye = c->cfg.seed_eye; ep.initial_r2 = c->cfg.initial_radius2; ep.double_nee = c->cfg.double_nee;
    ep.u = nullptr; ep.u_len = 0; ep.cnt = nullptr; ep.off = nullptr; ep.total = nullptr; ep.hcnt = nullptr; ep.hoff = nullptr;
    ep.seed_lens = c->cfg.seed_lens; ep.lens_u = nullptr;
    ep.bb_prev = nullptr; ep.bb_out = nullptr; ep.keys_out = nullptr;
    const bool fuse_order = c->fuse_order && c->gather_mode && c->n_own_pix == (int)c->P && c->tile_stride <= 1 && !c->own_pending && c->algo == 0 && c->sampler == 0 && !c->dscene.n_cells &&
                            c->view_chain[c->cur_view] >= 0;
    // the order chain that last read this view's positions must be done before the view is written again
    if (c->view_chain[nv] >= 0) HIPCHK(hipStreamWaitEvent(st, c->ev_order[c->view_chain[nv]], 0));
    if (c->algo == 1) {
        int rc = ppm_eye_emit(c, nv, st);
        if (rc) return rc;
    } else if (c->sampler == 1) {
        int rc = eye_pass_stream(c, ep);  // host-driven, on the context's stream
        if (rc) return rc;
    } else {
        if (c->dscene.n_cells) {  // large scenes: persistent wavefronts, lanes refilled (three waves per SIMD on every CU)
            HIPCHK(hipMemsetAsync(c->d_eye_next, 0, sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_eye_pass_refill<1>, dim3(std::min(nblk(c->P, 256), 256 * 3)), dim3(256), scene_lds(c), st, scene_of(c), ep, c->hp, c->view[nv], c->d_eye_next, c->d_counters);
        }
        else {
            // One GPU's whole image, and a pass before this one has left a bounding box: the eye pass makes the box and the
            // sort keys itself (keys over the box of the pass before; any order of the hitpoints is a valid one).  Two
            // launches and two sweeps over the positions fewer in the chain that the next gather's plan waits for.
            if (fuse_order) {
                const int slot = c->chain_n % 3;
                HIPCHK(hipMemsetAsync(c->d_hp_bb_r[slot], 0xFF, 3 * sizeof(unsigned long long), st));
                HIPCHK(hipMemsetAsync(c->d_hp_bb_r[slot] + 3, 0x00, 3 * sizeof(unsigned long long), st));
                ep.bb_prev = c->d_hp_bb_r[c->view_chain[c->cur_view]]; ep.bb_out = c->d_hp_bb_r[slot]; ep.keys_out = c->d_hp_keys;
            }
            hipLaunchKernelGGL(k_eye_pass<0>, dim3(nblk(c->P, 256)), dim3(256), scene_lds(c), st, scene_of(c), ep, c->hp, c->view[nv], c->d_counters);
        }
        HIPCHK(hipGetLastError());
    }
    if (c->gather_mode && c->n_own_pix > 0) {
        // re-form the gather tiles: sort this rank's hitpoints along a Morton curve of their positions, into the next
        // of the three order buffers
        const int slot = c->chain_n % 3;
        c->chain_n++;
        c->view_chain[nv] = slot;
        if (!fuse_order) {
            // bounding box seed: min slots all ones, max slots zero (ordered-int doubles); two memsets, nothing host-sourced
            HIPCHK(hipMemsetAsync(c->d_hp_bb_r[slot], 0xFF, 3 * sizeof(unsigned long long), st));
            HIPCHK(hipMemsetAsync(c->d_hp_bb_r[slot] + 3, 0x00, 3 * sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_hp_bbox, dim3(std::min(nblk(c->n_own_pix, 256), 512)), dim3(256), 0, st, c->view[nv], c->d_own_pix, c->n_own_pix, c->d_hp_bb_r[slot]);
            hipLaunchKernelGGL(k_hp_sortkeys, dim3(nblk(c->n_own_pix, 256)), dim3(256), 0, st, c->view[nv], c->d_own_pix, c->n_own_pix, c->d_hp_bb_r[slot], c->d_hp_keys);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(rocprim::radix_sort_pairs<HpSortConfig>(c->d_sort_tmp, c->sort_tmp_bytes, c->d_hp_keys, c->d_hp_keys_out, c->d_own_pix, c->d_order_r[slot],
                                         (size_t)c->n_own_pix, 0, MORTON_KEY_BITS, st));
        if (c->own_pending) {
            // Spatial ownership: this rank keeps positions [rank*P/N, (rank+1)*P/N) of the order just made
            // (every rank computes the same order from the same eye pass, so the ranges partition the
            // pixels without any exchange).  A compact region of space per rank keeps tiles dense and lets
            // the binning drop the records far from the rank's hitpoints.
            const int64_t lo = (int64_t)c->tile_first * c->P / c->tile_stride, hi = (int64_t)(c->tile_first + 1) * c->P / c->tile_stride;
            const int n = (int)(hi - lo);
            if (c->all_maps) HIPCHK(hipMemcpyAsync(c->d_all_pix, c->d_order_r[slot], (size_t)c->P * 4, hipMemcpyDeviceToDevice, st));
            if (n > 0) {
                HIPCHK(hipMemcpyAsync(c->d_own_pix, c->d_order_r[slot] + lo, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(c->d_order_r[slot], c->d_own_pix, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            }
            HIPCHK(hipMemsetAsync(c->d_own_mask, 0, (size_t)c->P, st));
            if (n > 0) hipLaunchKernelGGL(k_mask_from_list, dim3(nblk(n, 256)), dim3(256), 0, st, c->d_own_pix, n, c->d_own_mask);
            c->n_own_pix = n; c->n_owned = (n + 63) / 64; c->own_pending = 0;
            if (n > 0) {  // the bounding box of what this rank keeps
                HIPCHK(hipMemsetAsync(c->d_hp_bb_r[slot], 0xFF, 3 * sizeof(unsigned long long), st));
                HIPCHK(hipMemsetAsync(c->d_hp_bb_r[slot] + 3, 0x00, 3 * sizeof(unsigned long long), st));
                hipLaunchKernelGGL(k_hp_bbox, dim3(std::min(nblk(n, 256), 512)), dim3(256), 0, st, c->view[nv], c->d_own_pix, n, c->d_hp_bb_r[slot]);
            }
            HIPCHK(hipGetLastError());
        }
        if (c->tile_stride > 1) {  // which coarse cells lie near this rank's hitpoints (drops far records, k_mark_near)
            HIPCHK(hipMemsetAsync(c->d_near_v[nv], 0, (size_t)NEAR_DIM * NEAR_DIM * NEAR_DIM, st));
            hipLaunchKernelGGL(k_mark_near, dim3(nblk(c->n_own_pix, 256)), dim3(256), 0, st, c->view[nv], c->d_own_pix, c->n_own_pix, c->d_hp_bb_r[slot], reach_of(c), c->cfg.gather_radius2, c->d_near_v[nv]);
            HIPCHK(hipGetLastError());
        }
        if (c->all_maps && !c->own_pending) {  // every rank's bounding box and occupancy map, for the exchange by destination
            const int N = c->tile_stride;
            hipLaunchKernelGGL(k_bb_seed, dim3(1), dim3(64), 0, st, c->d_bb_all_v[nv], N);
            HIPCHK(hipMemsetAsync(c->d_near_all_v[nv], 0, (size_t)N * NEAR_DIM * NEAR_DIM * NEAR_DIM, st));
            for (int pass2 = 0; pass2 < 2; pass2++)
                for (int r = 0; r < N; r++) {
                    const int64_t lo = (int64_t)r * c->P / N, hi = (int64_t)(r + 1) * c->P / N;
                    const int n = (int)(hi - lo);
                    if (n <= 0) continue;
                    if (pass2 == 0) hipLaunchKernelGGL(k_hp_bbox, dim3(std::min(nblk(n, 256), 512)), dim3(256), 0, st, c->view[nv], c->d_all_pix + lo, n, c->d_bb_all_v[nv] + 6 * r);
                    else hipLaunchKernelGGL(k_mark_near, dim3(nblk(n, 256)), dim3(256), 0, st, c->view[nv], c->d_all_pix + lo, n, c->d_bb_all_v[nv] + 6 * r, reach_of(c), c->cfg.gather_radius2,
                                            c->d_near_all_v[nv] + (size_t)r * NEAR_DIM * NEAR_DIM * NEAR_DIM);
                }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(c->ev_order[slot], st));
    }
    c->view_pass[nv] = pass;
    HIPCHK(hipEventRecord(c->ev_view[nv], st));
    return 0;
}

// the view an eye pass has written becomes the current one; what follows on the context's stream is ordered behind
// the tile order made with it
static int flip_view(sppm_ctx* c)
{
    c->cur_view ^= 1;
    c->hp = c->view[c->cur_view]; c->d_near = c->d_near_v[c->cur_view];
    const int own = c->view_chain[c->cur_view];
    if (own >= 0) {
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_order[own], 0));
        c->d_order = c->d_order_r[own]; c->d_hp_bb = c->d_hp_bb_r[own];
    }
    c->order_valid = 1;
    return 0;
}

// Eye pass of pass+1 ahead of its call, on the eye stream, while the gather of `pass` runs: eye paths do not
// depend on photons.  Assumes the caller goes on with sppm_eye_pass(pass + 1); anything else drops it.
static int eye_prefetch(sppm_ctx* c, int32_t pass)
{
    if (!c->prefetch || c->sampler == 1 || c->pre_eye.valid || c->algo == 1) return 0;
    // the other view was last read by the pass before this one, whose work is all on the context's stream by now
    HIPCHK(hipEventRecord(c->ev_eye_go, c->stream));
    HIPCHK(hipStreamWaitEvent(c->eye_stream, c->ev_eye_go, 0));
    int rc = eye_launch(c, pass, c->eye_stream);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_eye_done, c->eye_stream));
    c->pre_eye.valid = 1; c->pre_eye.pass = pass;
    return 0;
}

extern "C" int sppm_eye_pass(sppm_ctx* c, int32_t pass)
{
    if (!c || pass < 0) return fail(SPPM_ERR_INVALID, "sppm_eye_pass: bad argument");
    CTX_DEVICE(c, "sppm_eye_pass");
    if (c->algo == 1 && pass != 0) return fail(SPPM_ERR_INVALID, "sppm_eye_pass: classic progressive photon mapping traces its hitpoints once (pass 0)");
    if (c->pre_eye.valid) {
        c->pre_eye.valid = 0;
        if (c->pre_eye.pass == pass) {  // computed ahead: order it before what follows on the context's stream
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_eye_done, 0));
            int rcf = flip_view(c);
            if (rcf) return rcf;
            c->last_eye_pass = pass;
            return SPPM_OK;
        }
        HIPCHK(hipStreamSynchronize(c->eye_stream));  // not the pass that was expected: its work is dropped
    }
    int rc = eye_launch(c, pass, c->stream);
    if (rc) return rc;
    rc = flip_view(c);
    if (rc) return rc;
    c->last_eye_pass = pass;
    return SPPM_OK;
}

static void set_cur(sppm_ctx* c, int b)
{
    c->cur = b; c->rec_u = c->rec_buf[b]; c->d_rec_count = c->d_rec_count_buf[b]; c->d_trace_next = c->d_trace_next_buf[b];
    const sppm_ctx::BinSet& B = c->bins[b];
    c->rec_s = B.rec_s; c->d_rec_bucket = B.rec_bucket; c->d_cell_cnt = B.cell_cnt; c->d_cell_start = B.cell_start; c->d_cursor = B.cursor;
    c->d_block_sum = B.block_sum; c->d_total = B.total;
}

// launch the tracer for photons [first, first+count) of `pass` into record buffer b on stream st
static int trace_into(sppm_ctx* c, int b, int32_t pass, int64_t first, int64_t count, hipStream_t st)
{
    TraceParams tp;
    tp.pass = pass; tp.first = first; tp.count = count;
    tp.photons_per_pass = c->cfg.photons_per_pass; tp.seed_photon = c->cfg.seed_photon; tp.flux_scale = c->flux_scale; tp.rec_cap = c->rec_buf[b].cap;
    // persistent wavefronts that take photons from one counter, never more than the photons need
    HIPCHK(hipMemsetAsync(c->d_trace_next_buf[b], 0, sizeof(unsigned long long), st));
    // A launch lasts about as long as its longest photon path whatever it is given (3-6 ms from 64 workgroups to 1024,
    // measured), and launches overlap each other and the gather: on the 10-sphere scene 128 workgroups do the work and
    // leave the SIMDs to the gather (+2-3 % photons/s); the grid walk of a large scene is real work and takes them all.
    const int trace_blocks = (int)std::min<int64_t>(nblk(count, 256), env_i64("SPPM_TRACE_BLOCKS", c->dscene.n_cells ? 256 * 4 : 128));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    tp.u = nullptr; tp.u_len = 0; tp.start = nullptr; tp.consumed = nullptr; tp.ppm = c->algo; tp.miss = c->cfg.tracer_variant;
    if (c->sampler == 1) {
        if (c->stream_pass != pass) return fail(SPPM_ERR_STATE, "reference-stream mode: the pass's stream offsets are made by sppm_photon_pass(pass, 0, photons)");
        tp.u = c->d_u_ph; tp.u_len = c->stream_len; tp.start = c->d_ph_start;
        if (c->dscene.n_cells) hipLaunchKernelGGL((k_photon_trace<1, 1>), dim3(trace_blocks), dim3(256), scene_lds(c), st, scene_of(c), tp, c->rec_buf[b], c->d_rec_count_buf[b], c->d_trace_next_buf[b], c->d_counters, c->d_err);
        else hipLaunchKernelGGL((k_photon_trace<1, 0>), dim3(trace_blocks), dim3(256), scene_lds(c), st, scene_of(c), tp, c->rec_buf[b], c->d_rec_count_buf[b], c->d_trace_next_buf[b], c->d_counters, c->d_err);
    } else {
        if (c->dscene.n_cells) hipLaunchKernelGGL((k_photon_trace<0, 1>), dim3(trace_blocks), dim3(256), scene_lds(c), st, scene_of(c), tp, c->rec_buf[b], c->d_rec_count_buf[b], c->d_trace_next_buf[b], c->d_counters, c->d_err);
        else hipLaunchKernelGGL((k_photon_trace<0, 0>), dim3(trace_blocks), dim3(256), scene_lds(c), st, scene_of(c), tp, c->rec_buf[b], c->d_rec_count_buf[b], c->d_trace_next_buf[b], c->d_counters, c->d_err);
    }
    if (c->timing) { HIPCHK(hipEventRecord(e1, st)); c->ev_trace.push_back({e0, e1}); }
    HIPCHK(hipGetLastError());
    c->photons_traced += (uint64_t)count;
    return 0;
}

static int discard_prefetch(sppm_ctx* c);

extern "C" int sppm_trace(sppm_ctx* c, int32_t pass, int64_t first, int64_t count)
{
    if (!c || pass < 0 || first < 0 || count < 0) return fail(SPPM_ERR_INVALID, "sppm_trace: bad argument");
    CTX_DEVICE(c, "sppm_trace");
    if (c->cfg.photons_per_pass > 0 && first + count > c->cfg.photons_per_pass)
        return fail(SPPM_ERR_INVALID, "sppm_trace: photon range exceeds photons_per_pass");
    if (count == 0) return SPPM_OK;
    if (!c->ahead.empty()) { int rc2 = discard_prefetch(c); if (rc2) return rc2; }
    if (c->pre.valid && c->pre.pass == pass && c->pre.first == first && c->pre.count == count && c->rec_end == c->rec_first) {
        // traced ahead (sppm_look_ahead): order it before what follows and make its buffer the current one
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_traced[c->pre.buf], 0));
        set_cur(c, c->pre.buf);
        c->bins[c->pre.buf].binned = 0;
        c->pre.valid = 0;
        c->rec_first = first; c->rec_end = first + count;
        return SPPM_OK;
    }
    if (c->pre.valid && c->pre.buf == c->cur) set_cur(c, (c->cur + 1) % NBUF);  // keep away from a batch traced ahead
    int rc = trace_into(c, c->cur, pass, first, count, c->stream);
    if (rc) return rc;
    if (c->rec_end == c->rec_first) { c->rec_first = first; c->rec_end = first + count; }
    else { c->rec_first = std::min(c->rec_first, first); c->rec_end = std::max(c->rec_end, first + count); }
    return SPPM_OK;
}

static int discard_prefetch(sppm_ctx* c);
static int eye_prefetch(sppm_ctx* c, int32_t pass);

extern "C" int sppm_look_ahead(sppm_ctx* c, int32_t pass, int64_t first, int64_t count)
{
    if (!c || pass < 0 || first < 0 || count < 0) return fail(SPPM_ERR_INVALID, "sppm_look_ahead: bad argument");
    CTX_DEVICE(c, "sppm_look_ahead");
    if (c->cfg.photons_per_pass > 0 && first + count > c->cfg.photons_per_pass)
        return fail(SPPM_ERR_INVALID, "sppm_look_ahead: photon range exceeds photons_per_pass");
    if (c->sampler == 1 || count == 0) return SPPM_OK;  // the sequential reference stream cannot run ahead
    int rc = discard_prefetch(c);
    if (rc) return rc;
    const int nb = (c->cur + 1) % NBUF;
    // the other buffer is free once everything enqueued on the context's stream so far has run
    HIPCHK(hipEventRecord(c->ev_free[nb], c->stream));
    HIPCHK(hipStreamWaitEvent(TSTREAM(c, nb), c->ev_free[nb], 0));
    HIPCHK(hipMemsetAsync(c->d_rec_count_buf[nb], 0, sizeof(unsigned long long), TSTREAM(c, nb)));
    rc = trace_into(c, nb, pass, first, count, TSTREAM(c, nb));
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_traced[nb], TSTREAM(c, nb)));
    c->pre.valid = 1; c->pre.buf = nb; c->pre.pass = pass; c->pre.first = first; c->pre.count = count;
    if (pass == c->last_eye_pass + 1) { rc = eye_prefetch(c, pass); if (rc) return rc; }
    return SPPM_OK;
}

// sppm_look_ahead followed, on the tracer's stream, by the packing of what was traced into d_rows: the rows
// of this rank's block of the NEXT batch get ready while the context's stream is busy with the gather of this
// one, and the caller can exchange them in the meantime.  sppm_rows_ready waits for them (host side).
extern "C" int sppm_look_ahead_rows(sppm_ctx* c, int32_t pass, int64_t first, int64_t count, double* d_rows, int64_t cap_rows)
{
    if (!c || !d_rows || cap_rows < 0) return fail(SPPM_ERR_INVALID, "sppm_look_ahead_rows: bad argument");
    CTX_DEVICE(c, "sppm_look_ahead_rows");
    if (c->sampler == 1) return fail(SPPM_ERR_STATE, "sppm_look_ahead_rows: the sequential reference stream cannot run ahead");
    c->rows_pending = 0;
    int rc = sppm_look_ahead(c, pass, first, count);
    if (rc) return rc;
    const int nb = count > 0 ? c->pre.buf : (c->cur + 1) % NBUF;
    hipStream_t ts = TSTREAM(c, nb);
    HIPCHK(hipMemsetAsync(c->d_rows_n, 0, sizeof(unsigned long long), ts));
    if (count > 0) {
        hipLaunchKernelGGL(k_pack_records, dim3(nblk(c->rec_buf[nb].cap, 4 * WAVE_ROWS)), dim3(256), 0, ts, c->rec_buf[nb], c->d_rec_count_buf[nb], d_rows, cap_rows, c->d_rows_n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(c->d_rec_count_buf[nb], 0, sizeof(unsigned long long), ts));  // the records live in the rows now
        c->pre.valid = 0;
    }
    HIPCHK(hipMemcpyAsync(c->h_rows_n, c->d_rows_n, sizeof(unsigned long long), hipMemcpyDeviceToHost, ts));
    HIPCHK(hipEventRecord(c->ev_rows, ts));
    c->rows_pending = 1; c->rows_cap = cap_rows;
    return SPPM_OK;
}

extern "C" int sppm_rows_ready(sppm_ctx* c, int64_t* n)
{
    if (!c || !n) return fail(SPPM_ERR_INVALID, "sppm_rows_ready: null argument");
    if (c->rows_pending != 1) return fail(SPPM_ERR_STATE, "sppm_rows_ready: no sppm_look_ahead_rows is outstanding");
    HIPCHK(hipEventSynchronize(c->ev_rows));
    c->rows_pending = 0;
    *n = (int64_t)*c->h_rows_n;
    if (*n > c->rows_cap) return fail(SPPM_ERR_OVERFLOW, "sppm_rows_ready: the row buffer was too small (n holds the count needed)");
    return SPPM_OK;
}

// sppm_look_ahead_rows with the rows sorted by DESTINATION: d_rows is [tile_stride][cap_rows][10], block d holds the
// records rank d can need -- those whose cell of rank d's occupancy map (made by the eye pass of `pass`, which every
// rank runs) is set.  A superset of what d will use; d applies its own map again when it appends them.  If the eye pass
// of `pass` has not been started on this context yet, every record goes to every rank.
extern "C" int sppm_look_ahead_rows_to(sppm_ctx* c, int32_t pass, int64_t first, int64_t count, double* d_rows, int64_t cap_rows)
{
    if (!c || !d_rows || cap_rows < 0) return fail(SPPM_ERR_INVALID, "sppm_look_ahead_rows_to: bad argument");
    CTX_DEVICE(c, "sppm_look_ahead_rows_to");
    if (!c->all_maps) return fail(SPPM_ERR_STATE, "sppm_look_ahead_rows_to: needs several ranks with spatial ownership (tile_stride > 1, ownership 1) and the tile gather");
    if (c->sampler == 1) return fail(SPPM_ERR_STATE, "sppm_look_ahead_rows_to: the sequential reference stream cannot run ahead");
    c->rows_pending = 0;
    int rc = sppm_look_ahead(c, pass, first, count);  // starts the eye pass of `pass` too, if it is the next one
    if (rc) return rc;
    const int nb = count > 0 ? c->pre.buf : (c->cur + 1) % NBUF;
    hipStream_t ts = TSTREAM(c, nb);
    const int N = c->tile_stride;
    int v = -1;
    if (!c->own_pending) {
        if (c->last_eye_pass == pass && c->view_pass[c->cur_view] == pass) v = c->cur_view;
        else if (c->pre_eye.valid && c->pre_eye.pass == pass && c->view_pass[1 - c->cur_view] == pass) v = 1 - c->cur_view;
    }
    if (v >= 0) HIPCHK(hipStreamWaitEvent(ts, c->ev_view[v], 0));
    HIPCHK(hipMemsetAsync(c->d_rows_n_to, 0, (size_t)N * sizeof(unsigned long long), ts));
    if (count > 0) {
        hipLaunchKernelGGL(k_pack_records_to, dim3(nblk(c->rec_buf[nb].cap, 4 * WAVE_ROWS)), dim3(256), 0, ts, c->rec_buf[nb], c->d_rec_count_buf[nb], d_rows, cap_rows, c->d_rows_n_to, N,
                           v >= 0 ? c->d_bb_all_v[v] : (const unsigned long long*)nullptr, v >= 0 ? c->d_near_all_v[v] : (const uint8_t*)nullptr, reach_of(c));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(c->d_rec_count_buf[nb], 0, sizeof(unsigned long long), ts));  // the records live in the rows now
        c->pre.valid = 0;
    }
    HIPCHK(hipMemcpyAsync(c->h_rows_n_to, c->d_rows_n_to, (size_t)N * sizeof(unsigned long long), hipMemcpyDeviceToHost, ts));
    HIPCHK(hipEventRecord(c->ev_rows, ts));
    c->rows_pending = 2; c->rows_cap = cap_rows;
    return SPPM_OK;
}

// waits for the rows of sppm_look_ahead_rows_to; h_counts[tile_stride]: rows in each destination's block
extern "C" int sppm_rows_ready_to(sppm_ctx* c, int64_t* h_counts)
{
    if (!c || !h_counts) return fail(SPPM_ERR_INVALID, "sppm_rows_ready_to: null argument");
    if (c->rows_pending != 2) return fail(SPPM_ERR_STATE, "sppm_rows_ready_to: no sppm_look_ahead_rows_to is outstanding");
    HIPCHK(hipEventSynchronize(c->ev_rows));
    c->rows_pending = 0;
    bool over = false;
    for (int d = 0; d < c->tile_stride; d++) { h_counts[d] = (int64_t)c->h_rows_n_to[d]; over |= h_counts[d] > c->rows_cap; }
    if (over) return fail(SPPM_ERR_OVERFLOW, "sppm_rows_ready_to: a destination's row block was too small (h_counts holds the counts needed)");
    return SPPM_OK;
}

// number of record slots in use in the current buffer (holes included)
static int slots_in_use(sppm_ctx* c, int64_t* n)
{
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, c->d_rec_count, sizeof cnt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((int64_t)cnt > c->rec_u.cap) return fail(SPPM_ERR_OVERFLOW, "record buffer overflow: raise SPPM_RECORD_CAP or lower SPPM_BATCH_PHOTONS");
    *n = (int64_t)cnt;
    return 0;
}

extern "C" int sppm_pack_records(sppm_ctx* c, double* d_rows, int64_t cap_rows, int64_t* n)
{
    if (!c || !n || cap_rows < 0 || (cap_rows > 0 && !d_rows)) return fail(SPPM_ERR_INVALID, "sppm_pack_records: bad argument");
    CTX_DEVICE(c, "sppm_pack_records");
    HIPCHK(hipMemsetAsync(c->d_r2max, 0, sizeof(unsigned long long), c->stream));  // reused as the row counter
    hipLaunchKernelGGL(k_pack_records, dim3(nblk(c->rec_u.cap, 4 * WAVE_ROWS)), dim3(256), 0, c->stream, c->rec_u, c->d_rec_count, d_rows, cap_rows, c->d_r2max);
    HIPCHK(hipGetLastError());
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, c->d_r2max, sizeof cnt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *n = (int64_t)cnt;
    if ((int64_t)cnt > cap_rows) return fail(SPPM_ERR_OVERFLOW, "sppm_pack_records: the row buffer is too small (n holds the count needed)");
    HIPCHK(hipMemsetAsync(c->d_rec_count, 0, sizeof(unsigned long long), c->stream));
    c->rec_first = c->rec_end = 0;
    return SPPM_OK;
}

extern "C" int sppm_append_records(sppm_ctx* c, const double* d_rows, int64_t n, int64_t photon_first, int64_t photon_end)
{
    if (!c || n < 0 || (n > 0 && !d_rows) || photon_first < 0 || photon_end < photon_first) return fail(SPPM_ERR_INVALID, "sppm_append_records: bad argument");
    CTX_DEVICE(c, "sppm_append_records");
    if (photon_end > photon_first) {
        if (c->rec_end == c->rec_first) { c->rec_first = photon_first; c->rec_end = photon_end; }
        else { c->rec_first = std::min(c->rec_first, photon_first); c->rec_end = std::max(c->rec_end, photon_end); }
    }
    if (n == 0) return SPPM_OK;
    const bool drop_far = c->gather_mode && c->tile_stride > 1 && c->order_valid && c->n_own_pix > 0;
    hipLaunchKernelGGL(k_append_records, dim3(nblk(n, 4 * WAVE_ROWS)), dim3(256), 0, c->stream, c->rec_u, c->d_rec_count, d_rows, n, c->d_err, near_map(c, drop_far));
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

static inline Own own_of(const sppm_ctx* c) { Own o; o.mask = c->d_own_mask; return o; }

static int run_scan(sppm_ctx* c, const sppm_ctx::BinSet& B, uint32_t n_padded, hipStream_t st, unsigned long long* records_acc)
{
    hipLaunchKernelGGL(k_scan_blocks, dim3(n_padded / SCAN_BLOCK), dim3(SCAN_BLOCK), 0, st, B.cell_cnt, B.cell_start, B.block_sum);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(SCAN_BLOCK), 0, st, B.block_sum, (int)(n_padded / SCAN_BLOCK), B.total, records_acc);
    hipLaunchKernelGGL(k_scan_add, dim3(n_padded / SCAN_BLOCK), dim3(SCAN_BLOCK), 0, st, B.cell_start, B.block_sum, B.cursor, (size_t)n_padded, B.total, B.cell_cnt);
    HIPCHK(hipGetLastError());
    return 0;
}

static int finish_batch(sppm_ctx* c)
{
    hipStream_t st = c->stream;
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, c->d_rec_count, sizeof cnt, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if ((int64_t)cnt > c->rec_u.cap) return fail(SPPM_ERR_OVERFLOW, "record buffer overflow: raise SPPM_RECORD_CAP or lower SPPM_BATCH_PHOTONS");
    if (cnt > 0) {  // slots include the holes of the tracer's chunked reservation: the scan total counts the records
        uint32_t valid = 0;
        HIPCHK(hipMemcpy(&valid, c->d_total, sizeof valid, hipMemcpyDeviceToHost));
        c->records_total += valid;
    }
    c->gather_batches++;
    c->photons_applied += (uint64_t)(c->rec_end - c->rec_first);
    c->rec_first = c->rec_end = 0;
    HIPCHK(hipMemsetAsync(c->d_rec_count, 0, sizeof(unsigned long long), st));
    return SPPM_OK;
}

// Simple gather (cross-check path, SPPM_GATHER_TILES=0): hash grid over the records, one thread per
// hitpoint selecting its next keys by rescanning its neighbourhood.
static int gather_simple(sppm_ctx* c)
{
    hipStream_t st = c->stream;
    const uint32_t nb = c->n_buckets;
    if (c->P == 0) return finish_batch(c);
    HIPCHK(hipMemsetAsync(c->d_r2max, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_r2max, dim3(std::min(nblk(c->P, 256), 2048)), dim3(256), 0, st, c->hp, own_of(c), c->cfg.gather_radius2, c->d_r2max);
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, c->d_r2max, sizeof bits, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double r2max;
    memcpy(&r2max, &bits, 8);
    if (!(r2max > 0.0)) return finish_batch(c);  // no valid hitpoint on this rank: nothing can be accepted
    Grid g;
    g.inv_cell = 1.0 / std::sqrt(r2max);
    g.mask = nb - 1;
    HIPCHK(hipMemsetAsync(c->d_cell_cnt, 0, sizeof(uint32_t) * nb, st));
    hipLaunchKernelGGL(k_grid_count, dim3(2048), dim3(256), 0, st, c->rec_u, c->d_rec_count, g, c->d_cell_cnt, c->d_rec_bucket);
    int rc = run_scan(c, c->bins[c->cur], nb, st, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_grid_scatter, dim3(2048), dim3(256), 0, st, c->rec_u, c->rec_s, c->d_rec_count, c->d_rec_bucket, c->d_cursor, 0, 0.0, c->d_err);
    GatherParams gp;
    gp.g = g; gp.gather_r2 = c->cfg.gather_radius2; gp.alpha = c->cfg.alpha; gp.rule = c->cfg.update_rule;
    gp.own = own_of(c); gp.unit_weight = c->algo; gp.fx_scale = c->fx_scale; gp.err = c->d_err;
    gp.tap_key = c->d_tap_key; gp.tap_pix = c->d_tap_pix; gp.tap_n = c->d_tap_n; gp.tap_cap = c->tap_cap;
    gp.veto = c->veto_dev;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    hipLaunchKernelGGL(k_gather, dim3(nblk(c->P, 256)), dim3(256), 0, st, c->hp, scene_of(c), c->rec_s, c->d_cell_start, gp, c->d_counters);
    if (c->timing) { HIPCHK(hipEventRecord(e1, st)); c->ev.push_back({e0, e1}); }
    HIPCHK(hipGetLastError());
    return finish_batch(c);
}

// ---- tile gather, host side ---------------------------------------------------------------------------
// Statistics of the tiles' regions (extent histogram, bounding box, counters): launched on the context's
// stream into a pinned slot, read by the host one batch later.
static int stats_launch(sppm_ctx* c)
{
    hipStream_t st = c->stream;
    TileParams tp;
    memset(&tp, 0, sizeof tp);
    tp.gather_r2 = c->cfg.gather_radius2; tp.order = c->d_order; tp.n_pix = c->n_own_pix; tp.n_owned = c->n_owned;
    // the tiles' extents and the mean estimate move slowly once a render is under way: measured every fourth batch then
    // (the counters below are copied every time; the slab length is derived from their differences batch by batch)
    if (c->stat_n < 16 || (c->stat_n & 3) == 0) {
        hipLaunchKernelGGL(k_tile_extent, dim3(nblk(c->n_owned, 4)), dim3(256), 0, st, c->hp, tp, c->d_ext_bin);
        hipLaunchKernelGGL(k_tile_hist, dim3(1), dim3(1024), 0, st, c->d_ext_bin, c->est_valid ? c->d_est : (const uint32_t*)nullptr, c->n_owned, c->d_stats);
        HIPCHK(hipGetLastError());
        c->hist_est_range = c->est_valid ? c->est_range : 0;  // the batch whose plan wrote d_est
    }
    c->stat_est_range[c->stat_w] = c->hist_est_range;
    hipLaunchKernelGGL(k_stat_snapshot, dim3(1), dim3(64), 0, st, c->d_stats, c->d_hp_bb, c->d_counters, c->d_stat_map + c->stat_w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_stat[c->stat_w], st));
    c->stat_w ^= 1; c->stat_n++;
    return 0;
}

static double undo_ord(unsigned long long o)
{
    const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

// Wait for the most recent statistics (already there in steady state) and derive what depends on them:
// the spatial part of the grid, the cluster distance, the mean record estimate per photon and the slab length.
static int stats_read(sppm_ctx* c)
{
    if (c->stat_n == 0) { int rc = stats_launch(c); if (rc) return rc; }
    const int r = c->stat_w ^ 1;  // the slot written last
    HIPCHK(hipEventSynchronize(c->ev_stat[r]));
    const sppm_ctx::StatSlot& sl = c->h_stat[r];
    const TileStats& hs = sl.hs;
    if (hs.live_tiles > 0) {
        // median of the tiles' region extents, interpolated inside its octave so the cell size moves smoothly
        int med = 0;
        double med_frac = 1.0;
        { unsigned int acc = 0; for (int e = 0; e < 64; e++) { acc += hs.hist[e]; if (2 * acc >= hs.live_tiles) { med = e; med_frac = hs.hist[e] ? 1.0 - (double)(acc - (hs.live_tiles + 1) / 2) / (double)hs.hist[e] : 1.0; break; } } }
        double cell = std::ldexp(1.0, med - 33) * std::exp2(med_frac) * (c->cfg.update_rule == 1 ? c->cell_factor_dep : c->cell_factor);
        double hi[3];
        // grid box: the hitpoints' bounding box (made by the eye pass) grown by the largest search radius;
        // positions outside it (a later pass may move hitpoints) fall into the border cells, on both sides alike
        const double rgrow = std::sqrt(c->cfg.gather_radius2) * (1.0 + 1e-9) + 1e-9;
        for (int k = 0; k < 3; k++) { c->gs.lo[k] = undo_ord(sl.bbq[k]) - rgrow; hi[k] = undo_ord(sl.bbq[3 + k]) + rgrow; }
        for (;;) {
            c->gs.ncells = 1;
            for (int k = 0; k < 3; k++) { c->gs.dims[k] = std::max<int64_t>(1, (int64_t)std::ceil((hi[k] - c->gs.lo[k]) / cell)); c->gs.ncells *= c->gs.dims[k]; if (c->gs.ncells > (1ll << 40)) break; }
            if (c->gs.ncells <= c->table_cap / 4) break;
            cell *= 1.25;
        }
        c->gs.cell = cell;
        c->gs.cluster_d = std::ldexp(1.0, med - 32);  // upper edge of the median tile extent
        c->gs.valid = 1;
        // (per photon of the batch the estimates were made for -- batches differ in size, and the slot is read a batch or
        // two after it was written: dividing by the latest batch's size made the split threshold of the next plan 10x too
        // small or too large whenever a short batch at the end of a pass was followed by a long one)
        if (c->stat_est_range[r] > 0) c->est_per_photon = (double)hs.est_sum / (double)hs.live_tiles / (double)c->stat_est_range[r];
    } else if (!c->gs.valid) {  // no live hitpoint yet: any grid will do
        for (int k = 0; k < 3; k++) { c->gs.lo[k] = 0.0; c->gs.dims[k] = 1; }
        c->gs.ncells = 1; c->gs.cell = 1.0; c->gs.cluster_d = 1.0; c->gs.valid = 1;
    }
    // slab length: aim at ~20 in-region candidates per (tile, slab), far from the 64 of one round.  The
    // snapshot of slot r was taken when gather j-1 began, the one before it when gather j-2 began: their
    // difference is what gather j-2 did, with the slab length it used.
    if (c->snap_prev_valid && c->gather_j >= 2 && c->cfg.update_rule == 0 && !getenv("SPPM_SLAB_PHOTONS")) {
        const double tiles_run = (double)(sl.cnt[7] - c->snap_prev[7]);
        const int64_t S = c->hist_S[(c->gather_j - 2) & 3], B = c->hist_B[(c->gather_j - 2) & 3];
        if (tiles_run > 0 && S > 0) {
            const double avg = (double)(sl.cnt[5] - c->snap_prev[5]) / (tiles_run * (double)S);
            double f = avg > 0 ? 20.0 / avg : 4.0;
            f = std::min(4.0, std::max(0.25, f));
            c->slab_B = std::max<int64_t>(64, (int64_t)((double)B * f));
        }
    }
    if (c->snap_prev_valid && c->gather_j >= 2 && c->cfg.update_rule == 1) {
        const int64_t range = c->hist_range[(c->gather_j - 2) & 3];
        if (range > 0 && sl.cnt[3] >= c->snap_prev[3]) c->pairs_per_photon = (double)(sl.cnt[3] - c->snap_prev[3]) / (double)range;
    }
    memcpy(c->snap_prev, sl.cnt, sizeof c->snap_prev);
    c->snap_prev_valid = 1;
    return 0;
}

// the (slab, cell) grid of a batch of photons [first, first + range): space from the statistics, time from the slab length
static TileGrid grid_for(const sppm_ctx* c, int64_t first, int64_t range)
{
    int64_t B = std::max<int64_t>(64, std::min<int64_t>(c->slab_B, range));
    if (c->cfg.update_rule == 1) B = std::max<int64_t>(1, range);  // the per-pass rule keeps no order: one slab
    int64_t S = (range + B - 1) / B;
    if (S * c->gs.ncells > c->table_cap) { S = std::max<int64_t>(1, c->table_cap / c->gs.ncells); B = (range + S - 1) / S; S = (range + B - 1) / B; }
    TileGrid g;
    g.ox = c->gs.lo[0]; g.oy = c->gs.lo[1]; g.oz = c->gs.lo[2]; g.inv_c = 1.0 / c->gs.cell;
    g.nx = (int)c->gs.dims[0]; g.ny = (int)c->gs.dims[1]; g.nz = (int)c->gs.dims[2]; g.ncells = (int)c->gs.ncells;
    g.S = (int)S; g.B = B; g.first = first;
    return g;
}

// counting sort of record buffer b into its (slab, cell) table, on stream st
static int bin_into(sppm_ctx* c, int b, const TileGrid& g, hipStream_t st, bool drop_far = false)
{
    sppm_ctx::BinSet& B = c->bins[b];
    const uint32_t n_table = (uint32_t)((int64_t)g.S * g.ncells);
    const uint32_t n_padded = (n_table + SCAN_BLOCK - 1) / SCAN_BLOCK * SCAN_BLOCK;
    // B.cell_cnt is all zero here: allocated so, and every scan clears what it has read (k_scan_add)
    // drop_far: only where the current view's occupancy map is final on this stream (not for batches binned ahead)
    // (binning ahead of its gather, on a tracer stream, has batches of slack: a narrow grid leaves the window between
    // two gathers to the kernels the next gather waits for)
    static const int bin_blocks_ahead = (int)env_i64("SPPM_BIN_BLOCKS", 256);
    const int bin_blocks = st == c->stream ? 2048 : bin_blocks_ahead;
    hipLaunchKernelGGL(k_bin_count, dim3(bin_blocks), dim3(256), 0, st, c->rec_buf[b], c->d_rec_count_buf[b], g, B.cell_cnt, B.rec_bucket, c->d_err,
                       near_map(c, drop_far));
    int rc = run_scan(c, B, n_padded, st, c->d_counters + 2);
    if (rc) return rc;
    hipLaunchKernelGGL(k_grid_scatter, dim3(bin_blocks), dim3(256), 0, st, c->rec_buf[b], B.rec_s, c->d_rec_count_buf[b], B.rec_bucket, B.cursor, 1,
                       c->cfg.update_rule == 1 ? c->fx_scale : 0.0, c->d_err);  // (rule 1 here: deposit_tiles is the reader)
    // the per-photon rule's plan reads the records of a region from a summed-area table over the cells: it depends on the
    // binning alone, so it is made here -- for a batch binned ahead, on the tracer's stream, outside the window between
    // two gathers (round 3: four launches and ~280 us less in that window)
    if (B.cell_tot) {
        hipLaunchKernelGGL(k_cell_totals, dim3(nblk(g.ncells, 256)), dim3(256), 0, st, B.cell_start, g.ncells, g.S, B.cell_tot);
        hipLaunchKernelGGL(k_sat_axis, dim3(nblk(g.ny * g.nz, 64)), dim3(64), 0, st, B.cell_tot, g.nx, g.ny, g.nz, 0);
        hipLaunchKernelGGL(k_sat_axis, dim3(nblk(g.nx * g.nz, 64)), dim3(64), 0, st, B.cell_tot, g.nx, g.ny, g.nz, 1);
        hipLaunchKernelGGL(k_sat_axis, dim3(nblk(g.nx * g.ny, 64)), dim3(64), 0, st, B.cell_tot, g.nx, g.ny, g.nz, 2);
    }
    HIPCHK(hipGetLastError());
    B.g = g; B.binned = 1;
    return 0;
}

// Tile gather of the current record buffer: (bin it if that has not happened on the tracer's stream,)
// plan the work items from the current radii, replay.  No stream synchronisation: errors are flagged
// on the device and reported by sppm_check.
static int gather_tiles(sppm_ctx* c)
{
    hipStream_t st = c->stream;
    const int64_t range = c->rec_end - c->rec_first;
    sppm_ctx::BinSet& BS = c->bins[c->cur];
    if (c->n_owned == 0 || range == 0) {
        BS.binned = 0;
        c->gather_batches++; c->photons_applied += (uint64_t)range; c->rec_first = c->rec_end = 0;
        HIPCHK(hipMemsetAsync(c->d_rec_count, 0, sizeof(unsigned long long), st));
        return SPPM_OK;
    }
    if (!c->order_valid) return fail(SPPM_ERR_STATE, "sppm_gather: no eye pass has run on this context yet");
    int rc = stats_read(c);
    if (rc) return rc;
    rc = stats_launch(c);  // for the batch after this one
    if (rc) return rc;
    if (!BS.binned) { rc = bin_into(c, c->cur, grid_for(c, c->rec_first, range), st, c->tile_stride > 1); if (rc) return rc; }
    const TileGrid g = BS.g;
    TileParams tp;
    memset(&tp, 0, sizeof tp);
    tp.gather_r2 = c->cfg.gather_radius2; tp.alpha = c->cfg.alpha; tp.rule = c->cfg.update_rule;
    tp.order = c->d_order; tp.n_pix = c->n_own_pix; tp.n_owned = c->n_owned;
    tp.tap_key = c->d_tap_key; tp.tap_pix = c->d_tap_pix; tp.tap_n = c->d_tap_n; tp.tap_cap = c->tap_cap;
    tp.g = g;
    tp.cluster_d = c->gs.cluster_d;
    tp.veto = c->veto_dev;
    const double cell = 1.0 / g.inv_c;
    const int64_t S = g.S, B = g.B;
    c->hist_S[c->gather_j & 3] = S; c->hist_B[c->gather_j & 3] = B; c->gather_j++;
    const bool dbg = getenv("SPPM_DEBUG") != nullptr;
    unsigned long long cnt0[24];
    if (dbg) { HIPCHK(hipMemcpyAsync(cnt0, c->d_counters, sizeof cnt0, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
    // --- work items: cut tiles whose region holds far more records than the average tile's.  The
    // threshold comes from the previous batch's mean (scaled by the photon count); the first batch is not cut.
    const double mean_est = c->est_per_photon * (double)range;
    const double thr_d = c->est_per_photon > 0.0 ? std::max(256.0, mean_est * c->split_factor) : 4.0e9;
    const uint32_t split_thr = (uint32_t)std::min(4.0e9, thr_d);
    HIPCHK(hipMemsetAsync(c->d_item_q, 0, sizeof(unsigned int) * (2 + ITEM_CLASSES), st));
    hipLaunchKernelGGL(k_tile_plan, dim3(nblk(c->n_owned, 4)), dim3(256), 0, st, c->hp, BS.cell_tot, tp, split_thr, c->d_items_raw, c->d_n_pieces, c->d_est);
    hipLaunchKernelGGL(k_items_hist, dim3(nblk(c->n_owned, 256)), dim3(256), 0, st, c->d_items_raw, c->d_n_pieces, c->n_owned, c->d_item_q);
    hipLaunchKernelGGL(k_items_bases, dim3(1), dim3(64), 0, st, c->d_item_q, c->d_rec_count);
    hipLaunchKernelGGL(k_items_scatter, dim3(nblk(c->n_owned, 256)), dim3(256), 0, st, c->d_items_raw, c->d_n_pieces, c->n_owned, c->d_item_q, c->d_items);
    c->est_valid = 1; c->est_range = range;
    // persistent wavefronts: 4 per SIMD on every CU (fewer if there are few tiles)
    const unsigned int n_items = (unsigned int)c->n_owned * 64u + 64u;  // capacity of the item list (diagnostics only)
    const int gather_blocks = std::max(1, std::min(nblk(c->n_owned, 4), (int)env_i64("SPPM_GATHER_BLOCKS", 256 * GATHER_WAVES)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (dbg) { HIPCHK(hipMemsetAsync(c->d_counters + 8, 0, 16 * sizeof(unsigned long long), st)); }
    if (c->timing || dbg) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    unsigned long long* d_item_times = nullptr;
    if (getenv("SPPM_ITEM_TIMES") && getenv("SPPM_STAMPS")) {
        HIPCHK(hipMalloc((void**)&d_item_times, (size_t)n_items * 32));
        HIPCHK(hipMemsetAsync(d_item_times, 0, (size_t)n_items * 32, st));
        tp.item_times = d_item_times;
    }
    if (getenv("SPPM_STAMPS"))
        hipLaunchKernelGGL((k_gather_tiles<true, false>), dim3(gather_blocks), dim3(256), 0, st, c->hp, scene_of(c), BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters);
    else if (c->algo == 1)
        hipLaunchKernelGGL((k_gather_tiles<false, true>), dim3(gather_blocks), dim3(256), 0, st, c->hp, scene_of(c), BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters);
    else
        hipLaunchKernelGGL((k_gather_tiles<false, false>), dim3(gather_blocks), dim3(256), 0, st, c->hp, scene_of(c), BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters);
    if (c->timing || dbg) { HIPCHK(hipEventRecord(e1, st)); }
    HIPCHK(hipGetLastError());
    float dbg_ms = 0.f;
    if (dbg) { HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&dbg_ms, e0, e1)); }
    if (d_item_times) {
        std::vector<unsigned long long> t((size_t)n_items * 4);
        HIPCHK(hipMemcpy(t.data(), d_item_times, t.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("SPPM_ITEM_TIMES"), "wb")) { fwrite(t.data(), 8, t.size(), f); fclose(f); }
        hipFree(d_item_times);
        tp.item_times = nullptr;
    }
    if (c->timing) c->ev.push_back({e0, e1});
    else if (dbg) { hipEventDestroy(e0); hipEventDestroy(e1); }
    BS.binned = 0;
    c->gather_batches++;
    c->photons_applied += (uint64_t)range;
    c->rec_first = c->rec_end = 0;
    // (the buffer's record counter was cleared by k_items_bases)
    if (!dbg) return SPPM_OK;
    unsigned long long cnt1[24];
    HIPCHK(hipMemcpyAsync(cnt1, c->d_counters, sizeof cnt1, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const double tiles_run = (double)(cnt1[7] - cnt0[7]);
    if (dbg)
        fprintf(stderr, "[sppm] batch %lld photons: %u items for %d tiles (mean est %.0f), gather %.2f ms, max per tile: sel %llu cand %llu rounds %llu cycles %llu (avg cycles %.0f); cell %.3f grid %dx%dx%d S=%d B=%lld rounds/tile-slab %.2f sel/tile-slab %.2f cand/sel %.2f next B %lld\n",
                (long long)range, n_items, c->n_owned, mean_est, dbg_ms, cnt1[8], cnt1[9], cnt1[10], cnt1[11], (double)cnt1[12] / std::max(1.0, tiles_run), cell, g.nx, g.ny, g.nz, g.S, (long long)B, (double)(cnt1[6] - cnt0[6]) / std::max(1.0, tiles_run * S),
                (double)(cnt1[5] - cnt0[5]) / std::max(1.0, tiles_run * S), (double)(cnt1[14]) / std::max(1.0, (double)(cnt1[5] - cnt0[5])),
                (long long)c->slab_B);
    if (dbg && getenv("SPPM_STAMPS"))
        fprintf(stderr, "[sppm]   stamps (share of item cycles): scan %.1f%% rank+load %.1f%% apply %.1f%% (of which walk %.1f%%); walk trips per round %.2f, updates per round %.1f\n", 100.0 * cnt1[16] / std::max(1.0, (double)cnt1[12]),
                100.0 * cnt1[17] / std::max(1.0, (double)cnt1[12]), 100.0 * cnt1[18] / std::max(1.0, (double)cnt1[12]), 100.0 * cnt1[19] / std::max(1.0, (double)cnt1[12]),
                (double)cnt1[20] / std::max(1.0, (double)(cnt1[6] - cnt0[6])), (double)(cnt1[4] - cnt0[4]) / std::max(1.0, (double)(cnt1[6] - cnt0[6])));
    if (dbg) {
        const int hk = (int)(cnt1[13] & 0xffffff);
        std::vector<int32_t> ord(64), obj(64);
        const int nn = std::min(64, c->n_own_pix - hk * 64);
        hipMemcpy(ord.data(), c->d_order + (size_t)hk * 64, nn * 4, hipMemcpyDeviceToHost);
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, rmin = 1e300, rmax = 0;
        int objs[16] = {0};
        for (int i = 0; i < nn; i++) {
            double v[3], r2; int32_t o, val;
            hipMemcpy(&val, c->hp.valid + ord[i], 4, hipMemcpyDeviceToHost);
            if (!val) continue;
            for (int a = 0; a < 3; a++) { hipMemcpy(&v[a], c->hp.pos + (size_t)a * c->P + ord[i], 8, hipMemcpyDeviceToHost); mn[a] = std::min(mn[a], v[a]); mx[a] = std::max(mx[a], v[a]); }
            hipMemcpy(&r2, c->hp.r2 + ord[i], 8, hipMemcpyDeviceToHost); rmin = std::min(rmin, r2); rmax = std::max(rmax, r2);
            hipMemcpy(&o, c->hp.obj + ord[i], 4, hipMemcpyDeviceToHost); if (o >= 0 && o < 16) objs[o]++;
        }
        uint32_t hest = 0;
        fprintf(stderr, "[sppm]   est %u thr %u; heaviest tile %d (sel %llu): bbox [%.2f %.2f %.2f]-[%.2f %.2f %.2f] r2 %.4f..%.4f objs", hest, split_thr, hk, cnt1[13] >> 24, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], rmin, rmax);
        for (int o = 0; o < 16; o++) if (objs[o]) fprintf(stderr, " %d:%d", o, objs[o]);
        fprintf(stderr, "\n");
    }
    return SPPM_OK;
}

// The per-pass rule's gather of the current record buffer (sppm_deposit.hpp): bin by cell (one slab), plan the work
// items from the summed-area table, deposit.  Same stream discipline as gather_tiles.
static int deposit_tiles(sppm_ctx* c)
{
    hipStream_t st = c->stream;
    const int64_t range = c->rec_end - c->rec_first;
    sppm_ctx::BinSet& BS = c->bins[c->cur];
    if (c->n_owned == 0 || range == 0) {
        BS.binned = 0;
        c->gather_batches++; c->photons_applied += (uint64_t)range; c->rec_first = c->rec_end = 0;
        HIPCHK(hipMemsetAsync(c->d_rec_count, 0, sizeof(unsigned long long), st));
        return SPPM_OK;
    }
    if (!c->order_valid) return fail(SPPM_ERR_STATE, "sppm_gather: no eye pass has run on this context yet");
    int rc = stats_read(c);
    if (rc) return rc;
    // (the statistics for the batch after this one are made by the plan and its two item kernels below: no launches of their own)
    if (!BS.binned) { rc = bin_into(c, c->cur, grid_for(c, c->rec_first, range), st, c->tile_stride > 1); if (rc) return rc; }
    const TileGrid g = BS.g;
    TileParams tp;
    memset(&tp, 0, sizeof tp);
    tp.gather_r2 = c->cfg.gather_radius2; tp.alpha = c->cfg.alpha; tp.rule = 1;
    tp.order = c->d_order; tp.n_pix = c->n_own_pix; tp.n_owned = c->n_owned;
    tp.tap_key = c->d_tap_key; tp.tap_pix = c->d_tap_pix; tp.tap_n = c->d_tap_n; tp.tap_cap = c->tap_cap;
    tp.g = g;
    tp.cluster_d = c->gs.cluster_d;
    tp.veto = c->veto_dev;
    c->hist_range[c->gather_j & 3] = range;
    c->gather_j++;
    const int n_dep_tiles = (c->n_own_pix + DEP_TILE - 1) / DEP_TILE;  // DEP_H hitpoints per lane
    hipLaunchKernelGGL(k_deposit_plan, dim3(nblk(n_dep_tiles, 4)), dim3(256), 0, st, c->hp, BS.cell_start, tp, n_dep_tiles, (uint32_t)c->dep_chunk, c->d_items_raw, c->d_n_pieces, c->d_est, c->d_item_q,
                       c->d_ext_bin, c->d_stats);
    hipLaunchKernelGGL(k_items_count, dim3(nblk(n_dep_tiles, 256)), dim3(256), 0, st, c->d_items_raw, c->d_n_pieces, n_dep_tiles, c->d_item_q, c->d_ext_bin, c->d_stats);
    hipLaunchKernelGGL(k_items_place, dim3(nblk(n_dep_tiles, 256)), dim3(256), 0, st, c->d_items_raw, c->d_n_pieces, n_dep_tiles, c->d_item_q, c->d_items, c->d_rec_count,
                       c->d_stats, c->d_hp_bb, c->d_counters, c->d_stat_map + c->stat_w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_stat[c->stat_w], st));
    c->stat_w ^= 1; c->stat_n++;
    c->est_valid = 0;  // (the estimates of the ordered gather's plan are not made here)
    // Persistent workgroups: three of the four wavefront slots of every SIMD while the deposit is the pass's long pole
    // (the first passes of a render: ~2800 pairs examined per photon at 1080p), down to 2.25 of them once it is not
    // (~1600 pairs per photon from pass ~100 on): what runs beside the kernel -- the eye pass of the next pass with its
    // sort, the tracer, the binning -- gets one workgroup slot per CU next to three resident deposit workgroups, and late
    // in a render the next deposit waits for that eye pass, not for this kernel.  Measured (photons/s at 20 / 256
    // passes): 768 workgroups 267M / 305M, 704 266M / 310M, 640 260M / 313M, 576 249M / 317M, 512 237M / 313M.
    int dep_blocks = 256 * 3;
    if (c->pairs_per_photon > 0.0) {
        const double t = std::min(1.0, std::max(0.0, (c->pairs_per_photon - c->dep_light) / (c->dep_heavy - c->dep_light)));
        dep_blocks = 576 + 64 * (int)std::lround(3.0 * t);
    }
    const int blocks = std::max(1, std::min(nblk(n_dep_tiles, 4) * 4, (int)env_i64("SPPM_DEP_BLOCKS", dep_blocks)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    const bool stamps = getenv("SPPM_STAMPS") != nullptr;
    // SPPM_DEP_WG=1: the workgroup form with the distance test on the matrix pipe (sppm_deposit_wg.hpp) -- the same deposits
    // bit for bit, measured SLOWER than the default; kept as the record of that experiment
    const bool dep_wg = env_i64("SPPM_DEP_WG", 0) != 0 && DEP_H == 1;
    if (stamps) {
        HIPCHK(hipMemsetAsync(c->d_counters + 15, 0, 9 * sizeof(unsigned long long), st));
        HIPCHK(hipMemsetAsync(c->d_counters + 8, 0, 2 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL((k_deposit_tiles<false, true>), dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
        unsigned long long t[24];
        HIPCHK(hipMemcpyAsync(t, c->d_counters, sizeof t, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double tot = (double)(t[16] + t[17] + t[18] + t[19] + t[20] + t[21]);
        fprintf(stderr, "[sppm] deposit: items %llu, shares of wave cycles: setup %.1f%% count %.1f%% issue %.1f%% append %.1f%% (of it waiting for the group's loads %.1f%%) process %.1f%% flush %.1f%%; grid %dx%dx%d\n", t[7],
                100.0 * t[16] / tot, 100.0 * t[17] / tot, 100.0 * t[18] / tot, 100.0 * t[19] / tot, 100.0 * t[23] / tot, 100.0 * t[20] / tot, 100.0 * t[21] / tot, g.nx, g.ny, g.nz);
        fprintf(stderr, "[sppm]   staged records by class: distance test only %llu, distance + pair flatness bound %llu (groups that needed the exact expressions: %llu), full test %llu (records rejected for the whole cluster are not staged)\n", t[22], t[8], t[9], t[15]);
    } else if (c->veto_dev.n != nullptr) {  // the K cap acts on this batch: the pairs it dropped are withheld
        if (c->tap_cap > 0) hipLaunchKernelGGL((k_deposit_tiles<true, false, true>), dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
        else hipLaunchKernelGGL((k_deposit_tiles<false, false, true>), dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
    } else if (dep_wg) {
        if (c->tap_cap > 0) hipLaunchKernelGGL(k_deposit_wg<true>, dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
        else hipLaunchKernelGGL(k_deposit_wg<false>, dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
    } else if (c->tap_cap > 0)
        hipLaunchKernelGGL(k_deposit_tiles<true>, dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
    else
        hipLaunchKernelGGL(k_deposit_tiles<false>, dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
    if (c->timing) { HIPCHK(hipEventRecord(e1, st)); c->ev.push_back({e0, e1}); }
    HIPCHK(hipGetLastError());
    BS.binned = 0;
    c->gather_batches++;
    c->photons_applied += (uint64_t)range;
    c->rec_first = c->rec_end = 0;
    // (the buffer's record counter was cleared by k_items_bases)
    return SPPM_OK;
}

// ---- K cap, host side (sppm_cap.hpp) --------------------------------------------------------------------
// The reference's tree: KDTree::BalanceKDTreeNode (src/kdtree.h:165-198) sorts a range of its vector with std::sort on
// one coordinate, takes the middle as the node, and recurses into the halves.  The same std::sort (this is libstdc++'s,
// as in the reference's build) on the same sequence with the same comparison leaves the same permutation, ties
// included -- so the vector is kept here as the reference keeps it: a hitpoint is appended when its pixel first finds
// one, in render order (src/sppm2.h:204-209, the y/x loops of :625-646), and every pass sorts the vector in place from
// the order the pass before left (kdtree.h:87).  Classic PPM builds once from the list's order.  Subtrees below the
// first levels are sorted by separate threads (their ranges are disjoint).
static void cap_sort_range(int32_t* v, const double* pos, int64_t P, int begin, int end, int depth, int par_levels)
{
    if (end - begin <= 0) return;
    const double* ax = pos + (int64_t)(depth % 3) * P;
    std::sort(v + begin, v + end, [ax](int32_t a, int32_t b) { return ax[a] < ax[b]; });
    const int mid = begin + (end - begin) / 2;
    if (par_levels > 0 && end - begin > 4096) {
        std::thread t(cap_sort_range, v, pos, P, begin, mid, depth + 1, par_levels - 1);
        cap_sort_range(v, pos, P, mid + 1, end, depth + 1, par_levels - 1);
        t.join();
    } else {
        cap_sort_range(v, pos, P, begin, mid, depth + 1, 0);
        cap_sort_range(v, pos, P, mid + 1, end, depth + 1, 0);
    }
}
static int cap_build_tree(sppm_ctx* c)
{
    if (c->kd_pass == c->last_eye_pass && !c->kd_order.empty()) return 0;
    const int64_t P = c->P;
    HIPCHK(hipStreamSynchronize(c->eye_stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const auto t_build = std::chrono::steady_clock::now();
    std::vector<int32_t> valid((size_t)P);
    std::vector<double> pos((size_t)P * 3);
    HIPCHK(hipMemcpy(valid.data(), c->hp.valid, (size_t)P * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pos.data(), c->hp.pos, (size_t)P * 24, hipMemcpyDeviceToHost));
    if (c->kd_listed.size() != (size_t)P) { c->kd_listed.assign((size_t)P, 0); c->kd_order.clear(); }
    if (c->algo == 1) {
        if (c->kd_order.empty()) for (int64_t i = 0; i < P; i++) if (valid[(size_t)i]) c->kd_order.push_back((int32_t)i);
    } else {
        const int W = c->cfg.width, H = c->cfg.height;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const int index = (H - y - 1) * W + x;
                if (valid[(size_t)index] && !c->kd_listed[(size_t)index]) { c->kd_listed[(size_t)index] = 1; c->kd_order.push_back(index); }
            }
    }
    const int n = (int)c->kd_order.size();
    cap_sort_range(c->kd_order.data(), pos.data(), P, 0, n, 0, 4);
    if (grow(c, &c->d_kd, &c->kd_cap, std::max<int64_t>(n, 1))) return SPPM_ERR_HIP;
    {   // a node as the walk reads it: one 32-byte word (position, hitpoint slot)
        std::vector<double4> nodes((size_t)std::max(n, 1));
        for (int i = 0; i < n; i++) {
            const size_t h = (size_t)c->kd_order[(size_t)i];
            nodes[(size_t)i] = make_double4(pos[h], pos[(size_t)P + h], pos[2 * (size_t)P + h], (double)c->kd_order[(size_t)i]);
        }
        HIPCHK(hipMemcpy(c->d_kd, nodes.data(), (size_t)n * sizeof(double4), hipMemcpyHostToDevice));
    }
    {   // the filter's grid: the tree's hitpoints counted in cells of half the search radius (at most 2^22 cells)
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) { const double v = pos[(size_t)a * P + c->kd_order[(size_t)i]]; lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
        if (n == 0) for (int a = 0; a < 3; a++) lo[a] = hi[a] = 0.0;
        double cell = 0.5 * std::sqrt(c->cfg.gather_radius2);
        if (!(cell > 0.0)) cell = 1.0;
        int64_t dims[3];
        for (;;) {
            int64_t cells = 1;
            for (int a = 0; a < 3; a++) { dims[a] = std::max<int64_t>(1, (int64_t)std::floor((hi[a] - lo[a]) / cell) + 1); cells *= dims[a]; if (cells > (1ll << 40)) break; }
            if (cells <= (1ll << 22)) break;
            cell *= 1.26;
        }
        CapGrid& g = c->cap_grid;
        g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2]; g.inv_c = 1.0 / cell; g.nx = (int)dims[0]; g.ny = (int)dims[1]; g.nz = (int)dims[2];
        std::vector<uint32_t> cnt((size_t)(dims[0] * dims[1] * dims[2]), 0u);
        auto coord = [](double v, double o, double inv, int nn) { int q = (int)std::floor((v - o) * inv); return q < 0 ? 0 : (q >= nn ? nn - 1 : q); };  // as grid_coord on the device
        for (int i = 0; i < n; i++) {
            const size_t h = (size_t)c->kd_order[(size_t)i];
            const int ix = coord(pos[h], g.ox, g.inv_c, g.nx), iy = coord(pos[(size_t)P + h], g.oy, g.inv_c, g.ny), iz = coord(pos[2 * (size_t)P + h], g.oz, g.inv_c, g.nz);
            cnt[((size_t)iz * g.ny + iy) * g.nx + ix]++;
        }
        if (grow(c, &c->d_cap_cnt, &c->cap_cnt_cap, (int64_t)cnt.size())) return SPPM_ERR_HIP;
        HIPCHK(hipMemcpy(c->d_cap_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice));
        g.cnt = c->d_cap_cnt;
    }
    c->kd_pass = c->last_eye_pass;
    if (getenv("SPPM_DEBUG")) fprintf(stderr, "[sppm] K cap: tree of %d hitpoints built on the host in %.1f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count());
    return 0;
}
// the capped search of every record in the current buffer; leaves the dropped pairs in the Veto table
static int cap_search(sppm_ctx* c)
{
    int rc = cap_build_tree(c);
    if (rc) return rc;
    const int n = (int)c->kd_order.size();
    if (c->cfg.gather_max >= (int64_t)n) return 0;  // fewer hitpoints than K: no heap can overflow
    hipStream_t st = c->stream;
    const int64_t K = c->cfg.gather_max;
    if (K < 0) return fail(SPPM_ERR_INVALID, "gather_max is negative");
    // the photon hits the cap can act on at all (k_cap_filter): usually a few percent, often none
    if (grow(c, &c->d_cap_surv, &c->cap_surv_cap, c->rec_u.cap)) return SPPM_ERR_HIP;
    if (c->cap_cell_cap < c->rec_u.cap) {
        if (c->d_cap_cell) { hipFree(c->d_cap_cell); hipFree(c->d_cap_surv_s); hipFree(c->d_cap_cell_s); c->d_cap_cell = nullptr; }
        HIPCHK(hipMalloc((void**)&c->d_cap_cell, (size_t)c->rec_u.cap * 4));
        HIPCHK(hipMalloc((void**)&c->d_cap_surv_s, (size_t)c->rec_u.cap * 4));
        HIPCHK(hipMalloc((void**)&c->d_cap_cell_s, (size_t)c->rec_u.cap * 4));
        c->cap_cell_cap = c->rec_u.cap;
    }
    if (!c->d_cap_nsurv) HIPCHK(hipMalloc((void**)&c->d_cap_nsurv, 8));
    HIPCHK(hipMemsetAsync(c->d_cap_nsurv, 0, 8, st));
    const double reach = std::sqrt(c->cfg.gather_radius2) * (1.0 + 1e-9) + 1e-9;
    hipLaunchKernelGGL(k_cap_filter, dim3(1024), dim3(256), 0, st, c->rec_u, c->d_rec_count, c->cap_grid, reach, (long long)K, c->d_cap_surv, c->d_cap_cell, c->d_cap_nsurv);
    HIPCHK(hipGetLastError());
    unsigned long long n_surv = 0;
    HIPCHK(hipMemcpyAsync(&n_surv, c->d_cap_nsurv, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    c->cap_filtered += n_surv;
    if (n_surv == 0) return 0;  // no photon hit of this batch can see more than K hitpoints: the cap cannot act
    // Photon hits of one wavefront should walk the same part of the tree (their node reads then fall into the same
    // cache lines and their walks are about equally long): take the survivors cell by cell.
    {
        size_t need = 0;
        if (rocprim::radix_sort_pairs(nullptr, need, c->d_cap_cell, c->d_cap_cell_s, c->d_cap_surv, c->d_cap_surv_s, (size_t)n_surv, 0, 32, st) != hipSuccess)
            return fail(SPPM_ERR_HIP, "K cap: sort size query failed");
        if (need > c->cap_sort_bytes) {
            if (c->d_cap_sort_tmp) hipFree(c->d_cap_sort_tmp);
            c->d_cap_sort_tmp = nullptr; c->cap_sort_bytes = 0;
            HIPCHK(hipMalloc(&c->d_cap_sort_tmp, need));
            c->cap_sort_bytes = need;
        }
        size_t bytes = c->cap_sort_bytes;
        HIPCHK(rocprim::radix_sort_pairs(c->d_cap_sort_tmp, bytes, c->d_cap_cell, c->d_cap_cell_s, c->d_cap_surv, c->d_cap_surv_s, (size_t)n_surv, 0, 32, st));
    }
    // One heap of K + 1 entries per photon hit in flight.  The walk is a chain of dependent reads, so the more of them
    // are in flight the better: up to a quarter of the free device memory (SPPM_CAP_HEAP_MB overrides), 64 GiB at most.
    const int64_t stride = K + 1;
    size_t mem_free = 0, mem_total = 0;
    HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
    const int64_t held = c->heap_cap * (int64_t)sizeof(double2);
    const int64_t budget = env_i64("SPPM_CAP_HEAP_MB", std::min<int64_t>(65536, (int64_t)(((int64_t)mem_free + held) / 4) >> 20)) << 20;
    int64_t lanes = std::min<int64_t>(1 << 20, std::max<int64_t>(64, budget / ((int64_t)sizeof(double2) * stride)));
    lanes = std::min<int64_t>(lanes, ((int64_t)n_surv + 63) / 64 * 64);  // heaps for the photon hits that need one
    lanes = std::max<int64_t>(64, lanes / 64 * 64);
    if (grow(c, &c->d_heap, &c->heap_cap, lanes * stride)) return SPPM_ERR_HIP;
    if (!c->d_veto_n) { HIPCHK(hipMalloc((void**)&c->d_veto_n, 8)); HIPCHK(hipMalloc((void**)&c->d_cap_stat, 32)); }
    for (int attempt = 0;; attempt++) {
        if (!c->d_veto_tab) {
            if (c->veto_entries < (1 << 16)) c->veto_entries = 1 << 16;
            HIPCHK(hipMalloc((void**)&c->d_veto_tab, (size_t)c->veto_entries * 16));
        }
        HIPCHK(hipMemsetAsync(c->d_veto_tab, 0, (size_t)c->veto_entries * 16, st));
        HIPCHK(hipMemsetAsync(c->d_veto_n, 0, 8, st));
        HIPCHK(hipMemsetAsync(c->d_cap_stat, 0, 32, st));
        CapParams cp;
        memset(&cp, 0, sizeof cp);
        cp.surv = c->d_cap_surv_s; cp.n_surv = (long long)n_surv;
        cp.kd = c->d_kd; cp.n = n; cp.R0 = c->cfg.gather_radius2; cp.K = K;
        cp.heap = c->d_heap; cp.heap_stride = stride; cp.lanes = (int)lanes;
        cp.veto.tab = c->d_veto_tab; cp.veto.mask = (unsigned long long)c->veto_entries - 1ull; cp.veto.n = c->d_veto_n;
        cp.veto_n = c->d_veto_n; cp.veto_cap = (unsigned long long)c->veto_entries / 2ull;
        cp.stat = c->d_cap_stat; cp.own = own_of(c);
        hipLaunchKernelGGL(k_cap_search, dim3((unsigned)(lanes / 64)), dim3(64), 0, st, c->hp, c->rec_u, c->d_rec_count, cp, c->d_err);
        HIPCHK(hipGetLastError());
        unsigned long long got = 0, stat[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(&got, c->d_veto_n, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(stat, c->d_cap_stat, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (got <= cp.veto_cap) {
            c->cap_capped += stat[0]; c->cap_nodes += stat[1]; c->cap_max_fill = std::max<uint64_t>(c->cap_max_fill, stat[2]); c->cap_vetoes += got;
            c->veto_dev = cp.veto;
            if (getenv("SPPM_DEBUG")) fprintf(stderr, "[sppm] K cap: %llu photon hits past the filter, %llu queries capped, %llu nodes walked, %llu pairs dropped (table %lld entries, %lld lanes)\n", n_surv, stat[0], stat[1], got, (long long)c->veto_entries, (long long)lanes);
            return 0;
        }
        if (attempt >= 2) return fail(SPPM_ERR_OVERFLOW, "K cap: the table of dropped pairs could not be sized");
        int64_t need = 1 << 16;
        while ((unsigned long long)need < got * 5ull / 2ull) need <<= 1;  // fill of at most 0.4
        hipFree(c->d_veto_tab); c->d_veto_tab = nullptr;
        c->veto_entries = need;
    }
}

// grid build + ordered gather over whatever is in the unsorted record buffer; clears the buffer
extern "C" int sppm_gather(sppm_ctx* c)
{
    if (!c) return fail(SPPM_ERR_INVALID, "sppm_gather: null context");
    CTX_DEVICE(c, "sppm_gather");
    if (c->cfg.check_gather_max == 1 && c->P > 0 && c->rec_end != c->rec_first) {
        hipLaunchKernelGGL(k_count_queries, dim3(nblk(c->rec_u.cap, 256)), dim3(256), 0, c->stream, c->hp, c->rec_u, c->d_rec_count, c->cfg.gather_radius2,
                           (long long)c->cfg.gather_max, c->d_err, c->d_qmax);
        HIPCHK(hipGetLastError());
    }
    c->veto_dev.n = nullptr;
    // check_gather_max 0 (default): the K cap is enforced, exactly, whenever it could act (K below the number of
    // hitpoints in the tree); 1: counted only (above); 2: ignored (the uncapped set, a documented deviation)
    if (c->cfg.check_gather_max == 0 && c->cfg.gather_max < (int64_t)c->P && c->P > 0 && c->rec_end != c->rec_first) {
        int rc = cap_search(c);
        if (rc) return rc;
    }
    if (!c->gather_mode) return gather_simple(c);
    return c->cfg.update_rule == 1 ? deposit_tiles(c) : gather_tiles(c);
}

static int64_t batch_size(const sppm_ctx* c, uint64_t applied)
{
    // Batches grow with the photons already applied: the radii shrink like N^-0.43, so a batch about
    // half as long as the history changes them little while it runs.
    int64_t n = c->batch_photons;
    if (c->gather_mode && c->cfg.update_rule == 0) n = std::min<int64_t>(n, std::max<int64_t>(c->batch_min, (int64_t)((double)applied * c->batch_growth)));
    return n;
}

// drop the batches that were traced ahead but are not the ones now asked for
static int discard_prefetch(sppm_ctx* c)
{
    std::vector<sppm_ctx::Pre> all(c->ahead.begin(), c->ahead.end());
    if (c->pre.valid) all.push_back(c->pre);
    if (!all.empty() && getenv("SPPM_DEBUG")) fprintf(stderr, "[sppm] look-ahead dropped: %zu batches (first: pass %d photons %lld+%lld)\n", all.size(), all[0].pass, (long long)all[0].first, (long long)all[0].count);
    for (const sppm_ctx::Pre& q : all) {
        const int b = q.buf;
        HIPCHK(hipStreamSynchronize(TSTREAM(c, b)));
        HIPCHK(hipMemsetAsync(c->d_rec_count_buf[b], 0, sizeof(unsigned long long), c->stream));
        HIPCHK(hipEventRecord(c->ev_free[b], c->stream));
        c->bins[b].binned = 0;
        c->photons_traced -= (uint64_t)q.count;
    }
    c->ahead.clear();
    c->pre.valid = 0;
    return 0;
}

// trace and bin photons [first, first+count) of `pass` into ring buffer `buf` on that buffer's stream, once the gather
// that last used the buffer is done
static int launch_ahead(sppm_ctx* c, int buf, int32_t pass, int64_t first, int64_t count)
{
    HIPCHK(hipStreamWaitEvent(TSTREAM(c, buf), c->ev_free[buf], 0));
    int rc = trace_into(c, buf, pass, first, count, TSTREAM(c, buf));
    if (rc) return rc;
    if (c->gather_mode && c->n_owned > 0) { rc = bin_into(c, buf, grid_for(c, first, count), TSTREAM(c, buf)); if (rc) return rc; }
    HIPCHK(hipEventRecord(c->ev_traced[buf], TSTREAM(c, buf)));
    return 0;
}

// The batch expected after `cur`, given that the caller asked for photons [first, first + count) of `pass` and is
// assumed to go on the same way: the rest of this call, then the same range of the next pass (SPPM) or the photons
// that follow (classic PPM, one run fed in several calls).  `applied`: photons gathered once `cur` has been.
static bool batch_after(const sppm_ctx* c, int64_t first, int64_t count, const sppm_ctx::Pre& cur, uint64_t applied, sppm_ctx::Pre* out)
{
    out->valid = 1; out->buf = -1;
    const int64_t cur_end = cur.first + cur.count;
    if (c->algo == 1) {  // consecutive photons of the one run; calls are assumed to stay `count` long
        if (cur_end >= c->cfg.photons_per_pass) return false;
        const bool same_call = (cur_end - first) % count != 0;
        if (!same_call && !c->prefetch) return false;
        const int64_t call_end = std::min<int64_t>(c->cfg.photons_per_pass, first + ((cur_end - first) / count + 1) * count);
        out->pass = cur.pass; out->first = cur_end; out->count = std::min(batch_size(c, applied), call_end - cur_end);
        return out->count > 0;
    }
    if (cur_end < first + count) { out->pass = cur.pass; out->first = cur_end; out->count = std::min(batch_size(c, applied), first + count - cur_end); return true; }
    if (!c->prefetch) return false;
    out->pass = cur.pass + 1; out->first = first; out->count = std::min(batch_size(c, applied), count);
    return true;
}

// Photon pass in batches.  While the gather of a batch runs on the context's stream, the tracer already fills the
// next record buffers of the ring with the batches expected after it, each on a stream of its own -- across the end
// of the pass and of the call too: photon paths do not depend on hitpoints, so the batches of the NEXT passes are
// traced ahead on the assumption that the caller will ask for the same photon range again (dropped if it does
// not).  A tracer launch lasts as long as its longest photon path -- a few photons bounce thousands of times inside
// the glass sphere, 10 ms and more against ~4 ms per pass -- so the look-ahead is several batches deep (NBUF - 1):
// with two, every third pass waited for its tracer (profiles/r02_pass_gaps_*.txt).
extern "C" int sppm_photon_pass(sppm_ctx* c, int32_t pass, int64_t first, int64_t count)
{
    if (!c || pass < 0 || first < 0 || count < 0) return fail(SPPM_ERR_INVALID, "sppm_photon_pass: bad argument");
    CTX_DEVICE(c, "sppm_photon_pass");
    if (c->cfg.photons_per_pass > 0 && first + count > c->cfg.photons_per_pass)
        return fail(SPPM_ERR_INVALID, "sppm_photon_pass: photon range exceeds photons_per_pass");
    if (count == 0) return SPPM_OK;
    if (c->rec_end != c->rec_first) return fail(SPPM_ERR_STATE, "sppm_photon_pass: records of an earlier sppm_trace are still waiting for sppm_gather");
    int rc;
    if (c->sampler == 1) {
        if (first != 0) return fail(SPPM_ERR_INVALID, "sppm_photon_pass: reference-stream mode traces a pass from photon 0 (the stream is sequential)");
        rc = prepare_photon_stream(c, pass, count);
        if (rc) return rc;
    }
    if (c->gather_mode && c->n_owned > 0) {  // the grid's spatial part must exist before a batch is binned
        if (!c->order_valid) return fail(SPPM_ERR_STATE, "sppm_photon_pass: no eye pass has run on this context yet");
        if (!c->gs.valid) { rc = stats_read(c); if (rc) return rc; }
    }
    if (c->pre.valid) {  // a batch started through sppm_look_ahead: the first of the queue, if nothing else is in it
        if (c->ahead.empty()) { c->ahead.push_back(c->pre); c->pre.valid = 0; }
        else { rc = discard_prefetch(c); if (rc) return rc; }
    }
    const int depth = c->sampler == 1 ? 0 : c->ahead_depth;  // the sequential reference stream cannot run ahead
    sppm_ctx::Pre cur;
    cur.valid = 1; cur.buf = -1; cur.pass = pass; cur.first = first; cur.count = std::min(batch_size(c, c->photons_applied), count);
    for (;;) {
        // the current batch: traced ahead during an earlier batch or call, or traced now
        if (!c->ahead.empty() && c->ahead.front().pass == cur.pass && c->ahead.front().first == cur.first && c->ahead.front().count == cur.count) {
            cur.buf = c->ahead.front().buf;
            c->ahead.pop_front();
        } else {
            rc = discard_prefetch(c);
            if (rc) return rc;
            cur.buf = c->cur;
            rc = launch_ahead(c, cur.buf, cur.pass, cur.first, cur.count);
            if (rc) return rc;
        }
        // top the queue up with the batches expected next (ring buffers cur.buf + 1, + 2, ...)
        {
            sppm_ctx::Pre last = cur;
            uint64_t applied = c->photons_applied + (uint64_t)cur.count;
            for (const sppm_ctx::Pre& q : c->ahead) { last = q; applied += (uint64_t)q.count; }
            while ((int)c->ahead.size() < depth) {
                sppm_ctx::Pre nb;
                if (!batch_after(c, first, count, last, applied, &nb)) break;
                nb.buf = (cur.buf + 1 + (int)c->ahead.size()) % NBUF;
                rc = launch_ahead(c, nb.buf, nb.pass, nb.first, nb.count);
                if (rc) return rc;
                c->ahead.push_back(nb);
                last = nb; applied += (uint64_t)nb.count;
            }
        }
        // the caller is expected to go on with the next pass: its eye pass runs ahead too (eye paths do not depend on photons).
        // (Both start where the gather BEFORE this one ends and so run against the small kernels between two gather
        // kernels.  Holding the tracer, or both, back until those kernels are done was measured in round 3: the window
        // shrinks to 70 us of kernels, but the next window then waits for the eye pass, which runs 3 ms in the gather's
        // shadow instead of 0.4 ms -- 249 and 255 M photons/s against 267.)
        if (c->algo == 0 && !c->ahead.empty() && c->ahead.front().pass == pass + 1) { rc = eye_prefetch(c, pass + 1); if (rc) return rc; }
        // gather the current batch
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_traced[cur.buf], 0));
        set_cur(c, cur.buf);
        c->rec_first = cur.first; c->rec_end = cur.first + cur.count;
        rc = sppm_gather(c);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c->ev_free[cur.buf], c->stream));
        if (cur.first + cur.count >= first + count) break;
        sppm_ctx::Pre nx;
        const int64_t nf = cur.first + cur.count;
        nx.valid = 1; nx.buf = -1; nx.pass = pass; nx.first = nf; nx.count = std::min(batch_size(c, c->photons_applied), first + count - nf);
        cur = nx;
    }
    return SPPM_OK;
}

extern "C" int sppm_end_pass(sppm_ctx* c)
{
    if (!c) return fail(SPPM_ERR_INVALID, "sppm_end_pass: null context");
    CTX_DEVICE(c, "sppm_end_pass");
    if (c->cfg.update_rule != 1 || c->P == 0) return SPPM_OK;
    hipLaunchKernelGGL(k_end_pass, dim3(nblk(c->P, 256)), dim3(256), 0, c->stream, c->hp, scene_of(c), own_of(c), c->cfg.alpha, 1.0 / c->fx_scale, c->d_err);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

extern "C" int sppm_resolve(sppm_ctx* c, int32_t pass, int64_t photon_num, double* d_image)
{
    if (!c || !d_image || pass < 0 || photon_num < 1) return fail(SPPM_ERR_INVALID, "sppm_resolve: bad argument");
    CTX_DEVICE(c, "sppm_resolve");
    if (c->algo == 1)
        hipLaunchKernelGGL(k_ppm_resolve, dim3(nblk(c->npix, 256)), dim3(256), 0, c->stream, c->hp, scene_of(c), c->d_hoff, c->cfg.width, c->cfg.height, photon_num,
                           (double)(c->cfg.samples * (c->cfg.supersamples * c->cfg.supersamples)), d_image);
    else
        hipLaunchKernelGGL(k_resolve, dim3(nblk(c->P, 256)), dim3(256), 0, c->stream, c->hp, pass, photon_num, c->cfg.update_rule, d_image);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// sppm_end_pass followed by sppm_resolve; under the per-pass rule both in ONE sweep over the hitpoints
extern "C" int sppm_end_pass_resolve(sppm_ctx* c, int32_t pass, int64_t photon_num, double* d_image)
{
    if (!c || !d_image || pass < 0 || photon_num < 1) return fail(SPPM_ERR_INVALID, "sppm_end_pass_resolve: bad argument");
    CTX_DEVICE(c, "sppm_end_pass_resolve");
    if (c->cfg.update_rule != 1 || c->algo == 1 || c->P == 0) {
        int rc = sppm_end_pass(c);
        if (rc) return rc;
        return sppm_resolve(c, pass, photon_num, d_image);
    }
    hipLaunchKernelGGL(k_end_pass_resolve, dim3(nblk(c->P, 256)), dim3(256), 0, c->stream, c->hp, scene_of(c), own_of(c), c->cfg.alpha, 1.0 / c->fx_scale, c->d_err, pass, photon_num, d_image,
                       0, (int)c->P);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// The same for hitpoints [first, first + count) only (update_rule 1, SPPM): one slice of a sliced all-reduce of the
// deposits gets its update and its part of the image while the collective of the next slice is in flight.
// d_image: the WHOLE image [P][3] (rows of the range are written), or null for the update alone.
extern "C" int sppm_end_pass_resolve_range(sppm_ctx* c, int32_t pass, int64_t photon_num, double* d_image, int64_t first, int64_t count)
{
    if (!c || pass < 0 || photon_num < 1 || first < 0 || count < 0) return fail(SPPM_ERR_INVALID, "sppm_end_pass_resolve_range: bad argument");
    CTX_DEVICE(c, "sppm_end_pass_resolve_range");
    if (c->cfg.update_rule != 1 || c->algo == 1) return fail(SPPM_ERR_STATE, "sppm_end_pass_resolve_range: update_rule 1 (SPPM) only");
    if (first + count > c->P) return fail(SPPM_ERR_INVALID, "sppm_end_pass_resolve_range: range exceeds the hitpoints");
    if (count == 0) return SPPM_OK;
    hipLaunchKernelGGL(k_end_pass_resolve, dim3(nblk(count, 256)), dim3(256), 0, c->stream, c->hp, scene_of(c), own_of(c), c->cfg.alpha, 1.0 / c->fx_scale, c->d_err, pass, photon_num, d_image,
                       (int)first, (int)count);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

extern "C" int sppm_hdr_pixels(sppm_ctx* c, const double* d_rgb, int64_t n, uint8_t* d_rgbe)
{
    if (!c || n < 0 || (n > 0 && (!d_rgb || !d_rgbe))) return fail(SPPM_ERR_INVALID, "sppm_hdr_pixels: bad argument");
    CTX_DEVICE(c, "sppm_hdr_pixels");
    if (n == 0) return SPPM_OK;
    hipLaunchKernelGGL(k_hdr_pixels, dim3(nblk(n, 256)), dim3(256), 0, c->stream, d_rgb, n, d_rgbe);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// save_hdr_file, src/hdr.h:48-89: header text, then per scanline 02 02 hi lo and the four channel
// planes as literal chunks of at most 127 bytes.
extern "C" int sppm_save_hdr_file(const char* filename, const double* h_image, int32_t width, int32_t height)
{
    if (!filename || !h_image || width < 1 || height < 1) return fail(SPPM_ERR_INVALID, "sppm_save_hdr_file: bad argument");
    FILE* fp = fopen(filename, "wb");
    if (!fp) return fail(SPPM_ERR_INVALID, std::string("sppm_save_hdr_file: cannot open ") + filename);  // the reference logs and returns (hdr.h:51-55)
    fprintf(fp, "#?RADIANCE\n# Made with 100%% pure HDR Shop\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0000000000000\n\n-Y %d +X %d\n", height, width);
    std::vector<uint8_t> line((size_t)width * 4), out;
    for (int i = 0; i < height; i++) {
        for (int j = 0; j < width; j++) {
            const double* px = h_image + 3 * ((size_t)j + (size_t)i * width);
            hdr_pixel(px[0], px[1], px[2], &line[4 * (size_t)j]);
        }
        out.clear();
        out.push_back(0x02); out.push_back(0x02); out.push_back((uint8_t)((width >> 8) & 0xFF)); out.push_back((uint8_t)(width & 0xFF));
        for (int ch = 0; ch < 4; ch++)
            for (int cur = 0; cur < width;) {
                const int mv = std::min(127, width - cur);
                out.push_back((uint8_t)mv);
                for (int j = cur; j < cur + mv; j++) out.push_back(line[4 * (size_t)j + ch]);
                cur += mv;
            }
        fwrite(out.data(), 1, out.size(), fp);
    }
    fclose(fp);
    return SPPM_OK;
}

// save_ppm_file, src/ppm.h:13-29 (the reference does not check fopen; this one reports it)
extern "C" int sppm_save_ppm_file(const char* filename, const double* h_image, int32_t width, int32_t height)
{
    if (!filename || !h_image || width < 1 || height < 1) return fail(SPPM_ERR_INVALID, "sppm_save_ppm_file: bad argument");
    FILE* f = fopen(filename, "wb");
    if (!f) return fail(SPPM_ERR_INVALID, std::string("sppm_save_ppm_file: cannot open ") + filename);
    fprintf(f, "P3\n%d %d\n%d\n", width, height, 255);
    auto to_int = [](double x) {
        if (x < 0.0) x = 0.0;
        if (x > 1.0) x = 1.0;
        return int(pow(x, 1 / 2.2) * 255 + 0.5);
    };
    for (int64_t i = 0; i < (int64_t)width * height; i++)
        fprintf(f, "%d %d %d ", to_int(h_image[3 * i]), to_int(h_image[3 * i + 1]), to_int(h_image[3 * i + 2]));
    fclose(f);
    return SPPM_OK;
}

static int resolve_events(sppm_ctx* c)
{
    for (auto& e : c->ev) {
        HIPCHK(hipEventSynchronize(e.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        c->gather_ns += (uint64_t)((double)ms * 1e6);
        c->gather_launches++;
        hipEventDestroy(e.first); hipEventDestroy(e.second);
    }
    c->ev.clear();
    for (auto& e : c->ev_trace) {
        HIPCHK(hipEventSynchronize(e.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
        c->trace_ns += (uint64_t)((double)ms * 1e6);
        c->trace_launches++;
        hipEventDestroy(e.first); hipEventDestroy(e.second);
    }
    c->ev_trace.clear();
    return 0;
}

extern "C" int sppm_check(sppm_ctx* c)
{
    if (!c) return fail(SPPM_ERR_INVALID, "sppm_check: null context");
    CTX_DEVICE(c, "sppm_check");
    int32_t err[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < NTS; b++) HIPCHK(hipStreamSynchronize(c->tstream[b]));
    HIPCHK(hipMemcpyAsync(err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = resolve_events(c);
    if (rc) return rc;
    if (err[0]) return fail(SPPM_ERR_OVERFLOW, "record buffer overflow: raise SPPM_RECORD_CAP or lower SPPM_BATCH_PHOTONS");
    if (err[1]) return fail(SPPM_ERR_OVERFLOW, "a photon path exceeded 524287 bounces (the depth field of the record key)");
    if (err[2]) return fail(SPPM_ERR_OVERFLOW, "a photon hit had more than gather_max candidate hitpoints: the reference's K cap would have acted (check_gather_max = 1 only counts; 0 enforces the cap; sppm_counters [14] has the largest count)");
    if (err[3]) return fail(SPPM_ERR_STATE, "a record's photon index lies outside the range declared for the batch");
    if (err[4]) return fail(SPPM_ERR_OVERFLOW, "reference-stream mode: a path ran past the uploaded part of the stream");
    if (err[5]) return fail(SPPM_ERR_STATE, "reference-stream mode: a pixel consumed a different number of draws than the offset walk found");
    if (err[7]) return fail(SPPM_ERR_OVERFLOW, "K cap: the walk of the hitpoint tree ran out of stack");
    if (err[6]) return fail(SPPM_ERR_OVERFLOW, "per-pass rule: a deposit left the fixed-point range (a photon flux above 8192 x the emitted flux, or 2^21 deposits on one hitpoint in one pass)");
    return SPPM_OK;
}

extern "C" int sppm_download_hitpoints(sppm_ctx* c, int32_t* h_valid, int32_t* h_obj, double* h_hp_f, double* h_r2, int32_t* h_n,
                                       double* h_flux, double* h_ld)
{
    if (!c) return fail(SPPM_ERR_INVALID, "sppm_download_hitpoints: null context");
    const size_t P = (size_t)c->P;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<int32_t> valid(P), obj(P);
    HIPCHK(hipMemcpy(valid.data(), c->hp.valid, P * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(obj.data(), c->hp.obj, P * 4, hipMemcpyDeviceToHost));
    if (h_valid) memcpy(h_valid, valid.data(), P * 4);
    if (h_obj) for (size_t i = 0; i < P; i++) h_obj[i] = valid[i] ? obj[i] : -1;
    if (h_hp_f) {
        std::vector<double> pos(3 * P), nrm(3 * P), w(3 * P), dist(P);
        HIPCHK(hipMemcpy(pos.data(), c->hp.pos, 3 * P * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(nrm.data(), c->hp.nrm, 3 * P * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(w.data(), c->hp.w, 3 * P * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(dist.data(), c->hp.dist, P * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < P; i++) {
            double* f = h_hp_f + 10 * i;
            f[0] = dist[i];
            for (int k = 0; k < 3; k++) { f[1 + k] = pos[k * P + i]; f[4 + k] = nrm[k * P + i]; f[7 + k] = w[k * P + i]; }
        }
    }
    // radius, count and flux of a pixel that has no hitpoint (yet) are reported as zero: the eye pass of the NEXT pass,
    // running ahead, may already have initialised them for a pixel that becomes valid then
    if (h_r2) { HIPCHK(hipMemcpy(h_r2, c->hp.r2, P * 8, hipMemcpyDeviceToHost)); for (size_t i = 0; i < P; i++) if (!valid[i]) h_r2[i] = 0.0; }
    if (h_n) {
        if (c->cfg.update_rule == 0) HIPCHK(hipMemcpy(h_n, c->hp.n, P * 4, hipMemcpyDeviceToHost));
        else {
            std::vector<double> nd(P);
            HIPCHK(hipMemcpy(nd.data(), c->hp.nd, P * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < P; i++) h_n[i] = (int32_t)nd[i];
        }
        for (size_t i = 0; i < P; i++) if (!valid[i]) h_n[i] = 0;
    }
    auto soa3 = [&](const double* d, double* h) -> int {
        std::vector<double> t(3 * P);
        HIPCHK(hipMemcpy(t.data(), d, 3 * P * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < P; i++) for (int k = 0; k < 3; k++) h[3 * i + k] = t[k * P + i];
        return 0;
    };
    if (h_flux) {
        int rc = soa3(c->cfg.update_rule == 0 ? c->hp.flux : c->hp.tau, h_flux);
        if (rc) return rc;
        for (size_t i = 0; i < P; i++) if (!valid[i]) h_flux[3 * i] = h_flux[3 * i + 1] = h_flux[3 * i + 2] = 0.0;
    }
    if (h_ld) { int rc = soa3(c->hp.ld, h_ld); if (rc) return rc; }
    return SPPM_OK;
}

extern "C" int sppm_hitpoint_count(sppm_ctx* c, int64_t* n)
{
    if (!c || !n) return fail(SPPM_ERR_INVALID, "sppm_hitpoint_count: null argument");
    *n = c->P;
    return SPPM_OK;
}

extern "C" int sppm_download_hitpoint_pixels(sppm_ctx* c, int32_t* h_pix)
{
    if (!c || !h_pix) return fail(SPPM_ERR_INVALID, "sppm_download_hitpoint_pixels: null argument");
    if (c->algo != 1) return fail(SPPM_ERR_STATE, "sppm_download_hitpoint_pixels: algorithm 0 keeps one hitpoint per image index");
    const int W = c->cfg.width, H = c->cfg.height;
    for (int t = 0; t < c->npix; t++) {
        const int32_t index = (H - t / W - 1) * W + t % W;
        for (long long h = c->h_hoff[(size_t)t]; h < c->h_hoff[(size_t)t + 1]; h++) h_pix[h] = index;
    }
    return SPPM_OK;
}

extern "C" int sppm_download_records(sppm_ctx* c, uint64_t* h_key, double* h_f, int64_t* n)
{
    if (!c || !n) return fail(SPPM_ERR_INVALID, "sppm_download_records: null argument");
    int64_t cnt = 0;
    int rc = slots_in_use(c, &cnt);
    if (rc) return rc;
    std::vector<uint64_t> key((size_t)cnt);
    std::vector<double> f(9 * (size_t)cnt);
    HIPCHK(hipMemcpy(key.data(), c->rec_u.key, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < 9; k++) HIPCHK(hipMemcpy(f.data() + (size_t)k * cnt, c->rec_u.f + (size_t)k * c->rec_u.cap, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    std::vector<int64_t> order;
    for (int64_t i = 0; i < cnt; i++) if (key[i] != REC_INVALID) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
    *n = (int64_t)order.size();
    if (!h_key && !h_f) return SPPM_OK;
    for (int64_t i = 0; i < (int64_t)order.size(); i++) {
        if (h_key) h_key[i] = key[order[i]];
        if (h_f) for (int k = 0; k < 9; k++) h_f[9 * i + k] = f[(size_t)k * cnt + order[i]];
    }
    return SPPM_OK;
}

extern "C" int sppm_owned_mask(sppm_ctx* c, uint8_t* d_mask)
{
    if (!c || !d_mask) return fail(SPPM_ERR_INVALID, "sppm_owned_mask: null argument");
    CTX_DEVICE(c, "sppm_owned_mask");
    if (c->own_pending) return fail(SPPM_ERR_STATE, "sppm_owned_mask: spatial ownership is decided by the first eye pass");
    HIPCHK(hipMemcpyAsync(d_mask, c->d_own_mask, (size_t)c->P, hipMemcpyDeviceToDevice, c->stream));
    return SPPM_OK;
}

extern "C" int sppm_export_owned_state(sppm_ctx* c, double* d_out)
{
    if (!c || !d_out) return fail(SPPM_ERR_INVALID, "sppm_export_owned_state: null argument");
    CTX_DEVICE(c, "sppm_export_owned_state");
    if (c->cfg.update_rule != 0) return fail(SPPM_ERR_STATE, "sppm_export_owned_state: update_rule 0 only");
    if (c->P == 0) return SPPM_OK;
    hipLaunchKernelGGL(k_export_state, dim3(nblk(c->P, 256)), dim3(256), 0, c->stream, c->hp, own_of(c), d_out);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

extern "C" int sppm_import_state(sppm_ctx* c, const double* d_in)
{
    if (!c || !d_in) return fail(SPPM_ERR_INVALID, "sppm_import_state: null argument");
    CTX_DEVICE(c, "sppm_import_state");
    if (c->cfg.update_rule != 0) return fail(SPPM_ERR_STATE, "sppm_import_state: update_rule 0 only");
    if (c->P == 0) return SPPM_OK;
    hipLaunchKernelGGL(k_import_state, dim3(nblk(c->P, 256)), dim3(256), 0, c->stream, c->hp, d_in);
    HIPCHK(hipGetLastError());
    return SPPM_OK;
}

// The per-pass rule between ranks: the pass's deposits, dep[4][P] 64-bit integers (fixed-point flux x, y, z and the
// count M).  Every rank deposits the photons it traced into its own copy; the sum over the ranks -- integers, so
// exact and independent of the number of ranks -- is stored back before sppm_end_pass.
extern "C" int sppm_export_deposits(sppm_ctx* c, int64_t* d_out)
{
    if (!c || !d_out) return fail(SPPM_ERR_INVALID, "sppm_export_deposits: null argument");
    CTX_DEVICE(c, "sppm_export_deposits");
    if (c->cfg.update_rule != 1) return fail(SPPM_ERR_STATE, "sppm_export_deposits: update_rule 1 only");
    if (c->P > 0) HIPCHK(hipMemcpyAsync(d_out, c->hp.dep, 4 * (size_t)c->P * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    return SPPM_OK;
}

// hitpoints [first, first + count) only: d_out / d_in are [4][count]
extern "C" int sppm_export_deposits_range(sppm_ctx* c, int64_t* d_out, int64_t first, int64_t count)
{
    if (!c || (!d_out && count > 0) || first < 0 || count < 0) return fail(SPPM_ERR_INVALID, "sppm_export_deposits_range: bad argument");
    CTX_DEVICE(c, "sppm_export_deposits_range");
    if (c->cfg.update_rule != 1) return fail(SPPM_ERR_STATE, "sppm_export_deposits_range: update_rule 1 only");
    if (first + count > c->P) return fail(SPPM_ERR_INVALID, "sppm_export_deposits_range: range exceeds the hitpoints");
    if (count > 0) HIPCHK(hipMemcpy2DAsync(d_out, (size_t)count * 8, c->hp.dep + first, (size_t)c->P * 8, (size_t)count * 8, 4, hipMemcpyDeviceToDevice, c->stream));
    return SPPM_OK;
}
extern "C" int sppm_import_deposits_range(sppm_ctx* c, const int64_t* d_in, int64_t first, int64_t count)
{
    if (!c || (!d_in && count > 0) || first < 0 || count < 0) return fail(SPPM_ERR_INVALID, "sppm_import_deposits_range: bad argument");
    CTX_DEVICE(c, "sppm_import_deposits_range");
    if (c->cfg.update_rule != 1) return fail(SPPM_ERR_STATE, "sppm_import_deposits_range: update_rule 1 only");
    if (first + count > c->P) return fail(SPPM_ERR_INVALID, "sppm_import_deposits_range: range exceeds the hitpoints");
    if (count > 0) HIPCHK(hipMemcpy2DAsync(c->hp.dep + first, (size_t)c->P * 8, d_in, (size_t)count * 8, (size_t)count * 8, 4, hipMemcpyDeviceToDevice, c->stream));
    return SPPM_OK;
}

extern "C" int sppm_import_deposits(sppm_ctx* c, const int64_t* d_in)
{
    if (!c || !d_in) return fail(SPPM_ERR_INVALID, "sppm_import_deposits: null argument");
    CTX_DEVICE(c, "sppm_import_deposits");
    if (c->cfg.update_rule != 1) return fail(SPPM_ERR_STATE, "sppm_import_deposits: update_rule 1 only");
    if (c->P > 0) HIPCHK(hipMemcpyAsync(c->hp.dep, d_in, 4 * (size_t)c->P * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    return SPPM_OK;
}

extern "C" int sppm_fixed_point_scale(sppm_ctx* c, double* scale)
{
    if (!c || !scale) return fail(SPPM_ERR_INVALID, "sppm_fixed_point_scale: null argument");
    *scale = c->fx_scale;
    return SPPM_OK;
}

extern "C" int sppm_download_tile_order(sppm_ctx* c, int32_t* h_order, int64_t* n)
{
    if (!c || !n) return fail(SPPM_ERR_INVALID, "sppm_download_tile_order: null argument");
    *n = c->n_own_pix;
    if (!h_order) return SPPM_OK;
    if (!c->order_valid) return fail(SPPM_ERR_STATE, "sppm_download_tile_order: no eye pass has run yet");
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(h_order, c->d_order, (size_t)c->n_own_pix * 4, hipMemcpyDeviceToHost));
    return SPPM_OK;
}

extern "C" int sppm_accept_tap(sppm_ctx* c, int64_t cap)
{
    if (!c || cap < 0) return fail(SPPM_ERR_INVALID, "sppm_accept_tap: bad argument");
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->d_tap_key) { hipFree(c->d_tap_key); hipFree(c->d_tap_pix); hipFree(c->d_tap_n); c->d_tap_key = nullptr; c->d_tap_pix = nullptr; c->d_tap_n = nullptr; }
    c->tap_cap = cap;
    if (cap > 0) {
        HIPCHK(hipMalloc((void**)&c->d_tap_key, (size_t)cap * 8));
        HIPCHK(hipMalloc((void**)&c->d_tap_pix, (size_t)cap * 4));
        HIPCHK(hipMalloc((void**)&c->d_tap_n, 8));
        HIPCHK(hipMemset(c->d_tap_n, 0, 8));
    }
    return SPPM_OK;
}

extern "C" int sppm_download_accepted(sppm_ctx* c, uint64_t* h_key, int32_t* h_pix, int64_t* n)
{
    if (!c || !n) return fail(SPPM_ERR_INVALID, "sppm_download_accepted: null argument");
    if (c->tap_cap <= 0) return fail(SPPM_ERR_STATE, "sppm_download_accepted: the tap is not enabled");
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpy(&cnt, c->d_tap_n, 8, hipMemcpyDeviceToHost));
    *n = (int64_t)cnt;
    if ((int64_t)cnt > c->tap_cap) return fail(SPPM_ERR_OVERFLOW, "accept tap overflow");
    if (!h_key || !h_pix) return SPPM_OK;
    std::vector<uint64_t> key((size_t)cnt);
    std::vector<int32_t> pix((size_t)cnt);
    HIPCHK(hipMemcpy(key.data(), c->d_tap_key, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pix.data(), c->d_tap_pix, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> order((size_t)cnt);
    for (int64_t i = 0; i < (int64_t)cnt; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key[a] != key[b] ? key[a] < key[b] : pix[a] < pix[b]; });
    for (int64_t i = 0; i < (int64_t)cnt; i++) { h_key[i] = key[order[i]]; h_pix[i] = pix[order[i]]; }
    return SPPM_OK;
}

extern "C" int sppm_counters(sppm_ctx* c, uint64_t h_out[16])
{
    if (!c || !h_out) return fail(SPPM_ERR_INVALID, "sppm_counters: null argument");
    unsigned long long d[8], qmax = 0;
    for (int b = 0; b < NTS; b++) HIPCHK(hipStreamSynchronize(c->tstream[b]));
    HIPCHK(hipMemcpyAsync(d, c->d_counters, sizeof d, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&qmax, c->d_qmax, sizeof qmax, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = resolve_events(c);
    if (rc) return rc;
    h_out[0] = d[0] + d[1]; h_out[1] = c->photons_traced; h_out[2] = c->records_total + d[2]; h_out[3] = d[3]; h_out[4] = d[4];
    h_out[5] = c->gather_batches; h_out[6] = c->gather_ns; h_out[7] = c->gather_launches;
    h_out[8] = d[5]; h_out[9] = d[6]; h_out[10] = d[7]; h_out[11] = c->trace_ns; h_out[12] = c->trace_launches; h_out[13] = d[1];
    h_out[14] = c->cfg.check_gather_max == 1 ? qmax : c->cap_capped; h_out[15] = c->cap_vetoes;
    return SPPM_OK;
}

extern "C" int sppm_set_timing(sppm_ctx* c, int32_t on)
{
    if (!c) return fail(SPPM_ERR_INVALID, "sppm_set_timing: null context");
    c->timing = on ? 1 : 0;
    return SPPM_OK;
}

// ================================ the entry point that allocates nothing ============================
// sppm_intersect_static: sppm_intersect for callers that cannot allocate (code under HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc).  sppm_intersect reads the scene from device memory that
// sppm_create allocated; here the sphere table and the hit counter live in a file-scope device array of fixed size.
// The table reaches it as KERNEL ARGUMENTS (chunks of SCRATCH_CHUNK spheres by value), so there is no host-sourced
// copy either: the call is a memset of the counter block and a few kernel launches on the caller's stream.
#define SPPM_STATIC_MAX_SPHERES 512
#define SCRATCH_HEAD 256  // bytes: [0] rays that hit (unsigned long long), rest reserved
#define SCRATCH_CHUNK 40  // spheres per upload launch (3520 bytes of kernel arguments)
static __device__ __align__(256) unsigned char scratch[SCRATCH_HEAD + SPPM_STATIC_MAX_SPHERES * sizeof(sppm_sphere)];
// the array's address on each device (every GPU holds its own copy): found once per device with that device
// current; host code uses these pointers, never the array's name
#define SPPM_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[SPPM_MAX_DEVICES];
static std::mutex g_scratch_mutex;

struct SphereChunk { sppm_sphere s[SCRATCH_CHUNK]; };
__global__ void k_static_put_spheres(SphereChunk chunk, int first, int count, unsigned char* __restrict__ base)
{
    double* dst = (double*)(base + SCRATCH_HEAD) + (size_t)first * 11;
    const double* src = (const double*)chunk.s;
    for (int i = threadIdx.x; i < count * 11; i += blockDim.x) dst[i] = src[i];
}
// k_intersect over the table in the scratch array, plus the number of rays that hit (one atomic per wavefront)
__global__ void __launch_bounds__(256) k_static_intersect(unsigned char* __restrict__ base, int ns, const double* __restrict__ rays, int64_t n,
                                                          int32_t* __restrict__ id, double* __restrict__ out)
{
    DevScene sc;
    sc.sp = (const sppm_sphere*)(base + SCRATCH_HEAD); sc.ns = ns; sc.n_cells = 0;
    STAGE_SCENE_T(sc, 0);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < n) {
        Hit h;
        hit = intersect_scene<0>(sc, ld3(rays + 6 * i), ld3(rays + 6 * i + 3), h);
        id[i] = hit ? h.object_id : -1;
        double* o = out + 7 * i;
        o[0] = h.distance;
        if (hit) { o[1] = h.position.x; o[2] = h.position.y; o[3] = h.position.z; o[4] = h.normal.x; o[5] = h.normal.y; o[6] = h.normal.z; }
        else { o[1] = o[2] = o[3] = o[4] = o[5] = o[6] = 0.0; }
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd((unsigned long long*)base, (unsigned long long)__popcll(m));
}

static int static_scratch(unsigned char** out)
{
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= SPPM_MAX_DEVICES) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: device index beyond " + std::to_string(SPPM_MAX_DEVICES - 1));
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        unsigned char* ptr = nullptr;
        HIPCHK(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));  // this device's copy of the array
        if (!ptr) return fail(SPPM_ERR_HIP, "sppm_intersect_static: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int sppm_intersect_static(const sppm_sphere* h_spheres, int32_t n_spheres, const double* d_rays, int64_t n, int32_t* d_id, double* d_hit,
                                     uint64_t* d_n_hit, void* stream)
{
    if (!h_spheres || n_spheres < 1 || n < 0 || (n > 0 && (!d_rays || !d_id || !d_hit))) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: bad argument");
    if (n_spheres > SPPM_STATIC_MAX_SPHERES)
        return fail(SPPM_ERR_INVALID, "sppm_intersect_static: at most " + std::to_string(SPPM_STATIC_MAX_SPHERES) + " spheres fit the static table (got " +
                                          std::to_string(n_spheres) + "); larger scenes go through sppm_create + sppm_intersect");
    if (n > (1ll << 40)) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: more than 2^40 rays");
    hipStream_t st = (hipStream_t)stream;
    if (st) {  // the array used is the CURRENT device's copy: a stream of another GPU would be handed the wrong address
        int cur = -1;
        hipDevice_t sdev = -1;
        HIPCHK(hipGetDevice(&cur));
        HIPCHK(hipStreamGetDevice(st, &sdev));
        if ((int)sdev != cur) return fail(SPPM_ERR_INVALID, "sppm_intersect_static: the stream belongs to device " + std::to_string((int)sdev) + ", the current device is " + std::to_string(cur));
    }
    unsigned char* ptr = nullptr;
    int rc = static_scratch(&ptr);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(ptr, 0, SCRATCH_HEAD, st));  // the array keeps the last call's contents: the counter must start at zero
    for (int first = 0; first < n_spheres; first += SCRATCH_CHUNK) {
        SphereChunk chunk;
        const int count = std::min(SCRATCH_CHUNK, n_spheres - first);
        memset(&chunk, 0, sizeof chunk);
        memcpy(chunk.s, h_spheres + first, sizeof(sppm_sphere) * (size_t)count);
        hipLaunchKernelGGL(k_static_put_spheres, dim3(1), dim3(256), 0, st, chunk, first, count, ptr);
    }
    HIPCHK(hipGetLastError());
    if (n > 0) {
        hipLaunchKernelGGL(k_static_intersect, dim3(nblk(n, 256)), dim3(256), (size_t)n_spheres * sizeof(sppm_sphere), st, ptr, n_spheres, d_rays, n, d_id, d_hit);
        HIPCHK(hipGetLastError());
    }
    if (d_n_hit) HIPCHK(hipMemcpyAsync(d_n_hit, ptr, sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    return SPPM_OK;
}

// ================================ measured peak (bench.py) ===========================================
// No counterpart in the reference.  The FP64 vector FMA rate this device sustains, measured beside the data-sheet
// figure (SURVEY.md 8d asks for both denominators): every thread runs 8 independent chains of fused multiply-adds.
__global__ void __launch_bounds__(256) k_fma_peak(double* __restrict__ out, int iters, double a, double b)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
    const double s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if (s == 12345.678) out[0] = s;  // keeps the chains alive; never true for the arguments used
}
extern "C" int sppm_measure_fma_peak(void* stream, double* h_tflops)
{
    if (!h_tflops) return fail(SPPM_ERR_INVALID, "sppm_measure_fma_peak: null argument");
    hipStream_t st = (hipStream_t)stream;
    double* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(double)));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const int blocks = 256 * 32, iters = 4096;
    hipLaunchKernelGGL(k_fma_peak, dim3(blocks), dim3(256), 0, st, d, 64, 0.999999, 1e-7);  // warm-up
    HIPCHK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_fma_peak, dim3(blocks), dim3(256), 0, st, d, iters, 0.999999, 1e-7);
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *h_tflops = 2.0 * 8.0 * (double)iters * 256.0 * (double)blocks / ((double)ms * 1e-3) / 1e12;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    return SPPM_OK;
}
