// sppm_host_ctx.hpp -- the context, error reporting and small host helpers.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
    // spatial order of this rank's hitpoints (tile gather): own pixel list, Hilbert keys, sorted order
    int32_t *d_own_pix, *d_order; int n_own_pix;
    uint8_t* d_own_mask;   // [P]
    int own_pending;       // spatial ownership: the pixel list is cut out of the first eye pass's Hilbert order
    hp_key_t *d_hp_keys, *d_hp_keys_out; unsigned long long* d_hp_bb;
    void* d_sort_tmp; size_t sort_tmp_bytes;
    uint32_t *d_est, *d_items; unsigned long long* d_items_raw; unsigned int* d_item_q; uint8_t* d_n_pieces; unsigned long long* d_est_sum; uint8_t* d_ext_bin; int est_valid;  // d_item_q: [0] items [1] queue head [2..33] per cost class
    double split_factor, est_per_photon; int64_t est_range;
    int64_t hist_est_range, stat_est_range[2];  // photons of the batch whose estimates the statistics (and each pinned slot) hold
    int64_t batch_min; double batch_growth;  // per-photon rule: first batch and growth of the batches with the photons applied (batch_size)
    double fx_scale;          // per-pass rule: fixed point of the deposits (sppm_deposit.hpp), a power of two
    double cell_factor_dep;   // per-pass rule: cell size of the record grid as a share of the median tile extent
    int64_t dep_chunk;        // per-pass rule: records per work item
    double dep_light, dep_heavy;  // per-pass rule: pairs per photon at which the deposit kernel runs on dep_min / 768 workgroups (deposit_tiles)
    int dep_min;
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
    // only to the ranks whose map holds its cell.  d_all_pix: the first eye pass's whole Hilbert order (rank r owns
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

// The sort of the hitpoints' Hilbert keys runs beside the gather, whose persistent wavefronts leave one wavefront slot of
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
