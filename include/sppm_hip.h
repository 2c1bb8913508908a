/* sppm_hip.h -- C ABI of libsppm_hip.so: the MI355X (gfx950) implementation of the reference's
 * stochastic progressive photon mapping path.
 *
 * The reference (0V/raytracer) has no plugin or FFI interface; its boundary is source-level C++
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference root).  include/sppm2_hip.hpp keeps the reference's C++
 * names (edupt::Sphere, photonmap::sppm2::StochasticPpm<W,H>::render, save_hdr_file ...) on top of
 * this ABI; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only.  Pointers named d_* are DEVICE pointers (HBM, the
 * current HIP device of the context), h_* are host pointers.  Every function returns 0 on success
 * and a negative sppm_status otherwise; sppm_last_error() gives the text.  A context is bound to
 * the HIP device that was current when it was created and to one HIP stream; calls on one context
 * must not overlap.  Nothing here falls back to a CPU path: without a GPU sppm_create fails.
 *
 * Arithmetic is IEEE double throughout (the reference is double end to end, src/vec.h:9; FP32
 * cannot reproduce first-hit ids on the radius-1e5 wall spheres), built with
 * -ffp-contract=off, with cos/sin/pow5 from raytracer_amd/csrc/detmath.h, so results are
 * bit-identical to oracle/sppm_oracle.c in its (sampler=XorShift, math=detmath) mode.
 */
#ifndef SPPM_HIP_H
#define SPPM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sppm_status {
    SPPM_OK = 0,
    SPPM_ERR_INVALID = -1,   /* bad argument */
    SPPM_ERR_HIP = -2,       /* a HIP call failed (no device, out of memory, launch failure) */
    SPPM_ERR_OVERFLOW = -3,  /* a device buffer was too small; nothing was dropped silently */
    SPPM_ERR_STATE = -4      /* calls made in an order the path does not allow */
} sppm_status;

/* edupt::ReflectionType, src/material.h:10-15 */
enum { SPPM_DIFFUSE = 0, SPPM_SPECULAR = 1, SPPM_REFRACTION = 2 };

/* edupt::Sphere, src/sphere.h:15-27 (88 bytes, same field order) */
typedef struct sppm_sphere {
    double radius;
    double position[3];
    double emission[3];
    double color[3];
    int32_t reflection_type;
    int32_t pad_;
} sppm_sphere;

/* Everything StochasticPpm<W,H> holds as members or takes in render():
 * src/sppm2.h:38-64 (tunables, camera), :562-564 (render arguments). */
typedef struct sppm_config {
    int32_t width, height, samples, supersamples;
    double camera_position[3], camera_dir[3], camera_up[3]; /* camera_dir already normalised, as src/sppm2.h:49 passes it */
    double screen_height, screen_dist;                       /* src/sppm2.h:54-56 */
    double alpha;            /* src/sppm2.h:39 */
    double initial_radius2;  /* src/sppm2.h:44 */
    double gather_radius2;   /* render()'s gather_photon_radius; a SQUARED bound (src/kdtree.h:57,126) */
    int64_t gather_max;      /* render()'s gahter_max_photon_num; the kd search's K cap (src/kdtree.h:58,126-133).
                                Enforced exactly (check_gather_max = 0, the default): whenever K is below the number of
                                hitpoints, every photon hit that can see more than K hitpoints walks the reference's
                                tree -- built on the host with the reference's std::sort from the reference's vector
                                order, uploaded per eye pass -- with the reference's heap, and the pairs the cap drops
                                are withheld from the gather (raytracer_amd/csrc/sppm_cap.hpp).  Both update rules
                                (src/stochastic_ppm.h:425-427 takes the same K) and classic PPM; ~0.8 s per pass at
                                1080p with K = 10000 (one heap of K + 1 entries per such photon hit: up to a quarter
                                of the free device memory), nothing when no photon hit can see more than K
                                hitpoints.  See check_gather_max. */
    int32_t update_rule;     /* 0: per-photon shrink, src/sppm2.h:391-421 (default, the live SPPM)
                                1: per-pass (N, tau) update, src/stochastic_ppm.h:721-741: photons deposit into
                                   per-hitpoint accumulators (:439-456) kept in 64-bit fixed point (exact integer
                                   sums: independent of the order and of the number of GPUs), sppm_end_pass applies
                                   the update.  Over several GPUs every rank keeps all hitpoints, deposits the
                                   photons it traced, and the accumulators are summed with ONE all-reduce per pass
                                   (sppm_export_deposits / sppm_import_deposits). */
    int32_t double_nee;      /* 1 reproduces the reference's first-sight double direct sample
                                (`break` leaves the switch, src/sppm2.h:204-227); 0 switches it off */
    uint32_t seed_eye, seed_photon;
    int64_t photons_per_pass; /* photons of one pass over ALL ranks; keys the per-photon XorShift
                                 streams (key = pass*photons_per_pass + i), so a photon's path does
                                 not depend on which GPU traces it */
    int32_t tile_first, tile_stride; /* this rank gathers for the 8x8 pixel tiles tile_first + k*tile_stride
                                        (tiles numbered row-major over the image-index grid); 0,0 = all tiles.
                                        The eye pass and tracing always cover what the calls ask for. */
    int32_t check_gather_max; /* 0: enforce gather_max (above).  1: do not enforce; count each query's candidates by brute
                                 force over all hitpoints (O(records x hitpoints), validation only) and let sppm_check()
                                 report SPPM_ERR_OVERFLOW if any exceeded K (sppm_counters [14] = the largest count).
                                 2: ignore gather_max -- the uncapped set {d2 < R^2, |dt| <= 0.01 R^2}; a deviation from
                                 the reference wherever a query has more than K candidates, chosen explicitly (bench.py). */
    int32_t sampler;         /* 0: one edupt::XorShift stream (src/random.h:9-33) per (pass, pixel) and per photon --
                                the parallel, GPU-count independent definition (default).
                                1: the reference's own sampler (src/sampler/value_sampler.h:8-15): ONE sequential
                                std::mt19937 stream for all eye passes seeded seed_eye (src/sppm2.h:621) and one per
                                photon pass seeded seed_photon + pass (src/sppm2.h:530; the reference takes both
                                seeds from std::random_device).  Reproduces the reference's image for matched
                                seeds; single GPU, whole passes only, for validation rather than speed. */
    int32_t ownership;       /* which hitpoints this rank gathers for, when tile_stride > 1:
                                0: the 8x8 pixel tiles tile_first + k*tile_stride (fixed, known without an eye pass);
                                1: positions [tile_first*P/tile_stride, (tile_first+1)*P/tile_stride) of the Hilbert
                                   order of the FIRST eye pass's hitpoints -- a compact region of space per rank, the
                                   same partition on every rank without an exchange (the eye pass is replicated).
                                   Tiles stay as dense as on one GPU and records far from the rank's hitpoints are
                                   dropped when a batch is binned.  Results are identical either way. */
    int32_t algorithm;       /* 0: stochastic progressive photon mapping, StochasticPpm<W,H>::render (src/sppm2.h) -- default.
                                1: classic progressive photon mapping, photonmap::progressive::render
                                   (src/progressive_photonmapping.h:709-821; what src/main.cpp:87 runs): ONE eye pass
                                   whose recursion (create_point, :496-615) leaves a LIST of hitpoints, several per
                                   pixel; photons deposit from their first diffuse hit on (:326,339,344) with the
                                   photon surface's colour and no hitpoint weight (:385-387); the image sums every
                                   pixel's hitpoints (:807-813).  sppm_create counts the list (sppm_hitpoint_count);
                                   then sppm_eye_pass(ctx, 0), sppm_photon_pass(ctx, 0, first, count) for any split of
                                   the photons [0, photons_per_pass), sppm_resolve(ctx, 0, photon_num, image).
                                   initial_radius2 is progressive::InitialRadius (:32, set by src/main.cpp:77).
                                   update_rule 0, one GPU.  With sampler 1 the eye pass reads the reference's one
                                   sequential stream (:760) and the photons its second one (:619). */
    int32_t camera_model;    /* 0: PinholeCamera (src/cameras.h:11-64), what src/sppm2.h:81-86 constructs -- default.
                                1: thin lens, DoFCamera::get_ray (src/cameras.h:66-139) with PointSamplerDisk
                                   (src/sampler/point_sampler_disk.h:28-71): the alternative member src/sppm2.h:73-79
                                   keeps commented out (aperture 4, focus distance 180).  Algorithm 0 only. */
    uint32_t seed_lens;      /* the lens draws come from a generator of their own (StochasticPpm::engine_,
                                src/sppm2.h:70-71): sampler 0 keys one XorShift stream per (pass, pixel) with it,
                                sampler 1 seeds the one sequential mt19937 with it */
    double aperture;         /* lens diameter (lens_radius = aperture * 0.5, src/cameras.h:117) */
    double focus_dist;       /* distance of the plane in focus along camera_dir (src/cameras.h:133) */
    int32_t tracer_variant;  /* 0: create_photon of the live src/sppm2.h:317-525 -- default, under either update rule.
                                1: create_photon of src/stochastic_ppm.h:371-572 AS SHIPPED: no two discarded draws
                                   (:377 against src/sppm2.h:322-324) and every diffuse bounce samples about the light's
                                   emission basis (:482-483 use u, v, w of :382-393 where diffuse_u/v/w were meant).
                                   For reproducing runs of that compiled file (tests/golden/ref_pp_*.npz); sampler 1 only. */
    int32_t pad2_;
} sppm_config;

typedef struct sppm_ctx sppm_ctx;

const char* sppm_last_error(void);
int sppm_device_count(void);

/* Replaces constructing StochasticPpm<W,H>(samples, supersamples) (src/sppm2.h:90-97) plus the
 * allocations at the top of render() (src/sppm2.h:581,615-619).  `stream` is a hipStream_t (NULL =
 * the default stream).  The scene is copied to the device (the reference compiles it in,
 * src/scene.h:12-24); sphere 0 is the light (LightID, src/sppm2.h:132).  Up to 512 spheres the closest
 * hit is the reference's loop over all spheres (src/scene.h:53-76); above that (at most 2^20) it walks a
 * uniform grid over the small spheres and returns the same sphere and distance, bit for bit. */
int sppm_create(const sppm_config* cfg, const sppm_sphere* h_spheres, int32_t n_spheres, void* stream, sppm_ctx** out);
int sppm_destroy(sppm_ctx* ctx);

/* edupt::PinholeCamera::get_ray for every pixel (src/cameras.h:52-64): d_rays[(y*W+x)*6 + 0..5] =
 * origin xyz, direction xyz. */
int sppm_camera_rays(sppm_ctx* ctx, int32_t supersample_x, int32_t supersample_y, double* d_rays);

/* edupt::intersect_scene over a batch of rays (src/scene.h:53-76 calling src/sphere.h:31-56).
 * d_rays[n][6]; d_id[n] = object id or -1; d_hit[n][7] = distance, position xyz, normal xyz
 * (distance = 1e128 and zeros on a miss). */
int sppm_intersect(sppm_ctx* ctx, const double* d_rays, int64_t n, int32_t* d_id, double* d_hit);

/* NO COUNTERPART IN THE REFERENCE.  sppm_intersect for callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same closest hit over the same ray batch, but without a
 * context: the sphere table (h_spheres, a HOST array read during the call; at most 512 spheres) and a hit counter
 * live in one file-scope device array of the library (45 312 bytes per GPU, held until the process ends), the
 * table travels as kernel arguments, and the call consists of one memset and a few kernel launches on `stream`
 * (a hipStream_t, NULL = the default stream): no hipMalloc / hipFree / host-sourced copy.  d_rays[n][6],
 * d_id[n], d_hit[n][7] as for sppm_intersect; d_n_hit (device, may be NULL) receives the number of rays that hit.
 * More than 512 spheres: SPPM_ERR_INVALID (such scenes need sppm_create, which builds a grid).  Every call on a
 * GPU shares that GPU's array: calls that overlap on one GPU (different streams or threads) race -- serialise them.
 * The array is the CURRENT device's: `stream` must belong to it (checked).  The first call on a device (or sppm_create
 * on it) looks the array's address up; make that call outside stream capture, later calls can be captured. */
int sppm_intersect_static(const sppm_sphere* h_spheres, int32_t n_spheres, const double* d_rays, int64_t n, int32_t* d_id,
                          double* d_hit, uint64_t* d_n_hit, void* stream);

/* One eye pass: the y/x/sy/sx/s loops of render() calling create_point_loop (src/sppm2.h:625-646,
 * :156-315) including next_event_est (src/radiance.h:29-70).  One XorShift stream per
 * (pass, image index).  Replaces CreateKDtree too (src/sppm2.h:649): nothing is built over the
 * hitpoints, the photon records are gridded instead. */
int sppm_eye_pass(sppm_ctx* ctx, int32_t pass);

/* Photon pass = update_photons (src/sppm2.h:527-558): photons [first, first+count) of this pass are
 * emitted and traced (create_photon, src/sppm2.h:317-525) and their diffuse hits are applied to
 * this rank's hitpoints in photon order with the update rule of the config.  Equivalent to
 * sppm_trace + sppm_gather in batches sized to the record buffer. */
int sppm_photon_pass(sppm_ctx* ctx, int32_t pass, int64_t first, int64_t count);

/* The two halves of a photon pass, for multi-GPU use (trace a shard, exchange records, gather):
 * sppm_trace appends the photon-hit records of photons [first, first+count) of a pass to the context's
 * current record buffer; sppm_gather bins the buffer's records and applies them to this rank's
 * hitpoints in key order, then clears the buffer.
 * sppm_pack_records moves the buffer's records into d_rows[n][10] (row = key bits, position,
 * geometric normal, flux; any order) and clears the buffer -- what a rank contributes to an all-gather;
 * if cap_rows is too small nothing is cleared, n holds the count needed and SPPM_ERR_OVERFLOW returns.
 * sppm_append_records adds rows (typically another rank's) to the buffer; [photon_first, photon_end) is
 * the photon index range those rows can come from (it sizes the time slabs of the gather).  On a rank that owns
 * part of the hitpoints (tile_stride > 1) rows that lie farther than the search radius from the bounding box of its
 * hitpoints are not stored: none of its hitpoints can accept them. */
int sppm_trace(sppm_ctx* ctx, int32_t pass, int64_t first, int64_t count);
/* No counterpart in the reference.  Starts now, on the context's internal streams, work that a later call will
 * ask for: the trace of photons [first, first+count) of `pass` into the record buffer that is not in use (a
 * following sppm_trace with the same arguments then only adopts it) and, if `pass` is the pass after the last eye
 * pass, its eye pass (adopted by sppm_eye_pass(pass)).  Photon paths do not depend on hitpoints and eye paths do
 * not depend on photons, so results do not change; work that is not asked for afterwards is dropped. */
int sppm_look_ahead(sppm_ctx* ctx, int32_t pass, int64_t first, int64_t count);
/* sppm_look_ahead plus, behind the trace on the same internal stream, sppm_pack_records of what it traced into
 * d_rows (device, cap_rows x 10 doubles): this rank's rows of the NEXT batch become ready while the context's stream
 * is busy, so the exchange between ranks can overlap the gather.  sppm_rows_ready blocks the host until they are
 * there and returns their number; d_rows may then be read on any stream.  No counterpart in the reference. */
int sppm_look_ahead_rows(sppm_ctx* ctx, int32_t pass, int64_t first, int64_t count, double* d_rows, int64_t cap_rows);
int sppm_rows_ready(sppm_ctx* ctx, int64_t* n);
/* The same with the rows sorted by DESTINATION rank, for an all-to-all exchange (no counterpart in the reference).
 * Needs tile_stride > 1 with ownership 1: the eye pass is replicated, so every rank knows where every rank's hitpoints
 * lie and keeps a coarse occupancy map per rank.  d_rows is [tile_stride][cap_rows][10]; block d receives the records
 * whose cell in rank d's map (of the eye pass of `pass`) is set -- a superset of what d will use, d filters again with
 * its own map in sppm_append_records.  sppm_rows_ready_to blocks the host until the rows are there and returns the
 * number of rows of each block in h_counts[tile_stride]. */
int sppm_look_ahead_rows_to(sppm_ctx* ctx, int32_t pass, int64_t first, int64_t count, double* d_rows, int64_t cap_rows);
int sppm_rows_ready_to(sppm_ctx* ctx, int64_t* h_counts);
int sppm_gather(sppm_ctx* ctx);
int sppm_pack_records(sppm_ctx* ctx, double* d_rows, int64_t cap_rows, int64_t* n);
int sppm_append_records(sppm_ctx* ctx, const double* d_rows, int64_t n, int64_t photon_first, int64_t photon_end);

/* End-of-pass update of rule 1 (src/stochastic_ppm.h:721-741) from the pass's deposits, which it clears; a no-op
 * for rule 0. */
int sppm_end_pass(sppm_ctx* ctx);

/* No counterpart in the reference (it is single-threaded): the per-pass rule between GPUs.  d_out / d_in are
 * DEVICE arrays of 4 * sppm_hitpoint_count 64-bit integers, [4][P] = the fixed-point sums of the deposited photon
 * flux (x, y, z; sppm_fixed_point_scale units per unit of flux) and the deposit count M of the pass so far.  A
 * rank exports its deposits, the ranks' arrays are summed element-wise (an all-reduce; integer sums are exact,
 * so the result does not depend on the number of ranks), the sum is imported, then sppm_end_pass.  update_rule 1. */
int sppm_export_deposits(sppm_ctx* ctx, int64_t* d_out);
int sppm_import_deposits(sppm_ctx* ctx, const int64_t* d_in);
/* The same for the hitpoints [first, first + count) by image index: d_out / d_in are [4][count].  For a SLICED
 * all-reduce: export every slice, start the collective of each on a side stream as soon as its export is done, and
 * as each sum arrives import it and update that slice (sppm_end_pass_resolve_range) while the next collective is in
 * flight.  Slices together must cover every hitpoint exactly once per pass. */
int sppm_export_deposits_range(sppm_ctx* ctx, int64_t* d_out, int64_t first, int64_t count);
int sppm_import_deposits_range(sppm_ctx* ctx, const int64_t* d_in, int64_t first, int64_t count);
int sppm_fixed_point_scale(sppm_ctx* ctx, double* scale);

/* Image assembly, src/sppm2.h:658-670: d_image[P][3] = ld/(pass+1) + flux/(pi r^2 (pass+1) photon_num). */
int sppm_resolve(sppm_ctx* ctx, int32_t pass, int64_t photon_num, double* d_image);
/* sppm_end_pass then sppm_resolve in one call (src/stochastic_ppm.h:721-741 then :743-765; src/sppm2.h:658-670 under
 * rule 0).  Under the per-pass rule both run as ONE sweep over the hitpoints; results are those of the two calls. */
int sppm_end_pass_resolve(sppm_ctx* ctx, int32_t pass, int64_t photon_num, double* d_image);
/* update_rule 1, SPPM: the same for hitpoints [first, first + count) only.  d_image is the WHOLE image [P][3] (the rows
 * of the range are written) or null for the update alone.  No counterpart in the reference (see
 * sppm_export_deposits_range). */
int sppm_end_pass_resolve_range(sppm_ctx* ctx, int32_t pass, int64_t photon_num, double* d_image, int64_t first, int64_t count);

/* get_hdr_pixel on the device (src/hdr.h:37-45): d_rgbe[n][4]. */
int sppm_hdr_pixels(sppm_ctx* ctx, const double* d_rgb, int64_t n, uint8_t* d_rgbe);
/* save_hdr_file / save_ppm_file (src/hdr.h:48-89, src/ppm.h:22-29) on a HOST image [h][w][3]. */
int sppm_save_hdr_file(const char* filename, const double* h_image, int32_t width, int32_t height);
int sppm_save_ppm_file(const char* filename, const double* h_image, int32_t width, int32_t height);

/* Waits for the context's stream and reports a deferred device-side error (record overflow, K cap). */
int sppm_check(sppm_ctx* ctx);

/* No counterpart in the reference: the hitpoint state between ranks.  d_out[P][5] (P = sppm_hitpoint_count) =
 * (r^2, flux xyz, photon count) of the hitpoints this rank owns and zeros elsewhere; the sum over the ranks
 * (one all-reduce; exactly one non-zero term per hitpoint, so it is exact) is the state of a single-GPU run,
 * which sppm_import_state stores into every hitpoint.  After that sppm_resolve gives the full image on every
 * rank (classic PPM sums several hitpoints per pixel, which different ranks may own).  update_rule 0. */
int sppm_export_owned_state(sppm_ctx* ctx, double* d_out);
int sppm_import_state(sppm_ctx* ctx, const double* d_in);

/* Number of hitpoint slots: width*height for algorithm 0 (one per image index); the length of the
 * hitpoint list for algorithm 1 (point_map.GetData().size() after the loops of
 * src/progressive_photonmapping.h:760-793), known from sppm_create on. */
int sppm_hitpoint_count(sppm_ctx* ctx, int64_t* n);
/* algorithm 1: ProgressiveIntersection::index of every hitpoint, in creation order: h_pix[sppm_hitpoint_count]. */
int sppm_download_hitpoint_pixels(sppm_ctx* ctx, int32_t* h_pix);

/* Taps for parity tests.  Host output arrays, any may be NULL.  P = sppm_hitpoint_count; for algorithm 1
 * the index is the hitpoint's place in creation order, the three weights are the scalar weight, ld is 0.
 * Hitpoints by image index: valid[P], obj[P] (-1 if none), hp_f[P][10] = distance, position,
 * normal, weight; r2[P]; n[P]; flux[P][3]; ld[P][3]. */
int sppm_download_hitpoints(sppm_ctx* ctx, int32_t* h_valid, int32_t* h_obj, double* h_hp_f, double* h_r2,
                            int32_t* h_n, double* h_flux, double* h_ld);
/* Records currently in the buffer, sorted by key: key[n] (photon<<20 | depth<<1 | orient),
 * f[n][9] = position, geometric normal, flux.  Returns the count through n (call with NULL arrays
 * to size). */
int sppm_download_records(sppm_ctx* ctx, uint64_t* h_key, double* h_f, int64_t* n);
/* d_mask[P] (device, uint8): 1 where this rank owns the pixel's hitpoint (sppm_resolve is meaningful there).
 * No counterpart in the reference: the image of a multi-GPU run is the sum of the ranks' masked images. */
int sppm_owned_mask(sppm_ctx* ctx, uint8_t* d_mask);
/* This rank's image indices in gather-tile order (64 consecutive entries form a tile): the Hilbert
 * order of the hitpoint positions made by the last eye pass.  Call with NULL to get the count. */
int sppm_download_tile_order(sppm_ctx* ctx, int32_t* h_order, int64_t* n);
/* Accepted (record key, image index) pairs of the gathers since the tap was enabled, sorted by
 * (key, index).  Enable with cap > 0 before the photon pass. */
int sppm_accept_tap(sppm_ctx* ctx, int64_t cap);
int sppm_download_accepted(sppm_ctx* ctx, uint64_t* h_key, int32_t* h_pix, int64_t* n);

/* counters since creation: [0] rays cast, [1] photons traced, [2] photon-hit records, [3] (record, hitpoint)
 * pairs whose distance test ran in the gather, [4] accepted hitpoint updates, [5] gather batches,
 * [6] nanoseconds of device time in the gather kernel (HIP events on the context's stream, when
 * timing is on), [7] gather kernel launches timed, [8] in-region candidates staged, [9] staging rounds,
 * [10] live tile launches, [11] nanoseconds of device time in the photon-trace kernel (HIP events on the
 * stream it runs on), [12] trace kernel launches timed, [13] rays cast by photon tracing ([0] includes
 * them), [14] check_gather_max 1: the largest number of candidate hitpoints any photon hit had; otherwise the photon hits
 * whose search the K cap acted on, [15] the (record, hitpoint) pairs the cap withheld from the gather. */
int sppm_counters(sppm_ctx* ctx, uint64_t h_out[16]);
/* No counterpart in the reference: the FP64 vector FMA rate (TFLOP/s, an FMA = 2) this device sustains, measured with
 * a kernel of independent FMA chains -- the measured denominator bench.py prints beside the data-sheet peak. */
int sppm_measure_fma_peak(void* stream, double* h_tflops);
int sppm_set_timing(sppm_ctx* ctx, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
