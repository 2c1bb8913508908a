// sppm_host_eye.hpp -- eye pass host side: classic PPM hitpoint list, eye launch, views.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
        // re-form the gather tiles: sort this rank's hitpoints along a Hilbert curve of their positions, into the next
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
