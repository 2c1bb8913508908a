// sppm_host_trace.hpp -- photon tracer host side: sppm_trace, the look-ahead entry points, record rows, pack / append.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
