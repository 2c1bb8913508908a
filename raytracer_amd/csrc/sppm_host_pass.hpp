// sppm_host_pass.hpp -- the pass driver: sppm_gather, batch sizing and the look-ahead ring, sppm_photon_pass, end of pass.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
