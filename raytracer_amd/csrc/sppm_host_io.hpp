// sppm_host_io.hpp -- image files, downloads, exports / imports for several GPUs, taps, counters and timing.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
