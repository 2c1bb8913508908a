// sppm_host_gather.hpp -- gather host side: batch statistics, binning, tile gather (rule 0) and deposits (rule 1).
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
    // (three and a half of the four wavefront slots of every SIMD: with the tallies kept per wavefront the kernel is short
    // enough that what runs beside it -- tracer, binning, the next eye pass -- needs the room: 896 against 1024 workgroups,
    // 165M against 163M photons/s at 20 passes, 320M against 311M at 256)
    const int gather_blocks = std::max(1, std::min(nblk(c->n_owned, 4), (int)env_i64("SPPM_GATHER_BLOCKS", 224 * GATHER_WAVES)));
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
    // (the first passes of a render: ~2800 pairs examined per photon at 1080p), down to two of them once it is not
    // (~1600 pairs per photon from pass ~100 on): what runs beside the kernel -- the eye pass of the next pass with its
    // sort, the tracer, the binning -- gets the other slots, and late in a render the next deposit waits for that eye
    // pass, not for this kernel.  Measured with fixed counts (photons/s at 20 / 256 passes, before the walk by rows of
    // cells): 768 workgroups 267M / 305M, 704 266M / 310M, 640 260M / 313M, 576 249M / 317M, 512 237M / 313M; with the
    // walk by rows and the single-precision tests the kernel is shorter and the late optimum moved from 576 to 448
    // (DESIGN.md section 3c).
    int dep_blocks = 256 * 3;
    if (c->pairs_per_photon > 0.0) {
        const double t = std::min(1.0, std::max(0.0, (c->pairs_per_photon - c->dep_light) / (c->dep_heavy - c->dep_light)));
        dep_blocks = c->dep_min + 64 * (int)std::lround((768 - c->dep_min) / 64.0 * t);
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
        HIPCHK(hipMemsetAsync(c->d_counters + 8, 0, 6 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL((k_deposit_tiles<false, true>), dim3(blocks), dim3(256), 0, st, c->hp, BS.rec_s, BS.cell_start, tp, c->d_items, c->d_item_q, c->d_counters, c->d_err);
        unsigned long long t[24];
        HIPCHK(hipMemcpyAsync(t, c->d_counters, sizeof t, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double tot = (double)(t[16] + t[17] + t[18] + t[19] + t[20] + t[21]);
        fprintf(stderr, "[sppm] deposit: items %llu, shares of wave cycles: setup %.1f%% count %.1f%% issue %.1f%% append %.1f%% (of it waiting for the group's loads %.1f%%) process %.1f%% flush %.1f%%; grid %dx%dx%d\n", t[7],
                100.0 * t[16] / tot, 100.0 * t[17] / tot, 100.0 * t[18] / tot, 100.0 * t[19] / tot, 100.0 * t[23] / tot, 100.0 * t[20] / tot, 100.0 * t[21] / tot, g.nx, g.ny, g.nz);
        fprintf(stderr, "[sppm]   lanes in their cluster per staged row (cumulative): %.1f of 64; staged rows per item %.0f\n", (double)t[3] / std::max(1.0, (double)t[5]), (double)t[5] / std::max(1.0, (double)t[7]));
        fprintf(stderr, "[sppm]   this launch: windows of 64 cells %llu, groups loaded %llu, records loaded %llu (%.1f per group), cells in the regions %llu\n", t[10], t[11], t[12], (double)t[12] / std::max(1.0, (double)t[11]), t[13]);
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
