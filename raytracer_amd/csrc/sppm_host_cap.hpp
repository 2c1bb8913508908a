// sppm_host_cap.hpp -- K cap host side.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
