// sppm_host_create.hpp -- scene grid build, sppm_create / sppm_destroy, sppm_camera_rays and sppm_intersect.
// Part of the one translation unit sppm_kernels.hip (included there, after the kernels; not a header to
// include anywhere else): the host side of the C ABI declared in include/sppm_hip.h.

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
    if (cfg->ownership != 0 && cfg->ownership != 1) return fail(SPPM_ERR_INVALID, "sppm_create: ownership must be 0 (screen tiles) or 1 (ranges of the hitpoints' Hilbert order)");
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
        return fail(SPPM_ERR_INVALID, "sppm_create: classic progressive photon mapping runs with update_rule 0; over several ranks its hitpoint list is shared out by ownership 1 (ranges of the Hilbert order)");
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
    c->dep_min = (int)std::min<int64_t>(768, std::max<int64_t>(64, env_i64("SPPM_DEP_MIN", 640) / 64 * 64));
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
    // (per-photon rule: 8192 photons first, then 0.35 of the photons applied so far -- with the gather a third shorter since its
    // tallies stay in the wavefronts, smaller first batches pay: 188-190M photons/s over the first 20 passes against 170-173M
    // with 32768 / 0.5, 337M against 334M over 256)
    c->batch_min = std::max<int64_t>(64, env_i64("SPPM_BATCH_MIN", 8192));
    { const char* e = getenv("SPPM_BATCH_GROWTH"); c->batch_growth = e && *e && atof(e) > 0.0 ? atof(e) : 0.35; }
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
        // ownership 1 (ranges of the Hilbert order of the first eye pass's hitpoints): every pixel until that
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
