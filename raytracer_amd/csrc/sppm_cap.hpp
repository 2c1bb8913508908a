// sppm_cap.hpp -- the K cap of the reference's kd search (src/kdtree.h:101-151), reproduced exactly.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
//
// SearchNodeKNN keeps a max-heap of the hitpoints that passed its two tests; once the heap holds more than K
// (render()'s gahter_max_photon_num) it drops the farthest and tightens BOTH tests -- distance and flatness -- to the new
// farthest distance for the rest of the walk (kdtree.h:126-133).  What survives depends on the order in which the walk
// meets the hitpoints, that is on the tree: medians of ranges that std::sort ordered axis by axis (kdtree.h:165-198).
// So the tree is the reference's tree (built on the host with the same std::sort from the same input order, see
// cap_build_tree in sppm_kernels.hip; `kd` below is the sorted vector, a node is the middle of its range), and one lane
// per photon-hit record walks it the way the recursion does, with the heap in global memory.
//
// The gathers keep working from the grid: they compute the UNCAPPED set {d2 < R^2, |dt| <= 0.01 R^2}.  This kernel
// hands them the difference, the pairs (record, hitpoint) that the uncapped search holds and the capped one drops:
//   * an entry pushed out of the heap,
//   * a node visited after the bound had tightened that fails the tightened tests but passes the original ones,
//   * a node in a subtree the tightened bound pruned (walked here in "shadow" mode: no heap, original bound).
// Only pairs the gather could still accept are kept (hitpoint owned by this rank, normals agree, d2 <= the hitpoint's
// radius now -- radii only shrink, so this is a superset of what the gather would accept later).  They go into the
// Veto table (sppm_gather_simple.hpp) that the gathers consult before they apply a pair.
#pragma once

// Which photon hits can the cap act on at all?  The heap overflows only if MORE than K hitpoints pass the search's
// original tests, and those all lie within sqrt(R0) of the photon hit.  A coarse grid of hitpoint counts (cells of half
// the search radius, made on the host together with the tree) bounds that number from above by the hitpoints in the
// cells the ball's bounding box touches; a photon hit whose bound is <= K is searched as if there were no cap -- by
// the gathers alone -- and never walks the tree.  At 1080p with the reference's K = 10000 that leaves a few percent of
// the photon hits; at 320x240 none.
struct CapGrid { const uint32_t* cnt; double ox, oy, oz, inv_c; int nx, ny, nz; };
// cell coordinate of x, clamped to [0, n-1]; monotone in x (the host counts the hitpoints with the same expression)
__device__ __forceinline__ int cap_coord(double x, double o, double inv_c, int n)
{
    const double f = fmin(fmax(floor((x - o) * inv_c), 0.0), 2147483000.0);
    const int i = (int)f;
    return i > n - 1 ? n - 1 : i;
}
__global__ void __launch_bounds__(256) k_cap_filter(RecArrays rec, const unsigned long long* __restrict__ rec_count, CapGrid g, double reach, long long K,
                                                    uint32_t* __restrict__ surv, uint32_t* __restrict__ surv_cell, unsigned long long* __restrict__ n_surv)
{
    const long long slots = min((long long)*rec_count, (long long)rec.cap);
    const long long cap = rec.cap;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < slots; r += (long long)gridDim.x * blockDim.x) {
        if (rec.key[r] == REC_INVALID) continue;
        const double qx = rec.f[r], qy = rec.f[cap + r], qz = rec.f[2 * cap + r];
        const int x0 = cap_coord(qx - reach, g.ox, g.inv_c, g.nx), x1 = cap_coord(qx + reach, g.ox, g.inv_c, g.nx);
        const int y0 = cap_coord(qy - reach, g.oy, g.inv_c, g.ny), y1 = cap_coord(qy + reach, g.oy, g.inv_c, g.ny);
        const int z0 = cap_coord(qz - reach, g.oz, g.inv_c, g.nz), z1 = cap_coord(qz + reach, g.oz, g.inv_c, g.nz);
        long long sum = 0;
        for (int iz = z0; iz <= z1; iz++)
            for (int iy = y0; iy <= y1; iy++)
                for (int ix = x0; ix <= x1; ix++) sum += (long long)g.cnt[((size_t)iz * g.ny + iy) * g.nx + ix];
        if (sum > K) {  // survivors carry their own cell, so that the search can take them in spatial order (cap_search sorts on it)
            const unsigned long long slot = atomicAdd(n_surv, 1ull);
            surv[slot] = (uint32_t)r;
            surv_cell[slot] = (uint32_t)(((size_t)cap_coord(qz, g.oz, g.inv_c, g.nz) * g.ny + cap_coord(qy, g.oy, g.inv_c, g.ny)) * g.nx + cap_coord(qx, g.ox, g.inv_c, g.nx));
        }
    }
}

struct CapParams {
    const uint32_t* surv; long long n_surv;  // the photon hits (record slots) the cap can act on (k_cap_filter)
    const double4* kd; int n;          // the reference's node_values_ after CreateKDtree, sorted range by range: position and (w) hitpoint slot
    double R0; long long K;            // gather_photon_radius (squared), gahter_max_photon_num
    double2* heap; long long heap_stride; int lanes;  // one heap of K+1 entries (x: squared distance, y: hitpoint slot) per lane
    Veto veto; unsigned long long* veto_n; unsigned long long veto_cap;      // table and fill; entries beyond veto_cap are counted, not stored
    unsigned long long* stat;          // [0] queries whose cap acted, [1] nodes visited, [2] largest heap fill
    Own own;
};

__device__ __forceinline__ void veto_put(const CapParams& cp, unsigned long long key, int pix)
{
    const unsigned long long slot = atomicAdd(cp.veto_n, 1ull);
    if (slot >= cp.veto_cap) return;  // the host sees the count, grows the table and runs the search again
    for (unsigned long long h = veto_hash(key, (unsigned int)pix) & cp.veto.mask;; h = (h + 1ull) & cp.veto.mask) {
        if (atomicCAS(&cp.veto.tab[2 * h], 0ull, key) == 0ull) { cp.veto.tab[2 * h + 1] = (unsigned long long)pix; return; }
    }
}

// std::push_heap / std::pop_heap of libstdc++ (bits/stl_heap.h: __push_heap, __adjust_heap) on (d2, pix), ordered by d2.
// An entry is one 16-byte word: a move is one load and one store.
__device__ __forceinline__ void cap_heap_push(double2* h, long long n /* size before */, double d2, int32_t pix)
{
    long long hole = n;
    while (hole > 0) {
        const long long par = (hole - 1) / 2;
        const double2 p = h[par];
        if (!(p.x < d2)) break;
        h[hole] = p;
        hole = par;
    }
    h[hole] = make_double2(d2, (double)pix);
}
// removes the top; n = size before; the caller has read the top already
__device__ __forceinline__ void cap_heap_pop(double2* h, long long n)
{
    const long long len = n - 1;  // the last element becomes `value`, the heap [0, len) gets a hole at the top
    if (len <= 0) return;
    const double2 v = h[len];
    long long hole = 0, child = 0;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        const double2 a = h[child], b = h[child - 1];
        if (a.x < b.x) { child--; h[hole] = b; } else h[hole] = a;
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    while (hole > 0) {  // __push_heap(first, hole, 0, value)
        const long long par = (hole - 1) / 2;
        const double2 p = h[par];
        if (!(p.x < v.x)) break;
        h[hole] = p;
        hole = par;
    }
    h[hole] = v;
}

#define CAP_STACK 96
__global__ void __launch_bounds__(64) k_cap_search(HpArrays hp, RecArrays rec, const unsigned long long* __restrict__ rec_count, CapParams cp, int32_t* __restrict__ err)
{
    const int lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane_id >= cp.lanes) return;
    double2* heap = cp.heap + (long long)lane_id * cp.heap_stride;
    const long long cap = rec.cap, P = hp.P;
    const double thr0 = cp.R0 * 0.01;
    unsigned long long n_capped = 0, n_nodes = 0, max_fill = 0;
    // stack of ranges still to walk: begin, end, depth | flags, and for a far child the squared plane distance
    int s_b[CAP_STACK], s_e[CAP_STACK], s_f[CAP_STACK];
    double s_d[CAP_STACK];
    const int F_SHADOW = 1 << 8, F_FAR = 1 << 9;
    for (long long si = lane_id; si < cp.n_surv; si += cp.lanes) {
        const long long r = (long long)cp.surv[si];
        const unsigned long long key = rec.key[r];
        if (key == REC_INVALID) continue;
        const d3 q = mk(rec.f[r], rec.f[cap + r], rec.f[2 * cap + r]);
        const d3 gn = mk(rec.f[3 * cap + r], rec.f[4 * cap + r], rec.f[5 * cap + r]);
        const d3 on = (key & 1ull) ? gn * -1.0 : gn;
        long long nh = 0;
        double B = cp.R0;
        bool capped = false;
        int sp = 0;
        s_b[0] = 0; s_e[0] = cp.n; s_f[0] = 0; s_d[0] = 0.0; sp = 1;
        while (sp > 0) {
            sp--;
            const int begin = s_b[sp], end = s_e[sp];
            int f = s_f[sp];
            if (f & F_FAR) {  // `if (delta * delta < query.max_distance2)` with the bound as it is NOW (kdtree.h:138,146)
                const double dd = s_d[sp];
                if (!(f & F_SHADOW)) {
                    if (!(dd < B)) {
                        if (capped && dd < cp.R0) f |= F_SHADOW;  // the uncapped walk would have gone in
                        else continue;
                    }
                } else if (!(dd < cp.R0)) continue;
            }
            if (end - begin <= 0) continue;
            const bool shadow = (f & F_SHADOW) != 0;
            const int depth = f & 0xFF;
            const int mid = begin + (end - begin) / 2;
            const double4 node = cp.kd[mid];
            const int pix = (int)node.w;
            n_nodes++;
            const d3 np = mk(node.x, node.y, node.z);
            const int axis = depth % 3;
            const double delta = axis == 0 ? q.x - np.x : (axis == 1 ? q.y - np.y : q.z - np.z);
            const d3 dir = np - q;
            const double d2 = len2(dir);
            const double dt = dot(on, dir / sqrt(d2));  // NaN at d2 == 0: fails both forms of the test
            const bool pass0 = d2 < cp.R0 && fabs(dt) <= thr0;
            int drop = -1; double drop_d2 = 0.0;  // a hitpoint the uncapped search holds and this one does not
            if (!shadow && d2 < B && fabs(dt) <= B * 0.01) {
                cap_heap_push(heap, nh, d2, pix);
                nh++;
                if ((unsigned long long)nh > max_fill) max_fill = (unsigned long long)nh;
                if (nh > cp.K) {
                    const double2 top = heap[0];
                    drop = (int)top.y; drop_d2 = top.x;
                    cap_heap_pop(heap, nh);
                    nh--;
                    B = heap[0].x;
                    capped = true;
                }
            } else if (pass0 && (shadow || capped)) { drop = pix; drop_d2 = d2; }
            if (drop >= 0 && hp.valid[drop] && owned(cp.own, drop) && drop_d2 <= hp.r2[drop]) {
                const d3 hn = mk(hp.nrm[drop], hp.nrm[P + drop], hp.nrm[2 * P + drop]);
                if (dot(hn, gn) > 1e-3) veto_put(cp, key, drop);
            }
            // children: the near side first, then the far side if the plane is closer than the bound is THEN
            const int nb = delta > 0.0 ? mid + 1 : begin, ne = delta > 0.0 ? end : mid;
            const int fb = delta > 0.0 ? begin : mid + 1, fe = delta > 0.0 ? mid : end;
            const int child_f = ((depth + 1) % 3) | (f & F_SHADOW);  // only depth % 3 is ever used: keep it small
            if (sp + 2 > CAP_STACK) { err[7] = 1; break; }
            if (fe - fb > 0) { s_b[sp] = fb; s_e[sp] = fe; s_f[sp] = child_f | F_FAR; s_d[sp] = delta * delta; sp++; }
            if (ne - nb > 0) { s_b[sp] = nb; s_e[sp] = ne; s_f[sp] = child_f; s_d[sp] = 0.0; sp++; }
        }
        if (capped) n_capped++;
    }
    if (n_capped) atomicAdd(&cp.stat[0], n_capped);
    if (n_nodes) atomicAdd(&cp.stat[1], n_nodes);
    if (max_fill) atomicMax(&cp.stat[2], max_fill);
}
