"""Multi-GPU sharding of the SPPM path: one process per GPU, torch.distributed over RCCL.

What shards and what does not (DESIGN.md "Multi-GPU"):
  * The reference's per-photon update (src/sppm2.h:391-421) is sequential per HITPOINT, not per
    photon: a hitpoint's state depends on the time-ordered photons near it, and on nothing else.
    So the hitpoints are partitioned across ranks and every rank replays, for its own hitpoints, the
    photon-hit records of the WHOLE pass in key order.  The result is bit-identical for every world size.
    Partition (sppm_config.ownership): 1 = contiguous ranges of the Hilbert order of the first eye pass's
    hitpoints, i.e. a compact region of space per rank (default here: tiles stay dense and a rank only
    bins the records near its region); 0 = 8x8 pixel tiles dealt round-robin.
  * Photon tracing is independent per photon (streams keyed by global photon index).  Two ways to
    give every rank all records of a pass:
      exchange="alltoall" (default with ownership 1): every batch is a contiguous photon range cut into one block per rank;
          a rank sorts the records of its block by DESTINATION -- the eye pass is replicated, so every rank keeps a coarse
          occupancy map of every rank's hitpoints and knows which ranks can use a record -- and one RCCL all-to-all per
          batch delivers to each rank only the records near its hitpoints (80 bytes each).
      exchange="allgather": the same blocks, every rank's records to every rank with one all-gather per batch
          (1.58 x 80 bytes per photon of the whole batch arrive at every rank); the receiver drops what is far.
      exchange="replicate": every rank traces all photons of the pass itself.  No data-path
          collective, but the tracing does not scale (measured: it is the bottleneck from 2 ranks on).
  * The image: every rank resolves its own hitpoints, zeroes the rest, one all-reduce (x + 0 is exact).
"""
import numpy as np

from . import scene, sppm


def tile_grid(width, height):
    """8x8 pixel tiles over the image-index grid: (tiles_x, tiles_y)."""
    return (width + 7) // 8, (height + 7) // 8


def owner_of_pixels(width, height, world):
    """int32 array [H*W]: the rank that owns each image index (tile t belongs to rank t % world)."""
    tx, _ = tile_grid(width, height)
    idx = np.arange(width * height)
    tile = (idx // width // 8) * tx + (idx % width) // 8
    return (tile % world).astype(np.int32)


class ShardedSppm:
    """rule 0 (default): the reference's per-photon update, exact for any world size: hitpoints partitioned, photon
    records exchanged (`exchange`).  rule 1: the per-pass update -- north_star's split: every rank keeps ALL hitpoints,
    traces photons [rank*M/world, (rank+1)*M/world) of the pass (streams keyed by global photon index) and deposits
    them into 64-bit fixed-point accumulators; ONE all-reduce (sum) of the accumulators per pass (4 x 8 bytes per
    hitpoint: 66 MB at 1080p), then every rank applies the same end-of-pass update.  Integer sums are exact, so the
    result is bit-identical for every world size.  The all-reduce is issued in `slices` parts (ranges of the image
    index): every slice's collective starts as soon as its export is done, and a slice is imported and updated
    (sppm_end_pass_resolve_range) while the collectives of the later slices are still in flight."""

    def __init__(self, width, height, photons_per_rank, rank=0, world=1, exchange=None, ownership=1,
                 spheres=scene.CORNELL_SPHERES, photons_total=None, rule=0, slices=4, **cfg_kw):
        if rule not in (0, 1):
            raise ValueError("rule must be 0 (per-photon update, sppm2.h) or 1 (per-pass update, stochastic_ppm.h)")
        if exchange is None:
            exchange = "allreduce" if rule == 1 else ("alltoall" if ownership == 1 and world > 1 else "allgather")
        if rule == 1 and exchange != "allreduce":
            raise ValueError("rule 1 exchanges the per-hitpoint accumulators: exchange='allreduce'")
        if rule == 0 and exchange not in ("replicate", "allgather", "alltoall"):
            raise ValueError("rule 0: exchange must be 'replicate', 'allgather' or 'alltoall'")
        if exchange == "alltoall" and (ownership != 1 or world < 2):
            raise ValueError("exchange='alltoall' needs several ranks with spatial ownership (ownership=1)")
        self.W, self.H, self.P = width, height, width * height
        self.rank, self.world = rank, world
        self.M_rank = photons_per_rank
        self.M = photons_per_rank * world
        self.rule = rule
        self.slices = max(1, int(slices))
        self.exchange = exchange if (world > 1 or rule == 1) else "replicate"
        self.replicated_trace = self.exchange == "replicate"
        self.ownership = ownership
        self.exchanged_bytes = 0   # bytes this rank handed to collectives (payload, both directions counted once)
        self.stage_events = []     # (name, start event, end event) on the current stream, when time_stages is set
        self.time_stages = False
        # photons_total: the photon streams are keyed over this many photons (classic PPM: every step's photons belong to ONE run)
        if rule == 1:  # all hitpoints on every rank
            cfg = sppm.make_config(width, height, photons_per_pass=photons_total or self.M, update_rule=1, **cfg_kw)
        else:
            cfg = sppm.make_config(width, height, photons_per_pass=photons_total or self.M, tile_first=rank, tile_stride=world, ownership=ownership, **cfg_kw)
        self.ctx = sppm.SppmContext(cfg, spheres)
        self.image = None
        self._dep = None
        self._rows = self._all = self._side = self._next = None
        self.total_passes = None   # if set, nothing is started ahead for passes >= total_passes
        self.batch_photons = 1 << 20 if world == 1 else (1 << 20) * world  # photons of all ranks per batch
        self.applied = 0

    def describe(self):
        if self.world == 1:
            return "1 GPU"
        if self.rule == 1:
            return f"{self.world} ranks: photons sharded by global index, hitpoints replicated, one all-reduce of the fixed-point deposits per pass (in {self.slices} slices)"
        how = "ranges of the hitpoints' Hilbert order" if self.ownership == 1 else "8x8 hitpoint tiles dealt round-robin"
        return f"{self.world} ranks: hitpoints by {how}, photon records by {self.exchange}"

    def step(self, i, resolve=True):
        """One SPPM pass: eye pass, photon pass of self.M photons, progressive update, image resolve."""
        ctx = self.ctx
        ctx.eye_pass(i)
        if self.rule == 1:
            ctx.photon_pass(i, self.M_rank, first=self.rank * self.M_rank)
            if self.world > 1 and ctx.n_hitpoints >= self.slices:
                return self._allreduce_and_end_pass(i, resolve)
        elif self.replicated_trace:
            ctx.photon_pass(i, self.M)
        else:
            self._photon_pass_allgather(i)
        if resolve:
            self.image = ctx.end_pass_resolve(i, self.M)
        else:
            ctx.end_pass()
        return self.image

    # ---- per-pass rule: one all-reduce of the deposits, in slices -------------------------------------------
    def _allreduce_and_end_pass(self, i, resolve):
        """Sum the ranks' accumulators [4][P] (int64) and apply the end of the pass, slice by slice: export every slice
        and start its collective at once (RCCL: asynchronously, on the process group's stream behind the export); then,
        in order, wait for a slice's sum, import it and update its hitpoints (+ their part of the image) while the later
        collectives are still running.  The next pass's tracing and eye pass already run on the context's look-ahead
        streams meanwhile.  gloo (tests): the same calls, each collective synchronous through pinned host memory."""
        import torch
        import torch.distributed as dist
        ctx, S, P = self.ctx, self.slices, self.ctx.n_hitpoints
        bounds = [P * s // S for s in range(S + 1)]
        if self._dep is None:
            self._dep = [torch.empty((4, bounds[s + 1] - bounds[s]), dtype=torch.int64, device=ctx.device) for s in range(S)]
        image = torch.empty((P, 3), dtype=torch.float64, device=ctx.device) if resolve else None
        staged = _host_staged(self._dep[0])
        works, evs = [], []
        for s in range(S):
            ctx.export_deposits_range(self._dep[s], bounds[s], bounds[s + 1] - bounds[s])
            if self.time_stages:
                e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e[0].record()
                evs.append(e)
            works.append(None if staged else dist.all_reduce(self._dep[s], op=dist.ReduceOp.SUM, async_op=True))
        for s in range(S):
            if staged:
                all_reduce_sum(self._dep[s])
            else:
                works[s].wait()   # the current stream waits for the collective; the host does not
            if self.time_stages:
                evs[s][1].record()
                self.stage_events.append((f"exchange_slice{s}",) + evs[s])
            ctx.import_deposits_range(self._dep[s], bounds[s], bounds[s + 1] - bounds[s])
            ctx.end_pass_resolve_range(i, self.M, image, bounds[s], bounds[s + 1] - bounds[s])
            self.exchanged_bytes += self._dep[s].numel() * 8
        if resolve:
            self.image = image
        return self.image

    def _stage_begin(self):
        """events around an exchange on the current stream (time_stages): device time of the collective as enqueued"""
        if not self.time_stages:
            return None
        import torch
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def _stage_end(self, ev):
        if ev:
            ev[1].record()
            self.stage_events.append(("exchange",) + ev)

    # ---- all-gather exchange, pipelined --------------------------------------------------------------------
    def _batch_size(self, done):
        """photons per rank of the batch that starts at photon `done` of a pass (same rule on every rank)"""
        batch = min(self.batch_photons, max(32768, self.applied // 2))
        return min(max(batch // self.world, 1024), (self.M - done) // self.world)

    def _buffers(self):
        import torch
        if self._rows is None:
            cap = 4 * (self.batch_photons // self.world) + 4096
            dev = self.ctx.device
            if self.exchange == "alltoall":  # one block of rows per destination
                self._rows = [torch.empty((self.world, cap, 10), dtype=torch.float64, device=dev) for _ in range(2)]
            else:
                self._rows = [torch.empty((cap, 10), dtype=torch.float64, device=dev) for _ in range(2)]
            self._all = [torch.empty((self.world * cap, 10), dtype=torch.float64, device=dev) for _ in range(2)]
            self._cap = cap
            self._side = torch.cuda.Stream(device=dev)

    def _where(self, i):
        """(pass, first photon) of step i: an SPPM step is a pass of its own"""
        return i, 0

    def _start_block(self, i, done, n, buf):
        """trace this rank's block of the batch (and the eye pass of pass i, if it is the next one) on the context's
        look-ahead streams and pack it into rows[buf]"""
        p, base = self._where(i)
        if self.exchange == "alltoall":
            self.ctx.look_ahead_rows_to(p, base + done + self.rank * n, n, self._rows[buf])
        else:
            self.ctx.look_ahead_rows(p, base + done + self.rank * n, n, self._rows[buf])

    def _exchange_alltoall(self, i, done, n, buf):
        """wait for the block's rows, then the counts (one small all-gather) and ONE all-to-all of the ragged blocks on the
        side stream: rank s's records for this rank land at recv[s * cap : s * cap + got[s]]"""
        import torch
        send = self.ctx.rows_ready_to()
        cap = self._cap
        with torch.cuda.stream(self._side):
            ev = self._stage_begin()
            table = exchange_count_rows(send, self.world, self.ctx.device)   # table[s][d]: rows rank s holds for rank d
            got = [table[s][self.rank] for s in range(self.world)]
            if max(got) > cap:
                raise RuntimeError("record exchange: a block received from another rank exceeds the row buffer")
            rows, recv = self._rows[buf], self._all[buf]
            all_to_all_rows([recv[s * cap:s * cap + got[s]] for s in range(self.world)], [rows[d, :send[d]] for d in range(self.world)], self.rank)
            self._stage_end(ev)
        self.exchanged_bytes += 80 * sum(got[s] for s in range(self.world) if s != self.rank)
        self._next = dict(key=(i, done, n), counts=got, m=cap, buf=buf)

    def _exchange(self, i, done, n, buf):
        """wait for the block's rows, then one all-gather (padded to the largest count) on the side stream"""
        import torch
        import torch.distributed as dist
        if self.exchange == "alltoall":
            return self._exchange_alltoall(i, done, n, buf)
        cnt = self.ctx.rows_ready()
        with torch.cuda.stream(self._side):
            ev = self._stage_begin()
            counts = exchange_counts(cnt, self.world, self.ctx.device)
            m = max(1, int(max(counts)))
            all_gather_into(self._all[buf][:self.world * m], self._rows[buf][:m])
            self._stage_end(ev)
        self.exchanged_bytes += 80 * m * (self.world - 1)   # received from the other ranks, padding included
        self._next = dict(key=(i, done, n), counts=counts, m=m, buf=buf)

    def _photon_pass_allgather(self, i):
        """Every batch is a contiguous range of the pass's photons cut into one block per rank (so batches are applied in
        key order and a batch's records are dense in time).  A rank traces its block, the packed records of all ranks
        are combined with one RCCL all-gather, and every rank applies the ones near its hitpoints
        (sppm_append_records drops the rest).  The block of the NEXT batch is traced and exchanged while the gather
        of this one runs.  Batches start small and grow with the photons already applied, like sppm_photon_pass's:
        the radii are large in the first passes, when a rank keeps most of the records."""
        import torch
        ctx = self.ctx
        self._buffers()
        done = 0
        while done < self.M:
            n = self._batch_size(done)
            if self._next is None or self._next["key"] != (i, done, n):   # nothing was started ahead (first batch)
                self._start_block(i, done, n, 0)
                self._exchange(i, done, n, 0)
            ex, self._next = self._next, None
            torch.cuda.current_stream().wait_stream(self._side)
            rows, m = self._all[ex["buf"]], ex["m"]
            base = self._where(i)[1]
            for r in range(self.world):
                ctx.append_records(rows[r * m:(r + 1) * m], ex["counts"][r], base + done + r * n, base + done + (r + 1) * n)
            self.applied += n * self.world
            done += n * self.world
            ni, ndone = (i, done) if done < self.M else (i + 1, 0)
            ahead = self.total_passes is None or ni < self.total_passes
            if ahead:
                nn = self._batch_size(ndone)
                self._start_block(ni, ndone, nn, 1 - ex["buf"])
            ctx.gather()
            if ahead:
                self._exchange(ni, ndone, nn, 1 - ex["buf"])   # overlaps the gather just enqueued

    def gather_image(self):
        """Full image on every rank: a rank's resolve is valid on the hitpoints it owns; the others are zeroed and
        one all-reduce sums the ranks (exactly one non-zero term per pixel)."""
        if self.world == 1 or self.rule == 1:   # rule 1: every rank holds all hitpoints
            return self.image
        out = self.image * self.ctx.owned_mask().to(self.image.dtype).unsqueeze(1)
        all_reduce_sum(out)
        return out

    def close(self):
        self.ctx.close()


class ShardedPpm(ShardedSppm):
    """Classic progressive photon mapping (src/progressive_photonmapping.h) over ranks.  The hitpoint LIST is shared
    out like SPPM's pixels (every rank traces the whole eye pass, then keeps a range of the list's Hilbert order);
    a step is one batch of photons_per_rank x world photons of the single run of `steps` batches, exchanged like
    SPPM's.  The image sums several hitpoints per pixel, which different ranks may own: finish() first sums the
    ranks' owned state (one all-reduce, exact: one non-zero term per hitpoint) and resolves on every rank."""

    def __init__(self, width, height, photons_per_rank, steps, rank=0, world=1, **kw):
        kw.setdefault("gather_radius2", 5.0)
        kw.setdefault("initial_radius2", kw["gather_radius2"])
        super().__init__(width, height, photons_per_rank, rank=rank, world=world, ownership=1 if world > 1 else 0,
                         photons_total=photons_per_rank * world * steps, algorithm=1, **kw)
        self.steps = steps
        self.total_passes = steps

    def _where(self, i):
        return 0, i * self.M

    def step(self, i, resolve=False):
        ctx = self.ctx
        if i == 0:
            ctx.eye_pass(0)
        if self.replicated_trace:
            ctx.photon_pass(0, self.M, first=i * self.M)
        else:
            self._photon_pass_allgather(i)
        return None

    def finish(self):
        import torch.distributed as dist
        if self.world > 1:
            state = self.ctx.export_owned_state()
            all_reduce_sum(state)
            self.ctx.import_state(state)
        self.image = self.ctx.resolve(0, self.M * self.steps)
        return self.image

    def gather_image(self):
        return self.finish()


def select_by_owner(stacked, owner):
    """stacked[world][P][3] -> [P][3], taking pixel p from rank owner[p] (works on CPU and GPU tensors)."""
    import torch
    idx = torch.as_tensor(owner, dtype=torch.int64, device=stacked.device)
    return stacked[idx, torch.arange(stacked.shape[1], device=stacked.device)]


# ---- collectives -------------------------------------------------------------------------------------------
# RCCL (backend "nccl") takes device tensors as they are.  With gloo -- the backend that rehearses the N > 1 flow on
# a box with fewer GPUs than ranks -- NO CUDA tensor enters the collective: gloo's own handling of CUDA tensors
# between local processes was seen to stall (round 1: both ranks stuck inside all_gather_into_tensor with > 100 MB
# messages, reproduced without this package), so device tensors are staged through pinned host memory there.
def _host_staged(t):
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def _pinned_like(t):
    import torch
    return torch.empty(t.shape, dtype=t.dtype, pin_memory=t.is_cuda)


def all_gather_into(out, inp):
    import torch.distributed as dist
    if _host_staged(inp):
        h_in, h_out = _pinned_like(inp), _pinned_like(out)
        h_in.copy_(inp)
        dist.all_gather_into_tensor(h_out, h_in)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, inp)


def all_reduce_sum(t):
    import torch.distributed as dist
    if _host_staged(t):
        h = _pinned_like(t)
        h.copy_(t)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def all_to_all_rows(out_list, in_list, rank):
    """ragged all-to-all: in_list[d] goes to rank d, out_list[s] receives rank s's block (sizes agreed beforehand).
    RCCL: one grouped all_to_all.  gloo (rehearsal): point-to-point through pinned host memory."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "gloo":
        dist.all_to_all(out_list, in_list)
        return
    world = len(in_list)
    out_list[rank].copy_(in_list[rank])
    h_in = [None if d == rank else _pinned_like(in_list[d]).copy_(in_list[d]) for d in range(world)]
    h_out = [None if s == rank else _pinned_like(out_list[s]) for s in range(world)]
    if any(t.is_cuda for t in in_list):
        torch.cuda.current_stream().synchronize()
    ops = []
    for k in range(1, world):   # the same pairing order on every rank
        d, s = (rank + k) % world, (rank - k) % world
        if h_in[d].numel():
            ops.append(dist.P2POp(dist.isend, h_in[d], d))
        if h_out[s].numel():
            ops.append(dist.P2POp(dist.irecv, h_out[s], s))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for s in range(world):
        if s != rank and h_out[s].numel():
            out_list[s].copy_(h_out[s])


def exchange_count_rows(counts, world, device):
    """all-gather `world` integers per rank: returns table[s][d] = counts of rank s"""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        device = "cpu"
    mine = torch.tensor(list(counts), dtype=torch.int64, device=device)
    out = torch.empty((world * world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return out.view(world, world).tolist()


def exchange_counts(count, world, device):
    """all-gather one integer per rank"""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        device = "cpu"
    mine = torch.tensor([count], dtype=torch.int64, device=device)
    out = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return [int(x) for x in out.tolist()]
