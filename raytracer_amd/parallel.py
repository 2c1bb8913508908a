"""Multi-GPU sharding of the SPPM path: one process per GPU, torch.distributed over RCCL.

What shards and what does not (DESIGN.md "Multi-GPU"):
  * The reference's per-photon update (src/sppm2.h:391-421) is sequential per HITPOINT, not per
    photon: a hitpoint's state depends on the time-ordered photons near it, and on nothing else.
    So the hitpoints are partitioned across ranks (contiguous slices of image indices) and every
    rank replays, for its own hitpoints, the photon-hit records of the WHOLE pass in key order.
    The result is bit-identical for every world size.
  * Photon tracing is independent per photon (streams keyed by global photon index).  Two ways to
    give every rank all records of a pass:
      exchange="replicate": every rank traces all photons of the pass itself.  No data-path
          collective.  On MI355X an FP64 trace of a photon costs far less than moving its ~2.6
          80-byte records over xGMI, so this is the default.
      exchange="allgather": rank r traces photons [r*M, (r+1)*M) and the packed records are
          combined with one RCCL all-gather per batch.
  * The image is assembled per slice and all-gathered once when it is needed.
"""
import numpy as np

from . import scene, sppm


def pixel_slice(P, rank, world):
    """Contiguous slice [first, first+count) of image indices owned by `rank`; sizes differ by <= 1."""
    base, rem = divmod(P, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class ShardedSppm:
    def __init__(self, width, height, photons_per_rank, rank=0, world=1, exchange="replicate", spheres=scene.CORNELL_SPHERES,
                 **cfg_kw):
        if exchange not in ("replicate", "allgather"):
            raise ValueError("exchange must be 'replicate' or 'allgather'")
        self.W, self.H, self.P = width, height, width * height
        self.rank, self.world = rank, world
        self.M_rank = photons_per_rank
        self.M = photons_per_rank * world
        self.exchange = exchange if world > 1 else "replicate"
        self.replicated_trace = self.exchange == "replicate"
        first, count = pixel_slice(self.P, rank, world)
        self.first, self.count = first, count
        cfg = sppm.make_config(width, height, photons_per_pass=self.M, pixel_first=first, pixel_count=count, **cfg_kw)
        self.ctx = sppm.SppmContext(cfg, spheres)
        self.image = None

    def describe(self):
        if self.world == 1:
            return "1 GPU"
        return f"{self.world} ranks: hitpoints sliced by image index, photon records by {self.exchange}"

    def step(self, i, resolve=True):
        """One SPPM pass: eye pass, photon pass of self.M photons, progressive update, image resolve."""
        ctx = self.ctx
        ctx.eye_pass(i)
        if self.replicated_trace:
            ctx.photon_pass(i, self.M)
        else:
            self._photon_pass_allgather(i)
        ctx.end_pass()
        if resolve:
            self.image = ctx.resolve(i, self.M)
        return self.image

    def _photon_pass_allgather(self, i):
        raise NotImplementedError("exchange='allgather' needs sppm_pack_records/sppm_append_records (next step)")

    def gather_image(self):
        """Full image on every rank: each rank contributes its slice (one all-gather)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return self.image
        base = -(-self.P // self.world)
        mine = torch.zeros((base, 3), dtype=torch.float64, device=self.image.device)
        mine[: self.count] = self.image[self.first: self.first + self.count]
        out = torch.empty((self.world * base, 3), dtype=torch.float64, device=self.image.device)
        dist.all_gather_into_tensor(out, mine)
        full = torch.empty((self.P, 3), dtype=torch.float64, device=self.image.device)
        for r in range(self.world):
            f, c = pixel_slice(self.P, r, self.world)
            full[f: f + c] = out[r * base: r * base + c]
        return full

    def close(self):
        self.ctx.close()
