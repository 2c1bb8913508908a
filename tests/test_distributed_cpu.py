"""World-size-2 gloo test (CPU) of the host-side sharding logic: tile ownership, the padded
all-gather of ragged per-rank rows used for the photon records, and the image merge by owner.
The GPU kernels are not involved (no GPU here); tests/test_gpu_parity.py covers them, including a
two-rank record exchange emulated on one GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, W, H, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import parallel
    own = parallel.owner_of_pixels(W, H, world)
    # image merge: every rank's "resolve" is valid only on its own pixels
    truth = torch.arange(W * H * 3, dtype=torch.float64).reshape(W * H, 3)
    mine = torch.where(torch.from_numpy(own == rank)[:, None], truth, torch.full_like(truth, -1.0))
    stacked = torch.empty((world * W * H, 3), dtype=torch.float64)
    dist.all_gather_into_tensor(stacked, mine)
    merged = parallel.select_by_owner(stacked.view(world, W * H, 3), own)
    ok_img = bool(torch.equal(merged, truth))
    # ragged rows: rank r contributes 5 + 3 r rows tagged with its rank
    cnt = 5 + 3 * rank
    counts = parallel.exchange_counts(cnt, world, torch.device("cpu"))
    m = max(counts)
    rows = torch.zeros((m, 10), dtype=torch.float64)
    rows[:cnt] = rank + 1
    out = torch.empty((world * m, 10), dtype=torch.float64)
    dist.all_gather_into_tensor(out, rows)
    out = out.view(world, m, 10)
    ok_rows = counts == [5 + 3 * r for r in range(world)] and all(bool((out[r, :counts[r]] == r + 1).all()) for r in range(world))
    # ragged all-to-all of row blocks (the rule-0 record exchange by destination): rank s sends 2 + s + 3 d rows tagged 100 s + d to rank d
    send = [2 + rank + 3 * d for d in range(world)]
    table = parallel.exchange_count_rows(send, world, torch.device("cpu"))
    got = [table[s][rank] for s in range(world)]
    ins = [torch.full((send[d], 10), 100.0 * rank + d, dtype=torch.float64) for d in range(world)]
    outs = [torch.empty((got[s], 10), dtype=torch.float64) for s in range(world)]
    parallel.all_to_all_rows(outs, ins, rank)
    ok_a2a = table == [[2 + s + 3 * d for d in range(world)] for s in range(world)] and all(
        outs[s].shape[0] == 2 + s + 3 * rank and bool((outs[s] == 100.0 * s + rank).all()) for s in range(world))
    # the per-pass rule's exchange: an exact integer sum of the fixed-point deposits
    dep = torch.arange(4 * W * H, dtype=torch.int64).reshape(4, W * H) * (rank + 1) + (1 << 40) * rank
    parallel.all_reduce_sum(dep)
    want = torch.arange(4 * W * H, dtype=torch.int64).reshape(4, W * H) * sum(r + 1 for r in range(world)) + (1 << 40) * sum(range(world))
    ok_sum = bool(torch.equal(dep, want))
    ret[rank] = (ok_img, ok_rows, ok_a2a, ok_sum)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_gloo(world):
    W, H = 40, 24
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, W, H, ret), nprocs=world, join=True)
    assert all(ret[r] == (True, True, True, True) for r in range(world)), dict(ret)
