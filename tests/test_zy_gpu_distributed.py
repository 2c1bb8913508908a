"""ShardedSppm end to end with two ranks (two processes on one GPU, gloo): the image every rank ends up with and the
hitpoint state merged by ownership equal a single-GPU run of the same passes bit for bit, for both record exchanges
and both ownership rules.  RCCL itself needs one GPU per rank and is exercised by bench.py --gpus N on a node.
Collected late (zy) on purpose: it starts subprocesses."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import sppm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env, timeout=150):
    """A hang is a failure (it used to be a skip): the workers' output is part of the message.  No CUDA tensor enters
    gloo any more -- raytracer_amd.parallel stages them through pinned host memory -- which is what stalled before."""
    try:
        return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        out = (e.stdout or b"").decode(errors="replace")[-2000:] + (e.stderr or b"").decode(errors="replace")[-4000:]
        pytest.fail(f"the two-rank run did not finish within {timeout} s:\n{out}")


def _single(W, H, M, passes):
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M))
    for i in range(passes):
        ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
    img = ctx.resolve(passes - 1, M).cpu().numpy()
    hp = ctx.hitpoints()
    ctx.check(); ctx.close()
    return img, hp


_EXCHANGED = {}


@pytest.mark.parametrize("exchange,ownership,world", [("allgather", 1, 2), ("alltoall", 1, 2), ("replicate", 0, 2), ("allgather", 0, 2), ("alltoall", 1, 4)])
def test_ranks_equal_one_gpu(tmp_path, exchange, ownership, world):
    """(world 4: the rehearsal of the node run -- four gloo ranks on this one GPU, the ragged all-to-all between all pairs)"""
    W, H, M_rank, passes = 96, 64, 24000, 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 1000 + {"allgather": 0, "replicate": 1, "alltoall": 4}[exchange] + 2 * ownership + 10 * world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path), exchange, str(ownership)]
    res = _run(cmd, env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    ref_img, ref = _single(W, H, M_rank * world, passes)
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    masks = [g["mask"].astype(bool) for g in got]
    assert (np.sum(masks, axis=0) == 1).all()
    for g in got:
        assert (g["image"].view(np.uint64) == ref_img.view(np.uint64)).all(), "image on a rank differs from the single-GPU image"
    for k in ("n", "r2", "flux"):
        merged = np.zeros_like(ref[k])
        for m, g in zip(masks, got):
            merged[m] = g[k][m]
        a, b = np.ascontiguousarray(merged), np.ascontiguousarray(ref[k])
        assert (a.view(np.uint64) == b.view(np.uint64)).all() if a.dtype.kind == "f" else (a == b).all(), k
    if world != 2:
        return
    _EXCHANGED[(exchange, ownership)] = [int(g["exchanged"]) for g in got]
    if ("allgather", 1) in _EXCHANGED and ("alltoall", 1) in _EXCHANGED:
        # sorted by destination, a rank receives only the records near its hitpoints: fewer bytes than the all-gather brings
        assert sum(_EXCHANGED[("alltoall", 1)]) < sum(_EXCHANGED[("allgather", 1)]), _EXCHANGED
        assert min(_EXCHANGED[("alltoall", 1)]) > 0


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_per_pass_rule_equal_one_gpu(tmp_path, world):
    """north_star's multi-GPU path (ShardedSppm(rule=1)): each rank traces half of every pass's photons and deposits them
    into its copy of all hitpoints, one all-reduce of the 64-bit fixed-point deposits per pass, the same end-of-pass
    update on every rank.  State and image on BOTH ranks equal a single-GPU run of the per-pass rule bit for bit."""
    W, H, M_rank, passes = 96, 64, 24000, 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29900 + os.getpid() % 1000 + 10 * world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path), "allreduce", "0"]
    res = _run(cmd, env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    M = M_rank * world
    ctx = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M, update_rule=1))
    for i in range(passes):
        ctx.eye_pass(i); ctx.photon_pass(i, M); ctx.end_pass()
    ref_img, ref = ctx.resolve(passes - 1, M).cpu().numpy(), ctx.hitpoints()
    ctx.check(); ctx.close()
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert (g["image"].view(np.uint64) == ref_img.view(np.uint64)).all(), "image on a rank differs from the single-GPU image"
        for k in ("n", "r2", "flux"):
            a, b = np.ascontiguousarray(g[k]), np.ascontiguousarray(ref[k])
            assert (a.view(np.uint64) == b.view(np.uint64)).all() if a.dtype.kind == "f" else (a == b).all(), k
        assert int(g["exchanged"]) == passes * 4 * 8 * W * H   # 32 bytes per hitpoint and pass


def test_two_ranks_classic_ppm_equal_one_gpu(tmp_path):
    """ShardedPpm: the hitpoint list shared out by ranges of its Hilbert order, photon records by all-gather, the owned
    state summed over the ranks at the end: state and image on every rank equal a single-GPU render, bit for bit."""
    world, W, H, M_rank, steps = 2, 96, 64, 24000, 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 1000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path), "ppm", "1"]
    res = _run(cmd, env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    total = M_rank * world * steps
    ctx = sppm.SppmContext(sppm.make_config(W, H, supersamples=2, photons_per_pass=total, gather_radius2=5.0, initial_radius2=5.0, algorithm=1))
    ctx.eye_pass(0)
    ctx.photon_pass(0, total)
    ref_img, ref = ctx.resolve(0, total).cpu().numpy(), ctx.hitpoints()
    ctx.check(); ctx.close()
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    masks = [g["mask"].astype(bool) for g in got]
    assert (np.sum(masks, axis=0) == 1).all() and abs(int(masks[0].sum()) - int(masks[1].sum())) <= 1
    for g in got:
        assert (g["image"].view(np.uint64) == ref_img.view(np.uint64)).all(), "image on a rank differs from the single-GPU image"
        for k in ("n", "r2", "flux"):
            a, b = np.ascontiguousarray(g[k]), np.ascontiguousarray(ref[k])
            assert (a.view(np.uint64) == b.view(np.uint64)).all() if a.dtype.kind == "f" else (a == b).all(), k


def test_allgather_flow_over_rccl_with_one_rank(tmp_path):  # also: dist.all_to_all (ragged lists) and the int64 all-reduce on RCCL
    """The exchange of ShardedSppm on the real backend: a process group of ONE rank over RCCL (all a one-GPU box allows)
    runs the pipelined flow -- block traced ahead, packed rows, counts and records all-gathered on the side stream,
    append, gather, masked all-reduce of the image -- and must reproduce the plain single-GPU passes bit for bit."""
    W, H, M, passes = 96, 64, 24000, 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 1000), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path), "nccl1", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    ref_img, ref = _single(W, H, M, passes)
    g = np.load(tmp_path / "rank0.npz")
    assert g["mask"].all()
    assert (g["image"].view(np.uint64) == ref_img.view(np.uint64)).all()
    for k in ("n", "r2", "flux"):
        a, b = np.ascontiguousarray(g[k]), np.ascontiguousarray(ref[k])
        assert (a.view(np.uint64) == b.view(np.uint64)).all() if a.dtype.kind == "f" else (a == b).all(), k
