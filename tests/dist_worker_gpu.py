"""Worker of tests/test_gpu_distributed.py: one rank of a ShardedSppm run (both ranks share GPU 0; the process
group is gloo, which carries CUDA tensors through the host -- RCCL needs one GPU per rank)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, exchange, ownership = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if exchange == "nccl1":  # ONE rank over RCCL: the collectives, streams and buffers of the all-gather flow on the real backend
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from raytracer_amd.parallel import ShardedSppm
        W, H, M_rank, passes = 96, 64, 24000, 3
        job = ShardedSppm(W, H, photons_per_rank=M_rank, rank=0, world=1)
        job.exchange, job.replicated_trace = "allgather", False   # a world of one still exchanges with itself
        job.batch_photons = 8192
        job.total_passes = passes
        for i in range(passes):
            job.step(i)
        job.ctx.check()
        img = job.image * job.ctx.owned_mask().to(job.image.dtype).unsqueeze(1)
        dist.all_reduce(img, op=dist.ReduceOp.SUM)
        hp = job.ctx.hitpoints()
        np.savez(os.path.join(out_dir, "rank0.npz"), image=img.cpu().numpy(), n=hp["n"], r2=hp["r2"], flux=hp["flux"], mask=job.ctx.owned_mask().cpu().numpy())
        job.close()
        # the ragged all-to-all of the exchange by destination and the int64 all-reduce of the per-pass rule, on RCCL
        from raytracer_amd import parallel
        src = torch.arange(70, dtype=torch.float64, device="cuda").reshape(7, 10)
        dst = torch.empty((7, 10), dtype=torch.float64, device="cuda")
        table = parallel.exchange_count_rows([7], 1, torch.device("cuda"))
        parallel.all_to_all_rows([dst[:table[0][0]]], [src], 0)
        dep = torch.arange(4 * 64, dtype=torch.int64, device="cuda").reshape(4, 64) + (1 << 40)
        want = dep.clone()
        parallel.all_reduce_sum(dep)
        torch.cuda.synchronize()
        assert table == [[7]] and torch.equal(dst, src) and torch.equal(dep, want)
        # the sliced all-reduce of the per-pass rule on RCCL: asynchronous collectives behind each slice's export, the
        # context's stream waiting for each sum (a world of one still runs every call) == the plain passes, bit for bit
        from raytracer_amd import sppm
        job = ShardedSppm(W, H, photons_per_rank=M_rank, rank=0, world=1, rule=1, slices=3)
        plain = sppm.SppmContext(sppm.make_config(W, H, photons_per_pass=M_rank, update_rule=1))
        job.time_stages = True
        for i in range(passes):
            job.ctx.eye_pass(i); job.ctx.photon_pass(i, M_rank)
            img = job._allreduce_and_end_pass(i, True)
            plain.eye_pass(i); plain.photon_pass(i, M_rank)
            ref = plain.end_pass_resolve(i, M_rank)
            assert torch.equal(img, ref), f"pass {i}: image of the sliced flow"
        job.ctx.check(); plain.check()
        a, b = job.ctx.hitpoints(), plain.hitpoints()
        for k in ("n", "r2", "flux"):
            assert (np.ascontiguousarray(a[k]).view(np.uint8) == np.ascontiguousarray(b[k]).view(np.uint8)).all(), k
        assert len(job.stage_events) == 3 * passes and all(e0.elapsed_time(e1) >= 0.0 for _, e0, e1 in job.stage_events)
        job.close(); plain.close()
        dist.destroy_process_group()
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd.parallel import ShardedPpm, ShardedSppm
    W, H, M_rank, passes = 96, 64, 24000, 3
    if exchange == "ppm":  # classic PPM: one run of `passes` batches; after finish() every rank holds the whole state
        job = ShardedPpm(W, H, M_rank, passes, rank=rank, world=world, supersamples=2)
        for i in range(passes):
            job.step(i)
        mask = job.ctx.owned_mask().cpu().numpy()
        img = job.finish().cpu().numpy()
        job.ctx.check()
        hp = job.ctx.hitpoints()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=img, n=hp["n"], r2=hp["r2"], flux=hp["flux"], mask=mask)
        job.close()
        dist.destroy_process_group()
        return
    if exchange == "allreduce":  # north_star's split: per-pass rule, photons sharded, hitpoints replicated, one all-reduce per pass
        job = ShardedSppm(W, H, photons_per_rank=M_rank, rank=rank, world=world, rule=1)
        for i in range(passes):
            job.step(i)
        job.ctx.check()
        hp = job.ctx.hitpoints()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=job.gather_image().cpu().numpy(), n=hp["n"], r2=hp["r2"], flux=hp["flux"],
                 exchanged=job.exchanged_bytes)
        job.close()
        dist.destroy_process_group()
        return
    job = ShardedSppm(W, H, photons_per_rank=M_rank, rank=rank, world=world, exchange=exchange, ownership=ownership)
    job.total_passes = passes
    for i in range(passes):
        job.step(i)
    job.ctx.check()
    img = job.gather_image().cpu().numpy()
    hp = job.ctx.hitpoints()
    mask = job.ctx.owned_mask().cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), image=img, n=hp["n"], r2=hp["r2"], flux=hp["flux"], mask=mask, exchanged=job.exchanged_bytes)
    job.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
