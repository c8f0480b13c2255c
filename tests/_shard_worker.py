"""Worker of test_gpu_sharded.py::test_run_sharded_world2_gloo_on_the_gpu: every rank of a
torch.distributed `gloo` group (ranks share the one GPU of the box) owns its share of the candidate
axis of a WithinPlan and drives plan.run_sharded() -- local search on the device, exchange, merge,
commit, chain walk -- and then compares its own result with the CPU oracle's sequential chain.
MM_EXCHANGE = gather | device selects the exchange (multimoda_rs_amd.distributed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import multimoda_rs_amd as mm
    from oracle import oracle as orc
    from helpers import geoms_equal, to_oracle

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    eng = mm.Engine(0)
    eng.set_bound_min_candidates(0)
    cases = [(True, 1.0, 180.0, 501, mm.MM_PRECISION_F32_FAST), (False, 0.05, 45.0, 200, mm.MM_PRECISION_F32_FAST),
             (True, 0.5, 180.0, 501, mm.MM_PRECISION_F32_BOUNDED), (True, 45.0, 90.0, 64, mm.MM_PRECISION_F32)]
    for bruteforce, step, rng_deg, ss, prec in cases:
        geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((10, 7, 12, 6))]
        og = [to_oracle(orc, g) for g in geoms]
        plan = mm.WithinPlan(eng, geoms, step, rng_deg, bruteforce, ss, precision=prec)
        plan.set_shard(rank, world)
        logs, evals, unres = plan.run_sharded()
        plan.close()
        for g, o, lg in zip(geoms, og, logs):
            ol = orc.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=4)
            assert lg == ol, (rank, step)
            assert geoms_equal(g, o), (rank, step)
        if bruteforce:
            assert evals == sum(orc.count_evals(step, rng_deg, True) * (g.n_frames - 1) for g in geoms), (rank, evals)
        assert unres == 0
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("SHARD_WORKER_OK")


if __name__ == "__main__":
    main()
