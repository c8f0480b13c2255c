"""Worker of test_gpu_sharded.py::test_sharded_finish_world2_gloo: two `gloo` ranks (sharing the one GPU) search a 4-pullback
case sharded, then FINISH it sharded -- pullback g walked on rank g mod 2, between pair k aligned on rank k mod 2, logs,
coordinates and rotations broadcast from the owner (multimoda_rs_amd.distributed.walk_sharded / align_between_sharded) --
and every rank compares everything it holds with the CPU oracle's sequential 4-phase alignment."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import multimoda_rs_amd as mm
    from multimoda_rs_amd import distributed as D
    from oracle import oracle as orc
    from helpers import geoms_equal, to_oracle

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    fin = dist.new_group(backend="gloo")                 # the finishing collectives on a group of their own, as bench.py does
    torch.cuda.set_device(0)
    eng = mm.Engine(0)
    for bruteforce, step, rng_deg, ss, prec in ((True, 1.0, 180.0, 501, mm.MM_PRECISION_F32_MATRIX), (False, 0.5, 90.0, 200, mm.MM_PRECISION_F32_BOUNDED)):
        geoms = [mm.synthetic_pullback(f, 501, pullback_id=i) for i, f in enumerate((10, 7, 12, 6))]
        og = [to_oracle(orc, g) for g in geoms]
        plan = mm.WithinPlan(eng, geoms, step, rng_deg, bruteforce, ss, precision=prec)
        plan.set_shard(rank, world)
        plan.search()
        logs, evals, unres = D.walk_sharded(plan, group=fin)
        plan.close()
        a, b, c, d = geoms
        r1, e1 = D.align_between_sharded(eng, [(a, b), (c, d)], rng_deg, step, ss, prec, group=fin)
        r2, e2 = D.align_between_sharded(eng, [(a, c), (b, d)], rng_deg, step, ss, prec, group=fin)
        ologs = [orc.align_within_chain(o, step, rng_deg, bruteforce, ss, n_threads=4) for o in og]
        orot = [orc.align_between(og[i], og[j], rng_deg, step, ss, n_threads=4) for i, j in ((0, 1), (2, 3), (0, 2), (1, 3))]
        for g, o, lg, ol in zip(geoms, og, logs, ologs):
            assert lg == ol, (rank, step)
            assert geoms_equal(g, o), (rank, step)
        assert list(np.concatenate([r1, r2])) == orot, (rank, list(np.concatenate([r1, r2])), orot)
        assert unres == 0
        if bruteforce:
            assert evals == sum(orc.count_evals(step, rng_deg, True) * (g.n_frames - 1) for g in geoms), (rank, evals)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("FINISH_WORKER_OK")


if __name__ == "__main__":
    main()
