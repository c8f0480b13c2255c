"""RCCL smoke + exchange timing on ONE GPU (world = 1 over the `nccl` backend, i.e. RCCL's own launch
path without a peer): a config3-sized WithinPlan is searched with the device-side exchange
(multimoda_rs_amd.distributed.search_device: export_cost -> all_reduce(MIN) -> export_keys ->
all_reduce(MIN) -> one D2H -> commit) and, for comparison, the gather exchange through the host.
Prints the per-level time of each piece measured with the plan's search already finished, so what is
timed is the exchange alone.  Results must equal plan.run().
Usage: python tools/rccl_smoke.py   (on a GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np
import torch
import torch.distributed as dist

import __graft_entry__ as ge

ge.build()
import multimoda_rs_amd as mm
from multimoda_rs_amd import distributed as D

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = mm.Engine(0)
base = mm.synthetic_case(512, 501)


def plan():
    return mm.WithinPlan(eng, [g.copy() for g in base], 0.5, 180.0, True, 501, precision=mm.MM_PRECISION_F32_FAST)


ref = plan()
ref_logs, _, _ = ref.run()
ref.close()

# device exchange, piece by piece
for rep in range(3):
    p = plan()
    n_jobs, n_levels, tol = p.dims()
    b = D._buffers(p, n_jobs)
    p.level_launch(0)
    eng.synchronize()                                   # the search is done: time the exchange alone
    t = [time.perf_counter()]
    p.level_export_cost(0, b.cost.data_ptr()); t.append(time.perf_counter())
    D._all_reduce_min(b.cost, None, b.stream); t.append(time.perf_counter())
    p.level_export_keys(0, b.cost.data_ptr(), b.keys.data_ptr()); t.append(time.perf_counter())
    D._all_reduce_min(b.keys, None, b.stream); t.append(time.perf_counter())
    p.level_commit_dev(0, b.cost.data_ptr(), b.keys.data_ptr()); t.append(time.perf_counter())
    logs, _, unres = p.walk()
    assert [list(l) for l in logs] == [list(l) for l in ref_logs] and unres == 0
    p.close()
    names = ("export_cost (enqueue)", "all_reduce cost (enqueue)", "export_keys (enqueue)", "all_reduce keys (enqueue)",
             "D2H + sync + commit")
    print("device exchange, level 0, %d jobs: " % n_jobs + ", ".join("%s %.3f ms" % (n, 1e3 * (t[i + 1] - t[i])) for i, n in enumerate(names))
          + "; total %.3f ms" % (1e3 * (t[-1] - t[0])))

# the same exchange issued by the LIBRARY on its own communicator (mm_comm_*, mm_within_plan_search_sharded's pieces)
comm = mm.Comm(mm.Comm.unique_id(), 0, 1, 0)
for rep in range(4):
    p = plan()
    n_jobs, n_levels, tol = p.dims()
    b = D._buffers(p, n_jobs)
    p.level_launch(0)
    eng.synchronize()
    t = [time.perf_counter()]
    p.level_export_cost(0, b.cost.data_ptr()); t.append(time.perf_counter())
    comm.all_reduce_min_f64(b.cost.data_ptr(), n_jobs, eng.stream); t.append(time.perf_counter())
    p.level_export_keys(0, b.cost.data_ptr(), b.keys.data_ptr()); t.append(time.perf_counter())
    comm.all_reduce_min_i64(b.keys.data_ptr(), 3 * n_jobs, eng.stream); t.append(time.perf_counter())
    p.level_commit_dev(0, b.cost.data_ptr(), b.keys.data_ptr()); t.append(time.perf_counter())
    logs, _, unres = p.walk()
    assert [list(l) for l in logs] == [list(l) for l in ref_logs] and unres == 0
    p.close()
    names = ("export_cost (enqueue)", "ncclAllReduce cost (enqueue)", "export_keys (enqueue)", "ncclAllReduce keys (enqueue)",
             "D2H + sync + commit")
    print("library exchange (mm_comm), level 0, %d jobs: " % n_jobs + ", ".join("%s %.3f ms" % (n, 1e3 * (t[i + 1] - t[i])) for i, n in enumerate(names))
          + "; total %.3f ms" % (1e3 * (t[-1] - t[0])))
for rep in range(3):
    p = plan()
    p.level_launch(0)
    eng.synchronize()
    t0 = time.perf_counter()
    p.search_sharded(comm)                              # level 0 is already launched: exports, collectives, commit only
    t1 = time.perf_counter()
    logs, _, unres = p.walk()
    assert [list(l) for l in logs] == [list(l) for l in ref_logs] and unres == 0
    p.close()
    print("mm_within_plan_search_sharded after the launch has finished (exchange alone): %.3f ms" % (1e3 * (t1 - t0)))

# gather exchange through the host
for rep in range(3):
    p = plan()
    n_jobs, n_levels, tol = p.dims()
    t0 = time.perf_counter()
    local = p.level_local(0, n_jobs)                    # search + fetch
    t1 = time.perf_counter()
    world_backup = D.world_size
    rec = torch.from_numpy(np.stack([local["cost"], local["angle"], local["idx"].astype(np.float64),
                                     local["uniform"].astype(np.float64)], axis=0)).cuda()
    out = [torch.empty_like(rec)]
    dist.all_gather(out, rec)
    g = torch.stack(out, dim=0).cpu().numpy()
    ok, angle, _i, _c = D.merge_shards(1, g[:, 0, :], g[:, 3, :].astype(np.int32), g[:, 1, :], g[:, 2, :].astype(np.int32), tol)
    p.level_commit(0, ok, angle)
    t2 = time.perf_counter()
    logs, _, unres = p.walk()
    assert [list(l) for l in logs] == [list(l) for l in ref_logs] and unres == 0
    p.close()
    print("gather exchange, level 0: search + fetch %.3f ms, pack + H2D + all_gather + D2H + merge + commit %.3f ms" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1)))

# the whole sharded search through WithinPlan.search (device exchange), against run()
for rep in range(3):
    p = plan()
    t0 = time.perf_counter()
    D.search_device(p, None)
    t1 = time.perf_counter()
    p.walk()
    p.close()
    q = plan()
    t2 = time.perf_counter()
    q.search()
    t3 = time.perf_counter()
    q.walk()
    q.close()
    r = plan()
    t4 = time.perf_counter()
    r.search_sharded(comm)
    t5 = time.perf_counter()
    r.walk()
    r.close()
    print("search with the device exchange through torch (world = 1 over RCCL) %.3f ms; through the library's communicator "
          "%.3f ms; plain single-rank search %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t5 - t4), 1e3 * (t3 - t2)))
comm.close()

dist.barrier()
dist.destroy_process_group()
eng.close()
print("RCCL_SMOKE_OK")
