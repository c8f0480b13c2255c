import os, sys, threading, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge; ge.build()
import multimoda_rs_amd as mm
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = mm.Engine(0)
stop = False
def hammer():
    rng = np.random.default_rng(0)
    ang, _, _ = mm.search_angles(1.0, 180.0)
    while not stop:
        a = rng.normal(0, 2, (300, 2)); b = rng.normal(0, 2, (300, 2))
        eng.best_rotation(a, b, ang, (0.0, 0.0))
th = threading.Thread(target=hammer); th.start()
t0 = time.perf_counter()
for i in range(200):
    rec = torch.from_numpy(np.random.rand(4, 2044)).cuda()
    out = [torch.empty_like(rec)]
    dist.all_gather(out, rec)
    g = rec[0].clone(); dist.all_reduce(g, op=dist.ReduceOp.MIN)
    _ = torch.stack(out).cpu().numpy(); _ = g.cpu()
dt = (time.perf_counter() - t0) / 200
stop = True; th.join()
dist.barrier(); dist.destroy_process_group(); eng.close()
print("rccl world=1 exchange with a concurrent search thread: %.3f ms per exchange" % (1e3 * dt))
