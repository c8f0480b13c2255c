"""Worker of test_distributed_gloo.py::test_broadcast_state_world3_gloo: the collective of the sharded finish on the CPU -- three
`gloo` ranks hold different geometries and logs; `distributed.broadcast_state` hands the owner's to the others in one broadcast
per geometry, in place, for numpy arrays and ctypes log buffers alike."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import multimoda_rs_amd as mm  # noqa: E402
from multimoda_rs_amd import _native as N  # noqa: E402
from multimoda_rs_amd import distributed as D  # noqa: E402

fin = dist.new_group(backend="gloo")
ok = True
for g in range(5):
    owner = g % world
    truth = mm.synthetic_pullback(4 + g, 64, pullback_id=g, seed=100 + g)            # what the owner holds after "its walk"
    mine = truth.copy() if rank == owner else mm.synthetic_pullback(4 + g, 64, pullback_id=g, seed=999)   # everybody else: something else
    logs = (N.MMAlignLog * (3 + g))()
    if rank == owner:
        for i in range(3 + g):
            logs[i].contour_id, logs[i].matched_to, logs[i].rot_deg, logs[i].tx = i + 1, i, 0.25 * i - g, 1e-3 * g
    D.broadcast_state(D._mutable_arrays(mine) + [logs], owner, fin)
    ok = ok and np.array_equal(mine.lumen, truth.lumen) and np.array_equal(mine.centroids, truth.centroids)
    ok = ok and (truth.cath is None or np.array_equal(mine.cath, truth.cath)) and np.array_equal(mine.ref, truth.ref)
    ok = ok and all(logs[i].contour_id == i + 1 and logs[i].rot_deg == 0.25 * i - g and logs[i].tx == 1e-3 * g for i in range(3 + g))
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank}: {'BCAST_OK' if ok else 'BCAST_MISMATCH'}")
