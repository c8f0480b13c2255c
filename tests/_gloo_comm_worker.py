"""Worker of test_native_comm_setup_is_decided_by_all_ranks: two gloo ranks ask for the library's RCCL communicator; rank 1
cannot load RCCL (MM_RCCL_LIB names a file that does not exist).  Both must raise, promptly -- rank 0, which could load
it, must not be left waiting for a broadcast that never comes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

rank = int(os.environ["RANK"])
if rank == 1:
    os.environ["MM_RCCL_LIB"] = "/nonexistent/librccl.so"
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dist.init_process_group("gloo")
torch.cuda.current_device = lambda: 0          # no device on the CPU box; the setup must fail before it needs one
from multimoda_rs_amd import distributed as D  # noqa: E402

try:
    D.native_comm()
    print(f"rank {rank}: NO ERROR")
except RuntimeError as ex:
    print(f"rank {rank}: COMM_SETUP_REFUSED {ex}")
dist.barrier()
dist.destroy_process_group()
