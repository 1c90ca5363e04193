"""Launched by tests/test_rccl_one_gpu.py under torch.distributed.run (not collected by pytest): the caller's own HIP kernel as
operator (dav_set_operator_device; tests/helpers/user_operator.hip) over a REAL RCCL communicator whose ranks share GPU 0
(NCCL_HOSTID per rank, loopback socket transport - bench.py: DAVIDSON_TRANSPORT=rccl-one-gpu).  The block the callback is handed
is gathered column by column in one grouped ncclAllGather.  Rank 0 prints one JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ["NCCL_HOSTID"] = f"davidson-rehearsal-host-{rank}"
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("DAVIDSON_COLLECTIVE_TIMEOUT", "120")
import numpy as np                                  # noqa: E402
import torch                                        # noqa: E402
import torch.distributed as dist                    # noqa: E402
import fortran_davidson_amd as fd                   # noqa: E402

dist.init_process_group(backend="gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
fd.hip_lib()
user = C.CDLL(os.path.join(ROOT, "fortran_davidson_amd", "lib", "test", "libuser_operator.so"))
user.user_op_create.restype = C.c_void_p
user.user_op_create.argtypes = [C.c_double, C.c_double, C.c_double]
n, lowest, gev = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1"
eng = fd.DavidsonEngine(n, lowest, None, gev=gev, device=0, rank=rank, nranks=world)
ident = [fd.CEngine.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ident, src=0)
eng.comm_init(ident[0])
eng.set_device_operator(1, user.user_op_apply, user.user_op_create(1.0, 1.0, 0.3), 1.0 + np.arange(n, dtype=np.float64))
if gev:
    eng.set_device_operator(2, user.user_op_apply, user.user_op_create(1.0, 0.0, 0.05), np.ones(n))
lam, _, it = eng.solve("DPR", 200, 1e-8, want_vectors=False)
st = eng.c.stats()
dist.barrier()
if rank == 0:
    print(json.dumps({"ranks": world, "n": n, "iters": int(it), "eigenvalues": [float(x) for x in lam], "collectives": int(st.collectives)}), flush=True)
eng.close()
dist.destroy_process_group()
