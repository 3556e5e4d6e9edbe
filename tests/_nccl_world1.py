"""Helper of tests/test_gpu_multirank.py (not a test): bench.py's collective calls over the "nccl" backend (= RCCL) in a world of
one — process-group creation bound to the device, barrier, the ragged record gather and the max over ranks on device tensors."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from speech_signal_processing_amd.dist import all_gather_rows, max_over_ranks, pack_records, unpack_records  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
dist.barrier()
am = torch.arange(37, dtype=torch.int32, device=device)
rec = all_gather_rows(pack_records(am, am.float() * 0.5, -am.float()), force=True)
a, b, u = unpack_records(rec)
ok = bool(torch.equal(a, am.repeat(world)) and torch.equal(b, (am.float() * 0.5).repeat(world)) and rec.is_cuda)
mx = max_over_ranks(1.25 + rank, device, force=True)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(json.dumps({"ok": ok, "backend": "nccl", "rows": int(rec.shape[0]), "max": mx}))
