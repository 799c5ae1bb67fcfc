#!/usr/bin/env python
"""What RCCL's flight recorder shows of the watchdog's work list on this box (lsps_amd/dist.py: drain_watchdog):
one rank on RCCL, a few eager all-reduces, then the dump polled until no entry is active.  Run with and without
TORCH_NCCL_TRACE_BUFFER_SIZE to see torch's default."""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29531')
print("TORCH_NCCL_TRACE_BUFFER_SIZE =", os.environ.get('TORCH_NCCL_TRACE_BUFFER_SIZE'))
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0), rank=0, world_size=1)
from torch._C._distributed_c10d import _dump_nccl_trace as dump  # noqa: E402
x = torch.ones(1 << 20, device='cuda')
for _ in range(5):
    dist.all_reduce(x)
t0 = time.time()
d = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=False))
print("keys:", sorted(d.keys()) if isinstance(d, dict) else type(d))
ent = d.get('entries') if isinstance(d, dict) else None
print("entries (all):", None if ent is None else len(ent))
if ent:
    print("last entry:", {k: ent[-1][k] for k in ent[-1] if k in ('state', 'profiling_name', 'collective_seq_id', 'retired', 'time_discovered_completed_ns')})
for i in range(60):
    a = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=True)).get('entries')
    print("t=%.3f s active entries: %s, not retired: %s" % (time.time() - t0, None if a is None else len(a), len([e for e in pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries") if not e.get("retired")])))
    if not [e for e in pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries") if not e.get("retired")]:
        break
    time.sleep(0.02)
from lsps_amd import dist as ldist  # noqa: E402
for _ in range(3):
    dist.all_reduce(x)
t1 = time.time()
print("drain_watchdog ->", ldist.drain_watchdog(), "in %.3f s" % (time.time() - t1))
dist.destroy_process_group()
