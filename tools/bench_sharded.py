#!/usr/bin/env python3
"""BASELINE.json configs[3]: one long sequence sharded over the GPUs of a node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
        --master-port 29533 tools/bench_sharded.py            # N_PER_GPU=1000000 windows per rank

Rank g generates ITS rows of the (G*N_PER_GPU + 149, 54) sequence (rows [g*N, (g+1)*N + 149): the
149-row halo is regenerated, not communicated), runs dce_infer_sequence on them, and the (N,16)
logits + (N,4) contacts are gathered to rank 0 with ONE RCCL gather each (64 MB + 4 MB per rank at
N = 1e6).  Prints windows/s including and excluding the gather.  Works with G = 1 as well."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from deep_contact_estimator_amd import contact_cnn, synth
from deep_contact_estimator_amd.distributed import gather_rows

N = int(os.environ.get("N_PER_GPU", 1_000_000))
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group("nccl", device_id=dev)
m = contact_cnn(device=local, max_batch=32768)
m.load_state_dict(synth.make_state_dict(1)).eval()
# row r of the global sequence is a pure function of r: ranks regenerate their halo rows identically
g = torch.Generator(device=dev)
def rows(lo, hi, chunk=1 << 20):
    parts = []
    for a in range(lo - lo % chunk, hi, chunk):          # chunk-aligned streams -> identical overlaps
        g.manual_seed(1000 + a // chunk)
        blk = torch.randn((chunk, 54), generator=g, device=dev, dtype=torch.float32)
        parts.append(blk[max(lo - a, 0): min(hi - a, chunk)])
    return torch.cat(parts)
seq = rows(rank * N, (rank + 1) * N + 149)
m.infer_sequence(seq[:4096 + 149]); torch.cuda.synchronize()
if world > 1: dist.barrier()
t0 = time.perf_counter()
out = m.infer_sequence(seq)
torch.cuda.synchronize()
t1 = time.perf_counter()
if world > 1:
    logits = gather_rows(out["logits"], dst=0); contacts = gather_rows(out["contacts"], dst=0)
    torch.cuda.synchronize(); dist.barrier()
else:
    logits, contacts = out["logits"], out["contacts"]
t2 = time.perf_counter()
if world > 1:
    t = torch.tensor([t1 - t0, t2 - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); tc, tt = t.tolist()
else:
    tc, tt = t1 - t0, t2 - t0
if rank == 0:
    assert logits.shape == (world * N, 16) and contacts.shape == (world * N, 4)
    print(json.dumps({"workload": f"configs[3]: {world} x {N} windows, halo-sharded, one gather of logits+contacts to rank 0",
                      "n_gpus": world, "windows_per_s_compute": world * N / tc, "windows_per_s_incl_gather": world * N / tt,
                      "gather_ms": (tt - tc) * 1e3, "gathered_MB": world * N * 20 / 1e6}))
if world > 1:
    dist.destroy_process_group()
