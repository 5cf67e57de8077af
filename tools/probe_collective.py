#!/usr/bin/env python3
"""Host and device cost of the collectives the N > 1 path issues, at world 1 on one GPU (RCCL through torch.distributed):
what one all_gather_into_tensor call costs the enqueuing thread, with the stream idle and with kernels queued in front of it."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
side = torch.cuda.Stream()
busy = torch.zeros(64 << 20, device="cuda")
for nbytes in (1 << 10, 1 << 17, 1 << 20, 10 << 20):
    src = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    for mode in ("idle", "queued", "ext_stream"):
        ts = []
        for it in range(12):
            torch.cuda.synchronize()
            if mode != "idle":
                with torch.cuda.stream(side):
                    for _ in range(20):
                        busy.add_(1.0)      # ~20 x 60 us of queued work in front of the collective
            t0 = time.perf_counter()
            if mode == "ext_stream":
                ext = torch.cuda.ExternalStream(side.cuda_stream)
                with torch.cuda.stream(ext):
                    dist.all_gather_into_tensor(dst, src, async_op=True).wait()
            elif mode == "queued":
                with torch.cuda.stream(side):
                    dist.all_gather_into_tensor(dst, src, async_op=True).wait()
            else:
                dist.all_gather_into_tensor(dst, src, async_op=True).wait()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if it >= 2:
                ts.append((t1 - t0, t2 - t0))
        print("all_gather %8d B  %-10s host %.1f us   until done %.1f us" % (nbytes, mode, 1e6 * sum(t[0] for t in ts) / len(ts), 1e6 * sum(t[1] for t in ts) / len(ts)))
t = torch.zeros(4000, dtype=torch.int32, device="cuda")
ts = []
for it in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    if it >= 2:
        ts.append((t1 - t0, time.perf_counter() - t0))
print("all_reduce 16 KB host %.1f us until done %.1f us" % (1e6 * sum(a for a, _ in ts) / len(ts), 1e6 * sum(b for _, b in ts) / len(ts)))
dist.destroy_process_group()
