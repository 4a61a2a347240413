"""Achievable HBM bandwidth on this device for the access shapes of the iteration's kernels: a streaming read of a
tensor the size of the tile stream (157 MB, B = 4096, N = 200, fp32), a streaming write of the same size and a copy,
timed with HIP events over 50 launches after 5 warm-ups.  The buffers rotate through a pool larger than the 256 MB
Infinity Cache so that every pass comes from / goes to HBM.  (torch elementwise kernels: plumbing, not the product.)"""
import torch, json
dev = "cuda:0"
n = 4096 * 200 * 48            # fp32 elements = 157.3 MB
pool = [torch.empty(n, device=dev, dtype=torch.float32).normal_() for _ in range(6)]
out = {}
def timed(fn, bytes_per):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): fn(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    return {"us": round(us, 2), "GBs": round(bytes_per / us * 1e-3, 1)}
out["read_157MB"] = timed(lambda i: pool[i % 6].sum(), n * 4)
out["write_157MB"] = timed(lambda i: pool[i % 6].fill_(1.0), n * 4)
out["copy_157MB"] = timed(lambda i: pool[i % 6].copy_(pool[(i + 3) % 6]), 2 * n * 4)
print(json.dumps(out))
