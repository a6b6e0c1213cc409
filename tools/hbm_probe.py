#!/usr/bin/env python3
"""Dev tool: what a plain streaming read / copy of classifier-weight size achieves on this chip when the data cannot come from the
Infinity Cache (rotating through 1.2 GB of distinct buffers): the ceiling the HBM-bound rows of the bench are read against."""
import torch
dev = torch.device("cuda:0")
n = 18432 * 1024                      # 75.5 MB fp32 = classifier.0.weight
bufs = [torch.randn(n, device=dev) for _ in range(16)]
out = torch.empty(n, device=dev)
def timed(fn, reps=64):
    for i in range(8):
        fn(bufs[i % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(bufs[i % 16])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = timed(lambda b: b.sum())
print(f"torch sum over 75.5 MB (read only): {t:.1f} us = {n * 4 / t / 1e6:.2f} TB/s")
t = timed(lambda b: torch.add(b, 1.0, out=out))
print(f"torch add into a second buffer (75.5 MB read + 75.5 MB write): {t:.1f} us = {2 * n * 4 / t / 1e6:.2f} TB/s")
t = timed(lambda b: out.fill_(0.0))
print(f"torch fill of 75.5 MB (write only): {t:.1f} us = {n * 4 / t / 1e6:.2f} TB/s")
t = timed(lambda b: b.fill_(0.0))
print(f"torch fill of 75.5 MB, rotating through 1.2 GB (write only, past the Infinity Cache): {t:.1f} us = {n * 4 / t / 1e6:.2f} TB/s")
half = n // 2
t = timed(lambda b: b[:half].fill_(0.0))
print(f"torch fill of 37.7 MB, rotating: {t:.1f} us = {half * 4 / t / 1e6:.2f} TB/s")
t = timed(lambda b: torch.add(b[:half], 1.0, out=bufs[(int(b.data_ptr()) // 7) % 16][half:]))
print(f"torch add 37.7 MB read + 37.7 MB write, both rotating: {t:.1f} us = {2 * half * 4 / t / 1e6:.2f} TB/s")
