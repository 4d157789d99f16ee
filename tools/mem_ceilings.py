#!/usr/bin/env python3
"""Device-memory rates of the box with plain torch kernels over 4 GiB tensors: fill (write only), sum (read only), copy
(1 read : 1 write).  Round 2 on an MI355X box: 6.8 / 4.0 (torch's reduction, not a ceiling) / 4.9 TB/s."""
import torch

n = 4 * 1024 ** 3 // 4            # 4 GiB of fp32
a = torch.empty(n, device="cuda")
b = torch.empty(n, device="cuda")
a.normal_()


def timed(fn, nbytes, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return nbytes / ms / 1e6, ms


for name, fn, nb in (("fill (write only)", lambda: b.fill_(1.0), 4 * n),
                     ("sum (read only)", lambda: a.sum(), 4 * n),
                     ("copy (1R:1W)", lambda: b.copy_(a), 8 * n)):
    gbps, ms = timed(fn, nb)
    print("%-60s %8.0f GB/s  (%.2f ms)" % (name, gbps, ms), flush=True)
