"""Measured HBM bandwidth of the box (confirm the roofline denominator):
device-to-device copy, STREAM triad a = b + s*c and a fill, on 1 GiB fp32 buffers.

    python tools/hbm_triad.py

Prints one JSON line (GB/s, algorithmic bytes: copy 2x, triad 3x, fill 1x the buffer size).
The roofline in bench.py keeps the guide's 8 TB/s peak as its denominator; this number is the
achievable streaming rate next to it.
"""
import json

import torch

dev = torch.device("cuda:0")
n = 1 << 28  # 1 GiB of fp32
a, b, c = (torch.empty(n, device=dev) for _ in range(3))
b.fill_(1.0)
c.fill_(2.0)


def timed(fn, bytes_moved, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return bytes_moved / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9


out = {"buffer_bytes": 4 * n, "device": torch.cuda.get_device_name(0),
       "copy_GBps": round(timed(lambda: a.copy_(b), 8 * n)),
       "triad_GBps": round(timed(lambda: torch.add(b, c, alpha=3.0, out=a), 12 * n)),
       "fill_GBps": round(timed(lambda: a.fill_(0.5), 4 * n))}
print(json.dumps(out))
