#!/usr/bin/env python3
"""Probe: raw pinned-host -> device copy rate, one stream vs several streams with copies in flight at once."""
import time, torch
n = 236 * 1024 * 1024          # one 256-frame batch of 640x480 BGR
host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
dev = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        for i in range(16):
            with torch.cuda.stream(streams[i % ns]):
                dev[i % 4].copy_(host[i % 4], non_blocking=True)
        torch.cuda.synchronize(); dt = time.time() - t
    print("%d stream(s): %.1f GB/s" % (ns, 16 * n / dt / 1e9))
# split one batch copy in two halves on two streams
s2 = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize(); t = time.time()
for i in range(16):
    h, d = host[i % 4], dev[i % 4]
    for j in range(2):
        with torch.cuda.stream(s2[j]):
            d[j * n // 2:(j + 1) * n // 2].copy_(h[j * n // 2:(j + 1) * n // 2], non_blocking=True)
torch.cuda.synchronize(); dt = time.time() - t
print("halves on two streams: %.1f GB/s" % (16 * n / dt / 1e9))
