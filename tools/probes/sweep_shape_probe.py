#!/usr/bin/env python3
"""How the two sweep kernels respond to the number of instance streams walking at once and to the span of one instance's
records: ANYmal trot at horizons N = 6 / 10 / 20 / 40 and batches 512 .. 8192, backward and forward timed alone
(min of 12), reported as time per grid point per resident wave and as algorithmic bytes per second.  A kernel bound by the
memory system's throughput gets faster per wave as the batch shrinks; one bound by its own chain (latency / issue) does
not; a kernel held back by address translation gets faster per byte as the records of an instance span fewer pages."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from robotoc_amd import capi, problems as pr
from robotoc_amd.types import BUF_DIR, BUF_DX0, BUF_KKT, BUF_RIC

dev = "cuda:0"
print("%5s %6s | %9s %9s %9s | %9s %9s %9s | record span of an instance" % (
    "N", "batch", "bwd ms", "us/grid", "GB/s", "fwd ms", "us/grid", "GB/s"))
for N in (40, 20, 10, 6):
    dims, grids, _ = pr.config_anymal_trot(N=N)
    n = len(grids)
    for batch in (8192, 4096, 2048, 1024, 512):
        ctx = capi.Context(dims, n, batch, 0)
        L = ctx.L
        ctx.set_grid(grids)
        z = lambda w: torch.zeros((batch, n, getattr(L, w).stride), dtype=torch.float64, device=dev)
        kkt = pr.make_kkt_batch_unique(L, grids, batch, seed=0, backend="torch", device=dev, out=z("kkt"))
        dx0 = pr.make_dx0_unique(L, batch, seed=0, backend="torch", device=dev).contiguous()
        ric, d = z("ric"), z("dir")
        for b_, t_ in ((BUF_KKT, kkt), (BUF_DX0, dx0), (BUF_RIC, ric), (BUF_DIR, d)):
            ctx.bind(b_, t_.data_ptr())
        torch.cuda.synchronize()
        for _ in range(3):
            ctx.riccati_backward()
            ctx.riccati_forward()
        ctx.sync()
        tb = min(ctx.time_phase(0, 1) for _ in range(12))
        tf = min(ctx.time_phase(1, 1) for _ in range(12))
        # algorithmic bytes as bench.py counts them (regular grid point: 5244 doubles backward, 3346 forward; the few
        # event grid points differ by a few per cent -- the same count on every line of this table)
        bb, fb = 8.0 * 5244 * n * batch, 8.0 * 3346 * n * batch
        rounds_b = max(1.0, batch / 2048.0)   # backward: 2 waves per SIMD resident; forward: 16 per CU
        rounds_f = max(1.0, batch / 4096.0)
        print("%5d %6d | %9.4f %9.2f %9.0f | %9.4f %9.2f %9.0f | kkt %.2f MB, ric %.2f MB   status != 0: %d" % (
            N, batch, tb, tb * 1e3 / n / rounds_b, bb / tb / 1e6, tf, tf * 1e3 / n / rounds_f, fb / tf / 1e6,
            n * L.kkt.stride * 8 / 1e6, n * L.ric.stride * 8 / 1e6, int((ctx.status() != 0).sum())))
        ctx.close()
        del kkt, ric, d, dx0
        torch.cuda.empty_cache()
