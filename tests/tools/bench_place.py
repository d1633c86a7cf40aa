"""ON THE GPU BOX: the placement kernel (sim5gpu_image_place_shares, k_assemble.hip) on the shares of one 4096^2 image for
2, 4 and 8 ranks: time per launch (HIP events, working clock), algorithmic bytes (every peer row read once and written once)
and the HBM rate they amount to.  Pure data movement: HBM-bound."""
import sys, math, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sim5_amd.capi as capi
from sim5_amd import sharding
n = 4096
out = []
for world in (2, 4, 8):
    rows_max = sharding.max_local_rows(n, world)
    descs = [capi.image_desc(n, n, 0.9, 1.0, **sharding.job_rows(n, r, world)) for r in range(1, world)]
    shares = capi.DeviceBuffer(world * 2 * rows_max * n * 4); shares.zero()
    img = capi.DeviceBuffer(2 * n * n * 4)
    fn = lambda: capi.image_place_shares(descs, shares.ptr + 2 * rows_max * n * 4, rows_max, img.ptr, img.ptr + n * n * 4)
    for _ in range(300): fn()
    capi.synchronize()
    e0 = capi.Event(); e1 = capi.Event(); e0.record()
    for _ in range(300): fn()
    e1.record(); ms = e0.elapsed_ms(e1) / 300
    rows = sum(capi.image_rows(d) for d in descs)
    byts = 2 * rows * n * 4 * 2                                   # two planes, read + write
    out.append("%d ranks: %d rows, %.1f MB moved, %.4f ms, %.0f GB/s (%.2f of 8 TB/s)" % (world, rows, byts / 1e6, ms, byts / ms / 1e6, byts / ms / 1e6 / 8000))
print(" | ".join(out))
