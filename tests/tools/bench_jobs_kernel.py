"""ON THE GPU BOX: the job-list launch (sim5gpu_disk_image_jobs -> disk_image_jobs_kernel) against one launch per image
(sim5gpu_disk_image -> disk_image_mirror_kernel<false>), at the working clock, with md5 sums of the planes: C2 alone and in
lists of 2..16, a 512-row share of the 4096^2 image (the share of one of 8 ranks), the headline image, the C5 scan."""
import sys, math, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sim5_amd.capi as capi
from sim5_amd import sharding

W = 1300.0 / 78.6e12


def timed(fn, reps, warm):
    for _ in range(warm): fn()
    capi.synchronize()
    e0, e1 = capi.Event(), capi.Event(); e0.record()
    for _ in range(reps): fn()
    e1.record()
    return e0.elapsed_ms(e1) / reps


def md5(bufs, shape):
    h = hashlib.md5()
    for b in bufs: h.update(b.to_numpy(np.float32, shape).tobytes())
    return h.hexdigest()[:8]


def case(name, descs, rows, nx, reps, warm):
    n = len(descs)
    F = [capi.DeviceBuffer(r * nx * 4) for r in rows]; G = [capi.DeviceBuffer(r * nx * 4) for r in rows]
    rays = sum(r * nx for r in rows)
    def single():
        for d, f, g in zip(descs, F, G): capi.disk_image_device(d, f.ptr, g.ptr)
    def jobs():
        capi.disk_image_jobs(descs, [f.ptr for f in F], [g.ptr for g in G])
    for f in F + G: f.zero()
    ms_s = timed(single, reps, warm); m_s = [md5([f, g], (r, nx)) for f, g, r in zip(F, G, rows)]
    for f in F + G: f.zero()
    ms_j = timed(jobs, reps, warm); m_j = [md5([f, g], (r, nx)) for f, g, r in zip(F, G, rows)]
    ms_s2 = timed(single, reps, 5); ms_j2 = timed(jobs, reps, 5)
    print("%-34s rays %9d  per-image launches %.4f / %.4f ms (frac %.3f)   job list %.4f / %.4f ms (frac %.3f)   same bits: %s" % (
        name, rays, ms_s, ms_s2, rays * W / (min(ms_s, ms_s2) * 1e-3), ms_j, ms_j2, rays * W / (min(ms_j, ms_j2) * 1e-3), m_s == m_j), flush=True)


inc = math.radians(70.0)
c2 = capi.image_desc(1024, 1024, 0.998, inc)
for k in (1, 2, 4, 8, 16):
    case("C2 x %d" % k, [c2] * k, [1024] * k, 1024, max(40, 1600 // k), max(100, 3000 // k))
case("2048^2", [capi.image_desc(2048, 2048, 0.998, inc)], [2048], 2048, 400, 800)
case("4096^2 headline", [capi.image_desc(4096, 4096, 0.998, inc)], [4096], 4096, 150, 250)
kw = sharding.job_rows(4096, 1, 8)
share = capi.image_desc(4096, 4096, 0.998, inc, **kw)
case("512-row share of 4096^2 (rank 1/8)", [share], [capi.image_rows(share)], 4096, 600, 1200)
kw = sharding.job_rows(4096, 1, 2)
share = capi.image_desc(4096, 4096, 0.998, inc, **kw)
case("2048-row share of 4096^2 (rank 1/2)", [share], [capi.image_rows(share)], 4096, 300, 500)
c5 = [capi.image_desc(8192, 8192, 0.998, math.radians(i)) for i in range(10, 90, 10)]
case("C5 scan 8 x 8192^2", c5, [8192] * 8, 8192, 4, 6)
