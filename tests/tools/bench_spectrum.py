"""Spectrum job on the GPU box: time per job and the roofline fraction of bench.py's f3 line, for a few
energy counts (python tests/tools/bench_spectrum.py [image side, default 1024] [log | uniform | all])."""
import ctypes as C, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sim5_amd.capi as capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm, reps = (300, 50) if n <= 1024 else (40, 10)
d = capi.image_desc(n, n, 0.998, math.radians(70.0))
# grids: "log" (default: 1, 128, 256 energies on a logarithmic grid -- what tests/tools/prof_spectrum.sh profiles: its summary takes the
# middle third of the launches for the 128-energy job), "uniform" (64, 128, 256 energies in equal steps: the recurrence along the
# energies; same three-job shape), "all" (both, and 512 uniform)
mode = sys.argv[2] if len(sys.argv) > 2 else "log"
jobs = {"log": ((1, False), (128, False), (256, False)), "uniform": ((64, True), (128, True), (256, True)),
        "all": ((1, False), (128, False), (256, False), (64, True), (128, True), (256, True), (512, True))}[mode]
for ne, uniform in jobs:
    E = np.linspace(0.1, 30.0, ne) if uniform else 10.0 ** np.linspace(-1, 1.5, ne)
    dE = capi.DeviceBuffer(E.nbytes); dE.from_numpy(E); dS = capi.DeviceBuffer(E.nbytes)
    capi._lib.sim5gpu_disk_spectrum_workspace.restype = capi.SZ
    ws = capi.DeviceBuffer(max(capi._lib.sim5gpu_disk_spectrum_workspace(C.byref(d), capi.I(ne)), 8))
    run = lambda: capi._check(capi._lib.sim5gpu_disk_spectrum(C.byref(d), capi.I(ne), capi.VP(dE.ptr), capi.D(1.7), capi.I(1), capi.VP(dS.ptr), capi.VP(ws.ptr), capi.VP(0)), "spectrum")
    for _ in range(warm): run()
    capi.synchronize(); e0 = capi.Event(); e1 = capi.Event(); e0.record()
    for _ in range(reps): run()
    e1.record(); ms = e0.elapsed_ms(e1) / reps
    w = n * n * 1.6e3 + n * n * ne * 6.0
    S = dS.to_numpy(np.float64, (ne,))
    print("%3d energies (%s grid): %.4f ms  roofline %.3f  sum %.12e" % (ne, "uniform" if uniform else "log", ms, w / (ms * 1e-3) / 78.6e12, S.sum()))
