"""Where the cycles of one raytrace() call go.  ON THE GPU BOX with the instrumented variant of the library:
    tests/tools/ab_build.sh dbg S5_TORUS_FAST_EXTRA="-DS5_TORUS_DEBUG"          (build container)
    SIM5GPU_LIB=sim5_amd/lib/ab_dbg.so python tests/tools/torus_phases.py [out.json]
The march kernel of that build passes a cycle counter to the phase marks of s5_raytrace.hpp (s_memtime between scheduling
barriers; sums per phase in LDS, added to a debug buffer at the end).  Two jobs: (a) TWO rays alone on the GPU (one wave, the
other 3 071 waves asleep: the latency of the dependent chain) and (b) the C4 job (1024^2 rays: the same phases with three waves
per SIMD taking turns)."""
import sys, json, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T

PH = ["stepsize", "predict (incl. sincos, sqrt)", "metric+connection", "corrector 1", "corrector 2", "corrector 3", "error check (k.k, k_t)",
      "acceleration at the new point", "rk4 head", "rk4 stage 1", "rk4 stage 2", "rk4 stage 3", "rk4 tail (update, metric+connection, accel)",
      "state load (LDS, kernarg)", "state store + transfer + end test", "queues, loop control, waits",
      "(two marks with nothing in between)"]
AT = 12000


def job(n, rows, label):
    d = T.torus_desc(capi, n, 0.9, 70.0, r0=100.0, precision=1.0, max_steps=100000)
    if rows is not None:
        d.img.y0, d.img.y1 = rows
    N = d.img.nx * (d.img.y1 - d.img.y0)
    sb = capi.DeviceBuffer(max(N, 1) * 40); steps = capi.DeviceBuffer(max(N, 1) * 4); dbg = capi.DeviceBuffer(16 * 8192 * 8)
    out = None
    for rep in range(2):                       # (first pass: code and tables loaded)
        dbg.from_numpy(np.zeros(16 * 8192, dtype=np.float64))
        capi.synchronize(); t0 = time.perf_counter()
        capi.torus_image_device(d, sb.ptr, aux={"steps": steps.ptr, "k_end": dbg.ptr}); capi.synchronize()
        wall = time.perf_counter() - t0
    w = dbg.to_numpy(np.uint64, (16 * 8192,))
    acc = w[AT:AT + 17].astype(np.float64); cnt = w[AT + 17:AT + 34].astype(np.float64)
    cyc, ticks = float(w[AT + 34]), float(w[AT + 35])
    s = steps.to_numpy(np.int32, (max(N, 1),))[:N]
    calls = float(s.sum())
    mhz = 100.0 * cyc / ticks if ticks else float("nan")
    rec = {"job": label, "rays": int(N), "raytrace_calls": int(calls), "calls_of_the_longest_ray": int(s.max()), "host_wall_ms": 1e3 * wall,
           "s_memtime_MHz (cycles per 100 MHz tick, summed over workgroups)": mhz,
           "phases": [{"phase": PH[i], "marks": int(cnt[i]), "cycles_per_mark": acc[i] / cnt[i] if cnt[i] else 0.0,
                       "cycles_per_call": acc[i] / calls if calls else 0.0} for i in range(17)],
           "cycles_per_call_sum": float(acc[:16].sum() / calls) if calls else 0.0,
           "V_batches": int(w[0]), "V_lanes_avg": float(w[1]) / max(float(w[0]), 1), "R_batches": int(w[2]), "R_lanes_avg": float(w[3]) / max(float(w[2]), 1)}
    return rec

out = {"what": __doc__, "jobs": [job(2, (0, 1), "two rays alone (alpha = -+ rmax / 2, beta = -rmax / 2; one row of a 2 x 2 image)"),
                                 job(1024, None, "C4: 1024^2 rays")]}
for j in out["jobs"]:
    print("== %s: %d calls, longest ray %d, %.3f ms wall, clock %.0f MHz, %.0f cycles per call (sum of phases)" % (
        j["job"], j["raytrace_calls"], j["calls_of_the_longest_ray"], j["host_wall_ms"], j["s_memtime_MHz (cycles per 100 MHz tick, summed over workgroups)"], j["cycles_per_call_sum"]))
    for p in j["phases"]:
        print("   %-48s marks %10d  cycles/mark %9.1f  cycles/call %9.1f" % (p["phase"], p["marks"], p["cycles_per_mark"], p["cycles_per_call"]))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
