#!/usr/bin/env python3
"""bench.py -- null geodesics per second on the headline workload of BASELINE.json:
a 4096 x 4096 thin-disk image of a Kerr black hole (a = 0.998, i = 70 deg), elliptic-integral
path, one GPU lane per ray (hand-written HIP, sim5_amd/csrc), through the C-ABI of
include/sim5gpu.h.

  python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one complete image per GPU.  Rays are independent, so the path shards without any exchange
(SURVEY.md 8(e)): with N > 1 (launched by torch.distributed.run, one rank per GPU) every rank traces its
own complete 4096 x 4096 image -- the shape of a multi-image job such as the inclination scan of
BASELINE.json configs[4], one image per GPU -- with NO collective on the data path; the images stay in the
HBM of the GPU that made them.  Per-GPU work is fixed, so "scaling" is "weak" and
`value` = N * 4096*4096*K / max-over-ranks time (barrier + synchronize on both sides of the timed region).

`--mode stripes` times the other way to use N GPUs, ONE image sharded by 64-row stripes over the ranks
(sim5_amd/sharding.py; one kernel launch per rank and image) and assembled on rank 0 by ONE RCCL gather
per image inside the timed region (double-buffered, so the gather of image i overlaps the tracing of
image i+1; all gathers complete before the clock stops): total work fixed, "scaling": "strong".

Rank 0 prints one JSON line.  At N = 1 it also carries
  roofline:     FP64-VALU roofline of the image kernel.  achieved = W_ell (1.3e3 algorithmic FP64
                operations per ray) x rays per launch / mean kernel time measured
                with HIP events on the launch stream; peak = 78.6 TFLOP/s FP64 vector (256 CU x 128
                FLOP/clk x 2.4 GHz).  The path is scalar ODE/special-function work: no MFMA, and HBM
                traffic is 8 B/ray of output (reported next to it as hbm_*).
  cpu_baseline: the unmodified reference (oracle/_ref/libsim5ref.so, built from the reference
                sources in the build container; falls back to our C port when absent) timed on this
                box's host cores over a bounded row sample of the same image.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX = NY = 4096
SPIN, INCL_DEG = 0.998, 70.0
W_ELL = 1.3e3                     # algorithmic FP64 ops per elliptic thin-disk ray (an estimate)
# the same quantity counted exactly on the reference binary (oracle/opcount.c, profiles/r01_opcount_image.json):
# 363 add + 207 sub + 419 mul + 165 div + 152 sqrt + 308 compare + 11 x87 + 139 library calls per ray
W_ELL_MEASURED = 1764.8
PEAK_FP64_VALU_TFLOPS = 78.6      # 256 CU x 128 FLOP/clk x 2.4 GHz
PEAK_HBM_GBPS = 8000.0


def cpu_baseline(budget_s=12.0):
    """Reference (or port) on the host cores over a bounded sample of the headline image."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as ol
    kind = "reference" if ol.have_reference() else "port"
    cores = os.cpu_count() or 1
    # calibrate on 1/128 of the rows, then size the sample for ~budget_s of wall time
    cal = ol.cpu_disk_image(kind, NX, NY, SPIN, INCL_DEG, y0=16, ystride=128, nthreads=cores, full=False)
    rate = cal["rays"] / max(cal["seconds"], 1e-6)
    rows = int(min(NY, max(32, budget_s * rate / NX)))
    stride = max(1, NY // rows)
    rays = 0
    secs = 0.0
    reps = 0
    while secs < 0.6 * budget_s and reps < 64:       # on many-core hosts one pass is short: repeat it
        run = ol.cpu_disk_image(kind, NX, NY, SPIN, INCL_DEG, y0=stride // 2, ystride=stride, nthreads=cores, full=False)
        rays += run["rays"]; secs += run["seconds"]; reps += 1
    one = ol.cpu_disk_image(kind, NX, NY, SPIN, INCL_DEG, y0=16, ystride=128, nthreads=1, full=False)
    return {
        "value": rays / secs, "unit": "null geodesics/s", "cores": cores, "kind": kind,
        "sample": "every %d-th row of the %dx%d image, %d pass(es), %d rays, %d threads, %.1f s wall" % (
            stride, NX, NY, reps, rays, cores, secs),
        "single_thread_value": one["rays"] / one["seconds"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["images", "stripes"], default="images",
                    help="N > 1: one complete image per GPU, no collective (default) | one image in row stripes + gather")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # test hook: SIM5_BENCH_ONE_GPU=1 runs all ranks on GPU 0 over gloo, to exercise the N > 1 control flow on a
    # one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    one_gpu_test = os.environ.get("SIM5_BENCH_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from sim5_amd.build import build
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        build()                             # no-op when the in-tree library is up to date
    if world > 1:
        dist.barrier()                      # nobody loads the library while rank 0 may be writing it
    import sim5_amd.capi as capi            # raises if libsim5gpu.so is missing
    from sim5_amd import sharding
    capi.set_device(local_rank)

    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    inc = INCL_DEG / 180.0 * math.pi
    stripes = world > 1 and args.mode == "stripes"
    if stripes and one_gpu_test:
        sys.exit("bench.py: the one-GPU test hook covers --mode images only (gloo cannot gather device tensors asynchronously)")
    if not stripes:
        # one complete image per rank: [2 planes (F g^4 | g), NY, NX] f32, resident in this GPU's HBM
        desc = capi.image_desc(NX, NY, SPIN, inc)
        image = torch.zeros((2, NY, NX), dtype=torch.float32, device=dev)

        def trace(buf):
            capi.disk_image_device(desc, buf[0].data_ptr(), buf[1].data_ptr(), stream=stream)

        def step(i):
            trace(image)

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        def last_image():
            return image
    else:
        # One launch per rank and image: the rank's 64-row stripes (rank, rank+world, ...) in a single grid.
        # Tile buffers are [2 planes, rows, NX] f32 = one contiguous gather payload; two of them so that the
        # gather of image i (RCCL, its own stream) overlaps the tracing of image i+1.
        desc = capi.image_desc(NX, NY, SPIN, inc, y0=rank * sharding.STRIPE, y1=NY,
                               stripe_rows=sharding.STRIPE, stripe_step=world * sharding.STRIPE)
        assert capi.image_rows(desc) == sharding.local_rows(NY, rank, world)
        pipe = sharding.TilePipeline(torch, dist, rank, world, NY, NX, dev)

        def trace(buf):
            capi.disk_image_device(desc, buf[0].data_ptr(), buf[1].data_ptr(), stream=stream)

        def step(i):
            pipe.step(lambda buf: trace(buf))

        def fence():
            pipe.drain()
            dist.barrier()
            torch.cuda.synchronize()

        def last_image():
            return pipe.last_image()

    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events around every kernel launch on rank 0 (on the stream the kernel is launched on)
    ev = [(capi.Event(), capi.Event()) for _ in range(args.steps)] if rank == 0 else None
    if ev:
        def trace(buf, _i=[0]):                  # noqa: B006,F811 -- the timed flavour of trace()
            a, b = ev[_i[0] % len(ev)]
            a.record(stream)
            capi.disk_image_device(desc, buf[0].data_ptr(), buf[1].data_ptr(), stream=stream)
            b.record(stream)
            _i[0] += 1
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu_test else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    rays = NX * NY * (1 if stripes or world == 1 else world)     # rays of one step of the whole job
    value = rays * args.steps / dt
    # sanity: the image that came out is the Kerr disk (known hit count of the reference)
    img = last_image()
    hits = int((img[1] > 0).sum().item())
    out = {
        "metric": "null geodesics/sec, 4096x4096 Kerr disk image (a=0.998, i=70)",
        "value": value, "unit": "null geodesics/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if stripes else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "4096x4096 thin-disk image, a=0.998, i=70deg, elliptic-integral path, "
                               "g-factor + Novikov-Thorne flux (BASELINE.json headline / configs[1] at 4096^2)",
                   "rays_per_step": rays,
                   "parallelism": ("1 GPU" if world == 1 else
                                   "row stripes x%d + 1 RCCL gather per image" % world if stripes else
                                   "%d independent images, one per GPU, no collective" % world),
                   "disk_hits": hits, "disk_hits_reference": 15865362},
    }
    if ev:
        rays_launch = capi.image_rows(desc) * NX             # rays of one launch of rank 0 (its stripes)
        kms = [a.elapsed_ms(b) for (a, b) in ev]
        kavg = sum(kms) / len(kms)
        achieved = rays_launch * W_ELL / (kavg * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "fp64_valu", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
            "kernel": "disk_image_grid_kernel", "kernel_ms_avg": kavg, "algorithmic_flops_per_ray": W_ELL,
            "rays_per_launch": rays_launch, "per": "GPU (rank 0)",
            "algorithmic_flops_per_ray_counted_on_reference": W_ELL_MEASURED,
            "achieved_with_counted_flops": achieved * W_ELL_MEASURED / W_ELL,
            "frac_with_counted_flops": achieved * W_ELL_MEASURED / W_ELL / PEAK_FP64_VALU_TFLOPS,
            "hbm_algorithmic_bytes_per_launch": rays_launch * 8,
            "hbm_achieved_GBps": rays_launch * 8 / (kavg * 1e-3) / 1e9, "hbm_peak_GBps": PEAK_HBM_GBPS,
            "note": "scalar FP64 special-function work per ray: no MFMA; HBM carries only 8 B/ray of output",
        }
        if world > 1:
            out["roofline"]["traffic"] = None          # the PMC traffic figure was collected for the 1-GPU launch
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:                       # the baseline is a report, never a blocker
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
