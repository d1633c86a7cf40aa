#!/usr/bin/env python3
"""bench.py -- null geodesics per second on the headline workload of BASELINE.json:
a 4096 x 4096 thin-disk image of a Kerr black hole (a = 0.998, i = 70 deg), elliptic-integral
path, one GPU lane per ray (hand-written HIP, sim5_amd/csrc), through the C-ABI of
include/sim5gpu.h.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|c5] [--mode stripes|images]

Before the W warm-up steps the program runs ~0.3 s of the same steps untimed ("spin_up_steps" in the line; --no-spin-up
skips them): an idle MI355X needs some 50 ms of work to reach its working clock, and W = 3 images are 1.3 ms.
N = 1: a step is one complete image.
N > 1: one rank per GPU.  `python bench.py --gpus N` by itself starts its ranks (a child `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`; the parent never touches a GPU); under a launcher (WORLD_SIZE set)
it is a rank.  Default `--mode stripes`, the split BASELINE.json's
north_star names: ONE image per step, its rows dealt to the ranks in 64-row stripes round-robin -- in mirrored pairs,
so that every rank runs the pairing kernel (sim5_amd/sharding.py; one kernel launch per rank and image) -- and
assembled on rank 0 INSIDE the timed region: rank 0 traces its own rows in place in the image (SIM5GPU_IMG_INPLACE), ONE
RCCL gather per image brings the peers' stripes (both planes of a rank are one contiguous payload; double-buffered, so
the gather of image i overlaps the tracing of image i+1), and ONE placement kernel (sim5gpu_image_place_shares) copies
them to their image rows: every step ends with a row-major [2, ny, nx] image on rank 0, and every gather and placement
has completed before the clock stops.  A GPU writes image rows several times faster than one xGMI link carries them
and rank 0's own rows need no link, so the split is weighted (`--root-band auto`, sharding.plan_dealt_rows): only the
outer rows are dealt and gathered, a centred band stays with rank 0, which traces it straight into the assembled image
while the gather is in flight; the band's size balances a kernel time and a gather time measured (HIP events, medians
of 12) before the timed region (recorded in per_rank.root_band_plan with the predicted step time of every candidate;
`--root-band off` deals everything).  Total work is fixed: "scaling": "strong", `value` = rays of one image * K /
max-over-ranks time; `value_kernel_only` = the same rays x K over a wall-clocked region of its own in which every rank
traces its share of K images back to back between barriers, no gather ("kernel time incl. image write;
gather ... separately"; `kernel_only_region`), `per_rank.gather_ms_alone` / `place_ms_alone` the exchange.
`--mode images` (opt-in) is the other way to use N GPUs: N independent images, one per GPU, no collective
("scaling": "weak").
`--workload c5` is BASELINE.json configs[4]: a step is the inclination scan 10..80 deg of 8192 x 8192 images
(8 images, 5.4e8 rays), each image striped over the ranks and gathered like the headline image.

Rank 0 prints one JSON line.  Beside the contract's fields it carries
  roofline:     FP64-VALU roofline of the image kernel on rank 0.  achieved = W_ell (1.3e3 algorithmic FP64
                operations per ray) x rays per launch / mean kernel time measured with HIP events
                on the launch stream; peak = 78.6 TFLOP/s FP64 vector (256 CU x 128 FLOP/clk x 2.4 GHz).  The
                path is scalar ODE/special-function work: no MFMA, and HBM traffic is 8 B/ray of output (hbm_*).
                executed_frac: the FP64 operations the kernel actually executes (PMC counts of the same command,
                profiles/traffic.json) over the same time -- the hardware-utilisation figure next to the algorithmic one.
  cold_clock:   N = 1: the same image timed on an idle GPU (3 launches after 0.3 s of idle, no spin-up).
  per_rank:     N > 1: mean kernel ms of every rank (HIP events, timed region) and the time of one gather measured
                on its own after the timed region (rank 0, the receiver), so compute and exchange can be told apart.
  cpu_baseline: N = 1: the unmodified reference (oracle/_ref/libsim5ref.so, built from the reference sources in the
                build container; our C port when absent) timed on this box's host cores over a bounded row sample
                of the same image; `cores` = the cores this process may run on (sched_getaffinity).
  extra:        the other BASELINE.json configurations, timed after the headline's timed region (a few launches
                each): C2, C3 (polarized), C4 (torus, raytrace() steps), C5 (N = 1: one 8192^2 image per inclination;
                N > 1: the whole striped + gathered scan), each with kernel ms, rays/s and its roofline fraction.
`ok` is false (and the exit code 1) if the image that came out does not have the reference's disk-hit count.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPIN, INCL_DEG = 0.998, 70.0
W_ELL = 1.3e3                     # algorithmic FP64 ops per elliptic thin-disk ray (an estimate)
IMAGE_KERNEL = "disk_image_jobs_kernel"     # whole images, centred bands and mirrored stripe shares (k_disk_image.hip: the pairing kernel reading its
                                            # job from the argument segment, a list of one); other row sets: disk_image_grid_kernel
# the same quantity counted exactly on the reference binary (oracle/opcount.c, profiles/r01_opcount_image.json):
# 363 add + 207 sub + 419 mul + 165 div + 152 sqrt + 308 compare + 11 x87 + 139 library calls per ray
W_ELL_MEASURED = 1764.8
W_POL = 3.0e2                     # polarization add-on per ray
W_STEP = 7.5e2                    # algorithmic FP64 ops per raytrace() step, Verlet part
W_STEP_MEASURED = 1354.0          # counted on the reference incl. its RK4 fallbacks (profiles/r01_opcount_verlet.json)
PEAK_FP64_VALU_TFLOPS = 78.6      # 256 CU x 128 FLOP/clk x 2.4 GHz
PEAK_HBM_GBPS = 8000.0
HEADLINE_HITS = 15865362          # disk hits of the reference on the headline image
C5_INCLINATIONS = (10, 20, 30, 40, 50, 60, 70, 80)


def reference_hits_c5():
    """Disk-hit counts of the reference on the eight 8192^2 images (tests/golden/c5_hit_counts.json, data)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "c5_hit_counts.json")) as fh:
            return {int(k): v["hits_image_g_gt_0"] for k, v in json.load(fh)["counts"].items()}
    except Exception:
        return {}


def usable_cores():
    """(cores this process may actually use, how that was found): the scheduler affinity mask, capped by the
    cgroup CPU quota of the container when there is one (cpu.max = "quota period" under cgroup v2)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except Exception:
            continue
    cores = aff if quota is None else max(1, min(aff, int(math.ceil(quota))))
    return cores, {"sched_getaffinity": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count()}


def cpu_baseline(nx, ny, budget_s=12.0):
    """Reference (or port) on the host cores over a bounded sample of the headline image."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib as ol
    kind = "reference" if ol.have_reference() else "port"
    cores, how = usable_cores()
    # calibrate on 1/128 of the rows, then size the sample for ~budget_s of wall time
    cal = ol.cpu_disk_image(kind, nx, ny, SPIN, INCL_DEG, y0=16, ystride=128, nthreads=cores, full=False)
    rate = cal["rays"] / max(cal["seconds"], 1e-6)
    rows = int(min(ny, max(32, budget_s * rate / nx)))
    stride = max(1, ny // rows)
    rays = 0
    secs = 0.0
    reps = 0
    while secs < 0.6 * budget_s and reps < 64:       # on many-core hosts one pass is short: repeat it
        run = ol.cpu_disk_image(kind, nx, ny, SPIN, INCL_DEG, y0=stride // 2, ystride=stride, nthreads=cores, full=False)
        rays += run["rays"]; secs += run["seconds"]; reps += 1
    one = ol.cpu_disk_image(kind, nx, ny, SPIN, INCL_DEG, y0=16, ystride=128, nthreads=1, full=False)
    single = one["rays"] / one["seconds"]
    return {
        "value": rays / secs, "unit": "null geodesics/s", "cores": cores, "kind": kind,
        "sample": "every %d-th row of the %dx%d image, %d pass(es), %d rays, %d threads, %.1f s wall" % (
            stride, nx, ny, reps, rays, cores, secs),
        "single_thread_value": single, "parallel_speedup": (rays / secs) / single,
        "cores_found_by": how,
    }


class ImageJob:
    """One thin-disk image per step: this rank's launches (descriptor + launch each).  N = 1 or --mode images: the whole
    image.  Striped: the rank's mirrored stripe pairs (sharding.job_rows) -- on rank 0 written IN PLACE into the image it
    assembles (SIM5GPU_IMG_INPLACE), on the peers packed into the payload of the gather -- and on rank 0, when the plan
    keeps a band of the middle rows with it (sharding.root_band), that band as a second launch into the same image."""

    def __init__(self, capi, sharding, n, incl_deg, rank, world, striped, stream, dealt=None):
        self.capi, self.n, self.stream = capi, n, stream
        inc = incl_deg / 180.0 * math.pi
        self.desc = self.band_desc = None
        if striped:
            kw = sharding.job_rows(n, rank, world, dealt=dealt)
            if kw["y0"] < kw["y1"]:
                self.desc = capi.image_desc(n, n, SPIN, inc, inplace=(rank == 0), **kw)
                assert capi.image_rows(self.desc) == sharding.local_rows(n, rank, world, dealt=dealt)
            band = sharding.root_band(n, dealt) if rank == 0 else None
            if band:
                self.band_desc = capi.image_desc(n, n, SPIN, inc, y0=band[0], y1=band[1])
            self.rays = sharding.rank_rows(n, rank, world, dealt=dealt) * n      # rays this rank traces per image
        else:
            self.desc = capi.image_desc(n, n, SPIN, inc)
            self.rays = capi.image_rows(self.desc) * n
        self.inplace = bool(striped and rank == 0)
        self.bracket, self.expected = False, 0
        self.events = None
        self.used = 0
        self.band_events = None
        self.band_used = 0

    def _launch(self, desc, buf):
        self.capi.disk_image_device(desc, buf[0].data_ptr(), buf[1].data_ptr(), stream=self.stream)

    def trace(self, buf, inplace=False):
        if self.desc is None:
            return
        assert inplace == self.inplace or not self.inplace      # a whole image (N = 1) is in place either way
        if self.events is not None and self.bracket:
            # N = 1: ONE pair of events around all launches of the timed region (an event pair per launch costs ~8 us of
            # stream time per step -- 2 % of a 0.37 ms image -- measured against back-to-back launches)
            if self.used == 0:
                self.events[0][0].record(self.stream)
            self._launch(self.desc, buf)
            self.used += 1
            if self.used == self.expected:
                self.events[0][1].record(self.stream)
        elif self.events is not None and self.used < len(self.events):
            a, b = self.events[self.used]
            self.used += 1
            a.record(self.stream)
            self._launch(self.desc, buf)
            b.record(self.stream)
        else:
            self._launch(self.desc, buf)

    def trace_both(self, buf, view):
        """rank 0 with a band: its in-place share and the band as ONE job-list launch (two jobs streaming back to back)"""
        a = b = None
        if self.events is not None and self.used < len(self.events):
            a, b = self.events[self.used]
            self.used += 1
            a.record(self.stream)
        self.capi.disk_image_jobs([self.desc, self.band_desc], [buf[0].data_ptr(), view[0].data_ptr()],
                                  [buf[1].data_ptr(), view[1].data_ptr()], stream=self.stream)
        if b is not None:
            b.record(self.stream)

    def trace_band(self, view):
        if self.band_events is not None and self.band_used < len(self.band_events):
            a, b = self.band_events[self.band_used]
            self.band_used += 1
            a.record(self.stream)
            self._launch(self.band_desc, view)
            b.record(self.stream)
        else:
            self._launch(self.band_desc, view)

    def start_timing(self, launches, bracket=False):
        """HIP events (created here, outside the timed region) around the next `launches` launches: one pair per launch
        (N > 1: kernel and exchange are told apart per rank), or with bracket=True one pair around all of them"""
        self.bracket, self.expected = bool(bracket), launches
        self.events = [(self.capi.Event(), self.capi.Event()) for _ in range(1 if bracket else launches)]
        self.used = 0
        if self.band_desc is not None:
            self.band_events = [(self.capi.Event(), self.capi.Event()) for _ in range(launches)]
            self.band_used = 0

    def collect(self):
        """mean kernel ms per image over the launches recorded since start_timing() (waits for them)"""
        if self.events is not None and self.bracket:
            ms = self.events[0][0].elapsed_ms(self.events[0][1]) / max(self.used, 1) if self.used == self.expected else float("nan")
            self.events = None
            return ms
        kms = [a.elapsed_ms(b) for (a, b) in (self.events or [])[:self.used]]
        bms = [a.elapsed_ms(b) for (a, b) in (self.band_events or [])[:self.band_used]]
        self.events = self.band_events = None
        if not kms and not bms:
            return 0.0 if self.desc is None and self.band_desc is None else float("nan")
        return (sum(kms) / len(kms) if kms else 0.0) + (sum(bms) / len(bms) if bms else 0.0)


def make_placer(capi, sharding, n, world, dealt, stream):
    """rank 0: the peers' gathered rows to their image rows with ONE kernel (sim5gpu_image_place_shares); the job
    descriptions only carry the row geometry here, which is the same for every image of this size and split"""
    descs = []
    for r in range(1, world):
        kw = sharding.job_rows(n, r, world, dealt=dealt)
        if kw["y0"] < kw["y1"]:
            descs.append(capi.image_desc(n, n, SPIN, 1.0, **kw))
        else:
            descs = None                                   # a rank without rows (tiny images): fall back to slicing
            break

    def place(gathered, image):
        if not descs:
            return sharding.place_shares(gathered, n, world, image, dealt=dealt)
        capi.image_place_shares(descs, gathered[1].data_ptr(), gathered.shape[2], image[0].data_ptr(), image[1].data_ptr(), stream=stream)
    return place


def plane_checksums(torch, img):
    """[sum of the 32-bit patterns of plane 0, of plane 1] as Python ints: equal images have equal checksums, and a single
    differing pixel changes one (tests: the image assembled from the ranks' shares against one launch)"""
    v = img.view(torch.int32).to(torch.int64)
    return [int(v[0].sum().item()), int(v[1].sum().item())]


class RankFailure(Exception):
    pass


def agree(dist, torch, cdev, rank, world, ok_local, what, err=None):
    """Collective-safe abort (N > 1): every rank joins one all-reduce of an error flag at the end of a phase that can fail
    on one rank alone; if any rank failed, EVERY rank leaves with a non-zero exit code instead of walking into the next
    collective and waiting for the others until the process-group timeout.  `ok_local` False on this rank: `err` is printed."""
    if not ok_local:
        sys.stderr.write("bench.py: rank %d failed in phase '%s': %s\n" % (rank, what, err))
        sys.stderr.flush()
    if world <= 1:
        if not ok_local:
            raise RankFailure(what)
        return
    t = torch.tensor([0.0 if ok_local else 1.0], dtype=torch.float32, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if float(t.item()) > 0.0:
        if rank == 0:
            print(json.dumps({"ok": False, "error": "a rank failed in phase '%s' (see stderr); all %d ranks stopped before the next collective" % (what, world),
                              "n_gpus": world}))
            sys.stdout.flush()
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        sys.exit(3)


def guarded(dist, torch, cdev, rank, world, what, fn):
    """fn() on this rank, then the agreement above: an exception on one rank stops all ranks (fn must not contain a
    collective that the failing rank would skip -- a phase that does is bounded by the process-group timeout instead)"""
    try:
        out = fn()
        err = None
    except Exception as e:                                 # noqa: BLE001 -- reported, and every rank leaves
        import traceback
        out, err = None, "%r\n%s" % (e, traceback.format_exc())
    agree(dist, torch, cdev, rank, world, err is None, what, err)
    return out


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def time_gather(torch, dist, capi, pipe, stream, reps, host_clock):
    """ms of ONE gather on its own (not overlapped), median of `reps`: HIP events on the stream the collective is
    synchronised with (work.wait() makes torch's current stream wait for RCCL's), or the host clock for the host-staged
    gather of the one-GPU test hook"""
    out = []
    for _ in range(reps):
        pipe.drain(); torch.cuda.synchronize(); dist.barrier()
        if host_clock:
            g0 = time.perf_counter()
            pipe.gather(0, async_op=False)
            torch.cuda.synchronize()
            out.append(1e3 * (time.perf_counter() - g0))
        else:
            e0, e1 = capi.Event(), capi.Event()
            e0.record(stream)
            pipe.gather(0, async_op=False)
            e1.record(stream)
            out.append(e0.elapsed_ms(e1))
        pipe.unplaced[0] = False                           # a measurement, not an image: nothing to place
    return median(out), out


def plan_root_band(torch, dist, capi, sharding, rank, world, n, dev, cdev, stream, one_gpu_test, setting):
    """Rows of the upper half to deal over the ranks (sharding.plan_dealt_rows); the band left in the middle stays with
    rank 0.  `setting`: "auto" measures the full-image kernel on rank 0 (HIP events, median of 12 batches at the working
    clock) and the gather of an equal split (HIP events around the collective, median of 12) before the timed region and
    balances the two; "off" deals the whole upper half; a number fixes the dealt rows (tests).  Every rank gets rank 0's
    answer.  Returns (dealt, record); the record carries the samples and the predicted step time of every candidate."""
    half = sharding.upper_half(n)
    rec = {"setting": setting}
    if setting == "off":
        dealt = half
    elif setting != "auto":
        dealt = max(1, min(half, int(setting)))
    else:
        job = ImageJob(capi, sharding, n, INCL_DEG, rank, 1, False, stream)
        img = torch.zeros((2, n, n), dtype=torch.float32, device=dev)
        timed_kernel(capi, stream, lambda: job.trace(img), 1, 150)             # ~60 ms of launches first: working clock
        ksamples = [timed_kernel(capi, stream, lambda: job.trace(img), 5, 0) for _ in range(12)]
        kms = median(ksamples)
        torch.cuda.synchronize()
        del img
        pipe = sharding.TilePipeline(torch, dist, rank, world, n, n, dev, host_staged=one_gpu_test)
        gms, gsamples = time_gather(torch, dist, capi, pipe, stream, 3 if one_gpu_test else 12, one_gpu_test)
        del pipe
        dealt = sharding.plan_dealt_rows(n, world, kms, gms)
        c_k, c_g = kms / n, gms / (n / world)
        cands = {}
        for cand in sorted({dealt, half} | {d for d in (dealt - 64 * world, dealt + 64 * world) if 64 * world <= d < half}):
            root_rows = sharding.rank_rows(n, 0, world, dealt=cand)
            peer_rows = sharding.rank_rows(n, 1, world, dealt=cand)
            cands[str(cand)] = {"root_trace_ms": root_rows * c_k, "peer_gather_ms": peer_rows * c_g,
                                "predicted_step_ms": max(root_rows * c_k, peer_rows * c_g)}
        rec.update({"kernel_ms_full_image": kms, "gather_ms_equal_split": gms,
                    "kernel_ms_samples": ksamples, "gather_ms_samples": gsamples, "timing": "HIP events, median",
                    "candidates_dealt_rows": cands})
    t = torch.tensor([float(dealt)], dtype=torch.float64, device=cdev)
    dist.broadcast(t, src=0)                 # rank 0's measurement decides for everybody
    dealt = int(t.item())
    rec["dealt_rows_of_upper_half"] = dealt
    rec["root_band_rows"] = list(sharding.root_band(n, dealt) or ())
    return dealt, rec


def all_ok(dist, torch, cdev, world, ok_local):
    """soft agreement: True on every rank iff every rank says ok (one all-reduce); nobody exits"""
    if world <= 1:
        return bool(ok_local)
    t = torch.tensor([0.0 if ok_local else 1.0], dtype=torch.float32, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()) == 0.0


class LineGuard:
    """N > 1, after the timed region: the measurements that follow (the gather on its own, the peer-to-peer A/B, the C5 scan) are
    optional, collective, and -- on the first node with several GPUs -- have never run on hardware.  None of them may take the headline
    line with it.  Rank 0 hands the guard a function that makes the line from what the timed region measured; a helper thread on every
    rank then waits for (a) SIGTERM -- the launcher's reaction to a rank that died, e.g. of a GPU fault while storing into a peer's
    memory: rank 0 prints the line without the optional records and every rank leaves with code 1 -- or (b) a deadline: the same, code
    0 on every rank (their timers start within a collective of each other).  The thread does not need the main thread, which may sit in
    a collective that will never complete (the signal reaches it through signal.set_wakeup_fd).  disarm() when the phases are through."""

    def __init__(self, rank, deadline_s):
        import select, signal, threading
        self.rank, self.make_line, self.done, self.fallback = rank, None, False, None
        self.lock = threading.Lock()
        self.r, self.w = os.pipe()
        os.set_blocking(self.w, False)
        self.old_fd = signal.set_wakeup_fd(self.w, warn_on_full_buffer=False)
        self.old_handler = signal.signal(signal.SIGTERM, lambda *a: None)     # (a Python-level handler: the wake-up byte is written for it)
        self.deadline = time.time() + deadline_s

        def watch():
            while True:
                left = self.deadline - time.time()
                ready = select.select([self.r], [], [], max(0.0, left))[0] if left > 0 else []
                with self.lock:
                    if self.done:
                        return
                    if ready:
                        try:
                            got = os.read(self.r, 64)
                        except OSError:
                            got = b""
                        if signal.SIGTERM not in got:
                            continue                                   # another signal's byte
                        why = "SIGTERM from the launcher (a rank died)"
                    elif time.time() >= self.deadline:
                        why = "no end after %d s" % deadline_s
                    else:
                        continue
                    self._leave(why, 1 if ready else 0)
        self.thread = threading.Thread(target=watch, daemon=True)
        self.thread.start()

    def _leave(self, why, code):
        if self.rank == 0 and self.make_line is not None:
            # the main thread may be writing to the records the line is made from (this is the helper thread): a line that cannot be
            # made now ("dictionary changed size during iteration") is replaced by the one serialised when the guard was armed
            try:
                line = self.make_line(why)
            except Exception as e:                                     # noqa: BLE001
                sys.stderr.write("bench.py: could not make the line in the guard (%r): printing the one made before the optional phases\n" % (e,))
                line = self.fallback
            if line:
                sys.stdout.write(line + "\n")
                sys.stdout.flush()
        sys.stderr.write("bench.py: rank %d: optional phases cut short: %s\n" % (self.rank, why))
        sys.stderr.flush()
        os._exit(code)

    def failed(self, exc):
        """(c) an optional phase raised on this rank -- a collective that lost its peer raises at once with some backends: the line
        without the optional records, code 1"""
        with self.lock:
            if not self.done:
                self._leave("an optional phase raised %s" % repr(exc)[:200], 1)

    def disarm(self):
        import signal
        with self.lock:
            self.done = True
        try:
            os.write(self.w, b"\0")                                    # wake the thread: it sees `done` and returns
        except OSError:
            pass
        signal.set_wakeup_fd(self.old_fd if self.old_fd is not None else -1)
        signal.signal(signal.SIGTERM, self.old_handler if self.old_handler is not None else signal.SIG_DFL)


def direct_store_ab(torch, dist, capi, sharding, rank, world, n, dev, cdev, stream, dealt, steps, single):
    """A/B of the exchange, AFTER the timed region (it never enters `value`): the peer-to-peer form SURVEY 8(e) allows.  Rank 0
    exports the inter-process handle of ONE whole-image allocation, every peer maps it and traces its mirrored stripes IN
    PLACE on the mapped planes (SIM5GPU_IMG_INPLACE) -- its rows go straight into rank 0's image over xGMI: no payload buffer,
    no gather, no placement pass; rank 0 traces its share and its band into the same image.  A step ends with a stream
    synchronisation and a barrier (the completion signal the gather form gets from the collective), so the figure includes
    that latency and is not pipelined: a lower bound on what the form can do.  Any rank that cannot take part (no IPC between
    the two devices, an allocation that fails) makes EVERY rank skip the phase with the reason in the record -- nothing here
    can take the headline line with it.  Returns the record on rank 0 (None elsewhere)."""
    rec = {"what": "peers store their rows straight into rank 0's IPC-mapped image (SIM5GPU_IMG_INPLACE on the mapped planes); "
                   "a step = every rank's launch + stream synchronisation + barrier; measured after the timed region, never part of `value`"}
    state = {"img": None, "base": None, "err": None, "mapped": False}

    def attempt(fn):
        try:
            fn()
            ok = True
        except Exception as e:                                 # noqa: BLE001 -- reported in the record
            state["err"] = repr(e)[:300]
            ok = False
        return all_ok(dist, torch, cdev, world, ok)

    handle = [None]

    def export():
        if rank == 0:
            state["img"] = capi.DeviceBuffer(2 * n * n * 4)
            state["img"].fill(0)
            state["base"] = state["img"].ptr
            handle[0] = capi.ipc_export(state["img"].ptr)
    if not attempt(export):
        rec["skipped"] = "rank 0 could not export its image: %s" % state["err"]
        return rec if rank == 0 else None
    dist.broadcast_object_list(handle, src=0)

    def open_():
        if rank != 0:
            state["base"] = capi.ipc_open(handle[0])
            state["mapped"] = True                             # THIS rank holds a mapping, whatever the others say
            if os.environ.get("SIM5_BENCH_TEST_PEER_DIES") == "1":
                os.abort()                                     # test hook (tests/test_gpu_bench.py): what a GPU fault on a peer does
            if os.environ.get("SIM5_BENCH_TEST_PEER_DIES") == "hang":
                time.sleep(3600)                               # test hook: a peer that HANGS (the others wait in a collective)
    opened = attempt(open_)
    if opened:
        inc = INCL_DEG / 180.0 * math.pi
        kw = sharding.job_rows(n, rank, world, dealt=dealt)
        descs = []
        if kw["y0"] < kw["y1"]:
            descs.append(capi.image_desc(n, n, SPIN, inc, inplace=True, **kw))
        band = sharding.root_band(n, dealt) if rank == 0 else None
        f_ptr, g_ptr = state["base"], state["base"] + n * n * 4

        bd = capi.image_desc(n, n, SPIN, inc, y0=band[0], y1=band[1]) if band else None

        def launch():
            if bd is not None and descs:
                capi.disk_image_jobs([descs[0], bd], [f_ptr, f_ptr + band[0] * n * 4], [g_ptr, g_ptr + band[0] * n * 4], stream=stream)
            else:
                for d in descs:
                    capi.disk_image_device(d, f_ptr, g_ptr, stream=stream)
                if bd is not None:                             # a band but no dealt stripes on rank 0: the band alone
                    capi.disk_image_device(bd, f_ptr + band[0] * n * 4, g_ptr + band[0] * n * 4, stream=stream)

        # a rank whose launch fails keeps taking part in every barrier (a rank that left the loop would meet the others'
        # barrier with another collective): it stops launching, the failure is agreed on after the loop
        failed = [False]

        def one_step():
            if not failed[0]:
                try:
                    launch()
                    torch.cuda.synchronize()
                except Exception as e:                         # noqa: BLE001 -- reported in the record
                    failed[0] = True
                    state["err"] = repr(e)[:300]
            # (a barrier that raises -- a peer is gone -- is caught by attempt(), whose agreement all-reduce then raises as well:
            # up to main() and into guard.failed(): the line first, then out)
            dist.barrier()

        times = []

        def run():
            for _ in range(2):
                one_step()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            times.append(time.perf_counter() - t0)
            if failed[0]:
                raise RuntimeError(state["err"])
        ran = attempt(run)
        if ran:
            t = torch.tensor([times[0]], dtype=torch.float64, device=cdev)
            allv = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allv, t)
            dt = max(float(v.item()) for v in allv)
            rec.update({"steps": steps, "ms_per_image": 1e3 * dt / steps, "rays_per_s": n * n * steps / dt})
            if rank == 0:
                if single is not None:
                    rec["words_differing_from_single_launch"] = capi.words_differ(state["img"].ptr, single.data_ptr(), 2 * n * n)
                else:
                    import numpy as np
                    g = state["img"].to_numpy(np.float32, (2, n, n))[1]
                    hits = int((g > 0).sum())
                    rec["disk_hits"] = hits
                    rec["disk_hits_reference"] = HEADLINE_HITS
        else:
            rec["skipped"] = "a rank failed while tracing into the mapped image: %s" % state["err"]
    else:
        rec["skipped"] = "a peer could not map rank 0's image (no IPC / peer access between the devices?): %s" % state["err"]

    def close():
        if rank != 0 and state["mapped"]:                       # whoever mapped unmaps, also when another rank could not
            torch.cuda.synchronize()
            capi.ipc_close(state["base"])
            state["mapped"] = False
    attempt(close)
    if world > 1:
        dist.barrier()                                          # the peers have unmapped before rank 0 frees
    state["img"] = None
    return rec if rank == 0 else None


def timed_kernel(capi, stream, launch, reps, warm=1):
    """mean ms per launch, HIP events on the launch stream"""
    for _ in range(warm):
        launch()
    if reps <= 0:
        return 0.0
    e0, e1 = capi.Event(), capi.Event()
    e0.record(stream)
    for _ in range(reps):
        launch()
    e1.record(stream)
    return e0.elapsed_ms(e1) / reps


def executed_from_profiles():
    """FP64 operations the job kernels EXECUTE, from the PMC passes of profiles/collect_jobs.sh (the newest committed
    profiles/r*_jobs_summary.json; SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes, an FMA counted twice): per ray of the
    polarized image, per raytrace() call of the march (the C4 launches of the profiled program), per ray of the surface search
    (its four kernels, the walk kernel eight launches per job), per launch of the spectrum kernel -- and the VALU
    wave-instructions per 64 units next to them.  {} when no summary is there."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_jobs_summary.json")))
    if not found:
        return {}
    try:
        d = json.load(open(found[-1]))
        src = "profiles/" + os.path.basename(found[-1])
        fl = lambda p: (p["SQ_INSTS_VALU_ADD_F64"] + p["SQ_INSTS_VALU_MUL_F64"] + 2.0 * p["SQ_INSTS_VALU_FMA_F64"] + p["SQ_INSTS_VALU_TRANS_F64"]) * 64.0
        out = {}
        k = d.get("s5f::disk_image_polarized_mirror_kernel")
        if k:
            p_ = k["pmc_mean_per_launch"]; n = 2048 * 2048
            out["c3"] = {"flops_per_ray": fl(p_) / n, "valu_wave_instr_per_64_rays": p_["SQ_INSTS_VALU"] / (n / 64.0), "valu_busy_pct": p_.get("VALUBusy"),
                         "lanes_used_pct": p_.get("VALUUtilization"), "source": src}
        k = d.get("s5f::torus_pool_kernel")
        if k and "pmc_mean_per_launch_c4_precision_1" in k:
            p_ = k["pmc_mean_per_launch_c4_precision_1"]
            calls = None
            for ln in d.get("_program_output_under_rocprof", []):
                if "precision 1:" in ln and "mean steps" in ln:
                    calls = float(ln.split("mean steps")[1].split()[0]) * 1024 * 1024
            if calls:
                out["c4"] = {"flops_per_call": fl(p_) / calls, "valu_wave_instr_per_64_calls": p_["SQ_INSTS_VALU"] / (calls / 64.0),
                             "valu_busy_pct": p_.get("VALUBusy"), "lanes_used_pct": p_.get("VALUUtilization"), "source": src}
        ks = [d.get("s5f::surface_%s_kernel" % w) for w in ("setup", "walk", "slow", "finish")]
        if all(ks):
            jobs = ks[0]["launches"]                                    # one set-up launch per job
            tot = sum(fl(k_["pmc_mean_per_launch"]) * k_["launches"] for k_ in ks) / jobs
            valu = sum(k_["pmc_mean_per_launch"]["SQ_INSTS_VALU"] * k_["launches"] for k_ in ks) / jobs
            n = 1024 * 1024
            out["f1"] = {"flops_per_ray": tot / n, "valu_wave_instr_per_64_rays": valu / (n / 64.0),
                         "source": src}
        k = d.get("s5f::disk_spectrum_fast_kernel")
        if k:
            p_ = k["pmc_mean_per_launch"]
            out["f3"] = {"flops_per_launch_1024x1024x128": fl(p_), "valu_wave_instr_per_launch": p_["SQ_INSTS_VALU"], "valu_busy_pct": p_.get("VALUBusy"),
                         "lanes_used_pct": p_.get("VALUUtilization"), "source": src}
        return out
    except Exception:
        return {}


def surface_flops_counted():
    """FP64 operations per ray of the surface search COUNTED on the unmodified reference (oracle/opcount.c `surface`: the C calls
    under python/sim5diskraytrace.py:228-335, ptrace single-stepped), or None where the record is missing"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_opcount_surface.json")), reverse=True):
        try:
            return float(json.load(open(f))["fp64_ops_per_unit"]), "profiles/" + os.path.basename(f)
        except Exception:
            continue
    return None, None


def extra_configs(torch, capi, dev, stream):
    """The other BASELINE.json configurations on ONE GPU, a few launches each (well under a second in total)."""
    out = {}
    rad = math.pi / 180.0
    # C2: 1024^2, a = 0.998, i = 70: g-factor + flux
    n = 1024
    img = torch.zeros((2, n, n), dtype=torch.float32, device=dev)
    d = capi.image_desc(n, n, 0.998, 70.0 * rad)
    ms = timed_kernel(capi, stream, lambda: capi.disk_image_device(d, img[0].data_ptr(), img[1].data_ptr(), stream=stream), 300, 1500)     # ~50 ms of launches first: working clock
    out["c2_1024_thin_disk"] = {"kernel": IMAGE_KERNEL, "kernel_ms": ms, "rays": n * n, "rays_per_s": n * n / ms * 1e3,
                                "roofline_frac": n * n * W_ELL / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                "disk_hits": int((img[1] > 0).sum().item()), "disk_hits_reference": 991579}
    # the same image as one of EIGHT jobs of one job-list launch (sim5gpu_disk_image_jobs: the jobs stream through the GPU back
    # to back -- what a caller with several small images, or a rank with several shares, should use): ms per image
    imgs = torch.zeros((8, 2, n, n), dtype=torch.float32, device=dev)
    fp, gp = [imgs[k, 0].data_ptr() for k in range(8)], [imgs[k, 1].data_ptr() for k in range(8)]
    ms8 = timed_kernel(capi, stream, lambda: capi.disk_image_jobs([d] * 8, fp, gp, stream=stream), 60, 200) / 8
    out["c2_1024_thin_disk"]["job_list_of_8"] = {
        "kernel_ms_per_image": ms8, "rays_per_s": n * n / ms8 * 1e3,
        "roofline_frac": n * n * W_ELL / (ms8 * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
        "same_bits_as_single_launch": bool(all(torch.equal(imgs[k].view(torch.int32), img.view(torch.int32)) for k in range(8))),
        "what": "8 x C2 in ONE launch of disk_image_jobs_kernel; a single launch pays ~6 us of launch gap, ramp and ragged last round"}
    del imgs
    # the share of one of 8 ranks of the 4096^2 headline image (512 rows: mirrored 64-row stripes), one launch: the regime of
    # the multi-GPU value_kernel_only at N = 8
    from sim5_amd import sharding
    dsh = capi.image_desc(4096, 4096, SPIN, INCL_DEG * rad, **sharding.job_rows(4096, 1, 8))
    rows = capi.image_rows(dsh)
    sh = torch.zeros((2, rows, 4096), dtype=torch.float32, device=dev)
    mss = timed_kernel(capi, stream, lambda: capi.disk_image_device(dsh, sh[0].data_ptr(), sh[1].data_ptr(), stream=stream), 300, 1000)
    out["share_512_rows_of_4096"] = {"kernel": IMAGE_KERNEL, "kernel_ms": mss, "rays": rows * 4096, "rays_per_s": rows * 4096 / mss * 1e3,
                                     "roofline_frac": rows * 4096 * W_ELL / (mss * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS}
    del sh
    # C3: 2048^2, a = 0.9, i = 70, Stokes I, Q, U in f64
    n = 2048
    st = torch.zeros((3, n, n), dtype=torch.float64, device=dev)
    d = capi.image_desc(n, n, 0.9, 70.0 * rad, pol_degree=0.1)
    ms = timed_kernel(capi, stream, lambda: capi.disk_image_polarized_device(d, st.data_ptr(), None, stream=stream), 100, 300)   # I, Q, U planes
    out["c3_2048_polarized"] = {"kernel": "disk_image_polarized_mirror_kernel", "kernel_ms": ms, "rays": n * n, "rays_per_s": n * n / ms * 1e3,
                                "roofline_frac": n * n * (W_ELL + W_POL) / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                "disk_hits_reference": 3871553 + 5993}
    gpl = torch.zeros((n, n), dtype=torch.float64, device=dev)       # one more launch with the g plane: hits = g > 0 (a hit inside the
    capi.disk_image_polarized_device(d, st.data_ptr(), None, aux={"g": gpl.data_ptr()}, stream=stream)     # zero-flux band has I = 0)
    out["c3_2048_polarized"]["disk_hits"] = int((gpl > 0).sum().item())
    ex = executed_from_profiles()
    if "c3" in ex:
        e = ex["c3"]
        out["c3_2048_polarized"].update({"executed_flops_per_ray": e["flops_per_ray"], "executed_frac": n * n * e["flops_per_ray"] / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                         "valu_wave_instr_per_64_rays": e["valu_wave_instr_per_64_rays"], "executed_flops_source": e["source"]})
    del st, gpl
    # C4: 1024^2 rays through the torus, raytrace() steps at precision 1.0 with transfer
    n = 1024
    N = n * n
    stokes = torch.zeros((N, 5), dtype=torch.float64, device=dev)
    steps = torch.zeros((N,), dtype=torch.int32, device=dev)
    td = capi.TorusDesc(img=capi.image_desc(n, n, 0.9, 70.0 * rad), r0=100.0, dl_max=1e9, precision=1.0, options=0,
                        max_steps=20000, max_error=1e-2, r_stop_in=1.05, r_stop_out=1.01, shape=0, torus_r=8.0,
                        torus_w=2.0, torus_l=3.5, emis0=1.0, absorb0=0.0)
    ms = timed_kernel(capi, stream, lambda: capi.torus_image_device(td, stokes.data_ptr(), aux={"steps": steps.data_ptr()},
                                                                    stream=stream), 3, 1)
    tot = int(steps.sum(dtype=torch.int64).item())
    out["c4_1024_torus_verlet"] = {"kernel": "torus_start_kernel + torus_pool_kernel", "variant": "fast (SIM5GPU_IMG_DEFAULT; the variant "
                                   "tests/test_gpu_raytrace.py holds to identical step counts and 1e-6 on every ray)", "job_ms": ms, "rays": N, "rays_per_s": N / ms * 1e3,
                                   "raytrace_calls": tot, "steps_per_ray": tot / N, "steps_per_s": tot / ms * 1e3,
                                   "roofline_frac": tot * W_STEP / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                   "roofline_frac_with_counted_flops": tot * W_STEP_MEASURED / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                   "stokes_I_sum": float(stokes[:, 0].sum().item())}
    if "c4" in ex:
        e = ex["c4"]
        out["c4_1024_torus_verlet"].update({"executed_flops_per_call": e["flops_per_call"], "executed_frac": tot * e["flops_per_call"] / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                            "valu_wave_instr_per_64_calls": e["valu_wave_instr_per_64_calls"], "valu_busy_pct": e["valu_busy_pct"],
                                            "lanes_used_pct": e["lanes_used_pct"], "executed_flops_source": e["source"]})
    del stokes, steps
    # the spectrum of the C2 image on 128 energies, fused into the image kernel (k_spectrum.hip).
    # Algorithmic work: the ray (W_ELL + the local frame ~ W_POL) per pixel + 5 FP64 operations and one exp per (pixel, energy)
    import ctypes as C
    import numpy as np
    n, ne = 1024, 128
    d = capi.image_desc(n, n, 0.998, 70.0 * rad)
    E = torch.tensor(10.0 ** np.linspace(-1, 1.5, ne), dtype=torch.float64, device=dev)
    S = torch.zeros(ne, dtype=torch.float64, device=dev)
    capi._lib.sim5gpu_disk_spectrum_workspace.restype = capi.SZ
    wsb = capi._lib.sim5gpu_disk_spectrum_workspace(C.byref(d), capi.I(ne))
    ws = torch.zeros(max(int(wsb), 8), dtype=torch.uint8, device=dev)
    spec = lambda: capi._check(capi._lib.sim5gpu_disk_spectrum(C.byref(d), capi.I(ne), capi.VP(E.data_ptr()), capi.D(1.7), capi.I(1),
                                                                capi.VP(S.data_ptr()), capi.VP(ws.data_ptr()), capi.VP(stream)), "sim5gpu_disk_spectrum")
    ms = timed_kernel(capi, stream, spec, 40, 60)
    w_spec = n * n * (W_ELL + W_POL) + n * n * ne * 6.0
    out["f3_spectrum_1024_x128"] = {"kernel": "disk_spectrum_fast_kernel + spectrum_sum_kernel", "job_ms": ms, "pixels": n * n, "energies": ne,
                                    "pixel_energy_pairs_per_s": n * n * ne / ms * 1e3, "algorithmic_flops": w_spec,
                                    "roofline_frac": w_spec / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                    "spectrum_sum": float(S.sum().item()),
                                    "note": "roofline: (W_ell + 3e2) per pixel + 6 FP64 operations (1 exp) per (pixel, energy) pair over the job time; the kernel spends 15.25 issue slots on a pair (k_spectrum.hip planck_sum), so the algorithmic fraction cannot pass ~0.42 even with the vector unit full"}
    if "f3" in ex:
        e = ex["f3"]
        out["f3_spectrum_1024_x128"].update({"executed_flops_per_launch": e["flops_per_launch_1024x1024x128"],
                                             "executed_frac": e["flops_per_launch_1024x1024x128"] / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                             "valu_wave_instr_per_launch": e["valu_wave_instr_per_launch"], "valu_busy_pct": e["valu_busy_pct"],
                                             "executed_flops_source": e["source"]})
    # the same job on the headline image (4096^2): the launch ramp, the ragged second round of a 1024^2 job (two rounds of resident
    # workgroups) and the summing launch are 1/16 of what they are above
    n4 = 4096
    d4 = capi.image_desc(n4, n4, 0.998, 70.0 * rad)
    wsb4 = capi._lib.sim5gpu_disk_spectrum_workspace(C.byref(d4), capi.I(ne))
    ws4 = torch.zeros(max(int(wsb4), 8), dtype=torch.uint8, device=dev)
    spec4 = lambda: capi._check(capi._lib.sim5gpu_disk_spectrum(C.byref(d4), capi.I(ne), capi.VP(E.data_ptr()), capi.D(1.7), capi.I(1),
                                                                 capi.VP(S.data_ptr()), capi.VP(ws4.data_ptr()), capi.VP(stream)), "sim5gpu_disk_spectrum")
    ms4 = timed_kernel(capi, stream, spec4, 10, 15)
    w_spec4 = n4 * n4 * (W_ELL + W_POL) + n4 * n4 * ne * 6.0
    out["f3_spectrum_4096_x128"] = {"kernel": "disk_spectrum_fast_kernel + spectrum_sum_kernel", "job_ms": ms4, "pixels": n4 * n4, "energies": ne,
                                    "pixel_energy_pairs_per_s": n4 * n4 * ne / ms4 * 1e3, "algorithmic_flops": w_spec4,
                                    "roofline_frac": w_spec4 / (ms4 * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                    "spectrum_sum": float(S.sum().item())}
    # the same two jobs on a UNIFORM energy grid (0.1 ... 30 keV in equal steps): the recurrence along the energies
    # (k_spectrum.hip planck_runs_uniform, ~7.5 issue slots per pair instead of 15.25; the kernel detects the grid itself)
    Eu = torch.tensor(np.linspace(0.1, 30.0, ne), dtype=torch.float64, device=dev)
    specu = lambda: capi._check(capi._lib.sim5gpu_disk_spectrum(C.byref(d), capi.I(ne), capi.VP(Eu.data_ptr()), capi.D(1.7), capi.I(1),
                                                                 capi.VP(S.data_ptr()), capi.VP(ws.data_ptr()), capi.VP(stream)), "sim5gpu_disk_spectrum")
    msu = timed_kernel(capi, stream, specu, 40, 60)
    su = float(S.sum().item())
    specu4 = lambda: capi._check(capi._lib.sim5gpu_disk_spectrum(C.byref(d4), capi.I(ne), capi.VP(Eu.data_ptr()), capi.D(1.7), capi.I(1),
                                                                  capi.VP(S.data_ptr()), capi.VP(ws4.data_ptr()), capi.VP(stream)), "sim5gpu_disk_spectrum")
    msu4 = timed_kernel(capi, stream, specu4, 10, 15)
    out["f3_spectrum_uniform_grid"] = {
        "what": "the two jobs above on 128 energies 0.1 ... 30 keV in EQUAL steps: e^-x of a pixel follows a recurrence along the energies",
        "1024_x128": {"job_ms": msu, "pixel_energy_pairs_per_s": n * n * ne / msu * 1e3, "roofline_frac": w_spec / (msu * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                      "spectrum_sum": su},
        "4096_x128": {"job_ms": msu4, "pixel_energy_pairs_per_s": n4 * n4 * ne / msu4 * 1e3,
                      "roofline_frac": w_spec4 / (msu4 * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS, "spectrum_sum": float(S.sum().item())}}
    del E, Eu, S, ws, ws4
    # the surface search of the reference's Python DiskRaytrace for a thick disk H(R) = 0.25 (R - 2), 1024^2
    # rays (k_surface.hip).  Algorithmic work per ray: ~550 sub-steps of geodesic_follow (ref src/sim5kerr-geod.c:891-925), each
    # r(P) + mu(P) = two jacobi_sncndn (~4 AGM levels, a sincos, the back recurrence) ~ 3.6e2 FP64 operations by SURVEY 8(d)'s
    # counting convention -> W_SURF ~ 2.0e5 per ray that walks (an estimate, stated in DESIGN.md)
    n = 1024
    ax = ((np.arange(n) + .5) / n - .5) * 2 * 20.0
    al, be = np.meshgrid(ax, ax)
    tR = np.linspace(2.0, 60.0, 256); tH = 0.25 * (tR - 2.0)
    tb = {k: torch.tensor(np.ascontiguousarray(v).ravel(), dtype=torch.float64, device=dev) for k, v in (("tR", tR), ("tH", tH), ("al", al), ("be", be))}
    N = n * n
    ob = {k: torch.zeros(N * w, dtype=torch.float64, device=dev) for k, w in (("P", 1), ("r", 1), ("m", 1), ("k", 4))}
    stt = torch.zeros(N, dtype=torch.int32, device=dev)
    surf = lambda: capi._check(capi._lib.sim5gpu_disk_surface_rays(
        capi.D(0.9), capi.D(70.0 * rad), capi.I(tR.size), capi.VP(tb["tR"].data_ptr()), capi.VP(tb["tH"].data_ptr()), capi.SZ(N),
        capi.VP(tb["al"].data_ptr()), capi.VP(tb["be"].data_ptr()), capi.VP(ob["P"].data_ptr()), capi.VP(ob["r"].data_ptr()),
        capi.VP(ob["m"].data_ptr()), capi.VP(ob["k"].data_ptr()), capi.VP(stt.data_ptr()), capi.I(capi.SURFACE_TABLE_CHECKED), capi.VP(stream)),
        "sim5gpu_disk_surface_rays")
    ms = timed_kernel(capi, stream, surf, 10, 10)
    # W_surf: COUNTED on the unmodified reference where the record exists (oracle/opcount.c `surface`: the C library calls under
    # python/sim5diskraytrace.py:228-335 for this very job on a sample grid, ptrace single-stepped); the round-4 estimate otherwise
    w_counted, w_src = surface_flops_counted()
    W_SURF = w_counted if w_counted else 2.0e5
    out["f1_surface_search_1024"] = {"kernel": "surface_setup / walk / slow / finish kernels", "job_ms": ms, "rays": N, "rays_per_s": N / ms * 1e3,
                                     "surface_hits": int((stt == 1).sum().item()),
                                     "algorithmic_flops_per_ray": W_SURF, "algorithmic_flops_source": w_src if w_counted else "estimate: 550 sub-steps x 3.6e2 operations",
                                     "roofline_frac": N * W_SURF / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                     "note": "thick disk H(R) = 0.25 (R - 2), a = 0.9, i = 70 deg, field of view +-20; the reference walks the ray by "
                                             "geodesic_follow (two jacobi_sncndn per sub-step); the kernel walks it by addition theorems re-anchored every "
                                             "48 sub-steps, so it executes far fewer operations than the reference does: executed_frac is its utilisation"}
    if "f1" in ex:
        e = ex["f1"]
        out["f1_surface_search_1024"].update({"executed_flops_per_ray": e["flops_per_ray"], "executed_frac": N * e["flops_per_ray"] / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                              "valu_wave_instr_per_64_rays": e["valu_wave_instr_per_64_rays"], "executed_flops_source": e["source"]})
    del tb, ob, stt
    out["scalar_api_example04_loop"] = scalar_api_rate(capi)
    out["scalar_api_raytrace_loop"] = scalar_raytrace_loop(capi)
    return out


def scalar_api_rate(capi):
    """The SIM5 SCALAR API over the GPU library, the way an unmodified caller uses it: tests/c/shim_probe.c `quiet` is the loop
    of ref examples/04-disk-image-eqplane/disk-image.c:53-105 (geodesic_init_inf, midplane crossing, position_rad, gfactorK,
    disk_nt_flux per pixel, results into two float images) compiled against sim5_amd/host/sim5lib.c and timed INSIDE the
    program, around the loop alone (process start-up, library load and GPU context are not in it).  The image is traced twice in
    one process: the first pass carries the library's one-off set-up of its staging memory, the second does not.  The shim
    answers the calls of a raster-order caller from records it asks for a row (or several rows) ahead, each call after a
    bit-for-bit comparison of its arguments with the ones its record was made for (sim5lib.c: LOOK-AHEAD).  A host-side figure;
    skipped, with the reason, where no C compiler is at hand."""
    import shutil
    import subprocess
    import tempfile
    try:
        cc = shutil.which("gcc") or shutil.which("cc")
        if not cc:
            return {"skipped": "no C compiler on this box"}
        tmp = tempfile.mkdtemp(prefix="s5shim_")
        exe = os.path.join(tmp, "probe")
        host = os.path.join(ROOT, "sim5_amd", "host")
        subprocess.run([cc, os.path.join(ROOT, "tests", "c", "shim_probe.c"), os.path.join(host, "sim5lib.c"), "-I", host, "-o", exe,
                        "-lm", "-O3", "-w", "-fgnu89-inline"], check=True, capture_output=True, timeout=120)
        env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
        runs = {}
        for n in (64, 1024):
            p = subprocess.run([exe, "0.998", "70", str(n), "quiet"], env=env, check=True, capture_output=True, text=True, timeout=300)
            for ln in p.stdout.splitlines():
                w = ln.split()
                if w[:2] == ["#", "quiet"]:
                    runs["%d_pass%s" % (n, w[3])] = {"rays": int(w[5]), "hits": int(w[7]), "loop_s": float(w[-1]), "rays_per_s": int(w[5]) / float(w[-1])}
        env1 = dict(env, SIM5_SHIM_NO_LOOKAHEAD="1")
        p = subprocess.run([exe, "0.998", "70", "64", "quiet"], env=env1, check=True, capture_output=True, text=True, timeout=300)
        one = [float(ln.split()[-1]) for ln in p.stdout.splitlines() if ln.startswith("# quiet")]
        shutil.rmtree(tmp, ignore_errors=True)
        head = runs["1024_pass0"]
        return {"what": "tests/c/shim_probe.c quiet (the example-04 loop, ~5 SIM5 calls per ray, results into two float images) through "
                        "sim5_amd/host/sim5lib.c, timed around the loop inside the program; strict arithmetic, look-ahead by rows",
                "rays_per_s": head["rays_per_s"], "us_per_ray": 1e6 / head["rays_per_s"], "image": "1024 x 1024, first pass of the process",
                "disk_hits": head["hits"], "disk_hits_reference": 991579, "runs": runs,
                "one_launch_per_ray_rays_per_s": 4096 / min(one) if one else None, "a": 0.998, "incl_deg": 70.0}
    except Exception as e:                      # a host-side extra must never take the bench line with it
        return {"skipped": "%s: %s" % (type(e).__name__, str(e)[:200])}


def scalar_raytrace_loop(capi):
    """The OTHER scalar loop north_star names -- raytrace_prepare / raytrace() one call at a time (ref README.md:184-193; the
    timing loop of ref src/sim5unittests.c:116-127) -- through sim5_amd/host/sim5lib.c: tests/c/raytrace_loop.c, six rays, timed
    inside the program around the loops.  Two modes of the shim: look-ahead (sim5gpu_raytrace_record: up to 64 consecutive calls
    of the ray per launch, each call served after a bit-for-bit check of x, k, *step and *rtd) and one launch per call.  The
    SAME program linked against the unmodified reference library runs on one host core beside it (the cpu_baseline leg's
    checker library: timed, never on the product path) -- a ray alone advances one call per ~6 us on the GPU, the dependent chain
    of one lane, so this loop is the one use of the API that stays far behind a CPU core."""
    import shutil
    import subprocess
    import tempfile
    try:
        cc = shutil.which("gcc") or shutil.which("cc")
        if not cc:
            return {"skipped": "no C compiler on this box"}
        tmp = tempfile.mkdtemp(prefix="s5rt_")
        host = os.path.join(ROOT, "sim5_amd", "host")
        src = os.path.join(ROOT, "tests", "c", "raytrace_loop.c")
        exe = os.path.join(tmp, "rtloop")
        subprocess.run([cc, src, os.path.join(host, "sim5lib.c"), "-I", host, "-o", exe, "-lm", "-O3", "-w", "-fgnu89-inline"],
                       check=True, capture_output=True, timeout=120)

        def run(cmd, env):
            p = subprocess.run(cmd, env=env, check=True, capture_output=True, text=True, timeout=300)
            rows = [ln.split() for ln in p.stdout.splitlines() if ln and not ln.startswith("#")]
            tail = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("# raytrace loop")][0]
            return {"calls": int(tail[6]), "loop_s": float(tail[8]), "calls_per_s": float(tail[10]),
                    "calls_per_ray": [int(r[1]) for r in rows], "r_end": [float(r[2]) for r in rows]}
        env = dict(os.environ, SIM5GPU_LIB=capi.LIB_PATH)
        ahead = run([exe, "0.9", "60", "6", "quiet"], env)
        single = run([exe, "0.9", "60", "2", "quiet"], dict(env, SIM5_SHIM_NO_LOOKAHEAD="1"))
        out = {"what": "tests/c/raytrace_loop.c: raytrace() one call at a time (precision 0.01, cap 1e9, 6 rays from r0 = 50), through sim5_amd/host/sim5lib.c",
               "calls_per_s": ahead["calls_per_s"], "us_per_call": 1e6 / ahead["calls_per_s"], "calls": ahead["calls"],
               "calls_per_ray": ahead["calls_per_ray"],
               "one_launch_per_call_calls_per_s": single["calls_per_s"], "one_launch_per_call_us": 1e6 / single["calls_per_s"]}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oraclelib as ol
        if ol.have_reference():
            ref_exe = os.path.join(tmp, "rtloop_ref")
            rdir = os.path.dirname(ol.REF_SO)
            subprocess.run([cc, src, "-I", host, "-o", ref_exe, "-L", rdir, "-lsim5ref", "-Wl,-rpath," + rdir, "-lm", "-O3", "-w", "-fgnu89-inline"],
                           check=True, capture_output=True, timeout=120)
            ref = run([ref_exe, "0.9", "60", "6", "quiet"], dict(os.environ))
            out.update({"reference_one_core_calls_per_s": ref["calls_per_s"], "reference_one_core_us_per_call": 1e6 / ref["calls_per_s"],
                        "same_calls_per_ray_as_the_reference": ref["calls_per_ray"] == ahead["calls_per_ray"],
                        "gpu_over_reference_one_core": ahead["calls_per_s"] / ref["calls_per_s"]})
        shutil.rmtree(tmp, ignore_errors=True)
        return out
    except Exception as e:                      # a host-side extra must never take the bench line with it
        return {"skipped": "%s: %s" % (type(e).__name__, str(e)[:200])}


def c5_on_one_gpu(torch, capi, dev, stream):
    n = 8192
    img = torch.zeros((2, n, n), dtype=torch.float32, device=dev)
    ref = reference_hits_c5()
    per = {}
    tot = 0.0
    ok = True
    for inc in C5_INCLINATIONS:
        d = capi.image_desc(n, n, 0.998, inc * math.pi / 180.0)
        ms = timed_kernel(capi, stream, lambda: capi.disk_image_device(d, img[0].data_ptr(), img[1].data_ptr(), stream=stream), 2, 1)
        hits = int((img[1] > 0).sum().item())
        per[str(inc)] = {"kernel_ms": ms, "rays_per_s": n * n / ms * 1e3, "disk_hits": hits, "disk_hits_reference": ref.get(inc)}
        ok = ok and (ref.get(inc) is None or ref[inc] == hits)
        tot += ms
    return {"kernel": IMAGE_KERNEL, "images": len(C5_INCLINATIONS), "rays": len(C5_INCLINATIONS) * n * n, "scan_ms": tot,
            "rays_per_s": len(C5_INCLINATIONS) * n * n / tot * 1e3,
            "roofline_frac": len(C5_INCLINATIONS) * n * n * W_ELL / (tot * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
            "per_inclination": per, "hits_ok": ok}


def launch_ranks(n):
    """`python bench.py --gpus N ...` without a launcher around it: start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`
    as a child process, pass its stdout / stderr through (inherited: rank 0's one JSON line is this program's one JSON line)
    and return its exit code.  SIGTERM / SIGINT sent to this process are forwarded to the launcher, which ends its ranks."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL and sim5gpu_ipc_* between processes need it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, cwd=os.getcwd())
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, lambda signum, frame: child.send_signal(signum))
    while True:
        try:
            rc = child.wait()
            break
        except KeyboardInterrupt:
            continue
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spin-up", action="store_true", help="skip the ~0.3 s of untimed steps that bring the GPU to its working clock")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2/C3/C4/C5 timings after the headline region")
    ap.add_argument("--workload", choices=["headline", "c5"], default="headline")
    ap.add_argument("--root-band", default="auto",
                    help="N > 1, stripes: rows of the upper half dealt over the ranks; the band left in the middle stays with rank 0, "
                         "which needs no link for it (auto: balanced from a measured kernel and gather | off: deal everything | <rows>)")
    ap.add_argument("--no-direct-ab", action="store_true",
                    help="N > 1: skip the A/B of the exchange after the timed region (peers storing their rows straight into rank 0's IPC-mapped image)")
    ap.add_argument("--mode", choices=["stripes", "images"], default="stripes",
                    help="N > 1: one image in row stripes + one RCCL gather per image (default) | one complete image per GPU, no collective")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: this process becomes the launcher.  It has not imported torch and never
        # touches a GPU; the ranks are CHILD processes of torch.distributed.run (no exec of anything that initialised HIP)
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # test hook: SIM5_BENCH_ONE_GPU=1 runs all ranks on GPU 0 over gloo (tiles staged through the host for the gather),
    # to exercise the N > 1 control flow on a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    one_gpu_test = os.environ.get("SIM5_BENCH_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from sim5_amd.build import build
    dev = torch.device("cuda", local_rank)
    cdev = "cpu" if one_gpu_test else dev   # where small control tensors of a collective live
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # every collective of this program is bounded: a rank that dies or never arrives ends the job after `timeout`
        # (RCCL's watchdog aborts the communicator and the process) instead of holding the node
        timeout = datetime.timedelta(seconds=float(os.environ.get("SIM5_BENCH_TIMEOUT_S", "180")))
        if one_gpu_test:
            dist.init_process_group("gloo", timeout=timeout)
        else:
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")       # a timed-out collective tears the process down
            dist.init_process_group("nccl", device_id=dev, timeout=timeout)
    # rank 0 builds (a no-op when the in-tree library is up to date); nobody loads the library while it may be written
    guarded(dist, torch, cdev, rank, world, "build", (lambda: build()) if rank == 0 else (lambda: None))

    def load():
        import sim5_amd.capi as capi_       # raises if libsim5gpu.so is missing
        capi_.set_device(local_rank)
        return capi_
    capi = guarded(dist, torch, cdev, rank, world, "load libsim5gpu.so", load)
    from sim5_amd import sharding
    # who takes part: world size as the process group reports it, and every rank's device (PCI bus id) -- "RCCL saw N ranks
    # on N distinct GPUs" is in the record, not assumed
    me = {"rank": rank, "local_rank": local_rank, "device_index": local_rank, "pci_bus_id": capi.device_bus_id(local_rank),
          "device_name": torch.cuda.get_device_name(local_rank), "pid": os.getpid()}
    if world > 1:
        roster = [None] * world
        dist.all_gather_object(roster, me)
        group = {"world_size_reported_by_process_group": dist.get_world_size(), "backend": dist.get_backend(),
                 "ranks": roster, "distinct_devices": len({r["pci_bus_id"] for r in roster}),
                 "collective_timeout_s": timeout.total_seconds()}
        if not one_gpu_test:
            agree(dist, torch, cdev, rank, world, group["distinct_devices"] == world and group["world_size_reported_by_process_group"] == world,
                  "one GPU per rank", "ranks share a device: %r" % [r["pci_bus_id"] for r in roster])
    else:
        group = {"world_size_reported_by_process_group": 1, "backend": None, "ranks": [me], "distinct_devices": 1}
    stream = torch.cuda.current_stream().cuda_stream
    striped = world > 1 and args.mode == "stripes"
    c5 = args.workload == "c5"
    n = 8192 if c5 else 4096
    inclinations = C5_INCLINATIONS if c5 else (INCL_DEG,)
    dealt, plan = None, None
    if striped:
        # preflight (no collective inside): this rank can allocate and trace a share of the image at all; a rank that cannot
        # stops everybody here, before the first gather
        def preflight():
            job = ImageJob(capi, sharding, n, INCL_DEG, rank, world, True, stream)
            buf = torch.zeros((2, n if rank == 0 else max(1, sharding.max_local_rows(n, world)), n), dtype=torch.float32, device=dev)
            job.trace(buf, rank == 0)
            torch.cuda.synchronize()
        guarded(dist, torch, cdev, rank, world, "preflight: trace one share", preflight)
        dealt, plan = plan_root_band(torch, dist, capi, sharding, rank, world, n, dev, cdev, stream, one_gpu_test, args.root_band)

    def make_jobs():
        jobs_ = [ImageJob(capi, sharding, n, inc, rank, world, striped, stream, dealt=dealt) for inc in inclinations]
        pipe_ = sharding.TilePipeline(torch, dist, rank, world if striped else 1, n, n, dev, host_staged=one_gpu_test, dealt=dealt,
                                      place=make_placer(capi, sharding, n, world, dealt, stream) if (striped and rank == 0) else None)
        return jobs_, pipe_
    jobs, pipe = guarded(dist, torch, cdev, rank, world, "allocate the image and gather buffers", make_jobs)
    cold_ms = None
    if world == 1 and not c5:
        # what a caller sees who renders ONE image on an idle GPU: after the first launch (code and tables are loaded) the
        # GPU is left idle for 0.3 s, then three images are timed with HIP events -- the clock has not ramped up yet
        jobs[0].trace(pipe.full[0], True)
        torch.cuda.synchronize()
        time.sleep(0.3)
        cold_ms = timed_kernel(capi, stream, lambda: jobs[0].trace(pipe.full[0], True), 3, 0)
    check_every_step = os.environ.get("SIM5_BENCH_CHECK_EVERY_STEP") == "1"        # tests: every assembled image, bit for bit
    step_hits, step_sums = [], []
    single = single_sums = None
    if check_every_step and rank == 0 and not c5:
        # the image of ONE launch on this GPU: what every assembled image must equal, bit for bit (checksum of both planes)
        single = torch.zeros((2, n, n), dtype=torch.float32, device=dev)
        capi.disk_image_device(capi.image_desc(n, n, SPIN, INCL_DEG / 180.0 * math.pi), single[0].data_ptr(), single[1].data_ptr(), stream=stream)
        torch.cuda.synchronize()
        single_sums = plane_checksums(torch, single)

    def step(i):
        for job in jobs:                    # one image per inclination: trace my share, gather (overlapped), rank 0: its band,
            pipe.step(job.trace, job.trace_band if job.band_desc is not None else None,      # then the previous image's rows in place
                      job.trace_both if (job.band_desc is not None and job.desc is not None) else None)
            if check_every_step and rank == 0 and pipe.count > 1:
                # the image of the previous step is complete on this stream from here on (no drain: that is the claim tested)
                prev = pipe.full[(pipe.count - 2) % pipe.nbuf]
                step_hits.append(int((prev[1] > 0).sum().item()))
                if single is not None:
                    step_sums.append(plane_checksums(torch, prev) + [bool(torch.equal(prev.view(torch.int32), single.view(torch.int32)))])

    def fence():
        pipe.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Spin-up, before the W warm-up steps of the contract: about 0.3 s of the very same steps (a number fixed by the workload
    # size, the same on every rank), so that the timed region measures the GPU at its working clock.  Measured on MI355X:
    # the first ~50 ms after idle run at a lower clock -- 0.449 ms per headline image against 0.389 ms in the steady state.
    spin_up = max(1, int(0.3 / (n * n * len(inclinations) / 4.0e10))) if not args.no_spin_up else 0
    if one_gpu_test:
        spin_up = min(spin_up, 4)           # the test hook stages every gather through the host: keep it short
    for i in range(spin_up):
        step(i)
        if i % 64 == 63:
            pipe.drain(); torch.cuda.synchronize()      # keep the launch queue short
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    for job in jobs:
        job.start_timing(args.steps, bracket=(world == 1))   # HIP events on the launch stream over the timed region, on every rank
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    kavg = [job.collect() for job in jobs]  # mean kernel ms per image of the step, this rank
    kstep = sum(kavg)                       # kernel ms per step, this rank
    per_rank = None
    kstep_max = kstep
    if world > 1:
        t = torch.tensor([dt, kstep], dtype=torch.float64, device=cdev)
        allv = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        dt = max(float(v[0].item()) for v in allv)
        kstep_max = max(float(v[1].item()) for v in allv)
        per_rank = {"kernel_ms_per_step": [float(v[1].item()) for v in allv],
                    "rays_per_launch": [sharding.rank_rows(n, r, world, dealt=dealt) * n if striped else n * n for r in range(world)]}
    kernel_only = {}                        # N > 1: filled by the kernel-only region, the first of the phases behind the guard (below)
    # the timed region is over on every rank (the all_gather above); the phases below are optional measurements, each entered
    # only after all ranks have agreed that they are still sound
    agree(dist, torch, cdev, rank, world, bool(kstep == kstep), "timed region", "kernel timing is NaN")
    last = pipe.last_image().clone() if (striped and rank == 0) else None      # (the gather measurement below reuses buffer 0)
    hits = None
    if rank == 0:
        img = last if striped else pipe.last_image()
        hits = int((img[1] > 0).sum().item())
    # N > 1: from here on nothing may take the line with it (LineGuard)
    # (the deadline lies well inside the collective timeout: under RCCL a peer that HANGS in an optional phase makes the watchdog
    # abort the process at `timeout`; the guard must have printed the line and left before that)
    guard = LineGuard(rank, float(os.environ.get("SIM5_BENCH_GUARD_S", min(240.0, 0.5 * timeout.total_seconds())))) if world > 1 else None
    if rank == 0:
        def build_out(direct, c5_scan, cut):
            """the line, from what the timed region measured plus the optional records; cut: the guard's reason when it makes the line"""
            images_per_step = len(inclinations) * (1 if striped or world == 1 else world)
            rays = n * n * images_per_step                     # rays of one step of the whole job
            value = rays * args.steps / dt
            # sanity: the image that came out is the Kerr disk (known hit count of the reference; counted before the optional phases)
            hits_ref = reference_hits_c5().get(inclinations[-1]) if c5 else HEADLINE_HITS
            ok = hits_ref is None or hits == hits_ref
            if c5:
                workload = ("8192x8192 thin-disk images, a=0.998, inclination scan 10..80 deg (8 images per step), elliptic-integral "
                            "path, g-factor + Novikov-Thorne flux (BASELINE.json configs[4])")
            else:
                workload = ("4096x4096 thin-disk image, a=0.998, i=70deg, elliptic-integral path, g-factor + Novikov-Thorne flux "
                            "(BASELINE.json headline / configs[1] at 4096^2)")
            out = {
                "metric": "null geodesics/sec, 4096x4096 Kerr disk image (a=0.998, i=70)" if not c5 else
                          "null geodesics/sec, 8192x8192 Kerr disk images x 8 inclinations (a=0.998)",
                "value": value, "unit": "null geodesics/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
                # BASELINE.md section 3 / SURVEY 8(d): "kernel time incl. image write; gather and D2H copy separately" -- `value` is the
                # whole job (tracing + gather + assembly into a row-major image, every step), `value_kernel_only` the same rays over
                # the slowest rank's kernel time per step (HIP events in the timed region)
                "value_kernel_only": (rays * args.steps / kernel_only["wall_s_max_over_ranks"] if kernel_only.get("wall_s_max_over_ranks") else
                                      rays * 1e3 / kstep_max if kstep_max > 0 else None),
                "value_kernel_only_from_events": rays * 1e3 / kstep_max if kstep_max > 0 else None,
                "kernel_ms_per_step_max_over_ranks": kstep_max,
                "scaling": "strong" if (striped or world == 1) else "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ok": ok, "spin_up_steps": spin_up,
                "config": {"workload": workload, "rays_per_step": rays,
                           "parallelism": ("1 GPU" if world == 1 else
                                           "mirrored pairs of 64-row stripes (rows < %d of the upper half) round-robin over %d GPUs + 1 RCCL gather "
                                           "per image (%d per step), overlapped with the next image; the centred band of %d rows stays on rank 0"
                                           % (dealt, world, len(inclinations), n - 2 * dealt) if striped else
                                           "%d independent images, one per GPU, no collective" % world),
                           "disk_hits": hits, "disk_hits_reference": hits_ref},
            }
            if world == 1:
                out["scaling"] = "strong"
            if cold_ms is not None:
                out["cold_clock"] = {"kernel_ms": cold_ms, "rays_per_s": n * n / cold_ms * 1e3,
                                     "what": "3 images after 0.3 s of idle (no spin-up): what a caller who renders one image sees; "
                                             "`value` is the steady state at the working clock"}
            if check_every_step and rank == 0 and not c5:
                out["hits_of_every_assembled_image"] = step_hits
                ok = ok and all(h == hits_ref for h in step_hits) and len(step_hits) > 0
                # [checksum of the F g^4 plane, of the g plane (sums of the 32-bit patterns), every word equal to the single launch]
                out["plane_checksums_single_launch"] = single_sums
                out["plane_checksums_of_every_assembled_image"] = step_sums
                ok = ok and len(step_sums) > 0 and all(c[2] and c[:2] == single_sums for c in step_sums)
                out["ok"] = ok
            rays_launch = sum(job.rays for job in jobs)            # rays rank 0 traces per step
            achieved = rays_launch * W_ELL / (kstep * 1e-3) / 1e12
            traffic = executed = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if world == 1 and not c5:                      # the PMC traffic figure was collected for the 1-GPU headline launch
                        traffic = tj.get("hbm_bytes_per_launch")
                    executed = tj.get("executed_fp64_flops_per_ray")     # per ray: holds for any row set of the same kernel
                    executed_src = tj.get("source")
                except Exception:
                    traffic = executed = None
            out["roofline"] = {
                "bound": "fp64_valu", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                "kernel": IMAGE_KERNEL, "kernel_ms_avg": kstep / len(jobs), "algorithmic_flops_per_ray": W_ELL,
                "kernel_ms_how": ("one HIP event pair on the launch stream around the K launches of the timed region, / K (launch gaps included)"
                                  if world == 1 else "HIP event pair around every launch of the timed region, mean"),
                "rays_per_launch": rays_launch // len(jobs), "per": "GPU (rank 0)",
                # what the hardware actually does: FP64 add + mul + 2 x fma instructions x 64 lanes counted by the PMC run of the same
                # command (profiles/traffic.json), over the kernel time measured here.  `frac` above is the contract's ALGORITHMIC
                # figure -- the reference's per-pixel work over the kernel time, i.e. an algorithmic speed-up measure once a kernel
                # shares work between rays; `executed_frac` is the fraction of the FP64 VALU peak the kernel's own instructions reach.
                "executed_flops_per_ray": executed,
                "executed_achieved": (rays_launch * executed / (kstep * 1e-3) / 1e12) if executed else None,
                "executed_frac": (rays_launch * executed / (kstep * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS) if executed else None,
                "executed_flops_source": executed_src if executed else None,
                "reference_flops_per_ray_counted": W_ELL_MEASURED,
                "hbm_algorithmic_bytes_per_launch": rays_launch // len(jobs) * 8,
                "hbm_achieved_GBps": rays_launch * 8 / (kstep * 1e-3) / 1e9, "hbm_peak_GBps": PEAK_HBM_GBPS,
                "note": "scalar FP64 special-function work per ray: no MFMA; HBM carries only 8 B/ray of output.  achieved = "
                        "algorithmic flops (SURVEY 8(d): 1.3e3 per ray, the reference's per-pixel work; 1 765 counted on the reference "
                        "binary, oracle/opcount.c) / kernel time; the kernel traces a ray and its mirror image in beta in one lane -- "
                        "they share l, q, the roots and the three R_F integrals -- and reads the flux and K(m) from tables, so it "
                        "EXECUTES fewer FP64 operations per ray than the algorithmic count: executed_* is the hardware-utilisation figure",
            }
            out["process_group"] = group
            if per_rank:
                out["per_rank"] = per_rank
            if kernel_only.get("wall_s_max_over_ranks"):
                out["kernel_only_region"] = dict(kernel_only)
            if world > 1:
                # which number is which: SURVEY 8(d) / BASELINE.md 3 define the metric on KERNEL time (image write
                # included) with the gather reported separately
                out["value_definition"] = ("value = rays x K / max-over-ranks wall time of the timed region, INCLUDING the gather to rank 0 and the "
                                           "placement of the gathered rows (a row-major image on rank 0 every step); value_kernel_only = the same rays "
                                           "x K / max-over-ranks wall time of the kernel-only region (every rank's launches of K images back to back "
                                           "between barriers: kernel_only_region) = the metric of SURVEY 8(d) (kernel time incl. image write; the gather "
                                           "is reported separately: per_rank.gather_ms_alone, link_bound); value_kernel_only_from_events = the same rays "
                                           "over the slowest rank's mean kernel time per step, HIP events inside the overlapped pipeline")
                if direct is not None:
                    out["exchange_ab_direct_stores"] = direct
                    if "ms_per_image" in direct:
                        direct["gather_form_ms_per_image_timed_region"] = 1e3 * dt / args.steps / len(inclinations)
            if striped and plan:
                # Why `value` does not follow the GPU count: a gather to ONE GPU moves (N-1)/N of every image over rank 0's inbound
                # xGMI links, one link per peer, and a GPU writes image rows several times faster than a link carries them
                # (DESIGN.md 8).  The plan's prediction for the chosen split (from the kernel and gather times measured before the
                # timed region) next to what the timed region measured; the kernels themselves scale with the rows (value_kernel_only).
                cand = (plan.get("candidates_dealt_rows") or {}).get(str(dealt))
                measured = 1e3 * dt / args.steps / len(inclinations)
                out["link_bound"] = {
                    "predicted_ms_per_image": cand["predicted_step_ms"] if cand else None,
                    "predicted_root_trace_ms": cand["root_trace_ms"] if cand else None,
                    "predicted_peer_gather_ms": cand["peer_gather_ms"] if cand else None,
                    "measured_ms_per_image": measured,
                    "kernel_ms_per_image_slowest_rank": kstep_max / len(inclinations),
                    "gather_ms_alone": per_rank.get("gather_ms_alone"),
                    "one_gpu_kernel_ms_full_image": plan.get("kernel_ms_full_image"),
                    "statement": "gather-to-root over point-to-point xGMI: beyond N = 2 the step is bound by the inbound links of rank 0, "
                                 "not by tracing; value_kernel_only is the rate of the kernels alone"}
            if not args.no_extra and not c5 and cut is None:
                try:
                    if world == 1:
                        extra = extra_configs(torch, capi, dev, stream)
                        extra["c5_8192_x8_inclinations"] = c5_on_one_gpu(torch, capi, dev, stream)
                        ok = ok and extra["c5_8192_x8_inclinations"]["hits_ok"]
                    elif striped:
                        extra = {"c5_8192_x8_inclinations": c5_scan}
                        ok = ok and extra["c5_8192_x8_inclinations"]["hits_ok"]
                    else:
                        extra = None
                    if extra:
                        out["extra"] = extra
                        out["ok"] = ok
                except Exception as e:                             # the extras are a report, never a blocker for the headline line
                    out["extra"] = {"error": repr(e)}
            if world == 1 and not args.no_cpu_baseline and not c5:
                try:
                    out["cpu_baseline"] = cpu_baseline(n, n)
                except Exception as e:                             # the baseline is a report, never a blocker
                    out["cpu_baseline"] = {"value": None, "error": repr(e)}
            if cut is not None:
                out["optional_phases"] = "cut short: %s -- the records of the gather alone, the peer-to-peer A/B and the C5 scan are missing or partial" % cut
            return out, ok

        if guard:
            guard.fallback = json.dumps(build_out(None, None, "line serialised before the optional phases started")[0])
            guard.make_line = lambda why: json.dumps(build_out(None, None, why)[0])
    # N > 1: SURVEY 8(d) / BASELINE.md 3 quote the scaling metric on KERNEL time (image write included, gather reported apart).  A
    # wall-clocked region of its own: every rank traces its share of K images back to back -- the launches of the timed region, no
    # gather, no placement -- between a barrier + device synchronisation on both sides; max over ranks.  (Buffer: the one that does
    # NOT hold the last assembled image; a rank's rows are rewritten with the same values.)  It runs BEHIND the guard, as the first of
    # the phases after the timed region: it is collective (barriers), so a rank that failed in it would leave the others waiting --
    # the headline line must be safe before it starts; if it does not come through, value_kernel_only falls back to the events.
    def measure_kernel_only():
        def kernel_only_region():
            b = pipe.count % pipe.nbuf
            fence()
            k0 = time.perf_counter()
            for _ in range(args.steps):
                for job in jobs:
                    if not striped:
                        job.trace(pipe.full[0], True)
                    elif rank == 0:
                        view = pipe.full[b][:, pipe.band[0]:pipe.band[1]] if pipe.band else None
                        if job.band_desc is not None and job.desc is not None:
                            job.trace_both(pipe.full[b], view)
                        else:
                            job.trace(pipe.full[b], True)
                            if job.band_desc is not None:
                                job.trace_band(view)
                    else:
                        job.trace(pipe.tiles[b], False)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            return time.perf_counter() - k0
        kdt = kernel_only_region()
        t = torch.tensor([kdt], dtype=torch.float64, device=cdev)
        allk = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allk, t)
        kernel_only.update({"wall_s_max_over_ranks": max(float(v[0].item()) for v in allk), "wall_s_per_rank": [float(v[0].item()) for v in allk],
                       "steps": args.steps,
                       "what": "every rank's launches of the timed region (its share of K images, written to its buffers) back to back, "
                               "no gather and no placement, between barrier + device synchronisation on both sides; host wall clock, max over ranks"})
    direct = c5_scan = None
    try:
        if world > 1:
            measure_kernel_only()
        # one gather on its own (not overlapped), after the timed region: the exchange time next to the compute time
        if striped:
            gms, gsamples = time_gather(torch, dist, capi, pipe, stream, 3 if one_gpu_test else 10, one_gpu_test)
            per_rank["gather_ms_alone"] = gms               # meaningful on rank 0 (the receiver); rank 0 reports its own
            per_rank["gather_ms_samples"] = gsamples
            per_rank["gather_payload_bytes_per_rank"] = 2 * sharding.max_local_rows(n, world, dealt=dealt) * n * 4
            if rank == 0:
                # the placement kernel on its own: the peers' rows of one image to their image rows
                pl = make_placer(capi, sharding, n, world, dealt, stream)
                src = pipe.staged if one_gpu_test else pipe.gathered[0]
                per_rank["place_ms_alone"] = timed_kernel(capi, stream, lambda: pl(src, pipe.full[0]), 10, 2)
                per_rank["assemblies_in_timed_region"] = args.steps * len(inclinations)
            per_rank["root_band_plan"] = plan
        if striped and not c5 and not args.no_direct_ab:
            direct = direct_store_ab(torch, dist, capi, sharding, rank, world, n, dev, cdev, stream, dealt, max(3, min(args.steps, 20)), single)
        if world > 1 and not args.no_extra and not c5 and striped:
            # collective: every rank enters it, and every rank learns whether all came through
            c5_scan = run_c5_scan(torch, dist, capi, sharding, rank, world, dev, cdev, stream, one_gpu_test, dealt_4096=dealt)
    except BaseException as e:                                 # noqa: BLE001 -- with a guard: the line first, then out
        if guard:
            guard.failed(e)
        raise
    if rank != 0:
        if guard:
            guard.disarm()
        if world > 1:
            dist.destroy_process_group()
        return
    if guard:
        guard.disarm()
    out, ok = build_out(direct, c5_scan, None)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def run_c5_scan(torch, dist, capi, sharding, rank, world, dev, cdev, stream, one_gpu_test, reps=2, dealt_4096=None):
    """BASELINE.json configs[4] on all ranks: 8192^2 x 8 inclinations, each image in 64-row stripes over the ranks,
    gathered to rank 0 with one collective per image (overlapped with the tracing of the next image) and put into row
    order there by the placement kernel, inside the timed region.  Called by every rank (collective); rank 0 returns the record."""
    n = 8192
    # the plan of the 4096^2 image carries over: kernel and gather times per row both double with the row length, so the
    # balance point is the same fraction of the image
    dealt = None if dealt_4096 is None or dealt_4096 >= sharding.upper_half(4096) else 2 * dealt_4096
    def make():
        jobs_ = [ImageJob(capi, sharding, n, inc, rank, world, True, stream, dealt=dealt) for inc in C5_INCLINATIONS]
        pipe_ = sharding.TilePipeline(torch, dist, rank, world, n, n, dev, host_staged=one_gpu_test, dealt=dealt,
                                      place=make_placer(capi, sharding, n, world, dealt, stream) if rank == 0 else None)
        return jobs_, pipe_
    jobs, pipe = guarded(dist, torch, cdev, rank, world, "C5 scan: allocate 8192^2 buffers", make)
    ref = reference_hits_c5()
    hits = {}

    def scan(check):
        for job, inc in zip(jobs, C5_INCLINATIONS):
            pipe.step(job.trace, job.trace_band if job.band_desc is not None else None,
                      job.trace_both if (job.band_desc is not None and job.desc is not None) else None)
            if check and rank == 0:
                pipe.drain()
                torch.cuda.synchronize()         # the band is traced by this rank, after the gather was issued
                hits[inc] = int((pipe.last_image()[1] > 0).sum().item())
        pipe.drain()
        dist.barrier()
        torch.cuda.synchronize()

    scan(True)                               # warm-up pass, also the hit-count check of every image
    for job in jobs:
        job.start_timing(reps)
    t0 = time.perf_counter()
    for _ in range(reps):
        scan(False)
    dt = time.perf_counter() - t0
    k = sum(job.collect() for job in jobs)
    t = torch.tensor([dt, k], dtype=torch.float64, device=cdev)
    allv = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allv, t)
    if rank != 0:
        return None
    dt = max(float(v[0].item()) for v in allv)
    rays = len(C5_INCLINATIONS) * n * n
    return {"images": len(C5_INCLINATIONS), "rays": rays, "scan_ms": 1e3 * dt / reps, "rays_per_s": rays * reps / dt,
            "n_gpus": world, "gathers_per_scan": len(C5_INCLINATIONS),
            "gather_payload_bytes_per_rank": 2 * sharding.max_local_rows(n, world, dealt=dealt) * n * 4,
            "dealt_rows_of_upper_half": dealt if dealt is not None else sharding.upper_half(n),
            "kernel_ms_per_scan_per_rank": [float(v[1].item()) for v in allv],
            "disk_hits": {str(k_): v for k_, v in hits.items()},
            "hits_ok": all(ref.get(i) is None or ref[i] == h for i, h in hits.items())}


if __name__ == "__main__":
    main()
