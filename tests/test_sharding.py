"""Row-stripe sharding over ranks: index arithmetic, and the gather/assemble path with
torch.distributed (gloo, world_size 2) on CPU."""
import os
import socket

import numpy as np
import pytest

from sim5_amd import sharding


@pytest.mark.parametrize("ny,world", [(4096, 1), (4096, 2), (4096, 8), (8192, 8), (100, 3), (101, 3), (64, 8), (65, 2),
                                      (129, 2), (1, 4), (2, 2), (1000, 3), (1001, 3)])
def test_stripes_partition_the_rows(ny, world):
    seen = np.zeros(ny, int)
    for r in range(world):
        rows, last = 0, 0
        owned = np.zeros(ny, bool)
        for (y0, y1) in sharding.stripes_for_rank(ny, r, world):
            assert 0 <= y0 < y1 <= ny and y0 >= last           # increasing row order: the order of the packed output
            last = y1
            seen[y0:y1] += 1
            owned[y0:y1] = True
            rows += y1 - y0
        assert rows == sharding.local_rows(ny, r, world) <= sharding.max_local_rows(ny, world)
        assert np.array_equal(owned, owned[::-1])              # a rank owns the mirror image of every row it owns
        top = sum(y1 - y0 for (y0, y1) in sharding.top_stripes_for_rank(ny, r, world))
        assert rows in (2 * top, 2 * top - 1)
    assert (seen == 1).all()


@pytest.mark.parametrize("ny,world,dealt", [(4096, 8, 1536), (4096, 2, 640), (4096, 4, 1024), (1001, 3, 192), (200, 2, 128 - 64),
                                            (8192, 8, 2560), (4096, 8, 2048)])
def test_root_band_plus_dealt_rows_partition_the_image(ny, world, dealt):
    """weighted split: rows [0, dealt) of the upper half and their mirrors are dealt, the band in the middle is rank 0's"""
    seen = np.zeros(ny, int)
    for r in range(world):
        for (y0, y1) in sharding.stripes_for_rank(ny, r, world, dealt=dealt):
            seen[y0:y1] += 1
        assert sharding.rank_rows(ny, r, world, dealt=dealt) == sharding.local_rows(ny, r, world, dealt=dealt) + (
            (ny - 2 * dealt) if (r == 0 and sharding.root_band(ny, dealt)) else 0)
    band = sharding.root_band(ny, dealt)
    if band:
        assert band == (dealt, ny - dealt)
        seen[band[0]:band[1]] += 1
    assert (seen == 1).all()
    assert sum(sharding.rank_rows(ny, r, world, dealt=dealt) for r in range(world)) == ny


def test_plan_balances_root_tracing_and_gather():
    # links much slower than the kernel: most of the image stays on rank 0; fast links: equal split, no band
    for world in (2, 4, 8):
        d = sharding.plan_dealt_rows(4096, world, 0.48, 1.1 * 2 / world)
        assert d % (64 * world) == 0 and 64 * world <= d < 2048
        root = sharding.rank_rows(4096, 0, world, dealt=d) * 0.48 / 4096
        peer = sharding.rank_rows(4096, 1, world, dealt=d) * (1.1 * 2 / world) / (4096 / world)
        assert 0.5 < root / peer < 2.0, (world, d, root, peer)
        assert sharding.plan_dealt_rows(4096, world, 0.48, 1e-4) == 2048 and sharding.root_band(4096, 2048) is None
    assert sharding.plan_dealt_rows(4096, 1, 0.5, 0.5) == 2048
    assert sharding.plan_dealt_rows(100, 3, 1.0, 5.0) == 50            # fewer rows than one round of stripes: plain split
    assert sharding.plan_dealt_rows(4096, 8, float("nan"), 1.0) == 2048


def test_library_counts_the_rows_of_a_mirrored_job():
    """sim5gpu_image_rows (host arithmetic, no GPU needed) agrees with the dealing for the job description bench.py makes"""
    from sim5_amd import capi
    for ny, world in [(4096, 8), (1000, 3), (1001, 3), (129, 2), (257, 8), (65, 2)]:
        for r in range(world):
            kw = sharding.job_rows(ny, r, world)
            want = sharding.local_rows(ny, r, world)
            if kw["y0"] >= kw["y1"]:
                assert want == 0
                continue
            d = capi.image_desc(ny, ny, 0.9, 1.0, **kw)
            assert capi.image_rows(d) == want, (ny, world, r)


def test_assemble_restores_row_order():
    ny, nx, world = 200, 7, 3
    full = np.arange(2 * ny * nx, dtype=np.float32).reshape(2, ny, nx)
    rmax = sharding.max_local_rows(ny, world)
    tiles = []
    for r in range(world):
        t = np.zeros((2, rmax, nx), np.float32)
        off = 0
        for (y0, y1) in sharding.stripes_for_rank(ny, r, world):
            t[:, off:off + y1 - y0] = full[:, y0:y1]
            off += y1 - y0
        tiles.append(t)
    assert np.array_equal(sharding.assemble(tiles, ny, world), full)


def _worker(rank, world, port, ny, nx, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # stand-in for the kernel: pixel value = global row * nx + column (plane 0), its negative (plane 1)
    rmax = sharding.max_local_rows(ny, world)
    tile = torch.zeros((2, rmax, nx), dtype=torch.float32)
    off = 0
    for (y0, y1) in sharding.stripes_for_rank(ny, rank, world):
        rows = torch.arange(y0, y1, dtype=torch.float32)[:, None] * nx + torch.arange(nx, dtype=torch.float32)[None, :]
        tile[0, off:off + y1 - y0] = rows
        tile[1, off:off + y1 - y0] = -rows
        off += y1 - y0
    gathered = [torch.zeros_like(tile) for _ in range(world)] if rank == 0 else None
    dist.barrier()
    dist.gather(tile, gathered, dst=0)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        img = sharding.assemble(gathered, ny, world)
        expect = torch.arange(ny * nx, dtype=torch.float32).reshape(ny, nx)
        q.put((bool(torch.equal(img[0], expect) and torch.equal(img[1], -expect)), float(t.item())))
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, ny, nx, nimages, q, host_staged=False, dealt=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe = sharding.TilePipeline(torch, dist, rank, world, ny, nx, torch.device("cpu"), host_staged=host_staged, dealt=dealt)
    state = {"img": 0}

    def trace_band(view):                         # rank 0 only: the band [dealt, ny - dealt) of the image being made
        y0, y1 = sharding.root_band(ny, dealt)
        rows = torch.arange(y0, y1, dtype=torch.float32)[:, None] * nx + torch.arange(nx, dtype=torch.float32)[None, :]
        view[0] = rows + 1e6 * (state["img"] - 1)
        view[1] = -rows

    def trace(buf, inplace):                      # stand-in kernel: value = image number * 1e6 + row * nx + col
        off = 0
        for (y0, y1) in sharding.stripes_for_rank(ny, rank, world, dealt=dealt):
            rows = torch.arange(y0, y1, dtype=torch.float32)[:, None] * nx + torch.arange(nx, dtype=torch.float32)[None, :]
            lo = y0 if inplace else off             # rank 0 writes its rows where they belong in the image
            buf[0, lo:lo + y1 - y0] = rows + 1e6 * state["img"]
            buf[1, lo:lo + y1 - y0] = -rows
            off += y1 - y0
        assert inplace == (rank == 0) and buf.shape[1] == (ny if inplace else sharding.max_local_rows(ny, world, dealt=dealt))
        state["img"] += 1

    expect = torch.arange(ny * nx, dtype=torch.float32).reshape(ny, nx)
    good = True
    for i in range(nimages):
        pipe.step(trace, trace_band if sharding.root_band(ny, dealt) else None)
        if rank == 0 and i > 0:                   # every step ends with the previous image complete, in row order
            prev = pipe.full[(pipe.count - 2) % pipe.nbuf]
            good = good and bool(torch.equal(prev[0], expect + 1e6 * (i - 1)) and torch.equal(prev[1], -expect)) and pipe.placed == i
    pipe.drain()
    dist.barrier()
    if rank == 0:
        img = pipe.last_image()
        q.put(good and pipe.placed == nimages and bool(torch.equal(img[0], expect + 1e6 * (nimages - 1)) and torch.equal(img[1], -expect)))
    dist.destroy_process_group()


@pytest.mark.parametrize("nimages,host_staged,dealt", [(1, False, None), (4, False, None), (5, False, None), (3, True, None),
                                                       (4, False, 64), (3, True, 64), (2, False, 100)])
def test_overlapped_gather_pipeline_gloo(nimages, host_staged, dealt):
    """The bench's double-buffered trace/gather loop with world size 2 on CPU (host_staged: the synchronous,
    host-staged gather of bench.py's one-GPU test hook)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, 200, 16, nimages, q, host_staged, dealt)) for r in range(2)]
    for p in ps:
        p.start()
    ok = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_pipeline_refuses_a_missing_band_tracer():
    """rank 0 keeps a band but the caller passes no trace_band: an error before any collective is issued"""
    import torch

    class NoDist:
        def gather(self, *a, **k):
            raise AssertionError("a collective was issued")
    pipe = sharding.TilePipeline(torch, NoDist(), 0, 2, 200, 16, torch.device("cpu"), dealt=64)
    with pytest.raises(ValueError, match="trace_band"):
        pipe.step(lambda buf, inplace: None)


def test_place_shares_restores_row_order():
    ny, nx, world = 201, 5, 3
    full = np.arange(2 * ny * nx, dtype=np.float32).reshape(2, ny, nx)
    for dealt in (None, 64):
        rmax = sharding.max_local_rows(ny, world, dealt=dealt)
        tiles = np.zeros((world, 2, rmax, nx), np.float32)
        out = np.zeros_like(full)
        for r in range(world):
            off = 0
            for (y0, y1) in sharding.stripes_for_rank(ny, r, world, dealt=dealt):
                if r == 0:
                    out[:, y0:y1] = full[:, y0:y1]                     # rank 0 traced in place
                tiles[r][:, off:off + y1 - y0] = full[:, y0:y1]
                off += y1 - y0
        band = sharding.root_band(ny, dealt)
        if band:
            out[:, band[0]:band[1]] = full[:, band[0]:band[1]]
        assert np.array_equal(sharding.place_shares(tiles, ny, world, out, dealt=dealt), full)


def test_library_row_map_is_the_dealing_rule():
    """sim5gpu_image_row_map (host arithmetic of the library: the rule its kernels and sim5gpu_image_place_shares use)
    gives every packed row of a rank's job the image row sharding.stripes_for_rank says"""
    from sim5_amd import capi
    for ny, world, dealt in [(4096, 8, None), (4096, 4, 1024), (1001, 3, 192), (129, 2, None), (257, 8, None), (65, 2, None)]:
        for r in range(world):
            kw = sharding.job_rows(ny, r, world, dealt=dealt)
            if kw["y0"] >= kw["y1"]:
                continue
            rows = capi.image_row_map(capi.image_desc(ny, ny, 0.9, 1.0, **kw))
            want = [y for (y0, y1) in sharding.stripes_for_rank(ny, r, world, dealt=dealt) for y in range(y0, y1)]
            assert rows.tolist() == want, (ny, world, dealt, r)


def test_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, 200, 16, q)) for r in range(2)]
    for p in ps:
        p.start()
    ok, tmax = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and tmax == 2.0


_ABORT_SCRIPT = r'''
import os, sys, datetime, time
sys.path.insert(0, os.environ["S5_ROOT"])
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
bench.guarded(dist, torch, "cpu", rank, world, "phase one", lambda: None)          # everybody fine: goes on
print("rank %d passed phase one" % rank, flush=True)
def boom():
    if rank == 1:
        raise RuntimeError("synthetic failure on rank 1")
bench.guarded(dist, torch, "cpu", rank, world, "phase two", boom)                 # one rank fails: nobody returns
print("rank %d passed phase two" % rank, flush=True)
t = torch.zeros(1)
dist.all_reduce(t)                                                                # would hang a survivor without the agreement
'''


def test_one_failing_rank_stops_all_ranks_before_the_next_collective(tmp_path):
    """bench.py's collective-safe abort: an exception on ONE rank inside a guarded phase makes EVERY
    rank exit non-zero at the agreement that ends the phase -- nobody walks into the next collective and hangs -- and rank 0
    prints a JSON line saying so.  World size 2 over gloo, on CPU."""
    import subprocess, sys, time
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "abort.py"
    script.write_text(_ABORT_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    t0 = time.time()
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), S5_ROOT=root)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert time.time() - t0 < 200
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 3, (r, p.returncode, so[-500:], se[-1500:])
        assert "passed phase one" in so and "passed phase two" not in so
    assert '"ok": false' in outs[0][0] and "phase two" in outs[0][0]
    assert "synthetic failure on rank 1" in outs[1][1]
