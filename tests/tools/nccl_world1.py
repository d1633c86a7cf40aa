"""RCCL sanity on one GPU: the collectives bench.py issues for N > 1 (async gather of a [2, rows, nx] f32 tile, barrier,
all_gather of a small f64 tensor, synchronous gather) through the real `nccl` backend with a world of ONE rank -- the API
usage and stream semantics are exercised even though nothing crosses a link."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sim5_amd import sharding
import sim5_amd.capi as capi, math
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
n = 4096
stream = torch.cuda.current_stream().cuda_stream
desc = capi.image_desc(n, n, 0.998, math.radians(70.0), y0=0, y1=n, stripe_rows=64, stripe_step=64)
tile = [torch.zeros((2, n, n), dtype=torch.float32, device=dev) for _ in range(2)]
# the receive side as sim5_amd/sharding.TilePipeline lays it out: ONE [world, 2, rows, nx] tensor, the gather list = its views
flat = [torch.zeros((1, 2, n, n), dtype=torch.float32, device=dev) for _ in range(2)]
out = [list(f.unbind(0)) for f in flat]
pending = [None, None]
t0 = time.perf_counter()
for i in range(6):
    b = i % 2
    if pending[b] is not None: pending[b].wait()
    capi.disk_image_device(desc, tile[b][0].data_ptr(), tile[b][1].data_ptr(), stream=stream)
    pending[b] = dist.gather(tile[b], out[b], dst=0, async_op=True)
for w in pending:
    if w is not None: w.wait()
dist.barrier(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
hits = int((out[1][0][1] > 0).sum().item())
t = torch.tensor([dt, 1.0], dtype=torch.float64, device=dev)
allv = [torch.zeros_like(t)]
dist.all_gather(allv, t)
# one synchronous gather between HIP events on the launch stream (how bench.py times the exchange on its own)
e0, e1 = capi.Event(), capi.Event()
e0.record(stream)
dist.gather(tile[0], out[0], dst=0)
e1.record(stream)
gms = e0.elapsed_ms(e1)
torch.cuda.synchronize()
assert gms > 0 and int((flat[0][0][1] > 0).sum().item()) == 15865362
# the placement kernel on the gathered block (a world of one: the share is the whole image, striped)
img = torch.full((2, n, n), float("nan"), dtype=torch.float32, device=dev)
capi.image_place_shares([desc], flat[1].data_ptr(), n, img[0].data_ptr(), img[1].data_ptr(), stream=stream)
torch.cuda.synchronize()
assert torch.equal(img, flat[1][0])
print("nccl world=1 ok: hits %d (reference 15865362), %.2f ms per image incl. self-gather" % (hits, 1e3 * dt / 6))
assert hits == 15865362
dist.destroy_process_group()
