"""Row-tile sharding of an image over the GPUs of one node.

Rays are independent, so the data path needs no exchange while tracing; the only collective is
one gather of the finished tiles to rank 0.  Rows are dealt in stripes of STRIPE rows round-robin
over the ranks: the expensive band around the black-hole shadow (second crossings, RC geodesics)
is then shared by all ranks instead of landing on the one or two ranks that own the middle rows.

Stripes are dealt in MIRRORED PAIRS: the upper half of the image is cut into stripes, stripe s goes to rank
s mod world, and the rank that owns rows [y0, y1) also owns their mirror images [ny - y1, ny - y0).  A ray and its
mirror image in beta share the geodesic, and the image kernel traces such a pair in one lane
(SIM5GPU_IMG_MIRROR, include/sim5gpu.h) -- the split keeps that saving on every rank.  A rank's rows are packed
in increasing row order: its stripes of the upper half, then their mirrors.

Pure index arithmetic (no GPU, no torch) so that it can be checked on CPU with gloo.
"""

STRIPE = 64


def upper_half(ny):
    """rows [0, upper_half(ny)) are dealt; the others are their mirrors (the middle row of an odd ny is its own)"""
    return (ny + 1) // 2


def top_stripes_for_rank(ny, rank, world, stripe=STRIPE):
    """[(y0, y1), ...] row ranges of the upper half owned by `rank`, in increasing row order."""
    out = []
    half = upper_half(ny)
    nstripes = (half + stripe - 1) // stripe
    for s in range(rank, nstripes, world):
        out.append((s * stripe, min(half, (s + 1) * stripe)))
    return out


def stripes_for_rank(ny, rank, world, stripe=STRIPE):
    """[(y0, y1), ...] ALL image-row ranges owned by `rank` (upper-half stripes and their mirrors), in increasing
    row order -- the order of the rows in the rank's packed output."""
    top = top_stripes_for_rank(ny, rank, world, stripe)
    mid = (ny - 1) // 2 if ny % 2 == 1 else -1
    bottom = []
    for (y0, y1) in reversed(top):
        lo, hi = ny - y1, ny - y0
        if y0 <= mid < y1:            # the middle row is its own mirror: it is already in the upper stripe
            lo += 1
        if hi > lo:
            bottom.append((lo, hi))
    return top + bottom


def job_rows(ny, rank, world, stripe=STRIPE):
    """keyword arguments of capi.image_desc for this rank's share: rows of the upper half + SIM5GPU_IMG_MIRROR"""
    return dict(y0=rank * stripe, y1=upper_half(ny), stripe_rows=stripe, stripe_step=world * stripe, mirror=True)


def local_rows(ny, rank, world, stripe=STRIPE):
    return sum(y1 - y0 for (y0, y1) in stripes_for_rank(ny, rank, world, stripe))


def max_local_rows(ny, world, stripe=STRIPE):
    return max(local_rows(ny, r, world, stripe) for r in range(world))


def assemble(tiles, ny, world, stripe=STRIPE):
    """Rank-0 side: put the gathered per-rank tile buffers back into image row order.

    tiles[r] has shape [..., >= local_rows(r), nx] (rows of rank r's stripes, concatenated).
    Works for numpy arrays and torch tensors alike (slicing + assignment only).
    """
    first = tiles[0]
    out = first.new_zeros(first.shape[:-2] + (ny, first.shape[-1])) if hasattr(first, "new_zeros") \
        else __import__("numpy").zeros(first.shape[:-2] + (ny, first.shape[-1]), dtype=first.dtype)
    for r in range(world):
        off = 0
        for (y0, y1) in stripes_for_rank(ny, r, world, stripe):
            out[..., y0:y1, :] = tiles[r][..., off:off + (y1 - y0), :]
            off += y1 - y0
    return out


class TilePipeline:
    """Double-buffered "trace my stripes, gather them to rank 0" loop shared by bench.py and the CPU
    (gloo) test.  `trace(buffer)` must enqueue the work that fills `buffer` ([2, rows_max, nx] tensor);
    the gather of image i is issued asynchronously and overlaps the tracing of image i+1; `drain()`
    waits for every outstanding gather.  With world == 1 there is no gather and a single buffer.

    host_staged=True is the one-GPU test hook of bench.py: the tile is copied to the host and gathered
    synchronously over gloo (RCCL refuses two ranks on one device); same control flow, same buffers."""

    def __init__(self, torch, dist, rank, world, ny, nx, device, dtype=None, host_staged=False):
        self.dist, self.rank, self.world, self.ny = dist, rank, world, ny
        self.host_staged = host_staged and world > 1
        dtype = dtype or torch.float32
        rows_max = max_local_rows(ny, world)
        self.nbuf = 2 if world > 1 else 1
        self.tiles = [torch.zeros((2, rows_max, nx), dtype=dtype, device=device) for _ in range(self.nbuf)]
        gdev = "cpu" if self.host_staged else device
        self.gathered = [[torch.zeros((2, rows_max, nx), dtype=dtype, device=gdev) for _ in range(world)]
                         for _ in range(self.nbuf)] if (world > 1 and rank == 0) else [None] * self.nbuf
        self.pending = [None] * self.nbuf
        self.count = 0
        self.gathers = 0

    def step(self, trace):
        b = self.count % self.nbuf
        if self.pending[b] is not None:
            self.pending[b].wait()              # the gather that last read this buffer has finished
            self.pending[b] = None
        trace(self.tiles[b])
        if self.world > 1:
            self.gather(b)
        self.count += 1

    def gather(self, b, async_op=True):
        """ONE collective per image: both planes of this rank's stripes in a single contiguous payload."""
        if self.host_staged:
            self.dist.gather(self.tiles[b].cpu(), self.gathered[b], dst=0)
        else:
            w = self.dist.gather(self.tiles[b], self.gathered[b], dst=0, async_op=async_op)
            if async_op:
                self.pending[b] = w
        self.gathers += 1

    def drain(self):
        for b in range(self.nbuf):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def last_image(self):
        """Rank 0: the most recent complete image, [2, ny, nx] (call after drain())."""
        b = (self.count - 1) % self.nbuf
        if self.world == 1:
            return self.tiles[b][:, :self.ny]
        return assemble(self.gathered[b], self.ny, self.world)
