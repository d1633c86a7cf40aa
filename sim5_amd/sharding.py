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


def _dealt(ny, dealt):
    return upper_half(ny) if dealt is None else dealt


def top_stripes_for_rank(ny, rank, world, stripe=STRIPE, dealt=None):
    """[(y0, y1), ...] row ranges of the upper half owned by `rank`, in increasing row order.  Only the rows
    [0, dealt) are dealt (default: the whole upper half); see root_band()."""
    out = []
    half = _dealt(ny, dealt)
    nstripes = (half + stripe - 1) // stripe
    for s in range(rank, nstripes, world):
        out.append((s * stripe, min(half, (s + 1) * stripe)))
    return out


def stripes_for_rank(ny, rank, world, stripe=STRIPE, dealt=None):
    """[(y0, y1), ...] ALL dealt image-row ranges owned by `rank` (upper-half stripes and their mirrors), in increasing
    row order -- the order of the rows in the rank's packed output."""
    top = top_stripes_for_rank(ny, rank, world, stripe, dealt)
    mid = (ny - 1) // 2 if ny % 2 == 1 else -1
    bottom = []
    for (y0, y1) in reversed(top):
        lo, hi = ny - y1, ny - y0
        if y0 <= mid < y1:            # the middle row is its own mirror: it is already in the upper stripe
            lo += 1
        if hi > lo:
            bottom.append((lo, hi))
    return top + bottom


def job_rows(ny, rank, world, stripe=STRIPE, dealt=None):
    """keyword arguments of capi.image_desc for this rank's share: rows of the upper half + SIM5GPU_IMG_MIRROR"""
    return dict(y0=rank * stripe, y1=_dealt(ny, dealt), stripe_rows=stripe, stripe_step=world * stripe, mirror=True)


def root_band(ny, dealt=None):
    """(y0, y1) of the centred band of rows that is NOT dealt and stays with rank 0, or None.

    A gather to rank 0 moves every row but rank 0's own over the links, and a GPU writes image rows several times
    faster than one xGMI link carries them, so an equal split leaves rank 0 waiting for the gather.  With dealt <
    upper_half(ny) only the rows [0, dealt) and their mirrors are dealt round-robin (and gathered); the band
    [dealt, ny - dealt) in the middle is traced by rank 0 straight into the assembled image while the gather of the dealt
    rows is in flight (plan_dealt_rows() balances the two).  A centred band is a symmetric row range, so it takes the
    pairing kernel without a flag."""
    d = _dealt(ny, dealt)
    return (d, ny - d) if ny - d > d else None


def plan_dealt_rows(ny, world, kernel_ms_full_image, gather_ms_equal_split, stripe=STRIPE):
    """Rows of the upper half to deal so that rank 0's tracing (its share + the band) takes as long as the gather of the
    other ranks' shares.  Inputs: the time of ONE full-image kernel on one GPU and of ONE gather of an equal split
    (payload ny / world rows per rank), both measured by the caller.  Per row: c_k = kernel / ny on a GPU, c_g =
    gather / (ny / world) over a link (the links of rank 0 work in parallel: the time is set by one rank's payload).
    Rank 0 traces ny - 2 d (1 - 1/world) rows, a peer sends 2 d / world rows:
        (ny - 2 d (1 - 1/world)) c_k = (2 d / world) c_g   ->   d = ny c_k / (2 ((1 - 1/world) c_k + c_g / world)).
    Taken at the whole number of rounds of stripes (every rank the same number of stripes) next to that point which
    gives the shorter step, at least one round; no band at all if the links keep up."""
    half = upper_half(ny)
    unit = stripe * world
    if world <= 1 or half < unit or not (kernel_ms_full_image > 0) or not (gather_ms_equal_split >= 0):
        return half
    c_k = kernel_ms_full_image / ny
    c_g = gather_ms_equal_split / (ny / world)
    d = ny * c_k / (2.0 * ((1.0 - 1.0 / world) * c_k + c_g / world))
    top = (half // unit) * unit
    lo = int(d // unit) * unit
    best, best_t = half, None
    for cand in (lo, lo + unit, half):              # whole rounds of stripes next to the balance point, or no band at all
        if cand != half and not (unit <= cand <= top and cand < half):
            continue
        root_rows = ny - 2 * cand * (1.0 - 1.0 / world) if cand != half else ny / world
        peer_rows = 2.0 * cand / world if cand != half else ny / world
        t = max(root_rows * c_k, peer_rows * c_g)
        if best_t is None or t < best_t:
            best, best_t = cand, t
    return best


def local_rows(ny, rank, world, stripe=STRIPE, dealt=None):
    return sum(y1 - y0 for (y0, y1) in stripes_for_rank(ny, rank, world, stripe, dealt))


def max_local_rows(ny, world, stripe=STRIPE, dealt=None):
    return max(local_rows(ny, r, world, stripe, dealt) for r in range(world))


def rank_rows(ny, rank, world, stripe=STRIPE, dealt=None):
    """rows a rank traces per image: its dealt share, plus the band on rank 0"""
    band = root_band(ny, dealt) if rank == 0 else None
    return local_rows(ny, rank, world, stripe, dealt) + ((band[1] - band[0]) if band else 0)


def assemble(tiles, ny, world, stripe=STRIPE, dealt=None, out=None):
    """Rank-0 side: put the gathered per-rank tile buffers back into image row order.

    tiles[r] has shape [..., >= local_rows(r), nx] (rows of rank r's stripes, concatenated).  `out`, if given, is the
    [..., ny, nx] image that already holds rank 0's band (root_band()); the dealt rows are written into it.
    Works for numpy arrays and torch tensors alike (slicing + assignment only).
    """
    first = tiles[0]
    if out is None:
        out = first.new_zeros(first.shape[:-2] + (ny, first.shape[-1])) if hasattr(first, "new_zeros") \
            else __import__("numpy").zeros(first.shape[:-2] + (ny, first.shape[-1]), dtype=first.dtype)
    for r in range(world):
        off = 0
        for (y0, y1) in stripes_for_rank(ny, r, world, stripe, dealt):
            out[..., y0:y1, :] = tiles[r][..., off:off + (y1 - y0), :]
            off += y1 - y0
    return out


def place_shares(tiles, ny, world, out, stripe=STRIPE, dealt=None, first=1):
    """Reference form of the root's assembly: rows of the gathered per-rank blocks tiles[first:] to their image rows in
    `out` ([..., ny, nx]); rank 0's own rows are already there (it traces in place).  Slicing + assignment only (numpy or
    torch); bench.py does the same on the GPU with ONE kernel (sim5gpu_image_place_shares)."""
    for r in range(first, world):
        off = 0
        for (y0, y1) in stripes_for_rank(ny, r, world, stripe, dealt):
            out[..., y0:y1, :] = tiles[r][..., off:off + (y1 - y0), :]
            off += y1 - y0
    return out


class TilePipeline:
    """Double-buffered "trace my share, gather the shares to rank 0, put them in row order" loop shared by bench.py and
    the CPU (gloo) test.  One step = one image:

      every rank   trace(buffer, inplace) enqueues the work that fills its share.  Peers: `buffer` is the [2, rows_max, nx]
                   payload of the gather, rows packed (inplace False).  Rank 0: `buffer` is its [2, ny, nx] IMAGE and the
                   share is written at its image rows (inplace True: SIM5GPU_IMG_INPLACE) -- rank 0's rows never move.
      every rank   ONE gather per image (both planes of a rank's stripes are one contiguous payload), asynchronous.
      rank 0       issues the gather BEFORE it traces (it only receives), then traces its share in place and the centred band
                   root_band(ny, dealt) into the same image while the gather is in flight: trace_both(image, band_view) in
                   one launch if given, else trace() then trace_band(view).
      rank 0       when the gather of an image has completed, place(shares, image) copies the peers' rows to their image
                   rows -- so every step ends with a complete row-major [2, ny, nx] image on rank 0.  The placement of
                   image i is issued after the tracing of image i+1 has been enqueued, so gather i overlaps tracing i+1;
                   drain() completes everything outstanding.

    `place(gathered, image)`: gathered is the [world, 2, rows_max, nx] tensor the gather filled (block 0, rank 0's own
    contribution, is not used); default: place_shares() above.  host_staged=True is the one-GPU test hook of bench.py:
    the tile is copied to the host and gathered synchronously over gloo (RCCL refuses two ranks on one device); same
    control flow, same buffers."""

    def __init__(self, torch, dist, rank, world, ny, nx, device, dtype=None, host_staged=False, dealt=None, place=None):
        self.dist, self.rank, self.world, self.ny = dist, rank, world, ny
        self.host_staged = host_staged and world > 1
        self.dealt = dealt if world > 1 else None
        dtype = dtype or torch.float32
        rows_max = max(1, max_local_rows(ny, world, dealt=self.dealt))
        self.rows_max = rows_max
        self.nbuf = 2 if world > 1 else 1
        root = world > 1 and rank == 0
        # the payload of the gather (rank 0 contributes a block nobody reads: its rows are traced in place)
        self.tiles = [torch.zeros((2, rows_max, nx), dtype=dtype, device=device) for _ in range(self.nbuf if not root else 1)]
        gdev = "cpu" if self.host_staged else device
        self.gathered = [torch.zeros((world, 2, rows_max, nx), dtype=dtype, device=gdev) for _ in range(self.nbuf)] if root else [None] * self.nbuf
        self.staged = torch.zeros((world, 2, rows_max, nx), dtype=dtype, device=device) if (root and self.host_staged) else None
        self.band = root_band(ny, self.dealt) if root else None
        self.full = [torch.zeros((2, ny, nx), dtype=dtype, device=device) for _ in range(self.nbuf)] if (root or world == 1) else None
        self.pending = [None] * self.nbuf       # gather in flight on this buffer
        self.unplaced = [False] * self.nbuf     # rank 0: gathered[b] holds rows that are not in full[b] yet
        self._place = place or (lambda gathered, image: place_shares(gathered, ny, world, image, dealt=self.dealt))
        self.count = 0
        self.gathers = 0
        self.placed = 0

    def step(self, trace, trace_band=None, trace_both=None):
        if self.band and trace_band is None and trace_both is None:
            raise ValueError("TilePipeline.step: rank 0 keeps the band %r of the image: trace_band is required" % (self.band,))
        b = self.count % self.nbuf
        self.finish(b)                          # buffer b is free: its gather has completed and its rows are placed
        if self.world == 1:
            trace(self.full[b], True)
        elif self.rank == 0:
            # The root RECEIVES only (its block of the gather is never read: its rows are traced in place), so its part of
            # the collective does not depend on its own tracing: the gather is issued FIRST and runs while rank 0 traces its
            # share and its band -- with `trace_both`, in ONE job-list launch (sim5gpu_disk_image_jobs: the two jobs stream
            # through the GPU back to back instead of paying a launch gap and a ragged last round each).
            self.gather(b)
            if trace_both is not None and self.band:
                trace_both(self.full[b], self.full[b][:, self.band[0]:self.band[1]])
            else:
                trace(self.full[b], True)       # in place: rank 0's rows are final
                if self.band:
                    trace_band(self.full[b][:, self.band[0]:self.band[1]])
            if self.count > 0:
                self.finish((self.count - 1) % self.nbuf)      # the previous image is complete from here on
        else:
            trace(self.tiles[b], False)
            self.gather(b)
        self.count += 1

    def gather(self, b, async_op=True):
        """ONE collective per image: both planes of this rank's stripes in a single contiguous payload."""
        mine = self.tiles[0] if self.rank == 0 else self.tiles[b]
        glist = list(self.gathered[b].unbind(0)) if self.rank == 0 else None
        if self.host_staged:
            self.dist.gather(mine.cpu(), glist, dst=0)
        else:
            w = self.dist.gather(mine, glist, dst=0, async_op=async_op)
            if async_op:
                self.pending[b] = w
        self.unplaced[b] = self.rank == 0
        self.gathers += 1

    def finish(self, b):
        """the gather that last used buffer b has completed; rank 0: the peers' rows are in the image"""
        if self.pending[b] is not None:
            self.pending[b].wait()              # the current stream waits for the collective (no host block on RCCL)
            self.pending[b] = None
        if self.unplaced[b]:
            src = self.gathered[b]
            if self.host_staged:
                self.staged.copy_(src)
                src = self.staged
            self._place(src, self.full[b])
            self.unplaced[b] = False
            self.placed += 1

    def drain(self):
        for b in range(self.nbuf):
            self.finish(b)

    def last_image(self):
        """Rank 0: the most recent complete image, [2, ny, nx] (call after drain() and a device synchronisation)."""
        return self.full[(self.count - 1) % self.nbuf]
