"""Row-tile sharding of an image over the GPUs of one node.

Rays are independent, so the data path needs no exchange while tracing; the only collective is
one gather of the finished tiles to rank 0.  Rows are dealt in stripes of STRIPE rows round-robin
over the ranks: the expensive band around the black-hole shadow (second crossings, RC geodesics)
is then shared by all ranks instead of landing on the one or two ranks that own the middle rows.

Pure index arithmetic (no GPU, no torch) so that it can be checked on CPU with gloo.
"""

STRIPE = 64


def stripes_for_rank(ny, rank, world, stripe=STRIPE):
    """[(y0, y1), ...] image-row ranges owned by `rank`, in increasing row order."""
    out = []
    nstripes = (ny + stripe - 1) // stripe
    for s in range(rank, nstripes, world):
        out.append((s * stripe, min(ny, (s + 1) * stripe)))
    return out


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
