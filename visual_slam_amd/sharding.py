"""Frame sharding over GPUs and the host-side gather of results.

Frames (and stereo pairs) are independent, so a batch is cut into contiguous ranges, one per rank,
and no collective runs on the data path. `group` keeps the images of a stereo pair (group=2) on
the same rank so a later on-device stereo matcher needs no exchange. After extraction every rank
holds its own frames' results; `gather_results` brings them to rank 0 through
torch.distributed's object gather (host side; gloo or the CPU side of any backend).
"""


def shard_range(nframes, world, rank, group=1):
    """Contiguous [lo, hi) of frames for `rank`; sizes differ by at most one group."""
    if nframes % group:
        raise ValueError("nframes must be a multiple of group")
    ngroups = nframes // group
    base, rem = divmod(ngroups, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo * group, hi * group


def extract_sharded(extract_fn, frames, world, rank, group=1):
    """Runs extract_fn(frames[lo:hi]) -> list of per-frame results on this rank's shard."""
    lo, hi = shard_range(len(frames), world, rank, group)
    return lo, (extract_fn(frames[lo:hi]) if hi > lo else [])


def gather_results(local, dist=None, dst=0):
    """Host-side gather: returns the frame-ordered list of all ranks' results on rank `dst`, else None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(list(local), out, dst=dst)
    if out is None:
        return None
    merged = []
    for part in out:       # ranks own increasing contiguous ranges
        merged.extend(part)
    return merged
