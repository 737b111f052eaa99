"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): shard, 'extract', host-side gather.

The extraction itself needs a GPU, so the per-frame work is replaced by the CPU oracle on tiny
frames; what is covered is everything N > 1 adds: the partition, the absence of any data-path
collective, and the frame-ordered gather on rank 0.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, nframes, q):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from oracle import orb_oracle as O
    from visual_slam_amd import sharding, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = synth.batch("blocks", 160, 120, nframes)        # every rank can build the same batch
    e = O.OracleExtractor(200, 1.2, 4, 20, 7)

    def extract(fr):
        return [e.extract(f) for f in fr]

    lo, local = sharding.extract_sharded(extract, frames, world, rank, group=2)
    merged = sharding.gather_results(local, dist, dst=0)
    if rank == 0:
        ref = extract(frames)
        ok = len(merged) == nframes and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(merged, ref))
        q.put(bool(ok))
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
