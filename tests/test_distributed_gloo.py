"""The N > 1 path on CPU: world_size-2 (and 4) gloo process groups exercise stripe mapping, the single
gather and the de-interleave exactly as bench.py / a multi-GPU host run them (the render itself needs a
GPU, so each rank fills its slab with the values the kernel would write: a function of the image column)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fraytracer_amd import distributed as ftd


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _pixel(x, H):
    y = torch.arange(H, dtype=torch.float32)
    return torch.stack([x * 1000.0 + y, x - y, torch.full((H,), float(x))], dim=1)     # [H, 3]


def _worker(rank, world, port, W, H, stripe, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols = ftd.stripe_columns(W, world, rank, stripe)
        slab = torch.stack([_pixel(x, H) for x in cols])                                # what this rank would render
        frame = ftd.gather_frame(slab, world, rank, stripe)
        if rank == 0:
            want = torch.stack([_pixel(x, H) for x in range(W)])
            q.put(bool(torch.equal(frame, want)))
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,stripe", [(2, 64, 16), (2, 32, 1), (4, 64, 8)])
def test_gather_reassembles_frame(world, W, stripe):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, 5, stripe, q)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) is True


def _pipeline_worker(rank, world, port, W, H, stripe, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols = ftd.stripe_columns(W, world, rank, stripe)
        frame_no = [0]

        def render(slab):                                   # frame k = the column pattern + k
            slab.copy_(torch.stack([_pixel(x, H) for x in cols]) + float(frame_no[0]))
            frame_no[0] += 1

        got = []
        lanes = [render, render] if W == 64 else render      # two render lanes alternate frames
        pipe = ftd.FramePipeline(lanes, len(cols), H, world, rank, stripe, "cpu", on_frame=lambda k, f: got.append((k, f.clone())))
        for _ in range(5):
            pipe.submit()
        pipe.drain()
        if rank == 0:
            want = torch.stack([_pixel(x, H) for x in range(W)])
            q.put(len(got) == 5 and all(k == i and torch.equal(f, want + float(i)) for i, (k, f) in enumerate(got)))
        else:
            assert not got
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frame_pipeline_delivers_every_frame_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, 64, 5, 8, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) is True


def test_stripe_mapping_partitions_the_columns():
    for world, W, S in [(1, 64, 16), (2, 64, 16), (4, 128, 8), (8, 4096, 16)]:
        seen = sorted(x for r in range(world) for x in ftd.stripe_columns(W, world, r, S))
        assert seen == list(range(W))
    with pytest.raises(ValueError):
        ftd.stripe_columns(100, 8, 0, 16)
    assert ftd.tiling(4096, 1, 0, 16) == {}
    assert ftd.tiling(4096, 8, 3, 16) == dict(stripe_width=16, stripe_ranks=8, stripe_rank=3, n_columns=512)
