"""Multi-GPU tiling of Image.render: interleaved column stripes + ONE gather.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU in
tests).  Pixels are independent (Image.fs:28-35), so ranks share nothing while rendering; the only
exchange is the collection of the finished column slabs on rank 0.  Because FColor[X,Y] is x-major
(Array2D.fs:30-38) a stripe of S columns is one contiguous run of S*H*3 floats both in a rank's slab
and in the final frame.
"""
import torch
import torch.distributed as dist


def stripe_columns(width, world, rank, stripe):
    """Image columns rendered by `rank`, in slab order — the mapping of ft_render_params:
    x = (c // S) * S * world + rank * S + c % S  for local column c."""
    if width % (stripe * world) != 0:
        raise ValueError(f"width {width} must be a multiple of stripe {stripe} x world {world}")
    cols = width // world
    return [(c // stripe) * stripe * world + rank * stripe + c % stripe for c in range(cols)]


def tiling(width, world, rank, stripe):
    """keyword arguments for DeviceScene.render / render_device describing this rank's share"""
    if world == 1:
        return {}
    if width % (stripe * world) != 0:
        raise ValueError(f"width {width} must be a multiple of stripe {stripe} x world {world}")
    return dict(stripe_width=stripe, stripe_ranks=world, stripe_rank=rank, n_columns=width // world)


def gather_buffers(slab, world, rank, force=False):
    """Pre-allocate (on rank 0) the receive buffer [world, W/world, H, 3] and the frame [W, H, 3]."""
    if (world == 1 and not force) or rank != 0:
        return None, None
    cols, H, C = slab.shape
    return (torch.empty((world, cols, H, C), dtype=slab.dtype, device=slab.device),
            torch.empty((cols * world, H, C), dtype=slab.dtype, device=slab.device))


def gather_frame(slab, world, rank, stripe, frame=None, recv=None, group=None, force=False, mark=None):
    """Collect the per-rank slabs [W/world, H, 3] on rank 0 and de-interleave them into the full
    [W, H, 3] frame.  Returns the frame on rank 0, None elsewhere.  `recv` / `frame` may be the buffers of
    gather_buffers() (bench.py reuses them across steps): the slabs land directly in `recv`, and ONE
    strided copy de-interleaves them — no intermediate copies.  `mark(name)` (optional) is called before the gather
    ("gather"), after it ("deinterleave") and at the end ("done"): FramePipeline records timing events there."""
    if world == 1 and not force:          # force: run the collective even in a 1-rank group (rehearsal on one GPU)
        return slab
    cols, H, C = slab.shape
    if rank == 0 and recv is None:
        recv, frame = gather_buffers(slab, world, rank, force)
    if mark: mark("gather")
    dist.gather(slab, list(recv.unbind(0)) if rank == 0 else None, dst=0, group=group)   # the ONE collective of the path
    if mark: mark("deinterleave")
    if rank != 0:
        if mark: mark("done")
        return None
    if frame is None:
        frame = torch.empty((cols * world, H, C), dtype=slab.dtype, device=slab.device)
    g = recv.view(world, cols // stripe, stripe, H, C)                                # [rank, stripe j, s, y, c]
    frame.view(cols // stripe, world, stripe, H, C).copy_(g.permute(1, 0, 2, 3, 4))
    if mark: mark("done")
    return frame


class FramePipeline:
    """A stream of frames on N GPUs: every rank renders its column stripes of frame k+1 while the slabs of frame k
    travel to rank 0 (the path's ONE collective) and are de-interleaved there — and, with two render lanes, while
    the stragglers of frame k are still marching.

    `renders` is one callable or two: `renders[i](slab)` fills this rank's slab [W/world, H, 3] asynchronously on
    `streams[i]` (for the HIP path: a Device whose stream was set to streams[i]).  With two lanes (two contexts on
    two streams) consecutive frames alternate between them, so frame k+1 starts filling the GPU while the longest
    rays of frame k drain: a persistent-wave kernel ends with a tail of a few long rays (C3: mean 29, maximum > 130
    SDF evaluations per pixel), which costs 2 % of a 4096^2 frame on one GPU but 20 % of the 1/8 share of an 8-GPU
    run (tools/overlap_probe.py).  The gather and the de-interleave run on a side stream.  Order kept by events:
      render k   waits for   gather k-2 (it reuses that slab);   gather k   waits for   render k.
    On CPU tensors (gloo tests) there are no streams and everything runs in order.
    `on_frame(k, frame)` (rank 0, optional) is called with the de-interleaved frame while the side stream is current:
    work it enqueues consumes frame k before frame k+1 overwrites the buffer."""

    def __init__(self, renders, cols, H, world, rank, stripe, device, streams=None, dtype=torch.float32, group=None,
                 force=False, on_frame=None, timed=False):
        self.renders = list(renders) if isinstance(renders, (list, tuple)) else [renders]
        self.world, self.rank, self.stripe, self.group, self.force = world, rank, stripe, group, force
        self.on_frame = on_frame
        self.timed, self._marks = timed, []           # timed: (gather, deinterleave, done) events per frame on the side stream
        self.slabs = [torch.empty((cols, H, 3), dtype=dtype, device=device) for _ in range(2)]
        self.recv, self.frame = gather_buffers(self.slabs[0], world, rank, force)
        self.cuda = self.slabs[0].is_cuda
        self.k = 0
        if self.cuda:
            self.streams = list(streams) if streams is not None else [torch.cuda.current_stream(device)] * len(self.renders)
            if len(self.streams) != len(self.renders):
                raise ValueError("one stream per render lane")
            self.side = torch.cuda.Stream(device=device)
            self.rendered = [torch.cuda.Event() for _ in range(2)]
            self.gathered = [None, None]

    def submit(self):
        """enqueue frame k: render on its lane's stream, gather behind it on the side stream"""
        b = self.k & 1
        lane = b % len(self.renders)
        slab = self.slabs[b]
        if not self.cuda:
            self.renders[lane](slab)
            out = gather_frame(slab, self.world, self.rank, self.stripe, frame=self.frame, recv=self.recv, group=self.group, force=self.force)
            if self.on_frame is not None and self.rank == 0:
                self.on_frame(self.k, out)
        else:
            st = self.streams[lane]
            if self.gathered[b] is not None:
                st.wait_event(self.gathered[b])            # frame k-2 has left this slab
            with torch.cuda.stream(st):
                self.renders[lane](slab)
            self.rendered[b].record(st)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.rendered[b])
                mark = None
                if self.timed:
                    evs = {}
                    self._marks.append(evs)

                    def mark(name, evs=evs):
                        evs[name] = torch.cuda.Event(enable_timing=True)
                        evs[name].record(self.side)
                out = gather_frame(slab, self.world, self.rank, self.stripe, frame=self.frame, recv=self.recv, group=self.group, force=self.force, mark=mark)
                if self.on_frame is not None and self.rank == 0:
                    self.on_frame(self.k, out)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.gathered[b] = ev
        self.k += 1

    def timings(self, reset=True):
        """(gather ms, de-interleave ms) summed over the frames submitted since the last call; call after drain() + synchronize.
        The gather time of a rank includes waiting for the slowest rank's render of that frame."""
        g = d = 0.0
        for evs in self._marks:
            if "gather" in evs and "deinterleave" in evs:
                g += evs["gather"].elapsed_time(evs["deinterleave"])
            if "deinterleave" in evs and "done" in evs:
                d += evs["deinterleave"].elapsed_time(evs["done"])
        if reset:
            self._marks = []
        return g, d

    def drain(self):
        """make the current stream wait for every render and gather submitted so far"""
        if self.cuda:
            cur = torch.cuda.current_stream()
            for st in self.streams:
                if st != cur:
                    cur.wait_stream(st)
            cur.wait_stream(self.side)
