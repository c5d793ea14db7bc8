"""Run by tests/test_gpu_parity.py::test_frame_pipeline_over_rccl in its own process: a 1-rank RCCL group on
cuda:0 drives fraytracer_amd.distributed.FramePipeline exactly as bench.py does for N > 1 (render on the main
stream, gather + de-interleave on the side stream) and checks every delivered frame against a monolithic render."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from fraytracer_amd import distributed as ftd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W = H = 256
STRIPE = 16
cam = syn.default_camera()
descs = [syn.config3(n=24 + 8 * i, size=W)[0] for i in range(3)]                   # three different frames
streams = [torch.cuda.Stream(), torch.cuda.Stream()]                               # two render lanes, as bench.py uses for N > 1
devs = [ft.Device(0), ft.Device(0)]
for d, st in zip(devs, streams):
    d.set_stream(st.cuda_stream)
scenes = [[d.scene(x) for x in descs] for d in devs]
want = [s.render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(W, H), cam)[0] for s in scenes[0]]
k = [0]


def lane(i):
    def render(slab):
        scenes[i][k[0] % 3].render_device(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(W, H), cam, slab.data_ptr(), **ftd.tiling(W, 1, 0, STRIPE))
        k[0] += 1
    return render


got = []
pipe = ftd.FramePipeline([lane(0), lane(1)], W, H, 1, 0, STRIPE, torch.device("cuda", 0), streams=streams, force=True,
                         on_frame=lambda i, f: got.append(f.clone()))
for _ in range(7):
    pipe.submit()
pipe.drain()
dist.barrier()
torch.cuda.synchronize()
ok = len(got) == 7 and all(np.array_equal(g.cpu().numpy().view(np.uint32), want[i % 3].view(np.uint32)) for i, g in enumerate(got))
dist.destroy_process_group()
print("PIPELINE_OK" if ok else "PIPELINE_MISMATCH")
sys.exit(0 if ok else 1)
