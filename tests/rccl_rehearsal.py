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
dev = ft.Device(0)
dev.set_stream(torch.cuda.current_stream().cuda_stream)
cam = syn.default_camera()
scenes = [dev.scene(syn.config3(n=24 + 8 * i, size=W)[0]) for i in range(3)]      # three different frames
want = [s.render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(W, H), cam)[0] for s in scenes]
k = [0]


def render(slab):
    scenes[k[0] % 3].render_device(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(W, H), cam, slab.data_ptr(), **ftd.tiling(W, 1, 0, STRIPE))
    k[0] += 1


got = []
pipe = ftd.FramePipeline(render, W, H, 1, 0, STRIPE, torch.device("cuda", 0), force=True,
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
