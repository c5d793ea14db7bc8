#!/usr/bin/env python3
"""Divergence of the grid-union loop (eval_union): loop trips and candidate evaluations per lane against what the
wave executes.  Needs the diagnostic build:  make -C fraytracer_amd/csrc profile
    FRAYTRACER_HIP_LIB=fraytracer_amd/libfraytracer_hip_profile.so python tools/union_divergence.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FRAYTRACER_HIP_LIB", os.path.join(ROOT, "fraytracer_amd", "libfraytracer_hip_profile.so"))
import fraytracer_amd as ft
from fraytracer_amd import synthetic as syn
from fraytracer_amd import _lib

dbg = _lib.lib.ft_debug_union_counters
dbg.argtypes = [C.c_void_p]
dev = ft.Device(0)
cam = syn.default_camera()
out = (C.c_uint64 * 12)()
for name, scene, n, kw in (("console 1000 tori", syn.console_scene()[0], 2000, {}), ("C2 union32", syn.config2()[0], 2048, {}),
                           ("C5 glass", syn.config5()[0], 1024, dict(spp=4, spectral=4, max_bounces=4))):
    ds = dev.scene(scene)
    dbg(out)
    img, st = ds.render(syn.EPSILON, syn.RAY_LENGTH, ft.ImageSize(n, n), cam, **kw)
    dbg(out)
    lane_trips, wave_trips, lane_evals, wave_evals, cyc_walk, cyc_eval, n_wave_evals, cyc_round, cyc_load, cyc_prim, uni_cell, same_cell = (int(v) for v in out)
    n_wave_evals //= 64
    ev = st["sdf_evals"]
    print(json.dumps({"scene": name, "size": n, "sdf_evals": ev,
                      "loop_trips_per_eval_lane": round(lane_trips / ev, 2), "loop_trips_per_eval_wave": round(wave_trips / ev, 2),
                      "candidate_evals_per_eval_lane": round(lane_evals / ev, 2), "candidate_eval_blocks_per_eval_wave": round(wave_evals / ev, 2),
                      "shader_cycles_per_wave_eval": round(cyc_eval / max(1, n_wave_evals)), "of_which_union_walk": round(cyc_walk / max(1, n_wave_evals)),
                      "walk_cycles_waiting_for_records": round(cyc_load / max(1, n_wave_evals)), "walk_cycles_in_candidate_evaluations": round(cyc_prim / max(1, n_wave_evals)),
                      "shader_cycles_per_round_incl_state_machine": round(cyc_round / max(1, n_wave_evals)), "kernel_ms": round(st["kernel_ms"], 2),
                      "wave_evals_in_one_cell": round(uni_cell / 64 / max(1, n_wave_evals), 3), "and_in_the_cell_of_the_evaluation_before": round(same_cell / 64 / max(1, n_wave_evals), 3),
                      "trip_efficiency": round(lane_trips / max(1, wave_trips), 3), "eval_efficiency": round(lane_evals / max(1, wave_evals), 3)}), flush=True)
