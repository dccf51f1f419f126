#!/usr/bin/env python3
"""Realtime 1080p frame rate of a scene with and without per-stage profiling: tools/rt_rate.py [scene] [frames].
(PT_CFG_PROFILE_STAGES brackets every n-th frame with events and reads its counters back, one host wait per such frame.)"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
W, H = 1920, 1080
scene = pt.Scene.named(name, W / H, seed=0x5EED0001)
for label, flags, tuning, rflags in (("profiled every 8th frame", pt.PT_CFG_PROFILE_STAGES, {}, 0), ("no profiling", 0, {}, 0),
                                     ("no profiling, frames issued without waiting (PT_RENDER_NO_WAIT)", 0, {}, pt.PT_RENDER_NO_WAIT),
                                     ("profiled every 8th frame, issued without waiting", pt.PT_CFG_PROFILE_STAGES, {}, pt.PT_RENDER_NO_WAIT)):
    ctx = pt.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 22, flags=flags, **tuning)
    ctx.upload(scene)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    ctx.bind_image(image.data_ptr(), W, H)
    for k in range(200):
        ctx.render(W, H, 1, 10, mode=pt.PT_MODE_REALTIME, frame=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(frames):
        ctx.render(W, H, 1, 10, mode=pt.PT_MODE_REALTIME, frame=200 + k, flags=rflags)
    ctx.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name} {label}: {frames / dt:.0f} frames/s, {dt / frames * 1e6:.1f} us per frame")
    ctx.close()
