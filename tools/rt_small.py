#!/usr/bin/env python3
"""Realtime frame rate of small frames through the pool and through the queue form: tools/rt_small.py [scene]."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
for W, H in ((320, 180), (480, 270), (640, 360), (800, 450), (1280, 720)):
    scene = pt.Scene.named(name, W / H, seed=0x5EED0001)
    line = f"{name} {W}x{H}:"
    for label, form in (("default", pt.PT_PASS_NONE), ("queue form", pt.PT_PASS_PERSISTENT)):
        ctx = pt.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 22, pass_form=form)
        ctx.upload(scene)
        for k in range(200):
            ctx.render(W, H, 1, 10, mode=pt.PT_MODE_REALTIME, frame=k, flags=pt.PT_RENDER_NO_WAIT)
        ctx.wait()
        t0 = time.perf_counter()
        frames = 3000
        for k in range(frames):
            ctx.render(W, H, 1, 10, mode=pt.PT_MODE_REALTIME, frame=200 + k, flags=pt.PT_RENDER_NO_WAIT)
        ctx.wait()
        dt = time.perf_counter() - t0
        line += f" {label} {frames / dt:.0f} frames/s"
        ctx.close()
    print(line)
