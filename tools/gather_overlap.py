#!/usr/bin/env python3
"""What the asynchronous gather hides, as far as one GPU can show it: realtime frames of the Cornell box (1920 x 1080, one
sample, depth 10) traced (a) alone, (b) with the synchronous loopback gather after every frame (odd rows packed, sent from
the context to itself through RCCL, scattered, host synchronisation), (c) with pt_group_loopback_async after every frame
(the same rows through RCCL on the second stream while the next frame is traced; both halves assembled into a frame)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  one copy of the HIP runtime and RCCL in the process

pt = importlib.import_module("path-tracer_amd")
W, H, FRAMES = 1920, 1080, 400
scene = pt.Scene.named("cornell_box", W / H)
out = {}
for form in ("alone", "synchronous loopback", "asynchronous loopback"):
    ctx = pt.Context(device=0, max_paths_in_flight=1 << 22)
    ctx.upload(scene)
    ctx.group_join(pt.group_unique_id(), 0, 1)
    after = {"alone": lambda: None, "synchronous loopback": ctx.group_loopback, "asynchronous loopback": ctx.group_loopback_async}[form]
    for k in range(20):
        ctx.render(W, H, 1, 10, mode=pt.PT_MODE_REALTIME, frame=k)
        after()
    ctx.group_wait()
    t0 = time.perf_counter()
    for k in range(FRAMES):
        ctx.render(W, H, 1, 10, mode=pt.PT_MODE_REALTIME, frame=20 + k)
        after()
    ctx.group_wait()
    ctx.readback()
    dt = time.perf_counter() - t0
    out[form] = {"frames_per_s": FRAMES / dt, "ms_per_frame": dt / FRAMES * 1e3}
    ctx.group_leave()
    ctx.close()
print(json.dumps({"workload": f"cornell_box {W}x{H} realtime frames, {FRAMES} frames", **out}))
