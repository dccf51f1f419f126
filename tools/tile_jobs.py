#!/usr/bin/env python3
"""Cost of keeping the reference's tile jobs (src/lib.rs:380-394: the image cut into 10 x 10 tiles, one
TraceSceneBatchJob each): one pt_render per job, the batch through pt_render_jobs, and one job for the whole image.
usage: tile_jobs.py [scene]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")
pt.register_earthmap()
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"


def tiles(w, h, n=10):
    tw, th = (w + n - 1) // n, (h + n - 1) // n
    return [(x, y, min(x + tw, w), min(y + th, h)) for y in range(0, h, th) for x in range(0, w, tw)]


for label, w, h, spp, depth, mode, reps in (("offline 1024x1024 @ 256 spp", 1024, 1024, 256, 50, pt.PT_MODE_OFFLINE, 2),
                                            ("realtime 1920x1080 frame", 1920, 1080, 1, 10, pt.PT_MODE_REALTIME, 50)):
    scene = pt.Scene.named(name, w / h)
    ctx = pt.Context(device=0, max_paths_in_flight=1 << 28)
    ctx.upload(scene)
    out = {}
    for how, jobs in (("one job", [None]), ("100 tile jobs", tiles(w, h)), ("batch", tiles(w, h))):
        for rep in range(reps + 1):
            if rep == 1:
                t0 = time.perf_counter()
            if how == "batch":
                ctx.render_jobs(w, h, spp, depth, jobs, mode=mode, frame=rep)
            else:
                for t in jobs:
                    ctx.render(w, h, spp, depth, tile=t, mode=mode, frame=rep)
        out[how] = (time.perf_counter() - t0) / reps
    print(f"{name} {label}: one job {out['one job'] * 1e3:.2f} ms, 100 tile jobs one by one {out['100 tile jobs'] * 1e3:.2f} ms "
          f"({out['100 tile jobs'] / out['one job']:.2f}x), as one batch {out['batch'] * 1e3:.2f} ms")
    ctx.close()
