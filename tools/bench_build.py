#!/usr/bin/env python3
"""Tree build on the device against the host-built SAH tree: build time and the speed of what it builds, on the
one-million-sphere scene (extend batch of 2^20 camera rays + their secondary rays, and a 1920x1080 render).

usage: bench_build.py [n_spheres]
"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
pt.register_earthmap()  # texels of the image texture (tests/golden/make_earthmap.py)
import oracle_binding as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
scene = pt.Scene.named("sphere_cloud", 16 / 9, param=n)
ctx = pt.Context(device=0, max_paths_in_flight=1 << 24)
t0 = time.perf_counter()
ctx.upload(scene)
upload_s = time.perf_counter() - t0
W = H = 1024
primary = ob.camera_rays(scene.camera, W, H, np.arange(W * H, dtype=np.uint64), np.zeros(W * H, dtype=np.uint32))
hits = ctx.intersect(primary)
secondary, valid = ob.secondary_rays(hits, primary)  # a ray generator only: cosine samples about the hit normals
secondary = secondary[valid]


def measure(label):
    out = {"tree": label}
    for name, rays in (("primary", primary), ("secondary", secondary)):
        soa = np.concatenate([rays["origin"][:, k] for k in range(3)] + [rays["direction"][:, k] for k in range(3)] + [rays["time"]])
        d = torch.from_numpy(np.ascontiguousarray(soa)).cuda()
        ctx.intersect_device(d.data_ptr(), len(rays), repeats=3)
        ms = ctx.intersect_device(d.data_ptr(), len(rays), repeats=20)
        out[name + "_Mrays_s"] = len(rays) / ms / 1e3
    ctx.render(1920, 1080, 8, 10, frame=0)
    t = time.perf_counter()
    ctx.render(1920, 1080, 8, 10, frame=1)
    out["render_Mpaths_s"] = 1920 * 1080 * 8 / (time.perf_counter() - t) / 1e6
    return out


a = measure("host SAH")
a["upload_incl_build_s"] = upload_s
base = ctx.intersect(secondary)
ctx.build_tree_on_device(0.0, 1.0)  # warm-up (allocations, first launches)
t0 = time.perf_counter()
ctx.build_tree_on_device(0.0, 1.0)
build_ms = (time.perf_counter() - t0) * 1e3
b = measure("device Morton")
b["build_ms"] = build_ms
got = ctx.intersect(secondary)
b["ids_equal_to_host_tree"] = float((got["prim"] == base["prim"]).mean())
b["t_equal_to_host_tree"] = bool(np.array_equal(got["t"], base["t"], equal_nan=True))
t0 = time.perf_counter()
ctx.refit(0.0, 1.0)
b["refit_ms"] = (time.perf_counter() - t0) * 1e3
print(json.dumps({"n_spheres": n, "host": a, "device": b}))
ctx.close()
