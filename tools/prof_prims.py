#!/usr/bin/env python3
"""Profiler workload with a synthetic scene of N copies of one primitive kind, to measure the per-primitive cost of
the list walk: prof_prims.py <rect|sphere|box|xbox> <n> [spp]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")

kind, n = sys.argv[1], int(sys.argv[2])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
s = pt.Scene()
white = s.lambertian(s.texture_constant((0.7, 0.7, 0.7)))
light = s.diffuse_light(s.texture_constant((15, 15, 15)))
nodes = []
lrect = s.flip_normals(s.rect(213, 343, 227, 332, 554, 1, light))
nodes.append(lrect)
for k in range(n):
    z = 100.0 + 400.0 * k / max(n, 1)
    if kind == "rect":
        nodes.append(s.rect(0, 555, 0, 555, z, 2, white))
    elif kind == "sphere":
        nodes.append(s.sphere((278, 278, z), 60.0, white))
    elif kind == "box":
        nodes.append(s.box((100, 100, z), (300, 300, z + 10), white))
    elif kind == "xbox":
        nodes.append(s.translate(s.rotate_y(s.box((0, 0, 0), (165, 330, 10)), 15.0), (100, 0, z)) if False else
                     s.translate(s.rotate_y(s.box((0, 0, 0), (165, 330, 10), white), 15.0), (100, 0, z)))
s.set_root(s.bvh(nodes))
s.set_lights([s.rect(213, 343, 227, 332, 554, 1, white)])
s.camera = pt.make_camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
ctx = pt.Context(device=0, max_paths_in_flight=1 << 24)
ctx.upload(s)
t0 = time.perf_counter()
ctx.render(1024, 1024, spp, 50)
dt = time.perf_counter() - t0
st = ctx.stats()
print(f"{kind} x {n}: {dt * 1e3:.1f} ms, {st.segments / max(st.paths, 1):.3f} segments/path, flatten {s.flatten_info()}")
ctx.close()
