#!/usr/bin/env python3
"""Small, torch-free workload for profiler passes: renders one image through the C ABI.

usage: prof_driver.py [scene] [width] [height] [spp] [max_depth] [paths_in_flight_log2] [scene param] [book]
       (book = 1: PT_RENDER_BOOK_SEMANTICS)
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")
pt.register_earthmap()  # texels of the image texture (tests/golden/make_earthmap.py)

scene_name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 8
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 50
pif = int(sys.argv[6]) if len(sys.argv) > 6 else 22
param = int(sys.argv[7]) if len(sys.argv) > 7 else 0
book = len(sys.argv) > 8 and sys.argv[8] == "1"
flags = int(os.environ.get("PTGPU_FLAGS", "0"))

scene = pt.Scene.named(scene_name, w / h, param=param)
ctx = pt.Context(device=0, max_paths_in_flight=1 << pif, flags=flags)
ctx.upload(scene)
t0 = time.perf_counter()
ctx.render(w, h, spp, depth, flags=pt.PT_RENDER_BOOK_SEMANTICS if book else 0)
dt = time.perf_counter() - t0
st = ctx.stats()
print(f"{scene_name} {w}x{h}@{spp} depth {depth}: {dt * 1e3:.1f} ms, {w * h * spp / dt / 1e6:.1f} Mpaths/s, "
      f"{st.segments / max(st.paths, 1):.3f} segments/path segments={int(st.segments)} paths={int(st.paths)}")
ctx.close()
