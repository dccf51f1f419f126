#!/usr/bin/env python3
"""Per-stripe render time of the headline workload on one GPU: how even is the 64-row round-robin split?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")
pt.register_earthmap()  # texels of the image texture (tests/golden/make_earthmap.py)
W = H = 1024
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene = pt.Scene.named("cornell_box", 1.0)
ctx = pt.Context(device=0, max_paths_in_flight=1 << 26)
ctx.upload(scene)
ctx.render(W, H, 4, 50)
ts = []
for y0 in range(0, H, rows):
    ctx.render(W, H, spp, 50, tile=(0, y0, W, y0 + rows))  # warm
    t0 = time.perf_counter()
    ctx.render(W, H, spp, 50, tile=(0, y0, W, y0 + rows))
    ts.append((time.perf_counter() - t0) * 1e3)
print("stripe ms:", " ".join(f"{t:.2f}" for t in ts))
for n in (2, 4, 8):
    per = [sum(ts[k] for k in range(r, len(ts), n)) for r in range(n)]
    print(f"N={n}: max/mean = {max(per) / (sum(per) / n):.3f}")
t0 = time.perf_counter(); ctx.render(W, H, spp, 50); print("whole image ms:", round((time.perf_counter() - t0) * 1e3, 2), "sum of stripes:", round(sum(ts), 2))
