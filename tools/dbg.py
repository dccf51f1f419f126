import sys, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
pt = importlib.import_module("path-tracer_amd")
import oracle_binding as ob, numpy as np
s = pt.Scene.named("cornell_box", 1.0)
ctx = pt.Context(max_paths_in_flight=1<<16)
ctx.upload(s)
rays = np.zeros(3, dtype=pt.RAY_DTYPE)
rays["origin"][:] = [278,278,-800]
rays["direction"][0] = [0,0.3,1]   # ceiling
rays["direction"][1] = [0.3,0,1]   # left wall (x=555, flipped)
rays["direction"][2] = [0.05,0.0,1]   # box / back wall
ref,_ = ob.intersect(s.desc(), rays)
got = ctx.intersect(rays)
for i in range(3): print("ref", ref[i]); print("got", got[i])
