import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
sys.modules["ptgpu"] = pt
import oracle_binding as ob
import test_gpu_parity as T
scene = T._kitchen_sink_scene()
ctx = pt.Context(device=0)
ctx.upload(scene)
rng = np.random.default_rng(123)
n = 200_000
rays = np.zeros(n, dtype=pt.RAY_DTYPE)
rays["origin"] = rng.uniform(-20, 120, size=(n, 3))
d = rng.normal(size=(n, 3))
rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 5.0, size=(n, 1))
rays["time"] = rng.uniform(0, 1, size=n)
k = n // 10
rays["direction"][:k, 1] = 0.0
ref, _ = ob.intersect(scene.desc(), rays)
got = ctx.intersect(rays)
bad = np.nonzero(got["prim"] != ref["prim"])[0]
print("mismatches", bad.size, "of which in first tenth", (bad < k).sum())
import collections
print(collections.Counter(zip(ref["prim"][bad].tolist(), got["prim"][bad].tolist())).most_common(20))
for i in bad[:8]:
    print(i, rays[i], "ref", ref["prim"][i], ref["t"][i], ref["normal"][i], "got", got["prim"][i], got["t"][i], got["normal"][i])
