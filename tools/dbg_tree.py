import importlib, os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import oracle_binding as ob
name, lo, hi = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
scene = pt.Scene.named(name, 1.0)
ctx = pt.Context(device=0)
ctx.upload(scene)
rng = np.random.default_rng(99)
n = 100_000
rays = np.zeros(n, dtype=pt.RAY_DTYPE)
rays["origin"] = rng.uniform(lo, hi, size=(n, 3))
d = rng.normal(size=(n, 3))
rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.5, 2.0, size=(n, 1))
rays["time"] = rng.uniform(0, 1, size=n)
ref, _ = ob.intersect(scene.desc(), rays, medium_seed=3)
got = ctx.intersect(rays, medium_seed=3)
bad = np.nonzero(got["prim"] != ref["prim"])[0]
print("mismatches", bad.size)
same_t = np.isclose(got["t"][bad], ref["t"][bad], rtol=1e-9)
print("same t:", same_t.sum(), " gpu miss:", (got["prim"][bad] < 0).sum(), " ref miss:", (ref["prim"][bad] < 0).sum(), "gpu nearer:", (got["t"][bad] < ref["t"][bad] * (1 - 1e-9)).sum(), "gpu farther:", (got["t"][bad] > ref["t"][bad] * (1 + 1e-9)).sum())
hk = scene.desc().hitables
def kind(i): return hk[i].kind if i >= 0 else -1
print(collections.Counter((kind(a), kind(b)) for a, b in zip(ref["prim"][bad].tolist(), got["prim"][bad].tolist())).most_common(10))
for i in bad[~same_t][:6]:
    print(i, rays[i], "ref", ref["prim"][i], kind(ref["prim"][i]), ref["t"][i], "got", got["prim"][i], kind(got["prim"][i]), got["t"][i])
