#!/usr/bin/env python3
"""What reordering a bounce's rays could buy at most, measured on the extend stage alone: the 2^20 cosine-sampled
secondary rays of the million-sphere scene (the incoherent batch), traced in three orders:
  as made        order of the primary rays that spawned them (image order)
  shuffled       a random permutation (the order a pool kernel's waves see: paths at every depth, mixed)
  sorted         by (30-bit Morton cell of the origin on the scene's box, direction octant)
The sort is done on the host and NOT timed: an upper bound on what a per-bounce sort could gain, before its own cost
(a radix sort of the keys and a gather of 56-byte rays per bounce).

usage: sort_rays_experiment.py [order]    order in {all, made, shuffled, sorted}; one order per process suits a
                                          rocprofv3 --pmc pass (TCC_HIT_sum TCC_REQ_sum: L2 hit rate of that order)
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import oracle_binding as ob

which = sys.argv[1] if len(sys.argv) > 1 else "all"
W = H = 1024
scene = pt.Scene.named("sphere_cloud", 1.0, param=1000000)
ctx = pt.Context(device=0, max_paths_in_flight=1 << 20)
ctx.upload(scene)
pix = np.arange(W * H, dtype=np.uint64)
primary = ob.camera_rays(scene.camera, W, H, pix, np.zeros(W * H, dtype=np.uint32))
hits = ctx.intersect(primary, medium_seed=5)
secondary, valid = ob.secondary_rays(hits, primary)
secondary = secondary[valid]
n = len(secondary)


def spread(v):  # 10 bits -> every third bit
    v = v.astype(np.uint64) & 0x3FF
    v = (v | (v << 16)) & 0x30000FF
    v = (v | (v << 8)) & 0x300F00F
    v = (v | (v << 4)) & 0x30C30C3
    v = (v | (v << 2)) & 0x9249249
    return v


o, d = secondary["origin"], secondary["direction"]
cell = np.clip((o - o.min(axis=0)) / (o.max(axis=0) - o.min(axis=0)) * 1024.0, 0, 1023).astype(np.uint64)
morton = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << 1) | ((d[:, 2] < 0).astype(np.uint64) << 2))
orders = {
    "made": np.arange(n),
    "shuffled": np.random.default_rng(1).permutation(n),
    "sorted": np.argsort((morton << 3) | octant, kind="stable"),
    "sorted_octant_first": np.argsort((octant << 30) | morton, kind="stable"),
}
want = ctx.intersect(secondary[:65536], medium_seed=5)
for label, perm in orders.items():
    if which not in ("all", label):
        continue
    rays = secondary[perm]
    soa = np.concatenate([rays["origin"][:, k] for k in range(3)] + [rays["direction"][:, k] for k in range(3)] + [rays["time"]])
    d_soa = torch.from_numpy(np.ascontiguousarray(soa)).cuda()
    ctx.intersect_device(d_soa.data_ptr(), n, medium_seed=5, repeats=3)
    ms = ctx.intersect_device(d_soa.data_ptr(), n, medium_seed=5, repeats=20)
    # the same rays find the same primitives in any order
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    got = ctx.intersect(rays[np.sort(inv[:65536])], medium_seed=5)
    back = np.argsort(np.argsort(inv[:65536]))
    same = bool((got["prim"][back] == want["prim"]).all())
    print(json.dumps({"order": label, "rays": n, "ms": ms, "Mrays_s": n / ms / 1e3, "same_hits": same}), flush=True)
ctx.close()
