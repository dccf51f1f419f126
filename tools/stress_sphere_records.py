#!/usr/bin/env python3
"""Random scenes of plain spheres (sizes from a handful to thousands, radii over six decades, scenes far from the origin,
with and without a ground sphere that spans the scene, with and without checker colours) traced through the compact
records over 32-byte nodes with f32 slab tests, against the general records over 64-byte nodes with f64 tests. Boxes
only cull, so the two must give the same image bit for bit, offline and in realtime frames, also after a rebuild of the
tree on the device. usage: tools/stress_sphere_records.py [n_scenes] [first_seed]   (needs an MI355X)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")


def scene_of(seed):
    rng = np.random.default_rng(seed)
    s = pt.Scene()
    n = int(rng.choice([3, 7, 30, 200, 1500, 6000]))
    scale = float(10.0 ** rng.uniform(-2.0, 3.0))           # size of the cloud
    offset = rng.uniform(-1.0, 1.0, 3) * float(rng.choice([0.0, 1.0, 1.0e3, 1.0e5])) * scale
    checker = s.texture_checker(s.texture_constant((0.2, 0.3, 0.1)), s.texture_constant((0.9, 0.9, 0.9)))
    nodes = []
    if rng.random() < 0.6:  # a ground sphere that spans the scene
        ground_r = scale * float(rng.choice([50.0, 1000.0]))
        tex = checker if rng.random() < 0.5 else s.texture_constant((0.5, 0.5, 0.5))
        nodes.append(s.sphere(offset + np.array([0.0, -ground_r - 0.5 * scale, 0.0]), ground_r, s.lambertian(tex, 0.0)))
    for k in range(n):
        c = offset + rng.uniform(-0.5, 0.5, 3) * scale
        r = scale * float(10.0 ** rng.uniform(-3.5, -0.7))
        pick = rng.integers(0, 6)
        if pick == 0:
            m = s.lambertian(s.texture_constant(rng.uniform(0.1, 0.9, 3)), 0.0)
        elif pick == 1:
            m = s.lambertian(s.texture_constant(rng.uniform(0.1, 0.9, 3)), 8.0)
        elif pick == 2:
            m = s.metal(rng.uniform(0.5, 1.0, 3), float(rng.uniform(0.0, 0.3)))
        elif pick == 3:
            m = s.dielectric(1.5)
        elif pick == 4:
            m = s.diffuse_light(s.texture_constant(rng.uniform(1.0, 4.0, 3)))
        else:
            m = s.lambertian(checker, 0.0)
        nodes.append(s.sphere(c, r, m))
    s.set_root(s.bvh(nodes))
    dummy = s.lambertian(s.texture_constant((0, 0, 0)), 0.0)
    s.set_lights([s.sphere(offset + np.array([0.0, 0.7 * scale, 0.0]), 0.05 * scale, dummy)])
    eye = offset + np.array([0.0, 0.1, -1.6]) * scale
    s.camera = pt.make_camera(tuple(eye), tuple(offset), vfov=40.0, aspect=1.5)
    return s, n


def images(scene, flags):
    ctx = pt.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 18, flags=flags)
    ctx.upload(scene)
    ctx.render(96, 64, 3, 12)
    out = [ctx.readback().copy()]
    records = ctx.last_pass_records() if int(ctx.stats().last_pass_form) == pt.PT_PASS_TREE_POOL else -1
    ctx.reset()
    for frame in range(2):
        ctx.render(96, 64, 1, 12, mode=pt.PT_MODE_REALTIME, frame=frame)
    out.append(ctx.readback().copy())
    if records >= 0:
        ctx.build_tree_on_device(0.0, 1.0)
        ctx.render(96, 64, 3, 12)
        out.append(ctx.readback().copy())
    ctx.close()
    return out, records


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = compact = 0
    for seed in range(first, first + n_scenes):
        scene, n = scene_of(seed)
        a, ra = images(scene, 0)
        b, rb = images(scene, pt.PT_CFG_FAT_RECORDS)
        compact += ra == (pt.PT_RECORDS_COMPACT_SPHERES | pt.PT_RECORDS_NODES32)
        same = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))
        if not same:
            bad += 1
            diff = [float(np.mean(x.view(np.uint32) != y.view(np.uint32))) for x, y in zip(a, b)]
            print(f"seed {seed}: {n} spheres, records {ra} / {rb}: images differ in {diff} of their words", flush=True)
        elif seed % 10 == 0:
            lit = float(np.mean(np.nan_to_num(a[0][..., :3]).sum(axis=-1) > 0.0))
            print(f"seed {seed}: {n} spheres, records {ra} / {rb}: identical ({lit:.2f} of the pixels lit, "
                  f"{float(np.mean(np.isnan(a[0][..., 0]))):.2f} NaN)", flush=True)
    print(f"{n_scenes} scenes, {compact} through the compact records over 32-byte nodes, {bad} with differing images")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
