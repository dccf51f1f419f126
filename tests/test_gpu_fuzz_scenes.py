"""Parity on scenes no recipe holds: random compositions of every primitive, wrapper, material and texture the builder
API offers (src/scene.rs:13-63, src/material.rs:80-131), built from a seed, traced on the device in both modes and by
the CPU oracle. Small compositions are walked as a primitive list, large ones through the tree; both sizes are drawn.
The bars are those of the recipe scenes (tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import ptgpu

pytestmark = pytest.mark.gpu


def random_composition(seed, n_prims):
    """A box-shaped stage lit by a rectangle, filled with n_prims random objects."""
    rng = np.random.default_rng(seed)
    s = ptgpu.Scene(seed=seed)
    U = lambda lo, hi: float(rng.uniform(lo, hi))

    def texture(depth=0):
        k = rng.integers(0, 5 if depth < 2 else 1)
        if k == 0 or k == 4:
            return s.texture_constant((U(0.1, 0.9), U(0.1, 0.9), U(0.1, 0.9)))
        if k == 1:
            return s.texture_checker(texture(depth + 1), texture(depth + 1))
        if k == 2:
            return s.texture_noise(U(0.5, 8.0))
        return s.texture_image(rng.integers(0, 256, size=(int(rng.integers(2, 9)), int(rng.integers(2, 17)), 3), dtype=np.uint8))

    def material():
        k = rng.integers(0, 6)
        if k <= 2:
            return s.lambertian(texture(), emissive=U(0.5, 2.0) if rng.random() < 0.15 else 0.0)
        if k == 3:
            return s.metal((U(0.4, 1.0), U(0.4, 1.0), U(0.4, 1.0)), U(0.0, 0.6))
        if k == 4:
            return s.dielectric(U(1.2, 1.9))
        return s.diffuse_light(s.texture_constant((U(1, 6), U(1, 6), U(1, 6))))

    def shape(mat):
        """One geometry node about the origin (a wrapper moves it), or placed directly."""
        k = rng.integers(0, 5)
        if k == 0:
            return s.sphere((U(-40, 40), U(-40, 40), U(-40, 40)), U(8, 45), mat)
        if k == 1:
            c0 = np.array([U(-40, 40), U(-40, 40), U(-40, 40)])
            return s.moving_sphere(c0, c0 + rng.uniform(-20, 20, 3), 0.0, 1.0, U(8, 35), mat)
        if k == 2:
            a0, b0 = U(-60, 20), U(-60, 20)
            node = s.rect(a0, a0 + U(20, 90), b0, b0 + U(20, 90), U(-40, 40), int(rng.integers(0, 3)), mat)
            return s.flip_normals(node) if rng.random() < 0.5 else node
        p0 = np.array([U(-50, 10), U(-50, 10), U(-50, 10)])
        return s.box(p0, p0 + rng.uniform(15, 70, 3), mat)

    def placed(node):
        k = rng.integers(0, 4)
        t = (U(100, 455), U(60, 455), U(100, 455))
        if k == 0:
            return s.translate(node, t)
        if k == 1:
            return s.translate(s.rotate_y(node, U(-80, 80)), t)
        if k == 2:
            return s.translate(s.rotate_y(s.translate(node, (U(-20, 20), 0.0, U(-20, 20))), U(-180, 180)), t)
        return s.translate(s.rotate_y(s.rotate_y(node, U(-40, 40)), U(-40, 40)), t)

    white = s.lambertian(s.texture_constant((0.73, 0.73, 0.73)))
    light = s.diffuse_light(s.texture_constant((12.0, 12.0, 12.0)))
    nodes = [s.flip_normals(s.rect(180, 380, 180, 380, 554, 1, light)),  # the stage: light, floor, back wall
             s.rect(0, 555, 0, 555, 0, 1, white),
             s.flip_normals(s.rect(0, 555, 0, 555, 555, 2, s.lambertian(texture())))]
    sphere_light = None
    while len(nodes) < n_prims:
        r = rng.random()
        if r < 0.12:  # a medium inside a sphere or a box boundary, the boundary wrapped or not
            inner = s.sphere((0.0, 0.0, 0.0), U(25, 60), white) if rng.random() < 0.5 else \
                s.box((-30.0, -30.0, -30.0), (U(10, 50), U(10, 50), U(10, 50)), white)
            nodes.append(s.constant_medium(placed(inner), U(0.005, 0.05), texture(2)))
        else:
            mat = material()
            nodes.append(placed(shape(mat)))
    if rng.random() < 0.5:  # a glass ball that is also a light shape, as in the reference's hard-coded list
        c, rad = (U(150, 400), U(80, 200), U(150, 400)), U(30, 70)
        nodes.append(s.sphere(c, rad, s.dielectric(1.5)))
        sphere_light = s.sphere(c, rad, white)
    s.set_root(s.bvh(nodes, 0.0, 1.0))
    lights = [s.rect(180, 380, 180, 380, 554, 1, white)]
    if sphere_light is not None:
        lights.append(sphere_light)
    s.set_lights(lights)
    s.camera = ptgpu.make_camera((278, 278, -760), (278, 278, 0), (0, 1, 0), 40.0, 1.0, U(0.0, 8.0) if rng.random() < 0.5 else 0.0,
                                 760.0, 0.0, 1.0)
    return s


CASES = [(101, 6), (102, 11), (103, 17), (104, 23), (105, 24), (201, 26), (202, 40), (203, 90), (204, 300)] + \
        [(300 + k, n) for k, n in enumerate([4, 5, 7, 8, 9, 10, 12, 13, 14, 15, 16, 18, 19, 20, 21, 22, 30, 35, 50, 64, 128, 200, 500, 1000])]


# PT_FUZZ_EXTRA=n: n more compositions (seeds 5000 ..., sizes cycling through lists and trees), for a one-off wider run
if os.environ.get("PT_FUZZ_EXTRA"):
    CASES = CASES + [(5000 + k, [4, 9, 17, 24, 25, 40, 90, 300, 700][k % 9]) for k in range(int(os.environ["PT_FUZZ_EXTRA"]))]


@pytest.mark.parametrize("seed,n_prims", CASES)
def test_random_composition_matches_the_oracle(any_ctx, seed, n_prims):
    ctx, exact = any_ctx
    scene = random_composition(seed, n_prims)
    desc = scene.desc()
    ctx.upload(scene)
    # extend: camera rays and the cosine-sampled rays off their hit points
    w = h = 40
    pix = np.arange(w * h, dtype=np.uint64)
    primary = ob.camera_rays(scene.camera, w, h, pix, np.zeros(w * h, dtype=np.uint32))
    ohits, _ = ob.intersect(desc, primary, medium_seed=7)
    secondary, valid = ob.secondary_rays(ohits, primary)
    for batch in (primary, secondary[valid]):
        ref, _ = ob.intersect(desc, batch, medium_seed=7)
        got = ctx.intersect(batch, medium_seed=7)
        same = got["prim"] == ref["prim"]
        # a ray through an edge shared by two faces of the composition may take either in fast mode
        assert same.mean() >= (0.9999 if exact else 0.998), f"ids agree on {same.mean():.5f}"
        both = same & (ref["prim"] >= 0)
        rel = np.abs(got["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-300)
        assert rel.size == 0 or rel.max() <= (1e-11 if exact else 1e-6), f"t differs by {rel.max():.3e}"
        assert np.allclose(got["normal"][both], ref["normal"][both], rtol=1e-6, atol=1e-6)
        uv_ok = np.isclose(got["u"][both], ref["u"][both], rtol=1e-8, atol=1e-8, equal_nan=True) & \
                np.isclose(got["v"][both], ref["v"][both], rtol=1e-8, atol=1e-8, equal_nan=True)
        assert uv_ok.mean() >= 0.999
        assert (got["material"][both] == ref["material"][both]).all()
    # render, in the reference's semantics and in the book's
    for flags in (0, ptgpu.PT_RENDER_BOOK_SEMANTICS):
        ctx.render(w, h, 3, 12, flags=flags)
        got = ctx.readback()[..., :3].astype(np.float64)
        ref, _ = ob.render(desc, scene.camera, w, h, 3, 12, seed=ctx.seed, threads=8, flags=flags)
        ref = ref[..., :3].astype(np.float64)
        finite = np.isfinite(got) & np.isfinite(ref)
        nan_same = (np.isnan(got) == np.isnan(ref)).mean()
        close = np.isclose(got, ref, rtol=1e-5, atol=1e-7) | ~finite
        assert nan_same >= 0.99, f"NaN masks agree on {nan_same:.4f}"
        assert close.mean() >= (0.995 if exact else 0.98), f"{close.mean():.4f} of channels agree"


@pytest.mark.parametrize("seed,n_prims", [(102, 11), (105, 24), (202, 40), (320, 128)])
def test_random_composition_realtime_frames_tiles_and_rows(any_ctx, seed, n_prims):
    """On the same compositions: progressive realtime frames (80 % pixel skip, running blend, src/trace.rs:144-177)
    against the oracle; a tile of the image and the interleaved rows of a group of three against the whole render."""
    ctx, exact = any_ctx
    scene = random_composition(seed, n_prims)
    desc = scene.desc()
    ctx.upload(scene)
    w, h = 48, 36
    ctx.reset()
    ref = np.zeros((h, w, 4), dtype=np.float32)
    counts = np.zeros((h, w), dtype=np.uint32)
    for frame in range(5):
        ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=frame)
        ob.render(desc, scene.camera, w, h, 1, 10, seed=ctx.seed, mode=1, frame=frame, threads=4, image=ref, counts=counts)
    got = ctx.readback()
    touched = counts > 0
    assert 0.45 < touched.mean() < 0.9  # 1 - 0.8^5 = 0.67
    assert (got[~touched][..., :3] == 0).all()
    close = np.isclose(got[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6, equal_nan=True)
    assert close.mean() >= (0.995 if exact else 0.98)
    # offline: parts against the whole
    ctx.reset()
    ctx.render(w, h, 3, 12)
    whole = ctx.readback()
    x0, y0, x1, y1 = 5, 3, 41, 30
    for phase in range(3):
        ctx.reset()
        ctx.render(w, h, 3, 12, tile=(x0, y0, x1, y1), rows=(3, phase))
        part = ctx.readback()
        mask = np.zeros((h, w), dtype=bool)
        mask[y0 + phase:y1:3, x0:x1] = True
        assert np.array_equal(part[mask], whole[mask], equal_nan=True)
        assert not np.nan_to_num(part[~mask]).any()


@pytest.mark.parametrize("seed,n_prims", [(103, 17), (203, 90)])
def test_random_composition_pass_forms_agree(seed, n_prims):
    """The one-launch pool forms and a launch per stage and bounce trace the same paths on compositions too."""
    scene = random_composition(seed, n_prims)
    images = []
    for tuning in ({}, {"pass_form": ptgpu.PT_PASS_PER_BOUNCE}, {"pool_chunk": 64, "pool_refill": 1, "tree_refill": 1}):
        ctx = ptgpu.Context(device=0, max_paths_in_flight=1 << 18, **tuning)
        ctx.upload(scene)
        ctx.render(96, 64, 4, 20)
        images.append(ctx.readback())
        ctx.close()
    assert np.array_equal(images[0], images[1], equal_nan=True)
    assert np.array_equal(images[0], images[2], equal_nan=True)


@pytest.mark.parametrize("with_medium", [False, True])
def test_wrapped_subtree_matches_the_oracle(any_ctx, with_medium):
    """A BVH under translate(rotate_y(...)), the construction around the sphere cluster of the final scene
    (src/lib.rs:1063-1075), in a scene small enough for every form: rays and a render against the oracle."""
    from test_gpu_configs import _wrapped_subtree_scene

    ctx, exact = any_ctx
    scene = _wrapped_subtree_scene(with_medium)
    desc = scene.desc()
    ctx.upload(scene)
    w = h = 48
    pix = np.arange(w * h, dtype=np.uint64)
    primary = ob.camera_rays(scene.camera, w, h, pix, np.zeros(w * h, dtype=np.uint32))
    ohits, _ = ob.intersect(desc, primary, medium_seed=3)
    secondary, valid = ob.secondary_rays(ohits, primary)
    for batch in (primary, secondary[valid]):
        ref, _ = ob.intersect(desc, batch, medium_seed=3)
        got = ctx.intersect(batch, medium_seed=3)
        same = got["prim"] == ref["prim"]
        assert same.mean() >= 0.9995
        both = same & (ref["prim"] >= 0)
        rel = np.abs(got["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-300)
        assert rel.max() <= (1e-11 if exact else 1e-7)
    ctx.render(w, h, 4, 20)
    got = ctx.readback()[..., :3].astype(np.float64)
    ref, _ = ob.render(desc, scene.camera, w, h, 4, 20, seed=ctx.seed, threads=8)
    ref = ref[..., :3].astype(np.float64)
    finite = np.isfinite(got) & np.isfinite(ref)
    assert (np.isnan(got) == np.isnan(ref)).mean() >= 0.99
    assert (np.isclose(got, ref, rtol=1e-5, atol=1e-7) | ~finite).mean() >= (0.995 if exact else 0.98)


def nested_composition(seed, boundary):
    """Compositions the builder API allows beyond the recipes (src/hitable.rs:40-78, src/volume.rs:30-31): HitableLists
    inside lists and under wrappers, and (boundary = "list" / "tree" / "wrapped_tree") a medium whose boundary is a list of
    two shapes, a BvhNode of spheres, or a BvhNode under wrappers instead of one primitive."""
    rng = np.random.default_rng(seed)
    s = ptgpu.Scene(seed=seed)
    U = lambda lo, hi: float(rng.uniform(lo, hi))
    white = s.lambertian(s.texture_constant((0.73, 0.73, 0.73)))
    light = s.diffuse_light(s.texture_constant((12.0, 12.0, 12.0)))

    def mat():
        k = rng.integers(0, 4)
        if k <= 1:
            return s.lambertian(s.texture_constant((U(0.1, 0.9), U(0.1, 0.9), U(0.1, 0.9))))
        if k == 2:
            return s.metal((U(0.4, 1.0), U(0.4, 1.0), U(0.4, 1.0)), U(0.0, 0.4))
        return s.dielectric(1.5)

    def pair():  # two shapes about the origin, as a list
        a = s.sphere((U(-60, 0), U(-40, 40), U(-40, 40)), U(30, 60), mat())
        p0 = np.array([U(0, 40), U(-60, 0), U(-60, 0)])
        b = s.box(p0, p0 + rng.uniform(40, 100, 3), mat())
        return s.list([a, s.rotate_y(b, U(-40, 40))])

    stage = s.list([s.flip_normals(s.rect(180, 380, 180, 380, 554, 1, light)), s.rect(0, 555, 0, 555, 0, 1, white),
                    s.flip_normals(s.rect(0, 555, 0, 555, 555, 2, white))])
    top = [stage]
    # lists under wrappers, one of them nested once more (RotateY::new takes its child's bounding box, which a list does not
    # have, src/hitable.rs:63-65,148: a list can stand under Translate and FlipNormals only; its members may be rotated)
    for _ in range(3):
        top.append(s.translate(pair(), (U(120, 430), U(60, 400), U(120, 430))))
    top.append(s.translate(s.list([pair(), s.flip_normals(s.translate(pair(), (70.0, 0.0, 10.0)))]), (U(150, 400), U(100, 300), U(150, 400))))
    if boundary == "list":
        shell = s.list([s.sphere((-30.0, 0.0, 0.0), 90.0, white), s.rotate_y(s.box((0.0, -60.0, -60.0), (140.0, 60.0, 60.0), white), 25.0)])
        top.append(s.constant_medium(s.translate(shell, (280.0, 250.0, 300.0)), 0.02, s.texture_constant((0.9, 0.9, 0.9))))
    elif boundary == "tree":
        balls = [s.translate(s.sphere((U(-80, 80), U(-80, 80), U(-80, 80)), U(50, 90), white), (280.0, 250.0, 300.0)) for _ in range(5)]
        top.append(s.constant_medium(s.bvh(balls, 0.0, 1.0), 0.03, s.texture_constant((0.2, 0.4, 0.9))))
    elif boundary == "wrapped_tree":  # a BvhNode under Translate and RotateY, with another wrapped BvhNode among its members
        balls = [s.sphere((U(-80, 80), U(-80, 80), U(-80, 80)), U(50, 90), white) for _ in range(5)]
        inner = s.translate(s.rotate_y(s.bvh(balls[3:], 0.0, 1.0), U(-40, 40)), (U(-30, 30), 0.0, U(-30, 30)))
        tree = s.bvh(balls[:3] + [inner], 0.0, 1.0)
        top.append(s.constant_medium(s.translate(s.rotate_y(tree, U(-40, 40)), (280.0, 250.0, 300.0)), 0.03, s.texture_constant((0.2, 0.4, 0.9))))
    s.set_root(s.list(top))
    s.set_lights([s.rect(180, 380, 180, 380, 554, 1, white)])
    s.camera = ptgpu.make_camera((278, 278, -760), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 760.0, 0.0, 1.0)
    return s


@pytest.mark.parametrize("seed,boundary", [(401, None), (402, None), (403, "list"), (404, "list"), (405, "tree"), (406, "tree"),
                                           (407, "wrapped_tree"), (408, "wrapped_tree")])
def test_nested_lists_and_general_medium_boundaries(any_ctx, seed, boundary):
    ctx, exact = any_ctx
    scene = nested_composition(seed, boundary)
    desc = scene.desc()
    ctx.upload(scene)
    w = h = 48
    pix = np.arange(w * h, dtype=np.uint64)
    primary = ob.camera_rays(scene.camera, w, h, pix, np.zeros(w * h, dtype=np.uint32))
    ohits, _ = ob.intersect(desc, primary, medium_seed=9)
    secondary, valid = ob.secondary_rays(ohits, primary)
    for batch in (primary, secondary[valid]):
        ref, _ = ob.intersect(desc, batch, medium_seed=9)
        got = ctx.intersect(batch, medium_seed=9)
        same = got["prim"] == ref["prim"]
        assert same.mean() >= (0.9999 if exact else 0.998), f"ids agree on {same.mean():.5f}"
        both = same & (ref["prim"] >= 0)
        rel = np.abs(got["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-300)
        assert rel.size == 0 or rel.max() <= (1e-11 if exact else 1e-6), f"t differs by {rel.max():.3e}"
        assert np.allclose(got["normal"][both], ref["normal"][both], rtol=1e-6, atol=1e-6)
    if boundary is not None:  # the medium is hit by some of the rays, or the case tests nothing
        medium_node = max(int(p) for p in np.unique(ohits["prim"]))
        assert (ohits["normal"][ohits["prim"] == medium_node] == (1.0, 0.0, 0.0)).all() and (ohits["prim"] == medium_node).sum() > 20, (ohits["prim"] == medium_node).sum()
    for flags in (0, ptgpu.PT_RENDER_BOOK_SEMANTICS):
        ctx.render(w, h, 3, 12, flags=flags)
        got = ctx.readback()[..., :3].astype(np.float64)
        ref, _ = ob.render(desc, scene.camera, w, h, 3, 12, seed=ctx.seed, threads=8, flags=flags)
        ref = ref[..., :3].astype(np.float64)
        finite = np.isfinite(got) & np.isfinite(ref)
        assert (np.isnan(got) == np.isnan(ref)).mean() >= 0.99
        assert (np.isclose(got, ref, rtol=1e-5, atol=1e-7) | ~finite).mean() >= (0.995 if exact else 0.98)
    # realtime frames take the same route
    ctx.reset()
    ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=0)
    assert np.isfinite(np.nan_to_num(ctx.readback())).all()
