"""Oracle parity of the pass forms and arithmetic that are BENCHMARKED: fast mode (f64 with refined reciprocals and
FMA), the library's default pass form (path pool for list scenes, tree pool for tree scenes), at the benchmark
configurations' image sizes or the largest the CPU oracle finishes in seconds, same seed on both sides.

The device and the oracle draw the same random numbers per path, so a pixel only differs beyond rounding when a few
ulp flip one of its paths' decisions (a hit / miss at a grazing angle, a sampler choice at a draw next to 0.5). The
bars below therefore have two parts: nearly every channel agrees to 1e-6 relative, and the image as a whole agrees
to a stated relative RMSE. Observed values are in the comments beside each bar.
"""
import numpy as np
import pytest

import oracle_binding as ob
import ptgpu

pytestmark = pytest.mark.gpu


def _compare(got, ref):
    """(fraction of channels within 1e-6 relative, relative RMSE over finite channels, NaN-mask agreement)."""
    got = got[..., :3].astype(np.float64)
    ref = ref[..., :3].astype(np.float64)
    nan_same = float((np.isnan(got) == np.isnan(ref)).mean())
    fin = np.isfinite(got) & np.isfinite(ref)
    close = np.isclose(got, ref, rtol=1e-6, atol=1e-9) | ~fin
    rmse = np.sqrt(np.mean((got[fin] - ref[fin]) ** 2))
    scale = max(np.sqrt(np.mean(ref[fin] ** 2)), 1e-300)
    return float(close.mean()), float(rmse / scale), nan_same


def _render_both(scene, w, h, spp, depth, flags=0, budget=1 << 22, rflags=0, threads=16):
    ctx = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=budget, flags=flags)
    ctx.upload(scene)
    ctx.render(w, h, spp, depth, flags=rflags)
    got, st = ctx.readback().copy(), ctx.stats()
    ctx.close()
    ref, ost = ob.render(scene.desc(), scene.camera, w, h, spp, depth, seed=0x5EED0001, threads=threads, flags=rflags)
    return got, ref, st, ost


def test_c1_rtiow_full_size_fast_tree_pool():
    """configs[0]: RTiOW random spheres, 1200 x 800, depth 50, one sample per pixel, fast mode through k_tree_pool."""
    scene = ptgpu.Scene.named("random_scene", 1200 / 800)
    got, ref, st, ost = _render_both(scene, 1200, 800, 1, 50)
    assert int(st.last_pass_form) == ptgpu.PT_PASS_TREE_POOL
    close, rel_rmse, nan_same = _compare(got, ref)
    print(f"C1 fast: close {close:.6f} rel_rmse {rel_rmse:.3e} nan_same {nan_same:.6f}")
    assert nan_same == 1.0
    assert close >= 0.9999     # observed 1.0
    assert rel_rmse <= 1e-6    # observed 6.3e-13 (one sample per pixel: a single flipped path would move a whole pixel)
    assert int(st.segments) == pytest.approx(ost.segments, rel=0.2)  # fast mode stops zero-throughput paths


def test_c4_million_spheres_fast_tree_pool():
    """configs[3] scene (10^6 spheres) at 256 x 144, depth 10, fast mode through k_tree_pool against the oracle, which
    walks the reference's own BVH. The cloud fills the reference's hard-coded light shapes (src/trace.rs:123-128), and
    Sphere::pdf_value / random evaluate sqrt(1 - r^2 / d^2) (src/sphere.rs:18,92), which cancels catastrophically for
    surface points next to the shape's sphere: there the reference's own rounding decides the value, a few ulp of
    difference in r^2 / d^2 become 1e-5 relative in the sampled direction, and such a path hits other spheres from then
    on. 0.3 % of the pixels hold such a path; every other pixel agrees to rounding."""
    scene = ptgpu.Scene.named("sphere_cloud", 16 / 9, param=1000000)
    got, ref, st, _ = _render_both(scene, 256, 144, 1, 10, budget=1 << 18)
    assert int(st.last_pass_form) == ptgpu.PT_PASS_TREE_POOL
    close, rel_rmse, nan_same = _compare(got, ref)
    print(f"C4 fast: close {close:.6f} rel_rmse {rel_rmse:.3e} nan_same {nan_same:.6f}")
    assert nan_same >= 0.999   # observed 0.99959
    assert close >= 0.995      # observed 0.99760
    # one sample per pixel and spheres that emit 30x: a path that goes elsewhere moves its pixel by its whole value
    assert rel_rmse <= 0.15    # observed 7.7e-2
    g, r = got[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
    assert abs(np.nanmean(g) / np.nanmean(r) - 1.0) <= 0.02


@pytest.mark.parametrize("name,param,depth,book", [("final_book_two", 0, 10, False), ("final_book_two", 0, 10, True),
                                                   ("cornell_smoke", 0, 20, False), ("cornell_smoke", 1, 20, True)])
def test_c3_media_scenes_fast_quirk_and_book(name, param, depth, book):
    """configs[2]: media, Perlin and image textures, moving sphere, nested transformed BVH; fast mode, default pass
    forms. Reference semantics (Isotropic yields 0 or NaN: the NaN masks must agree) and book semantics (uniform
    phase function: a finite image)."""
    scene = ptgpu.Scene.named(name, 1.0, param=param)
    rflags = ptgpu.PT_RENDER_BOOK_SEMANTICS if book else 0
    got, ref, st, _ = _render_both(scene, 96, 96, 4, depth, rflags=rflags, budget=1 << 18)
    assert int(st.last_pass_form) in (ptgpu.PT_PASS_TREE_POOL, ptgpu.PT_PASS_PATH_POOL)
    close, rel_rmse, nan_same = _compare(got, ref)
    print(f"C3 {name} book={book}: close {close:.6f} rel_rmse {rel_rmse:.3e} nan_same {nan_same:.6f} "
          f"nan_frac {np.isnan(ref[..., :3]).any(axis=-1).mean():.3f}")
    if book:
        # the smoke box with its light flipped is finite everywhere; the final scene keeps the reference's hard-coded
        # glass-sphere light shape, whose pdf is NaN for origins inside it (src/sphere.rs:92-94)
        if name == "cornell_smoke":
            assert not np.isnan(ref[..., :3]).any() and not np.isnan(got[..., :3]).any()
    else:
        assert np.isnan(ref[..., :3]).any()
    # Finite pixels agree to rounding (observed: all of them, relative RMSE 0). The NaN masks differ where a path
    # with throughput exactly 0 (a Lambertian sample below its surface) runs on INSIDE a floor box: its exit face
    # coincides with the neighbour box's entry face, the reference's tree order and the fast-mode tree break that tie
    # differently, and only one of the two continuations reaches a point inside the sphere light shape, where the
    # light pdf is NaN and 0 * NaN turns the pixel NaN. Observed agreement: 0.9969 (reference semantics), 0.9899
    # (book semantics, where fewer pixels are NaN to begin with); the smoke box: 1.0.
    assert nan_same >= (0.985 if name == "final_book_two" else 0.9999)
    assert close >= 0.9999
    assert rel_rmse <= 1e-6


def test_cornell_high_spp_same_seed_bias_bound():
    """configs[1] scene at 256 x 256 @ 256 spp, depth 50, same seed: the fast-mode path pool against the oracle. With
    identical sample streams the Monte Carlo noise cancels, so the relative RMSE bounds any bias of the fast
    arithmetic far below what a converged-image comparison with independent noise could."""
    scene = ptgpu.Scene.named("cornell_box", 1.0)
    got, ref, st, ost = _render_both(scene, 256, 256, 256, 50, budget=1 << 24)
    assert int(st.last_pass_form) == ptgpu.PT_PASS_PATH_POOL
    close, rel_rmse, nan_same = _compare(got, ref)
    g, r = got[..., :3].astype(np.float64), ref[..., :3].astype(np.float64)
    mean_ratio = g.mean() / r.mean()
    print(f"Cornell 256^2 @ 256: close {close:.6f} rel_rmse {rel_rmse:.3e} mean ratio - 1 {mean_ratio - 1:.3e}")
    assert nan_same == 1.0 and not np.isnan(g).any()
    # observed: every f32 pixel identical (the f64 means differ by ~1e-15 relative, far below f32 rounding)
    assert rel_rmse <= 1e-5
    assert abs(mean_ratio - 1.0) <= 1e-6
    assert close >= 0.999


@pytest.mark.parametrize("exact", [False, True])
def test_thin_lens_and_shutter_window(exact):
    """Camera::get_ray with aperture > 0 (lens_radius * random_in_unit_disk offsets, src/camera.rs:53-59,221-228) and a
    shutter window inside the moving spheres' motion: defocused RTiOW-style scene with MovingSpheres."""
    s = ptgpu.Scene(seed=77)
    ground = s.lambertian(s.texture_checker(s.texture_constant((0.2, 0.3, 0.1)), s.texture_constant((0.9, 0.9, 0.9))))
    nodes = [s.sphere((0, -1000, 0), 1000, ground)]
    rng = np.random.default_rng(3)
    for k in range(60):
        c = np.array([rng.uniform(-6, 6), 0.3, rng.uniform(-6, 6)])
        kind = k % 3
        mat = (s.lambertian(s.texture_constant(rng.uniform(0.1, 0.9, 3)), 8.0 if k % 10 == 0 else 0.0) if kind == 0
               else s.metal(rng.uniform(0.5, 1.0, 3), 0.1) if kind == 1 else s.dielectric(1.5))
        if k % 2:
            nodes.append(s.moving_sphere(c, c + (0, 0.6, 0), 0.0, 1.0, 0.3, mat))
        else:
            nodes.append(s.sphere(c, 0.3, mat))
    s.set_root(s.bvh(nodes, 0.2, 0.7))
    dummy = s.lambertian(s.texture_constant((0, 0, 0)))
    s.set_lights([s.rect(-2, 2, -2, 2, 6.0, ptgpu.AXIS_Y, dummy)])
    s.camera = ptgpu.make_camera((9, 2.5, 4), (0, 0.3, 0), (0, 1, 0), vfov=30.0, aspect=1.5, aperture=2.0, focus_dist=9.0,
                                 time0=0.2, time1=0.7)
    got, ref, st, _ = _render_both(s, 192, 128, 4, 20, flags=ptgpu.PT_CFG_EXACT if exact else 0, budget=1 << 18)
    close, rel_rmse, nan_same = _compare(got, ref)
    print(f"thin lens exact={exact}: close {close:.6f} rel_rmse {rel_rmse:.3e}")
    assert np.isfinite(ref[..., :3]).all() and ref[..., :3].max() > 0
    assert nan_same == 1.0
    assert close >= 0.9999     # observed 1.0 in both modes
    assert rel_rmse <= 1e-6
    # the lens really is open: the same scene through a pinhole gives a different image
    s.camera.aperture = 0.0
    pin, _ = ob.render(s.desc(), s.camera, 192, 128, 4, 20, seed=0x5EED0001, threads=16)
    assert np.abs(pin[..., :3] - ref[..., :3]).mean() > 1e-3


def _sphere_scene(n, seed=7, translated=False, textured=False):
    """n plain spheres of all four surface materials (enough of them for the tree form) built through the scene API."""
    rng = np.random.default_rng(seed)
    s = ptgpu.Scene()
    nodes = []
    checker = s.texture_checker(s.texture_constant((0.2, 0.3, 0.1)), s.texture_constant((0.9, 0.9, 0.9)))
    for k in range(n):
        c = rng.uniform(-40.0, 40.0, 3)
        pick = k % 5
        if pick == 0:
            m = s.lambertian(s.texture_constant(rng.uniform(0.1, 0.9, 3)), 0.0)
        elif pick == 1:
            m = s.lambertian(s.texture_constant(rng.uniform(0.1, 0.9, 3)), 12.0)
        elif pick == 2:
            m = s.metal(rng.uniform(0.5, 1.0, 3), float(rng.uniform(0.0, 0.3)))
        elif pick == 3:
            m = s.dielectric(1.5)
        else:
            m = s.diffuse_light(s.texture_constant(rng.uniform(1.0, 4.0, 3)))
        if textured and k == 0:
            m = s.lambertian(checker, 0.0)
        r = float(rng.uniform(1.0, 4.0))
        node = s.sphere(c, r, m)
        nodes.append(s.translate(node, (0.5, -0.25, 1.0)) if (translated and k == 1) else node)
    s.set_root(s.bvh(nodes))
    dummy = s.lambertian(s.texture_constant((0, 0, 0)), 0.0)
    s.set_lights([s.sphere((0.0, 60.0, 0.0), 5.0, dummy)])
    s.camera = ptgpu.make_camera((0.0, 0.0, -130.0), (0.0, 0.0, 0.0), vfov=40.0, aspect=1.5)
    return s


@pytest.mark.parametrize("case", ["cloud", "mixed_materials", "translated", "textured"])
def test_compact_sphere_records_give_the_same_image(case):
    """A scene of plain spheres alone is traced through 64-byte records that serve the leaf test and the vertex alike
    (k_tree_pool<.., SPH>); PT_CFG_FAT_RECORDS sends the same scene through the general primitive and leaf records.
    Same arithmetic on the same values: every f32 channel identical, NaN pixels included, offline and in realtime
    frames, also after the tree has been rebuilt on the device (which reorders the records). A checker colour (a texture
    evaluated from the hit point alone) travels in the record as a texture index; a scene that does not qualify (a sphere
    under a Translate) takes the general records either way. Trees of any size over compact records are read as 32-byte
    nodes with f32 slab tests (PT_CFG_WIDE_NODES: the 64-byte nodes and their f64 tests)."""
    if case == "cloud":
        scene, w, h, spp, depth = ptgpu.Scene.named("sphere_cloud", 16 / 9, param=200000), 256, 144, 2, 10
    else:
        scene = _sphere_scene(300, translated=case == "translated", textured=case == "textured")
        w, h, spp, depth = 96, 64, 4, 12
    images = {}
    forms = [("compact", 0), ("fat", ptgpu.PT_CFG_FAT_RECORDS), ("wide_nodes", ptgpu.PT_CFG_WIDE_NODES)]
    for label, flags in forms:
        ctx = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 18, flags=flags)
        ctx.upload(scene)
        ctx.render(w, h, spp, depth)
        assert int(ctx.stats().last_pass_form) == ptgpu.PT_PASS_TREE_POOL
        # which records the pass read: the compact ones unless the flag or the scene rules them out (a checker colour does
        # not), and with them the 32-byte nodes unless that flag is set
        want = 0 if (label == "fat" or case == "translated") else ptgpu.PT_RECORDS_COMPACT_SPHERES | (
            0 if label == "wide_nodes" else ptgpu.PT_RECORDS_NODES32)
        assert ctx.last_pass_records() == want, (label, case)
        offline = ctx.readback()
        ctx.reset()
        for frame in range(3):
            ctx.render(w, h, 1, depth, mode=ptgpu.PT_MODE_REALTIME, frame=frame)
        realtime = ctx.readback()
        ctx.build_tree_on_device(0.0, 1.0)
        ctx.render(w, h, spp, depth)
        rebuilt = ctx.readback()
        images[label] = (offline, realtime, rebuilt)
        ctx.close()
    for label, _ in forms[1:]:
        for a, b in zip(images["compact"], images[label]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), label
    # a tree over the same primitives finds the same hits (exact ties aside): the rebuilt tree's image is the same too
    assert np.array_equal(images["compact"][0].view(np.uint32), images["compact"][2].view(np.uint32))
    # and the compact form against the oracle on the small scenes
    if case != "cloud":
        ref, _ = ob.render(scene.desc(), scene.camera, w, h, spp, depth, seed=0x5EED0001, threads=4)
        close, rel_rmse, nan_same = _compare(images["compact"][0], ref)
        assert nan_same >= 0.99 and close >= 0.98, (close, rel_rmse, nan_same)


def test_sphere_recipes_take_the_compact_records():
    """random_scene (checker ground, a scene-wide leaf at the root) and the sphere cloud are traced through the compact
    records over 32-byte nodes, offline and in realtime frames; the final scene (boxes, media, wrappers) is not."""
    for name, param, want in (("random_scene", 0, ptgpu.PT_RECORDS_COMPACT_SPHERES | ptgpu.PT_RECORDS_NODES32),
                              ("sphere_cloud", 50000, ptgpu.PT_RECORDS_COMPACT_SPHERES | ptgpu.PT_RECORDS_NODES32),
                              ("final_book_two", 0, 0)):
        ctx = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 18)
        ctx.upload(ptgpu.Scene.named(name, 1.5, param=param))
        ctx.render(96, 64, 2, 10)
        assert int(ctx.stats().last_pass_form) == ptgpu.PT_PASS_TREE_POOL and ctx.last_pass_records() == want, name
        ctx.render(96, 64, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=1)
        assert ctx.last_pass_records() == want, name
        if name == "random_scene":
            # the 32-byte nodes of a small tree span the scene beside the root's leaf (the ground sphere), and rays start
            # past the root: the same image after a refit and after a rebuild on the device, whose root holds no such
            # leaf (both remake the nodes and the grid), and the image of the 64-byte nodes
            ctx.reset()
            ctx.render(96, 64, 2, 10)
            first = ctx.readback().copy()
            ctx.refit(0.0, 1.0)
            ctx.render(96, 64, 2, 10)
            assert np.array_equal(first.view(np.uint32), ctx.readback().view(np.uint32))
            ctx.build_tree_on_device(0.0, 1.0)
            ctx.render(96, 64, 2, 10)
            assert ctx.last_pass_records() == want
            assert np.array_equal(first.view(np.uint32), ctx.readback().view(np.uint32))
            wide = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 18, flags=ptgpu.PT_CFG_WIDE_NODES)
            wide.upload(ptgpu.Scene.named(name, 1.5, param=param))
            wide.render(96, 64, 2, 10)
            assert wide.last_pass_records() == ptgpu.PT_RECORDS_COMPACT_SPHERES
            assert np.array_equal(first.view(np.uint32), wide.readback().view(np.uint32))
            wide.close()
        ctx.close()


def test_tree_traversal_counters():
    """PT_CFG_COUNT_ACCEL: node records read and primitives tested per ray in the fast-mode tree, the figures bench.py
    prices the extend launches with. The counting form returns the hits of the timed form, every ray reads at least the
    root, and a ray that hits something has tested at least one primitive."""
    scene = ptgpu.Scene.named("random_scene", 1.5)
    pix = np.arange(128 * 96, dtype=np.uint64)
    rays = ob.camera_rays(scene.camera, 128, 96, pix, np.zeros(128 * 96, dtype=np.uint32))
    plain = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 16)
    plain.upload(scene)
    want = plain.intersect(rays)
    plain.close()
    ctx = ptgpu.Context(device=0, seed=0x5EED0001, max_paths_in_flight=1 << 16, flags=ptgpu.PT_CFG_COUNT_ACCEL)
    ctx.upload(scene)
    got = ctx.intersect(rays)
    st = ctx.stats()
    ctx.close()
    assert (got["prim"] == want["prim"]).all() and (got["t"].view(np.uint64) == want["t"].view(np.uint64)).all()
    n, n_hit = len(rays), int((want["prim"] >= 0).sum())
    assert st.node_visits >= n and st.prim_tests >= n_hit
    assert st.node_visits / n < 2 * np.log2(scene.accel_info()["n_nodes"]) + 20  # a tree walk, not a scan
