"""GPU tests at the sizes of the benchmark configurations: where the CPU oracle is too slow for the full workload,
the checks are size-independent properties (determinism, independence of the traversal form and of the math mode,
parity on a bounded sample of the same scene)."""
import numpy as np
import pytest

import oracle_binding as ob
import ptgpu

pytestmark = pytest.mark.gpu


def _ctx(**kw):
    return ptgpu.Context(device=0, seed=0x5EED0001, **kw)


def _rays(scene, w, h):
    pix = np.arange(w * h, dtype=np.uint64)
    primary = ob.camera_rays(scene.camera, w, h, pix, np.zeros(w * h, dtype=np.uint32))
    hits, _ = ob.intersect(scene.desc(), primary)
    secondary, valid = ob.secondary_rays(hits, primary)
    return primary, secondary[valid]


@pytest.fixture(scope="module")
def cloud():
    # configs[3]: one million spheres, BVH ~2 M ops deep ~21 (built on the host, src/bvh.rs:13-69)
    return ptgpu.Scene.named("sphere_cloud", 16 / 9, param=1000000)


@pytest.mark.parametrize("exact", [False, True])
def test_million_sphere_extend_parity(cloud, exact):
    ctx = _ctx(max_paths_in_flight=1 << 18, flags=ptgpu.PT_CFG_EXACT if exact else 0)
    ctx.upload(cloud)
    for batch in _rays(cloud, 256, 144):
        ref, _ = ob.intersect(cloud.desc(), batch)
        got = ctx.intersect(batch)
        same = got["prim"] == ref["prim"]
        assert same.mean() >= 0.9999
        both = same & (ref["prim"] >= 0)
        rel = np.abs(got["t"][both] - ref["t"][both]) / np.abs(ref["t"][both])
        assert rel.max() <= (1e-12 if exact else 1e-7)
    ctx.close()


def test_million_sphere_render_sample(cloud):
    ctx = _ctx(max_paths_in_flight=1 << 18, flags=ptgpu.PT_CFG_EXACT)
    ctx.upload(cloud)
    w, h = 128, 72
    ctx.render(w, h, 1, 10)
    got = ctx.readback()[..., :3].astype(np.float64)
    ref, _ = ob.render(cloud.desc(), cloud.camera, w, h, 1, 10, seed=ctx.seed, threads=8)
    close = np.isclose(got, ref[..., :3], rtol=1e-5, atol=1e-7, equal_nan=True)
    assert close.mean() >= 0.99
    ctx.close()


def test_rtiow_1200x800_one_sample_parity():
    # configs[0]: RTiOW random spheres at the reference's offline depth
    scene = ptgpu.Scene.named("random_scene", 1200 / 800)
    ctx = _ctx(max_paths_in_flight=1 << 20, flags=ptgpu.PT_CFG_EXACT)
    ctx.upload(scene)
    ctx.render(1200, 800, 1, 50)
    got = ctx.readback()[..., :3].astype(np.float64)
    ref, st = ob.render(scene.desc(), scene.camera, 1200, 800, 1, 50, seed=ctx.seed, threads=16)
    close = np.isclose(got, ref[..., :3], rtol=1e-5, atol=1e-7, equal_nan=True)
    assert close.mean() >= 0.995
    assert int(ctx.stats().segments) == pytest.approx(st.segments, rel=2e-3)
    ctx.close()


def test_cornell_full_size_properties():
    # configs[1] at full resolution and reduced spp
    scene = ptgpu.Scene.named("cornell_box", 1.0)
    w = h = 1024
    images = {}
    for name, tuning, flags in [("scan", {}, 0), ("scan_again", {}, 0), ("per_lane", {"scan_max_ops_plus1": 1}, 0),
                                ("exact", {}, ptgpu.PT_CFG_EXACT)]:
        ctx = _ctx(max_paths_in_flight=1 << 21, flags=flags, **tuning)
        ctx.upload(scene)
        ctx.render(w, h, 4, 50)
        images[name] = ctx.readback().copy()
        ctx.close()
    # deterministic: the same render twice is bit-identical (per-pixel sums run in sample order, no float atomics)
    assert np.array_equal(images["scan"], images["scan_again"], equal_nan=True)
    # the wave-uniform and the per-lane traversal visit the same ops in the same order: bit-identical images
    assert np.array_equal(images["scan"], images["per_lane"], equal_nan=True)
    # fast vs exact arithmetic: f64 both, a few ulp apart per operation; a path only differs where that flips a
    # hit / miss or sampling decision
    a, b = images["scan"][..., :3].astype(np.float64), images["exact"][..., :3].astype(np.float64)
    close = np.isclose(a, b, rtol=1e-5, atol=1e-7, equal_nan=True)
    assert close.mean() >= 0.99
    fin = np.isfinite(a) & np.isfinite(b)
    assert abs(a[fin].mean() - b[fin].mean()) / b[fin].mean() < 2e-3
    assert np.isnan(a).mean() < 1e-3


def test_final_scene_and_smoke_quirks():
    # configs[2]: media, Perlin and image textures, moving sphere, nested transformed BVH
    for name, depth in (("final_book_two", 10), ("cornell_smoke", 20)):
        scene = ptgpu.Scene.named(name, 1.0)
        ctx = _ctx(max_paths_in_flight=1 << 18, flags=ptgpu.PT_CFG_EXACT)
        ctx.upload(scene)
        w = h = 96
        ctx.render(w, h, 4, depth)
        got = ctx.readback()[..., :3].astype(np.float64)
        ref, _ = ob.render(scene.desc(), scene.camera, w, h, 4, depth, seed=ctx.seed, threads=16)
        ref = ref[..., :3].astype(np.float64)
        # the Isotropic quirk makes NaN pixels (0 * x / 0); the masks must agree
        assert (np.isnan(got) == np.isnan(ref)).mean() >= 0.995
        assert np.isnan(ref).any()
        close = np.isclose(got, ref, rtol=1e-5, atol=1e-7, equal_nan=True)
        assert close.mean() >= 0.99
        assert ctx.stats().nan_pixels > 0
        ctx.close()


def test_realtime_1080p_frames():
    # configs[4] on one GPU: 1920x1080, one sample per frame, max_depth 10, 80 % pixel skip (src/lib.rs:117-123)
    scene = ptgpu.Scene.named("cornell_box", 1920 / 1080)
    ctx = _ctx(max_paths_in_flight=1 << 22)
    ctx.upload(scene)
    w, h = 1920, 1080
    lit = []
    for frame in range(3):
        ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=frame)
        lit.append((ctx.readback()[..., :3] != 0).any(axis=-1).mean())
    st = ctx.stats()
    assert st.paths == 3 * w * h                      # paths offered; 80 % of them are skipped in raygen
    # a pixel is traced in a frame with probability 0.2 and a sample is non-black with some probability q, so the
    # lit fraction after k frames is 1 - (1 - 0.2 q)^k: the ratios to frame 1 lie between q = 1 and q -> 0
    assert 0.36 / 0.2 - 0.05 < lit[1] / lit[0] < 2.0
    assert 0.488 / 0.2 - 0.05 < lit[2] / lit[0] < 3.0
    assert st.segments < 0.25 * 3 * w * h * 11        # skipped pixels trace nothing
    ctx.reset()
    assert (ctx.readback() == 0).all()
    ctx.close()


def test_realtime_frames_of_a_tree_scene():
    """Realtime frames (80 % pixel skip, running blend, src/trace.rs:144-177) of a scene that is traced through the SAH
    tree: one pool launch per frame, whose waves evaluate the skip draw of 64 path ids at a time and hand the survivors
    to their free lanes, against ray generation plus an extend and a shade launch per bounce. Bit-identical images and
    segment counts over several frames, at widths that leave ragged chunks; and the frames agree with the oracle's."""
    scene = ptgpu.Scene.named("random_scene", 3 / 2)

    def run(w, h, frames, **tuning):
        ctx = _ctx(max_paths_in_flight=1 << 16, **tuning)
        ctx.upload(scene)
        ctx.reset()
        forms = set()
        for frame in range(frames):
            ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=frame)
            forms.add(int(ctx.stats().last_pass_form))
        img, st = ctx.readback(), ctx.stats()
        ctx.close()
        return img, st, forms

    for w, h in ((150, 100), (67, 31)):
        a, st_a, forms_a = run(w, h, 5, pass_form=ptgpu.PT_PASS_PER_BOUNCE)
        assert forms_a == {ptgpu.PT_PASS_PER_BOUNCE}
        for tuning in ({}, {"tree_refill": 1, "pool_chunk": 64}, {"tree_refill": 64, "tree_shade": 64, "pool_chunk": 4096}):
            b, st_b, forms_b = run(w, h, 5, **tuning)
            assert forms_b == {ptgpu.PT_PASS_TREE_POOL}, tuning
            assert st_b.launches_extend == 0 and st_b.launches_raygen == 0
            assert np.array_equal(a, b, equal_nan=True), (w, h, tuning)
            assert int(st_a.segments) == int(st_b.segments), tuning
    # against the oracle's frames
    w, h = 96, 64
    got, _, _ = run(w, h, 4)
    ref = np.zeros((h, w, 4), dtype=np.float32)
    counts = np.zeros((h, w), dtype=np.uint32)
    for frame in range(4):
        ob.render(scene.desc(), scene.camera, w, h, 1, 10, seed=0x5EED0001, mode=1, frame=frame, threads=8, image=ref, counts=counts)
    touched = counts > 0
    assert 0.4 < touched.mean() < 0.8            # 1 - 0.8^4 = 0.59
    assert (got[~touched][..., :3] == 0).all()
    close = np.isclose(got[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6)
    assert close.mean() >= 0.995


def test_pass_forms_give_the_same_image():
    """The ways a list scene's offline pass can be run -- one launch whose waves keep a pool of paths in registers and
    generate their own camera samples (the default), ray generation plus one launch of persistent waves over queue
    tiles (what realtime frames use), and ray generation plus an extend and a shade launch per bounce -- trace the
    same paths: images are bit-identical, whatever the pool's chunk size and refill threshold."""
    scene = ptgpu.Scene.named("cornell_box", 1.0)
    w = h = 512
    spp = 48  # 12.6 M paths: one pass with a 2^24 budget, 13 passes with 2^20

    def run(budget, **tuning):
        ctx = ptgpu.Context(device=0, max_paths_in_flight=budget, **tuning)
        ctx.upload(scene)
        ctx.render(w, h, spp, 50)
        img, st = ctx.readback(), ctx.stats()
        ctx.close()
        return img, st

    a, st = run(1 << 24)
    assert st.launches_raygen == 0 and st.launches_shade == st.launches_resolve == 1  # the pool: one launch per pass
    assert int(st.last_pass_form) == ptgpu.PT_PASS_PATH_POOL
    assert np.isfinite(a[..., :3]).all() and a[..., :3].mean() > 0.05
    b, st = run(1 << 24, pass_form=ptgpu.PT_PASS_PER_BOUNCE)
    assert st.launches_raygen == 1 and st.launches_extend == st.launches_shade == 51  # a launch per stage and bounce
    assert np.array_equal(a, b, equal_nan=True)
    segments = int(st.segments)
    for budget in (1 << 24, 1 << 20):
        c, st = run(budget, pass_form=ptgpu.PT_PASS_PERSISTENT)
        assert st.launches_raygen == st.launches_shade == st.launches_resolve  # persistent waves: raygen + one launch
        assert int(st.last_pass_form) == ptgpu.PT_PASS_PERSISTENT
        assert np.array_equal(a, c, equal_nan=True)
    for tuning in ({"pool_chunk": 64, "pool_refill": 1}, {"pool_chunk": 100000, "pool_refill": 64},
                   {"pool_grid": 3, "pool_refill": 40}):
        d, st = run(1 << 22, **tuning)
        assert np.array_equal(a, d, equal_nan=True), tuning
        assert int(st.segments) == segments, tuning


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke", "random_scene"])
def test_frames_issued_without_waiting_are_the_same_frames(name):
    """PT_RENDER_NO_WAIT: realtime frames are put on the context's stream one behind the other and the host waits once
    (the reference's main loop presents frame k while its workers trace frame k + 1, src/lib.rs:476-509). A frame of one
    sample per pixel through a pool kernel is one launch: its constants come with the launch packet and every path that
    ends blends its own pixel (src/trace.rs:166-177). A context that profiles every frame keeps the constants launch
    and the resolve launch. All three give the same image, segment count and NaN-pixel count."""
    scene = ptgpu.Scene.named(name, 16 / 9)
    w, h, frames = 800, 450, 12
    out = {}
    for key, cfg, rflags in (("waited", {}, 0), ("not waited", {}, ptgpu.PT_RENDER_NO_WAIT),
                             ("resolve launch", {"flags": ptgpu.PT_CFG_PROFILE_STAGES, "profile_period": 1}, 0)):
        ctx = ptgpu.Context(device=0, max_paths_in_flight=1 << 22, **cfg)
        ctx.upload(scene)
        for k in range(frames):
            ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=k, flags=rflags)
            if k == 5:  # a batch of rays between two frames uses the same counters
                rays = np.zeros(64, dtype=ptgpu.RAY_DTYPE)
                rays["direction"] = (0.0, 0.0, 1.0)
                ctx.intersect(rays)
        ctx.wait()
        st = ctx.stats()
        out[key] = (ctx.readback(), int(st.segments), int(st.nan_pixels), int(st.launches_resolve))
        ctx.close()
    for key in ("not waited", "resolve launch"):
        assert np.array_equal(out["waited"][0], out[key][0], equal_nan=True), key
        assert out["waited"][1:3] == out[key][1:3], key
    assert out["waited"][1] > 0 and out["waited"][3] == 0 and out["resolve launch"][3] == frames


@pytest.mark.parametrize("flags", [0, ptgpu.PT_CFG_EXACT])
def test_unit_sphere_points_made_by_the_wave_are_the_serial_loops(flags):
    """In the path pool of a scene with media the wave runs the rejection loop of random_in_unit_sphere together
    (src/material.rs:21-31: helper lanes make the candidates of a lane's iterations side by side); the launch-per-bounce
    form runs the loop lane by lane. Same draws, same points, same draw counts afterwards: the images are bit-identical,
    in book semantics, where an Isotropic scatter goes on in the sampled direction, and in the reference's."""
    scene = ptgpu.Scene.named("cornell_smoke", 1.0, param=1)
    w = h = 192
    images = {}
    for form in (ptgpu.PT_PASS_NONE, ptgpu.PT_PASS_PER_BOUNCE):
        ctx = ptgpu.Context(device=0, max_paths_in_flight=1 << 22, flags=flags, pass_form=form)
        ctx.upload(scene)
        for rflags in (ptgpu.PT_RENDER_BOOK_SEMANTICS, 0):
            ctx.render(w, h, 16, 50, flags=rflags)
            images[(form, rflags)] = ctx.readback()
            if form == ptgpu.PT_PASS_NONE:
                assert int(ctx.stats().last_pass_form) == ptgpu.PT_PASS_PATH_POOL
        ctx.close()
    for rflags in (ptgpu.PT_RENDER_BOOK_SEMANTICS, 0):
        a, b = images[(ptgpu.PT_PASS_NONE, rflags)], images[(ptgpu.PT_PASS_PER_BOUNCE, rflags)]
        assert np.array_equal(a, b, equal_nan=True)
    book = images[(ptgpu.PT_PASS_NONE, ptgpu.PT_RENDER_BOOK_SEMANTICS)][..., :3]
    assert np.isfinite(book).all() and book.mean() > 0.01  # the case traces smoke that scatters light on


@pytest.mark.parametrize("name,param", [("sphere_cloud", 50000), ("final_book_two", 0)])
def test_tree_traversal_forms_agree(name, param):
    """Fast mode walks the SAH tree either with one ray per thread or with waves that refill their free lanes from a
    run of the queue. Both test the same primitives with the same arithmetic, so ids and distances are identical,
    whatever the refill thresholds and however the rays are dealt to the waves (ragged batch sizes, one shard)."""
    scene = ptgpu.Scene.named(name, 1.0, param=param)
    primary, secondary = _rays(scene, 192, 160)
    batches = [primary, secondary, secondary[:4097], primary[:63], primary[:1]]

    def run(**tuning):
        ctx = _ctx(max_paths_in_flight=1 << 16, **tuning)
        ctx.upload(scene)
        out = [ctx.intersect(b, medium_seed=5) for b in batches]
        ctx.close()
        return out

    base = run(accel_form=1)
    for tuning in ({"accel_form": 2}, {"accel_form": 2, "accel_refill": 1, "accel_inner": 1},
                   {"accel_form": 2, "accel_refill": 64, "accel_inner": 65},
                   {"accel_form": 2, "shards": 1, "accel_grid": 1}, {"accel_form": 2, "shards": 64}, {}):
        got = run(**tuning)
        for a, b in zip(base, got):
            assert np.array_equal(a["prim"], b["prim"]), tuning
            assert np.array_equal(a["t"], b["t"], equal_nan=True), tuning
    assert (base[0]["prim"] >= 0).mean() > 0.1


def test_big_tree_in_32_byte_nodes():
    """Trees of 32 768 nodes and more are also kept with their boxes quantised to 16 bits on the scene's grid (half
    the loads per node visit), and the refilled-lane extend kernel walks that form. The quantised boxes contain the
    f32 ones, so the hits are those of the f32 tree, bit for bit: camera rays, incoherent secondary rays, rays that run
    parallel to one, two or all three axes (an infinite or undefined reciprocal direction), and the same again after
    a refit and after a rebuild of the tree on the device (both remake the quantised nodes)."""
    scene = ptgpu.Scene.named("sphere_cloud", 1.0, param=150000)
    assert scene.accel_info()["n_nodes"] >= 32768
    primary, secondary = _rays(scene, 160, 160)
    special = np.zeros(18, dtype=ptgpu.RAY_DTYPE)
    # ... and rays from far outside the grid of the quantised planes, with components of wildly different sizes (the slab
    # tests of the 32-byte nodes run in f32, widened by their rounding bound: they must not cull what the f64 tests keep)
    special["origin"] = ([(500.3, 500.1, -10.0)] * 4 + [(-10.0, 499.7, 500.2)] * 4 + [(500.0, 500.0, 500.0)] * 4 +
                         [(500.2, 499.9, -1.0e7), (1.0e6, 1.0e6, 1.0e6), (500.1, -100.0, 500.3), (499.0, 501.0, -3.0e4),
                          (2.0e5, 500.0, 500.0), (500.0, 500.0, 500.0)])
    special["direction"] = [(0, 0, 1), (0, 0, 2.5), (1e-300, 0, 1), (0, 1e-9, 1), (1, 0, 0), (3, 0, 0), (1, 0, 1e-12), (1, 1, 0),
                            (0, 0, 0), (0, -1, 0), (-1, 0, 0), (0, 0, -1),
                            (0, 0, 1), (-1, -1, -1), (1e-30, 1, 1e-30), (1e-5, -1e-5, 1), (-1e8, 0, 1e-8), (1e-200, 1e-200, 1e-200)]
    batches = [primary, secondary, special]

    def run(steps, **tuning):
        ctx = _ctx(max_paths_in_flight=1 << 16, **tuning)
        ctx.upload(scene)
        out = []
        for step in steps:
            if step == "refit":
                ctx.refit(0.0, 1.0)
            elif step == "build":
                ctx.build_tree_on_device(0.0, 1.0)
            out.append([ctx.intersect(b) for b in batches])
        ctx.close()
        return out

    f32_nodes = run(["upload", "refit", "build"], accel_form=1)   # a ray per thread reads the f32 nodes
    quantised = run(["upload", "refit", "build"])                 # the lane pool reads the 32-byte nodes
    for stage, (a_all, b_all) in enumerate(zip(f32_nodes, quantised)):
        for a, b in zip(a_all, b_all):
            assert np.array_equal(a["prim"], b["prim"]), stage
            assert np.array_equal(a["t"], b["t"], equal_nan=True), stage
    assert (f32_nodes[0][0]["prim"] >= 0).mean() > 0.1
    assert (f32_nodes[0][2]["prim"] >= 0).any()


@pytest.mark.parametrize("name,param,rflags", [("random_scene", 0, 0), ("final_book_two", 0, 0),
                                               ("final_book_two", 1, ptgpu.PT_RENDER_BOOK_SEMANTICS), ("sphere_cloud", 50000, 0)])
def test_tree_pool_gives_the_same_image(name, param, rflags):
    """Scenes traced through the SAH tree: an offline pass run as one launch whose lanes carry a path each (traversal
    and shading interleaved inside the wave) against ray generation plus an extend and a shade launch per bounce.
    Same paths, same arithmetic per path: bit-identical images and segment counts, whatever the thresholds. The final
    scene with its own light and book semantics has finite pixels all over its noise-textured sphere, whose turbulence
    the one-launch form evaluates with the lanes of a wave working together and the other form lane by lane."""
    scene = ptgpu.Scene.named(name, 4 / 3, param=param)
    w, h, spp = 160, 120, 6

    def run(**tuning):
        ctx = _ctx(max_paths_in_flight=1 << 16, **tuning)
        ctx.upload(scene)
        ctx.render(w, h, spp, 50, flags=rflags)
        img, st = ctx.readback(), ctx.stats()
        ctx.close()
        return img, st

    a, st_a = run(pass_form=ptgpu.PT_PASS_PER_BOUNCE)
    assert st_a.launches_extend > 0
    if rflags:
        assert np.isfinite(a[..., :3]).mean() > 0.99
    for tuning in ({}, {"tree_refill": 1, "tree_inner": 1, "tree_shade": 1, "tree_trav_plus1": 1},
                   {"tree_refill": 64, "tree_inner": 64, "tree_shade": 64, "tree_trav_plus1": 64, "pool_chunk": 64}):
        b, st_b = run(**tuning)
        assert st_b.launches_extend == 0 and st_b.launches_raygen == 0, tuning
        assert int(st_b.last_pass_form) == ptgpu.PT_PASS_TREE_POOL
        assert np.array_equal(a, b, equal_nan=True), tuning
        assert int(st_a.segments) == int(st_b.segments), tuning


def _wrapped_subtree_scene(with_medium):
    """A short stream WITH frames: a BVH of spheres under translate(rotate_y(...)) (ENTER / EXIT ops in the stream, as
    around the sphere cluster of the final scene), on a lit stage; optionally a medium in a wrapped box."""
    s = ptgpu.Scene(seed=77)
    white = s.lambertian(s.texture_constant((0.73, 0.73, 0.73)))
    green = s.lambertian(s.texture_constant((0.12, 0.45, 0.15)))
    light = s.diffuse_light(s.texture_constant((9.0, 9.0, 9.0)))
    rng = np.random.default_rng(5)
    balls = [s.sphere(tuple(rng.uniform(0, 120, 3)), 12.0, white if k % 2 else s.dielectric(1.5)) for k in range(6)]
    cluster = s.translate(s.rotate_y(s.bvh(balls, 0.0, 1.0), 15.0), (200.0, 150.0, 250.0))
    nodes = [s.flip_normals(s.rect(150, 400, 150, 400, 554, 1, light)), s.rect(0, 555, 0, 555, 0, 1, white),
             s.flip_normals(s.rect(0, 555, 0, 555, 555, 2, green)), cluster]
    if with_medium:
        box = s.translate(s.rotate_y(s.box((0, 0, 0), (120, 120, 120), white), -20.0), (60.0, 0.0, 120.0))
        nodes.append(s.constant_medium(box, 0.01, s.texture_constant((1.0, 1.0, 1.0))))
    s.set_root(s.bvh(nodes, 0.0, 1.0))
    s.set_lights([s.rect(150, 400, 150, 400, 554, 1, white)])
    s.camera = ptgpu.make_camera((278, 278, -760), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    assert s.flatten_info()["max_frame_depth"] == 1
    return s


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke", "wrapped_subtree", "wrapped_subtree_medium"])
def test_exact_mode_pool_gives_the_same_image(name):
    """Exact mode on short op streams: one pool launch per pass around the wave-uniform walk of the reference's own
    tree against ray generation plus an extend and a shade launch per bounce. Bit-identical images. The recipes'
    streams hold no ENTER / EXIT ops (the pool variants whose ray never changes frame); the two built scenes do."""
    scene = ptgpu.Scene.named(name, 1.0) if not name.startswith("wrapped") else _wrapped_subtree_scene(name.endswith("medium"))
    w = h = 96

    def run(**tuning):
        ctx = _ctx(max_paths_in_flight=1 << 16, flags=ptgpu.PT_CFG_EXACT, **tuning)
        ctx.upload(scene)
        ctx.render(w, h, 12, 50)
        img, st = ctx.readback(), ctx.stats()
        ctx.close()
        return img, st

    a, st_a = run(pass_form=ptgpu.PT_PASS_PER_BOUNCE)
    b, st_b = run()
    assert st_a.launches_extend > 0 and st_b.launches_extend == 0
    assert int(st_b.last_pass_form) == ptgpu.PT_PASS_PATH_POOL
    assert np.array_equal(a, b, equal_nan=True)
    assert int(st_a.segments) == int(st_b.segments)
    ref, _ = ob.render(scene.desc(), scene.camera, w, h, 12, 50, seed=0x5EED0001, threads=8)
    finite = np.isfinite(b[..., :3]) & np.isfinite(ref[..., :3])
    assert (np.isnan(b[..., :3]) == np.isnan(ref[..., :3])).mean() >= 0.995
    assert (np.isclose(b[..., :3], ref[..., :3], rtol=1e-5, atol=1e-7) | ~finite).mean() >= 0.99


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (7, 3, 5), (65, 2, 3), (256, 1, 257)])
def test_pool_on_tiny_and_ragged_passes(w, h, spp):
    """Passes far smaller than one wave per compute unit, and passes whose last chunk of path ids is ragged: the pool
    form and the launch-per-bounce form still give the same image, and every pixel is written."""
    scene = ptgpu.Scene.named("cornell_box", w / h)

    def run(**tuning):
        ctx = _ctx(max_paths_in_flight=1 << 12, **tuning)
        ctx.upload(scene)
        ctx.render(w, h, spp, 50)
        img, st = ctx.readback(), ctx.stats()
        ctx.close()
        return img, st

    a, st_a = run()
    b, st_b = run(pass_form=ptgpu.PT_PASS_PER_BOUNCE)
    assert int(st_a.last_pass_form) == ptgpu.PT_PASS_PATH_POOL
    assert int(st_b.last_pass_form) == ptgpu.PT_PASS_PER_BOUNCE
    assert int(st_a.paths) == int(st_b.paths) == w * h * spp
    assert np.array_equal(a, b, equal_nan=True)
    assert int(st_a.segments) == int(st_b.segments)


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke", "two_spheres", "two_perlin_spheres", "textured_sphere",
                                  "simple_light"])
def test_realtime_frames_of_a_list_scene_through_the_pool(name):
    """Realtime frames of a list scene run through the path pool (its waves evaluate the 80 % pixel skip of 64 path ids
    at a time before they make camera samples, take their chunks of ids by wave index, and blend the pixels of their
    finished paths themselves); the queue form can be forced. Either way, at a small, a ragged and a medium size, the
    accumulated frames are bit-identical, and they agree with the oracle."""
    scene = ptgpu.Scene.named(name, 1.0)
    for (w, h) in [(96, 64), (67, 3), (640, 400)]:
        images = {}
        for form in (ptgpu.PT_PASS_PATH_POOL, ptgpu.PT_PASS_PERSISTENT):
            ctx = ptgpu.Context(device=0, max_paths_in_flight=1 << 20, pass_form=form)
            ctx.upload(scene)
            for frame in range(4):
                ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=frame)
            images[form] = ctx.readback()
            assert int(ctx.stats().last_pass_form) == form
            ctx.close()
        assert np.array_equal(images[ptgpu.PT_PASS_PATH_POOL], images[ptgpu.PT_PASS_PERSISTENT], equal_nan=True), (w, h)
    # the rows of a group of two, frame by frame into one image, are the whole frames (the pool maps its path ids to the
    # member's rows)
    ctx = ptgpu.Context(device=0, max_paths_in_flight=1 << 20, pass_form=ptgpu.PT_PASS_PATH_POOL)
    ctx.upload(scene)
    for frame in range(4):
        for phase in (0, 1):
            ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=frame, rows=(2, phase))
    assert np.array_equal(ctx.readback(), images[ptgpu.PT_PASS_PATH_POOL], equal_nan=True)
    ctx.close()
    ref = np.zeros((h, w, 4), dtype=np.float32)
    counts = np.zeros((h, w), dtype=np.uint32)
    for frame in range(4):
        ob.render(scene.desc(), scene.camera, w, h, 1, 10, seed=0x5EED0001, mode=1, frame=frame, threads=8, image=ref, counts=counts)
    got = images[ptgpu.PT_PASS_PATH_POOL]
    assert (got[counts == 0][..., :3] == 0).all()
    assert np.isclose(got[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6, equal_nan=True).mean() >= 0.98


def _tile_grid(w, h, n):
    tw, th = (w + n - 1) // n, (h + n - 1) // n
    return [(x, y, min(x + tw, w), min(y + th, h)) for y in range(0, h, th) for x in range(0, w, tw)]


@pytest.mark.parametrize("name", ["cornell_box", "random_scene"])
def test_batch_of_tile_jobs(name):
    """pt_render_jobs, the reference's dispatch of one job per tile (src/lib.rs:380-399): a batch that tiles a rectangle
    is traced as one pass and gives, bit for bit, the image of the jobs run one by one; batches that cannot be merged
    (a hole, overlapping tiles, unequal sample counts) are run job by job with the same result as pt_render calls."""
    scene = ptgpu.Scene.named(name, 1.25)
    w, h = 150, 120
    ctx = ptgpu.Context(device=0, max_paths_in_flight=1 << 20)
    ctx.upload(scene)
    tiles = _tile_grid(w, h, 10)
    assert len(tiles) == 100

    def one_by_one(jobs, spp_of=None, mode=ptgpu.PT_MODE_OFFLINE, frames=1):
        ctx.reset()
        for f in range(frames):
            for k, t in enumerate(jobs):
                ctx.render(w, h, spp_of[k] if spp_of else 3, 12, tile=t, mode=mode, frame=f)
        return ctx.readback(), int(ctx.stats().launches_resolve)

    def batched(jobs, spp_of=None, mode=ptgpu.PT_MODE_OFFLINE, frames=1):
        ctx.reset()
        for f in range(frames):
            ctx.render_jobs(w, h, 3, 12, jobs, mode=mode, frame=f, spp_of=spp_of)
        return ctx.readback(), int(ctx.stats().launches_resolve)

    ctx.stats_reset()
    ref, n_ref = one_by_one(tiles)
    ctx.stats_reset()
    got, n_got = batched(tiles)
    assert np.array_equal(got, ref, equal_nan=True)
    assert n_ref == 100 and n_got == 1  # one pass for the whole grid
    # realtime frames accumulate the same way
    ref, _ = one_by_one(tiles, mode=ptgpu.PT_MODE_REALTIME, frames=3)
    got, _ = batched(tiles, mode=ptgpu.PT_MODE_REALTIME, frames=3)
    assert np.array_equal(got, ref, equal_nan=True)
    # a sub-rectangle of the image, shuffled
    sub = [t for t in tiles if 15 <= t[0] < 105 and 24 <= t[1] < 96][::-1]
    ref, _ = one_by_one(sub)
    ctx.stats_reset()
    got, n_got = batched(sub)
    assert np.array_equal(got, ref, equal_nan=True) and n_got == 1
    # not mergeable: a hole; overlapping tiles (the later job overwrites); unequal sample counts
    hole = tiles[:37] + tiles[38:]
    overlap = tiles + [(10, 10, 40, 30)]
    spp_of = [3 if k % 2 else 2 for k in range(len(tiles))]
    for jobs, spps in ((hole, None), (overlap, None), (tiles, spp_of)):
        ref, n_ref = one_by_one(jobs, spps)
        ctx.stats_reset()
        got, n_got = batched(jobs, spps)
        assert np.array_equal(got, ref, equal_nan=True)
        assert n_got == len(jobs)
    with pytest.raises(ptgpu.PtError):
        ctx.render_jobs(w, h, 3, 12, [(0, 0, w + 1, h)])
    ctx.close()
