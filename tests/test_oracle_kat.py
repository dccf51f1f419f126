"""Known-answer tests that pin the CPU oracle (oracle/pt_oracle.cpp).

The reference crate holds no golden vectors for the render path ("parity unpinned"); its only test checks Vec3
arithmetic (src/math/vec3.rs:495-577), restated here against the oracle's Vec3. Everything else is checked
against closed forms: analytic ray/sphere and ray/rect hits, the slab test, a rotated box, the camera frame,
a furnace (energy conservation) scene, and the committed golden fixtures under tests/golden/.
"""
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
import ptgpu

F64_MAX = np.finfo(np.float64).max
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def ray(o, d, time=0.0):
    r = np.zeros(1, dtype=ptgpu.RAY_DTYPE)
    r["origin"][0] = o
    r["direction"][0] = d
    r["time"][0] = time
    return r


# ------------------------------------------------------------------ Vec3 (src/math/vec3.rs:495-577)
def test_vec3_reference_asserts():
    assert ob.vec3_op(0, (1, 1, 1), (1, 2, 3)) == (2.0, 3.0, 4.0)            # add vec
    assert ob.vec3_op(0, (1, 1, 1), (1, 1, 1)) != (2.0, 3.0, 4.0)
    assert ob.vec3_op(0, (0, 0, 1), (1, 0, 0)) == (1.0, 0.0, 1.0)            # add assign
    assert ob.vec3_op(4, (0, 1, 2), (1, 0, 0)) == (1.0, 2.0, 3.0)            # add float
    assert ob.vec3_op(1, (1, 1, 1), (1, 2, 3)) == (0.0, -1.0, -2.0)          # sub vec
    assert ob.vec3_op(1, (0, 0, 1), (1, 0, 0)) == (-1.0, 0.0, 1.0)           # sub assign
    assert ob.vec3_op(5, (0, 1, 2), (1, 0, 0)) == (-1.0, 0.0, 1.0)           # sub float
    assert ob.vec3_op(2, (2, 2, 2), (1, 2, 3)) == (2.0, 4.0, 6.0)            # mul vec
    assert ob.vec3_op(2, (3, 0, 3), (2, 1, 3)) == (6.0, 0.0, 9.0)            # mul assign
    assert ob.vec3_op(6, (0, 1, 2), (2, 0, 0)) == (0.0, 2.0, 4.0)            # mul float
    assert ob.vec3_op(3, (2, 2, 2), (1, 2, 4)) == (2.0, 1.0, 0.5)            # div vec
    assert ob.vec3_op(3, (4, 0, 9), (2, 1, 3)) == (2.0, 0.0, 3.0)            # div assign
    assert ob.vec3_op(7, (0, 1, 2), (2, 0, 0)) == (0.0, 0.5, 1.0)            # div float
    assert ob.vec3_scalar(0, (2, 2, 2)) == 12.0                               # squared length
    assert ob.vec3_scalar(1, (2, 2, 2)) == math.sqrt(12.0)                    # length
    length = math.sqrt(3.0)
    assert ob.vec3_op(8, (1, 1, 1)) == (1.0 / length, 1.0 / length, 1.0 / length)  # normalise
    assert ob.vec3_op(9, (1, 0, 0), (0, 1, 0)) == (0.0, -0.0, 1.0)            # cross (src/math/vec3.rs:137-143)


# ------------------------------------------------------------------ primitives
def _single(scene, node):
    scene.set_root(node)
    scene.set_lights([node])
    return scene.desc()


def test_sphere_closed_form():
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((0.5, 0.5, 0.5)))
    d = _single(s, s.sphere((0, 0, -5), 2.0, m))
    hits, _ = ob.intersect(d, ray((0, 0, 0), (0, 0, -1)))
    assert hits["prim"][0] >= 0 and hits["t"][0] == 3.0
    assert tuple(hits["p"][0]) == (0.0, 0.0, -3.0) and tuple(hits["normal"][0]) == (0.0, 0.0, 1.0)
    # un-normalised direction halves t (src/sphere.rs:47-53)
    hits, _ = ob.intersect(d, ray((0, 0, 0), (0, 0, -2)))
    assert hits["t"][0] == 1.5
    # from inside: the far root (src/sphere.rs:66-67)
    hits, _ = ob.intersect(d, ray((0, 0, -5), (1, 0, 0)))
    assert hits["t"][0] == 2.0 and tuple(hits["normal"][0]) == (1.0, 0.0, 0.0)
    # grazing: discriminant == 0 is a miss (strict >, src/sphere.rs:51)
    hits, _ = ob.intersect(d, ray((2, 0, 0), (0, 0, -1)))
    assert hits["prim"][0] == -1
    # t must be strictly inside (t_min, t_max)
    hits, _ = ob.intersect(d, ray((0, 0, 0), (0, 0, -1)), t_min=0.001, t_max=3.0)
    assert hits["prim"][0] == -1 or hits["t"][0] != 3.0


def test_sphere_uv_quirk():
    # near root: uv from the negated unit normal; far root: from the un-normalised vector (src/sphere.rs:56 vs 69)
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((0.5, 0.5, 0.5)))
    d = _single(s, s.sphere((0, 0, 0), 2.0, m))
    hits, _ = ob.intersect(d, ray((0, 0, 5), (0, 0, -1)))
    # p = (0,0,2); (c-p)/r = (0,0,-1): phi = atan2(-1, 0) = -pi/2, theta = 0
    assert hits["u"][0] == pytest.approx(1.0 - (-math.pi / 2 + math.pi) / (2 * math.pi))
    assert hits["v"][0] == pytest.approx(0.5)
    hits, _ = ob.intersect(d, ray((0, 0.5, 0), (0, 1, 0)))   # inside, far root, c - p = (0,-2,0): asin(-2) = NaN
    assert math.isnan(hits["v"][0])


def test_rect_closed_form_and_inclusive_ends():
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((0.5, 0.5, 0.5)))
    d = _single(s, s.rect(0, 4, 0, 2, -3.0, ptgpu.AXIS_Z, m))
    hits, _ = ob.intersect(d, ray((1, 1, 0), (0, 0, -1)))
    assert hits["t"][0] == 3.0 and hits["u"][0] == 0.25 and hits["v"][0] == 0.5
    assert tuple(hits["normal"][0]) == (0.0, 0.0, 1.0)
    # edge points are accepted (src/rect.rs:103 uses < and >)
    hits, _ = ob.intersect(d, ray((4, 2, 0), (0, 0, -1)))
    assert hits["prim"][0] >= 0
    hits, _ = ob.intersect(d, ray((4.0000001, 2, 0), (0, 0, -1)))
    assert hits["prim"][0] == -1
    # t == t_max is accepted for rects (src/rect.rs:99), unlike spheres
    hits, _ = ob.intersect(d, ray((1, 1, 0), (0, 0, -1)), t_min=0.001, t_max=3.0)
    assert hits["prim"][0] >= 0 and hits["t"][0] == 3.0
    mn, mx = ob.bounding_box(d, d.root)
    assert tuple(mn) == (0.0, 0.0, -3.0 - 0.0001) and tuple(mx) == (4.0, 2.0, -3.0 + 0.0001)


def test_slab_test_semantics():
    mn, mx = (0, 0, 0), (1, 1, 1)
    assert ob.aabb_hit(mn, mx, (-1, 0.5, 0.5), (1, 0, 0), 0.001, F64_MAX)
    assert not ob.aabb_hit(mn, mx, (-1, 2.0, 0.5), (1, 0, 0), 0.001, F64_MAX)
    assert not ob.aabb_hit(mn, mx, (2, 0.5, 0.5), (1, 0, 0), 0.001, F64_MAX)       # behind the ray
    assert ob.aabb_hit(mn, mx, (2, 0.5, 0.5), (-1, 0, 0), 0.001, F64_MAX)          # swap when inv_d < 0
    assert not ob.aabb_hit(mn, mx, (-1, 0.5, 0.5), (1, 0, 0), 0.001, 1.0)          # tmax <= tmin rejects (aabb.rs:41)
    assert ob.aabb_hit(mn, mx, (0.5, 0.5, 0.5), (0, 0, 1), 0.001, F64_MAX)          # zero components: +-inf slabs


def test_rotated_translated_box():
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((0.7, 0.7, 0.7)))
    box = s.box((0, 0, 0), (2, 2, 2), m)
    rot = s.rotate_y(box, 90.0)
    node = s.translate(rot, (10, 0, 0))
    d = _single(s, node)
    # RotateY(90): object (x, z) -> world (z, -x); box [0,2]^3 -> x in [0,2], z in [-2,0]; then +10 in x
    mn, mx = ob.bounding_box(d, node)
    assert np.allclose(mn, (10, 0, -2), atol=1e-12) and np.allclose(mx, (12, 2, 0), atol=1e-12)
    hits, _ = ob.intersect(d, ray((11, 1, 5), (0, 0, -1)))
    assert hits["prim"][0] >= 0 and hits["t"][0] == pytest.approx(5.0, abs=1e-12)
    assert np.allclose(hits["normal"][0], (0, 0, 1), atol=1e-12)
    assert np.allclose(hits["p"][0], (11, 1, 0), atol=1e-12)
    hits, _ = ob.intersect(d, ray((11, 1, 5), (0, 1, 0)))
    assert hits["prim"][0] == -1
    # reported primitive is one of the six face rects of the box
    faces = {d.child_index[d.hitables[box].list_start + k] for k in range(6)}
    face_rects = {f if d.hitables[f].kind == ptgpu.H_RECT else d.hitables[f].child[0] for f in faces}
    hits, _ = ob.intersect(d, ray((11, 1, 5), (0, 0, -1)))
    assert int(hits["prim"][0]) in face_rects


def test_moving_sphere_center_and_box():
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((0.5, 0.5, 0.5)))
    node = s.moving_sphere((0, 0, 0), (4, 0, 0), 0.0, 1.0, 1.0, m)
    d = _single(s, node)
    hits, _ = ob.intersect(d, ray((2, 0, 5), (0, 0, -1), time=0.5))
    assert hits["t"][0] == 4.0
    hits, _ = ob.intersect(d, ray((2, 0, 5), (0, 0, -1), time=0.0))
    assert hits["prim"][0] == -1
    mn, mx = ob.bounding_box(d, node, 0.0, 1.0)
    assert tuple(mn) == (-1.0, -1.0, -1.0) and tuple(mx) == (5.0, 1.0, 1.0)


def test_constant_medium_statistics():
    # free-flight distance is exponential with rate = density (src/volume.rs:42)
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((0.5, 0.5, 0.5)))
    boundary = s.sphere((0, 0, 0), 100.0, m)
    density = 0.05
    med = s.constant_medium(boundary, density, s.texture_constant((1, 1, 1)))
    d = _single(s, med)
    n = 20000
    rays = np.zeros(n, dtype=ptgpu.RAY_DTYPE)
    rays["origin"][:] = (0, 0, 0)
    rays["direction"][:] = (0, 0, 2.0)           # un-normalised: distances are scaled by |d| (src/volume.rs:40-44)
    hits, _ = ob.intersect(d, rays, medium_seed=3)
    hit = hits["prim"] >= 0
    # inside the sphere from the centre: path length 100 -> P(scatter) = 1 - exp(-density * 100)
    assert hit.mean() == pytest.approx(1 - math.exp(-density * 100), abs=0.01)
    dist = hits["t"][hit] * 2.0
    assert dist.mean() == pytest.approx(1 / density, rel=0.05) or dist.max() <= 100.0
    assert (hits["prim"][hit] == med).all() and np.all(hits["normal"][hit] == (1.0, 0.0, 0.0))


# ------------------------------------------------------------------ camera / RNG / textures
def test_camera_frame_cornell():
    s = ptgpu.Scene.named("cornell_box", 1.0)
    f = ob.camera_frame(s.camera)
    origin, llc, horiz, vert, u, v, w = f
    assert tuple(origin) == (278.0, 278.0, -800.0)
    assert np.allclose(w, (0, 0, -1)) and np.allclose(u, (-1, 0, 0)) and np.allclose(v, (0, 1, 0))
    hh = math.tan(math.radians(40.0) / 2) * 10.0
    assert np.allclose(horiz, (-2 * hh, 0, 0)) and np.allclose(vert, (0, 2 * hh, 0))
    assert np.allclose(llc, (278 + hh, 278 - hh, -790))
    # centre ray points down +z
    rays = ob.camera_rays(s.camera, 2, 2, np.array([0]), np.array([0]))
    assert rays["direction"][0][2] == pytest.approx(10.0)


def test_rng_streams_are_counter_based():
    a = [ob.lib.oracle_path_draw(1, 0, 5, 7, n) for n in range(1, 6)]
    b = [ob.lib.oracle_path_draw(1, 0, 5, 7, n) for n in range(1, 6)]
    assert a == b and len(set(a)) == 5 and all(0.0 <= x < 1.0 for x in a)
    assert ob.lib.oracle_path_draw(1, 0, 5, 8, 1) != a[0] and ob.lib.oracle_path_draw(1, 1, 5, 7, 1) != a[0]
    xs = np.array([ob.lib.oracle_path_draw(9, 0, p, 0, 1) for p in range(4000)])
    assert abs(xs.mean() - 0.5) < 0.02 and abs(np.mean(xs < 0.8) - 0.8) < 0.02


def test_rng_stream_statistics():
    """The path stream (two PCG hash rounds per draw) is uniform and uncorrelated along each axis that the
    renderer varies: draw index, pixel, sample and frame."""
    draw = ob.lib.oracle_path_draw
    n = 20000
    along = {
        "draw": np.array([draw(3, 0, 11, 5, k) for k in range(1, n + 1)]),
        "pixel": np.array([draw(3, 0, p, 5, 1) for p in range(n)]),
        "sample": np.array([draw(3, 0, 11, s, 2) for s in range(n)]),
        "frame": np.array([draw(3, f, 11, 5, 3) for f in range(n)]),
    }
    for name, xs in along.items():
        assert xs.min() >= 0.0 and xs.max() < 1.0, name
        assert abs(xs.mean() - 0.5) < 0.01 and abs(xs.var() - 1.0 / 12.0) < 0.004, name
        for lag in (1, 2, 7):
            c = np.corrcoef(xs[:-lag], xs[lag:])[0, 1]
            assert abs(c) < 0.03, (name, lag, c)
        # consecutive pairs fill the unit square evenly: chi-square over a 16 x 16 grid (255 dof)
        h, _, _ = np.histogram2d(xs[:-1], xs[1:], bins=16, range=[[0, 1], [0, 1]])
        e = (n - 1) / 256.0
        chi2 = ((h - e) ** 2 / e).sum()
        assert chi2 < 360.0, (name, chi2)
    # every draw is a multiple of 2^-32
    assert all(float(x * 4294967296.0).is_integer() for x in along["draw"][:100])


def test_textures():
    s = ptgpu.Scene()
    c0, c1 = s.texture_constant((0.2, 0.3, 0.1)), s.texture_constant((0.9, 0.9, 0.9))
    chk = s.texture_checker(c0, c1)
    noise = s.texture_noise(4.0)
    img = np.zeros((4, 8, 3), dtype=np.uint8)
    img[1, 2] = (255, 128, 0)
    it = s.texture_image(img)
    m = s.lambertian(chk)
    d = _single(s, s.sphere((0, 0, 0), 1.0, m))
    p = (0.1, 0.1, 0.1)
    assert tuple(ob.texture_value(d, chk, 0, 0, p)) == (0.2, 0.3, 0.1)          # sines > 0 -> even (texture.rs:49-54)
    assert tuple(ob.texture_value(d, chk, 0, 0, (-0.1, 0.1, 0.1))) == (0.9, 0.9, 0.9)
    g = ob.texture_value(d, noise, 0, 0, (1.3, 2.1, -0.7))
    assert g[0] == g[1] == g[2] and 0.0 <= g[0] <= 1.0
    # nearest texel, i = u * W, j = v * H - 0.001, no v flip (texture.rs:102-113)
    assert np.allclose(ob.texture_value(d, it, 2.5 / 8, 1.5 / 4, p), (1.0, 128 / 255.0, 0.0))
    assert tuple(ob.texture_value(d, it, 5.0, 5.0, p)) == (0.0, 0.0, 0.0)          # clamped to the last texel
    # Perlin noise vanishes on the integer lattice and is bounded
    for q in [(0, 0, 0), (3, -2, 7)]:
        assert ob.lib.oracle_perlin_noise(d.perlin, ob._v(q)) == 0.0
    vals = [ob.lib.oracle_perlin_noise(d.perlin, ob._v((0.37 * k, 1.1 * k, -0.6 * k))) for k in range(1, 200)]
    assert max(abs(v) for v in vals) < 1.0 and np.std(vals) > 0.05
    assert ob.lib.oracle_perlin_turb(d.perlin, ob._v((0.3, 0.2, 0.9)), 7) >= 0.0


def test_tonemap_matches_reference_formula():
    rgba = np.zeros((2, 2, 4), dtype=np.float32)
    rgba[0, 0, :3] = (1.0, 0.25, 0.0)
    rgba[1, 1, :3] = (np.nan, 1e9, -0.1)
    out = ob.tonemap_rgb8(rgba)
    # row flip: buffer row 0 (bottom) is the last PPM row (src/lib.rs:620-630)
    exp = [int(255.99 * math.sqrt(1.5 * c / (1.5 * c + 1.0))) for c in (1.0, 0.25, 0.0)]
    assert list(out[1, 0]) == exp
    assert list(out[0, 1]) == [0, 255, 0]      # NaN -> 0, huge -> 255, negative -> sqrt(NaN) -> 0 (`as u8`)
    assert (ptgpu.tonemap_rgb8(rgba) == out).all()


# ------------------------------------------------------------------ estimator
def furnace_scene(a, e, size=100.0):
    """Closed cube of emissive Lambertian walls whose normals all face inwards."""
    s = ptgpu.Scene()
    m = s.lambertian(s.texture_constant((a, a, a)), emissive=e)
    walls = []
    for axis in (ptgpu.AXIS_X, ptgpu.AXIS_Y, ptgpu.AXIS_Z):
        walls.append(s.rect(0, size, 0, size, 0.0, axis, m))                       # normal +axis
        walls.append(s.flip_normals(s.rect(0, size, 0, size, size, axis, m)))      # normal -axis
    s.set_root(s.bvh(walls))
    dm = s.lambertian(s.texture_constant((0, 0, 0)))
    s.set_lights([s.rect(40, 60, 40, 60, size - 1.0, ptgpu.AXIS_Y, dm), s.sphere((30, 30, 30), 10.0, dm)])
    cam = ptgpu.make_camera((50, 50, 10), (50, 50, 90), vfov=60.0)
    return s, cam


def test_furnace_energy_series():
    """Inside a closed emissive Lambertian box every path sees emitted radiance a*e at every vertex, so the
    estimator's mean is a*e*(1 + a + ... + a^D) (src/trace.rs:217-221 with Lambertian::emitted, material.rs:303-305)."""
    a, e, depth = 0.5, 2.0, 6
    s, cam = furnace_scene(a, e)
    img, st = ob.render(s.desc(), cam, 24, 24, 64, depth, threads=8)
    expected = a * e * sum(a ** k for k in range(depth + 1))
    assert st.segments >= 0.99 * st.paths * (depth + 1)     # (almost) nothing escapes: paths run to max_depth
    assert np.nanmean(img[..., 0]) == pytest.approx(expected, rel=0.03)
    assert np.isnan(img[..., :3]).mean() < 0.01


def test_cornell_mean_luminance_is_stable_across_seeds():
    s = ptgpu.Scene.named("cornell_box", 1.0)
    means = []
    for seed in (1, 2):
        img, _ = ob.render(s.desc(), s.camera, 40, 40, 24, 50, seed=seed, threads=8)
        means.append(np.nanmean(img[..., :3]))
    assert means[0] == pytest.approx(means[1], rel=0.12) and 0.05 < means[0] < 1.0


def test_isotropic_medium_quirk_yields_nan_or_zero():
    # Isotropic has no scattering_pdf and a zero-vector DummyPDF (src/material.rs:346-352, 439-445): a medium scatter
    # contributes 0 when the light-sampled branch is taken and NaN when the zero direction is (0 * x / 0).
    s = ptgpu.Scene.named("cornell_smoke", 1.0)
    img, _ = ob.render(s.desc(), s.camera, 32, 32, 8, 10, threads=8)
    assert np.isnan(img[..., :3]).any()


# ------------------------------------------------------------------ golden fixtures
def test_golden_extend_fixture():
    g = np.load(os.path.join(GOLDEN, "cornell_extend_4096.npz"))
    s = ptgpu.Scene.named("cornell_box", 1.0, seed=int(g["seed"]))
    rays = np.zeros(g["origin"].shape[0], dtype=ptgpu.RAY_DTYPE)
    rays["origin"], rays["direction"], rays["time"] = g["origin"], g["direction"], g["time"]
    hits, _ = ob.intersect(s.desc(), rays)
    assert (hits["prim"] == g["prim"]).all()
    both = g["prim"] >= 0
    assert np.array_equal(hits["t"][both], g["t"][both])


def test_golden_image_fixture():
    g = np.load(os.path.join(GOLDEN, "cornell_image_32x32_8spp.npz"))
    s = ptgpu.Scene.named("cornell_box", 1.0, seed=int(g["seed"]))
    img, _ = ob.render(s.desc(), s.camera, 32, 32, 8, 50, seed=int(g["seed"]), threads=4)
    assert np.allclose(img[..., :3], g["rgb"], rtol=1e-6, atol=1e-8, equal_nan=True)


def test_display_transform_closed_forms():
    """ACES-fitted display transform (assets/shaders/tonemap.frag:37-70): black stays black, the curve saturates
    to 1, the vignette is 1 - distance from the screen centre, rows are flipped, alpha is 1."""
    h, w = 6, 8
    img = np.zeros((h, w, 4), dtype=np.float32)
    out = ob.display_transform(img, 1.3)
    assert np.all(out[..., :3] == 0.0) and np.all(out[..., 3] == 1.0)
    img[..., :3] = 1000.0  # far into the shoulder: every channel clamps to 1, leaving the vignette alone
    out = ob.display_transform(img, 1.3)
    jj, ii = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    vig = 1.0 - np.sqrt(((ii + 0.5) / w - 0.5) ** 2 + ((jj + 0.5) / h - 0.5) ** 2)
    assert np.allclose(out[..., 0], vig, atol=2e-6) and np.allclose(out[..., 2], vig, atol=2e-6)
    # a mid grey through the published fit: x = 0.18 * exposure on all channels (the matrices' rows sum to 1)
    img[..., :3] = 0.18
    out = ob.display_transform(img, 1.0)
    x = 0.18
    fit = (x * (x + 0.0245786) - 0.000090537) / (x * (0.983729 * x + 0.4329510) + 0.238081)
    rows = np.array([1.60475 - 0.53108 - 0.07367, -0.10208 + 1.10813 - 0.00605, -0.00327 - 0.07276 + 1.07602])
    centre = out[h // 2, w // 2, :3] / vig[h // 2, w // 2]
    assert np.allclose(centre, fit * rows, rtol=2e-4)
    # row flip: the bottom scanline of the buffer ends up in the last output row
    img[...] = 0.0
    img[0, :, 0] = 0.5
    out = ob.display_transform(img, 1.3)
    assert np.all(out[-1, :, 0] > 0.0) and np.all(out[:-1, :, 0] == 0.0)
