"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (from the north star): fixed ray batches agree on the hit primitive for >= 99.99 % of rays with hit t within
1e-4 relative; images agree within a stated RMSE. Because the device computes in f64 with the reference's operation
order, the observed agreement is far tighter and the tests also pin that.
"""
import numpy as np
import pytest

import oracle_binding as ob
import ptgpu

pytestmark = pytest.mark.gpu

SCENES_SMALL = ["cornell_box", "cornell_smoke", "random_scene", "final_book_two", "two_spheres", "four_spheres",
                "two_perlin_spheres", "textured_sphere", "simple_light"]


def _ray_batches(scene, w, h):
    pix = np.arange(w * h, dtype=np.uint64)
    smp = np.zeros(w * h, dtype=np.uint32)
    primary = ob.camera_rays(scene.camera, w, h, pix, smp)
    hits, _ = ob.intersect(scene.desc(), primary)
    secondary, valid = ob.secondary_rays(hits, primary)
    return primary, secondary[valid]


def _compare_hits(gpu, ref):
    same = gpu["prim"] == ref["prim"]
    frac = same.mean()
    both = same & (ref["prim"] >= 0)
    rel = np.abs(gpu["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-300)
    return frac, (rel.max() if rel.size else 0.0), both


@pytest.mark.parametrize("name", SCENES_SMALL)
def test_extend_parity(any_ctx, name):
    gpu_ctx, exact = any_ctx
    scene = ptgpu.Scene.named(name, 1.0)
    gpu_ctx.upload(scene)
    w = h = 64
    for batch in _ray_batches(scene, w, h):
        ref, _ = ob.intersect(scene.desc(), batch, medium_seed=11)
        got = gpu_ctx.intersect(batch, medium_seed=11)
        frac, max_rel, both = _compare_hits(got, ref)
        assert frac >= 0.9999, f"{name}: primitive ids agree on {frac:.6f} of rays"
        assert max_rel <= 1e-4, f"{name}: max relative t error {max_rel:.3e}"
        # f64 device arithmetic: the records agree to rounding, not just to the bar. Exact mode repeats the
        # reference's operation order; fast mode differs by a few ulp per operation (cancellation in the
        # sphere discriminant amplifies that for grazing hits).
        assert max_rel <= (1e-12 if exact else 1e-7), f"{name}: max relative t error {max_rel:.3e}"
        for field in ("p", "normal"):
            assert np.allclose(got[field][both], ref[field][both], rtol=1e-6, atol=1e-6), field
            if exact:
                assert np.allclose(got[field][both], ref[field][both], rtol=1e-9, atol=1e-9), field
        uv_ok = np.isclose(got["u"][both], ref["u"][both], rtol=1e-9, atol=1e-9, equal_nan=True) & \
                np.isclose(got["v"][both], ref["v"][both], rtol=1e-9, atol=1e-9, equal_nan=True)
        assert uv_ok.mean() >= 0.9999
        assert (got["material"][both] == ref["material"][both]).all()


@pytest.mark.parametrize("name,spp,depth", [("cornell_box", 4, 50), ("random_scene", 2, 50), ("cornell_smoke", 2, 50),
                                            ("final_book_two", 2, 10), ("simple_light", 2, 10)])
def test_render_parity_small(any_ctx, name, spp, depth):
    gpu_ctx, exact = any_ctx
    scene = ptgpu.Scene.named(name, 1.0)
    gpu_ctx.upload(scene)
    w = h = 48
    gpu_ctx.render(w, h, spp, depth)
    got = gpu_ctx.readback()[..., :3].astype(np.float64)
    ref, st = ob.render(scene.desc(), scene.camera, w, h, spp, depth, seed=gpu_ctx.seed, threads=8)
    ref = ref[..., :3].astype(np.float64)
    nan_same = (np.isnan(got) == np.isnan(ref)).mean()
    finite = np.isfinite(got) & np.isfinite(ref)
    # same RNG streams, f64 geometry: nearly every pixel follows the same paths as the oracle. Exact mode agrees
    # to 1e-5 per channel; fast mode samples directions with f32 sin/cos, so its radiance differs by ~1e-6 relative
    # per bounce and a few more paths flip a hit/miss decision.
    close = np.isclose(got, ref, rtol=1e-5 if exact else 2e-3, atol=1e-7 if exact else 1e-5) | ~finite
    assert nan_same >= 0.995, f"{name}: NaN masks agree on {nan_same:.4f}"
    assert close.mean() >= (0.99 if exact else 0.97), f"{name}: {close.mean():.4f} of channels agree"
    rmse = np.sqrt(np.mean((got[finite] - ref[finite]) ** 2))
    scale = max(np.sqrt(np.mean(ref[finite] ** 2)), 1e-12)
    assert rmse / scale <= 0.05, f"{name}: relative RMSE {rmse / scale:.3e}"


def test_segments_match_oracle(gpu_ctx):
    scene = ptgpu.Scene.named("cornell_box", 1.0)
    gpu_ctx.upload(scene)
    gpu_ctx.stats_reset()
    gpu_ctx.render(32, 32, 4, 50)
    st = gpu_ctx.stats()
    _, ost = ob.render(scene.desc(), scene.camera, 32, 32, 4, 50, seed=gpu_ctx.seed, threads=4)
    assert st.paths == ost.paths == 32 * 32 * 4
    assert abs(int(st.segments) - int(ost.segments)) <= max(4, ost.segments // 1000)
