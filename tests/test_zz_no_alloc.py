"""pt_intersect_static, the entry point that allocates no device memory, against pt_intersect.

Bar: the two entry points run the same kernels on the same rays, so every field of every hit record is bit-identical
(prim, material, t, u, v, p, normal; NaN payloads included). Two calls in a row share the GPU's fixed scratch array;
the second is smaller, so stale entries of the first call lie right behind its own.
"""
import numpy as np
import pytest

import oracle_binding as ob
import ptgpu

pytestmark = pytest.mark.gpu


def _batch(scene, w, h):
    pix = np.arange(w * h, dtype=np.uint64)
    primary = ob.camera_rays(scene.camera, w, h, pix, np.zeros(w * h, dtype=np.uint32))
    hits, _ = ob.intersect(scene.desc(), primary)
    secondary, valid = ob.secondary_rays(hits, primary)
    return np.concatenate([primary, secondary[valid]])


def _on_device(torch, host_bytes):
    return torch.from_numpy(host_bytes.view(np.uint8).copy()).to("cuda:0")


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke", "random_scene"])
def test_static_entry_point_matches_pt_intersect(mode, name, gpu_ctx, gpu_ctx_exact):
    import torch

    ctx = gpu_ctx if mode == "fast" else gpu_ctx_exact
    scene = ptgpu.Scene.named(name, 1.0)
    ctx.upload(scene)
    rays_all = _batch(scene, 96, 96)
    assert rays_all.shape[0] > 12000
    for n in (12000, 5000, 1, 64, 4097):  # each call after the first is smaller or ragged against the tile size
        rays = np.ascontiguousarray(rays_all[:n])
        want = ctx.intersect(rays, medium_seed=7)
        d_rays = _on_device(torch, rays)
        d_hits = torch.full((n * ptgpu.HIT_DTYPE.itemsize,), 0xA5, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        # the property the entry point exists for: the library allocates nothing, and the device's free memory does not move
        allocs, free_before = ptgpu.device_alloc_count(), torch.cuda.mem_get_info(0)[0]
        ctx.intersect_static(d_rays.data_ptr(), n, d_hits.data_ptr(), medium_seed=7)
        assert ptgpu.device_alloc_count() == allocs
        assert torch.cuda.mem_get_info(0)[0] == free_before
        got = d_hits.cpu().numpy().view(ptgpu.HIT_DTYPE)
        assert got.shape == want.shape
        assert (got["prim"] == want["prim"]).all() and (got["material"] == want["material"]).all()
        for field in ("t", "u", "v", "p", "normal"):
            assert (got[field].view(np.uint64) == want[field].view(np.uint64)).all(), field
        # and both agree with the oracle on which primitive is hit
        ref, _ = ob.intersect(scene.desc(), rays, medium_seed=7)
        assert (got["prim"] == ref["prim"]).mean() >= 0.9999


def test_static_entry_point_limits(gpu_ctx):
    import torch

    scene = ptgpu.Scene.named("cornell_box", 1.0)
    gpu_ctx.upload(scene)
    n = ptgpu.PT_INTERSECT_STATIC_MAX_RAYS
    rays = np.zeros(n + 1, dtype=ptgpu.RAY_DTYPE)
    rays["origin"] = (278.0, 278.0, -800.0)
    rays["direction"] = (0.0, 0.0, 1.0)
    d_rays = _on_device(torch, rays)
    d_hits = torch.zeros(((n + 1) * ptgpu.HIT_DTYPE.itemsize,), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    with pytest.raises(ptgpu.PtError) as e:  # one ray over the maximum: refused with a clear message, nothing runs
        gpu_ctx.intersect_static(d_rays.data_ptr(), n + 1, d_hits.data_ptr())
    assert e.value.code == ptgpu.PT_ERR_TOO_LARGE and "PT_INTERSECT_STATIC_MAX_RAYS" in str(e.value)
    allocs = ptgpu.device_alloc_count()
    gpu_ctx.intersect_static(d_rays.data_ptr(), n, d_hits.data_ptr())  # the maximum itself fits
    assert ptgpu.device_alloc_count() == allocs > 0  # ... without an allocation (the counter saw pt_create's)
    got = d_hits.cpu().numpy().view(ptgpu.HIT_DTYPE)[:n]
    want = gpu_ctx.intersect(rays[:n])
    assert (got["prim"] == want["prim"]).all() and (got["t"] == want["t"]).all()
    assert (got["prim"] >= 0).all()
    # host memory where device memory is expected is an error, not a GPU fault
    with pytest.raises(ptgpu.PtError) as e:
        gpu_ctx.intersect_static(rays.ctypes.data, 16, d_hits.data_ptr())
    assert e.value.code == ptgpu.PT_ERR_INVALID_ARG
    # a device buffer that is too short for n entries likewise
    # (torch serves small tensors from 2-MiB segments; 2^15 hit records are 2.5 MiB)
    short = torch.zeros((8 * ptgpu.HIT_DTYPE.itemsize,), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(ptgpu.PtError):
        gpu_ctx.intersect_static(d_rays.data_ptr(), 1 << 15, short.data_ptr())
    gpu_ctx.intersect_static(d_rays.data_ptr(), 0, d_hits.data_ptr())  # an empty batch is a no-op
