"""Presentation ring (pt_present_*): frames copied into the host's staging memory behind the render stream, as the
main loop hands SceneOutput.buffer to the CopyToImage node (src/lib.rs:496-509, src/node/copy_image.rs:104-165).
Every frame that lands in a slot must be, bit for bit, what pt_readback / pt_display_transform return for that frame."""
import numpy as np
import pytest

import ptgpu

pytestmark = pytest.mark.gpu

W, H = 160, 96


def _fresh(seed=0x5EED0001):
    scene = ptgpu.Scene.named("cornell_box", W / H)
    ctx = ptgpu.Context(device=0, seed=seed, max_paths_in_flight=1 << 18)
    ctx.upload(scene)
    return scene, ctx


@pytest.mark.parametrize("owner", ["library", "caller"])
def test_frames_land_in_their_slots_while_the_next_is_traced(owner):
    """Progressive realtime frames: frame k is submitted to slot k % 2 and the next frame is traced at once; a second
    context replays the same frames and reads each one back synchronously. The ring is the library's page-locked
    allocation, or the caller's own memory."""
    _, ctx = _fresh()
    _, ref = _fresh()
    mine = np.full((2, H, W, 4), np.nan, dtype=np.float32)
    frames = []
    try:
        if owner == "library":
            ring = ctx.present_attach(None, slots=2, width=W, height=H)
            assert ctx.present_pinned()
        else:
            ring = ctx.present_attach(mine)
        assert ring.shape == (2, H, W, 4)
        for k in range(5):
            ctx.render(W, H, spp=1, max_depth=10, mode=ptgpu.PT_MODE_REALTIME, frame=k)
            ctx.present_submit(k % 2)
            ref.render(W, H, spp=1, max_depth=10, mode=ptgpu.PT_MODE_REALTIME, frame=k)
            frames.append(ref.readback())
            if k >= 1:  # the frame before this one has had a whole render to cross PCIe; now it is needed
                ctx.present_wait((k - 1) % 2)
                assert np.array_equal(ring[(k - 1) % 2], frames[k - 1], equal_nan=True), f"frame {k - 1}"
        ctx.present_wait(4 % 2)
        assert np.array_equal(ring[0], frames[4], equal_nan=True)
        assert not np.array_equal(frames[3], frames[4])  # the frames do differ: the test can tell slots apart
        # a slot that was never submitted after a detach / attach cycle is ready at once
        ctx.present_detach()
        ctx.present_attach(mine)
        ctx.present_wait(1)
    finally:
        ctx.close()
        ref.close()


def test_display_kind_matches_the_display_transform():
    _, ctx = _fresh()
    ring = np.zeros((1, H, W, 4), dtype=np.float32)
    try:
        ctx.render(W, H, spp=4, max_depth=50)
        ctx.present_attach(ring)
        ctx.present_submit(0, kind=ptgpu.PT_PRESENT_DISPLAY, exposure=1.7)
        ctx.present_wait(0)
        assert np.array_equal(ring[0], ctx.display_transform(1.7))
        ctx.present_submit(0)  # raw, into the same slot
        ctx.present_wait(0)
        assert np.array_equal(ring[0], ctx.readback(), equal_nan=True)
    finally:
        ctx.close()


def test_argument_errors():
    _, ctx = _fresh()
    small = np.zeros((2, 8, 8, 4), dtype=np.float32)
    try:
        with pytest.raises(ptgpu.PtError):
            ctx.present_submit(0)  # nothing attached
        ctx.present_attach(small)
        with pytest.raises(ptgpu.PtError):
            ctx.present_attach(small)  # attached already
        with pytest.raises(ptgpu.PtError):
            ctx.present_submit(0)  # no image yet
        ctx.render(W, H, spp=1, max_depth=5)
        with pytest.raises(ptgpu.PtError) as e:
            ctx.present_submit(0)  # the image does not fit a slot
        assert e.value.code == ptgpu.PT_ERR_TOO_LARGE
        with pytest.raises(ptgpu.PtError):
            ctx.present_submit(2)  # no such slot
        ctx.present_detach()
        ctx.present_detach()  # idempotent
        with pytest.raises(ptgpu.PtError):
            ctx.present_attach(np.zeros((5, 8, 8, 4), dtype=np.float32))  # more than four slots
    finally:
        ctx.close()


def test_eight_bit_kinds_are_the_display_transform_quantised():
    """PT_PRESENT_DISPLAY8: every byte is floor(255 v + 0.5) of pt_display_transform's f32 value, alpha 255, a quarter of
    the bytes (a slot of width x height x 4 bytes suffices). PT_PRESENT_DISPLAY8_SRGB: the same after the sRGB transfer
    function; its power function is evaluated in f32 on both sides, so a byte may differ by one level where the two
    roundings of the power straddle a level boundary (bar: at most one level, on at most 0.1 % of the bytes)."""
    _, ctx = _fresh()
    try:
        ctx.render(W, H, spp=8, max_depth=50)
        ctx.present_attach(None, slots=2, width=W // 2, height=H // 2)  # a quarter of an f32 frame per slot
        with pytest.raises(ptgpu.PtError) as e:
            ctx.present_submit(0, kind=ptgpu.PT_PRESENT_DISPLAY)  # the f32 kinds do not fit such a slot
        assert e.value.code == ptgpu.PT_ERR_TOO_LARGE
        want = ctx.display_transform(1.3).astype(np.float32)
        ctx.present_submit(0, kind=ptgpu.PT_PRESENT_DISPLAY8, exposure=1.3)
        ctx.present_submit(1, kind=ptgpu.PT_PRESENT_DISPLAY8_SRGB, exposure=1.3)
        ctx.present_wait(0)
        ctx.present_wait(1)
        got = ctx.present_slot_rgba8(0, W, H).copy()
        quantised = np.floor(np.clip(want, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
        assert np.array_equal(got, quantised)
        assert (got[..., 3] == 255).all() and got[..., :3].max() > 100
        srgb = ctx.present_slot_rgba8(1, W, H).copy()
        c = np.clip(want[..., :3], 0.0, 1.0).astype(np.float32)
        curve = np.where(c <= np.float32(0.0031308), np.float32(12.92) * c,
                         np.float32(1.055) * np.power(c, np.float32(1.0 / 2.4), dtype=np.float32) - np.float32(0.055))
        want8 = np.floor(np.clip(curve, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.int32)
        diff = np.abs(srgb[..., :3].astype(np.int32) - want8)
        assert diff.max() <= 1 and (diff > 0).mean() <= 1e-3
        assert (srgb[..., 3] == 255).all()
    finally:
        ctx.close()
