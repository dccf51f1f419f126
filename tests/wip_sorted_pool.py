"""WIP: the regrouped path pool against the plain one."""
import time
import numpy as np
import pytest
import ptgpu

pytestmark = pytest.mark.gpu


def run(scene, w, h, spp, budget, depth=50, **tuning):
    ctx = ptgpu.Context(device=0, max_paths_in_flight=budget, **tuning)
    ctx.upload(scene)
    ctx.render(w, h, spp, depth)
    t0 = time.perf_counter()
    ctx.render(w, h, spp, depth)
    dt = time.perf_counter() - t0
    img, st = ctx.readback(), ctx.stats()
    ctx.close()
    return img, st, dt


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke"])
def test_sorted_small(name):
    scene = ptgpu.Scene.named(name, 1.0)
    for (w, h, spp) in [(1, 1, 1), (7, 3, 5), (65, 2, 3), (64, 64, 4), (256, 1, 257)]:
        a, sa, _ = run(scene, w, h, spp, 1 << 20)
        b, sb, _ = run(scene, w, h, spp, 1 << 20, pass_form=ptgpu.PT_PASS_PATH_POOL_SORTED)
        assert np.array_equal(a, b, equal_nan=True), (w, h, spp)
        assert int(sa.segments) == int(sb.segments)


def test_sorted_big():
    scene = ptgpu.Scene.named("cornell_box", 1.0)
    a, sa, ta = run(scene, 1024, 1024, 64, 1 << 26)
    b, sb, tb = run(scene, 1024, 1024, 64, 1 << 26, pass_form=ptgpu.PT_PASS_PATH_POOL_SORTED)
    print(f"\nplain {1024*1024*64/ta/1e6:.0f} Mpaths/s, sorted {1024*1024*64/tb/1e6:.0f} Mpaths/s")
    assert np.array_equal(a, b, equal_nan=True)
    assert int(sa.segments) == int(sb.segments)
