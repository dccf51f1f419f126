"""Groups of contexts (pt_group_*): what one GPU can check. Members beyond one need one GPU each; the multi-rank
row split and the assembly of the image are covered on CPU by tests/test_dist_gloo.py and on hardware by
`bench.py --gpus N`. Here: a group of one behaves like a lone context, and the gather's data path (pack rows ->
ncclSend -> ncclRecv -> scatter rows) returns an image unchanged when a context sends to itself."""
import numpy as np
import pytest

import ptgpu

pytestmark = pytest.mark.gpu


def _rendered(w, h):
    scene = ptgpu.Scene.named("cornell_box", w / h)
    ctx = ptgpu.Context(device=0, seed=7, max_paths_in_flight=1 << 18)
    ctx.upload(scene)
    ctx.render(w, h, 2, 10)
    return ctx, ctx.readback().copy()


@pytest.mark.parametrize("w,h", [(64, 48), (33, 7), (16, 1)])
def test_group_of_one_and_loopback(w, h):
    ctx, before = _rendered(w, h)
    assert np.isfinite(before[..., :3]).all() and before[..., :3].max() > 0
    with pytest.raises(ptgpu.PtError):
        ctx.group_loopback()                      # not in a group yet
    ident = ptgpu.group_unique_id()
    assert len(ident) == ptgpu.PT_GROUP_ID_BYTES and any(ident)
    ctx.group_join(ident, 0, 1)
    with pytest.raises(ptgpu.PtError):
        ctx.group_join(ident, 0, 1)               # already a member
    assert ctx.group_rows() == (1, 0)
    ctx.group_gather(0)                           # alone: nothing moves
    assert np.array_equal(ctx.readback(), before, equal_nan=True)
    with pytest.raises(ptgpu.PtError):
        ctx.group_gather(1)                       # no such member
    ctx.group_loopback()                          # odd rows: pack, wipe, send to self, receive, scatter
    assert np.array_equal(ctx.readback(), before, equal_nan=True)
    ctx.group_leave()
    assert ctx.group_rows() == (1, 0)
    ctx.close()


def test_local_group_of_one():
    ctx, before = _rendered(40, 24)
    ptgpu.group_join_local([ctx])
    assert ctx.group_rows() == (1, 0)
    ptgpu.group_gather_local([ctx], 0)
    ctx.group_loopback()
    assert np.array_equal(ctx.readback(), before, equal_nan=True)
    other = ptgpu.Context(device=0, seed=7, max_paths_in_flight=1 << 12)
    with pytest.raises(ptgpu.PtError):            # RCCL takes one rank per device
        ptgpu.group_join_local([other, ptgpu.Context(device=0, seed=7, max_paths_in_flight=1 << 12)])
    other.close()
    ctx.close()                                   # leaves the group on destruction


def test_interleaved_rows_of_a_virtual_group_compose():
    """What members of a group of three would render (row_step 3, row_phase r), done one after the other on one GPU
    into one image, is the image a lone context renders."""
    w, h = 48, 31
    scene = ptgpu.Scene.named("cornell_box", w / h)
    ctx = ptgpu.Context(device=0, seed=11, max_paths_in_flight=1 << 16)
    ctx.upload(scene)
    ctx.render(w, h, 3, 12)
    whole = ctx.readback().copy()
    ctx.reset()
    for r in range(3):
        ctx.render(w, h, 3, 12, rows=(3, r))
    assert np.array_equal(ctx.readback(), whole, equal_nan=True)
    ctx.close()


@pytest.mark.parametrize("form", ["loopback", "lone"])
@pytest.mark.parametrize("w,h", [(96, 64), (33, 7)])
def test_asynchronous_gather_frame_sequence(form, w, h):
    """pt_group_gather_async in a realtime frame loop on one GPU: frame k is queued for gathering and frame k + 1 is
    traced at once, with no host synchronisation in between; the frame the root's readers then see must be frame k, bit
    for bit, not a mixture. `loopback`: a virtual group of two (odd rows travel from the context to itself through RCCL
    on the second stream); `lone`: a context alone (a group of one: its own rows are packed and assembled, nothing is
    sent). The reference sequence comes from a second context that reads every frame back before it traces the next."""
    scene = ptgpu.Scene.named("cornell_box", w / h)
    frames = 6
    ref_ctx = ptgpu.Context(device=0, seed=21, max_paths_in_flight=1 << 16)
    ref_ctx.upload(scene)
    want = []
    for k in range(frames):
        ref_ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=k)
        want.append(ref_ctx.readback().copy())
    ref_ctx.close()
    assert not np.array_equal(want[0], want[1])  # progressive frames differ, so a stale or mixed frame would show

    ctx = ptgpu.Context(device=0, seed=21, max_paths_in_flight=1 << 16)
    ctx.upload(scene)
    if form == "loopback":
        ctx.group_join(ptgpu.group_unique_id(), 0, 1)
        gather = ctx.group_loopback_async
    else:
        gather = ctx.group_gather_async
    got = []
    ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=0)
    for k in range(frames):
        gather()                                                       # frame k: queued, not waited for
        if k + 1 < frames:
            ctx.render(w, h, 1, 10, mode=ptgpu.PT_MODE_REALTIME, frame=k + 1)  # traced while frame k's rows travel
        got.append(ctx.readback().copy())                              # the assembled frame k (the image holds k + 1 by now)
    ctx.group_wait()
    for k in range(frames):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), f"frame {k}"
    # the display transform and the presentation ring read the assembled frame too
    ring = ctx.present_attach(None, slots=2, width=w, height=h)
    ctx.present_submit(0)
    ctx.present_wait(0)
    assert np.array_equal(ring[0].view(np.uint32), want[-1].view(np.uint32))
    ctx.present_detach()
    # the synchronous gather puts the readers back on the context's own image
    ctx.group_gather(0)
    assert np.array_equal(ctx.readback().view(np.uint32), want[-1].view(np.uint32))
    if form == "loopback":
        ctx.group_leave()
    ctx.close()


def test_bench_multi_rank_rehearsal():
    """bench.py --gpus 2 on a one-GPU box: both ranks on GPU 0 over gloo (PTGPU_BENCH_REHEARSAL=1; RCCL takes one rank
    per device, so the gather goes through the host). The launcher path, the row split, the gather and the line's
    fields are the ones of a real two-GPU run; the numbers mean nothing."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PTGPU_BENCH_REHEARSAL"] = "1"
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--width", "96", "--height", "64",
                           "--spp", "4", "--steps", "2", "--warmup", "1", "--paths-in-flight", "65536", "--no-exact"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["steps"] == 2
    assert line["value"] > 0 and line["dtype"] == "f64" and line["roofline"]["bound"] == "fp64_valu"
    assert line["config"]["nan_pixel_fraction"] == 0.0
    assert line["gather"] == "torch.distributed.gather"  # the line says which gather moved the rows
