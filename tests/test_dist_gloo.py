"""Multi-process path on CPU: two ranks over the gloo backend each produce their stripes of the image and the
stripes are gathered to rank 0 with the same code bench.py uses over RCCL. The per-rank "render" is the CPU oracle
restricted to the rank's tile rectangles, so the assembled image must equal a single-process render bit for bit
(the RNG is keyed by global pixel and sample, never by rank)."""
import importlib
import os
from ctypes import c_uint32 as C_uint32, c_uint64 as C_uint64
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, ROWS = 24, 40, 2, 8, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_rows(rank, world, port, out_path):
    """The split bench.py uses: rows interleaved between ranks, one render call and one gather per rank."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pt = importlib.import_module("path-tracer_amd")
    tiles = importlib.import_module("path-tracer_amd.tiles")
    import oracle_binding as ob

    scene = pt.Scene.named("cornell_box", W / H)
    image = np.zeros((H, W, 4), dtype=np.float32)
    ob.render(scene.desc(), scene.camera, W, H, SPP, DEPTH, threads=1, image=image, rows=tiles.rows_for(rank, world))
    # only this rank's rows were written
    others = np.ones(H, dtype=bool)
    others[rank::world] = False
    assert not image[others].any()
    t = torch.from_numpy(image)
    tiles.gather_rows(t, rank, world, dist)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pt = importlib.import_module("path-tracer_amd")
    tiles = importlib.import_module("path-tracer_amd.tiles")
    import oracle_binding as ob

    scene = pt.Scene.named("cornell_box", W / H)
    image = np.zeros((H, W, 4), dtype=np.float32)
    for (y0, y1) in tiles.stripes_for(rank, world, H, ROWS):
        ob.render(scene.desc(), scene.camera, W, H, SPP, DEPTH, tile=(0, y0, W, y1), threads=1, image=image)
    t = torch.from_numpy(image)
    tiles.gather_stripes(t, rank, world, dist, rows=ROWS)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stripe_gather_matches_single_process(tmp_path, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import ptgpu

    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    scene = ptgpu.Scene.named("cornell_box", W / H)
    ref, _ = ob.render(scene.desc(), scene.camera, W, H, SPP, DEPTH, threads=1)
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.nanmax(ref) > 0


@pytest.mark.parametrize("world", [2, 3])
def test_interleaved_rows_gather_matches_single_process(tmp_path, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    import ptgpu

    out = str(tmp_path / "gathered_rows.npy")
    mp.spawn(_worker_rows, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    scene = ptgpu.Scene.named("cornell_box", W / H)
    ref, _ = ob.render(scene.desc(), scene.camera, W, H, SPP, DEPTH, threads=1)
    assert np.array_equal(got, ref, equal_nan=True)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`bench.py --gpus 2` run bare (no launcher, WORLD_SIZE unset) starts two ranks itself. Without a GPU each rank
    stops at the device check, after the world size has been checked against --gpus."""
    import subprocess

    if torch.cuda.is_available():  # before anything is started: with a GPU this would be a real two-rank benchmark
        pytest.skip("a GPU is present: covered by the rehearsal test")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    small = ["--width", "64", "--height", "64", "--spp", "1", "--paths-in-flight", str(1 << 16), "--no-exact", "--no-extend",
             "--no-cpu-baseline", "--no-workloads"]
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"] + small,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert proc.returncode != 0
    assert proc.stdout.count("bench.py needs an MI355X") >= 2, proc.stdout[-2000:]
    # a launcher that starts the wrong number of ranks is refused
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, env=env2, timeout=300)
    assert proc.returncode != 0 and "WORLD_SIZE=4" in proc.stdout


@pytest.mark.parametrize("world", range(2, 9))
def test_gather_layout_covers_every_row_once(world):
    """The row arithmetic the gather's kernels run (csrc/group_rows.h through pt_group_layout / pt_group_staged_row),
    replayed on the host: every member packs its rows, the root's staging buffer is their concatenation at the offsets
    the root posts its receives at, and the scatter must find every image row exactly where its owner put it. Heights
    the world size does not divide, and heights below it, included."""
    import ptgpu

    for height in (1, 2, 3, 5, 7, 8, 9, 16, 31, 64, 100, 1080):
        for root in sorted({0, world - 1, world // 2}):
            rows_of, before = zip(*[ptgpu.group_layout(height, world, root, r) for r in range(world)])
            assert sum(rows_of) == height
            assert [rows_of[r] for r in range(world)] == [len(range(r, height, world)) for r in range(world)]
            # the root's receive offsets: members in rank order, the root itself left out, nothing overlapping
            expect = 0
            for r in range(world):
                assert before[r] == expect
                if r != root:
                    expect += rows_of[r]
            staged = np.full(expect, -1, dtype=np.int64)  # image row held by each staging-buffer row
            for r in range(world):
                if r == root:
                    continue
                packed = np.arange(r, height, world)  # k_pack_rows: row k of the buffer = image row rank + k * world
                staged[before[r]:before[r] + rows_of[r]] = packed
            assert (staged >= 0).all()
            for y in range(height):  # k_unpack_rows
                idx = ptgpu.lib.pt_group_staged_row(height, world, root, y)
                if y % world == root:
                    assert idx == -1
                else:
                    assert 0 <= idx < expect and staged[idx] == y
    rows, before = C_uint32(), C_uint64()
    assert ptgpu.lib.pt_group_layout(8, world, world, 0, rows, before) == ptgpu.PT_ERR_INVALID_ARG  # root outside the group
