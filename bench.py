#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s (pixels x spp / s) of the wavefront path tracer on the Cornell box,
1024 x 1024 @ 1024 spp, max_depth 50 (the offline configuration of the reference: src/lib.rs:128-135),
on N GPUs of one node.

A step is one full render of the image. With N > 1 the scanlines are dealt to the ranks (one process per GPU) by
interleaving: rank r traces rows r, r + N, ... into its own copy of the f32 RGBA image in one pt_render call, and the
rows are gathered to rank 0 with one RCCL gather (torch.distributed, backend "nccl") inside the timed region.
The scene and the RNG are keyed by global pixel and sample index, so the image does not depend on N.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
SEED = 0x5EED0001


def algorithmic_bytes(V, P):
    """Per-unit HBM bytes of the compact layout the roofline is priced on: a ray is 28 B (f32 origin, direction,
    time), a hit 8 B, a BVH node / primitive / material record 32 B, a queue entry 48 B (ray + pixel + throughput).
    extend: read ray, write hit, fetch V nodes + P primitives. shade: read queue entry + hit, fetch primitive +
    material, write the next queue entry; together B_seg = 204 + 32 (V + P).
    The fused trace+shade kernel does both on one queue entry and never writes or re-reads the hit. It only runs for
    scenes whose whole geometry (a few KB) stays on chip, so no node or primitive fetch has to come from HBM: what
    must cross HBM per segment is the queue entry in, the primitive + material records, and the queue entry out."""
    b_ray = 28 + 8 + 32.0 * (V + P)
    b_shade = 48 + 8 + 32 + 32 + 48
    b_fused = 48 + 32 + 32 + 48
    return b_ray, b_shade, b_fused


def cpu_baseline(scene, width, height, max_depth, target_seconds=12.0, name="cornell_box"):
    """Times the CPU oracle (f64 restatement of the reference, tile jobs on a thread pool) on a bounded sample
    of the same workload: the same scene and image size at a reduced spp."""
    import oracle_binding as ob

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    desc = scene.desc()
    # calibrate on a 1-spp pass, then size the sample for ~target_seconds
    _, st = ob.render(desc, scene.camera, width, height, 1, max_depth, seed=SEED, threads=threads)
    rate = st.paths / max(st.seconds, 1e-9)
    spp = int(max(1, min(64, round(target_seconds * rate / (width * height)))))
    _, st = ob.render(desc, scene.camera, width, height, spp, max_depth, seed=SEED, threads=threads)
    return {
        "value": st.paths / st.seconds / 1e6,
        "unit": "Mpaths/s",
        "cores": int(st.threads),
        "kind": "port",
        "sample": f"{name} {width}x{height} @ {spp} spp, max_depth {max_depth}, {st.seconds:.1f} s "
                  f"(f64 recursive oracle, {st.tiles} tile jobs)",
        "S": st.segments / st.paths,
        "V": st.node_visits / st.segments,
        "P": st.prim_tests / st.segments,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--scene-param", type=int, default=0, help="recipe parameter (sphere_cloud: sphere count)")
    ap.add_argument("--paths-in-flight", type=int, default=1 << 28)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-mode side measurement")
    ap.add_argument("--realtime", action="store_true",
                    help="progressive frames instead of the offline render (src/lib.rs:118-123: spp 1, depth 10, "
                         "80 %% pixel skip, running mean); pass --width 1920 --height 1080 --steps <frames>")
    args = ap.parse_args()
    if args.realtime:
        args.spp, args.max_depth = 1, 10
        args.no_cpu_baseline = args.no_exact = True

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # PTGPU_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo, for rehearsing the multi-rank code on a one-GPU box
    # (the numbers mean nothing; RCCL refuses two ranks on one device). Never set by the driver.
    rehearsal = os.environ.get("PTGPU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pt = importlib.import_module("path-tracer_amd")
    pt.register_earthmap()  # texels of the image texture (tests/golden/make_earthmap.py)
    tiles = importlib.import_module("path-tracer_amd.tiles")
    W, H = args.width, args.height
    scene = pt.Scene.named(args.scene, W / H, seed=SEED, param=args.scene_param)
    ctx = pt.Context(device=local_rank, seed=SEED, max_paths_in_flight=args.paths_in_flight,
                     flags=pt.PT_CFG_PROFILE_STAGES)
    ctx.upload(scene)

    # The image lives in a torch tensor so that RCCL can move it; the library renders straight into it.
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    ctx.bind_image(image.data_ptr(), W, H)
    pack = gather_list = None
    if world > 1:
        pack = torch.zeros((tiles.max_rows(world, H), W, 4), dtype=torch.float32, device=image.device)
        if rank == 0:
            gather_list = [torch.zeros_like(pack) for _ in range(world)]

    def step(frame):
        mode = pt.PT_MODE_REALTIME if args.realtime else pt.PT_MODE_OFFLINE
        if world == 1:
            ctx.render(W, H, args.spp, args.max_depth, frame=frame, mode=mode)
            return
        ctx.render(W, H, args.spp, args.max_depth, frame=frame, mode=mode, rows=tiles.rows_for(rank, world))
        if rehearsal:  # gloo gathers host tensors
            host = image.cpu()
            tiles.gather_rows(host, rank, world, dist)
            image.copy_(host)
        else:
            tiles.gather_rows(image, rank, world, dist, pack=pack, gather_list=gather_list)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and not rehearsal:
        # communicator set-up (seconds the first time) must not land in the timed region even with --warmup 0
        tiles.gather_rows(image, rank, world, dist, pack=pack, gather_list=gather_list)
    for w in range(args.warmup):
        step(w)
    ctx.stats_reset()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else image.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = ctx.stats()
    exact_value = None
    if rank == 0 and world == 1 and not args.no_exact:
        # the same workload in exact mode (IEEE division, no contraction, every BVH box test): reported, not `value`
        ctx_x = pt.Context(device=local_rank, seed=SEED, max_paths_in_flight=args.paths_in_flight, flags=pt.PT_CFG_EXACT)
        ctx_x.upload(scene)
        spp_x = max(1, min(args.spp, 32))
        ctx_x.render(W, H, spp_x, args.max_depth, frame=0)  # warm-up
        t_x = time.perf_counter()
        ctx_x.render(W, H, spp_x, args.max_depth, frame=1)
        exact_value = float(W) * H * spp_x / (time.perf_counter() - t_x) / 1e6
        ctx_x.close()
    # a realtime frame traces the pixels that pass the 80 % skip (src/trace.rs:144): 20 % of them on average
    paths_total = float(W) * H * args.spp * args.steps * (0.2 if args.realtime else 1.0)
    value = paths_total / elapsed / 1e6
    result = None
    if rank == 0:
        nan_frac = float(torch.isnan(image[..., :3]).any(dim=-1).float().mean().item())
        cpu = None
        V, P = 6.8, 8.8  # Cornell box counters from the oracle; replaced by the live ones below
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scene, W, H, args.max_depth, name=args.scene)
            V, P = cpu["V"], cpu["P"]
        b_ray, b_shade, b_fused = algorithmic_bytes(V, P)
        n_launch = max(int(st.profiled_bounce_launches), 1)
        seg = float(st.profiled_segments)
        ext_ms, shd_ms = st.ms_extend, st.ms_shade
        # no extend launches: the scene is a short primitive list and every bounce is one fused trace+shade launch
        fused = int(st.launches_extend) == 0
        if fused:
            dominant, dom_ms, dom_bytes_unit = "shade_fused", shd_ms, b_fused
        else:
            dominant = "extend" if ext_ms >= shd_ms else "shade"
            dom_ms = max(ext_ms, shd_ms)
            dom_bytes_unit = b_ray if dominant == "extend" else b_shade
        algorithmic = seg * dom_bytes_unit
        # bounce 0 of these passes runs in k_first_bounce (camera sample + walk + shade in one launch); the library
        # times it apart and leaves its segments out of profiled_segments, so the figures below are k_shade's alone
        gen0 = fused and int(st.launches_raygen) == 0
        # one launch per pass and no ray generation launch: the path pool (k_path_pool), whose paths live in registers
        # from camera sample to last vertex. It is priced like the fused kernel, at the queue entry in and out and the
        # primitive + material record per segment that a wavefront formulation has to move; what it really moves is
        # `traffic` (a radiance value per PATH), far below that.
        form = int(st.last_pass_form)
        pool = form == pt.PT_PASS_PATH_POOL
        tree_pool = form == pt.PT_PASS_TREE_POOL
        if pool:
            gen0 = False
            dominant = "path_pool"
        if tree_pool:
            # tree scenes in one launch per pass: extend and shade of the same segment happen in the same lane, so
            # the segment is priced at both stages' bytes minus the hit record and the ray re-read that no longer exist
            gen0 = fused = False
            dominant, dom_ms = "tree_pool", shd_ms
            dom_bytes_unit = (b_ray - 36) + (b_shade - 8)
            algorithmic = seg * dom_bytes_unit
        achieved = algorithmic / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE per segment, see
        # profiles/pmc_traffic.json) scaled to this run's units per launch; null if no PMC summary is committed.
        traffic = None
        pmc = {}
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                pmc = json.load(open(prof))[dominant]
                traffic = pmc["hbm_bytes_per_unit"] * seg / n_launch
            except Exception:
                traffic = None
        result = {
            "metric": f"Mpaths/sec (traced pixels x spp / s), {args.scene} realtime frames" if args.realtime
            else "Mpaths/sec (pixels x spp / s), Cornell box offline render" if args.scene == "cornell_box"
            else f"Mpaths/sec (pixels x spp / s), {args.scene} offline render",
            "value": value,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.scene} {W}x{H} @ {args.spp} spp, max_depth {args.max_depth}, "
                            f"{'realtime progressive frames (80 % pixel skip), ' if args.realtime else ''}"
                            f"{'one tile' if world == 1 else f'rows interleaved over {world} GPUs + RCCL gather'}",
                "paths_in_flight": args.paths_in_flight,
                "frames_per_s": (args.steps / elapsed) if args.realtime else None,
                "segments_per_path": float(st.segments) / max(float(st.paths), 1.0),
                "nan_pixel_fraction": nan_frac,
                "arithmetic": "f64 rays, intersections and throughput; f32 direction sampling and pdf weights; "
                              "reciprocal + Newton instead of IEEE division (PT_CFG_EXACT switches all of that off)",
                "exact_mode_Mpaths_s": exact_value,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": {"extend": "k_extend_accel_pool / k_extend_scan", "shade": "k_shade", "shade_fused": "k_shade<FUSE=1>",
                           "path_pool": "k_path_pool<FUSE=1>", "tree_pool": "k_tree_pool"}[dominant],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "avg_launch_ms": dom_ms / n_launch,
                "algorithmic_bytes_per_launch": algorithmic / n_launch,
                "bytes_per_unit": dom_bytes_unit,
                "units_per_launch": seg / n_launch,
                "V": V,
                "P": P,
                "profiled_launches": n_launch,
                "other": {
                    "fused_trace_shade": fused,
                    "bounce0_generates_rays": gen0,
                    "path_pool": pool,
                    "tree_pool": tree_pool,
                    # from the committed PMC passes of this kernel: the pipe that actually bounds it
                    "valu_busy_frac": pmc.get("valu_busy_frac"),
                    "mean_active_lanes": pmc.get("mean_active_lanes"),
                    # the unfused two-kernel figure 204 + 32 (V + P), which prices every node / primitive visit as
                    # an HBM fetch, and the rate it would give; not used for `achieved`
                    "two_kernel_bytes_per_segment": b_ray + b_shade,
                    "two_kernel_model_GBs": seg * (b_ray + b_shade) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None,
                    "extend_ms_total": None if fused else ext_ms,
                    "shade_ms_total": shd_ms,
                    "raygen_ms_total": None if gen0 else st.ms_raygen,
                    "first_bounce_ms_total": st.ms_raygen if gen0 else None,
                    "resolve_ms_total": st.ms_resolve,
                    "extend_Mrays_s": seg / (ext_ms * 1e-3) / 1e6 if (ext_ms > 0 and not fused) else None,
                    "shade_Msegments_s": seg / (shd_ms * 1e-3) / 1e6 if shd_ms > 0 else None,
                },
            },
        }
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["config"]["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(result), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
