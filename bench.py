#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s (pixels x spp / s) of the path tracer on the Cornell box, 1024 x 1024 @ 1024 spp,
max_depth 50 (the offline configuration of the reference: src/lib.rs:128-135), on N GPUs of one node, all f64.

A step is one full render of the image. With N > 1 there is one process per GPU (this script starts them itself when
it is run without a launcher): the processes form a group through the library (pt_group_join, RCCL), rank r traces
rows r, r + N, ... into its own copy of the f32 RGBA image in one pt_render call, and pt_group_gather moves the rows
to rank 0 inside the timed region. The RNG is keyed by global pixel and sample index, so the image does not depend
on N.

Prints ONE JSON line on rank 0: the throughput (`value`), the same workload in exact mode, the extend stage alone on
fixed ray batches (Mrays/s with id / t agreement against the CPU oracle, and its bytes per ray against the HBM peak), the
roofline of the dominant kernel against the FP64 vector peak, the other single-GPU configurations of the reference
(`config.workloads`: RTiOW, the final scene and the smoke box in reference and book semantics, one million spheres), each
with its own throughput, segments per path, NaN fraction and roofline block, and the CPU oracle timed on the host cores.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X: 256 CUs x 4 SIMDs x 16 f64 FMA lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0
SEED = 0x5EED0001

# The other single-GPU configurations: (label, recipe, recipe parameter, width, height, spp, max_depth, book semantics).
# Book semantics (PT_RENDER_BOOK_SEMANTICS with the scene's own light as the light shape, recipe parameter 1) is the
# headline of the two media scenes: in reference semantics the final scene's image is NaN in every pixel and the smoke
# box's in 87 % of them (Isotropic has no scattering pdf in the reference, src/material.rs:142-144,346-352).
WORKLOADS = [
    ("C1 RTiOW random_scene (src/lib.rs:713-765)", "random_scene", 0, 1200, 800, 100, 50, False),
    ("C3 final_book_two, reference semantics (src/lib.rs:990-1077)", "final_book_two", 0, 1024, 1024, 64, 50, False),
    ("C3 final_book_two, book semantics", "final_book_two", 1, 1024, 1024, 64, 50, True),
    ("C3 cornell_smoke, reference semantics (src/lib.rs:894-988)", "cornell_smoke", 0, 1024, 1024, 64, 50, False),
    ("C3 cornell_smoke, book semantics", "cornell_smoke", 1, 1024, 1024, 64, 50, True),
    ("C4 one million spheres (sphere_cloud)", "sphere_cloud", 1000000, 3840, 2160, 16, 10, False),
]


def load_model(entry):
    """Per-segment counts of a workload's pass kernel from the committed counter passes (profiles/pmc_model.json, written
    by tools/pmc_model.py from a rocprofv3 --pmc summary under profiles/)."""
    path = os.path.join(ROOT, "profiles", "pmc_model.json")
    if not os.path.exists(path):
        return {}
    return json.load(open(path)).get(entry, {})


def roofline_block(pt, st, model):
    """Roofline of the pass kernel of the run `st` describes: FP64 instructions issued per segment (ADD + MUL + 2 FMA,
    x 64 lanes) and HBM bytes per segment (2 x FETCH_SIZE + WRITE_SIZE) from the committed counter pass, scaled to this
    run's segments per launch and divided by this run's launch time (HIP events on the library's stream)."""
    n_launch = max(int(st.profiled_bounce_launches), 1)
    seg = float(st.profiled_segments)
    form = int(st.last_pass_form)
    one_launch = form in (pt.PT_PASS_PATH_POOL, pt.PT_PASS_TREE_POOL, pt.PT_PASS_PERSISTENT)
    dom_ms = st.ms_shade if (one_launch or st.ms_shade >= st.ms_extend) else st.ms_extend
    kernel = {pt.PT_PASS_PATH_POOL: "k_path_pool", pt.PT_PASS_TREE_POOL: "k_tree_pool",
              pt.PT_PASS_PERSISTENT: "k_trace_persistent"}.get(form, "k_shade" if st.ms_shade >= st.ms_extend else "k_extend*")
    if model and model.get("kernel_full_name", kernel).split("<")[0] != kernel:
        model = {}  # the committed counters describe another kernel than the one that ran
    units_per_launch = seg / n_launch
    launch_s = dom_ms / n_launch * 1e-3
    flop_unit = model.get("fp64_flop_issued_per_unit")
    achieved = flop_unit * units_per_launch / launch_s / 1e12 if (flop_unit and launch_s > 0) else None
    hbm_unit = model.get("hbm_bytes_per_unit")
    traffic = hbm_unit * units_per_launch if hbm_unit is not None else None
    fp64_frac = (achieved / FP64_VECTOR_PEAK_TFLOPS) if achieved is not None else None
    hbm_gbs = (traffic / launch_s / 1e9) if (traffic is not None and launch_s > 0) else None
    hbm_frac = hbm_gbs / HBM_PEAK_GBS if hbm_gbs is not None else None
    # The roof a kernel is priced against is the one it stands closer to: the f64 issue rate of the vector pipe for the
    # kernels that keep their scene in caches and their paths in registers, the memory system for the kernel that walks a
    # tree of a million nodes (its slab tests are f32, so its f64 flop say little; what it moves is node and primitive
    # records, 32 and 64 bytes at a time, most of them from L2 and the last-level cache).
    memory_bound = hbm_frac is not None and fp64_frac is not None and hbm_frac > fp64_frac
    head = ({"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
             "achieved_is": "bytes the L2s request from the fabric (measured, see traffic_note), per launch, over the launch's "
                            "duration: node and primitive records as this kernel reads them, not a per-visit byte model"}
            if memory_bound else
            {"bound": "fp64_valu", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp64_frac})
    return {
        **head,
        "kernel": model.get("kernel_full_name", kernel),
        "fp64_valu_TFLOPs": achieved,
        "fp64_valu_frac": fp64_frac,
        "traffic": traffic,
        "traffic_note": "bytes the L2s request from the fabric per launch (2 x FETCH_SIZE + WRITE_SIZE): hits in the 256-MiB "
                        "last-level cache are counted, and the factor 2 is calibrated for coalesced 16-byte-per-lane reads only "
                        "(divergent node fetches of the tree kernels may be counted up to twice)",
        "avg_launch_ms": dom_ms / n_launch,
        "units_per_launch": units_per_launch,
        "unit_name": "path segment as the GPU traces it (fast mode stops a path whose throughput is exactly zero; the "
                     "reference follows it to the depth cap: see segments_per_path against cpu_baseline.S)",
        "fp64_flop_issued_per_unit": flop_unit,
        "hbm_bytes_per_unit": hbm_unit,
        "hbm_GBs": hbm_gbs,
        "hbm_frac_of_peak": hbm_frac,
        "valu_busy_frac": model.get("valu_busy_frac"),
        "mean_active_lanes": model.get("mean_active_lanes"),
        "valu_insts_per_unit": model.get("valu_insts_per_unit"),
        "fp64_share_of_valu_insts": model.get("fp64_share_of_valu_insts"),
        "wait_frac_of_wave_cycles": model.get("wait_frac_of_wave_cycles"),
        "l2_hit_frac": model.get("l2_hit_frac"),
        "model_source": model.get("source"),
        "profiled_launches": n_launch,
        "Msegments_s": seg / (dom_ms * 1e-3) / 1e6 if dom_ms > 0 else None,
        "stage_ms": {"raygen": st.ms_raygen, "extend": st.ms_extend, "shade_or_pass_kernel": st.ms_shade,
                     "resolve": st.ms_resolve},
    }


def workload_leg(pt, torch, device, spec, steps, warmup, paths_in_flight):
    """One of the other configurations on one GPU: warm-up renders, then `steps` timed renders between synchronisations."""
    import hashlib

    label, name, param, W, H, spp, depth, book = spec
    scene = pt.Scene.named(name, W / H, seed=SEED, param=param)
    ctx = pt.Context(device=device, seed=SEED, max_paths_in_flight=paths_in_flight, flags=pt.PT_CFG_PROFILE_STAGES, profile_period=1)
    ctx.upload(scene)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{device}")
    ctx.bind_image(image.data_ptr(), W, H)
    rflags = pt.PT_RENDER_BOOK_SEMANTICS if book else 0
    for k in range(warmup):
        ctx.render(W, H, spp, depth, frame=k, flags=rflags)
    ctx.stats_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ctx.render(W, H, spp, depth, frame=warmup + k, flags=rflags)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    out = {
        "workload": f"{label}: {name} {W}x{H} @ {spp} spp, max_depth {depth}",
        "Mpaths_s": float(W) * H * spp * steps / elapsed / 1e6,
        "ms_per_step": elapsed / steps * 1e3,
        "steps": steps,
        "warmup": warmup,
        "book_semantics": book,
        "headline_of_its_config": book or name in ("random_scene", "sphere_cloud"),
        "segments_per_path": float(st.segments) / max(float(st.paths), 1.0),
        "nan_pixel_fraction": float(torch.isnan(image[..., :3]).any(dim=-1).float().mean().item()),
        "image_sha1_16": hashlib.sha1(image.cpu().numpy().tobytes()).hexdigest()[:16],
        "roofline": roofline_block(pt, st, load_model(name + (":book" if book else ""))),
    }
    ctx.close()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--scene-param", type=int, default=0, help="recipe parameter (sphere_cloud: sphere count)")
    ap.add_argument("--paths-in-flight", type=int, default=1 << 30,
                    help="paths per pass (24 B of device memory each, 228 B with a pass form that keeps paths in memory: 2^30 = the "
                         "whole default render as one pass, 26 GB)")
    ap.add_argument("--book-semantics", action="store_true", help="PT_RENDER_BOOK_SEMANTICS (Isotropic scatters as in the book)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-mode leg")
    ap.add_argument("--no-extend", action="store_true", help="skip the extend-stage leg")
    ap.add_argument("--cfg-flags", type=int, default=0, help="extra PT_CFG_* bits for the timed context (8 = PT_CFG_FAT_RECORDS, 32 = PT_CFG_WIDE_NODES: A/B runs)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the legs over the other configurations (config.workloads)")
    ap.add_argument("--extend-cloud", type=int, default=1000000,
                    help="spheres of the second extend scene (0 skips it; its host build and oracle pass take ~1 min)")
    ap.add_argument("--realtime", action="store_true",
                    help="progressive frames instead of the offline render (src/lib.rs:118-123: spp 1, depth 10, "
                         "80 %% pixel skip, running mean); pass --width 1920 --height 1080 --steps <frames>")
    ap.add_argument("--present", choices=["none", "readback", "ring", "ring8"], default="none",
                    help="--realtime on one GPU: every frame also goes to host memory, as the reference hands its buffer to "
                         "the presentation graph (src/lib.rs:496-509): 'readback' = pt_readback after each frame, 'ring' = "
                         "pt_present_submit into a two-slot ring, waiting for a frame only when the next one has been traced; "
                         "'ring8' = the same with PT_PRESENT_DISPLAY8 (the display transform as 8-bit RGBA: a quarter of the bytes)")
    ap.add_argument("--gather", choices=["capi", "torch"], default="capi",
                    help="N > 1: rows to rank 0 through pt_group_gather (RCCL inside the library) or torch.distributed.gather")
    args = ap.parse_args(argv)
    if args.realtime:
        args.spp, args.max_depth = 1, 10
        args.no_cpu_baseline = args.no_exact = args.no_extend = True
    return args


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start N ranks as a child process (nothing here has touched the GPU or
    imported torch yet, and this process never does) and pass rank 0's line through."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in lines[-1:]:
        print(l, flush=True)
    if proc.returncode != 0 or not lines:
        sys.stderr.write(proc.stdout[-4000:])
        raise SystemExit(proc.returncode or 1)


def cpu_baseline(ob, scene, width, height, max_depth, target_seconds=12.0, name="cornell_box"):
    """Times the CPU oracle (f64 restatement of the reference, tile jobs on a thread pool) on a bounded sample
    of the same workload: the same scene and image size at a reduced spp."""
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


def extend_leg(pt, ob, torch, np, device, name, param, check_rays):
    """BVH extend alone (Hitable::hit on SceneState.world, src/hitable.rs:28): 2^20 primary camera rays (sample 0 of a
    1024 x 1024 image) and the secondary rays cosine-sampled about their hit normals, resident in HBM, through
    pt_intersect_device (fast mode). Agreement with the CPU oracle on the first `check_rays` rays of each batch:
    fraction of equal primitive ids and the largest relative difference of t where both hit."""
    W = H = 1024
    scene = pt.Scene.named(name, 1.0, param=param)
    ctx = pt.Context(device=device, seed=SEED, max_paths_in_flight=1 << 20)
    ctx.upload(scene)
    desc = scene.desc()
    pix = np.arange(W * H, dtype=np.uint64)
    primary = ob.camera_rays(scene.camera, W, H, pix, np.zeros(W * H, dtype=np.uint32))
    # secondary rays need the primary hits; the device's own (checked against the oracle below) keep this bounded
    hits = ctx.intersect(primary, medium_seed=5)
    secondary, valid = ob.secondary_rays(hits, primary)
    secondary = secondary[valid]
    out = {}
    for label, rays in (("primary", primary), ("secondary", secondary)):
        n = len(rays)
        soa = np.concatenate([rays["origin"][:, k] for k in range(3)] + [rays["direction"][:, k] for k in range(3)] + [rays["time"]])
        d_soa = torch.from_numpy(np.ascontiguousarray(soa)).to(f"cuda:{device}")
        ctx.intersect_device(d_soa.data_ptr(), n, medium_seed=5, repeats=3)
        ms = ctx.intersect_device(d_soa.data_ptr(), n, medium_seed=5, repeats=20)
        m = min(n, check_rays)
        got = hits[:m] if label == "primary" else ctx.intersect(rays[:m], medium_seed=5)
        ref, _ = ob.intersect(desc, rays[:m], medium_seed=5)
        same = got["prim"] == ref["prim"]
        both = same & (ref["prim"] >= 0)
        rel = np.abs(got["t"][both] - ref["t"][both]) / np.maximum(np.abs(ref["t"][both]), 1e-300)
        out[label] = {"rays": n, "Mrays_s": n / ms / 1e3, "ms": ms, "checked_rays": int(m),
                      "id_agreement": float(same.mean()), "max_rel_t_err": float(rel.max()) if rel.size else 0.0,
                      "hit_fraction": float((ref["prim"] >= 0).mean())}
    ctx.close()
    # Roofline of the extend launches against HBM: node records read (V) and primitives tested (P) per ray, counted by the
    # traversal itself on the first 65 536 rays of each batch (PT_CFG_COUNT_ACCEL), priced at the size of the records the
    # timed kernel reads, plus the ray in (7 f64) and the hit out (f64 + i32).
    cctx = pt.Context(device=device, seed=SEED, max_paths_in_flight=1 << 16, flags=pt.PT_CFG_COUNT_ACCEL)
    cctx.upload(scene)
    n_nodes = scene.accel_info()["n_nodes"]
    for label, rays in (("primary", primary), ("secondary", secondary)):
        m = min(len(rays), 1 << 16)
        cctx.stats_reset()
        cctx.intersect(rays[:m], medium_seed=5)
        cst = cctx.stats()
        V, P = cst.node_visits / m, cst.prim_tests / m
        tree = V > 0
        node_bytes = (32 if n_nodes >= 32768 else 64) if tree else 0  # big trees are read in their 32-byte form
        compulsory = 56 + 12
        model_bytes = compulsory + node_bytes * V + 64 * P
        rate = out[label]["Mrays_s"] * 1e6
        out[label]["roofline"] = {
            "bound": "hbm", "kernel": "k_extend_accel_pool" if tree else "k_extend_list",
            "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "node_visits_per_ray": V, "prim_tests_per_ray": P, "counted_rays": int(m),
            "bytes_per_ray": model_bytes, "bytes_per_ray_compulsory": compulsory,
            "achieved": rate * model_bytes / 1e9, "frac": rate * model_bytes / 1e9 / HBM_PEAK_GBS,
            "achieved_compulsory": rate * compulsory / 1e9, "frac_compulsory": rate * compulsory / 1e9 / HBM_PEAK_GBS,
            "note": ("records of 32 B per node visit and 64 B per primitive test, as the kernel reads them; most of these reads "
                     "are served by L2 and the last-level cache, so frac prices the memory system as a whole, not HBM alone")
            if tree else
            ("a list scene: every ray is tested against every primitive, and the records come through the scalar cache "
             "once per wave, not once per ray; only frac_compulsory (ray in, hit out) is HBM traffic"),
        }
    cctx.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: one rank per GPU expected")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # PTGPU_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo, for rehearsing the multi-rank code on a one-GPU box
    # (the numbers mean nothing; RCCL refuses two ranks on one device). Never set by the driver.
    rehearsal = os.environ.get("PTGPU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        args.gather = "torch"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world

    pt = importlib.import_module("path-tracer_amd")
    pt.register_earthmap()  # texels of the image texture (tests/golden/make_earthmap.py)
    tiles = importlib.import_module("path-tracer_amd.tiles")
    W, H = args.width, args.height
    scene = pt.Scene.named(args.scene, W / H, seed=SEED, param=args.scene_param)
    rflags = pt.PT_RENDER_BOOK_SEMANTICS if args.book_semantics else 0
    mode = pt.PT_MODE_REALTIME if args.realtime else pt.PT_MODE_OFFLINE
    # a realtime frame traces the pixels that pass the 80 % skip (src/trace.rs:144): 20 % of them on average
    paths_per_step = float(W) * H * args.spp * (0.2 if args.realtime else 1.0)

    gather_note = []

    def timed_leg(flags):
        """W warm-up steps, then K steps between barriers, max over ranks: (seconds, stats, image tensor, ctx)."""
        # a rank traces every world-th row: its passes never hold more than its share of the paths
        share = W * ((H + world - 1) // world) * args.spp
        # (a profiled pass ends with a host wait for its counters: every 8th pass by default, every 64th frame of a
        # realtime run, whose frames are otherwise issued without waiting)
        ctx = pt.Context(device=local_rank, seed=SEED, max_paths_in_flight=max(1 << 16, min(args.paths_in_flight, share)), flags=flags,
                         **({"profile_period": 64} if args.realtime else {}))
        ctx.upload(scene)
        # the image lives in a torch tensor (rank 0 inspects it; the torch gather moves it); the library renders into it
        image = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
        ctx.bind_image(image.data_ptr(), W, H)
        rows, pack, gather_list = None, None, None
        if world > 1:
            if args.gather == "capi":
                # A failure that every rank sees alike (RCCL not loadable beside PyTorch's copy, an id it rejects) falls
                # back to the torch.distributed gather instead of losing the run; the line then says so.
                ok = torch.ones(1, dtype=torch.int32, device=image.device)
                try:
                    ident = torch.zeros(pt.PT_GROUP_ID_BYTES, dtype=torch.uint8, device=image.device)
                    if rank == 0:
                        ident.copy_(torch.frombuffer(bytearray(pt.group_unique_id()), dtype=torch.uint8))
                except pt.PtError as e:
                    ok.zero_()
                    gather_note.append(str(e))
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()):
                    dist.broadcast(ident, src=0)
                    try:
                        ctx.group_join(ident.cpu().numpy().tobytes(), rank, world)
                    except pt.PtError as e:
                        ok.zero_()
                        gather_note.append(str(e))
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()):
                    rows = ctx.group_rows()
                else:
                    ctx.group_leave()
                    args.gather = "torch"
                    gather_note.append("fell back to torch.distributed.gather")
            if args.gather != "capi":
                rows = tiles.rows_for(rank, world)
                pack = torch.zeros((tiles.max_rows(world, H), W, 4), dtype=torch.float32, device=image.device)
                if rank == 0:
                    gather_list = [torch.zeros_like(pack) for _ in range(world)]

        def gather():
            if args.gather == "capi" and args.realtime:
                ctx.group_gather_async(0)  # queued behind an event on a second stream: the next frame is traced meanwhile
            elif args.gather == "capi":
                ctx.group_gather(0)
            elif rehearsal:  # gloo gathers host tensors
                host = image.cpu()
                tiles.gather_rows(host, rank, world, dist)
                image.copy_(host)
            else:
                tiles.gather_rows(image, rank, world, dist, pack=pack, gather_list=gather_list)

        ring = None
        if args.present in ("ring", "ring8") and world == 1:
            ring = ctx.present_attach(None, slots=2, width=W, height=H)  # the library's page-locked ring
        present_kind = pt.PT_PRESENT_DISPLAY8 if args.present == "ring8" else pt.PT_PRESENT_RAW

        # Realtime frames are issued one behind the other without a host wait in between (PT_RENDER_NO_WAIT: the
        # reference's main loop presents frame k while its workers trace frame k + 1, src/lib.rs:476-509); whatever reads
        # the image through the library is ordered behind them, and the timed region ends with a device synchronisation.
        # (The torch gather reads the bound image on torch's stream, so its frames are waited for.)
        def step(frame):
            # (args.gather may still fall back to "torch" at the first transfer below)
            no_wait = pt.PT_RENDER_NO_WAIT if (args.realtime and (world == 1 or args.gather == "capi")) else 0
            ctx.render(W, H, args.spp, args.max_depth, frame=frame, mode=mode, flags=rflags | no_wait, rows=rows)
            if world > 1:
                gather()
            elif args.present == "readback":
                ctx.readback()
            elif ring is not None:
                ctx.present_submit(frame & 1, kind=present_kind)
                ctx.present_wait((frame & 1) ^ 1)  # the frame before this one is shown now

        def barrier():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        if world > 1:
            # communicator set-up and staging buffers (seconds the first time) never land in the timed region,
            # also with --warmup 0
            ctx.render(W, H, 1, args.max_depth, frame=0, mode=mode, flags=rflags, rows=rows)
            if args.gather == "capi":  # the first transfer through the library's communicator: same fallback as above
                ok = torch.ones(1, dtype=torch.int32, device=image.device)
                try:
                    gather()
                except pt.PtError as e:
                    ok.zero_()
                    gather_note.append(str(e))
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if not int(ok.item()):
                    try:
                        ctx.group_leave()
                    except pt.PtError as e:
                        gather_note.append(str(e))
                    args.gather = "torch"
                    gather_note.append("fell back to torch.distributed.gather")
                    rows = tiles.rows_for(rank, world)
                    pack = torch.zeros((tiles.max_rows(world, H), W, 4), dtype=torch.float32, device=image.device)
                    if rank == 0:
                        gather_list = [torch.zeros_like(pack) for _ in range(world)]
                    ctx.render(W, H, 1, args.max_depth, frame=0, mode=mode, flags=rflags, rows=rows)
                    gather()
            else:
                gather()
        for w in range(args.warmup):
            step(w)
        ctx.stats_reset()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k)
        if world > 1 and args.gather == "capi" and args.realtime:
            ctx.group_wait()  # the last frame's rows have landed
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else image.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, ctx.stats(), image, ctx

    elapsed, st, image, ctx = timed_leg(pt.PT_CFG_PROFILE_STAGES | args.cfg_flags)
    value = paths_per_step * args.steps / elapsed / 1e6
    nan_frac, image_hash = 0.0, None
    if rank == 0:  # the image is a pure function of scene, seed, size and spp (never of GPU count, tiling or pass form)
        import hashlib

        # (after asynchronous gathers the assembled frame lives in the library: pt_readback returns it)
        final = torch.from_numpy(ctx.readback()) if (world > 1 and args.gather == "capi" and args.realtime) else image.cpu()
        nan_frac = float(torch.isnan(final[..., :3]).any(dim=-1).float().mean().item())
        image_hash = hashlib.sha1(final.numpy().tobytes()).hexdigest()[:16]
    ctx.close()
    del image

    exact_value = exact_ms = None
    if not args.no_exact:
        # the same workload, same steps / warm-up / spp, in exact mode (IEEE division and roots, no contraction,
        # every box test of the reference's own tree): reported beside `value`
        x_elapsed, _, x_image, x_ctx = timed_leg(pt.PT_CFG_EXACT)
        exact_value = paths_per_step * args.steps / x_elapsed / 1e6
        exact_ms = x_elapsed / args.steps * 1e3
        x_ctx.close()
        del x_image

    if rank == 0:
        import oracle_binding as ob

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(ob, scene, W, H, args.max_depth, name=args.scene)
        extend = None
        if world == 1 and not args.no_extend:
            extend = {"cornell_box": extend_leg(pt, ob, torch, np, local_rank, "cornell_box", 0, 1 << 20)}
            if args.extend_cloud > 0:
                extend[f"sphere_cloud_{args.extend_cloud}"] = extend_leg(pt, ob, torch, np, local_rank, "sphere_cloud",
                                                                         args.extend_cloud, 1 << 16)
        model = load_model("k_path_pool") if (args.scene == "cornell_box" and not args.realtime) else \
            load_model(args.scene + (":book" if args.book_semantics else "")) if not args.realtime else {}
        roofline = roofline_block(pt, st, model)
        workloads = None
        if world == 1 and not args.no_workloads and not args.realtime:
            workloads = [workload_leg(pt, torch, local_rank, spec, 2, 1, min(args.paths_in_flight, 1 << 27)) for spec in WORKLOADS]
        result = {
            "metric": f"Mpaths/sec (traced pixels x spp / s), {args.scene} realtime frames" if args.realtime
            else "Mpaths/sec (pixels x spp / s), Cornell box offline render; Mrays/s BVH extend in config.extend" if args.scene == "cornell_box"
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
            "gather": (("pt_group_gather (RCCL inside the library)" if args.gather == "capi" else "torch.distributed.gather")
                       if world > 1 else None),
            "config": {
                "workload": f"{args.scene} {W}x{H} @ {args.spp} spp, max_depth {args.max_depth}, "
                            f"{'realtime progressive frames (80 % pixel skip), ' if args.realtime else ''}"
                            f"{'one tile' if world == 1 else f'rows interleaved over {world} GPUs + RCCL gather ({args.gather})'}",
                "world_size": world,
                "gather_note": "; ".join(gather_note) or None,
                "paths_in_flight": args.paths_in_flight,
                "frames_per_s": (args.steps / elapsed) if args.realtime else None,
                "present": args.present,
                "segments_per_path": float(st.segments) / max(float(st.paths), 1.0),
                "nan_pixel_fraction": nan_frac,
                "image_sha1_16": image_hash,
                "book_semantics": bool(args.book_semantics),
                "arithmetic": "f64 throughout (rays, intersections, sampling, pdf values, weights, throughput); quotients and "
                              "roots by Newton-refined hardware reciprocals, a*b+c fused, sin/cos by quadrant reduction + "
                              "polynomials (PT_CFG_EXACT: IEEE division / sqrt / libm, no contraction)",
                "exact_mode_Mpaths_s": exact_value,
                "exact_mode_ms_per_step": exact_ms,
                "extend": extend,
                "workloads": workloads,
                "gather": (args.gather if world > 1 else None),
            },
            "roofline": roofline,
        }
        if cpu is not None:
            result["cpu_baseline"] = cpu
            result["config"]["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
