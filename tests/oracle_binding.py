"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference used as the checker.

Test infrastructure only: nothing under path-tracer_amd/ imports this module.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import importlib

ptgpu = importlib.import_module("path-tracer_amd") if ROOT in sys.path else None
if ptgpu is None:
    sys.path.insert(0, ROOT)
    ptgpu = importlib.import_module("path-tracer_amd")

LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} missing: run `make -C {ROOT}/oracle`")
lib = C.CDLL(LIB_PATH)


class OracleStats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64),
        ("segments", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("seconds", C.c_double),
        ("threads", C.c_uint32),
        ("tiles", C.c_uint32),
    ]


_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
lib.oracle_intersect.restype = C.c_int
lib.oracle_intersect.argtypes = [C.POINTER(ptgpu.SceneDesc), _vp, C.c_size_t, C.c_double, C.c_double, C.c_uint64, _vp, C.POINTER(OracleStats)]
lib.oracle_render.restype = C.c_int
lib.oracle_render.argtypes = [C.POINTER(ptgpu.SceneDesc), C.POINTER(ptgpu.Camera), C.POINTER(ptgpu.RenderArgs), C.c_uint64, C.c_int, _vp, _vp, C.POINTER(OracleStats)]
lib.oracle_trace_paths.restype = C.c_int
lib.oracle_trace_paths.argtypes = [C.POINTER(ptgpu.SceneDesc), C.POINTER(ptgpu.Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _vp, _vp, C.c_size_t, _vp]
lib.oracle_camera_rays.restype = C.c_int
lib.oracle_camera_rays.argtypes = [C.POINTER(ptgpu.Camera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _vp, _vp, C.c_size_t, _vp]
lib.oracle_camera_frame.restype = C.c_int
lib.oracle_camera_frame.argtypes = [C.POINTER(ptgpu.Camera), _dp]
lib.oracle_secondary_rays.restype = C.c_int
lib.oracle_secondary_rays.argtypes = [_vp, _vp, C.c_size_t, C.c_uint64, _vp, _vp]
lib.oracle_texture_value.restype = C.c_int
lib.oracle_texture_value.argtypes = [C.POINTER(ptgpu.SceneDesc), C.c_int, C.c_double, C.c_double, _dp, _dp]
lib.oracle_perlin_noise.restype = C.c_double
lib.oracle_perlin_noise.argtypes = [C.POINTER(ptgpu.Perlin), _dp]
lib.oracle_perlin_turb.restype = C.c_double
lib.oracle_perlin_turb.argtypes = [C.POINTER(ptgpu.Perlin), _dp, C.c_int]
lib.oracle_bounding_box.restype = C.c_int
lib.oracle_bounding_box.argtypes = [C.POINTER(ptgpu.SceneDesc), C.c_int, C.c_double, C.c_double, _dp, _dp]
lib.oracle_aabb_hit.restype = C.c_int
lib.oracle_aabb_hit.argtypes = [_dp, _dp, _vp, C.c_double, C.c_double]
lib.oracle_rotate_y_box.restype = C.c_int
lib.oracle_rotate_y_box.argtypes = [_dp, _dp, C.c_double, _dp, _dp, _dp]
lib.oracle_bvh_from_list.restype = C.c_int
lib.oracle_bvh_from_list.argtypes = [C.POINTER(ptgpu.SceneDesc), C.POINTER(C.c_int32), C.c_int, C.c_double, C.c_double, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), _dp]
lib.oracle_path_draw.restype = C.c_double
lib.oracle_path_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64]
lib.oracle_scene_draw.restype = C.c_double
lib.oracle_scene_draw.argtypes = [C.c_uint64, C.c_uint64]
lib.oracle_pixel_skip_draw.restype = C.c_double
lib.oracle_pixel_skip_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
lib.oracle_vec3_op.restype = C.c_int
lib.oracle_vec3_op.argtypes = [C.c_int, _dp, _dp, _dp]
lib.oracle_vec3_scalar.restype = C.c_double
lib.oracle_vec3_scalar.argtypes = [C.c_int, _dp, _dp]
lib.oracle_camera_apply_input.restype = C.c_int
lib.oracle_camera_apply_input.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_double]
lib.oracle_display_transform.restype = C.c_int
lib.oracle_display_transform.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_float, _vp]
lib.oracle_tonemap_rgb8.restype = C.c_int
lib.oracle_tonemap_rgb8.argtypes = [_vp, C.c_uint32, C.c_uint32, _vp]

F64_MAX = np.finfo(np.float64).max


def _v(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def intersect(desc, rays, t_min=0.001, t_max=F64_MAX, medium_seed=0):
    rays = np.ascontiguousarray(rays, dtype=ptgpu.RAY_DTYPE)
    hits = np.zeros(rays.shape[0], dtype=ptgpu.HIT_DTYPE)
    st = OracleStats()
    rc = lib.oracle_intersect(C.byref(desc), rays.ctypes.data, rays.shape[0], t_min, t_max, C.c_uint64(medium_seed), hits.ctypes.data, C.byref(st))
    assert rc == 0
    return hits, st


def render(desc, camera, width, height, spp, max_depth, seed=0x5EED0001, tile=None, mode=0, frame=0, threads=0, image=None, counts=None, flags=0, rows=None):
    x0, y0, x1, y1 = tile if tile is not None else (0, 0, width, height)
    step, phase = rows if rows is not None else (1, 0)
    a = ptgpu.RenderArgs(C.sizeof(ptgpu.RenderArgs), width, height, x0, y0, x1, y1, spp, max_depth, mode, frame, flags, step, phase)
    if image is None:
        image = np.zeros((height, width, 4), dtype=np.float32)
    st = OracleStats()
    rc = lib.oracle_render(C.byref(desc), C.byref(camera), C.byref(a), C.c_uint64(seed), threads, image.ctypes.data, counts.ctypes.data if counts is not None else None, C.byref(st))
    assert rc == 0
    return image, st


def trace_paths(desc, camera, width, height, max_depth, pixels, samples, seed=0x5EED0001, frame=0):
    pixels = np.ascontiguousarray(pixels, dtype=np.uint64)
    samples = np.ascontiguousarray(samples, dtype=np.uint32)
    out = np.zeros((pixels.shape[0], 3), dtype=np.float64)
    rc = lib.oracle_trace_paths(C.byref(desc), C.byref(camera), width, height, max_depth, C.c_uint64(seed), frame, pixels.ctypes.data, samples.ctypes.data, pixels.shape[0], out.ctypes.data)
    assert rc == 0
    return out


def camera_rays(camera, width, height, pixels, samples, seed=0x5EED0001, frame=0):
    pixels = np.ascontiguousarray(pixels, dtype=np.uint64)
    samples = np.ascontiguousarray(samples, dtype=np.uint32)
    out = np.zeros(pixels.shape[0], dtype=ptgpu.RAY_DTYPE)
    rc = lib.oracle_camera_rays(C.byref(camera), width, height, C.c_uint64(seed), frame, pixels.ctypes.data, samples.ctypes.data, pixels.shape[0], out.ctypes.data)
    assert rc == 0
    return out


def secondary_rays(hits, primary, seed=7):
    hits = np.ascontiguousarray(hits, dtype=ptgpu.HIT_DTYPE)
    primary = np.ascontiguousarray(primary, dtype=ptgpu.RAY_DTYPE)
    out = np.zeros(primary.shape[0], dtype=ptgpu.RAY_DTYPE)
    valid = np.zeros(primary.shape[0], dtype=np.uint8)
    rc = lib.oracle_secondary_rays(hits.ctypes.data, primary.ctypes.data, primary.shape[0], C.c_uint64(seed), out.ctypes.data, valid.ctypes.data)
    assert rc == 0
    return out, valid.astype(bool)


def camera_frame(camera):
    out = (C.c_double * 21)()
    lib.oracle_camera_frame(C.byref(camera), out)
    return np.array(out).reshape(7, 3)


def texture_value(desc, tex, u, v, p):
    out = (C.c_double * 3)()
    lib.oracle_texture_value(C.byref(desc), tex, u, v, _v(p), out)
    return np.array(out)


def bounding_box(desc, node, t0=0.0, t1=1.0):
    mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
    rc = lib.oracle_bounding_box(C.byref(desc), node, t0, t1, mn, mx)
    if rc != 0:
        return None
    return np.array(mn), np.array(mx)


def aabb_hit(mn, mx, origin, direction, tmin, tmax):
    r = np.zeros(1, dtype=ptgpu.RAY_DTYPE)
    r["origin"][0] = origin
    r["direction"][0] = direction
    return bool(lib.oracle_aabb_hit(_v(mn), _v(mx), r.ctypes.data, tmin, tmax))


def bvh_from_list(desc, nodes, time0, time1, seed, draws=0):
    n = len(nodes)
    cap = 2 * n + 2  # singleton sub-lists get a node of their own (src/bvh.rs:47-49), so nodes can outnumber leaves
    lr = (C.c_int32 * (2 * cap))()
    box = (C.c_double * (6 * cap))()
    d = C.c_uint64(draws)
    arr = (C.c_int32 * n)(*nodes)
    created = lib.oracle_bvh_from_list(C.byref(desc), arr, n, time0, time1, C.c_uint64(seed), C.byref(d), lr, box)
    return np.array(lr[: 2 * created]).reshape(-1, 2), np.array(box[: 6 * created]).reshape(-1, 6), d.value


def vec3_op(op, a, b=(0, 0, 0)):
    out = (C.c_double * 3)()
    rc = lib.oracle_vec3_op(op, _v(a), _v(b), out)
    assert rc == 0
    return tuple(out)


def vec3_scalar(op, a, b=(0, 0, 0)):
    return lib.oracle_vec3_scalar(op, _v(a), _v(b))


def camera_apply_input(camera, keys_pressed=0, keys_held=0, mouse=(0.0, 0.0), frame_time=1.0 / 60.0):
    """Returns (moved, origin, look_at) after update_from_input + update, from the oracle's restatement."""
    cam = np.array(list(camera.origin) + list(camera.look_at) + list(camera.vup), dtype=np.float64)
    moved = lib.oracle_camera_apply_input(cam.ctypes.data, keys_pressed, keys_held, mouse[0], mouse[1], frame_time)
    return bool(moved), cam[0:3].copy(), cam[3:6].copy()


def display_transform(rgba, exposure=1.3):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w, _ = rgba.shape
    out = np.empty((h, w, 4), dtype=np.float32)
    lib.oracle_display_transform(rgba.ctypes.data, w, h, float(exposure), out.ctypes.data)
    return out


def tonemap_rgb8(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w, _ = rgba.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib.oracle_tonemap_rgb8(rgba.ctypes.data, w, h, out.ctypes.data)
    return out
