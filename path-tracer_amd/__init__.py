"""ctypes binding of libptgpu.so, the MI355X render path for PathTracerRust.

The directory name contains a hyphen, so import it with
``importlib.import_module("path-tracer_amd")`` (tests/conftest.py does this and
exposes it as ``ptgpu``).  The binding is a thin mirror of include/ptgpu.h: the
library is the product, this file only moves pointers.  There is no CPU
fallback: if the shared library is missing the import raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptgpu.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()). "
        "The HIP library is the only implementation of the render path; there is no fallback."
    )

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

PT_OK = 0
PT_ERR_INVALID_ARG = -1
PT_ERR_UNSUPPORTED_SCENE = -2
PT_ERR_NO_DEVICE = -3
PT_ERR_HIP = -4
PT_ERR_NO_SCENE = -5
PT_ERR_NO_CAMERA = -6
PT_ERR_OUT_OF_MEMORY = -7
PT_ERR_TOO_LARGE = -8
PT_ERR_MISSING_ASSET = -9
PT_ERR_COMM = -10
PT_GROUP_ID_BYTES = 128

PT_MODE_OFFLINE = 0
PT_MODE_REALTIME = 1
PT_RENDER_BOOK_SEMANTICS = 1
PT_RENDER_NO_WAIT = 2
PT_CFG_PROFILE_STAGES = 1
PT_CFG_COUNT_TRAVERSAL = 2
PT_CFG_EXACT = 4
PT_CFG_FAT_RECORDS = 8
PT_CFG_WIDE_NODES = 32
PT_RECORDS_COMPACT_SPHERES = 1
PT_RECORDS_NODES32 = 2
PT_CFG_COUNT_ACCEL = 16

# pt_hitable_kind
H_SPHERE, H_MOVING_SPHERE, H_RECT, H_BOX, H_LIST, H_FLIP, H_TRANSLATE, H_ROTATE_Y, H_MEDIUM, H_BVH = range(1, 11)
AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2


class Hitable(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("material", C.c_int32),
        ("child", C.c_int32 * 2),
        ("list_start", C.c_int32),
        ("list_count", C.c_int32),
        ("axis", C.c_int32),
        ("reserved", C.c_int32),
        ("f", C.c_double * 10),
    ]


class Material(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("texture", C.c_int32),
        ("albedo", C.c_double * 3),
        ("fuzz", C.c_double),
        ("ref_idx", C.c_double),
        ("emissive", C.c_double),
    ]


class Texture(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("even", C.c_int32),
        ("odd", C.c_int32),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("reserved", C.c_int32),
        ("texel_offset", C.c_uint64),
        ("colour", C.c_double * 3),
        ("scale", C.c_double),
    ]


class Perlin(C.Structure):
    _fields_ = [
        ("ranvec", (C.c_double * 3) * 256),
        ("perm_x", C.c_int32 * 256),
        ("perm_y", C.c_int32 * 256),
        ("perm_z", C.c_int32 * 256),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("root", C.c_int32),
        ("hitables", C.POINTER(Hitable)),
        ("n_hitables", C.c_int32),
        ("n_child_index", C.c_int32),
        ("child_index", C.POINTER(C.c_int32)),
        ("materials", C.POINTER(Material)),
        ("textures", C.POINTER(Texture)),
        ("n_materials", C.c_int32),
        ("n_textures", C.c_int32),
        ("texels", C.POINTER(C.c_uint8)),
        ("n_texel_bytes", C.c_uint64),
        ("perlin", C.POINTER(Perlin)),
        ("lights", C.POINTER(C.c_int32)),
        ("n_lights", C.c_int32),
        ("reserved", C.c_int32),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("look_at", C.c_double * 3),
        ("vup", C.c_double * 3),
        ("vfov", C.c_double),
        ("aspect", C.c_double),
        ("aperture", C.c_double),
        ("focus_dist", C.c_double),
        ("time0", C.c_double),
        ("time1", C.c_double),
    ]


# key bits of UserInput (enum pt_key)
PT_KEY_W, PT_KEY_S, PT_KEY_A, PT_KEY_D, PT_KEY_E, PT_KEY_Q = (1 << k for k in range(6))
PT_KEY_RIGHT, PT_KEY_LEFT, PT_KEY_UP, PT_KEY_DOWN = (1 << k for k in range(6, 10))
PT_KEY_T, PT_KEY_R, PT_KEY_O, PT_KEY_P, PT_KEY_B = (1 << k for k in range(10, 15))


class UserInput(C.Structure):
    _fields_ = [("keys_pressed", C.c_uint32), ("keys_held", C.c_uint32), ("mouse_dx", C.c_float), ("mouse_dy", C.c_float)]


class FrameInfo(C.Structure):
    _fields_ = [
        ("camera_moved", C.c_int32),
        ("cleared", C.c_int32),
        ("exposure", C.c_double),
        ("numframes", C.c_double),
        ("time0", C.c_double),
        ("time1", C.c_double),
        ("sky_brightness", C.c_double),
        ("disable_emissive", C.c_int32),
        ("frame", C.c_uint32),
        ("camera", Camera),
    ]


TUNING_FIELDS = ("pass_form", "pool_refill", "pool_chunk", "pool_grid", "tree_refill", "tree_inner", "tree_shade",
                 "tree_trav_plus1", "accel_form", "accel_refill", "accel_inner", "accel_grid", "shards",
                 "scan_max_ops_plus1", "list_max_leaves_plus1", "profile_period")


class Tuning(C.Structure):
    _fields_ = [(name, C.c_uint32) for name in TUNING_FIELDS]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("seed", C.c_uint64),
        ("max_paths_in_flight", C.c_uint32),
        ("flags", C.c_uint32),
        ("tuning", Tuning),
    ]


class RenderArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("x0", C.c_uint32),
        ("y0", C.c_uint32),
        ("x1", C.c_uint32),
        ("y1", C.c_uint32),
        ("spp", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("mode", C.c_uint32),
        ("frame", C.c_uint32),
        ("flags", C.c_uint32),
        ("row_step", C.c_uint32),
        ("row_phase", C.c_uint32),
    ]


PT_PASS_NONE, PT_PASS_PER_BOUNCE, PT_PASS_PERSISTENT, PT_PASS_PATH_POOL, PT_PASS_TREE_POOL = 0, 1, 3, 4, 5


class Stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64),
        ("segments", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("prim_tests", C.c_uint64),
        ("nan_pixels", C.c_uint64),
        ("launches_raygen", C.c_uint64),
        ("launches_extend", C.c_uint64),
        ("launches_shade", C.c_uint64),
        ("launches_resolve", C.c_uint64),
        ("ms_raygen", C.c_double),
        ("ms_extend", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_resolve", C.c_double),
        ("profiled_segments", C.c_uint64),
        ("profiled_paths", C.c_uint64),
        ("profiled_passes", C.c_uint64),
        ("profiled_bounce_launches", C.c_uint64),
        ("last_pass_form", C.c_uint64),
    ]


# numpy views of pt_ray / pt_hit
RAY_DTYPE = np.dtype([("origin", "<f8", (3,)), ("direction", "<f8", (3,)), ("time", "<f8")], align=True)
HIT_DTYPE = np.dtype(
    [("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("p", "<f8", (3,)), ("normal", "<f8", (3,)), ("prim", "<i4"), ("material", "<i4")],
    align=True,
)
assert RAY_DTYPE.itemsize == 56 and HIT_DTYPE.itemsize == 80

# Every symbol include/ptgpu.h declares.
EXPORTS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_scene_upload", "pt_camera_set", "pt_render", "pt_accum_reset", "pt_wait",
    "pt_readback", "pt_image_device_ptr", "pt_image_bind", "pt_intersect", "pt_stats_get", "pt_stats_reset", "pt_last_pass_records",
    "pt_stream", "pt_intersect_device", "pt_scene_build_named", "pt_scene_free", "pt_scene_get_desc", "pt_scene_new",
    "pt_scene_texture_constant", "pt_scene_texture_checker", "pt_scene_texture_noise", "pt_scene_texture_image",
    "pt_scene_material_lambertian", "pt_scene_material_diffuse_light", "pt_scene_material_metal",
    "pt_scene_material_dielectric", "pt_scene_sphere", "pt_scene_moving_sphere", "pt_scene_rect", "pt_scene_box",
    "pt_scene_flip_normals", "pt_scene_translate", "pt_scene_rotate_y", "pt_scene_constant_medium", "pt_scene_list",
    "pt_scene_bvh", "pt_scene_set_root", "pt_scene_set_lights", "pt_scene_bounding_box", "pt_scene_flatten_info",
    "pt_tonemap_rgb8", "pt_tonemap_device", "pt_display_transform", "pt_camera_apply_input", "pt_session_new",
    "pt_session_free", "pt_session_frame", "pt_scene_accel_info", "pt_scene_refit", "pt_scene_build_device",
    "pt_abi_version", "pt_intersect_static", "pt_asset_register_image", "pt_group_unique_id", "pt_group_join",
    "pt_group_join_local", "pt_group_rows", "pt_group_gather", "pt_group_gather_local", "pt_group_leave",
    "pt_group_loopback", "pt_render_jobs", "pt_present_attach", "pt_present_slot", "pt_present_pinned", "pt_present_submit", "pt_present_wait", "pt_present_detach",
    "pt_device_alloc_count", "pt_group_layout", "pt_group_staged_row", "pt_group_gather_async", "pt_group_wait",
    "pt_group_loopback_async",
]

_vp = C.c_void_p
_d3 = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("pt_create", C.c_int, C.POINTER(Config), C.POINTER(_vp))
_sig("pt_destroy", None, _vp)
_sig("pt_last_error", C.c_char_p, _vp)
_sig("pt_scene_upload", C.c_int, _vp, C.POINTER(SceneDesc))
_sig("pt_camera_set", C.c_int, _vp, C.POINTER(Camera))
_sig("pt_render", C.c_int, _vp, C.POINTER(RenderArgs))
_sig("pt_render_jobs", C.c_int, _vp, C.POINTER(RenderArgs), C.c_size_t)
_sig("pt_accum_reset", C.c_int, _vp)
_sig("pt_wait", C.c_int, _vp)
_sig("pt_readback", C.c_int, _vp, C.POINTER(C.c_float), C.c_size_t)
_sig("pt_image_device_ptr", _vp, _vp)
_sig("pt_image_bind", C.c_int, _vp, _vp, C.c_uint32, C.c_uint32)
_sig("pt_intersect", C.c_int, _vp, _vp, C.c_size_t, C.c_double, C.c_double, C.c_uint64, _vp)
_sig("pt_stats_get", C.c_int, _vp, C.POINTER(Stats))
_sig("pt_stats_reset", C.c_int, _vp)
_sig("pt_last_pass_records", C.c_int, _vp)
_sig("pt_stream", _vp, _vp)
_sig("pt_intersect_device", C.c_int, _vp, _vp, C.c_size_t, C.c_double, C.c_double, C.c_uint64, C.c_int, _d3)
_sig("pt_scene_build_named", C.c_int, C.c_char_p, C.c_double, C.c_uint64, C.c_int64, C.POINTER(_vp), C.POINTER(Camera))
_sig("pt_scene_free", None, _vp)
_sig("pt_scene_get_desc", C.c_int, _vp, C.POINTER(SceneDesc))
_sig("pt_scene_new", C.c_int, C.c_uint64, C.POINTER(_vp))
_sig("pt_scene_texture_constant", C.c_int, _vp, _d3)
_sig("pt_scene_texture_checker", C.c_int, _vp, C.c_int, C.c_int)
_sig("pt_scene_texture_noise", C.c_int, _vp, C.c_double)
_sig("pt_scene_texture_image", C.c_int, _vp, C.POINTER(C.c_uint8), C.c_int, C.c_int)
_sig("pt_scene_material_lambertian", C.c_int, _vp, C.c_int, C.c_double)
_sig("pt_scene_material_diffuse_light", C.c_int, _vp, C.c_int)
_sig("pt_scene_material_metal", C.c_int, _vp, _d3, C.c_double)
_sig("pt_scene_material_dielectric", C.c_int, _vp, C.c_double)
_sig("pt_scene_sphere", C.c_int, _vp, _d3, C.c_double, C.c_int)
_sig("pt_scene_moving_sphere", C.c_int, _vp, _d3, _d3, C.c_double, C.c_double, C.c_double, C.c_int)
_sig("pt_scene_rect", C.c_int, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int)
_sig("pt_scene_box", C.c_int, _vp, _d3, _d3, C.c_int)
_sig("pt_scene_flip_normals", C.c_int, _vp, C.c_int)
_sig("pt_scene_translate", C.c_int, _vp, C.c_int, _d3)
_sig("pt_scene_rotate_y", C.c_int, _vp, C.c_int, C.c_double)
_sig("pt_scene_constant_medium", C.c_int, _vp, C.c_int, C.c_double, C.c_int)
_sig("pt_scene_list", C.c_int, _vp, _i32p, C.c_int)
_sig("pt_scene_bvh", C.c_int, _vp, _i32p, C.c_int, C.c_double, C.c_double)
_sig("pt_scene_set_root", C.c_int, _vp, C.c_int)
_sig("pt_scene_set_lights", C.c_int, _vp, _i32p, C.c_int)
_sig("pt_scene_bounding_box", C.c_int, _vp, C.c_int, C.c_double, C.c_double, _d3, _d3)
_sig("pt_scene_flatten_info", C.c_int, C.POINTER(SceneDesc), _i32p, _i32p, _i32p, _i32p)
_sig("pt_tonemap_rgb8", C.c_int, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8))
_sig("pt_tonemap_device", C.c_int, _vp, C.POINTER(C.c_uint8), C.c_size_t)
_sig("pt_display_transform", C.c_int, _vp, C.c_float, C.POINTER(C.c_float), C.c_size_t)
_sig("pt_camera_apply_input", C.c_int, C.POINTER(Camera), C.POINTER(UserInput), C.c_double)
_sig("pt_session_new", C.c_int, _vp, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp))
_sig("pt_session_free", None, _vp)
_sig("pt_session_frame", C.c_int, _vp, C.POINTER(UserInput), C.c_double, C.POINTER(FrameInfo))
_sig("pt_scene_accel_info", C.c_int, C.POINTER(SceneDesc), _i32p, _i32p, _i32p)
_sig("pt_scene_refit", C.c_int, _vp, C.c_double, C.c_double)
_sig("pt_scene_build_device", C.c_int, _vp, C.c_double, C.c_double)
_sig("pt_asset_register_image", C.c_int, C.c_char_p, C.POINTER(C.c_uint8), C.c_int, C.c_int)
_sig("pt_group_unique_id", C.c_int, C.POINTER(C.c_uint8), C.c_size_t)
_sig("pt_group_join", C.c_int, _vp, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_int)
_sig("pt_group_join_local", C.c_int, C.POINTER(_vp), C.c_int)
_sig("pt_group_rows", C.c_int, _vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
_sig("pt_group_layout", C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))
_sig("pt_group_staged_row", C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_uint32)
_sig("pt_device_alloc_count", C.c_uint64)
_sig("pt_group_gather", C.c_int, _vp, C.c_int)
_sig("pt_group_gather_local", C.c_int, C.POINTER(_vp), C.c_int, C.c_int)
_sig("pt_group_leave", C.c_int, _vp)
_sig("pt_group_gather_async", C.c_int, _vp, C.c_int)
_sig("pt_group_wait", C.c_int, _vp)
_sig("pt_group_loopback_async", C.c_int, _vp)
_sig("pt_group_loopback", C.c_int, _vp)
_sig("pt_present_attach", C.c_int, _vp, _vp, C.c_size_t, C.c_uint32)
_sig("pt_present_slot", _vp, _vp, C.c_uint32)
_sig("pt_present_pinned", C.c_int, _vp)
_sig("pt_present_submit", C.c_int, _vp, C.c_uint32, C.c_int, C.c_float)
_sig("pt_present_wait", C.c_int, _vp, C.c_uint32)
_sig("pt_present_detach", C.c_int, _vp)
PT_PRESENT_RAW, PT_PRESENT_DISPLAY, PT_PRESENT_DISPLAY8, PT_PRESENT_DISPLAY8_SRGB = 0, 1, 2, 3
_sig("pt_abi_version", C.c_uint32)
_sig("pt_intersect_static", C.c_int, _vp, _vp, C.c_size_t, C.c_double, C.c_double, C.c_uint64, _vp)
PT_INTERSECT_STATIC_MAX_RAYS = 65536


class PtError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"ptgpu error {code}: {msg}")
        self.code = code


def _vec(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _ints(v):
    return (C.c_int32 * len(v))(*[int(x) for x in v])


def _check_id(rc, what):
    if rc < 0:
        raise PtError(rc, what)
    return rc


EARTHMAP_FIXTURE = os.path.join(os.path.dirname(_HERE), "tests", "golden", "earthmap_1024x512_rgb8.npz")


def register_image(name, rgb8):
    """Hands the decoded RGB8 texels (H x W x 3, rows top-down) of an image file to the library under the file's name
    (pt_asset_register_image); the recipes "textured_sphere" and "final_book_two" need "earthmap.jpg"."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, c = a.shape
    if c != 3:
        raise ValueError("RGB8 texels expected")
    rc = lib.pt_asset_register_image(name.encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
    if rc != PT_OK:
        raise PtError(rc, f"pt_asset_register_image({name})")


def register_earthmap(path=None):
    """Registers the committed decode of the reference's earthmap.jpg (tests/golden/make_earthmap.py wrote it)."""
    register_image("earthmap.jpg", np.load(path or EARTHMAP_FIXTURE)["rgb8"])


class Scene:
    """Host-side scene: a named recipe of the reference or one built node by node (SceneBuilder mirror)."""

    def __init__(self, seed=0x5EED0001):
        self._h = _vp()
        self.camera = None
        rc = lib.pt_scene_new(C.c_uint64(seed), C.byref(self._h))
        if rc != PT_OK:
            raise PtError(rc, "pt_scene_new")

    @classmethod
    def named(cls, name, aspect=1.0, seed=0x5EED0001, param=0):
        self = cls.__new__(cls)
        self._h = _vp()
        cam = Camera()
        rc = lib.pt_scene_build_named(name.encode(), float(aspect), C.c_uint64(seed), C.c_int64(param), C.byref(self._h), C.byref(cam))
        if rc == PT_ERR_MISSING_ASSET:
            raise PtError(rc, f"pt_scene_build_named({name}): the recipe needs earthmap.jpg; call register_earthmap() first")
        if rc != PT_OK:
            raise PtError(rc, f"pt_scene_build_named({name})")
        self.camera = cam
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # at interpreter exit the module globals may already be gone
            lib.pt_scene_free(h)
            self._h = None

    def desc(self) -> SceneDesc:
        d = SceneDesc()
        rc = lib.pt_scene_get_desc(self._h, C.byref(d))
        if rc != PT_OK:
            raise PtError(rc, "pt_scene_get_desc")
        d._owner = self  # the arrays the description points into live as long as the scene does
        return d

    # textures / materials
    def texture_constant(self, rgb): return _check_id(lib.pt_scene_texture_constant(self._h, _vec(rgb)), "texture_constant")
    def texture_checker(self, even, odd): return _check_id(lib.pt_scene_texture_checker(self._h, even, odd), "texture_checker")
    def texture_noise(self, scale): return _check_id(lib.pt_scene_texture_noise(self._h, float(scale)), "texture_noise")

    def texture_image(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w, c = a.shape
        assert c == 3
        return _check_id(lib.pt_scene_texture_image(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h), "texture_image")

    def lambertian(self, texture, emissive=0.0): return _check_id(lib.pt_scene_material_lambertian(self._h, texture, float(emissive)), "lambertian")
    def diffuse_light(self, texture): return _check_id(lib.pt_scene_material_diffuse_light(self._h, texture), "diffuse_light")
    def metal(self, albedo, fuzz): return _check_id(lib.pt_scene_material_metal(self._h, _vec(albedo), float(fuzz)), "metal")
    def dielectric(self, ref_idx): return _check_id(lib.pt_scene_material_dielectric(self._h, float(ref_idx)), "dielectric")

    # hitables
    def sphere(self, center, radius, material): return _check_id(lib.pt_scene_sphere(self._h, _vec(center), float(radius), material), "sphere")

    def moving_sphere(self, c0, c1, t0, t1, radius, material):
        return _check_id(lib.pt_scene_moving_sphere(self._h, _vec(c0), _vec(c1), float(t0), float(t1), float(radius), material), "moving_sphere")

    def rect(self, amin, amax, bmin, bmax, c, axis, material):
        return _check_id(lib.pt_scene_rect(self._h, float(amin), float(amax), float(bmin), float(bmax), float(c), axis, material), "rect")

    def box(self, pmin, pmax, material): return _check_id(lib.pt_scene_box(self._h, _vec(pmin), _vec(pmax), material), "box")
    def flip_normals(self, node): return _check_id(lib.pt_scene_flip_normals(self._h, node), "flip_normals")
    def translate(self, node, t): return _check_id(lib.pt_scene_translate(self._h, node, _vec(t)), "translate")
    def rotate_y(self, node, degrees): return _check_id(lib.pt_scene_rotate_y(self._h, node, float(degrees)), "rotate_y")

    def constant_medium(self, boundary, density, texture):
        return _check_id(lib.pt_scene_constant_medium(self._h, boundary, float(density), texture), "constant_medium")

    def list(self, nodes): return _check_id(lib.pt_scene_list(self._h, _ints(nodes), len(nodes)), "list")

    def bvh(self, nodes, time0=0.0, time1=1.0):
        return _check_id(lib.pt_scene_bvh(self._h, _ints(nodes), len(nodes), float(time0), float(time1)), "bvh")

    def set_root(self, node):
        rc = lib.pt_scene_set_root(self._h, node)
        if rc != PT_OK:
            raise PtError(rc, "set_root")

    def set_lights(self, nodes):
        rc = lib.pt_scene_set_lights(self._h, _ints(nodes), len(nodes))
        if rc != PT_OK:
            raise PtError(rc, "set_lights")

    def bounding_box(self, node, t0=0.0, t1=1.0):
        mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
        rc = lib.pt_scene_bounding_box(self._h, node, float(t0), float(t1), mn, mx)
        if rc != PT_OK:
            raise PtError(rc, "bounding_box")
        return np.array(mn), np.array(mx)

    def accel_info(self):
        """Builds and verifies the fast-mode SAH tree on the host; returns node / primitive counts and depth."""
        d = self.desc()
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib.pt_scene_accel_info(C.byref(d), C.byref(a), C.byref(b), C.byref(c))
        if rc != PT_OK:
            raise PtError(rc, "pt_scene_accel_info")
        return {"n_nodes": a.value, "n_prims": b.value, "depth": c.value}

    def flatten_info(self):
        d = self.desc()
        v = [C.c_int32() for _ in range(4)]
        rc = lib.pt_scene_flatten_info(C.byref(d), *[C.byref(x) for x in v])
        if rc != PT_OK:
            raise PtError(rc, "flatten_info")
        return dict(zip(("n_ops", "n_leaves", "n_xforms", "max_frame_depth"), [x.value for x in v]))


def make_camera(origin, look_at, vup=(0, 1, 0), vfov=40.0, aspect=1.0, aperture=0.0, focus_dist=10.0, time0=0.0, time1=1.0):
    c = Camera()
    c.origin[:] = [float(x) for x in origin]
    c.look_at[:] = [float(x) for x in look_at]
    c.vup[:] = [float(x) for x in vup]
    c.vfov, c.aspect, c.aperture, c.focus_dist, c.time0, c.time1 = map(float, (vfov, aspect, aperture, focus_dist, time0, time1))
    return c


class Context:
    """One GPU's render state (pt_ctx)."""

    def __init__(self, device=0, seed=0x5EED0001, max_paths_in_flight=0, flags=0, **tuning):
        """tuning: fields of pt_tuning by name (pass_form=PT_PASS_PER_BOUNCE, pool_chunk=64, ...)."""
        cfg = Config(C.sizeof(Config), device, seed, max_paths_in_flight, flags, Tuning(**tuning))
        self._h = _vp()
        rc = lib.pt_create(C.byref(cfg), C.byref(self._h))
        if rc != PT_OK:
            raise PtError(rc, "pt_create (no usable HIP device?)" if rc == PT_ERR_NO_DEVICE else "pt_create")
        self.seed = seed
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != PT_OK:
            raise PtError(rc, f"{what}: {lib.pt_last_error(self._h).decode(errors='replace')}")

    def upload(self, scene: Scene):
        d = scene.desc()
        self._check(lib.pt_scene_upload(self._h, C.byref(d)), "pt_scene_upload")
        if scene.camera is not None:
            self.set_camera(scene.camera)

    def upload_desc(self, desc: SceneDesc):
        self._check(lib.pt_scene_upload(self._h, C.byref(desc)), "pt_scene_upload")

    def set_camera(self, cam: Camera):
        self._check(lib.pt_camera_set(self._h, C.byref(cam)), "pt_camera_set")

    def render(self, width, height, spp, max_depth, tile=None, mode=PT_MODE_OFFLINE, frame=0, flags=0, rows=None):
        """rows = (step, phase): trace only rows y0 + phase + m * step of the tile (interleaved split between GPUs)."""
        x0, y0, x1, y1 = tile if tile is not None else (0, 0, width, height)
        step, phase = rows if rows is not None else (1, 0)
        a = RenderArgs(C.sizeof(RenderArgs), width, height, x0, y0, x1, y1, spp, max_depth, mode, frame, flags, step, phase)
        self._check(lib.pt_render(self._h, C.byref(a)), "pt_render")
        self.width, self.height = width, height

    def render_jobs(self, width, height, spp, max_depth, tiles, mode=PT_MODE_OFFLINE, frame=0, flags=0, rows=None, spp_of=None):
        """A batch of tile jobs, one per rectangle (x0, y0, x1, y1) of `tiles` (pt_render_jobs). spp_of: optional list
        of per-job sample counts (a batch that does not agree is run job by job)."""
        step, phase = rows if rows is not None else (1, 0)
        jobs = (RenderArgs * len(tiles))()
        for k, (x0, y0, x1, y1) in enumerate(tiles):
            jobs[k] = RenderArgs(C.sizeof(RenderArgs), width, height, x0, y0, x1, y1, spp_of[k] if spp_of else spp, max_depth, mode,
                                 frame, flags, step, phase)
        self._check(lib.pt_render_jobs(self._h, jobs, len(tiles)), "pt_render_jobs")
        self.width, self.height = width, height

    def refit(self, time0, time1):
        """Refits the fast-mode tree for the shutter window [time0, time1] (pt_scene_refit)."""
        self._check(lib.pt_scene_refit(self._h, float(time0), float(time1)), "pt_scene_refit")

    def build_tree_on_device(self, time0, time1):
        """Builds the fast-mode tree on the device for the shutter window [time0, time1] (pt_scene_build_device)."""
        self._check(lib.pt_scene_build_device(self._h, float(time0), float(time1)), "pt_scene_build_device")

    def wait(self):
        """Blocks until everything issued on the context has completed (renders with PT_RENDER_NO_WAIT included)."""
        self._check(lib.pt_wait(self._h), "pt_wait")

    def reset(self):
        self._check(lib.pt_accum_reset(self._h), "pt_accum_reset")

    def readback(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(lib.pt_readback(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "pt_readback")
        return out

    def bind_image(self, device_ptr, width, height):
        self._check(lib.pt_image_bind(self._h, _vp(device_ptr), width, height), "pt_image_bind")
        self.width, self.height = width, height

    def image_device_ptr(self):
        return lib.pt_image_device_ptr(self._h)

    def stream(self):
        return lib.pt_stream(self._h)

    def intersect(self, rays, t_min=0.001, t_max=np.finfo(np.float64).max, medium_seed=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._check(
            lib.pt_intersect(self._h, rays.ctypes.data, rays.shape[0], float(t_min), float(t_max), C.c_uint64(medium_seed), hits.ctypes.data),
            "pt_intersect",
        )
        return hits

    def intersect_device(self, d_rays_soa_ptr, n, t_min=0.001, t_max=np.finfo(np.float64).max, medium_seed=0, repeats=1):
        ms = C.c_double()
        self._check(
            lib.pt_intersect_device(self._h, _vp(d_rays_soa_ptr), n, float(t_min), float(t_max), C.c_uint64(medium_seed), repeats, C.byref(ms)),
            "pt_intersect_device",
        )
        return ms.value

    def intersect_static(self, d_rays_ptr, n, d_hits_ptr, t_min=0.001, t_max=np.finfo(np.float64).max, medium_seed=0):
        """pt_intersect on caller-owned DEVICE arrays of n pt_ray / n pt_hit, through the library's fixed scratch
        array: no device allocation (pt_intersect_static)."""
        self._check(
            lib.pt_intersect_static(self._h, _vp(d_rays_ptr), n, float(t_min), float(t_max), C.c_uint64(medium_seed), _vp(d_hits_ptr)),
            "pt_intersect_static",
        )

    # ---- groups of contexts, one per GPU (pt_group_*) ----
    def group_join(self, unique_id: bytes, rank: int, world: int):
        """Collective over `world` processes; `unique_id` comes from group_unique_id() on one of them."""
        buf = (C.c_uint8 * PT_GROUP_ID_BYTES).from_buffer_copy(unique_id)
        self._check(lib.pt_group_join(self._h, buf, PT_GROUP_ID_BYTES, rank, world), "pt_group_join")

    def group_rows(self):
        """(row_step, row_phase) for pt_render: this member's rows of the image."""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(lib.pt_group_rows(self._h, C.byref(a), C.byref(b)), "pt_group_rows")
        return a.value, b.value

    def group_gather(self, root=0):
        self._check(lib.pt_group_gather(self._h, root), "pt_group_gather")

    def group_gather_async(self, root=0):
        """Queues the gather of the frame just traced and returns; the root's readers then see the assembled frame."""
        self._check(lib.pt_group_gather_async(self._h, root), "pt_group_gather_async")

    def group_wait(self):
        self._check(lib.pt_group_wait(self._h), "pt_group_wait")

    def group_loopback_async(self):
        self._check(lib.pt_group_loopback_async(self._h), "pt_group_loopback_async")

    def group_leave(self):
        self._check(lib.pt_group_leave(self._h), "pt_group_leave")

    def group_loopback(self):
        self._check(lib.pt_group_loopback(self._h), "pt_group_loopback")

    def tonemap_rgb8(self):
        """Reinhard + sqrt gamma + u8 of the context's image on the device, top row first (pt_tonemap_device)."""
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._check(lib.pt_tonemap_device(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size), "pt_tonemap_device")
        return out

    def display_transform(self, exposure=1.3):
        """ACES-fitted display transform with exposure and vignette (pt_display_transform): f32 RGBA, top row first."""
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(lib.pt_display_transform(self._h, float(exposure), out.ctypes.data_as(C.POINTER(C.c_float)), out.size),
                    "pt_display_transform")
        return out

    def present_attach(self, staging=None, slots=2, width=None, height=None):
        """Attaches the presentation ring (pt_present_attach) and returns it as a float32 array of shape
        (slots, height, width, 4). `staging`: a C-contiguous array of that shape owned by the caller (it must outlive
        the attachment); None: the library allocates page-locked memory for `slots` frames of width x height."""
        if staging is not None:
            assert staging.dtype == np.float32 and staging.flags["C_CONTIGUOUS"] and staging.ndim == 4
            self._check(lib.pt_present_attach(self._h, staging.ctypes.data, staging[0].nbytes, staging.shape[0]), "pt_present_attach")
            self._present_staging = staging
            return staging
        width, height = width or self.width, height or self.height
        slot_bytes = width * height * 16
        self._check(lib.pt_present_attach(self._h, None, slot_bytes, slots), "pt_present_attach")
        base = lib.pt_present_slot(self._h, 0)
        ring = np.ctypeslib.as_array((C.c_float * (slots * height * width * 4)).from_address(base)).reshape(slots, height, width, 4)
        self._present_staging = ring  # a view of the library's memory: invalid after present_detach / close
        return ring

    def present_slot_rgba8(self, slot, width=None, height=None):
        """A slot of the ring viewed as the (height, width, 4) uint8 frame the 8-bit present kinds write."""
        width, height = width or self.width, height or self.height
        base = lib.pt_present_slot(self._h, slot)
        return np.ctypeslib.as_array((C.c_uint8 * (height * width * 4)).from_address(base)).reshape(height, width, 4)

    def present_pinned(self):
        return bool(lib.pt_present_pinned(self._h))

    def present_submit(self, slot, kind=PT_PRESENT_RAW, exposure=1.3):
        self._check(lib.pt_present_submit(self._h, slot, kind, float(exposure)), "pt_present_submit")

    def present_wait(self, slot):
        self._check(lib.pt_present_wait(self._h, slot), "pt_present_wait")

    def present_detach(self):
        self._check(lib.pt_present_detach(self._h), "pt_present_detach")
        self._present_staging = None

    def stats(self) -> Stats:
        s = Stats()
        self._check(lib.pt_stats_get(self._h, C.byref(s)), "pt_stats_get")
        return s

    def stats_reset(self):
        self._check(lib.pt_stats_reset(self._h), "pt_stats_reset")

    def last_pass_records(self) -> int:
        """PT_RECORDS_* bits of the most recent pass (pt_last_pass_records)."""
        rc = lib.pt_last_pass_records(self._h)
        if rc < 0:
            self._check(rc, "pt_last_pass_records")
        return rc


def group_unique_id() -> bytes:
    """The identifier one member creates and every member passes to Context.group_join (pt_group_unique_id)."""
    buf = (C.c_uint8 * PT_GROUP_ID_BYTES)()
    rc = lib.pt_group_unique_id(buf, PT_GROUP_ID_BYTES)
    if rc != PT_OK:
        raise PtError(rc, "pt_group_unique_id")
    return bytes(buf)


def group_layout(height, world, root, rank):
    """(rows member `rank` traces, rows before them in the root's staging buffer): pt_group_layout."""
    rows, before = C.c_uint32(), C.c_uint64()
    rc = lib.pt_group_layout(height, world, root, rank, C.byref(rows), C.byref(before))
    if rc != PT_OK:
        raise PtError(rc, "pt_group_layout")
    return rows.value, before.value


def device_alloc_count() -> int:
    """Device / page-locked allocations the library has made in this process so far (pt_device_alloc_count)."""
    return int(lib.pt_device_alloc_count())


def group_join_local(contexts):
    """One process, one context per GPU: contexts[k] becomes member k (pt_group_join_local)."""
    arr = (_vp * len(contexts))(*[c._h for c in contexts])
    rc = lib.pt_group_join_local(arr, len(contexts))
    if rc != PT_OK:
        raise PtError(rc, "pt_group_join_local: " + lib.pt_last_error(contexts[0]._h).decode(errors="replace"))


def group_gather_local(contexts, root=0):
    arr = (_vp * len(contexts))(*[c._h for c in contexts])
    rc = lib.pt_group_gather_local(arr, len(contexts), root)
    if rc != PT_OK:
        raise PtError(rc, "pt_group_gather_local: " + lib.pt_last_error(contexts[0]._h).decode(errors="replace"))


def copy_camera(cam: Camera) -> Camera:
    out = Camera()
    C.memmove(C.byref(out), C.byref(cam), C.sizeof(Camera))
    return out


def camera_apply_input(cam: Camera, keys_pressed=0, keys_held=0, mouse=(0.0, 0.0), frame_time=1.0 / 60.0):
    """Camera::update_from_input + update on a copy of `cam`; returns (moved, new camera)."""
    out = copy_camera(cam)
    ui = UserInput(keys_pressed, keys_held, mouse[0], mouse[1])
    rc = lib.pt_camera_apply_input(C.byref(out), C.byref(ui), float(frame_time))
    if rc < 0:
        raise PtError(rc, "pt_camera_apply_input")
    return bool(rc), out


class Session:
    """Headless replay of the realtime main loop (pt_session_*): one call to frame() per displayed frame."""

    def __init__(self, ctx: "Context", camera: Camera, width, height, max_depth=10):
        self._ctx = ctx
        self._h = _vp()
        rc = lib.pt_session_new(ctx._h, C.byref(camera), width, height, max_depth, C.byref(self._h))
        if rc != PT_OK:
            raise PtError(rc, "pt_session_new")
        ctx.width, ctx.height = width, height

    def frame(self, keys_pressed=0, keys_held=0, mouse=(0.0, 0.0), frame_time=1.0 / 60.0) -> FrameInfo:
        ui = UserInput(keys_pressed, keys_held, mouse[0], mouse[1])
        info = FrameInfo()
        self._ctx._check(lib.pt_session_frame(self._h, C.byref(ui), float(frame_time), C.byref(info)), "pt_session_frame")
        return info

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.pt_session_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def tonemap_rgb8(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w, _ = rgba.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    rc = lib.pt_tonemap_rgb8(rgba.ctypes.data_as(C.POINTER(C.c_float)), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != PT_OK:
        raise PtError(rc, "pt_tonemap_rgb8")
    return out


def write_ppm(path, rgb8):
    """P6 writer with the header of src/lib.rs:603."""
    h, w, _ = rgb8.shape
    with open(path, "wb") as f:
        f.write(f"P6 {w} {h} 255\n".encode())
        f.write(np.ascontiguousarray(rgb8, dtype=np.uint8).tobytes())

