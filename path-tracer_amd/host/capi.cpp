// extern "C" surface of the host-side scene construction layer (declared in include/ptgpu.h).
#include "ptscene.hpp"

#include <cmath>
#include <cstring>
#include <new>

using namespace ptgpu;

namespace {
inline bool valid_node(const pt_scene *s, int n) { return s && n >= 0 && n < s->scene.n_hitables(); }
inline Vec3 v3(const double p[3]) { return Vec3(p[0], p[1], p[2]); }
inline int ok_or_err(const pt_scene *s, int id) { return s->scene.error().empty() ? id : (int)PT_ERR_INVALID_ARG; }
} // namespace

extern "C" {

uint32_t pt_abi_version(void) { return PT_ABI_VERSION; }

int pt_scene_new(uint64_t seed, pt_scene **out) {
    if (!out) return PT_ERR_INVALID_ARG;
    pt_scene *s = new (std::nothrow) pt_scene(seed);
    if (!s) return PT_ERR_OUT_OF_MEMORY;
    *out = s;
    return PT_OK;
}

int pt_scene_build_named(const char *name, double aspect, uint64_t seed, int64_t param, pt_scene **out,
                         pt_camera *camera_out) {
    if (!name || !out || !camera_out) return PT_ERR_INVALID_ARG;
    pt_scene *s = new (std::nothrow) pt_scene(seed);
    if (!s) return PT_ERR_OUT_OF_MEMORY;
    std::string error;
    bool ok = false;
    try {
        ok = build_named_scene(name, aspect, seed, param, s->scene, *camera_out, error);
    } catch (const std::bad_alloc &) {
        delete s;
        return PT_ERR_OUT_OF_MEMORY;
    } catch (...) {
        ok = false;
    }
    if (!ok) {
        delete s;
        return error.rfind("missing asset", 0) == 0 ? PT_ERR_MISSING_ASSET : PT_ERR_INVALID_ARG;
    }
    *out = s;
    return PT_OK;
}

int pt_asset_register_image(const char *name, const uint8_t *rgb8, int width, int height) {
    if (!name || !rgb8 || width <= 0 || height <= 0) return PT_ERR_INVALID_ARG;
    try {
        return register_image_asset(name, rgb8, width, height) ? PT_OK : PT_ERR_INVALID_ARG;
    } catch (const std::bad_alloc &) {
        return PT_ERR_OUT_OF_MEMORY;
    }
}

void pt_scene_free(pt_scene *scene) { delete scene; }

int pt_scene_get_desc(const pt_scene *scene, pt_scene_desc *out) {
    if (!scene || !out) return PT_ERR_INVALID_ARG;
    *out = scene->scene.desc();
    return PT_OK;
}

int pt_scene_texture_constant(pt_scene *s, const double rgb[3]) {
    if (!s || !rgb) return PT_ERR_INVALID_ARG;
    return s->scene.constant_texture(v3(rgb)).id;
}
int pt_scene_texture_checker(pt_scene *s, int even, int odd) {
    if (!s) return PT_ERR_INVALID_ARG;
    int id = s->scene.checker_texture(Texture{even}, Texture{odd}).id;
    return ok_or_err(s, id);
}
int pt_scene_texture_noise(pt_scene *s, double scale) {
    if (!s) return PT_ERR_INVALID_ARG;
    return s->scene.noise_texture(scale).id;
}
int pt_scene_texture_image(pt_scene *s, const uint8_t *rgb8, int width, int height) {
    if (!s || !rgb8 || width <= 0 || height <= 0) return PT_ERR_INVALID_ARG;
    return s->scene.image_texture(rgb8, width, height).id;
}
int pt_scene_material_lambertian(pt_scene *s, int texture, double emissive) {
    if (!s || texture < 0) return PT_ERR_INVALID_ARG;
    return s->scene.lambertian(Texture{texture}, emissive).id;
}
int pt_scene_material_diffuse_light(pt_scene *s, int texture) {
    if (!s || texture < 0) return PT_ERR_INVALID_ARG;
    return s->scene.diffuse_light(Texture{texture}).id;
}
int pt_scene_material_metal(pt_scene *s, const double albedo[3], double fuzz) {
    if (!s || !albedo) return PT_ERR_INVALID_ARG;
    return s->scene.metal(v3(albedo), fuzz).id;
}
int pt_scene_material_dielectric(pt_scene *s, double ref_idx) {
    if (!s) return PT_ERR_INVALID_ARG;
    return s->scene.dielectric(ref_idx).id;
}
int pt_scene_sphere(pt_scene *s, const double center[3], double radius, int material) {
    if (!s || !center) return PT_ERR_INVALID_ARG;
    return s->scene.sphere(v3(center), radius, Material{material}).id;
}
int pt_scene_moving_sphere(pt_scene *s, const double c0[3], const double c1[3], double t0, double t1,
                           double radius, int material) {
    if (!s || !c0 || !c1) return PT_ERR_INVALID_ARG;
    return s->scene.moving_sphere(v3(c0), v3(c1), t0, t1, radius, Material{material}).id;
}
int pt_scene_rect(pt_scene *s, double amin, double amax, double bmin, double bmax, double c, int axis,
                  int material) {
    if (!s || axis < 0 || axis > 2) return PT_ERR_INVALID_ARG;
    // AxisAlignedRect::new panics on a degenerate rectangle (src/rect.rs:39-44); here that is an error code.
    if (std::fabs(amin - amax) < 0.0000001 || std::fabs(bmin - bmax) < 0.0000001) return PT_ERR_INVALID_ARG;
    return s->scene.rect(amin, amax, bmin, bmax, c, (Axis)axis, Material{material}).id;
}
int pt_scene_box(pt_scene *s, const double pmin[3], const double pmax[3], int material) {
    if (!s || !pmin || !pmax) return PT_ERR_INVALID_ARG;
    for (int i = 0; i < 3; ++i)
        if (std::fabs(pmin[i] - pmax[i]) < 0.0000001) return PT_ERR_INVALID_ARG;
    return s->scene.axis_aligned_box(v3(pmin), v3(pmax), Material{material}).id;
}
int pt_scene_flip_normals(pt_scene *s, int node) {
    if (!valid_node(s, node)) return PT_ERR_INVALID_ARG;
    return s->scene.flip_normals(Hitable{node}).id;
}
int pt_scene_translate(pt_scene *s, int node, const double t[3]) {
    if (!valid_node(s, node) || !t) return PT_ERR_INVALID_ARG;
    return s->scene.translate(Hitable{node}, v3(t)).id;
}
int pt_scene_rotate_y(pt_scene *s, int node, double degrees) {
    if (!valid_node(s, node)) return PT_ERR_INVALID_ARG;
    Aabb bb;
    if (!s->scene.bounding_box(Hitable{node}, 0.0, 1.0, bb)) return PT_ERR_INVALID_ARG;
    return s->scene.rotate_y(Hitable{node}, degrees).id;
}
int pt_scene_constant_medium(pt_scene *s, int boundary, double density, int texture) {
    if (!valid_node(s, boundary) || texture < 0) return PT_ERR_INVALID_ARG;
    return s->scene.constant_medium(Hitable{boundary}, density, Texture{texture}).id;
}
int pt_scene_list(pt_scene *s, const int32_t *nodes, int n) {
    if (!s || (n > 0 && !nodes) || n < 0) return PT_ERR_INVALID_ARG;
    std::vector<Hitable> list;
    for (int i = 0; i < n; ++i) {
        if (!valid_node(s, nodes[i])) return PT_ERR_INVALID_ARG;
        list.push_back(Hitable{nodes[i]});
    }
    return s->scene.hitable_list(list).id;
}
int pt_scene_bvh(pt_scene *s, const int32_t *nodes, int n, double time0, double time1) {
    if (!s || !nodes || n <= 0) return PT_ERR_INVALID_ARG;
    std::vector<Hitable> list;
    for (int i = 0; i < n; ++i) {
        if (!valid_node(s, nodes[i])) return PT_ERR_INVALID_ARG;
        Aabb bb;
        if (!s->scene.bounding_box(Hitable{nodes[i]}, 0.0, 0.0, bb)) return PT_ERR_INVALID_ARG;
        list.push_back(Hitable{nodes[i]});
    }
    int id = s->scene.bvh_from_list(list, time0, time1).id;
    return id < 0 ? (int)PT_ERR_INVALID_ARG : id;
}
int pt_scene_set_root(pt_scene *s, int node) {
    if (!valid_node(s, node)) return PT_ERR_INVALID_ARG;
    s->scene.set_root(Hitable{node});
    return PT_OK;
}
int pt_scene_set_lights(pt_scene *s, const int32_t *nodes, int n) {
    if (!s || (n > 0 && !nodes) || n < 0) return PT_ERR_INVALID_ARG;
    std::vector<Hitable> list;
    for (int i = 0; i < n; ++i) {
        if (!valid_node(s, nodes[i])) return PT_ERR_INVALID_ARG;
        list.push_back(Hitable{nodes[i]});
    }
    s->scene.set_lights(list);
    return PT_OK;
}
int pt_scene_bounding_box(const pt_scene *s, int node, double t0, double t1, double min_out[3],
                          double max_out[3]) {
    if (!valid_node(s, node) || !min_out || !max_out) return PT_ERR_INVALID_ARG;
    Aabb bb;
    if (!s->scene.bounding_box(Hitable{node}, t0, t1, bb)) return PT_ERR_INVALID_ARG;
    min_out[0] = bb.min.x; min_out[1] = bb.min.y; min_out[2] = bb.min.z;
    max_out[0] = bb.max.x; max_out[1] = bb.max.y; max_out[2] = bb.max.z;
    return PT_OK;
}

// reinhard_tonemap (src/trace.rs:236-242), gamma + `as u8` (src/lib.rs:330-340), PPM row flip (src/lib.rs:617-630)
int pt_camera_apply_input(pt_camera *c, const pt_user_input *input, double frame_time) {
    if (!c || !input) return PT_ERR_INVALID_ARG;
    ptgpu::Camera cam(ptgpu::Vec3{c->origin[0], c->origin[1], c->origin[2]},
                      ptgpu::Vec3{c->look_at[0], c->look_at[1], c->look_at[2]},
                      ptgpu::Vec3{c->vup[0], c->vup[1], c->vup[2]}, c->vfov, c->aspect, c->aperture, c->focus_dist,
                      c->time0, c->time1);
    const bool moved = cam.update_from_input(*input, frame_time);
    if (moved) cam.update();
    *c = cam.params();
    return moved ? 1 : 0;
}

struct pt_session {
    ptgpu::RealtimeSession session;
    pt_session(pt_ctx *ctx, const pt_camera &cam, uint32_t w, uint32_t h, uint32_t d) : session(ctx, cam, w, h, d) {}
};
int pt_session_new(pt_ctx *ctx, const pt_camera *cam, uint32_t width, uint32_t height, uint32_t max_depth,
                   pt_session **out) {
    if (!ctx || !cam || !out || width == 0 || height == 0) return PT_ERR_INVALID_ARG;
    *out = new (std::nothrow) pt_session(ctx, *cam, width, height, max_depth);
    return *out ? PT_OK : PT_ERR_OUT_OF_MEMORY;
}
void pt_session_free(pt_session *s) { delete s; }
int pt_session_frame(pt_session *s, const pt_user_input *input, double frame_time, pt_frame_info *info) {
    if (!s || !input) return PT_ERR_INVALID_ARG;
    return s->session.frame(*input, frame_time, info);
}

int pt_tonemap_rgb8(const float *rgba, uint32_t width, uint32_t height, uint8_t *rgb8_out) {
    if (!rgba || !rgb8_out) return PT_ERR_INVALID_ARG;
    const double exposure = 1.5;
    for (uint32_t j = 0; j < height; ++j) {
        const uint32_t j_flipped = height - j - 1;
        for (uint32_t i = 0; i < width; ++i) {
            const float *px = rgba + ((size_t)j_flipped * width + i) * 4;
            uint8_t *dst = rgb8_out + ((size_t)j * width + i) * 3;
            for (int c = 0; c < 3; ++c) {
                double colour = (double)px[c] * exposure;
                double mapped = colour / (colour + 1.0);
                double scaled = 255.99 * std::sqrt(mapped);
                if (!(scaled > 0.0)) dst[c] = 0; // NaN and negatives saturate to 0 like Rust's `as u8`
                else if (scaled >= 255.0) dst[c] = 255;
                else dst[c] = (uint8_t)scaled;
            }
        }
    }
    return PT_OK;
}

} // extern "C"
