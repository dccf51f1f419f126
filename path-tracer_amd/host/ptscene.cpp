// Scene arena, builders and camera of the host layer (see ptscene.hpp).
#include "ptscene.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace ptgpu {

namespace {
const double kPi = 3.14159265358979323846264338327950288;
const uint64_t kGolden = 0x9E3779B97F4A7C15ULL;
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline double ffmin(double a, double b) { return a < b ? a : b; } // aabb.rs:70-72
inline double ffmax(double a, double b) { return a > b ? a : b; } // aabb.rs:66-68
inline Aabb aabb_union(const Aabb &a, const Aabb &b) {           // aabb.rs:49-56
    return Aabb{Vec3(ffmin(a.min.x, b.min.x), ffmin(a.min.y, b.min.y), ffmin(a.min.z, b.min.z)),
                Vec3(ffmax(a.max.x, b.max.x), ffmax(a.max.y, b.max.y), ffmax(a.max.z, b.max.z))};
}
pt_hitable blank(int kind) {
    pt_hitable h;
    std::memset(&h, 0, sizeof(h));
    h.kind = kind;
    h.material = -1;
    h.child[0] = h.child[1] = -1;
    return h;
}
} // namespace

double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vec3 cross(Vec3 a, Vec3 b) { // vec3.rs:137-143
    return Vec3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
double length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
Vec3 unit_vector(Vec3 a) { return a / length(a); }

SceneRng::SceneRng(uint64_t seed) : key_(mix64(seed ^ 0x5343454E45ULL)) {}
double SceneRng::rand() {
    n_ += 1;
    return (double)(mix64(key_ + n_ * kGolden) >> 11) * (1.0 / 9007199254740992.0);
}

Scene::Scene(uint64_t seed) : rng_(seed) { std::memset(&perlin_, 0, sizeof(perlin_)); }

void Scene::fail(const std::string &msg) {
    if (error_.empty()) error_ = msg;
}
int Scene::push(const pt_hitable &h) {
    hitables_.push_back(h);
    return (int)hitables_.size() - 1;
}

// ---------------------------------------------------------------- textures
Texture Scene::constant_texture(Vec3 colour) {
    pt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = PT_T_CONSTANT;
    t.even = t.odd = -1;
    t.colour[0] = colour.x;
    t.colour[1] = colour.y;
    t.colour[2] = colour.z;
    textures_.push_back(t);
    return Texture{(int)textures_.size() - 1};
}
Texture Scene::checker_texture(Texture even, Texture odd) {
    pt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = PT_T_CHECKER;
    t.even = even.id;
    t.odd = odd.id;
    if (even.id < 0 || odd.id < 0 || even.id >= (int)textures_.size() || odd.id >= (int)textures_.size())
        fail("checker_texture: bad child texture");
    textures_.push_back(t);
    return Texture{(int)textures_.size() - 1};
}
Texture Scene::noise_texture(double scale) {
    if (!have_perlin_) {
        // perlin_generate, noise.rs:47-57: 256 unit vectors from uniform [-1,1)^3
        for (int i = 0; i < 256; ++i) {
            double x = 2.0 * rng_.rand() - 1.0;
            double y = 2.0 * rng_.rand() - 1.0;
            double z = 2.0 * rng_.rand() - 1.0;
            Vec3 u = unit_vector(Vec3(x, y, z));
            perlin_.ranvec[i][0] = u.x;
            perlin_.ranvec[i][1] = u.y;
            perlin_.ranvec[i][2] = u.z;
        }
        // perlin_generate_perm + permute, noise.rs:59-76
        int32_t *perms[3] = {perlin_.perm_x, perlin_.perm_y, perlin_.perm_z};
        for (int k = 0; k < 3; ++k) {
            int32_t *p = perms[k];
            for (int i = 0; i < 256; ++i) p[i] = i;
            for (int i = 255; i >= 0; --i) {
                int target = (int)(rng_.rand() * (double)(i + 1));
                std::swap(p[i], p[target]);
            }
        }
        have_perlin_ = true;
    }
    pt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = PT_T_NOISE;
    t.even = t.odd = -1;
    t.scale = scale;
    textures_.push_back(t);
    return Texture{(int)textures_.size() - 1};
}
Texture Scene::image_texture(const uint8_t *rgb8, int width, int height) {
    pt_texture t;
    std::memset(&t, 0, sizeof(t));
    t.kind = PT_T_IMAGE;
    t.even = t.odd = -1;
    t.width = width;
    t.height = height;
    t.texel_offset = texels_.size();
    if (!rgb8 || width <= 0 || height <= 0) {
        fail("image_texture: empty image");
    } else {
        texels_.insert(texels_.end(), rgb8, rgb8 + (size_t)width * height * 3);
    }
    textures_.push_back(t);
    return Texture{(int)textures_.size() - 1};
}

// --------------------------------------------------------------- materials
static pt_material blank_material(int kind) {
    pt_material m;
    std::memset(&m, 0, sizeof(m));
    m.kind = kind;
    m.texture = -1;
    m.ref_idx = 1.0;
    return m;
}
Material Scene::lambertian(Texture albedo, double emissive) {
    pt_material m = blank_material(PT_M_LAMBERTIAN);
    m.texture = albedo.id;
    m.emissive = emissive;
    if (albedo.id < 0) fail("lambertian: no texture");
    materials_.push_back(m);
    return Material{(int)materials_.size() - 1};
}
Material Scene::diffuse_light(Texture texture) {
    pt_material m = blank_material(PT_M_DIFFUSE_LIGHT);
    m.texture = texture.id;
    if (texture.id < 0) fail("diffuse_light: no texture");
    materials_.push_back(m);
    return Material{(int)materials_.size() - 1};
}
Material Scene::metal(Vec3 albedo, double fuzz) {
    pt_material m = blank_material(PT_M_METAL);
    m.albedo[0] = albedo.x;
    m.albedo[1] = albedo.y;
    m.albedo[2] = albedo.z;
    m.fuzz = std::fmin(fuzz, 1.0); // material.rs:224
    materials_.push_back(m);
    return Material{(int)materials_.size() - 1};
}
Material Scene::dielectric(double ref_idx) {
    pt_material m = blank_material(PT_M_DIELECTRIC);
    m.ref_idx = ref_idx;
    materials_.push_back(m);
    return Material{(int)materials_.size() - 1};
}
Material Scene::dummy_material() {
    materials_.push_back(blank_material(PT_M_DUMMY));
    return Material{(int)materials_.size() - 1};
}

// ---------------------------------------------------------------- hitables
Hitable Scene::sphere(Vec3 center, double radius, Material m) {
    pt_hitable h = blank(PT_H_SPHERE);
    h.material = m.id;
    h.f[0] = center.x;
    h.f[1] = center.y;
    h.f[2] = center.z;
    h.f[3] = radius;
    return Hitable{push(h)};
}
Hitable Scene::moving_sphere(Vec3 c0, Vec3 c1, double t0, double t1, double radius, Material m) {
    pt_hitable h = blank(PT_H_MOVING_SPHERE);
    h.material = m.id;
    Vec3 range = c1 - c0;
    h.f[0] = c0.x;
    h.f[1] = c0.y;
    h.f[2] = c0.z;
    h.f[3] = radius;
    h.f[4] = range.x;
    h.f[5] = range.y;
    h.f[6] = range.z;
    h.f[7] = t0;
    h.f[8] = t1 - t0;
    return Hitable{push(h)};
}
Hitable Scene::rect(double amin, double amax, double bmin, double bmax, double c, Axis axis, Material m) {
    const double tolerance = 0.0000001; // rect.rs:26
    if (std::fabs(amin - amax) < tolerance) fail("rect: amin == amax, no axis-aligned plane supplied");
    if (std::fabs(bmin - bmax) < tolerance) fail("rect: bmin == bmax, no axis-aligned plane supplied");
    pt_hitable h = blank(PT_H_RECT);
    h.material = m.id;
    h.axis = (int)axis;
    h.f[0] = amin;
    h.f[1] = amax;
    h.f[2] = bmin;
    h.f[3] = bmax;
    h.f[4] = c;
    return Hitable{push(h)};
}
Hitable Scene::axis_aligned_box(Vec3 pmin, Vec3 pmax, Material m) {
    // six faces in the order of axis_aligned_box.rs:17-22: +Z, -Z (flipped), +Y, -Y (flipped), +X, -X (flipped)
    Hitable faces[6];
    faces[0] = rect(pmin.x, pmax.x, pmin.y, pmax.y, pmax.z, Axis::Z, m);
    faces[1] = flip_normals(rect(pmin.x, pmax.x, pmin.y, pmax.y, pmin.z, Axis::Z, m));
    faces[2] = rect(pmin.x, pmax.x, pmin.z, pmax.z, pmax.y, Axis::Y, m);
    faces[3] = flip_normals(rect(pmin.x, pmax.x, pmin.z, pmax.z, pmin.y, Axis::Y, m));
    faces[4] = rect(pmin.y, pmax.y, pmin.z, pmax.z, pmax.x, Axis::X, m);
    faces[5] = flip_normals(rect(pmin.y, pmax.y, pmin.z, pmax.z, pmin.x, Axis::X, m));
    pt_hitable h = blank(PT_H_BOX);
    h.material = m.id;
    h.list_start = (int)child_index_.size();
    h.list_count = 6;
    for (auto &f : faces) child_index_.push_back(f.id);
    h.f[0] = pmin.x;
    h.f[1] = pmin.y;
    h.f[2] = pmin.z;
    h.f[3] = pmax.x;
    h.f[4] = pmax.y;
    h.f[5] = pmax.z;
    return Hitable{push(h)};
}
Hitable Scene::flip_normals(Hitable child) {
    pt_hitable h = blank(PT_H_FLIP);
    h.child[0] = child.id;
    return Hitable{push(h)};
}
Hitable Scene::translate(Hitable child, Vec3 t) {
    pt_hitable h = blank(PT_H_TRANSLATE);
    h.child[0] = child.id;
    h.f[0] = t.x;
    h.f[1] = t.y;
    h.f[2] = t.z;
    return Hitable{push(h)};
}
Hitable Scene::rotate_y(Hitable child, double angle) {
    // RotateY::new, hitable.rs:148-180
    double radians = angle * (kPi / 180.0);
    double sin_theta = std::sin(radians);
    double cos_theta = std::cos(radians);
    Aabb bb;
    if (!bounding_box(child, 0.0, 1.0, bb)) fail("rotate_y: child has no bounding box");
    Vec3 mn = Vec3::from_float(DBL_MAX), mx = Vec3::from_float(-DBL_MAX);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                double fi = i, fj = j, fk = k;
                double x = fi * bb.max.x + (1.0 - fi) * bb.min.x;
                double y = fj * bb.max.y + (1.0 - fj) * bb.min.y;
                double z = fk * bb.max.z + (1.0 - fk) * bb.min.z;
                double new_x = cos_theta * x + sin_theta * z;
                double new_z = -sin_theta * x + cos_theta * z;
                mn = Vec3(std::fmin(new_x, mn.x), std::fmin(y, mn.y), std::fmin(new_z, mn.z));
                mx = Vec3(std::fmax(new_x, mx.x), std::fmax(y, mx.y), std::fmax(new_z, mx.z));
            }
    pt_hitable h = blank(PT_H_ROTATE_Y);
    h.child[0] = child.id;
    h.f[0] = sin_theta;
    h.f[1] = cos_theta;
    h.f[2] = mn.x;
    h.f[3] = mn.y;
    h.f[4] = mn.z;
    h.f[5] = mx.x;
    h.f[6] = mx.y;
    h.f[7] = mx.z;
    return Hitable{push(h)};
}
Hitable Scene::constant_medium(Hitable boundary, double density, Texture texture) {
    pt_material m = blank_material(PT_M_ISOTROPIC); // volume.rs:15
    m.texture = texture.id;
    materials_.push_back(m);
    pt_hitable h = blank(PT_H_MEDIUM);
    h.child[0] = boundary.id;
    h.material = (int)materials_.size() - 1;
    h.f[0] = density;
    return Hitable{push(h)};
}
Hitable Scene::hitable_list(const std::vector<Hitable> &list) {
    pt_hitable h = blank(PT_H_LIST);
    h.list_start = (int)child_index_.size();
    h.list_count = (int)list.size();
    for (auto &c : list) child_index_.push_back(c.id);
    return Hitable{push(h)};
}

bool Scene::bounding_box(Hitable hh, double t0, double t1, Aabb &out) const {
    if (hh.id < 0 || hh.id >= (int)hitables_.size()) return false;
    const pt_hitable &n = hitables_[hh.id];
    switch (n.kind) {
    case PT_H_SPHERE: { // sphere.rs:83-85
        Vec3 c(n.f[0], n.f[1], n.f[2]);
        out = Aabb{c - Vec3::from_float(n.f[3]), c + Vec3::from_float(n.f[3])};
        return true;
    }
    case PT_H_MOVING_SPHERE: { // sphere.rs:130-132,177-183
        Vec3 c0(n.f[0], n.f[1], n.f[2]), range(n.f[4], n.f[5], n.f[6]);
        Vec3 a = c0 + ((t0 - n.f[7]) / n.f[8]) * range;
        Vec3 b = c0 + ((t1 - n.f[7]) / n.f[8]) * range;
        Vec3 r = Vec3::from_float(n.f[3]);
        out = aabb_union(Aabb{a - r, a + r}, Aabb{b - r, b + r});
        return true;
    }
    case PT_H_RECT: { // rect.rs:116-122
        double amin = n.f[0], amax = n.f[1], bmin = n.f[2], bmax = n.f[3], c = n.f[4];
        if (n.axis == 0)
            out = Aabb{Vec3(c - 0.0001, amin, bmin), Vec3(c + 0.0001, amax, bmax)};
        else if (n.axis == 1)
            out = Aabb{Vec3(amin, c - 0.0001, bmin), Vec3(amax, c + 0.0001, bmax)};
        else
            out = Aabb{Vec3(amin, bmin, c - 0.0001), Vec3(amax, bmax, c + 0.0001)};
        return true;
    }
    case PT_H_BOX:
    case PT_H_BVH:
        out = Aabb{Vec3(n.f[0], n.f[1], n.f[2]), Vec3(n.f[3], n.f[4], n.f[5])};
        return true;
    case PT_H_LIST:
        return false;
    case PT_H_FLIP:
    case PT_H_MEDIUM:
        return bounding_box(Hitable{n.child[0]}, t0, t1, out);
    case PT_H_TRANSLATE: { // hitable.rs:133-137
        if (!bounding_box(Hitable{n.child[0]}, t0, t1, out)) return false;
        Vec3 t(n.f[0], n.f[1], n.f[2]);
        out.max = out.max + t;
        out.min = out.min + t;
        return true;
    }
    case PT_H_ROTATE_Y:
        out = Aabb{Vec3(n.f[2], n.f[3], n.f[4]), Vec3(n.f[5], n.f[6], n.f[7])};
        return true;
    default:
        return false;
    }
}

Hitable Scene::bvh_from_list(std::vector<Hitable> list, double time0, double time1) {
    if (list.empty()) {
        fail("bvh_from_list: empty list");
        return Hitable{-1};
    }
    // bvh.rs:15: random split axis
    int axis = (int)std::floor(rng_.rand() * 3.0);
    // bvh.rs:19-41: order by bounding_box(0,0).min()[axis]; the comparator only ever says Less or Greater,
    // so equal keys have no defined order in the reference -- a stable sort fixes one.
    std::vector<std::pair<double, Hitable>> keyed;
    keyed.reserve(list.size());
    for (auto &h : list) {
        Aabb b;
        if (!bounding_box(h, 0.0, 0.0, b)) {
            fail("bvh_from_list: child without bounding box (a HitableList cannot be a BVH leaf)");
            return Hitable{-1};
        }
        keyed.push_back({b.min[axis], h});
    }
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const std::pair<double, Hitable> &a, const std::pair<double, Hitable> &b) {
                         return a.first - b.first < 0.0;
                     });
    for (size_t i = 0; i < list.size(); ++i) list[i] = keyed[i].second;
    keyed.clear();
    keyed.shrink_to_fit();
    Hitable left, right;
    size_t n = list.size();
    if (n == 1) { // bvh.rs:47-49: the single element becomes both children
        left = right = list[0];
    } else if (n == 2) {
        left = list[0];
        right = list[1];
    } else {
        size_t half = n / 2;
        std::vector<Hitable> first(list.begin(), list.begin() + half), second(list.begin() + half, list.end());
        left = bvh_from_list(first, time0, time1);
        right = bvh_from_list(second, time0, time1);
        if (left.id < 0 || right.id < 0) return Hitable{-1};
    }
    Aabb bl, br;
    bounding_box(left, time0, time1, bl);
    bounding_box(right, time0, time1, br);
    Aabb u = aabb_union(bl, br);
    pt_hitable h = blank(PT_H_BVH);
    h.child[0] = left.id;
    h.child[1] = right.id;
    h.f[0] = u.min.x;
    h.f[1] = u.min.y;
    h.f[2] = u.min.z;
    h.f[3] = u.max.x;
    h.f[4] = u.max.y;
    h.f[5] = u.max.z;
    return Hitable{push(h)};
}

void Scene::set_lights(const std::vector<Hitable> &shapes) {
    lights_.clear();
    for (auto &s : shapes) lights_.push_back(s.id);
}
void Scene::set_default_lights() {
    Material dummy = dummy_material();
    Hitable light_shape = rect(213.0, 343.0, 227.0, 332.0, 554.0, Axis::Y, dummy);
    Hitable glass_sphere = sphere(Vec3(190.0, 90.0, 190.0), 90.0, dummy);
    set_lights({light_shape, glass_sphere});
}

pt_scene_desc Scene::desc() const {
    pt_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.struct_size = sizeof(pt_scene_desc);
    d.root = root_;
    d.hitables = hitables_.data();
    d.n_hitables = (int32_t)hitables_.size();
    d.child_index = child_index_.data();
    d.n_child_index = (int32_t)child_index_.size();
    d.materials = materials_.data();
    d.n_materials = (int32_t)materials_.size();
    d.textures = textures_.data();
    d.n_textures = (int32_t)textures_.size();
    d.texels = texels_.empty() ? nullptr : texels_.data();
    d.n_texel_bytes = texels_.size();
    d.perlin = have_perlin_ ? &perlin_ : nullptr;
    d.lights = lights_.data();
    d.n_lights = (int32_t)lights_.size();
    return d;
}

// --------------------------------------------------------- MaterialBuilder
MaterialBuilder::MaterialBuilder(Scene &scene) : scene_(scene) {
    texture_ = scene_.constant_texture(Vec3::from_float(0.0)); // material.rs:83
    albedo_ = Vec3::from_float(0.0);
}
MaterialBuilder &MaterialBuilder::with_texture(Texture t) {
    texture_ = t;
    return *this;
}
MaterialBuilder &MaterialBuilder::set_emissive(double e) {
    emissive_ = e;
    return *this;
}
MaterialBuilder &MaterialBuilder::set_albedo(Vec3 a) {
    albedo_ = a;
    return *this;
}
MaterialBuilder &MaterialBuilder::set_fuzz(double f) {
    fuzz_ = f;
    return *this;
}
MaterialBuilder &MaterialBuilder::set_refraction_index(double ri) {
    refraction_index_ = ri;
    return *this;
}
Material MaterialBuilder::lambertian() { return scene_.lambertian(texture_, emissive_); }
Material MaterialBuilder::diffuse_light() { return scene_.diffuse_light(texture_); }
Material MaterialBuilder::metal() { return scene_.metal(albedo_, fuzz_); }
Material MaterialBuilder::dielectric() { return scene_.dielectric(refraction_index_); }

// ------------------------------------------------------------ SceneBuilder
SceneBuilder &SceneBuilder::add_hitable(Hitable h) {
    list_.push_back(h);
    return *this;
}
SceneBuilder &SceneBuilder::flip_normals() {
    if (!list_.empty()) {
        Hitable h = list_.back();
        list_.pop_back();
        list_.push_back(scene_.flip_normals(h));
    }
    return *this;
}
SceneBuilder &SceneBuilder::translate(Vec3 translation) {
    if (!list_.empty()) {
        Hitable h = list_.back();
        list_.pop_back();
        list_.push_back(scene_.translate(h, translation));
    }
    return *this;
}
SceneBuilder &SceneBuilder::rotate_y(double angle) {
    if (!list_.empty()) {
        Hitable h = list_.back();
        list_.pop_back();
        list_.push_back(scene_.rotate_y(h, angle));
    }
    return *this;
}
Hitable SceneBuilder::as_bvh() { return scene_.bvh_from_list(list_, 0.0, 1.0); }
Hitable SceneBuilder::as_bvh_node() { return scene_.bvh_from_list(list_, 0.0, 1.0); }
Hitable SceneBuilder::as_hitable_list() { return scene_.hitable_list(list_); }
Hitable SceneBuilder::as_hitable() {
    if (list_.empty()) return Hitable{-1};
    Hitable h = list_.back();
    list_.pop_back();
    return h;
}

// ------------------------------------------------------------------ Camera
Camera::Camera(Vec3 origin, Vec3 look_at, Vec3 vup, double vfov, double aspect, double aperture,
               double focus_dist, double time0, double time1) {
    double theta = vfov * kPi / 180.0;
    half_height_ = std::tan(theta / 2.0);
    half_width_ = aspect * half_height_;
    origin_ = origin;
    look_at_ = look_at;
    world_up_ = vup;
    focus_dist_ = focus_dist;
    time0_ = time0;
    time1_ = time1;
    lens_radius_ = aperture / 2.0;
    vfov_ = vfov;
    aspect_ = aspect;
    update();
}
void Camera::update() { // camera.rs:89-96
    w_ = unit_vector(origin_ - look_at_);
    u_ = unit_vector(cross(world_up_, w_));
    v_ = cross(w_, u_);
    lower_left_corner_ =
        origin_ - (u_ * half_width_ * focus_dist_) - (v_ * half_height_ * focus_dist_) - (w_ * focus_dist_);
    horizontal_ = u_ * 2.0 * half_width_ * focus_dist_;
    vertical_ = v_ * 2.0 * half_height_ * focus_dist_;
}
void Camera::set_origin(Vec3 origin, bool update_look_at) { // camera.rs:77-83
    if (update_look_at) look_at_ = look_at_ + (origin - origin_);
    origin_ = origin;
    lower_left_corner_ =
        origin_ - (u_ * half_width_ * focus_dist_) - (v_ * half_height_ * focus_dist_) - (w_ * focus_dist_);
}
void Camera::set_look_at(Vec3 look_at, bool maintain_distance) { // camera.rs:98-107
    if (maintain_distance) {
        double look_at_dist = length(origin_ - look_at_);
        double new_look_at_dist = length(origin_ - look_at);
        look_at = look_at * (look_at_dist / new_look_at_dist);
    }
    look_at_ = look_at;
}
bool Camera::update_from_input(const pt_user_input &in, double frame_time) { // camera.rs:109-215
    const double CAM_SPEED = 40.0;
    const double MOUSE_LOOK_SPEED = 1.0;
    bool camera_moved = false;
    auto pressed = [&](uint32_t k) { return (in.keys_pressed & k) != 0; };
    auto held = [&](uint32_t k) { return (in.keys_held & k) != 0; };
    auto move = [&](Vec3 axis, double sign) { // camera.rs:121-167
        Vec3 diff = axis * sign * CAM_SPEED * frame_time;
        set_origin(get_origin() + diff, true);
        camera_moved = true;
    };
    if (pressed(PT_KEY_W)) move(get_forward(), 1.0);
    if (pressed(PT_KEY_S)) move(get_forward(), -1.0);
    if (pressed(PT_KEY_D)) move(get_right(), 1.0);
    if (pressed(PT_KEY_A)) move(get_right(), -1.0);
    if (pressed(PT_KEY_E)) move(get_up(), 1.0);
    if (pressed(PT_KEY_Q)) move(get_up(), -1.0);
    auto look = [&](Vec3 axis, double sign) { // camera.rs:169-192
        set_look_at(get_look_at() + axis * sign * CAM_SPEED * frame_time, true);
        camera_moved = true;
    };
    if (held(PT_KEY_RIGHT)) look(get_right(), 1.0);
    if (held(PT_KEY_LEFT)) look(get_right(), -1.0);
    if (held(PT_KEY_UP)) look(get_up(), 1.0);
    if (held(PT_KEY_DOWN)) look(get_up(), -1.0);
    if (in.mouse_dx != 0.0f || in.mouse_dy != 0.0f) { // camera.rs:193-211
        Vec3 cam_look_at = get_look_at();
        if (in.mouse_dx != 0.0f) cam_look_at = cam_look_at + get_right() * MOUSE_LOOK_SPEED * frame_time * (double)in.mouse_dx;
        if (in.mouse_dy != 0.0f) cam_look_at = cam_look_at + get_up() * MOUSE_LOOK_SPEED * frame_time * (double)in.mouse_dy;
        set_look_at(cam_look_at, true);
        camera_moved = true;
    }
    return camera_moved;
}
pt_camera Camera::params() const {
    pt_camera c;
    c.origin[0] = origin_.x; c.origin[1] = origin_.y; c.origin[2] = origin_.z;
    c.look_at[0] = look_at_.x; c.look_at[1] = look_at_.y; c.look_at[2] = look_at_.z;
    c.vup[0] = world_up_.x; c.vup[1] = world_up_.y; c.vup[2] = world_up_.z;
    c.vfov = vfov_;
    c.aspect = aspect_;
    c.aperture = lens_radius_ * 2.0;
    c.focus_dist = focus_dist_;
    c.time0 = time0_;
    c.time1 = time1_;
    return c;
}

// ------------------------------------------------------------------ realtime session
RealtimeSession::RealtimeSession(pt_ctx *ctx, const pt_camera &c, uint32_t width, uint32_t height, uint32_t max_depth)
    : ctx_(ctx),
      cam_(Vec3{c.origin[0], c.origin[1], c.origin[2]}, Vec3{c.look_at[0], c.look_at[1], c.look_at[2]},
           Vec3{c.vup[0], c.vup[1], c.vup[2]}, c.vfov, c.aspect, c.aperture, c.focus_dist, c.time0, c.time1),
      width_(width), height_(height), max_depth_(max_depth) {}

int RealtimeSession::frame(const pt_user_input &in, double frame_time, pt_frame_info *info) {
    bool clear_scene = false;
    time0_ = time1_; // lib.rs:412-417
    time1_ += frame_time;
    numframes_ += 1.0; // lib.rs:419
    auto pressed = [&](uint32_t k) { return (in.keys_pressed & k) != 0; };
    if (pressed(PT_KEY_T)) exposure_ += 0.1; // lib.rs:425-429
    else if (pressed(PT_KEY_R)) exposure_ -= 0.1;
    if (pressed(PT_KEY_O)) { // lib.rs:431-435
        clear_scene = true;
        sky_brightness_ = std::max(sky_brightness_ - 0.05, 0.0);
    }
    if (pressed(PT_KEY_P)) { // lib.rs:437-441
        clear_scene = true;
        sky_brightness_ += 0.05;
    }
    if (pressed(PT_KEY_B)) { // lib.rs:443-447
        disable_emissive_ = !disable_emissive_;
        clear_scene = true;
    }
    const bool camera_moved = cam_.update_from_input(in, frame_time); // lib.rs:456-460
    int rc = PT_OK;
    const bool cleared = camera_moved || clear_scene;
    if (cleared) { // lib.rs:462-468
        cam_.update();
        rc = pt_accum_reset(ctx_);
        numframes_ = 1.0;
    }
    const pt_camera params = cam_.params();
    if (rc == PT_OK && (cleared || frame_counter_ == 0)) rc = pt_camera_set(ctx_, &params);
    if (rc == PT_OK) { // lib.rs:475-478: dispatch the tile jobs and wait; one job covers the image here
        pt_render_args a;
        std::memset(&a, 0, sizeof(a));
        a.struct_size = sizeof(a);
        a.width = width_;
        a.height = height_;
        a.x1 = width_;
        a.y1 = height_;
        a.spp = 1;
        a.max_depth = max_depth_;
        a.mode = PT_MODE_REALTIME;
        a.frame = frame_counter_;
        rc = pt_render(ctx_, &a);
    }
    if (rc == PT_OK) frame_counter_ += 1;
    if (info) {
        info->camera_moved = camera_moved ? 1 : 0;
        info->cleared = cleared ? 1 : 0;
        info->exposure = exposure_;
        info->numframes = numframes_;
        info->time0 = time0_;
        info->time1 = time1_;
        info->sky_brightness = sky_brightness_;
        info->disable_emissive = disable_emissive_ ? 1 : 0;
        info->frame = frame_counter_;
        info->camera = params;
    }
    return rc;
}

// ------------------------------------------------------------------ Config
Config Config::from_cmdline(const std::vector<std::string> &args) { // lib.rs:125-146
    Config c;
    for (auto &a : args) {
        if (a == "-offline") {
            c.realtime = false;
            c.max_depth = 50;
            c.spp = 100;
        }
    }
    for (auto &a : args) {
        if (a.rfind("-spp=", 0) == 0) {
            long v = std::strtol(a.c_str() + 5, nullptr, 10);
            if (v > 0) c.spp = (uint32_t)v;
        }
    }
    return c;
}

// --------------------------------------------------------------- tile jobs
static uint32_t round_down_to_closest_factor(uint32_t factor_to_round, uint32_t factor_of) { // math/mod.rs:28-31
    if (factor_to_round == 0) return factor_of;
    double factor = (double)factor_of / (double)factor_to_round;
    uint32_t f = (uint32_t)std::floor(factor);
    return f == 0 ? factor_of : factor_of / f;
}
std::vector<std::array<uint32_t, 4>> make_tile_grid(uint32_t nx, uint32_t ny) {
    uint32_t tw = round_down_to_closest_factor(nx / 9, nx); // lib.rs:345-348
    uint32_t th = round_down_to_closest_factor(ny / 9, ny);
    if (tw == 0) tw = 1;
    if (th == 0) th = 1;
    std::vector<std::array<uint32_t, 4>> tiles;
    for (uint32_t y = 0; y < ny; y += th)
        for (uint32_t x = 0; x < nx; x += tw)
            tiles.push_back({x, y, std::min(x + tw, nx), std::min(y + th, ny)});
    return tiles;
}

TraceSceneBatchJob::TraceSceneBatchJob(uint32_t num_samples, std::array<uint32_t, 2> start_xy,
                                       std::array<uint32_t, 2> end_xy, std::array<uint32_t, 2> image_size,
                                       SceneState *state, SceneOutput *output, bool realtime)
    : num_samples_(num_samples), start_xy_(start_xy), end_xy_(end_xy), image_size_(image_size), state_(state),
      output_(output), realtime_(realtime) {}

pt_render_args TraceSceneBatchJob::render_args() const {
    pt_render_args a;
    std::memset(&a, 0, sizeof(a));
    a.struct_size = sizeof(a);
    a.width = image_size_[0];
    a.height = image_size_[1];
    a.x0 = start_xy_[0];
    a.y0 = start_xy_[1];
    a.x1 = end_xy_[0];
    a.y1 = end_xy_[1];
    a.spp = num_samples_;
    a.max_depth = (uint32_t)state_->config.max_depth;
    a.mode = realtime_ ? PT_MODE_REALTIME : PT_MODE_OFFLINE;
    a.frame = state_->frame;
    return a;
}
int TraceSceneBatchJob::run() {
    const pt_render_args a = render_args();
    int rc = pt_render(state_->ctx, &a);
    if (output_) output_->notify_task_completion();
    return rc;
}
int dispatch_jobs(std::vector<TraceSceneBatchJob> &jobs) {
    if (jobs.empty()) return PT_OK;
    std::vector<pt_render_args> args;
    args.reserve(jobs.size());
    for (const TraceSceneBatchJob &j : jobs) {
        if (j.context() != jobs[0].context()) return PT_ERR_INVALID_ARG; // one batch, one context
        args.push_back(j.render_args());
    }
    const int rc = pt_render_jobs(jobs[0].context(), args.data(), args.size());
    for (const TraceSceneBatchJob &j : jobs)
        if (j.output()) j.output()->notify_task_completion();
    return rc;
}
void TraceSceneBatchJob::clear_buffer() { pt_accum_reset(state_->ctx); }

} // namespace ptgpu
