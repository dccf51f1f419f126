/*
 * pt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A CPU restatement, in f64, of the render path of SimpsonGSD/path-tracer:
 * it keeps the reference's recursion, comparison forms and quirks, and
 * interprets the scene-graph wire format of include/ptgpu.h directly (no
 * flattening, no device).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library never does.
 *
 * PARITY UNPINNED: the reference holds no golden vectors or known-answer tests
 * for this path (its only test pins Vec3 arithmetic, src/math/vec3.rs:495-577,
 * which is restated in tests/), and neither rustc nor cargo exists in this
 * image, so the reference itself cannot be run.  The oracle is validated by
 * closed-form geometry and energy-conservation tests instead.
 *
 * The one deliberate departure: the reference draws from an OS-seeded
 * thread_rng (src/math/random.rs:3-6), which is irreproducible; here every
 * path owns a counter-based stream (two PCG hash rounds over a 64-bit key made
 * from seed, frame, pixel and sample) and draws from it in the reference's call order.  The free-flight
 * draw inside ConstantMedium::hit (src/volume.rs:42) is keyed by
 * (path, depth, medium node) so that it does not depend on traversal order.
 *
 * Build with -ffp-contract=off: Rust never fuses a*b+c.
 */
#include "../include/ptgpu.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

const double PI = 3.14159265358979323846264338327950288;      /* std::f64::consts::PI */
const double FRAC_PI_2 = 1.57079632679489661923132169163975144;
const double FRAC_1_PI = 0.318309886183790671537767526745028724;
const double F64_MAX = DBL_MAX;

/* ---------------------------------------------------------------- RNG -- */

const uint64_t GOLDEN = 0x9E3779B97F4A7C15ULL;

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline double to_unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

/* PCG output hash (32 -> 32 bits) */
inline uint32_t pcg32(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
/* n-th draw (n >= 1) of the path stream with 64-bit key: two PCG rounds keyed by the two key halves */
inline uint32_t path_draw_u32(uint64_t key, uint32_t n) {
    uint32_t x = pcg32((uint32_t)key + n * 0x9E3779B9u);
    return pcg32(x ^ (uint32_t)(key >> 32));
}
struct Rng {
    uint64_t key;
    uint64_t n;
    /* stands in for random::rand (src/math/random.rs:3-6): f64 in [0,1); 32 random bits per draw */
    double next() {
        n += 1;
        return (double)path_draw_u32(key, (uint32_t)n) * (1.0 / 4294967296.0);
    }
};

inline uint64_t path_key(uint64_t seed, uint64_t frame, uint64_t pixel, uint64_t sample) {
    uint64_t k = mix64(seed ^ (0xA0761D6478BD642FULL * (frame + 1)));
    k = mix64(k ^ pixel);
    k = mix64(k + GOLDEN * (sample + 1));
    return k;
}
inline double medium_draw(uint64_t pkey, uint32_t depth, int32_t medium_node) {
    uint64_t salt = mix64(0x4D454449554D0000ULL + ((uint64_t)depth << 32) + (uint64_t)(uint32_t)medium_node);
    return to_unit(mix64(pkey ^ salt));
}
inline double pixel_skip_draw(uint64_t seed, uint64_t frame, uint64_t pixel) {
    uint64_t k = mix64(seed ^ 0x534B495050495845ULL ^ (0xA0761D6478BD642FULL * (frame + 1)));
    return to_unit(mix64(k ^ pixel));
}
inline uint64_t intersect_key(uint64_t medium_seed, uint64_t i) { return mix64(medium_seed ^ mix64(i + GOLDEN)); }
inline uint64_t scene_stream_key(uint64_t seed) { return mix64(seed ^ 0x5343454E45ULL); }

/* --------------------------------------------------------------- Vec3 -- */
/* src/math/vec3.rs:4-155 */
struct V3 {
    double x, y, z;
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator+(V3 a, double s) { return V3{a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(V3 a, double s) { return V3{a.x - s, a.y - s, a.z - s}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x};
}
inline double squared_length(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double length(V3 a) { return std::sqrt(squared_length(a)); }
inline V3 unit_vector(V3 a) { return a / length(a); } /* vec3.rs:31-33 */
inline double idx(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* src/math/ray.rs:3-33 */
struct Ray {
    V3 o, d;
    double time;
};
inline V3 point_at(const Ray &r, double t) { return r.o + (r.d * t); }

/* src/math/aabb.rs:6-72 */
struct AABB {
    V3 mn, mx;
};
inline double ffmin(double a, double b) { return a < b ? a : b; }
inline double ffmax(double a, double b) { return a > b ? a : b; }
inline AABB aabb_union(const AABB &a, const AABB &b) {
    return AABB{v3(ffmin(a.mn.x, b.mn.x), ffmin(a.mn.y, b.mn.y), ffmin(a.mn.z, b.mn.z)),
                v3(ffmax(a.mx.x, b.mx.x), ffmax(a.mx.y, b.mx.y), ffmax(a.mx.z, b.mx.z))};
}
/* aabb.rs:27-47 */
inline bool aabb_hit(const AABB &b, const Ray &r, double tmin, double tmax) {
    for (int i = 0; i < 3; ++i) {
        double inv_d = 1.0 / idx(r.d, i);
        double t0 = (idx(b.mn, i) - idx(r.o, i)) * inv_d;
        double t1 = (idx(b.mx, i) - idx(r.o, i)) * inv_d;
        if (inv_d < 0.0) std::swap(t0, t1);
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}

/* src/hitable.rs:5-12, plus the graph node that produced the record */
struct HitRec {
    double t, u, v;
    V3 p, n;
    int mat;
    int prim;
};

struct Counters {
    uint64_t segments = 0, node_visits = 0, prim_tests = 0, paths = 0;
};

struct Ctx {
    const pt_scene_desc *s;
    Rng rng;
    uint64_t pkey;  /* key of the medium draws */
    uint32_t depth; /* bounce depth of the segment being intersected */
    Counters *c;
    bool counting;  /* false while evaluating pdf_value re-intersections */
    bool book = false; /* PT_RENDER_BOOK_SEMANTICS: Isotropic with the uniform-sphere pdf */
};

/* ----------------------------------------------------------- geometry -- */

/* sphere.rs:8-14 */
inline void get_sphere_uv(V3 p, double &u, double &v) {
    double phi = std::atan2(p.z, p.x);
    double theta = std::asin(p.y);
    u = 1.0 - (phi + PI) / (PI * 2.0);
    v = (theta + FRAC_PI_2) / PI;
}

bool hit(Ctx &cx, int h, const Ray &r, double t_min, double t_max, HitRec &rec);

/* sphere.rs:44-81 (normalise_second_uv = false) and 137-175 (moving: both roots un-normalised) */
inline bool sphere_hit_common(V3 center, double radius, int mat, int prim, bool first_uv_normalised,
                              const Ray &ray, double t_min, double t_max, HitRec &rec) {
    V3 oc = ray.o - center;
    double a = dot(ray.d, ray.d);
    double b = dot(oc, ray.d);
    double c = dot(oc, oc) - radius * radius;
    double discriminant = b * b - a * c;
    if (discriminant > 0.0) {
        double temp = (-b - std::sqrt(b * b - a * c)) / a;
        if (temp < t_max && temp > t_min) {
            V3 point = point_at(ray, temp);
            double u, v;
            if (first_uv_normalised)
                get_sphere_uv((center - point) / radius, u, v);
            else
                get_sphere_uv(center - point, u, v);
            rec = HitRec{temp, u, v, point, (point - center) / radius, mat, prim};
            return true;
        }
        temp = (-b + std::sqrt(b * b - a * c)) / a;
        if (temp < t_max && temp > t_min) {
            V3 point = point_at(ray, temp);
            double u, v;
            get_sphere_uv(center - point, u, v);
            rec = HitRec{temp, u, v, point, (point - center) / radius, mat, prim};
            return true;
        }
    }
    return false;
}

inline V3 moving_center(const pt_hitable &n, double time) { /* sphere.rs:130-132 */
    V3 c0 = v3(n.f[0], n.f[1], n.f[2]);
    V3 range = v3(n.f[4], n.f[5], n.f[6]);
    return c0 + ((time - n.f[7]) / n.f[8]) * range;
}

/* rect.rs:63-114 */
inline bool rect_hit(const pt_hitable &n, int prim, const Ray &ray, double t_min, double t_max, HitRec &rec) {
    double amin = n.f[0], amax = n.f[1], bmin = n.f[2], bmax = n.f[3], c = n.f[4];
    double t;
    switch (n.axis) {
    case 0: t = (c - ray.o.x) / ray.d.x; break;
    case 1: t = (c - ray.o.y) / ray.d.y; break;
    default: t = (c - ray.o.z) / ray.d.z; break;
    }
    if (t < t_min || t > t_max) return false;
    double a, b;
    switch (n.axis) {
    case 0: a = ray.o.y + t * ray.d.y; b = ray.o.z + t * ray.d.z; break;
    case 1: a = ray.o.x + t * ray.d.x; b = ray.o.z + t * ray.d.z; break;
    default: a = ray.o.x + t * ray.d.x; b = ray.o.y + t * ray.d.y; break;
    }
    if (a < amin || a > amax || b < bmin || b > bmax) return false;
    V3 normal = n.axis == 0 ? v3(1, 0, 0) : (n.axis == 1 ? v3(0, 1, 0) : v3(0, 0, 1));
    double a_size = amax - amin, b_size = bmax - bmin; /* rect.rs:46-47 */
    rec = HitRec{t, (a - amin) / a_size, (b - bmin) / b_size, point_at(ray, t), normal, n.material, prim};
    return true;
}

/* hitable.rs:51-62 */
inline bool list_hit(Ctx &cx, const pt_hitable &n, const Ray &r, double t_min, double t_max, HitRec &rec) {
    double closest = t_max;
    bool any = false;
    for (int k = 0; k < n.list_count; ++k) {
        HitRec tmp;
        if (hit(cx, cx.s->child_index[n.list_start + k], r, t_min, closest, tmp)) {
            closest = tmp.t;
            rec = tmp;
            any = true;
        }
    }
    return any;
}

bool hit(Ctx &cx, int h, const Ray &r, double t_min, double t_max, HitRec &rec) {
    const pt_hitable &n = cx.s->hitables[h];
    switch (n.kind) {
    case PT_H_SPHERE:
        if (cx.counting) cx.c->prim_tests++;
        return sphere_hit_common(v3(n.f[0], n.f[1], n.f[2]), n.f[3], n.material, h, true, r, t_min, t_max, rec);
    case PT_H_MOVING_SPHERE:
        if (cx.counting) cx.c->prim_tests++;
        return sphere_hit_common(moving_center(n, r.time), n.f[3], n.material, h, false, r, t_min, t_max, rec);
    case PT_H_RECT:
        if (cx.counting) cx.c->prim_tests++;
        return rect_hit(n, h, r, t_min, t_max, rec);
    case PT_H_BOX: /* axis_aligned_box.rs:33-35 */
    case PT_H_LIST:
        return list_hit(cx, n, r, t_min, t_max, rec);
    case PT_H_FLIP: /* hitable.rs:93-100 */
        if (hit(cx, n.child[0], r, t_min, t_max, rec)) {
            rec.n = -rec.n;
            return true;
        }
        return false;
    case PT_H_TRANSLATE: { /* hitable.rs:122-131 */
        V3 tr = v3(n.f[0], n.f[1], n.f[2]);
        Ray moved{r.o - tr, r.d, r.time};
        if (hit(cx, n.child[0], moved, t_min, t_max, rec)) {
            rec.p = rec.p + tr;
            return true;
        }
        return false;
    }
    case PT_H_ROTATE_Y: { /* hitable.rs:182-215 */
        double s = n.f[0], c = n.f[1];
        V3 o = r.o, d = r.d;
        V3 oo = v3(c * o.x - s * o.z, o.y, s * o.x + c * o.z);
        V3 dd = v3(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
        Ray rr{oo, dd, r.time};
        if (hit(cx, n.child[0], rr, t_min, t_max, rec)) {
            V3 p = rec.p, nn = rec.n;
            rec.p = v3(c * p.x + s * p.z, p.y, -s * p.x + c * p.z);
            rec.n = v3(c * nn.x + s * nn.z, nn.y, -s * nn.x + c * nn.z);
            return true;
        }
        return false;
    }
    case PT_H_MEDIUM: { /* volume.rs:24-65 */
        HitRec rec1, rec2;
        if (hit(cx, n.child[0], r, -F64_MAX, F64_MAX, rec1)) {
            if (hit(cx, n.child[0], r, rec1.t + 0.0001, F64_MAX, rec2)) {
                rec1.t = std::fmax(rec1.t, t_min); /* f64::max */
                rec2.t = std::fmin(rec2.t, t_max);
                if (rec1.t >= rec2.t) return false;
                double ray_length = length(r.d);
                double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
                double hit_distance = -(1.0 / n.f[0]) * std::log(medium_draw(cx.pkey, cx.depth, h));
                if (hit_distance < distance_inside_boundary) {
                    double time = rec1.t + hit_distance / ray_length;
                    rec = HitRec{time, 0.0, 0.0, point_at(r, time), v3(1.0, 0.0, 0.0), n.material, h};
                    return true;
                }
            }
        }
        return false;
    }
    case PT_H_BVH: { /* bvh.rs:74-99 */
        if (cx.counting) cx.c->node_visits++;
        AABB box{v3(n.f[0], n.f[1], n.f[2]), v3(n.f[3], n.f[4], n.f[5])};
        if (!aabb_hit(box, r, t_min, t_max)) return false;
        HitRec left;
        if (hit(cx, n.child[0], r, t_min, t_max, left)) {
            HitRec right;
            if (hit(cx, n.child[1], r, t_min, left.t, right))
                rec = right;
            else
                rec = left;
            return true;
        }
        return hit(cx, n.child[1], r, t_min, t_max, rec);
    }
    default:
        return false;
    }
}

/* bounding_box of every Hitable implementor; returns false for a LIST (unreachable!, hitable.rs:63-65) */
bool bounding_box(const pt_scene_desc *s, int h, double t0, double t1, AABB &out) {
    const pt_hitable &n = s->hitables[h];
    switch (n.kind) {
    case PT_H_SPHERE: { /* sphere.rs:83-85 */
        V3 c = v3(n.f[0], n.f[1], n.f[2]);
        out = AABB{c - v3(n.f[3], n.f[3], n.f[3]), c + v3(n.f[3], n.f[3], n.f[3])};
        return true;
    }
    case PT_H_MOVING_SPHERE: { /* sphere.rs:177-183 */
        V3 c0 = moving_center(n, t0), c1 = moving_center(n, t1);
        V3 rr = v3(n.f[3], n.f[3], n.f[3]);
        out = aabb_union(AABB{c0 - rr, c0 + rr}, AABB{c1 - rr, c1 + rr});
        return true;
    }
    case PT_H_RECT: { /* rect.rs:116-122 */
        double amin = n.f[0], amax = n.f[1], bmin = n.f[2], bmax = n.f[3], c = n.f[4];
        if (n.axis == 0)
            out = AABB{v3(c - 0.0001, amin, bmin), v3(c + 0.0001, amax, bmax)};
        else if (n.axis == 1)
            out = AABB{v3(amin, c - 0.0001, bmin), v3(amax, c + 0.0001, bmax)};
        else
            out = AABB{v3(amin, bmin, c - 0.0001), v3(amax, bmax, c + 0.0001)};
        return true;
    }
    case PT_H_BOX: /* axis_aligned_box.rs:37-39 */
        out = AABB{v3(n.f[0], n.f[1], n.f[2]), v3(n.f[3], n.f[4], n.f[5])};
        return true;
    case PT_H_LIST:
        return false;
    case PT_H_FLIP: /* hitable.rs:102-104 */
    case PT_H_MEDIUM: /* volume.rs:66-68 */
        return bounding_box(s, n.child[0], t0, t1, out);
    case PT_H_TRANSLATE: { /* hitable.rs:133-137, aabb.rs:58-61 */
        if (!bounding_box(s, n.child[0], t0, t1, out)) return false;
        V3 tr = v3(n.f[0], n.f[1], n.f[2]);
        out.mx = out.mx + tr;
        out.mn = out.mn + tr;
        return true;
    }
    case PT_H_ROTATE_Y: /* hitable.rs:199-201: the box cached by RotateY::new */
        out = AABB{v3(n.f[2], n.f[3], n.f[4]), v3(n.f[5], n.f[6], n.f[7])};
        return true;
    case PT_H_BVH: /* bvh.rs:101-103 */
        out = AABB{v3(n.f[0], n.f[1], n.f[2]), v3(n.f[3], n.f[4], n.f[5])};
        return true;
    default:
        return false;
    }
}

/* RotateY::new's corner sweep (hitable.rs:148-172) */
AABB rotate_y_box(const AABB &bb, double sin_theta, double cos_theta) {
    V3 mn = v3(F64_MAX, F64_MAX, F64_MAX), mx = v3(-F64_MAX, -F64_MAX, -F64_MAX);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                double fi = i, fj = j, fk = k;
                double x = fi * bb.mx.x + (1.0 - fi) * bb.mn.x;
                double y = fj * bb.mx.y + (1.0 - fj) * bb.mn.y;
                double z = fk * bb.mx.z + (1.0 - fk) * bb.mn.z;
                double nx = cos_theta * x + sin_theta * z;
                double nz = -sin_theta * x + cos_theta * z;
                V3 c = v3(nx, y, nz);
                mn = v3(std::fmin(c.x, mn.x), std::fmin(c.y, mn.y), std::fmin(c.z, mn.z));
                mx = v3(std::fmax(c.x, mx.x), std::fmax(c.y, mx.y), std::fmax(c.z, mx.z));
            }
    return AABB{mn, mx};
}

/* ------------------------------------------------------------ samplers -- */

/* onb.rs:18-37 */
struct ONB {
    V3 u, v, w;
};
inline ONB onb_from_w(V3 n) {
    V3 w = unit_vector(n);
    V3 a = std::fabs(w.x) > 0.9 ? v3(0.0, 1.0, 0.0) : v3(1.0, 0.0, 0.0);
    V3 v = unit_vector(cross(w, a));
    V3 u = cross(w, v);
    return ONB{u, v, w};
}
inline V3 onb_local(const ONB &b, V3 a) { return b.u * a.x + b.v * a.y + b.w * a.z; }

/* material.rs:10-19 */
inline V3 random_cosine_direction(Rng &g) {
    double r1 = g.next();
    double r2 = g.next();
    double z = std::sqrt(1.0 - r2);
    double phi = 2.0 * PI * r1;
    double r2_sqrt = std::sqrt(r2);
    double x = std::cos(phi) * r2_sqrt;
    double y = std::sin(phi) * r2_sqrt;
    return v3(x, y, z);
}
/* material.rs:21-31 */
inline V3 random_in_unit_sphere(Rng &g) {
    V3 p;
    for (;;) {
        double a = g.next();
        double b = g.next();
        double c = g.next();
        p = 2.0 * v3(a, b, c) - v3(1.0, 1.0, 1.0);
        if (squared_length(p) < 1.0) break;
    }
    return p;
}
/* camera.rs:221-228 */
inline V3 random_in_unit_disk(Rng &g) {
    double a = g.next();
    double b = g.next();
    V3 nv = v3(a, b, 0.0) * 2.0 - v3(1.0, 1.0, 0.0);
    while (dot(nv, nv) >= 1.0) {
        a = g.next();
        b = g.next();
        nv = v3(a, b, 0.0) * 2.0 - v3(1.0, 1.0, 0.0);
    }
    return nv;
}
/* sphere.rs:16-25 */
inline V3 random_to_sphere(Rng &g, double radius, double distance_squared) {
    double r1 = g.next();
    double r2 = g.next();
    double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / distance_squared) - 1.0);
    double phi = 2.0 * PI * r1;
    double c = std::sqrt(1.0 - z * z);
    double x = std::cos(phi) * c;
    double y = std::sin(phi) * c;
    return v3(x, y, z);
}

/* Hitable::pdf_value (hitable.rs:30-32 default; sphere.rs:87-95; rect.rs:124-133; hitable.rs:66-73) */
double pdf_value(Ctx &cx, int h, V3 origin, V3 direction) {
    const pt_hitable &n = cx.s->hitables[h];
    bool saved = cx.counting;
    cx.counting = false;
    double out = 0.0;
    switch (n.kind) {
    case PT_H_SPHERE: {
        HitRec rec;
        if (hit(cx, h, Ray{origin, direction, 0.0}, 0.001, F64_MAX, rec)) {
            V3 c = v3(n.f[0], n.f[1], n.f[2]);
            double cos_theta_max = std::sqrt(1.0 - n.f[3] * n.f[3] / squared_length(c - origin));
            double solid_angle = 2.0 * PI * (1.0 - cos_theta_max);
            out = 1.0 / solid_angle;
        }
        break;
    }
    case PT_H_RECT: {
        HitRec rec;
        if (hit(cx, h, Ray{origin, direction, 0.0}, 0.001, F64_MAX, rec)) {
            double area = (n.f[1] - n.f[0]) * (n.f[3] - n.f[2]); /* rect.rs:58 */
            double v_sq_length = squared_length(direction);
            double distance_sq = rec.t * rec.t * v_sq_length;
            double cosine = std::fabs(dot(direction, rec.n) / std::sqrt(v_sq_length));
            out = distance_sq / (cosine * area);
        }
        break;
    }
    case PT_H_LIST: {
        double weight = 1.0 / (double)n.list_count;
        double sum = 0.0;
        for (int k = 0; k < n.list_count; ++k)
            sum += weight * pdf_value(cx, cx.s->child_index[n.list_start + k], origin, direction);
        out = sum;
        break;
    }
    default:
        out = 0.0;
    }
    cx.counting = saved;
    return out;
}

/* Hitable::random (hitable.rs:33-35 default; sphere.rs:96-101; rect.rs:135-144; hitable.rs:74-77) */
V3 hitable_random(Ctx &cx, int h, V3 origin) {
    const pt_hitable &n = cx.s->hitables[h];
    switch (n.kind) {
    case PT_H_SPHERE: {
        V3 direction = v3(n.f[0], n.f[1], n.f[2]) - origin;
        double distance_sq = squared_length(direction);
        ONB uvw = onb_from_w(direction);
        return onb_local(uvw, random_to_sphere(cx.rng, n.f[3], distance_sq));
    }
    case PT_H_RECT: {
        double random_a = n.f[0] + cx.rng.next() * (n.f[1] - n.f[0]);
        double random_b = n.f[2] + cx.rng.next() * (n.f[3] - n.f[2]);
        V3 p = n.axis == 0 ? v3(n.f[4], random_a, random_b)
                           : (n.axis == 1 ? v3(random_a, n.f[4], random_b) : v3(random_a, random_b, n.f[4]));
        return p - origin;
    }
    case PT_H_LIST: {
        double x = (double)n.list_count * cx.rng.next();
        int index = (int)x;
        return hitable_random(cx, cx.s->child_index[n.list_start + index], origin);
    }
    default:
        return v3(1.0, 0.0, 0.0);
    }
}

/* The light-shape HitableList of trace.rs:123-128, held as desc->lights. */
double lights_pdf_value(Ctx &cx, V3 origin, V3 direction) {
    int n = cx.s->n_lights;
    double weight = 1.0 / (double)n;
    double sum = 0.0;
    for (int k = 0; k < n; ++k) sum += weight * pdf_value(cx, cx.s->lights[k], origin, direction);
    return sum;
}
V3 lights_random(Ctx &cx, V3 origin) {
    double x = (double)cx.s->n_lights * cx.rng.next();
    int index = (int)x;
    return hitable_random(cx, cx.s->lights[index], origin);
}

/* ------------------------------------------------------------ textures -- */

/* math/noise.rs:26-44 */
double perlin_interpolate(const V3 c[2][2][2], double u, double v, double w) {
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                double fi = i, fj = j, fk = k;
                V3 weight_v = v3(u - fi, v - fj, w - fk);
                accum += (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                         (fk * ww + (1.0 - fk) * (1.0 - ww)) * dot(c[i][j][k], weight_v);
            }
    return accum;
}
inline int32_t f64_to_i32_saturating(double x) { /* Rust `as i32` */
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
/* math/noise.rs:89-113 */
double perlin_noise(const pt_perlin *t, V3 p) {
    int32_t i = f64_to_i32_saturating(std::floor(p.x));
    int32_t j = f64_to_i32_saturating(std::floor(p.y));
    int32_t k = f64_to_i32_saturating(std::floor(p.z));
    double u = p.x - (double)i;
    double v = p.y - (double)j;
    double w = p.z - (double)k;
    V3 c[2][2][2];
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                int32_t ix = t->perm_x[(uint32_t)(i + di) & 255] ^ t->perm_y[(uint32_t)(j + dj) & 255] ^
                             t->perm_z[(uint32_t)(k + dk) & 255];
                c[di][dj][dk] = v3(t->ranvec[ix][0], t->ranvec[ix][1], t->ranvec[ix][2]);
            }
    return perlin_interpolate(c, u, v, w);
}
/* math/noise.rs:115-126 */
double perlin_turb(const pt_perlin *t, V3 p, int depth) {
    double accum = 0.0;
    V3 temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(t, temp_p);
        weight *= 0.5;
        temp_p = temp_p * 2.0;
    }
    return std::fabs(accum);
}
inline double clamp_ref(double a, double lo, double hi) { /* math/mod.rs:22-26 */
    double b = a > hi ? hi : a;
    b = b < lo ? lo : b;
    return b;
}
inline size_t f64_to_usize(double x) { /* Rust `as usize`: NaN -> 0, saturating */
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)x;
}

/* Texture::value (texture.rs:24-28, 47-56, 70-75, 101-114) */
V3 texture_value(const pt_scene_desc *s, int tex, double u, double v, V3 p) {
    const pt_texture &t = s->textures[tex];
    switch (t.kind) {
    case PT_T_CONSTANT:
        return v3(t.colour[0], t.colour[1], t.colour[2]);
    case PT_T_CHECKER: {
        double sines = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
        if (sines < 0.0) return texture_value(s, t.odd, u, v, p);
        return texture_value(s, t.even, u, v, p);
    }
    case PT_T_NOISE: {
        double noise = t.scale * p.z + 10.0 * perlin_turb(s->perlin, p, 7);
        return v3(1.0, 1.0, 1.0) * 0.5 * (1.0 + std::sin(noise));
    }
    case PT_T_IMAGE: {
        double wf = (double)t.width, hf = (double)t.height;
        double i = u * wf;
        double j = v * hf - 0.001;
        size_t ii = f64_to_usize(clamp_ref(i, 0.0, wf - 1.0));
        size_t jj = f64_to_usize(clamp_ref(j, 0.0, hf - 1.0));
        size_t off = 3 * ii + 3 * (size_t)t.width * jj;
        const uint8_t *d = s->texels + t.texel_offset;
        return v3(d[off] / 255.0, d[off + 1] / 255.0, d[off + 2] / 255.0);
    }
    default:
        return v3(0, 0, 0);
    }
}

/* ----------------------------------------------------------- materials -- */

inline V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n) * n); } /* material.rs:50-52 */
inline bool refract(V3 v, V3 n, double ni_over_nt, V3 &refracted) { /* material.rs:54-64 */
    V3 uv = unit_vector(v);
    double dt = dot(uv, n);
    double discriminant = 1.0 - ni_over_nt * ni_over_nt * (1.0 - dt * dt);
    if (discriminant > 0.0) {
        refracted = ni_over_nt * (uv - n * dt) - n * std::sqrt(discriminant);
        return true;
    }
    return false;
}
inline double schlick(double cosine, double ref_idx) { /* material.rs:66-70 */
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * std::pow(1.0 - cosine, 5.0);
}

enum PdfKind { PDF_DUMMY = 0, PDF_COSINE = 1, PDF_SPHERE = 2 /* book semantics only */ };
struct Scatter { /* material.rs:133-138 */
    Ray specular_ray;
    bool is_specular;
    V3 albedo;
    PdfKind pdf;
    ONB uvw;
};

/* Material::emitted (material.rs:145-147 default, 303-305, 325-331) */
V3 emitted(const pt_scene_desc *s, int mat, const Ray &ray, const HitRec &rec) {
    const pt_material &m = s->materials[mat];
    switch (m.kind) {
    case PT_M_LAMBERTIAN:
        if (m.emissive > 0.0) return texture_value(s, m.texture, rec.u, rec.v, rec.p) * m.emissive;
        return v3(0, 0, 0);
    case PT_M_DIFFUSE_LIGHT:
        if (dot(rec.n, ray.d) < 0.0) return texture_value(s, m.texture, rec.u, rec.v, rec.p);
        return v3(0, 0, 0);
    default:
        return v3(0, 0, 0);
    }
}

/* Material::scatter (material.rs:159-161, 177-212, 230-241, 289-301, 321-323, 347-351) */
bool scatter(Ctx &cx, int mat, const Ray &r_in, const HitRec &rec, Scatter &out) {
    const pt_material &m = cx.s->materials[mat];
    switch (m.kind) {
    case PT_M_DIELECTRIC: {
        V3 outward_normal;
        V3 reflected = reflect(r_in.d, rec.n);
        double ni_over_nt;
        V3 refracted = v3(0, 0, 0);
        double reflect_prob;
        double cosine;
        if (dot(r_in.d, rec.n) > 0.0) {
            outward_normal = -rec.n;
            ni_over_nt = m.ref_idx;
            cosine = m.ref_idx * dot(r_in.d, rec.n) / length(r_in.d);
        } else {
            outward_normal = rec.n;
            ni_over_nt = 1.0 / m.ref_idx;
            cosine = -dot(r_in.d, rec.n) / length(r_in.d);
        }
        if (refract(r_in.d, outward_normal, ni_over_nt, refracted))
            reflect_prob = schlick(cosine, m.ref_idx);
        else
            reflect_prob = 1.0;
        if (cx.rng.next() < reflect_prob)
            out.specular_ray = Ray{rec.p, reflected, r_in.time};
        else
            out.specular_ray = Ray{rec.p, refracted, r_in.time};
        out.is_specular = true;
        out.albedo = v3(1.0, 1.0, 1.0);
        out.pdf = PDF_DUMMY;
        return true;
    }
    case PT_M_METAL: {
        V3 reflected = reflect(unit_vector(r_in.d), rec.n);
        V3 dir = reflected + m.fuzz * random_in_unit_sphere(cx.rng);
        out.specular_ray = Ray{rec.p, dir, r_in.time};
        out.is_specular = true;
        out.albedo = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
        out.pdf = PDF_DUMMY;
        return true;
    }
    case PT_M_LAMBERTIAN: {
        out.albedo = texture_value(cx.s, m.texture, rec.u, rec.v, rec.p);
        out.specular_ray = Ray{v3(0, 0, 0), v3(1, 0, 0), 0.0};
        out.is_specular = false;
        out.pdf = PDF_COSINE;
        out.uvw = onb_from_w(rec.n);
        return true;
    }
    case PT_M_ISOTROPIC: {
        out.specular_ray = Ray{rec.p, random_in_unit_sphere(cx.rng), r_in.time};
        out.albedo = texture_value(cx.s, m.texture, rec.u, rec.v, rec.p);
        out.is_specular = false;
        out.pdf = cx.book ? PDF_SPHERE : PDF_DUMMY;
        return true;
    }
    default: /* DUMMY, DIFFUSE_LIGHT */
        return false;
    }
}

/* Material::scattering_pdf (material.rs:142-144 default, 268-275) */
double scattering_pdf(const pt_scene_desc *s, int mat, const HitRec &rec, const Ray &scattered, bool book) {
    if (book && s->materials[mat].kind == PT_M_ISOTROPIC) return 1.0 / (4.0 * PI); /* unit_sphere_pdf, material.rs:46-48 */
    if (s->materials[mat].kind == PT_M_LAMBERTIAN) {
        double cosine = dot(rec.n, unit_vector(scattered.d));
        if (cosine < 0.0) return 0.0;
        return cosine * FRAC_1_PI;
    }
    return 0.0;
}

/* PDF::value / generate of the scatter's own pdf (material.rs:374-386, 439-446) */
inline double scatter_pdf_value(const Scatter &sc, V3 direction) {
    if (sc.pdf == PDF_COSINE) {
        double cosine = dot(unit_vector(direction), sc.uvw.w);
        if (cosine > 0.0) return cosine * FRAC_1_PI;
        return 0.0;
    }
    if (sc.pdf == PDF_SPHERE) return 1.0 / (4.0 * PI);
    return 0.0;
}
inline V3 scatter_pdf_generate(Ctx &cx, const Scatter &sc) {
    if (sc.pdf == PDF_COSINE) return onb_local(sc.uvw, random_cosine_direction(cx.rng));
    if (sc.pdf == PDF_SPHERE) return sc.specular_ray.d; /* the unit vector Isotropic::scatter drew (material.rs:349) */
    return v3(0, 0, 0);
}

/* -------------------------------------------------------------- color -- */

/* trace.rs:198-234 */
V3 color(Ctx &cx, const Ray &r, int depth, int max_depth) {
    HitRec rec;
    cx.depth = (uint32_t)depth;
    cx.c->segments++;
    if (hit(cx, cx.s->root, r, 0.001, F64_MAX, rec)) {
        V3 emissive = emitted(cx.s, rec.mat, r, rec);
        if (depth < max_depth) {
            Scatter sc;
            if (scatter(cx, rec.mat, r, rec, sc)) {
                if (sc.is_specular) {
                    return sc.albedo * color(cx, sc.specular_ray, depth + 1, max_depth);
                } else {
                    /* MixturePDF(HittablePDF(lights, p), scatter.pdf): material.rs:423-433 */
                    V3 dir;
                    if (cx.rng.next() < 0.5)
                        dir = lights_random(cx, rec.p);
                    else
                        dir = scatter_pdf_generate(cx, sc);
                    Ray scattered{rec.p, dir, r.time};
                    double pdf_val = 0.5 * lights_pdf_value(cx, rec.p, scattered.d) +
                                     0.5 * scatter_pdf_value(sc, scattered.d);
                    double spdf = scattering_pdf(cx.s, rec.mat, rec, scattered, cx.book);
                    V3 colour = sc.albedo * spdf * color(cx, scattered, depth + 1, max_depth) / pdf_val;
                    return colour + emissive;
                }
            }
        }
        return emissive;
    }
    return v3(0.0, 0.0, 0.0);
}

/* ------------------------------------------------------------- camera -- */

/* camera.rs:6-59 */
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    double time0, time1, lens_radius;
};
Camera camera_new(const pt_camera *c) {
    V3 origin = v3(c->origin[0], c->origin[1], c->origin[2]);
    V3 look_at = v3(c->look_at[0], c->look_at[1], c->look_at[2]);
    V3 vup = v3(c->vup[0], c->vup[1], c->vup[2]);
    double theta = c->vfov * PI / 180.0;
    double half_height = std::tan(theta / 2.0);
    double half_width = c->aspect * half_height;
    V3 w = unit_vector(origin - look_at);
    V3 u = unit_vector(cross(vup, w));
    V3 v = cross(w, u);
    Camera cam;
    cam.origin = origin;
    cam.lower_left_corner =
        origin - (u * half_width * c->focus_dist) - (v * half_height * c->focus_dist) - (w * c->focus_dist);
    cam.horizontal = u * 2.0 * half_width * c->focus_dist;
    cam.vertical = v * 2.0 * half_height * c->focus_dist;
    cam.u = u;
    cam.v = v;
    cam.w = w;
    cam.time0 = c->time0;
    cam.time1 = c->time1;
    cam.lens_radius = c->aperture / 2.0;
    return cam;
}
Ray camera_get_ray(const Camera &cam, Rng &g, double s, double t) {
    V3 rd = random_in_unit_disk(g) * cam.lens_radius;
    V3 offset = cam.u * rd.x + cam.v * rd.y;
    double time = cam.time0 + g.next() * (cam.time1 - cam.time0);
    V3 direction = cam.lower_left_corner + cam.horizontal * s + cam.vertical * t - cam.origin - offset;
    return Ray{cam.origin + offset, direction, time};
}

/* -------------------------------------------------------- tile render -- */

struct RenderJob {
    const pt_scene_desc *s;
    Camera cam;
    pt_render_args a;
    uint64_t seed;
    float *rgba;
    uint32_t *counts; /* per-pixel num_frames_per_pixel (trace.rs:80), image-sized, may be NULL offline */
};

/* TraceSceneBatchJob::trace (trace.rs:117-189) over one tile [x0,x1) x [y0,y1) */
void trace_tile(const RenderJob &job, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, Counters &cnt) {
    const pt_render_args &a = job.a;
    const bool realtime = a.mode == PT_MODE_REALTIME;
    for (uint32_t jj = y1; jj-- > y0;) { /* rows top to bottom of the tile, trace.rs:135 */
        uint32_t j = jj;
        /* interleaved split of a tile between GPUs (no reference counterpart): only rows y0 + phase + m * step */
        if (a.row_step > 1 && (j < a.y0 + a.row_phase || (j - a.y0 - a.row_phase) % a.row_step != 0)) continue;
        float *row = job.rgba + ((size_t)j * a.width) * 4;
        for (uint32_t i = x0; i < x1; ++i) {
            uint64_t pixel = (uint64_t)j * a.width + i;
            if (realtime && pixel_skip_draw(job.seed, a.frame, pixel) < 0.8) continue; /* trace.rs:144 */
            uint32_t nframes = 0;
            if (job.counts) {
                uint32_t &c = job.counts[pixel];
                c += (c <= 1000) ? 1 : 0; /* trace.rs:149 */
                nframes = c;
            }
            V3 pixel_colour = v3(0, 0, 0);
            for (uint32_t s = 0; s < a.spp; ++s) {
                Ctx cx;
                cx.s = job.s;
                cx.pkey = path_key(job.seed, a.frame, pixel, s);
                cx.book = (a.flags & PT_RENDER_BOOK_SEMANTICS) != 0;
                cx.rng = Rng{cx.pkey, 0};
                cx.depth = 0;
                cx.c = &cnt;
                cx.counting = true;
                cnt.paths++;
                double random = cx.rng.next();
                double u = ((double)i + random) / (double)a.width;
                random = cx.rng.next();
                double v = ((double)j + random) / (double)a.height;
                Ray r = camera_get_ray(job.cam, cx.rng, u, v);
                pixel_colour = pixel_colour + color(cx, r, 0, (int)a.max_depth);
            }
            pixel_colour = pixel_colour / (double)a.spp;
            float *px = row + (size_t)i * 4;
            if (realtime) { /* trace.rs:170-177 */
                float weight = 1.0f / (float)nframes;
                float one_minus_weight = 1.0f - weight;
                px[0] = (float)pixel_colour.x * weight + px[0] * one_minus_weight;
                px[1] = (float)pixel_colour.y * weight + px[1] * one_minus_weight;
                px[2] = (float)pixel_colour.z * weight + px[2] * one_minus_weight;
            } else {
                px[0] = (float)pixel_colour.x;
                px[1] = (float)pixel_colour.y;
                px[2] = (float)pixel_colour.z;
            }
        }
    }
}

uint32_t round_down_to_closest_factor(uint32_t factor_to_round, uint32_t factor_of) { /* math/mod.rs:28-31 */
    if (factor_to_round == 0) return factor_of;
    double factor = (double)factor_of / (double)factor_to_round;
    uint32_t f = (uint32_t)std::floor(factor);
    return f == 0 ? factor_of : factor_of / f;
}

} // namespace

/* =========================================================== C exports == */

extern "C" {

typedef struct oracle_stats {
    uint64_t paths, segments, node_visits, prim_tests;
    double seconds;
    uint32_t threads;
    uint32_t tiles;
} oracle_stats;

/* Hitable::hit on desc->root for a batch of rays (hitable.rs:28). */
int oracle_intersect(const pt_scene_desc *desc, const pt_ray *rays, size_t n, double t_min, double t_max,
                     uint64_t medium_seed, pt_hit *out, oracle_stats *stats) {
    Counters cnt;
    for (size_t i = 0; i < n; ++i) {
        Ctx cx;
        cx.s = desc;
        cx.pkey = intersect_key(medium_seed, i);
        cx.rng = Rng{cx.pkey, 0};
        cx.depth = 0;
        cx.c = &cnt;
        cx.counting = true;
        Ray r{v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]),
              v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]), rays[i].time};
        HitRec rec;
        cnt.segments++;
        if (hit(cx, desc->root, r, t_min, t_max, rec)) {
            out[i].t = rec.t;
            out[i].u = rec.u;
            out[i].v = rec.v;
            out[i].p[0] = rec.p.x; out[i].p[1] = rec.p.y; out[i].p[2] = rec.p.z;
            out[i].normal[0] = rec.n.x; out[i].normal[1] = rec.n.y; out[i].normal[2] = rec.n.z;
            out[i].prim = rec.prim;
            out[i].material = rec.mat;
        } else {
            std::memset(&out[i], 0, sizeof(pt_hit));
            out[i].prim = -1;
            out[i].material = -1;
        }
    }
    if (stats) {
        stats->paths = 0;
        stats->segments = cnt.segments;
        stats->node_visits = cnt.node_visits;
        stats->prim_tests = cnt.prim_tests;
    }
    return 0;
}

/* The offline / realtime dispatch of lib.rs:345-399 + trace.rs:117-189: tiles of (nx/9) x (ny/9) pixels on a
 * FIFO pool of `threads` workers (jobs/mod.rs:66-124). Unlike lib.rs:350 the tile grid covers the whole
 * render rectangle (edge tiles are clipped instead of dropped). */
int oracle_render(const pt_scene_desc *desc, const pt_camera *camera, const pt_render_args *args, uint64_t seed,
                  int threads, float *rgba, uint32_t *counts, oracle_stats *stats) {
    if (!desc || !camera || !args || !rgba) return -1;
    RenderJob job;
    job.s = desc;
    job.cam = camera_new(camera);
    job.a = *args;
    job.seed = seed;
    job.rgba = rgba;
    job.counts = counts;
    uint32_t tw = round_down_to_closest_factor(args->width / 9, args->width);
    uint32_t th = round_down_to_closest_factor(args->height / 9, args->height);
    if (tw == 0) tw = 1;
    if (th == 0) th = 1;
    struct Tile {
        uint32_t x0, y0, x1, y1;
    };
    std::vector<Tile> tiles;
    for (uint32_t ty = args->y0; ty < args->y1; ty += th)
        for (uint32_t tx = args->x0; tx < args->x1; tx += tw)
            tiles.push_back(Tile{tx, ty, std::min(tx + tw, args->x1), std::min(ty + th, args->y1)});
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    std::atomic<size_t> next(0);
    std::vector<Counters> per(threads);
    auto t_begin = std::chrono::steady_clock::now();
    auto worker = [&](int w) {
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= tiles.size()) break;
            trace_tile(job, tiles[k].x0, tiles[k].y0, tiles[k].x1, tiles[k].y1, per[w]);
        }
    };
    if (threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int w = 0; w < threads; ++w) pool.emplace_back(worker, w);
        for (auto &t : pool) t.join();
    }
    auto t_end = std::chrono::steady_clock::now();
    if (stats) {
        Counters sum;
        for (auto &c : per) {
            sum.paths += c.paths;
            sum.segments += c.segments;
            sum.node_visits += c.node_visits;
            sum.prim_tests += c.prim_tests;
        }
        stats->paths = sum.paths;
        stats->segments = sum.segments;
        stats->node_visits = sum.node_visits;
        stats->prim_tests = sum.prim_tests;
        stats->seconds = std::chrono::duration<double>(t_end - t_begin).count();
        stats->threads = (uint32_t)threads;
        stats->tiles = (uint32_t)tiles.size();
    }
    return 0;
}

/* color() for explicit camera samples: path i uses pixel index pixel[i] and sample index sample[i];
 * returns the per-path radiance (before the per-pixel mean) so single paths can be compared. */
int oracle_trace_paths(const pt_scene_desc *desc, const pt_camera *camera, uint32_t width, uint32_t height,
                       uint32_t max_depth, uint64_t seed, uint32_t frame, const uint64_t *pixel,
                       const uint32_t *sample, size_t n, double *rgb_out) {
    Camera cam = camera_new(camera);
    Counters cnt;
    for (size_t k = 0; k < n; ++k) {
        uint64_t px = pixel[k];
        uint32_t i = (uint32_t)(px % width), j = (uint32_t)(px / width);
        Ctx cx;
        cx.s = desc;
        cx.pkey = path_key(seed, frame, px, sample[k]);
        cx.rng = Rng{cx.pkey, 0};
        cx.depth = 0;
        cx.c = &cnt;
        cx.counting = true;
        double random = cx.rng.next();
        double u = ((double)i + random) / (double)width;
        random = cx.rng.next();
        double v = ((double)j + random) / (double)height;
        Ray r = camera_get_ray(cam, cx.rng, u, v);
        V3 c = color(cx, r, 0, (int)max_depth);
        rgb_out[3 * k] = c.x;
        rgb_out[3 * k + 1] = c.y;
        rgb_out[3 * k + 2] = c.z;
    }
    return 0;
}

/* Camera::get_ray for pixel-space samples driven by the path stream (camera.rs:53-59, trace.rs:153-158). */
int oracle_camera_rays(const pt_camera *camera, uint32_t width, uint32_t height, uint64_t seed, uint32_t frame,
                       const uint64_t *pixel, const uint32_t *sample, size_t n, pt_ray *out) {
    Camera cam = camera_new(camera);
    for (size_t k = 0; k < n; ++k) {
        uint64_t px = pixel[k];
        uint32_t i = (uint32_t)(px % width), j = (uint32_t)(px / width);
        Rng g{path_key(seed, frame, px, sample[k]), 0};
        double random = g.next();
        double u = ((double)i + random) / (double)width;
        random = g.next();
        double v = ((double)j + random) / (double)height;
        Ray r = camera_get_ray(cam, g, u, v);
        out[k].origin[0] = r.o.x; out[k].origin[1] = r.o.y; out[k].origin[2] = r.o.z;
        out[k].direction[0] = r.d.x; out[k].direction[1] = r.d.y; out[k].direction[2] = r.d.z;
        out[k].time = r.time;
    }
    return 0;
}

/* Camera::new derived fields: origin, lower_left_corner, horizontal, vertical, u, v, w (7 x 3 doubles). */
int oracle_camera_frame(const pt_camera *camera, double *out21) {
    Camera c = camera_new(camera);
    V3 f[7] = {c.origin, c.lower_left_corner, c.horizontal, c.vertical, c.u, c.v, c.w};
    for (int i = 0; i < 7; ++i) {
        out21[3 * i] = f[i].x;
        out21[3 * i + 1] = f[i].y;
        out21[3 * i + 2] = f[i].z;
    }
    return 0;
}

/* Cosine-weighted secondary rays about a hit normal, for building extend test batches:
 * origin = hit p, direction = CosinePDF(normal).generate() drawn from stream (seed, i). */
int oracle_secondary_rays(const pt_hit *hits, const pt_ray *primary, size_t n, uint64_t seed, pt_ray *out,
                          uint8_t *valid) {
    for (size_t i = 0; i < n; ++i) {
        if (hits[i].prim < 0) {
            valid[i] = 0;
            out[i] = primary[i];
            continue;
        }
        valid[i] = 1;
        Rng g{mix64(seed ^ mix64(i + 1)), 0};
        ONB uvw = onb_from_w(v3(hits[i].normal[0], hits[i].normal[1], hits[i].normal[2]));
        V3 d = onb_local(uvw, random_cosine_direction(g));
        out[i].origin[0] = hits[i].p[0]; out[i].origin[1] = hits[i].p[1]; out[i].origin[2] = hits[i].p[2];
        out[i].direction[0] = d.x; out[i].direction[1] = d.y; out[i].direction[2] = d.z;
        out[i].time = primary[i].time;
    }
    return 0;
}

int oracle_texture_value(const pt_scene_desc *desc, int tex, double u, double v, const double p[3], double out[3]) {
    V3 c = texture_value(desc, tex, u, v, v3(p[0], p[1], p[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    return 0;
}
double oracle_perlin_noise(const pt_perlin *t, const double p[3]) { return perlin_noise(t, v3(p[0], p[1], p[2])); }
double oracle_perlin_turb(const pt_perlin *t, const double p[3], int depth) {
    return perlin_turb(t, v3(p[0], p[1], p[2]), depth);
}

int oracle_bounding_box(const pt_scene_desc *desc, int node, double t0, double t1, double mn[3], double mx[3]) {
    AABB b;
    if (!bounding_box(desc, node, t0, t1, b)) return -1;
    mn[0] = b.mn.x; mn[1] = b.mn.y; mn[2] = b.mn.z;
    mx[0] = b.mx.x; mx[1] = b.mx.y; mx[2] = b.mx.z;
    return 0;
}
int oracle_aabb_hit(const double mn[3], const double mx[3], const pt_ray *r, double tmin, double tmax) {
    AABB b{v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2])};
    Ray ray{v3(r->origin[0], r->origin[1], r->origin[2]), v3(r->direction[0], r->direction[1], r->direction[2]),
            r->time};
    return aabb_hit(b, ray, tmin, tmax) ? 1 : 0;
}
int oracle_rotate_y_box(const double mn[3], const double mx[3], double degrees, double out_sin_cos[2],
                        double out_mn[3], double out_mx[3]) {
    double radians = degrees * (PI / 180.0); /* f64::to_radians */
    double s = std::sin(radians), c = std::cos(radians);
    AABB b = rotate_y_box(AABB{v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2])}, s, c);
    out_sin_cos[0] = s; out_sin_cos[1] = c;
    out_mn[0] = b.mn.x; out_mn[1] = b.mn.y; out_mn[2] = b.mn.z;
    out_mx[0] = b.mx.x; out_mx[1] = b.mx.y; out_mx[2] = b.mx.z;
    return 0;
}

/* BvhNode::from_list (bvh.rs:13-69) over `nodes`, appending BVH nodes in post-order (left subtree, right subtree,
 * self). New node k gets index desc->n_hitables + k. Writes (left, right) to lr_out[2k..] and the box to
 * box_out[6k..]; returns the number of BVH nodes created (capacity must be >= 2n: singleton sub-lists get their own node).  The comparator
 * "a.min < b.min ? Less : Greater" is applied with a stable sort; the axis draw floor(3*rand) comes from the
 * scene stream (key derived from `seed`, `*draws` draws already consumed). */
struct SceneStream { /* SplitMix64 stream used while building scenes (BVH axis draws) */
    uint64_t key;
    uint64_t n;
    double next() {
        n += 1;
        return to_unit(mix64(key + n * GOLDEN));
    }
};
struct BvhBuild {
    const pt_scene_desc *s;
    SceneStream rng;
    int32_t *lr;
    double *box;
    int created;
    double t0, t1;
    std::vector<AABB> extra; /* boxes of created nodes */
};
static AABB bb_of(BvhBuild &b, int node, double t0, double t1) {
    if (node >= b.s->n_hitables) return b.extra[node - b.s->n_hitables];
    AABB o;
    bounding_box(b.s, node, t0, t1, o);
    return o;
}
static int bvh_from_list(BvhBuild &b, std::vector<int> list) {
    int axis = (int)std::floor(b.rng.next() * 3.0);
    std::stable_sort(list.begin(), list.end(), [&](int x, int y) {
        AABB bx = bb_of(b, x, 0.0, 0.0), by = bb_of(b, y, 0.0, 0.0);
        return idx(bx.mn, axis) - idx(by.mn, axis) < 0.0;
    });
    int left, right;
    size_t n = list.size();
    if (n == 1) {
        left = right = list[0];
    } else if (n == 2) {
        left = list[0];
        right = list[1];
    } else {
        size_t half = n / 2;
        std::vector<int> first(list.begin(), list.begin() + half), second(list.begin() + half, list.end());
        left = bvh_from_list(b, first);
        right = bvh_from_list(b, second);
    }
    AABB u = aabb_union(bb_of(b, left, b.t0, b.t1), bb_of(b, right, b.t0, b.t1));
    int k = b.created++;
    b.lr[2 * k] = left;
    b.lr[2 * k + 1] = right;
    double *bo = b.box + 6 * k;
    bo[0] = u.mn.x; bo[1] = u.mn.y; bo[2] = u.mn.z; bo[3] = u.mx.x; bo[4] = u.mx.y; bo[5] = u.mx.z;
    b.extra.push_back(u);
    return b.s->n_hitables + k;
}
int oracle_bvh_from_list(const pt_scene_desc *desc, const int32_t *nodes, int n, double time0, double time1,
                         uint64_t seed, uint64_t *draws, int32_t *lr_out, double *box_out) {
    BvhBuild b;
    b.s = desc;
    b.rng = SceneStream{scene_stream_key(seed), *draws};
    b.lr = lr_out;
    b.box = box_out;
    b.created = 0;
    b.t0 = time0;
    b.t1 = time1;
    std::vector<int> list(nodes, nodes + n);
    bvh_from_list(b, list);
    *draws = b.rng.n;
    return b.created;
}

/* The n-th draw (1-based) of a path stream / the scene stream, for pinning the RNG definition. */
double oracle_path_draw(uint64_t seed, uint32_t frame, uint64_t pixel, uint32_t sample, uint64_t n) {
    return (double)path_draw_u32(path_key(seed, frame, pixel, sample), (uint32_t)n) * (1.0 / 4294967296.0);
}
double oracle_scene_draw(uint64_t seed, uint64_t n) { return to_unit(mix64(scene_stream_key(seed) + n * GOLDEN)); }
double oracle_pixel_skip_draw(uint64_t seed, uint32_t frame, uint64_t pixel) {
    return pixel_skip_draw(seed, frame, pixel);
}

/* Vec3 known-answer helpers (math/vec3.rs:495-577): op 0 add, 1 sub, 2 mul, 3 div (vec), 4..7 same with scalar b[0] */
int oracle_vec3_op(int op, const double a[3], const double b[3], double out[3]) {
    V3 x = v3(a[0], a[1], a[2]), y = v3(b[0], b[1], b[2]), r;
    switch (op) {
    case 0: r = x + y; break;
    case 1: r = x - y; break;
    case 2: r = x * y; break;
    case 3: r = x / y; break;
    case 4: r = x + b[0]; break;
    case 5: r = x - b[0]; break;
    case 6: r = x * b[0]; break;
    case 7: r = x / b[0]; break;
    case 8: r = unit_vector(x); break;
    case 9: r = cross(x, y); break;
    default: return -1;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 0;
}
double oracle_vec3_scalar(int op, const double a[3], const double b[3]) {
    V3 x = v3(a[0], a[1], a[2]), y = v3(b[0], b[1], b[2]);
    switch (op) {
    case 0: return squared_length(x);
    case 1: return length(x);
    case 2: return dot(x, y);
    default: return 0.0;
    }
}

/* reinhard_tonemap + sqrt gamma + `as u8` (trace.rs:236-242, lib.rs:330-340) and the PPM row flip (lib.rs:617-630) */
int oracle_tonemap_rgb8(const float *rgba, uint32_t width, uint32_t height, uint8_t *rgb8_out) {
    for (uint32_t j = 0; j < height; ++j) {
        uint32_t jf = height - j - 1;
        for (uint32_t i = 0; i < width; ++i) {
            const float *px = rgba + ((size_t)jf * width + i) * 4;
            for (int c = 0; c < 3; ++c) {
                double col = (double)px[c] * 1.5;
                double tm = col / (col + 1.0);
                double val = 255.99 * std::sqrt(tm);
                uint8_t b; /* Rust `as u8`: saturating, NaN -> 0 */
                if (!(val > 0.0)) b = 0;
                else if (val >= 255.0) b = 255;
                else b = (uint8_t)val;
                rgb8_out[((size_t)j * width + i) * 3 + c] = b;
            }
        }
    }
    return 0;
}

/* Camera::update_from_input followed by Camera::update (camera.rs:89-96,109-215) on a camera held as plain numbers.
 * cam = origin[3], look_at[3], vup[3] (in/out; the other Camera::new parameters do not change). keys are the bit
 * masks of include/ptgpu.h (W S A D E Q = bits 0-5 read from `pressed`; Right Left Up Down = bits 6-9 read from
 * `held`). Returns camera_moved. Written against the reference text, independently of the host library. */
int oracle_camera_apply_input(double *cam, uint32_t pressed, uint32_t held, float mouse_dx, float mouse_dy,
                              double frame_time) {
    V3 origin = v3(cam[0], cam[1], cam[2]), look_at = v3(cam[3], cam[4], cam[5]), vup = v3(cam[6], cam[7], cam[8]);
    /* the frame the camera holds since its last update() (camera.rs:30-32) */
    V3 w = unit_vector(origin - look_at);
    V3 u = unit_vector(cross(vup, w));
    V3 v = cross(w, u);
    const double CAM_SPEED = 40.0, MOUSE_LOOK_SPEED = 1.0;
    bool moved = false;
    struct Move { uint32_t bit; int axis; double sign; };
    /* order of the `if`s in camera.rs:121-167: W S D A E Q; axis 0 = forward (-w), 1 = right (u), 2 = up (v) */
    const Move moves[6] = {{1u << 0, 0, 1.0}, {1u << 1, 0, -1.0}, {1u << 3, 1, 1.0}, {1u << 2, 1, -1.0}, {1u << 4, 2, 1.0}, {1u << 5, 2, -1.0}};
    for (const Move &m : moves) {
        if (!(pressed & m.bit)) continue;
        V3 axis = m.axis == 0 ? -w : (m.axis == 1 ? u : v);
        V3 diff = axis * m.sign * CAM_SPEED * frame_time;
        V3 new_origin = origin + diff;
        look_at = look_at + (new_origin - origin); /* set_origin(.., true), camera.rs:77-83 */
        origin = new_origin;
        moved = true;
    }
    auto set_look_at = [&](V3 target) { /* camera.rs:98-107 with maintain_distance: scales the POSITION vector */
        double look_at_dist = length(origin - look_at);
        double new_look_at_dist = length(origin - target);
        look_at = target * (look_at_dist / new_look_at_dist);
    };
    const Move looks[4] = {{1u << 6, 1, 1.0}, {1u << 7, 1, -1.0}, {1u << 8, 2, 1.0}, {1u << 9, 2, -1.0}};
    for (const Move &m : looks) { /* camera.rs:169-192: Right Left Up Down */
        if (!(held & m.bit)) continue;
        V3 axis = m.axis == 1 ? u : v;
        set_look_at(look_at + axis * m.sign * CAM_SPEED * frame_time);
        moved = true;
    }
    if (mouse_dx != 0.0f || mouse_dy != 0.0f) { /* camera.rs:193-211 */
        V3 target = look_at;
        if (mouse_dx != 0.0f) target = target + u * MOUSE_LOOK_SPEED * frame_time * (double)mouse_dx;
        if (mouse_dy != 0.0f) target = target + v * MOUSE_LOOK_SPEED * frame_time * (double)mouse_dy;
        set_look_at(target);
        moved = true;
    }
    cam[0] = origin.x; cam[1] = origin.y; cam[2] = origin.z;
    cam[3] = look_at.x; cam[4] = look_at.y; cam[5] = look_at.z;
    return moved ? 1 : 0;
}

/* The display transform of the rendy tonemap pass (assets/shaders/tonemap.frag:37-70), in f32 like the shader:
 * colour * exposure, ACESFitted (input matrix, RRTAndODTFit, output matrix, clamp), vignette 1 - distance(uv, 0.5)
 * at the fragment centre, uv.y flipped. out row 0 = top of the screen, alpha 1. */
static float oracle_aces_fit(float v) {
    float a = v * (v + 0.0245786f) - 0.000090537f;
    float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
int oracle_display_transform(const float *rgba, uint32_t width, uint32_t height, float exposure, float *out) {
    static const float in_m[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83777f}};
    static const float out_m[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
    for (uint32_t j = 0; j < height; ++j) {
        for (uint32_t i = 0; i < width; ++i) {
            const float *px = rgba + ((size_t)(height - j - 1) * width + i) * 4;
            float c[3] = {px[0] * exposure, px[1] * exposure, px[2] * exposure};
            float t[3], f[3];
            for (int k = 0; k < 3; ++k) t[k] = c[0] * in_m[k][0] + c[1] * in_m[k][1] + c[2] * in_m[k][2];
            for (int k = 0; k < 3; ++k) f[k] = oracle_aces_fit(t[k]);
            float u = ((float)i + 0.5f) / (float)width, v = 1.0f - ((float)j + 0.5f) / (float)height;
            float du = u - 0.5f, dv = v - 0.5f;
            float vignette = 1.0f - std::sqrt(du * du + dv * dv);
            float *o = out + ((size_t)j * width + i) * 4;
            for (int k = 0; k < 3; ++k) {
                float x = f[0] * out_m[k][0] + f[1] * out_m[k][1] + f[2] * out_m[k][2];
                x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
                o[k] = x * vignette;
            }
            o[3] = 1.0f;
        }
    }
    return 0;
}

} /* extern "C" */
