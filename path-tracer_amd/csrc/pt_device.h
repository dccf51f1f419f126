// Device-side arithmetic of the render path: vectors, the counter RNG, primitive tests, hit finalisation,
// textures, materials and light sampling. Everything is f64, the reference's arithmetic type, and is
// compiled with -ffp-contract=off so that products and sums round exactly as in the reference (Rust never
// fuses a*b+c). FP64 vector rate on gfx950 is half the FP32 rate, which is what makes this affordable.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptgpu.h"
#include "device_scene.h"

namespace ptdev {

#define PT_DEV __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kFracPi2 = 1.57079632679489661923132169163975144;
constexpr double kFrac1Pi = 0.318309886183790671537767526745028724;
constexpr double kF64Max = 1.7976931348623157e308;
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;

// ------------------------------------------------------------------- RNG --
// Stands in for random::rand (src/math/random.rs:3-6), which is an OS-seeded thread_rng and cannot be
// reproduced. Keys are built with the SplitMix64 finaliser (once per path); a path's n-th draw is two rounds
// of the 32-bit PCG output hash over (key, n): 32-bit integer multiplies only, which is what the VALU is fast at.
PT_DEV uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
PT_DEV double to_unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
PT_DEV uint32_t pcg32(uint32_t v) {
    const uint32_t s = v * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
struct Rng {
    uint64_t key;
    uint32_t n;
    PT_DEV uint32_t next_u32() {
        n += 1;
        const uint32_t x = pcg32((uint32_t)key + n * 0x9E3779B9u);
        return pcg32(x ^ (uint32_t)(key >> 32));
    }
    // the draw `ahead` places further on, without consuming anything
    PT_DEV uint32_t peek_u32(uint32_t ahead) const {
        const uint32_t x = pcg32((uint32_t)key + (n + ahead) * 0x9E3779B9u);
        return pcg32(x ^ (uint32_t)(key >> 32));
    }
    PT_DEV double next() { return (double)next_u32() * (1.0 / 4294967296.0); }
};
PT_DEV uint64_t path_key(uint64_t seed, uint64_t frame, uint64_t pixel, uint64_t sample) {
    uint64_t k = mix64(seed ^ (0xA0761D6478BD642FULL * (frame + 1)));
    k = mix64(k ^ pixel);
    k = mix64(k + kGolden * (sample + 1));
    return k;
}
PT_DEV double medium_draw(uint64_t pkey, uint32_t depth, int32_t medium_node) {
    uint64_t salt = mix64(0x4D454449554D0000ULL + ((uint64_t)depth << 32) + (uint64_t)(uint32_t)medium_node);
    return to_unit(mix64(pkey ^ salt));
}
PT_DEV double pixel_skip_draw(uint64_t seed, uint64_t frame, uint64_t pixel) {
    uint64_t k = mix64(seed ^ 0x534B495050495845ULL ^ (0xA0761D6478BD642FULL * (frame + 1)));
    return to_unit(mix64(k ^ pixel));
}
PT_DEV uint64_t intersect_key(uint64_t medium_seed, uint64_t i) { return mix64(medium_seed ^ mix64(i + kGolden)); }

// ------------------------------------------------------------------ Vec3 --
struct D3 {
    double x, y, z;
};
PT_DEV D3 d3(double x, double y, double z) { return D3{x, y, z}; }
PT_DEV D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV D3 operator*(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
PT_DEV D3 operator*(double s, D3 a) { return D3{a.x * s, a.y * s, a.z * s}; }
PT_DEV D3 operator/(D3 a, double s) { return D3{a.x / s, a.y / s, a.z / s}; }
PT_DEV D3 operator-(D3 a) { return D3{-a.x, -a.y, -a.z}; }
PT_DEV double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV D3 cross(D3 a, D3 b) { // src/math/vec3.rs:137-143
    return D3{a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x};
}
PT_DEV double squared_length(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PT_DEV double length(D3 a) { return sqrt(squared_length(a)); }
PT_DEV D3 unit_vector(D3 a) { return a / length(a); }

// ---------------------------------------------------------- math policy --
// X = true  : "exact" mode. Every quotient is a true IEEE division and nothing is fused, so results match an
//             f64 CPU evaluation of the reference's expressions bit for bit (up to libm differences).
// X = false : "fast" mode (default). Still f64 from end to end: quotients become a multiply by a Newton-refined
//             v_rcp_f64 (about 1 ulp), square roots a Newton-refined v_rsq_f64, sums of products use FMA, and
//             sin / cos of 2 pi r come from a quadrant reduction and two polynomials instead of libm's general
//             routines. Results differ from exact mode by a few ulp per operation, far inside the 1e-4 relative bar
//             on hit distances.
PT_DEV double fast_rcp(double y) {
    const double r0 = __builtin_amdgcn_rcp(y); // ~2^-26 relative error
    double e = __builtin_fma(-y, r0, 1.0);
    double r = __builtin_fma(e, r0, r0);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(e, r, r);
    // y = 0 or inf: the refinement produces NaN (0 * inf); keep the hardware's inf / 0, as IEEE division would
    return (__builtin_isinf(r0) || r0 == 0.0) ? r0 : r;
}
// 1 / sqrt(x) for finite x > 0: hardware estimate (about 2^-23 relative) and two Newton steps.
PT_DEV double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    return y;
}
// The same with ONE refinement step: relative error about 2^-50 (reciprocal) and 2^-44 (square roots). Used where the
// value only WEIGHTS a path (pdf values, cosines, sampled local directions): seven to nine decimal digits more than
// an f32 evaluation, at two thirds of the instructions of the fully refined forms. Geometry (hit distances, hit
// points, ray directions that are traced) never uses these.
PT_DEV double weight_rcp(double y) {
    const double r0 = __builtin_amdgcn_rcp(y);
    const double e = __builtin_fma(-y, r0, 1.0);
    const double r = __builtin_fma(e, r0, r0);
    return (__builtin_isinf(r0) || r0 == 0.0) ? r0 : r;
}
PT_DEV double weight_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(y * 0.5, e, y);
}
PT_DEV double weight_sqrt(double x) {
    const double r = x * weight_rsqrt(x);
    return x == 0.0 ? 0.0 : r;
}
// sqrt(x) for finite x >= 0 (x = 0 gives 0, not 0 * inf)
PT_DEV double fast_sqrt(double x) {
    const double r = x * fast_rsqrt(x);
    return x == 0.0 ? 0.0 : r;
}
// sin(2 pi r), cos(2 pi r) for r in [0, 1], f64, about 1 ulp: r is split exactly into a quarter turn count q and a
// remainder f in [-1/8, 1/8]; x = 2 pi f lies in [-pi/4, pi/4], where the two minimax polynomials below (the kernels
// every libm uses on that interval) are accurate to below one ulp; the quadrant then swaps and negates. No table, no
// branch, about thirty FMA-class instructions for both values.
PT_DEV void fast_sincos_2pi(double r, double &sn, double &cs) {
    const double q = __builtin_rint(r * 4.0);
    const double f = __builtin_fma(q, -0.25, r); // exact
    const double x = f * 6.283185307179586476925286766559;
    const double z = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double s0 = __builtin_fma(x * z, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double c0 = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int k = (int)q; // 0 .. 4
    const bool swap = (k & 1) != 0;
    const double a = swap ? c0 : s0, b = swap ? s0 : c0;
    sn = (k & 2) ? -a : a;
    cs = ((k + 1) & 2) ? -b : b;
}

// log(x) for x in [2^-53, 1), the range of a unit draw (x = 0 gives -inf, as the library does), about 1 ulp: x = 2^k m
// with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the seven
// minimax coefficients every libm uses for R, and k ln 2 added in two parts. Thirty-odd instructions where the library's
// general routine, which also serves denormals, negatives and huge arguments in double-double steps, takes a hundred.
PT_DEV double fast_log_unit(double x) {
    double m = __builtin_amdgcn_frexp_mant(x); // [1/2, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double den = 2.0 + f; // [1.7, 2.42)
    const double r0 = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r0, 1.0);
    double r = __builtin_fma(e, r0, r0);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(e, r, r);
    const double s = f * r;
    const double z = s * s, w = z * z;
    double t1 = __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01);
    t1 = __builtin_fma(w, t1, 3.999999999940941908e-01);
    t1 = w * t1;
    double t2 = __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01);
    t2 = __builtin_fma(w, t2, 2.857142874366239149e-01);
    t2 = __builtin_fma(w, t2, 6.666666666666735130e-01);
    const double big_r = __builtin_fma(z, t2, t1);
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    const double tail = __builtin_fma(s, hfsq + big_r, dk * 1.90821492927058770002e-10);
    const double v = __builtin_fma(dk, 6.93147180369123816490e-01, -((hfsq - tail) - f));
    return x == 0.0 ? -__builtin_inf() : v;
}

template <bool X> struct M {
    static PT_DEV double rcp(double y) {
        if constexpr (X) return 1.0 / y;
        else return fast_rcp(y);
    }
    static PT_DEV double div(double a, double b) {
        if constexpr (X) return a / b;
        else return a * fast_rcp(b);
    }
    static PT_DEV double mad(double a, double b, double c) { // a * b + c
        if constexpr (X) return a * b + c;
        else return __builtin_fma(a, b, c);
    }
    static PT_DEV double dot(D3 a, D3 b) {
        if constexpr (X) return a.x * b.x + a.y * b.y + a.z * b.z;
        else return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x));
    }
    static PT_DEV double squared_length(D3 a) { return dot(a, a); }
    static PT_DEV double sqrt_(double v) {
        if constexpr (X) return sqrt(v);
        else return fast_sqrt(v);
    }
    static PT_DEV double length(D3 a) {
        if constexpr (X) return sqrt(dot(a, a));
        else return fast_sqrt(dot(a, a));
    }
    static PT_DEV D3 divs(D3 a, double s) {
        if constexpr (X) return D3{a.x / s, a.y / s, a.z / s};
        else {
            const double r = fast_rcp(s);
            return D3{a.x * r, a.y * r, a.z * r};
        }
    }
    static PT_DEV D3 unit_vector(D3 a) {
        if constexpr (X) return divs(a, length(a));
        else {
            const double r = fast_rsqrt(dot(a, a));
            return D3{a.x * r, a.y * r, a.z * r};
        }
    }
    static PT_DEV D3 point_at(D3 o, D3 d, double t) { // Ray::point_at_parameter, src/math/ray.rs:30-32
        if constexpr (X) return D3{o.x + d.x * t, o.y + d.y * t, o.z + d.z * t};
        else return D3{__builtin_fma(d.x, t, o.x), __builtin_fma(d.y, t, o.y), __builtin_fma(d.z, t, o.z)};
    }
    // cos(2 pi r), sin(2 pi r) for r in [0, 1)
    static PT_DEV void sincos_2pi(double r, double &sn, double &cs) {
        if constexpr (X) {
            const double phi = 2.0 * kPi * r;
            sincos(phi, &sn, &cs); // one argument reduction for both; the values of sin(phi) and cos(phi)
        } else {
            fast_sincos_2pi(r, sn, cs);
        }
    }
};

// Arithmetic of the quantities that only weight a path. Exact mode: the same IEEE operations as M<true>.
template <bool X> struct W {
    static PT_DEV double rcp(double y) {
        if constexpr (X) return 1.0 / y;
        else return weight_rcp(y);
    }
    static PT_DEV double div(double a, double b) {
        if constexpr (X) return a / b;
        else return a * weight_rcp(b);
    }
    static PT_DEV double sqrt_(double v) {
        if constexpr (X) return sqrt(v);
        else return weight_sqrt(v);
    }
    static PT_DEV D3 divs(D3 a, double s) {
        if constexpr (X) return D3{a.x / s, a.y / s, a.z / s};
        else {
            const double r = weight_rcp(s);
            return D3{a.x * r, a.y * r, a.z * r};
        }
    }
};

// ---------------------------------------------------------- scene tables --
struct SceneView {
    const DevOp *ops;
    const DevXform *xforms;
    const DevLeaf *leaves;
    const DevLight *lights;
    const pt_material *materials;
    const pt_texture *textures;
    const uint8_t *texels;
    const pt_perlin *perlin;
    int32_t n_traversal_ops;
    int32_t n_leaves;
    int32_t n_lights;
    int32_t has_medium;
    int32_t has_frames;
    // A throughput of exactly 0 can still turn NaN further down the path (0 * x / 0 or 0 * NaN, src/trace.rs:217-220):
    // some material is Isotropic, or a diffuse surface reaches into a sphere light shape, where Sphere::pdf_value is
    // the square root of a negative number (src/sphere.rs:92-94). Fast mode then follows such paths to their end.
    int32_t zero_can_turn_nan;
};

// Records read at a wave-uniform address come through the scalar cache into scalar registers (the constant address
// space is how the compiler is told that it may do so; the scene's tables are never written while a kernel runs).
// Any trivially copyable record, word by word (the compiler merges the words into wide scalar loads).
template <class T> PT_DEV T load_uniform(const T *p) {
    static_assert(sizeof(T) % 4 == 0 && __is_trivially_copyable(T), "copied word by word");
    const uint32_t __attribute__((address_space(4))) *src = (const uint32_t __attribute__((address_space(4))) *)p;
    union { // (the words are read back as T through the union, not through a pointer of another type)
        T value;
        uint32_t words[sizeof(T) / 4];
    } out;
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) out.words[k] = src[k];
    return out.value;
}

// ------------------------------------------------------------ transforms --
// Ray into a wrapper's child space: Translate::hit (src/hitable.rs:124), then RotateY::hit (204-206, via 182-187).
template <bool X> PT_DEV void xform_ray(const DevXform &x, D3 &o, D3 &d) {
    o = o - d3(x.tx, x.ty, x.tz);
    const double s = x.s, c = x.c;
    if constexpr (X) {
        o = d3(c * o.x - s * o.z, o.y, s * o.x + c * o.z);
        d = d3(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
    } else {
        o = d3(__builtin_fma(c, o.x, -(s * o.z)), o.y, __builtin_fma(s, o.x, c * o.z));
        d = d3(__builtin_fma(c, d.x, -(s * d.z)), d.y, __builtin_fma(s, d.x, c * d.z));
    }
}
template <bool X> PT_DEV void apply_chain(const DevXform *xf, int start, int count, D3 &o, D3 &d) {
    for (int k = 0; k < count; ++k) xform_ray<X>(xf[start + k], o, d);
}
// Hit record back out of a wrapper: hitable.rs:209-210 (rotate, via 189-194), 126 (translate), 95 (flip).
template <bool X> PT_DEV void unxform_hit(const DevXform &x, D3 &p, D3 &n) {
    const double s = x.s, c = x.c;
    if constexpr (X) {
        p = d3(c * p.x + s * p.z, p.y, -s * p.x + c * p.z);
        n = d3(c * n.x + s * n.z, n.y, -s * n.x + c * n.z);
    } else {
        p = d3(__builtin_fma(c, p.x, s * p.z), p.y, __builtin_fma(-s, p.x, c * p.z));
        n = d3(__builtin_fma(c, n.x, s * n.z), n.y, __builtin_fma(-s, n.x, c * n.z));
    }
    p = p + d3(x.tx, x.ty, x.tz);
    n = n * x.sign;
}

// ------------------------------------------------------------ primitives --
// AABB::hit (src/math/aabb.rs:27-47). inv = 1/d per axis is hoisted out of the per-node loop: the reference
// recomputes the same quotient at every node, so the values are identical.
PT_DEV bool aabb_hit(const double *b, D3 o, D3 inv, double tmin, double tmax) {
    // The reference returns false at the first axis whose interval is empty; the later axes cannot turn that into a hit,
    // so all three are evaluated here without a branch (the same comparisons on the same values, NaN included).
    bool ok;
    {
        double t0 = (b[0] - o.x) * inv.x, t1 = (b[3] - o.x) * inv.x;
        const bool neg = inv.x < 0.0;
        const double lo = neg ? t1 : t0, hi = neg ? t0 : t1;
        tmin = lo > tmin ? lo : tmin;
        tmax = hi < tmax ? hi : tmax;
        ok = !(tmax <= tmin);
    }
    {
        double t0 = (b[1] - o.y) * inv.y, t1 = (b[4] - o.y) * inv.y;
        const bool neg = inv.y < 0.0;
        const double lo = neg ? t1 : t0, hi = neg ? t0 : t1;
        tmin = lo > tmin ? lo : tmin;
        tmax = hi < tmax ? hi : tmax;
        ok = ok & !(tmax <= tmin);
    }
    {
        double t0 = (b[2] - o.z) * inv.z, t1 = (b[5] - o.z) * inv.z;
        const bool neg = inv.z < 0.0;
        const double lo = neg ? t1 : t0, hi = neg ? t0 : t1;
        tmin = lo > tmin ? lo : tmin;
        tmax = hi < tmax ? hi : tmax;
        ok = ok & !(tmax <= tmin);
    }
    return ok;
}

// Sphere::hit / MovingSphere::hit root selection (src/sphere.rs:44-81, 137-175). which = 0 near root, 1 far root.
template <bool X>
PT_DEV bool sphere_t(D3 center, double radius, D3 o, D3 d, double t_min, double t_max, double &t_out, int &which) {
    D3 oc = o - center;
    double a = M<X>::dot(d, d);
    double b = M<X>::dot(oc, d);
    double c, discriminant;
    if constexpr (X) {
        c = M<X>::dot(oc, oc) - radius * radius;
        discriminant = b * b - a * c;
    } else {
        c = __builtin_fma(-radius, radius, M<X>::dot(oc, oc));
        discriminant = __builtin_fma(-a, c, b * b);
    }
    if (discriminant > 0.0) {
        double root = M<X>::sqrt_(discriminant);
        if constexpr (X) {
            double temp = (-b - root) / a;
            if (temp < t_max && temp > t_min) {
                t_out = temp;
                which = 0;
                return true;
            }
            temp = (-b + root) / a;
            if (temp < t_max && temp > t_min) {
                t_out = temp;
                which = 1;
                return true;
            }
        } else {
            const double inv_a = fast_rcp(a);
            double temp = (-b - root) * inv_a;
            if (temp < t_max && temp > t_min) {
                t_out = temp;
                which = 0;
                return true;
            }
            temp = (-b + root) * inv_a;
            if (temp < t_max && temp > t_min) {
                t_out = temp;
                which = 1;
                return true;
            }
        }
    }
    return false;
}

// AxisAlignedRect::hit acceptance (src/rect.rs:97-105); axis 0 = X, 1 = Y, 2 = Z (63-85).
// `inv` = 1 / d per axis; only the fast mode uses it (exact mode divides, as the reference does).
template <bool X>
PT_DEV bool rect_t(double amin, double amax, double bmin, double bmax, double c, int axis, D3 o, D3 d, D3 inv,
                   double t_min, double t_max, double &t_out) {
    double t, a, b;
    if (axis == 0) {
        t = X ? (c - o.x) / d.x : (c - o.x) * inv.x;
        if (t < t_min || t > t_max) return false;
        a = M<X>::mad(t, d.y, o.y);
        b = M<X>::mad(t, d.z, o.z);
    } else if (axis == 1) {
        t = X ? (c - o.y) / d.y : (c - o.y) * inv.y;
        if (t < t_min || t > t_max) return false;
        a = M<X>::mad(t, d.x, o.x);
        b = M<X>::mad(t, d.z, o.z);
    } else {
        t = X ? (c - o.z) / d.z : (c - o.z) * inv.z;
        if (t < t_min || t > t_max) return false;
        a = M<X>::mad(t, d.x, o.x);
        b = M<X>::mad(t, d.y, o.y);
    }
    if (a < amin || a > amax || b < bmin || b > bmax) return false;
    t_out = t;
    return true;
}

// Rect parameters of face k of AxisAlignedBox (src/axis_aligned_box.rs:17-22): +Z, -Z, +Y, -Y, +X, -X.
PT_DEV void box_face(const double *pb, int k, double &amin, double &amax, double &bmin, double &bmax, double &c,
                     int &axis) {
    if (k < 2) {
        amin = pb[0]; amax = pb[3]; bmin = pb[1]; bmax = pb[4]; axis = 2;
        c = (k == 0) ? pb[5] : pb[2];
    } else if (k < 4) {
        amin = pb[0]; amax = pb[3]; bmin = pb[2]; bmax = pb[5]; axis = 1;
        c = (k == 2) ? pb[4] : pb[1];
    } else {
        amin = pb[1]; amax = pb[4]; bmin = pb[2]; bmax = pb[5]; axis = 0;
        c = (k == 4) ? pb[3] : pb[0];
    }
}
// AxisAlignedBox::hit = HitableList::hit over its six faces with a shrinking t_max (src/hitable.rs:51-62).
template <bool X>
PT_DEV bool box_t(const double *pb, D3 o, D3 d, D3 inv, double t_min, double t_max, double &t_out, int &face) {
    double closest = t_max;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double amin, amax, bmin, bmax, c, t;
        int axis;
        box_face(pb, k, amin, amax, bmin, bmax, c, axis);
        if (rect_t<X>(amin, amax, bmin, bmax, c, axis, o, d, inv, t_min, closest, t)) {
            closest = t;
            face = k;
            any = true;
        }
    }
    t_out = closest;
    return any;
}

// AxisAlignedBox::hit (src/axis_aligned_box.rs:17-40): six rectangle tests with a shrinking t_max find the nearer
// of the (at most two) faces the ray crosses inside [t_min, t_max]. The slab form finds the same pair directly:
// entry = latest near plane, exit = earliest far plane; the face is the plane that set it. Ties go to X, then Y,
// then Z, as the reference's face order (+Z -Z +Y -Y +X -X, later face wins an equal t) gives.
PT_DEV bool box_pred(const double *pb, D3 o, D3 inv, double t_min, double t_max, double &t,
                                         int &face) {
    const double x0 = (pb[0] - o.x) * inv.x, x1 = (pb[3] - o.x) * inv.x;
    const double y0 = (pb[1] - o.y) * inv.y, y1 = (pb[4] - o.y) * inv.y;
    const double z0 = (pb[2] - o.z) * inv.z, z1 = (pb[5] - o.z) * inv.z;
    const double nx = fmin(x0, x1), fx = fmax(x0, x1);
    const double ny = fmin(y0, y1), fy = fmax(y0, y1);
    const double nz = fmin(z0, z1), fz = fmax(z0, z1);
    const double tn = fmax(nx, fmax(ny, nz)), tf = fmin(fx, fmin(fy, fz));
    const bool entry = !(tn < t_min);
    t = entry ? tn : tf;
    const int axis_n = (nx == tn) ? 0 : ((ny == tn) ? 1 : 2);
    const int axis_f = (fx == tf) ? 0 : ((fy == tf) ? 1 : 2);
    const int axis = entry ? axis_n : axis_f;
    // Entering through the max plane means travelling in -axis; leaving through it, in +axis: the max plane is hit
    // when (travelling in -axis) == entry. The direction of travel is the sign bit of 1 / d (an infinity keeps the
    // sign of its zero). Faces are numbered +Z -Z +Y -Y +X -X.
    const uint32_t negative = ((uint32_t)__double2hiint(inv.x) >> 31) | (((uint32_t)__double2hiint(inv.y) >> 31) << 1) |
                              (((uint32_t)__double2hiint(inv.z) >> 31) << 2);
    const uint32_t min_plane = ((negative >> axis) & 1u) ^ (entry ? 1u : 0u);
    face = 4 - 2 * axis + (int)min_plane;
    // A rectangle accepts t == t_max (src/rect.rs:99), but a box is reached through the BVH node that holds it,
    // whose bounds are exactly the box, and AABB::hit rejects an interval that has shrunk to a point
    // (`tmax <= tmin`, src/math/aabb.rs:41): a box ENTERED exactly at the current closest distance is never
    // tested. That is what decides coplanar pairs such as a box standing on a floor seen from below.
    // (entry: t = tn and tn >= t_min holds already; exit: t = tf must not lie below t_min nor beyond t_max)
    const bool ok_entry = entry & (tn < t_max);
    const bool ok_exit = !entry & !(tf < t_min) & !(tf > t_max);
    return (tn <= tf) & (ok_entry | ok_exit);
}

template <bool X> PT_DEV D3 moving_center(const double *pd, const double *e, double time) { // src/sphere.rs:130-132
    return d3(pd[0], pd[1], pd[2]) + M<X>::div(time - pd[4], pd[5]) * d3(e[0], e[1], e[2]);
}

// One primitive op against a ray that is already in the op's frame; applies the op's own chain first.
// `inv` is 1 / d of the frame ray (fast mode only; recomputed here if the op's chain rotates the ray).
// `sub` is the near/far root for spheres and the face for boxes.
// The test itself, for a ray that is already in the primitive's own frame.
template <bool X>
PT_DEV bool prim_hit_framed(const SceneView &sv, const DevOp &op, D3 o, D3 d, D3 inv, double time, double t_min,
                            double t_max, double &t_out, int &sub) {
    const int kind = op.kind & 0xff;
    sub = 0;
    switch (kind) {
    case OP_SPHERE:
        return sphere_t<X>(d3(op.d[0], op.d[1], op.d[2]), op.d[3], o, d, t_min, t_max, t_out, sub);
    case OP_MSPHERE:
        return sphere_t<X>(moving_center<X>(op.d, sv.leaves[op.leaf].e, time), op.d[3], o, d, t_min, t_max, t_out, sub);
    case OP_RECT:
        return rect_t<X>(op.d[0], op.d[1], op.d[2], op.d[3], op.d[4], (op.kind >> 8) & 3, o, d, inv, t_min, t_max,
                         t_out);
    case OP_BOXPRIM:
        if constexpr (X) return box_t<X>(op.d, o, d, inv, t_min, t_max, t_out, sub);
        else return box_pred(op.d, o, inv, t_min, t_max, t_out, sub); // fast mode: the slab form of the same test
    default:
        return false;
    }
}
template <bool X>
PT_DEV bool prim_hit(const SceneView &sv, const DevOp &op, D3 o, D3 d, D3 inv, double time, double t_min,
                     double t_max, double &t_out, int &sub) {
    const int kind = op.kind & 0xff;
    const int cc = (op.kind >> 16) & 0xff;
    if (cc) {
        apply_chain<X>(sv.xforms, op.chain_start, cc, o, d);
        if constexpr (!X) inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
    }
    sub = 0;
    switch (kind) {
    case OP_SPHERE:
        return sphere_t<X>(d3(op.d[0], op.d[1], op.d[2]), op.d[3], o, d, t_min, t_max, t_out, sub);
    case OP_MSPHERE:
        return sphere_t<X>(moving_center<X>(op.d, sv.leaves[op.leaf].e, time), op.d[3], o, d, t_min, t_max, t_out, sub);
    case OP_RECT:
        return rect_t<X>(op.d[0], op.d[1], op.d[2], op.d[3], op.d[4], (op.kind >> 8) & 3, o, d, inv, t_min, t_max,
                         t_out);
    case OP_BOXPRIM:
        if constexpr (X) return box_t<X>(op.d, o, d, inv, t_min, t_max, t_out, sub);
        else return box_pred(op.d, o, inv, t_min, t_max, t_out, sub); // fast mode: the slab form of the same test
    default:
        return false;
    }
}

// Closest hit over a medium's boundary when that is more than one primitive: the threaded walk of its sub-stream
// [first, end) of BOX and primitive ops (a HitableList is its members in order, a BvhNode its box and subtrees:
// src/hitable.rs:51-62, src/bvh.rs:74-99), the ray already in the medium's frame. Wrappers above a tree are the chain
// of its BOX ops and the front of its members' chains.
template <bool X>
PT_DEV bool boundary_hit(const SceneView &sv, int32_t first, int32_t end, D3 o, D3 d, D3 inv, double time, double t_min,
                         double t_max, double &t_out) {
    double closest = t_max;
    bool any = false;
    int32_t cur = first;
    while (cur < end) {
        const DevOp *bop = &sv.ops[cur];
        if ((bop->kind & 0xff) == OP_BOX) {
            const int cc = (bop->kind >> 16) & 0xff;
            bool in;
            if (cc) { // the box of a tree under wrappers: tested with the ray those wrappers hand down
                D3 bo = o, bd = d;
                apply_chain<X>(sv.xforms, bop->chain_start, cc, bo, bd);
                in = aabb_hit(bop->d, bo, d3(M<X>::rcp(bd.x), M<X>::rcp(bd.y), M<X>::rcp(bd.z)), t_min, closest);
            } else {
                in = aabb_hit(bop->d, o, inv, t_min, closest);
            }
            cur = in ? cur + 1 : bop->skip;
            continue;
        }
        double t;
        int sub;
        if (prim_hit<X>(sv, *bop, o, d, inv, time, t_min, closest, t, sub)) {
            closest = t;
            any = true;
        }
        cur += 1;
    }
    t_out = closest;
    return any;
}

// ConstantMedium::hit (src/volume.rs:24-65). The ray is in the medium's frame once the op's chain is applied.
// GENERAL: the boundary may be a sub-stream of several ops (op.d[1] of them); otherwise it is one primitive op.
// UNIFORM: every calling lane passes the SAME op (the wave-uniform walks): the medium's chain steps, its boundary op and
// its node id are then read once for the wave through the scalar cache instead of once per lane by vector loads that
// sit, one behind the other, in front of every lane's boundary test.
template <bool X, bool GENERAL = false, bool UNIFORM = false>
PT_DEV bool medium_hit(const SceneView &sv, const DevOp &op, D3 o, D3 d, D3 inv, double time, double t_min,
                       double t_max, uint64_t pkey, uint32_t depth, double &t_out) {
    static_assert(!(GENERAL && UNIFORM), "a boundary of several ops is walked per lane");
    const int cc = (op.kind >> 16) & 0xff;
    if (cc) {
        if constexpr (UNIFORM) {
            const int start = __builtin_amdgcn_readfirstlane(op.chain_start), steps = __builtin_amdgcn_readfirstlane(cc);
            for (int k = 0; k < steps; ++k) xform_ray<X>(load_uniform(sv.xforms + start + k), o, d);
        } else {
            apply_chain<X>(sv.xforms, op.chain_start, cc, o, d);
        }
        if constexpr (!X) inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
    }
    DevOp bop_uniform;
    if constexpr (UNIFORM) bop_uniform = load_uniform(sv.ops + __builtin_amdgcn_readfirstlane(op.skip));
    const DevOp &bop = UNIFORM ? bop_uniform : sv.ops[op.skip];
    double t1, t2;
    int sub;
    const int32_t n_boundary = GENERAL ? (int32_t)op.d[1] : 1;
    if (GENERAL && n_boundary > 1) {
        if (!boundary_hit<X>(sv, op.skip, op.skip + n_boundary, o, d, inv, time, -kF64Max, kF64Max, t1)) return false;
        if (!boundary_hit<X>(sv, op.skip, op.skip + n_boundary, o, d, inv, time, t1 + 0.0001, kF64Max, t2)) return false;
    } else if (!X && (bop.kind & 0xff) == OP_SPHERE && ((bop.kind >> 16) & 0xff) == 0) {
        // A sphere boundary without a chain of its own: both calls of Sphere::hit below evaluate the same two roots
        // (src/sphere.rs:44-81). The first, over (-MAX, MAX), takes the near root r0 (whenever it is finite); the
        // second, over (r0 + 0.0001, MAX), can then only take the far root. Same expressions as sphere_t<false>, once.
        const D3 oc = o - d3(bop.d[0], bop.d[1], bop.d[2]);
        const double a = M<X>::dot(d, d), b = M<X>::dot(oc, d);
        const double c = __builtin_fma(-bop.d[3], bop.d[3], M<X>::dot(oc, oc));
        const double discriminant = __builtin_fma(-a, c, b * b);
        if (!(discriminant > 0.0)) return false;
        const double root = M<X>::sqrt_(discriminant), inv_a = fast_rcp(a);
        const double r0 = (-b - root) * inv_a, r1 = (-b + root) * inv_a;
        if (r0 < kF64Max && r0 > -kF64Max) {
            t1 = r0;
            if (!(r1 < kF64Max && r1 > t1 + 0.0001)) return false;
            t2 = r1;
        } else {
            return false; // r0 not finite: the first call would take r1 and the second find nothing beyond it
        }
    } else {
        if (!prim_hit<X>(sv, bop, o, d, inv, time, -kF64Max, kF64Max, t1, sub)) return false;
        if (!prim_hit<X>(sv, bop, o, d, inv, time, t1 + 0.0001, kF64Max, t2, sub)) return false;
    }
    t1 = fmax(t1, t_min);
    t2 = fmin(t2, t_max);
    if (t1 >= t2) return false;
    double ray_length = M<X>::length(d);
    double distance_inside_boundary = (t2 - t1) * ray_length;
    int32_t medium_node;
    if constexpr (UNIFORM) medium_node = load_uniform(&sv.leaves[__builtin_amdgcn_readfirstlane(op.leaf)].prim[0]);
    else medium_node = sv.leaves[op.leaf].prim[0];
    double xi = medium_draw(pkey, depth, medium_node);
    double hit_distance = -M<X>::rcp(op.d[0]) * (X ? log(xi) : fast_log_unit(xi)); // (xi = 0: +inf, no scatter)
    if (hit_distance < distance_inside_boundary) {
        t_out = t1 + M<X>::div(hit_distance, ray_length);
        return true;
    }
    return false;
}

// ---------------------------------------------------------- finalisation --
struct HitRec { // src/hitable.rs:5-12
    double t, u, v;
    D3 p, n;
    int32_t mat;
    int32_t prim;
};

PT_DEV void get_sphere_uv(D3 p, double &u, double &v) { // src/sphere.rs:8-14
    double phi = atan2(p.z, p.x);
    double theta = asin(p.y);
    u = 1.0 - (phi + kPi) / (kPi * 2.0);
    v = (theta + kFracPi2) / kPi;
}

// Rebuilds the HitRecord of the closest hit from (world ray, t, leaf, sub): the leaf's full chain takes the
// ray to the primitive's space with the same sequence of operations the traversal used, the record is built
// there as the primitive's hit() does, then carried back out through the wrappers in reverse.
// TEX: what the scene's textures need (see texture_value). The surface (u, v) feed image textures and nothing else
// (a sphere's take an atan2 and an asin): TEX == 2 computes them for the leaves whose texture reads them, TEX == 3
// (the hit records pt_intersect returns) for every leaf, lower levels never.
template <bool X, int TEX>
PT_DEV HitRec finalize_hit(const SceneView &sv, const DevLeaf &lf, D3 o, D3 d, double time, double t, int sub) {
    const bool want_uv = TEX == 3 || (TEX == 2 && (lf.tex_flags & kLeafTexUv) != 0);
    apply_chain<X>(sv.xforms, lf.full_start, lf.full_count, o, d);
    HitRec r;
    r.t = t;
    r.u = 0.0;
    r.v = 0.0;
    r.mat = lf.material;
    r.prim = lf.prim[0];
    const double *g = lf.g;
    D3 point = M<X>::point_at(o, d, t);
    r.p = point;
    if (lf.kind == OP_SPHERE) {
        D3 c = d3(g[0], g[1], g[2]);
        if (want_uv) {
            if (sub == 0) get_sphere_uv(M<X>::divs(c - point, g[3]), r.u, r.v); // sphere.rs:56
            else get_sphere_uv(c - point, r.u, r.v);                               // sphere.rs:69
        }
        r.n = M<X>::divs(point - c, g[3]);
    } else if (lf.kind == OP_MSPHERE) {
        D3 c = moving_center<X>(g, lf.e, time);
        if (want_uv) get_sphere_uv(c - point, r.u, r.v); // sphere.rs:150,163
        r.n = M<X>::divs(point - c, g[3]);
    } else if (lf.kind == OP_RECT || lf.kind == OP_BOXPRIM) {
        double amin, amax, bmin, bmax, c;
        int axis;
        if (lf.kind == OP_RECT) {
            amin = g[0]; amax = g[1]; bmin = g[2]; bmax = g[3]; c = g[4];
            axis = lf.axis;
        } else {
            box_face(g, sub, amin, amax, bmin, bmax, c, axis);
            r.prim = lf.prim[sub];
        }
        if (want_uv) {
            double a, b;
            if (axis == 0) { a = M<X>::mad(t, d.y, o.y); b = M<X>::mad(t, d.z, o.z); }
            else if (axis == 1) { a = M<X>::mad(t, d.x, o.x); b = M<X>::mad(t, d.z, o.z); }
            else { a = M<X>::mad(t, d.x, o.x); b = M<X>::mad(t, d.y, o.y); }
            r.u = M<X>::div(a - amin, amax - amin); // rect.rs:46-47,108-109
            r.v = M<X>::div(b - bmin, bmax - bmin);
        }
        r.n = axis == 0 ? d3(1.0, 0.0, 0.0) : (axis == 1 ? d3(0.0, 1.0, 0.0) : d3(0.0, 0.0, 1.0));
        if (lf.kind == OP_BOXPRIM) r.n = r.n * ((sub & 1) ? -1.0 : 1.0); // faces 1, 3, 5 are wrapped in FlipNormals
    } else { // OP_MEDIUM, volume.rs:51-59
        r.n = d3(1.0, 0.0, 0.0);
    }
    for (int k = lf.full_count - 1; k >= 0; --k) unxform_hit<X>(sv.xforms[lf.full_start + k], r.p, r.n);
    return r;
}

// -------------------------------------------------------------- textures --
PT_DEV double perlin_noise(const pt_perlin *t, D3 p) { // src/math/noise.rs:26-44, 89-113
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    // Rust's `as i32` saturates and maps NaN to 0
    int32_t i = (fx != fx) ? 0 : (fx >= 2147483647.0 ? 2147483647 : (fx <= -2147483648.0 ? (-2147483647 - 1) : (int32_t)fx));
    int32_t j = (fy != fy) ? 0 : (fy >= 2147483647.0 ? 2147483647 : (fy <= -2147483648.0 ? (-2147483647 - 1) : (int32_t)fy));
    int32_t k = (fz != fz) ? 0 : (fz >= 2147483647.0 ? 2147483647 : (fz <= -2147483648.0 ? (-2147483647 - 1) : (int32_t)fz));
    double u = p.x - (double)i, v = p.y - (double)j, w = p.z - (double)k;
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                int32_t ix = t->perm_x[((uint32_t)i + (uint32_t)di) & 255u] ^ t->perm_y[((uint32_t)j + (uint32_t)dj) & 255u] ^
                             t->perm_z[((uint32_t)k + (uint32_t)dk) & 255u];
                D3 c = d3(t->ranvec[ix][0], t->ranvec[ix][1], t->ranvec[ix][2]);
                double fi = di, fj = dj, fk = dk;
                D3 weight_v = d3(u - fi, v - fj, w - fk);
                accum += (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                         (fk * ww + (1.0 - fk) * (1.0 - ww)) * dot(c, weight_v);
            }
    return accum;
}
PT_DEV double perlin_turb(const pt_perlin *t, D3 p, int depth) { // src/math/noise.rs:115-126
    double accum = 0.0;
    D3 temp_p = p;
    double weight = 1.0;
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(t, temp_p);
        weight *= 0.5;
        temp_p = temp_p * 2.0;
    }
    return fabs(accum);
}
// perlin_turb(p, 7) for the lanes of a wave that `want` it, evaluated by the whole wave: lane 7 s + i computes octave i
// (the noise at 2^i p, weighted 2^-i; both scalings are exact) for the s-th wanting lane of a batch of nine, and each
// wanting lane then adds its seven terms in the order of the serial loop, so the value has the bits perlin_turb gives.
// An octave is two dependent table reads behind some forty instructions; one lane walking seven of them in a row,
// with the other lanes of the wave idle, is what a noise-textured hit used to cost. `tab` may live in LDS. `ps` (64
// points) and `rs` (nine rows of eight) are scratch of the calling wave alone. Must be called by all 64 lanes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_turb7(const pt_perlin *tab, double (*ps)[3], double (*rs)[8], bool want, D3 p) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = __ballot(want);
    const uint32_t total = (uint32_t)__popcll(mask);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (want) {
        ps[rank][0] = p.x;
        ps[rank][1] = p.y;
        ps[rank][2] = p.z;
    }
    wave_lds_fence();
    const uint32_t slot = lane / 7u, oct = lane - slot * 7u;
    double result = 0.0;
    for (uint32_t base = 0; base < total; base += 9u) { // wave-uniform trip count
        const uint32_t k = base + slot;
        if (lane < 63u && k < total) {
            const double s = (double)(1u << oct), w = 1.0 / s;
            rs[slot][oct] = w * perlin_noise(tab, d3(ps[k][0] * s, ps[k][1] * s, ps[k][2] * s));
        }
        wave_lds_fence();
        if (want && rank >= base && rank < base + 9u) {
            double accum = 0.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) accum += rs[rank - base][i];
            result = fabs(accum);
        }
        wave_lds_fence();
    }
    return result;
}
PT_DEV double clamp_ref(double a, double lo, double hi) { // src/math/mod.rs:22-26
    double b = a > hi ? hi : a;
    b = b < lo ? lo : b;
    return b;
}
// Texture::value (src/texture.rs:24-28, 47-56, 70-75, 101-114). Checker children are resolved iteratively.
// TEX is the scene's texture level, a property of the uploaded scene that the kernels are specialised on:
//   0  every material's texture is a constant colour (already copied into the leaf records: nothing to evaluate);
//   1  checker textures over constant colours occur (three sines of the hit point);
//   2  noise or image textures occur too (Perlin turbulence; the surface u, v of the hit).
// The Perlin evaluation alone is several thousand instructions and some forty registers; a kernel for a scene without
// it must not carry it.
// CheckerTexture::value picks `odd` when sin(10 x) sin(10 y) sin(10 z) < 0 (src/texture.rs:49-54). Only the sign of
// the product is used, and the sign of sin(y) is the parity of floor(y / pi): fast mode takes the three parities
// instead of three general-argument sines (a few hundred instructions each). A factor that is exactly zero makes the
// product zero, i.e. not negative.
template <bool X> PT_DEV bool checker_is_odd(D3 p) {
    if constexpr (X) {
        double sines = sin(10.0 * p.x) * sin(10.0 * p.y) * sin(10.0 * p.z);
        return sines < 0.0;
    } else {
        const double ax = 10.0 * p.x, ay = 10.0 * p.y, az = 10.0 * p.z;
        const double kx = floor(ax * kFrac1Pi), ky = floor(ay * kFrac1Pi), kz = floor(az * kFrac1Pi);
        // parity of k: k - 2 floor(k / 2) is 0 or 1, exactly, for every finite k
        const double odd = (kx - 2.0 * floor(0.5 * kx)) + (ky - 2.0 * floor(0.5 * ky)) + (kz - 2.0 * floor(0.5 * kz));
        const bool any_zero = (ax == 0.0) | (ay == 0.0) | (az == 0.0);
        return (odd == 1.0 || odd == 3.0) && !any_zero;
    }
}
PT_DEV pt_texture load_texture_uniform(const pt_texture *p) {
    const pt_texture __attribute__((address_space(4))) *cp = (const pt_texture __attribute__((address_space(4))) *)p;
    pt_texture t;
    t.kind = cp->kind;
    t.even = cp->even;
    t.odd = cp->odd;
    t.width = cp->width;
    t.height = cp->height;
    t.reserved = 0;
    t.texel_offset = cp->texel_offset;
    t.colour[0] = cp->colour[0];
    t.colour[1] = cp->colour[1];
    t.colour[2] = cp->colour[2];
    t.scale = cp->scale;
    return t;
}
PT_DEV DevLight load_light_uniform(const DevLight *p) {
    const DevLight __attribute__((address_space(4))) *cp = (const DevLight __attribute__((address_space(4))) *)p;
    DevLight l;
    l.kind = cp->kind;
    l.axis = cp->axis;
#pragma unroll
    for (int k = 0; k < 5; ++k) l.d[k] = cp->d[k];
    return l;
}
// `turbulence`: perlin_turb(p, 7) already evaluated for this hit (by the whole wave, see wave_turb7), or null.
template <bool X, int TEX>
PT_DEV D3 texture_value(const SceneView &sv, int tex, double u, double v, D3 p, const double *turbulence = nullptr) {
    if constexpr (TEX == 0) return d3(0.0, 0.0, 0.0);
    // Texture records come through the scalar cache: the calling lanes hold a few distinct indices (one checker and
    // its two colours, a noise and an image texture), so the wave reads one index at a time at a uniform address and
    // the lanes that asked for it keep the record. Read at each lane's own index, a checker is three dependent vector
    // loads in front of the vertex's sampling.
    auto fetch = [&](int index, bool want, pt_texture &rec) {
        bool pending = want;
        for (;;) {
            const unsigned long long left = __ballot(pending);
            if (!left) break;
            const int k = __builtin_amdgcn_readlane(index, __ffsll((long long)left) - 1); // the first unserved lane's index
            const pt_texture rk = load_texture_uniform(sv.textures + k);
            if (pending && index == k) {
                rec = rk;
                pending = false;
            }
        }
    };
    pt_texture rec;
    fetch(tex, true, rec);
    if (__ballot(rec.kind == PT_T_CHECKER)) { // the cell is the same at every level of nested checkers
        const bool odd = checker_is_odd<X>(p);
#pragma unroll 1
        for (int guard = 0; guard < 16 && __ballot(rec.kind == PT_T_CHECKER); ++guard)
            fetch(odd ? rec.odd : rec.even, rec.kind == PT_T_CHECKER, rec);
    }
    const pt_texture *t = &rec;
    if (t->kind == PT_T_CONSTANT) return d3(t->colour[0], t->colour[1], t->colour[2]);
    if (TEX == 2 && t->kind == PT_T_NOISE) {
        double noise = t->scale * p.z + 10.0 * (turbulence ? *turbulence : perlin_turb(sv.perlin, p, 7));
        double g = 1.0 * 0.5 * (1.0 + sin(noise));
        return d3(g, g, g);
    }
    if (TEX == 2 && t->kind == PT_T_IMAGE) {
        double wf = (double)t->width, hf = (double)t->height;
        double ci = clamp_ref(u * wf, 0.0, wf - 1.0);
        double cj = clamp_ref(v * hf - 0.001, 0.0, hf - 1.0);
        uint64_t ii = (ci > 0.0) ? (uint64_t)ci : 0; // `as usize`: NaN -> 0
        uint64_t jj = (cj > 0.0) ? (uint64_t)cj : 0;
        const uint8_t *px = sv.texels + t->texel_offset + 3 * ii + 3 * (uint64_t)t->width * jj;
        return d3((double)px[0] / 255.0, (double)px[1] / 255.0, (double)px[2] / 255.0);
    }
    return d3(0.0, 0.0, 0.0);
}
// True when a texture's value depends on the surface (u, v) of the hit.
PT_DEV bool texture_is_constant(const SceneView &sv, int tex) { return sv.textures[tex].kind == PT_T_CONSTANT; }

// -------------------------------------------------------------- samplers --
struct Onb { // src/onb.rs:18-37
    D3 u, v, w;
};
// a x b with each component as one multiply and one FMA (fast mode)
PT_DEV D3 cross_fma(D3 a, D3 b) {
    return D3{__builtin_fma(a.y, b.z, -(a.z * b.y)), -__builtin_fma(a.x, b.z, -(a.z * b.x)), __builtin_fma(a.x, b.y, -(a.y * b.x))};
}
// Fast mode's basis about n, with 1 / |n| from the caller: the same basis as the exact form's, the products by the
// zeros and ones of `a` written out: w x (0,1,0) = (-w.z, 0, w.x), w x (1,0,0) = (0, w.z, -w.y)
PT_DEV Onb onb_from_w_scaled(D3 n, double inv_len) {
    const D3 w = n * inv_len;
    const bool big = fabs(w.x) > 0.9;
    const double p = big ? w.x : w.y;
    const double r = weight_rsqrt(__builtin_fma(w.z, w.z, p * p));
    const D3 v = big ? d3(-w.z * r, 0.0, w.x * r) : d3(0.0, w.z * r, -w.y * r);
    const D3 u = cross_fma(w, v);
    return Onb{u, v, w};
}
template <bool X> PT_DEV Onb onb_from_w(D3 n) {
    if constexpr (X) {
        D3 w = M<X>::unit_vector(n);
        D3 a = fabs(w.x) > 0.9 ? d3(0.0, 1.0, 0.0) : d3(1.0, 0.0, 0.0);
        D3 v = M<X>::unit_vector(cross(w, a));
        D3 u = cross(w, v);
        return Onb{u, v, w};
    } else {
        return onb_from_w_scaled(n, weight_rsqrt(M<X>::squared_length(n)));
    }
}
PT_DEV D3 onb_local(const Onb &b, D3 a) { return b.u * a.x + b.v * a.y + b.w * a.z; }
PT_DEV D3 onb_local_fma(const Onb &b, D3 a) { // the same sums with fused multiply-adds (fast mode)
    return D3{__builtin_fma(b.w.x, a.z, __builtin_fma(b.v.x, a.y, b.u.x * a.x)),
              __builtin_fma(b.w.y, a.z, __builtin_fma(b.v.y, a.y, b.u.y * a.x)),
              __builtin_fma(b.w.z, a.z, __builtin_fma(b.v.z, a.y, b.u.z * a.x))};
}

// The direction samplers below take the raw bits u1, u2 of the NEXT two draws of `g` (Rng::peek_u32), which the
// caller computes once for every lane of the wave, whichever sampler a lane ends up in: the integer hashing then runs
// once with all lanes active instead of once per branch with the lanes of that branch. A sampler that uses the draws
// consumes them (g.n += 2); one that draws nothing (the default Hitable::random) leaves the stream where it is. The
// values are those of two g.next() calls.
PT_DEV double bits_to_f64(uint32_t u) { return (double)u * (1.0 / 4294967296.0); } // Rng::next
template <bool X> PT_DEV D3 random_cosine_direction(uint32_t u1, uint32_t u2, Rng &g) { // src/material.rs:10-19
    g.n += 2;
    const double r1 = bits_to_f64(u1);
    const double r2 = bits_to_f64(u2);
    double z = W<X>::sqrt_(1.0 - r2);
    double sn, cs;
    M<X>::sincos_2pi(r1, sn, cs);
    double r2_sqrt = W<X>::sqrt_(r2);
    double x = cs * r2_sqrt;
    double y = sn * r2_sqrt;
    return d3(x, y, z);
}
PT_DEV D3 random_in_unit_sphere(Rng &g) { // src/material.rs:21-31
    D3 p;
    for (;;) {
        double a = g.next();
        double b = g.next();
        double c = g.next();
        p = 2.0 * d3(a, b, c) - d3(1.0, 1.0, 1.0);
        if (squared_length(p) < 1.0) break;
    }
    return p;
}
// random_in_unit_sphere for the lanes of a wave that `want` it, evaluated by the whole wave. A lane's draws are a pure
// function of (key, count), so the candidates of the rejection loop can be made side by side: with m wanting lanes every
// one of them gets C = 8, 4, 2 or 1 helper lanes (C m <= 64), helper k makes the candidate of iteration k from draws
// 3 k + 1 ... 3 k + 3 behind the lane's count, and the lane takes the first accepted one, which is the one the serial
// loop stops at: same draws, same arithmetic, same point, same count afterwards (n + 3 per iteration tried). Further
// rounds of C iterations run only while some lane has found none (C = 8: one case in 400). The serial loop costs the
// wave as many iterations as its unluckiest lane needs, four to five with ten lanes in it, each three draws long.
// `src_tab`: 64 words of LDS of the calling wave alone. Must be called by all 64 lanes; p_out and n_out are written for
// the wanting lanes only.
__device__ __forceinline__ void wave_unit_sphere(bool want, uint64_t key, uint32_t n, uint32_t *src_tab, D3 &p_out,
                                                 uint32_t &n_out) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long wmask = __ballot(want);
    const uint32_t m = (uint32_t)__popcll(wmask);
    const uint32_t rank = want ? (uint32_t)__popcll(wmask & ((1ull << lane) - 1ull)) : 0u;
    if (want) src_tab[rank] = lane;
    wave_lds_fence();
    const uint32_t shift = m <= 8u ? 3u : (m <= 16u ? 2u : (m <= 32u ? 1u : 0u));
    const uint32_t per = 1u << shift;
    const uint32_t w = lane >> shift, k0 = lane & (per - 1u);
    const bool helper = w < m;
    const int src = (int)src_tab[helper ? w : 0u];
    const uint64_t hkey = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(key >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)key, src);
    const uint32_t hn = (uint32_t)__shfl((int)n, src);
    bool open = want;
    for (uint32_t base = 0;; base += per) {
        Rng g{hkey, hn + 3u * (base + k0)};
        const double a = g.next();
        const double b = g.next();
        const double c = g.next();
        const D3 p = 2.0 * d3(a, b, c) - d3(1.0, 1.0, 1.0);
        const bool accepted = helper && squared_length(p) < 1.0;
        const unsigned long long amask = __ballot(accepted);
        const uint32_t mine = (uint32_t)(amask >> (rank << shift)) & ((1u << per) - 1u); // (per <= 8)
        const uint32_t first = mine ? (uint32_t)__builtin_ctz(mine) : 0u;
        const int from = (int)((rank << shift) + first);
        const double px = __shfl(p.x, from), py = __shfl(p.y, from), pz = __shfl(p.z, from);
        if (open && mine) {
            p_out = d3(px, py, pz);
            n_out = n + 3u * (base + first + 1u);
            open = false;
        }
        if (__ballot(open) == 0ull) break;
    }
    wave_lds_fence(); // the table may be written again by the next call
}
PT_DEV D3 random_in_unit_disk(Rng &g) { // src/camera.rs:221-228
    double a = g.next();
    double b = g.next();
    D3 nv = d3(a, b, 0.0) * 2.0 - d3(1.0, 1.0, 0.0);
    while (dot(nv, nv) >= 1.0) {
        a = g.next();
        b = g.next();
        nv = d3(a, b, 0.0) * 2.0 - d3(1.0, 1.0, 0.0);
    }
    return nv;
}
template <bool X> PT_DEV D3 random_to_sphere(uint32_t u1, uint32_t u2, double radius, double distance_squared) { // src/sphere.rs:16-25
    const double r1 = bits_to_f64(u1);
    const double r2 = bits_to_f64(u2);
    double z = 1.0 + r2 * (W<X>::sqrt_(1.0 - W<X>::div(radius * radius, distance_squared)) - 1.0);
    double sn, cs;
    M<X>::sincos_2pi(r1, sn, cs);
    double c = W<X>::sqrt_(1.0 - z * z);
    double x = cs * c;
    double y = sn * c;
    return d3(x, y, z);
}

// ---------------------------------------------------------------- lights --
// Hitable::pdf_value of one light shape (src/sphere.rs:87-95, src/rect.rs:124-133, default src/hitable.rs:30-32).
// `len2` = |direction|^2 and `inv_len` = 1 / |direction| come from the caller, which needs them anyway (fast mode; exact
// mode recomputes them the way the reference does).
template <bool X> PT_DEV double light_pdf_value(const DevLight &l, D3 origin, D3 direction, double len2, double inv_len) {
    if (l.kind == LIGHT_SPHERE) {
        D3 c = d3(l.d[0], l.d[1], l.d[2]);
        bool hit;
        double dist2 = 0.0;
        if constexpr (X) {
            double t;
            int which;
            hit = sphere_t<X>(c, l.d[3], origin, direction, 0.001, kF64Max, t, which);
        } else {
            // Sphere::hit accepts a root in (0.001, MAX); the larger root passes whenever the smaller does, so the ray
            // hits iff disc > 0 and (-b + sqrt(disc)) / a > 0.001, i.e. sqrt(disc) > y with y = 0.001 a + b: true for
            // y < 0, else disc > y^2. No square root, no quotient.
            const D3 oc = origin - c;
            const double b = M<X>::dot(oc, direction);
            dist2 = M<X>::dot(oc, oc); // = |c - origin|^2, bit for bit
            const double cc = __builtin_fma(-l.d[3], l.d[3], dist2);
            const double disc = __builtin_fma(-len2, cc, b * b);
            const double y = __builtin_fma(0.001, len2, b);
            hit = (disc > 0.0) & ((y < 0.0) | (disc > y * y));
        }
        if (hit) {
            if constexpr (X) dist2 = M<X>::squared_length(c - origin);
            double cos_theta_max = W<X>::sqrt_(1.0 - W<X>::div(l.d[3] * l.d[3], dist2));
            double solid_angle = 2.0 * kPi * (1.0 - cos_theta_max);
            return W<X>::rcp(solid_angle);
        }
        return 0.0;
    }
    if (l.kind == LIGHT_RECT) {
        double t;
        D3 inv = d3(0.0, 0.0, 0.0);
        const double dk = l.axis == 0 ? direction.x : (l.axis == 1 ? direction.y : direction.z);
        if constexpr (!X) {
            const double r = weight_rcp(dk); // t only enters the pdf value
            inv = d3(r, r, r);
        }
        if (rect_t<X>(l.d[0], l.d[1], l.d[2], l.d[3], l.d[4], l.axis, origin, direction, inv, 0.001, kF64Max, t)) {
            double area = (l.d[1] - l.d[0]) * (l.d[3] - l.d[2]);
            if constexpr (X) {
                double v_sq_length = M<X>::squared_length(direction);
                double distance_sq = t * t * v_sq_length;
                D3 normal = l.axis == 0 ? d3(1.0, 0.0, 0.0) : (l.axis == 1 ? d3(0.0, 1.0, 0.0) : d3(0.0, 0.0, 1.0));
                double cosine = fabs(dot(direction, normal) / sqrt(v_sq_length));
                return distance_sq / (cosine * area);
            } else {
                const double distance_sq = t * t * len2;
                const double cosine = fabs(dk * inv_len);
                return distance_sq * weight_rcp(cosine * area);
            }
        }
        return 0.0;
    }
    return 0.0;
}
// Hitable::random of one light shape (src/sphere.rs:96-101, src/rect.rs:135-144, default src/hitable.rs:33-35).
template <bool X> PT_DEV D3 light_random(const DevLight &l, uint32_t u1, uint32_t u2, Rng &g, D3 origin) {
    if (l.kind == LIGHT_SPHERE) {
        g.n += 2;
        D3 direction = d3(l.d[0], l.d[1], l.d[2]) - origin;
        double distance_sq = M<X>::squared_length(direction);
        Onb uvw = onb_from_w<X>(direction);
        return onb_local(uvw, random_to_sphere<X>(u1, u2, l.d[3], distance_sq));
    }
    if (l.kind == LIGHT_RECT) {
        g.n += 2;
        double random_a = l.d[0] + bits_to_f64(u1) * (l.d[1] - l.d[0]);
        double random_b = l.d[2] + bits_to_f64(u2) * (l.d[3] - l.d[2]);
        D3 p = l.axis == 0 ? d3(l.d[4], random_a, random_b)
                           : (l.axis == 1 ? d3(random_a, l.d[4], random_b) : d3(random_a, random_b, l.d[4]));
        return p - origin;
    }
    return d3(1.0, 0.0, 0.0);
}
// HitableList::pdf_value over the light list (src/hitable.rs:66-73).
template <bool X> PT_DEV double lights_pdf_value(const SceneView &sv, D3 origin, D3 direction, double len2, double inv_len) {
    double weight = 1.0 / (double)sv.n_lights;
    double sum = 0.0;
    for (int k = 0; k < sv.n_lights; ++k) // (k is the same in every lane: the record comes through the scalar cache)
        sum += weight * light_pdf_value<X>(load_light_uniform(sv.lights + k), origin, direction, len2, inv_len);
    return sum;
}

// ------------------------------------------------------------- materials --
template <bool X> PT_DEV D3 reflect(D3 v, D3 n) { return v - (2.0 * M<X>::dot(v, n) * n); } // src/material.rs:50-52
template <bool X> PT_DEV bool refract(D3 v, D3 n, double ni_over_nt, D3 &refracted) { // src/material.rs:54-64
    D3 uv = M<X>::unit_vector(v);
    double dt = M<X>::dot(uv, n);
    double discriminant = 1.0 - ni_over_nt * ni_over_nt * (1.0 - dt * dt);
    if (discriminant > 0.0) {
        refracted = ni_over_nt * (uv - n * dt) - n * M<X>::sqrt_(discriminant);
        return true;
    }
    return false;
}
template <bool X> PT_DEV double schlick(double cosine, double ref_idx) { // src/material.rs:66-70
    double r0 = M<X>::div(1.0 - ref_idx, 1.0 + ref_idx);
    r0 = r0 * r0;
    const double m = 1.0 - cosine;
    if constexpr (X) {
        // (1 - cosine).powf(5.0), src/material.rs:69, as the CORRECTLY ROUNDED fifth power: m^2, m^4 and m^5 as
        // double-double products (each error term exact by FMA; about 2^-100 relative in all), rounded once at the end.
        // Checked on the host against a binary128 evaluation: equal on all of 2 x 10^7 random arguments in [0, 2);
        // glibc's pow, which the CPU restatement calls, is one ulp off that value on 0.085 % of them (it is within
        // 0.52 ulp, not correctly rounded), so the two agree as far as two libms agree. A tenth of a general pow's
        // instructions. Fast mode's three plain multiplies are within 1.5 ulp, which is not the same number.
        const double h2 = m * m, l2 = __builtin_fma(m, m, -h2);                       // m^2 = h2 + l2
        const double p4 = h2 * h2;
        const double e4 = __builtin_fma(2.0 * h2, l2, __builtin_fma(h2, h2, -p4));    // m^4 = p4 + e4 (+ l2^2, below 2^-104)
        const double h4 = p4 + e4, l4 = e4 - (h4 - p4);
        const double p5 = h4 * m;
        const double e5 = __builtin_fma(l4, m, __builtin_fma(h4, m, -p5));            // m^5 = p5 + e5
        return r0 + (1.0 - r0) * (p5 + e5);
    } else {
        // three multiplies: within 1.5 ulp of the power; the value only decides reflection against a random draw
        const double m2 = m * m;
        return r0 + (1.0 - r0) * (m2 * m2 * m);
    }
}

// The material's texture at a hit: the colour inlined in the leaf record, or the general evaluator.
template <bool X, int TEX>
PT_DEV D3 leaf_texture(const SceneView &sv, const DevLeaf &lf, const HitRec &rec, const double *turbulence = nullptr) {
    if (TEX == 0 || lf.texture < 0) return d3(lf.albedo[0], lf.albedo[1], lf.albedo[2]);
    return texture_value<X, TEX>(sv, lf.texture, rec.u, rec.v, rec.p, turbulence);
}
// Material::emitted (src/material.rs:145-147 default, 303-305 Lambertian, 325-331 DiffuseLight).
template <bool X, int TEX>
PT_DEV D3 emitted(const SceneView &sv, const DevLeaf &lf, D3 ray_d, const HitRec &rec, const double *turbulence = nullptr) {
    if (lf.mat_kind == PT_M_LAMBERTIAN) {
        if (lf.emissive > 0.0) return leaf_texture<X, TEX>(sv, lf, rec, turbulence) * lf.emissive;
        return d3(0.0, 0.0, 0.0);
    }
    if (lf.mat_kind == PT_M_DIFFUSE_LIGHT) {
        if (dot(rec.n, ray_d) < 0.0) return leaf_texture<X, TEX>(sv, lf, rec, turbulence);
        return d3(0.0, 0.0, 0.0);
    }
    return d3(0.0, 0.0, 0.0);
}

} // namespace ptdev
