// Device-side arithmetic of the render step for gfx950: vector helpers in the
// reference's (GLM's) operation order, the two numeric policies, the
// counter-based random numbers, ray set-up, the two surface tests and the
// three material samplers.  Included by pt_kernels.hip only.
//
// The translation unit is compiled with -ffp-contract=off, so `a*b + c` below
// is a multiply followed by an add.  Fused operations appear only where a
// FAST policy asks for them by name (__builtin_fmaf).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct V3 { float x, y, z; };

#define PT_DEV __device__ __forceinline__

PT_DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
PT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }

// Numeric policy.  EXACT (FAST = false) reproduces the reference's x86-64
// build: no contraction, correctly rounded divide and square root.  FAST
// fuses multiply-adds and uses the hardware reciprocal, reciprocal square
// root, sine and cosine.
template <bool FAST>
struct Num {
    // a*b + c
    static PT_DEV float mad(float a, float b, float c)
    {
        if constexpr (FAST) return __builtin_fmaf(a, b, c);
        else return a * b + c;
    }
    // a*b - c*d
    static PT_DEV float msub(float a, float b, float c, float d)
    {
        if constexpr (FAST) return __builtin_fmaf(a, b, -(c * d));
        else return a * b - c * d;
    }
    static PT_DEV float div(float a, float b)
    {
        if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
        else return a / b;
    }
    static PT_DEV float sqrt(float a)
    {
        if constexpr (FAST) return __builtin_amdgcn_sqrtf(a);
        else return __builtin_sqrtf(a);
    }
    // 1 / sqrt(a)
    static PT_DEV float rsqrt(float a)
    {
        if constexpr (FAST) return __builtin_amdgcn_rsqf(a);
        else return 1.0f / __builtin_sqrtf(a);
    }
    // cos(2*pi*x), sin(2*pi*x) for x in [0,1)
    static PT_DEV void sincos_turns(float x, float &s, float &c)
    {
        if constexpr (FAST) {
            s = __builtin_amdgcn_sinf(x);   // v_sin_f32 takes its argument in turns
            c = __builtin_amdgcn_cosf(x);
        } else {
            s = sinpif(2.0f * x);
            c = cospif(2.0f * x);
        }
    }
    // dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z
    static PT_DEV float dot(V3 a, V3 b)
    {
        if constexpr (FAST) return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
        else return (a.x * b.x + a.y * b.y) + a.z * b.z;
    }
    // normalize(v) = v * (1/sqrt(dot(v,v)))
    static PT_DEV V3 normalize(V3 a) { return a * rsqrt(dot(a, a)); }
    // o + d*t
    static PT_DEV V3 along(V3 o, V3 d, float t)
    {
        return v3(mad(d.x, t, o.x), mad(d.y, t, o.y), mad(d.z, t, o.z));
    }
    // determinant of the matrix with columns c0,c1,c2, GLM's expansion:
    // c0.x*(c1.y*c2.z - c2.y*c1.z) - c1.x*(c0.y*c2.z - c2.y*c0.z) + c2.x*(c0.y*c1.z - c1.y*c0.z)
    static PT_DEV float det(V3 c0, V3 c1, V3 c2)
    {
        float m0 = msub(c1.y, c2.z, c2.y, c1.z);
        float m1 = msub(c0.y, c2.z, c2.y, c0.z);
        float m2 = msub(c0.y, c1.z, c1.y, c0.z);
        if constexpr (FAST) return __builtin_fmaf(c2.x, m2, __builtin_fmaf(c0.x, m0, -(c1.x * m1)));
        else return c0.x * m0 - c1.x * m1 + c2.x * m2;
    }
};

// ---------------------------------------------------------------- random numbers
// Counter-based replacement for random_real (src/main.cpp:42-44): the stream
// of a path is a pure function of (seed, pixel, sample); draw i of the path is
// a pure function of the stream and i.  Values are odd multiples of 2^-24, so
// they lie strictly inside (0,1).
//
// A stream is named by 64 bits: a key, pcg(pixel_key ^ sample * G), and the counter its draws
// start from, pixel * A + sample * B (A, B odd); draw i is pcg(key ^ (start + i * G)), G = 2^32 /
// golden ratio.  Neither level of the key is additive: with sums in place of the exclusive-ors,
// two pixels whose keys lie less than spp apart (some 10^5 pairs in an 800 x 800 image) would
// trace hundreds of identical paths at shifted sample numbers, and two paths with neighbouring
// keys would draw each other's numbers a few draws later.  The starting counter is what keeps two
// paths whose 32-bit keys meet (the birthday bound: of an image's 10^9 paths some 10^8 pairs,
// and every key eight times over in a 3840 x 2160 image at 4096 samples per pixel) from drawing
// the same numbers: their draws coincide only where key ^ counter does, one draw at a time, as
// the draws of any two paths may.  It costs two integer multiplications per path and no
// register: the counter is part of the stream's state anyway.

PT_DEV uint32_t pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

constexpr uint32_t RNG_STEP = 0x9E3779B9u;

constexpr uint32_t RNG_PIXEL_MUL = 0x85EBCA6Bu, RNG_SAMPLE_MUL = 0xC2B2AE35u;

struct Rng { uint32_t key, ctr; };      // ctr = the stream's start + RNG_STEP * (draws made so far)

// Per-pixel part of the key: computed once per work item.
PT_DEV uint32_t rng_pixel(uint32_t seed_mix, uint32_t pixel) { return pcg(seed_mix ^ pixel); }

PT_DEV Rng rng_path(uint32_t pixel_key, uint32_t pixel, uint32_t sample)
{
    Rng r;
    r.key = pcg(pixel_key ^ (sample * RNG_STEP));
    r.ctr = pixel * RNG_PIXEL_MUL + sample * RNG_SAMPLE_MUL;
    return r;
}

PT_DEV float rng_at(const Rng &r, uint32_t draw)
{
    const uint32_t x = pcg(r.key ^ (r.ctr + draw * RNG_STEP));
    return (float)((x >> 8) | 1u) * 0x1p-24f;
}

PT_DEV float rng_next(Rng &r)
{
    const uint32_t x = pcg(r.key ^ r.ctr);
    r.ctr += RNG_STEP;
    return (float)((x >> 8) | 1u) * 0x1p-24f;
}

// ---------------------------------------------------------------- scene records

// Geometry record, 16 floats per surface, read with wave-uniform indices.
//  triangle: [0..2] v0  [3] kind  [4..6] e1  [7] e1.y*e2.z - e2.y*e1.z  [8..10] e2  [12] run
//  sphere:   [0..2] position  [3] kind  [4] radius2                                 [12] run
// Element 7 is the first minor of both determinant(-d,e1,e2) and
// determinant(b,e1,e2); it does not depend on the ray.
// run (an integer, >= 1): how many surfaces from this one on, in scene order, have
// this one's kind; a scan of the whole scene reads it at the start of a run and
// then tests that many surfaces without looking at their kinds.
constexpr int GEOM_STRIDE = 16;
constexpr int GEOM_RUN = 12;
// Kinds in a geometry record.
constexpr int GEOM_TRIANGLE = 0, GEOM_SPHERE = 1;
// A triangle whose edges are so long that its determinants can leave the range
// in which the hardware reciprocal is accurate (|det| up to 6 |d| |e1| |e2|;
// v_rcp_f32 flushes a denormal result to zero) carries a NaN in element 7: every
// estimate the error-band filter forms for it is then a NaN, no band decides,
// and the ray is evaluated by closest_direct, which computes that minor itself.
// (The FAST policy has no such fallback: the NaN fails its test of t, and the triangle is never hit.)
constexpr float GEOM_DIRECT_EDGE_PRODUCT = 1e18f;     // |e1|max * |e2|max above this

// Shading record, 12 floats per surface, read with per-lane indices.
//  [0..2] color (diffuse: 2 * color, what Material::brdf multiplies by)  [3] type | kind<<8
//  [4..6] emittance  [8..10] triangle normal | sphere position
constexpr int SHADE_STRIDE = 12;

struct Camera {
    float pos[3];
    float rot[9];       // column-major
    float focal;
    int width, height;
    int dof_on;
    float dof_focal_length;
    float dof_aperture;
};

typedef const __attribute__((address_space(4))) float *cfloat_p;   // constant address space: scalar loads

PT_DEV cfloat_p as_constant(const float *p) { return (cfloat_p)(uintptr_t)p; }

// ---------------------------------------------------------------- ray set-up

// mat3 * vec3, row r = (m[0][r]*v.x + m[1][r]*v.y) + m[2][r]*v.z
template <bool FAST>
PT_DEV V3 mat_vec(const Camera &c, V3 v)
{
    using N = Num<FAST>;
    return v3(N::mad(c.rot[6], v.z, N::mad(c.rot[3], v.y, c.rot[0] * v.x)),
              N::mad(c.rot[7], v.z, N::mad(c.rot[4], v.y, c.rot[1] * v.x)),
              N::mad(c.rot[8], v.z, N::mad(c.rot[5], v.y, c.rot[2] * v.x)));
}

// get_ray_direction_by_pixel, src/main.cpp:50-54 (W/2 and H/2 are integer divisions)
template <bool FAST>
PT_DEV V3 ray_direction_by_pixel(const Camera &c, float u, float v)
{
    V3 p = v3(u - (float)(c.width / 2), v - (float)(c.height / 2), c.focal);
    return Num<FAST>::normalize(mat_vec<FAST>(c, p));
}

// The render kernels' form of the jittered pinhole direction: rot * (u - W/2, v - H/2, focal) for
// the pixel's integer corner once per work item (pixel_base), and per path the jitter along the
// matrix's first two columns (a continuous function of random numbers: fused arithmetic, not the
// reference's order of additions; pt_camera_rays keeps that order).
PT_DEV V3 pixel_base(const Camera &c, int u, int v)
{
    return mat_vec<true>(c, v3((float)(u - c.width / 2), (float)(v - c.height / 2), c.focal));
}

PT_DEV V3 jittered_direction(const Camera &c, V3 base, float xi_u, float xi_v)
{
    const V3 p = v3(__builtin_fmaf(c.rot[3], xi_v, __builtin_fmaf(c.rot[0], xi_u, base.x)),
                    __builtin_fmaf(c.rot[4], xi_v, __builtin_fmaf(c.rot[1], xi_u, base.y)),
                    __builtin_fmaf(c.rot[5], xi_v, __builtin_fmaf(c.rot[2], xi_u, base.z)));
    return Num<true>::normalize(p);
}

template <bool FAST>
PT_DEV void lens_ray(const Camera &c, float xi_r, float xi_phi, V3 &origin, V3 &dir);

// src/main.cpp:120-136: jitter, then the thin-lens origin/direction when DOF is on.
template <bool FAST>
PT_DEV void camera_ray(const Camera &c, int u, int v, float xi_u, float xi_v,
                       float xi_r, float xi_phi, V3 &origin, V3 &dir)
{
    origin = v3(c.pos[0], c.pos[1], c.pos[2]);
    dir = ray_direction_by_pixel<FAST>(c, (float)u + xi_u, (float)v + xi_v);
    lens_ray<FAST>(c, xi_r, xi_phi, origin, dir);
}

// The thin lens of src/main.cpp:124-136 applied to a pinhole ray (origin = the camera's position).
template <bool FAST>
PT_DEV void lens_ray(const Camera &c, float xi_r, float xi_phi, V3 &origin, V3 &dir)
{
    using N = Num<FAST>;
    const V3 pos = v3(c.pos[0], c.pos[1], c.pos[2]);
    if (c.dof_on) {
        // camera.hpp:91-101: forward/downward/right = rot * unit axes = the columns
        // (the products with the 0 and 1 components are exact)
        const V3 right = v3(c.rot[0], c.rot[1], c.rot[2]);
        const V3 downward = v3(c.rot[3], c.rot[4], c.rot[5]);
        const V3 forward = v3(c.rot[6], c.rot[7], c.rot[8]);
        float t = N::div(c.dof_focal_length, N::dot(forward, dir));
        V3 p = N::along(pos, dir, t);
        float r = xi_r * c.dof_aperture;       // uniform in radius (main.cpp:129)
        float sn, cs;
        N::sincos_turns(xi_phi, sn, cs);       // phi = 2*pi*xi (main.cpp:130)
        float dx = r * cs, dy = r * sn;
        origin = N::along(N::along(pos, right, dx), downward, dy);
        dir = N::normalize(p - origin);
    }
}

// ---------------------------------------------------------------- surface tests

struct Hit { int idx; float t; float dist; };

// Comparisons against the reference's `const double EPSILON = 0.00001` promote
// the float operand to double.  The fp32 equivalents (1e-5f is the float just
// below 1e-5): x < EPSILON <=> x <= 1e-5f;  x < -EPSILON <=> x < -1e-5f;
// -EPSILON < x <=> x >= -1e-5f.
constexpr float EPS_F = 1e-5f;

// The four determinants of Triangle::intersects (src/surface.hpp:51-65) in
// GLM's expansion, sharing the minors that recur (each shared value is the
// same sequence of roundings wherever the reference computes it):
//   det_a = determinant(-d, e1, e2)   du = determinant(-d, b, e2)
//   dv    = determinant(-d, e1, b)    dt = determinant(b, e1, e2)
struct TriDets { float det_a, du, dv, dt; };

template <bool FAST>
PT_DEV TriDets triangle_dets_v(V3 v0, V3 e1, float m_e, V3 e2, V3 o, V3 nd)
{
    using N = Num<FAST>;
    const V3 b = o - v0;                                 // :56
    const float m_de2 = N::msub(nd.y, e2.z, e2.y, nd.z);
    const float m_de1 = N::msub(nd.y, e1.z, e1.y, nd.z);
    const float m_be2 = N::msub(b.y, e2.z, e2.y, b.z);
    const float m_db = N::msub(nd.y, b.z, b.y, nd.z);
    const float m_e1b = N::msub(e1.y, b.z, b.y, e1.z);
    // b.y*e1.z - e1.y*b.z: the same two products subtracted the other way round, and
    // rounding to nearest is symmetric, so this is exact (the negation folds into the
    // instruction that uses it)
    const float m_be1 = -m_e1b;
    TriDets r;
    if constexpr (FAST) {
        r.det_a = __builtin_fmaf(e2.x, m_de1, __builtin_fmaf(nd.x, m_e, -(e1.x * m_de2)));
        r.du = __builtin_fmaf(e2.x, m_db, __builtin_fmaf(nd.x, m_be2, -(b.x * m_de2)));
        r.dv = __builtin_fmaf(b.x, m_de1, __builtin_fmaf(nd.x, m_e1b, -(e1.x * m_db)));
        r.dt = __builtin_fmaf(e2.x, m_be1, __builtin_fmaf(b.x, m_e, -(e1.x * m_be2)));
    } else {
        r.det_a = nd.x * m_e - e1.x * m_de2 + e2.x * m_de1;
        r.du = nd.x * m_be2 - b.x * m_de2 + e2.x * m_db;
        r.dv = nd.x * m_e1b - e1.x * m_db + b.x * m_de1;
        r.dt = b.x * m_e - e1.x * m_be2 + e2.x * m_be1;
    }
    return r;
}

// The same from a geometry record read with a wave-uniform index.
template <bool FAST>
PT_DEV TriDets triangle_dets(cfloat_p g, V3 o, V3 nd)
{
    return triangle_dets_v<FAST>(v3(g[0], g[1], g[2]), v3(g[4], g[5], g[6]), g[7], v3(g[8], g[9], g[10]), o, nd);
}

// Floats 0..11 of a geometry record, read with a wave-uniform index in one go
// (scalar loads of 8 + 4 dwords, one wait), whichever kind the surface has.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct GeomRec { f32x4 q0, q1, q2; int kind; };

PT_DEV GeomRec geom_load(cfloat_p g)
{
    typedef const __attribute__((address_space(4))) f32x4 *cf32x4_p;
    typedef const __attribute__((address_space(4))) i32x4 *ci32x4_p;
    const cf32x4_p p = (cf32x4_p)g;
    // The kind is read as an integer lane of an integer vector: a bit cast of one
    // float lane of the loaded vector makes this compiler narrow the load to a
    // single dword at the vector's base (the lane offset is lost).
    const i32x4 first = ((ci32x4_p)g)[0];
    GeomRec r;
    r.kind = first.w;
    r.q0 = __builtin_bit_cast(f32x4, first);
    r.q1 = p[1]; r.q2 = p[2];
    return r;
}

// The same when the kind is known: 12 floats of a triangle, 5 of a sphere.
PT_DEV GeomRec geom_load_triangle(cfloat_p g)
{
    typedef const __attribute__((address_space(4))) f32x4 *cf32x4_p;
    const cf32x4_p p = (cf32x4_p)g;
    GeomRec r;
    r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2];
    r.kind = GEOM_TRIANGLE;
    return r;
}

PT_DEV GeomRec geom_load_sphere(cfloat_p g)
{
    typedef const __attribute__((address_space(4))) f32x4 *cf32x4_p;
    GeomRec r;
    r.q0 = ((cf32x4_p)g)[0];
    r.q1 = f32x4{g[4], 0.0f, 0.0f, 0.0f};
    r.q2 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    r.kind = GEOM_SPHERE;
    return r;
}

// Triangle::intersects, src/surface.hpp:47-73, evaluated directly: IEEE
// quotients in EXACT.  Returns true on a hit; t, distance = length(t*ray_dir).
template <bool FAST>
PT_DEV bool triangle_direct(cfloat_p g, V3 o, V3 d, V3 nd, float &t_out, float &dist_out)
{
    using N = Num<FAST>;
    // the record's element 7 is this same minor, or a NaN (see GEOM_DIRECT_EDGE_PRODUCT)
    const V3 e1 = v3(g[4], g[5], g[6]), e2 = v3(g[8], g[9], g[10]);
    const TriDets q = triangle_dets_v<FAST>(v3(g[0], g[1], g[2]), e1, N::msub(e1.y, e2.z, e2.y, e1.z), e2, o, nd);
    bool ok = !(q.det_a >= -EPS_F && q.det_a <= EPS_F);             // :53
    const float u = N::div(q.du, q.det_a);                          // :57
    ok = ok && !(u < -EPS_F || u >= 1.0f);                          // :58
    const float v = N::div(q.dv, q.det_a);                          // :61
    ok = ok && !(v < -EPS_F || u + v >= 1.0f);                      // :62
    const float t = N::div(q.dt, q.det_a);                          // :65
    ok = ok && !(t <= EPS_F);                                       // :66
    const V3 td = d * t;                                            // :70
    t_out = t;
    dist_out = N::sqrt(N::dot(td, td));
    return ok;
}

// The part of Sphere::intersects (src/surface.hpp:91-101) before the square
// root: tca, and arg = radius2 - d2.  pre_ok is false when the reference has
// already returned (tca < 0 or d2 > radius2).
template <bool FAST>
PT_DEV void sphere_pre_terms(V3 c, float radius2, V3 o, V3 d, float &tca, float &d2, float &arg)
{
    using N = Num<FAST>;
    const V3 L = c - o;
    tca = N::dot(L, d);
    if constexpr (FAST) d2 = __builtin_fmaf(-tca, tca, N::dot(L, L));
    else d2 = N::dot(L, L) - tca * tca;                             // :97
    arg = radius2 - d2;
}

template <bool FAST>
PT_DEV bool sphere_pre_v(V3 c, float radius2, V3 o, V3 d, float &tca, float &arg)
{
    float d2;
    sphere_pre_terms<FAST>(c, radius2, o, d, tca, d2, arg);
    return !(tca < 0.0f) && !(d2 > radius2);                        // :94, :98
}

template <bool FAST>
PT_DEV bool sphere_pre(cfloat_p g, V3 o, V3 d, float &tca, float &arg)
{
    return sphere_pre_v<FAST>(v3(g[0], g[1], g[2]), g[4], o, d, tca, arg);
}

// Sphere::intersects, src/surface.hpp:91-112, evaluated directly.
template <bool FAST>
PT_DEV bool sphere_direct(cfloat_p g, V3 o, V3 d, float &t_out, float &dist_out)
{
    using N = Num<FAST>;
    float tca, arg;
    bool ok = sphere_pre<FAST>(g, o, d, tca, arg);
    const float thc = N::sqrt(arg);                                 // :101
    float t = tca - thc;                                            // :103
    if (t <= EPS_F) {                                               // :105
        t = tca + thc;
        ok = ok && !(t <= EPS_F);                                   // :107
    }
    t_out = t;
    dist_out = t;                                                   // :111
    return ok;
}

// closest_intersection, src/main.cpp:56-67, evaluated directly: scene order,
// a later surface replaces the current one only when strictly closer.
template <bool FAST>
PT_DEV Hit closest_direct(cfloat_p geom, int n_surfaces, V3 o, V3 d, const int *index_map = nullptr)
{
    Hit best;
    best.idx = -1;
    best.t = 0.0f;
    best.dist = 0.0f;
    const V3 nd = -d;
    for (int j = 0; j < n_surfaces; j++) {
        // index_map (ascending scene indices) restricts the scan to a subset of the scene
        const int i = index_map ? index_map[j] : j;
        cfloat_p g = geom + (size_t)i * GEOM_STRIDE;
        float t, dist;
        bool ok;
        if (__builtin_bit_cast(int, g[3]) == GEOM_TRIANGLE)    // wave-uniform branch
            ok = triangle_direct<FAST>(g, o, d, nd, t, dist);
        else
            ok = sphere_direct<FAST>(g, o, d, t, dist);
        if (ok && (best.idx < 0 || dist < best.dist)) {
            best.idx = i;
            best.t = t;
            best.dist = dist;
        }
    }
    return best;
}

// ---- filtered evaluation ----------------------------------------------------
//
// The direct form spends most of its instructions on three correctly rounded
// divisions and a square root per triangle (a square root per sphere), whose
// results are only compared against thresholds and against each other.  The
// filtered form computes the same numerators and denominators, in the same
// arithmetic, but replaces each quotient by numerator * v_rcp(denominator) and
// each root by v_sqrt (both within 1 ulp), and takes a decision from such an
// approximation only when it lies outside an error band round the threshold
// that is several times wider than the approximation's worst error.  A lane
// with any value inside a band, or two candidate distances closer than their
// error bounds, or a NaN, is re-evaluated by closest_direct; that happens for
// about one ray in 10^5.  The winner's t (and distance) is then computed once,
// with the reference's correctly rounded operations, from the exact numerator
// and denominator kept for it.  The result is identical to closest_direct's;
// the bit-exact parity tests run through this function.
//
// Error budget (relative, fp32): quotient' = n * rcp(d): 2^-23 + 2^-24 ~ 1.8e-7,
// plus the exact quotient's own 6e-8.  Bands: 1e-5 relative round +-EPSILON,
// 2e-6 absolute round 1 (u and v are at most about 1 there), 2e-6 relative on
// triangle distances (which also covers |d|' and the reference's roundings of
// length(t*d)), 4e-7 * (thc + |t|) absolute on sphere distances (v_sqrt's 1 ulp
// against the correctly rounded root, and one rounding of tca -+ thc).

constexpr float BAND_LO_REJ = -1.00001e-5f, BAND_LO_ACC = -0.99999e-5f;   // round -EPSILON
constexpr float BAND_T_REJ = 0.99999e-5f, BAND_T_ACC = 1.00001e-5f;       // round +EPSILON
constexpr float BAND_ONE_REJ = 1.0f + 2e-6f, BAND_ONE_ACC = 1.0f - 2e-6f; // round 1
constexpr float BAND_SUM_ACC = BAND_ONE_ACC - 1.1e-5f;                    // u + v, standing for u < 1 as well
constexpr float TRI_DIST_ERR = 2e-6f, SPH_DIST_ERR = 4e-7f;

// Lanes whose x compares with the constant K as named, as a wave mask.  The
// constant travels in the instruction: measured on this GPU (tools/micro), a
// vector instruction with a scalar-register operand issues at little more than
// half the rate of one with a literal or vector-register operand, and left alone
// the compiler keeps such thresholds in scalar registers.  (The compare writes
// VCC, the only destination the encoding with a literal has.)
typedef unsigned long long LaneMask;
constexpr uint32_t f32_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
// NAME<K>(m, x) = m OP lanes(x CMP K); the compare is written with K as its first operand.
#define PT_LANES_FN(NAME, VCMP, SOP, OPERANDS)                                                                  \
    template <uint32_t K> PT_DEV LaneMask NAME(LaneMask m, float x)                                              \
    {                                                                                                            \
        LaneMask r;                                                                                              \
        asm("v_cmp_" VCMP "_f32 vcc, %2, %3\n\ts_" SOP "_b64 %0, " OPERANDS : "=s"(r) : "s"(m), "n"(K), "v"(x) : "vcc"); \
        return r;                                                                                                \
    }
PT_LANES_FN(or_lt, "gt", "or", "%1, vcc")          // m | (x < K)
PT_LANES_FN(or_gt, "lt", "or", "%1, vcc")          // m | (x > K)
PT_LANES_FN(or_le, "ge", "or", "%1, vcc")          // m | (x <= K)
PT_LANES_FN(and_lt, "gt", "and", "%1, vcc")        // m & (x < K)
PT_LANES_FN(and_gt, "lt", "and", "%1, vcc")        // m & (x > K)
PT_LANES_FN(gt_andn, "lt", "andn2", "vcc, %1")     // (x > K) & ~m
#undef PT_LANES_FN
template <uint32_t K> PT_DEV LaneMask lanes_lt(float x) { return or_lt<K>(0ull, x); }
template <uint32_t K> PT_DEV LaneMask lanes_gt(float x) { return or_gt<K>(0ull, x); }

// Best candidate so far.  [lo, hi] brackets its exact distance; a and b are the
// exact operands of the final evaluation; code = 4 * surface index + mode, with
// mode 0: triangle, t = a / b;  mode 1: sphere, t = a - sqrt(b);  mode 2:
// sphere, t = a + sqrt(b).  code < 0: no hit yet.
struct Best { int code; float lo, hi, a, b; };

// A closest-hit scan in progress: the best candidate so far and, as a wave mask
// (one bit per lane, kept in scalar registers: no vector instruction per surface
// to accumulate it), whether every decision taken for a lane so far fell outside
// the error bands.  All lanes that want a result must run the scan together.
// (Scalar instructions issue at half the rate of vector ones on this GPU, and the
// scan's loop has one for every two vector instructions: their count matters.)
struct Scan { Best best; unsigned long long sure; V3 nd; float len_d; };

// This lane must be evaluated directly / some running lane must.
PT_DEV bool scan_unsure(const Scan &st) { return !__builtin_amdgcn_inverse_ballot_w64(st.sure); }
PT_DEV bool scan_any_unsure(const Scan &st) { return (__builtin_amdgcn_ballot_w64(true) & ~st.sure) != 0; }

template <bool FAST>
PT_DEV void scan_begin(Scan &st, V3 d)
{
    // no candidate yet: brackets that compare false with everything, so that the first
    // hit is taken whatever its distance (an overflowed, infinite one included), as
    // src/main.cpp:61 takes it
    const float NONE = __builtin_nanf("");
    st.best = {-4, NONE, NONE, 0.0f, 0.0f};
    st.nd = -d;
    st.len_d = __builtin_amdgcn_sqrtf(Num<FAST>::dot(d, d));
    // distances are estimated as t * |d|: a direction whose squared length leaves the
    // normal range (or is not a number) gets no estimates, only the direct evaluation
    st.sure = FAST ? ~0ull : __builtin_amdgcn_ballot_w64(st.len_d >= 1e-18f) & __builtin_amdgcn_ballot_w64(st.len_d <= 1e18f);
}

// One surface against the ray: i = its index in scene order.  Surfaces must
// be offered in ascending scene order (src/main.cpp:61 keeps the earlier one of
// two equally distant hits).
// WAVE_SKIP: before a sphere's square root, ask whether any lane of the wave
// got past the reference's first two tests (tca >= 0, d2 <= radius2), and
// before a triangle's v and t whether any lane can still be inside on u; if
// none, nothing can change.  Worth it when most waves miss most surfaces (many
// small ones), a loss on a few large ones: the render kernel is compiled both
// ways and chosen by scene (pt_amd.hip, WAVE_SKIP_MIN_*).
// KIND: the surface's kind when the caller knows it (rec.kind is then not read).
// SELECT_UPDATE: the new candidate replaces the old by five selects instead of five moves under a
// branch.  Measured: no difference in the render kernels (the branch is nearly always taken: some
// lane of the wave finds nearly every surface closer), 3 % slower on 16 surfaces, and 7-9 % faster
// in the closest-hit queries with several rays per lane, where the compiler otherwise turns every
// ray's update into a branch with register copies of its own.
// PRUNE (EXACT, with WAVE_SKIP): ask the wave, once a triangle's t is known, whether any lane can
// still find it closer than the candidate it has; for callers whose rays are neighbours across the
// wave (camera rays), where whole waves are past most surfaces after the first few.
template <bool FAST, bool WAVE_SKIP = false, int KIND = -1, bool SELECT_UPDATE = false, bool PRUNE = false>
PT_DEV void scan_surface(Scan &st, int i, const GeomRec &rec, V3 o, V3 d)
{
    // Decisions are kept as wave masks (ballots of single comparisons, combined with
    // scalar logic): the compiler then holds them in scalar registers across the
    // triangle / sphere branch and the accumulation of `unsure` costs no vector work.
    // A lane that is not running has 0 in every ballot; complements are only used
    // under an AND with a ballot or with the mask of running lanes.
    typedef unsigned long long Mask;
#define PT_B(x) __builtin_amdgcn_ballot_w64(x)
    Best &best = st.best;
    const float len_d = st.len_d;
    Mask acc;           // surely a hit (FAST: not used)
    Mask rej;           // surely not a hit (FAST: not a hit)
    int mode;
    float c_lo, c_hi, a, b;
    if (KIND == GEOM_TRIANGLE || (KIND < 0 && rec.kind == GEOM_TRIANGLE)) {             // wave-uniform branch
        const TriDets q = triangle_dets_v<FAST>(v3(rec.q0.x, rec.q0.y, rec.q0.z), v3(rec.q1.x, rec.q1.y, rec.q1.z),
                                                rec.q1.w, v3(rec.q2.x, rec.q2.y, rec.q2.z), o, st.nd);
        const float inv = __builtin_amdgcn_rcpf(q.det_a);
        const float u = q.du * inv;
        const Mask det_bad = PT_B(__builtin_fabsf(q.det_a) <= EPS_F);      // false for a NaN, as in the reference
        if constexpr (PRUNE && !FAST) {
            // Every lane either surely without a hit on t alone (or parallel), or with a candidate
            // that this triangle, hit or not, is surely no closer than: nothing can change, not the
            // candidate and not what is known for sure (a surface that cannot win need not be
            // decided).  A NaN anywhere compares false: that lane keeps the wave here.
            const float t_early = q.dt * inv;
            const float lo_early = (t_early * len_d) * (1.0f - TRI_DIST_ERR);
            const Mask past = or_lt<f32_bits(BAND_T_REJ)>(det_bad, t_early) | PT_B(lo_early >= best.hi);
            if ((PT_B(true) & ~past) == 0) return;
        }
        if constexpr (WAVE_SKIP) {
            // every lane surely outside on u alone (or parallel): v and t need not be formed
            const Mask out_on_u = or_gt<f32_bits(BAND_ONE_REJ)>(or_lt<f32_bits(BAND_LO_REJ)>(det_bad, u), u);
            if ((PT_B(true) & ~out_on_u) == 0) return;
        }
        const float v = q.dv * inv, t = q.dt * inv;
        const float s = u + v;
        const float dist = t * len_d;
        if constexpr (FAST) {
            rej = det_bad | PT_B(u >= 1.0f) | PT_B(s >= 1.0f);
            // t by a positive compare: a NaN (an extreme triangle's record, a non-finite ray) is no hit
            rej = or_lt<f32_bits(-EPS_F)>(or_lt<f32_bits(-EPS_F)>(rej, u), v) | ~PT_B(t > EPS_F);
            acc = ~rej;
            c_lo = c_hi = dist;
        } else {
            // min/max drop a NaN operand; a NaN in u or v still fails s < BAND_ONE_ACC,
            // so it can make neither acc nor (alone) rej true and the lane is redone
            const float low_uv = __builtin_fminf(u, v);
            rej = or_lt<f32_bits(BAND_LO_REJ)>(det_bad, low_uv);
            rej = or_gt<f32_bits(BAND_ONE_REJ)>(rej, __builtin_fmaxf(u, s));
            rej = or_lt<f32_bits(BAND_T_REJ)>(rej, t);
            // (u < 1 needs no test of its own: with v above -1e-5, u is below s + 1.00001e-5
            // plus a rounding of s, and s is held that much farther from 1)
            acc = gt_andn<f32_bits(BAND_LO_ACC)>(det_bad, low_uv);
            acc = and_lt<f32_bits(BAND_SUM_ACC)>(acc, s);
            acc = and_gt<f32_bits(BAND_T_ACC)>(acc, t);
            c_lo = dist * (1.0f - TRI_DIST_ERR);
            c_hi = dist * (1.0f + TRI_DIST_ERR);
        }
        mode = 0;
        a = q.dt;
        b = q.det_a;
    } else {
        float tca, d2, arg;
        sphere_pre_terms<FAST>(v3(rec.q0.x, rec.q0.y, rec.q0.z), rec.q1.x, o, d, tca, d2, arg);
        const Mask pre_ok = PT_B(true) & ~(PT_B(tca < 0.0f) | PT_B(d2 > rec.q1.x));     // surface.hpp:94, :98
        if constexpr (WAVE_SKIP) {
            if (pre_ok == 0) return;
        }
        const float thc = __builtin_amdgcn_sqrtf(arg);
        const float t1 = tca - thc, t2 = tca + thc;
        if constexpr (FAST) {
            const bool near = !(t1 <= EPS_F);
            c_lo = c_hi = near ? t1 : t2;
            rej = or_le<f32_bits(EPS_F)>(~pre_ok, c_lo);
            acc = ~rej;
            mode = near ? 1 : 2;
        } else {
            // The near root unless it surely lies behind the ray's start, else the far one; the
            // chosen root surely beyond the start: a hit, surely behind it: none (only the far
            // root can be), in between: undecided.
            const float e1 = SPH_DIST_ERR * (thc + __builtin_fabsf(t1));
            const bool far_lane = t1 + e1 < BAND_T_REJ;
            const float t = far_lane ? t2 : t1;
            const float e = SPH_DIST_ERR * (thc + __builtin_fabsf(t));
            c_lo = t - e;
            c_hi = t + e;
            acc = and_gt<f32_bits(BAND_T_ACC)>(pre_ok, c_lo);
            rej = or_lt<f32_bits(BAND_T_REJ)>(~pre_ok, c_hi);
            mode = far_lane ? 2 : 1;
        }
        a = tca;
        b = arg;
    }
    // src/main.cpp:61: a later surface replaces the current one only when strictly closer;
    // a candidate that is not a hit never wins and never ties
    const Mask maybe_farther = PT_B(c_hi >= best.lo);
    Mask closer;
    if constexpr (FAST) {
        closer = ~(rej | maybe_farther);        // bits of lanes that are not running are never looked at
    } else {
        closer = acc & ~maybe_farther;
        const Mask surely_farther = PT_B(c_lo >= best.hi);
        // decided: surely no hit, or a hit that is surely closer or surely not closer
        st.sure &= rej | (acc & ~(maybe_farther & ~surely_farther));
    }
    const bool take = __builtin_amdgcn_inverse_ballot_w64(closer);
    if constexpr (SELECT_UPDATE) {
        best.code = take ? 4 * i + mode : best.code;
        best.lo = take ? c_lo : best.lo;
        best.hi = take ? c_hi : best.hi;
        best.a = take ? a : best.a;
        best.b = take ? b : best.b;
    } else {
        if (take) best = {4 * i + mode, c_lo, c_hi, a, b};
    }
#undef PT_B
}

// The hit a finished scan found, with the reference's t (and distance).  Not
// for a lane whose scan is unsure: that lane is evaluated by closest_direct.
template <bool FAST, bool NEED_DIST>
PT_DEV Hit scan_finish(const Scan &st, V3 d)
{
    using N = Num<FAST>;
    const Best &best = st.best;
    Hit h;
    h.idx = best.code >> 2;
    h.t = 0.0f;
    h.dist = 0.0f;
    const int mode = best.code & 3;
    if constexpr (FAST) {
        h.t = mode == 0 ? best.a * __builtin_amdgcn_rcpf(best.b) : best.lo;
        h.dist = best.lo;
    } else if (h.idx >= 0) {
        if (mode == 0) {
            h.t = best.a / best.b;                              // surface.hpp:65
            if constexpr (NEED_DIST) {
                const V3 td = d * h.t;                          // surface.hpp:70
                h.dist = __builtin_sqrtf(N::dot(td, td));
            }
        } else {
            const float thc = __builtin_sqrtf(best.b);          // surface.hpp:101
            h.t = mode == 1 ? best.a - thc : best.a + thc;
            h.dist = h.t;                                       // surface.hpp:111
        }
    }
    return h;
}

template <bool FAST, bool NEED_DIST, bool WAVE_SKIP = false>
PT_DEV Hit closest_intersection(cfloat_p geom, int n_surfaces, V3 o, V3 d,
                                unsigned long long *redo_count = nullptr, const int *index_map = nullptr)
{
    Scan st;
    scan_begin<FAST>(st, d);
    if (index_map || !WAVE_SKIP) {
        // index_map (ascending scene indices) restricts the scan to a subset of the scene
        for (int j = 0; j < n_surfaces; j++) {
            const int i = index_map ? index_map[j] : j;
            scan_surface<FAST, WAVE_SKIP>(st, i, geom_load(geom + (size_t)i * GEOM_STRIDE), o, d);
        }
    } else {
        // Scenes of many surfaces (the ones WAVE_SKIP is for), run by run: one loop per
        // kind, no look at the kind of each surface.  Measured, 800 x 800: 80 spheres
        // 58.6 -> 51.3 ms, 242 triangles 28.7 -> 27.7 ms; a scene of 8 or 15 surfaces
        // loses 6 % to the reads at the head of each run, hence not there.  (Two surfaces
        // per trip of these loops: 2 % slower.)
        for (int i = 0; i < n_surfaces;) {
            cfloat_p head = geom + (size_t)i * GEOM_STRIDE;
            const int end = i + __builtin_bit_cast(int, head[GEOM_RUN]);
            if (__builtin_bit_cast(int, head[3]) == GEOM_TRIANGLE) {
                for (; i < end; i++)
                    scan_surface<FAST, WAVE_SKIP, GEOM_TRIANGLE>(st, i, geom_load_triangle(geom + (size_t)i * GEOM_STRIDE), o, d);
            } else {
                for (; i < end; i++)
                    scan_surface<FAST, WAVE_SKIP, GEOM_SPHERE>(st, i, geom_load_sphere(geom + (size_t)i * GEOM_STRIDE), o, d);
            }
        }
    }
    if constexpr (!FAST) {
        if (scan_any_unsure(st)) {                   // rare: redo those lanes directly
            if (scan_unsure(st)) {
                if (redo_count) atomicAdd(redo_count, 1ull);
                return closest_direct<false>(geom, n_surfaces, o, d, index_map);
            }
        }
    }
    return scan_finish<FAST, NEED_DIST>(st, d);
}

// ---- the FAST policy's own scan ------------------------------------------------
//
// PT_NUM_FAST promises images that agree statistically, not the reference's decisions, so
// its linear scan does not evaluate the reference's four determinants at all: a triangle is
// a plane (unit normal n, offset d0 = n.v0) and two affine functions of the hit point that
// give its barycentric coordinates,
//     t = (d0 - n.o) / (n.d)    P = o + t d    u = Ku.P + cu    v = Kv.P + cv
// with Ku = (e2 x N) / |N|^2, Kv = (N x e1) / |N|^2, N = e1 x e2 (so that
// o + t d = v0 + u e1 + v e2, src/surface.hpp:47-66), all fused multiply-adds: 17 arithmetic
// instructions against 29 for the fused determinants (10, seven of them packed, in the
// closest-hit queries: lean_triangle's PACKED).  The reference's acceptance tests
// (src/surface.hpp:53-66) become one signed margin, positive inside:
//     min(min(u, v) + eps, 1 - (u + v), t - eps, |n.d| - eps / |N|)
// (u < 1 follows from the other two), one comparison.  A NaN anywhere (a degenerate
// triangle's record, a non-finite ray, a hit point that overflows) is carried into the
// margin and fails that comparison: no hit.
//
// Lean record, 16 floats per surface, scene order, read with wave-uniform indices:
//  triangle: [0..2] n  [3] kind  [4] d0  [5] eps / |N| + eps  [6,7] Ku.x Kv.x  [8,9] Ku.y Kv.y  [10,11] Ku.z Kv.z
//            [12] run  [14,15] cu cv
//  sphere:   [0..2] position  [3] kind  [4] radius2                       [12] run
// The pairs sit in even-aligned scalar registers, as operands of the packed instructions below.
constexpr int LEAN_STRIDE = 16;

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct LeanRec { f32x4 q0, q1, q2, q3; int kind; };

PT_DEV LeanRec lean_load(cfloat_p g)
{
    typedef const __attribute__((address_space(4))) f32x4 *cf32x4_p;
    typedef const __attribute__((address_space(4))) i32x4 *ci32x4_p;
    const cf32x4_p p = (cf32x4_p)g;
    // the kind is read as an integer lane of an integer vector (see geom_load)
    const i32x4 first = ((ci32x4_p)g)[0];
    LeanRec r;
    r.kind = first.w;
    r.q0 = __builtin_bit_cast(f32x4, first);
    r.q1 = p[1]; r.q2 = p[2]; r.q3 = p[3];
    // all 64 bytes requested before the kind is looked at: left alone, the compiler moves the
    // second half's load into the triangle branch, which then waits for it in mid-flight
    asm volatile("" : "+s"(r.q2), "+s"(r.q3));
    return r;
}

// Best hit so far of a lean scan: code = 4 * surface index + mode (0 triangle: dist = t |d|;
// 1 sphere: dist = t), or negative; dist = +infinity before the first hit.
struct LeanBest { int code; float dist; };

// The ray as a lean scan holds it: (d.k, o.k) pairs, so that one packed instruction forms a
// term of n.d and of n.o together.
struct LeanRay { f32x2 dox, doy, doz; float len_d; };

// Packed fp32: two results per instruction, a = (a.x, a.y).  Measured (tools/micro): a vector
// instruction with a scalar-register operand occupies its issue slot for 4.1 cycles when the next
// one has such an operand too, plain or packed alike, so of a triangle's 15 record values 14
// arrive two per instruction.  LO/HI: which half of the named operand feeds BOTH results.
#define PT_PK(NAME, TEXT, OUT, ...)                                                                  \
    PT_DEV f32x2 NAME { f32x2 r; asm(TEXT : OUT(r) : __VA_ARGS__); return r; }
// s.lo * v, s.hi * v (s: scalar pair)
PT_PK(pk_mul_slo(f32x2 s, f32x2 v), "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]", "=v", "s"(s), "v"(v))
// s * v.lo (both results use v's low half)
PT_PK(pk_mul_vlo(f32x2 s, f32x2 v), "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]", "=v", "s"(s), "v"(v))
#undef PT_PK
// s.lo * v + a;  s.hi * v + a;  s * v.lo + a;  s * v.hi + a
PT_DEV f32x2 pk_fma_slo(f32x2 s, f32x2 v, f32x2 a) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "s"(s), "v"(v)); return a; }
PT_DEV f32x2 pk_fma_shi(f32x2 s, f32x2 v, f32x2 a) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a) : "s"(s), "v"(v)); return a; }
PT_DEV f32x2 pk_fma_vlo(f32x2 s, f32x2 v, f32x2 a) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "s"(s), "v"(v)); return a; }
PT_DEV f32x2 pk_fma_vhi(f32x2 s, f32x2 v, f32x2 a) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a) : "s"(s), "v"(v)); return a; }
PT_DEV f32x2 pk_add_s(f32x2 v, f32x2 s) { asm("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "s"(s)); return v; }

// (written out: the compiler's own minimum of three first passes each operand through a
// canonicalising v_max, two instructions more)
PT_DEV float lean_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// PACKED: the record's pairs as operands of packed instructions (7 of them for 14 of the 15 record
// values); otherwise one fused multiply-add per value.  Measured: where every wave of a SIMD runs
// this code (the closest-hit queries) the packed form is 12 % faster; in the render kernels, whose
// waves are spread over scan, shading and path set-up, other waves' plain instructions already
// fill the slots behind a scalar operand, and the packed form is 3 % slower.
// PRUNE: ask the wave, once t is known, whether any lane can still be hit closer than what it has
// (for callers whose rays are neighbours across the wave: camera rays).
template <bool WAVE_SKIP, bool PACKED, bool PRUNE = false>
PT_DEV void lean_triangle(LeanBest &best, int i, const LeanRec &r, const LeanRay &ray)
{
    float den, num, t;
    f32x2 uv;
    if constexpr (PACKED) {
        const f32x2 nxy = {r.q0.x, r.q0.y}, nzk = {r.q0.z, r.q0.w};
        // (n.d, n.o)
        f32x2 dn = pk_mul_slo(nxy, ray.dox);
        dn = pk_fma_shi(nxy, ray.doy, dn);
        dn = pk_fma_slo(nzk, ray.doz, dn);
        den = dn.x;
        num = r.q1.x - dn.y;
        t = num * __builtin_amdgcn_rcpf(den);
        f32x2 pxy, pz;
        pxy.x = __builtin_fmaf(t, ray.dox.x, ray.dox.y);
        pxy.y = __builtin_fmaf(t, ray.doy.x, ray.doy.y);
        pz.x = __builtin_fmaf(t, ray.doz.x, ray.doz.y);
        pz.y = pz.x;
        // (u, v) = (Ku, Kv).P + (cu, cv)
        uv = pk_mul_vlo(f32x2{r.q1.z, r.q1.w}, pxy);
        uv = pk_fma_vhi(f32x2{r.q2.x, r.q2.y}, pxy, uv);
        uv = pk_fma_vlo(f32x2{r.q2.z, r.q2.w}, pz, uv);
        uv = pk_add_s(uv, f32x2{r.q3.z, r.q3.w});
    } else {
        const float dx = ray.dox.x, dy = ray.doy.x, dz = ray.doz.x, ox = ray.dox.y, oy = ray.doy.y, oz = ray.doz.y;
        den = __builtin_fmaf(r.q0.z, dz, __builtin_fmaf(r.q0.y, dy, r.q0.x * dx));
        num = __builtin_fmaf(-r.q0.z, oz, __builtin_fmaf(-r.q0.y, oy, __builtin_fmaf(-r.q0.x, ox, r.q1.x)));
        t = num * __builtin_amdgcn_rcpf(den);
        const float px = __builtin_fmaf(t, dx, ox), py = __builtin_fmaf(t, dy, oy), pz = __builtin_fmaf(t, dz, oz);
        uv.x = __builtin_fmaf(r.q2.z, pz, __builtin_fmaf(r.q2.x, py, __builtin_fmaf(r.q1.z, px, r.q3.z)));
        uv.y = __builtin_fmaf(r.q2.w, pz, __builtin_fmaf(r.q2.y, py, __builtin_fmaf(r.q1.w, px, r.q3.w)));
    }
    if constexpr (PRUNE) {
        // no lane of the wave can still be hit closer than what it has (t behind the ray's start, or
        // the plane farther than the lane's best hit; half of eps: never stricter than the margin
        // below): the hit point and its coordinates need not be formed
        if (!__any(t > 0.5f * EPS_F && t * ray.len_d < best.dist)) return;
    }
    // the margin of the comment above, less eps throughout: min(u, v, 1 - eps - (u + v), t - 2 eps,
    // |n.d| - (eps / |N| + eps)) > -eps.  (v_min3 drops a NaN operand: 0 * (u + v) carries a NaN
    // or an overflowed hit point into the margin.)
    const float s = uv.x + uv.y;
    const float m_in = lean_min3(uv.x, uv.y, (1.0f - EPS_F) - s);
    const float m_all = lean_min3(m_in, t - 2.0f * EPS_F, __builtin_fabsf(den) - r.q1.y);
    const float margin = __builtin_fmaf(s, 0.0f, m_all);
    const float dist = t * ray.len_d;
    if (margin > -EPS_F && dist < best.dist) best = {4 * i, dist};
}

template <bool WAVE_SKIP>
PT_DEV void lean_sphere(LeanBest &best, int i, const LeanRec &r, const LeanRay &ray)
{
    const V3 d = v3(ray.dox.x, ray.doy.x, ray.doz.x);
    const float lx = r.q0.x - ray.dox.y, ly = r.q0.y - ray.doy.y, lz = r.q0.z - ray.doz.y;
    const float tca = __builtin_fmaf(lz, d.z, __builtin_fmaf(ly, d.y, lx * d.x));
    const float ll = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
    const float arg = r.q1.x - __builtin_fmaf(-tca, tca, ll);    // radius2 - d2, src/surface.hpp:97-101
    if constexpr (WAVE_SKIP) {
        // no lane of the wave gets past the reference's first two tests: no square root
        if (!__any(tca >= 0.0f && arg >= 0.0f)) return;
    }
    const float thc = __builtin_amdgcn_sqrtf(arg);
    const float t1 = tca - thc, t2 = tca + thc;
    const float t = t1 > EPS_F ? t1 : t2;                        // src/surface.hpp:103-109
    // tca >= 0, d2 <= radius2, t > eps: one margin, positive when all hold (a NaN root, when
    // d2 > radius2, drops out of the minimum, but arg is negative then; tca = 0 and d2 = radius2
    // exactly, accepted by the reference, are not: sets of measure zero)
    const float margin = __builtin_fminf(__builtin_fminf(tca, arg), t - EPS_F);
    if (margin > 0.0f && t < best.dist) best = {4 * i + 1, t};
}

// closest_intersection (src/main.cpp:56-67) for the FAST policy over lean records.
template <bool WAVE_SKIP, bool PACKED = false>
PT_DEV Hit closest_intersection_lean(cfloat_p lean, int n_surfaces, V3 o, V3 d)
{
    LeanBest best = {-4, __builtin_inff()};
    const float len2 = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
    const float len_d = __builtin_amdgcn_sqrtf(len2);
    const LeanRay ray = {f32x2{d.x, o.x}, f32x2{d.y, o.y}, f32x2{d.z, o.z}, len_d};
    if constexpr (!WAVE_SKIP) {
        for (int i = 0; i < n_surfaces; i++) {
            const LeanRec r = lean_load(lean + (size_t)i * LEAN_STRIDE);
            if (r.kind == GEOM_TRIANGLE) lean_triangle<false, PACKED>(best, i, r, ray);   // wave-uniform
            else lean_sphere<false>(best, i, r, ray);
        }
    } else {
        // many surfaces: run by run, one loop per kind (see closest_intersection)
        for (int i = 0; i < n_surfaces;) {
            cfloat_p head = lean + (size_t)i * LEAN_STRIDE;
            const int end = i + __builtin_bit_cast(int, head[GEOM_RUN]);
            if (__builtin_bit_cast(int, head[3]) == GEOM_TRIANGLE) {
                for (; i < end; i++) lean_triangle<true, PACKED>(best, i, lean_load(lean + (size_t)i * LEAN_STRIDE), ray);
            } else {
                for (; i < end; i++) lean_sphere<true>(best, i, lean_load(lean + (size_t)i * LEAN_STRIDE), ray);
            }
        }
    }
    Hit h;
    h.idx = best.code >> 2;
    h.dist = h.idx < 0 ? 0.0f : best.dist;
    h.t = (best.code & 3) == 0 ? best.dist * __builtin_amdgcn_rcpf(len_d) : best.dist;
    if (h.idx < 0) h.t = 0.0f;
    return h;
}

// ---------------------------------------------------------------- materials

// sample_diffuse, src/material.hpp:13-27: a uniform direction on the sphere,
// theta = 2*pi*u, phi = acos(2v-1), flipped into the normal's hemisphere.
// cos(phi) = 2v-1 and sin(phi) = sqrt((1-z)(1+z)) without the arc cosine.
template <bool FAST>
PT_DEV V3 sample_diffuse(V3 normal, float u, float v)
{
    using N = Num<FAST>;
    float st, ct;
    N::sincos_turns(u, st, ct);
    const float z = 2.0f * v - 1.0f;
    const float sp = N::sqrt(fmaxf((1.0f - z) * (1.0f + z), 0.0f));
    V3 dir = v3(ct * sp, st * sp, z);
    if (N::dot(dir, normal) < 0.0f) dir = -dir;
    return dir;
}

// sample_mirror, src/material.hpp:29-31.  (sample_pdf reaches the same direction through
// sample_glass's reflection branch; this form documents what that branch must equal.)
template <bool FAST>
PT_DEV V3 sample_mirror(V3 normal, V3 incident)
{
    using N = Num<FAST>;
    const float k = 2.0f * N::dot(normal, incident);
    return v3(N::mad(-k, normal.x, incident.x), N::mad(-k, normal.y, incident.y),
              N::mad(-k, normal.z, incident.z));
}

// sample_glass, src/material.hpp:33-67
// With mirror = true the reflectance is taken as 1: the reflected direction
// i + 2 cosI n' (n' = the normal turned against the ray) is sample_mirror's
// i - 2 (n.i) n, bit for bit (the two differ by exact negations only), so a
// wave whose lanes hit mirrors and glass runs one block instead of two.
template <bool FAST>
PT_DEV V3 sample_glass(V3 normal, V3 incident, float xi, bool mirror = false)
{
    using N = Num<FAST>;
    // The reference turns the normal against the ray when the ray arrives from inside
    // (material.hpp:38-44) and goes on with the turned normal n' = side * n.  Both directions it
    // can return are incident (times a scalar) plus a multiple of n': the side goes into that
    // multiple, k * n' = (side * k) * n, an exact change of sign, and the normal stays as it is.
    // cosI = -dot(n', i) = |dot(n, i)| on either side.
    const float ni = N::dot(normal, incident);
    const bool outside = ni < 0.0f;
    const float n1 = outside ? 1.0f : 1.5f, n2 = outside ? 1.5f : 1.0f;
    // n1 / n2: one constant per side (a quotient of the two selected values would be a division
    // sequence per lane)
    const float n = outside ? 1.0f / 1.5f : 1.5f / 1.0f;
    const float cosI = outside ? -ni : ni;
    const float sin2T = n * n * (1.0f - cosI * cosI);
    const float cosT = N::sqrt(1.0f - sin2T);     // NaN under total internal reflection
    float reflectance;
    if (sin2T > 1.0f) {
        reflectance = 1.0f;
    } else {
        const float rOrth = N::div(n1 * cosI - n2 * cosT, n1 * cosI + n2 * cosT);
        const float rPar = N::div(n2 * cosI - n1 * cosT, n2 * cosI + n1 * cosT);
        reflectance = (rOrth * rOrth + rPar * rPar) / 2.0f;
    }
    if (mirror | (xi < reflectance)) {
        const float k = 2.0f * cosI;
        const float ks = outside ? k : -k;
        return v3(N::mad(ks, normal.x, incident.x), N::mad(ks, normal.y, incident.y),
                  N::mad(ks, normal.z, incident.z));
    }
    const float k = n * cosI - cosT;
    const float ks = outside ? k : -k;
    return v3(N::mad(ks, normal.x, n * incident.x), N::mad(ks, normal.y, n * incident.y),
              N::mad(ks, normal.z, n * incident.z));
}

// Material::sample_pdf, src/material.hpp:78-87
template <bool FAST>
PT_DEV V3 sample_pdf(int type, V3 normal, V3 incident, float xi0, float xi1)
{
    if (type == 0) return sample_diffuse<FAST>(normal, xi0, xi1);
    return sample_glass<FAST>(normal, incident, xi0, type == 1);
}

// Material::brdf, src/material.hpp:89-96: diffuse 2*color*dot(reflected, normal)
template <bool FAST>
PT_DEV V3 brdf(int type, V3 color, V3 normal, V3 reflected)
{
    if (type == 0) return (color * 2.0f) * Num<FAST>::dot(reflected, normal);
    return color;
}

}  // namespace ptd
