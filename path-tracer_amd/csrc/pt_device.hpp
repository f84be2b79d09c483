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

PT_DEV uint32_t pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

struct Rng { uint32_t k0, k1, ctr; };

PT_DEV Rng rng_path(uint32_t seed_mix, uint32_t pixel, uint32_t sample)
{
    Rng r;
    r.k0 = pcg(pcg(seed_mix ^ pixel) + sample);
    r.k1 = pcg(pcg(~seed_mix ^ sample) + pixel);
    r.ctr = 0;
    return r;
}

PT_DEV float rng_at(const Rng &r, uint32_t draw)
{
    uint32_t x = pcg(r.k0 + draw) ^ r.k1;
    return (float)((x >> 8) | 1u) * 0x1p-24f;
}

PT_DEV float rng_next(Rng &r) { return rng_at(r, r.ctr++); }

// ---------------------------------------------------------------- scene records

// Geometry record, 16 floats per surface, read with wave-uniform indices.
//  triangle: [0..2] v0  [3] kind  [4..6] e1  [7] e1.y*e2.z - e2.y*e1.z  [8..10] e2
//  sphere:   [0..2] position  [3] kind  [4] radius2
// Element 7 is the first minor of both determinant(-d,e1,e2) and
// determinant(b,e1,e2); it does not depend on the ray.
constexpr int GEOM_STRIDE = 16;

// Shading record, 12 floats per surface, read with per-lane indices.
//  [0..2] color  [3] type | kind<<8  [4..6] emittance  [8..10] triangle normal | sphere position
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

// src/main.cpp:120-136: jitter, then the thin-lens origin/direction when DOF is on.
template <bool FAST>
PT_DEV void camera_ray(const Camera &c, int u, int v, float xi_u, float xi_v,
                       float xi_r, float xi_phi, V3 &origin, V3 &dir)
{
    using N = Num<FAST>;
    const V3 pos = v3(c.pos[0], c.pos[1], c.pos[2]);
    origin = pos;
    dir = ray_direction_by_pixel<FAST>(c, (float)u + xi_u, (float)v + xi_v);
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

// Triangle::intersects, src/surface.hpp:47-73.  g = the surface's geometry
// record (wave-uniform).  nd = -ray_dir.  Returns true on a hit and sets t and
// distance = length(t*ray_dir).
template <bool FAST>
PT_DEV bool triangle_test(cfloat_p g, V3 o, V3 d, V3 nd, float &t_out, float &dist_out)
{
    using N = Num<FAST>;
    const V3 v0 = v3(g[0], g[1], g[2]);
    const V3 e1 = v3(g[4], g[5], g[6]);
    const V3 e2 = v3(g[8], g[9], g[10]);
    const float m_e = g[7];                    // e1.y*e2.z - e2.y*e1.z

    // determinant(mat3(-d, e1, e2))
    const float m1a = N::msub(nd.y, e2.z, e2.y, nd.z);
    const float m2a = N::msub(nd.y, e1.z, e1.y, nd.z);
    float det_a;
    if constexpr (FAST) det_a = __builtin_fmaf(e2.x, m2a, __builtin_fmaf(nd.x, m_e, -(e1.x * m1a)));
    else det_a = nd.x * m_e - e1.x * m1a + e2.x * m2a;
    bool ok = !(det_a >= -EPS_F && det_a <= EPS_F);                 // :53

    const V3 b = o - v0;                                            // :56
    const float u = N::div(N::det(nd, b, e2), det_a);               // :57
    ok = ok && !(u < -EPS_F || u >= 1.0f);                          // :58

    const float v = N::div(N::det(nd, e1, b), det_a);               // :61
    ok = ok && !(v < -EPS_F || u + v >= 1.0f);                      // :62

    // determinant(mat3(b, e1, e2))
    const float m1t = N::msub(b.y, e2.z, e2.y, b.z);
    const float m2t = N::msub(b.y, e1.z, e1.y, b.z);
    float det_t;
    if constexpr (FAST) det_t = __builtin_fmaf(e2.x, m2t, __builtin_fmaf(b.x, m_e, -(e1.x * m1t)));
    else det_t = b.x * m_e - e1.x * m1t + e2.x * m2t;
    const float t = N::div(det_t, det_a);                           // :65
    ok = ok && !(t <= EPS_F);                                       // :66

    const V3 td = d * t;                                            // :70
    t_out = t;
    dist_out = N::sqrt(N::dot(td, td));
    return ok;
}

// Sphere::intersects, src/surface.hpp:91-112.
template <bool FAST>
PT_DEV bool sphere_test(cfloat_p g, V3 o, V3 d, float &t_out, float &dist_out)
{
    using N = Num<FAST>;
    const V3 c = v3(g[0], g[1], g[2]);
    const float radius2 = g[4];
    const V3 L = c - o;
    const float tca = N::dot(L, d);
    bool ok = !(tca < 0.0f);                                        // :94
    float d2;
    if constexpr (FAST) d2 = __builtin_fmaf(-tca, tca, N::dot(L, L));
    else d2 = N::dot(L, L) - tca * tca;                             // :97
    ok = ok && !(d2 > radius2);                                     // :98
    const float thc = N::sqrt(radius2 - d2);                        // :101
    float t = tca - thc;                                            // :103
    if (t <= EPS_F) {                                               // :105
        t = tca + thc;
        ok = ok && !(t <= EPS_F);                                   // :107
    }
    t_out = t;
    dist_out = t;                                                   // :111
    return ok;
}

// closest_intersection, src/main.cpp:56-67: scene order, a later surface
// replaces the current one only when strictly closer.
template <bool FAST>
PT_DEV Hit closest_intersection(cfloat_p geom, int n_surfaces, V3 o, V3 d)
{
    Hit best;
    best.idx = -1;
    best.t = 0.0f;
    best.dist = 0.0f;
    const V3 nd = -d;
    for (int i = 0; i < n_surfaces; i++) {
        cfloat_p g = geom + (size_t)i * GEOM_STRIDE;
        float t, dist;
        bool ok;
        if (__builtin_bit_cast(int, g[3]) == 0)      // wave-uniform branch
            ok = triangle_test<FAST>(g, o, d, nd, t, dist);
        else
            ok = sphere_test<FAST>(g, o, d, t, dist);
        if (ok && (best.idx < 0 || dist < best.dist)) {
            best.idx = i;
            best.t = t;
            best.dist = dist;
        }
    }
    return best;
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

// sample_mirror, src/material.hpp:29-31
template <bool FAST>
PT_DEV V3 sample_mirror(V3 normal, V3 incident)
{
    using N = Num<FAST>;
    const float k = 2.0f * N::dot(normal, incident);
    return v3(N::mad(-k, normal.x, incident.x), N::mad(-k, normal.y, incident.y),
              N::mad(-k, normal.z, incident.z));
}

// sample_glass, src/material.hpp:33-67
template <bool FAST>
PT_DEV V3 sample_glass(V3 normal, V3 incident, float xi)
{
    using N = Num<FAST>;
    float n1 = 1.0f, n2 = 1.5f;
    if (!(N::dot(normal, incident) < 0.0f)) {
        normal = -normal;
        n1 = 1.5f;
        n2 = 1.0f;
    }
    const float n = n1 / n2;      // both branches fold to constants
    const float cosI = -N::dot(normal, incident);
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
    if (xi < reflectance) {
        const float k = 2.0f * cosI;
        return v3(N::mad(k, normal.x, incident.x), N::mad(k, normal.y, incident.y),
                  N::mad(k, normal.z, incident.z));
    }
    const float k = n * cosI - cosT;
    return v3(N::mad(k, normal.x, n * incident.x), N::mad(k, normal.y, n * incident.y),
              N::mad(k, normal.z, n * incident.z));
}

// Material::sample_pdf, src/material.hpp:78-87
template <bool FAST>
PT_DEV V3 sample_pdf(int type, V3 normal, V3 incident, float xi0, float xi1)
{
    if (type == 0) return sample_diffuse<FAST>(normal, xi0, xi1);
    if (type == 1) return sample_mirror<FAST>(normal, incident);
    return sample_glass<FAST>(normal, incident, xi0);
}

// Material::brdf, src/material.hpp:89-96: diffuse 2*color*dot(reflected, normal)
template <bool FAST>
PT_DEV V3 brdf(int type, V3 color, V3 normal, V3 reflected)
{
    if (type == 0) return (color * 2.0f) * Num<FAST>::dot(reflected, normal);
    return color;
}

}  // namespace ptd
