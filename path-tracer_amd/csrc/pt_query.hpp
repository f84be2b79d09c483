// Closest-hit queries with several rays per lane.  Included by pt_kernels.hpp only.
//
// The linear scan of closest_intersection (src/main.cpp:56-67) reads its records through the
// scalar cache, so every arithmetic instruction of a surface test has a scalar-register
// operand; two such instructions in a row hold a SIMD's issue slot for 4.1 cycles each
// where instructions on vector registers take 2.2 (tools/micro).  In the render kernels
// other waves' plain instructions fill those slots; in a query kernel every wave of a SIMD
// runs the scan at once and nothing does.  Here a lane carries N rays and a record's values
// are brought into vector registers once per surface and N rays (broadcast reads of an LDS
// copy of the records: no vector-ALU slot at all), so that the N tests are plain
// vector-register arithmetic.  The arithmetic is scan_surface's / lean_triangle's own: the
// results are those of the one-ray kernels bit for bit.
#pragma once

#include "pt_device.hpp"

namespace ptd {

// Records the query kernels keep in LDS (64 bytes each): scenes up to this many scanned
// surfaces take the N-rays-per-lane kernels, larger ones the one-ray kernels.
constexpr int QUERY_MAX_SURFACES = 256;

// Copies n records of 16 floats into LDS; every thread of the workgroup must call it.
PT_DEV void query_stage_records(const float *src, int n, float *recs)
{
    const float4 *s = reinterpret_cast<const float4 *>(src);
    float4 *d = reinterpret_cast<float4 *>(recs);
    for (int i = threadIdx.x; i < n * (GEOM_STRIDE / 4); i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

PT_DEV int uniform_kind(float w)
{
    return __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w));
}

// closest_intersection for N rays of one lane, EXACT policy: the error-band filter of
// scan_surface per ray, the rare undecided ray re-evaluated by closest_direct from the
// records in global memory (geom).  WAVE_SKIP: scan_surface's wave-level early-outs, for callers
// whose k-th rays are neighbours across the wave (camera rays).
template <int N, bool WAVE_SKIP = false>
PT_DEV void closest_hits_exact(const float *recs, cfloat_p geom, int n_surfaces, const V3 (&o)[N], const V3 (&d)[N],
                               Hit (&h)[N], unsigned long long *redo_count = nullptr)
{
    Scan st[N];
#pragma unroll
    for (int k = 0; k < N; k++) scan_begin<false>(st[k], d[k]);
    for (int i = 0; i < n_surfaces; i++) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(recs + i * GEOM_STRIDE);
        GeomRec rec;
        rec.q0 = p[0]; rec.q1 = p[1];
        rec.kind = uniform_kind(rec.q0.w);
        if (rec.kind == GEOM_TRIANGLE) {                         // wave-uniform
            rec.q2 = p[2];
#pragma unroll
            for (int k = 0; k < N; k++) scan_surface<false, WAVE_SKIP, GEOM_TRIANGLE, true, WAVE_SKIP>(st[k], i, rec, o[k], d[k]);
        } else {
            rec.q2 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < N; k++) scan_surface<false, WAVE_SKIP, GEOM_SPHERE, true>(st[k], i, rec, o[k], d[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        bool redo = false;
        if (scan_any_unsure(st[k])) redo = scan_unsure(st[k]);
        if (redo) {
            if (redo_count) atomicAdd(redo_count, 1ull);
            h[k] = closest_direct<false>(geom, n_surfaces, o[k], d[k]);
        } else {
            h[k] = scan_finish<false, true>(st[k], d[k]);
        }
    }
}

// The same for the FAST policy: lean records, lean_triangle / lean_sphere.
template <int N, bool WAVE_SKIP = false>
PT_DEV void closest_hits_lean(const float *recs, int n_surfaces, const V3 (&o)[N], const V3 (&d)[N], Hit (&h)[N])
{
    LeanBest best[N];
    LeanRay ray[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        best[k] = {-4, __builtin_inff()};
        const float len2 = __builtin_fmaf(d[k].z, d[k].z, __builtin_fmaf(d[k].y, d[k].y, d[k].x * d[k].x));
        ray[k] = {f32x2{d[k].x, o[k].x}, f32x2{d[k].y, o[k].y}, f32x2{d[k].z, o[k].z}, __builtin_amdgcn_sqrtf(len2)};
    }
    for (int i = 0; i < n_surfaces; i++) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(recs + i * LEAN_STRIDE);
        LeanRec r;
        r.q0 = p[0]; r.q1 = p[1];
        r.kind = uniform_kind(r.q0.w);
        if (r.kind == GEOM_TRIANGLE) {                           // wave-uniform
            r.q2 = p[2]; r.q3 = p[3];
#pragma unroll
            for (int k = 0; k < N; k++) lean_triangle<false, false, WAVE_SKIP>(best[k], i, r, ray[k]);
        } else {
            r.q2 = r.q3 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < N; k++) lean_sphere<WAVE_SKIP>(best[k], i, r, ray[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        h[k].idx = best[k].code >> 2;
        h[k].dist = h[k].idx < 0 ? 0.0f : best[k].dist;
        h[k].t = (best[k].code & 3) == 0 ? best[k].dist * __builtin_amdgcn_rcpf(ray[k].len_d) : best[k].dist;
        if (h[k].idx < 0) h[k].t = 0.0f;
    }
}

template <bool FAST, int N, bool WAVE_SKIP = false>
PT_DEV void closest_hits(const float *recs, const float *geom, int n_surfaces, const V3 (&o)[N], const V3 (&d)[N], Hit (&h)[N])
{
    if constexpr (FAST) closest_hits_lean<N, WAVE_SKIP>(recs, n_surfaces, o, d, h);
    else closest_hits_exact<N, WAVE_SKIP>(recs, as_constant(geom), n_surfaces, o, d, h);
}

}  // namespace ptd
