// gfx950 kernels of the render step.  Included by pt_amd.hip only.
#pragma once

#include "pt_device.hpp"
#include "pt_bvh.hpp"
#include "pt_query.hpp"

namespace ptd {

constexpr int BLOCK = 256;           // 4 waves of 64
constexpr uint32_t POOL_BATCH_MIN = 64;    // work items a wave takes from the global queue at a time:
constexpr uint32_t POOL_BATCH_MAX = 256;   // at least a wave's width; the host picks (RenderParams::pool_batch)
constexpr uint32_t NO_ITEM = 0xFFFFFFFFu;
constexpr int MAX_DEPTH = 1024;      // PT_MAX_DEPTH
// The hierarchy megakernel runs its shading block when at least NUM/DEN of a wave's live lanes wait
// for it. Measured on the million-triangle heightfield (800x800, 64 spp): 1/4 21.6 ms, 3/8 18.9 ms,
// 1/2 18.0 ms, 5/8 18.5 ms, 3/4 20.0 ms.
constexpr int BVH_REFILL_NUM = 1, BVH_REFILL_DEN = 2;
constexpr int MAX_LDS_SURFACES = 64 * 1024 / (12 * 4);   // shading records kept in LDS (64 KiB)

// How a kernel sees the scene.  Without a hierarchy every surface is scanned
// in scene order.  With one (scenes of many triangles) the spheres and the
// large triangles are scanned, by their scene indices in `loose`, and the
// small triangles are searched through the hierarchy.
struct SceneView {
    const float *geom;        // every surface's geometry record, scene order, read through the scalar cache
    const float *lean;        // every surface's lean record (the FAST policy's scan), same order
    const float *shade;       // every surface's shading record
    int n_scan;               // surfaces scanned linearly
    int n_surfaces;           // surfaces in the scene
    const int *loose;         // with a hierarchy: the n_scan scene indices to scan, ascending
    const float *loose_geom;  // and their geometry records, copied out in that order (a scan of
                              // consecutive records: no look-up of an index before each one)
    const float *loose_lean;  // and their lean records (the FAST policy's scan)
    BvhView bvh;
    float bg[3];
    int max_depth;            // a path ends after this many segments (1..MAX_DEPTH)
};

// Scene kinds a kernel is compiled for.
constexpr int SCENE_LDS = 0;      // shading records staged in LDS, linear scan
constexpr int SCENE_GLOBAL = 1;   // shading records read from global memory, linear scan
constexpr int SCENE_BVH = 2;      // shading records from global memory, scan + hierarchy

// lds: the kernel's dynamic LDS; a hierarchy scene keeps the traversal stacks
// there, BVH_STACK levels of BLOCK lanes (level-major, so a level's 64 lanes
// fall in 64 different banks).
// QUERY: called by a closest-hit query kernel, every wave of which runs the scan at the same time
// (lean_triangle's PACKED).
template <bool FAST, bool NEED_DIST, int KIND, bool WAVE_SKIP = false, bool QUERY = false>
PT_DEV Hit closest_hit(const SceneView &sc, V3 o, V3 d, float *lds, unsigned long long *redo_count = nullptr)
{
    if constexpr (KIND != SCENE_BVH) {
        if constexpr (FAST) return closest_intersection_lean<WAVE_SKIP, QUERY>(as_constant(sc.lean), sc.n_scan, o, d);
        else return closest_intersection<FAST, NEED_DIST, WAVE_SKIP>(as_constant(sc.geom), sc.n_scan, o, d, redo_count);
    } else {
        Hit h;
        if constexpr (FAST) h = closest_intersection_lean<false, QUERY>(as_constant(sc.loose_lean), sc.n_scan, o, d);
        else h = closest_intersection<FAST, true>(as_constant(sc.loose_geom), sc.n_scan, o, d, redo_count);
        if (h.idx >= 0) h.idx = sc.loose[h.idx];
        if constexpr (BVH_STACK_IN_LDS) {
            closest_in_bvh<FAST>(sc.bvh, sc.geom, sc.n_surfaces, o, d, h,
                                 reinterpret_cast<int *>(lds) + threadIdx.x, BLOCK);
        } else {
            int stack[BVH_STACK];
            closest_in_bvh<FAST>(sc.bvh, sc.geom, sc.n_surfaces, o, d, h, stack, 1);
        }
        if (h.idx < 0) h.dist = 0.0f;
        return h;
    }
}

// Division of a number below 2^31 by a constant d >= 1 as a multiplication's high half and a shift
// (the host's fast_div): two instructions where the general 32-bit division is sixteen.
// With s = ceil(log2 d) and m = ceil(2^(31+s) / d) <= 2^32 the error m d - 2^(31+s) is below
// d <= 2^s, so n m / 2^(31+s) has the integer part of n / d for every n < 2^31.  (d = 1: s = 0, and
// the quotient is n itself; a power of two makes m = 2^31.)
struct FastDiv { uint32_t m, shift, d; };

PT_DEV uint32_t fast_div(uint32_t n, FastDiv f)
{
    return f.d == 1u ? n : __umulhi(n, f.m) >> f.shift;
}

struct RenderParams {
    Camera cam;
    SceneView scene;
    uint32_t seed_mix;
    uint32_t first_sample;
    uint32_t spp;
    uint32_t chunk_len;       // samples per work item
    uint32_t n_pixels;        // pixels this launch renders: the whole image, or a band of rows of it
    uint32_t first_pixel;     // the first of them, in the image's row-major order
    uint32_t n_items;         // n_pixels * ceil(spp / chunk_len); item = chunk * n_pixels + (pixel - first_pixel)
    FastDiv by_pixels, by_width;   // division by n_pixels and by the image's width (item_open)
    float *slabs;             // n_items * 3 partial sums, slab of chunk k at k * n_pixels * 3
    uint32_t pool_batch;      // work items a wave takes from the queue per atomic (POOL_BATCH_MIN..MAX)
    uint32_t *queue_head;     // zeroed before the launch
    unsigned long long *stats;  // [0] += segments, [1] += segments re-evaluated directly, [3] += hierarchy nodes
                                // opened, [4] += hierarchy leaves tested (counting instantiations only)
};

// ---------------------------------------------------------------- parity kernels

// One thread per pixel: the ray through the pixel's integer corner, closest hit.
template <bool FAST, int KIND>
__global__ void __launch_bounds__(BLOCK)
k_primary_hits(Camera cam, SceneView scene, int32_t *idx, float *dist)
{
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= cam.width * cam.height) return;
    const int x = p % cam.width, y = p / cam.width;
    const V3 o = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    const V3 d = ray_direction_by_pixel<FAST>(cam, (float)x, (float)y);
    extern __shared__ __align__(16) float lds[];
    const Hit h = closest_hit<FAST, true, KIND, false, true>(scene, o, d, lds);
    idx[p] = h.idx;
    dist[p] = h.idx < 0 ? 0.0f : h.dist;
}

template <bool FAST, int KIND>
__global__ void __launch_bounds__(BLOCK)
k_intersect_rays(SceneView scene, const float *org, const float *dir,
                 long long n, int32_t *idx, float *dist, float *point)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3 o = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    const V3 d = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    extern __shared__ __align__(16) float lds[];
    const Hit h = closest_hit<FAST, true, KIND, false, true>(scene, o, d, lds);
    idx[i] = h.idx;
    dist[i] = h.idx < 0 ? 0.0f : h.dist;
    V3 pt = v3(0, 0, 0);
    if (h.idx >= 0) pt = Num<FAST>::along(o, d, h.t);
    point[3 * i] = pt.x; point[3 * i + 1] = pt.y; point[3 * i + 2] = pt.z;
}

// The two query kernels with N rays per lane (pt_query.hpp): scenes of up to QUERY_MAX_SURFACES
// scanned surfaces and no hierarchy.  A workgroup takes BLOCK * N consecutive rays, lane t the rays
// t, t + BLOCK, ... of them (so a wave's loads and stores of one k are contiguous).
template <bool FAST, int N>
__global__ void __launch_bounds__(BLOCK)
k_primary_hits_multi(Camera cam, SceneView scene, int32_t *idx, float *dist)
{
    extern __shared__ __align__(16) float lds[];
    query_stage_records(FAST ? scene.lean : scene.geom, scene.n_scan, lds);
    const uint32_t n = (uint32_t)cam.width * (uint32_t)cam.height;
    const uint32_t width = (uint32_t)cam.width;
    const uint32_t step_y = (uint32_t)BLOCK / width, step_x = (uint32_t)BLOCK - step_y * width;
    const uint32_t p0 = blockIdx.x * (uint32_t)(BLOCK * N) + threadIdx.x;
    // pixel of ray k: p0 + k * BLOCK, stepped in (x, y) without a division per ray
    uint32_t y = p0 / width, x = p0 - y * width;
    V3 o[N], d[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        o[k] = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
        d[k] = ray_direction_by_pixel<FAST>(cam, (float)x, (float)y);
        x += step_x; y += step_y;
        if (x >= width) { x -= width; y++; }
    }
    Hit h[N];
    // Ray k of a wave's 64 lanes goes through 64 neighbouring pixels of one image row: most surfaces
    // are missed by all of them together, and the scan asks the wave before it forms what only a
    // possible hit needs (scan_surface's WAVE_SKIP; 12 % faster on the 8 surfaces of cornell_2, 15 %
    // on the 16 of cornell, where the same early-outs cost the render kernels' unrelated rays 3-10 %).
    closest_hits<FAST, N, true>(lds, scene.geom, scene.n_scan, o, d, h);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t p = p0 + (uint32_t)(k * BLOCK);
        if (p < n) {
            idx[p] = h[k].idx;
            dist[p] = h[k].idx < 0 ? 0.0f : h[k].dist;
        }
    }
}

template <bool FAST, int N>
__global__ void __launch_bounds__(BLOCK)
k_intersect_rays_multi(SceneView scene, const float *org, const float *dir,
                       long long n, int32_t *idx, float *dist, float *point)
{
    extern __shared__ __align__(16) float lds[];
    query_stage_records(FAST ? scene.lean : scene.geom, scene.n_scan, lds);
    const long long i0 = (long long)blockIdx.x * (BLOCK * N) + threadIdx.x;
    V3 o[N], d[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        long long i = i0 + k * BLOCK;
        if (i >= n) i = n - 1;                                  // a ray to keep the lane company; not stored
        o[k] = v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        d[k] = v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    }
    Hit h[N];
    closest_hits<FAST, N>(lds, scene.geom, scene.n_scan, o, d, h);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const long long i = i0 + k * BLOCK;
        if (i < n) {
            idx[i] = h[k].idx;
            dist[i] = h[k].idx < 0 ? 0.0f : h[k].dist;
            V3 pt = v3(0, 0, 0);
            if (h[k].idx >= 0) pt = Num<FAST>::along(o[k], d[k], h[k].t);
            point[3 * i] = pt.x; point[3 * i + 1] = pt.y; point[3 * i + 2] = pt.z;
        }
    }
}

// ---- the linear scan with records staged through LDS ---------------------------
//
// The workgroup copies the geometry records into LDS in tiles of SCAN_TILE,
// two buffers (tile k + 1 arrives by LDS-DMA while tile k is scanned), and every
// lane scans the tile with broadcast LDS reads.  This is the textbook layout
// for a brute-force scan; it exists here to be measured against the scan through
// the scalar cache (closest_intersection), which needs no LDS, no barriers and no
// vector registers for the records.  Closest-hit queries only.
constexpr int SCAN_TILE = 256;                                   // records per tile: 16 KiB
constexpr int SCAN_TILE_FLOATS = SCAN_TILE * GEOM_STRIDE;

// Every thread of the workgroup must call this, with the same scene (o, d: its own ray).
template <bool FAST>
PT_DEV Hit closest_hit_tiled(const SceneView &sc, V3 o, V3 d, float *tiles /* 2 * SCAN_TILE_FLOATS */)
{
    typedef const __attribute__((address_space(1))) void *global_p;
    typedef __attribute__((address_space(3))) void *lds_p;
    const int n = sc.n_surfaces;
    const int n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const int wave_first = (int)(threadIdx.x & ~63u);            // first thread of this wave
    // One instruction copies 16 bytes per lane: BLOCK lanes move BLOCK / 4 records.
    constexpr int COPIES = SCAN_TILE * (GEOM_STRIDE / 4) / BLOCK;
    auto request = [&](int k) {
        float *dst = tiles + (k & 1) * SCAN_TILE_FLOATS;
        const float *src = sc.geom + (size_t)k * SCAN_TILE_FLOATS;
#pragma unroll
        for (int m = 0; m < COPIES; m++) {
            const int quad = m * BLOCK + (int)threadIdx.x;                     // 16-byte piece of the tile
            if ((size_t)k * SCAN_TILE + quad / 4 < (size_t)n)             // no read beyond the last record
                __builtin_amdgcn_global_load_lds((global_p)(uintptr_t)(src + quad * 4),
                                                 (lds_p)(uintptr_t)(dst + (m * BLOCK + wave_first) * 4), 16, 0, 0);
        }
    };

    Scan st;
    scan_begin<FAST>(st, d);
    if (n_tiles > 0) request(0);
    for (int k = 0; k < n_tiles; k++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's pieces of tile k have landed
        __syncthreads();                                        // everyone's have, and tile k - 1 is no longer read
        if (k + 1 < n_tiles) request(k + 1);
        const float *tile = tiles + (k & 1) * SCAN_TILE_FLOATS;
        const int count = n - k * SCAN_TILE < SCAN_TILE ? n - k * SCAN_TILE : SCAN_TILE;
        for (int j = 0; j < count; j++) {
            const f32x4 *p = reinterpret_cast<const f32x4 *>(tile + j * GEOM_STRIDE);
            GeomRec rec;
            rec.q0 = p[0]; rec.q1 = p[1]; rec.q2 = p[2];
            rec.kind = __builtin_bit_cast(int, tile[j * GEOM_STRIDE + 3]);
            scan_surface<FAST>(st, k * SCAN_TILE + j, rec, o, d);
        }
    }
    if constexpr (!FAST) {
        if (scan_unsure(st)) return closest_direct<false>(as_constant(sc.geom), n, o, d);
    }
    return scan_finish<FAST, true>(st, d);
}

template <bool FAST>
__global__ void __launch_bounds__(BLOCK)
k_primary_hits_tiled(Camera cam, SceneView scene, int32_t *idx, float *dist)
{
    __shared__ __align__(16) float tiles[2 * SCAN_TILE_FLOATS];
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = p < cam.width * cam.height;
    const int x = live ? p % cam.width : 0, y = live ? p / cam.width : 0;
    const V3 o = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    const V3 d = ray_direction_by_pixel<FAST>(cam, (float)x, (float)y);
    const Hit h = closest_hit_tiled<FAST>(scene, o, d, tiles);
    if (!live) return;
    idx[p] = h.idx;
    dist[p] = h.idx < 0 ? 0.0f : h.dist;
}

template <bool FAST>
__global__ void __launch_bounds__(BLOCK)
k_intersect_rays_tiled(SceneView scene, const float *org, const float *dir,
                       long long n, int32_t *idx, float *dist, float *point)
{
    __shared__ __align__(16) float tiles[2 * SCAN_TILE_FLOATS];
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = i < n;
    const long long r = live ? i : 0;                           // idle threads still help with the tiles
    const V3 o = v3(org[3 * r], org[3 * r + 1], org[3 * r + 2]);
    const V3 d = v3(dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]);
    const Hit h = closest_hit_tiled<FAST>(scene, o, d, tiles);
    if (!live) return;
    idx[i] = h.idx;
    dist[i] = h.idx < 0 ? 0.0f : h.dist;
    V3 pt = v3(0, 0, 0);
    if (h.idx >= 0) pt = Num<FAST>::along(o, d, h.t);
    point[3 * i] = pt.x; point[3 * i + 1] = pt.y; point[3 * i + 2] = pt.z;
}

template <bool FAST>
__global__ void __launch_bounds__(BLOCK)
k_camera_rays(Camera cam, const int32_t *px, const float *xi, long long n, float *org, float *dir)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    V3 o, d;
    camera_ray<FAST>(cam, px[2 * i], px[2 * i + 1], xi[4 * i], xi[4 * i + 1], xi[4 * i + 2],
                     xi[4 * i + 3], o, d);
    org[3 * i] = o.x; org[3 * i + 1] = o.y; org[3 * i + 2] = o.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
}

template <bool FAST>
__global__ void __launch_bounds__(BLOCK)
k_eval_bsdf(const int32_t *type, const float *color, const float *normal, const float *incident,
            const float *xi, long long n, float *out_dir, float *out_brdf)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3 nrm = v3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
    const V3 inc = v3(incident[3 * i], incident[3 * i + 1], incident[3 * i + 2]);
    const V3 col = v3(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
    const V3 d = sample_pdf<FAST>(type[i], nrm, inc, xi[2 * i], xi[2 * i + 1]);
    const V3 f = brdf<FAST>(type[i], col, nrm, d);
    out_dir[3 * i] = d.x; out_dir[3 * i + 1] = d.y; out_dir[3 * i + 2] = d.z;
    out_brdf[3 * i] = f.x; out_brdf[3 * i + 1] = f.y; out_brdf[3 * i + 2] = f.z;
}

__global__ void __launch_bounds__(BLOCK)
k_eval_brdf(const int32_t *type, const float *color, const float *normal, const float *reflected,
            long long n, float *out_brdf)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3 nrm = v3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
    const V3 refl = v3(reflected[3 * i], reflected[3 * i + 1], reflected[3 * i + 2]);
    const V3 col = v3(color[3 * i], color[3 * i + 1], color[3 * i + 2]);
    const V3 f = brdf<false>(type[i], col, nrm, refl);
    out_brdf[3 * i] = f.x; out_brdf[3 * i + 1] = f.y; out_brdf[3 * i + 2] = f.z;
}

__global__ void __launch_bounds__(BLOCK)
k_rng_uniform(uint32_t seed_mix, const uint32_t *pixel, const uint32_t *sample,
              const uint32_t *draw, long long n, float *out)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Rng r = rng_path(rng_pixel(seed_mix, pixel[i]), pixel[i], sample[i]);
    out[i] = rng_at(r, draw[i]);
}

// The lean records of the FAST policy's scan (pt_device.hpp), one thread per surface, from
// the geometry records: a triangle's unit normal n and d0 = n.v0, and the affine maps
// P -> u, v with o + t d = v0 + u e1 + v e2 at the hit point P, evaluated in double and
// rounded once.  A degenerate triangle gets NaNs: never hit.
__global__ void __launch_bounds__(BLOCK)
k_lean_records(const float *geom, int n, float *lean)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *g = geom + (size_t)i * GEOM_STRIDE;
    float *ln = lean + (size_t)i * LEAN_STRIDE;
    float out[LEAN_STRIDE];
    for (int k = 0; k < LEAN_STRIDE; k++) out[k] = 0.0f;
    out[3] = g[3];                      // kind
    out[GEOM_RUN] = g[GEOM_RUN];
    if (__float_as_int(g[3]) == GEOM_TRIANGLE) {
        const double v0[3] = {g[0], g[1], g[2]}, e1[3] = {g[4], g[5], g[6]}, e2[3] = {g[8], g[9], g[10]};
        const double N[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double nn = N[0] * N[0] + N[1] * N[1] + N[2] * N[2], len = sqrt(nn);
        const double ku[3] = {(e2[1] * N[2] - e2[2] * N[1]) / nn, (e2[2] * N[0] - e2[0] * N[2]) / nn, (e2[0] * N[1] - e2[1] * N[0]) / nn};
        const double kv[3] = {(N[1] * e1[2] - N[2] * e1[1]) / nn, (N[2] * e1[0] - N[0] * e1[2]) / nn, (N[0] * e1[1] - N[1] * e1[0]) / nn};
        double d0 = 0, cu = 0, cv = 0;
        for (int k = 0; k < 3; k++) {
            d0 += N[k] / len * v0[k];
            cu -= ku[k] * v0[k];
            cv -= kv[k] * v0[k];
            out[k] = (float)(N[k] / len);
            out[6 + 2 * k] = (float)ku[k];
            out[7 + 2 * k] = (float)kv[k];
        }
        out[4] = (float)d0;
        // |determinant(-d, e1, e2)| = |n.d| |N| against EPSILON (surface.hpp:53); the scan's margin is
        // formed less eps throughout (lean_triangle)
        out[5] = (float)(1e-5 / len + (double)EPS_F);
        out[14] = (float)cu;
        out[15] = (float)cv;
    } else {
        out[0] = g[0]; out[1] = g[1]; out[2] = g[2];
        out[4] = g[4];
    }
    for (int k = 0; k < LEAN_STRIDE; k++) ln[k] = out[k];
}

// ---------------------------------------------------------------- one path segment

// A path between two segments: what both kernel layouts carry per path.
struct PathState {
    V3 o, d;            // ray of the next segment
    V3 thr, col;        // throughput and radiance so far
    Rng rng;
    int depth;
};

// The start of a path: trace_rays' per-pixel set-up, src/main.cpp:120-136.
// base: pixel_base of the path's pixel (the part of the ray set-up that is the same for every
// sample of a pixel is formed once per work item).
PT_DEV void path_begin(PathState &ps, const Camera &cam, Rng stream, V3 base)
{
    ps.rng = stream;
    const float xu = rng_next(ps.rng), xv = rng_next(ps.rng);
    float xr = 0.0f, xp = 0.0f;
    if (cam.dof_on) { xr = rng_next(ps.rng); xp = rng_next(ps.rng); }
    ps.o = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    ps.d = jittered_direction(cam, base, xu, xv);
    lens_ray<true>(cam, xr, xp, ps.o, ps.d);
    ps.col = v3(0, 0, 0);
    ps.thr = v3(1, 1, 1);
    ps.depth = 0;
}

// One iteration of trace_ray's loop, src/main.cpp:84-109, in two parts: the
// closest hit (closest_hit), then shade_hit below, which returns true when the
// path has ended (ps.col is then its radiance).  shade = the scene's shading
// records (LDS or global memory).
//
// FAST selects the arithmetic of the closest-hit search and of the hit point:
// EXACT finds, for every ray, the surface and parameter t the reference's
// arithmetic finds.  Quantities that are continuous functions of random
// numbers - the jittered camera ray, sphere normals, sampled directions,
// throughput - use fused multiply-adds and the hardware reciprocal/root/sine
// (each within 1 ulp) in both modes: the reference evaluates them through libm
// in double, which no fp32 path matches bit for bit, and they are checked to
// 1e-5 instead.
template <bool FAST>
PT_DEV bool shade_hit(PathState &ps, const SceneView &sc, const float *shade, const Hit &h)
{
    using N = Num<FAST>;
    using S = Num<true>;
    const float (&bg)[3] = sc.bg;
    if (h.idx < 0) {                                // main.cpp:105-108
        ps.col = v3(S::mad(bg[0], ps.thr.x, ps.col.x), S::mad(bg[1], ps.thr.y, ps.col.y),
                    S::mad(bg[2], ps.thr.z, ps.col.z));
        return true;
    }
    const float4 *rec = reinterpret_cast<const float4 *>(shade + (size_t)h.idx * SHADE_STRIDE);
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    const int flags = __builtin_bit_cast(int, r0.w);
    const int type = flags & 0xff;
    const V3 point = N::along(ps.o, ps.d, h.t);
    V3 normal = v3(r2.x, r2.y, r2.z);
    if (flags >> 8) normal = S::normalize(point - normal);       // sphere, surface.hpp:86-88
    float xi0 = 0.0f, xi1 = 0.0f;
    // diffuse draws two numbers, glass one, a mirror none: the first draw in one block for both
    if (type != 1) xi0 = rng_next(ps.rng);
    if (type == 0) xi1 = rng_next(ps.rng);
    const V3 new_dir = sample_pdf<true>(type, normal, ps.d, xi0, xi1);
    ps.o = point;
    ps.col = v3(S::mad(r1.x, ps.thr.x, ps.col.x), S::mad(r1.y, ps.thr.y, ps.col.y),
                S::mad(r1.z, ps.thr.z, ps.col.z));             // emission at every hit, main.cpp:92
    // Material::brdf, material.hpp:89-96: the record holds 2 * color for a diffuse surface (an exact
    // doubling, done once per scene), color otherwise
    V3 f = v3(r0.x, r0.y, r0.z);
    if (type == 0) f = f * S::dot(new_dir, normal);
    ps.thr = ps.thr * f;
    ps.d = new_dir;
    bool ended = false;
    if (ps.depth + 1 >= sc.max_depth) {             // the cap on path length (the reference has none)
        ended = true;
    } else if (ps.depth >= 5) {                     // russian roulette, main.cpp:97-104
        // clamp(max(throughput), 0, 0.95): one maximum of three, then the two bounds (0.95 as a
        // literal; compares and selects, as the reference's std::max and std::clamp read, are eight)
        float p;
        asm("v_max3_f32 %0, %1, %2, %3\n\tv_min_f32 %0, 0x3f733333, %0\n\tv_max_f32 %0, 0, %0"
            : "=&v"(p) : "v"(ps.thr.x), "v"(ps.thr.y), "v"(ps.thr.z));
        if (rng_next(ps.rng) > p) ended = true;
        else ps.thr = ps.thr * S::div(1.0f, p);
    }
    ps.depth++;
    return ended;
}

template <bool FAST, int KIND, bool WAVE_SKIP = false>
PT_DEV bool path_segment(PathState &ps, const SceneView &sc, const float *shade, float *lds,
                         unsigned long long *redo_count)
{
    const Hit h = closest_hit<FAST, false, KIND, WAVE_SKIP>(sc, ps.o, ps.d, lds, redo_count);
    return shade_hit<FAST>(ps, sc, shade, h);
}

// ---------------------------------------------------------------- work queue

// A wave's share of the global work queue: items [next, end) already taken
// from the queue head and not yet handed to a lane.  All fields are equal in
// every live lane of the wave.
struct WorkPool { uint32_t next, end; bool drained; };

// Hands one work item to every lane with want = true, refilling the pool with
// `batch` items (>= 64) from the global head when it runs short (one atomic per
// refill).  Must be called by every live lane of the wave together.  Returns
// NO_ITEM to a lane once the queue is empty.
PT_DEV uint32_t pool_take(WorkPool &pool, bool want, uint32_t *queue_head, uint32_t n_items, uint32_t batch)
{
    const uint32_t lane = __lane_id();
    const unsigned long long mask = __ballot(want);
    uint32_t cnt = (uint32_t)__popcll(mask);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    uint32_t got = NO_ITEM;
    const uint32_t avail = pool.end - pool.next;
    if (want && rank < avail) got = pool.next + rank;
    if (cnt <= avail) {
        pool.next += cnt;
        return got;
    }
    cnt -= avail;                               // 1..64 lanes still without an item
    uint32_t base = n_items;
    if (!pool.drained) {
        uint32_t b = 0;
        if (lane == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) b = atomicAdd(queue_head, batch);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    }
    uint32_t end = base + batch;
    if (base >= n_items) { base = end = n_items; pool.drained = true; }
    else if (end > n_items) end = n_items;
    if (want && rank >= avail && base + (rank - avail) < end) got = base + (rank - avail);
    pool.next = base + cnt < end ? base + cnt : end;
    pool.end = end;
    return got;
}

// What a lane keeps about its current work item (pixel, chunk of samples).
struct ItemState {
    bool have;
    uint32_t item, pixel_key;
    uint32_t left;      // samples of the item not yet started
    // the next sample's terms of rng_path, kept by addition from sample to sample (the same
    // integers modulo 2^32 as its multiplications: a 32-bit multiply takes two issue slots)
    uint32_t key_term;  // sample * RNG_STEP
    uint32_t ctr;       // pixel * RNG_PIXEL_MUL + sample * RNG_SAMPLE_MUL
    V3 base;            // pixel_base of the item's pixel
    V3 sum;
};

// The random stream of the item's next sample (rng_path of the pixel and that sample).
PT_DEV Rng item_stream(const ItemState &it)
{
    Rng r;
    r.key = pcg(it.pixel_key ^ it.key_term);
    r.ctr = it.ctr;
    return r;
}

// One sample of the item has been started and finished.
PT_DEV void item_next_sample(ItemState &it)
{
    it.left--;
    it.key_term += RNG_STEP;
    it.ctr += RNG_SAMPLE_MUL;
}

PT_DEV void item_open(ItemState &it, uint32_t item, const RenderParams &P)
{
    const Camera &cam = P.cam;
    const uint32_t n_pixels = P.n_pixels, first_pixel = P.first_pixel, first_sample = P.first_sample, spp = P.spp,
                   chunk_len = P.chunk_len, seed_mix = P.seed_mix;
    it.item = item;
    it.have = true;
    const uint32_t chunk = fast_div(item, P.by_pixels);
    const uint32_t pixel = first_pixel + (item - chunk * n_pixels);
    const uint32_t y = fast_div(pixel, P.by_width), x = pixel - y * (uint32_t)cam.width;
    it.base = pixel_base(cam, (int)x, (int)y);
    const uint32_t s = first_sample + chunk * chunk_len, last = first_sample + spp;
    it.left = (s + chunk_len < last ? s + chunk_len : last) - s;
    it.key_term = s * RNG_STEP;
    it.ctr = pixel * RNG_PIXEL_MUL + s * RNG_SAMPLE_MUL;
    it.sum = v3(0, 0, 0);
    it.pixel_key = rng_pixel(seed_mix, pixel);
}

PT_DEV void item_close(ItemState &it, float *slabs)
{
    float *dst = slabs + 3ull * it.item;
    dst[0] = it.sum.x; dst[1] = it.sum.y; dst[2] = it.sum.z;
    it.have = false;
}

// ---------------------------------------------------------------- megakernel

// Persistent lanes.  A lane owns one work item (pixel, chunk of samples) at a
// time, traces its paths one segment per loop iteration, and starts the next
// path - or fetches the next item - in the iteration after a path ends, so
// lanes of a wave stay busy whatever the lengths of their paths.  Items come
// from a global counter, P.pool_batch at a time per wave, and are handed to
// lanes from a wave-uniform pool.  A lane leaves the loop when the queue is
// empty: every wave reaches that state, so the grid always drains.
//
// KIND: SCENE_LDS when the shading records fit in LDS (staged there once per
// workgroup), SCENE_GLOBAL when lanes read them from global memory (scenes
// above MAX_LDS_SURFACES), SCENE_BVH when the scene has a hierarchy.
// (At most 96 SGPRs, so that eight waves fit a SIMD: left alone the EXACT variant
// takes 102 and runs seven, measured 3.8 % slower.)
// SPHERE_SKIP: the scan's wave-level early-outs (scan_surface's WAVE_SKIP).
template <bool FAST, int KIND, bool SPHERE_SKIP = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(96)))
k_render_mega(RenderParams P)
{
    extern __shared__ __align__(16) float lds_shade[];
    const float *shade = P.scene.shade;
    if constexpr (KIND == SCENE_LDS) {
        for (int i = threadIdx.x; i < P.scene.n_surfaces * SHADE_STRIDE; i += BLOCK)
            lds_shade[i] = P.scene.shade[i];
        __syncthreads();
        shade = lds_shade;
    }

    WorkPool pool = {0, 0, false};
    ItemState it = {false, 0, 0, 0, 0, 0, v3(0, 0, 0), v3(0, 0, 0)};
    bool need_path = true;
    PathState ps = {v3(0, 0, 0), v3(0, 0, 1), v3(1, 1, 1), v3(0, 0, 0), {0, 0}, 0};
    unsigned long long n_seg = 0;

    for (;;) {
        const bool want = need_path && it.left == 0;
        if (__any(want)) {                      // wave-uniform: every live lane enters
            if (want && it.have) item_close(it, P.slabs);
            const uint32_t got = pool_take(pool, want, P.queue_head, P.n_items, P.pool_batch);
            if (want) {
                if (got == NO_ITEM) break;      // queue empty: this lane retires
                item_open(it, got, P);
            }
        }
        if (need_path) {
            path_begin(ps, P.cam, item_stream(it), it.base);
            need_path = false;
        }
        const bool ended = path_segment<FAST, KIND, SPHERE_SKIP>(ps, P.scene, shade, lds_shade, P.stats + 1);
        n_seg++;
        if (ended) {
            it.sum = it.sum + ps.col;           // buffer[u][v] += color, main.cpp:139
            item_next_sample(it);
            need_path = true;
        }
    }
    atomicAdd(P.stats, n_seg);
}

// The megakernel with the scan's records staged through LDS tiles shared by the workgroup
// (PT_ACCEL_LDS_TILES; closest_hit_tiled): the layout to measure against the scan through
// the scalar cache.  The tiles are loaded by all 256 threads between barriers, so the
// workgroup iterates in step: a lane that finds the queue empty stays in the loop, idle
// (its ray is a fixed dummy), until no lane of the workgroup has work left.  Shading
// records are read from global memory: the tiles take the LDS.
template <bool FAST>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(96)))
k_render_mega_tiled(RenderParams P)
{
    __shared__ __align__(16) float tiles[2 * SCAN_TILE_FLOATS];
    const float *shade = P.scene.shade;
    WorkPool pool = {0, 0, false};
    ItemState it = {false, 0, 0, 0, 0, 0, v3(0, 0, 0), v3(0, 0, 0)};
    bool need_path = true, alive = true;
    PathState ps = {v3(0, 0, 0), v3(0, 0, 1), v3(1, 1, 1), v3(0, 0, 0), {0, 0}, 0};
    unsigned long long n_seg = 0;

    for (;;) {
        const bool want = alive && need_path && it.left == 0;
        if (__any(want)) {                      // wave-uniform: every lane of the wave enters
            if (want && it.have) item_close(it, P.slabs);
            const uint32_t got = pool_take(pool, want, P.queue_head, P.n_items, P.pool_batch);
            if (want) {
                if (got == NO_ITEM) alive = false;
                else item_open(it, got, P);
            }
        }
        if (__syncthreads_count(alive) == 0) break;             // workgroup-uniform
        if (alive && need_path) {
            path_begin(ps, P.cam, item_stream(it), it.base);
            need_path = false;
        }
        const V3 o = alive ? ps.o : v3(0, 0, 0), d = alive ? ps.d : v3(0, 0, 1);
        const Hit h = closest_hit_tiled<FAST>(P.scene, o, d, tiles);
        if (alive) {
            n_seg++;
            if (shade_hit<FAST>(ps, P.scene, shade, h)) {
                it.sum = it.sum + ps.col;           // buffer[u][v] += color, main.cpp:139
                item_next_sample(it);
                need_path = true;
            }
        }
    }
    atomicAdd(P.stats, n_seg);
}

// The megakernel for scenes with a hierarchy.  A traversal takes tens of
// dependent steps whose number differs from lane to lane, so lanes are
// decoupled at the granularity of one traversal step: a lane whose traversal
// has finished waits until at least BVH_REFILL_NUM/DEN of the wave's live
// lanes wait too; then they all shade their hits, start their next segments
// (or paths, or work items) and scan the surfaces outside the hierarchy
// together, while in every iteration each traversing lane advances by one node
// or leaf.  The records of a step are fetched by the wave as a whole
// (bvh_fetch_records), which needs all 64 lanes: a lane that finds the queue
// empty stays in the loop, idle, until the whole wave is done.
// (Held to 80 VGPRs, 6 waves per SIMD: left alone it takes 83 and runs 5, measured
// 5 % slower; forced down to 64 VGPRs for 8 waves it spills, 9 % slower.  WAVES = 7,
// 72 VGPRs, keeps eight loop-invariant values in scratch memory: 3.4 % faster on a
// launch large enough to fill seven waves per SIMD, 4 % slower on one that is not;
// the host picks by launch size.)
// Where its instructions go, per 64 segments on the million-triangle field (counted: 15.4 loop
// iterations of some 170 vector instructions - record fetch, node step, leaf test - for 9.4
// traversal steps per lane, and 2.0 runs of the shading block of some 890, over half of it the
// scan of the 8 surfaces outside the hierarchy).  Also measured and not kept: two paths per lane,
// the one whose traversal has finished parked while the other steps, so that the shading block
// runs with most lanes (1.4 runs per 64 segments instead of 2.0).  The parked paths' state (21
// words each) has nowhere to wait: in LDS it halves the waves per SIMD, in private memory it is
// 66 MB that every segment reads and writes through the caches, and the kernel took 3x as long.
// COUNT: also counts the node records opened and the leaf triangles tested (stats[3], stats[4]):
// an instantiation of its own (PT_MODE_COUNT_STEPS), the two counters cost registers.
// SPHERES: the hierarchy holds spheres too (bvh_test_leaf).
template <bool FAST, int WAVES = 6, bool COUNT = false, bool SPHERES = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
k_render_mega_bvh(RenderParams P)
{
    __shared__ __align__(16) char fetch_lds[(BLOCK / 64) * BVH_FETCH_LDS];
    char *lds_wave = fetch_lds + (threadIdx.x / 64) * BVH_FETCH_LDS;
    const float *shade = P.scene.shade;
    WorkPool pool = {0, 0, false};
    ItemState it = {false, 0, 0, 0, 0, 0, v3(0, 0, 0), v3(0, 0, 0)};
    PathState ps = {v3(0, 0, 0), v3(0, 0, 1), v3(1, 1, 1), v3(0, 0, 0), {0, 0}, 0};
    BvhBest best = {BVH_NO_HIT, 0.0f, 0.0f};
    BvhCursor cur;
    cur.node = 0; cur.sp = 0; cur.top = 0; cur.inv = v3(0, 0, 0); cur.len_d = 1.0f; cur.inv_len = 1.0f; cur.slack = 0.0f; cur.direct = false;
    // the traversal stack, and behind it the ray parameter of the segment's hit outside the
    // hierarchy: needed once, after the traversal, so it waits in memory, not in a register
    int stack[BVH_STACK + 1];
    bool alive = true;              // false: the queue is empty and this lane has nothing left
    bool traversing = false;        // false: waiting to shade / start a segment
    bool have_hit = false;          // a finished segment waits to be shaded
    unsigned long long n_seg = 0;
    uint32_t n_nodes = 0, n_leaves = 0;

    for (;;) {
        const int n_alive = __popcll(__ballot(alive));
        if (n_alive == 0) break;                                 // wave-uniform
        const int n_waiting = __popcll(__ballot(alive && !traversing));
        if (BVH_REFILL_DEN * n_waiting >= BVH_REFILL_NUM * n_alive) {          // wave-uniform
            const bool waits = alive && !traversing;
            bool need_path = false;
            if (waits) {
                need_path = !have_hit;
                if (have_hit) {
                    // the winner's t and distance, once per segment (bvh_finish), then shading
                    Hit h = bvh_finish<FAST, SPHERES>(cur, P.scene.bvh, P.scene.geom, P.scene.n_surfaces, ps.o, ps.d, best,
                                             FAST ? 0.0f : __builtin_bit_cast(float, stack[BVH_STACK]));
                    if (h.idx < 0) h.dist = 0.0f;
                    if (shade_hit<FAST>(ps, P.scene, shade, h)) {
                        it.sum = it.sum + ps.col;            // buffer[u][v] += color, main.cpp:139
                        item_next_sample(it);
                        need_path = true;
                    }
                    have_hit = false;
                }
            }
            const bool want = need_path && it.left == 0;
            if (__any(want)) {
                if (want && it.have) item_close(it, P.slabs);
                const uint32_t got = pool_take(pool, want, P.queue_head, P.n_items, P.pool_batch);
                if (want) {
                    if (got == NO_ITEM) alive = false;       // queue empty: this lane retires
                    else item_open(it, got, P);
                }
            }
            if (waits && alive) {
                if (need_path) path_begin(ps, P.cam, item_stream(it), it.base);
                // the surfaces outside the hierarchy first, then the hierarchy
                Hit outside;
                if constexpr (FAST)
                    outside = closest_intersection_lean<false>(as_constant(P.scene.loose_lean), P.scene.n_scan, ps.o, ps.d);
                else
                    outside = closest_intersection<FAST, true>(as_constant(P.scene.loose_geom), P.scene.n_scan,
                                                               ps.o, ps.d, P.stats + 1);
                if (outside.idx >= 0) outside.idx = P.scene.loose[outside.idx];
                if constexpr (!FAST) stack[BVH_STACK] = __builtin_bit_cast(int, outside.t);
                bvh_begin<FAST, SPHERES>(cur, P.scene.bvh, ps.o, ps.d, outside);
                best = bvh_best_from<FAST>(outside);
                traversing = true;
                n_seg++;
            }
        }
        float4 q0, q1, q2, q3;
        bvh_fetch_records(P.scene.bvh.nodes, traversing ? bvh_record_of(P.scene.bvh, cur.node) : 0u, lds_wave,
                          q0, q1, q2, q3);
        if (traversing) {
            if constexpr (COUNT) {
                if (!cur.direct) { if (cur.node >= 0) n_nodes++; else n_leaves++; }
            }
            // (a ray that needs the reference's own scan skips the traversal)
            if (cur.direct || !bvh_advance<FAST, SPHERES>(cur, P.scene.bvh, q0, q1, q2, q3, ps.o, ps.d, best, stack, 1)) {
                traversing = false;
                have_hit = true;
            }
        }
    }
    atomicAdd(P.stats, n_seg);
    if constexpr (COUNT) {
        atomicAdd(P.stats + 3, (unsigned long long)n_nodes);
        atomicAdd(P.stats + 4, (unsigned long long)n_leaves);
    }
}

// accum[p] += slab_0[p] + slab_1[p] + ... in chunk order (n = n_pixels * 3 floats per slab)
__global__ void __launch_bounds__(BLOCK)
k_reduce_slabs(const float *slabs, uint32_t n_chunks, size_t n, float *accum)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float a = accum[i];
    for (uint32_t k = 0; k < n_chunks; k++) a += slabs[(size_t)k * n + i];
    accum[i] = a;
}

// ---------------------------------------------------------------- wavefront layout

// Queue-per-bounce layout: the state of every live path sits in HBM, 64 bytes
// per path as four 16-byte quarters (each read or written by one coalesced
// 1 KiB wave instruction); one kernel advances every live path by one segment
// and writes the survivors, compacted, to the other set of arrays.
//   a = (o.x, o.y, o.z, d.x)   b = (d.y, d.z, thr.x, thr.y)
//   c = (thr.z, col.x, col.y, col.z)   d = (rng key, rng counter, depth, slot)
// slot = sample_in_batch * n_pixels + pixel: where the path's radiance goes.
struct WfArrays { float4 *a, *b, *c; uint4 *d; };

// The arrays are cut into WF_SHARDS equal regions; workgroup g works on region
// g % WF_SHARDS and appends its survivors to the same region of the other set,
// through that region's own counter.  One counter for the whole queue would
// serialise every wave of the launch on one word; 64 counters on 64 different
// 128-byte lines take the adds in parallel.  A region never overflows: it
// receives at most as many paths as were read from it.
constexpr uint32_t WF_SHARDS = 64;
constexpr uint32_t WF_COUNT_STRIDE = 32;     // uint32 words between two counters (128 bytes)

struct WfParams {
    Camera cam;
    SceneView scene;
    uint32_t seed_mix;
    uint32_t first_sample;     // global number of the batch's first sample
    uint32_t n_pixels;
    uint32_t n_paths;          // generate: paths in the batch
    uint32_t shard_cap;        // paths a region can hold (a multiple of BLOCK)
    WfArrays in, out;
    const uint32_t *n_in;      // bounce: live paths per region of `in` (WF_COUNT_STRIDE apart)
    uint32_t *n_out;           // bounce: survivors per region of `out` (zeroed before the launch)
    float *result;             // radiance of finished paths, 3 floats per slot
    unsigned long long *stats; // [1] += segments re-evaluated directly
};

PT_DEV void wf_store(const WfArrays &w, uint32_t i, const PathState &ps, uint32_t slot)
{
    w.a[i] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
    w.b[i] = make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y);
    w.c[i] = make_float4(ps.thr.z, ps.col.x, ps.col.y, ps.col.z);
    w.d[i] = make_uint4(ps.rng.key, ps.rng.ctr, (uint32_t)ps.depth, slot);
}

// Path i of the batch starts in region (i / BLOCK) % WF_SHARDS, at the place
// the same workgroup index reads in the first bounce.
__global__ void __launch_bounds__(BLOCK)
k_wf_generate(WfParams P)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P.n_paths) return;
    const uint32_t s = i / P.n_pixels, pixel = i - s * P.n_pixels;
    PathState ps;
    path_begin(ps, P.cam, rng_path(rng_pixel(P.seed_mix, pixel), pixel, P.first_sample + s),
               pixel_base(P.cam, (int)(pixel % (uint32_t)P.cam.width), (int)(pixel / (uint32_t)P.cam.width)));
    const uint32_t shard = blockIdx.x % WF_SHARDS, local = blockIdx.x / WF_SHARDS;
    wf_store(P.in, shard * P.shard_cap + local * BLOCK + threadIdx.x, ps, i);
}

template <bool FAST, int KIND>
__global__ void __launch_bounds__(BLOCK)
k_wf_bounce(WfParams P)
{
    extern __shared__ __align__(16) float lds_shade[];
    const float *shade = P.scene.shade;
    if constexpr (KIND == SCENE_LDS) {
        for (int i = threadIdx.x; i < P.scene.n_surfaces * SHADE_STRIDE; i += BLOCK)
            lds_shade[i] = P.scene.shade[i];
        __syncthreads();
        shade = lds_shade;
    }
    const uint32_t shard = blockIdx.x % WF_SHARDS, local = blockIdx.x / WF_SHARDS;
    const uint32_t k = local * BLOCK + threadIdx.x;          // position within the region
    const bool live = k < P.n_in[shard * WF_COUNT_STRIDE];
    const uint32_t region = shard * P.shard_cap;
    bool survives = false;
    PathState ps;
    uint32_t slot = 0;
    if (live) {
        const uint32_t i = region + k;
        const float4 a = P.in.a[i], b = P.in.b[i], c = P.in.c[i];
        const uint4 d = P.in.d[i];
        ps.o = v3(a.x, a.y, a.z);
        ps.d = v3(a.w, b.x, b.y);
        ps.thr = v3(b.z, b.w, c.x);
        ps.col = v3(c.y, c.z, c.w);
        ps.rng.key = d.x;
        ps.rng.ctr = d.y;
        ps.depth = (int)d.z;
        slot = d.w;
        const bool ended = path_segment<FAST, KIND>(ps, P.scene, shade, lds_shade, P.stats + 1);
        if (ended) {
            float *r = P.result + 3ull * slot;
            r[0] = ps.col.x; r[1] = ps.col.y; r[2] = ps.col.z;
        }
        survives = !ended;
    }
    // compaction: one counter add per wave, survivors packed in lane order
    const unsigned long long mask = __ballot(survives);
    if (mask) {
        const uint32_t lane = __lane_id();
        const int first = __ffsll((long long)mask) - 1;
        uint32_t base = 0;
        if ((int)lane == first) base = atomicAdd(P.n_out + shard * WF_COUNT_STRIDE, (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, first);
        if (survives) wf_store(P.out, region + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)), ps, slot);
    }
}

// accum[p] += result of sample 0, 1, ... of the batch, in sample order (n = n_pixels * 3)
__global__ void __launch_bounds__(BLOCK)
k_wf_reduce(const float *result, uint32_t batch_spp, size_t n, float *accum)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float a = accum[i];
    for (uint32_t s = 0; s < batch_spp; s++) a += result[(size_t)s * n + i];
    accum[i] = a;
}

// to_ldr, src/main.cpp:146-148, on sum / float(TOTAL_SAMPLES) (main.cpp:160),
// then the Uint8 truncation of sdl_helpers.hpp:68-70.
__global__ void __launch_bounds__(BLOCK)
k_tonemap(const float *accum, size_t n, float total_samples, uint8_t *out)
{
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = accum[i] / total_samples;
    const float c = fminf(fmaxf(x, 0.0f), 1.0f);
    const float v = (float)((double)(powf(c, 1 / 2.2f) * 255) + 0.5);
    out[i] = (uint8_t)v;
}

}  // namespace ptd
