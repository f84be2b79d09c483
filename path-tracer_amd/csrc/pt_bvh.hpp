// Bounding-volume hierarchy over a scene's small triangles and spheres, built on the GPU
// (Morton codes, radix sort, Karras' parallel hierarchy, bottom-up refit) and
// traversed per lane with a private stack.  The reference has no acceleration
// structure (an "octree" is a to-do note, src/main.cpp:75); this one returns
// exactly what the reference's linear scan returns: the surface of smallest
// Intersection::distance, and among equal distances the one earliest in scene
// order (src/main.cpp:58-66).  Included by pt_kernels.hpp.
#pragma once

#include "pt_device.hpp"

namespace ptd {

// A node holds the boxes of its two children:
//   q[0] = (left min xyz, left ref)   q[1] = (left max xyz, right ref)
//   q[2] = (right min xyz, -)         q[3] = (right max xyz, -)
// A child ref >= 0 is an internal node, < 0 is leaf ~ref (position in Morton order).
constexpr int BVH_NODE_QUADS = 4;
constexpr int BVH_MAX_LEAVES = 1 << 25;          // nodes + leaves, 64 bytes each, within 4 GiB
constexpr float BVH_ORIGIN_ROUNDING = 8e-6f;     // of |o|: some 60 ulps
constexpr float BVH_GRAZING = 0.05f;             // some 8 eps / 1e-5
// A leaf is one surface: its geometry record (GEOM_STRIDE floats) with the surface's index in
// scene order in element 3 (where a geometry record keeps the kind) and the kind in element 11.
constexpr int BVH_LEAF_KIND = 11;
// Spheres (src/surface.hpp:91-112).  The reference forms d2 = L.L - tca^2 in fp32 and accepts
// d2 <= radius2, so a hit it reports can lie outside the sphere by the rounding of d2 - some
// 8 eps |L|^2 in d2, i.e. 4 eps |L|^2 / radius + 2 eps |L| in distance from the centre, L = centre -
// origin - which grows with the ray's distance, not with the sphere.  The boxes are not padded
// for it; every ray widens its box tests instead (BvhCursor::slack) by
// BVH_SPHERE_ROUNDING * (|o| + |c|max)^2 / radius_min, twice (both ends of an interval move) and
// with a factor of four in hand.
// The reference's test also takes the direction for a unit vector: with |d|^2 = 1 + e it forms
// d2 = L.L - (1 + e) tca^2, and what it accepts is the sphere only up to e |L|^2 / (2 radius) - for
// the directions a render produces (e of some 1e-6: a normalisation, the hardware sine) one more
// term of the same kind, which the ray adds from its own |d|; a ray with |e| above
// BVH_SPHERE_UNIT_TOLERANCE (only a caller's own rays: pt_intersect_rays) asks for something no box
// bounds, and is answered by the scan of every surface.
constexpr float BVH_SPHERE_ROUNDING = 4e-6f;
constexpr float BVH_SPHERE_UNIT_TOLERANCE = 1e-3f;
constexpr int BVH_STACK = 64;        // deepest possible tree: one level per key bit
// Where the per-lane traversal stacks live.  In LDS they cost 64 KiB per
// workgroup, which leaves two workgroups per CU; in private memory they cost
// write-through traffic but no occupancy.  Measured on the million-triangle
// scene: private 1.8x faster.  (Descending all lanes to their next leaf before
// testing leaves - two nested loops - was also measured, 14 % slower than the
// single loop below, in which every lane advances by one node or one leaf per
// iteration.)
constexpr bool BVH_STACK_IN_LDS = false;
// Also measured and not kept: folding two binary levels into four-wide nodes of
// 128 bytes with two-triangle leaves (half the dependent steps, twice the bytes
// per step, 81 VGPRs): 11 % slower on the same scene.

struct BvhView {
    const float4 *nodes;     // n_leaves - 1 internal nodes, root = 0, in the first n_leaves records
    const float *leaves;     // n_leaves * GEOM_STRIDE, directly behind the nodes' n_leaves records
    int n_leaves;            // 0: the scene has no hierarchy
    float ray_reach;         // the boxes allow for ray origins with |x| + |y| + |z| up to this
    float sphere_term;       // 0: no spheres in the hierarchy; else 1 / smallest radius among them
    float sphere_reach;      // largest |x| + |y| + |z| of a sphere centre in the hierarchy
};

// ---------------------------------------------------------------- build

// Sort key of a hierarchy triangle: Morton code of its centroid, BVH_CODE_BITS per axis,
// above its number among the hierarchy's triangles (so keys are unique and the number
// can be read back from a sorted key).
constexpr int BVH_INDEX_BITS = 25;                       // BVH_MAX_LEAVES = 1 << 25
constexpr int BVH_CODE_BITS = 13;                        // 3 * 13 + 25 = 64
static_assert((1 << BVH_INDEX_BITS) == BVH_MAX_LEAVES && 3 * BVH_CODE_BITS + BVH_INDEX_BITS <= 64, "key layout");

PT_DEV unsigned long long spread_bits(unsigned long long v)     // up to 21 bits -> every third bit
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

PT_DEV int bvh_key_index(unsigned long long key) { return (int)(key & ((1ull << BVH_INDEX_BITS) - 1ull)); }

// Per hierarchy triangle j (scene index tri_idx[j]): padded box, 64-bit sort key
// = Morton code of the centroid, then j (keys are unique).
// The pad covers the reference's tolerance outside the triangle (u, v >= -1e-5
// of an edge length), the rounding of hit points, and the rounding of o - v0 in
// the reference's test for ray origins up to ray_reach from the coordinate
// origin (a hit it reports can lie off the triangle by a few ulps of |o|).
__global__ void __launch_bounds__(256)
k_bvh_prepare(const float *geom, const int *tri_idx, int n, float3 cmin, float3 cinv, float ray_reach,
              unsigned long long *keys, float4 *box_min, float4 *box_max)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float *g = geom + (size_t)tri_idx[j] * GEOM_STRIDE;
    if (__float_as_int(g[3]) == GEOM_SPHERE) {
        // the sphere's own box, with room for the rounding of the radius (the record holds its
        // square) and of hit points; what the reference's rounding adds is the rays' business
        const V3 c = v3(g[0], g[1], g[2]);
        const float r = sqrtf(g[4]);
        const float reach = fmaxf(fabsf(c.x), fmaxf(fabsf(c.y), fabsf(c.z))) + r;
        const float pad = 1e-5f * (1.0f + reach) + 1e-5f * r + BVH_ORIGIN_ROUNDING * ray_reach;
        box_min[j] = make_float4(c.x - r - pad, c.y - r - pad, c.z - r - pad, 0.0f);
        box_max[j] = make_float4(c.x + r + pad, c.y + r + pad, c.z + r + pad, 0.0f);
        constexpr float CELLS_S = (float)(1 << BVH_CODE_BITS);
        const uint32_t sx = (uint32_t)fminf(fmaxf((c.x - cmin.x) * cinv.x * CELLS_S, 0.0f), CELLS_S - 1.0f);
        const uint32_t sy = (uint32_t)fminf(fmaxf((c.y - cmin.y) * cinv.y * CELLS_S, 0.0f), CELLS_S - 1.0f);
        const uint32_t sz = (uint32_t)fminf(fmaxf((c.z - cmin.z) * cinv.z * CELLS_S, 0.0f), CELLS_S - 1.0f);
        const unsigned long long code_s = (spread_bits(sx) << 2) | (spread_bits(sy) << 1) | spread_bits(sz);
        keys[j] = (code_s << BVH_INDEX_BITS) | (unsigned long long)j;
        return;
    }
    const V3 v0 = v3(g[0], g[1], g[2]);
    const V3 v1 = v0 + v3(g[4], g[5], g[6]), v2 = v0 + v3(g[8], g[9], g[10]);
    V3 lo = v3(fminf(v0.x, fminf(v1.x, v2.x)), fminf(v0.y, fminf(v1.y, v2.y)), fminf(v0.z, fminf(v1.z, v2.z)));
    V3 hi = v3(fmaxf(v0.x, fmaxf(v1.x, v2.x)), fmaxf(v0.y, fmaxf(v1.y, v2.y)), fmaxf(v0.z, fmaxf(v1.z, v2.z)));
    const float size = (hi.x - lo.x) + (hi.y - lo.y) + (hi.z - lo.z);
    const float reach = fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fmaxf(fabsf(lo.y), fabsf(hi.y)), fmaxf(fabsf(lo.z), fabsf(hi.z))));
    // ... and for rays nearly parallel to the triangle's plane: the reference accepts any
    // |determinant| above 1e-5, and its fp32 u and v then carry an error of some
    // eps |o - v0| |e1| |e2| / 1e-5 edge lengths, so a hit it reports can lie that far off the
    // triangle: BVH_GRAZING * (ray_reach + reach) * |e1| |e2| * longest edge (negligible for small
    // triangles).
    const V3 e1 = v3(g[4], g[5], g[6]), e2 = v3(g[8], g[9], g[10]), e3 = e2 - e1;
    const float l1 = sqrtf(Num<true>::dot(e1, e1)), l2 = sqrtf(Num<true>::dot(e2, e2)), l3 = sqrtf(Num<true>::dot(e3, e3));
    const float pad = 1e-3f * size + 1e-5f * (1.0f + reach) + BVH_ORIGIN_ROUNDING * ray_reach +
                      BVH_GRAZING * (ray_reach + reach) * l1 * l2 * fmaxf(l1, fmaxf(l2, l3));
    box_min[j] = make_float4(lo.x - pad, lo.y - pad, lo.z - pad, 0.0f);
    box_max[j] = make_float4(hi.x + pad, hi.y + pad, hi.z + pad, 0.0f);
    const float cx = ((lo.x + hi.x) * 0.5f - cmin.x) * cinv.x;
    const float cy = ((lo.y + hi.y) * 0.5f - cmin.y) * cinv.y;
    const float cz = ((lo.z + hi.z) * 0.5f - cmin.z) * cinv.z;
    constexpr float CELLS = (float)(1 << BVH_CODE_BITS);
    const uint32_t qx = (uint32_t)fminf(fmaxf(cx * CELLS, 0.0f), CELLS - 1.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf(cy * CELLS, 0.0f), CELLS - 1.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf(cz * CELLS, 0.0f), CELLS - 1.0f);
    const unsigned long long code = (spread_bits(qx) << 2) | (spread_bits(qy) << 1) | spread_bits(qz);
    keys[j] = (code << BVH_INDEX_BITS) | (unsigned long long)j;
}

// Length of the common prefix of sorted keys i and j; -1 outside the array.
PT_DEV int bvh_delta(const unsigned long long *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// One thread per internal node: its key range, its split, its children
// (Karras, "Maximizing parallelism in the construction of BVHs...", 2012).
// child >= 0: internal node; child < 0: leaf ~child.
__global__ void __launch_bounds__(256)
k_bvh_hierarchy(const unsigned long long *keys, int n, int2 *children, int *parent_of_node, int *parent_of_leaf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = bvh_delta(keys, n, i, i + 1) - bvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = bvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (bvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (bvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = bvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (bvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    int2 c;
    if (lo == gamma) { c.x = ~gamma; parent_of_leaf[gamma] = i; }
    else { c.x = gamma; parent_of_node[gamma] = i; }
    if (hi == gamma + 1) { c.y = ~(gamma + 1); parent_of_leaf[gamma + 1] = i; }
    else { c.y = gamma + 1; parent_of_node[gamma + 1] = i; }
    children[i] = c;
    if (i == 0) parent_of_node[0] = -1;
}

// Bottom-up boxes, one launch per level.  One thread per internal node; in launch number
// `pass` (1, 2, ...) a node whose two children were complete before this launch (leaves
// always are; done[child] in 1..pass-1) forms its children's boxes, writes its record and
// marks itself done in this pass.  Launches on one stream are ordered and each sees the
// memory of the one before, so nothing is handed over inside a launch: no fences, no
// atomics.  (The one-launch form, in which the second child to arrive at a node carries on
// upwards, needs two agent-scope fences per node visit; on this GPU such a fence writes an
// XCD's L2 back, and that form took 7.4 ms for a million triangles against 0.4 ms for some
// 25 of these launches.)  done[] is zeroed before the first pass; the root is node 0.
PT_DEV void bvh_child_box(int child, const unsigned long long *keys, const float4 *box_min, const float4 *box_max,
                          const float4 *nodes, float4 &lo, float4 &hi)
{
    if (child < 0) {
        const int j = bvh_key_index(keys[~child]);
        lo = box_min[j];
        hi = box_max[j];
    } else {
        const float4 *q = nodes + (size_t)child * BVH_NODE_QUADS;
        const float4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3];
        lo = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), 0.0f);
        hi = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), 0.0f);
    }
}

__global__ void __launch_bounds__(256)
k_bvh_refit_pass(const unsigned long long *keys, int n, const int2 *children, const float4 *box_min,
                 const float4 *box_max, int pass, int *done, float4 *nodes)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1 || done[i] != 0) return;
    const int2 c = children[i];
    const int dl = c.x < 0 ? 1 : done[c.x], dr = c.y < 0 ? 1 : done[c.y];
    // a child finished in THIS launch does not count: its record may not have landed yet
    const bool left_ready = c.x < 0 || (dl != 0 && dl < pass), right_ready = c.y < 0 || (dr != 0 && dr < pass);
    if (!(left_ready && right_ready)) return;
    float4 llo, lhi, rlo, rhi;
    bvh_child_box(c.x, keys, box_min, box_max, nodes, llo, lhi);
    bvh_child_box(c.y, keys, box_min, box_max, nodes, rlo, rhi);
    float4 *q = nodes + (size_t)i * BVH_NODE_QUADS;
    // the refs travel in the w components of the first two quads
    q[0] = make_float4(llo.x, llo.y, llo.z, __int_as_float(c.x));
    q[1] = make_float4(lhi.x, lhi.y, lhi.z, __int_as_float(c.y));
    q[2] = make_float4(rlo.x, rlo.y, rlo.z, 0.0f);
    q[3] = make_float4(rhi.x, rhi.y, rhi.z, 0.0f);
    done[i] = pass;
}

// Leaf records in Morton order: the triangle's geometry record with the scene
// index in place of the kind.
__global__ void __launch_bounds__(256)
k_bvh_leaves(const unsigned long long *keys, int n, const float *geom, const int *tri_idx, float *leaves)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int i = tri_idx[bvh_key_index(keys[k])];
    const float4 *src = reinterpret_cast<const float4 *>(geom + (size_t)i * GEOM_STRIDE);
    float4 *dst = reinterpret_cast<float4 *>(leaves + (size_t)k * GEOM_STRIDE);
    float4 a = src[0], c = src[2];
    c.w = a.w;                       // the kind moves to BVH_LEAF_KIND
    a.w = __int_as_float(i);
    dst[0] = a; dst[1] = src[1]; dst[2] = c; dst[3] = src[3];
}

// The geometry records of the surfaces outside the hierarchy, copied out in scan order,
// with the run lengths of that order (GEOM_RUN).  One thread per record.
__global__ void __launch_bounds__(256)
k_bvh_gather_loose(const float *geom, const int *loose, const int *runs, int n, float *out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float4 *src = reinterpret_cast<const float4 *>(geom + (size_t)loose[j] * GEOM_STRIDE);
    float4 *dst = reinterpret_cast<float4 *>(out + (size_t)j * GEOM_STRIDE);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    dst[3] = make_float4(__int_as_float(runs[j]), 0.0f, 0.0f, 0.0f);
}

// ---------------------------------------------------------------- traversal

// Entry parameter of the ray into a box, or a negative value if it misses or
// the box lies wholly beyond t_far.  inv = 1/d per component; slack widens the
// test by the ray's own rounding (BvhCursor::slack).  (Slab crossings as one fused
// b * inv - o * inv each: 1 % faster, and wrong where b * inv overflows but (b - o) * inv does
// not; not kept.)
PT_DEV float box_entry(float4 bmin, float4 bmax, V3 o, V3 inv, float t_far, float slack)
{
    const float x1 = (bmin.x - o.x) * inv.x, x2 = (bmax.x - o.x) * inv.x;
    const float y1 = (bmin.y - o.y) * inv.y, y2 = (bmax.y - o.y) * inv.y;
    const float z1 = (bmin.z - o.z) * inv.z, z2 = (bmax.z - o.z) * inv.z;
    // fminf/fmaxf drop a NaN operand (0 * inf when the ray lies in a box face)
    const float t_in = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.0f));
    const float t_out = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), t_far));
    return t_in <= t_out + slack ? t_in : -1.0f;
}

// The same against the box grown by `pad` on every side (a distance, the ray's own: hierarchies
// that hold spheres, BVH_SPHERE_ROUNDING): every slab crossing moves by pad / |d_k|; an axis the
// ray is parallel to (1/d_k infinite) keeps its crossings, which are infinite already.
PT_DEV float box_entry_padded(float4 bmin, float4 bmax, V3 o, V3 inv, float t_far, float pad)
{
    const float INF = __builtin_inff();
    const float ix = __builtin_fabsf(inv.x), iy = __builtin_fabsf(inv.y), iz = __builtin_fabsf(inv.z);
    const float px = ix < INF ? pad * ix : 0.0f, py = iy < INF ? pad * iy : 0.0f, pz = iz < INF ? pad * iz : 0.0f;
    const float x1 = (bmin.x - o.x) * inv.x, x2 = (bmax.x - o.x) * inv.x;
    const float y1 = (bmin.y - o.y) * inv.y, y2 = (bmax.y - o.y) * inv.y;
    const float z1 = (bmin.z - o.z) * inv.z, z2 = (bmax.z - o.z) * inv.z;
    const float t_in = fmaxf(fmaxf(fminf(x1, x2) - px, fminf(y1, y2) - py), fmaxf(fminf(z1, z2) - pz, 0.0f));
    const float t_out = fminf(fminf(fmaxf(x1, x2) + px, fmaxf(y1, y2) + py), fminf(fmaxf(z1, z2) + pz, t_far));
    return t_in <= t_out ? t_in : -1.0f;
}

// A lane's traversal in progress.
struct BvhCursor {
    V3 inv;              // 1/d per component
    float len_d, inv_len;   // |d|; ray parameter per unit of distance (an upper bound of it: bvh_begin)
    float slack;         // what the box tests allow for this ray beyond the boxes' own padding: in a hierarchy
                         // of triangles a ray parameter (what the origin's rounding can move a slab crossing,
                         // rays from beyond ray_reach only); in one that holds spheres a distance by which
                         // every box is grown (bvh_begin)
    int node;            // >= 0: internal node to open; < 0: leaf ~node to test
    int sp;              // entries on the stack
    int top;             // the newest entry, kept in a register: a pop then costs no memory
                         // round trip (the entry below it is fetched while the popped node is)
    bool direct;         // the answer must come from the reference's scan over every surface
};

// The best candidate of a traversal in progress.
//   ref >= 0: a surface outside the hierarchy, by scene index;
//   ref <  0: leaf ~ref (position in Morton order);  BVH_NO_HIT: none yet.
// EXACT: [lo, hi] brackets the distance the reference computes for it; lo == hi
// when that distance is known exactly (every candidate from outside the
// hierarchy, and leaves that had to be evaluated exactly).  A leaf is evaluated
// exactly - three IEEE divisions and a root - only when the error-band filter
// cannot decide; the winner's t and distance are computed once, in bvh_finish.
// FAST: ref is always a scene index, lo the distance and hi the ray parameter.
struct BvhBest { int ref; float lo, hi; };
constexpr int BVH_NO_HIT = (int)0x80000000;

template <bool FAST>
PT_DEV BvhBest bvh_best_from(const Hit &outside)
{
    const float INF = __builtin_inff();
    if (outside.idx < 0) return {BVH_NO_HIT, INF, INF};
    return {outside.idx, outside.dist, FAST ? outside.t : outside.dist};
}

// The reference keeps the first hit it finds whatever its distance, replaces it
// only by a strictly smaller one, and a distance that is not a number compares
// false both ways (src/main.cpp:61): with such distances the answer depends on
// the order in which the surfaces are visited, which a hierarchy does not keep.
// So a ray whose origin is not finite or whose direction's length leaves the
// normal range, a hit outside the hierarchy (`outside`) whose distance is not a
// number, and (in bvh_test_leaf) a leaf hit whose distance is not finite all set
// c.direct, and bvh_finish then scans every surface in scene order instead.
template <bool FAST, bool SPHERES = true>
PT_DEV void bvh_begin(BvhCursor &c, const BvhView &bvh, V3 o, V3 d, const Hit &outside)
{
    c.inv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    c.len_d = __builtin_amdgcn_sqrtf(Num<FAST>::dot(d, d));
    c.inv_len = __builtin_amdgcn_rcpf(c.len_d);
    // A triangle's distance is |t d| (src/surface.hpp:70), a sphere's is t itself whatever the
    // direction's length (surface.hpp:111): with spheres in the hierarchy a box entered at
    // parameter t may hold a hit at distance t or t |d|, and only the smaller can be used to prune.
    if (SPHERES && bvh.sphere_term > 0.0f) c.inv_len = fmaxf(c.inv_len, 1.0f);
    c.sp = 0;
    c.top = 0;
    c.node = bvh.n_leaves > 1 ? 0 : ~0;
    c.direct = false;
    c.slack = 0.0f;
    if constexpr (!FAST) {
        const float INF = __builtin_inff();
        const float reach = __builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z);
        c.direct = !(c.len_d >= 1e-18f && c.len_d <= 1e18f) | !(reach <= 1e18f) |
                   ((outside.idx >= 0) & !(__builtin_fabsf(outside.dist) < INF));
        // The boxes are padded for origins up to bvh.ray_reach away.  A ray from
        // farther out widens every box test instead: a slab crossing (b - o) / d_k
        // moves by the rounding of o over |d_k|; an axis the ray is parallel to
        // (1/d_k infinite) has no crossing to move.  (One slack for all axes: such
        // rays are slow, not wrong.)
        const bool far_out = reach > bvh.ray_reach;
        if (SPHERES && bvh.sphere_term > 0.0f) {            // the same for every ray of a launch
            // spheres in the hierarchy: every box grows by what the reference's rounding can add to
            // a sphere for this ray (BVH_SPHERE_ROUNDING)
            const float far = reach + bvh.sphere_reach;
            const float off_unit = __builtin_fabsf(__builtin_fmaf(c.len_d, c.len_d, -1.0f));
            c.direct = c.direct | !(off_unit <= BVH_SPHERE_UNIT_TOLERANCE);
            c.slack = (far_out ? BVH_ORIGIN_ROUNDING * reach : 0.0f) +
                      (BVH_SPHERE_ROUNDING + 2.0f * off_unit) * bvh.sphere_term * far * far;
        } else if (far_out) {
            const float ix = __builtin_fabsf(c.inv.x), iy = __builtin_fabsf(c.inv.y), iz = __builtin_fabsf(c.inv.z);
            const float steepest = fmaxf(fmaxf(ix < INF ? ix : 0.0f, iy < INF ? iy : 0.0f), iz < INF ? iz : 0.0f);
            c.slack = BVH_ORIGIN_ROUNDING * reach * steepest;
        }
    }
}

// The exact evaluation of one leaf triangle (record quads g0..g2), as
// Triangle::intersects does it (src/surface.hpp:53-70).
PT_DEV bool bvh_leaf_exact(const TriDets &q, V3 d, float &t, float &dist)
{
    bool ok = !(q.det_a >= -EPS_F && q.det_a <= EPS_F);
    const float ue = q.du / q.det_a;
    ok = ok & !(ue < -EPS_F || ue >= 1.0f);
    const float ve = q.dv / q.det_a;
    ok = ok & !(ve < -EPS_F || ue + ve >= 1.0f);
    t = q.dt / q.det_a;
    ok = ok & !(t <= EPS_F);
    const V3 td = d * t;
    dist = __builtin_sqrtf(Num<false>::dot(td, td));
    return ok;
}

PT_DEV TriDets bvh_leaf_dets(float4 g0, float4 g1, float4 g2, V3 o, V3 d)
{
    return triangle_dets_v<false>(v3(g0.x, g0.y, g0.z), v3(g1.x, g1.y, g1.z), g1.w, v3(g2.x, g2.y, g2.z), o, -d);
}

// The exact evaluation of a leaf sphere (centre g0.xyz, radius2 g1.x), as Sphere::intersects
// does it (src/surface.hpp:91-112).
PT_DEV bool bvh_sphere_exact(float4 g0, float4 g1, V3 o, V3 d, float &t, float &dist)
{
    float tca, arg;
    bool ok = sphere_pre_v<false>(v3(g0.x, g0.y, g0.z), g1.x, o, d, tca, arg);
    const float thc = __builtin_sqrtf(arg);
    t = tca - thc;
    if (t <= EPS_F) {
        t = tca + thc;
        ok = ok & !(t <= EPS_F);
    }
    dist = t;
    return ok;
}

// Leaf `leaf` (position in Morton order) evaluated exactly, whichever kind it has.
template <bool SPHERES = true>
PT_DEV bool bvh_leaf_exact_any(const BvhView &bvh, int leaf, V3 o, V3 d, int &scene_index, float &t, float &dist)
{
    const float4 *g = reinterpret_cast<const float4 *>(bvh.leaves + (size_t)leaf * GEOM_STRIDE);
    const float4 g0 = g[0], g1 = g[1], g2 = g[2];
    scene_index = __float_as_int(g0.w);
    if (SPHERES && __float_as_int(g2.w) == GEOM_SPHERE) return bvh_sphere_exact(g0, g1, o, d, t, dist);
    return bvh_leaf_exact(bvh_leaf_dets(g0, g1, g2, o, d), d, t, dist);
}

// After the traversal: the closest hit overall, with the reference's t and
// distance (idx < 0: none).  geom, n_surfaces: every surface's geometry record.
// outside_t: the ray parameter of the hit outside the hierarchy the traversal
// began with (bvh_best_from), which is the answer when no leaf came closer.
template <bool FAST, bool SPHERES = true>
PT_DEV Hit bvh_finish(const BvhCursor &c, const BvhView &bvh, const float *geom, int n_surfaces, V3 o, V3 d,
                      const BvhBest &best, float outside_t)
{
    Hit h = {-1, 0.0f, 0.0f};
    if constexpr (FAST) {
        if (best.ref != BVH_NO_HIT) h = {best.ref, best.hi, best.lo};
        return h;
    } else {
        if (c.direct) return closest_direct<false>(as_constant(geom), n_surfaces, o, d);
        if (best.ref == BVH_NO_HIT) return h;
        if (best.ref >= 0) {
            // a surface outside the hierarchy won: the scan that found it has its exact t and distance
            h = {best.ref, outside_t, best.lo};
        } else {
            bvh_leaf_exact_any<SPHERES>(bvh, ~best.ref, o, d, h.idx, h.t, h.dist);
        }
        return h;
    }
}

// Opens an internal node (its four quads): true when the traversal descends
// into a child (c.node), false when neither child can hold a closer hit.
PT_DEV bool bvh_open_node(BvhCursor &c, bool padded, float4 q0, float4 q1, float4 q2, float4 q3, V3 o, float bound,
                          int *stack, int stride)
{
    // nothing beyond `bound`, an upper bound of the best distance, can win; the margin
    // covers the rounding of t * |d|
    const float t_far = bound * c.inv_len * (1.0f + 1e-5f);
    float tl, tr;
    if (padded) {                                   // wave-uniform: the hierarchy holds spheres
        tl = box_entry_padded(q0, q1, o, c.inv, t_far, c.slack);
        tr = box_entry_padded(q2, q3, o, c.inv, t_far, c.slack);
    } else {
        tl = box_entry(q0, q1, o, c.inv, t_far, c.slack);
        tr = box_entry(q2, q3, o, c.inv, t_far, c.slack);
    }
    const int left = __float_as_int(q0.w), right = __float_as_int(q1.w);
    if (tl >= 0.0f && tr >= 0.0f) {                 // both: nearer first
        const bool left_first = tl <= tr;
        if (c.sp > 0) stack[(c.sp - 1) * stride] = c.top;
        c.top = left_first ? right : left;
        c.sp++;
        c.node = left_first ? left : right;
        return true;
    }
    if (tl >= 0.0f) { c.node = left; return true; }
    if (tr >= 0.0f) { c.node = right; return true; }
    return false;
}

// Tests leaf ~c.node (the first three quads of its record) against the best candidate so far.
// SPHERES: the hierarchy may hold spheres (false: the render kernel compiled for hierarchies of
// triangles only, without the sphere's code and the padded box test in its loop - 8 % faster
// on the million-triangle field than the general one).
template <bool FAST, bool SPHERES = true>
PT_DEV void bvh_test_leaf(BvhCursor &c, const BvhView &bvh, float4 g0, float4 g1, float4 g2, V3 o, V3 d, BvhBest &best)
{
    const int idx = __float_as_int(g0.w);
    const bool is_sphere = SPHERES && __float_as_int(g2.w) == GEOM_SPHERE;
    // the same bands as the scan's filter (scan_surface): rej = surely no hit, acc = surely a hit,
    // [c_lo, c_hi] brackets the distance the reference computes (FAST: c_lo = the distance, c_hi = t)
    bool rej, acc;
    float c_lo, c_hi;
    if (is_sphere) {
        float tca, d2, arg;
        sphere_pre_terms<FAST>(v3(g0.x, g0.y, g0.z), g1.x, o, d, tca, d2, arg);
        const bool pre_ok = !(tca < 0.0f) & !(d2 > g1.x);                       // surface.hpp:94, :98
        const float thc = __builtin_amdgcn_sqrtf(arg);
        const float t1 = tca - thc, t2 = tca + thc;
        if constexpr (FAST) {
            const float t = !(t1 <= EPS_F) ? t1 : t2;
            acc = pre_ok & (t > EPS_F);
            rej = !acc;
            c_lo = c_hi = t;
        } else {
            const float e1 = SPH_DIST_ERR * (thc + __builtin_fabsf(t1)), e2 = SPH_DIST_ERR * (thc + __builtin_fabsf(t2));
            const float lo1 = t1 - e1, hi1 = t1 + e1, lo2 = t2 - e2, hi2 = t2 + e2;
            const bool near_ok = lo1 > BAND_T_ACC, near_bad = hi1 < BAND_T_REJ;
            const bool far_ok = lo2 > BAND_T_ACC, far_bad = hi2 < BAND_T_REJ;
            acc = pre_ok & (near_ok | (near_bad & far_ok));
            rej = !pre_ok | (near_bad & far_bad);
            // the bracket also serves a leaf that is neither accepted nor rejected yet: the near
            // root unless it is surely out, the far root unless the near one is surely in
            c_lo = near_bad ? lo2 : lo1;
            c_hi = near_ok ? hi1 : hi2;
        }
    } else {
        const TriDets q = triangle_dets_v<FAST>(v3(g0.x, g0.y, g0.z), v3(g1.x, g1.y, g1.z), g1.w,
                                                v3(g2.x, g2.y, g2.z), o, -d);
        const float rinv = __builtin_amdgcn_rcpf(q.det_a);
        const float u = q.du * rinv, v = q.dv * rinv, t = q.dt * rinv;
        const float s = u + v;
        const bool det_bad = __builtin_fabsf(q.det_a) <= EPS_F;
        const float dist = t * c.len_d;
        if constexpr (FAST) {
            // (t by a positive compare: a NaN is no hit)
            acc = !(det_bad | (u < -EPS_F) | (u >= 1.0f) | (v < -EPS_F) | (s >= 1.0f)) & (t > EPS_F);
            rej = !acc;
            c_lo = dist;
            c_hi = t;
        } else {
            const float low_uv = __builtin_fminf(u, v);
            rej = det_bad | (low_uv < BAND_LO_REJ) | (__builtin_fmaxf(u, s) > BAND_ONE_REJ) | (t < BAND_T_REJ);
            acc = !det_bad & (low_uv > BAND_LO_ACC) & (u < BAND_ONE_ACC) & (s < BAND_ONE_ACC) & (t > BAND_T_ACC);
            c_lo = dist * (1.0f - TRI_DIST_ERR);
            c_hi = dist * (1.0f + TRI_DIST_ERR);
        }
    }
    if constexpr (FAST) {
        if (acc & (c_lo < best.lo)) best = {idx, c_lo, c_hi};
    } else {
        // surely no hit, or (if a hit at all) surely farther than the best
        if (rej | (c_lo > best.hi)) return;
        if (acc & (c_hi < best.lo)) {               // surely a hit and surely closer
            best = {c.node, c_lo, c_hi};
            return;
        }
        // Rare: a value inside a band, or distances too close to tell apart.  Both
        // candidates exactly; equal distances go to the earlier surface in scene order.
        float te, dist_e;
        const bool ok = is_sphere ? bvh_sphere_exact(g0, g1, o, d, te, dist_e)
                                  : bvh_leaf_exact(bvh_leaf_dets(g0, g1, g2, o, d), d, te, dist_e);
        c.direct = c.direct | (ok & !(dist_e < __builtin_inff()));
        if (!ok) return;
        int best_idx = best.ref;
        if (best.ref < 0 && best.ref != BVH_NO_HIT) {
            float tb, dist_b;
            bvh_leaf_exact_any<SPHERES>(bvh, ~best.ref, o, d, best_idx, tb, dist_b);
            if (best.lo != best.hi) best.lo = best.hi = dist_b;
        }
        if ((dist_e < best.lo) | ((dist_e == best.lo) & (idx < best_idx) & (best.ref != BVH_NO_HIT)))
            best = {c.node, dist_e, dist_e};
    }
}

// Takes the next node off the stack; false when the traversal is complete.
PT_DEV bool bvh_pop(BvhCursor &c, const int *stack, int stride)
{
    if (c.sp == 0) return false;
    c.sp--;
    c.node = c.top;
    if (c.sp > 0) c.top = stack[(c.sp - 1) * stride];
    return true;
}

// Advances a traversal by one node or one leaf, given the record c.node names.
template <bool FAST, bool SPHERES = true>
PT_DEV bool bvh_advance(BvhCursor &c, const BvhView &bvh, float4 q0, float4 q1, float4 q2, float4 q3, V3 o, V3 d,
                        BvhBest &best, int *stack, int stride)
{
    if (c.node >= 0) {
        if (bvh_open_node(c, SPHERES && !FAST && bvh.sphere_term > 0.0f, q0, q1, q2, q3, o, FAST ? best.lo : best.hi, stack, stride))
            return true;
    } else {
        bvh_test_leaf<FAST, SPHERES>(c, bvh, q0, q1, q2, o, d, best);
    }
    return bvh_pop(c, stack, stride);
}

// Advances a traversal by one node or one leaf.  Returns false when the
// traversal is complete.  stack: BVH_STACK ints per lane, lane-strided
// (stack[level * stride]).
template <bool FAST, bool SPHERES = true>
PT_DEV bool bvh_step(BvhCursor &c, const BvhView &bvh, V3 o, V3 d, BvhBest &best, int *stack, int stride)
{
    if (c.node >= 0) {
        const float4 *q = bvh.nodes + (size_t)c.node * BVH_NODE_QUADS;
        if (bvh_open_node(c, SPHERES && !FAST && bvh.sphere_term > 0.0f, q[0], q[1], q[2], q[3], o, FAST ? best.lo : best.hi, stack, stride))
            return true;
    } else {
        const float4 *g = reinterpret_cast<const float4 *>(bvh.leaves + (size_t)(~c.node) * GEOM_STRIDE);
        bvh_test_leaf<FAST, SPHERES>(c, bvh, g[0], g[1], g[2], o, d, best);
    }
    return bvh_pop(c, stack, stride);
}

// ---- records fetched by the whole wave together ------------------------------
//
// A lane's traversal step reads one 64-byte record (node or leaf) at an address
// of its own.  Read lane by lane, as four 16-byte loads, that is four cache-line
// look-ups per lane and step in the vector L1, which serves one look-up per
// cycle: measured on the million-triangle scene, the L1 was busy for 79 % of
// the kernel's cycles.  Here the four lanes of a quad read the four quarters of
// one record in one instruction - one look-up - for each of the quad's lanes in
// turn, straight into LDS (global_load_lds_dwordx4 places lane l's 16 bytes at
// base + 16 l), and every lane then reads its own record back from LDS.
// Measured: L1 look-ups per launch 6.4e9 -> 3.4e9, kernel time -4 % (what
// remains is the latency of dependent fetches at 6 waves per SIMD).
//
// LDS per wave: four slabs, one per instruction, of 64 lanes x 16 bytes, each
// followed by 16 bytes of skew, so that the four read-backs (lane m reads slab
// m & 3, quad m >> 2) spread over all banks.
constexpr int BVH_FETCH_SLAB = 1024 + 16;
constexpr int BVH_FETCH_LDS = 4 * BVH_FETCH_SLAB;      // bytes per wave

// Index of the record c.node names, in an array that holds the nodes, then the
// leaves (the hierarchy is allocated that way).
PT_DEV uint32_t bvh_record_of(const BvhView &bvh, int node)
{
    return node >= 0 ? (uint32_t)node : (uint32_t)bvh.n_leaves + (uint32_t)~node;
}

// Must be called by all 64 lanes of a wave together.  rec: this lane's record
// index (any valid index for a lane that needs none).  lds_wave: the wave's
// BVH_FETCH_LDS bytes.
template <int K>
PT_DEV void bvh_fetch_issue(const float4 *records, uint32_t rec, char *lds_wave, uint32_t quarter)
{
    typedef const __attribute__((address_space(1))) void *global_p;
    typedef __attribute__((address_space(3))) void *lds_p;
    // quad_perm [K,K,K,K]: every lane of a quad gets the value of the quad's lane K
    const uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)rec, K * 0x55, 0xf, 0xf, true);
    // a 32-bit byte offset from the (scalar) base: the hierarchy holds at most BVH_MAX_LEAVES
    const uint32_t offset = r * (uint32_t)(BVH_NODE_QUADS * sizeof(float4)) + quarter * (uint32_t)sizeof(float4);
    const char *src = reinterpret_cast<const char *>(records) + offset;
    __builtin_amdgcn_global_load_lds((global_p)(uintptr_t)src, (lds_p)(uintptr_t)(lds_wave + K * BVH_FETCH_SLAB), 16, 0, 0);
}

PT_DEV void bvh_fetch_records(const float4 *records, uint32_t rec, char *lds_wave,
                              float4 &q0, float4 &q1, float4 &q2, float4 &q3)
{
    const uint32_t lane = __lane_id();
    const uint32_t quarter = lane & 3u;
    bvh_fetch_issue<0>(records, rec, lds_wave, quarter);
    bvh_fetch_issue<1>(records, rec, lds_wave, quarter);
    bvh_fetch_issue<2>(records, rec, lds_wave, quarter);
    bvh_fetch_issue<3>(records, rec, lds_wave, quarter);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float4 *mine = reinterpret_cast<const float4 *>(lds_wave + quarter * BVH_FETCH_SLAB + (lane >> 2) * 64);
    q0 = mine[0]; q1 = mine[1]; q2 = mine[2]; q3 = mine[3];
}

// Continues a closest-hit search through the hierarchy.  On entry `hit` is
// the exact closest hit among the surfaces outside the hierarchy (idx < 0:
// none); on exit it is the closest hit overall, with the reference's t and
// distance (idx < 0: none).
template <bool FAST, bool SPHERES>
PT_DEV void closest_in_bvh_of(const BvhView &bvh, const float *geom, int n_surfaces, V3 o, V3 d, Hit &hit,
                              int *stack, int stride)
{
    BvhCursor c;
    bvh_begin<FAST, SPHERES>(c, bvh, o, d, hit);
    BvhBest best = bvh_best_from<FAST>(hit);
    const float outside_t = hit.t;
    if (!c.direct)
        while (bvh_step<FAST, SPHERES>(c, bvh, o, d, best, stack, stride)) {}
    hit = bvh_finish<FAST, SPHERES>(c, bvh, geom, n_surfaces, o, d, best, outside_t);
}

template <bool FAST>
PT_DEV void closest_in_bvh(const BvhView &bvh, const float *geom, int n_surfaces, V3 o, V3 d, Hit &hit,
                           int *stack, int stride)
{
    // the same for every ray of a launch: the traversal with or without the sphere's code in its loop
    if (bvh.sphere_term > 0.0f) closest_in_bvh_of<FAST, true>(bvh, geom, n_surfaces, o, d, hit, stack, stride);
    else closest_in_bvh_of<FAST, false>(bvh, geom, n_surfaces, o, d, hit, stack, stride);
}

}  // namespace ptd
