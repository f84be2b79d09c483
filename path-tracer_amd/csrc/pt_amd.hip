// C ABI of the MI355X-native render step (declared in include/pt_amd.h):
// context, scene flattening, launches.  Build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-slp-vectorize -fPIC -shared
#include "../../include/pt_amd.h"
#include "pt_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstring>   // rocprim uses memset without including it
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace ptd;

// Device memory for pt_render_into_static, the entry point that allocates nothing: the
// work queue's words in the first STATIC_HEADER_BYTES, the work items' partial sums after
// them.  Every GPU the library runs on holds its own copy, at its own address, until the
// process ends.  Host code never uses the array's name as an address (on the host it is
// a stand-in for the symbol): pt_create asks the runtime for the device address once per
// device and everything below goes through that pointer.
constexpr size_t STATIC_SCRATCH_BYTES = PT_STATIC_SCRATCH_BYTES;
constexpr size_t STATIC_HEADER_BYTES = 256;
constexpr size_t QUEUE_BYTES = 64;          // the work queue's head and the launch's counters (RenderParams::stats)
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {

thread_local std::string g_error;

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES];      // per device: where its copy of `scratch` lives

struct HostV3 { float x, y, z; };
inline HostV3 hsub(HostV3 a, HostV3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float hdot(HostV3 a, HostV3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline HostV3 hcross(HostV3 x, HostV3 y)
{
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline HostV3 hnormalize(HostV3 a)
{
    const float inv = 1.0f / std::sqrt(hdot(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}

// A triangle outside the error-band filter's range (GEOM_DIRECT_EDGE_PRODUCT).
bool extreme_triangle(const pt_surface &s)
{
    float m1 = 0, m2 = 0;
    for (int k = 0; k < 3; k++) {
        m1 = std::max(m1, std::fabs(s.b[k] - s.a[k]));
        m2 = std::max(m2, std::fabs(s.c[k] - s.a[k]));
    }
    return !(m1 * m2 <= GEOM_DIRECT_EDGE_PRODUCT);      // also when not a number
}

// The constants of the kernels' fast_div for the divisor d (1 <= d < 2^31).
FastDiv fast_div_by(uint32_t d)
{
    uint32_t s = 0;
    while ((1ull << s) < d) s++;
    const unsigned long long two = 1ull << (31 + s);
    const unsigned long long m = (two + d - 1) / d;          // <= 2^32; exactly 2^31 for a power of two
    return FastDiv{(uint32_t)std::min<unsigned long long>(m, 0xFFFFFFFFull), s > 0 ? s - 1 : 0, d};
}

uint32_t host_pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
uint32_t host_seed_mix(uint64_t seed)
{
    return host_pcg(host_pcg((uint32_t)seed) ^ (uint32_t)(seed >> 32));
}

}  // namespace

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    bool have_scene = false, have_camera = false;
    int n_surfaces = 0, n_tri = 0, n_sph = 0;
    float bg[3] = {0, 0, 0};
    Camera cam{};
    float *d_geom = nullptr, *d_shade = nullptr, *d_lean = nullptr;

    // hierarchy over the small triangles (n_bvh == 0: none) and the surfaces left outside it
    int accel = PT_ACCEL_AUTO;
    int max_depth = PT_MAX_DEPTH;    // segments per path at most (pt_set_max_depth)
    int n_bvh = 0, n_loose = 0;
    float bvh_ray_reach = 0.0f;
    float bvh_sphere_term = 0.0f, bvh_sphere_reach = 0.0f;     // BvhView
    int *d_loose = nullptr;
    float *d_loose_geom = nullptr;
    float *d_loose_lean = nullptr;      // the FAST policy's records of the same surfaces, same order
    std::vector<pt_surface> host_surfaces;      // kept to rebuild when pt_set_accel changes
    float4 *d_bvh_nodes = nullptr;
    float *d_bvh_leaves = nullptr;

    float *d_accum = nullptr;        // height*width*3 running sums
    float *h_mapped = nullptr;       // pt_map_accum's page-locked copy of them (allocated on first use)
    size_t mapped_floats = 0;
    size_t accum_floats = 0;
    uint64_t total_samples = 0;

    float *d_slabs = nullptr;        // partial sums of the work items
    size_t slab_floats = 0;
    uint32_t *d_queue = nullptr;     // [0] work counter; 64-bit counters: [2..3] segments, [4..5] redone,
                                     // [8..9] hierarchy nodes opened, [10..11] hierarchy leaves tested
    const uint32_t *last_queue = nullptr;   // the queue words the last render counted in (d_queue or the static scratch)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};

    // wavefront layout: two sets of path-state arrays, per-path results, live counts
    void *d_wf = nullptr;
    size_t wf_paths = 0;
    uint32_t *d_wf_counts = nullptr; // [0], [1]: live paths in set 0 / set 1

    int num_cus = 0;
    pt_counters last{};
    bool last_valid = false, last_pending = false;
};

namespace {

int fail(pt_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    g_error = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(ctx, PT_E_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                            \
    } while (0)

int use_device(pt_ctx *ctx)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the runtime's "last error" is per thread and sticky: whatever another library left there
    // (RCCL probes devices when a communicator is made) must not be taken for a failed launch here
    (void)hipGetLastError();
    return PT_OK;
}

template <typename T>
int dev_alloc(pt_ctx *ctx, T **p, size_t count)
{
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(ctx, PT_E_NOMEM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T),
                    hipGetErrorString(e));
    }
    return PT_OK;
}

// Scratch device buffer that lives for one call.
template <typename T>
struct Scratch {
    T *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
};

template <typename T>
int upload(pt_ctx *ctx, Scratch<T> &s, const T *host, size_t count)
{
    int rc = dev_alloc(ctx, &s.p, std::max<size_t>(count, 1));
    if (rc) return rc;
    if (count) HIP_TRY(ctx, hipMemcpyAsync(s.p, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return PT_OK;
}

unsigned blocks_for(long long n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

int check_ready(pt_ctx *ctx, bool need_camera)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!ctx->have_scene) return fail(ctx, PT_E_INVALID, "pt_set_scene has not been called");
    if (need_camera && !ctx->have_camera) return fail(ctx, PT_E_INVALID, "pt_set_camera has not been called");
    return use_device(ctx);
}

// Builds the hierarchy over the scene's small triangles on the GPU, or clears
// it when the scene has too few of them (or PT_ACCEL_NONE is set).  geom is
// the host copy of the geometry records already uploaded to ctx->d_geom.
// PT_ACCEL_AUTO: from this many small triangles on.  Measured at 800 x 800 (scan / hierarchy, ms
// of kernel at 64, 8 and 1 spp): 8 small triangles + 7 other surfaces 3.9 / 3.2 (64 spp),
// 0.65 / 0.61, 0.21 / 0.24; 18 + 7: 4.67 / 3.21, 0.84 / 0.63, 0.29 / 0.25; 242 + 7: 26.2 / 3.7,
// 4.25 / 0.87, 1.42 / 0.34.
constexpr int BVH_AUTO_MIN_TRIANGLES = 16;
constexpr int BVH_AUTO_SPHERES_PER_TRIANGLE = 8;
constexpr int BVH_MIN_SPHERES_INSIDE = 32;
// Measured, 800 x 800: 80 spheres 46.2 -> 32.2 ms and 249 small triangles 45.0 -> 30.3 ms with the
// early-outs; 15 triangles 16.4 -> 16.8 ms and the three large spheres of cornell_2 -10 % (hence off).
constexpr int WAVE_SKIP_MIN_SPHERES = 16, WAVE_SKIP_MIN_TRIANGLES = 32;

int build_hierarchy(pt_ctx *ctx, const pt_surface *surfaces, int n_surfaces)
{
    if (ctx->d_loose) (void)hipFree(ctx->d_loose);
    if (ctx->d_loose_geom) (void)hipFree(ctx->d_loose_geom);
    if (ctx->d_loose_lean) (void)hipFree(ctx->d_loose_lean);
    if (ctx->d_bvh_nodes) (void)hipFree(ctx->d_bvh_nodes);      // the leaves live in the same allocation
    ctx->d_loose = nullptr;
    ctx->d_loose_geom = nullptr;
    ctx->d_loose_lean = nullptr;
    ctx->d_bvh_nodes = nullptr;
    ctx->d_bvh_leaves = nullptr;
    ctx->n_bvh = ctx->n_loose = 0;
    if (ctx->accel == PT_ACCEL_NONE || ctx->accel == PT_ACCEL_LDS_TILES) return PT_OK;
    if (ctx->accel == PT_ACCEL_AUTO && ctx->n_tri + ctx->n_sph < BVH_AUTO_MIN_TRIANGLES) return PT_OK;

    // bounds of the surfaces' centres (triangle centroids, sphere centres); a surface much larger
    // than that spread stays outside
    auto centre = [&](const pt_surface &s, int k) {
        return s.kind == PT_TRIANGLE ? (s.a[k] + s.b[k] + s.c[k]) / 3.0f : s.a[k];
    };
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n_surfaces; i++)
        for (int k = 0; k < 3; k++) {
            const float c = centre(surfaces[i], k);
            if (!std::isfinite(c)) continue;
            cmin[k] = std::min(cmin[k], c);
            cmax[k] = std::max(cmax[k], c);
        }
    const float spread = (cmax[0] - cmin[0]) + (cmax[1] - cmin[1]) + (cmax[2] - cmin[2]);
    std::vector<int> loose, inside;
    float radius_min = INFINITY, sphere_reach = 0.0f;
    // spheres join only where there are many: with any of them inside, every box test of every ray
    // carries the ray's own padding (box_entry_padded), which a few spheres scanned beside the
    // hierarchy do not cost
    int small_spheres = 0;
    auto sphere_fits = [&](const pt_surface &s) {
        const float r = s.b[0], reach = std::fabs(s.a[0]) + std::fabs(s.a[1]) + std::fabs(s.a[2]);
        return std::isfinite(r) && std::isfinite(reach) && r > 0.0f && 6.0f * r <= 0.25f * spread &&
               r >= 1e-3f * spread && r * r >= 1e-30f && reach <= 1e15f;
    };
    for (int i = 0; i < n_surfaces; i++) small_spheres += surfaces[i].kind == PT_SPHERE && sphere_fits(surfaces[i]);
    const bool spheres_inside = small_spheres >= BVH_MIN_SPHERES_INSIDE;
    for (int i = 0; i < n_surfaces; i++) {
        const pt_surface &s = surfaces[i];
        bool in = false;
        if (s.kind == PT_TRIANGLE) {
            float size = 0;
            for (int k = 0; k < 3; k++) {
                const float lo = std::min({s.a[k], s.b[k], s.c[k]});
                const float hi = std::max({s.a[k], s.b[k], s.c[k]});
                size += hi - lo;
            }
            in = std::isfinite(size) && size <= 0.1f * spread && !extreme_triangle(s);
        } else {
            // a sphere: its box is 2 r a side; not so small that the allowance every ray makes for
            // the reference's rounding (BVH_SPHERE_ROUNDING / smallest radius) grows large
            in = spheres_inside && sphere_fits(s);
            if (in) {
                radius_min = std::min(radius_min, s.b[0]);
                sphere_reach = std::max(sphere_reach, std::fabs(s.a[0]) + std::fabs(s.a[1]) + std::fabs(s.a[2]));
            }
        }
        (in ? inside : loose).push_back(i);
    }
    const int n = (int)inside.size();
    // (a sphere the scan rejects after a third of its test, and whole waves skip it together: the
    // hierarchy pays from some 130 small spheres on, against 16 small triangles)
    int n_in_spheres = 0;
    for (int i : inside) n_in_spheres += surfaces[i].kind == PT_SPHERE;
    const int weight = (n - n_in_spheres) + n_in_spheres / BVH_AUTO_SPHERES_PER_TRIANGLE;
    if (n < 2 || (ctx->accel == PT_ACCEL_AUTO && weight < BVH_AUTO_MIN_TRIANGLES)) return PT_OK;
    if (n > BVH_MAX_LEAVES) {
        if (ctx->accel == PT_ACCEL_AUTO) return PT_OK;           // the scan has no such limit
        return fail(ctx, PT_E_LIMIT, "%d surfaces exceed the hierarchy's maximum of %d", n, BVH_MAX_LEAVES);
    }

    // bounds of the box centres of the surfaces that go in, for the Morton grid
    for (int k = 0; k < 3; k++) { cmin[k] = INFINITY; cmax[k] = -INFINITY; }
    for (int i : inside)
        for (int k = 0; k < 3; k++) {
            const pt_surface &s = surfaces[i];
            const float lo = s.kind == PT_TRIANGLE ? std::min({s.a[k], s.b[k], s.c[k]}) : s.a[k];
            const float hi = s.kind == PT_TRIANGLE ? std::max({s.a[k], s.b[k], s.c[k]}) : s.a[k];
            cmin[k] = std::min(cmin[k], 0.5f * (lo + hi));
            cmax[k] = std::max(cmax[k], 0.5f * (lo + hi));
        }
    // How far from the coordinate origin rays are expected to start: four times the scene's
    // own extent (|x| + |y| + |z| over the finite surfaces); see BvhView::ray_reach.
    float extent[3] = {0, 0, 0};
    for (int i = 0; i < n_surfaces; i++) {
        const pt_surface &s = surfaces[i];
        for (int k = 0; k < 3; k++) {
            float m = std::fabs(s.a[k]);
            if (s.kind == PT_TRIANGLE) m = std::max({m, std::fabs(s.b[k]), std::fabs(s.c[k])});
            else m += std::fabs(s.b[0]);
            if (std::isfinite(m)) extent[k] = std::max(extent[k], m);
        }
    }
    const float ray_reach = std::min(4.0f * (extent[0] + extent[1] + extent[2]), 1e30f);
    float3 gmin = make_float3(cmin[0], cmin[1], cmin[2]), ginv;
    // one scale for the three axes (the grid is a cube round the centroids): an axis along
    // which the scene is thin then spends few code bits, and the hierarchy's early splits
    // fall along the axes on which the triangles actually spread
    const float widest = std::max({cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]});
    ginv.x = ginv.y = ginv.z = widest > 0.0f ? 1.0f / widest : 0.0f;

    int rc;
    Scratch<int> d_inside, d_parent_node, d_parent_leaf;
    Scratch<unsigned long long> d_keys, d_sorted;
    Scratch<float4> d_bmin, d_bmax;
    Scratch<int2> d_children;
    Scratch<unsigned int> d_arrivals;
    Scratch<char> d_temp;
    if ((rc = upload(ctx, d_inside, inside.data(), (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_keys.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_sorted.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_bmin.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_bmax.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_children.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_parent_node.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_parent_leaf.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_arrivals.p, (size_t)n))) return rc;
    // one array of 64-byte records: n for the nodes (n - 1 used), then the n leaves
    static_assert(BVH_NODE_QUADS * sizeof(float4) == GEOM_STRIDE * sizeof(float), "nodes and leaves share a record size");
    if ((rc = dev_alloc(ctx, &ctx->d_bvh_nodes, (size_t)2 * n * BVH_NODE_QUADS))) return rc;
    ctx->d_bvh_leaves = reinterpret_cast<float *>(ctx->d_bvh_nodes + (size_t)n * BVH_NODE_QUADS);
    if ((rc = dev_alloc(ctx, &ctx->d_loose, std::max<size_t>(loose.size(), 1)))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_loose_geom, std::max<size_t>(loose.size(), 1) * GEOM_STRIDE))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_loose_lean, std::max<size_t>(loose.size(), 1) * LEAN_STRIDE))) return rc;
    Scratch<int> d_loose_runs;
    std::vector<int> loose_runs(loose.size());
    if (!loose.empty()) {
        for (int j = (int)loose.size() - 1, run = 0; j >= 0; j--) {
            run = (j + 1 < (int)loose.size() && surfaces[loose[j + 1]].kind == surfaces[loose[j]].kind) ? run + 1 : 1;
            loose_runs[j] = run;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_loose, loose.data(), loose.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = upload(ctx, d_loose_runs, loose_runs.data(), loose_runs.size()))) return rc;
        hipLaunchKernelGGL(k_bvh_gather_loose, dim3((unsigned)((loose.size() + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->d_geom, ctx->d_loose, d_loose_runs.p, (int)loose.size(), ctx->d_loose_geom);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_lean_records, dim3(blocks_for((long long)loose.size())), dim3(BLOCK), 0, ctx->stream,
                           ctx->d_loose_geom, (int)loose.size(), ctx->d_loose_lean);
        HIP_TRY(ctx, hipGetLastError());
    }

    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_bvh_prepare, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_geom, d_inside.p, n, gmin, ginv,
                       ray_reach, d_keys.p, d_bmin.p, d_bmax.p);
    HIP_TRY(ctx, hipGetLastError());
    size_t temp_bytes = 0;
    HIP_TRY(ctx, rocprim::radix_sort_keys(nullptr, temp_bytes, d_keys.p, d_sorted.p, (size_t)n, 0, 64, ctx->stream));
    if ((rc = dev_alloc(ctx, &d_temp.p, std::max<size_t>(temp_bytes, 1)))) return rc;
    HIP_TRY(ctx, rocprim::radix_sort_keys(d_temp.p, temp_bytes, d_keys.p, d_sorted.p, (size_t)n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_bvh_hierarchy, dim3(blocks), dim3(256), 0, ctx->stream, d_sorted.p, n, d_children.p,
                       d_parent_node.p, d_parent_leaf.p);
    HIP_TRY(ctx, hipGetLastError());
    // boxes bottom-up, one launch per level (k_bvh_refit_pass); the root's flag is looked at after
    // every few launches.  A tree over distinct 64-bit keys is at most 64 levels deep.
    HIP_TRY(ctx, hipMemsetAsync(d_arrivals.p, 0, (size_t)n * sizeof(unsigned int), ctx->stream));
    {
        int root_done = n < 2 ? 1 : 0;
        for (int pass = 1; !root_done; pass++) {
            if (pass > 2 * BVH_STACK + 2) return fail(ctx, PT_E_HIP, "hierarchy boxes did not reach the root");
            hipLaunchKernelGGL(k_bvh_refit_pass, dim3(blocks), dim3(256), 0, ctx->stream, d_sorted.p, n, d_children.p,
                               d_bmin.p, d_bmax.p, pass, reinterpret_cast<int *>(d_arrivals.p), ctx->d_bvh_nodes);
            HIP_TRY(ctx, hipGetLastError());
            if (pass % 8 == 0 || pass >= 24) {
                HIP_TRY(ctx, hipMemcpyAsync(&root_done, d_arrivals.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
    }
    hipLaunchKernelGGL(k_bvh_leaves, dim3(blocks), dim3(256), 0, ctx->stream, d_sorted.p, n, ctx->d_geom, d_inside.p,
                       ctx->d_bvh_leaves);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_bvh = n;
    ctx->n_loose = (int)loose.size();
    ctx->bvh_ray_reach = ray_reach;
    ctx->bvh_sphere_term = std::isfinite(radius_min) ? 1.0f / radius_min : 0.0f;
    ctx->bvh_sphere_reach = sphere_reach;
    return PT_OK;
}

SceneView scene_view(const pt_ctx *ctx)
{
    SceneView v{};
    v.geom = ctx->d_geom;
    v.lean = ctx->d_lean;
    v.shade = ctx->d_shade;
    v.n_surfaces = ctx->n_surfaces;
    v.n_scan = ctx->n_bvh > 0 ? ctx->n_loose : ctx->n_surfaces;
    v.loose = ctx->d_loose;
    v.loose_geom = ctx->d_loose_geom;
    v.loose_lean = ctx->d_loose_lean;
    v.bvh.nodes = ctx->d_bvh_nodes;
    v.bvh.leaves = ctx->d_bvh_leaves;
    v.bvh.n_leaves = ctx->n_bvh;
    v.bvh.ray_reach = ctx->bvh_ray_reach;
    v.bvh.sphere_term = ctx->bvh_sphere_term;
    v.bvh.sphere_reach = ctx->bvh_sphere_reach;
    memcpy(v.bg, ctx->bg, sizeof v.bg);
    v.max_depth = ctx->max_depth;
    return v;
}

// Dynamic LDS a kernel of scene kind KIND needs: the shading records, or the traversal stacks.
size_t lds_bytes_for(const pt_ctx *ctx, int kind)
{
    if (kind == SCENE_LDS) return (size_t)ctx->n_surfaces * SHADE_STRIDE * sizeof(float);
    if (kind == SCENE_BVH) return BVH_STACK_IN_LDS ? (size_t)BVH_STACK * BLOCK * sizeof(int) : 0;
    return 0;
}

int scene_kind(const pt_ctx *ctx)
{
    if (ctx->n_bvh > 0) return SCENE_BVH;
    return ctx->n_surfaces <= MAX_LDS_SURFACES ? SCENE_LDS : SCENE_GLOBAL;
}

// Calls f.template operator()<FAST, KIND>() for the run-time numeric mode and scene kind.
template <class F>
int with_variant(bool fast, int kind, F &&f)
{
    if (fast) {
        if (kind == SCENE_LDS) return f.template operator()<true, SCENE_LDS>();
        if (kind == SCENE_GLOBAL) return f.template operator()<true, SCENE_GLOBAL>();
        return f.template operator()<true, SCENE_BVH>();
    }
    if (kind == SCENE_LDS) return f.template operator()<false, SCENE_LDS>();
    if (kind == SCENE_GLOBAL) return f.template operator()<false, SCENE_GLOBAL>();
    return f.template operator()<false, SCENE_BVH>();
}

// Rays per lane of a closest-hit query (k_primary_hits_multi, k_intersect_rays_multi), or 0 for
// the one-ray kernels.  Measured on 16.8 M camera rays of cornell_2, ms with one ray per lane
// through the scalar cache / 2 / 4 / 8 rays per lane: FAST 0.115 / 0.103 / 0.097 / 0.094; EXACT
// 0.189 / 0.186 / 0.176 / 0.186 (124 registers at 8), and 0.350 / 0.340 / 0.320 / 0.343 on the 16
// surfaces of cornell (before the camera-ray kernel's wave-level early-outs, which the N-ray forms
// alone have: with them EXACT 0.147 / 0.138 at 2 / 4, FAST 0.070 at 8).  Scenes with a hierarchy, scenes whose records do not fit the kernels' LDS
// copy, and the LDS-tiled scan keep the one-ray kernels.  More rays per lane mean fewer
// workgroups, so small queries take fewer.  PT_QUERY_RAYS (0, 1, 2, 4, 8) overrides the choice,
// for either policy: for measurements and for the parity tests of every form.
int query_rays_per_lane(const pt_ctx *ctx, long long n_rays, bool fast)
{
    if (ctx->accel == PT_ACCEL_LDS_TILES || ctx->n_bvh > 0 || ctx->n_surfaces < 1 ||
        ctx->n_surfaces > QUERY_MAX_SURFACES)
        return 0;
    if (const char *e = getenv("PT_QUERY_RAYS")) {
        const int v = atoi(e);
        if (v == 0 || v == 1 || v == 2 || v == 4 || v == 8) return v;
    }
    // rays per lane if every CU had one workgroup
    const long long per_lane = n_rays / ((long long)std::max(ctx->num_cus, 1) * BLOCK);
    if (!fast) return per_lane >= 32 ? 4 : 0;
    if (per_lane >= 64) return 8;
    if (per_lane >= 32) return 4;
    if (per_lane >= 16) return 2;
    return 0;
}

template <class F>
int with_rays_per_lane(int n, F &&f)
{
    switch (n) {
    case 1: return f.template operator()<1>();
    case 2: return f.template operator()<2>();
    case 4: return f.template operator()<4>();
    default: return f.template operator()<8>();
    }
}

// Surfaces every segment scans linearly: all of them, or those outside the hierarchy.
static double ctx_scan_count(const pt_ctx *ctx) { return ctx->n_bvh > 0 ? ctx->n_loose : ctx->n_surfaces; }

// Where a render keeps its work queue and the work items' partial sums: the context's own
// buffers (the partial sums grow on demand), or a fixed area the call may not outgrow.
struct Workspace {
    uint32_t *queue = nullptr;      // 8 words, zeroed before every launch
    float *slabs = nullptr;         // fixed area: room for slab_floats partial sums
    size_t slab_floats = 0;
    bool fixed = false;             // false: ctx->d_queue and ctx->d_slabs
    // the band of image rows to render (pt_render_rows); n_rows = 0: the whole image
    uint32_t first_row = 0, n_rows = 0;
};

template <bool FAST, int KIND>
int launch_render(pt_ctx *ctx, float *accum, uint32_t spp, uint32_t first_sample, uint64_t seed, Workspace ws,
                  bool count_steps)
{
    const uint32_t width = (uint32_t)ctx->cam.width;
    const uint32_t n_pixels = ws.n_rows ? ws.n_rows * width : width * (uint32_t)ctx->cam.height;
    const uint32_t first_pixel = ws.n_rows ? ws.first_row * width : 0u;
    accum += (size_t)first_pixel * 3;
    size_t lds_bytes = lds_bytes_for(ctx, KIND);

    // scenes with a hierarchy have a megakernel of their own
    void (*kernel)(RenderParams) = nullptr;
    if constexpr (KIND != SCENE_BVH) {
        kernel = &k_render_mega<FAST, KIND>;
        // many spheres or many (then small) triangles: most waves miss most of them (see scan_surface)
        if (ctx->n_sph >= WAVE_SKIP_MIN_SPHERES || ctx->n_tri >= WAVE_SKIP_MIN_TRIANGLES)
            kernel = &k_render_mega<FAST, KIND, true>;
        if (ctx->accel == PT_ACCEL_LDS_TILES) {     // the scan's records through LDS tiles (its own static LDS)
            kernel = &k_render_mega_tiled<FAST>;
            lds_bytes = 0;
        }
    }
    const double samples = (double)n_pixels * spp;
    // Workgroups per CU (= waves per SIMD) for this launch: one more for every doubling of the
    // launch, from 2e5 samples on.  A small launch ends when its longest path does, and a wave's
    // iteration takes the longer the more waves share its SIMD, so fewer, busier waves finish
    // such a launch sooner.  Measured (kernel us; 800x800 cornell_2 at 1/2/4/8/16 spp): all
    // that fit 317/355/457/623/938, this rule 219/294/401/587/925; the million-triangle
    // field at 1 spp 2949 -> 2733, 80 spheres 1113 -> 1089; 356x200 at 1 spp 259 -> 124.
    const int blocks_wanted = std::max(1, (int)std::lround(0.9 + std::log2(std::max(samples, 1.0) / 2e5)));
    if constexpr (KIND == SCENE_BVH) {
        const bool spheres = ctx->bvh_sphere_term > 0.0f;                   // spheres among the leaves
        kernel = spheres ? &k_render_mega_bvh<FAST, 6, false, true> : &k_render_mega_bvh<FAST>;
        if (blocks_wanted >= 8)                                             // from 2e7 samples on
            kernel = spheres ? &k_render_mega_bvh<FAST, 7, false, true> : &k_render_mega_bvh<FAST, 7>;
        if (count_steps) kernel = spheres ? &k_render_mega_bvh<FAST, 6, true, true> : &k_render_mega_bvh<FAST, 6, true>;
    }
    int blocks_per_cu = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, BLOCK, lds_bytes));
    if (blocks_per_cu < 1) return fail(ctx, PT_E_HIP, "render kernel does not fit on a compute unit");
    blocks_per_cu = std::min(blocks_per_cu, blocks_wanted);
    const uint32_t grid = (uint32_t)blocks_per_cu * (uint32_t)ctx->num_cus;
    const uint64_t lanes = (uint64_t)grid * BLOCK;

    // Work items: enough per lane (about 32) that the last ones to finish cost
    // little, but no more chunks than samples.  (Headline launch: 8 per lane 47.3 ms,
    // 16: 44.6, 32: 44.0, 64: 43.8 with twice the partial sums again.)
    uint64_t want_chunks = (32 * lanes + n_pixels - 1) / n_pixels;
    if (ws.fixed) {
        // a fixed area holds so many chunks of partial sums and no more
        const uint64_t room = ws.slab_floats / ((uint64_t)n_pixels * 3);
        if (room < 1)
            return fail(ctx, PT_E_LIMIT, "image of %u pixels exceeds the static scratch area (%zu bytes hold %zu pixels)",
                        n_pixels, STATIC_SCRATCH_BYTES, (STATIC_SCRATCH_BYTES - STATIC_HEADER_BYTES) / 12);
        want_chunks = std::min(want_chunks, room);
    }
    uint32_t n_chunks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want_chunks, 1), spp);
    const uint32_t chunk_len = (spp + n_chunks - 1) / n_chunks;
    n_chunks = (spp + chunk_len - 1) / chunk_len;
    const uint64_t n_items64 = (uint64_t)n_pixels * n_chunks;
    // (below 2^31: the queue counts in 32 bits, and the kernels divide item and pixel numbers by fast_div)
    if (n_items64 >= 0x80000000ull)
        return fail(ctx, PT_E_LIMIT, "%llu work items exceed the queue's 2^31", (unsigned long long)n_items64);

    const size_t need = (size_t)n_items64 * 3;
    if (ws.fixed) {
        if (need > ws.slab_floats) return fail(ctx, PT_E_LIMIT, "partial sums exceed the static scratch area");
    } else {
        if (need > ctx->slab_floats) {
            if (ctx->d_slabs) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_slabs); }
            ctx->d_slabs = nullptr;
            ctx->slab_floats = 0;
            int rc = dev_alloc(ctx, &ctx->d_slabs, need);
            if (rc) return rc;
            ctx->slab_floats = need;
        }
        ws.queue = ctx->d_queue;
        ws.slabs = ctx->d_slabs;
    }

    RenderParams P{};
    P.cam = ctx->cam;
    P.scene = scene_view(ctx);
    P.seed_mix = host_seed_mix(seed);
    P.first_sample = first_sample;
    P.spp = spp;
    P.chunk_len = chunk_len;
    P.n_pixels = n_pixels;
    P.first_pixel = first_pixel;
    P.n_items = (uint32_t)n_items64;
    P.by_pixels = fast_div_by(n_pixels);
    P.by_width = fast_div_by(width);
    // Items a wave takes per atomic on the queue head.  Every wave of the launch adds to that
    // one word, which takes some 140 adds per microsecond; the batch is sized for about 40 at
    // the rate this scene is expected to consume samples (samples per microsecond, from what
    // a segment costs: the surfaces scanned and the depth of the hierarchy; 15 700 measured on
    // the 8-surface headline scene, 3 200 on 85 surfaces, 2 200 on the million-triangle field).
    // One-sample items in batches of 64 asked for more than the word can take (800x800 at
    // 16 spp: 2046 us, now 912 us; a hierarchy over 18 triangles at 8 spp 1.13 -> 0.58 ms),
    // while batches larger than needed leave the last waves with too much (1024 spp: 64 items
    // 41.7 ms, 256 items 43.8 ms; the million-triangle field at 4 spp 1.99 -> 2.23 ms).
    // And never so many that a launch of few items feeds only some of its waves (356x200 at
    // 1 spp is 71k items for 1024 waves: 143 us with batches of 256, 121 us with 69).
    const double samples_per_us = 2.5e5 / (8.0 + ctx_scan_count(ctx) + 6.0 * std::log2((double)ctx->n_bvh + 1.0));
    const uint64_t fair = n_items64 / (lanes / 64);
    P.pool_batch = (uint32_t)std::min<uint64_t>((uint64_t)(samples_per_us / (40.0 * chunk_len)), fair);
    P.pool_batch = std::min(std::max(P.pool_batch, POOL_BATCH_MIN), POOL_BATCH_MAX);
    P.slabs = ws.slabs;
    P.queue_head = ws.queue;
    P.stats = reinterpret_cast<unsigned long long *>(ws.queue + 2);
    ctx->last_queue = ws.queue;

    // Under stream capture (the launches become nodes of a HIP graph and run when the graph is
    // replayed) the timing events are left out: an event recorded into a graph cannot be waited
    // for or read afterwards, and the counters of such a call do not exist.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (ctx->stream) HIP_TRY(ctx, hipStreamIsCapturing(ctx->stream, &capture));
    const bool timed = capture == hipStreamCaptureStatusNone;

    HIP_TRY(ctx, hipMemsetAsync(ws.queue, 0, QUEUE_BYTES, ctx->stream));
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds_bytes, ctx->stream, P);
    HIP_TRY(ctx, hipGetLastError());
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    const size_t n = (size_t)n_pixels * 3;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks_for((long long)n)), dim3(BLOCK), 0, ctx->stream,
                       ws.slabs, n_chunks, n, accum);
    HIP_TRY(ctx, hipGetLastError());
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));

    hipFuncAttributes attr{};
    HIP_TRY(ctx, hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel)));
    ctx->last = pt_counters{};
    ctx->last.samples = (uint64_t)n_pixels * spp;
    ctx->last.vgprs = (uint32_t)attr.numRegs;
    ctx->last.grid_blocks = grid;
    ctx->last.accel = KIND == SCENE_BVH ? PT_ACCEL_BVH : PT_ACCEL_NONE;
    ctx->last_valid = timed;
    ctx->last_pending = timed;
    return PT_OK;
}

// Queue-per-bounce layout.  Samples are rendered in batches of whole samples per
// pixel, as many as fit the path-state budget; per batch: one generate launch,
// one bounce launch per path depth until no path is alive (the live counts are
// read back after every bounce to size the next launch), one reduce launch.
constexpr size_t WF_MAX_PATHS = 48u << 20;          // 48 Mi paths: 2 x 3 GiB of state + 576 MiB of results
constexpr size_t WF_BYTES_PER_PATH = 2 * 64 + 12;
constexpr size_t WF_COUNT_WORDS = (size_t)WF_SHARDS * WF_COUNT_STRIDE;

template <bool FAST, int KIND>
int launch_wavefront(pt_ctx *ctx, float *accum, uint32_t spp, uint32_t first_sample, uint64_t seed)
{
    const uint32_t n_pixels = (uint32_t)ctx->cam.width * (uint32_t)ctx->cam.height;
    if (n_pixels > WF_MAX_PATHS)
        return fail(ctx, PT_E_LIMIT, "image of %u pixels exceeds the wavefront layout's %zu paths", n_pixels, WF_MAX_PATHS);
    const uint32_t batch_spp_max = (uint32_t)std::min<size_t>(spp, WF_MAX_PATHS / n_pixels);
    const size_t max_paths = (size_t)batch_spp_max * n_pixels;
    // a region holds the workgroups g with g % WF_SHARDS == shard, whole workgroups
    auto shard_cap_for = [](size_t paths) {
        const size_t groups = (paths + BLOCK - 1) / BLOCK;
        return (uint32_t)(((groups + WF_SHARDS - 1) / WF_SHARDS) * BLOCK);
    };
    const size_t alloc_paths = (size_t)shard_cap_for(max_paths) * WF_SHARDS;
    if (alloc_paths > ctx->wf_paths) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_wf) (void)hipFree(ctx->d_wf);
        ctx->d_wf = nullptr;
        ctx->wf_paths = 0;
        hipError_t e = hipMalloc(&ctx->d_wf, alloc_paths * WF_BYTES_PER_PATH);
        if (e != hipSuccess) return fail(ctx, PT_E_NOMEM, "hipMalloc of %zu bytes of path state failed: %s",
                                         alloc_paths * WF_BYTES_PER_PATH, hipGetErrorString(e));
        ctx->wf_paths = alloc_paths;
    }
    if (!ctx->d_wf_counts) {
        int rc = dev_alloc(ctx, &ctx->d_wf_counts, 2 * WF_COUNT_WORDS);
        if (rc) return rc;
    }
    // carve: set 0 {a,b,c,d}, set 1 {a,b,c,d}, results
    char *base = static_cast<char *>(ctx->d_wf);
    const size_t q = ctx->wf_paths * 16;
    WfArrays sets[2];
    for (int k = 0; k < 2; k++) {
        sets[k].a = reinterpret_cast<float4 *>(base + (4 * k + 0) * q);
        sets[k].b = reinterpret_cast<float4 *>(base + (4 * k + 1) * q);
        sets[k].c = reinterpret_cast<float4 *>(base + (4 * k + 2) * q);
        sets[k].d = reinterpret_cast<uint4 *>(base + (4 * k + 3) * q);
    }
    float *result = reinterpret_cast<float *>(base + 8 * q);

    WfParams P{};
    P.cam = ctx->cam;
    P.scene = scene_view(ctx);
    P.seed_mix = host_seed_mix(seed);
    P.n_pixels = n_pixels;
    P.result = result;
    P.stats = reinterpret_cast<unsigned long long *>(ctx->d_queue + 2);
    const size_t lds_bytes = lds_bytes_for(ctx, KIND);
    std::vector<uint32_t> counts(WF_COUNT_WORDS);

    ctx->last_queue = ctx->d_queue;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue, 0, QUEUE_BYTES, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    uint64_t segments = 0;
    uint32_t launches = 0;
    for (uint32_t done = 0; done < spp;) {
        const uint32_t batch_spp = std::min(batch_spp_max, spp - done);
        const uint32_t n_paths = batch_spp * n_pixels;
        const uint32_t n_groups = blocks_for(n_paths);
        P.first_sample = first_sample + done;
        P.n_paths = n_paths;
        P.shard_cap = shard_cap_for(n_paths);
        P.in = sets[0];
        hipLaunchKernelGGL(k_wf_generate, dim3(n_groups), dim3(BLOCK), 0, ctx->stream, P);
        HIP_TRY(ctx, hipGetLastError());
        // initial live counts: region s holds workgroups s, s + WF_SHARDS, ...; the last workgroup may be partial
        std::fill(counts.begin(), counts.end(), 0u);
        for (uint32_t g = 0; g < n_groups; g++)
            counts[(g % WF_SHARDS) * WF_COUNT_STRIDE] += std::min<uint32_t>(BLOCK, n_paths - g * BLOCK);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_wf_counts, counts.data(), WF_COUNT_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        int cur = 0;
        for (;;) {
            uint64_t live = 0;
            uint32_t most = 0;
            for (uint32_t sh = 0; sh < WF_SHARDS; sh++) {
                live += counts[sh * WF_COUNT_STRIDE];
                most = std::max(most, counts[sh * WF_COUNT_STRIDE]);
            }
            if (live == 0) break;
            P.in = sets[cur];
            P.out = sets[cur ^ 1];
            P.n_in = ctx->d_wf_counts + cur * WF_COUNT_WORDS;
            P.n_out = ctx->d_wf_counts + (cur ^ 1) * WF_COUNT_WORDS;
            HIP_TRY(ctx, hipMemsetAsync(P.n_out, 0, WF_COUNT_WORDS * 4, ctx->stream));
            hipLaunchKernelGGL((k_wf_bounce<FAST, KIND>), dim3(blocks_for(most) * WF_SHARDS), dim3(BLOCK), lds_bytes, ctx->stream, P);
            HIP_TRY(ctx, hipGetLastError());
            segments += live;
            launches++;
            HIP_TRY(ctx, hipMemcpyAsync(counts.data(), P.n_out, WF_COUNT_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            cur ^= 1;
        }
        const size_t n = (size_t)n_pixels * 3;
        hipLaunchKernelGGL(k_wf_reduce, dim3(blocks_for((long long)n)), dim3(BLOCK), 0, ctx->stream,
                           result, batch_spp, n, accum);
        HIP_TRY(ctx, hipGetLastError());
        done += batch_spp;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    // the bounce launches know their own segment counts: put the total where pt_get_counters reads it
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_queue + 2, &segments, 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    hipFuncAttributes attr{};
    HIP_TRY(ctx, hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_wf_bounce<FAST, KIND>)));
    ctx->last = pt_counters{};
    ctx->last.samples = (uint64_t)n_pixels * spp;
    ctx->last.vgprs = (uint32_t)attr.numRegs;
    ctx->last.grid_blocks = launches;
    ctx->last.accel = KIND == SCENE_BVH ? PT_ACCEL_BVH : PT_ACCEL_NONE;
    ctx->last_valid = true;
    ctx->last_pending = true;
    return PT_OK;
}

// No exception crosses the C boundary: a failed host allocation becomes PT_E_NOMEM.
template <class F>
int guarded(pt_ctx *ctx, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(ctx, PT_E_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(ctx, PT_E_INVALID, "unexpected failure: %s", e.what());
    }
}

int render_dispatch_body(pt_ctx *ctx, float *accum, uint32_t spp, uint32_t first_sample, uint64_t seed, int mode,
                         Workspace ws);

int render_dispatch(pt_ctx *ctx, float *accum, uint32_t spp, uint32_t first_sample, uint64_t seed, int mode,
                    Workspace ws = Workspace{})
{
    return guarded(ctx, [&] { return render_dispatch_body(ctx, accum, spp, first_sample, seed, mode, ws); });
}

int render_dispatch_body(pt_ctx *ctx, float *accum, uint32_t spp, uint32_t first_sample, uint64_t seed, int mode,
                         Workspace ws)
{
    if (spp == 0) return fail(ctx, PT_E_INVALID, "spp must be positive");
    if ((uint64_t)first_sample + spp > 0xFFFFFFFFull)
        return fail(ctx, PT_E_LIMIT, "sample numbers must fit in 32 bits");
    if (mode & ~(PT_MODE_WAVEFRONT | PT_NUM_FAST | PT_MODE_COUNT_STEPS))
        return fail(ctx, PT_E_INVALID, "unknown mode bits 0x%x", mode);
    const bool fast = (mode & PT_NUM_FAST) != 0;
    if (mode & PT_MODE_WAVEFRONT) {
        if (ws.fixed) return fail(ctx, PT_E_INVALID, "the wavefront layout allocates its path state: megakernel only");
        if (ws.n_rows) return fail(ctx, PT_E_INVALID, "a band of rows is rendered by the megakernel only");
        return with_variant(fast, scene_kind(ctx), [&]<bool FAST, int KIND>() {
            return launch_wavefront<FAST, KIND>(ctx, accum, spp, first_sample, seed);
        });
    }
    return with_variant(fast, scene_kind(ctx), [&]<bool FAST, int KIND>() {
        return launch_render<FAST, KIND>(ctx, accum, spp, first_sample, seed, ws, (mode & PT_MODE_COUNT_STEPS) != 0);
    });
}

}  // namespace

extern "C" {

const char *pt_global_error(void) { return g_error.c_str(); }

int pt_create(int device, pt_ctx **out)
{
    if (!out) return fail(nullptr, PT_E_INVALID, "out is null");
    *out = nullptr;
    int count = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return fail(nullptr, PT_E_INVALID, "device %d out of range (%d visible)", device, count);
    pt_ctx *ctx = new (std::nothrow) pt_ctx;
    if (!ctx) return fail(nullptr, PT_E_NOMEM, "out of host memory");
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop{};
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        delete ctx;
        return fail(nullptr, PT_E_HIP, "device %d is %s; this library holds gfx950 code only", device, prop.gcnArchName);
    }
    if (e == hipSuccess && device < MAX_DEVICES) {
        // this device's copy of the static scratch area: asked for once, with the device current
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void *p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e == hipSuccess) g_scratch_ptr[device] = static_cast<unsigned char *>(p);
        }
    }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_queue, QUEUE_BYTES);
    for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventCreate(&ctx->ev[i]);
    if (e != hipSuccess) {
        int rc = fail(nullptr, PT_E_HIP, "pt_create on device %d: %s", device, hipGetErrorString(e));
        pt_destroy(ctx);
        return rc;
    }
    ctx->num_cus = prop.multiProcessorCount;
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->d_geom) (void)hipFree(ctx->d_geom);
    if (ctx->d_shade) (void)hipFree(ctx->d_shade);
    if (ctx->d_lean) (void)hipFree(ctx->d_lean);
    if (ctx->d_loose) (void)hipFree(ctx->d_loose);
    if (ctx->d_loose_geom) (void)hipFree(ctx->d_loose_geom);
    if (ctx->d_loose_lean) (void)hipFree(ctx->d_loose_lean);
    if (ctx->d_bvh_nodes) (void)hipFree(ctx->d_bvh_nodes);      // the leaves live in the same allocation
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    if (ctx->h_mapped) (void)hipHostFree(ctx->h_mapped);
    if (ctx->d_slabs) (void)hipFree(ctx->d_slabs);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_wf) (void)hipFree(ctx->d_wf);
    if (ctx->d_wf_counts) (void)hipFree(ctx->d_wf_counts);
    delete ctx;
}

const char *pt_last_error(pt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

int pt_set_stream(pt_ctx *ctx, void *hip_stream)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    int rc = use_device(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return PT_OK;
}

static int set_scene_body(pt_ctx *ctx, const pt_surface *surfaces, int32_t n_surfaces,
                          const pt_material *materials, int32_t n_materials, const float bg[3]);

int pt_set_scene(pt_ctx *ctx, const pt_surface *surfaces, int32_t n_surfaces,
                 const pt_material *materials, int32_t n_materials, const float bg[3])
{
    return guarded(ctx, [&] { return set_scene_body(ctx, surfaces, n_surfaces, materials, n_materials, bg); });
}

static int set_scene_body(pt_ctx *ctx, const pt_surface *surfaces, int32_t n_surfaces,
                          const pt_material *materials, int32_t n_materials, const float bg[3])
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (n_surfaces < 0 || n_materials < 0 || (n_surfaces > 0 && (!surfaces || !materials)) || !bg)
        return fail(ctx, PT_E_INVALID, "bad scene arguments");
    if (n_surfaces > (1 << 26))
        return fail(ctx, PT_E_LIMIT, "%d surfaces exceed the maximum of %d", n_surfaces, 1 << 26);
    int rc = use_device(ctx);
    if (rc) return rc;

    std::vector<float> geom((size_t)std::max(n_surfaces, 1) * GEOM_STRIDE, 0.0f);
    std::vector<float> shade((size_t)std::max(n_surfaces, 1) * SHADE_STRIDE, 0.0f);
    int n_tri = 0, n_sph = 0;
    for (int i = 0; i < n_surfaces; i++) {
        const pt_surface &s = surfaces[i];
        if (s.material < 0 || s.material >= n_materials)
            return fail(ctx, PT_E_INVALID, "surface %d names material %d of %d", i, s.material, n_materials);
        const pt_material &m = materials[s.material];
        if (m.type < PT_DIFFUSE || m.type > PT_GLASS)
            return fail(ctx, PT_E_INVALID, "material %d has unknown type %d", s.material, m.type);
        float *g = &geom[(size_t)i * GEOM_STRIDE];
        float *sh = &shade[(size_t)i * SHADE_STRIDE];
        const int32_t kind = s.kind;
        memcpy(g + 3, &kind, 4);
        if (kind == PT_TRIANGLE) {
            const HostV3 v0{s.a[0], s.a[1], s.a[2]}, v1{s.b[0], s.b[1], s.b[2]}, v2{s.c[0], s.c[1], s.c[2]};
            const HostV3 e1 = hsub(v1, v0), e2 = hsub(v2, v0);          // surface.hpp:48-49
            const HostV3 nrm = hnormalize(hcross(hsub(v2, v0), hsub(v1, v0)));  // surface.hpp:40
            g[0] = v0.x; g[1] = v0.y; g[2] = v0.z;
            g[4] = e1.x; g[5] = e1.y; g[6] = e1.z;
            g[7] = e1.y * e2.z - e2.y * e1.z;
            g[8] = e2.x; g[9] = e2.y; g[10] = e2.z;
            if (extreme_triangle(s)) g[7] = NAN;
            sh[8] = nrm.x; sh[9] = nrm.y; sh[10] = nrm.z;
            n_tri++;
        } else if (kind == PT_SPHERE) {
            g[0] = s.a[0]; g[1] = s.a[1]; g[2] = s.a[2];
            g[4] = s.b[0] * s.b[0];                                      // surface.hpp:83
            sh[8] = s.a[0]; sh[9] = s.a[1]; sh[10] = s.a[2];
            n_sph++;
        } else {
            return fail(ctx, PT_E_INVALID, "surface %d has unknown kind %d", i, kind);
        }
        const int32_t flags = m.type | (kind << 8);
        for (int k = 0; k < 3; k++) sh[k] = m.type == PT_DIFFUSE ? m.color[k] * 2.0f : m.color[k];
        memcpy(sh + 3, &flags, 4);
        memcpy(sh + 4, m.emittance, 12);
    }

    // run lengths by kind, for the scan of the whole scene
    for (int i = n_surfaces - 1, run = 0; i >= 0; i--) {
        run = (i + 1 < n_surfaces && surfaces[i + 1].kind == surfaces[i].kind) ? run + 1 : 1;
        memcpy(&geom[(size_t)i * GEOM_STRIDE + GEOM_RUN], &run, 4);
    }

    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_geom) (void)hipFree(ctx->d_geom);
    if (ctx->d_shade) (void)hipFree(ctx->d_shade);
    if (ctx->d_lean) (void)hipFree(ctx->d_lean);
    ctx->d_geom = ctx->d_shade = ctx->d_lean = nullptr;
    ctx->have_scene = false;
    if ((rc = dev_alloc(ctx, &ctx->d_geom, geom.size()))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_shade, shade.size()))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_lean, (size_t)std::max(n_surfaces, 1) * LEAN_STRIDE))) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_geom, geom.data(), geom.size() * sizeof(float), hipMemcpyHostToDevice));
    if (n_surfaces > 0) {
        // the FAST policy's records are derived from the geometry records on the device
        hipLaunchKernelGGL(k_lean_records, dim3(blocks_for(n_surfaces)), dim3(BLOCK), 0, ctx->stream, ctx->d_geom,
                           n_surfaces, ctx->d_lean);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpy(ctx->d_shade, shade.data(), shade.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->n_surfaces = n_surfaces;
    ctx->n_tri = n_tri;
    ctx->n_sph = n_sph;
    memcpy(ctx->bg, bg, sizeof ctx->bg);
    ctx->host_surfaces.assign(surfaces, surfaces + n_surfaces);
    if ((rc = build_hierarchy(ctx, ctx->host_surfaces.data(), n_surfaces))) return rc;
    ctx->have_scene = true;
    return PT_OK;
}

int pt_set_accel(pt_ctx *ctx, int accel)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (accel != PT_ACCEL_AUTO && accel != PT_ACCEL_NONE && accel != PT_ACCEL_BVH && accel != PT_ACCEL_LDS_TILES)
        return fail(ctx, PT_E_INVALID, "unknown acceleration mode %d", accel);
    int rc = use_device(ctx);
    if (rc) return rc;
    ctx->accel = accel;
    if (!ctx->have_scene) return PT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return guarded(ctx, [&] { return build_hierarchy(ctx, ctx->host_surfaces.data(), ctx->n_surfaces); });
}

int pt_set_max_depth(pt_ctx *ctx, int max_segments)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (max_segments < 1 || max_segments > PT_MAX_DEPTH)
        return fail(ctx, PT_E_INVALID, "path length cap %d outside 1..%d", max_segments, PT_MAX_DEPTH);
    ctx->max_depth = max_segments;
    return PT_OK;
}

int pt_set_camera(pt_ctx *ctx, const pt_camera *cam)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!cam) return fail(ctx, PT_E_INVALID, "camera is null");
    if (cam->width < 1 || cam->height < 1 || (int64_t)cam->width * cam->height > (1ll << 28))
        return fail(ctx, PT_E_LIMIT, "image size %d x %d out of range", cam->width, cam->height);
    int rc = use_device(ctx);
    if (rc) return rc;
    const size_t floats = (size_t)cam->width * cam->height * 3;
    if (floats != ctx->accum_floats) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_accum) (void)hipFree(ctx->d_accum);
        ctx->d_accum = nullptr;
        ctx->accum_floats = 0;
        if ((rc = dev_alloc(ctx, &ctx->d_accum, floats))) return rc;
        ctx->accum_floats = floats;
    }
    memcpy(ctx->cam.pos, cam->pos, sizeof cam->pos);
    memcpy(ctx->cam.rot, cam->rot, sizeof cam->rot);
    ctx->cam.focal = cam->focal;
    ctx->cam.width = cam->width;
    ctx->cam.height = cam->height;
    ctx->cam.dof_on = cam->dof_on != 0;
    ctx->cam.dof_focal_length = cam->dof_focal_length;
    ctx->cam.dof_aperture = cam->dof_aperture;
    ctx->have_camera = true;
    return pt_clear(ctx);
}

int pt_clear(pt_ctx *ctx)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!ctx->have_camera) return fail(ctx, PT_E_INVALID, "pt_set_camera has not been called");
    int rc = use_device(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, ctx->accum_floats * sizeof(float), ctx->stream));
    ctx->total_samples = 0;
    return PT_OK;
}

int pt_render(pt_ctx *ctx, uint32_t spp, uint32_t first_sample, uint64_t seed, int mode)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    rc = render_dispatch(ctx, ctx->d_accum, spp, first_sample, seed, mode);
    if (rc == PT_OK) ctx->total_samples += spp;
    return rc;
}

// A caller-owned image must be device memory of the context's GPU.
static int check_accum_device(pt_ctx *ctx, const float *accum_device)
{
    if (!accum_device) return fail(ctx, PT_E_INVALID, "accum_device is null");
    hipPointerAttribute_t pa{};
    if (hipPointerGetAttributes(&pa, accum_device) != hipSuccess || pa.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(ctx, PT_E_INVALID, "accum_device is not device memory");
    }
    if (pa.device != ctx->device)
        return fail(ctx, PT_E_INVALID, "accum_device lives on device %d, the context on %d", pa.device, ctx->device);
    return PT_OK;
}

int pt_render_into(pt_ctx *ctx, float *accum_device, uint32_t spp, uint32_t first_sample,
                   uint64_t seed, int mode)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if ((rc = check_accum_device(ctx, accum_device))) return rc;
    return render_dispatch(ctx, accum_device, spp, first_sample, seed, mode);
}

int pt_render_rows(pt_ctx *ctx, float *accum_device, int32_t first_row, int32_t n_rows, uint32_t spp,
                   uint32_t first_sample, uint64_t seed, int mode)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if ((rc = check_accum_device(ctx, accum_device))) return rc;
    if (first_row < 0 || n_rows < 0 || (int64_t)first_row + n_rows > ctx->cam.height)
        return fail(ctx, PT_E_INVALID, "rows %d..%lld outside an image of %d rows", first_row,
                    (long long)first_row + n_rows, ctx->cam.height);
    if (n_rows == 0) return PT_OK;          // an empty share
    Workspace ws;
    ws.first_row = (uint32_t)first_row;
    ws.n_rows = (uint32_t)n_rows;
    return render_dispatch(ctx, accum_device, spp, first_sample, seed, mode, ws);
}

int pt_render_into_static(pt_ctx *ctx, float *accum_device, uint32_t spp, uint32_t first_sample,
                          uint64_t seed, int mode)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if ((rc = check_accum_device(ctx, accum_device))) return rc;
    unsigned char *area = ctx->device < MAX_DEVICES ? g_scratch_ptr[ctx->device] : nullptr;
    if (!area) return fail(ctx, PT_E_INVALID, "device %d has no static scratch area", ctx->device);
    if ((uint64_t)ctx->cam.width * ctx->cam.height > (STATIC_SCRATCH_BYTES - STATIC_HEADER_BYTES) / 12)
        return fail(ctx, PT_E_LIMIT, "image %d x %d exceeds the static scratch area (%zu bytes hold %zu pixels)",
                    ctx->cam.width, ctx->cam.height, STATIC_SCRATCH_BYTES, (STATIC_SCRATCH_BYTES - STATIC_HEADER_BYTES) / 12);
    Workspace ws;
    ws.fixed = true;
    ws.queue = reinterpret_cast<uint32_t *>(area);
    ws.slabs = reinterpret_cast<float *>(area + STATIC_HEADER_BYTES);
    ws.slab_floats = (STATIC_SCRATCH_BYTES - STATIC_HEADER_BYTES) / sizeof(float);
    return render_dispatch(ctx, accum_device, spp, first_sample, seed, mode, ws);
}

int pt_read_accum(pt_ctx *ctx, float *rgb_sum_host, uint64_t *total_samples)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!ctx->have_camera) return fail(ctx, PT_E_INVALID, "pt_set_camera has not been called");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (rgb_sum_host) {
        HIP_TRY(ctx, hipMemcpyAsync(rgb_sum_host, ctx->d_accum, ctx->accum_floats * sizeof(float),
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (total_samples) *total_samples = ctx->total_samples;
    return PT_OK;
}

int pt_map_accum(pt_ctx *ctx, const float **rgb_sum_host, uint64_t *total_samples)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!rgb_sum_host) return fail(ctx, PT_E_INVALID, "rgb_sum_host is null");
    if (!ctx->have_camera) return fail(ctx, PT_E_INVALID, "pt_set_camera has not been called");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (ctx->mapped_floats != ctx->accum_floats) {
        if (ctx->h_mapped) (void)hipHostFree(ctx->h_mapped);
        ctx->h_mapped = nullptr;
        ctx->mapped_floats = 0;
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_mapped), ctx->accum_floats * sizeof(float), hipHostMallocDefault));
        ctx->mapped_floats = ctx->accum_floats;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mapped, ctx->d_accum, ctx->accum_floats * sizeof(float),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *rgb_sum_host = ctx->h_mapped;
    if (total_samples) *total_samples = ctx->total_samples;
    return PT_OK;
}

int pt_write_accum(pt_ctx *ctx, const float *rgb_sum, uint64_t total_samples)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!ctx->have_camera) return fail(ctx, PT_E_INVALID, "pt_set_camera has not been called");
    if (!rgb_sum) return fail(ctx, PT_E_INVALID, "rgb_sum is null");
    int rc = use_device(ctx);
    if (rc) return rc;
    // host or device memory: the runtime tells them apart
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_accum, rgb_sum, ctx->accum_floats * sizeof(float),
                                hipMemcpyDefault, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->total_samples = total_samples;
    return PT_OK;
}

int pt_tonemap_rgb8(pt_ctx *ctx, uint8_t *out_host)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!ctx->have_camera || !out_host) return fail(ctx, PT_E_INVALID, "no camera or null output");
    if (ctx->total_samples == 0) return fail(ctx, PT_E_INVALID, "nothing rendered yet");
    int rc = use_device(ctx);
    if (rc) return rc;
    Scratch<uint8_t> d_out;
    if ((rc = dev_alloc(ctx, &d_out.p, ctx->accum_floats))) return rc;
    hipLaunchKernelGGL(k_tonemap, dim3(blocks_for((long long)ctx->accum_floats)), dim3(BLOCK), 0, ctx->stream,
                       ctx->d_accum, ctx->accum_floats, (float)ctx->total_samples, d_out.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_host, d_out.p, ctx->accum_floats, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

namespace {
// Counters of a closest-hit query (pt_primary_hits, pt_intersect_rays): one segment per ray.
// Call after the stream has been synchronised; ev[0] and ev[1] bracket the kernel.
int note_query(pt_ctx *ctx, long long n_rays, const void *kernel, unsigned grid, int kind)
{
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    hipFuncAttributes attr{};
    HIP_TRY(ctx, hipFuncGetAttributes(&attr, kernel));
    ctx->last = pt_counters{};
    ctx->last.samples = (uint64_t)n_rays;
    ctx->last.segments = (uint64_t)n_rays;
    ctx->last.tests = (uint64_t)n_rays * (uint64_t)ctx->n_surfaces;
    ctx->last.kernel_ms = ms;
    ctx->last.total_ms = ms;
    ctx->last.flops = (double)n_rays * (63.0 * ctx->n_tri + 18.0 * ctx->n_sph);
    ctx->last.vgprs = (uint32_t)attr.numRegs;
    ctx->last.grid_blocks = grid;
    ctx->last.accel = kind == SCENE_BVH ? PT_ACCEL_BVH : PT_ACCEL_NONE;
    ctx->last_valid = true;
    ctx->last_pending = false;
    return PT_OK;
}
}  // namespace

int pt_primary_hits(pt_ctx *ctx, int32_t *surf_idx, float *distance, int numeric)
{
    int rc = check_ready(ctx, true);
    if (rc) return rc;
    if (!surf_idx || !distance) return fail(ctx, PT_E_INVALID, "null output");
    const long long n = (long long)ctx->cam.width * ctx->cam.height;
    Scratch<int32_t> d_idx;
    Scratch<float> d_dist;
    if ((rc = dev_alloc(ctx, &d_idx.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_dist.p, (size_t)n))) return rc;
    const void *kernel = nullptr;
    const int per_lane = query_rays_per_lane(ctx, n, (numeric & PT_NUM_FAST) != 0);
    unsigned grid = blocks_for(n);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    rc = with_variant((numeric & PT_NUM_FAST) != 0, scene_kind(ctx), [&]<bool FAST, int KIND>() {
        if (per_lane > 0) {
            return with_rays_per_lane(per_lane, [&]<int N>() {
                grid = (unsigned)((n + (long long)BLOCK * N - 1) / ((long long)BLOCK * N));
                kernel = reinterpret_cast<const void *>(&k_primary_hits_multi<FAST, N>);
                hipLaunchKernelGGL((k_primary_hits_multi<FAST, N>), dim3(grid), dim3(BLOCK),
                                   (size_t)ctx->n_surfaces * GEOM_STRIDE * sizeof(float), ctx->stream,
                                   ctx->cam, scene_view(ctx), d_idx.p, d_dist.p);
                return (int)PT_OK;
            });
        }
        if (ctx->accel == PT_ACCEL_LDS_TILES) {
            kernel = reinterpret_cast<const void *>(&k_primary_hits_tiled<FAST>);
            hipLaunchKernelGGL(k_primary_hits_tiled<FAST>, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream,
                               ctx->cam, scene_view(ctx), d_idx.p, d_dist.p);
            return PT_OK;
        }
        kernel = reinterpret_cast<const void *>(&k_primary_hits<FAST, KIND>);
        hipLaunchKernelGGL((k_primary_hits<FAST, KIND>), dim3(blocks_for(n)), dim3(BLOCK), lds_bytes_for(ctx, KIND), ctx->stream,
                           ctx->cam, scene_view(ctx), d_idx.p, d_dist.p);
        return PT_OK;
    });
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(surf_idx, d_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(distance, d_dist.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return note_query(ctx, n, kernel, grid, scene_kind(ctx));
}

int pt_intersect_rays(pt_ctx *ctx, const float *origin, const float *dir, int64_t n,
                      int32_t *surf_idx, float *distance, float *point, int numeric)
{
    int rc = check_ready(ctx, false);
    if (rc) return rc;
    if (n < 0 || n > (1ll << 31)) return fail(ctx, PT_E_LIMIT, "ray count %lld out of range", (long long)n);
    if (n == 0) return PT_OK;
    if (!origin || !dir || !surf_idx || !distance || !point) return fail(ctx, PT_E_INVALID, "null argument");
    Scratch<float> d_o, d_d, d_dist, d_pt;
    Scratch<int32_t> d_idx;
    if ((rc = upload(ctx, d_o, origin, (size_t)n * 3))) return rc;
    if ((rc = upload(ctx, d_d, dir, (size_t)n * 3))) return rc;
    if ((rc = dev_alloc(ctx, &d_idx.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_dist.p, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_pt.p, (size_t)n * 3))) return rc;
    const void *kernel = nullptr;
    const int per_lane = query_rays_per_lane(ctx, n, (numeric & PT_NUM_FAST) != 0);
    unsigned grid = blocks_for(n);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    rc = with_variant((numeric & PT_NUM_FAST) != 0, scene_kind(ctx), [&]<bool FAST, int KIND>() {
        if (per_lane > 0) {
            return with_rays_per_lane(per_lane, [&]<int N>() {
                grid = (unsigned)((n + (long long)BLOCK * N - 1) / ((long long)BLOCK * N));
                kernel = reinterpret_cast<const void *>(&k_intersect_rays_multi<FAST, N>);
                hipLaunchKernelGGL((k_intersect_rays_multi<FAST, N>), dim3(grid), dim3(BLOCK),
                                   (size_t)ctx->n_surfaces * GEOM_STRIDE * sizeof(float), ctx->stream,
                                   scene_view(ctx), d_o.p, d_d.p, (long long)n, d_idx.p, d_dist.p, d_pt.p);
                return (int)PT_OK;
            });
        }
        if (ctx->accel == PT_ACCEL_LDS_TILES) {
            kernel = reinterpret_cast<const void *>(&k_intersect_rays_tiled<FAST>);
            hipLaunchKernelGGL(k_intersect_rays_tiled<FAST>, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream,
                               scene_view(ctx), d_o.p, d_d.p, (long long)n, d_idx.p, d_dist.p, d_pt.p);
            return PT_OK;
        }
        kernel = reinterpret_cast<const void *>(&k_intersect_rays<FAST, KIND>);
        hipLaunchKernelGGL((k_intersect_rays<FAST, KIND>), dim3(blocks_for(n)), dim3(BLOCK), lds_bytes_for(ctx, KIND), ctx->stream,
                           scene_view(ctx), d_o.p, d_d.p, (long long)n, d_idx.p, d_dist.p, d_pt.p);
        return PT_OK;
    });
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(surf_idx, d_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(distance, d_dist.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(point, d_pt.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return note_query(ctx, n, kernel, grid, scene_kind(ctx));
}

int pt_camera_rays(pt_ctx *ctx, const int32_t *px, const float *xi, int64_t n,
                   float *origin, float *dir, int numeric)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!ctx->have_camera) return fail(ctx, PT_E_INVALID, "pt_set_camera has not been called");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n < 0 || n > (1ll << 31)) return fail(ctx, PT_E_LIMIT, "ray count %lld out of range", (long long)n);
    if (n == 0) return PT_OK;
    if (!px || !xi || !origin || !dir) return fail(ctx, PT_E_INVALID, "null argument");
    Scratch<int32_t> d_px;
    Scratch<float> d_xi, d_o, d_d;
    if ((rc = upload(ctx, d_px, px, (size_t)n * 2))) return rc;
    if ((rc = upload(ctx, d_xi, xi, (size_t)n * 4))) return rc;
    if ((rc = dev_alloc(ctx, &d_o.p, (size_t)n * 3))) return rc;
    if ((rc = dev_alloc(ctx, &d_d.p, (size_t)n * 3))) return rc;
    if (numeric & PT_NUM_FAST)
        hipLaunchKernelGGL(k_camera_rays<true>, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream,
                           ctx->cam, d_px.p, d_xi.p, (long long)n, d_o.p, d_d.p);
    else
        hipLaunchKernelGGL(k_camera_rays<false>, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream,
                           ctx->cam, d_px.p, d_xi.p, (long long)n, d_o.p, d_d.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(origin, d_o.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dir, d_d.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_eval_bsdf(pt_ctx *ctx, const int32_t *type, const float *color, const float *normal,
                 const float *incident, const float *xi, int64_t n,
                 float *out_dir, float *out_brdf, int numeric)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n < 0 || n > (1ll << 31)) return fail(ctx, PT_E_LIMIT, "row count %lld out of range", (long long)n);
    if (n == 0) return PT_OK;
    if (!type || !color || !normal || !incident || !xi || !out_dir || !out_brdf)
        return fail(ctx, PT_E_INVALID, "null argument");
    for (int64_t i = 0; i < n; i++)
        if (type[i] < PT_DIFFUSE || type[i] > PT_GLASS)
            return fail(ctx, PT_E_INVALID, "row %lld has unknown material type %d", (long long)i, type[i]);
    Scratch<int32_t> d_type;
    Scratch<float> d_col, d_nrm, d_inc, d_xi, d_dir, d_f;
    if ((rc = upload(ctx, d_type, type, (size_t)n))) return rc;
    if ((rc = upload(ctx, d_col, color, (size_t)n * 3))) return rc;
    if ((rc = upload(ctx, d_nrm, normal, (size_t)n * 3))) return rc;
    if ((rc = upload(ctx, d_inc, incident, (size_t)n * 3))) return rc;
    if ((rc = upload(ctx, d_xi, xi, (size_t)n * 2))) return rc;
    if ((rc = dev_alloc(ctx, &d_dir.p, (size_t)n * 3))) return rc;
    if ((rc = dev_alloc(ctx, &d_f.p, (size_t)n * 3))) return rc;
    if (numeric & PT_NUM_FAST)
        hipLaunchKernelGGL(k_eval_bsdf<true>, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream, d_type.p, d_col.p,
                           d_nrm.p, d_inc.p, d_xi.p, (long long)n, d_dir.p, d_f.p);
    else
        hipLaunchKernelGGL(k_eval_bsdf<false>, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream, d_type.p, d_col.p,
                           d_nrm.p, d_inc.p, d_xi.p, (long long)n, d_dir.p, d_f.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_dir, d_dir.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_brdf, d_f.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_eval_brdf(pt_ctx *ctx, const int32_t *type, const float *color, const float *normal,
                 const float *reflected, int64_t n, float *out_brdf)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n < 0 || n > (1ll << 31)) return fail(ctx, PT_E_LIMIT, "row count %lld out of range", (long long)n);
    if (n == 0) return PT_OK;
    if (!type || !color || !normal || !reflected || !out_brdf) return fail(ctx, PT_E_INVALID, "null argument");
    for (int64_t i = 0; i < n; i++)
        if (type[i] < PT_DIFFUSE || type[i] > PT_GLASS)
            return fail(ctx, PT_E_INVALID, "row %lld has unknown material type %d", (long long)i, type[i]);
    Scratch<int32_t> d_type;
    Scratch<float> d_col, d_nrm, d_refl, d_f;
    if ((rc = upload(ctx, d_type, type, (size_t)n))) return rc;
    if ((rc = upload(ctx, d_col, color, (size_t)n * 3))) return rc;
    if ((rc = upload(ctx, d_nrm, normal, (size_t)n * 3))) return rc;
    if ((rc = upload(ctx, d_refl, reflected, (size_t)n * 3))) return rc;
    if ((rc = dev_alloc(ctx, &d_f.p, (size_t)n * 3))) return rc;
    hipLaunchKernelGGL(k_eval_brdf, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream, d_type.p, d_col.p,
                       d_nrm.p, d_refl.p, (long long)n, d_f.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_brdf, d_f.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_rng_uniform(pt_ctx *ctx, uint64_t seed, const uint32_t *pixel, const uint32_t *sample,
                   const uint32_t *draw, int64_t n, float *out)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (n < 0 || n > (1ll << 31)) return fail(ctx, PT_E_LIMIT, "count %lld out of range", (long long)n);
    if (n == 0) return PT_OK;
    if (!pixel || !sample || !draw || !out) return fail(ctx, PT_E_INVALID, "null argument");
    Scratch<uint32_t> d_p, d_s, d_d;
    Scratch<float> d_out;
    if ((rc = upload(ctx, d_p, pixel, (size_t)n))) return rc;
    if ((rc = upload(ctx, d_s, sample, (size_t)n))) return rc;
    if ((rc = upload(ctx, d_d, draw, (size_t)n))) return rc;
    if ((rc = dev_alloc(ctx, &d_out.p, (size_t)n))) return rc;
    hipLaunchKernelGGL(k_rng_uniform, dim3(blocks_for(n)), dim3(BLOCK), 0, ctx->stream, host_seed_mix(seed),
                       d_p.p, d_s.p, d_d.p, (long long)n, d_out.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_get_counters(pt_ctx *ctx, pt_counters *out)
{
    if (!ctx) return fail(nullptr, PT_E_INVALID, "null context");
    if (!out) return fail(ctx, PT_E_INVALID, "out is null");
    if (!ctx->last_valid) return fail(ctx, PT_E_INVALID, "nothing rendered yet");
    int rc = use_device(ctx);
    if (rc) return rc;
    if (ctx->last_pending) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2]));
        float ms_kernel = 0, ms_total = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms_kernel, ctx->ev[0], ctx->ev[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms_total, ctx->ev[0], ctx->ev[2]));
        unsigned long long st[5] = {0, 0, 0, 0, 0};
        HIP_TRY(ctx, hipMemcpy(st, ctx->last_queue + 2, sizeof st, hipMemcpyDeviceToHost));
        const unsigned long long seg = st[0];
        ctx->last.segments = seg;
        ctx->last.redone = st[1];
        ctx->last.node_steps = st[3];
        ctx->last.leaf_steps = st[4];
        ctx->last.tests = seg * (uint64_t)ctx->n_surfaces;
        ctx->last.kernel_ms = ms_kernel;
        ctx->last.total_ms = ms_total;
        ctx->last.flops = (double)seg * (63.0 * ctx->n_tri + 18.0 * ctx->n_sph);
        ctx->last_pending = false;
    }
    *out = ctx->last;
    return PT_OK;
}

}  // extern "C"
