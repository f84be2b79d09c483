/*
 * pt_amd.h -- C ABI of the MI355X-native render step.
 *
 * The reference (jvgille/Path-Tracer) has no plugin or FFI interface: its
 * render step is the free function `void trace_rays(Scene&, BUFFER&)`
 * (src/main.cpp:114-143), called once per sample-per-pixel pass from
 * `update()` (src/main.cpp:270).  This header is the flat C boundary a host
 * binds in its place; each entry point names the reference code it replaces.
 *
 * Conventions: every function returns 0 on success and a negative PT_E_* code
 * on failure; pt_last_error() then describes the failure.  Nothing throws
 * across this boundary.  One context drives one GPU; calls on one context
 * must be serialised by the caller; different contexts are independent.
 * Host pointers are owned by the caller.  The HIP library is the only
 * implementation: there is no CPU fallback.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK            0
#define PT_E_INVALID    -1   /* bad argument or call order */
#define PT_E_HIP        -2   /* a HIP runtime call failed */
#define PT_E_NOMEM      -3
#define PT_E_LIMIT      -4   /* size beyond a documented maximum */

/* MaterialType, src/material.hpp:9-11 */
#define PT_DIFFUSE  0
#define PT_MIRROR   1
#define PT_GLASS    2
/* class Triangle / class Sphere, src/surface.hpp:31,76 */
#define PT_TRIANGLE 0
#define PT_SPHERE   1

/* pt_render mode bits */
#define PT_MODE_MEGAKERNEL 0x0   /* one persistent kernel, whole paths in registers */
#define PT_MODE_WAVEFRONT  0x1   /* queue-per-bounce kernels, path state in HBM */
#define PT_MODE_COUNT_STEPS 0x100 /* megakernel over a hierarchy: also count the nodes opened and the leaf
                                    triangles tested (pt_counters.node_steps, leaf_steps); a slower
                                    instantiation, for measurement */
#define PT_NUM_EXACT       0x0   /* no FMA contraction, IEEE divide/sqrt: the parity arithmetic */
#define PT_NUM_FAST        0x10  /* FMA + hardware rcp/rsq/sin/cos */

/* pt_set_accel modes */
#define PT_ACCEL_AUTO 0   /* hierarchy when the scene has 16 small triangles or more, eight small spheres
                             counting as one triangle (and no more leaves than it can hold: 2^25;
                             beyond that, the scan) */
#define PT_ACCEL_NONE 1   /* always the reference's linear scan over every surface */
#define PT_ACCEL_BVH  2   /* hierarchy whenever the scene has two small triangles or more (small spheres
                             join them from 32 of them on); PT_E_LIMIT beyond 2^25 leaves */
#define PT_ACCEL_LDS_TILES 3  /* linear scan with the records staged through LDS in tiles shared by a
                                 workgroup (pt_render's megakernel, pt_primary_hits, pt_intersect_rays):
                                 the layout kept to be measured against PT_ACCEL_NONE's scan through
                                 the scalar cache, which is faster on every scene measured; the
                                 wavefront layout scans as with PT_ACCEL_NONE */

/* One Surface, in scene order (Scene::surfaces, src/scene.hpp:18).
 * Triangle (src/surface.hpp:33-35): a,b,c = v0,v1,v2.
 * Sphere   (src/surface.hpp:78-79): a = position, b[0] = radius. */
typedef struct pt_surface {
    int32_t kind;        /* PT_TRIANGLE | PT_SPHERE */
    int32_t material;    /* index into the material table */
    float a[3];
    float b[3];
    float c[3];
} pt_surface;

/* class Material, src/material.hpp:69-76 */
typedef struct pt_material {
    int32_t type;        /* PT_DIFFUSE | PT_MIRROR | PT_GLASS */
    float color[3];
    float emittance[3];
} pt_material;

/* class Camera (src/camera.hpp:12-16) as trace_rays reads it, plus the
 * Scene's depth-of-field fields (src/scene.hpp:20-22) and the image size,
 * compile-time in the reference (src/constants.hpp:6-7), runtime here.
 * rot is the GLM column-major rotation matrix: rot[3*c + r] = m[c][r]. */
typedef struct pt_camera {
    float pos[3];
    float rot[9];
    float focal;               /* focal length in pixels */
    int32_t width;
    int32_t height;
    int32_t dof_on;
    float dof_focal_length;
    float dof_aperture;
} pt_camera;

/* Counters of the last pt_render, or of the last closest-hit query
 * (pt_primary_hits, pt_intersect_rays: one segment per ray, redone = 0). */
typedef struct pt_counters {
    uint64_t samples;          /* paths traced by the last pt_render */
    uint64_t segments;         /* closest-hit queries (src/main.cpp:85) in the last pt_render */
    uint64_t tests;            /* segments x surfaces */
    uint64_t redone;           /* segments whose closest hit was re-evaluated with IEEE divisions */
    double kernel_ms;          /* device time of the last pt_render's render kernel (HIP events) */
    double total_ms;           /* the same including the partial-sum reduction */
    double flops;              /* algorithmic flops: segments x (63 n_tri + 18 n_sph) */
    uint32_t vgprs;            /* registers per lane of the render kernel */
    uint32_t grid_blocks;      /* megakernel: workgroups launched; wavefront: bounce launches */
    uint32_t accel;            /* PT_ACCEL_BVH if the last pt_render went through a hierarchy (tests and
                                  flops then count what the reference's scan would do, not what ran),
                                  else PT_ACCEL_NONE */
    uint32_t reserved;
    uint64_t node_steps;       /* PT_MODE_COUNT_STEPS: hierarchy node records opened (two box tests each) */
    uint64_t leaf_steps;       /* PT_MODE_COUNT_STEPS: hierarchy leaves (triangles, spheres) tested */
} pt_counters;

typedef struct pt_ctx pt_ctx;

#define PT_MAX_DEPTH 1024      /* safety cap on path length; the reference has none */

/* Bytes of the static device area pt_render_into_static works in: 256 for the work
 * queue, the rest for partial sums of 12 bytes per pixel and chunk of samples, so an
 * image may have at most (PT_STATIC_SCRATCH_BYTES - 256) / 12 = 2 796 181 pixels. */
#define PT_STATIC_SCRATCH_BYTES (32u << 20)

/* Library-wide: message for a failed pt_create (no context exists yet). */
const char *pt_global_error(void);

/* Lifetime.  device = HIP device ordinal.  Replaces nothing in the reference
 * (it has no device); owns the device copies of the scene and the buffers. */
int  pt_create(int device, pt_ctx **out);
void pt_destroy(pt_ctx *ctx);
const char *pt_last_error(pt_ctx *ctx);

/* Stream all later work is enqueued on (a hipStream_t; NULL = default). */
int pt_set_stream(pt_ctx *ctx, void *hip_stream);

/* Uploads Scene::surfaces and Scene::BG_COLOR (src/scene.hpp:15-18).  Derived
 * values are computed here in the reference's arithmetic: Triangle::normal =
 * normalize(cross(v2-v0, v1-v0)) (src/surface.hpp:40), Sphere::radius2
 * (src/surface.hpp:83). */
int pt_set_scene(pt_ctx *ctx, const pt_surface *surfaces, int32_t n_surfaces,
                 const pt_material *materials, int32_t n_materials, const float bg[3]);

/* Chooses how the closest hit is searched; rebuilds for the current scene.
 * The reference scans every surface for every segment (src/main.cpp:58-66;
 * an "octree" is a to-do note, src/main.cpp:75).  The hierarchy, built on the
 * GPU over the scene's small triangles and spheres, returns the same surface,
 * t and distance as the scan, ties included: candidates are evaluated in the
 * reference's arithmetic, and the boxes are padded for what that arithmetic
 * accepts outside a triangle (its 1e-5 tolerances, and the error of its u and
 * v for rays nearly parallel to a triangle, where its determinant is just
 * above the 1e-5 cut-off) and grown, ray by ray, by what it accepts outside a
 * sphere (the rounding of its d2, which grows with the ray's distance).
 * Sphere::intersects takes the direction for a unit vector; a ray whose
 * direction is not one to within 1e-3 (pt_intersect_rays only) is answered by
 * the scan whenever the hierarchy holds spheres.  Checked on random scenes
 * against the scan, rays aimed at edges and vertices, along triangle planes
 * and along sphere silhouettes included (tools/soak_parity.py). */
int pt_set_accel(pt_ctx *ctx, int accel);

/* Ends every path of later renders after max_segments closest-hit queries
 * (1..PT_MAX_DEPTH; default PT_MAX_DEPTH).  Not the reference's behaviour: trace_ray's loop
 * (src/main.cpp:84) has no bound, and the default here has none in practice; the image the
 * reference ships of its default scene (images/orange_plane_alt_800_55s.png) is reproduced
 * without one.  The setting bounds the cost of a preview, and it is how the tests reproduce
 * the six other shipped images, which an earlier revision rendered with paths of 7 (8)
 * segments. */
int pt_set_max_depth(pt_ctx *ctx, int max_segments);

/* Uploads the camera and DOF state; (re)allocates the accumulation buffer
 * when the image size changes and clears it, as src/main.cpp:262-269 does
 * when the camera moved. */
int pt_set_camera(pt_ctx *ctx, const pt_camera *cam);

/* Zeroes the accumulation buffer and the sample counter
 * (src/main.cpp:262-269: TOTAL_SAMPLES = 0; buffer = 0). */
int pt_clear(pt_ctx *ctx);

/* trace_rays (src/main.cpp:114-143) run spp times: adds spp samples per pixel
 * to the context's accumulation buffer and spp to its sample counter
 * (TOTAL_SAMPLES, src/main.cpp:142).  Samples are numbered first_sample ..
 * first_sample+spp-1; the random stream of a path depends only on (seed,
 * pixel, sample number), so disjoint sample ranges rendered on different GPUs
 * add up to the same image as one range rendered on one GPU. */
int pt_render(pt_ctx *ctx, uint32_t spp, uint32_t first_sample, uint64_t seed, int mode);

/* The same, adding into a caller-owned DEVICE buffer of height*width*3 floats
 * (row-major, pixel (u,v) at 3*(v*width+u)); the context's own buffer and
 * counter are untouched.  For hosts that hold the image in their own device
 * memory (for instance to reduce it across GPUs). */
int pt_render_into(pt_ctx *ctx, float *accum_device, uint32_t spp, uint32_t first_sample,
                   uint64_t seed, int mode);

/* pt_render_into for a band of the image: rows first_row .. first_row + n_rows - 1 (0 = the
 * top row) receive spp samples per pixel, the rest of accum_device - still the whole image,
 * height*width*3 floats - is left as it is.  A pixel's random streams depend on its place in
 * the whole image, so bands rendered by different GPUs (or one after another) make up the
 * image one call would render, to the order of fp32 additions.  The reference has no
 * counterpart (its loops run over the whole buffer, src/main.cpp:118-119); this is the other way
 * to share an image among GPUs besides sharing its samples.  Megakernel only (PT_E_INVALID with
 * PT_MODE_WAVEFRONT); n_rows = 0 does nothing. */
int pt_render_rows(pt_ctx *ctx, float *accum_device, int32_t first_row, int32_t n_rows, uint32_t spp,
                   uint32_t first_sample, uint64_t seed, int mode);

/* pt_render_into for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in
 * the reference.  Same arguments and the same image; the megakernel's work queue and
 * the work items' partial sums live in a static device array of
 * PT_STATIC_SCRATCH_BYTES that every GPU the library has a context on holds until the
 * process ends, and the call makes no hipMalloc, hipFree or other allocating call.
 * PT_E_LIMIT for an image of more pixels than the area holds one partial sum for;
 * PT_E_INVALID for PT_MODE_WAVEFRONT (that layout allocates its path state).  With
 * less room than pt_render_into would take, samples are cut into fewer, longer
 * chunks: the sums then differ from pt_render_into's by the order of fp32 additions
 * only (equal bit for bit when both cut alike, as for small images).  All calls on a
 * GPU share that GPU's area: calls that overlap on one GPU, from any context or
 * stream, race; pt_get_counters after the call reads the area too.  A call made while
 * the context's stream is being captured becomes nodes of the graph (a memset and two
 * kernels) and leaves no counters: pt_get_counters then fails with PT_E_INVALID. */
int pt_render_into_static(pt_ctx *ctx, float *accum_device, uint32_t spp, uint32_t first_sample,
                          uint64_t seed, int mode);

/* Copies the accumulation buffer (BUFFER, src/main.cpp:31, as running sums)
 * to host memory, height*width*3 floats row-major, and the sample counter.
 * Either pointer may be NULL. */
int pt_read_accum(pt_ctx *ctx, float *rgb_sum_host, uint64_t *total_samples);

/* The same image without a copy of the caller's: the sums are read back into
 * page-locked host memory that the context owns (the fast path across PCIe) and
 * *rgb_sum_host points at it, height*width*3 floats row-major.  The memory stays
 * valid, and keeps these values, until the next pt_map_accum on this context,
 * a pt_set_camera that changes the image size, or pt_destroy.  For a caller that
 * adds every pass to a host-side BUFFER, as trace_rays does (src/main.cpp:139).
 * total_samples may be NULL. */
int pt_map_accum(pt_ctx *ctx, const float **rgb_sum_host, uint64_t *total_samples);

/* Replaces the accumulation buffer and the sample counter with caller data
 * (height*width*3 floats, row-major, in host or device memory): resumes an
 * image read back earlier with pt_read_accum, hands a host-side BUFFER to
 * pt_tonemap_rgb8, or adopts the sum of several GPUs' images (a device buffer
 * after the reduce) together with their total sample count. */
int pt_write_accum(pt_ctx *ctx, const float *rgb_sum, uint64_t total_samples);

/* draw()'s per-pixel conversion (src/main.cpp:146-162) with PutPixelSDL's
 * truncation (src/sdl_helpers.hpp:68-70): rgb8 = uint8(to_ldr(sum/N)).
 * out_host: height*width*3 bytes. */
int pt_tonemap_rgb8(pt_ctx *ctx, uint8_t *out_host);

/* Parity entry points.  Host pointers.  numeric = PT_NUM_EXACT | PT_NUM_FAST. */

/* get_ray_direction_by_pixel (src/main.cpp:50-54) on integer pixel corners,
 * then closest_intersection (src/main.cpp:56-67).  surf_idx[v*W+u] = index in
 * scene order or -1; distance = Intersection::distance (0 on a miss). */
int pt_primary_hits(pt_ctx *ctx, int32_t *surf_idx, float *distance, int numeric);

/* closest_intersection (src/main.cpp:56-67) with Triangle::intersects
 * (src/surface.hpp:47-73) and Sphere::intersects (src/surface.hpp:91-112) on
 * caller-supplied rays.  origin, dir, point: n x 3 floats. */
int pt_intersect_rays(pt_ctx *ctx, const float *origin, const float *dir, int64_t n,
                      int32_t *surf_idx, float *distance, float *point, int numeric);

/* The ray set-up of trace_rays (src/main.cpp:120-136) with forced random
 * numbers.  px: n x 2 (u,v); xi: n x 4 = jitter u, jitter v, lens radius
 * draw, lens angle draw.  origin, dir: n x 3.  numeric = PT_NUM_EXACT|FAST. */
int pt_camera_rays(pt_ctx *ctx, const int32_t *px, const float *xi, int64_t n,
                   float *origin, float *dir, int numeric);

/* Material::sample_pdf and Material::brdf (src/material.hpp:78-96) with
 * forced random numbers.  type: n; color, normal, incident: n x 3; xi: n x 2
 * (diffuse uses both, glass the first).  numeric = PT_NUM_EXACT|FAST. */
int pt_eval_bsdf(pt_ctx *ctx, const int32_t *type, const float *color, const float *normal,
                 const float *incident, const float *xi, int64_t n,
                 float *out_dir, float *out_brdf, int numeric);

/* Material::brdf (src/material.hpp:89-96) alone, for a caller-supplied
 * outgoing direction.  type: n; color, normal, reflected, out_brdf: n x 3. */
int pt_eval_brdf(pt_ctx *ctx, const int32_t *type, const float *color, const float *normal,
                 const float *reflected, int64_t n, float *out_brdf);

/* The uniform (0,1) variate that draw number `draw` of path (pixel, sample)
 * uses: the counter-based replacement for random_real (src/main.cpp:42-44).
 * Evaluated on the device.  out: n floats; pixel, sample, draw: n each. */
int pt_rng_uniform(pt_ctx *ctx, uint64_t seed, const uint32_t *pixel, const uint32_t *sample,
                   const uint32_t *draw, int64_t n, float *out);

/* Statistics of the last pt_render / pt_render_into (waits for it). */
int pt_get_counters(pt_ctx *ctx, pt_counters *out);

/* ---- several GPUs of one node ------------------------------------------------------
 * No counterpart in the reference (one thread, one image).  A group owns one context per
 * device; a render call's samples are partitioned by global sample number - device r of n
 * renders samples [first + r*spp/n, first + (r+1)*spp/n) of every pixel - and the per-device
 * running sums are combined on the group's first device by ONE ncclReduce (RCCL, loaded
 * with dlopen when the first group is created).  Because a path's random stream depends
 * only on (seed, pixel, sample number) and the image is a sum (src/main.cpp:139), the
 * group's image equals one GPU's up to the order of fp32 additions.  One host thread
 * drives a group; calls on a group must be serialised. */
typedef struct pt_group pt_group;

int  pt_group_create(const int *devices, int n_devices, pt_group **out);   /* distinct HIP ordinals */
void pt_group_destroy(pt_group *g);
const char *pt_group_last_error(pt_group *g);     /* g may be NULL after a failed create */
int  pt_group_size(pt_group *g);
/* Context of member i (0 = the device the image is combined on), for pt_get_counters and
 * the parity entry points; do not set its stream, scene or camera directly. */
pt_ctx *pt_group_context(pt_group *g, int i);
/* The share of rank `rank` of `n` in samples [first_sample, first_sample + spp_total); an empty
 * share (count 0) for n < 1 or a rank outside 0..n-1. */
void pt_group_share(uint32_t spp_total, int n, int rank, uint32_t first_sample, uint32_t *first, uint32_t *count);

/* pt_set_scene / pt_set_accel / pt_set_max_depth / pt_set_camera / pt_clear on every member
 * (the scene is replicated; the camera call sizes and clears the per-device images). */
int pt_group_set_scene(pt_group *g, const pt_surface *surfaces, int32_t n_surfaces,
                       const pt_material *materials, int32_t n_materials, const float bg[3]);
int pt_group_set_accel(pt_group *g, int accel);
int pt_group_set_max_depth(pt_group *g, int max_segments);
int pt_group_set_camera(pt_group *g, const pt_camera *cam);
int pt_group_clear(pt_group *g);

/* trace_rays (src/main.cpp:114-143) run spp_total times, shared among the devices: every
 * member's launch is enqueued on its own stream; the call does not wait for them. */
int pt_group_render(pt_group *g, uint32_t spp_total, uint32_t first_sample, uint64_t seed, int mode);
/* The one collective: sums the members' images onto the first device (ncclReduce, fp32 sum,
 * height*width*3 values: 7.68 MB at 800 x 800) and makes the result, with the sample count
 * of all render calls since the last clear, the first context's accumulation buffer.  Waits. */
int pt_group_reduce(pt_group *g);
/* pt_read_accum / pt_tonemap_rgb8 of the combined image (reducing first when render calls
 * have been made since the last reduce). */
int pt_group_read_accum(pt_group *g, float *rgb_sum_host, uint64_t *total_samples);
int pt_group_tonemap_rgb8(pt_group *g, uint8_t *out_host);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
