/*
 * pt_oracle.c -- CPU restatement of the reference path tracer's render step.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  PARITY: pinned at image level
 * by the five renders the reference ships (tests/test_reference_images.py);
 * unpinned at bit level (GLM's operation order) - the reference holds no tests
 * or golden vectors for this path and cannot be built here.
 *
 * Every routine cites the reference file:line it follows (paths relative to
 * the reference's root).  GLM is an un-vendored, un-pinned dependency of the
 * reference (CMakeLists.txt:14); its scalar formulas are restated here in the
 * operation order GLM publishes for vec3/mat3 with default qualifiers:
 *   dot(a,b)       = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   cross(x,y)     = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
 *   length(v)      = sqrt(dot(v,v))
 *   normalize(v)   = v * (1/sqrt(dot(v,v)))
 *   determinant(m) = m00*(m11*m22 - m21*m12) - m10*(m01*m22 - m21*m02)
 *                  + m20*(m01*m12 - m11*m02)          with m[col][row]
 *   (M*v)[r]       = (m[0][r]*v.x + m[1][r]*v.y) + m[2][r]*v.z
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fopenmp -fPIC -shared (x86-64,
 * SSE2 scalar fp32, no FMA, FLT_EVAL_METHOD 0).
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif
#ifndef M_PI_4
#define M_PI_4 0.78539816339744830962
#endif

/* constants.hpp:4 -- a double, so every comparison against it promotes. */
static const double EPSILON = 0.00001;

typedef struct { float x, y, z; } v3;
typedef struct { v3 c0, c1, c2; } m3;  /* columns */

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float length(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }
static inline float determinant(v3 c0, v3 c1, v3 c2)
{
    return + c0.x * (c1.y * c2.z - c2.y * c1.z)
           - c1.x * (c0.y * c2.z - c2.y * c0.z)
           + c2.x * (c0.y * c1.z - c1.y * c0.z);
}
static inline v3 mat_vec(const m3 *m, v3 v)
{
    return V((m->c0.x * v.x + m->c1.x * v.y) + m->c2.x * v.z,
             (m->c0.y * v.x + m->c1.y * v.y) + m->c2.y * v.z,
             (m->c0.z * v.x + m->c1.z * v.y) + m->c2.z * v.z);
}
static inline m3 cam_matrix(const orc_camera *cam)
{
    m3 m = { ld3(cam->rot), ld3(cam->rot + 3), ld3(cam->rot + 6) };
    return m;
}

/* ------------------------------------------------------------------ camera */

/* camera.hpp:20-27.  cos/sin are called on float arguments from a class body
 * that precedes any using-directive for std, so the C double overloads are
 * the ones found; each result is narrowed to float on assignment.  The two
 * mat3 literals are GLM column-major (first three scalars = column 0). */
static void rotation_matrix(float rot_x, float rot_y, float out[9])
{
    float cx = (float)cos((double)-rot_x);
    float sx = (float)sin((double)-rot_x);
    float cy = (float)cos((double)-rot_y);
    float sy = (float)sin((double)-rot_y);
    /* a = mat3(cy,0,sy, 0,1,0, -sy,0,cy); b = mat3(1,0,0, 0,cx,-sx, 0,sx,cx) */
    const v3 a0 = V(cy, 0, sy), a1 = V(0, 1, 0), a2 = V(-sy, 0, cy);
    const float b[3][3] = { {1, 0, 0}, {0, cx, -sx}, {0, sx, cx} };
    /* (a*b) column j = a0*b[j][0] + a1*b[j][1] + a2*b[j][2] */
    for (int j = 0; j < 3; j++) {
        v3 col = add(add(scale(a0, b[j][0]), scale(a1, b[j][1])), scale(a2, b[j][2]));
        st3(out + 3 * j, col);
    }
}

void orc_camera_init(orc_camera *cam, const float pos[3], float rot_x, float rot_y,
                     float focal, int add_offset, int width, int height)
{
    memset(cam, 0, sizeof *cam);
    memcpy(cam->pos, pos, sizeof cam->pos);
    if (add_offset) {
        /* camera.hpp:10,36: OFFSET{2*EPSILON, EPSILON} narrowed to vec2. */
        rot_x = rot_x + (float)(2 * EPSILON);
        rot_y = rot_y + (float)EPSILON;
    }
    rotation_matrix(rot_x, rot_y, cam->rot);
    cam->focal = focal;
    cam->width = width;
    cam->height = height;
    cam->dof_on = 0;
    cam->dof_focal_length = 2.0f;   /* scene.hpp:25 defaults */
    cam->dof_aperture = 0.15f;
}

/* The interactive camera, camera.hpp:41-58 and :76-89, driven as the main loop drives it
 * (handle_input's rotate and move, main.cpp:178-203, then update, main.cpp:259) for n steps
 * from the stored pose (pos, rot).  moves: n x 5 = d_rot_x, d_rot_y, dx, dy, dz.  out: n x 15 =
 * pos (3), rot (2), update()'s return value, rotation matrix (9) after each step.
 * cos and sin take their double forms as in rotation_matrix above; 2*M_PI and the yaw
 * comparisons are double, the stores narrow to float. */
void orc_camera_walk(const float pos0[3], const float rot0[2], int n, const float *moves, float *out)
{
    float pos[3] = {pos0[0], pos0[1], pos0[2]}, rot[2] = {rot0[0], rot0[1]};
    float matrix[9];
    int dirty = 0;
    rotation_matrix(rot[0], rot[1], matrix);
    for (int k = 0; k < n; k++) {
        const float *m = moves + 5 * k;
        if (!(m[0] == 0 && m[1] == 0)) {                      /* rotate, camera.hpp:41-48 */
            rot[0] += m[0];
            rot[1] += m[1];
            dirty = 1;
        }
        if (!(m[2] == 0 && m[3] == 0 && m[4] == 0)) {         /* move, camera.hpp:50-58 */
            const double c = cos((double)-rot[1]), sn = sin((double)-rot[1]);
            pos[0] = (float)(pos[0] + (m[2] * c - m[4] * sn));
            pos[2] = (float)(pos[2] + (m[2] * sn + m[4] * c));
            pos[1] += m[3];
            dirty = 1;
        }
        int updated = 0;
        if (dirty) {                                          /* update, camera.hpp:76-89 */
            dirty = 0;
            updated = 1;
            rotation_matrix(rot[0], rot[1], matrix);          /* before the clamp and the wrap */
            const float lo = (float)-M_PI_2, hi = (float)M_PI_2;
            rot[0] = rot[0] < lo ? lo : (hi < rot[0] ? hi : rot[0]);     /* glm::clamp = min(max(x, lo), hi) */
            while (rot[1] > 2 * M_PI) rot[1] = (float)(rot[1] - 2 * M_PI);
            while (rot[1] < 0) rot[1] = (float)(rot[1] + 2 * M_PI);
        }
        float *o = out + 15 * k;
        o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
        o[3] = rot[0]; o[4] = rot[1];
        o[5] = (float)updated;
        memcpy(o + 6, matrix, sizeof matrix);
    }
}

/* main.cpp:50-54.  SCREEN_WIDTH/2 and SCREEN_HEIGHT/2 are integer divisions. */
static v3 ray_direction_by_pixel(const orc_camera *cam, const m3 *rot, float u, float v)
{
    v3 p = V(u - (float)(cam->width / 2), v - (float)(cam->height / 2), cam->focal);
    return normalize(mat_vec(rot, p));
}

/* main.cpp:120-136.  xi_* are the four random_real results in [0,1]:
 * jitter for u, jitter for v, lens radius, lens angle. */
static void camera_ray(const orc_camera *cam, const m3 *rot, int u, int v,
                       double xi_u, double xi_v, double xi_r, double xi_phi,
                       v3 *origin, v3 *dir)
{
    const v3 camera_pos = ld3(cam->pos);
    *origin = camera_pos;
    /* int + double, narrowed to the float parameters (main.cpp:122) */
    *dir = ray_direction_by_pixel(cam, rot, (float)(u + xi_u), (float)(v + xi_v));
    if (cam->dof_on) {
        const v3 forward  = mat_vec(rot, V(0, 0, 1));    /* camera.hpp:91-93 */
        const v3 downward = mat_vec(rot, V(0, 1, 0));    /* camera.hpp:95-97 */
        const v3 right    = mat_vec(rot, V(1, 0, 0));    /* camera.hpp:99-101 */
        float t = cam->dof_focal_length / dot(forward, *dir);          /* :125 */
        v3 p = add(scale(*dir, t), camera_pos);                         /* :126 */
        float r = (float)(0 + xi_r * ((double)cam->dof_aperture - 0));  /* :129 */
        float phi = (float)(0 + xi_phi * (2 * M_PI - 0));               /* :130 */
        float dx = r * cosf(phi);   /* std::cos(float) after main.cpp:21 */
        float dy = r * sinf(phi);
        *origin = add(add(camera_pos, scale(right, dx)), scale(downward, dy)); /* :134 */
        *dir = normalize(sub(p, *origin));                                      /* :135 */
    }
}

void orc_camera_rays(const orc_camera *cam, const int32_t *px, const float *xi,
                     int64_t n, float *origin, float *dir)
{
    const m3 rot = cam_matrix(cam);
    for (int64_t i = 0; i < n; i++) {
        v3 o, d;
        camera_ray(cam, &rot, px[2 * i], px[2 * i + 1], xi[4 * i], xi[4 * i + 1],
                   xi[4 * i + 2], xi[4 * i + 3], &o, &d);
        st3(origin + 3 * i, o);
        st3(dir + 3 * i, d);
    }
}

/* ------------------------------------------------------------ intersection */

typedef struct { int idx; float distance; v3 point; } hit_t;   /* intersection.hpp:12-16 */

/* surface.hpp:47-73 */
static int triangle_intersects(const orc_surface *s, v3 ray_start, v3 ray_dir, hit_t *out)
{
    const v3 v0 = ld3(s->a), v1 = ld3(s->b), v2 = ld3(s->c);
    v3 e1 = sub(v1, v0);
    v3 e2 = sub(v2, v0);

    v3 nd = neg(ray_dir);
    float det_A = determinant(nd, e1, e2);
    if (-EPSILON < (double)det_A && (double)det_A < EPSILON)
        return 0;

    v3 b = sub(ray_start, v0);
    float u = determinant(nd, b, e2) / det_A;
    if ((double)u < -EPSILON || u >= 1)
        return 0;

    float v = determinant(nd, e1, b) / det_A;
    if ((double)v < -EPSILON || u + v >= 1)
        return 0;

    float t = determinant(b, e1, e2) / det_A;
    if ((double)t < EPSILON)
        return 0;

    out->point = add(ray_start, scale(ray_dir, t));
    out->distance = length(scale(ray_dir, t));
    return 1;
}

/* surface.hpp:91-112 */
static int sphere_intersects(const orc_surface *s, v3 ray_start, v3 ray_dir, hit_t *out)
{
    const v3 position = ld3(s->a);
    const float radius = s->b[0];
    const float radius2 = radius * radius;          /* surface.hpp:83 */

    v3 L = sub(position, ray_start);
    float tca = dot(L, ray_dir);
    if (tca < 0)
        return 0;

    float d2 = dot(L, L) - tca * tca;
    if (d2 > radius2)
        return 0;

    float thc = sqrtf(radius2 - d2);
    float t = tca - thc;
    if ((double)t < EPSILON) {
        t = tca + thc;
        if ((double)t < EPSILON)
            return 0;
    }
    out->point = add(ray_start, scale(ray_dir, t));
    out->distance = t;
    return 1;
}

/* main.cpp:56-67: scene order, strictly smaller distance replaces. */
static int closest_intersection(const orc_scene *scene, v3 start, v3 dir, hit_t *closest)
{
    int found = 0;
    for (int i = 0; i < scene->n_surfaces; i++) {
        const orc_surface *s = &scene->surfaces[i];
        hit_t res;
        int ok = (s->kind == ORC_TRIANGLE) ? triangle_intersects(s, start, dir, &res)
                                           : sphere_intersects(s, start, dir, &res);
        if (ok) {
            if (!found || res.distance < closest->distance) {
                *closest = res;
                closest->idx = i;
                found = 1;
            }
        }
    }
    return found;
}

/* surface.hpp:38-45 (Triangle: constant normal) and :86-88 (Sphere). */
static v3 surface_normal(const orc_surface *s, v3 point)
{
    if (s->kind == ORC_TRIANGLE) {
        const v3 v0 = ld3(s->a), v1 = ld3(s->b), v2 = ld3(s->c);
        return normalize(cross(sub(v2, v0), sub(v1, v0)));
    }
    return normalize(sub(point, ld3(s->a)));
}

void orc_primary_hits(const orc_scene *scene, const orc_camera *cam,
                      int32_t *idx, float *dist)
{
    const m3 rot = cam_matrix(cam);
    const v3 pos = ld3(cam->pos);
    for (int y = 0; y < cam->height; y++) {
        for (int x = 0; x < cam->width; x++) {
            v3 d = ray_direction_by_pixel(cam, &rot, (float)x, (float)y);
            hit_t h;
            int64_t p = (int64_t)y * cam->width + x;
            if (closest_intersection(scene, pos, d, &h)) {
                idx[p] = h.idx;
                dist[p] = h.distance;
            } else {
                idx[p] = -1;
                dist[p] = 0.0f;
            }
        }
    }
}

void orc_intersect_rays(const orc_scene *scene, const float *o, const float *d,
                        int64_t n, int32_t *idx, float *dist, float *point)
{
    for (int64_t i = 0; i < n; i++) {
        hit_t h;
        if (closest_intersection(scene, ld3(o + 3 * i), ld3(d + 3 * i), &h)) {
            idx[i] = h.idx;
            dist[i] = h.distance;
            st3(point + 3 * i, h.point);
        } else {
            idx[i] = -1;
            dist[i] = 0.0f;
            st3(point + 3 * i, V(0, 0, 0));
        }
    }
}

/* --------------------------------------------------------------- materials */

/* material.hpp:13-27.  The function precedes any using-directive for std, so
 * cos/sin/acos are the C double functions; u, v, theta, phi are floats. */
static v3 sample_diffuse(v3 normal, double xi_u, double xi_v)
{
    float u = (float)xi_u;
    float v = (float)xi_v;
    float theta = (float)(2 * M_PI * (double)u);
    float phi = (float)acos((double)(2 * v - 1));
    v3 direction = V((float)(cos((double)theta) * sin((double)phi)),
                     (float)(sin((double)theta) * sin((double)phi)),
                     (float)cos((double)phi));
    if (dot(direction, normal) < 0)
        direction = scale(direction, -1);
    return direction;
}

/* material.hpp:29-31 */
static v3 sample_mirror(v3 normal, v3 incident)
{
    return sub(incident, scale(normal, 2 * dot(normal, incident)));
}

/* material.hpp:33-67 */
static v3 sample_glass(v3 normal, v3 incident, double xi)
{
    const float air_refractive_index = 1.0f;
    const float glass_refractive_index = 1.5f;
    float n1, n2;
    if (dot(normal, incident) < 0) {
        n1 = air_refractive_index;
        n2 = glass_refractive_index;
    } else {
        normal = scale(normal, -1);
        n1 = glass_refractive_index;
        n2 = air_refractive_index;
    }
    float n = n1 / n2;
    float cosI = -dot(normal, incident);
    float sin2T = n * n * (1 - cosI * cosI);
    float cosT = (float)sqrt((double)(1 - sin2T));   /* NaN under total internal reflection */

    float reflectance;
    if (sin2T > 1) {
        reflectance = 1;
    } else {
        float rOrth = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
        float rPar = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
        reflectance = (rOrth * rOrth + rPar * rPar) / 2;
    }
    if (xi < (double)reflectance)
        return add(incident, scale(normal, 2 * cosI));
    return add(scale(incident, n), scale(normal, n * cosI - cosT));
}

/* material.hpp:78-87 */
static v3 sample_pdf(int type, v3 normal, v3 incident, double xi0, double xi1)
{
    switch (type) {
    case ORC_DIFFUSE: return sample_diffuse(normal, xi0, xi1);
    case ORC_MIRROR:  return sample_mirror(normal, incident);
    default:          return sample_glass(normal, incident, xi0);
    }
}

/* material.hpp:89-96 */
static v3 brdf(int type, v3 color, v3 normal, v3 reflected)
{
    if (type == ORC_DIFFUSE)
        return scale(scale(color, 2.0f), dot(reflected, normal));
    return color;
}

void orc_eval_bsdf(const int32_t *type, const float *color, const float *normal,
                   const float *incident, const float *xi, int64_t n,
                   float *out_dir, float *out_brdf)
{
    for (int64_t i = 0; i < n; i++) {
        v3 nrm = ld3(normal + 3 * i);
        v3 dir = sample_pdf(type[i], nrm, ld3(incident + 3 * i), xi[2 * i], xi[2 * i + 1]);
        st3(out_dir + 3 * i, dir);
        st3(out_brdf + 3 * i, brdf(type[i], ld3(color + 3 * i), nrm, dir));
    }
}

/* --------------------------------------------------------- random numbers */

static inline uint32_t pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

/* The counter-based generator the HIP path uses in place of libc rand() (a design change: the
 * reference's generator is serial global state).  key = pcg(pcg(seed ^ pixel) ^ sample * G),
 * start = pixel * A + sample * B, draw i = pcg(key ^ (start + i * G)): neither level of the key is
 * additive, so paths whose keys lie close do not replay each other in runs, and the starting
 * counter keeps two paths whose 32-bit keys meet from drawing the same numbers. */
#define RNG_STEP 0x9E3779B9u
#define RNG_PIXEL_MUL 0x85EBCA6Bu
#define RNG_SAMPLE_MUL 0xC2B2AE35u
typedef struct { uint32_t key, ctr; } rng_t;       /* ctr = start + RNG_STEP * draws made so far */

static inline uint32_t seed_mix(uint64_t seed)
{
    return pcg(pcg((uint32_t)seed) ^ (uint32_t)(seed >> 32));
}

static inline rng_t rng_path(uint32_t smix, uint32_t pixel, uint32_t sample)
{
    rng_t r;
    r.key = pcg(pcg(smix ^ pixel) ^ (sample * RNG_STEP));
    r.ctr = pixel * RNG_PIXEL_MUL + sample * RNG_SAMPLE_MUL;
    return r;
}

static inline float rng_at(const rng_t *r, uint32_t draw)
{
    uint32_t x = pcg(r->key ^ (r->ctr + draw * RNG_STEP));
    return (float)((x >> 8) | 1u) * 0x1p-24f;    /* odd/2^24: in (0,1), exact */
}

static inline double rng_next(rng_t *r)
{
    double x = (double)rng_at(r, 0);
    r->ctr += RNG_STEP;
    return x;
}

float orc_rng_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw)
{
    rng_t r = rng_path(seed_mix(seed), pixel, sample);
    return rng_at(&r, draw);
}

/* main.cpp:42-44 */
static double random_real_libc(double min, double max)
{
    return min + ((double)rand() / RAND_MAX) * (max - min);
}

/* --------------------------------------------------------------- path loop */

typedef struct { rng_t rng; int libc; } rand_src;

static inline double draw(rand_src *rs)
{
    return rs->libc ? random_real_libc(0, 1) : rng_next(&rs->rng);
}

/* Path-length cap of the next renders, in segments; 0 = none (the reference's loop,
 * main.cpp:84, has none).  Set through orc_set_path_cap; see pt_oracle.h. */
static int g_path_cap = 0;

void orc_set_path_cap(int segments) { g_path_cap = segments > 0 ? segments : 0; }

/* 1: russian roulette as the source has it (a survivor's throughput is divided by its survival
 * probability, main.cpp:104); 0: without that division - a hypothesis about the revision that
 * rendered the shipped images, for tests/test_reference_images.py to weigh against the others. */
static int g_rr_compensation = 1;

void orc_set_rr_compensation(int on) { g_rr_compensation = on != 0; }

/* First depth at which russian roulette is played: 5 in the source (main.cpp:97).  A test knob:
 * with a start beyond every path (and a cap) the estimator has no roulette at all, which is what
 * the roulette's estimate must equal in the mean. */
static int g_rr_start = 5;

void orc_set_rr_start(int depth) { g_rr_start = depth >= 0 ? depth : 5; }

/* main.cpp:80-111 */
static v3 trace_ray(const orc_scene *scene, v3 origin, v3 dir, rand_src *rs,
                    uint64_t *segments, uint64_t *max_depth)
{
    v3 color = V(0, 0, 0);
    v3 throughput = V(1, 1, 1);
    const v3 bg = ld3(scene->bg);

    for (int depth = 0; ; depth++) {
        hit_t h;
        (*segments)++;
        if ((uint64_t)(depth + 1) > *max_depth)
            *max_depth = (uint64_t)(depth + 1);
        if (closest_intersection(scene, origin, dir, &h)) {
            const orc_surface *s = &scene->surfaces[h.idx];
            const orc_material *m = &scene->materials[s->material];
            v3 normal = surface_normal(s, h.point);

            double xi0 = 0, xi1 = 0;
            if (m->type == ORC_DIFFUSE) { xi0 = draw(rs); xi1 = draw(rs); }  /* material.hpp:14-15 */
            else if (m->type == ORC_GLASS) { xi0 = draw(rs); }               /* material.hpp:62 */
            v3 new_dir = sample_pdf(m->type, normal, dir, xi0, xi1);
            origin = h.point;

            color = add(color, mul(ld3(m->emittance), throughput));
            throughput = mul(throughput, brdf(m->type, ld3(m->color), normal, new_dir));
            dir = new_dir;

            if (g_path_cap && depth + 1 >= g_path_cap)   /* test knob, absent from the reference */
                break;
            if (depth < g_rr_start)
                continue;
            if (depth + 1 >= ORC_MAX_DEPTH)   /* safety cap, absent from the reference */
                break;

            /* std::clamp(std::max(...), 0.0f, 0.95f), main.cpp:100 */
            float mx = throughput.y < throughput.z ? throughput.z : throughput.y;
            mx = throughput.x < mx ? mx : throughput.x;
            float p = mx < 0.0f ? 0.0f : (0.95f < mx ? 0.95f : mx);
            if (draw(rs) > (double)p)
                break;
            if (g_rr_compensation)
                throughput = scale(throughput, 1 / p);
        } else {
            color = add(color, mul(bg, throughput));
            break;
        }
    }
    return color;
}

/* main.cpp:114-143, one pixel-sample. */
static v3 pixel_sample(const orc_scene *scene, const orc_camera *cam, const m3 *rot,
                       int u, int v, rand_src *rs, uint64_t *segments, uint64_t *max_depth)
{
    double xu, xv, xr = 0, xp = 0;
    if (rs->libc) {
        /* g++ on x86-64 evaluates the arguments of main.cpp:122 right to left */
        xv = draw(rs);
        xu = draw(rs);
    } else {
        xu = draw(rs);
        xv = draw(rs);
    }
    if (cam->dof_on) { xr = draw(rs); xp = draw(rs); }
    v3 origin, dir;
    camera_ray(cam, rot, u, v, xu, xv, xr, xp, &origin, &dir);
    return trace_ray(scene, origin, dir, rs, segments, max_depth);
}

void orc_render(const orc_scene *scene, const orc_camera *cam, uint32_t spp,
                uint32_t first_sample, uint64_t seed, int threads,
                float *accum, orc_counters *counters)
{
    const m3 rot = cam_matrix(cam);
    const uint32_t smix = seed_mix(seed);
    const int W = cam->width, H = cam->height;
    uint64_t tot_seg = 0, tot_max = 0;
    if (threads < 1) threads = 1;

    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads) \
            reduction(+:tot_seg) reduction(max:tot_max)
    for (int u = 0; u < W; u++) {              /* columns outermost, main.cpp:118 */
        for (int v = 0; v < H; v++) {
            const uint32_t pixel = (uint32_t)v * (uint32_t)W + (uint32_t)u;
            float *px = accum + 3 * (size_t)pixel;
            for (uint32_t s = 0; s < spp; s++) {
                rand_src rs;
                rs.libc = 0;
                rs.rng = rng_path(smix, pixel, first_sample + s);
                uint64_t seg = 0, mx = 0;
                v3 c = pixel_sample(scene, cam, &rot, u, v, &rs, &seg, &mx);
                px[0] += c.x; px[1] += c.y; px[2] += c.z;      /* main.cpp:139 */
                tot_seg += seg;
                if (mx > tot_max) tot_max = mx;
            }
        }
    }
    if (counters) {
        counters->samples = (uint64_t)W * H * spp;
        counters->segments = tot_seg;
        counters->max_depth = tot_max;
    }
}

void orc_render_libc_rand(const orc_scene *scene, const orc_camera *cam, uint32_t spp,
                          unsigned seed, float *accum, orc_counters *counters)
{
    const m3 rot = cam_matrix(cam);
    const int W = cam->width, H = cam->height;
    uint64_t tot_seg = 0, tot_max = 0;
    rand_src rs;
    memset(&rs, 0, sizeof rs);
    rs.libc = 1;
    srand(seed);
    for (uint32_t s = 0; s < spp; s++) {        /* one trace_rays pass per sample */
        for (int u = 0; u < W; u++) {
            for (int v = 0; v < H; v++) {
                float *px = accum + 3 * ((size_t)v * W + u);
                v3 c = pixel_sample(scene, cam, &rot, u, v, &rs, &tot_seg, &tot_max);
                px[0] += c.x; px[1] += c.y; px[2] += c.z;
            }
        }
    }
    if (counters) {
        counters->samples = (uint64_t)W * H * spp;
        counters->segments = tot_seg;
        counters->max_depth = tot_max;
    }
}

/* ---------------------------------------------------------------- tone map */

/* main.cpp:146-148: powf(clamp(x,0,1), 1/2.2f)*255 in float, + 0.5 in double,
 * narrowed to the float return type; main.cpp:160 divides the sum by
 * float(TOTAL_SAMPLES); sdl_helpers.hpp:68-70 truncates to Uint8. */
static float to_ldr(float x)
{
    float c = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);   /* glm::clamp = min(max(x,lo),hi) */
    return (float)((double)(powf(c, 1 / 2.2f) * 255) + 0.5);
}

void orc_tonemap(const float *sum, int64_t n, uint32_t total_samples, uint8_t *rgb)
{
    const float denom = (float)total_samples;
    for (int64_t i = 0; i < 3 * n; i++)
        rgb[i] = (uint8_t)to_ldr(sum[i] / denom);
}

/* ------------------------------------------------------------------ scenes */

typedef struct {
    orc_surface *surfaces;
    orc_material *materials;
    int n;
} builder;

static int add_material(builder *b, int type, float r, float g, float bl,
                        float er, float eg, float eb)
{
    orc_material *m = &b->materials[b->n];
    m->type = type;
    m->color[0] = r; m->color[1] = g; m->color[2] = bl;
    m->emittance[0] = er; m->emittance[1] = eg; m->emittance[2] = eb;
    return b->n;
}

typedef struct { int type; float c[3]; float e[3]; } mat;

static mat M(int type, float r, float g, float b, float er, float eg, float eb)
{
    mat m = { type, {r, g, b}, {er, eg, eb} };
    return m;
}

static void add_sphere(builder *b, float x, float y, float z, float radius, mat m)
{
    orc_surface *s = &b->surfaces[b->n];
    memset(s, 0, sizeof *s);
    s->kind = ORC_SPHERE;
    s->material = add_material(b, m.type, m.c[0], m.c[1], m.c[2], m.e[0], m.e[1], m.e[2]);
    s->a[0] = x; s->a[1] = y; s->a[2] = z;
    s->b[0] = radius;
    b->n++;
}

static void add_triangle(builder *b, const float v0[3], const float v1[3], const float v2[3], mat m)
{
    orc_surface *s = &b->surfaces[b->n];
    memset(s, 0, sizeof *s);
    s->kind = ORC_TRIANGLE;
    s->material = add_material(b, m.type, m.c[0], m.c[1], m.c[2], m.e[0], m.e[1], m.e[2]);
    memcpy(s->a, v0, 12); memcpy(s->b, v1, 12); memcpy(s->c, v2, 12);
    b->n++;
}

#define TRI(b, x0,y0,z0, x1,y1,z1, x2,y2,z2, m) do { \
    const float p0_[3] = {x0,y0,z0}, p1_[3] = {x1,y1,z1}, p2_[3] = {x2,y2,z2}; \
    add_triangle(b, p0_, p1_, p2_, m); } while (0)

int orc_builtin_scene(const char *name, int width, int height,
                      orc_surface *surfaces, orc_material *materials,
                      int32_t *n_surfaces, float bg[3], orc_camera *cam)
{
    builder B = { surfaces, materials, 0 };
    builder *b = &B;
    const float focal = (float)height;   /* every scene passes SCREEN_HEIGHT */
    float pos[3] = {0, 0, 0};
    float rx = 0, ry = 0, bgv[3] = {0, 0, 0};
    int dof_on = 0;
    float aperture = 0.15f, focal_length = 2.0f;     /* scene.hpp:25 */

    if (!strcmp(name, "dof_demo")) {                 /* scene.hpp:34-57 */
        mat redish  = M(ORC_DIFFUSE, 1, 0, 0.5f, 0, 0, 0);
        mat blueish = M(ORC_DIFFUSE, 0.5f, 0, 1, 0, 0, 0);
        mat white   = M(ORC_DIFFUSE, 1, 1, 1, 0.5f * 1, 0.5f * 1, 0.5f * 1);
        mat red     = M(ORC_DIFFUSE, 1, 0, 0, 1.0f * 1, 1.0f * 0, 1.0f * 0);
        mat blue    = M(ORC_DIFFUSE, 0, 0, 1, 1.0f * 0, 1.0f * 0, 1.0f * 1);
        mat purple  = M(ORC_DIFFUSE, 1, 0, 1, 1.0f * 1, 1.0f * 0, 1.0f * 1);
        float r = 1000;
        add_sphere(b, 0, r, 0, r, white);
        add_sphere(b, -1, -1, 1, 1, blueish);
        add_sphere(b, 1.5f, -1, 2, 1, redish);
        add_sphere(b, -35, -20, 50, 1, red);
        add_sphere(b, 3, -12, 50, 1, blue);
        add_sphere(b, 35, -16, 50, 1, purple);
        pos[0] = 0; pos[1] = -1; pos[2] = -3;
        bgv[0] = bgv[1] = bgv[2] = 0.1f * 1;
        dof_on = 1; aperture = (float)0.3; focal_length = 5;
    } else if (!strcmp(name, "test")) {              /* scene.hpp:59-71 */
        for (int i = 0; i <= 10; i++) {
            float v = (float)(i / 10.0);
            add_sphere(b, (float)i, (float)-0.5, 0, (float)0.5, M(ORC_DIFFUSE, 1, v, 0, 0, 0, 0));
        }
        pos[0] = 5; pos[1] = 0; pos[2] = -7;
        bgv[0] = bgv[1] = bgv[2] = 1;
    } else if (!strcmp(name, "orange_plane")) {      /* scene.hpp:73-93 */
        mat orange = M(ORC_DIFFUSE, 1.00f, 0.40f, 0.00f, 0, 0, 0);
        mat white  = M(ORC_DIFFUSE, 1, 1, 1, 0, 0, 0);
        mat lamp   = M(ORC_DIFFUSE, 1, 1, 1, 5.0f * 1.0f, 5.0f * 1.0f, 5.0f * 1.0f);
        mat mirror = M(ORC_MIRROR, 1, 1, 1, 0, 0, 0);
        mat glass  = M(ORC_GLASS, 1, 1, 1, 0, 0, 0);
        add_sphere(b, -1.5f, 0.0f, 0.0f, (float)0.75, mirror);
        add_sphere(b,  1.5f, 0.0f, 0.0f, (float)0.75, glass);
        add_sphere(b,  0.0f, 0.0f, 2.0f, (float)0.75, white);
        add_sphere(b,  0.0f, -10.0f, 0.0f, 5, lamp);
        TRI(b, -100, (float)0.75, 100,  100, (float)0.75, 100,  0, (float)0.75, -100, orange);
        pos[0] = 0; pos[1] = 0; pos[2] = -3;
        dof_on = 1; aperture = (float)0.15; focal_length = 5;
    } else if (!strcmp(name, "red_corner")) {        /* scene.hpp:95-112 */
        mat red    = M(ORC_DIFFUSE, 1.00f, 0.05f, 0.05f, 0, 0, 0);
        mat black  = M(ORC_DIFFUSE, 0.05f, 0.05f, 0.05f, 0, 0, 0);
        mat lamp   = M(ORC_DIFFUSE, 1, 1, 1, 1.5f * 1.0f, 1.5f * 1.0f, 1.5f * 1.0f);
        mat mirror = M(ORC_MIRROR, 1, 1, 1, 0, 0, 0);
        mat glass  = M(ORC_GLASS, 1, 1, 1, 0, 0, 0);
        add_sphere(b, 0, -1, 2, 1, black);
        add_sphere(b, (float)-1.25, -1, 0, 1, mirror);
        add_sphere(b, (float)1.25, -1, 0, 1, glass);
        TRI(b, 0, 0, -100,  -10, 0, 3,  10, 0, 3, red);
        TRI(b, 0, -100, 3,  -10, 0, 3,  10, 0, 3, lamp);
        pos[0] = 0; pos[1] = (float)-4.5; pos[2] = -3;
        rx = (float)-M_PI_4;
    } else if (!strcmp(name, "sunrise")) {           /* scene.hpp:114-127 */
        mat orange = M(ORC_DIFFUSE, 1.00f, 0.40f, 0.00f, 0, 0, 0);
        mat white  = M(ORC_DIFFUSE, 0.95f, 0.95f, 0.95f, 0, 0, 0);
        mat black  = M(ORC_DIFFUSE, 0.05f, 0.05f, 0.05f, 0, 0, 0);
        TRI(b, -100, 0, 100,  100, 0, 100,  0, 0, -100, black);
        add_sphere(b, -2, -1, 0, 1, white);
        add_sphere(b, -1, -1, 7, 1, orange);
        pos[0] = 0; pos[1] = -1; pos[2] = -2;
        ry = (float)5.78;
        bgv[0] = (float)0.7; bgv[1] = (float)0.7; bgv[2] = 1;
        dof_on = 1; aperture = (float)0.2; focal_length = 9;
    } else if (!strcmp(name, "cornell_2")) {         /* scene.hpp:129-156 */
        mat red    = M(ORC_DIFFUSE, 1.00f, 0.05f, 0.05f, 0, 0, 0);
        mat purple = M(ORC_DIFFUSE, 1, 0.05f, 1, 0, 0, 0);
        mat yellow = M(ORC_DIFFUSE, 1, 1, 0.05f, 0, 0, 0);
        mat cyan   = M(ORC_DIFFUSE, 0.05f, 1, 1, 0, 0, 0);
        mat white  = M(ORC_DIFFUSE, 1, 1, 1, 0, 0, 0);
        mat light  = M(ORC_DIFFUSE, 1, 1, 1, 2.0f * 1, 2.0f * 1, 2.0f * 1);
        mat glass  = M(ORC_GLASS, 1, 1, 1, 0, 0, 0);
        mat mirror = M(ORC_MIRROR, 1, 1, 1, 0, 0, 0);
        TRI(b, -2, 1, -1,   0, 1, 10,   2, 1, -1, red);
        TRI(b, -1, -2, -1,  -1, 0, 10,  -1, 2, -1, purple);
        TRI(b, 1, -2, -1,   1, 2, -1,   1, 0, 10, yellow);
        TRI(b, -2, -1, -1,  2, -1, -1,  0, -1, 10, cyan);
        TRI(b, -2, 1, 1,    0, -10, 1,  2, 1, 1, white);
        add_sphere(b, (float)-0.32, (float)0.6, (float)0.26, (float)0.4, mirror);
        add_sphere(b, (float)0.35, (float)0.6, (float)-0.39, (float)0.4, glass);
        add_sphere(b, 0, (float)-1.8, 0, 1, light);
        pos[0] = 0; pos[1] = 0; pos[2] = -3;
        bgv[0] = bgv[1] = bgv[2] = 0.0f * 1;
    } else if (!strcmp(name, "cornell")) {           /* scene.hpp:158-201 */
        mat red    = M(ORC_DIFFUSE, 1.00f, 0.05f, 0.05f, 0, 0, 0);
        mat blue   = M(ORC_DIFFUSE, 0.05f, 0.05f, 1.00f, 0, 0, 0);
        mat green  = M(ORC_DIFFUSE, 0.05f, 1, 0.05f, 0, 0, 0);
        mat purple = M(ORC_DIFFUSE, 1, 0.05f, 1, 0, 0, 0);
        mat yellow = M(ORC_DIFFUSE, 1, 1, 0.05f, 0, 0, 0);
        mat cyan   = M(ORC_DIFFUSE, 0.05f, 1, 1, 0.7f * 1, 0.7f * 1, 0.7f * 1);
        mat white  = M(ORC_DIFFUSE, 1, 1, 1, 0, 0, 0);
        TRI(b, -2, 1, -1,   0, 1, 10,   2, 1, -1, green);
        TRI(b, -1, -2, -1,  -1, 0, 10,  -1, 2, -1, purple);
        TRI(b, 1, -2, -1,   1, 2, -1,   1, 0, 10, yellow);
        TRI(b, -2, -1, -1,  2, -1, -1,  0, -1, 10, cyan);
        TRI(b, -2, 1, 1,    0, -10, 1,  2, 1, 1, white);
        add_sphere(b, (float)0.35, (float)0.7, (float)-0.39, (float)0.3, red);
        const float A[3] = {-0.5243243243243243f, 1.0f, -0.1099099099099099f};
        const float Bv[3] = {0.04504504504504503f, 1.0f, 0.06666666666666665f};
        const float C[3] = {-0.7009009009009008f, 1.0f, 0.463063063063063f};
        const float D[3] = {-0.13153153153153152f, 1.0f, 0.6432432432432433f};
        const float E[3] = {-0.5243243243243243f, -0.18918918918918926f, -0.1099099099099099f};
        const float F[3] = {0.04504504504504503f, -0.18918918918918926f, 0.06666666666666665f};
        const float G[3] = {-0.7009009009009008f, -0.18918918918918926f, 0.463063063063063f};
        const float Hh[3] = {-0.13153153153153152f, -0.18918918918918926f, 0.6432432432432433f};
        add_triangle(b, E, Bv, A, blue);
        add_triangle(b, E, F, Bv, blue);
        add_triangle(b, F, D, Bv, blue);
        add_triangle(b, F, Hh, D, blue);
        add_triangle(b, Hh, C, D, blue);
        add_triangle(b, Hh, G, C, blue);
        add_triangle(b, G, E, C, blue);
        add_triangle(b, E, A, C, blue);
        add_triangle(b, G, F, E, blue);
        add_triangle(b, G, Hh, F, blue);
        pos[0] = 0; pos[1] = 0; pos[2] = -3;
        bgv[0] = bgv[1] = bgv[2] = 0.1f * 1;
    } else {
        return -1;
    }

    orc_camera_init(cam, pos, rx, ry, focal, 1, width, height);
    cam->dof_on = dof_on;
    cam->dof_aperture = aperture;
    cam->dof_focal_length = focal_length;
    memcpy(bg, bgv, sizeof bgv);
    *n_surfaces = B.n;
    return 0;
}
