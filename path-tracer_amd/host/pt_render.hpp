// The render step on the GPU behind the reference's call surface.
//
//   void trace_rays(Scene&, BUFFER&)   -- src/main.cpp:114-143: adds one sample
//       per pixel to the host buffer and increments TOTAL_SAMPLES.
//   pt::Renderer                        -- the batched form: the running sums
//       stay in device memory, many samples per call, read back on demand.
//
// Both flatten the Scene into the plain arrays of include/pt_amd.h and call
// the HIP library; nothing here computes radiance on the CPU.  Include this
// header in exactly one translation unit (it defines TOTAL_SAMPLES, as the
// reference's main.cpp does).
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <typeinfo>

#include "pt_types.hpp"

// src/main.cpp:31-32
typedef std::vector<std::vector<vec3>> BUFFER;   // buffer[u = x][v = y], running sums
inline unsigned TOTAL_SAMPLES = 0;

namespace pt {

class Error : public std::runtime_error {
  public:
    Error(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};

// A Scene in the flat form of the C ABI.
struct FlatScene {
    std::vector<pt_surface> surfaces;
    std::vector<pt_material> materials;   // one per surface: each Surface owns its Material by value
    float bg[3];
    pt_camera camera;

    bool same_geometry(const FlatScene &o) const
    {
        return surfaces.size() == o.surfaces.size() && std::memcmp(bg, o.bg, sizeof bg) == 0 &&
               (surfaces.empty() ||
                (std::memcmp(surfaces.data(), o.surfaces.data(), surfaces.size() * sizeof(pt_surface)) == 0 &&
                 std::memcmp(materials.data(), o.materials.data(), materials.size() * sizeof(pt_material)) == 0));
    }
    bool same_camera(const FlatScene &o) const { return std::memcmp(&camera, &o.camera, sizeof camera) == 0; }
};

// The camera and depth-of-field settings of a Scene for a width x height image.
inline pt_camera flatten_camera(const Scene &scene, int width, int height)
{
    pt_camera out{};
    const Camera &cam = scene.camera;
    const vec3 &p = cam.get_position();
    out.pos[0] = p.x; out.pos[1] = p.y; out.pos[2] = p.z;
    const mat3 &r = cam.get_rotation_matrix();
    for (int c = 0; c < 3; c++) {
        out.rot[3 * c + 0] = r[c].x;
        out.rot[3 * c + 1] = r[c].y;
        out.rot[3 * c + 2] = r[c].z;
    }
    out.focal = cam.get_focal_length();
    out.width = width;
    out.height = height;
    out.dof_on = scene.DOF_on ? 1 : 0;
    out.dof_focal_length = scene.DOF_focal_length;
    out.dof_aperture = scene.DOF_aperture_size;
    return out;
}

inline FlatScene flatten(const Scene &scene, int width, int height)
{
    FlatScene f{};
    f.surfaces.reserve(scene.surfaces.size());
    f.materials.reserve(scene.surfaces.size());
    for (const auto &ptr : scene.surfaces) {          // scene order is kept: it breaks distance ties
        pt_surface s = ptr->flat();
        s.material = static_cast<int32_t>(f.materials.size());
        pt_material m{};
        m.type = ptr->material.type;
        const vec3 &c = ptr->material.color, &e = ptr->material.emittance;
        m.color[0] = c.x; m.color[1] = c.y; m.color[2] = c.z;
        m.emittance[0] = e.x; m.emittance[1] = e.y; m.emittance[2] = e.z;
        f.surfaces.push_back(s);
        f.materials.push_back(m);
    }
    f.bg[0] = scene.BG_COLOR.x; f.bg[1] = scene.BG_COLOR.y; f.bg[2] = scene.BG_COLOR.z;
    f.camera = flatten_camera(scene, width, height);
    return f;
}

// RAII owner of one GPU context, or of a group of them (several GPUs of one node rendering
// one image: pt_group_* of the C ABI; samples are shared by sample number and the images
// combined on the first device by one RCCL reduce whenever the sums are read).
class Renderer {
    pt_ctx *ctx_ = nullptr;          // the context, or the group's first one
    pt_group *group_ = nullptr;
    FlatScene loaded_{};
    bool have_scene_ = false;
    uint32_t next_sample_ = 0;

    void check(int rc) const
    {
        if (rc != PT_OK) throw Error(rc, pt_last_error(ctx_));
    }
    void gcheck(int rc) const
    {
        if (rc != PT_OK) throw Error(rc, pt_group_last_error(group_));
    }
    // a group's sums live on its members until they are reduced
    void settle() const
    {
        if (group_) gcheck(pt_group_read_accum(group_, nullptr, nullptr));
    }

  public:
    explicit Renderer(int device = 0)
    {
        const int rc = pt_create(device, &ctx_);
        if (rc != PT_OK) throw Error(rc, pt_global_error());
    }
    // One renderer over several devices (a single device makes a group of one).
    explicit Renderer(const std::vector<int> &devices)
    {
        const int rc = pt_group_create(devices.data(), static_cast<int>(devices.size()), &group_);
        if (rc != PT_OK) throw Error(rc, pt_group_last_error(nullptr));
        ctx_ = pt_group_context(group_, 0);
    }
    ~Renderer()
    {
        if (group_) pt_group_destroy(group_);
        else pt_destroy(ctx_);
    }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    pt_ctx *context() const { return ctx_; }
    int devices() const { return group_ ? pt_group_size(group_) : 1; }

    // Uploads whatever changed since the last call.  A changed camera, DOF
    // setting, image size or scene clears the accumulated image, the condition
    // under which the reference clears its buffer (src/main.cpp:258-269).
    // Returns true when the image was cleared.
    bool sync(const Scene &scene, int width, int height)
    {
        FlatScene f = flatten(scene, width, height);
        bool cleared = false;
        if (!have_scene_ || !f.same_geometry(loaded_)) {
            const int32_t n = static_cast<int32_t>(f.surfaces.size()), m = static_cast<int32_t>(f.materials.size());
            if (group_) gcheck(pt_group_set_scene(group_, f.surfaces.data(), n, f.materials.data(), m, f.bg));
            else check(pt_set_scene(ctx_, f.surfaces.data(), n, f.materials.data(), m, f.bg));
            cleared = true;
        }
        if (cleared || !f.same_camera(loaded_)) {
            set_camera(f.camera);   // clears
            cleared = true;
        }
        loaded_ = std::move(f);
        have_scene_ = true;
        return cleared;
    }

    // The same for a scene whose surfaces have not changed since the last sync
    // (an interactive session moves only the camera): compares and uploads the
    // camera and depth-of-field settings alone, without flattening the surfaces.
    bool sync_camera(const Scene &scene, int width, int height)
    {
        if (!have_scene_) return sync(scene, width, height);
        const pt_camera cam = flatten_camera(scene, width, height);
        if (std::memcmp(&cam, &loaded_.camera, sizeof cam) == 0) return false;
        set_camera(cam);   // clears
        loaded_.camera = cam;
        return true;
    }

    void set_camera(const pt_camera &cam)
    {
        if (group_) gcheck(pt_group_set_camera(group_, &cam));
        else check(pt_set_camera(ctx_, &cam));
    }

    void clear()
    {
        if (group_) gcheck(pt_group_clear(group_));
        else check(pt_clear(ctx_));
    }

    // Adds spp samples per pixel; sample numbers continue across calls so no
    // two passes repeat a random stream.
    void render(uint32_t spp, uint64_t seed = 1, int mode = PT_MODE_MEGAKERNEL | PT_NUM_EXACT)
    {
        if (group_) gcheck(pt_group_render(group_, spp, next_sample_, seed, mode));
        else check(pt_render(ctx_, spp, next_sample_, seed, mode));
        next_sample_ += spp;
    }

    uint64_t total_samples() const
    {
        uint64_t n = 0;
        settle();
        check(pt_read_accum(ctx_, nullptr, &n));
        return n;
    }

    // Running sums, row-major (pixel (u,v) at 3*(v*width+u)).
    std::vector<float> read_sums() const
    {
        std::vector<float> sums(static_cast<size_t>(loaded_.camera.width) * loaded_.camera.height * 3);
        settle();
        check(pt_read_accum(ctx_, sums.data(), nullptr));
        return sums;
    }

    // The same without a copy of the caller's: a view of page-locked memory the context
    // owns, valid until the next call of this function.
    const float *map_sums() const
    {
        const float *sums = nullptr;
        settle();
        check(pt_map_accum(ctx_, &sums, nullptr));
        return sums;
    }

    // Replaces the device image with host sums (row-major) and their sample count.
    void write_sums(const std::vector<float> &sums, uint64_t total_samples)
    {
        if (sums.size() != static_cast<size_t>(loaded_.camera.width) * loaded_.camera.height * 3)
            throw Error(PT_E_INVALID, "write_sums: size does not match the image");
        if (group_) throw Error(PT_E_INVALID, "write_sums: a group's image is the sum of its members' shares");
        check(pt_write_accum(ctx_, sums.data(), total_samples));
    }

    // 8-bit image as the reference's draw() shows it (src/main.cpp:146-162).
    std::vector<uint8_t> tonemap() const
    {
        std::vector<uint8_t> rgb(static_cast<size_t>(loaded_.camera.width) * loaded_.camera.height * 3);
        settle();
        check(pt_tonemap_rgb8(ctx_, rgb.data()));
        return rgb;
    }

    // Counters of the last render on the context (a group's first member).
    pt_counters counters() const
    {
        pt_counters c{};
        check(pt_get_counters(ctx_, &c));
        return c;
    }
};

// The renderers behind the free functions below (trace_rays, the Material and Surface gateways):
// made on first use, released by pt::shutdown().  They are deliberately NOT objects with static
// storage: such an object's destructor would call into the HIP runtime while the process is being
// torn down, in an order relative to the runtime's own teardown that nothing guarantees (a
// segmentation fault at exit, one run in some tens, is what that looks like).  A program that
// never calls shutdown() leaves the contexts to the operating system.
struct SharedRenderers {
    std::mutex lock;
    std::unique_ptr<Renderer> render, probe;
};

inline SharedRenderers &shared_renderers()
{
    static SharedRenderers *shared = new SharedRenderers;     // never deleted, see above
    return *shared;
}

inline Renderer &default_renderer()
{
    SharedRenderers &s = shared_renderers();
    std::lock_guard<std::mutex> hold(s.lock);
    if (!s.render) s.render.reset(new Renderer(0));
    return *s.render;
}

// Releases the renderers of the free functions (their device memory, streams and events); they
// are made again on the next use.  Call it before main returns, not from a static destructor.
inline void shutdown()
{
    SharedRenderers &s = shared_renderers();
    std::lock_guard<std::mutex> hold(s.lock);
    s.render.reset();
    s.probe.reset();
}

// ---- single-query gateways used by Material and Surface methods ----

inline vec3 gpu_sample_pdf(MaterialType type, vec3 normal, vec3 incident, float xi0, float xi1)
{
    const int32_t t = type;
    const float col[3] = {1, 1, 1}, n[3] = {normal.x, normal.y, normal.z};
    const float in[3] = {incident.x, incident.y, incident.z}, xi[2] = {xi0, xi1};
    float dir[3], f[3];
    Renderer &r = default_renderer();
    const int rc = pt_eval_bsdf(r.context(), &t, col, n, in, xi, 1, dir, f, PT_NUM_EXACT);
    if (rc != PT_OK) throw Error(rc, pt_last_error(r.context()));
    return vec3(dir[0], dir[1], dir[2]);
}

inline vec3 gpu_brdf(MaterialType type, vec3 color, vec3 normal, vec3 reflected)
{
    const int32_t t = type;
    const float col[3] = {color.x, color.y, color.z}, n[3] = {normal.x, normal.y, normal.z};
    const float refl[3] = {reflected.x, reflected.y, reflected.z};
    float f[3];
    Renderer &r = default_renderer();
    const int rc = pt_eval_brdf(r.context(), &t, col, n, refl, 1, f);
    if (rc != PT_OK) throw Error(rc, pt_last_error(r.context()));
    return vec3(f[0], f[1], f[2]);
}

inline bool gpu_intersect(const pt_surface &surface, vec3 start, vec3 dir, vec3 &point, float &distance)
{
    // A context of its own, so that one-surface queries never disturb the scene
    // loaded in the default renderer.
    Renderer *probe_ptr;
    {
        SharedRenderers &shared = shared_renderers();
        std::lock_guard<std::mutex> hold(shared.lock);
        if (!shared.probe) shared.probe.reset(new Renderer(0));
        probe_ptr = shared.probe.get();
    }
    Renderer &probe = *probe_ptr;
    pt_surface s = surface;
    s.material = 0;
    pt_material m{};
    m.type = PT_DIFFUSE;
    const float bg[3] = {0, 0, 0};
    int rc = pt_set_scene(probe.context(), &s, 1, &m, 1, bg);
    if (rc != PT_OK) throw Error(rc, pt_last_error(probe.context()));
    const float o[3] = {start.x, start.y, start.z}, d[3] = {dir.x, dir.y, dir.z};
    int32_t idx = -1;
    float pt3[3] = {0, 0, 0};
    rc = pt_intersect_rays(probe.context(), o, d, 1, &idx, &distance, pt3, PT_NUM_EXACT);
    if (rc != PT_OK) throw Error(rc, pt_last_error(probe.context()));
    point = vec3(pt3[0], pt3[1], pt3[2]);
    return idx == 0;
}

inline float gpu_uniform()
{
    static uint32_t draw = 0;
    const uint32_t zero = 0, d = draw++;
    float out = 0;
    Renderer &r = default_renderer();
    const int rc = pt_rng_uniform(r.context(), 0x5eedull, &zero, &zero, &d, 1, &out);
    if (rc != PT_OK) throw Error(rc, pt_last_error(r.context()));
    return out;
}

}  // namespace pt

// src/main.cpp:114-143 for spp samples per pixel in one call.  buffer is indexed buffer[u][v]
// with u the column; its size gives the image size.  spp samples per pixel are added to it and
// TOTAL_SAMPLES grows by spp: spp calls of the reference's function.
inline void trace_rays(Scene &scene, BUFFER &buffer, uint32_t spp)
{
    if (buffer.empty() || buffer[0].empty() || spp == 0) return;
    const int width = static_cast<int>(buffer.size()), height = static_cast<int>(buffer[0].size());
    pt::Renderer &r = pt::default_renderer();
    r.sync(scene, width, height);
    r.clear();                                   // the host buffer holds the running sums
    r.render(spp);
    const float *sums = r.map_sums();
    // the device image is row-major, BUFFER is a vector of columns: a few columns at a
    // time, so that a row's pixels are read while their cache lines are still there
    // (column blocks are independent: spread over the host's cores when built with OpenMP, which
    // the reference links too, CMakeLists.txt:7)
    constexpr int COLUMNS = 16;
#pragma omp parallel for schedule(static)
    for (int u0 = 0; u0 < width; u0 += COLUMNS) {
        const int u1 = u0 + COLUMNS < width ? u0 + COLUMNS : width;
        for (int v = 0; v < height; v++) {
            const float *p = sums + 3 * (static_cast<size_t>(v) * width + u0);
            for (int u = u0; u < u1; u++, p += 3) buffer[u][v] += vec3(p[0], p[1], p[2]);
        }
    }
    TOTAL_SAMPLES += spp;
}

// The reference's own signature, void(Scene &, BUFFER &) (src/main.cpp:114): one sample per pixel.
// A function of its own, not a default argument of the one above, so that its address is a
// void (*)(Scene &, BUFFER &) like the reference's.
inline void trace_rays(Scene &scene, BUFFER &buffer) { trace_rays(scene, buffer, 1u); }
