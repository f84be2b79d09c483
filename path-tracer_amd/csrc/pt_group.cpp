// Several GPUs of one node behind the C ABI (pt_group_* in include/pt_amd.h): one context
// per GPU, the samples of every render call partitioned by global sample number, the
// per-GPU images combined on the first GPU with ONE ncclReduce over RCCL (xGMI between the
// GPUs of a node).  One host thread drives all devices: every device has a stream of its
// own, renders are enqueued on all of them before anything waits, and the reduce calls are
// fused between ncclGroupStart and ncclGroupEnd.
//
// The reference is a single thread and has no counterpart; what makes the partition exact
// is that its image is a running sum (src/main.cpp:139) and that a path's random stream
// here depends only on (seed, pixel, global sample number).
//
// RCCL is loaded with dlopen when the first group is created, so the library itself does
// not depend on it (a host with one GPU never loads its 500 MB).
#include "../../include/pt_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the functions are resolved at run time

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

// Loads RCCL once per process.  Returns nullptr (g_rccl.error says why) when it is absent.
Rccl *rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        g_rccl.error = std::string("cannot load librccl.so: ") + dlerror();
        return nullptr;
    }
    Rccl r;
    r.handle = h;
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) { ok = false; g_rccl.error = std::string("librccl.so lacks ") + name; }
        return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(h);
        return nullptr;
    }
    g_rccl = r;
    return &g_rccl;
}

thread_local std::string g_group_error;

}  // namespace

struct pt_group {
    std::vector<int> devices;
    std::vector<pt_ctx *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<float *> part;        // per device: its share of the image, height*width*3 running sums
    float *sum = nullptr;             // first device: the combined image
    size_t floats = 0;
    uint64_t total_samples = 0;
    bool dirty = false;               // the parts hold samples the first context's image lacks
    bool poisoned = false;            // a render failed after some members had added their shares (until pt_group_clear)
    std::string error;
};

namespace {

int gfail(pt_group *g, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->error = buf;
    g_group_error = buf;
    return code;
}

// A failed call on member i: its context's message becomes the group's.
int member_failed(pt_group *g, size_t i, int rc)
{
    return gfail(g, rc, "device %d: %s", g->devices[i], pt_last_error(g->ctx[i]));
}

#define GROUP_HIP(g, call)                                                                           \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return gfail(g, PT_E_HIP, "%s: %s", #call, hipGetErrorString(e_));     \
    } while (0)

void free_images(pt_group *g)
{
    for (size_t i = 0; i < g->part.size(); i++)
        if (g->part[i]) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipFree(g->part[i]);
            g->part[i] = nullptr;
        }
    if (g->sum) {
        (void)hipSetDevice(g->devices[0]);
        (void)hipFree(g->sum);
        g->sum = nullptr;
    }
    g->floats = 0;
}

int wait_all(pt_group *g)
{
    for (size_t i = 0; i < g->ctx.size(); i++) {
        GROUP_HIP(g, hipSetDevice(g->devices[i]));
        GROUP_HIP(g, hipStreamSynchronize(g->streams[i]));
    }
    return PT_OK;
}

}  // namespace

extern "C" {

const char *pt_group_last_error(pt_group *g) { return g ? g->error.c_str() : g_group_error.c_str(); }

void pt_group_share(uint32_t spp_total, int n, int rank, uint32_t first_sample, uint32_t *first, uint32_t *count)
{
    if (n < 1 || rank < 0 || rank >= n) {           // no such member: an empty share
        if (first) *first = first_sample;
        if (count) *count = 0;
        return;
    }
    const uint64_t lo = (uint64_t)rank * spp_total / (uint64_t)n, hi = (uint64_t)(rank + 1) * spp_total / (uint64_t)n;
    if (first) *first = first_sample + (uint32_t)lo;
    if (count) *count = (uint32_t)(hi - lo);
}

void pt_group_destroy(pt_group *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); i++) {
        (void)hipSetDevice(g->devices[i]);
        if (i < g->streams.size() && g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
    }
    Rccl *r = g->comms.empty() ? nullptr : rccl();
    if (r)
        for (ncclComm_t c : g->comms)
            if (c) (void)r->CommDestroy(c);
    free_images(g);
    for (size_t i = 0; i < g->ctx.size(); i++) {
        if (g->ctx[i]) {
            (void)pt_set_stream(g->ctx[i], nullptr);
            pt_destroy(g->ctx[i]);
        }
        if (i < g->streams.size() && g->streams[i]) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipStreamDestroy(g->streams[i]);
        }
    }
    delete g;
}

int pt_group_create(const int *devices, int n_devices, pt_group **out)
{
    if (!out) return gfail(nullptr, PT_E_INVALID, "out is null");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64)
        return gfail(nullptr, PT_E_INVALID, "a group has 1 to 64 devices");
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j])
                return gfail(nullptr, PT_E_INVALID, "device %d named twice", devices[i]);
    Rccl *r = rccl();
    if (!r) return gfail(nullptr, PT_E_HIP, "%s", g_rccl.error.c_str());
    pt_group *g = new (std::nothrow) pt_group;
    if (!g) return gfail(nullptr, PT_E_NOMEM, "out of host memory");
    try {
        g->devices.assign(devices, devices + n_devices);
        g->ctx.assign(n_devices, nullptr);
        g->streams.assign(n_devices, nullptr);
        g->part.assign(n_devices, nullptr);
    } catch (const std::bad_alloc &) {
        delete g;
        return gfail(nullptr, PT_E_NOMEM, "out of host memory");
    }
    for (int i = 0; i < n_devices; i++) {
        int rc = pt_create(devices[i], &g->ctx[i]);
        if (rc != PT_OK) {
            gfail(nullptr, rc, "device %d: %s", devices[i], pt_global_error());
            pt_group_destroy(g);
            return rc;
        }
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) {
            gfail(nullptr, PT_E_HIP, "stream on device %d: %s", devices[i], hipGetErrorString(e));
            pt_group_destroy(g);
            return PT_E_HIP;
        }
        if ((rc = pt_set_stream(g->ctx[i], g->streams[i])) != PT_OK) {
            gfail(nullptr, rc, "device %d: %s", devices[i], pt_last_error(g->ctx[i]));
            pt_group_destroy(g);
            return rc;
        }
    }
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    const ncclResult_t nr = r->CommInitAll(comms.data(), n_devices, devices);
    if (nr != ncclSuccess) {
        gfail(nullptr, PT_E_HIP, "ncclCommInitAll over %d devices: %s", n_devices, r->GetErrorString(nr));
        pt_group_destroy(g);
        return PT_E_HIP;
    }
    g->comms = comms;
    *out = g;
    return PT_OK;
}

int pt_group_size(pt_group *g) { return g ? (int)g->ctx.size() : 0; }

pt_ctx *pt_group_context(pt_group *g, int i)
{
    return g && i >= 0 && i < (int)g->ctx.size() ? g->ctx[i] : nullptr;
}

int pt_group_set_scene(pt_group *g, const pt_surface *surfaces, int32_t n_surfaces,
                       const pt_material *materials, int32_t n_materials, const float bg[3])
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    for (size_t i = 0; i < g->ctx.size(); i++) {
        const int rc = pt_set_scene(g->ctx[i], surfaces, n_surfaces, materials, n_materials, bg);
        if (rc != PT_OK) return member_failed(g, i, rc);
    }
    return PT_OK;
}

int pt_group_set_accel(pt_group *g, int accel)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    for (size_t i = 0; i < g->ctx.size(); i++) {
        const int rc = pt_set_accel(g->ctx[i], accel);
        if (rc != PT_OK) return member_failed(g, i, rc);
    }
    return PT_OK;
}

int pt_group_set_max_depth(pt_group *g, int max_segments)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    for (size_t i = 0; i < g->ctx.size(); i++) {
        const int rc = pt_set_max_depth(g->ctx[i], max_segments);
        if (rc != PT_OK) return member_failed(g, i, rc);
    }
    return PT_OK;
}

int pt_group_clear(pt_group *g)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    if (!g->floats) return gfail(g, PT_E_INVALID, "pt_group_set_camera has not been called");
    for (size_t i = 0; i < g->ctx.size(); i++) {
        GROUP_HIP(g, hipSetDevice(g->devices[i]));
        GROUP_HIP(g, hipMemsetAsync(g->part[i], 0, g->floats * sizeof(float), g->streams[i]));
    }
    const int rc = pt_clear(g->ctx[0]);
    if (rc != PT_OK) return member_failed(g, 0, rc);
    g->total_samples = 0;
    g->dirty = false;
    g->poisoned = false;
    return PT_OK;
}

int pt_group_set_camera(pt_group *g, const pt_camera *cam)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    for (size_t i = 0; i < g->ctx.size(); i++) {
        const int rc = pt_set_camera(g->ctx[i], cam);      // validates; clears the context's own image
        if (rc != PT_OK) return member_failed(g, i, rc);
    }
    const size_t floats = (size_t)cam->width * cam->height * 3;
    if (floats != g->floats) {
        int rc = wait_all(g);
        if (rc != PT_OK) return rc;
        free_images(g);
        for (size_t i = 0; i < g->ctx.size(); i++) {
            GROUP_HIP(g, hipSetDevice(g->devices[i]));
            if (hipMalloc((void **)&g->part[i], floats * sizeof(float)) != hipSuccess) {
                free_images(g);
                return gfail(g, PT_E_NOMEM, "hipMalloc of %zu bytes on device %d failed", floats * sizeof(float), g->devices[i]);
            }
        }
        GROUP_HIP(g, hipSetDevice(g->devices[0]));
        if (hipMalloc((void **)&g->sum, floats * sizeof(float)) != hipSuccess) {
            free_images(g);
            return gfail(g, PT_E_NOMEM, "hipMalloc of %zu bytes on device %d failed", floats * sizeof(float), g->devices[0]);
        }
        g->floats = floats;
    }
    return pt_group_clear(g);
}

int pt_group_render(pt_group *g, uint32_t spp_total, uint32_t first_sample, uint64_t seed, int mode)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    if (!g->floats) return gfail(g, PT_E_INVALID, "pt_group_set_camera has not been called");
    if (spp_total == 0) return gfail(g, PT_E_INVALID, "spp must be positive");
    if (g->poisoned) return gfail(g, PT_E_INVALID, "an earlier pt_group_render failed part-way: call pt_group_clear");
    const int n = (int)g->ctx.size();
    for (int i = 0; i < n; i++) {           // enqueued on every device's stream before anything waits
        uint32_t first = 0, count = 0;
        pt_group_share(spp_total, n, i, first_sample, &first, &count);
        if (count == 0) continue;
        const int rc = pt_render_into(g->ctx[i], g->part[i], count, first, seed, mode);
        if (rc != PT_OK) {
            // members 0..i-1 have already added their shares to their images: the images no longer
            // add up to a whole number of samples per pixel, and stay unusable until they are cleared
            if (i > 0) g->poisoned = true;
            return member_failed(g, (size_t)i, rc);
        }
    }
    g->total_samples += spp_total;
    g->dirty = true;
    return PT_OK;
}

int pt_group_reduce(pt_group *g)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    if (!g->floats) return gfail(g, PT_E_INVALID, "pt_group_set_camera has not been called");
    Rccl *r = rccl();
    if (!r) return gfail(g, PT_E_HIP, "%s", g_rccl.error.c_str());
    if (g->poisoned) return gfail(g, PT_E_INVALID, "an earlier pt_group_render failed part-way: call pt_group_clear");
    const int n = (int)g->ctx.size();
    ncclResult_t nr = r->GroupStart();
    if (nr != ncclSuccess) return gfail(g, PT_E_HIP, "ncclGroupStart: %s", r->GetErrorString(nr));
    for (int i = 0; i < n && nr == ncclSuccess; i++)
        nr = r->Reduce(g->part[i], i == 0 ? g->sum : nullptr, g->floats, ncclFloat, ncclSum, 0, g->comms[i], g->streams[i]);
    const ncclResult_t ne = r->GroupEnd();          // closes the group that GroupStart opened, also after a failed call
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return gfail(g, PT_E_HIP, "ncclReduce: %s", r->GetErrorString(nr));
    int rc = wait_all(g);
    if (rc != PT_OK) return rc;
    // the combined image and the sample count of all devices become the first context's own
    rc = pt_write_accum(g->ctx[0], g->sum, g->total_samples);
    if (rc != PT_OK) return member_failed(g, 0, rc);
    g->dirty = false;
    return PT_OK;
}

int pt_group_read_accum(pt_group *g, float *rgb_sum_host, uint64_t *total_samples)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    if (g->dirty) {
        const int rc = pt_group_reduce(g);
        if (rc != PT_OK) return rc;
    }
    const int rc = pt_read_accum(g->ctx[0], rgb_sum_host, total_samples);
    if (rc != PT_OK) return member_failed(g, 0, rc);
    return PT_OK;
}

int pt_group_tonemap_rgb8(pt_group *g, uint8_t *out_host)
{
    if (!g) return gfail(nullptr, PT_E_INVALID, "null group");
    if (g->dirty) {
        const int rc = pt_group_reduce(g);
        if (rc != PT_OK) return rc;
    }
    const int rc = pt_tonemap_rgb8(g->ctx[0], out_host);
    if (rc != PT_OK) return member_failed(g, 0, rc);
    return PT_OK;
}

}  // extern "C"
