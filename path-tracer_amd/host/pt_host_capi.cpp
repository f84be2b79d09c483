// C entry points over the C++ host scene model, so that non-C++ hosts (the
// Python tests and the benchmark driver) build scenes through the same code a
// C++ caller uses: Scenes::*, Camera's matrix, pt::flatten.
#include <fstream>

#include "pt_render.hpp"
#include "pt_scene_io.hpp"
#include "pt_scenes.hpp"

namespace {
thread_local std::string g_host_error;
}  // namespace

extern "C" {

// Flattens scene `name` (Scenes::by_name) for a width x height image (focal
// length = height pixels).  surfaces/materials hold `capacity` entries.
// Returns 0, -1 for an unknown name, -2 if capacity is too small; *n_surfaces
// is set to the scene's size in both of the last two cases, so a caller can
// size its arrays with a first call of capacity 0; -3 if building the scene
// failed (out of memory for a huge synthetic scene; pt_host_last_error says).
int pt_host_builtin_scene(const char *name, int width, int height, pt_surface *surfaces,
                          pt_material *materials, int32_t capacity, int32_t *n_surfaces,
                          float bg[3], pt_camera *camera)
{
    set_screen_size(width, height);
    try {
        std::optional<Scene> scene;
        if (!Scenes::by_name(name, scene)) return -1;
        const pt::FlatScene flat = pt::flatten(*scene, width, height);
        *n_surfaces = static_cast<int32_t>(flat.surfaces.size());
        if (*n_surfaces > capacity) return -2;
        for (size_t i = 0; i < flat.surfaces.size(); i++) {
            surfaces[i] = flat.surfaces[i];
            materials[i] = flat.materials[i];
        }
        std::memcpy(bg, flat.bg, sizeof flat.bg);
        *camera = flat.camera;
        return 0;
    } catch (const std::exception &e) {     // no exception may unwind through a C caller
        g_host_error = e.what();
        return -3;
    }
}

namespace {
int emit(const Scene &scene, int width, int height, pt_surface *surfaces, pt_material *materials,
         int32_t capacity, int32_t *n_surfaces, float bg[3], pt_camera *camera)
{
    const pt::FlatScene flat = pt::flatten(scene, width, height);
    *n_surfaces = static_cast<int32_t>(flat.surfaces.size());
    std::memcpy(bg, flat.bg, sizeof flat.bg);
    *camera = flat.camera;
    if (*n_surfaces > capacity) return -2;
    for (size_t i = 0; i < flat.surfaces.size(); i++) {
        surfaces[i] = flat.surfaces[i];
        materials[i] = flat.materials[i];
    }
    return 0;
}
}  // namespace

const char *pt_host_last_error(void) { return g_host_error.c_str(); }

// Loads a scene file (pt_scene_io.hpp) and flattens it like pt_host_builtin_scene.
// Returns 0, -2 if capacity is too small (*n_surfaces set), -3 if the file cannot
// be read or parsed (pt_host_last_error has the message).
int pt_host_load_scene(const char *path, int width, int height, pt_surface *surfaces,
                       pt_material *materials, int32_t capacity, int32_t *n_surfaces,
                       float bg[3], pt_camera *camera)
{
    set_screen_size(width, height);
    try {
        std::ifstream in(path);
        if (!in) throw std::runtime_error(std::string("cannot open ") + path);
        const std::string file(path);
        const size_t cut = file.find_last_of('/');
        const Scene scene = pt::load_scene(in, cut == std::string::npos ? std::string() : file.substr(0, cut));
        return emit(scene, width, height, surfaces, materials, capacity, n_surfaces, bg, camera);
    } catch (const std::exception &e) {
        g_host_error = e.what();
        return -3;
    }
}

// Writes scene `name` (Scenes::by_name, built for a width x height image) as a scene file.
// Returns 0, -1 for an unknown name, -3 if the file cannot be written.
int pt_host_save_scene(const char *name, int width, int height, const char *path)
{
    set_screen_size(width, height);
    try {
        std::optional<Scene> scene;
        if (!Scenes::by_name(name, scene)) return -1;
        std::ofstream out(path);
        if (!out) { g_host_error = std::string("cannot write ") + path; return -3; }
        pt::save_scene(*scene, out);
        if (!out.good()) { g_host_error = std::string("error while writing ") + path; return -3; }
        return 0;
    } catch (const std::exception &e) {
        g_host_error = e.what();
        return -3;
    }
}

// Camera pose -> rotation matrix through class Camera (src/camera.hpp:20-39).
// add_offset != 0 uses the three-argument constructor's OFFSET.
void pt_host_camera_matrix(float rot_x, float rot_y, int add_offset, float out_rot[9])
{
    Camera cam;                       // pose (0,0), no offset
    if (add_offset) {
        cam = Camera(vec3(0, 0, 0), vec2(rot_x, rot_y), 1.0f);
    } else {
        cam.rotate(rot_x, rot_y);
        cam.update();
    }
    const mat3 &r = cam.get_rotation_matrix();
    for (int c = 0; c < 3; c++) {
        out_rot[3 * c + 0] = r[c].x;
        out_rot[3 * c + 1] = r[c].y;
        out_rot[3 * c + 2] = r[c].z;
    }
}

// class Camera driven as the main loop drives it: rotate, move (handle_input) and update for
// n steps from the stored pose.  moves: n x 5 = d_rot_x, d_rot_y, dx, dy, dz.  out: n x 15 =
// position, rotation, update()'s result, rotation matrix (column-major) after each step.
void pt_host_camera_walk(const float pos0[3], const float rot0[2], int n, const float *moves, float *out)
{
    Camera cam = Camera::from_pose(vec3(pos0[0], pos0[1], pos0[2]), vec2(rot0[0], rot0[1]), 1.0f);
    for (int k = 0; k < n; k++) {
        const float *m = moves + 5 * k;
        cam.rotate(m[0], m[1]);
        cam.move(m[2], m[3], m[4]);
        const bool updated = cam.update();
        float *o = out + 15 * k;
        const vec3 &p = cam.get_position();
        const vec2 &r = cam.get_rotation();
        o[0] = p.x; o[1] = p.y; o[2] = p.z;
        o[3] = r.x; o[4] = r.y;
        o[5] = updated ? 1.0f : 0.0f;
        const mat3 &rm = cam.get_rotation_matrix();
        for (int c = 0; c < 3; c++) { o[6 + 3 * c] = rm[c].x; o[7 + 3 * c] = rm[c].y; o[8 + 3 * c] = rm[c].z; }
    }
}

}  // extern "C"
