// The reference's seven built-in scenes (src/scene.hpp:32-203) as data for the
// host scene model: same names, same surface order (the surface index is the
// tie-break of the closest-hit search), same constants.  Every camera uses a
// focal length of SCREEN_HEIGHT pixels and the constructor that adds OFFSET.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "pt_types.hpp"

namespace Scenes {

namespace detail {

using SurfaceList = std::vector<std::unique_ptr<Surface>>;

inline Material diffuse(float r, float g, float b, float emit = 0.0f)
{
    return Material(DIFFUSE, vec3(r, g, b), emit * vec3(1, 1, 1));
}

// The five wall triangles shared by both Cornell boxes: floor, left, right,
// ceiling, back.  The box is open towards the camera (walls start at z = -1).
inline void cornell_walls(SurfaceList &out, const Material &floor, const Material &left,
                          const Material &right, const Material &ceiling, const Material &back)
{
    add_surface(out, Triangle(vec3(-2, 1, -1), vec3(0, 1, 10), vec3(2, 1, -1), floor));
    add_surface(out, Triangle(vec3(-1, -2, -1), vec3(-1, 0, 10), vec3(-1, 2, -1), left));
    add_surface(out, Triangle(vec3(1, -2, -1), vec3(1, 2, -1), vec3(1, 0, 10), right));
    add_surface(out, Triangle(vec3(-2, -1, -1), vec3(2, -1, -1), vec3(0, -1, 10), ceiling));
    add_surface(out, Triangle(vec3(-2, 1, 1), vec3(0, -10, 1), vec3(2, 1, 1), back));
}

inline Scene finish(SurfaceList &surfaces, vec3 cam_pos, vec2 cam_rot, vec3 background, bool dof,
                    float aperture = 0.15, float focus = 2)
{
    Camera camera(cam_pos, cam_rot, static_cast<float>(SCREEN_HEIGHT));
    return Scene(camera, background, surfaces, dof, aperture, focus);
}

}  // namespace detail

// Three large spheres on a huge floor sphere, coloured lights far away; DOF on.
inline Scene dof_demo()
{
    using namespace detail;
    SurfaceList s;
    const float R = 1000;
    add_surface(s, Sphere(vec3(0, R, 0), R, Material(DIFFUSE, vec3(1, 1, 1), 0.5f * vec3(1, 1, 1))));
    add_surface(s, Sphere(vec3(-1, -1, 1), 1, Material(DIFFUSE, vec3(0.5, 0, 1))));
    add_surface(s, Sphere(vec3(1.5, -1, 2), 1, Material(DIFFUSE, vec3(1, 0, 0.5))));
    add_surface(s, Sphere(vec3(-35, -20, 50), 1, Material(DIFFUSE, vec3(1, 0, 0), 1.0f * vec3(1, 0, 0))));
    add_surface(s, Sphere(vec3(3, -12, 50), 1, Material(DIFFUSE, vec3(0, 0, 1), 1.0f * vec3(0, 0, 1))));
    add_surface(s, Sphere(vec3(35, -16, 50), 1, Material(DIFFUSE, vec3(1, 0, 1), 1.0f * vec3(1, 0, 1))));
    return finish(s, vec3(0, -1, -3), vec2(0, 0), 0.1f * vec3(1, 1, 1), true, 0.3, 5);
}

// Eleven spheres in a row shading from red to yellow under a white sky.
inline Scene test()
{
    using namespace detail;
    SurfaceList s;
    for (int i = 0; i <= 10; i++) {
        const float green = i / 10.0;
        add_surface(s, Sphere(vec3(i, -0.5, 0), 0.5, Material(DIFFUSE, vec3(1, green, 0))));
    }
    return finish(s, vec3(5, 0, -7), vec2(0, 0), vec3(1, 1, 1), false);
}

// Mirror, glass and white spheres on an orange ground triangle under a big lamp; DOF on.
inline Scene orange_plane()
{
    using namespace detail;
    SurfaceList s;
    add_surface(s, Sphere(vec3(-1.5f, 0.0f, 0.0f), 0.75, Material(MIRROR)));
    add_surface(s, Sphere(vec3(1.5f, 0.0f, 0.0f), 0.75, Material(GLASS)));
    add_surface(s, Sphere(vec3(0.0f, 0.0f, 2.0f), 0.75, diffuse(1, 1, 1)));
    add_surface(s, Sphere(vec3(0.0f, -10.0f, 0.0f), 5, diffuse(1, 1, 1, 5.0f)));
    add_surface(s, Triangle(vec3(-100, 0.75, 100), vec3(100, 0.75, 100), vec3(0, 0.75, -100),
                            diffuse(1.00f, 0.40f, 0.00f)));
    return finish(s, vec3(0, 0, -3), vec2(0, 0), vec3(0, 0, 0), true, 0.15, 5);
}

// Black, mirror and glass spheres in the corner of a red floor and a lamp wall.
inline Scene red_corner()
{
    using namespace detail;
    SurfaceList s;
    add_surface(s, Sphere(vec3(0, -1, 2), 1, diffuse(0.05f, 0.05f, 0.05f)));
    add_surface(s, Sphere(vec3(-1.25, -1, 0), 1, Material(MIRROR)));
    add_surface(s, Sphere(vec3(1.25, -1, 0), 1, Material(GLASS)));
    add_surface(s, Triangle(vec3(0, 0, -100), vec3(-10, 0, 3), vec3(10, 0, 3), diffuse(1.00f, 0.05f, 0.05f)));
    add_surface(s, Triangle(vec3(0, -100, 3), vec3(-10, 0, 3), vec3(10, 0, 3), diffuse(1, 1, 1, 1.5f)));
    return finish(s, vec3(0, -4.5, -3), vec2(-M_PI_4, 0), vec3(0, 0, 0), false);
}

// Two spheres on a dark ground under a pale sky, camera turned round; DOF on.
inline Scene sunrise()
{
    using namespace detail;
    SurfaceList s;
    add_surface(s, Triangle(vec3(-100, 0, 100), vec3(100, 0, 100), vec3(0, 0, -100), diffuse(0.05f, 0.05f, 0.05f)));
    add_surface(s, Sphere(vec3(-2, -1, 0), 1, diffuse(0.95f, 0.95f, 0.95f)));
    add_surface(s, Sphere(vec3(-1, -1, 7), 1, diffuse(1.00f, 0.40f, 0.00f)));
    return finish(s, vec3(0, -1, -2), vec2(0, 5.78), vec3(0.7, 0.7, 1), true, 0.2, 9);
}

// Cornell box with a mirror sphere, a glass sphere and a sphere light that
// pokes through the ceiling: the benchmark scene.
inline Scene cornell_2()
{
    using namespace detail;
    SurfaceList s;
    cornell_walls(s, diffuse(1.00f, 0.05f, 0.05f), diffuse(1, 0.05f, 1), diffuse(1, 1, 0.05f),
                  diffuse(0.05f, 1, 1), diffuse(1, 1, 1));
    add_surface(s, Sphere(vec3(-0.32, 0.6, 0.26), 0.4, Material(MIRROR)));
    add_surface(s, Sphere(vec3(0.35, 0.6, -0.39), 0.4, Material(GLASS)));
    add_surface(s, Sphere(vec3(0, -1.8, 0), 1, diffuse(1, 1, 1, 2.0f)));
    return finish(s, vec3(0, 0, -3), vec2(0, 0), 0.0f * vec3(1, 1, 1), false);
}

// All-diffuse Cornell box lit by its ceiling, with a red sphere and a blue
// box of ten triangles (no bottom face).
inline Scene cornell()
{
    using namespace detail;
    SurfaceList s;
    cornell_walls(s, diffuse(0.05f, 1, 0.05f), diffuse(1, 0.05f, 1), diffuse(1, 1, 0.05f),
                  diffuse(0.05f, 1, 1, 0.7f), diffuse(1, 1, 1));
    add_surface(s, Sphere(vec3(0.35, 0.7, -0.39), 0.3, diffuse(1.00f, 0.05f, 0.05f)));

    // box corners: index bit 2 = top (y = -0.189..) else bottom (y = 1); bits 1,0 pick the footprint corner
    const float foot[4][2] = {{-0.5243243243243243f, -0.1099099099099099f},
                              {0.04504504504504503f, 0.06666666666666665f},
                              {-0.7009009009009008f, 0.463063063063063f},
                              {-0.13153153153153152f, 0.6432432432432433f}};
    const float y_bottom = 1.0f, y_top = -0.18918918918918926f;
    vec3 corner[8];
    for (int i = 0; i < 8; i++) corner[i] = vec3(foot[i & 3][0], i & 4 ? y_top : y_bottom, foot[i & 3][1]);
    // faces as corner indices, in the reference's order (A..H = 0..7)
    const int faces[10][3] = {{4, 1, 0}, {4, 5, 1}, {5, 3, 1}, {5, 7, 3}, {7, 2, 3},
                              {7, 6, 2}, {6, 4, 2}, {4, 0, 2}, {6, 5, 4}, {6, 7, 5}};
    const Material blue = diffuse(0.05f, 0.05f, 1.00f);
    for (const auto &f : faces) add_surface(s, Triangle(corner[f[0]], corner[f[1]], corner[f[2]], blue));
    return finish(s, vec3(0, 0, -3), vec2(0, 0), 0.1f * vec3(1, 1, 1), false);
}

// ---- synthetic stress scenes (not in the reference; built from its types) ----

namespace detail {

inline uint32_t hash32(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
// uniform in [0,1) from an integer key
inline float hash01(uint32_t key) { return static_cast<float>(hash32(key) >> 8) * (1.0f / 16777216.0f); }

// smooth value noise on the integer lattice, in [0,1)
inline float value_noise(float x, float z, uint32_t seed)
{
    const float fx = std::floor(x), fz = std::floor(z);
    const int ix = static_cast<int>(fx), iz = static_cast<int>(fz);
    const float tx = x - fx, tz = z - fz;
    const float sx = tx * tx * (3 - 2 * tx), sz = tz * tz * (3 - 2 * tz);
    auto corner = [&](int cx, int cz) {
        return hash01(seed ^ (static_cast<uint32_t>(cx) * 0x9E3779B1u) ^ (static_cast<uint32_t>(cz) * 0x85EBCA77u));
    };
    const float a = corner(ix, iz), b = corner(ix + 1, iz), c = corner(ix, iz + 1), d = corner(ix + 1, iz + 1);
    return (a + (b - a) * sx) + ((c + (d - c) * sx) - (a + (b - a) * sx)) * sz;
}

}  // namespace detail

// Glass-heavy stress scene: the cornell_2 walls and sphere light, a 5 x 5 x 3
// lattice of glass spheres (radius 0.12, positions jittered by a hash of seed 4)
// and four mirror spheres.  Mirror and glass leave the throughput at 1, so
// russian roulette keeps paths alive at its 0.95 cap: long paths, divergent
// materials.
inline Scene glass_stress()
{
    using namespace detail;
    SurfaceList s;
    cornell_walls(s, diffuse(1.00f, 0.05f, 0.05f), diffuse(1, 0.05f, 1), diffuse(1, 1, 0.05f),
                  diffuse(0.05f, 1, 1), diffuse(1, 1, 1));
    add_surface(s, Sphere(vec3(0, -1.8, 0), 1, diffuse(1, 1, 1, 2.0f)));
    const uint32_t seed = 4;
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 5; j++)
            for (int i = 0; i < 5; i++) {
                const uint32_t id = static_cast<uint32_t>((k * 5 + j) * 5 + i);
                const float jx = (hash01(seed * 1000003u + 3 * id) - 0.5f) * 0.08f;
                const float jy = (hash01(seed * 1000003u + 3 * id + 1) - 0.5f) * 0.08f;
                const float jz = (hash01(seed * 1000003u + 3 * id + 2) - 0.5f) * 0.08f;
                const vec3 c(-0.72f + 0.36f * i + jx, -0.45f + 0.33f * j + jy, -0.55f + 0.5f * k + jz);
                add_surface(s, Sphere(c, 0.12f, Material(GLASS)));
            }
    const float m[4][3] = {{-0.75f, 0.8f, 0.7f}, {0.75f, 0.8f, 0.7f}, {-0.75f, -0.6f, 0.75f}, {0.75f, -0.6f, 0.75f}};
    for (const auto &c : m) add_surface(s, Sphere(vec3(c[0], c[1], c[2]), 0.2f, Material(MIRROR)));
    return finish(s, vec3(0, 0, -3), vec2(0, 0), 0.0f * vec3(1, 1, 1), false);
}

// Large-mesh scene: cornell_2 with its floor triangle replaced by a height
// field of quads x quads cells (two triangles each; 707 gives 999,698
// triangles), heights up to 0.15 above y = 1 (y is down) from value noise,
// per-triangle diffuse colours from a hash of seed 5.  Cells are 0.03 wide
// whatever their number, so the field spreads beyond the box: the reference's
// triangle test rejects |determinant| < 1e-5 (src/surface.hpp:53), which is
// twice the projected area of the triangle, so much smaller cells would never
// be hit.  Winding is chosen so that normalize(cross(v2-v0, v1-v0)) points up.
inline Scene heightfield_mesh(int quads = 707)
{
    using namespace detail;
    SurfaceList s;
    s.reserve(static_cast<size_t>(quads) * quads * 2 + 8);
    const uint32_t seed = 5;
    const float cell = 0.03f, x0 = -0.5f * cell * quads, z0 = -1.0f;
    auto vertex = [&](int i, int k) {
        const float x = x0 + cell * i, z = z0 + cell * k;
        return vec3(x, 1.0f - 0.15f * value_noise(4.0f * x, 4.0f * z, seed), z);
    };
    for (int k = 0; k < quads; k++)
        for (int i = 0; i < quads; i++) {
            const vec3 p00 = vertex(i, k), p10 = vertex(i + 1, k), p01 = vertex(i, k + 1), p11 = vertex(i + 1, k + 1);
            const uint32_t id = static_cast<uint32_t>(k * quads + i);
            const Material m0 = diffuse(0.3f + 0.7f * hash01(seed + 6 * id), 0.3f + 0.7f * hash01(seed + 6 * id + 1),
                                        0.3f + 0.7f * hash01(seed + 6 * id + 2));
            const Material m1 = diffuse(0.3f + 0.7f * hash01(seed + 6 * id + 3), 0.3f + 0.7f * hash01(seed + 6 * id + 4),
                                        0.3f + 0.7f * hash01(seed + 6 * id + 5));
            // v1 along +z, v2 along +x: cross(v2-v0, v1-v0) = (+x) x (+z) = -y, up
            add_surface(s, Triangle(p00, p01, p10, m0));
            add_surface(s, Triangle(p10, p01, p11, m1));
        }
    add_surface(s, Triangle(vec3(-1, -2, -1), vec3(-1, 0, 10), vec3(-1, 2, -1), diffuse(1, 0.05f, 1)));
    add_surface(s, Triangle(vec3(1, -2, -1), vec3(1, 2, -1), vec3(1, 0, 10), diffuse(1, 1, 0.05f)));
    add_surface(s, Triangle(vec3(-2, -1, -1), vec3(2, -1, -1), vec3(0, -1, 10), diffuse(0.05f, 1, 1)));
    add_surface(s, Triangle(vec3(-2, 1, 1), vec3(0, -10, 1), vec3(2, 1, 1), diffuse(1, 1, 1)));
    add_surface(s, Sphere(vec3(-0.32, 0.6, 0.26), 0.4, Material(MIRROR)));
    add_surface(s, Sphere(vec3(0.35, 0.6, -0.39), 0.4, Material(GLASS)));
    add_surface(s, Sphere(vec3(0, -1.8, 0), 1, diffuse(1, 1, 1, 2.0f)));
    return finish(s, vec3(0, 0, -3), vec2(0, 0), 0.0f * vec3(1, 1, 1), false);
}

// Looks a scene up by name: the reference's seven, "glass_stress", and
// "heightfield" or "heightfield:<quads per side>".  Returns false for an
// unknown name.
inline bool by_name(const char *name, std::optional<Scene> &out)
{
    struct Entry { const char *name; Scene (*make)(); };
    static const Entry table[] = {{"dof_demo", dof_demo}, {"test", test}, {"orange_plane", orange_plane},
                                  {"red_corner", red_corner}, {"sunrise", sunrise},
                                  {"cornell_2", cornell_2}, {"cornell", cornell}};
    for (const Entry &e : table)
        if (!std::strcmp(e.name, name)) { out.emplace(e.make()); return true; }
    if (!std::strcmp(name, "glass_stress")) { out.emplace(glass_stress()); return true; }
    if (!std::strcmp(name, "heightfield")) { out.emplace(heightfield_mesh()); return true; }
    if (!std::strncmp(name, "heightfield:", 12)) {
        const int quads = std::atoi(name + 12);
        if (quads < 1 || quads > 4000) return false;
        out.emplace(heightfield_mesh(quads));
        return true;
    }
    return false;
}

}  // namespace Scenes
