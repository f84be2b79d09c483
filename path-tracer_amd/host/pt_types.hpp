// Host-side scene model with the reference's type names and call surface
// (Material, Surface, Triangle, Sphere, Intersection, Camera, Scene,
// add_surface), so scene-building code written against the reference compiles
// against this header.  The types are data carriers: the arithmetic of the
// render step runs on the GPU.  The per-object methods the reference computes
// on the CPU (Surface::intersects, Material::sample_pdf, Material::brdf) are
// kept and forward single queries to the GPU library.
//
// Image size is a runtime value here (SCREEN_WIDTH / SCREEN_HEIGHT are
// variables, set with set_screen_size), where the reference fixes it at
// compile time (src/constants.hpp:5-7).
#pragma once

#include <functional>
#include <memory>
#include <optional>
#include <utility>
#include <vector>

#include "../../include/pt_amd.h"
#include "ptm.hpp"

using ptm::vec2;
using ptm::vec3;
using ptm::mat3;

// src/constants.hpp:4-7
inline constexpr double EPSILON = 0.00001;
inline int SCREEN_HEIGHT = 200;
inline int SCREEN_WIDTH = 356;
inline void set_screen_size(int width, int height)
{
    SCREEN_WIDTH = width;
    SCREEN_HEIGHT = height;
}

// ---------------------------------------------------------------- materials

// src/material.hpp:9-11
enum MaterialType { DIFFUSE = PT_DIFFUSE, MIRROR = PT_MIRROR, GLASS = PT_GLASS };

namespace pt {
// Single-query gateways to the GPU library (defined in pt_render.hpp).
vec3 gpu_sample_pdf(MaterialType type, vec3 normal, vec3 incident, float xi0, float xi1);
vec3 gpu_brdf(MaterialType type, vec3 color, vec3 normal, vec3 reflected);
bool gpu_intersect(const pt_surface &surface, vec3 start, vec3 dir, vec3 &point, float &distance);
float gpu_uniform();   // next variate of a host-side counter stream, evaluated on the GPU
}  // namespace pt

// src/material.hpp:69-97
class Material {
  public:
    const MaterialType type;
    const vec3 color;
    const vec3 emittance;

    Material(MaterialType type, const vec3 &color = vec3(1, 1, 1), const vec3 &emittance = vec3(0, 0, 0))
        : type(type), color(color), emittance(emittance) {}

    // One sampled outgoing direction for a ray arriving along `incident`.
    vec3 sample_pdf(const vec3 &normal, const vec3 &incident) const
    {
        const float xi0 = type == MIRROR ? 0.0f : pt::gpu_uniform();
        const float xi1 = type == DIFFUSE ? pt::gpu_uniform() : 0.0f;
        return pt::gpu_sample_pdf(type, normal, incident, xi0, xi1);
    }
    // Throughput weight for leaving along `reflected`.
    vec3 brdf(const vec3 &normal, const vec3 &reflected) const
    {
        return pt::gpu_brdf(type, color, normal, reflected);
    }
};

// ---------------------------------------------------------------- surfaces

class Surface;

// src/intersection.hpp:12-16
struct Intersection {
    vec3 point;
    float distance;
    std::reference_wrapper<const Surface> surface;
};

// src/surface.hpp:20-29
class Surface {
  public:
    const Material material;

    explicit Surface(const Material &material) : material(material) {}
    virtual ~Surface() = default;

    virtual vec3 get_normal(const vec3 &intersection_point) const = 0;
    // Geometry in the flat form the GPU library takes (material index left 0).
    virtual pt_surface flat() const = 0;

    std::optional<Intersection> intersects(vec3 ray_start, vec3 ray_dir) const
    {
        vec3 point;
        float distance = 0;
        if (!pt::gpu_intersect(flat(), ray_start, ray_dir, point, distance)) return std::nullopt;
        return Intersection{point, distance, *this};
    }
};

// src/surface.hpp:31-74
class Triangle : public Surface {
  public:
    const vec3 v0, v1, v2;
    const vec3 normal;

    Triangle(vec3 v0, vec3 v1, vec3 v2, const Material &material)
        : Surface(material), v0(v0), v1(v1), v2(v2),
          normal(ptm::normalize(ptm::cross(v2 - v0, v1 - v0))) {}

    vec3 get_normal(const vec3 &) const override { return normal; }

    pt_surface flat() const override
    {
        pt_surface s{};
        s.kind = PT_TRIANGLE;
        s.a[0] = v0.x; s.a[1] = v0.y; s.a[2] = v0.z;
        s.b[0] = v1.x; s.b[1] = v1.y; s.b[2] = v1.z;
        s.c[0] = v2.x; s.c[1] = v2.y; s.c[2] = v2.z;
        return s;
    }
};

// src/surface.hpp:76-113
class Sphere : public Surface {
  public:
    const vec3 position;
    const float radius;
    const float radius2;

    Sphere(vec3 position, float radius, const Material &material)
        : Surface(material), position(position), radius(radius), radius2(radius * radius) {}

    vec3 get_normal(const vec3 &intersection_point) const override
    {
        return ptm::normalize(intersection_point - position);
    }

    pt_surface flat() const override
    {
        pt_surface s{};
        s.kind = PT_SPHERE;
        s.a[0] = position.x; s.a[1] = position.y; s.a[2] = position.z;
        s.b[0] = radius;
        return s;
    }
};

// ---------------------------------------------------------------- camera

// src/camera.hpp:10: a small tilt that keeps rays off exact axis alignment
inline const vec2 OFFSET{2 * EPSILON, EPSILON};

// src/camera.hpp:12-102
class Camera {
    vec3 pos;
    vec2 rot;            // pitch (x), yaw (y)
    mat3 rot_matrix;
    float focal_length;
    bool dirty = true;

    // R = Ry * Rx of the negated angles; trig in double, narrowed to float
    // (src/camera.hpp:20-27)
    void update_rotation_matrix()
    {
        const float cx = static_cast<float>(std::cos(static_cast<double>(-rot.x)));
        const float sx = static_cast<float>(std::sin(static_cast<double>(-rot.x)));
        const float cy = static_cast<float>(std::cos(static_cast<double>(-rot.y)));
        const float sy = static_cast<float>(std::sin(static_cast<double>(-rot.y)));
        const mat3 yaw(cy, 0, sy, 0, 1, 0, -sy, 0, cy);
        const mat3 pitch(1, 0, 0, 0, cx, -sx, 0, sx, cx);
        rot_matrix = yaw * pitch;
    }

  public:
    Camera() : pos(), rot(), focal_length(static_cast<float>(SCREEN_HEIGHT)) { update(); }

    Camera(vec3 pos_, vec2 rot_, float focal_length_)
        : pos(pos_), rot(rot_ + OFFSET), focal_length(focal_length_)
    {
        update();
    }

    // A camera with exactly this stored pose (no OFFSET added): for restoring
    // a pose read back with get_position/get_rotation.  Not in the reference.
    static Camera from_pose(vec3 pos_, vec2 rot_, float focal_length_)
    {
        Camera c;
        c.pos = pos_;
        c.rot = rot_;
        c.focal_length = focal_length_;
        c.dirty = true;
        c.update_rotation_matrix();
        c.dirty = false;
        return c;
    }

    void rotate(float dx, float dy)
    {
        if (dx == 0 && dy == 0) return;
        rot.x += dx;
        rot.y += dy;
        dirty = true;
    }

    // dx strafes, dz walks in the yaw direction, dy is vertical (src/camera.hpp:50-58)
    void move(float dx, float dy, float dz)
    {
        if (dx == 0 && dy == 0 && dz == 0) return;
        const double c = std::cos(static_cast<double>(-rot.y)), s = std::sin(static_cast<double>(-rot.y));
        pos.x = static_cast<float>(pos.x + (dx * c - dz * s));
        pos.z = static_cast<float>(pos.z + (dx * s + dz * c));
        pos.y += dy;
        dirty = true;
    }

    const vec3 &get_position() const { return pos; }
    const vec2 &get_rotation() const { return rot; }
    const mat3 &get_rotation_matrix() const { return rot_matrix; }
    float get_focal_length() const { return focal_length; }

    // Rebuilds the matrix if the pose changed; true means the image must restart.
    // The matrix is built before pitch is clamped and yaw wrapped, as in the
    // reference (src/camera.hpp:76-89).
    bool update()
    {
        if (!dirty) return false;
        dirty = false;
        update_rotation_matrix();
        rot.x = ptm::clamp(rot.x, static_cast<float>(-M_PI_2), static_cast<float>(M_PI_2));
        while (rot.y > 2 * M_PI) rot.y = static_cast<float>(rot.y - 2 * M_PI);
        while (rot.y < 0) rot.y = static_cast<float>(rot.y + 2 * M_PI);
        return true;
    }

    vec3 get_forward() const { return rot_matrix * vec3(0, 0, 1); }
    vec3 get_downward() const { return rot_matrix * vec3(0, 1, 0); }
    vec3 get_right() const { return rot_matrix * vec3(1, 0, 0); }
};

// ---------------------------------------------------------------- scene

// src/scene.hpp:9-12
template <class T>
void add_surface(std::vector<std::unique_ptr<Surface>> &surfaces, const T &surface)
{
    surfaces.push_back(std::make_unique<T>(surface));
}

// src/scene.hpp:15-30.  Note the argument order: aperture before focal length.
struct Scene {
    Camera camera;
    vec3 BG_COLOR;
    std::vector<std::unique_ptr<Surface>> surfaces;

    bool DOF_on;
    float DOF_focal_length;
    float DOF_aperture_size;

    Scene(Camera &camera, vec3 BG_COLOR, std::vector<std::unique_ptr<Surface>> &surfaces, bool DOF_on,
          float aperture_size = 0.15, float focal_length = 2)
        : camera(camera), BG_COLOR(BG_COLOR), surfaces(std::move(surfaces)), DOF_on(DOF_on),
          DOF_focal_length(focal_length), DOF_aperture_size(aperture_size) {}
};
