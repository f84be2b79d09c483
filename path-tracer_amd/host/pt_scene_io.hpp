// Text scene files for the host scene model.  The reference has only
// hard-coded scenes (src/scene.hpp:32-203); this format lets scenes be data.
// One statement per line, '#' starts a comment:
//
//   camera  px py pz  rot_x rot_y  focal     pose as scene code gives it: OFFSET is added,
//                                            focal "auto" = image height in pixels
//   camera_pose  px py pz  rot_x rot_y  focal   exactly this stored pose (what save_scene writes)
//   background  r g b
//   dof  off | aperture focal_length
//   material  NAME  diffuse|mirror|glass  r g b  [er eg eb]
//   sphere  cx cy cz  radius  MATERIAL
//   triangle  x0 y0 z0  x1 y1 z1  x2 y2 z2  MATERIAL
//   mesh  FILE.obj  MATERIAL  [scale [tx ty tz]]   the faces of a Wavefront OBJ file as triangles
//                                            (vertex * scale + t; FILE relative to the scene file)
//
// Of an OBJ file the `v` and `f` statements are read: a face's vertices may be
// written v, v/vt, v//vn or v/vt/vn, counted from 1 or, negative, back from the
// last vertex read so far; a face of more than three vertices becomes a fan
// round its first vertex.  Everything else in the file is skipped.
//
// Surfaces keep file order (the order breaks distance ties).  Numbers are
// written with 9 significant digits, which round-trips every fp32 value.
#pragma once

#include <cstdio>
#include <fstream>
#include <istream>
#include <map>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>

#include "pt_types.hpp"

namespace pt {

class SceneFileError : public std::runtime_error {
  public:
    SceneFileError(int line, const std::string &what)
        : std::runtime_error("scene file line " + std::to_string(line) + ": " + what), line(line) {}
    int line;
};

// Appends the faces of an OBJ stream to `surfaces`, in file order.  Returns the
// number of triangles added.  `what` names the stream in error messages.
inline int load_obj(std::istream &in, std::vector<std::unique_ptr<Surface>> &surfaces, const Material &material,
                    float scale = 1.0f, vec3 offset = vec3(0, 0, 0), const std::string &what = "OBJ")
{
    std::vector<vec3> vertices;
    std::string line;
    int line_no = 0, added = 0;
    auto fail = [&](const std::string &why) {
        throw std::runtime_error(what + " line " + std::to_string(line_no) + ": " + why);
    };
    while (std::getline(in, line)) {
        line_no++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream ss(line);
        std::string word;
        if (!(ss >> word)) continue;
        if (word == "v") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) fail("a vertex needs three coordinates");
            vertices.push_back(vec3(x * scale + offset.x, y * scale + offset.y, z * scale + offset.z));
        } else if (word == "f") {
            std::vector<int> corner;
            std::string ref;
            while (ss >> ref) {
                long index = 0;
                try { index = std::stol(ref.substr(0, ref.find('/'))); }
                catch (const std::exception &) { fail("bad vertex reference '" + ref + "'"); }
                const long n = static_cast<long>(vertices.size());
                const long at = index > 0 ? index - 1 : n + index;
                if (index == 0 || at < 0 || at >= n) fail("vertex reference '" + ref + "' out of range");
                corner.push_back(static_cast<int>(at));
            }
            if (corner.size() < 3) fail("a face needs three vertices");
            for (size_t k = 1; k + 1 < corner.size(); k++) {
                add_surface(surfaces, Triangle(vertices[corner[0]], vertices[corner[k]], vertices[corner[k + 1]], material));
                added++;
            }
        }
    }
    return added;
}

// base_dir: where `mesh` statements look for their files ("" = the working directory).
inline Scene load_scene(std::istream &in, const std::string &base_dir = "")
{
    std::vector<std::unique_ptr<Surface>> surfaces;
    std::map<std::string, Material> materials;
    Camera camera;
    vec3 background(0, 0, 0);
    bool dof = false, have_camera = false;
    float aperture = 0.15f, focus = 2.0f;
    std::string line;
    int line_no = 0;

    auto need = [&](std::istringstream &ss, auto &...v) {
        if (!((ss >> v) && ...)) throw SceneFileError(line_no, "too few or malformed values");
    };
    auto material_named = [&](const std::string &name) -> const Material & {
        auto it = materials.find(name);
        if (it == materials.end()) throw SceneFileError(line_no, "unknown material '" + name + "'");
        return it->second;
    };

    while (std::getline(in, line)) {
        line_no++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream ss(line);
        std::string word;
        if (!(ss >> word)) continue;
        if (word == "camera" || word == "camera_pose") {
            float px, py, pz, rx, ry, focal;
            std::string focal_word;
            need(ss, px, py, pz, rx, ry, focal_word);
            if (focal_word == "auto") {
                focal = static_cast<float>(SCREEN_HEIGHT);
            } else {
                try { focal = std::stof(focal_word); }
                catch (const std::exception &) { throw SceneFileError(line_no, "bad focal length '" + focal_word + "'"); }
            }
            camera = word == "camera" ? Camera(vec3(px, py, pz), vec2(rx, ry), focal)
                                      : Camera::from_pose(vec3(px, py, pz), vec2(rx, ry), focal);
            have_camera = true;
        } else if (word == "background") {
            float r, g, b;
            need(ss, r, g, b);
            background = vec3(r, g, b);
        } else if (word == "dof") {
            std::string first;
            need(ss, first);
            if (first == "off") {
                dof = false;
            } else {
                dof = true;
                try { aperture = std::stof(first); }
                catch (const std::exception &) { throw SceneFileError(line_no, "bad aperture '" + first + "'"); }
                need(ss, focus);
            }
        } else if (word == "material") {
            std::string name, kind;
            float r, g, b, er = 0, eg = 0, eb = 0;
            need(ss, name, kind, r, g, b);
            if (ss >> er) need(ss, eg, eb);
            MaterialType type;
            if (kind == "diffuse") type = DIFFUSE;
            else if (kind == "mirror") type = MIRROR;
            else if (kind == "glass") type = GLASS;
            else throw SceneFileError(line_no, "unknown material type '" + kind + "'");
            materials.erase(name);
            materials.emplace(name, Material(type, vec3(r, g, b), vec3(er, eg, eb)));
        } else if (word == "sphere") {
            float x, y, z, radius;
            std::string name;
            need(ss, x, y, z, radius, name);
            add_surface(surfaces, Sphere(vec3(x, y, z), radius, material_named(name)));
        } else if (word == "triangle") {
            float v[9];
            std::string name;
            need(ss, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], name);
            add_surface(surfaces, Triangle(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), vec3(v[6], v[7], v[8]),
                                           material_named(name)));
        } else if (word == "mesh") {
            std::string file, name;
            float scale = 1.0f, tx = 0, ty = 0, tz = 0;
            need(ss, file, name);
            if (ss >> scale) { if (ss >> tx) need(ss, ty, tz); }
            const Material &m = material_named(name);
            const std::string path = (base_dir.empty() || file[0] == '/') ? file : base_dir + "/" + file;
            std::ifstream obj(path);
            if (!obj) throw SceneFileError(line_no, "cannot open mesh '" + path + "'");
            try { load_obj(obj, surfaces, m, scale, vec3(tx, ty, tz), path); }
            catch (const std::runtime_error &e) { throw SceneFileError(line_no, e.what()); }
        } else {
            throw SceneFileError(line_no, "unknown statement '" + word + "'");
        }
    }
    if (!have_camera) throw SceneFileError(line_no, "no camera statement");
    return Scene(camera, background, surfaces, dof, aperture, focus);
}

inline void save_scene(const Scene &scene, std::ostream &out)
{
    char buf[512];
    const vec3 &p = scene.camera.get_position();
    const vec2 &r = scene.camera.get_rotation();
    std::snprintf(buf, sizeof buf, "camera_pose %.9g %.9g %.9g %.9g %.9g %.9g\n", p.x, p.y, p.z, r.x, r.y,
                  scene.camera.get_focal_length());
    out << buf;
    std::snprintf(buf, sizeof buf, "background %.9g %.9g %.9g\n", scene.BG_COLOR.x, scene.BG_COLOR.y, scene.BG_COLOR.z);
    out << buf;
    if (scene.DOF_on) {
        std::snprintf(buf, sizeof buf, "dof %.9g %.9g\n", scene.DOF_aperture_size, scene.DOF_focal_length);
        out << buf;
    } else {
        // keep the parameters a later "dof on" toggle would use
        std::snprintf(buf, sizeof buf, "dof %.9g %.9g\ndof off\n", scene.DOF_aperture_size, scene.DOF_focal_length);
        out << buf;
    }
    static const char *kinds[] = {"diffuse", "mirror", "glass"};
    int i = 0;
    for (const auto &ptr : scene.surfaces) {
        const Material &m = ptr->material;
        std::snprintf(buf, sizeof buf, "material m%d %s %.9g %.9g %.9g %.9g %.9g %.9g\n", i, kinds[m.type], m.color.x,
                      m.color.y, m.color.z, m.emittance.x, m.emittance.y, m.emittance.z);
        out << buf;
        const pt_surface s = ptr->flat();
        if (s.kind == PT_SPHERE)
            std::snprintf(buf, sizeof buf, "sphere %.9g %.9g %.9g %.9g m%d\n", s.a[0], s.a[1], s.a[2], s.b[0], i);
        else
            std::snprintf(buf, sizeof buf, "triangle %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g m%d\n", s.a[0], s.a[1],
                          s.a[2], s.b[0], s.b[1], s.b[2], s.c[0], s.c[1], s.c[2], i);
        out << buf;
        i++;
    }
}

}  // namespace pt
