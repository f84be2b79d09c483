// Exercises the host scene model (scene builders, flattening, scene files,
// camera matrices) through its C entry points.  Built with AddressSanitizer and
// UndefinedBehaviorSanitizer by tests/test_host_sanitize.py; needs no GPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "pt_amd.h"
#include "pt_scenes.hpp"
#include "pt_viewer.hpp"

extern "C" {
int pt_host_builtin_scene(const char *, int, int, pt_surface *, pt_material *, int32_t, int32_t *, float[3], pt_camera *);
int pt_host_load_scene(const char *, int, int, pt_surface *, pt_material *, int32_t, int32_t *, float[3], pt_camera *);
int pt_host_save_scene(const char *, int, int, const char *);
const char *pt_host_last_error(void);
void pt_host_camera_matrix(float, float, int, float[9]);
}

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

int main(int argc, char **argv)
{
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    const char *names[] = {"cornell", "cornell_2", "orange_plane", "red_corner", "glass_stress", "heightfield:9"};
    for (const char *name : names) {
        int32_t n = -1;
        float bg[3];
        pt_camera cam;
        int rc = pt_host_builtin_scene(name, 64, 48, nullptr, nullptr, 0, &n, bg, &cam);
        CHECK(rc == -2 && n > 0);
        std::vector<pt_surface> s(n);
        std::vector<pt_material> m(n);
        rc = pt_host_builtin_scene(name, 64, 48, s.data(), m.data(), n, &n, bg, &cam);
        CHECK(rc == 0 && cam.width == 64 && cam.height == 48);

        const std::string path = tmp + "/sanitize_scene.txt";
        CHECK(pt_host_save_scene(name, 64, 48, path.c_str()) == 0);
        int32_t n2 = -1;
        float bg2[3];
        pt_camera cam2;
        std::vector<pt_surface> s2(n);
        std::vector<pt_material> m2(n);
        rc = pt_host_load_scene(path.c_str(), 64, 48, s2.data(), m2.data(), n, &n2, bg2, &cam2);
        CHECK(rc == 0 && n2 == n);
        if (rc == 0 && n2 == n) {
            CHECK(std::memcmp(s.data(), s2.data(), sizeof(pt_surface) * n) == 0);
            CHECK(std::memcmp(m.data(), m2.data(), sizeof(pt_material) * n) == 0);
            CHECK(std::memcmp(&cam, &cam2, sizeof cam) == 0);
        }
        // a capacity one short must report the size and write nothing past it
        if (n > 1) {
            std::vector<pt_surface> s3(n - 1);
            std::vector<pt_material> m3(n - 1);
            CHECK(pt_host_load_scene(path.c_str(), 64, 48, s3.data(), m3.data(), n - 1, &n2, bg2, &cam2) == -2 && n2 == n);
        }
    }
    {
        int32_t n = 0;
        float bg[3];
        pt_camera cam;
        CHECK(pt_host_builtin_scene("no_such_scene", 8, 8, nullptr, nullptr, 0, &n, bg, &cam) == -1);
        CHECK(pt_host_builtin_scene("heightfield:0", 8, 8, nullptr, nullptr, 0, &n, bg, &cam) != 0);
        CHECK(pt_host_builtin_scene("heightfield:-3", 8, 8, nullptr, nullptr, 0, &n, bg, &cam) != 0);
        CHECK(pt_host_builtin_scene("heightfield:x", 8, 8, nullptr, nullptr, 0, &n, bg, &cam) != 0);
        CHECK(pt_host_save_scene("cornell", 8, 8, (tmp + "/no/such/dir/x.txt").c_str()) == -3);
        CHECK(pt_host_load_scene((tmp + "/does_not_exist.txt").c_str(), 8, 8, nullptr, nullptr, 0, &n, bg, &cam) == -3);
        CHECK(std::strlen(pt_host_last_error()) > 0);
    }
    // malformed scene files: every one must be refused with a message, none may crash
    const char *bad[] = {
        "",
        "background 0 0 0\n",
        "camera 0 0 0 0 0\n",
        "camera 0 0 0 0 0 fast\n",
        "camera 0 0 0 0 0 auto\nsphere 0 0 0 1 nobody\n",
        "camera 0 0 0 0 0 auto\nmaterial a shiny 1 1 1\n",
        "camera 0 0 0 0 0 auto\nmaterial a diffuse 1 1\n",
        "camera 0 0 0 0 0 auto\nmaterial a diffuse 1 1 1 2\n",
        "camera 0 0 0 0 0 auto\nmaterial a diffuse 1 1 1\ntriangle 0 0 0 1 1 1 a\n",
        "camera 0 0 0 0 0 auto\ndof\n",
        "camera 0 0 0 0 0 auto\ndof wide 2\n",
        "camera 0 0 0 0 0 auto\ndof 0.1\n",
        "camera 0 0 0 0 0 auto\nteapot 1 2 3\n",
        "camera 0 0 0 0 0 auto\nmaterial a diffuse 1 1 1\nsphere 0 0 0 1e999 a\n",
    };
    for (const char *text : bad) {
        const std::string path = tmp + "/sanitize_bad.txt";
        { std::ofstream out(path); out << text; }
        int32_t n = 0;
        float bg[3];
        pt_camera cam;
        pt_surface s[4];
        pt_material m[4];
        const int rc = pt_host_load_scene(path.c_str(), 8, 8, s, m, 4, &n, bg, &cam);
        if (rc != -3) std::printf("accepted: %s", text);
        CHECK(rc == -3 && std::strlen(pt_host_last_error()) > 0);
    }
    // a well-formed file with comments, redefinition of a material, non-finite but parseable values
    {
        const std::string path = tmp + "/sanitize_good.txt";
        { std::ofstream out(path);
          out << "# comment\ncamera 0 0 -3 0 0 auto   # trailing\nbackground 0.1 0.2 0.3\ndof 0.15 5\n"
                 "material a diffuse 1 0 0\nmaterial a mirror 1 1 1\nsphere 0 0 0 1 a\n"
                 "material lamp diffuse 0 0 0 5 5 5\ntriangle 0 0 0 1 0 0 0 1 0 lamp\n\n"; }
        int32_t n = 0;
        float bg[3];
        pt_camera cam;
        pt_surface s[4];
        pt_material m[4];
        CHECK(pt_host_load_scene(path.c_str(), 32, 16, s, m, 4, &n, bg, &cam) == 0);
        CHECK(n == 2 && m[0].type == 1 && m[1].emittance[0] == 5.0f && cam.dof_on == 1 && cam.focal == 16.0f);
        CHECK(bg[1] == 0.2f);
    }
    // OBJ meshes through the `mesh` statement: good and malformed files
    {
        const std::string obj = tmp + "/sanitize_mesh.obj", path = tmp + "/sanitize_mesh.txt";
        { std::ofstream out(path); out << "camera 0 0 -3 0 0 auto\nmaterial a diffuse 1 1 1\nmesh sanitize_mesh.obj a 0.5 1 2 3\n"; }
        const char *meshes[] = {
            "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\nf -1/2/3 -2//1 -3\n",   // 3 triangles
            "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nv 1 1 1\nv 2 2 2\nf 1 2\n", "v 1 2\n", "v 0 0 0\nv 1 1 1\nv 2 2 2\nf 0 1 2\n",
            "v 0 0 0\nv 1 1 1\nv 2 2 2\nf 1 2 99999999999999999999\n", "v 0 0 0\nv 1 1 1\nv 2 2 2\nf -4 1 2\n", "f\n",
        };
        for (size_t k = 0; k < sizeof meshes / sizeof *meshes; k++) {
            { std::ofstream out(obj); out << meshes[k]; }
            int32_t n = 0;
            float bg[3];
            pt_camera cam;
            pt_surface s[4];
            pt_material m[4];
            const int rc = pt_host_load_scene(path.c_str(), 8, 8, s, m, 4, &n, bg, &cam);
            if (k == 0) CHECK(rc == 0 && n == 3 && s[0].a[0] == 1.0f && s[0].b[0] == 1.5f && s[2].kind == PT_TRIANGLE);
            else CHECK(rc == -3 && std::strlen(pt_host_last_error()) > 0);
        }
    }
    for (int off = 0; off < 2; off++) {
        float rot[9];
        pt_host_camera_matrix(0.3f, -1.2f, off, rot);
        float det = rot[0] * (rot[4] * rot[8] - rot[7] * rot[5]) - rot[3] * (rot[1] * rot[8] - rot[7] * rot[2]) +
                    rot[6] * (rot[1] * rot[5] - rot[4] * rot[2]);
        CHECK(std::fabs(det - 1.0f) < 1e-5f);
    }
    // the interactive loop's input handling (no window, no GPU)
    {
        pt::KeyState keys;
        CHECK(pt::KeyState::parse("w+left+u", keys) && keys[pt::KEY_W] && keys[pt::KEY_LEFT] && keys[pt::KEY_U] && !keys[pt::KEY_S]);
        CHECK(pt::KeyState::parse("-", keys) && keys.down.none());
        CHECK(!pt::KeyState::parse("w+x", keys));

        set_screen_size(64, 48);
        Scene scene = Scenes::cornell_2();
        pt::Logging logging;
        std::ostringstream msg;
        CHECK(!scene.camera.update());                         // a fresh scene's camera is clean
        const vec3 p0 = scene.camera.get_position();
        const vec2 r0 = scene.camera.get_rotation();

        pt::KeyState::parse("-", keys);
        CHECK(!pt::handle_input(scene, 16.0f, keys, logging, msg) && !scene.camera.update());

        pt::KeyState::parse("q", keys);                        // straight up by dt/1000
        CHECK(!pt::handle_input(scene, 500.0f, keys, logging, msg));
        CHECK(scene.camera.update() && !scene.camera.update());
        CHECK(scene.camera.get_position().y == p0.y + 0.5f && scene.camera.get_position().x == p0.x);

        pt::KeyState::parse("w", keys);                        // forward along the yaw direction
        pt::handle_input(scene, 1000.0f, keys, logging, msg);
        CHECK(scene.camera.update());
        const vec3 p1 = scene.camera.get_position();
        const double c = std::cos(-double(r0.y)), sn = std::sin(-double(r0.y));
        CHECK(std::fabs(p1.x - float(p0.x - 1.0 * sn)) < 1e-6f && std::fabs(p1.z - float(p0.z + 1.0 * c)) < 1e-6f);

        pt::KeyState::parse("up+right", keys);
        pt::handle_input(scene, 250.0f, keys, logging, msg);
        CHECK(scene.camera.update());
        CHECK(std::fabs(scene.camera.get_rotation().x - (r0.x + 0.25f)) < 1e-6f);
        CHECK(std::fabs(scene.camera.get_rotation().y - (r0.y + 0.25f)) < 1e-6f);
        pt::KeyState::parse("up", keys);                       // pitch is clamped at pi/2
        for (int i = 0; i < 4; i++) { pt::handle_input(scene, 1000.0f, keys, logging, msg); scene.camera.update(); }
        CHECK(scene.camera.get_rotation().x == float(M_PI_2));
        pt::KeyState::parse("left", keys);                     // yaw wraps into [0, 2 pi]
        for (int i = 0; i < 3; i++) { pt::handle_input(scene, 1000.0f, keys, logging, msg); scene.camera.update(); }
        CHECK(scene.camera.get_rotation().y >= 0.0f && scene.camera.get_rotation().y <= float(2 * M_PI));

        const bool dof0 = scene.DOF_on;
        pt::KeyState::parse(dof0 ? "u" : "j", keys);           // a DOF key that changes nothing still restarts the image
        CHECK(pt::handle_input(scene, 16.0f, keys, logging, msg) && scene.DOF_on == dof0);
        pt::KeyState::parse("u+i+o", keys);
        const float f0 = scene.DOF_focal_length, a0 = scene.DOF_aperture_size;
        CHECK(pt::handle_input(scene, 16.0f, keys, logging, msg) && scene.DOF_on);
        CHECK(scene.DOF_focal_length == f0 + 0.1f && scene.DOF_aperture_size == a0 + 0.01f);
        pt::KeyState::parse("j+k+l", keys);
        CHECK(pt::handle_input(scene, 16.0f, keys, logging, msg) && !scene.DOF_on);
        CHECK(scene.DOF_focal_length == f0 + 0.1f - 0.1f && scene.DOF_aperture_size == a0 + 0.01f - 0.01f);
        pt::KeyState::parse("1", keys);
        CHECK(!pt::handle_input(scene, 16.0f, keys, logging, msg) && logging.save_image);
        pt::KeyState::parse("2", keys);
        pt::handle_input(scene, 16.0f, keys, logging, msg);
        CHECK(!logging.save_image && logging.next_target == 100 && !logging.exit);
        CHECK(msg.str().find("depth of field enabled") != std::string::npos);
    }
    // the BMP writer: 3 x 2 pixels, rows padded to 12 bytes, bottom row first, B G R
    {
        const std::vector<uint8_t> rgb = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18};
        const std::string path = tmp + "/sanitize.bmp";
        CHECK(pt::write_bmp(path, 3, 2, rgb));
        std::ifstream in(path, std::ios::binary);
        std::vector<unsigned char> b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        CHECK(b.size() == 54 + 24 && b[0] == 'B' && b[1] == 'M' && b[2] == 78 && b[10] == 54 && b[14] == 40);
        CHECK(b[18] == 3 && b[22] == 2 && b[26] == 1 && b[28] == 24);
        CHECK(b[54] == 12 && b[55] == 11 && b[56] == 10 && b[54 + 9] == 0 && b[54 + 12] == 3 && b[54 + 13] == 2);
        CHECK(!pt::write_bmp(path, 3, 2, std::vector<uint8_t>(5)));
        CHECK(!pt::write_bmp(tmp + "/no/such/dir/x.bmp", 3, 2, rgb));
    }
    std::printf(failures ? "%d check(s) failed\n" : "host checks passed\n", failures);
    return failures ? 1 : 0;
}
