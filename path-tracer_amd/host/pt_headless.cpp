// The reference's main loop (src/main.cpp:296-326) without a window: build a
// scene, call trace_rays once per pass, report progress every 10 passes as
// log() does (src/main.cpp:275-277), stop after a fixed number of passes (the
// reference's -exit stops at 100) and write the tone-mapped image as a PPM.
//
//   pt_headless [-scene NAME | -file SCENE.txt] [-width W] [-height H] [-passes N]
//               [-batch B] [-gpus N] [-fast] [-out FILE.ppm]
//               [-log] [-exit] [-keys SCRIPT] [-images DIR]
//
// -batch B > 0 renders B samples per pixel in one GPU call per pass through
// pt::Renderer, keeping the sums on the device, instead of one sample per
// trace_rays call.  -gpus N (with -batch) shares every pass's B samples among
// devices 0..N-1 and combines their images with one RCCL reduce at the end.
//
// -log, -exit, -keys or -images select the interactive loop (pt_viewer.hpp)
// with its window replaced by a replay script: -log saves DIR/<passes>.bmp at
// 100, 200, 400, ... passes and -exit stops at the first of them, as in the
// reference (src/main.cpp:275-308); every line of SCRIPT is
// "<frames> <milliseconds per frame> <keys>", keys like "w+left" or "-" for
// none, named as in src/main.cpp:178-247.  The loop ends with the script (or
// after -passes frames without one).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "pt_render.hpp"
#include "pt_scene_io.hpp"
#include "pt_scenes.hpp"
#include "pt_viewer.hpp"

static bool write_ppm(const std::string &path, int width, int height, const std::vector<uint8_t> &rgb)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    const bool ok = std::fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok;
}

int main(int argc, char *argv[])
{
    // the renderers behind trace_rays and the gateways are released when main ends, while the HIP
    // runtime is still whole (pt::shutdown)
    struct Release { ~Release() { pt::shutdown(); } } release_at_end;
    std::string scene_name = "orange_plane", scene_file, out;   // the reference's default scene (src/main.cpp:311)
    int width = 356, height = 200, passes = 100, batch = 0, gpus = 0;
    bool fast = false, viewer = false, log_images = false, exit_at_target = false;
    std::string key_script, image_dir = "../images/";
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { std::cerr << "missing value for " << what << std::endl; std::exit(2); }
            return argv[++i];
        };
        if (a == "-scene") scene_name = next("-scene");
        else if (a == "-file") scene_file = next("-file");
        else if (a == "-width") width = std::atoi(next("-width"));
        else if (a == "-height") height = std::atoi(next("-height"));
        else if (a == "-passes") passes = std::atoi(next("-passes"));
        else if (a == "-batch") batch = std::atoi(next("-batch"));
        else if (a == "-gpus") gpus = std::atoi(next("-gpus"));
        else if (a == "-out") out = next("-out");
        else if (a == "-fast") fast = true;
        else if (a == "-log") { viewer = log_images = true; }
        else if (a == "-exit") { viewer = exit_at_target = true; }
        else if (a == "-keys") { viewer = true; key_script = next("-keys"); }
        else if (a == "-images") { viewer = true; image_dir = next("-images"); if (!image_dir.empty() && image_dir.back() != '/') image_dir += '/'; }
        else { std::cerr << "unknown argument " << a << std::endl; return 2; }
    }
    if (width < 1 || height < 1 || passes < 1 || batch < 0 || gpus < 0) { std::cerr << "bad size" << std::endl; return 2; }
    if (gpus > 0 && (batch == 0 || viewer)) { std::cerr << "-gpus needs -batch and no viewer loop" << std::endl; return 2; }

    try {
        set_screen_size(width, height);
        std::optional<Scene> scene;
        if (!scene_file.empty()) {
            std::ifstream in(scene_file);
            if (!in) { std::cerr << "cannot open " << scene_file << std::endl; return 2; }
            const size_t cut = scene_file.find_last_of('/');
            try { scene.emplace(pt::load_scene(in, cut == std::string::npos ? std::string() : scene_file.substr(0, cut))); }
            catch (const pt::SceneFileError &e) { std::cerr << e.what() << std::endl; return 2; }
            scene_name = scene_file;
        } else if (!Scenes::by_name(scene_name.c_str(), scene)) {
            std::cerr << "unknown scene " << scene_name << std::endl;
            return 2;
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> rgb;
        double mean[3] = {0, 0, 0};
        if (viewer) {
            // frames to replay: (count, dt, keys)
            struct Span { long frames; float dt; pt::KeyState keys; };
            std::vector<Span> spans;
            if (key_script.empty()) {
                spans.push_back({passes, 16.0f, pt::KeyState{}});
            } else {
                std::ifstream in(key_script);
                if (!in) { std::cerr << "cannot open " << key_script << std::endl; return 2; }
                std::string line;
                int line_no = 0;
                while (std::getline(in, line)) {
                    line_no++;
                    const size_t hash = line.find('#');
                    if (hash != std::string::npos) line.erase(hash);
                    std::istringstream ss(line);
                    Span sp{};
                    std::string keys;
                    if (!(ss >> sp.frames)) continue;
                    if (!(ss >> sp.dt >> keys) || sp.frames < 0 || !(sp.dt >= 0) || !pt::KeyState::parse(keys, sp.keys)) {
                        std::cerr << key_script << " line " << line_no << ": expected <frames> <ms> <keys>" << std::endl;
                        return 2;
                    }
                    spans.push_back(sp);
                }
            }
            pt::Viewer view(*scene, width, height, batch > 0 ? static_cast<uint32_t>(batch) : 1u, 0, image_dir,
                            PT_MODE_MEGAKERNEL | (fast ? PT_NUM_FAST : PT_NUM_EXACT));
            view.logging.save_image = log_images;
            view.logging.exit = exit_at_target;
            if (exit_at_target) std::cerr << "will exit at " << view.logging.next_target << std::endl;
            bool go = true;
            long frames = 0;
            for (const Span &sp : spans)
                for (long f = 0; f < sp.frames && go; f++, frames++) go = view.frame(sp.keys, sp.dt);
            const std::vector<float> sums = view.renderer().read_sums();
            for (size_t i = 0; i < sums.size(); i += 3) { mean[0] += sums[i]; mean[1] += sums[i + 1]; mean[2] += sums[i + 2]; }
            if (!out.empty()) rgb = view.renderer().tonemap();
            const vec3 &p = scene->camera.get_position();
            const vec2 &r = scene->camera.get_rotation();
            std::printf("frames=%ld camera_pos=%.9g,%.9g,%.9g camera_rot=%.9g,%.9g dof=%d,%.9g,%.9g\n", frames, p.x, p.y, p.z,
                        r.x, r.y, scene->DOF_on ? 1 : 0, scene->DOF_aperture_size, scene->DOF_focal_length);
        } else if (batch == 0) {
            BUFFER buffer(width, std::vector<vec3>(height, vec3()));
            // the render step through a pointer of the reference's own type (src/main.cpp:114)
            void (*const render_step)(Scene &, BUFFER &) = &trace_rays;
            for (int pass = 0; pass < passes; pass++) {
                scene->camera.update();
                render_step(*scene, buffer);
                if (TOTAL_SAMPLES % 10 == 0) std::cerr << TOTAL_SAMPLES << std::endl;
            }
            for (const auto &column : buffer)
                for (const vec3 &c : column) { mean[0] += c.x; mean[1] += c.y; mean[2] += c.z; }
            if (!out.empty()) {               // draw(): tone-map buffer / TOTAL_SAMPLES
                std::vector<float> sums(static_cast<size_t>(width) * height * 3);
                for (int u = 0; u < width; u++)
                    for (int v = 0; v < height; v++) {
                        float *p = &sums[3 * (static_cast<size_t>(v) * width + u)];
                        p[0] = buffer[u][v].x; p[1] = buffer[u][v].y; p[2] = buffer[u][v].z;
                    }
                pt::Renderer &r = pt::default_renderer();
                r.write_sums(sums, TOTAL_SAMPLES);
                rgb = r.tonemap();
            }
        } else {
            std::optional<pt::Renderer> many;
            if (gpus > 0) {
                std::vector<int> devices;
                for (int d = 0; d < gpus; d++) devices.push_back(d);
                many.emplace(devices);
            }
            pt::Renderer &r = many ? *many : pt::default_renderer();
            r.sync(*scene, width, height);
            for (int pass = 0; pass < passes; pass++) {
                r.render(static_cast<uint32_t>(batch), 1, fast ? PT_NUM_FAST : PT_NUM_EXACT);
                TOTAL_SAMPLES += batch;
                if ((pass + 1) % 10 == 0) std::cerr << TOTAL_SAMPLES << std::endl;
            }
            const std::vector<float> sums = r.read_sums();
            for (size_t i = 0; i < sums.size(); i += 3) { mean[0] += sums[i]; mean[1] += sums[i + 1]; mean[2] += sums[i + 2]; }
            if (!out.empty()) rgb = r.tonemap();
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double n = static_cast<double>(width) * height * TOTAL_SAMPLES;
        std::printf("scene=%s size=%dx%d gpus=%d samples_per_pixel=%u seconds=%.3f Msamples_per_s=%.2f mean_rgb=%.6f,%.6f,%.6f\n",
                    scene_name.c_str(), width, height, gpus > 0 ? gpus : 1, TOTAL_SAMPLES, secs, n / secs / 1e6,
                    mean[0] / n, mean[1] / n, mean[2] / n);
        if (!out.empty()) {
            if (!write_ppm(out, width, height, rgb)) { std::cerr << "cannot write " << out << std::endl; return 1; }
            std::cerr << "saved image to " << out << std::endl;
        }
    } catch (const pt::Error &e) {
        std::cerr << "render failed (" << e.code << "): " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
