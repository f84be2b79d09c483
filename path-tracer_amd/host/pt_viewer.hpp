// The reference's interactive loop (src/main.cpp:170-326) with the window taken
// out.  The reference reads SDL's key state, moves the camera and the depth-of-
// field settings, clears the image when they changed, traces one pass, draws,
// and logs; here the key state is plain data (filled by a replay script in
// pt_headless, or by whatever window toolkit a front-end has), the pass runs on
// the GPU through pt::Renderer, "draw" is the device tone map, and saved images
// are BMP files named like the reference's.
#pragma once

#include <bitset>
#include <cstdio>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>

#include "pt_render.hpp"

namespace pt {

// The keys handle_input looks at (src/main.cpp:178-247).
enum Key {
    KEY_UP, KEY_DOWN, KEY_LEFT, KEY_RIGHT,          // rotate
    KEY_W, KEY_A, KEY_S, KEY_D, KEY_Q, KEY_E,       // move
    KEY_1, KEY_2,                                   // image saving on / off
    KEY_0,                                          // print the camera pose
    KEY_U, KEY_J,                                   // depth of field on / off
    KEY_I, KEY_K,                                   // focal length + / -
    KEY_O, KEY_L,                                   // aperture + / -
    KEY_COUNT
};

struct KeyState {
    std::bitset<KEY_COUNT> down;
    bool operator[](Key k) const { return down[k]; }

    // "w+left+u" -> those keys held; "-" or "" -> none.  False on an unknown name.
    static bool parse(const std::string &spec, KeyState &out)
    {
        static const char *names[KEY_COUNT] = {"up", "down", "left", "right", "w", "a", "s", "d", "q", "e",
                                                "1", "2", "0", "u", "j", "i", "k", "o", "l"};
        out.down.reset();
        if (spec.empty() || spec == "-") return true;
        std::stringstream ss(spec);
        std::string name;
        while (std::getline(ss, name, '+')) {
            int found = -1;
            for (int k = 0; k < KEY_COUNT; k++)
                if (name == names[k]) found = k;
            if (found < 0) return false;
            out.down.set(found);
        }
        return true;
    }
};

// src/main.cpp:25-29
struct Logging {
    bool save_image = false;
    bool exit = false;
    unsigned next_target = 100;
};

// The key bindings of src/main.cpp:170-251 as data.  Every control is a pair of keys; the
// first wins when both are held (the reference's `if ... else if`).
struct KeyPair { Key first, second; };
constexpr KeyPair PITCH_KEYS = {KEY_UP, KEY_DOWN}, YAW_KEYS = {KEY_RIGHT, KEY_LEFT};
constexpr KeyPair STRAFE_KEYS = {KEY_D, KEY_A}, LIFT_KEYS = {KEY_Q, KEY_E}, WALK_KEYS = {KEY_W, KEY_S};
constexpr KeyPair SAVING_KEYS = {KEY_1, KEY_2}, DOF_KEYS = {KEY_U, KEY_J};
constexpr KeyPair FOCAL_KEYS = {KEY_I, KEY_K}, APERTURE_KEYS = {KEY_O, KEY_L};
constexpr float FOCAL_STEP = 0.1f, APERTURE_STEP = 0.01f;      // src/main.cpp:228-247

// +1 when the pair's first key is held, -1 when only its second is, 0 when neither.
inline int pair_sign(const KeyState &keys, KeyPair p) { return keys[p.first] ? 1 : keys[p.second] ? -1 : 0; }

// One frame's worth of input (src/main.cpp:170-251).  dt in milliseconds; speeds are one
// unit (radian) per second.  Returns true when the image must restart because a
// depth-of-field key was held, whether or not a value changed; camera motion reports
// through Camera::update(), as in the reference.
inline bool handle_input(Scene &scene, float dt, const KeyState &keys, Logging &logging, std::ostream &msg = std::cerr)
{
    const float step = 1 * dt / 1000;
    scene.camera.rotate(pair_sign(keys, PITCH_KEYS) * step, pair_sign(keys, YAW_KEYS) * step);
    scene.camera.move(pair_sign(keys, STRAFE_KEYS) * step, pair_sign(keys, LIFT_KEYS) * step,
                      pair_sign(keys, WALK_KEYS) * step);

    if (const int saving = pair_sign(keys, SAVING_KEYS)) {
        logging.save_image = saving > 0;
        msg << (saving > 0 ? "image saving enabled" : "image saving disabled") << std::endl;
    }
    if (keys[KEY_0]) {
        const vec3 &p = scene.camera.get_position();
        const vec2 &r = scene.camera.get_rotation();
        char buf[160];
        std::snprintf(buf, sizeof buf, "Camera pos: (%f,%f,%f)\nCamera rot: (%f, %f)", p.x, p.y, p.z, r.x, r.y);
        std::cout << buf << std::endl;
    }

    bool restart = false;
    if (const int dof = pair_sign(keys, DOF_KEYS)) {
        restart = true;
        scene.DOF_on = dof > 0;
        msg << (dof > 0 ? "depth of field enabled" : "depth of field disabled") << std::endl;
    }
    if (const int focal = pair_sign(keys, FOCAL_KEYS)) {
        restart = true;
        if (focal > 0) scene.DOF_focal_length += FOCAL_STEP; else scene.DOF_focal_length -= FOCAL_STEP;
        msg << "DoF focal length: " << scene.DOF_focal_length << std::endl;
    }
    if (const int aperture = pair_sign(keys, APERTURE_KEYS)) {
        restart = true;
        if (aperture > 0) scene.DOF_aperture_size += APERTURE_STEP; else scene.DOF_aperture_size -= APERTURE_STEP;
        msg << "DoF aperture size: " << scene.DOF_aperture_size << std::endl;
    }
    return restart;
}

// A 24-bit uncompressed BMP (bottom-up rows of B,G,R padded to four bytes), the
// form SDL_SaveBMP gives the reference's 32-bit screen surface (src/main.cpp:283).
// rgb: width*height*3 bytes, row-major from the top.
inline bool write_bmp(const std::string &path, int width, int height, const std::vector<uint8_t> &rgb)
{
    if (width < 1 || height < 1 || rgb.size() != static_cast<size_t>(width) * height * 3) return false;
    const uint32_t row = (static_cast<uint32_t>(width) * 3 + 3) & ~3u;
    const uint32_t data = row * static_cast<uint32_t>(height);
    uint8_t head[54] = {0};
    auto put32 = [&](int at, uint32_t v) { for (int k = 0; k < 4; k++) head[at + k] = static_cast<uint8_t>(v >> (8 * k)); };
    head[0] = 'B'; head[1] = 'M';
    put32(2, 54 + data);        // file size
    put32(10, 54);              // offset of the pixel data
    put32(14, 40);              // BITMAPINFOHEADER
    put32(18, static_cast<uint32_t>(width));
    put32(22, static_cast<uint32_t>(height));
    head[26] = 1;               // planes
    head[28] = 24;              // bits per pixel
    put32(34, data);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(head, 1, sizeof head, f) == sizeof head;
    std::vector<uint8_t> line(row, 0);
    for (int y = height - 1; y >= 0 && ok; y--) {
        const uint8_t *src = &rgb[static_cast<size_t>(y) * width * 3];
        for (int x = 0; x < width; x++) {
            line[3 * x + 0] = src[3 * x + 2];
            line[3 * x + 1] = src[3 * x + 1];
            line[3 * x + 2] = src[3 * x + 0];
        }
        ok = std::fwrite(line.data(), 1, row, f) == row;
    }
    return std::fclose(f) == 0 && ok;
}

// The loop body of the reference's main() (src/main.cpp:319-323) for one scene.
class Viewer {
    Scene &scene_;
    Renderer renderer_;
    int width_, height_;
    uint32_t passes_per_frame_;
    int mode_;
    std::string image_dir_;
    std::vector<uint8_t> screen_;

  public:
    Logging logging;
    // Called with the 8-bit image after every frame's draw, if set: a window
    // front-end blits it.  When unset, the image is tone-mapped only to be saved.
    std::function<void(const std::vector<uint8_t> &)> on_draw;

    // passes_per_frame: samples per pixel added per frame; the reference adds 1.
    Viewer(Scene &scene, int width, int height, uint32_t passes_per_frame = 1, int device = 0,
           std::string image_dir = "../images/", int mode = PT_MODE_MEGAKERNEL | PT_NUM_EXACT)
        : scene_(scene), renderer_(device), width_(width), height_(height),
          passes_per_frame_(passes_per_frame ? passes_per_frame : 1), mode_(mode), image_dir_(std::move(image_dir))
    {
        set_screen_size(width, height);
        TOTAL_SAMPLES = 0;
        renderer_.sync(scene_, width_, height_);
    }

    Renderer &renderer() { return renderer_; }
    const std::vector<uint8_t> &screen() const { return screen_; }

    // update() (src/main.cpp:253-273): input, camera, clear when needed, one pass.
    void update(const KeyState &keys, float dt)
    {
        bool clear_buffer = handle_input(scene_, dt, keys, logging);
        clear_buffer |= scene_.camera.update();
        const bool cleared = renderer_.sync_camera(scene_, width_, height_);   // a changed camera or DOF setting clears
        if (clear_buffer && !cleared) renderer_.clear();                  // a DOF key that changed nothing clears too
        if (clear_buffer || cleared) TOTAL_SAMPLES = 0;
        renderer_.render(passes_per_frame_, 1, mode_);
        TOTAL_SAMPLES += passes_per_frame_;
    }

    // draw() (src/main.cpp:153-168)
    void draw() { screen_ = renderer_.tonemap(); }

    // log() (src/main.cpp:275-294); needs_screen says whether this frame's image is wanted.
    bool needs_screen() const { return on_draw || (TOTAL_SAMPLES >= logging.next_target && logging.save_image); }

    bool log()
    {
        if (TOTAL_SAMPLES % 10 == 0) std::cerr << TOTAL_SAMPLES << std::endl;
        if (TOTAL_SAMPLES >= logging.next_target) {
            if (logging.save_image) {
                const std::string path = image_dir_ + std::to_string(TOTAL_SAMPLES) + ".bmp";
                if (write_bmp(path, width_, height_, screen_)) std::cerr << "saved image to " << path << std::endl;
                else std::cerr << "could not save image to " << path << std::endl;
                while (logging.next_target <= TOTAL_SAMPLES) logging.next_target *= 2;
            }
            if (logging.exit) return false;
        }
        return true;
    }

    // One turn of the main loop; false when log() asks to stop.
    bool frame(const KeyState &keys, float dt)
    {
        update(keys, dt);
        if (needs_screen()) {
            draw();
            if (on_draw) on_draw(screen_);
        }
        return log();
    }
};

}  // namespace pt
