"""Finds the camera pose and depth-of-field setting under which the reference rendered its two
images of Scenes::orange_plane (src/scene.hpp:73-93, the scene src/main.cpp:311 starts with).

Build container only (reads /root/reference/images, needs PIL and the oracle); nothing is
written: the result is recorded in tests/test_reference_images.py (CASES), and the images' pixels
reach the tests through make_reference_images.py.

The images could have been taken after the author had moved the camera with the keys of
src/main.cpp:170-251: W/S/A/D/Q/E move it (continuously: the step is speed x frame time), the
arrows turn it, I/K change the focal length of the lens in steps of 0.1 and O/L the aperture in
steps of 0.01 (src/main.cpp:220-248).  So the pose is seven numbers - position, pitch, yaw, focal
length, aperture - of which the last two lie on a grid.  Two stages:

1. all seven fitted by rendering with the oracle and minimising the squared difference of 4 x 4
   block means of the tone-mapped image against the PNG (a simplex search with a fixed random
   seed).  Result: the camera never moved - position within 0.04 and angles within 0.01 rad of
   Scenes::orange_plane's own (0, 0, -3), (0, 0) for both images, which is the noise of the fit;
2. at that pose, the lens numbers on their grid, the loop of src/main.cpp:84-109 as it stands.
   Result: orange_plane_alt_800_55s.png is Scenes::orange_plane exactly as the source has it
   (focal length 5, aperture 0.15); orange_plane_1600_104s.png is the same with the focal length
   lowered to 3.0 (key K, twenty steps) and the aperture untouched.  The file names give the
   samples per pixel (800, 1600) and the seconds the render took.

    python tests/golden/fit_orange_plane.py
"""
import os
import sys

import numpy as np
from PIL import Image
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import pt_oracle as orc  # noqa: E402

SRC = "/root/reference/images"
FILES = {"orange_plane": "orange_plane_1600_104s.png", "orange_plane_alt": "orange_plane_alt_800_55s.png"}
# first guesses: the camera some 0.25 nearer than Scenes::orange_plane puts it; the first image
# is focused on the two front spheres, the second on the white one behind (the scene's own 5)
START = {"orange_plane": [0.0, 0.0, -2.75, 0.0, 0.0, 2.8, 0.15],
         "orange_plane_alt": [0.0, 0.0, -2.75, 0.0, 0.0, 5.0, 0.15]}
BLOCK = 4


def block_means(img):
    h, w = img.shape[:2]
    hh, ww = h // BLOCK * BLOCK, w // BLOCK * BLOCK
    return img[:hh, :ww].astype(float).reshape(hh // BLOCK, BLOCK, ww // BLOCK, BLOCK, 3).mean((1, 3))


def scene_at(theta, w, h):
    sc = orc.builtin_scene("orange_plane", w, h)
    # the key handlers change the stored angles, to which the constructor had added OFFSET
    # (src/camera.hpp:10,36): the fitted angles are the stored ones, so no offset is added again
    cam = orc.camera_init(theta[:3], theta[3], theta[4], float(h), 0, w, h)
    cam["dof_on"] = 1
    cam["dof_focal_length"] = theta[5]
    cam["dof_aperture"] = theta[6]
    sc.camera = cam
    return sc


def render(theta, w, h, spp, seed=1):
    orc.set_path_cap(7)
    try:
        acc, _ = orc.render(scene_at(theta, w, h), spp, seed=seed, threads=8)
    finally:
        orc.set_path_cap(0)
    return orc.tonemap(acc, spp)


def fit(image, start):
    h, w = image.shape[:2]
    target = block_means(image)
    evals = [0]

    def cost(theta, spp):
        if theta[6] <= 0.0 or theta[5] <= 0.5:
            return 1e9
        evals[0] += 1
        return float(((block_means(render(theta, w, h, spp)) - target) ** 2).mean())

    theta = np.array(start, float)
    step = np.array([0.05, 0.05, 0.1, 0.02, 0.02, 0.3, 0.03])
    for spp, iters in ((16, 250), (64, 200)):
        simplex = np.vstack([theta] + [theta + step * np.eye(7)[k] for k in range(7)])
        res = minimize(cost, theta, args=(spp,), method="Nelder-Mead",
                       options={"initial_simplex": simplex, "maxfev": iters, "xatol": 1e-4, "fatol": 1e-3})
        theta, step = res.x, step / 3
        print("  %d spp: cost %.3f after %d renders, theta %s" % (spp, res.fun, evals[0], np.round(theta, 4)), flush=True)
    # the lens numbers lie on the key handlers' grid: 5 - 0.1 k and 0.15 + 0.01 m
    lens = np.array([round((theta[5] - 5.0) / 0.1) * 0.1 + 5.0, round((theta[6] - 0.15) / 0.01) * 0.01 + 0.15])

    def cost5(t5, spp):
        return cost(np.concatenate([t5, lens]), spp)

    simplex = np.vstack([theta[:5]] + [theta[:5] + step[:5] * np.eye(5)[k] for k in range(5)])
    res = minimize(cost5, theta[:5], args=(128,), method="Nelder-Mead",
                   options={"initial_simplex": simplex, "maxfev": 150, "xatol": 1e-5, "fatol": 1e-4})
    theta = np.concatenate([res.x, lens])
    print("  lens on its grid %s: cost %.3f, theta %s" % (lens, res.fun, np.round(theta, 4)), flush=True)
    return theta, res.fun


def lens_grid(image, spp):
    """Stage 2: cost over the key handlers' grid of lens numbers at the scene's own pose, paths
    as the source traces them (no cap)."""
    h, w = image.shape[:2]
    target = block_means(image)
    best = None
    for f in np.arange(2.6, 5.61, 0.1):
        for a in np.arange(0.13, 0.191, 0.01):
            sc = scene_at(np.array([0, 0, -3, 2e-5, 1e-5, f, a]), w, h)
            acc, _ = orc.render(sc, spp, seed=4, threads=8)
            c = float(((block_means(orc.tonemap(acc, spp)) - target) ** 2).mean())
            if best is None or c < best[0]:
                best = (c, round(float(f), 1), round(float(a), 2))
    return best


if __name__ == "__main__":
    for key, name in FILES.items():
        image = np.array(Image.open(os.path.join(SRC, name)).convert("RGB"), dtype=np.uint8)
        print(key, name, image.shape, flush=True)
        theta, c = fit(image, START[key])
        print("  stage 1: position %s, pitch and yaw %s, lens %s" % (
            np.round(theta[:3], 3), np.round(theta[3:5], 4), np.round(theta[5:], 2)), flush=True)
        c, f, a = lens_grid(image, 200)
        print("  stage 2 (the scene's own pose): focal length %.1f, aperture %.2f, cost %.2f" % (f, a, c), flush=True)
