"""The oracle and the HIP path against the images the reference itself ships.

The reference has no tests, but images/ holds outputs of its own renderer at its native sizes
(tests/golden/make_reference_images.py copies their pixels into reference_images.npz).  Two
checks per image, on the CPU for the oracle and on the GPU for the HIP path:

* silhouette: every pixel of the image is classified by colour into one of the scene's diffuse
  materials (prototypes = the image's own median colour over each material's region); away
  from region boundaries the classes must agree with the primary-hit index map of
  closest_intersection (src/main.cpp:56-67) on the pinhole rays of src/main.cpp:50-54 for
  >= 99.5 % of the pixels (>= 99 % of those within 6 px of a silhouette), and the agreement
  must collapse when the map is displaced by 4 px or turned upside down (so the check has
  teeth).  This pins camera, y-down, scene constants and scene order to something the
  reference produced.
* radiance: a render, tone-mapped like draw() (src/main.cpp:146-162), must equal the image in
  10 x 10 (8 x 8) block means within a bound derived from the two images' own pixel noise.

orange_plane_alt_800_55s.png - the scene the program starts with, untouched - is reproduced by the
loop of src/main.cpp:84-109 exactly as the source has it (mean levels to 0.02 of a grey level),
which pins russian roulette and its 1/p.  The other six are reproduced when paths stop after 7
segments (8 for red_corner) - dof_demo then to 0.03 of a grey level in every channel's mean,
cornell to 0.1, the second orange_plane image to 0.14 - and NOT by the unbounded loop (long paths add 20 % of red light in cornell_2, a
quarter of a grey level in dof_demo and red_corner): those were evidently rendered by an earlier
revision with a bounded path length (test_which_path_loop_rendered_the_image weighs caps 6, 7, 8,
the source's loop and russian roulette without its 1/p against each image at the image's own
noise level).  What no image can pin is the bit-level operation order of GLM.
"""
import os

import numpy as np
import pytest
from scipy.ndimage import maximum_filter, minimum_filter, uniform_filter

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_images.npz")

# key -> (scene, camera overrides, path cap (0 = the source's unbounded loop), block size,
#         bound on the mean difference per channel in grey levels)
# Measured with the oracle at 512 spp: |bias| 0.05 (red_corner), 0.02, 0.01 (dof_demo), 0.3
# (cornell), 1.4 red / 0.2 green and blue (cornell_2: its image fits a cap of 7 less exactly,
# -2.4 % of red on the floor, than cornell's does).
CASES = {
    "red_corner": ("red_corner", {}, 8, 8, 0.5),
    "dof_demo_on": ("dof_demo", {}, 7, 8, 0.15),
    "dof_demo_off": ("dof_demo", {"dof_on": 0}, 7, 8, 0.15),
    "cornell": ("cornell", {}, 7, 10, 0.8),
    "cornell_2": ("cornell_2", {}, 7, 10, 2.5),
    # the scene src/main.cpp:311 starts with, exactly as the source has it - paths of any length,
    # russian roulette with its 1/p; and the same scene with the lens focused at 3.0 (key K),
    # which like the five above shows paths of 7 segments.  Poses found by
    # tests/golden/fit_orange_plane.py
    "orange_plane_alt": ("orange_plane", {}, 0, 8, 0.4),
    "orange_plane": ("orange_plane", {"dof_focal_length": 3.0}, 7, 8, 0.4),
}


# Silhouette bars (agreement of all scored pixels, of those within 6 px of a silhouette, share of
# the image scored, agreement of a map moved by 4 px at most) where they differ from 0.995 / 0.99 /
# 0.7 / 0.9: the lens of Scenes::orange_plane (aperture 0.15) blurs whatever is out of focus over
# tens of pixels, and the map compared with is the pinhole's; what pins these two images' geometry
# is the radiance check, whose 8 x 8 block means move by tens of grey levels with the pose.
SILHOUETTE_BARS = {"orange_plane_alt": (0.97, 0.85, 0.6, 0.86), "orange_plane": (0.97, 0.85, 0.6, 0.86)}


@pytest.fixture(scope="module")
def images():
    return np.load(GOLDEN, allow_pickle=False)


def scene_for(mod, key, image):
    name, overrides, cap, block, _ = CASES[key]
    h, w = image.shape[:2]
    sc = mod.builtin_scene(name, w, h)
    for k, v in overrides.items():
        sc.camera[k][0] = v
    return sc, cap, block


def material_classes(sc, idx):
    """Per pixel: the class of the surface hit (surfaces with equal diffuse materials share a
    class), -1 for the background, < -1 for mirror and glass surfaces (which have no colour of
    their own and are not scored)."""
    mats = sc.materials[sc.surfaces["material"]]
    classes = {}
    surf_cls = np.zeros(len(mats), int)
    for i, m in enumerate(mats):
        if m["type"] != 0:
            surf_cls[i] = -2 - i
            continue
        key = (tuple(m["color"]), tuple(m["emittance"]))
        surf_cls[i] = classes.setdefault(key, len(classes))
    return np.where(idx >= 0, surf_cls[np.maximum(idx, 0)], -1)


def colour_features(img):
    """Chromaticity and (weakly) brightness of the 5 x 5 box-filtered image."""
    b = np.stack([uniform_filter(img[..., c].astype(float), 5) for c in range(3)], -1)
    s = b.sum(-1, keepdims=True) + 30.0
    return np.concatenate([b / s, 0.6 * s / (3 * 255.0)], -1)


def silhouette_agreement(image, cls, proto_cls=None):
    """Classifies the image's pixels by colour (prototypes: the image's median colour over the
    deep interior of each class of proto_cls) and compares with cls: the share of agreeing
    pixels among those 2 px or more from a boundary of cls, and among those of them that are
    also within 6 px of one (where a misplaced silhouette shows first)."""
    proto_cls = cls if proto_cls is None else proto_cls
    feat = colour_features(image)
    deep = minimum_filter(proto_cls, 9) == maximum_filter(proto_cls, 9)          # 4 px from any boundary
    keys, protos = [], []
    for c in np.unique(proto_cls):
        m = (proto_cls == c) & deep
        if c < -1 or m.sum() < 50:
            continue
        keys.append(c)
        protos.append(np.median(feat[m], 0))
    protos = np.stack(protos)
    d = ((feat[:, :, None, :] - protos[None, None]) ** 2).sum(-1)
    pred = np.array(keys)[d.argmin(-1)]
    interior = (minimum_filter(cls, 5) == maximum_filter(cls, 5)) & np.isin(cls, keys)
    band = interior & (minimum_filter(cls, 13) != maximum_filter(cls, 13))
    return float((pred == cls)[interior].mean()), float((pred == cls)[band].mean()), float(interior.mean()), len(keys)


def check_silhouette(key, image, sc, idx):
    cls = material_classes(sc, idx)
    agree, agree_band, scored, n_classes = silhouette_agreement(image, cls)
    # the same prototypes against a map displaced by 4 pixels, and against the map upside down
    _, shifted_band, _, _ = silhouette_agreement(image, np.roll(cls, 4, axis=1), cls)
    _, lifted_band, _, _ = silhouette_agreement(image, np.roll(cls, 4, axis=0), cls)
    flipped, _, _, _ = silhouette_agreement(image, cls[::-1], cls)
    print("%s: %d classes, %.4f of %.0f %% scored pixels agree, %.4f within 6 px of a silhouette "
          "(map moved 4 px sideways: %.3f, 4 px down: %.3f; upside down: %.3f of all)"
          % (key, n_classes, agree, 100 * scored, agree_band, shifted_band, lifted_band, flipped))
    agree_min, band_min, scored_min, moved_max = SILHOUETTE_BARS.get(key, (0.995, 0.99, 0.7, 0.9))
    assert n_classes >= 3 and scored > scored_min
    assert agree >= agree_min and agree_band >= band_min
    assert min(shifted_band, lifted_band) < moved_max and flipped < 0.9


def block_means(img, block):
    h, w = img.shape[:2]
    hh, ww = h // block * block, w // block * block
    return img[:hh, :ww].astype(float).reshape(hh // block, block, ww // block, block, 3).mean((1, 3))


def pixel_noise(img):
    """Standard deviation of the pixel noise of a tone-mapped render: the residual against the
    3 x 3 local mean (x sqrt(9/8)), median over the lit part of the image (local mean above 16
    grey levels: a black background has no noise to show) so that edges do not count."""
    local = np.stack([uniform_filter(img[..., c].astype(float), 3) for c in range(3)], -1)
    r = img.astype(float) - local
    lit = local.max(-1) > 16
    return float(np.sqrt(9.0 / 8.0) * 1.4826 * np.median(np.abs(r[lit])))


def check_radiance(key, image, ldr, block, what):
    d = block_means(ldr, block) - block_means(image, block)
    # block means of two independent noisy renders differ by noise of this expected magnitude
    floor = np.sqrt(2 / np.pi) * np.hypot(pixel_noise(image), pixel_noise(ldr)) / block
    abs_mean = float(np.abs(d).mean())
    bias = d.mean((0, 1))
    print("%s %s: block |diff| mean %.2f LSB (noise floor %.2f), bias per channel %s, max %.1f"
          % (key, what, abs_mean, floor, bias.round(2), np.abs(d).max()))
    assert abs_mean <= 2.0 * floor + 0.4
    assert np.abs(bias).max() <= CASES[key][4]
    return bias


# ------------------------------------------------------------------ the oracle (CPU)

@pytest.mark.parametrize("key", list(CASES))
def test_oracle_silhouettes_match_reference_image(orc, images, key):
    sc, _, _ = scene_for(orc, key, images[key])
    idx, _ = orc.primary_hits(sc)
    check_silhouette(key, images[key], sc, idx)


def test_cornell_images_were_not_rendered_by_the_unbounded_loop(orc, images):
    """What the images do not pin: as the source stands (no bound on path length) the box
    holds more red light than the shipped image shows."""
    sc, _, block = scene_for(orc, "cornell_2", images["cornell_2"])
    acc, _ = orc.render(sc, 128, seed=1, threads=8)
    d = block_means(orc.tonemap(acc, 128), block) - block_means(images["cornell_2"], block)
    print("cornell_2, unbounded paths: bias per channel", d.mean((0, 1)).round(2))
    assert d.mean((0, 1))[0] > 5.0


# Samples per pixel of the shipped image: the tone map is concave, so the mean grey level of a
# noisy render lies below that of a converged one (cornell_2, cap 7: -5.5 grey levels of red at 64
# spp, -2.0 at 256, -1.2 at 1024), and hypotheses about an image can only be weighed at its own
# noise.  The file names give the number for four images (cornell_800_276s.png: 800 samples per
# pixel in 276 s, and so on); for the others it is the number at which the oracle's render has the
# image's pixel noise (which gives 740 and 655 for the two 800-sample images whose noise can be
# estimated; DOF_off.png is converged beyond what the estimate resolves and is given the samples
# of its twin).
MATCHED_SPP = {"red_corner": 800, "dof_demo_on": 1024, "dof_demo_off": 1024, "cornell": 107, "cornell_2": 800,
               "orange_plane_alt": 800, "orange_plane": 1600}

# What could have rendered the images: the loop of src/main.cpp:84-109 as it stands ("source"), the
# same with paths stopped after N segments ("cap N"), and the same with russian roulette's survivors
# not divided by their survival probability (src/main.cpp:104 left out: "no 1/p").
HYPOTHESES = [("cap 6", 6, True), ("cap 7", 7, True), ("cap 8", 8, True), ("source", 0, True), ("no 1/p", 0, False)]


@pytest.mark.parametrize("key", list(CASES))
def test_which_path_loop_rendered_the_image(orc, images, key):
    """Every hypothesis rendered at the image's own noise level, scored by the largest mean
    difference of a channel in grey levels.  Measured (cap 6 / 7 / 8 / source / no 1/p):
    orange_plane_alt 1.8 / 1.0 / 0.6 / 0.02 / 0.24: this image of the default scene is what the
    source renders, paths of any length and russian roulette with its 1/p, to a fiftieth of a grey
    level - which pins the energy of long paths (the orange plane reflects all red light, so its
    red channel is the most sensitive to them of any shipped image).  The six other images are
    best reproduced by paths bounded at 7 segments (8 for red_corner, whose paths are short) and
    NOT by the unbounded loop: orange_plane 0.59 / 0.14 / 0.52 / 1.16 / 0.89, dof_demo 0.20 /
    0.03 / 0.13 / 0.24 / 0.11, cornell 1.0 / 0.1 / 0.5 / 2.0 / 0.9, cornell_2 5 / 1.3 / 1.9 / 12 /
    6.8, red_corner 0.36 / 0.17 / 0.04 / 0.23 / 0.06 - they were rendered by a revision that
    bounded its paths.  cornell_2's best fit still leaves about -1.3 grey levels of red (and cap 8
    +1.9): the one residue no hypothesis here explains."""
    sc, cap, block = scene_for(orc, key, images[key])
    spp = MATCHED_SPP[key]
    score = {}
    for name, n, comp in HYPOTHESES:
        orc.set_path_cap(n)
        orc.set_rr_compensation(comp)
        try:
            acc, _ = orc.render(sc, spp, seed=2, threads=8)
        finally:
            orc.set_path_cap(0)
            orc.set_rr_compensation(True)
        ldr = orc.tonemap(acc, spp)
        if name == "cap 7" and pixel_noise(images[key]) > 0.05 and key == "cornell":
            ratio = pixel_noise(ldr) / pixel_noise(images[key])
            assert 0.85 <= ratio <= 1.15, "render noise %.2fx the image's: recalibrate MATCHED_SPP" % ratio
        if (n, comp) == (cap, True):
            # the hypothesis that fits: the image in block means, within the two images' own noise
            check_radiance(key, images[key], ldr, block, "oracle, " + name)
        bias = (block_means(ldr, block) - block_means(images[key], block)).mean((0, 1))
        score[name] = float(np.abs(bias).max())
        print("%s at %d spp, %-7s mean difference per channel %s" % (key, spp, name + ":", bias.round(2)))
    best = min(score, key=score.get)
    assert best == ("cap %d" % cap if cap else "source"), score
    if cap == 0:
        # the source's loop, and nothing shorter: a bound of 7 segments or a roulette without its
        # 1/p each lose light the image has
        assert score["source"] <= 0.15, score
        for other in ("cap 7", "no 1/p"):
            assert score[other] >= 3 * score["source"] and score[other] >= score["source"] + 0.15, (other, score)
        return
    # the loop as the source has it is not the best fit here, and not by a whisker
    assert score["source"] >= 3 * score[best] and score["source"] >= score[best] + 0.15, score
    if cap == 7:
        margin = 1.3 if key == "cornell_2" else 2.5
        for other in ("cap 6", "cap 8", "no 1/p"):
            assert score[other] >= margin * score["cap 7"], (other, score)
    else:
        assert score["cap 7"] >= 3 * score[best] and score["cap 6"] >= 3 * score[best], score


# ------------------------------------------------------------------ the HIP path (GPU)

@pytest.mark.gpu
@pytest.mark.parametrize("key", list(CASES))
def test_gpu_silhouettes_match_reference_image(ctx, pt, images, key):
    sc, _, _ = scene_for(pt, key, images[key])
    ctx.load(sc)
    idx, _ = ctx.primary_hits(pt.NUM_EXACT)
    check_silhouette(key, images[key], sc, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("numeric", ["exact", "fast"])
@pytest.mark.parametrize("key", list(CASES))
def test_gpu_radiance_matches_reference_image(ctx, pt, images, key, numeric):
    sc, cap, block = scene_for(pt, key, images[key])
    spp = 1024
    ctx.load(sc)
    ctx.set_max_depth(cap if cap else pt.MAX_DEPTH)
    try:
        ctx.render(spp, 0, 1, pt.NUM_FAST if numeric == "fast" else pt.NUM_EXACT)
        ldr = ctx.tonemap_rgb8()
    finally:
        ctx.set_max_depth(pt.MAX_DEPTH)
    check_radiance(key, images[key], ldr, block, "HIP %s, path cap %d" % (numeric, cap))
