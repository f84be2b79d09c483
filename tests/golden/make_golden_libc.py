"""Mints tests/golden/render_libc_64_4096spp.npz: converged 64 x 64 images rendered by the
oracle with the REFERENCE'S OWN random numbers - libc rand() through random_real
(src/main.cpp:42-44), one serial stream over the whole image as trace_rays draws it
(src/main.cpp:114-143) - for the statistical gate of tests/test_gpu_parity.py.  The HIP path
uses a counter-based generator instead, so its images are statistically independent of these.

Per scene: the mean image of 4096 samples per pixel (8 worker processes x 512 samples, each a
serial rand() stream of its own seed; glibc's rand() is process-global state) and the mean of a
second, independent 4096 (other seeds): the distance between the two is the noise floor.

Build container only (minutes of CPU); the oracle is test infrastructure.
   python tests/golden/make_golden_libc.py
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

SIZE = 64
SPP = 4096
WORKERS = 8
SCENES = ("cornell_2", "orange_plane", "cornell", "glass_stress", "heightfield:6", "heightfield:40")


def oracle_scene(name):
    """Built-in scenes from the oracle's own catalogue; the synthetic ones from the host mirror
    (plain data: surfaces, materials, background, camera)."""
    import pt_oracle
    if name in pt_oracle.SCENE_NAMES:
        return pt_oracle.builtin_scene(name, SIZE, SIZE)
    import pt_amd
    sc = pt_amd.builtin_scene(name, SIZE, SIZE)
    return pt_oracle.Scene(sc.surfaces, sc.materials, sc.bg, sc.camera)


def work(args):
    name, seed, spp = args
    import pt_oracle
    acc, cnt = pt_oracle.render_libc_rand(oracle_scene(name), spp, seed=seed)
    return acc.astype(np.float64), cnt.segments, cnt.samples


if __name__ == "__main__":
    out = {}
    with mp.Pool(WORKERS) as pool:
        for name in SCENES:
            key = name.replace(":", "_")
            for tag, base in (("mean", 1000), ("mean_other_seeds", 2000)):
                parts = pool.map(work, [(name, base + i, SPP // WORKERS) for i in range(WORKERS)])
                total = sum(p[0] for p in parts)
                out["%s_%s" % (key, tag)] = (total / SPP).astype(np.float32)
                if tag == "mean":
                    out[key + "_segments_per_sample"] = np.float64(sum(p[1] for p in parts) / sum(p[2] for p in parts))
            a, b = out[key + "_mean"], out[key + "_mean_other_seeds"]
            print(name, "noise floor per channel", np.sqrt(((a - b) ** 2).mean((0, 1))), "mean", a.mean((0, 1)),
                  "segments/sample", out[key + "_segments_per_sample"], flush=True)
    np.savez_compressed(os.path.join(HERE, "render_libc_64_4096spp.npz"), **out)
