"""Copies the pixels of the images the reference ships into tests/golden/reference_images.npz.

Build container only (reads /root/reference, needs PIL); the tests read the .npz.  The PNGs are
outputs of the reference's own renderer (src/main.cpp:275-288 saves the window as a BMP; the
author converted them), at the reference's native sizes:

  diffuse.png              300 x 300   Scenes::cornell     (src/scene.hpp:158-201)
  cornell_800_276s.png     300 x 300   Scenes::cornell_2   (src/scene.hpp:129-156)
  red_corner_800_64s.png   356 x 200   Scenes::red_corner  (src/scene.hpp:95-112)
  DOF_on.png               355 x 200   Scenes::dof_demo    (src/scene.hpp:34-57), depth of field on
  DOF_off.png              355 x 200   the same with depth of field switched off (key J)

  orange_plane_alt_800_55s.png  356 x 200  Scenes::orange_plane (src/scene.hpp:73-93), as the source has it
  orange_plane_1600_104s.png    356 x 200  the same with the lens focused at 3.0 instead of 5 (key K)

(The poses of the last two were found by tests/golden/fit_orange_plane.py; the numbers in the
file names are samples per pixel and seconds.)
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/images"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_images.npz")
FILES = {
    "cornell": "diffuse.png",
    "cornell_2": "cornell_800_276s.png",
    "red_corner": "red_corner_800_64s.png",
    "dof_demo_on": "DOF_on.png",
    "dof_demo_off": "DOF_off.png",
    "orange_plane_alt": "orange_plane_alt_800_55s.png",
    "orange_plane": "orange_plane_1600_104s.png",
}

if __name__ == "__main__":
    arrays = {}
    for key, name in FILES.items():
        arrays[key] = np.array(Image.open(os.path.join(SRC, name)).convert("RGB"), dtype=np.uint8)
        print(key, name, arrays[key].shape)
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
