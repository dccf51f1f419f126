"""Decodes the reference's assets/textures/earthmap.jpg (the image behind ImageTexture::new(EARTH_TEXTURE_BYTES),
src/lib.rs:767, src/texture.rs:84-98) into RGB8 texels, rows top-down, and stores them as
tests/golden/earthmap_1024x512_rgb8.npz.

Run in the build container, where /root/reference exists and PIL is importable; the fixture travels to the GPU box,
PIL and the reference need not. The reference decodes with the `image` crate (^0.22.3, version unpinned: no
Cargo.lock); JPEG decoders may differ by one code value per channel, so the texel VALUES are "parity unpinned" against
the crate's own decode; the oracle and the device share these exact texels.
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/textures/earthmap.jpg"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "earthmap_1024x512_rgb8.npz")

im = Image.open(SRC).convert("RGB")
rgb = np.asarray(im, dtype=np.uint8)
assert rgb.shape == (512, 1024, 3), rgb.shape
np.savez_compressed(OUT, rgb8=rgb, source=np.array("assets/textures/earthmap.jpg"),
                    sha256_of_texels=np.array(hashlib.sha256(rgb.tobytes()).hexdigest()))
print(OUT, rgb.shape, hashlib.sha256(rgb.tobytes()).hexdigest())
