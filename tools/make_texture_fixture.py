"""Makes tests/golden/dancer/textures/Material.001_normal.png: the dancer asset's normal map (4096x4096 RGB, 4 MB in the
reference's assets/models/a_contortionist_dancer/textures/) reduced to 1024x1024 by a 4x4 box filter and re-encoded,
so the glTF's `images[2].uri` resolves inside the fixture directory at a size fit for the repository (CC-BY-4.0, credit
in tests/golden/dancer/license.txt).  The other two images the glTF names (baseColor, metallicRoughness) are absent
from the reference's own assets directory, and stay absent here: the loader has to cope with that.

usage: python tools/make_texture_fixture.py   (needs /root/reference and Pillow; run once, output committed)"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/assets/models/a_contortionist_dancer/textures/Material.001_normal.png"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dancer", "textures", "Material.001_normal.png")

src = np.asarray(Image.open(SRC).convert("RGB")).astype(np.uint32)
h, w, _ = src.shape
f = 4
box = src.reshape(h // f, f, w // f, f, 3).sum(axis=(1, 3))
out = ((box + f * f // 2) // (f * f)).astype(np.uint8)
os.makedirs(os.path.dirname(DST), exist_ok=True)
Image.fromarray(out, "RGB").save(DST, "PNG", optimize=True)
print(DST, out.shape, os.path.getsize(DST), "bytes")
