#!/usr/bin/env python3
"""Writes tests/golden/small_cases.json: integer-exact digests of the oracle's output on the seeded small cases
(winning-primitive image, stored depth bits, covered-pixel count, sRGB8 checksum).  The GPU tests compare the HIP path
against these committed vectors as well as against the live oracle, so a change that moved both would still be caught."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge   # noqa: E402
import oracle_binding as ob    # noqa: E402


def digest(scene):
    r = ob.render(scene, want_bgra8=True)
    covered = r["prim"] != 0xFFFFFFFF
    depth_bits = np.where(covered, r["depth"].view(np.uint32), 0).astype(np.uint32)
    return {"name": scene.name, "width": scene.width, "height": scene.height, "triangles": scene.num_triangles,
            "covered": int(covered.sum()), "prim_sha256": hashlib.sha256(r["prim"].tobytes()).hexdigest(),
            "depth_sha256": hashlib.sha256(depth_bits.tobytes()).hexdigest(),
            "bgra8_sum": [int(x) for x in r["bgra8"].reshape(-1, 4).astype(np.uint64).sum(axis=0)]}


if __name__ == "__main__":
    m = ge.load_package()
    out = {k: digest(fn()) for k, fn in m.scenes.SMALL_CASES.items()}
    out["dancer_320x180"] = digest(m.scenes.gltf_model(os.path.join(ROOT, "tests", "golden", "dancer", "scene.gltf"), 320, 180))
    path = os.path.join(ROOT, "tests", "golden", "small_cases.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(out), "cases")
