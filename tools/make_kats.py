#!/usr/bin/env python3
"""Derives the known-answer fixture tests/golden/kats.json from the reference checkout.

Run in the build container only (needs /root/reference).  The fixture holds DATA the reference's own
artefacts pin for the hot path -- no reference source text:
  K1  statistics of screenshots/Hello Triangle.png (the only golden render)
  K2  analytic coverage of the hello triangle at 256x256 under the Vulkan top-left rule
  K3  shader constants of shaders/hlsl/lights.hlsli / pixel/model.hlsl
  K4  default-camera matrix values (crates/scene/src/camera.rs:43-56,110-142 + glam definitions)
  K5  counts of the bundled glTF asset (crates/resources/tests/integration_test.rs:7-83)
  K6  vertex / UBO layouts asserted by the reference's unit tests
"""
import json
import math
import os
import struct
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kats.json")


def k1():
    img = np.array(Image.open(os.path.join(REF, "screenshots", "Hello Triangle.png")).convert("RGBA"))
    h, w = img.shape[:2]
    bg = img[h - 10, 10, :3].tolist()
    # client area: rows whose left-most pixel has the background colour
    rows = [y for y in range(h) if img[y, 5, :3].tolist() == bg]
    y0, y1 = rows[0], rows[-1]
    client = img[y0:y1 + 1, :, :3].astype(np.int32)
    tri = np.any(np.abs(client - np.array(bg)) > 6, axis=2)
    ys, xs = np.where(tri)
    cx, cy = int(round(xs.mean())), int(round(ys.mean()))
    return {
        "source": "screenshots/Hello Triangle.png", "image_size": [w, h], "client_rows": [y0, y1],
        "client_size": [w, y1 - y0 + 1], "background_srgb8": bg,
        "triangle_bbox_x": [int(xs.min()), int(xs.max())], "triangle_bbox_y_client": [int(ys.min()), int(ys.max())],
        "centroid_client": [cx, cy], "centroid_srgb8": client[cy, cx].tolist(),
        "top_row_rgb": client[int(ys.min()) + 3, int(round(xs[ys == ys.min() + 3].mean()))].tolist(),
        "bottom_left_rgb": client[int(ys.max()) - 3, int(xs[ys == ys.max() - 3].min()) + 8].tolist(),
        "bottom_right_rgb": client[int(ys.max()) - 3, int(xs[ys == ys.max() - 3].max()) - 8].tolist(),
        "covered_pixels": int(tri.sum()),
        "note": "window grab of unknown DPI scaling: geometry +-2 px, colour +-2 LSB",
    }


def k2():
    # verts (0,-.5),(-.5,.5),(.5,.5) @256^2 -> snapped (128,64),(64,192),(192,192) px; integer edge functions,
    # pixel centre (x+.5,y+.5), top-left rule -- computed independently of the oracle
    X = [128 * 256, 64 * 256, 192 * 256]
    Y = [64 * 256, 192 * 256, 192 * 256]
    S = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0])
    if S < 0:
        X[1], X[2], Y[1], Y[2] = X[2], X[1], Y[2], Y[1]
    rows = {}
    total = 0
    for y in range(256):
        for x in range(256):
            px, py = 256 * x + 128, 256 * y + 128
            inside = True
            for i in range(3):
                a, b = i, (i + 1) % 3
                dx, dy = X[b] - X[a], Y[b] - Y[a]
                e = dx * (py - Y[a]) - dy * (px - X[a])
                tl = dy < 0 or (dy == 0 and dx > 0)
                if e < 0 or (e == 0 and not tl):
                    inside = False
                    break
            if inside:
                total += 1
                rows.setdefault(y, []).append(x)
    return {"covered_pixels": total, "first_row": min(rows), "last_row": max(rows),
            "row65": rows[65], "row191_range": [rows[191][0], rows[191][-1]],
            "row_counts": {str(y): len(v) for y, v in rows.items() if y in (65, 66, 100, 128, 190, 191)}}


def k3():
    return {"fallback_shininess": 2048.0 + (2.0 - 2048.0) * 0.5, "ambient_per_channel": 0.03 * 0.7,
            "fallback_light_dir": [1 / math.sqrt(3)] * 3, "aligned_NLV_color": 0.03 * 0.7 + 0.7 + 1.0,
            "attenuation": [[0.0, 10.0, 1.0], [10.0, 10.0, 0.0], [1.0, 10.0, 0.5 * 0.81]],
            "roughness_to_shininess": [[0.0, 2048.0], [1.0, 2.0], [0.5, 1025.0], [-1.0, 2048.0], [2.0, 2.0]],
            "spot_default_radius": 50.0}


def k4():
    fovy, aspect, n, f = math.radians(45.0), 16.0 / 9.0, 0.1, 1000.0
    h = math.cos(fovy / 2) / math.sin(fovy / 2)
    return {"perspective": {"fovy_deg": 45.0, "aspect": aspect, "near": n, "far": f, "h": h, "w": h / aspect,
                            "r": f / (n - f), "m32": (f / (n - f)) * n, "m11_after_flip": -h},
            "default_eye": [0.0, 0.0, 5.0], "view_times_origin": [0.0, 0.0, -5.0]}


def k5():
    path = os.path.join(REF, "assets", "models", "a_contortionist_dancer", "scene.gltf")
    g = json.load(open(path))
    prim = g["meshes"][0]["primitives"][0]
    acc = g["accessors"]
    pos = acc[prim["attributes"]["POSITION"]]
    idx = acc[prim["indices"]]
    return {"asset": "assets/models/a_contortionist_dancer/scene.gltf", "meshes": len(g["meshes"]),
            "primitives": len(g["meshes"][0]["primitives"]), "vertices": pos["count"], "indices": idx["count"],
            "triangles": idx["count"] // 3, "position_min": pos["min"], "position_max": pos["max"],
            "index_component_type": idx["componentType"], "attributes": sorted(prim["attributes"].keys()),
            "bin_bytes": os.path.getsize(os.path.join(os.path.dirname(path), "scene.bin"))}


def k6():
    return {"TriangleVertex": {"size": 24, "position": 0, "color": 12},
            "Vertex": {"size": 48, "position": 0, "normal": 12, "tex_coord": 24, "tangent": 32},
            "CameraUbo": {"size": 208, "view": 0, "projection": 64, "view_projection": 128, "camera_position": 192},
            "ObjectUbo": {"size": 128, "model": 0, "normal_matrix": 64},
            "DirectionalLightUbo": {"size": 32}, "PointLight": {"size": 32}, "SpotLight_hlsl": {"size": 48},
            "LightUBO_hlsl": {"size": 48}, "MaterialData_hlsl": {"size": 32},
            "pipeline_defaults": {"topology": "TriangleList", "cull": "Back", "front_face": "CounterClockwise",
                                  "depth_test": True, "depth_write": True, "depth_compare": "Less", "samples": 1},
            "color_attachment_default": {"load": "CLEAR", "store": "STORE", "clear": [0, 0, 0, 1]},
            "depth_attachment_default": {"load": "CLEAR", "store": "DONT_CARE", "clear_depth": 1.0},
            "hello_triangle_clear": [0.1, 0.1, 0.15, 1.0], "MAX_FRAMES_IN_FLIGHT": 2}


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    kats = {"K1_screenshot": k1(), "K2_hello_coverage_256": k2(), "K3_shader_constants": k3(), "K4_matrices": k4(),
            "K5_asset": k5(), "K6_layouts": k6()}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(kats, open(OUT, "w"), indent=1)
    print(json.dumps(kats["K1_screenshot"], indent=1))
    print(json.dumps(kats["K2_hello_coverage_256"]))
    print(json.dumps(kats["K5_asset"]))
