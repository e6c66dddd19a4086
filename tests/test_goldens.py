"""Committed golden vectors (tests/golden/small_cases.json, made by tools/make_goldens.py): the oracle must still
reproduce them (CPU), and the HIP path must hit the same integer digests (GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "small_cases.json")))
DANCER = os.path.join(HERE, "golden", "dancer", "scene.gltf")


def _scene(scenes, key):
    if key == "dancer_320x180":
        return scenes.gltf_model(DANCER, 320, 180)
    return scenes.SMALL_CASES[key]()


def _digest(prim, depth):
    covered = prim != 0xFFFFFFFF
    depth_bits = np.where(covered, depth.view(np.uint32), 0).astype(np.uint32)
    return int(covered.sum()), hashlib.sha256(prim.tobytes()).hexdigest(), hashlib.sha256(depth_bits.tobytes()).hexdigest()


@pytest.mark.parametrize("key", sorted(GOLD))
def test_oracle_reproduces_goldens(oracle, scenes, key):
    g = GOLD[key]
    r = oracle.render(_scene(scenes, key), want_bgra8=True)
    cov, ph, dh = _digest(r["prim"], r["depth"])
    assert (cov, ph, dh) == (g["covered"], g["prim_sha256"], g["depth_sha256"])
    assert [int(x) for x in r["bgra8"].reshape(-1, 4).astype(np.uint64).sum(axis=0)] == g["bgra8_sum"]


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(GOLD))
def test_gpu_matches_goldens(mirhi, device, scenes, key):
    g = GOLD[key]
    scene = _scene(scenes, key)
    depth_tested = any(d.depth_test for d in scene.draws)
    res = mirhi.SceneResources(device, scene, mirhi.Format.B8G8R8A8_SRGB, want_prim=True, want_depth=True)
    res.render()
    out = res.read()
    res.destroy()
    cov, ph, dh = _digest(out["prim"], out["depth"])
    assert cov == g["covered"] and ph == g["prim_sha256"]
    if depth_tested:
        assert dh == g["depth_sha256"]
    # sRGB8 is within 1 LSB per pixel of the oracle's encoding, so the channel sums differ by at most the pixel count
    sums = [int(x) for x in out["color"].reshape(-1, 4).astype(np.uint64).sum(axis=0)]
    assert all(abs(a - b) <= scene.width * scene.height for a, b in zip(sums, g["bgra8_sum"]))
