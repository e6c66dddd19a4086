"""the glTF loader with the reference's semantics (crates/resources/src/model.rs:111-270), pinned by
K5 -- the asset the reference's own integration test loads (crates/resources/tests/integration_test.rs:7-83), committed as
a data fixture under tests/golden/dancer (CC-BY-4.0, credit in license.txt)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
K5 = json.load(open(os.path.join(HERE, "golden", "kats.json")))["K5_asset"]
DANCER = os.path.join(HERE, "golden", "dancer", "scene.gltf")


def test_k5_dancer_counts_and_bounds(mirhi):
    from renderer_rs_amd import gltf
    m = gltf.load(DANCER)
    assert len(m.meshes) == K5["primitives"] == 1
    mesh = m.meshes[0]
    assert mesh.vertex_count == K5["vertices"] == 11865
    assert mesh.indices.size == K5["indices"] == 51630 and mesh.triangle_count == K5["triangles"] == 17210
    assert np.allclose(m.aabb_min, K5["position_min"], atol=1e-6) and np.allclose(m.aabb_max, K5["position_max"], atol=1e-6)
    assert mesh.indices.max() < mesh.vertex_count
    assert mesh.normals.shape == (11865, 3) and mesh.tex_coords.shape == (11865, 2) and mesh.tangents.shape == (11865, 4)
    assert np.allclose(np.linalg.norm(mesh.normals, axis=1), 1.0, atol=1e-3)
    assert set(np.unique(np.round(mesh.tangents[:, 3]))) <= {-1.0, 1.0}
    assert m.materials[0].metallic == pytest.approx(0.0594287, abs=1e-6) and mesh.material_index == 0
    v = mesh.interleave()
    assert v.shape == (11865, 12) and v.dtype == np.float32 and v.nbytes == 11865 * 48
    assert os.path.getsize(os.path.join(os.path.dirname(DANCER), "scene.bin")) == K5["bin_bytes"]


def test_loader_defaults_and_errors(mirhi, tmp_path):
    """Missing attributes take the reference's defaults; missing file / mesh-less file raise (model.rs:113-115,262-264)."""
    from renderer_rs_amd import gltf
    import base64
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    uri = "data:application/octet-stream;base64," + base64.b64encode(pos.tobytes()).decode()
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": 36, "uri": uri}],
           "bufferViews": [{"buffer": 0, "byteLength": 36}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}]}
    p = tmp_path / "tri.gltf"
    p.write_text(json.dumps(doc))
    m = gltf.load(str(p))
    mesh = m.meshes[0]
    assert np.array_equal(mesh.indices, [0, 1, 2])                     # sequential indices
    assert np.array_equal(mesh.normals, np.tile([0, 1, 0], (3, 1)))    # +Y
    assert np.array_equal(mesh.tangents, np.tile([1, 0, 0, 1], (3, 1))) and not mesh.tex_coords.any()
    with pytest.raises(gltf.ResourceError):
        gltf.load(str(tmp_path / "missing.gltf"))
    doc["meshes"] = []
    p.write_text(json.dumps(doc))
    with pytest.raises(gltf.ResourceError):
        gltf.load(str(p))


def test_dancer_scene_renders_with_oracle(oracle, scenes):
    s = scenes.gltf_model(DANCER, 320, 180)
    assert s.num_triangles == 17210
    r = oracle.render(s, want_bgra8=False)
    cov = r["prim"] != 0xFFFFFFFF
    assert 0.03 < cov.mean() < 0.6 and not np.isnan(r["rgba"]).any()
