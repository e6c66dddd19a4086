"""Builds and runs the C++ host mirror tests (renderer-rs_amd/host/test_host.cpp): the reference's own unit-test
assertions restated against mirhi.hpp, and -- on the GPU -- Renderer::render_frame / FrameManager over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "renderer-rs_amd", "host")


def _build(mirhi):
    mirhi.lib()   # makes sure libmirhi.so exists
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return os.path.join(HOST, "test_host")


def test_host_mirror_cpu(mirhi):
    exe = _build(mirhi)
    env = dict(os.environ, MIRHI_TEST_GLTF=os.path.join(ROOT, "tests", "golden", "dancer", "scene.gltf"))   # Model::load, K5 counts
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0 and "host tests: ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_host_mirror_renderer_gpu(mirhi):
    exe = _build(mirhi)
    out = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host tests: ok" in out.stdout, out.stdout + out.stderr
