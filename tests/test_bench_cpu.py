"""bench.py's host logic that needs no GPU: the timeline summary (overlap / gap / period arithmetic), the traffic figure's build-hash
gate, the launcher command for `--gpus N`, and the refusal to report a line whose n_gpus would not be what was asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ("geometry", "raster", "vertex", "fragment_count")


def test_timeline_summary_serial_lane():
    # two frames on one lane, strictly serial: geometry 10 us, gap 2, raster 12 us, gap 1
    tl = [(0, 0, 0.0, 10.0), (1, 0, 12.0, 24.0), (0, 0, 25.0, 35.0), (1, 0, 37.0, 49.0), (0, 0, 50.0, 60.0), (1, 0, 62.0, 74.0)]
    s = bench.summarize_timeline(tl, NAMES)
    assert s["dispatches"] == 6 and s["geometry_us"] == 10.0 and s["raster_us"] == 12.0
    assert s["raster_overlap_us"] == 0.0 and s["geometry_to_raster_gap_us"] == 2.0 and s["frame_period_us"] == 25.0


def test_timeline_summary_overlapping_lanes():
    # two lanes, raster kernels of consecutive frames overlap by 3 us; each raster pairs with the geometry of ITS lane
    tl = [(0, 0, 0.0, 5.0), (0, 1, 1.0, 6.0), (1, 0, 6.0, 16.0), (1, 1, 13.0, 23.0), (0, 0, 17.0, 22.0), (1, 0, 24.0, 34.0)]
    s = bench.summarize_timeline(tl, NAMES)
    assert s["raster_us"] == 10.0
    assert s["raster_overlap_us"] == pytest.approx((3.0 + 0.0) / 2, abs=1e-3)
    assert s["geometry_to_raster_gap_us"] == pytest.approx((1.0 + 7.0 + 2.0) / 3, abs=1e-3)
    assert bench.summarize_timeline([], NAMES) is None


def test_traffic_is_quoted_only_for_the_same_kernel_build(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r09_c2_hbm_traffic.json").write_text(json.dumps({"kernel_source_sha16": "aaaa", "frame_hbm_bytes": 123}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    hit, note = bench.measured_traffic("c2", "aaaa")
    assert hit["frame_hbm_bytes"] == 123 and note == "r09_c2_hbm_traffic.json"
    miss, note = bench.measured_traffic("c2", "bbbb")
    assert miss is None and "not quoted" in note
    none, note = bench.measured_traffic("c5", "aaaa")
    assert none is None and "no counters" in note


def test_gpus_flag_starts_the_ranks_itself(monkeypatch):
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    args = bench.parse_args()
    assert bench.spawn_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing to report" in (r.stderr + r.stdout)
