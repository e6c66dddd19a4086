"""TEST INFRASTRUCTURE (child process of tests/test_gpu_api.py::test_band_exchange_with_n_ranks_on_one_gpu): N mirhi devices of one
process on one GPU play the N ranks of a tile-row split; the band exchange of the C ABI (mirhi_comm_all_gather_bands) runs against
tests/mock/mock_rccl.cpp, loaded through MIRHI_RCCL_LIBRARY.  Every rank's gathered frame must be the unsplit frame, byte for byte,
for even and uneven bands, both exchange algorithms and more ranks than tile rows.  usage: split_exchange_check.py <libmock_rccl.so>"""
import ctypes
import os
import sys

import numpy as np

mock_path = os.path.abspath(sys.argv[1])
os.environ["MIRHI_RCCL_LIBRARY"] = mock_path
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

m = ge.load_package()
mock = ctypes.CDLL(mock_path)
for fn in ("mock_rccl_pending", "mock_rccl_matched", "mock_rccl_bytes"):
    getattr(mock, fn).restype = ctypes.c_size_t

CASES = [  # width, height, ranks, algorithm
    (320, 200, 2, "DIRECT"), (320, 200, 3, "DIRECT"), (333, 217, 4, "DIRECT"), (640, 360, 8, "DIRECT"),      # 360 rows = 12 tile rows: uneven bands of 2 / 1
    (320, 200, 3, "BROADCAST"), (640, 360, 8, "BROADCAST"),
    (256, 96, 5, "DIRECT"),                                                                                     # 3 tile rows for 5 ranks: two ranks own nothing
]
failures = 0
# every case with one contiguous band per rank and with interleaved tile rows (rank r: rows r, r + N, ...: one piece per owned row in the exchange)
for (W, H, N, algo_name, layout) in [c + (lay,) for lay in ("bands", "interleaved") for c in CASES]:
    scene = m.scenes.random_triangles(500, W, H, seed=1000 + N, rmin=3, rmax=40)
    solo = m.Device(0)
    res0 = m.SceneResources(solo, scene, m.Format.B8G8R8A8_SRGB)
    res0.render()
    ref = res0.read()["color"].copy()
    res0.destroy(); solo.destroy()
    devs = [m.Device(0) for _ in range(N)]
    for d in devs:
        d.set_split_layout(layout)
    uid = m.Comm.unique_id()
    comms = [m.Comm(devs[r], uid, r, N) for r in range(N)]          # (sets the device's tile split: the frames are recorded behind it)
    assert all(c.world() == N for c in comms)
    from renderer_rs_amd import multigpu
    assert all(devs[r].split_rows(H) == multigpu.split_tile_rows(H, r, N, layout) for r in range(N))
    ress = [m.SceneResources(devs[r], scene, m.Format.B8G8R8A8_SRGB) for r in range(N)]
    algo = getattr(m.GatherAlgo, algo_name)
    moved0 = mock.mock_rccl_bytes()
    for frame in range(2):                                           # twice: the second exchange reuses events and streams
        for r in range(N):
            ress[r].render()
        for r in range(N):
            comms[r].all_gather_bands(ress[r].color, ress[r].cmd, algo)
        assert mock.mock_rccl_pending() == 0, f"{mock.mock_rccl_pending()} operations never met their peer"
    for r in range(N):
        devs[r].wait_idle()
    moved = mock.mock_rccl_bytes() - moved0
    for c in comms:
        c.destroy()                                                  # (synchronises the exchange stream)
    ok = True
    for r in range(N):
        got = ress[r].read()["color"]
        if not np.array_equal(got, ref):
            bad = np.argwhere((got != ref).any(axis=2))
            print(f"MISMATCH {W}x{H} N={N} {algo_name} {layout}: rank {r} differs in {len(bad)} pixels, first at row {bad[0][0]}")
            ok = False
    want = 2 * (N - 1) * W * H * 4
    if moved != want:
        print(f"BYTES {W}x{H} N={N} {algo_name} {layout}: moved {moved}, expected {want}")
        ok = False
    failures += 0 if ok else 1
    for r in range(N):
        ress[r].destroy(); devs[r].destroy()
    print(f"{W}x{H} ranks {N} {algo_name} {layout}: {'ok' if ok else 'FAILED'} ({moved} bytes exchanged)", flush=True)
# A failing ncclSend inside the grouped exchange (ADVICE r2 / VERDICT r2 item 6b): the call reports it, the thread's RCCL group is closed
# again (an open group would swallow every later RCCL call of the thread), and the next exchange works.
mock.mock_rccl_group_depth.restype = ctypes.c_int
scene = m.scenes.random_triangles(300, 320, 200, seed=77, rmin=3, rmax=40)
solo = m.Device(0)
res0 = m.SceneResources(solo, scene, m.Format.B8G8R8A8_SRGB); res0.render(); ref = res0.read()["color"].copy(); res0.destroy(); solo.destroy()
N = 3
devs = [m.Device(0) for _ in range(N)]
for d in devs:
    d.set_split_layout("bands")        # (the group handling is what is tested here; see mock_rccl.cpp on the packed exchange under this stand-in)
uid = m.Comm.unique_id()
comms = [m.Comm(devs[r], uid, r, N) for r in range(N)]
ress = [m.SceneResources(devs[r], scene, m.Format.B8G8R8A8_SRGB) for r in range(N)]
for r in range(N):
    ress[r].render()
mock.mock_rccl_fail_send_in(2)                       # rank 0's second send
raised = False
try:
    comms[0].all_gather_bands(ress[0].color, ress[0].cmd, m.GatherAlgo.DIRECT)
except m.RhiError as e:
    raised = "RCCL" in str(e)
ok = raised and mock.mock_rccl_group_depth() == 0
mock.mock_rccl_reset()
for r in range(N):
    devs[r].wait_idle()
for r in range(N):
    ress[r].render()
for r in range(N):
    comms[r].all_gather_bands(ress[r].color, ress[r].cmd, m.GatherAlgo.DIRECT)
for r in range(N):
    devs[r].wait_idle()
ok = ok and mock.mock_rccl_pending() == 0 and all(np.array_equal(ress[r].read()["color"], ref) for r in range(N))
for c in comms:
    c.destroy()
for r in range(N):
    ress[r].destroy(); devs[r].destroy()
print(f"failing send inside the group: {'ok' if ok else 'FAILED'} (error reported: {raised}, group depth afterwards {mock.mock_rccl_group_depth()})", flush=True)
failures += 0 if ok else 1
sys.exit(1 if failures else 0)
