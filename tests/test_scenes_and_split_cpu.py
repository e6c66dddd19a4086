"""Host-side logic on CPU: scene generators (BASELINE configs), algorithmic byte counts and the
tile-row split + all-gather path with world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_scenes_match_survey_numbers(scenes):
    c2 = scenes.random_triangles()
    assert c2.num_triangles == 10000 and (c2.width, c2.height) == (1920, 1080)
    assert c2.algorithmic_bytes(4) == 720000 + 1920 * 1080 * 4 == 9014400           # 9.01 MB
    c3 = scenes.displaced_sphere()
    assert c3.num_triangles == 70312 and c3.draws[0].vertices.shape[0] == 35532
    assert abs(c3.algorithmic_bytes(4) - 10.84e6) < 0.02e6
    # same seed -> same bytes
    again = scenes.random_triangles()
    assert np.array_equal(c2.draws[0].vertices, again.draws[0].vertices)
    flat = c2.draws[0].vertices.reshape(10000, 3, 6)
    assert np.array_equal(flat[:, 0, 3:], flat[:, 1, 3:]) and np.array_equal(flat[:, 0, 3:], flat[:, 2, 3:])   # flat shade
    assert np.array_equal(flat[:, 0, 2], flat[:, 2, 2]) and flat[:, :, 2].min() >= 0.05 and flat[:, :, 2].max() <= 0.95


def test_big_config_scenes_counts(scenes):
    c4 = scenes.heightfield_grid(100, 50, 640, 360)      # same generator, small instance
    assert c4.num_triangles == 100 * 50 * 2 and c4.draws[0].vertices.shape[0] == 101 * 51
    c5 = scenes.box_hall(8, 320, 180, tex_size=64)
    assert c5.num_triangles == 8 * 512 and len(c5.draws) == 4
    assert all(d.albedo_map.rgba8.shape == (64, 64, 4) for d in c5.draws)


def test_pcg32_reference_vector(scenes):
    """PCG32 XSH-RR: first outputs for seed 42, sequence 54 (pcg-random.org demo vector)."""
    rng = scenes.PCG32(42, 54)
    got = [rng.next_u32() for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_band_rows_partition(mirhi):
    from renderer_rs_amd import multigpu
    for height in (1080, 2160, 64, 33):
        for world in (1, 2, 4, 8):
            bands = [multigpu.band_rows(height, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == height or any(b[1] == height for b in bands)
            covered = sum(e - b for b, e in bands)
            assert covered == height
            for (b0, e0), (b1, e1) in zip(bands, bands[1:]):
                assert e0 == b1 or (b1 == e1 == height)
            assert multigpu.padded_rows(height, world) >= height and multigpu.padded_rows(height, world) % world == 0


def test_split_tile_rows_partition_both_layouts(mirhi):
    """every tile row of a frame belongs to exactly one rank, with one band per rank and with interleaved rows; the pixel-row runs a rank sends
    (multigpu.owned_pixel_rows = the pieces of mirhi_comm_all_gather_bands) cover the frame exactly once"""
    from renderer_rs_amd import multigpu
    for layout in ("bands", "interleaved"):
        for height in (1080, 2160, 64, 33, 97):
            for world in (1, 2, 3, 4, 8):
                owner = np.full(multigpu.tiles_y(height), -1)
                rows = np.zeros(height, dtype=np.int32)
                for r in range(world):
                    first, step, count = multigpu.split_tile_rows(height, r, world, layout)
                    for k in range(count):
                        assert owner[first + k * step] == -1
                        owner[first + k * step] = r
                    for b, e in multigpu.owned_pixel_rows(height, r, world, layout):
                        rows[b:e] += 1
                assert (owner >= 0).all() and (rows == 1).all()
    assert multigpu.split_tile_rows(2160, 3, 8, "interleaved") == (3, 8, 9) and multigpu.split_tile_rows(2160, 4, 8, "interleaved") == (4, 8, 8)
    assert multigpu.owned_pixel_rows(1080, 1, 4, "interleaved")[-1] == (33 * 32, 1080)      # the frame's last tile row is 24 pixel rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _split_worker(rank, world, port, out_dir, width=200, height=150):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    m = ge.load_package()
    from renderer_rs_amd import multigpu
    import oracle_binding as ob
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = m.scenes.random_triangles(400, width, height, seed=21, rmin=3, rmax=30)
    r0, r1 = multigpu.band_rows(scene.height, rank, world)
    per = multigpu.rows_per_rank(scene.height, world)
    frame = torch.zeros((multigpu.padded_rows(scene.height, world), scene.width, 4), dtype=torch.uint8)
    # each rank renders only its own band (here with the CPU oracle standing in for the GPU band render)
    band = ob.render(scene, rows=(r0, r1))["bgra8"]
    frame[rank * per:rank * per + (r1 - r0)] = torch.from_numpy(band[r0:r1])
    multigpu.all_gather_bands(frame, rank, world)
    full = ob.render(scene)["bgra8"]
    ok = bool((frame[:scene.height].numpy() == full).all())
    # the direct pattern of mirhi_comm_all_gather_bands (MIRHI_GATHER_DIRECT): unpadded frame, bands of unequal size
    direct = torch.zeros((scene.height, scene.width, 4), dtype=torch.uint8)
    direct[r0:r1] = torch.from_numpy(band[r0:r1])
    multigpu.exchange_bands_direct(direct, scene.height, rank, world)
    ok = ok and bool((direct.numpy() == full).all())
    # ... and with interleaved tile rows: a rank's share is one piece per tile row it owns
    inter = torch.zeros((scene.height, scene.width, 4), dtype=torch.uint8)
    for b, e in multigpu.owned_pixel_rows(scene.height, rank, world, "interleaved"):
        inter[b:e] = torch.from_numpy(ob.render(scene, rows=(b, e))["bgra8"][b:e])
    multigpu.exchange_bands_direct(inter, scene.height, rank, world, layout="interleaved")
    ok = ok and bool((inter.numpy() == full).all())
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_tile_row_split_all_gather_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_split_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "rank0.txt").read() == "ok" and open(tmp_path / "rank1.txt").read() == "ok"


def test_tile_row_split_world4_uneven_last_band(tmp_path):
    """1080 rows = 34 tile rows over 4 ranks: bands of 9, 9, 9 and 7 tile rows (288, 288, 288, 216 pixel rows) -- the padded
    in-place all-gather and the direct exchange of unequal bands both reassemble the oracle's full frame on every rank."""
    import torch.multiprocessing as mp
    from renderer_rs_amd import multigpu
    assert [multigpu.band_rows(1080, r, 4) for r in range(4)] == [(0, 288), (288, 576), (576, 864), (864, 1080)]
    port = _free_port()
    mp.spawn(_split_worker, args=(4, port, str(tmp_path), 96, 1080), nprocs=4, join=True)
    assert [open(tmp_path / f"rank{r}.txt").read() for r in range(4)] == ["ok"] * 4
