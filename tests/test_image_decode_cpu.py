"""PNG / JPEG decode (renderer-rs_amd/host/image_decode.hpp through include/miresources.h) -- SURVEY.md 8f rank 1.

No GPU.  Expected pixels come from two independent sources:
  * PNG: files are ENCODED here by a small writer (zlib + struct) that exercises every colour type, bit depth, all five
    scanline filters and Adam7, with the expected RGBA computed in numpy straight from the PNG specification's sample
    scaling rules (16 -> 8 bit by the `image` crate's (c + 128) / 257); Pillow is a second opinion where it has the mode.
  * JPEG: files are encoded by Pillow (libjpeg-turbo), sequential and progressive, with different sampling, quality, restart and Huffman options and
    decoded by Pillow again; the decoder here restates the same integer IDCT / colour conversion / triangle upsampling,
    so the comparison is bit for bit, not a tolerance.
"""
import io
import os
import re
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as _ge  # noqa: E402

_ge.load_package()
from renderer_rs_amd import images  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _built():
    images.build()


def _chunk(tag: bytes, body: bytes) -> bytes:
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, rng):
    """rows: list of bytes (one per scanline) -> filtered stream with a random filter type per row."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for r in rows:
        f = int(rng.integers(0, 5))
        out.append(f)
        for i, x in enumerate(r):
            a = r[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            out.append((x - pred) & 0xFF)
        prev = r
    return bytes(out)


def _pack_row(samples, depth):
    """samples: 1-D array of one scanline's samples (all channels) -> packed bytes."""
    if depth == 8:
        return bytes(samples.astype(np.uint8))
    if depth == 16:
        return samples.astype(">u2").tobytes()
    bits = np.zeros(((samples.size * depth + 7) // 8) * 8, dtype=np.uint8)
    for k in range(depth):
        bits[k:samples.size * depth:depth] = (samples >> (depth - 1 - k)) & 1
    return bytes(np.packbits(bits))


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def encode_png(samples, color, depth, rng, interlace=False, plte=None, trns=None, idat_split=3):
    """samples: (h, w, channels) integer array of raw sample values."""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    stream = b""
    passes = ADAM7 if interlace else [(0, 0, 1, 1)]
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = [_pack_row(sub[y].reshape(-1), depth) for y in range(sub.shape[0])]
        stream += _filter_rows(rows, bpp, rng)
    z = zlib.compress(stream, 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    out += _chunk(b"gAMA", struct.pack(">I", 45455))                   # ancillary chunk: must be skipped
    if plte is not None:
        out += _chunk(b"PLTE", bytes(np.asarray(plte, dtype=np.uint8).reshape(-1)))
    if trns is not None:
        out += _chunk(b"tRNS", trns)
    step = max(1, len(z) // idat_split)
    for i in range(0, len(z), step):
        out += _chunk(b"IDAT", z[i:i + step])
    return out + _chunk(b"IEND", b"")


def _narrow(v, depth):
    v = v.astype(np.int64)
    if depth == 16:
        return ((v + 128) // 257).astype(np.uint8)
    return (v * 255 // ((1 << depth) - 1)).astype(np.uint8)


PNG_CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("color,depth", PNG_CASES)
@pytest.mark.parametrize("size", [(1, 1), (5, 3), (37, 23)])
def test_png_every_colour_type_depth_filter_and_interlace(color, depth, size, interlace):
    rng = np.random.default_rng(hash((color, depth, size, interlace)) & 0xFFFFFFFF)
    w, h = size
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    samples = rng.integers(0, 1 << depth, size=(h, w, ch), dtype=np.int64)
    if w > 8:
        samples[:, : w // 2] = samples[:, :1]          # flat runs so the filters and LZ77 matches see repeats
    plte = trns = None
    exp = np.zeros((h, w, 4), dtype=np.uint8)
    with_trns = (w + h) % 2 == 0
    if color == 3:
        plte = rng.integers(0, 256, size=(1 << depth, 3), dtype=np.int64)
        alpha = np.full(1 << depth, 255, dtype=np.int64)
        if with_trns:
            k = max(1, (1 << depth) // 2)
            alpha[:k] = rng.integers(0, 256, size=k)
            trns = bytes(alpha[:k].astype(np.uint8))
        exp[..., :3] = plte[samples[..., 0]]
        exp[..., 3] = alpha[samples[..., 0]]
    elif color == 0:
        exp[..., :3] = _narrow(samples[..., :1], depth)
        exp[..., 3] = 255
        if with_trns:
            key = int(samples[h // 2, w // 2, 0])
            trns = struct.pack(">H", key)
            exp[..., 3] = np.where(samples[..., 0] == key, 0, 255)
    elif color == 2:
        exp[..., :3] = _narrow(samples, depth)
        exp[..., 3] = 255
        if with_trns:
            key = samples[h // 2, w // 2]
            trns = struct.pack(">HHH", *[int(v) for v in key])
            exp[..., 3] = np.where((samples == key).all(axis=-1), 0, 255)
    elif color == 4:
        exp[..., :3] = _narrow(samples[..., :1], depth)
        exp[..., 3] = _narrow(samples[..., 1], depth)
    else:
        exp[...] = _narrow(samples, depth)
    data = encode_png(samples, color, depth, rng, interlace=interlace, plte=plte, trns=trns)
    got = images.decode_image(data)
    assert got.rgba.shape == (h, w, 4)
    assert np.array_equal(got.rgba, exp)
    assert got.source_channels == ({0: 1, 2: 3, 3: 3, 4: 2, 6: 4}[color] + (1 if trns is not None and color in (0, 2, 3) else 0))
    if depth == 8 and color in (0, 2, 3, 4, 6):          # second opinion where Pillow's conversion rule is the same
        from PIL import Image
        ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
        assert np.array_equal(got.rgba, ref)


def test_png_stored_and_fixed_huffman_blocks():
    rng = np.random.default_rng(5)
    samples = rng.integers(0, 256, size=(9, 11, 3), dtype=np.int64)
    rows = b"".join(b"\x00" + bytes(samples[y].reshape(-1).astype(np.uint8)) for y in range(9))
    for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (9, zlib.Z_RLE), (1, zlib.Z_HUFFMAN_ONLY)):
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
        z = co.compress(rows) + co.flush()
        data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 11, 9, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", z) + _chunk(b"IEND", b"")
        got = images.decode_image(data)
        assert np.array_equal(got.rgba[..., :3], samples.astype(np.uint8)) and (got.rgba[..., 3] == 255).all()


def test_png_large_window_matches():
    """distances up to 32 KiB and long-code Huffman symbols: a 300x300 image with period-structured noise"""
    rng = np.random.default_rng(6)
    base = rng.integers(0, 256, size=(30, 300, 4), dtype=np.int64)
    samples = np.concatenate([base] * 10, axis=0)
    samples[::7, ::5] = rng.integers(0, 256, size=samples[::7, ::5].shape)
    data = encode_png(samples, 6, 8, rng, idat_split=7)
    assert np.array_equal(images.decode_image(data).rgba, samples.astype(np.uint8))


def _png_rgb(w=4, h=4):
    rng = np.random.default_rng(9)
    return encode_png(rng.integers(0, 256, size=(h, w, 3), dtype=np.int64), 2, 8, rng)


def test_png_errors_are_reported_not_decoded():
    good = _png_rgb()
    with pytest.raises(images.ImageDecodeError, match="CRC mismatch"):
        bad = bytearray(good)
        bad[8 + 8 + 2] ^= 1                                # inside IHDR body
        images.decode_image(bytes(bad))
    with pytest.raises(images.ImageDecodeError):
        images.decode_image(good[: len(good) // 2])
    with pytest.raises(images.ImageDecodeError, match="Unsupported image format"):
        images.decode_image(b"GIF89a" + bytes(32))
    # Adler-32 of the zlib stream
    i = good.index(b"IDAT")
    n = struct.unpack(">I", good[i - 4:i])[0]
    body = bytearray(good[i + 4:i + 4 + n])
    if good.count(b"IDAT") == 1:
        body[-1] ^= 0xFF
        bad = good[:i - 4] + _chunk(b"IDAT", bytes(body)) + good[i + 8 + n:]
        with pytest.raises(images.ImageDecodeError, match="Adler"):
            images.decode_image(bad)
    # bit depth / colour type combination that the specification forbids
    bad = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 4, 2, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(bytes(20))) + _chunk(b"IEND", b"")
    with pytest.raises(images.ImageDecodeError, match="bit depth"):
        images.decode_image(bad)
    # unknown critical chunk
    bad = good[:33] + _chunk(b"XyZW", b"abc") + good[33:]
    with pytest.raises(images.ImageDecodeError, match="critical"):
        images.decode_image(bad)
    with pytest.raises(images.ImageDecodeError) as e:
        images.load_image("/nonexistent/texture.png")
    assert e.value.code == 2


def test_png_header_cannot_force_huge_allocations():
    """a ~70-byte PNG announcing 16384 x 16384 x RGBA16 must fail on its (short) data without first reserving
    the 2 GiB its header asks for; surplus data behind a complete image is ignored, as libpng and the `png` crate do."""
    import resource
    ihdr = struct.pack(">IIBBBBB", 16384, 16384, 16, 6, 0, 0, 0)
    bomb = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(bytes(64))) + _chunk(b"IEND", b"")
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(images.ImageDecodeError, match="too short"):
        images.decode_image(bomb)
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 64 * 1024        # KiB
    rng = np.random.default_rng(3)
    px = rng.integers(0, 256, size=(3, 5, 3), dtype=np.int64)
    rows = b"".join(b"\x00" + bytes(px[y].reshape(-1).tolist()) for y in range(3))
    surplus = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 5, 3, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(rows + bytes(40))) + _chunk(b"IEND", b"")
    img = images.decode_image(surplus)
    assert np.array_equal(img.rgba[..., :3], px.astype(np.uint8))


def _jpeg(arr, mode, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, "JPEG", **kw)
    data = buf.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
    return data, ref


def _photo(h, w, rng, channels=3):
    """smooth gradients + edges + noise: exercises DC prediction, long AC runs, EOB and ZRL symbols"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 100 * np.sin(x / (7.0 + 3 * c) + c) * np.cos(y / (5.0 + 2 * c)) for c in range(channels)], axis=-1)
    img[h // 3: h // 2, w // 4: w // 2] = 250 - img[h // 3: h // 2, w // 4: w // 2]
    img += rng.normal(0, 12, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (8, 8), (16, 16), (17, 9), (33, 47), (130, 70)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("quality", [35, 90, 100])
def test_jpeg_ycbcr_matches_libjpeg_bit_for_bit(size, subsampling, quality):
    rng = np.random.default_rng(hash((size, subsampling, quality)) & 0xFFFFFFFF)
    w, h = size
    data, ref = _jpeg(_photo(h, w, rng), "RGB", quality=quality, subsampling=subsampling, optimize=(quality == 90))
    got = images.decode_image(data)
    assert got.source_channels == 3 and got.rgba.shape == ref.shape
    assert np.array_equal(got.rgba, ref), f"max diff {np.abs(got.rgba.astype(int) - ref.astype(int)).max()}"


@pytest.mark.parametrize("size", [(1, 1), (9, 8), (64, 31)])
def test_jpeg_greyscale(size):
    rng = np.random.default_rng(size[0])
    data, ref = _jpeg(_photo(size[1], size[0], rng, 1)[..., 0], "L", quality=80)
    got = images.decode_image(data)
    assert got.source_channels == 1 and np.array_equal(got.rgba, ref)


@pytest.mark.parametrize("subsampling", [0, 2])
@pytest.mark.parametrize("blocks", [1, 3, 8])
def test_jpeg_restart_intervals(subsampling, blocks):
    rng = np.random.default_rng(blocks)
    data, ref = _jpeg(_photo(75, 100, rng), "RGB", quality=85, subsampling=subsampling, restart_marker_blocks=blocks)
    assert b"\xff\xdd" in data                          # DRI present
    assert np.array_equal(images.decode_image(data).rgba, ref)


@pytest.mark.parametrize("size", [(1, 1), (8, 8), (17, 9), (33, 47), (130, 70)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("quality", [30, 92])
def test_jpeg_progressive_matches_libjpeg_bit_for_bit(size, subsampling, quality):
    """SOF2: DC first / refinement scans, AC band first passes with end-of-band runs, AC refinement with correction bits
    (libjpeg's default progression script: 10 scans for a colour image)"""
    rng = np.random.default_rng(hash((size, subsampling, quality, 1)) & 0xFFFFFFFF)
    w, h = size
    data, ref = _jpeg(_photo(h, w, rng), "RGB", quality=quality, subsampling=subsampling, progressive=True)
    assert b"\xff\xc2" in data and data.count(b"\xff\xda") > 5
    got = images.decode_image(data)
    assert np.array_equal(got.rgba, ref), f"max diff {np.abs(got.rgba.astype(int) - ref.astype(int)).max()}"


def test_jpeg_progressive_greyscale_and_restarts():
    rng = np.random.default_rng(8)
    data, ref = _jpeg(_photo(45, 61, rng, 1)[..., 0], "L", quality=75, progressive=True)
    assert np.array_equal(images.decode_image(data).rgba, ref)
    data, ref = _jpeg(_photo(75, 100, rng), "RGB", quality=85, subsampling=2, progressive=True, restart_marker_blocks=2)
    assert b"\xff\xdd" in data
    assert np.array_equal(images.decode_image(data).rgba, ref)
    data, ref = _jpeg(_photo(64, 64, rng), "RGB", quality=100, subsampling=0, progressive=True, optimize=True)
    assert np.array_equal(images.decode_image(data).rgba, ref)


def test_jpeg_custom_quantisation_tables():
    rng = np.random.default_rng(3)
    qt = [[min(255, 1 + 3 * i) for i in range(64)], [min(255, 2 + 5 * i) for i in range(64)]]
    data, ref = _jpeg(_photo(40, 56, rng), "RGB", qtables=qt, subsampling=0)
    assert np.array_equal(images.decode_image(data).rgba, ref)


def test_jpeg_unsupported_variants_are_refused():
    rng = np.random.default_rng(4)
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(_photo(16, 16, rng, 4), "CMYK").save(buf, "JPEG")
    with pytest.raises(images.ImageDecodeError, match="4-component"):
        images.decode_image(buf.getvalue())
    good, _ = _jpeg(_photo(32, 32, rng), "RGB")
    with pytest.raises(images.ImageDecodeError):
        images.decode_image(good[:200])
    # truncated entropy data decodes as far as the bits go (zero-fed, T.81 F.2.2.5) or fails -- never crashes
    try:
        images.decode_image(good[: len(good) - 40])
    except images.ImageDecodeError:
        pass


def test_fuzzed_files_never_crash():
    """bit flips anywhere in a PNG / JPEG: either an ImageDecodeError or some image, never a fault"""
    rng = np.random.default_rng(11)
    png = _png_rgb(19, 13)
    jpg, _ = _jpeg(_photo(24, 40, rng), "RGB", subsampling=2)
    pjpg, _ = _jpeg(_photo(24, 40, rng), "RGB", subsampling=1, progressive=True)
    for base in (png, jpg, pjpg):
        for _ in range(300):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            try:
                images.decode_image(bytes(b))
            except images.ImageDecodeError:
                pass


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "miresources.h")).read()
    names = set(re.findall(r"\b(mires_[a-z_]+)\s*\(", header))
    assert names == {"mires_image_decode", "mires_image_load", "mires_image_free", "mires_last_error_message"}
    lib = images.lib()
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.skipif(not os.path.isdir("/root/reference/assets/textures"), reason="reference assets are not on this machine")
def test_reference_asset_textures_decode_like_pillow():
    """every texture file the reference ships (assets/textures/*, the dancer's normal map): PNG and JPEG, bit for bit"""
    import glob
    from PIL import Image
    files = sorted(glob.glob("/root/reference/assets/textures/*/*.*") + glob.glob("/root/reference/assets/models/*/textures/*.*"))
    assert len(files) >= 22                                           # includes the 4096x4096 normal map
    for f in files:
        got = images.load_image(f)
        ref = np.asarray(Image.open(f).convert("RGBA"))
        assert np.array_equal(got.rgba, ref), f
