"""Image decoders of the host program (src/imgio.h), checked on the CPU through mgm_amd/bin/imgconv.

The reference reads images through iio on libpng/libtiff (img_tools.h:18-34); what a file decodes to is fixed by
iio.c:1487-1547 (PNG: PACKING | EXPAND) and iio.c:1657-1880 (TIFF scanlines as stored).  Expected values come from
PIL where PIL decodes the same way, and are written out by hand where it does not (sub-byte grey scaling, tRNS as an
alpha channel) or cannot write the variant (big-endian, BigTIFF, tiles, planar, predictor 3): those files are
assembled here byte by byte from the format specifications.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV = os.path.join(ROOT, "mgm_amd", "bin", "imgconv")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(CONV):
        from mgm_amd import build
        build.build_cli()
    assert os.path.exists(CONV)


def decode(path, tmp_path, expect_fail=False):
    out = str(tmp_path / "out.npy")
    r = subprocess.run([CONV, str(path), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if expect_fail:
        assert r.returncode == 2 and "imgconv:" in r.stderr, (r.returncode, r.stderr)
        return None
    assert r.returncode == 0, r.stderr
    a = np.load(out)
    assert a.dtype == np.float32 and a.ndim == 3
    return a


def same(a, b):
    b = np.asarray(b, dtype=np.float32)
    if b.ndim == 2:
        b = b[..., None]
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def smooth(rng, h, w, c=None, hi=255):
    shape = (h, w) if c is None else (h, w, c)
    x = rng.random(shape)
    for ax in (0, 1):
        x = (x + np.roll(x, 1, ax) + np.roll(x, 2, ax) + np.roll(x, 3, ax)) / 4
    x = (x - x.min()) / (x.max() - x.min())
    return np.round(x * hi)


# ---------------------------------------------------------------------------------------------- PNG
@pytest.mark.parametrize("mode", ["L", "LA", "RGB", "RGBA", "I;16"])
def test_png_modes(tmp_path, mode):
    rng = np.random.default_rng(1)
    h, w = 37, 53
    if mode == "I;16":
        a = smooth(rng, h, w, None, 65535).astype(np.uint16)
        im = PIL.fromarray(a)
    else:
        c = {"L": None, "LA": 2, "RGB": 3, "RGBA": 4}[mode]
        a = smooth(rng, h, w, c).astype(np.uint8)
        im = PIL.fromarray(a)
    p = tmp_path / "x.png"
    im.save(p)
    same(decode(p, tmp_path), a)


def test_png_palette_and_transparency(tmp_path):
    rng = np.random.default_rng(2)
    idx = rng.integers(0, 7, (20, 31)).astype(np.uint8)
    pal = rng.integers(0, 256, (7, 3)).astype(np.uint8)
    im = PIL.fromarray(idx)
    im.putpalette(pal.tobytes())
    p = tmp_path / "p.png"
    im.save(p)
    same(decode(p, tmp_path), pal[idx])
    # tRNS on a palette: alpha per entry, 255 where the table is short
    p2 = tmp_path / "pt.png"
    im.save(p2, transparency=bytes([0, 128, 255, 7]))
    alpha = np.array([0, 128, 255, 7, 255, 255, 255], np.uint8)
    same(decode(p2, tmp_path), np.concatenate([pal[idx], alpha[idx][..., None]], -1))


def test_png_trns_grey_and_rgb(tmp_path):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 6, (9, 14)).astype(np.uint8)
    p = tmp_path / "g.png"
    PIL.fromarray(g).save(p, transparency=3)
    same(decode(p, tmp_path), np.stack([g, np.where(g == 3, 0, 255)], -1))
    c = rng.integers(0, 2, (9, 14, 3)).astype(np.uint8)
    p = tmp_path / "c.png"
    PIL.fromarray(c).save(p, transparency=(1, 0, 1))
    key = (c == np.array([1, 0, 1])).all(-1)
    same(decode(p, tmp_path), np.concatenate([c, np.where(key, 0, 255)[..., None]], -1))


def png_bytes(w, h, depth, ctype, rows, filters=None):
    """A PNG assembled by hand: rows = list of per-row byte strings (already packed)."""
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    raw = b"".join(bytes([0 if filters is None else filters[y]]) + rows[y] for y in range(h))
    idat = zlib.compress(raw)
    cut = len(idat) // 2  # two IDAT chunks: the stream must be concatenated
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", idat[:cut]) + chunk(b"IDAT", idat[cut:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_png_subbyte_grey_is_scaled(tmp_path, depth):
    rng = np.random.default_rng(depth)
    h, w = 6, 13  # a ragged last byte
    v = rng.integers(0, 1 << depth, (h, w))
    rows = []
    for y in range(h):
        bits = "".join(format(int(s), "0%db" % depth) for s in v[y])
        bits += "0" * (-len(bits) % 8)
        rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    p = tmp_path / "s.png"
    p.write_bytes(png_bytes(w, h, depth, 0, rows))
    same(decode(p, tmp_path), v * (255 // ((1 << depth) - 1)))


def test_png_every_filter_type(tmp_path):
    """Rows filtered with each of the five filter types by a small encoder written from the PNG specification."""
    rng = np.random.default_rng(5)
    h, w, c = 10, 17, 3
    a = smooth(rng, h, w, c).astype(np.uint8)
    flat = a.reshape(h, w * c).astype(np.int32)
    bpp = c
    rows, filters = [], []
    for y in range(h):
        ft = y % 5
        cur = flat[y]
        up = flat[y - 1] if y else np.zeros_like(cur)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - up
        elif ft == 3:
            f = cur - (left + up) // 2
        else:
            pp = left + up - upleft
            pa, pb, pc = abs(pp - left), abs(pp - up), abs(pp - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            f = cur - pred
        rows.append(bytes((f % 256).astype(np.uint8)))
        filters.append(ft)
    p = tmp_path / "f.png"
    p.write_bytes(png_bytes(w, h, 8, 2, rows, filters))
    assert np.array_equal(np.array(PIL.open(p)), a)  # the hand-made file is a valid PNG
    same(decode(p, tmp_path), a)


def adam7_png(a, depth):
    """An interlaced PNG assembled by hand: the seven reduced images of PNG 1.2 section 8.2, filter type 0 or 2."""
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    a3 = a.reshape(h, w, ch)
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a3[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        prev = None
        for y in range(sub.shape[0]):
            vals = sub[y].reshape(-1)
            if depth == 16:
                row = vals.astype(">u2").tobytes()
            elif depth == 8:
                row = vals.astype(np.uint8).tobytes()
            else:
                bits = "".join(format(int(v), "0%db" % depth) for v in vals)
                bits += "0" * (-len(bits) % 8)
                row = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
            if prev is not None and y % 2:  # filter type 2 (Up) on odd rows of the pass
                raw += b"\x02" + bytes((c - p) % 256 for c, p in zip(row, prev))
            else:
                raw += b"\x00" + row
            prev = row

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("shape,depth", [((13, 21), 8), ((9, 10, 3), 8), ((17, 5), 16), ((3, 2), 8), ((1, 1), 8), ((11, 19), 2),
                                         ((8, 8, 4), 8), ((5, 33, 2), 16)])
def test_png_adam7(tmp_path, shape, depth):
    rng = np.random.default_rng(sum(shape) + depth)
    a = rng.integers(0, 1 << depth, shape)
    p = tmp_path / "a7.png"
    p.write_bytes(adam7_png(a, depth))
    if depth == 8 or (depth == 16 and a.ndim == 2):  # the hand-made file is a valid interlaced PNG for PIL as well
        b = np.array(PIL.open(p))  # (PIL reduces 16-bit multi-channel files to 8 bits: not compared)
        assert np.array_equal(b.astype(np.int64), a)
    same(decode(p, tmp_path), a * (255 // ((1 << depth) - 1)) if depth < 8 else a)


def test_png_unknown_interlace_and_garbage_are_refused(tmp_path):
    good = png_bytes(2, 2, 8, 0, [b"\x01\x02", b"\x03\x04"])
    bad = bytearray(good)
    bad[8 + 8 + 12] = 2  # IHDR interlace byte: only 0 and 1 exist (CRC not checked by this reader)
    p = tmp_path / "i.png"
    p.write_bytes(bytes(bad))
    decode(p, tmp_path, expect_fail=True)
    p.write_bytes(b"this is not an image at all")
    decode(p, tmp_path, expect_fail=True)
    p.write_bytes(good[:40])
    decode(p, tmp_path, expect_fail=True)


# ---------------------------------------------------------------------------------------------- TIFF
@pytest.mark.parametrize("comp", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
@pytest.mark.parametrize("kind", ["u8", "u16", "f32", "rgb"])
def test_tiff_pil(tmp_path, comp, kind):
    rng = np.random.default_rng(7)
    h, w = 61, 45
    if kind == "u8":
        a = smooth(rng, h, w).astype(np.uint8)
    elif kind == "u16":
        a = smooth(rng, h, w, None, 65535).astype(np.uint16)
    elif kind == "f32":
        a = (smooth(rng, h, w) / 7 - 3).astype(np.float32)
        a[3, 4] = np.nan
        a[5, 6] = np.inf
    else:
        a = smooth(rng, h, w, 3).astype(np.uint8)
    p = tmp_path / "x.tif"
    PIL.fromarray(a).save(p, compression=comp)
    same(decode(p, tmp_path), a)


def test_tiff_lzw_long_runs_and_table_resets(tmp_path):
    """Enough data to walk the code width through 9..12 bits and to force ClearCodes."""
    rng = np.random.default_rng(8)
    a = rng.integers(0, 256, (300, 400)).astype(np.uint8)  # incompressible: the table fills and resets
    a[100:200] = 7  # long runs: strings far longer than a row
    p = tmp_path / "l.tif"
    PIL.fromarray(a).save(p, compression="tiff_lzw")
    same(decode(p, tmp_path), a)


def test_tiff_lzw_with_horizontal_predictor(tmp_path):
    rng = np.random.default_rng(9)
    for a in (smooth(rng, 40, 50, 3).astype(np.uint8), smooth(rng, 40, 50, None, 65535).astype(np.uint16)):
        p = tmp_path / "p.tif"
        PIL.fromarray(a).save(p, compression="tiff_lzw", tiffinfo={317: 2})
        assert PIL.open(p).tag_v2.get(317) == 2
        same(decode(p, tmp_path), a)


def tiff_bytes(a, big_endian=False, bigtiff=False, tile=None, planar=False, predictor=1, deflate=False, rows_per_strip=None):
    """A TIFF assembled by hand from the TIFF 6.0 / BigTIFF layouts."""
    E = ">" if big_endian else "<"
    h, w = a.shape[:2]
    spp = 1 if a.ndim == 2 else a.shape[2]
    a3 = a.reshape(h, w, spp)
    bps = a.dtype.itemsize * 8
    fmt = 3 if a.dtype.kind == "f" else 2 if a.dtype.kind == "i" else 1
    planes = [a3[:, :, c:c + 1] for c in range(spp)] if planar else [a3]

    def encode(block):  # block: (rows, cols, samples) of a.dtype
        rows, cols, cs = block.shape
        be = block.astype(block.dtype.newbyteorder(E))
        if predictor == 2:
            d = block.astype(np.uint64)
            d[:, 1:, :] = d[:, 1:, :] - d[:, :-1, :]
            be = (d & ((1 << bps) - 1)).astype("u%d" % (bps // 8)).astype(np.dtype("u%d" % (bps // 8)).newbyteorder(E))
            data = be.tobytes()
        elif predictor == 3:
            out = b""
            for r in range(rows):
                by = block[r].astype(block.dtype.newbyteorder(">")).reshape(-1).view(np.uint8).reshape(cols * cs, bps // 8)
                pl = by.T.reshape(-1).astype(np.int32)  # byte planes, most significant first
                pl[cs:] = pl[cs:] - pl[:-cs]
                out += bytes((pl % 256).astype(np.uint8))
            data = out
        else:
            data = be.tobytes()
        return zlib.compress(data) if deflate else data

    chunks = []
    if tile:
        tw, th = tile
        for pl in planes:
            for y in range(0, h, th):
                for x in range(0, w, tw):
                    blk = np.zeros((th, tw, pl.shape[2]), a.dtype)
                    part = pl[y:y + th, x:x + tw]
                    blk[:part.shape[0], :part.shape[1]] = part
                    chunks.append(encode(blk))
    else:
        rps = rows_per_strip or h
        for pl in planes:
            for y in range(0, h, rps):
                chunks.append(encode(pl[y:y + rps]))

    osz = 8 if bigtiff else 4
    ofmt = "Q" if bigtiff else "I"
    head = (b"MM" if big_endian else b"II") + (struct.pack(E + "HHHQ", 43, 8, 0, 16) if bigtiff else struct.pack(E + "HI", 42, 8))
    tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [bps] * spp), (259, 3, [8 if deflate else 1]), (262, 3, [1]),
            (277, 3, [spp]), (284, 3, [2 if planar else 1]), (317, 3, [predictor]), (339, 3, [fmt] * spp)]
    if tile:
        tags += [(322, 4, [tile[0]]), (323, 4, [tile[1]]), (324, 16 if bigtiff else 4, None), (325, 16 if bigtiff else 4, [len(c) for c in chunks])]
    else:
        tags += [(278, 4, [rows_per_strip or h]), (273, 16 if bigtiff else 4, None), (279, 16 if bigtiff else 4, [len(c) for c in chunks])]
    tags.sort()
    tsz = {3: "H", 4: "I", 16: "Q"}
    ifd_len = (8 if bigtiff else 2) + len(tags) * (20 if bigtiff else 12) + osz
    extra_off = len(head) + ifd_len
    # lay the out-of-line arrays, then the data
    extra = b""
    placed = {}
    for tag, typ, vals in tags:
        n = len(chunks) if vals is None else len(vals)
        size = n * struct.calcsize(tsz[typ])
        if size > osz:
            placed[tag] = extra_off + len(extra)
            extra += b"\0" * size
    data_off = extra_off + len(extra)
    offs, o = [], data_off
    for c in chunks:
        offs.append(o)
        o += len(c)
    ifd = struct.pack(E + ("Q" if bigtiff else "H"), len(tags))
    extra = b""
    for tag, typ, vals in tags:
        vals = offs if vals is None else vals
        packed = struct.pack(E + tsz[typ] * len(vals), *vals)
        ifd += struct.pack(E + "HH" + ofmt, tag, typ, len(vals))
        if tag in placed:
            ifd += struct.pack(E + ofmt, placed[tag])
            extra += packed
        else:
            ifd += packed + b"\0" * (osz - len(packed))
    ifd += struct.pack(E + ofmt, 0)
    return head + ifd + extra + b"".join(chunks)


@pytest.mark.parametrize("variant", [
    dict(), dict(big_endian=True), dict(bigtiff=True), dict(bigtiff=True, big_endian=True), dict(rows_per_strip=7),
    dict(tile=(16, 16)), dict(tile=(32, 16), deflate=True), dict(planar=True), dict(planar=True, tile=(16, 32), big_endian=True),
    dict(deflate=True, rows_per_strip=5), dict(predictor=2, deflate=True), dict(predictor=2, big_endian=True, rows_per_strip=11),
])
@pytest.mark.parametrize("dtype", ["u1", "u2", "i2", "u4", "i4", "f4", "f8"])
def test_tiff_handmade_variants(tmp_path, variant, dtype):
    if variant.get("predictor") == 2 and dtype in ("f4", "f8"):
        pytest.skip("predictor 2 is an integer predictor")
    rng = np.random.default_rng(11)
    h, w, c = 37, 45, 3
    if dtype in ("f4", "f8"):
        a = (smooth(rng, h, w, c) * 1.37 - 100).astype(dtype)
        a[1, 2, 0] = np.nan
    elif dtype in ("i2", "i4"):
        a = (smooth(rng, h, w, c, 30000) - 15000).astype(dtype)
    else:
        a = smooth(rng, h, w, c, 2 ** (8 * int(dtype[1])) - 1 if dtype != "u4" else 2 ** 31).astype(dtype)
    p = tmp_path / "h.tif"
    p.write_bytes(tiff_bytes(a, **variant))
    same(decode(p, tmp_path), a.astype(np.float32))


@pytest.mark.parametrize("variant", [dict(predictor=3), dict(predictor=3, deflate=True, big_endian=True), dict(predictor=3, tile=(16, 16)),
                                     dict(predictor=3, planar=True, rows_per_strip=9)])
@pytest.mark.parametrize("dtype", ["f4", "f8"])
def test_tiff_float_predictor(tmp_path, variant, dtype):
    rng = np.random.default_rng(12)
    a = (smooth(rng, 29, 41, 2) * 0.731 - 17).astype(dtype)
    a[0, 0, 0] = np.inf
    p = tmp_path / "fp.tif"
    p.write_bytes(tiff_bytes(a, **variant))
    same(decode(p, tmp_path), a.astype(np.float32))


def test_tiff_handmade_is_valid_for_other_readers(tmp_path):
    """The hand assembler itself is checked against PIL on the variants PIL reads."""
    rng = np.random.default_rng(13)
    a = smooth(rng, 20, 30).astype(np.uint8)
    for variant in (dict(), dict(big_endian=True), dict(bigtiff=True), dict(deflate=True, rows_per_strip=6), dict(tile=(16, 16))):
        p = tmp_path / "v.tif"
        p.write_bytes(tiff_bytes(a, **variant))
        assert np.array_equal(np.array(PIL.open(p)), a), variant


def test_tiff_unsupported_is_refused(tmp_path):
    a = np.zeros((4, 4), np.uint8)
    raw = bytearray(tiff_bytes(a))
    i = raw.index(struct.pack("<HHI", 259, 3, 1)) + 8
    raw[i:i + 2] = struct.pack("<H", 7)  # JPEG-in-TIFF
    p = tmp_path / "j.tif"
    p.write_bytes(bytes(raw))
    decode(p, tmp_path, expect_fail=True)
    p.write_bytes(bytes(tiff_bytes(a))[:60])
    decode(p, tmp_path, expect_fail=True)


# ---------------------------------------------------------------------------------------------- writers, Netpbm
@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_written_tiff_is_read_by_pil_and_by_us(tmp_path, c):
    rng = np.random.default_rng(14)
    a = (rng.standard_normal((23, 31, c)) * 50).astype(np.float32)
    a[2, 3, 0] = np.nan
    src = tmp_path / "a.npy"
    np.save(src, a)
    out = tmp_path / "a.tif"
    r = subprocess.run([CONV, str(src), str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    same(decode(out, tmp_path), a)
    if c == 1:
        b = np.array(PIL.open(out))
        assert b.dtype == np.float32 and np.array_equal(b.view(np.uint32), a[..., 0].view(np.uint32))
    else:  # PIL has no multi-channel float mode: check the tags and the raw strip instead
        t = PIL.open(out).tag_v2 if c == 1 else None
        raw = out.read_bytes()
        assert raw[:4] == b"II*\0" and raw.endswith(a.tobytes())


def test_pfm_round_trip_and_netpbm(tmp_path):
    rng = np.random.default_rng(15)
    a = rng.standard_normal((9, 11, 1)).astype(np.float32)
    np.save(tmp_path / "a.npy", a)
    assert subprocess.run([CONV, str(tmp_path / "a.npy"), str(tmp_path / "a.pfm")]).returncode == 0
    same(decode(tmp_path / "a.pfm", tmp_path), a)
    g = rng.integers(0, 256, (7, 5)).astype(np.uint8)
    (tmp_path / "g.pgm").write_bytes(b"P5\n# a comment\n5 7\n255\n" + g.tobytes())
    same(decode(tmp_path / "g.pgm", tmp_path), g)
    g16 = rng.integers(0, 65536, (7, 5)).astype(">u2")
    (tmp_path / "g16.pgm").write_bytes(b"P5 5 7 65535\n" + g16.tobytes())
    same(decode(tmp_path / "g16.pgm", tmp_path), g16.astype(np.float32))
    c = rng.integers(0, 256, (4, 6, 3)).astype(np.uint8)
    (tmp_path / "c.ppm").write_bytes(b"P6\n6 4\n255\n" + c.tobytes())
    same(decode(tmp_path / "c.ppm", tmp_path), c)
    (tmp_path / "t.pgm").write_text("P2\n# plain\n3 2\n9\n0 1 2\n3 4 9\n")
    same(decode(tmp_path / "t.pgm", tmp_path), np.array([[0, 1, 2], [3, 4, 9]]))


def test_unknown_output_suffix_is_refused(tmp_path):
    np.save(tmp_path / "a.npy", np.zeros((2, 2), np.float32))
    r = subprocess.run([CONV, str(tmp_path / "a.npy"), str(tmp_path / "a.png")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 2 and "float images" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/data"), reason="the reference's sample images live in the build container only")
def test_reference_sample_images_decode_as_pil_decodes_them(tmp_path):
    import glob
    files = sorted(glob.glob("/root/reference/data/*.png") + glob.glob("/root/reference/data/*.tif"))
    assert len(files) >= 8
    for f in files:
        im = PIL.open(f)
        if im.mode == "P":
            im = im.convert("RGB")
        same(decode(f, tmp_path), np.array(im))


# ---------------------------------------------------------------------------------------------- against iio itself
REF_IMG = os.path.join(ROOT, "oracle", "_ref", "mgm_img")  # the reference CLI with iio's libpng/libtiff readers


def _ref_run(args, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="2", TSGM="2", **(env or {}))
    r = subprocess.run([REF_IMG, "-r", "-6", "-R", "6", "-O", "4"] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout


def _pair(rng, h=40, w=56, c=None, hi=255):
    u = smooth(rng, h, w, c, hi)
    v = np.roll(u, 3, 1)
    return u, v


@pytest.mark.skipif(not os.path.exists(REF_IMG), reason="oracle/_ref/mgm_img was not built (needs libpng/libtiff headers)")
def test_inputs_decode_as_iio_decodes_them(tmp_path):
    """The reference CLI run on the image files themselves and on our decoding of them (.npy) must agree bit for bit:
    AD costs against fixed P1/P2 depend on every sample value, channel count and scaling."""
    rng = np.random.default_rng(21)
    files = []
    u, v = _pair(rng, c=3)
    for n, a in (("u", u), ("v", v)):
        PIL.fromarray(a.astype(np.uint8)).save(tmp_path / ("rgb_%s.png" % n))
    files.append(("rgb_u.png", "rgb_v.png"))
    u, v = _pair(rng, hi=65535)
    for n, a in (("u", u), ("v", v)):
        PIL.fromarray(a.astype(np.uint16)).save(tmp_path / ("g16_%s.png" % n))
        PIL.fromarray((a / 3 - 1000).astype(np.float32)).save(tmp_path / ("f_%s.tif" % n), compression="tiff_lzw")
        PIL.fromarray(a.astype(np.uint16)).save(tmp_path / ("u16_%s.tif" % n), compression="tiff_adobe_deflate")
    files += [("g16_u.png", "g16_v.png"), ("f_u.tif", "f_v.tif"), ("u16_u.tif", "u16_v.tif")]
    # 2-bit grey (scaled by 85), grey + tRNS (becomes 2 channels), palette (becomes RGB)
    u, v = _pair(rng, hi=3)
    for n, a in (("u", u), ("v", v)):
        rows = []
        for y in range(a.shape[0]):
            bits = "".join(format(int(s), "02b") for s in a[y])
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        (tmp_path / ("g2_%s.png" % n)).write_bytes(png_bytes(a.shape[1], a.shape[0], 2, 0, rows))
        PIL.fromarray((a * 60).astype(np.uint8)).save(tmp_path / ("gt_%s.png" % n), transparency=120)
        im = PIL.fromarray(a.astype(np.uint8))
        im.putpalette(bytes([10, 200, 30, 90, 0, 250, 255, 255, 0, 77, 77, 78]))
        im.save(tmp_path / ("pal_%s.png" % n))
    files += [("g2_u.png", "g2_v.png"), ("gt_u.png", "gt_v.png"), ("pal_u.png", "pal_v.png")]
    u, v = _pair(rng, c=3)
    for n, a in (("u", u), ("v", v)):
        (tmp_path / ("a7_%s.png" % n)).write_bytes(adam7_png(a.astype(np.int64), 8))
    files.append(("a7_u.png", "a7_v.png"))
    for fu, fv in files:
        a = decode(tmp_path / fu, tmp_path)
        np.save(tmp_path / "du.npy", a)
        np.save(tmp_path / "dv.npy", decode(tmp_path / fv, tmp_path))
        _ref_run([tmp_path / fu, tmp_path / fv, tmp_path / "o1.npy", tmp_path / "c1.npy"])
        _ref_run([tmp_path / "du.npy", tmp_path / "dv.npy", tmp_path / "o2.npy", tmp_path / "c2.npy"])
        for x in ("o", "c"):
            r1, r2 = np.load(tmp_path / (x + "1.npy")), np.load(tmp_path / (x + "2.npy"))
            assert np.isfinite(r1).any()
            assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32)), (fu, x, a.shape)


@pytest.mark.skipif(not os.path.exists(REF_IMG), reason="oracle/_ref/mgm_img was not built (needs libpng/libtiff headers)")
def test_tiffs_written_by_iio_are_read_back(tmp_path):
    """iio writes float BigTIFFs, LZW-compressed below 2000x2000 (iio.c:3972-4043): its .tif and .npy outputs of the
    same run must decode to the same samples here."""
    rng = np.random.default_rng(22)
    u, v = _pair(rng, 50, 70)
    np.save(tmp_path / "u.npy", u.astype(np.float32))
    np.save(tmp_path / "v.npy", v.astype(np.float32))
    for env in ({}, {"IIOTIFF_PLAIN": "1"}):
        _ref_run([tmp_path / "u.npy", tmp_path / "v.npy", tmp_path / "o.tif", tmp_path / "c.tif"], env)
        _ref_run([tmp_path / "u.npy", tmp_path / "v.npy", tmp_path / "o.npy", tmp_path / "c.npy"], env)
        for x in ("o", "c"):
            raw = (tmp_path / (x + ".tif")).read_bytes()
            assert raw[:4] == b"II+\0"  # BigTIFF
            same(decode(tmp_path / (x + ".tif"), tmp_path), np.load(tmp_path / (x + ".npy")))


@pytest.mark.skipif(not os.path.exists(REF_IMG), reason="oracle/_ref/mgm_img was not built (needs libpng/libtiff headers)")
def test_tiffs_written_here_are_read_by_iio(tmp_path):
    rng = np.random.default_rng(23)
    u, v = _pair(rng, 50, 70, 3)
    for n, a in (("u", u), ("v", v)):
        np.save(tmp_path / (n + ".npy"), (a / 7).astype(np.float32))
        assert subprocess.run([CONV, str(tmp_path / (n + ".npy")), str(tmp_path / (n + ".tif"))], stdout=subprocess.DEVNULL).returncode == 0
    _ref_run([tmp_path / "u.tif", tmp_path / "v.tif", tmp_path / "o1.npy"])
    _ref_run([tmp_path / "u.npy", tmp_path / "v.npy", tmp_path / "o2.npy"])
    r1, r2 = np.load(tmp_path / "o1.npy"), np.load(tmp_path / "o2.npy")
    assert np.isfinite(r1).any() and np.array_equal(r1.view(np.uint32), r2.view(np.uint32))


def test_npy_fortran_order_and_squeeze(tmp_path):
    """np.save of an F-contiguous array writes 'fortran_order': True (a transposed view, or what fancy indexing leaves behind): iio
    reads it (iio.c:3209-3252: swap the sides, read, transpose), and so does src/npyio.h -- the image, not its transpose."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 255, size=(23, 31)).astype(np.float32)
    f = np.asfortranarray(a)
    assert f.flags.f_contiguous and not f.flags.c_contiguous
    np.save(tmp_path / "f.npy", f)
    assert b"'fortran_order': True" in open(tmp_path / "f.npy", "rb").read(200)
    same(decode(tmp_path / "f.npy", tmp_path), a)
    np.save(tmp_path / "s.npy", a[None])  # shape (1, 23, 31): iio squeezes the leading one away (iio.c:3202-3207)
    same(decode(tmp_path / "s.npy", tmp_path), a)


@pytest.mark.skipif(not os.path.exists(REF_IMG), reason="oracle/_ref/mgm_img was not built (needs libpng/libtiff headers)")
def test_npy_fortran_order_decodes_as_iio_decodes_it(tmp_path):
    rng = np.random.default_rng(22)
    for c in (None, 3):  # (with a third axis iio keeps treating it as the interleaved pixel dimension: same bytes, same image here)
        u, v = _pair(rng, c=c)
        np.save(tmp_path / "fu.npy", np.asfortranarray(u.astype(np.float32)))
        np.save(tmp_path / "fv.npy", np.asfortranarray(v.astype(np.float32)))
        np.save(tmp_path / "du.npy", decode(tmp_path / "fu.npy", tmp_path))
        np.save(tmp_path / "dv.npy", decode(tmp_path / "fv.npy", tmp_path))
        _ref_run([tmp_path / "fu.npy", tmp_path / "fv.npy", tmp_path / "o1.npy", tmp_path / "c1.npy"])
        _ref_run([tmp_path / "du.npy", tmp_path / "dv.npy", tmp_path / "o2.npy", tmp_path / "c2.npy"])
        for x in ("o", "c"):
            r1, r2 = np.load(tmp_path / (x + "1.npy")), np.load(tmp_path / (x + "2.npy"))
            assert r1.shape == r2.shape and np.array_equal(r1.view(np.uint32), r2.view(np.uint32)), (c, x)
