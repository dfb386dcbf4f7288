"""Byte-level fuzzing of the image decoders (src/imgio.h): corrupt valid PNG/TIFF/PNM files and require the
converter to end with exit code 0 or 2 -- never a crash, a sanitizer report or a hang.

    g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined src/imgconv.cc -lz -o /tmp/imgconv_asan
    python tools/fuzz_imgio.py /tmp/imgconv_asan [cases per seed]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_imgio as T  # noqa: E402  (the hand assemblers)


def main():
    exe = sys.argv[1]
    per_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    seeds = []
    a = T.smooth(rng, 20, 30, 3).astype(np.uint8)
    p = os.path.join(tmp, "s")
    Image.fromarray(a).save(p + ".png")
    seeds.append(open(p + ".png", "rb").read())
    im = Image.fromarray(rng.integers(0, 4, (10, 12)).astype(np.uint8))
    im.putpalette(bytes(range(12)))
    im.save(p + ".png", transparency=bytes([1, 2]))
    seeds.append(open(p + ".png", "rb").read())
    for comp in (None, "tiff_lzw", "tiff_adobe_deflate", "packbits"):
        Image.fromarray(a).save(p + ".tif", compression=comp)
        seeds.append(open(p + ".tif", "rb").read())
    seeds.append(T.tiff_bytes(a.astype("f4"), predictor=3, tile=(16, 16)))
    seeds.append(T.tiff_bytes(a.astype("u2"), predictor=2, planar=True, bigtiff=True, big_endian=True))
    seeds.append(T.adam7_png(a, 8))
    seeds.append(b"P5\n5 7\n255\n" + bytes(35))
    seeds.append(b"P2\n2 2\n9\n1 2 3 4\n")
    seeds.append(b"Pf\n3 2\n-1\n" + bytes(24))
    bad = n = 0
    fin, fout = os.path.join(tmp, "in"), os.path.join(tmp, "out.npy")
    for s in seeds:
        for _ in range(per_seed):
            b = bytearray(s)
            for _ in range(rng.integers(1, 6)):
                mode = rng.integers(0, 3)
                if mode == 0:
                    b[rng.integers(0, len(b))] = rng.integers(0, 256)
                elif mode == 1 and len(b) > 2:
                    b = b[:rng.integers(1, len(b))]
                else:
                    i = rng.integers(0, max(1, len(b) - 4))
                    b[i:i + 4] = bytes(rng.integers(0, 256, 4).astype(np.uint8))
            open(fin, "wb").write(bytes(b))
            try:
                r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=20)
                rc, err = r.returncode, r.stderr
            except subprocess.TimeoutExpired:
                rc, err = "timeout", ""
            n += 1
            if rc not in (0, 2):
                bad += 1
                if bad <= 5:
                    keep = os.path.join(tmp, "bad%d" % bad)
                    open(keep, "wb").write(bytes(b))
                    print("rc", rc, keep, err[-500:])
    print(n, "cases,", bad, "bad")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
