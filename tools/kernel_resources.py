#!/usr/bin/env python3
"""Register / scratch / LDS budget of every device kernel of libmgm_hip.so, from hipcc's own resource remarks.

    python tools/kernel_resources.py [--filter k_pass2] [--out profiles/r04_kernel_resources.txt] [-D MGM_P2_ONEB_WPE=5 ...]

Compiles every translation unit of mgm_amd/build.py with -Rpass-analysis=kernel-resource-usage (objects go to a
scratch directory, the product library is not touched) and prints one line per kernel.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mgm_amd import build as B  # noqa: E402


def demangle(names):
    import shutil
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt):
        return names
    r = subprocess.run([filt], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return r.stdout.splitlines()


def parse(text):
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = blk.split("\n")[0].split(" [-R")[0].strip()

        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         occ=g(r"Occupancy \[waves/SIMD\]"), sspill=g("SGPRs Spill"), vspill=g("VGPRs Spill"),
                         lds=g(r"LDS Size \[bytes/block\]")))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--units", default="", help="substring of the unit's object suffix / source, e.g. lpl4")
    ap.add_argument("--out", default="")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    cc = B.hipcc()
    tmp = tempfile.mkdtemp(prefix="mgmres")
    jobs = []
    for src, suffix, extra in B.UNITS:
        tag = src.replace(".hip", suffix)
        if a.units and a.units not in tag:
            continue
        cmd = [cc] + B.COMMON + extra + ["-D" + d for d in a.D] + ["-Rpass-analysis=kernel-resource-usage", "-c",
                                                                   os.path.join(B.CSRC, src), "-o", os.path.join(tmp, tag + ".o")]
        jobs.append((tag, cmd))

    def run(j):
        r = subprocess.run(j[1], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError(r.stdout[-4000:])
        return j[0], parse(r.stdout)
    lines = []
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for tag, rows in ex.map(run, jobs):
            names = demangle([r["name"] for r in rows])
            for r, n in zip(rows, names):
                n = re.sub(r"\(.*$", "", n).replace("void mgm::", "").replace("mgm::", "")
                if a.filter and a.filter not in n:
                    continue
                lines.append("%-22s %-64s vgpr %3d agpr %3d sgpr %3d  vgpr-spill %3d sgpr-spill %3d scratch %4d B/lane  occupancy %d" %
                             (tag, n, r["vgpr"], r["agpr"], r["sgpr"], r["vspill"], r["sspill"], r["scratch"], r["occ"]))
    hdr = ("# hipcc --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, flags of mgm_amd/build.py%s\n"
           "# k_pass2<LPL, FH, WEIGHTED, MGM, C8, SUBV, DEEP, XCDQ, ONEB>\n" % ((" + -D" + " -D".join(a.D)) if a.D else ""))
    text = hdr + "\n".join(lines) + "\n"
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
