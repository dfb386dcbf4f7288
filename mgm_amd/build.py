"""Build libmgm_hip.so for gfx950 with hipcc, in-tree (mgm_amd/lib/).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build
container; the resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libmgm_hip.so")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          "-Wall", "-Wno-unused-function"]
# per-translation-unit extras (see the header comment of each file)
# (source, object suffix, extra flags)
REL_EXTRA = os.environ.get("MGM_REL_DEFINES", "").split()  # e.g. "-DMGM_REL_PHASES=1" (development build of the range-proportional kernels)
UNITS = [("mgm_pass.hip", "", ["-fno-honor-nans"]), ("mgm_pass_rel.hip", "", ["-fno-honor-nans"] + REL_EXTRA)]
P2_EXTRA = os.environ.get("MGM_P2_DEFINES", "").split()  # e.g. "-DMGM_P2_MAXD=3" (tuning experiments)
UNITS += [("mgm_pass2.hip", "_lpl%d" % n, ["-fno-honor-nans", "-DMGM_P2_LPL=%d" % n] + P2_EXTRA) for n in (1, 2, 3, 4, 6, 8, 12, 16)]
UNITS += [("mgm_pass2_dispatch.hip", "", P2_EXTRA), ("mgm_cost.hip", "", []), ("mgm_cost_fast.hip", "", []), ("mgm_wta.hip", "", []), ("mgm_post.hip", "", []),
          ("mgm_api.hip", "", []), ("mgm_ctx.hip", "", []), ("mgm_plan.hip", "", []), ("mgm_multi.hip", "", []),
          ("mgm_pass_exact.hip", "", [])]  # (no -fno-honor-nans: this one exists for the NaNs)


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, "mgm_device.h"), os.path.join(CSRC, "mgm_pass_common.h"), os.path.join(CSRC, "mgm_host.h"), os.path.join(CSRC, "mgm_cost_common.h"),
               os.path.join(HERE, "..", "include", "mgm_hip.h"),
               os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src, suffix, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", suffix + ".o"))
        objs.append(o)
        cmd = [cc] + COMMON + extra + ["-c", s, "-o", o]
        # an object built with other flags (MGM_P2_DEFINES of a tuning or development build) is stale too
        if force or _stale(o, [s] + headers) or _read(o + ".cmd") != " ".join(cmd):
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        if "-c" in cmd:
            with open(cmd[-1] + ".cmd", "w") as f:
                f.write(" ".join(cmd))
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for out in ex.map(run, jobs):
                if verbose and out.strip():
                    print(out)
    if jobs or force or _stale(LIB, objs):
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB])
    build_cli(force)
    return LIB


def build_cli(force=False):
    """The host program: src/mgm_main.cc (plain g++) linked against libmgm_hip.so -> mgm_amd/bin/mgm."""
    src = os.path.join(HERE, "..", "src", "mgm_main.cc")
    hdrs = [os.path.join(HERE, "..", "src", "npyio.h"), os.path.join(HERE, "..", "src", "imgio.h"),
            os.path.join(HERE, "..", "include", "mgm_hip.h")]
    exe = os.path.join(HERE, "bin", "mgm")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if force or _stale(exe, [src, LIB] + hdrs):
        rocm = os.path.dirname(os.path.dirname(hipcc()))
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", src, "-L" + LIBDIR, "-lmgm_hip", "-L" + os.path.join(rocm, "lib"),
               "-lamdhip64", "-lz", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("g++ failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
    # the image decoders alone, no GPU library: format conversion and the CPU tests of src/imgio.h
    conv_src = os.path.join(HERE, "..", "src", "imgconv.cc")
    conv = os.path.join(HERE, "bin", "imgconv")
    if force or _stale(conv, [conv_src] + hdrs[:2]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", conv_src, "-lz", "-o", conv]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("g++ failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
