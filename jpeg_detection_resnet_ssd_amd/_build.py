"""In-tree build of the gfx950 C-ABI library (csrc/*.hip -> csrc/libdj_hip.so).

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to the GPU
box with the working tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libdj_hip.so")
JPEG_LIB_PATH = os.path.join(CSRC, "libdj_jpeg.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_library(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link them into libdj_hip.so."""
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(CSRC, "..", "..", "include", "dj_hip.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17"]

    def compile_one(src):
        obj = os.path.join(objdir, src[:-4] + ".o")
        srcp = os.path.join(CSRC, src)
        if force or _newer([srcp] + headers, obj):
            cmd = [hipcc] + flags + ["-c", srcp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 6, 8, max(1, len(sources)))) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or _newer(objs, LIB_PATH):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    build_jpeg_library(force)
    return LIB_PATH


def build_jpeg_library(force=False):
    """Host-only JPEG coefficient reader (csrc/dj_jpeg.cpp -> csrc/libdj_jpeg.so), plain g++."""
    src = os.path.join(CSRC, "dj_jpeg.cpp")
    hdr = os.path.join(CSRC, "..", "..", "include", "dj_jpeg.h")
    if force or _newer([src, hdr], JPEG_LIB_PATH):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", src, "-o",
               JPEG_LIB_PATH]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for dj_jpeg.cpp:\n%s\n%s" % (r.stdout, r.stderr))
    return JPEG_LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
