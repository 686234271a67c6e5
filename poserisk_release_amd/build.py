"""Build recipe for libposerisk_hip.so (gfx950 only).

`python -m poserisk_release_amd.build` compiles every .hip translation unit under csrc/ with
hipcc (cross-compiles without a GPU) and links them into the in-tree shared library that
`poserisk_release_amd._lib` loads with ctypes.  Objects are rebuilt only when a source or header
is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "_build")
LIB_PATH = os.path.join(HERE, "libposerisk_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall",
            "-Wno-unused-function", "-Wno-tautological-overlap-compare",
            "-ffp-contract=fast-honor-pragmas"]
# ablation builds only (timing hooks that make results wrong stay out of the shipped library): POSERISK_CXXFLAGS=-DPR_TIMING_HOOKS
CXXFLAGS_EXTRA = os.environ.get("POSERISK_CXXFLAGS", "").split()
CXXFLAGS += CXXFLAGS_EXTRA


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cc"))


def _newest_header_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "poserisk_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, hdr_mtime, verbose):
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_mtime):
        return obj
    # .cc = device-free host code (host_plan.cc): plain C++, no offload pass
    flags = [f for f in CXXFLAGS if not f.startswith("--offload-arch") and f != "-fno-gpu-rdc"] if src.endswith(".cc") else CXXFLAGS
    cmd = [HIPCC, *flags, "-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(verbose=False, force=False, out=None):
    """out: another library path (A/B and ablation builds, selected at run time with POSERISK_LIB_PATH): its objects go
    to `<out>.objs/`, the shipped library and its objects are not touched."""
    global OBJ_DIR
    lib_path = out or LIB_PATH
    if out:
        OBJ_DIR = out + ".objs"
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    hdr_mtime = _newest_header_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr_mtime, verbose), srcs))
    if (not os.path.exists(lib_path)) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib_path


if __name__ == "__main__":
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    if not out and CXXFLAGS_EXTRA:
        raise SystemExit("POSERISK_CXXFLAGS is set: name the experiment build's file with --out <path>.so "
                         "(the shipped library is only ever built with the release flags)")
    print(build(verbose=True, force="--force" in sys.argv, out=out))
