"""Build the native library in-tree: hipcc for gfx950 -> gecco_amd/lib/libgecco_crf.so.

The .so is git-ignored (history stays source-only) but travels to the GPU box with the
gpurun snapshot.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgecco_crf.so")
SOURCES = ["crf_model.cpp", "crf_plan.cpp", "crf_session.cpp", "crf_tables.cpp", "capi.cpp", "crf_kernels.hip", "crf_sequence.hip", "crf_segment.hip", "crf_general.hip", "crf_composition.hip", "crf_exact.hip"]
HEADERS = ["crf_model.hpp", "crf_plan.hpp", "crf_device.hpp", "crf_scan.hpp", "crf_vd_short.hpp", "crf_session.hpp", "crf_tables.hpp", "crf_exact_exp.hpp", os.path.join("..", "..", "include", "gecco_crf.h")]
# reference-bits mode and the correctly rounded exp rely on every multiply and add being rounded on its own: no fused multiply-add
# where the source has none
EXTRA_FLAGS = {"crf_exact.hip": ["-ffp-contract=off"], "capi.cpp": ["-ffp-contract=off"]}
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build gecco_amd)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


LAST_BUILD = {}  # what the last build_native() / build_objpath() of this process did: {"libgecco_crf": "compiled" | "reused", ...}


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        LAST_BUILD["libgecco_crf"] = "reused (up to date against every source and header)"
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    hipcc = _hipcc()
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra",
              "-Wno-unused-parameter"]
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *common, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    # translation units are independent: compile them side by side (the two big kernel files dominate: ~1 min instead of ~3)
    with ThreadPoolExecutor(max_workers=max(1, min(len(SOURCES), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    LAST_BUILD["libgecco_crf"] = f"compiled {len(SOURCES)} translation units for {ARCH}"
    tmp = f"{LIB}.{os.getpid()}.tmp"
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs])
    os.replace(tmp, LIB)
    return LIB


OBJPATH_SRC = os.path.join(CSRC, "objpath.c")


def objpath_path() -> str:
    import sysconfig

    return os.path.join(LIBDIR, "_objpath" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_objpath(force: bool = False, verbose: bool = False) -> str:
    """The CPython extension with the object-model loops of the drop-in class (csrc/objpath.c): gcc + Python headers."""
    import sysconfig

    out = objpath_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(OBJPATH_SRC):
        LAST_BUILD["_objpath"] = "reused (up to date)"
        return out
    LAST_BUILD["_objpath"] = "compiled"
    os.makedirs(LIBDIR, exist_ok=True)
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler for gecco_amd/csrc/objpath.c")
    tmp = f"{out}.{os.getpid()}.tmp"  # (several ranks may build at once: everyone links into a file of its own)
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I" + sysconfig.get_paths()["include"],
           OBJPATH_SRC, "-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build_objpath(force="--force" in sys.argv, verbose=True))
    print(build_native(force="--force" in sys.argv, verbose=True))
