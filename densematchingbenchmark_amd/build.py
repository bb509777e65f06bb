"""Build recipe for libdmb_hip.so (the C-ABI library declared in include/dmb_hip.h).

Plain ``hipcc --offload-arch=gfx950`` per translation unit, then one ``-shared`` link, all in-tree under
``densematchingbenchmark_amd/lib/`` so that the built library travels with a repository snapshot.
hipcc cross-compiles without a GPU.  Run as ``python -m densematchingbenchmark_amd.build``.
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdmb_hip.so")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")

SOURCES = ["core.cpp", "volume.hip", "regression.hip", "conv3d.hip", "confhead.hip", "gwc_mfma.hip", "conv2d.hip", "losses.hip", "conv3d_x6.hip", "wgrad.hip", "norm.hip", "path_bwd.hip", "catconv.hip", "warp_volume.hip", "deconv3d_zy.hip", "spn.hip", "preprocess.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# warp_volume.hip restates the reference's FP32 sampler arithmetic operation by operation: no fused multiply-adds there
EXTRA_FLAGS = {"warp_volume.hip": ["-ffp-contract=off"], "spn.hip": ["-ffp-contract=off"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, dev=False):
    """Compile every HIP source for gfx950 and link libdmb_hip.so.  Returns the library path.

    ``dev=True`` builds the DEVELOPMENT variant instead: the same sources with -DDMB_DEV (kernel-variant switches and diagnostic
    branches for the A/B measurements of scripts/, reached through ``dmb_dev_set_option``) into lib/libdmb_hip_dev.so.  Nothing in
    the package, the tests or bench.py loads it (scripts select it with DMB_LIB=dev); __graft_entry__.build() does not build it."""
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    lib_path = os.path.join(LIB_DIR, "libdmb_hip_dev.so") if dev else LIB_PATH
    suffix, extra = (".dev.o", ["-DDMB_DEV"]) if dev else (".o", [])
    # build-time experiments (development build only): DMB_BUILD_TAG=st16 DMB_BUILD_DEFS="-DDMB_ZY_ST=16" -> lib/libdmb_hip_dev_st16.so,
    # loaded by scripts with DMB_LIB=dev_st16
    tag = os.environ.get("DMB_BUILD_TAG", "") if dev else ""
    if tag:
        lib_path = os.path.join(LIB_DIR, "libdmb_hip_dev_%s.so" % tag)
        suffix, extra = ".dev_%s.o" % tag, extra + os.environ.get("DMB_BUILD_DEFS", "").split()
    headers = [os.path.join(CSRC, "dmb_common.h"), os.path.join(CSRC, "interp.h"), os.path.join(INCLUDE, "dmb_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError("libdmb_hip.so source listed in build.SOURCES is missing: %s" % sp)
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + suffix)
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [hipcc, "--offload-arch=" + ARCH] + FLAGS + extra + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", sp, "-o", obj]
            jobs.append(cmd)
    if jobs:   # the translation units are independent: compile them side by side (bounded by the host's cores)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[dmb build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(lib_path, objs):
        # -Bsymbolic: references between the library's own translation units bind inside the library (two builds of it can then
        # be loaded side by side in one process)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", lib_path] + objs
        if verbose:
            print("[dmb build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, dev="--dev" in sys.argv))
