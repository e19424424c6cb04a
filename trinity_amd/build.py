"""In-tree build of the native libraries (explicit hipcc / g++; no JIT cache, so the .so files travel with the
repo snapshot to the GPU box)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_HIP = os.path.join(PKG, "libtrinity_hip.so")
LIB_HOST = os.path.join(PKG, "libtrinity_host.so")
HIP_SRCS = [os.path.join(PKG, "csrc", "trinity_hip.hip"), os.path.join(PKG, "csrc", "commit_sort.hip")]
HOST_SRCS = [os.path.join(PKG, "csrc", "host", "synth.cpp"), os.path.join(PKG, "csrc", "host", "plan_host.cpp")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(srcs):
    out = list(srcs) + [os.path.join(ROOT, "include", "trinity_hip.h")]
    for d in {os.path.dirname(s) for s in srcs} | {os.path.join(PKG, "csrc")}:  # (the host tools include the planner's headers of csrc/)
        out += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hpp", ".h", ".cuh"))]
    return out


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force=False):
    if force or _newer(LIB_HIP, _deps(HIP_SRCS)):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-o", LIB_HIP] + HIP_SRCS + ["-ldl"]
        subprocess.run(cmd, check=True)
    return LIB_HIP


def build_host(force=False):
    if force or _newer(LIB_HOST, _deps(HOST_SRCS)):
        cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", LIB_HOST] + HOST_SRCS
        subprocess.run(cmd, check=True)
    return LIB_HOST


MIRROR_TEST_SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
MIRROR_TEST_BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def build_mirror_test(force=False):
    """The C++ operator surface (csrc/host/trinity_gpu.hpp) compiled into its test driver, linked to libtrinity_hip.so."""
    deps = [MIRROR_TEST_SRC, os.path.join(PKG, "csrc", "host", "trinity_gpu.hpp"), os.path.join(PKG, "csrc", "host", "google_encoder.hpp"), os.path.join(ROOT, "include", "trinity_hip.h")]
    if force or _newer(MIRROR_TEST_BIN, deps):
        build_hip()
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", MIRROR_TEST_BIN, MIRROR_TEST_SRC, "-L" + PKG, "-ltrinity_hip", "-Wl,-rpath,$ORIGIN/../../trinity_amd"]
        subprocess.run(cmd, check=True)
    return MIRROR_TEST_BIN


MIRROR_WRITE_SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_write_test.cpp")
MIRROR_WRITE_BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_write_test")


def build_mirror_write_test(force=False):
    """The write side of the operator surface (csrc/host/trinity_gpu_write.hpp: SegmentIndexSession begin / insert / commit, merge) compiled into its
    driver; in-tree, so that it travels to the GPU box, where tests/test_host_mirror.py runs it."""
    deps = [MIRROR_WRITE_SRC, os.path.join(PKG, "csrc", "host", "trinity_gpu_write.hpp"), os.path.join(PKG, "csrc", "host", "google_encoder.hpp"), os.path.join(ROOT, "include", "trinity_hip.h")]
    if force or _newer(MIRROR_WRITE_BIN, deps):
        build_hip()
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", MIRROR_WRITE_BIN, MIRROR_WRITE_SRC, "-L" + PKG, "-ltrinity_hip", "-Wl,-rpath,$ORIGIN/../../trinity_amd"]
        subprocess.run(cmd, check=True)
    return MIRROR_WRITE_BIN


def build_all(force=False):
    return build_hip(force), build_host(force), build_mirror_test(force), build_mirror_write_test(force)


def kernels_stamp():
    """SHA-256 (16 hex digits) over the device sources (csrc/*.hip, csrc/*.hpp): what a committed profile is stamped with — bench.py quotes
    profiles/pmc_latest.json's traffic only while the stamp it carries is that of the kernels it runs."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]
