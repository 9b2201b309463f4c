"""Builds gpu_raytracer_amd/librt_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

-ffp-contract=off: the kernels keep the reference's f32 operation order (see kernels.hip);
HIP's default correctly-rounded f32 divide/sqrt is left on.
-fno-slp-vectorize: the SLP vectoriser pairs scalar f32 operations of the triangle test and the shading code into v_pk_* at the
price of register moves to line the pairs up: 72 instead of 80 VGPRs and 2.3 % less frame time without it (profiles/ab_r02.json);
the packed operations of the box test are written out in device_common.h and stay.
-mllvm -enable-post-misched=0: the traversal loops are VALU-issue bound with 6 waves per SIMD hiding latency; the post-RA
machine scheduler's reordering costs 1 % there (same A/B).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt_hip.so")
SOURCES = ["kernels.hip", "wavefront.hip", "device_build.hip", "shadow_grid.hip", "rt_api.cpp", "bvh_builder.cpp", "rt_host_api.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=0",
         "-Wall", "-Wno-unused-function", "-pthread", "-I" + os.path.join(HERE, "..", "include")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
    deps += [os.path.join(HERE, "..", "include", f) for f in ("rt_hip.h", "rt_shared.h")]
    return any(os.path.isfile(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra=(), out=None):
    global LIB
    if out:
        LIB = out
        force = True
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc] + FLAGS + list(extra) + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


EXAMPLE_SRC = os.path.join(HERE, "..", "examples", "rt_render.cpp")
EXAMPLE_BIN = os.path.join(HERE, "..", "build", "rt_render")


def build_examples(force=False, verbose=True):
    """examples/rt_render.cpp: the native headless front end over the C ABI (host-only C++, links librt_hip.so)."""
    deps = [EXAMPLE_SRC, LIB] + [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
    if not force and os.path.exists(EXAMPLE_BIN) and all(os.path.getmtime(d) <= os.path.getmtime(EXAMPLE_BIN) for d in deps):
        return EXAMPLE_BIN
    os.makedirs(os.path.dirname(EXAMPLE_BIN), exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", EXAMPLE_SRC, "-I" + os.path.join(HERE, "..", "include"), "-I" + CSRC,
           "-L" + HERE, "-lrt_hip", "-Wl,-rpath,$ORIGIN/../gpu_raytracer_amd", "-o", EXAMPLE_BIN]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return EXAMPLE_BIN


if __name__ == "__main__":
    out = None
    args = sys.argv[1:]
    if "--out" in args:
        i = args.index("--out")
        out = args[i + 1]
        del args[i:i + 2]
    build(force="--force" in args, extra=[a for a in args if a.startswith("-") and a != "--force"], out=out)
    if out is None:
        build_examples(force="--force" in args)
