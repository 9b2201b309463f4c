"""The device BVH builder (host code, gpu_raytracer_amd/csrc/bvh_builder.cpp) under AddressSanitizer + UBSan, checked by a
structural validator that decodes the 8-wide nodes the way the kernels do (tests/check_bvh.cpp): slot / mask consistency, empty
slots inverted, each triangle in exactly one leaf, conservative boxes, depth within the kernels' stack."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("bvh") / "check_bvh"
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread",
                         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gpu_raytracer_amd", "csrc"),
                         os.path.join(HERE, "check_bvh.cpp"), os.path.join(ROOT, "gpu_raytracer_amd", "csrc", "bvh_builder.cpp"), "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-3000:]
    return str(exe)


KINDS = {0: "soup", 1: "coplanar", 2: "coincident points", 3: "collinear chain", 4: "huge + tiny", 5: "NaN / inf vertices"}


@pytest.mark.parametrize("method", [0, 1], ids=["binned_sah", "ploc"])
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_device_bvh_structure(checker, kind, method):
    """method 1 is the host statement of the device build (device_build.hip): PLOC must stay logarithmic in rounds and shallow
    enough for the kernels' stacks on coincident points and on chains, where its nearest-neighbour rule degenerates."""
    env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for n in (0, 1, 2, 5, 37, 1000, 20000):
        run = subprocess.run([checker, str(n), str(17 * kind + n), str(kind), str(method)], capture_output=True, text=True, env=env, timeout=300)
        assert run.returncode == 0, f"{KINDS[kind]} n={n}:\n{run.stdout[-2000:]}\n{run.stderr[-3000:]}"
        assert " 0 failures" in run.stdout
