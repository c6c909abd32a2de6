"""CPU only: the product's host-side helpers under AddressSanitizer / UBSan (SURVEY section 5 "sanitizers").  Listed in
.gpurunignore -- the GPU pool refuses any call whose tree holds a test that builds with -fsanitize (GPU sanitizers are
disabled there), and nothing in this file touches a GPU."""

import os

import pytest


def test_host_side_packing_under_sanitizers(tmp_path):
    """SURVEY section 5 "sanitizers" for the product's own host code: the key packing, probe-sequence and INT4 scale-slot
    helpers that the index build, the match kernels and the table kernels share between host and device
    (scone_amd/csrc/scone_common.h), compiled for the host only with -fsanitize=address,undefined and checked for the
    properties the exact-key index relies on (tests/host_pack_check.cpp).  CPU only: hipcc cross-compiles, nothing runs on
    a GPU."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_pack_check")
    subprocess.run(["hipcc", "-x", "hip", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", os.path.join(root, "tests", "host_pack_check.cpp"),
                    "-o", exe], check=True, capture_output=True, timeout=300)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))    # the HIP runtime's own start-up allocations are not ours
    assert p.returncode == 0 and "0 problem(s)" in p.stdout, p.stdout + p.stderr[-2000:]
