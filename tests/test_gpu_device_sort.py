"""GPU: csrc/device_sort.h -- the library's own stable radix sort and exclusive scan (they replaced hipcub in round 5 and carry the
device-side leaf planner) against std::stable_sort and a serial scan, compiled here with hipcc (tests/device_sort_check.hip)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_radix_sort_and_scan_against_the_standard_library(tmp_path):
    exe = str(tmp_path / "device_sort_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "device_sort_check.hip"), "-o", exe], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), p.stdout[-800:] + p.stderr[-400:]
