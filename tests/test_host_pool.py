"""csrc/host_pool.hpp (the thread pool behind the chunk pipeline of ugvc_filter_variants): a native stress test, built here
with g++ - every task of every job runs exactly once, also when jobs follow each other faster than workers leave them."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
@pytest.mark.parametrize("threads", [2, 8, 16])
def test_pool_runs_every_task_exactly_once(tmp_path, threads):
    exe = str(tmp_path / "host_pool_stress")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "host_pool_stress.cpp"), "-o", exe],
                   check=True)
    r = subprocess.run([exe, str(threads), "3000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_copy_moves_every_byte_and_no_more(tmp_path):
    """csrc/host_copy.hpp (the CPU side of every pinned-slot copy, devmem.hip): sizes around the threading threshold and around
    multiples of 64 x threads, 1 ... 7 threads - the first version lost the last n % 4 bytes of a 6 239 490-byte upload."""
    exe = str(tmp_path / "host_copy_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "host_copy_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-2000:] + r.stderr
