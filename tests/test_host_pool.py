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
