import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# launch breadcrumbs of the HIP library (csrc/devmem.hip): if a GPU memory fault aborts the process, the last kernels launched
# are printed behind the runtime's message - the record of a red run names a kernel
os.environ.setdefault("UGVC_BREADCRUMB", "1")


_CONFIG = None


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_runtest_logstart(nodeid, location):
    # GPU runs only (UGVC_TEST_TRACE=1, or a visible device): the name of every test goes to the REAL stderr, past pytest's
    # capture and unbuffered - a process the runtime aborts (rc 134) still leaves the name of the test it died in
    if os.environ.get("UGVC_TEST_TRACE", "1" if os.path.exists("/dev/kfd") else "0") == "0" or _CONFIG is None:
        return
    capman = _CONFIG.pluginmanager.getplugin("capturemanager")
    if capman is None:
        return
    with capman.global_and_fixture_disabled():
        sys.stderr.write(f"[test] {nodeid}\n")
        sys.stderr.flush()


@pytest.fixture(scope="session")
def frozen_models():
    from variantcalling_amd import model_io
    return model_io.load_models(os.path.join(GOLDEN, "synth_rf_v1.npz"))


@pytest.fixture(scope="session")
def small_callset():
    from variantcalling_amd import synth
    return synth.make_callset(60_000, genome_len=30_000_000, n_contigs=5, seed=1234)


@pytest.fixture(scope="session")
def engine():
    """One GPU context for the session; the HIP library is the only path (no fallback)."""
    from variantcalling_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


def real_chr1_reference():
    """Real hg38 chr1:1-5,000,000 (with its N runs) + chr20 sample as a 2-contig reference."""
    from variantcalling_amd import schema as S, synth
    a = synth.load_hg38_slice("chr1")
    b = synth.load_hg38_slice("chr20")
    codes = np.concatenate([a, b])
    return S.Reference(codes, np.array([0, a.size, a.size + b.size], dtype=np.int64), ["chr1", "chr20"])
