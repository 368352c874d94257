"""pytest configuration: markers and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks (CPU only).
`-m gpu`       : parity of the HIP path (through the C-ABI) against the oracle, on an MI355X.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """the native pieces must exist before any test runs: build them if a fresh checkout has none
    (hipcc cross-compiles gfx950 without a GPU; on the GPU box the prebuilt files travel with the tree)"""
    import __graft_entry__ as ge
    lib = os.path.join(ROOT, "nvbio-gpl_amd", "lib", "libnvbio_amd.so")
    orc_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc_so)):
        ge.build()


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """the reference's own host code; only exists in the development container"""
    import oracle
    if not oracle.Reference.available():
        pytest.skip("oracle/_ref not built (reference sources are not on this machine)")
    return oracle.Reference()


@pytest.fixture(scope="session")
def fm_golden():
    return np.load(os.path.join(GOLDEN, "fm_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def dp_golden():
    return np.load(os.path.join(GOLDEN, "dp_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def ed_golden():
    return np.load(os.path.join(GOLDEN, "ed_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def bt_golden():
    return np.load(os.path.join(GOLDEN, "bt_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def best2_golden():
    return np.load(os.path.join(GOLDEN, "best2_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def myers_golden():
    return np.load(os.path.join(GOLDEN, "myers_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def swtb_golden():
    return np.load(os.path.join(GOLDEN, "swtb_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def fswtb_golden():
    return np.load(os.path.join(GOLDEN, "fswtb_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def staged_golden():
    return np.load(os.path.join(GOLDEN, "staged_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def rankdict_golden():
    return np.load(os.path.join(GOLDEN, "rankdict_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def deque_golden():
    return np.load(os.path.join(GOLDEN, "deque_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def sw_golden():
    return np.load(os.path.join(GOLDEN, "sw_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def ftb_golden():
    return np.load(os.path.join(GOLDEN, "ftb_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def tb_golden():
    return np.load(os.path.join(GOLDEN, "tb_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def amd():
    """the product package (HIP kernels behind the C-ABI); GPU tests only"""
    import __graft_entry__ as ge
    return ge.load_package()
