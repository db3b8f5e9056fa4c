import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "data")
LW_FSCK = os.path.join(DATA, "ecckd-1.2_lw_ckd-definition_climate_fsck-tol0.0161.nc")
LW_RRTMGP = os.path.join(DATA, "ecckd-1.2_lw_ckd-definition_climate_rrtmgp-tol0.061.nc")
SW_WIDE = os.path.join(DATA, "ecckd-1.2_sw_ckd-definition_climate_wide-tol0.05.nc")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    """The product package with its HIP library built (hipcc cross-compiles without a GPU)."""
    import rte_ecckd_amd as p
    p.build()
    return p


@pytest.fixture(scope="session")
def gpu(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
