import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def restated_host_libm():
    """Skips the caller unless the libm of THIS host is the one LTP_POW_LIBM restates (glibc >= 2.28, FMA variant): the oracle is
    compiled against the installed libm, so "the device has the oracle's BITS" can only hold where the two are the same library.
    (ltp_host_libm_pow_rule: 2^18 planner-sized powers through the installed pow; the 1e-9 parity tests do not depend on this.)"""
    from longtermplanner_amd import LongTermPlanner
    rule, n_libm, n_exact = LongTermPlanner.powRuleMatchingHostLibm()
    if rule != "libm":
        pytest.skip(f"this host's libm is not the glibc variant LTP_POW_LIBM restates ({n_libm} of 262144 probe powers differ; "
                    f"{n_exact} differ from the correctly rounded rule): bit-identity with a libm oracle is not defined here")
    return True
