import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """The native pieces are built artefacts kept out of git: a fresh checkout has none.  Build what is missing or stale before the
    first test asks for it (no-ops when current: content stamps; the same calls as __graft_entry__.build())."""
    from pf3plat_amd import _lib

    try:
        _lib.build()
        _lib.build_torch_ext()
    except Exception as e:  # (no compiler on this machine: the tests that need the libraries will say so themselves)
        print(f"[conftest] native build skipped: {type(e).__name__}: {str(e)[:300]}", file=sys.stderr)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import load_oracle

    return load_oracle()


@pytest.fixture()
def oracle_backend():
    """Install the oracle-backed test backend behind pf3plat_amd's wrappers (CPU tensors)."""
    from tests.oracle_backend import OracleBackend
    from tests.util import install_backend

    be = OracleBackend()
    old = install_backend(be)
    yield be
    install_backend(old)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The give in the parity tolerance, made visible: per test, the pixels beyond 1e-4, the threshold flips and the gradient
    rows set aside (tests/parity_checks.py REPORT).  Also written to gpurun_out/parity_counts.json."""
    import json

    from tests import parity_checks

    rep = parity_checks.REPORT
    if not rep:
        return
    tr = terminalreporter
    tr.write_sep("-", "parity counts (pixels > 1e-4 | final-T flips | gradient rows set aside | worst rel-L2 over ALL rows)")
    for test, row in rep.items():
        aside = sum(v for k, v in row.items() if k.endswith("_set_aside"))
        tr.write_line(f"{test.split('::')[-1][:70]:70s} outlier_px {row.get('outlier_pixels_1e-4', 0):5d}  T_flips {row.get('final_T_flipped_pixels', 0):5d}  "
                      f"flipped_px {row.get('flipped_pixels', 0):5d}  set_aside {aside:5d}  img {row.get('color_rel_l2_all', 0.0):.2e}  "
                      f"grad {row.get('grad_rel_l2_all_max', 0.0):.2e}")
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_counts.json"), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
