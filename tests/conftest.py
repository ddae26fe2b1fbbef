import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from cpu_checkers import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from cpu_checkers import Ref, ref_available
    if not ref_available():
        pytest.skip("compiled reference (oracle/_ref/libmlref.so) not available here")
    return Ref()


def pytest_terminal_summary(terminalreporter):
    """The hardware-approximate operations (divideApprox, sqrtApprox, Peak, RMS) are compared under a tolerance; say what was measured."""
    try:
        from inputs import REL_MEASURED
    except Exception:
        return
    if REL_MEASURED:
        worst = max(REL_MEASURED.values())
        terminalreporter.write_line(f"toleranced comparisons: largest relative difference seen {worst:.3e} = 2^{__import__('math').log2(worst) if worst > 0 else float('-inf'):.2f} "
                                    f"(tolerance 2^-11 = {2.0 ** -11:.3e}); per label: "
                                    + ", ".join(f"{k} {v:.2e}" for k, v in sorted(REL_MEASURED.items(), key=lambda kv: -kv[1])[:8]))
