import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_PARITY_LINES = []


def parity_report(line: str):
    """Measured-error lines of the parity tests: printed as one table in the terminal summary (also under -q, where
    passing tests' stdout is swallowed) and written to gpurun_out/parity_report.txt on the GPU box."""
    _PARITY_LINES.append(line)
    print(line)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _PARITY_LINES:
        return
    terminalreporter.section("parity report (measured errors)")
    for l in _PARITY_LINES:
        terminalreporter.write_line(l)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.txt"), "w") as f:
            f.write("\n".join(_PARITY_LINES) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
