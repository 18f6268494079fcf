import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # The oracle is torch-CPU: hundreds of small ops per texture.  On the GPU box (256 cores, torch defaults to 128
    # threads) their fork / join dominates: tests/test_parity_report.py [5-4-128] takes 99 s with the default and
    # 9.7 s with 16 threads (profiles/r06/README.md) — the reference's own scripts assume 16 (scripts/volsurfs.sh:47).
    # (oracle/raytrace_ref.c asks for every core itself.)
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 16))


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
