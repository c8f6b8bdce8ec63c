import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    if os.environ.get("EAT_POISON_EMPTY") == "1":
        # debugging aid: every torch.empty / empty_like is filled with NaN, so a kernel that reads memory it (or its producer)
        # never wrote turns a test red instead of depending on what the allocator happened to hand out (this is how the
        # unordered memset nodes of the captured DyMN step were found: DESIGN 5, "Correctness fix of round 6")
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
