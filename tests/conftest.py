"""Shared pytest configuration.

* ``-m "not gpu"``: oracle vs the reference's golden vectors, host logic, C-ABI symbol check
  (no compute calls) -- runs in the CPU-only build container.
* ``-m gpu``: parity tests proper; they call the HIP path through the C-ABI on a real MI355X.
"""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_ranks: starts torch.distributed.run children that share the GPU; run as its own "
                                       "pytest process (tests/test_multirank_gpu.py)")


def _denan(x):
    if isinstance(x, str) and x == "nan":
        return math.nan
    if isinstance(x, list):
        return [_denan(v) for v in x]
    if isinstance(x, dict):
        return {k: _denan(v) for k, v in x.items()}
    return x


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return _denan(json.load(f))
