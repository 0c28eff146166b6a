import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# The stem-kernel tests of rounds 3-5 are tests OF THE bf16 x 3 ARITHMETIC (kernel names, limb counts, "at or below the
# fp32 kernel's error" bounds that hold for an exact three-way split): they pin it.  The arithmetic a new executor
# takes by default since round 6 -- two fp16 limbs, three products -- has its own tests (tests/test_gpu_round6.py).
BF16X3_MODULES = ("test_gpu_round3", "test_gpu_round4", "test_gpu_round5")


@pytest.fixture(autouse=True)
def _pin_bf16x3_for_the_tests_of_that_arithmetic(request, monkeypatch):
    if request.module.__name__.split(".")[-1] in BF16X3_MODULES and "CTG_STEM_ARITH" not in os.environ:
        monkeypatch.setenv("CTG_STEM_ARITH", "bf16x3")
    yield
