import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # tests may use the oracle (checker only)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# The library sends launch groups of fewer than 96 blocks to the tiled kernels (a block kernel gives one compute unit a whole
# block; fdc_api.hip, kBlockMinBlocks).  The parity tests use a handful of blocks and are about the block kernels too:
# they run with the threshold at 1; test_short_calls_take_the_tiled_kernels checks the default.
import gr_fdc_amd as _G                                # noqa: E402  (does not load the library yet)
_G.defaults.setdefault("FDC_BLOCK_MIN_BLOCKS", "1")     # -> fdc_pipeline_cfg.min_block_launch of every pipeline the tests create
for _k in ("FDC_FORCE_GENERIC", "FDC_NO_POLY", "FDC_NO_BLOCK", "FDC_NO_FUSED"):     # the whole suite under a forced path: FDC_TEST_FORCE=FDC_NO_POLY pytest ...
    if os.environ.get("FDC_TEST_FORCE") == _k:
        _G.defaults[_k] = "1"


@pytest.fixture(autouse=True)
def _defaults_are_per_test():
    """gr_fdc_amd.defaults as every test found it: a test that forces a path for a comparison (sets a key, deletes it afterwards) would otherwise take the
    key of a forced-path run of the SUITE (FDC_TEST_FORCE) with it, and everything behind it would run unforced."""
    saved = dict(_G.defaults)
    yield
    _G.defaults.clear()
    _G.defaults.update(saved)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Timing assertions run behind every parity test: test_plan_choice_gpu.py compares HIP-event timings (10 %), and the driver's GPU run
    stops at the first failure (-x) — a noisy box must not end the run in front of the tests that are about results."""
    items.sort(key=lambda it: 1 if "test_plan_choice_gpu" in it.nodeid else 0)      # stable: everything else keeps its order


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O
