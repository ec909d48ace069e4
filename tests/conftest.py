import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the test-suite runs the library in its strict mode: a HIP error that is pending in the calling thread when the library
# is about to launch a kernel -- somebody swallowed a failing call -- fails the test (include/at_debug.h: strict_errors)
os.environ.setdefault("AT_STRICT_ERRORS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def be():
    """The product backend (HIP only).  GPU tests call through the C ABI with it."""
    from audio_tokens_amd.backend import default_backend
    return default_backend()


_NATIVE_DEFAULTS = {"assign_variant": 0, "filter_fused": 1, "filter_sync": 0, "prune_kernel": 1, "prune_nb": 0,
                    "filter_screen": 1, "filter_nb": 0, "filter_wps2": 0, "dmin_kernel": 1, "resample_simple": 0,
                    "accum_buckets": 1, "filter_stats": 0, "visit_bits": 8, "filter_timing": 0}


@pytest.fixture()
def switches(be):
    """Sets A/B switches for one test -- native ones through at_debug_set, host-side ones in be.switches -- and puts
    the defaults back afterwards.  (The library reads the AT_* environment only once, in at_create.)"""
    host_before = dict(be.switches)

    def set_(**kw):
        for name, value in kw.items():
            if name in be.switches:
                be.switches[name] = bool(value)
            else:
                be.debug_set(name, value)
    yield set_
    be.switches.update(host_before)
    for name, value in _NATIVE_DEFAULTS.items():
        be.debug_set(name, value)
