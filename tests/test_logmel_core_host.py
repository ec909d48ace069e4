"""The per-lane FFT phases of audio_tokens_amd/csrc/logmel_core.h executed on the host (16 lanes
one after another) against numpy: checks the index algebra of the kernel without a GPU."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
fp = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    so = tmp_path_factory.mktemp("h") / "liblogmel_host.so"
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(so),
                    str(ROOT / "tests" / "host_harness" / "logmel_host.cpp")], check=True)
    return ctypes.CDLL(str(so))


def test_dft16(harness):
    rng = np.random.default_rng(0)
    for _ in range(10):
        v = rng.standard_normal(32).astype(np.float32)
        ref = np.fft.fft(v[0::2].astype(np.float64) + 1j * v[1::2])
        w = v.copy()
        harness.logmel_host_dft16(w.ctypes.data_as(fp))
        assert np.abs((w[0::2] + 1j * w[1::2]) - ref).max() < 1e-5


def test_frame_power_spectrum(harness):
    rng = np.random.default_rng(1)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512)).astype(np.float32)
    t = np.arange(512)
    frames = [rng.standard_normal(512), np.sin(2 * np.pi * 37.3 * t / 512), np.zeros(512), np.eye(512)[200],
              np.ones(512)]
    for f in frames:
        f = f.astype(np.float32)
        P = np.zeros(257, np.float32)
        harness.logmel_host_power(f.ctypes.data_as(fp), win.ctypes.data_as(fp), P.ctypes.data_as(fp))
        ref = np.abs(np.fft.rfft(f.astype(np.float64) * win)) ** 2
        assert np.abs(P - ref).max() <= 3e-6 * max(ref.max(), 1e-30) + 1e-30
