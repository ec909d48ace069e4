"""Resampler (processors/spectrogram_generator.py:117-121): the oracle
and the library's host tap builder against an independent torch restatement of torchaudio 2.4.1's
published algorithm (functional._get_sinc_resample_kernel / _apply_sinc_resample_kernel), plus
known-answer properties.  torchaudio itself is not installed here, so no reference output exists:
parity for this row is pinned by formula and by properties only."""
import math

import numpy as np
import pytest
import torch

import oracle
from audio_tokens_amd.backend import HostHelpers

RATES = [(44100, 22050), (48000, 22050), (16000, 22050), (8000, 22050), (22050, 16000), (32000, 22050)]


def torch_resample(wave: torch.Tensor, orig_freq: int, new_freq: int):
    """wave [B, L] float32 -> ([B, ceil(L*new/orig)], taps [new, K]) with torch ops only."""
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    lpw, rolloff = 6, 0.99
    base = min(orig, new) * rolloff
    width = math.ceil(lpw * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lpw, lpw)
    window = torch.cos(t * math.pi / lpw / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kern = (kern * window * (base / orig)).to(torch.float32)
    L = wave.shape[-1]
    x = torch.nn.functional.pad(wave, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kern, stride=orig)
    y = y.transpose(1, 2).reshape(wave.shape[0], -1)
    return y[:, : math.ceil(new * L / orig)], kern[:, 0]


@pytest.mark.parametrize("orig_freq,new_freq", RATES)
def test_taps_match_torch_restatement(orig_freq, new_freq):
    taps, orig, new, width = HostHelpers().resample_taps(orig_freq, new_freq)
    _, kern = torch_resample(torch.zeros(1, 8), orig_freq, new_freq)
    assert taps.shape == tuple(kern.shape) == (new, 2 * width + orig)
    # double-precision sin/cos of two libms, then one rounding to float
    np.testing.assert_allclose(taps, kern.numpy(), rtol=0, atol=2e-7)


@pytest.mark.parametrize("orig_freq,new_freq", RATES)
@pytest.mark.parametrize("L", [1, 7, 1000, 22051])
def test_oracle_matches_torch_restatement(orig_freq, new_freq, L):
    rng = np.random.default_rng(L + orig_freq)
    w = rng.standard_normal(L).astype(np.float32)
    got = oracle.resample(w, orig_freq, new_freq)
    want, _ = torch_resample(torch.from_numpy(w)[None], orig_freq, new_freq)
    assert got.shape == (HostHelpers().resample_length(L, orig_freq, new_freq),) == tuple(want.shape[1:])
    np.testing.assert_allclose(got, want[0].numpy(), rtol=0, atol=2e-5)   # fp32 conv1d vs double dot


@pytest.mark.parametrize("orig_freq,new_freq", [(44100, 22050), (48000, 22050), (16000, 22050)])
def test_low_frequency_sine_is_preserved(orig_freq, new_freq):
    f = 440.0
    n = orig_freq // 2
    w = np.sin(2 * np.pi * f * np.arange(n) / orig_freq).astype(np.float32)
    y = oracle.resample(w, orig_freq, new_freq)
    want = np.sin(2 * np.pi * f * np.arange(y.shape[0]) / new_freq)
    edge = 64
    assert np.abs(y[edge:-edge] - want[edge:-edge]).max() < 2e-3


def test_equal_rates_and_dc_gain():
    w = np.ones(4096, np.float32)
    assert np.array_equal(oracle.resample(w, 22050, 22050), w)
    y = oracle.resample(w, 44100, 22050)
    assert np.abs(y[32:-32] - 1.0).max() < 2e-3            # unit DC gain away from the zero padding
