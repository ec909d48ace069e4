"""Synthetic 22.05 kHz clips for benchmarks and tests (there is no dataset in the build or on the GPU
box).  Counter-based: clip i depends only on (seed, i), so any rank can generate exactly its shard.

Recipe: 1-4 sinusoids/chirps with f ~ LogU(50, 10 000) Hz and amplitude
~ U(0.05, 0.5), white noise with sigma ~ LogU(1e-4, 1e-1), and one exact-zero segment of random
length (digital silence: exercises the 1e-10 clamp / constant -100 dB frames).
This is input generation, not part of the measured path; it uses torch ops on whatever device it is
given.
"""
from __future__ import annotations

import numpy as np
import torch

_M32 = 0xFFFFFFFF


def _mix(x: torch.Tensor) -> torch.Tensor:
    """32-bit finaliser (murmur3 fmix32) on int64 tensors holding uint32 values."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    x = x ^ (x >> 16)
    return x


def clip_params(seed: int, clip_ids: np.ndarray, noisy: bool = False):
    """Per-clip parameters from a numpy Generator keyed by (seed, clip id).  noisy: the harder stream of bench.py's
    `hard_workload` -- the same draws, with the tones at a fifth of their amplitude under white noise of
    sigma ~ LogU(0.05, 0.5): frames that cluster poorly."""
    n = len(clip_ids)
    ntone = np.zeros(n, np.int64)
    f0 = np.zeros((n, 4)); f1 = np.zeros((n, 4)); amp = np.zeros((n, 4)); ph = np.zeros((n, 4))
    sigma = np.zeros(n); z0 = np.zeros(n); zlen = np.zeros(n)
    for j, cid in enumerate(clip_ids):
        r = np.random.default_rng([int(seed), int(cid)])
        ntone[j] = r.integers(1, 5)
        f0[j] = np.exp(r.uniform(np.log(50.0), np.log(10000.0), 4))
        chirp = r.random(4) < 0.5
        f1[j] = np.where(chirp, np.exp(r.uniform(np.log(50.0), np.log(10000.0), 4)), f0[j])
        amp[j] = r.uniform(0.05, 0.5, 4) * (np.arange(4) < ntone[j])
        ph[j] = r.uniform(0, 2 * np.pi, 4)
        sigma[j] = np.exp(r.uniform(np.log(1e-4), np.log(1e-1)))
        z0[j] = r.random()
        zlen[j] = r.random() * 0.04  # ~2 % of all frames are digital silence (identical points)
        if noisy:
            amp[j] *= 0.2
            sigma[j] = np.exp(np.log(0.05) + (np.log(sigma[j]) - np.log(1e-4)) / (np.log(1e-1) - np.log(1e-4)) * (np.log(0.5) - np.log(0.05)))
    return dict(f0=f0, f1=f1, amp=amp, ph=ph, sigma=sigma, z0=z0, zlen=zlen)


def synth_clips(n_clips: int, L: int = 220500, seed: int = 4242, first_clip: int = 0, sr: int = 22050,
                device="cpu", chunk: int = 256, out: torch.Tensor | None = None, noisy: bool = False) -> torch.Tensor:
    """-> float32 [n_clips, L] in [-1, 1] on `device`."""
    device = torch.device(device)
    if out is None:
        out = torch.empty((n_clips, L), dtype=torch.float32, device=device)
    t = torch.arange(L, device=device, dtype=torch.float64) / sr
    dur = L / sr
    idx = torch.arange(L, device=device, dtype=torch.int64)
    for c0 in range(0, n_clips, chunk):
        c1 = min(n_clips, c0 + chunk)
        ids = np.arange(first_clip + c0, first_clip + c1)
        p = clip_params(seed, ids, noisy)
        P = {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in p.items()}
        w = torch.zeros((c1 - c0, L), dtype=torch.float64, device=device)
        for j in range(4):
            f0, f1 = P["f0"][:, j:j + 1], P["f1"][:, j:j + 1]
            phase = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / dur * t * t) + P["ph"][:, j:j + 1]
            w += P["amp"][:, j:j + 1] * torch.sin(phase)
        # counter-based gaussian noise: two hashed uniforms per sample -> Box-Muller
        cid = torch.from_numpy(ids).to(device).unsqueeze(1)
        key = _mix(cid * 0x9E3779B1 + seed)
        h1 = _mix(idx.unsqueeze(0) * 2 + 1 + key * 0x632BE5AB)
        h2 = _mix(idx.unsqueeze(0) * 2 + 2 + key * 0x7F4A7C15)
        u1 = (h1.double() + 1.0) / 4294967297.0
        u2 = h2.double() / 4294967296.0
        w += P["sigma"].unsqueeze(1) * torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2 * np.pi * u2)
        w.clamp_(-1.0, 1.0)
        # exact-zero segment
        zs = (P["z0"] * L).long().unsqueeze(1)
        ze = zs + (P["zlen"] * L).long().unsqueeze(1)
        w = torch.where((idx.unsqueeze(0) >= zs) & (idx.unsqueeze(0) < ze), torch.zeros_like(w), w)
        out[c0:c1] = w.float()
    return out
