"""CPU stand-in for audio_tokens_amd.backend.HipBackend, built on the oracle.  TEST ONLY.

It exists so that the HOST logic of audio_tokens_amd.ops (FAISS training recipe, subsample
sharding, the per-iteration exchange under torch.distributed) can run in this GPU-less container
and under gloo with world_size 2.  The product never imports it; the sequential host helpers
(rand_perm, split_clusters) are the product's own native ones."""
import numpy as np
import torch

import oracle
from audio_tokens_amd.backend import HostHelpers


class OracleBackend(HostHelpers):
    def __init__(self):
        super().__init__()
        self.device = torch.device("cpu")
        self.assign_trace = None

    def _f32(self, t, name="tensor"):
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        return t.float().contiguous()

    def empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype)

    def zeros(self, shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype)

    def from_host(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dtype) if dtype is not None else t

    def to_host(self, t):
        return t.detach().numpy()

    def host_staging(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def synchronize(self):
        pass

    def record_event(self):
        class _Done:
            def synchronize(self):
                pass
        return _Done()

    def resample(self, wave, orig_freq, new_freq):
        w = self._f32(wave)
        if w.dim() == 1:
            return torch.from_numpy(oracle.resample(w.numpy(), orig_freq, new_freq))
        return torch.from_numpy(np.stack([oracle.resample(r.numpy(), orig_freq, new_freq) for r in w]))

    def l2norm_rows(self, x, out=None):
        return torch.from_numpy(oracle.l2norm_rows(self._f32(x).numpy()))

    def assign(self, x, c, want_dist=True):
        ids, dis = oracle.assign(self._f32(x).numpy(), self._f32(c).numpy())
        return torch.from_numpy(ids), (torch.from_numpy(dis) if want_dist else None)

    def assign_hinted(self, x, c, hint_ids, order=None, want_dist=True):
        return self.assign(x, c, want_dist)          # hints never change the answer

    def gather_rows(self, x, idx):
        idx = idx if isinstance(idx, np.ndarray) else idx.numpy()
        return self._f32(x)[torch.from_numpy(idx.astype(np.int64))].contiguous()

    def centroid_accum(self, x, ids, k, out=None, want_order=False):
        x = self._f32(x).numpy()
        d = x.shape[1]
        sums = np.zeros((k, d), np.float32)
        counts = np.zeros(k, np.float32)
        np.add.at(sums, ids.numpy(), x)          # unbuffered, ascending i: the sequential fp32 sum
        np.add.at(counts, ids.numpy(), np.float32(1))
        part = torch.from_numpy(np.concatenate([sums.ravel(), counts]))
        if out is not None:
            out[: k * d + k] = part
            part = out
        return (part, None) if want_order else part

    def rand_perm_prefix_device(self, n, seed, m):
        return torch.from_numpy(self.rand_perm_prefix(n, seed, m))

    def split_clusters_device(self, hassign, cent, n, nsplit_out):
        h, c = hassign.numpy(), cent.numpy()       # (views: updated in place, as the device kernel does)
        nsplit_out[0] = self.split_clusters(h, c, n) if (h == 0).any() else 0

    def sum_parts(self, parts):
        tot = np.zeros(parts.shape[1], np.float32)
        for p in parts.numpy():                  # ascending rank order
            tot = tot + p
        return torch.from_numpy(tot)

    def lloyd_stats(self, hassign, parts, k, d, stats_row, objs=None):
        off, total = self.part_layout(k, d)
        obj = 0.0
        if objs is not None:
            for o in objs.tolist():
                obj += o
        else:
            for p in parts.reshape(-1, total):   # ascending rank order
                obj += float(p[off:off + 2].view(torch.float64)[0])
        h = hassign.double().numpy()
        stats_row[0] = obj
        stats_row[1] = float((h * h).sum() * k / (h.sum() ** 2))

    def to_host_async(self, t):
        class _Done:
            def synchronize(self):
                pass
        return t.clone(), _Done()

    def centroid_finalize(self, parts, k, d):
        parts = parts.reshape(-1, self.part_layout(k, d)[1]).numpy()
        tot = np.zeros(k * d, np.float32)
        cnt = np.zeros(k, np.float32)
        for p in parts:                          # ascending rank order
            tot = tot + p[: k * d]
            cnt = cnt + p[k * d: k * d + k]
        cent = tot.reshape(k, d).copy()
        nz = cnt != 0
        cent[nz] = cent[nz] * (np.float32(1.0) / cnt[nz])[:, None]
        return torch.from_numpy(cent), torch.from_numpy(cnt)

    def sum_f64(self, v, out=None):
        r = torch.tensor([float(self._f32(v).double().sum())], dtype=torch.float64)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def any_nonfinite(self, v):
        return not bool(torch.isfinite(self._f32(v)).all())
