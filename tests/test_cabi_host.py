"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares, and
its host helpers (no GPU involved) agree with the oracle."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from audio_tokens_amd import _lib
    declared = set()
    for name in ("audio_tokens_amd.h", "at_debug.h"):       # the operator surface, and the test / A-B hooks
        header = (ROOT / "include" / name).read_text()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # (comments mention functions too)
        found = set(re.findall(r"\b(at_[a-z0-9_]+)\s*\(", header)) - {"at_ctx"}
        assert found, f"no declarations parsed in {name}"
        declared |= found
    op_header = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "audio_tokens_amd.h").read_text(), flags=re.S)
    for hook in ("at_filter_probe_f32", "at_filter_stats", "at_prune_stats", "at_debug_set"):
        assert hook not in op_header, f"{hook} belongs in at_debug.h, not in the operator surface"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    # and the ctypes table binds exactly that set
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().at_version() == 100


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from audio_tokens_amd.backend import HipBackend, default_backend
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipBackend()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        default_backend()


def test_product_never_imports_the_oracle():
    for p in (ROOT / "audio_tokens_amd").rglob("*.py"):
        text = p.read_text()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f"{p} imports the oracle"
    for p in list((ROOT / "audio_tokens_amd" / "csrc").glob("*")):
        if p.suffix in (".hip", ".cpp", ".h"):
            assert "oracle" not in p.read_text().replace("oracle/oracle.c orc_assign", ""), p


@pytest.fixture(scope="module")
def host():
    from audio_tokens_amd.backend import HostHelpers
    return HostHelpers()


@pytest.mark.parametrize("n,seed", [(0, 1234), (1, 1234), (2, 1235), (1000, 1234), (200003, 1235)])
def test_rand_perm_matches_oracle_and_libstdcxx(host, oracle, n, seed):
    p = host.rand_perm(n, seed)
    assert np.array_equal(p, oracle.rand_perm(n, seed))
    assert np.array_equal(p, oracle.std_rand_perm(n, seed))


@pytest.mark.parametrize("n,m", [(10, 0), (10, 10), (1000, 37), (300000, 4096)])
def test_rand_perm_prefix(host, oracle, n, m):
    assert np.array_equal(host.rand_perm_prefix(n, 1234, m), oracle.rand_perm(n, 1234)[:m])


@pytest.mark.parametrize("n_mels", [64, 128, 40])
def test_mel_filterbank(host, oracle, n_mels):
    fb = host.mel_filterbank(22050, 512, n_mels)
    assert np.array_equal(fb, oracle.mel_filterbank(22050, 512, n_mels))
    if n_mels == 64:
        assert (fb > 0).sum() == 499 and (fb > 0).sum(0).max() == 22
    if n_mels == 128:
        assert ((fb > 0).sum(0) == 0).sum() == 3                            # three empty filters


def test_num_frames(host):
    assert host.num_frames(220500, 128) == 1723
    assert host.num_frames(22050, 128) == 173


def test_split_clusters_matches_oracle(host, oracle):
    rng = np.random.default_rng(0)
    k, d, n = 200, 16, 50000
    h = rng.integers(0, 600, k).astype(np.float32)
    h[rng.choice(k, 40, replace=False)] = 0
    h[0] = 1.0  # a donor candidate with probability 0
    n = int(h.sum())
    c = rng.standard_normal((k, d)).astype(np.float32)
    c[h == 0] = 0
    ns_o, h_o, c_o = oracle.split_clusters(h, c, n)
    h2, c2 = h.copy(), c.copy()
    ns = host.split_clusters(h2, c2, n)
    assert ns == ns_o == 40
    assert np.array_equal(h2, h_o)
    assert np.array_equal(c2.view(np.uint32), c_o.view(np.uint32))
    # no empty cluster: nothing happens, no random numbers drawn
    h3 = np.ones(k, np.float32)
    c3 = c.copy()
    assert host.split_clusters(h3, c3, k) == 0 and np.array_equal(c3, c)
