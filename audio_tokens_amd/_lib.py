"""ctypes binding of libaudio_tokens_amd.so (the C ABI declared in include/audio_tokens_amd.h).

There is no fallback of any kind: if the shared library is missing, or the device is not a
gfx950, every operator raises.  Tensors are PyTorch-ROCm tensors used as storage only -- the
library receives `data_ptr()` and the current HIP stream handle.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libaudio_tokens_amd.so"

AT_LAYOUT_MEL_MAJOR = 0
AT_LAYOUT_FRAME_MAJOR = 1

_c = ctypes
_vp, _i32, _i64 = _c.c_void_p, _c.c_int, _c.c_int64

class PrunedArgs(_c.Structure):
    """struct at_pruned_args of include/audio_tokens_amd.h."""
    _fields_ = [("x", _vp), ("n", _i64), ("d", _i32), ("c", _vp), ("k", _i32), ("order", _vp), ("hint_sorted", _vp),
                ("cperm", _vp), ("ng", _i32), ("bounds", _vp), ("guess_only", _i32), ("use_filter", _i32),
                ("prepass_done", _i32), ("image_current", _i32), ("ids", _vp), ("dist_or_null", _vp)]


# name -> (restype, argtypes); kept in one table so tests can check every declared symbol exports
SIGNATURES = {
    "at_version": (_i32, []),
    "at_last_error": (_c.c_char_p, []),
    "at_create": (_i32, [_i32, _c.POINTER(_vp)]),
    "at_destroy": (None, [_vp]),
    "at_workspace_bytes": (_i64, [_vp]),
    "at_background_stream": (_i32, [_vp, _c.POINTER(_vp)]),
    "at_debug_set": (_i32, [_vp, _c.c_char_p, _i32]),
    "at_debug_get": (_i32, [_vp, _c.c_char_p, _c.POINTER(_i32)]),
    "at_diag_errors": (_i32, [_c.POINTER(_i64), _c.POINTER(_i32), _c.POINTER(_i64), _c.POINTER(_i32), _c.c_char_p, _i32, _i32]),
    "at_debug_leave_error_pending": (_i32, []),
    "at_rand_perm_mt19937": (_i32, [_i64, _i64, _vp]),
    "at_rand_perm_prefix_mt19937": (_i32, [_i64, _i64, _i64, _vp]),
    "at_rand_perm_prefix_device": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "at_mel_filterbank_host": (_i32, [_i32, _i32, _i32, _vp]),
    "at_num_frames": (_i64, [_i64, _i32]),
    "at_split_clusters_host": (_i32, [_i32, _i32, _i64, _vp, _vp, _c.POINTER(_i32)]),
    "at_logmel_f32": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "at_resample_length": (_i64, [_i64, _i32, _i32]),
    "at_resample_taps_host": (_i32, [_i32, _i32, _c.POINTER(_i32), _c.POINTER(_i32), _c.POINTER(_i32), _vp, _i64]),
    "at_resample_f32": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "at_minmax_scale_clips_f32": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "at_conv1d_mel_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "at_logmel_minmax_f32": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "at_l2norm_rows_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "at_assign_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp]),
    "at_assign_hinted_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "at_group_rows_kd_host": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "at_group_min_dist_f32": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "at_visit_order_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "at_prune_stats": (_i32, [_vp, _c.POINTER(_i64), _c.POINTER(_i64), _i32]),
    "at_assign_coarse_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "at_filter_stats": (_i32, [_vp, _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_c.c_double), _c.POINTER(_i64),
                               _c.POINTER(_i64), _c.POINTER(_i64), _i32]),
    "at_filter_probe_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                   _c.POINTER(_i64), _vp]),
    "at_group_means_f32": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "at_group_neighbours_f32": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "at_assign_pruned_f32": (_i32, [_vp, _vp, _vp]),   # (ctx, const at_pruned_args*, stream)
    "at_prune_mask_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp]),
    "at_gather_rows_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _vp, _vp]),
    "at_centroid_accum_f32": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "at_centroid_finalize_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "at_split_clusters_f32": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "at_lloyd_stats_f64": (_i32, [_vp, _vp, _i32, _vp, _i64, _i32, _vp, _vp]),
    "at_comm_allgather_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "at_comm_allreduce_ordered_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "at_lloyd_stats_split_f32": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "at_sum_parts_f32": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "at_sum_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "at_any_nonfinite_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "at_logmel_nonfinite_take": (_i32, [_vp, _vp, _vp]),
    "at_centroid_accum_defer": (_i32, [_vp, _i32]),
    "at_centroid_accum_join": (_i32, [_vp, _vp]),
    "at_token_histogram_i64": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "at_token_stats_f64": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
}

_lib = None


class NativeLibraryError(ImportError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises NativeLibraryError if it is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {_PKG / 'csrc'}`). "
                "audio_tokens_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    return load().at_last_error().decode("utf-8", "replace")


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"audio_tokens_amd native error {code}: {msg}")
        self.code = code


def check(rc: int) -> None:
    if rc != 0:
        raise NativeError(rc, last_error())


class Context:
    """One at_ctx (per-device workspace).  Not re-entrant: one per host thread."""

    def __init__(self, device_index: int):
        self.lib = load()
        self.device_index = int(device_index)
        h = _vp()
        check(self.lib.at_create(self.device_index, ctypes.byref(h)))
        self.handle = h

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.at_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self) -> int:
        return int(self.lib.at_workspace_bytes(self.handle))


_contexts: dict[int, Context] = {}


def context(device_index: int) -> Context:
    ctx = _contexts.get(device_index)
    if ctx is None:
        ctx = _contexts[device_index] = Context(device_index)
    return ctx
