"""ctypes binding of the gfx950 tokenizer library (include/lipvq.h).

This is the stub a maintainer of the reference would add next to
robomimic/models/vq_vae/backbone_lfqvae_v5.py: it passes raw device
pointers and the current HIP stream; PyTorch keeps ownership of every buffer.

There is NO fallback: if the shared object is missing the import of the product fails, and a
non-zero status from the library raises RuntimeError with the library's message.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# torch ships its own libamdhip64; it must be mapped BEFORE our library so that the loader binds
# our NEEDED libamdhip64.so.7 to that same runtime (one HIP runtime per process: device pointers
# and streams handed over by PyTorch are only meaningful to the runtime that created them).
import torch  # noqa: F401  (import order matters)

import os as _os

# LIPVQ_HIP_LIBRARY: development hook (A/B builds of the same library, scripts/ab_build.sh); the product loads the in-tree build
_LIB_PATH = Path(_os.environ.get("LIPVQ_HIP_LIBRARY") or (Path(__file__).resolve().parent / "_lipvq_hip.so"))

ACT_NONE, ACT_GELU, ACT_SIGMOID, ACT_RELU = 0, 1, 2, 3
DIST_NORM, DIST_SQSUM = 0, 1
ABI_VERSION = 1

_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

# name -> (restype, argtypes); mirrors include/lipvq.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "lipvq_abi_version": (_i, []),
    "lipvq_last_error": (C.c_char_p, []),
    "lipvq_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "lipvq_get_option": (C.c_char_p, [C.c_char_p]),
    "lipvq_lipschitz_scale_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "lipvq_mlp3_packed_floats": (_sz, [_i, _i, _i, _i]),
    "lipvq_mlp3_pack_f32": (_i, [_vp] * 7 + [_i] * 4 + [_vp]),
    "lipvq_mlp3_f32": (_i, [_vp] * 7 + [_i64] + [_i] * 7 + [_vp]),
    "lipvq_nearest_f32": (_i, [_vp] * 6 + [_i64, _i, _i, _i, _vp]),
    "lipvq_ste_f32": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "lipvq_mse_workspace_bytes": (_sz, []),
    "lipvq_mse_pair_f32": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "lipvq_mse_pair_loss_f32": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, C.c_float, _i, _vp, _vp]),
    "lipvq_nearest_prep_bytes": (_sz, [_i, _i]),
    "lipvq_nearest_prepare_f32": (_i, [_vp, _vp, _i, _i, _vp]),
    "lipvq_nearest_screened_supported": (_i, [_i, _i]),
    "lipvq_nearest_workspace_bytes": (_sz, [_i64]),
    "lipvq_screen_is_coarse": (_i, [_i, _i]),
    "lipvq_nearest_screened_f32": (_i, [_vp] * 7 + [_i64, _i, _i, _vp]),
    "lipvq_nearest_rows_f32": (_i, [_vp] * 5 + [_i64, _i, _i, _vp]),
    "lipvq_vq_nearest_screened_f32": (_i, [_vp] * 7 + [_i64, _i, _i, _vp]),
    "lipvq_vq_nearest_rows_f32": (_i, [_vp] * 5 + [_i64, _i, _i, _vp]),
    "lipvq_screen_debug_f32": (_i, [_vp] * 8 + [C.c_float, _i64, _i, _i, _vp]),
    "lipvq_tokenize_supported": (_i, [_i] * 5),
    "lipvq_tokenize_fast_supported": (_i, [_i] * 5),
    "lipvq_tokenize_workspace_bytes": (_sz, [_i64, _i]),
    "lipvq_tokenize_workspace_init": (_i, [_vp, _vp]),
    "lipvq_tokenize_f32": (_i, [_vp] * 10 + [_i64] + [_i] * 5 + [_vp]),
    "lipvq_tokenize_tune_f32": (_i, [_vp] * 10 + [_i64] + [_i] * 5 + [_vp, _i, _vp, _vp]),
    "lipvq_nearest_small_supported": (_i, [_i64, _i, _i]),
    "lipvq_nearest_small_workspace_bytes": (_sz, [_i64, _i]),
    "lipvq_nearest_small_f32": (_i, [_vp] * 6 + [_i64, _i, _i, _i, _vp]),
    "lipvq_vq_tokenize_f32": (_i, [_vp] * 9 + [_i64] + [_i] * 5 + [_vp]),
    "lipvq_vq_tokenize_train_f32": (_i, [_vp] * 12 + [_i64] + [_i] * 5 + [_vp]),
    "lipvq_tokenize_train_f32": (_i, [_vp] * 13 + [_i64] + [_i] * 5 + [_vp]),
    "lipvq_mlp3_packed_f16_bytes": (_sz, [_i] * 4),
    "lipvq_mlp3_pack_f16_f32": (_i, [_vp] * 4 + [_i] * 4 + [_vp]),
    "lipvq_tokenize_fast_f32": (_i, [_vp] * 10 + [_i64] + [_i] * 5 + [_vp]),
    "lipvq_mlp3_packed_bwd_floats": (_sz, [_i, _i, _i, _i]),
    "lipvq_mlp3_pack_bwd_f32": (_i, [_vp] * 4 + [_i] * 4 + [_vp]),
    "lipvq_mlp3_pack_bwd2_f32": (_i, [_vp] * 4 + [_i] * 4 + [_vp] * 4 + [_i] * 4 + [_vp]),
    "lipvq_mlp3_bwd_f32": (_i, [_vp] * 9 + [_i64] + [_i] * 7 + [_vp]),
    "lipvq_mlp3_loss_supported": (_i, [_i64, _i, _i, _i, _i]),
    "lipvq_mlp3_loss_f32": (_i, [_vp] * 7 + [_i64] + [_i] * 7 + [_vp] * 4 + [C.c_float, _i, _vp, _vp]),
    "lipvq_mlp3_bwd_vq_supported": (_i, [_i64, _i, _i, _i, _i]),
    "lipvq_mlp3_bwd_vq_f32": (_i, [_vp] * 9 + [_i64] + [_i] * 7 + [_vp] * 2 + [C.c_float] + [_vp] * 4 + [C.c_float, _vp, _vp]),
    "lipvq_wgrad_workspace_bytes": (_sz, [_i64, _i, _i]),
    "lipvq_wgrad_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "lipvq_scatter_add_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "lipvq_scatter_add_det_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "lipvq_scatter_add_sorted_supported": (_i, [_i64, _i, _i]),
    "lipvq_scatter_add_sorted_workspace_bytes": (_sz, [_i64, _i, _i]),
    "lipvq_scatter_add_sorted_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "lipvq_scatter_add_sorted_vq_f32": (_i, [_vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "lipvq_lipschitz_bwd_f32": (_i, [_vp] * 5 + [_i, _i, _vp]),
    "lipvq_scaled_diff_f32": (_i, [_vp, _vp, _vp, C.c_float, _vp, _vp, _i64, _vp]),
    "lipvq_linear_f32": (_i, [_vp] * 4 + [_i64, _i, _i, _vp]),
    "lipvq_embed_rows_f32": (_i, [_vp] * 5 + [C.c_float, _vp, _vp, _i64, _i, _i] + [_i64] * 4 + [_vp]),
    "lipvq_embed_rows_bwd_f32": (_i, [_vp] * 10 + [_i64, _i, _i] + [_i64] * 4 + [_vp]),
    "lipvq_embed_rows_bwd_ws_supported": (_i, [_i64, _i, _i, _i64]),
    "lipvq_embed_rows_bwd_workspace_bytes": (_sz, [_i64, _i, _i, _i64]),
    "lipvq_embed_rows_bwd_ws_f32": (_i, [_vp] * 11 + [_i64, _i, _i] + [_i64] * 4 + [_vp]),
    "lipvq_linear_act_f32": (_i, [_vp] * 5 + [_i64, _i, _i, _i, _vp]),
    "lipvq_bin_minmax_f32": (_i, [_vp] * 3 + [_i64, _i, _vp]),
    "lipvq_bin_discretize_f32": (_i, [_vp] * 4 + [_i64, _i, _i, _vp]),
    "lipvq_bin_boundaries_f32": (_i, [_vp] * 3 + [_i, _i, _vp]),
    "lipvq_bin_hidden_f32": (_i, [_vp] * 5 + [_i64, _i, _i, _i, _vp]),
    "lipvq_act_bwd_f32": (_i, [_vp] * 3 + [_i64, _i, _vp]),
    "lipvq_spectral_norm_f32": (_i, [_vp] * 5 + [_i, _i, _i, C.c_float, _vp]),
    "lipvq_spectral_norm_bwd_f32": (_i, [_vp] * 6 + [_i, _i, _vp]),
    "lipvq_attention_f32": (_i, [_vp] * 4 + [C.c_float, _i64, _i, _i, _vp]),
    "lipvq_attention_bwd_f32": (_i, [_vp] * 7 + [C.c_float, _i64, _i, _i, _vp]),
    "lipvq_add_layernorm_f32": (_i, [_vp] * 4 + [C.c_float] + [_vp] * 3 + [_i64, _i, _vp]),
    "lipvq_layernorm_bwd_f32": (_i, [_vp] * 7 + [_i64, _i, _vp]),
    "lipvq_adamw_workspace_bytes": (_sz, []),
    "lipvq_adamw_f32": (_i, [_vp] * 6 + [_i] + [C.c_double] * 5 + [_vp, _vp]),
    "lipvq_ema_update_f32": (_i, [_vp] * 5 + [C.c_float, C.c_float, _i, _i, _vp, _vp]),
    "lipvq_comm_unique_id": (_i, [_vp]),
    "lipvq_comm_init": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "lipvq_comm_destroy": (_i, [_vp]),
    "lipvq_allreduce_counts": (_i, [_vp, _i, _vp, _vp]),
    "lipvq_allreduce_f32": (_i, [_vp, _i64, _vp, _vp]),
}


class LipvqLibraryError(RuntimeError):
    pass


def _load():
    if not _LIB_PATH.exists():
        raise ImportError(
            f"{_LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C lipvq-vae_amd/csrc). "
            "There is no CPU fallback for the tokenizer path.")
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype, fn.argtypes = res, args
    got = lib.lipvq_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"lipvq ABI version {got} != expected {ABI_VERSION}")
    return lib


lib = _load()


OPTIONS = ("screen_mode", "tok_shape", "tok_ze_rows", "tok_grid", "rows_grid", "wgrad_chunk", "wgrad_per_tile", "wgrad_no_wg5",
           "wgrad_rows", "embed_bwd_grid", "mlp3_small_tiles", "mlp3_sub", "mlp3_lds_rows", "tok_inplace", "tok_defer_ze", "tok_nt_ze", "tok_ze_ring")


def set_option(name: str, value=None) -> None:
    """lipvq_set_option (include/lipvq.h): a process-global switch for tests and measurements; value None = the default.
    Results are the same under every setting."""
    v = None if value is None else str(value).encode()
    if lib.lipvq_set_option(name.encode(), v) != 0:
        raise LipvqLibraryError(lib.lipvq_last_error().decode(errors="replace"))


def get_option(name: str):
    v = lib.lipvq_get_option(name.encode())
    return None if v is None else v.decode()


# Development hook: with LIPVQ_DEV_KNOBS=1 in the environment, LIPVQ_<OPTION> variables are forwarded ONCE, here, through
# lipvq_set_option (scripts/dev/*: one-off measurements without touching code).  Without it the environment is ignored: the
# library itself reads none.
if _os.environ.get("LIPVQ_DEV_KNOBS") == "1":
    for _o in OPTIONS:
        _v = _os.environ.get("LIPVQ_" + _o.upper())
        if _v is not None:
            set_option(_o, _v)


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib.lipvq_last_error().decode(errors="replace")
        raise LipvqLibraryError(f"{what} failed with status {status}: {msg}")
