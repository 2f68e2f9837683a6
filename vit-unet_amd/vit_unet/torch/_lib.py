"""ctypes binding of libvitunet_amd.so (the C ABI declared in include/vit_unet_amd.h).

The product path has no CPU fallback: if the shared library is missing, or a kernel is asked to
run on a non-GPU tensor, this module raises.  `__graft_entry__.build()` (or `make -C
vit-unet_amd/csrc`) produces the library next to this file.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VU_LIB_PATH: a measurement build of the same sources (tools/build_variant.sh: -D switches of csrc/vu_flash.hip etc.); the product
# and the tests load the in-tree library
LIB_PATH = os.environ.get("VU_LIB_PATH") or os.path.join(_HERE, "libvitunet_amd.so")
if os.environ.get("VU_LIB_PATH"):      # never silently: a measurement build is not the product
    import sys as _sys
    print(f"vit_unet: loading the HIP library from VU_LIB_PATH={LIB_PATH} (a measurement build, not the in-tree library)", file=_sys.stderr)

VU_OK = 0
ABI_VERSION = 200        # include/vit_unet_amd.h: vu_version()


class VuError(RuntimeError):
    pass


class vu_config(C.Structure):
    _fields_ = [("depth", C.c_int), ("depth_te", C.c_int), ("size_bottleneck", C.c_int),
                ("im_size", C.c_int), ("patch_size", C.c_int), ("num_channels", C.c_int),
                ("hidden_dim", C.c_int), ("num_heads", C.c_int),
                ("attn_drop", C.c_float), ("proj_drop", C.c_float), ("linear_drop", C.c_float),
                ("out_conv", C.c_int), ("dtype", C.c_int), ("attn_operands", C.c_int)]


class vu_param_entry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", C.c_longlong), ("ndim", C.c_int),
                ("shape", C.c_int * 4), ("bn_index", C.c_int)]


class vu_attn_params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("mix_w", "mix_b", "bn_w", "bn_b", "wq", "wk", "wv",
                                          "proj_w", "proj_b", "run_mean", "run_var")] + [("operands", C.c_int)]


class vu_attn_grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("mix_w", "mix_b", "bn_w", "bn_b", "wq", "wk", "wv",
                                          "proj_w", "proj_b")]


_vp, _i, _ll, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint64, C.c_size_t
_cfgp = C.POINTER(vu_config)

# name -> (restype, argtypes); mirrors include/vit_unet_amd.h one to one
SIGNATURES = {
    "vu_version": (_i, []),
    "vu_config_size": (_i, []),
    "vu_last_error": (C.c_char_p, []),
    "vu_model_validate": (_i, [_cfgp]),
    "vu_model_param_elems": (_ll, [_cfgp]),
    "vu_model_num_params": (_i, [_cfgp]),
    "vu_model_param_table": (_i, [_cfgp, C.POINTER(vu_param_entry), _i]),
    "vu_model_num_attn": (_i, [_cfgp]),
    "vu_model_workspace_bytes": (_sz, [_cfgp, _i]),
    "vu_model_workspace_bytes_ex": (_sz, [_cfgp, _i, _i]),
    "vu_model_pcache_bytes": (_sz, [_cfgp, _i]),
    "vu_model_workspace_describe": (_i, [_cfgp, _i, C.c_char_p, _i]),
    "vu_model_prefers_eager": (_i, [_cfgp, _i]),
    "vu_set_flash_key_split": (_i, [_i]),
    "vu_set_flash_pcache": (_i, [_i]),
    "vu_set_flash_pcache_budget": (_i, [C.c_ulonglong]),
    "vu_set_deferred_reductions": (_i, [_i]),
    "vu_model_forward": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _u64, _vp, _vp]),
    "vu_model_backward": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _u64, _vp, _i, _vp]),
    "vu_model_num_backward_units": (_i, [_cfgp]),
    "vu_model_backward_unit_ranges": (_i, [_cfgp, C.POINTER(C.c_longlong), _i]),
    "vu_model_backward_units": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _u64, _vp, _i, _i, _vp]),
    "vu_retile": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vu_retile_add": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vu_conv3x3_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "vu_conv3x3_bwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "vu_conv3x3_qkv_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "vu_conv3x3_qkv_dgrad": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "vu_conv3x3_qkv_wgrad": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _ll, _i, _i, _vp]),
    "vu_attn_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "vu_attn_forward": (_i, [_i, C.POINTER(vu_attn_params), _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i,
                             _f, _f, _i, _u64, _u64, _vp]),
    "vu_attn_backward": (_i, [_i, C.POINTER(vu_attn_params), C.POINTER(vu_attn_grads), _vp, _vp, _vp, _vp, _vp,
                              _vp, _sz, _i, _i, _i, _i, _i, _f, _f, _i, _u64, _u64, _vp]),
    "vu_layernorm_workspace_floats": (_sz, [_i, _ll]),
    "vu_add_layernorm_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp]),
    "vu_layernorm_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp]),
    "vu_gemm": (_i, [_i, _i, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _ll, _ll, _i, _i,
                     _ll, _ll, _ll, _ll, _ll, _ll, _f, _vp, _i, _vp]),
    "vu_ff_scratch_bytes": (_sz, [_i, _ll, _i, _i]),
    "vu_ff_forward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _f, _i, _u64, _u64, _vp]),
    "vu_ff_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _f, _i, _u64, _u64,
                            _vp]),
    "vu_mse_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _f, _vp]),
    "vu_adamw": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _f, _vp]),
    "vu_round_e4m3": (_i, [_i, _vp, _ll, _vp]),
    "vu_colsum": (_i, [_i, _vp, _vp, _ll, _i, _ll, _vp]),
    "vu_cast_bf16": (_i, [_vp, _vp, _ll, _vp]),
    "vu_dice_partials_floats": (_sz, []),
    "vu_dice_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _vp]),
    "vu_psnr_partials_floats": (_sz, [_i]),
    "vu_psnr": (_i, [_vp, _vp, _vp, _vp, _i, _ll, _f, _vp]),
    "vu_ssim_partials_floats": (_sz, [_i, _i, _i, _i, _i]),
    "vu_ssim": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "vu_denoise_prepare_scratch_bytes": (_sz, [_i, _i, _i]),
    "vu_denoise_prepare": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "vu_seg_prepare_scratch_bytes": (_sz, [_i, _i, _i]),
    "vu_seg_prepare": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "vu_add": (_i, [_i, _vp, _vp, _vp, _ll, _vp]),
    "vu_dropout": (_i, [_i, _vp, _vp, _ll, _f, _u64, _u64, _vp]),
    "vu_gelu_fwd": (_i, [_i, _vp, _vp, _ll, _vp]),
    "vu_gelu_bwd": (_i, [_i, _vp, _vp, _vp, _ll, _vp]),
    "vu_token_pool4_fwd": (_i, [_i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vu_token_pool4_bwd": (_i, [_i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vu_softmax_rows_fwd": (_i, [_i, _vp, _vp, _vp, _ll, _i, _i, _f, _f, _u64, _u64, _vp]),
    "vu_softmax_rows_bwd": (_i, [_i, _vp, _vp, _vp, _ll, _i, _i, _f, _f, _u64, _u64, _vp]),
    "vu_add_layernorm_fwd_eps": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _f, _vp]),
    "vu_set_attn_form": (_i, [_i, _i]),
    "vu_prof_enable": (_i, [_vp]),
    "vu_prof_report": (C.c_char_p, []),
    "vu_prof_gate": (_i, [_i]),
    "vu_dp_unique_id": (_i, [_vp]),
    "vu_dp_init": (_i, [_i, _i, _vp]),
    "vu_dp_allreduce_bucket": (_i, [_vp, _ll, _i, _vp]),
    "vu_dp_world": (_i, []),
    "vu_dp_finalize": (_i, []),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VuError(f"{LIB_PATH} is missing: build it with `make -C vit-unet_amd/csrc` "
                          "(or __graft_entry__.build()); the ViT-UNet path has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)     # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        # ABI guard: a stale binding (or a stale .so) must not pass a short struct - the C side would read past its end
        if L.vu_version() != ABI_VERSION or L.vu_config_size() != C.sizeof(vu_config):
            raise VuError(f"{LIB_PATH}: ABI mismatch (library version {L.vu_version()}, vu_config {L.vu_config_size()} bytes; "
                          f"binding expects version {ABI_VERSION}, {C.sizeof(vu_config)} bytes): rebuild with `make -C vit-unet_amd/csrc`")
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = lib().vu_last_error().decode("utf-8", "replace")
        if rc == -1 and msg and ("Depth must" in msg or "Patch size" in msg):
            raise AssertionError(msg)               # reference error convention (model.py:281-283)
        raise VuError(f"{what or 'vit_unet_amd'} failed (rc={rc}): {msg}")
    return rc


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VuError("the ViT-UNet HIP path needs GPU tensors (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise VuError("non-contiguous tensor passed to the HIP path")
    if t.device.index != torch.cuda.current_device():
        # launches and events target the CURRENT HIP device: a tensor on another card would get a foreign stream
        raise VuError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                      "wrap the call in `with torch.cuda.device(tensor.device):` (or torch.cuda.set_device)")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}


# what q, k, v are rounded to before the attention products (vu_config.attn_operands): the storage dtype itself, or
# OCP e4m3 (BASELINE config 5)
OPERAND_CODE = {"storage": 0, "e4m3": 1}


def operand_code(name) -> int:
    if name not in OPERAND_CODE:
        raise ValueError(f"attn_operands must be one of {sorted(OPERAND_CODE)}, got {name!r}")
    return OPERAND_CODE[name]


def set_attn_form(flash: int = -1, centered: int = 0) -> None:
    """Process-level choice of the re-attention form (include/vit_unet_amd.h: vu_set_attn_form): flash -1 auto / 0 never /
    1 wherever covered; centered 1: the stand-alone op in the centred-map form.  Tests and experiments only."""
    check(lib().vu_set_attn_form(int(flash), int(centered)), "vu_set_attn_form")


def set_flash_key_split(ks: int = 0) -> None:
    """Recompute attention only: waves of a workgroup that share one tile and split the streamed keys / queries (0 = by launch
    size, 1, 2, 3; include/vit_unet_amd.h: vu_set_flash_key_split).  Tests and experiments only."""
    check(lib().vu_set_flash_key_split(int(ks)), "vu_set_flash_key_split")


def set_flash_pcache(on: int = -1) -> None:
    """Recompute attention, 8 heads: the probability cache (include/vit_unet_amd.h: vu_set_flash_pcache; -1 default, 0 off, 1 on).
    Changes the workspace size: call it before a workspace is sized, never between a forward and its backward."""
    check(lib().vu_set_flash_pcache(int(on)), "vu_set_flash_pcache")


def make_config(depth, depth_te, size_bottleneck, preprocessing, im_size, patch_size, num_channels,
                hidden_dim, num_heads, attn_drop, proj_drop, linear_drop, dtype, attn_operands="storage") -> vu_config:
    if dtype not in DTYPE_CODE:
        raise ValueError(f"dtype must be torch.float32 or torch.bfloat16, got {dtype}")
    return vu_config(int(depth), int(depth_te), int(size_bottleneck), int(im_size), int(patch_size),
                     int(num_channels), int(hidden_dim), int(num_heads), float(attn_drop), float(proj_drop),
                     float(linear_drop), 1 if preprocessing == "conv" else 0, DTYPE_CODE[dtype],
                     operand_code(attn_operands))


def backward_unit_ranges(cfg: vu_config):
    """[(lo, hi)] arena range of the parameter gradients of every backward unit, in backward order."""
    L = lib()
    n = check(L.vu_model_num_backward_units(C.byref(cfg)), "vu_model_num_backward_units")
    arr = (C.c_longlong * (2 * n))()
    check(L.vu_model_backward_unit_ranges(C.byref(cfg), arr, n), "vu_model_backward_unit_ranges")
    return [(int(arr[2 * i]), int(arr[2 * i + 1])) for i in range(n)]


def param_table(cfg: vu_config):
    L = lib()
    n = check(L.vu_model_num_params(C.byref(cfg)), "vu_model_num_params")
    arr = (vu_param_entry * n)()
    check(L.vu_model_param_table(C.byref(cfg), arr, n), "vu_model_param_table")
    out = []
    for e in arr:
        out.append((e.name.decode(), int(e.offset), tuple(e.shape[i] for i in range(e.ndim)), int(e.bn_index)))
    return out
