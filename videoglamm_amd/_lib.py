"""ctypes binding of libvgkernels.so (include/vg_kernels.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C videoglamm_amd/csrc``.
There is NO fallback: if the shared object is missing or a call fails, this module raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VG_KERNELS_SO") or os.path.join(_HERE, "csrc", "libvgkernels.so")      # (override: diagnostic builds)

_lib = None


class VGKernelError(RuntimeError):
    pass


def _sig(lib):
    P, I, L, F = c_void_p, c_int, c_int64, c_float
    S = {
        "vg_version": ([], c_int),
        "vg_last_error": ([], c_char_p),
        "vg_init": ([I], c_int),
        "vg_graph_node_counts": ([P, ctypes.POINTER(c_int64)], c_int),
        "vg_gemm": ([P, L, L, P, L, L, P, L, L, P, P, P, L, L, I, I, I, I, I, I, I, I, P], c_int),
        "vg_gemm_splitk": ([P, L, P, L, P, L, P, P, P, L, I, I, I, I, I, I, I, P, L, P], c_int),
        "vg_quantize_fp8_rows": ([P, L, P, L, P, L, I, I, P], c_int),
        "vg_gemm_f8": ([P, L, P, P, L, P, P, L, P, P, L, L, L, L, I, I, P], c_int),
        "vg_gemm_route": ([L, L, L, I, I, I], c_int),
        "vg_mlp_rows_supported": ([I, I], c_int),
        "vg_mlp_rows": ([P, L, P, L, P, P, F, P, P, P, P, L, I, I, I, P], c_int),
        "vg_gemm_rows": ([P, L, P, L, P, L, P, P, L, L, I, I, I, P, P, F, P, L, I, P, P, I, I, I, L, I, I, I, I, P], c_int),
        "vg_gemm_window": ([P, L, P, L, P, L, P, P, P, L, I, I, I, I, I, I, I, I, I, I, P, P], c_int),
        "vg_gemm_ln": ([P, L, P, L, P, L, P, P, P, F, L, I, I, I, I, I, I, I, I, P, I, P], c_int),
        "vg_attention": ([P, P, P, P, I, I, I, I, I, I] + [L] * 12 + [F, I, I, P], c_int),
        "vg_attention_splitkv": ([P, P, P, P, I, I, I, I, I, I] + [L] * 12 + [F, I, I, P, L, I, P, P], c_int),
        "vg_attention_dv": ([P, P, P, P, I, I, I, I, I, I] + [L] * 12 + [F, I, P, L, I, P], c_int),
        "vg_window_attention": ([P, P, P, P, I, I, I, I, I] + [L] * 12 + [F, I, P], c_int),
        "vg_rope_kv_append": ([P, L, P, P, P, P, I, I, I, I, I, P, I, P], c_int),
        "vg_store_row": ([P, P, L, P, I, I, P], c_int),
        "vg_add_int": ([P, I, P], c_int),
        "vg_decode_gemv": ([P, P, L, P, P, F, P, I, I, I, I, I, P], c_int),
        "vg_decode_gemv_w8": ([P, P, L, P, P, P, F, P, I, I, I, I, P], c_int),
        "vg_decode_attention_ws_floats": ([I, I, I, I], c_int64),
        "vg_decode_attention": ([P, P, P, P, P, P, I, I, I, I, I, F, P, P, L, I, I, P], c_int),
        "vg_decode_qkv_rope_supported": ([I, I, I, I, I], c_int),
        "vg_decode_qkv_rope": ([P, P, L, P, F, P, P, P, P, P, I, I, I, I, I, P], c_int),
        "vg_decode_attention2_supported": ([I, I, I, I], c_int),
        "vg_decode_attention2": ([P, P, P, P, I, I, I, I, I, F, P, P, L, I, I, P], c_int),
        "vg_decode_advance": ([P, P, P, P, I, P, P, I, P, P, P, I, I, P], c_int),
        "vg_decode_step_begin": ([P, P, P, I, I, P, P, P, P, I, P], c_int),
        "vg_argmax_partial": ([P, L, I, P, I, P], c_int),
        "vg_decode_step_end": ([P, P, P, P, P, I, P, P, I, P, P, I, I, P], c_int),
        "vg_mlp3_grouped": ([P, L, L, P, P, P, P, P, P, P, L, L, I, I, I, I, I, I, ctypes.c_uint, P], c_int),
        "vg_decode_layer_roles": ([I, I, I, I, I, I], c_int),
        "vg_decode_layer_flag_ints": ([], c_int64),
        "vg_decode_layer": ([P, P, P, P, P, P, I, I, I, I, I, F, P, P, L, P, P, L, P, P, P, F, P, L, P, P, L, P, I, I, I, P], c_int),
        "vg_layernorm": ([P, L, P, P, P, L, L, I, F, I, I, P], c_int),
        "vg_mask_upscale": ([P, P, P, P, P, P, F, P, P, P, P, P, I, I, I, I, P], c_int),
        "vg_twoway_image_update": ([P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, I, I, I, P], c_int),
        "vg_heads_blockdiag": ([P, P, I, I, I, I, I, P], c_int),
        "vg_rmsnorm": ([P, L, P, P, L, L, I, F, I, I, P], c_int),
        "vg_axpby": ([P, P, P, L, F, F, L, I, I, I, P], c_int),
        "vg_activation": ([P, P, L, I, I, I, P], c_int),
        "vg_swiglu": ([P, P, L, I, I, P], c_int),
        "vg_cast": ([P, P, L, I, I, P], c_int),
        "vg_where_rows": ([P, P, P, P, L, L, L, F, L, I, P], c_int),
        "vg_mask_for_mem": ([P, P, L, I, F, F, I, P], c_int),
        "vg_threshold": ([P, P, L, P], c_int),
        "vg_rope_half": ([P, L, L, P, P, I, I, I, I, I, P], c_int),
        "vg_rope_axial": ([P, P, P, I, I, I, I, I, I, P], c_int),
        "vg_rope_axial_heads": ([P, L, L, P, P, I, I, I, I, I, I, P], c_int),
        "vg_embed": ([P, P, P, L, I, I, P], c_int),
        "vg_argmax": ([P, L, I, P, I, P], c_int),
        "vg_multimask_select": ([P, P, P, P, P, P, P, I, L, I, F, F, I, I, P], c_int),
        "vg_permute5": ([P, P, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), I, P], c_int),
        "vg_im2col": ([P, P, I, I, I, I, I, I, I, I, I, I, P], c_int),
        "vg_dwconv": ([P, P, P, P, I, I, I, I, I, I, P], c_int),
        "vg_conv3s2_ln_gelu": ([P, P, L, P, P, P, F, P, I, I, I, I, I, I, P], c_int),
        "vg_pixel_shuffle2": ([P, P, P, I, I, I, I, I, P], c_int),
        "vg_pool2": ([P, P, I, I, I, I, L, I, I, P], c_int),
        "vg_window_partition": ([P, P, I, I, I, I, I, I, P], c_int),
        "vg_window_unpartition": ([P, P, I, I, I, I, I, I, P], c_int),
        "vg_bilinear": ([P, P, I, I, I, I, I, P], c_int),
        "vg_bilinear_mask": ([P, P, I, I, I, I, I, P], c_int),
        "vg_upsample2_add": ([P, P, P, I, I, I, I, I, P], c_int),
        "vg_connected_components": ([P, P, P, I, I, I, I, P], c_int),
        "vg_remove_small_blobs": ([P, P, P, P, I, I, I, I, P], c_int),
        "vg_fill_holes": ([P, P, P, P, I, I, I, I, P], c_int),
        "vg_mask_pair_counts": ([P, P, P, P, I, I, L, I, P], c_int),
        "vg_boundary_counts": ([P, P, P, I, I, I, I, P], c_int),
        "vg_resample_u8": ([P, P, I, I, I, I, I, I, P, P, I, P], c_int),
        "vg_resize_cv_linear_u8": ([P, P, I, I, I, I, I, I, P, P, P, P, P], c_int),
        "vg_normalize_u8": ([P, P, I, I, I, I, I, I, I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), I, I, P], c_int),
    }
    for name, (args, res) in S.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.argtypes = args
        fn.restype = res
    return list(S)


EXPORTS = []


def load():
    """Load (once) and return the ctypes handle; raises VGKernelError when the .so is absent."""
    global _lib, EXPORTS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VGKernelError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C videoglamm_amd/csrc). "
                "There is no CPU fallback for the VideoGLaMM hot path."
            )
        lib = ctypes.CDLL(LIB_PATH)
        EXPORTS = _sig(lib)
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().vg_last_error()
        raise VGKernelError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
