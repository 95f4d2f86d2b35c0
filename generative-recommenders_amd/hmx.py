"""ctypes binding of libhmx.so (C ABI declared in include/hmx.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMX_LIBRARY", os.path.join(_HERE, "libhmx.so"))

P = c_void_p
I64 = c_int64
F = c_float

class WgradDesc(ctypes.Structure):
    """hmx_wgrad_desc of include/hmx.h."""
    _fields_ = [("dY", c_void_p), ("ldy", c_int64), ("X", c_void_p), ("ldx", c_int64), ("in_rowmask", c_void_p),
                ("T", c_int64), ("N", c_int64), ("K", c_int64), ("dW", c_void_p), ("ldw", c_int64),
                ("dbias", c_void_p), ("accumulate", c_int32), ("n_group", c_int32), ("n_group_padded", c_int32),
                ("k_group", c_int32), ("k_group_padded", c_int32)]


class AdamwTable(ctypes.Structure):
    """hmx_adamw_table of include/hmx.h."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n_rows", c_int64), ("d", c_int64),
                ("row_flags", c_void_p), ("bf16_out", c_void_p), ("bf16_ld", c_int64), ("bf16_max_norm", c_void_p)]


class BlockWeights(ctypes.Structure):
    """hmx_block_weights of include/hmx.h."""
    _fields_ = ([(n, c_void_p) for n in ("W1", "W2", "Wg", "Wu", "Wd")] + [("Win", c_void_p * 8), ("Wout", c_void_p * 8)] +
                [(n, c_void_p) for n in ("scale1", "b1", "b2", "scale2")])


_BLOCK_PTRS_IN = ("mask", "x", "drop", "scale1", "b1", "rope_cos", "rope_sin")


class InputAssemblyArgs(ctypes.Structure):
    """hmx_input_assembly_args of include/hmx.h."""
    _fields_ = ([(k, c_int64) for k in ("T", "L", "ds", "d", "time_span")] +
                [(k, c_void_p) for k in ("ids", "token_type", "timestamps", "action", "dwell", "item_table", "user_table",
                                         "pos_table")] +
                [(k, c_int64) for k in ("item_rows", "user_rows", "pos_rows")] +
                [("W", c_void_p), ("ldw", c_int64), ("bias", c_void_p), ("pre", c_void_p), ("scale", c_float)] +
                [(k, c_void_p) for k in ("time_rows", "action_rows", "dwell_rows")] +
                [(k, c_int64) for k in ("time_n", "action_n", "dwell_n")] +
                [(k, c_void_p) for k in ("out", "mask", "pos_idx", "time_bin", "e", "dout", "g", "gs", "de")])


class BlockArgs(ctypes.Structure):
    """hmx_block_args of include/hmx.h."""
    _fields_ = ([("T", c_int64), ("L", c_int64), ("d", c_int64), ("H", c_int64), ("first", c_int32), ("eps", c_float),
                 ("packed", c_void_p)] +
                [(n, c_void_p) for n in ("mask", "x", "drop", "scale1", "b1", "rope_cos", "rope_sin",
                                         "rstd1", "x_in", "z", "y2", "yg", "b2", "scale2", "out", "m", "rstd2", "dout",
                                         "dyg", "dy2", "dm", "hmid", "n2", "dg", "du", "p", "dy1", "dscale_part",
                                         "dx_in", "dz", "dx", "s", "dx1", "n")] + [("d_real", c_int64)])


# name -> (restype, argtypes); must list every symbol declared in include/hmx.h
PROTOTYPES = {
    "hmx_abi_version": (c_int, []),
    "hmx_error_string": (c_char_p, [c_int]),
    "hmx_gather_rows_f32": (c_int, [P, I64, I64, P, I64, P, P]),
    "hmx_gather_rows_bf16": (c_int, [P, I64, I64, P, I64, P, P]),
    "hmx_row_any_nonzero_f32": (c_int, [P, I64, I64, P, P]),
    "hmx_gather_routed_f32": (c_int, [P, I64, P, I64, I64, P, P, I64, P, P]),
    "hmx_gather_gelu_rows_f32": (c_int, [P, I64, I64, P, I64, P, P]),
    "hmx_scatter_add_rows_workspace_bytes": (c_size_t, [I64, I64, I64]),
    "hmx_scatter_add_rows_f32": (c_int, [P, P, P, I64, I64, I64, P, I64, I64, P, P, c_size_t, P]),
    "hmx_scatter_add_rows_static_scratch_bytes": (c_size_t, []),
    "hmx_scatter_add_rows_f32_static": (c_int, [P, P, P, I64, I64, I64, P, I64, I64, P]),
    "hmx_table_prepare_bf16": (c_int, [P, I64, I64, P, P, P]),
    "hmx_score_topk_workspace_bytes": (c_size_t, [I64, I64, I64, I64]),
    "hmx_score_topk_f32": (c_int, [P, I64, P, I64, I64, I64, P, I64, I64, P, P, P, c_size_t, P]),
    "hmx_score_topk_f32_flagged": (c_int, [P, I64, P, I64, I64, I64, P, I64, I64, P, P, P, P, c_size_t, P]),
    "hmx_score_topk_bf16": (c_int, [P, I64, P, P, P, I64, I64, I64, P, I64, I64, P, P, P, P, c_size_t, P]),
    "hmx_topk_sample_thresholds_bf16": (c_int, [P, I64, P, I64, I64, I64, I64, I64, P, P, c_size_t, P]),
    "hmx_kth_largest_f32": (c_int, [P, I64, I64, I64, I64, I64, P, P]),
    "hmx_score_topk_bf16_seeded": (c_int, [P, I64, P, P, P, I64, I64, I64, I64, P, P, P, P, P, P, c_size_t, P]),
    "hmx_topk_merge": (c_int, [P, P, I64, I64, I64, P, P, P]),
    "hmx_debug_ring_skip_publish": (c_int, [I64]),
    "hmx_attn1q_fwd": (c_int, [P, P, P, I64, I64, I64, I64, P, P, P]),
    "hmx_attn1q_bwd": (c_int, [P, P, P, P, P, I64, I64, I64, I64, P, P, P]),
    "hmx_pscan_fwd": (c_int, [P, P, I64, I64, I64, P, P]),
    "hmx_pscan_bwd": (c_int, [P, P, P, I64, I64, I64, P, P, P]),
    "hmx_nce_fwd_f32": (c_int, [P, P, I64, I64, P, I64, P, I64, P, I64, I64, I64, F, c_int, P, P, P, P, P, P, P]),
    "hmx_nce_bwd_workspace_bytes": (c_size_t, [I64, I64, I64, I64, I64]),
    "hmx_nce_bwd_f32": (c_int, [P, P, I64, I64, P, I64, P, I64, P, I64, I64, I64, F, c_int, P, P, P, P, P, P, P, P, P, P, P,
                                c_size_t, P]),
    "hmx_nce_fwd_periodic_f32": (c_int, [P, P, I64, I64, P, I64, P, I64, I64, P, I64, I64, I64, F, c_int, P, P, P, P, P,
                                         P, P]),
    "hmx_nce_bwd_grouped_workspace_bytes": (c_size_t, [I64, I64, I64, I64, I64, I64, I64]),
    "hmx_nce_bwd_grouped_f32": (c_int, [P, P, I64, I64, P, I64, P, I64, P, I64, I64, I64, F, c_int, P, P, P, P, P, P, P, P,
                                        P, P, P, I64, I64, P, c_size_t, P]),
    "hmx_adamw_chunk_elems": (I64, []),
    "hmx_grad_norm_workspace_bytes": (c_size_t, [I64]),
    "hmx_adamw_table_blocks": (I64, [I64]),
    "hmx_grad_norm_f32": (c_int, [P, P, I64, I64, ctypes.POINTER(AdamwTable), I64, F, P, c_size_t, P, P]),
    "hmx_adamw_tables_step_f32": (c_int, [ctypes.POINTER(AdamwTable), I64, F, F, F, F, F, I64, P, P]),
    "hmx_adamw_step_f32": (c_int, [P, P, I64, I64, F, F, F, F, F, I64, P, c_int, P]),
    "hmx_rmsnorm_fwd": (c_int, [P, P, I64, I64, F, P, P, P]),
    "hmx_rmsnorm_bwd_blocks": (I64, [I64]),
    "hmx_rmsnorm_bwd": (c_int, [P, P, P, P, I64, I64, P, P, P]),
    "hmx_colsum_workspace_bytes": (c_size_t, [I64, I64]),
    "hmx_colsum_f32": (c_int, [P, I64, I64, I64, P, P, c_size_t, P]),
    "hmx_set_linear_precision": (c_int, [c_int]),
    "hmx_set_wgrad_precision": (c_int, [c_int]),
    "hmx_linear_f32": (c_int, [P, I64, P, I64, I64, P, I64, I64, I64, P, I64, P, I64, F, P, I64, F, P, P, c_int, P]),
    "hmx_heads_linear_f32": (c_int, [P, I64, I64, P, I64, I64, I64, I64, I64, P, I64, I64, c_int, c_int, P]),
    "hmx_linear_wgrad_workspace_bytes": (c_size_t, [I64, I64, I64]),
    "hmx_linear_wgrad_f32": (c_int, [P, I64, P, I64, P, I64, I64, I64, P, I64, P, P, c_size_t, P]),
    "hmx_linear_wgrad_batch_workspace_bytes": (c_size_t, [ctypes.POINTER(WgradDesc), I64]),
    "hmx_linear_wgrad_batch_f32": (c_int, [ctypes.POINTER(WgradDesc), I64, P, c_size_t, P]),
    "hmx_silu_rope_fwd": (c_int, [P, P, P, I64, I64, I64, I64, P, P, P]),
    "hmx_silu_rope_bwd": (c_int, [P, P, P, P, P, I64, I64, I64, I64, P, P]),
    "hmx_binary_fwd": (c_int, [P, P, I64, F, c_int, P, P]),
    "hmx_binary_bwd": (c_int, [P, P, P, I64, F, c_int, P, P, P]),
    "hmx_combine_f32": (c_int, [P, P, P, P, F, F, F, F, P, I64, I64, P, P]),
    "hmx_seq_index": (c_int, [P, P, I64, I64, I64, P, P, P, P]),
    "hmx_conv_silu_fwd": (c_int, [P, P, P, I64, I64, I64, P, P]),
    "hmx_conv_silu_bwd": (c_int, [P, P, P, P, I64, I64, I64, P, P, P]),
    "hmx_scan_fwd": (c_int, [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, P, P, P]),
    "hmx_scan_groups_per_head": (I64, [I64]),
    "hmx_scan_state_floats": (I64, [I64, I64, I64]),
    "hmx_scan_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, P, P, P, P, P]),
    "hmx_sum_groups_f32": (c_int, [P, I64, I64, I64, P, P]),
    "hmx_block_supported": (c_int, [I64, I64]),
    "hmx_block_packed_bytes": (c_size_t, [I64, I64]),
    "hmx_block_scale_partial_rows": (I64, [I64]),
    "hmx_block_padded_width": (I64, [I64, I64]),
    "hmx_block_pack_f32": (c_int, [ctypes.POINTER(BlockWeights), I64, I64, I64, P, P]),
    "hmx_input_assembly_supported": (c_int, [I64, I64]),
    "hmx_input_assembly_fwd": (c_int, [ctypes.POINTER(InputAssemblyArgs), P]),
    "hmx_input_assembly_bwd": (c_int, [ctypes.POINTER(InputAssemblyArgs), P]),
    "hmx_block_pre_fwd": (c_int, [ctypes.POINTER(BlockArgs), P]),
    "hmx_block_post_fwd": (c_int, [ctypes.POINTER(BlockArgs), P]),
    "hmx_block_post_bwd": (c_int, [ctypes.POINTER(BlockArgs), P]),
    "hmx_block_pre_bwd": (c_int, [ctypes.POINTER(BlockArgs), P]),
    "hmx_gather_sum_rows_f32": (c_int, [P, I64, I64, P, I64, I64, P, P]),
    "hmx_act_fwd": (c_int, [P, I64, c_int, P, P]),
    "hmx_act_bwd": (c_int, [P, P, I64, c_int, P, P]),
    "hmx_gate_fwd": (c_int, [P, P, I64, P, P]),
    "hmx_gate_bwd": (c_int, [P, P, P, I64, P, P, P]),
    "hmx_gate_mix_fwd": (c_int, [P, P, P, I64, P, P]),
    "hmx_gate_mix_bwd": (c_int, [P, P, P, P, I64, P, P, P, P]),
    "hmx_layernorm_bwd_blocks": (I64, [I64]),
    "hmx_layernorm_fwd": (c_int, [P, P, P, I64, I64, F, P, P, P, P]),
    "hmx_layernorm_bwd": (c_int, [P, P, P, P, P, I64, I64, P, P, P]),
    "hmx_row_inv_norm_f32": (c_int, [P, I64, I64, F, P, P]),
    "hmx_dropout_multiplier_f32": (c_int, [ctypes.c_uint64, ctypes.c_uint64, I64, F, P, P]),
    "hmx_attn_fwd": (c_int, [P, P, P, I64, I64, I64, I64, I64, P, P, P]),
    "hmx_attn_bwd": (c_int, [P, P, P, P, P, I64, I64, I64, I64, I64, P, P, P, P, P]),
    "hmx_profile_enable": (c_int, [c_int]),
    "hmx_profile_filter": (c_int, [c_char_p]),
    "hmx_profile_collect": (c_int, [c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)]),
}

_lib = None


class HmxError(RuntimeError):
    pass


def lib():
    """Load libhmx.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HmxError(f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                continue                      # reported by check_exports(); calling it raises AttributeError
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise HmxError(f"{what} failed: {lib().hmx_error_string(code).decode()} ({code})")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def profile_enable(on=True, only=None):
    """Event brackets around the tagged kernels on / off; only = a tag: record that tag alone."""
    check(lib().hmx_profile_filter(only.encode() if only else None), "hmx_profile_filter")
    check(lib().hmx_profile_enable(1 if on else 0), "hmx_profile_enable")


def profile_collect(tag):
    """-> (total milliseconds, launches) of the spans recorded under `tag` since the last collect."""
    ms, n = ctypes.c_double(0.0), c_int64(0)
    check(lib().hmx_profile_collect(tag.encode(), ctypes.byref(ms), ctypes.byref(n)), "hmx_profile_collect")
    return ms.value, n.value


SPLIT_BF16 = True


def set_linear_precision(split_bf16=True):
    """True (default): split-bf16 projections (fp32-class accuracy); False: single bf16 MFMA per product (the fused
    block kernels are split-bf16 only: in single mode blocks run on the per-operation kernels)."""
    global SPLIT_BF16
    check(lib().hmx_set_linear_precision(1 if split_bf16 else 0), "hmx_set_linear_precision")
    SPLIT_BF16 = bool(split_bf16)


def set_wgrad_precision(fp32_mfma=False):
    """False (default): weight gradients multiply split-bf16 operands on the bf16 MFMA (fp32-class accuracy);
    True: on the fp32 MFMA."""
    check(lib().hmx_set_wgrad_precision(1 if fp32_mfma else 0), "hmx_set_wgrad_precision")
