"""ctypes binding of libflid_tg.so (include/flid_tg.h).  There is NO fallback: if the HIP library is missing or a call
fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLID_TG_LIB: another build of the same library (A/B timing of kernel variants, tools/); never a different implementation
LIB_PATH = os.environ.get("FLID_TG_LIB") or os.path.join(_HERE, "csrc", "libflid_tg.so")

c_i64, c_i32, c_f32, c_void = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class AttnDesc(C.Structure):
    """struct tg_attn_desc"""
    _fields_ = [("d_feat", c_void), ("feat_ld", c_i64), ("d_feat_idx", c_void),
                ("d_edge", c_void), ("edge_ld", c_i64), ("d_edge_idx", c_void),
                ("d_nbr", c_void), ("d_dt", c_void), ("d_te_w", c_void), ("d_te_b", c_void),
                ("m", c_i64), ("k", C.c_int), ("heads", C.c_int), ("dn", C.c_int), ("de", C.c_int), ("dt_dim", C.c_int),
                ("scale", c_f32), ("dropout_p", c_f32), ("seed", C.c_uint64), ("row0", c_i64)]


_PARAM_NAMES = ("Wq", "Wk", "Wv", "ln_g", "ln_b", "Wr", "br", "W1", "b1", "W2", "b2")


class LayerParams(C.Structure):
    """struct tg_layer_params / tg_layer_grads (same layout)"""
    _fields_ = [(n, c_void) for n in _PARAM_NAMES]


class LayerDesc(C.Structure):
    """struct tg_layer_desc"""
    _fields_ = [("attn", AttnDesc), ("params", LayerParams), ("own", c_void), ("own_ld", c_i64), ("raw", c_void), ("raw_ld", c_i64),
                ("cosb", c_void), ("res_dropout_p", c_f32), ("res_seed", C.c_uint64)] + \
               [(n, c_void) for n in ("qbias", "q", "u", "agg", "prob", "ctx", "res", "y", "mean", "rstd", "f1", "out", "wT")] + [("y_ld", c_i64)] + \
               [("compute_cosb", C.c_int), ("gather_table", c_void), ("gather_ld", c_i64), ("gather_idx", c_void)]


class LayerBwdDesc(C.Structure):
    """struct tg_layer_bwd_desc"""
    _fields_ = [("grads", LayerParams), ("dout", c_void)] + \
               [(n, c_void) for n in ("df1", "dy", "dsum", "dres", "dctx", "dagg", "du", "dq", "part", "vec", "d_cosb", "d_tew", "d_teb")] + \
               [("dfeat", c_void), ("dfeat_ld", c_i64), ("pad_row", c_i64), ("d_own", c_void), ("d_own_ld", c_i64),
                ("d_own_accumulate", C.c_int), ("d_raw", c_void), ("defer_join", C.c_int), ("finish_time_bias", C.c_int)]


class PackJob(C.Structure):
    """struct tg_pack_job"""
    _fields_ = [("src", c_void), ("ld", c_i64), ("N", C.c_int), ("K", C.c_int), ("trans", C.c_int), ("dst", c_void)] + \
               [(n, C.c_int) for n in ("src_N", "src_K", "n_len", "n_pad", "k_len", "k_pad")]


class WgradJob(C.Structure):
    """struct tg_wgrad_job"""
    _fields_ = [("A", c_void), ("lda", c_i64), ("M", C.c_int), ("B", c_void), ("ldb", c_i64), ("N", C.c_int), ("C", c_void), ("ldc", c_i64),
                ("colsum_A", c_void)]


class StepperCfg(C.Structure):
    """struct tg_stepper_cfg"""
    _fields_ = [("graph", c_void), ("d_node", c_void), ("node_ld", c_i64), ("d_edge", c_void), ("edge_ld", c_i64),
                ("dn", C.c_int), ("de", C.c_int), ("dt_dim", C.c_int), ("heads", C.c_int), ("layers", C.c_int), ("k", C.c_int),
                ("max_roots", c_i64), ("slots", C.c_int), ("d_param", c_void), ("param_floats", c_i64), ("dropout_p", c_f32),
                ("dedupe", C.c_int), ("extra_grad_floats", c_i64), ("tgn", C.c_int)]


class TgnBank(C.Structure):
    """struct tg_tgn_bank"""
    _fields_ = [("d_mem", c_void), ("mem_ld", c_i64), ("d_last_update", c_void), ("d_msg", c_void), ("msg_ld", c_i64), ("d_has", c_void),
                ("d_msg_time", c_void), ("d_last_idx_ws", c_void), ("h_has", c_void), ("h_msg_time", c_void), ("h_last", c_void),
                ("num_nodes", c_i64), ("past_violation", C.c_int)]


class AdamArgs(C.Structure):
    """struct tg_adam_args"""
    _fields_ = [("d_exp_avg", c_void), ("d_exp_avg_sq", c_void), ("n", c_i64), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double), ("step", c_i64)]


class Pack32Job(C.Structure):
    """struct tg_pack32_job"""
    _fields_ = [("src", c_void), ("ld", c_i64), ("N", C.c_int32), ("K", C.c_int32), ("trans", C.c_int32), ("dst", c_void)]


class DygCfg(C.Structure):
    """struct tg_dyg_cfg"""
    _fields_ = [("graph", c_void), ("d_node", c_void), ("node_ld", c_i64), ("d_edge", c_void), ("edge_ld", c_i64), ("num_edge_rows", c_i64),
                ("d_param", c_void), ("param_floats", c_i64), ("poff", c_i64 * 64)] + \
               [(n, C.c_int32) for n in ("dn", "de", "dt_dim", "channel", "layers", "heads", "max_len", "max_edges")]


GRAD_READY_FN = C.CFUNCTYPE(None, c_void, c_void, c_i64)          # tg_grad_ready_fn

# name -> (restype, argtypes); every symbol declared in include/flid_tg.h
SIGNATURES = {
    "tg_last_error": (C.c_char_p, []),
    "tg_version": (C.c_int, []),
    "tg_profile_enable": (None, [C.c_int]),
    "tg_profile_collect": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_i64), C.c_int]),
    "tg_stepper_param_floats": (c_i64, [C.POINTER(StepperCfg)]),
    "tg_stepper_arena_floats": (c_i64, [C.POINTER(StepperCfg)]),
    "tg_stepper_create": (C.c_int, [C.POINTER(StepperCfg), c_void, c_i64, C.POINTER(c_void)]),
    "tg_stepper_destroy": (None, [c_void]),
    "tg_stepper_regions": (C.c_int, [c_void, c_void, C.POINTER(c_i64)]),
    "tg_stepper_prepare_begin": (C.c_int, [c_void, C.c_int, c_void, c_void, c_i64]),
    "tg_stepper_prepare_finish": (C.c_int, [c_void, C.c_int, C.POINTER(c_i64)]),
    "tg_stepper_release": (C.c_int, [c_void, C.c_int]),
    "tg_stepper_set_graph": (C.c_int, [c_void, c_void]),
    "tg_stepper_slot_view": (C.c_int, [c_void, C.c_int, C.POINTER(c_void), C.POINTER(c_i64)]),
    "tg_stepper_forward": (C.c_int, [c_void, C.c_int, C.c_int, C.POINTER(C.c_uint64), c_void, C.POINTER(c_void)]),
    "tg_stepper_backward": (C.c_int, [c_void, C.c_int, c_void, c_void, GRAD_READY_FN, c_void, C.POINTER(AdamArgs), C.POINTER(c_void)]),
    "tg_stepper_tgn_prepare_begin": (C.c_int, [c_void, C.c_int, c_void, c_void, c_void, c_void, c_i64, c_i64, c_i64]),
    "tg_stepper_tgn_forward": (C.c_int, [c_void, C.c_int, C.POINTER(TgnBank), C.c_int, C.POINTER(C.c_uint64), c_void, C.POINTER(c_void), C.c_int]),
    "tg_stepper_tgn_backward": (C.c_int, [c_void, C.c_int, C.POINTER(TgnBank), c_void, C.c_int, c_void, C.POINTER(AdamArgs), C.POINTER(c_void),
                                          GRAD_READY_FN, c_void]),
    "tg_packed32_floats": (c_i64, [C.c_int, C.c_int]),
    "tg_pack32_weights": (C.c_int, [C.c_int, c_void, c_void]),
    "tg_gemm_pk_nt": (C.c_int, [c_i64, C.c_int, C.c_int, c_void, c_i64, c_void, c_void, c_i64, c_void, c_void]),
    "tg_dyg_arena_floats": (c_i64, [C.POINTER(DygCfg)]),
    "tg_dyg_create": (C.c_int, [C.POINTER(DygCfg), c_void, c_i64, C.POINTER(c_void)]),
    "tg_dyg_destroy": (None, [c_void]),
    "tg_dyg_regions": (C.c_int, [c_void, c_void, C.POINTER(c_i64)]),
    "tg_dyg_set_graph": (C.c_int, [c_void, c_void]),
    "tg_dyg_forward": (C.c_int, [c_void, c_void, c_void, c_void, c_i64, C.c_int, C.c_int, c_f32, C.POINTER(C.c_uint64), c_void, C.POINTER(c_void)]),
    "tg_dyg_backward": (C.c_int, [c_void, c_void, c_void, C.POINTER(AdamArgs), C.POINTER(c_void)]),
    "tg_add_layernorm_fwd_res": (C.c_int, [c_void, c_void, c_i64, C.c_int, c_void, c_void, c_f32, C.c_uint64, c_void, c_void, c_void, c_void, c_void]),
    "tg_add_layernorm_bwd_res": (C.c_int, [c_void, c_void, c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void, c_f32,
                                           C.c_uint64, c_void, c_i64, c_void]),
    "tg_adam_f32": (C.c_int, [c_void, c_void, c_void, c_void, c_i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_i64, c_void]),
    "tg_time_bias_finish": (C.c_int, [c_void, c_void, c_void, C.c_int, c_void]),
    "tg_bce_logits": (C.c_int, [c_void, c_i64, c_i64, c_void, c_void, c_void]),
    "tg_weighted_ce": (C.c_int, [c_void, c_i64, c_void, c_void, c_i64, C.c_int, c_void, c_void, c_i64, c_void]),
    "tg_weighted_sum": (C.c_int, [c_void, c_void, c_i64, c_f32, c_void, c_void]),
    "tg_hash_features": (C.c_int, [c_void, c_i64, c_i64, c_i64, C.c_int, C.c_uint64, c_void]),
    "tg_graph_create": (C.c_int, [c_void, c_void, c_void, c_void, c_i64, c_i64, C.POINTER(c_void)]),
    "tg_graph_destroy": (None, [c_void]),
    "tg_graph_num_rows": (c_i64, [c_void]),
    "tg_graph_num_entries": (c_i64, [c_void]),
    "tg_graph_export": (C.c_int, [c_void, c_void, c_void, c_void, c_void]),
    "tg_sample_recent": (C.c_int, [c_void, c_void, c_void, c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "tg_dedupe_capacity": (c_i64, [c_i64]),
    "tg_dedupe_pairs": (C.c_int, [c_void, c_void, c_i64, c_i64, c_void, c_void, c_void, c_i32, c_void, c_void, c_void, c_void, c_void]),
    "tg_graph_set_time_weights": (C.c_int, [c_void, C.c_double]),
    "tg_sample_random": (C.c_int, [c_void, c_void, c_void, c_void, c_i64, C.c_int, C.c_int, C.c_uint64, c_void, c_void, c_void, c_void, c_void, c_void]),
    "tg_host_count_before": (C.c_int, [c_void, c_void, c_i64, c_void, c_void, c_i64, c_void]),
    "tg_first_hop_window": (C.c_int, [c_void, c_void, c_void, c_i64, C.c_int, C.c_int, c_void, c_void, c_void, c_void, c_void]),
    "tg_time_encode": (C.c_int, [c_void, c_i64, c_void, c_void, C.c_int, C.c_int, c_void, c_void]),
    "tg_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), c_void, c_void, c_void, c_void]),
    "tg_attn_bwd_parts": (C.c_int, [c_i64]),
    "tg_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), c_void, c_void, c_void, c_void, c_void, c_void, c_i64, c_i64, c_void, c_i64, c_void, c_void]),
    "tg_set_attn_fast": (None, [C.c_int]),
    "tg_attn_dropped_scores": (C.c_int, [c_void, c_i64, C.c_int, C.c_int, c_f32, C.c_uint64, c_void, c_void]),
    "tg_set_overlap": (None, [C.c_int]),
    "tg_tgat_layer_fwd": (C.c_int, [C.POINTER(LayerDesc), c_void]),
    "tg_tgat_layer_wt_floats": (c_i64, [C.c_int, C.c_int, C.c_int]),
    "tg_tgat_layer_part_floats": (c_i64, [c_i64, C.c_int, C.c_int, C.c_int]),
    "tg_tgat_layer_vec_floats": (c_i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "tg_set_wgrad_grouped": (None, [C.c_int]),
    "tg_set_merged_min_rows": (None, [c_i64]),
    "tg_side_join": (C.c_int, [c_void]),
    "tg_set_layer_merged": (None, [C.c_int]),
    "tg_set_layer_chain": (None, [C.c_int]),
    "tg_tgat_layer_bwd": (C.c_int, [C.POINTER(LayerDesc), C.POINTER(LayerBwdDesc), c_void]),
    "tg_gemm_f32": (C.c_int, [C.c_int, C.c_int, c_i64, c_i64, c_i64, c_f32, c_void, c_i64, c_void, c_i64, c_void, c_i64,
                              c_void, C.c_int, C.c_int, c_void]),
    "tg_gemm_f32_nt_masked": (C.c_int, [c_i64, c_i64, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void]),
    "tg_packed_floats": (c_i64, [C.c_int, C.c_int]),
    "tg_pack_weights": (C.c_int, [C.c_int, C.POINTER(PackJob), c_void]),
    "tg_wgrad_group": (C.c_int, [C.c_int, C.POINTER(WgradJob), c_i64, c_void]),
    "tg_set_wgrad_form": (None, [C.c_int]),
    "tg_set_gemm_mode": (None, [C.c_int]),
    "tg_get_gemm_mode": (C.c_int, []),
    "tg_set_gemm_mode_thread": (None, [C.c_int]),
    "tg_get_gemm_mode_thread": (C.c_int, []),
    "tg_gemm_f32_batched": (C.c_int, [C.c_int, C.c_int, c_i64, c_i64, c_i64, c_f32, c_void, c_i64, c_i64, c_void, c_i64, c_i64, c_void,
                                      c_i64, c_i64, C.c_int, c_void, C.c_int, C.c_int, c_void]),
    "tg_gemm_f32_batched2": (C.c_int, [C.c_int, C.c_int, c_i64, c_i64, c_i64, c_f32, c_void, c_i64, c_i64, c_i64, c_void, c_i64, c_i64,
                                       c_i64, c_void, c_i64, c_i64, c_i64, C.c_int, C.c_int, C.c_int, c_void]),
    "tg_time_encode_masked": (C.c_int, [c_void, c_void, c_i64, c_void, c_void, C.c_int, c_void, c_void]),
    "tg_time_encode_bwd": (C.c_int, [c_void, c_void, c_i64, c_void, c_void, C.c_int, c_void, c_void, c_void]),
    "tg_cooccurrence": (C.c_int, [c_void, c_i64, C.c_int, c_void, c_i64, C.c_int, c_i64, c_void, c_void, c_void]),
    "tg_gelu_fwd": (C.c_int, [c_void, c_i64, c_void, c_void]),
    "tg_gelu_bwd": (C.c_int, [c_void, c_void, c_i64, c_void, c_void]),
    "tg_softmax_fwd": (C.c_int, [c_void, c_i64, C.c_int, c_void, c_void]),
    "tg_softmax_keymask_fwd": (C.c_int, [c_void, c_i64, C.c_int, c_void, c_i64, c_void, c_void]),
    "tg_tgn_rows_fwd": (C.c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, c_i64, c_void, C.c_int, C.c_int] + [c_void] * 4 + [C.c_int] + [c_void] * 6 + [c_void]),
    "tg_gru_gates_bwd_masked": (C.c_int, [c_void] * 6 + [c_i64, C.c_int, c_void, c_void, c_void]),
    "tg_tgn_host_advance": (C.c_int, [c_void, c_void, c_i64, c_void, c_void, c_void, c_i64, c_void]),
    "tg_tgn_prepare_layout": (C.c_int, [c_i64, c_i64, C.c_int, c_void]),
    "tg_tgn_prepare_batch": (C.c_int, [c_void] * 5 + [c_i64, c_i64, c_i64, C.c_int, c_i64] + [c_void] * 6 + [c_i64] + [c_void] * 11 + [c_void]),
    "tg_recent_window_mean": (C.c_int, [c_void, c_void, c_void, c_i64, C.c_int, c_void, c_i64, C.c_int, c_void, c_i64, c_void]),
    "tg_seq_attn_fwd": (C.c_int, [c_void, c_i64, C.c_int, C.c_int, C.c_int, c_f32, C.c_uint64, c_void, c_void, c_void]),
    "tg_seq_attn_bwd": (C.c_int, [c_void, c_void, c_void, c_i64, C.c_int, C.c_int, C.c_int, c_f32, C.c_uint64, c_void, c_void]),
    "tg_softmax_bwd": (C.c_int, [c_void, c_void, c_i64, C.c_int, c_void, c_void]),
    "tg_dropout": (C.c_int, [c_void, c_i64, c_f32, C.c_uint64, c_void, c_void]),
    "tg_gelu_dropout_fwd": (C.c_int, [c_void, c_i64, c_f32, C.c_uint64, c_void, c_void]),
    "tg_gelu_dropout_bwd": (C.c_int, [c_void, c_void, c_i64, c_f32, C.c_uint64, c_void, c_void]),
    "tg_dropout_add": (C.c_int, [c_void, c_void, c_i64, c_f32, C.c_uint64, c_void, c_void]),
    "tg_segment_mean_fwd": (C.c_int, [c_void, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "tg_segment_mean_bwd": (C.c_int, [c_void, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_void, c_void]),
    "tg_gather_rows": (C.c_int, [c_void, c_i64, c_void, c_i64, C.c_int, c_void, c_i64, c_void]),
    "tg_scatter_add_rows": (C.c_int, [c_void, c_i64, c_void, c_i64, C.c_int, c_void, c_i64, c_void]),
    "tg_add_layernorm_fwd": (C.c_int, [c_void, c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "tg_rowop_parts": (C.c_int, [c_i64]),
    "tg_add_layernorm_bwd": (C.c_int, [c_void, c_void, c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void, c_void, c_void]),
    "tg_colsum": (C.c_int, [c_void, c_i64, c_i64, C.c_int, c_void, C.c_int, c_void]),
    "tg_relu_bwd_inplace": (C.c_int, [c_void, c_void, c_i64, c_void]),
    "tg_gru_gates_fwd": (C.c_int, [c_void, c_void, c_void, c_i64, C.c_int, c_void, c_void]),
    "tg_gru_gates_bwd": (C.c_int, [c_void, c_void, c_void, c_void, c_i64, C.c_int, c_void, c_void, c_void, c_void]),
    "tg_tgn_persist": (C.c_int, [c_void, c_i64, c_void, c_void, c_void, c_void, c_void, c_i64, c_void, c_i64, C.c_int, c_void]),
    "tg_msg_scatter_last": (C.c_int, [c_void, c_void, c_i64, c_void, c_i64, C.c_int, c_void, c_i64, c_void, c_void, c_void, c_void]),
    "tg_build_messages": (C.c_int, [c_void, c_i64, c_void, c_void, c_void, c_void, c_void, c_i64, c_void, c_void, c_void, c_i64,
                                    C.c_int, C.c_int, C.c_int, c_void, c_void]),
}

_lib = None


class TgError(RuntimeError):
    pass


class TgShapeNotCovered(TgError):
    """TG_ESHAPE: the entry point does not cover this (valid) shape / alignment and launched nothing -- take the general form"""


def lib():
    """Load (once) and return the CDLL with typed entry points.  Raises if the HIP extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TgError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C flid_amd/csrc`.  flid_amd has no CPU fallback.")
    # torch first: it brings its own libamdhip64, and the library must bind to THAT runtime -- loaded before torch it pulls in the
    # system's copy, the process then holds two HIP runtimes and the second one finds "no ROCm-capable device"
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)      # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().tg_last_error().decode("utf-8", "replace")
        if rc == -1 and ("greater than 0" in msg or "greater than 1" in msg or "in the past" in msg):
            raise AssertionError(msg.split("invalid argument: ", 1)[-1])       # reference raises AssertionError there
        if rc == -5:
            raise TgShapeNotCovered(f"{what or 'libflid_tg'}: {msg}")
        if rc == -4:
            raise IndexError("list index out of range")                        # an id beyond the graph, as the reference's list lookup
        raise TgError(f"{what or 'libflid_tg'} failed (rc={rc}): {msg}")
