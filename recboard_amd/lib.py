"""ctypes binding of librecengine.so -- the C-ABI drop-in boundary (include/recengine.h).

The product path has NO fallback: if the shared library is missing or a call returns a negative code,
a RuntimeError is raised.  Nothing here imports `oracle/`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librecengine.so")

_vp, _i64, _i32, _f32, _u32, _sz, _f64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_uint32, ctypes.c_size_t, ctypes.c_double)

# name -> (restype, argtypes); mirrors include/recengine.h one to one
SIGNATURES = {
    "re_abi_version": (_i32, []),
    "re_error_string": (ctypes.c_char_p, [_i32]),
    "re_tile_wgs_per_cu": (_i32, [_i64]),
    "re_sasrec_tile_step_certain": (_i32, [_i64, _i64, _i64]),
    "re_gather_rows": (_i32, [_vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "re_sasrec_embed": (_i32, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _f32, _f32, _u32, _vp, _vp, _vp]),
    "re_sasrec_embed_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "re_sasrec_embed_bwd": (_i32, [_vp, _vp, _i64, _i64, _i64, _f32, _f32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "re_scatter_add_rows_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "re_scatter_plan": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _sz, _vp]),
    "re_scatter_apply": (_i32, [_vp, _i64, _i64, _i64, _f32, _vp, _i32, _vp, _sz, _vp]),
    "re_sparse_adam_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _f64, _vp, _sz, _vp]),
    "re_sparse_adam_rows_small": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64,
                                         _f64, _f64, _vp]),
    "re_sparse_adam_rows_dev": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _vp, _sz, _vp]),
    "re_scatter_add_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _i32, _vp, _sz, _vp]),
    "re_pair_loss_workspace_bytes": (_sz, [_i64]),
    "re_pair_loss_fwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_pair_loss_bwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "re_pair_loss_fwd_bwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "re_bpr_triplet_fwd_bwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_bpr_triplet_step_rows": (_i32, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_bpr_triplet_fwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "re_bpr_triplet_bwd": (_i32, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "re_score_dense": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "re_score_topk_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "re_score_topk": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "re_score_prepare_bytes": (_sz, [_i64, _i64]),
    "re_score_prepare": (_i32, [_vp, _i64, _i64, _vp, _sz, _vp]),
    "re_score_topk_prepared_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "re_score_topk_prepared": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "re_route_workspace_bytes": (_sz, [_i64, _i64]),
    "re_route_bucket": (_i32, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_sasrec_plan_bytes": (_sz, [_i64, _i64]),
    "re_sasrec_batch_prep": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u32, _i64,
                                    _f64, _f64, _f64, _vp]),
    "re_sasrec_batch_prep_w": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _u32, _i64,
                                      _f64, _f64, _f64, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp, _sz, _vp, _vp, _f32, _vp]),
    "re_seq_train_sample_prep": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _u32, _u32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _sz, _vp, _u32, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp, _sz, _vp, _vp, _f32, _vp]),
    "re_sasrec_tape_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "re_seq_train_sample": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "re_gen_train_sample": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _u32, _u32, _vp, _vp, _vp, _vp]),
    "re_sasrec_tape_layout": (_i32, [_i64, _i64, _i64, _i64, _vp, _i64]),
    "re_sasrec_encoder_fwd": (_i32, [_vp, _vp, _i64, _vp, _f32, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _u32, _vp, _vp, _i32,
                                     _vp, _vp, _sz, _i32, _vp]),
    "re_sasrec_encoder_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "re_sasrec_encoder_bwd_workspace_layout": (_i32, [_i64, _i64, _i64, _i64, ctypes.c_uint64, _vp]),
    "re_sasrec_encoder_bwd": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _u32, _vp, _vp, _vp, _i32, _f32, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_sasrec_plan_rows": (_i64, [_i64, _i64]),
    "re_sasrec_encoder_step": (_i32, [_vp, _i64, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _u32, _vp, _vp, _i32,
                                      _vp, _vp, _sz, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_sasrec_encoder_step_part": (_i32, [_vp, _i64, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _u32, _vp, _vp, _i32,
                                           _vp, _vp, _sz, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp, _vp]),
    "re_sasrec_encoder_fwd_loss": (_i32, [_vp, _i64, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _u32, _vp, _vp, _i32,
                                          _vp, _vp, _sz, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_sasrec_loss_rows_workspace_bytes": (_sz, []),
    "re_sasrec_loss_rows": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_scatter_add_rows_small": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _i64, _i64, _i64, _i64, _f32, _vp, _vp]),
    "re_scatter_adam_rows_small": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp]),
    "re_sasrec_step_tail": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp, _sz, _vp, _f32, _vp,
                                    _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "re_sasrec_step_tail_sparse": (_i32, [_vp, _vp, _i32, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _vp, _i64, _i64, _i64,
                                           _i64, _vp, _i32, _vp, _sz, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "re_sasrec_step_stage_sample": (_i32, [_vp, _u32, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _u32, _u32, _vp, _i64, _i64, _vp, _vp,
                                            _vp, _i64, _i64, _vp, _sz, _vp, _sz, _vp, _vp, _f32, _vp]),
    "re_sasrec_step_stage": (_i32, [_vp, _u32, _i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp, _sz,
                                     _vp, _vp, _f32, _vp]),
    "re_spmm_csr": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _f32, _vp, _f32, _vp, _sz, _vp]),
    "re_spmm_csr_split": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _f32, _vp, _f32, _vp, _sz,
                                 _vp]),
    "re_spmm_csr_masked": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _f32, _vp, _f32, _vp, _vp,
                                  _vp, _sz, _vp]),
    "re_row_mask": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "re_rows_sqnorm_workspace_bytes": (_sz, []),
    "re_rows_sqnorm": (_i32, [_vp, _i64, _i64, _vp, _i64, _f32, _vp, _i32, _vp, _sz, _vp]),
    "re_rank_metrics": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "re_grad_clip_workspace_bytes": (_sz, []),
    "re_grad_clip_coef": (_i32, [_vp, _i64, _f32, _vp, _vp, _sz, _vp]),
    "re_adam_step_scaled": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _f64, _vp, _f64, _f64, _f64, _f64, _vp, _vp]),
    "re_adam_step_clip2": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f64, _vp, _f64, _f64, _f64, _f64, _f64, _f32, _vp, _vp, _sz, _vp]),
    "re_adam_step_reduce": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _i64, _f64, _vp, _f64, _f64, _f64, _f64, _f64, _vp]),
    "re_score_pool": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "re_pool_topk": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "re_auc_workspace_bytes": (_sz, []),
    "re_auc": (_i32, [_vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "re_fm_bag_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "re_fm_bag_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "re_fm_table_grad": (_i32, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "re_bce_logits": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "re_gemm_f32_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "re_gemm_f32": (_i32, [_i32, _i32, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _i32, _vp, _sz, _vp]),
    "re_gemm_f32_slabs": (_i32, [_i32, _i32, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _sz, _vp, _vp]),
    "re_gemm_splitk_reduce_many": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "re_ce_rows": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "re_ce_chunk_stats": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "re_ce_chunk_loss": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "re_ce_chunk_grad": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "re_gemm_f32_colstats": (_i32, [_i32, _i32, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "re_bn_relu_drop_fwd_pre": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _u32, _vp, _u32, _vp, _vp, _vp, _i32, _vp]),
    "re_bn_relu_drop_fwd": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _u32, _vp, _u32, _vp, _vp, _vp, _sz, _vp]),
    "re_step_state": (_i32, [_vp, _u32, _i64, _f64, _f64, _f64, _vp]),
    "re_step_stage_inputs": (_i32, [_vp, _u32, _i64, _f64, _f64, _f64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "re_mlp_workspace_bytes": (_sz, [_i64]),
    "re_mlp_head_workspace_bytes": (_sz, [_i64, _i64]),
    "re_mlp_head_fwd": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_gemm_f32_gated": (_i32, [_i32, _i32, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _f32, _vp, _vp]),
    "re_mlp_head_bwd_gated": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _f32, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "re_bn_bwd_apply": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "re_mlp_head_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "re_bn_relu_drop_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "re_colsum": (_i32, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "re_scale_copy": (_i32, [_vp, _vp, _f32, _i64, _vp]),
    "re_adam_step_dev": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f64, _f64, _f64, _f64, _vp]),
    "re_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64, _vp]),
}

_LIB = None


def load():
    """Load librecengine.so (built by `__graft_entry__.build()` / `make -C recboard_amd/csrc`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"recengine: {LIB_PATH} not found -- the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C recboard_amd/csrc`.")
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7); it must be the one already mapped when
    # librecengine.so resolves the same SONAME, otherwise the process ends up with two HIP runtimes and every
    # launch on a torch stream fails.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


RE_EUNSUPPORTED = -4      # (include/recengine.h)


def check(code, what):
    if code != 0:
        msg = load().re_error_string(code).decode()
        raise RuntimeError(f"recengine: {what} failed: {msg} (code {code})")
