"""ctypes binding of ``csrc/librechub_hip.so`` (the C ABI declared in ``include/rechub_hip.h``).

There is deliberately no CPU fallback: if the shared library is missing or a symbol cannot be
resolved the import of any op fails loudly (``RuntimeError``), and every op refuses tensors that
are not on a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RECHUB_HIP_LIB: load another build of the same ABI (kernel experiments); default is the in-tree library
LIB_PATH = os.environ.get("RECHUB_HIP_LIB") or os.path.join(_HERE, "csrc", "librechub_hip.so")

c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p

# name -> argtypes; every function returns int (0 ok) unless listed in _RESTYPES
SIGNATURES = {
    "rh_abi_version": [],
    "rh_last_error": [],
    "rh_set_tuning": [c_int, c_int],
    "rh_embed_fwd": [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr,
                     c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr],
    "rh_embed_bwd": [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                     c_ptr, c_ptr, c_f32, c_int, c_ptr, c_int, c_ptr, c_ptr],
    "rh_embed_bwd_nchunks": [c_int, c_int],
    "rh_embed_scatter_rows": [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_f32, c_int, c_ptr, c_ptr],
    "rh_fm_fwd": [c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_fm_bwd": [c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_i64, c_ptr],
    "rh_seq_pool_fwd": [c_ptr, c_i64, c_ptr, c_int, c_i64, c_i64, c_int, c_int, c_int, c_int, c_i64, c_ptr, c_i64,
                        c_ptr, c_ptr],
    "rh_seq_pool_bwd": [c_ptr, c_i64, c_ptr, c_int, c_i64, c_i64, c_int, c_int, c_int, c_int, c_i64, c_i64, c_ptr,
                        c_i64, c_f32, c_ptr, c_ptr],
    "rh_cross_fwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr],
    "rh_cross_bwd_nblocks": [c_int],
    "rh_cross_max_layers": [c_int],
    "rh_cross_bwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr,
                     c_i64, c_int, c_ptr, c_ptr],
    "rh_cross_v2_epilogue_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr],
    "rh_cross_v2_epilogue_bwd": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_cross_mix_epilogue_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_cross_mix_epilogue_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_cross_mix_nblocks": [c_int],
    "rh_cross_mix_epilogue_bwd_b": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                    c_ptr],
    "rh_cross_moe_kp": [c_int, c_int],
    "rh_cross_moe_supported": [c_int, c_int, c_int, c_int],
    "rh_cross_moe_mid_blocks": [c_int, c_int, c_int],
    "rh_cross_moe_pack": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_cross_moe_mid_fwd": [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_cross_moe_mid_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_cross_moe_res_bwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_cross_moe_unpack": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                            c_ptr, c_ptr],
    "rh_dice_nblocks": [c_i64],
    "rh_dice_fwd": [c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_dice_bwd": [c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr],
    "rh_bn_stats_fwd": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_int, c_ptr, c_ptr, c_ptr],
    "rh_bn_finalize_bwd": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_bn_finalize_bwd_tail": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr],
    "rh_bn_dice_stats_blocks": [c_i64],
    "rh_bn_dice_bwd_stats": [c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_bn_dice_bwd_apply": [c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_bn_dice_head_fwd": [c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_bn_dice_head_bwd_stats": [c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_bn_dice_head_bwd_apply": [c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_din_att_input_fwd": [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_din_att_input_bwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_din_pool_fwd": [c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_din_pool_bwd": [c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_inbatch_logits_fwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_inbatch_logits_bwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr,
                              c_ptr],
    "rh_prelu_nblocks": [c_i64],
    "rh_prelu_fwd": [c_ptr, c_ptr, c_i64, c_ptr, c_ptr],
    "rh_prelu_bwd": [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr],
    "rh_din_att_l1_supported": [c_int, c_int],
    "rh_din_att_l1_chunk_rows": [c_i64],
    "rh_din_att_l1_fwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_bn_stats_from_partial": [c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_ptr, c_ptr],
    "rh_gemm_stats_rows": [c_int, c_int],
    "rh_linear_fwd": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                      c_ptr],
    "rh_linear_fwd_gate": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr,
                           c_ptr, c_ptr],
    "rh_linear_dgrad": [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr],
    "rh_cross_v2_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_cross_v2_dgrad": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr],
    "rh_gemm_chain_stats_rows": [c_int],
    "rh_linear_bnact_fwd": [c_ptr, c_i64, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_ptr,
                            c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_int, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_linear_dgrad_bnbwd": [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                              c_f32, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_head_bnact_fwd": [c_ptr, c_i64, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_ptr, c_ptr, c_ptr,
                          c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_linear_wgrad_workspace": [c_int, c_int, c_int],
    "rh_linear_wgrad_tiles": [c_int, c_int],
    "rh_linear_wgrad": [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_linear_wgrad_splits": [c_int, c_int, c_int],
    "rh_linear_wgrad_partial": [c_ptr, c_i64, c_ptr, c_i64, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_linear_wgrad_partial_group": [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_head_bwd_ex": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                       c_int, c_ptr],
    "rh_head_bwd_bn": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int,
                       c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_int, c_ptr, c_ptr],
    "rh_head_bwd_bn_scalars": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                               c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr,
                               c_ptr, c_ptr, c_int, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr],
    "rh_bn_relu_dropout_bwd_pre": [c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr,
                                   c_ptr, c_ptr, c_int, c_ptr],
    "rh_pack_grads": [c_ptr, c_int, c_ptr, c_ptr],
    "rh_pack_grads_adam": [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_pack_grads_adam_gate": [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_head_nblocks": [c_int],
    "rh_head_fwd": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr],
    "rh_head_bwd": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_head_loss_nblocks": [c_int],
    "rh_head_loss_fwd": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_head_loss_bwd": [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_step_scalars": [c_ptr, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64,
                        c_ptr],
    "rh_colsum": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr],
    "rh_bce_fwd": [c_ptr, c_ptr, c_i64, c_ptr, c_ptr],
    "rh_bce_bwd": [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr],
    "rh_bn_act_nchunks": [c_int],
    "rh_bn_relu_dropout_fwd": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_int, c_ptr,
                               c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr],
    "rh_bn_relu_dropout_bwd": [c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                               c_ptr, c_int, c_ptr],
    "rh_bn_prelu_dropout_fwd": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_int, c_ptr,
                                c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_bn_prelu_nblocks": [c_int, c_int],
    "rh_bn_prelu_dropout_bwd": [c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_adam_prepare": [c_ptr, c_ptr, c_ptr, c_int, c_ptr],
    "rh_adam_small": [c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_adam_lazy_step": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr,
                          c_ptr],
    "rh_adam_lazy_step_mode": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int,
                               c_ptr, c_int, c_ptr],
    "rh_adam_lazy_step_mode_idx": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_int,
                                   c_ptr, c_int, c_ptr],
    "rh_adam_lazy_touched": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_int, c_int,
                             c_ptr, c_ptr],
    "rh_adam_lazy_touched_group": [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int,
                                   c_ptr, c_ptr],
    "rh_adam_lazy_refresh_assemble": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr,
                                      c_ptr, c_i64, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr],
    "rh_adam_sweep_stagger": [c_ptr],
    "rh_adam_sweep_gate": [c_ptr, c_i64, c_i64, c_ptr, c_ptr],
    "rh_adam_sweep_gate_done": [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_ptr],
    "rh_host_device_pointer": [c_ptr, c_ptr],
    "rh_adam_sweep_gate_open": [c_ptr, c_ptr],
    "rh_adam_sweep_release": [c_ptr, c_ptr],
    "rh_adam_lazy_step_ahead": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr,
                                c_ptr, c_i64, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr],
    "rh_adam_lazy_step_ahead_touched": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr,
                                        c_ptr, c_ptr, c_i64, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr,
                                        c_int, c_ptr],
    "rh_adam_lazy_step_ahead_wgrad": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr,
                                      c_ptr, c_ptr, c_i64, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int,
                                      c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_adam_lazy_sweep": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_i64, c_ptr],
    "rh_l2norm_fwd": [c_ptr, c_i64, c_int, c_int, c_f32, c_ptr, c_ptr, c_ptr],
    "rh_l2norm_bwd": [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_f32, c_ptr, c_ptr],
    "rh_ce_nblocks": [c_int],
    "rh_ce_fwd": [c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_ce_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr],
    "rh_adam_dense": [c_ptr, c_int, c_ptr, c_ptr, c_int, c_ptr],
    "rh_batch_gather": [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_batch_advance": [c_ptr, c_i64, c_i64, c_ptr],
    "rh_inbatch_sample": [c_ptr, c_int, c_int, c_ptr, c_ptr],
    "rh_inbatch_sample_rows": [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_augru_max_dim": [],
    "rh_augru_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr],
    "rh_augru_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr],
    "rh_shard_localize": [c_ptr, c_int, c_i64, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "rh_shard_narrow": [c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr],
}
_RESTYPES = {"rh_last_error": ctypes.c_char_p, "rh_linear_wgrad_workspace": ctypes.c_int64}
# functions whose int return value is a result, not a status
_VALUE_RETURNING = {"rh_abi_version", "rh_embed_bwd_nchunks", "rh_cross_bwd_nblocks", "rh_cross_max_layers",
                    "rh_bn_act_nchunks", "rh_dice_nblocks", "rh_linear_wgrad_workspace", "rh_linear_wgrad_tiles", "rh_linear_wgrad_splits",
                    "rh_head_nblocks", "rh_ce_nblocks", "rh_bn_prelu_nblocks", "rh_head_loss_nblocks", "rh_cross_mix_nblocks", "rh_cross_moe_kp", "rh_cross_moe_supported", "rh_cross_moe_mid_blocks", "rh_din_att_l1_supported", "rh_din_att_l1_chunk_rows", "rh_prelu_nblocks", "rh_gemm_stats_rows", "rh_gemm_chain_stats_rows", "rh_bn_dice_stats_blocks", "rh_augru_max_dim"}

ABI_VERSION = 1
_lib = None


def ab(name, default=True):
    """Same-box A/B switches of benchmarks, ONE environment variable: RECHUB_AB="chain=0,headside=0,assemble=0" turns the named
    round-4 paths off (each has a bit- or tolerance-pinned twin; the tests flip the module attributes instead)."""
    for item in filter(None, os.environ.get("RECHUB_AB", "").split(",")):
        k, _, v = item.partition("=")
        if k.strip() == name:
            return v.strip() not in ("0", "false", "off")
    return default


class PackItem(ctypes.Structure):
    """RhPackItem of include/rechub_hip.h (one parameter's gradient sources for rh_pack_grads)."""
    _fields_ = [("src", ctypes.c_uint64), ("add", ctypes.c_uint64), ("nparts", c_i64), ("stride", c_i64),
                ("numel", c_i64), ("dst_offset", c_i64)]


def load():
    """Load the shared library once; raise RuntimeError (never fall back) when it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"librechub_hip.so not found at {LIB_PATH}. Build it with "
                           f"`python -c 'import __graft_entry__ as g; g.build()'` or `{_HERE}/csrc/build.sh` "
                           "(there is no CPU fallback for the HIP hot path).")
    # torch must be imported first so that the HIP runtime already mapped by torch (same SONAME
    # libamdhip64.so.7) is the one this library binds to: one runtime, shared streams.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"librechub_hip.so does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    got = lib.rh_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"librechub_hip.so ABI {got} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    # RECHUB_TUNE="key=value,key=value": rh_set_tuning knobs of include/rechub_hip.h (kernel experiments)
    for item in filter(None, os.environ.get("RECHUB_TUNE", "").split(",")):
        k, v = item.split("=")
        if lib.rh_set_tuning(int(k), int(v)) != 0:
            raise RuntimeError(f"RECHUB_TUNE: rh_set_tuning({k}, {v}) rejected")
    return lib


def call(name, *args):
    """Invoke a status-returning entry point; raise RuntimeError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _VALUE_RETURNING:
        return rc
    if rc != 0:
        msg = lib.rh_last_error()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")
    return 0
