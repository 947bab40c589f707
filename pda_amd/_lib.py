"""ctypes binding of libpda_hip.so (the C ABI declared in include/pda_hip.h).

There is NO CPU fallback: if the shared object is missing or a symbol is absent this module raises.
Device pointers come from torch ROCm tensors (``tensor.data_ptr()``); the launch stream is torch's
current HIP stream, so torch.cuda.Event / torch.cuda.graph see every kernel.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDA_HIP_LIB") or os.path.join(_HERE, "csrc", "libpda_hip.so")   # PDA_HIP_LIB: an A/B build of the same library (tools/)

# Constants mirrored from include/pda_hip.h
ABI_VERSION = 2
HEAD_RAW, HEAD_POP = 0, 1
HIST_BY_BLOCK_ROW, HIST_BY_USER_ID = 0, 1
UPD_NONE, UPD_SGD_FUSED, UPD_DENSE_GRAD = 0, 1, 2
UPD_ANY_ORDER = 0x100
UPD_USERS_DISTINCT = 0x200
ADAM_REPLAY_FAST = 0x10
MAX_K = 64
TOPK_CAP = 60

_vp, _i, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t

class ScorePlan(C.Structure):
    """struct pda_score_plan of include/pda_hip.h (pda_score_topk_plan)."""
    _fields_ = [("path", _i), ("sweep_mode", _i), ("n_splits", _i), ("early_stop", _i), ("order", _i), ("prep_with_pop", _i),
                ("workspace_bytes", _sz), ("keys_rows", _sz)]


class SampleJob(C.Structure):
    """struct pda_sample_job of include/pda_hip.h (arguments of pda_sample_triplets_dev, for pda_bpr_step_sample_f32)."""
    _fields_ = [("users", _vp), ("gen_users", _i), ("user_pool", _vp), ("n_pool", _i), ("B", _i),
                ("train_indptr", _vp), ("train_indices", _vp), ("train_slots", _vp),
                ("neg_lo", _i), ("neg_hi", _i), ("pop_matrix", _vp), ("n_slots", _i), ("seed", _u64),
                ("step_dev", _vp), ("step_next", _vp),
                ("pos", _vp), ("neg", _vp), ("pos_pop", _vp), ("neg_pop", _vp)]


# name -> (restype, argtypes); exactly the declarations of include/pda_hip.h
SIGNATURES = {
    "pda_abi_version": (_i, []),
    "pda_error_string": (C.c_char_p, [_i]),
    "pda_score_topk_auto_splits": (_i, [_i, _i]),
    "pda_score_topk_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pda_item_prep_bytes": (_sz, [_i, _i]),
    "pda_item_prep_f32": (_i, [_vp, _i, _i, _vp, _vp]),
    "pda_score_topk_workspace_bytes": (_sz, [_i]),
    "pda_score_topk4_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pda_score_topk_prepped_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pda_item_prep_ordered_bytes": (_sz, [_i, _i]),
    "pda_item_prep_ordered_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pda_item_prep_ordered_check": (_i, [_vp, _i, _i, _vp]),
    "pda_hist_reorder": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "pda_score_topk_ordered_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pda_item_prep_bf16_bytes": (_sz, [_i, _i]),
    "pda_item_prep_bf16": (_i, [_vp, _i, _i, _vp, _vp]),
    "pda_score_topk_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pda_item_prep_ordered_bf16_bytes": (_sz, [_i, _i]),
    "pda_item_prep_ordered_bf16": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pda_score_topk_ordered_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pda_item_prep4_bytes": (_sz, [_i, _i]),
    "pda_item_prep4_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pda_item_prep4_bf16": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pda_item_prep7_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pda_item_prep7_bf16": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pda_item_prep4_check": (_i, [_vp, _i, _i, _vp]),
    "pda_score_topk4_auto_splits": (_i, [_i, _i, _i]),
    "pda_score_topk4_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pda_score_topk4_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pda_peak_mfma_flops_per_launch": (C.c_double, [_i]),
    "pda_peak_mfma_bf16": (_i, [_vp, _i, _vp]),
    "pda_peak_mfma_bf16_const": (_i, [_vp, _i, _vp]),
    "pda_peak_mfma_lds_flops_per_launch": (C.c_double, [_i]),
    "pda_peak_mfma_lds_bf16": (_i, [_vp, _sz, _vp, _i, _vp]),
    "pda_peak_copy": (_i, [_vp, _vp, _sz, _vp]),
    "pda_topk_kth_value": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "pda_topk_remap_items": (_i, [_vp, _sz, _vp, _i, _vp]),
    "pda_topk_seed_refine": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "pda_topk_seed_bounds": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "pda_topk_seed_counts": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "pda_topk_seed_pick": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pda_score_topk4_phase_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pda_score_topk4_phase_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pda_score_topk_plan": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pda_score_topk_huge_splits": (_i, [_i, _i, _i]),
    "pda_score_topk7_workspace_bytes": (_sz, [_i, _i, _i]),
    "pda_score_topk7_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "pda_score_topk7_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "pda_topk_merge": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pda_bpr_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pda_triplet_plan_bytes": (_sz, [_i]),
    "pda_bpr_step_plan_scratch_bytes": (_sz, [_i, _i]),
    "pda_triplet_plan": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "pda_triplet_plan_large_workspace_bytes": (_sz, [_i]),
    "pda_triplet_plan_large": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "pda_bpr_step_plan_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _i, _vp, _vp]),
    "pda_bpr_step_plan_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "pda_sgd_apply_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "pda_bpr_step_shard_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _i, _vp, _vp, _vp]),
    "pda_apply_user_grads_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pda_bpr_step_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pda_refresh_rows_bf16": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "pda_group_triplets_by_pos": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pda_sort_triplets_by_pos": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pda_adam_dense_sweep_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    "pda_adam_dense_sweep2_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _vp]),
    "pda_adam_mark_rows": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "pda_adam_dense_sweep3_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _f, _f, _f, _f, _vp]),
    "pda_score_dense_f32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "pda_adam_dense_sweep4_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _f, _f, _f, _f, _i, _vp]),
    "pda_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _i, _f, _f, _f, _f,
                               _i, _i, _vp, _vp]),
    "pda_adam_rows_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp]),
    "pda_adam_lazy_f32": (_i, [_i] + [_vp] * 13 + [_i, _i, _i, _vp, _f, _f, _f, _vp]),
    "pda_adam_lazy_dev_f32": (_i, [_i] + [_vp] * 13 + [_i, _i, _vp, _vp, _vp, _i, _f, _f, _f, _vp]),
    "pda_adam_lazy_sync_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _vp, _f, _f, _f, _vp]),
    "pda_adam_lazy_sync_fast_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _vp, _f, _f, _f, _vp]),
    "pda_metrics": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "pda_sample_triplets_dev": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pda_sample_batches_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pda_group_triplets_by_pos_batches": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "pda_counter_add": (_i, [_vp, _u64, _vp]),
    "pda_bpr_step_sample_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp]),
    "pda_bpr_train_steps_f32": (_i, [_vp, _vp, _i, _f, _f, _f, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "pda_sample_triplets": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class PdaHipError(RuntimeError):
    pass


def load():
    """Load (once) and type every entry point.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PdaHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pda_amd/csrc`.  pda_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PdaHipError(f"libpda_hip.so does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    got = lib.pda_abi_version()
    if got != ABI_VERSION:
        raise PdaHipError(f"libpda_hip.so ABI {got} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().pda_error_string(code).decode()
        raise PdaHipError(f"{what}: {msg} ({code})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
