"""Command-line surface of the reference trainer (MF/parse.py:3-117), kept flag-for-flag so that the commands
in the reference README (README.md:41,69,93) run unchanged against `python -m pda_amd.train_new_api`.

Flags the reference marks "not used" are accepted and ignored (they still have to parse).  Three flags are
additions of this implementation and default to the reference's behaviour; they are listed last.
"""
from __future__ import annotations

import argparse

# (name, type, default, help)   -- type None means "nargs='?' string" exactly like the reference
_REFERENCE_FLAGS = [
    ("data_path", None, "./data/", "root directory that holds <dataset>/"),
    ("dataset", None, "kwai", "dataset directory name"),
    ("source", None, "normal", "(unused)"),
    ("train", None, "normal", "normal (BPRMF) | s_condition (PD/PDA)"),
    ("test", None, "normal", "normal (BPRMF/BPRMF-A) | s_condition (PD/PDA)"),
    ("valid_set", None, "test", "test | valid"),
    ("save_dir", None, "/data/zyang/save_model/", "checkpoint root"),
    ("alpha", float, 1e-3, "(unused; appears in the checkpoint directory name)"),
    ("beta", float, 1e-3, "(unused)"),
    ("pc_alpha", float, 0.1, "(unused)"),
    ("pc_beta", float, 0.1, "(unused)"),
    ("exp_init_values", float, 0.1, "(unused)"),
    ("pop_exp", float, 0.1, "popularity exponent gamma"),
    ("early_stop", int, 1, "1: stop when recall@Ks[0] stalls"),
    ("need_save", int, 1, "(unused)"),
    ("cores", int, 1, "(unused)"),
    ("verbose", int, 1, "print every `verbose` epochs between evaluations"),
    ("epoch", int, 400, "number of epochs"),
    ("load_epoch", int, 400, "(unused)"),
    ("embed_size", int, 64, "embedding width d"),
    ("batch_size", int, 1024, "triplets per step"),
    ("Ks", None, "[20]", "python list literal of cut-offs, max 50"),
    ("epochs", None, "[]", "(unused)"),
    ("regs", float, 1e-5, "L2 coefficient"),
    ("fregs", float, 1e-5, "(unused)"),
    ("c", float, 10.0, "(unused)"),
    ("train_c", str, "val", "(unused)"),
    ("lr", float, 1e-3, "learning rate"),
    ("wd", float, 1e-5, "(overwritten by --regs, MF/train_new_api.py:1020)"),
    ("model", None, "mf", "only 'mf' is implemented"),
    ("skew", int, 0, "(unused)"),
    ("model_type", None, "o", "(unused)"),
    ("devide_ratio", float, 0.8, "(unused)"),
    ("save_flag", int, 1, "1: also checkpoint every 50 epochs"),
    ("pop_used", int, -2, "(unused)"),
    ("cuda", str, "1", "visible GPU id (HIP_VISIBLE_DEVICES)"),
    ("pretrain", int, 0, "only 0 is implemented"),
    ("check_c", int, 1, "(unused)"),
    ("log_interval", int, 10, "evaluate every this many epochs"),
    ("pop_wd", float, 0.0, "(unused)"),
    ("base", float, -1.0, "(unused)"),
    ("cf_pen", float, 1.0, "(unused)"),
    ("saveID", None, "", "suffix of the checkpoint directory"),
    ("user_min", int, 1, "(unused)"),
    ("user_max", int, 1000, "(unused)"),
    ("data_type", None, "ori", "only 'ori' is implemented"),
    ("imb_type", None, "exp", "(unused)"),
    ("top_ratio", float, 0.1, "(unused)"),
    ("lam", float, 1.0, "(unused)"),
    ("check_epoch", None, "all", "(unused)"),
    ("start", float, -1.0, "(unused)"),
    ("end", float, 1.0, "(unused)"),
    ("step", int, 20, "(unused)"),
    ("out", int, 0, "(unused)"),
]

_EXTENSION_FLAGS = [
    ("optimizer", str, "adam", "adam = TF-1.14 dense-decay Adam (reference, MF/model_api.py:83) | lazy_adam | sgd (exact mini-batch step) | sgd_fused (one launch, asynchronous in-kernel update)"),
    ("adam_sweep", str, "auto", "how --optimizer adam applies the reference's dense decay: sweep = one pass over both tables per step | replay = the same arithmetic without the sweep (idle rows replay their decay when next needed; bit-identical after the sync) | replay_fast = that catch-up to 1e-6 instead of bit for bit (~4x less arithmetic; explicit opt-in) | auto = sweep up to 64 MB of tables, the bit-identical replay above"),
    ("sampler", str, "device", "device = HIP counter-based sampler | host = the reference's Python generators"),
    ("table_dtype", str, "f32", "f32 | bf16 (BASELINE config 5): bf16 embedding tables for the forward pass and the evaluation, fp32 masters take the updates"),
    ("eval_block", int, 262144, "users per score+top-K launch (the reference always uses 2048, MF/train_new_api.py:703); large blocks balance the early-terminating sweep: 92 M users/s at 65536, 109 M at 262144 (C3)"),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Run pop_bias (PDA BPR-MF) on MI355X.")
    for name, typ, default, hlp in _REFERENCE_FLAGS + _EXTENSION_FLAGS:
        if typ is None:
            p.add_argument("--" + name, nargs="?", default=default, help=hlp)
        else:
            p.add_argument("--" + name, type=typ, default=default, help=hlp)
    return p


def parse_args(argv=None) -> argparse.Namespace:
    return build_parser().parse_args(argv)


def reference_flag_names():
    return [f[0] for f in _REFERENCE_FLAGS]
