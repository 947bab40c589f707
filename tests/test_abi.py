"""CPU suite, part 2: the C-ABI library loads here (no GPU) and exports exactly what include/pda_hip.h declares;
argument validation returns error codes before anything is launched."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_in(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pda_[a-z0-9_]+)\s*\(", text)))


def declared_symbols():
    """Both headers: the stable drop-in surface and what pda_amd uses beyond it."""
    stable, exp = declared_in("pda_hip.h"), declared_in("pda_hip_experimental.h")
    assert not set(stable) & set(exp), "a symbol is declared in both headers"
    return sorted(stable + exp)


def test_the_stable_header_is_the_drop_in_surface_and_integration_section_2_needs_nothing_else():
    """the plan, the preps and score calls a plan may name, the merge, the train step, the reference's optimiser, metrics, sampler
    are in include/pda_hip.h; phases / seeds / planned and looped steps / lazy Adam / look-ahead / peaks are not.  The stub INTEGRATION.md section 2
    tells a maintainer of the reference to write binds stable symbols only (the blocks under "Beyond the stable header" are marked as such)."""
    stable, exp = set(declared_in("pda_hip.h")), set(declared_in("pda_hip_experimental.h"))
    core = {"pda_abi_version", "pda_error_string", "pda_score_topk_plan", "pda_item_prep4_f32", "pda_item_prep4_bf16", "pda_item_prep7_f32",
            "pda_score_topk4_f32", "pda_score_topk4_bf16", "pda_score_topk7_f32", "pda_score_topk7_bf16", "pda_score_topk_f32", "pda_topk_merge",
            "pda_bpr_step_f32", "pda_bpr_step_bf16", "pda_sgd_apply_f32", "pda_adam_step_f32", "pda_adam_dense_sweep_f32", "pda_adam_dense_sweep2_f32",
            "pda_adam_dense_sweep4_f32", "pda_metrics", "pda_sample_triplets", "pda_sample_triplets_dev", "pda_bpr_step_shard_f32"}
    assert core <= stable, core - stable
    for n in exp:
        assert re.search(r"phase|seed|kth|remap|huge_splits|plan|sort_|group_|mark_rows|sweep3|adam_rows|lazy|sample_f32|batches|train_steps|peak", n), n
    assert len(stable) <= 50 and len(exp) >= 30
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec2 = text[text.index("## 2."):text.index("## 3.")]
    head, beyond = sec2.split("### Beyond the stable header")
    used = set(re.findall(r"_lib\.(pda_[a-z0-9_]+)", head))
    assert used and used <= stable, used - stable
    assert set(re.findall(r"_lib\.(pda_[a-z0-9_]+)", beyond)) & exp


def test_library_exports_every_declared_symbol():
    from pda_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "libpda_hip.so does not export " + n
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.pda_abi_version() == _lib.ABI_VERSION == 2


def test_error_strings_and_argument_checks_without_gpu():
    from pda_amd import _lib
    lib = _lib.load()
    assert lib.pda_error_string(0) == b"ok"
    assert lib.pda_error_string(-1) == b"invalid argument"
    assert lib.pda_error_string(-2).startswith(b"unsupported")
    null = C.c_void_p(None)
    # null tables / bad sizes are rejected before any HIP call
    assert lib.pda_score_topk_f32(null, null, null, null, 0, 0, 0, 64, null, null, 0, 50, 0, 1, null, null) == -1
    assert lib.pda_topk_merge(null, 1, 1, 50, null, null, null, null, null, null, 0, null) == -1
    assert lib.pda_bpr_step_f32(null, null, null, null, null, null, null, 0, 64, 0.0, 1.0, 0.0, 0, null, null, null, null, null, null, null) == -1
    assert lib.pda_adam_dense_sweep_f32(null, null, null, null, 0, 0.0, 0.9, 0.999, 1e-8, null) == -1
    assert lib.pda_metrics(null, 0, 50, null, null, null, 0, null, null) == -1
    assert lib.pda_sample_triplets(null, 0, null, 0, 0, null, null, null, 0, 0, null, 0, 0, 0, null, null, null, null, null) == -1
    assert lib.pda_score_topk_auto_splits(2048, 200000) == 32
    assert lib.pda_score_topk_auto_splits(1 << 20, 200000) == 1


def test_product_refuses_to_run_without_the_extension(monkeypatch):
    from pda_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpda_hip.so")
    try:
        _lib.load()
        raise AssertionError("load() must fail loudly")
    except _lib.PdaHipError as e:
        assert "no CPU fallback" in str(e)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pda_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
