"""The C-ABI library builds, loads without a GPU and exports exactly what include/wft.h declares."""
import ctypes
import re
from pathlib import Path

from whisper_finetune.engine import lib as L

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "wft.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wft_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_functions():
    names = _declared()
    assert len(names) >= 20
    assert "wft_gemm_nt_bf16" in names and "wft_attn_bwd_bf16" in names and "wft_logmel" in names


def test_library_exports_every_declared_symbol():
    assert L.LIB_PATH.exists(), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(str(L.LIB_PATH))
    for name in _declared():
        assert hasattr(handle, name), f"{name} declared in wft.h but not exported by libwft.so"


def test_python_binding_covers_the_header():
    assert sorted(L.SIGNATURES) == _declared()
    handle = L.load()
    assert handle.wft_version().decode().endswith("gfx950")
    assert handle.wft_layernorm_bwd_workspace(1000, 384) > 0  # pure host function: callable without a GPU


def test_struct_sizes_match_header_layout():
    # 8-byte aligned C structs: pointers/int64 interleaved with ints exactly as in wft.h
    assert ctypes.sizeof(L.GemmArgs) % 8 == 0 and ctypes.sizeof(L.AttnArgs) % 8 == 0
    assert L.GemmArgs.M.offset - L.GemmArgs.alpha.offset == 4
    assert L.AttnArgs.d_o.offset % 8 == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", tmp_path / "nope.so")
    try:
        L.load()
    except L.WftError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when libwft.so is absent")
    finally:
        monkeypatch.undo()
        L._lib = None
        L.load()


def test_shipped_library_cannot_read_timing_switches():
    """WFT_GEMM_DIAG=6/22/23/24 drop stores / epilogues "for timing only", the *_VARIANT / *_VAR / thresholds are A/B switches: they
    exist only in libwft_timing.so (`make TIMING=1`, -DWFT_TIMING_BUILDS).  The shipped library does not contain their names at
    all, so no environment can change what it computes; the two launch-mode variables of a multi-GPU job are the only ones it
    reads, and wft_version() says which build is loaded."""
    blob = L.LIB_PATH.read_bytes()
    names = sorted(set(m.decode() for m in re.findall(rb"WFT_[A-Z0-9_]{3,}", blob)) - {"WFT_EPI_DGELU", "WFT_EPI_GELU_GRAD", "WFT_EPI_MUL_AUX"})
    assert names == ["WFT_ATTN_PERSISTENT", "WFT_NT256_PERSISTENT"], names  # (WFT_EPI_*: enum names inside error messages)
    # every getenv of a developer switch in the sources goes through wft_dev_getenv (compiled to nullptr without the flag)
    for src in (ROOT / "whisper-finetune_amd" / "csrc").glob("*.hip"):
        for var in re.findall(r'[^_]getenv\("(WFT_[A-Z0-9_]+)"\)', src.read_text()):
            assert var in ("WFT_ATTN_PERSISTENT", "WFT_NT256_PERSISTENT"), (src.name, var)
    assert b"timing-builds" not in blob
    assert L.load().wft_version().decode() == "wft 0.1 gfx950"


def test_library_holds_no_mutable_launch_state():
    """launch mode and kernel variants are per-call fields of the argument structs; no setter is exported."""
    handle = L.load()
    for name in ("wft_gemm_set_persistent", "wft_attn_set_persistent", "wft_gemm_set_nt_variant", "wft_gemm_set_tn_variant",
                 "wft_attn_set_dkdv_variant", "wft_attn_set_dq_variant", "wft_attn_set_fwd_variant"):
        assert not hasattr(handle, name), name
    for st in (L.GemmArgs, L.AttnArgs):
        assert st.launch_mode.size == 4 and st.variant.size == 4
    assert L.AttnArgs.q_prescaled.offset == L.AttnArgs.variant.offset + 4 == L.AttnArgs.launch_mode.offset + 8
    assert L.AttnArgs.launch_mode.offset == L.AttnArgs.colsum_ws.offset + 8
    assert L.GemmArgs.launch_mode.offset == L.GemmArgs.tn_seg_ptr.offset + 32 and L.GemmArgs.variant.offset == L.GemmArgs.launch_mode.offset + 4
    # the dispatch queries read the per-call fields: the headline fc1 GEMM goes to the 4-wave kernel unless `variant` says otherwise
    a = L.GemmArgs()
    a.M, a.N, a.K, a.batch, a.lda, a.ldb, a.ldc, a.alpha = 130500, 5120, 1280, 1, 1280, 1280, 5120, 1.0
    a.A = a.B = a.C = 1 << 20
    assert handle.wft_gemm_nt_variant(ctypes.byref(a)) == 4
    a.variant = 1
    assert handle.wft_gemm_nt_variant(ctypes.byref(a)) == 256
    b = L.AttnArgs()
    b.B, b.H, b.Tq, b.Tk, b.causal, b.scale = 87, 20, 1500, 1500, 0, 0.125
    b.ldq = b.ldk = b.ldv = 3840
    b.lddo = 1280
    assert [handle.wft_attn_variant(ctypes.byref(b), w) for w in (0, 1, 2)] == [2, 4, 4]
    b.variant = 7
    assert [handle.wft_attn_variant(ctypes.byref(b), w) for w in (0, 1, 2)] == [1, 8, 8]
