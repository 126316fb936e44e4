"""CPU: the C-ABI library loads, exports every symbol include/sy11.h declares, and the binding's table matches the
header.  No compute calls (no GPU here); argument validation paths that return before any launch ARE exercised."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "sy11.h").read_text()


def declared_symbols():
    return sorted(set(re.findall(r"\b(sy11_[a-z0-9_]+)\s*\(", HEADER)))


def test_header_declares_expected_families():
    syms = declared_symbols()
    for fam in ("sy11_conv2d_fwd", "sy11_conv2d_dgrad", "sy11_conv2d_wgrad", "sy11_stem_conv_fwd", "sy11_bn_act_fwd",
                "sy11_maxpool5_fwd", "sy11_attention_fwd", "sy11_detect_decode", "sy11_nms_sorted", "sy11_stft_logmel"):
        assert fam in syms


def test_library_exports_every_declared_symbol():
    from sy11 import _lib
    lib = _lib.load()
    for s in declared_symbols():
        assert hasattr(lib, s), f"libsy11.so does not export {s}"


def test_binding_table_matches_header():
    from sy11 import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.OTHER)
    assert bound == set(declared_symbols())
    # arity check: number of parameters in the header == number of ctypes argtypes
    for name, args in {**_lib.SIGNATURES, **{k: v[0] for k, v in _lib.OTHER.items()}}.items():
        m = re.search(rf"\b{name}\s*\(([^;]*?)\)\s*;", HEADER, re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))


def test_error_codes_and_messages_without_gpu():
    from sy11 import _lib
    lib = _lib.load()
    assert lib.sy11_version() == 100
    d = _lib.ConvDesc(dtype=7)
    rc = lib.sy11_conv2d_fwd(C.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b"dtype" in lib.sy11_last_error()
    d = _lib.ConvDesc(0, 1, 8, 8, 16, 16, 9, 9, 16, 16, 3, 3, 1, 1, 1, 1, 1, 1, 1, 0)      # OH/OW wrong (should be 8)
    rc = lib.sy11_conv2d_fwd(C.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b"OH/OW" in lib.sy11_last_error()
    rc = lib.sy11_stft_logmel(1, 100, 1000, 256, 1, 8, None, None, None, None, 8, None, None, None)
    assert rc == -1
    assert lib.sy11_nms_workspace_bytes(130) == 130 * 3 * 8
    with pytest.raises(_lib.Sy11Error):
        _lib.check(-1, "x")


def test_run_time_options_round_trip_and_reject_unknown_names():
    """Every option include/sy11.h documents can be set and read back without a GPU; an unknown name is an error, not a no-op."""
    from sy11 import _lib
    lib = _lib.load()
    names = re.findall(r'"([a-z0-9_]+)"', HEADER[HEADER.index("run-time options"):HEADER.index("int sy11_set_option")])
    assert {"tune", "tune_log", "igemm_cfg", "wgrad_cfg", "igemm_korder", "igemm_deep", "igemm_bpol", "dgrad_s2_halo", "row_map", "deterministic"} <= set(names)
    for n in set(names):
        old = _lib.get_option(n)
        try:
            _lib.set_option(n, 1)
            assert _lib.get_option(n) == 1, n
        finally:
            _lib.set_option(n, old)
    assert lib.sy11_set_option(b"no_such_option", 1) == -1 and b"unknown option" in lib.sy11_last_error()


def test_product_fails_loudly_on_cpu_tensors():
    import torch
    from sy11 import _lib
    from sy11.nn.modules import Conv
    m = Conv(8, 8, 3)
    with pytest.raises(_lib.Sy11Error):
        m(torch.zeros(1, 8, 4, 4))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from sy11 import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.Sy11Error):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "spectrogram-yolov11_amd"
    for f in pkg.rglob("*.py"):
        t = f.read_text()
        assert "import oracle" not in t and "from oracle" not in t, f


def test_tune_import_accepts_every_configuration_of_this_build_and_nothing_beyond():
    """sy11_tune_import validates a pick against the SAME table sizes the kernels' files use (csrc/tune.h): the highest configuration
    of each table imports, the next index is refused.  (r03: the few-channel conv and patch filter-gradient configurations were
    added to the kernels while the importer still had the old sizes — rank 0's picks then failed to import on the other ranks.)"""
    import struct
    from sy11 import _lib
    import re
    tune_h = (ROOT / "spectrogram-yolov11_amd" / "csrc" / "tune.h").read_text()
    igemm_n = int(re.search(r"SY11_IGEMM_NCFG = (\d+);", tune_h).group(1))
    wgrad_n = int(re.search(r"SY11_WGRAD_NCFG = (\d+);", tune_h).group(1))
    assert igemm_n >= 22 and wgrad_n >= 16                    # r04: the 8-wave pipeline added configurations 20 / 21
    _lib.tune_import(struct.pack("<Qii", 0xfeed0001, 0, igemm_n - 1) + struct.pack("<Qii", 0xfeed0002, 1, wgrad_n - 1))
    blob = _lib.tune_export()
    recs = {blob[i:i + 16] for i in range(0, len(blob), 16)}
    assert struct.pack("<Qii", 0xfeed0001, 0, igemm_n - 1) in recs and struct.pack("<Qii", 0xfeed0002, 1, wgrad_n - 1) in recs
    for kind, bad in ((0, igemm_n), (1, wgrad_n), (0, -1), (2, 0)):
        with pytest.raises(_lib.Sy11Error):
            _lib.tune_import(struct.pack("<Qii", 0xfeed0003, kind, bad))
