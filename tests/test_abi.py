"""The C-ABI library loads (no GPU needed) and exports every symbol include/emage_hip.h declares, with the
prototypes pantomatrix_amd/_lib.py binds."""
import os
import re

from pantomatrix_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(tools=False):
    text = open(os.path.join(ROOT, "include", "emage_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    guarded = re.findall(r"#ifdef EMAGE_TOOLS(.*?)#endif", text, flags=re.S)
    text = "".join(guarded) if tools else re.sub(r"#ifdef EMAGE_TOOLS.*?#endif", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:int|long|size_t|const char\*)\s+(emage_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        protos[m.group(1)] = args
    return protos


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    protos = _declared()
    assert len(protos) >= 16
    for name, args in protos.items():
        assert hasattr(lib, name), f"{name} declared in emage_hip.h but not exported"
        if name in ("emage_abi_version", "emage_target_arch"):
            continue
        assert name in _lib.SIGNATURES, f"{name} not bound in _lib.py"
        assert len(_lib.SIGNATURES[name]) == len(args), (name, len(_lib.SIGNATURES[name]), len(args))
    assert set(_lib.SIGNATURES) <= set(protos)
    assert lib.emage_abi_version() == _lib.ABI_VERSION and lib.emage_target_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    """Invalid arguments are rejected before any launch (EMAGE_EINVAL = -1), so this runs without a device."""
    lib = _lib.load()
    assert lib.emage_vq_argmin_f32(None, 0, None, None, 0, 0, 0, 0, 0, None) == -1
    assert lib.emage_gemm(1, None, 0, None, None, None, None, 0, 0, 0, None, 0, 0, None, 0, None, 0, 0, 0,
                          0, 0, 0, 0, 0, 0, 0, 0, 1.0, 1.0, None) == -1
    assert lib.emage_gather_rows(None, None, 0, 0, 1, None, 0, 0, 0, 0, 0, 0, None) == -1
    assert lib.emage_wav_conv_in(0, None, 0, 0, 1, 0, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, None) == -1
    assert lib.emage_pack_motion(0, None, None, 0, None, None, 0, 0, None, 0, 0, 0, 0, 0, None) == -1


def test_product_library_has_no_tuning_hooks_and_the_tools_twin_does():
    """the product .so carries no `emage_set_tuning` / tracer / mutable tuning globals; the -DEMAGE_TOOLS twin
    exports them (and everything the product exports)."""
    import ctypes
    tools_only = _declared(tools=True)
    assert set(tools_only) == set(_lib.TOOLS_SIGNATURES) == {"emage_set_tuning", "emage_h2_set_trace"}
    product = ctypes.CDLL(_lib.LIB_PATH)
    for name in tools_only:
        assert not hasattr(product, name), f"{name} must not be exported by the product library"
    try:
        tools = _lib.use_tools(True)
        for name in list(_declared()) + list(tools_only):
            assert hasattr(tools, name), name
        assert tools.emage_set_tuning(99, 0) == -1 and tools.emage_set_tuning(0, -1) == 0
    finally:
        _lib.use_tools(False)
