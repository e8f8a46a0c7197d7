"""CPU tier: the C-ABI library loads and exports every symbol include/rscm_gpu.h declares, and
the ctypes table binds exactly that set.  No compute call is made (no GPU here)."""
import os
import re

from rscm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(headers=("rscm_gpu.h", "rscm_gpu_internal.h")):
    """The boundary (rscm_gpu.h) and the test / A-B hooks the library also exports (rscm_gpu_internal.h)."""
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"RSCM_API\s+[\w\s\*]+?\b(rscm_\w+)\s*\(", text))
    return names


def test_test_hooks_are_not_in_the_public_header():
    public = _declared(("rscm_gpu.h",))
    hooks = _declared(("rscm_gpu_internal.h",))
    assert hooks and not (public & hooks)
    for name in ("rscm_gpu_set_lockstep_fusion", "rscm_gpu_lockstep_stats", "rscm_gpu_selftest_div", "rscm_gpu_ocean_fit_selftest"):
        assert name in hooks


def test_header_and_binding_table_agree():
    decl = _declared()
    assert len(decl) >= 30
    assert decl == set(_lib.SIGNATURES)


_C_BASE = {"int": "i4", "int32_t": "i4", "uint32_t": "u4", "int64_t": "i8", "uint64_t": "u8", "double": "f8", "float": "f4",
           "uint8_t": "u1", "char": "i1", "void": "void", "rscm_ens": "void", "rscm_sampler": "void"}   # opaque handles travel as void*


def _c_type(text):
    """Canonical form of one C parameter or return type as the header writes it: base type by kind and width, one "p:" per level of
    indirection (`double out[4]` is a pointer; `const` does not change the ABI); the parameter's name, if any, is dropped."""
    text = text.replace("const", " ").strip()
    depth = text.count("*") + text.count("[")
    words = re.findall(r"[A-Za-z_]\w*", re.sub(r"\[.*?\]", " ", text))
    assert words and words[0] in _C_BASE, text
    return "p:" * depth + _C_BASE[words[0]]


def _prototypes(headers=("rscm_gpu.h", "rscm_gpu_internal.h")):
    out = {}
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for ret, name, args in re.findall(r"RSCM_API\s+([\w\s\*]+?)\b(rscm_\w+)\s*\(([^)]*)\)\s*;", text):
            args = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
            out[name] = (_c_type(ret), [_c_type(a) for a in args])
    return out


def _ctypes_type(t):
    import ctypes as C
    if t is None:
        return "void"
    if t is C.c_void_p:
        return "p:void"
    if t is C.c_char_p:
        return "p:i1"
    if hasattr(t, "_type_") and not isinstance(t._type_, str):   # POINTER(X)
        return "p:" + _ctypes_type(t._type_)
    kind = {"i": "i", "l": "i", "q": "i", "I": "u", "L": "u", "Q": "u", "B": "u", "b": "i", "d": "f", "f": "f", "P": "p:void", "z": "p:i1"}[t._type_]
    return kind if kind.startswith("p:") else f"{kind}{C.sizeof(t)}"


def test_binding_table_agrees_with_the_header_by_type():
    """Every entry of the ctypes table against the C prototype in the headers, argument by argument: kind (signed / unsigned / float
    / pointer), width and level of indirection -- the compiler checks the header against the implementation, this checks the
    Python table against the header (an int64 passed as int32, a missing argument or a double taken for a pointer would corrupt a
    call silently)."""
    protos = _prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        want_ret, want_args = protos[name]
        assert _ctypes_type(restype) == want_ret, (name, "return", _ctypes_type(restype), want_ret)
        got = [_ctypes_type(a) for a in argtypes]
        assert got == want_args, (name, got, want_args)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()  # binds each symbol; AttributeError if one is missing
    for name in _declared():
        assert hasattr(lib, name)
    assert lib.rscm_gpu_abi_version() == 1
    assert lib.rscm_gpu_abi_minor() >= 4
    assert lib.rscm_gpu_last_error() is not None


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under rscm_amd/ or include/ may import, link
    or load it."""
    for top in ("rscm_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            if "build" in dirpath.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                    src = open(os.path.join(dirpath, f)).read()
                    for needle in ("import oracle", "from oracle", "librscm_oracle", "rscm_oracle",
                                   "oracle/"):
                        assert needle not in src, f"{top}/{f} references {needle!r}"


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without the HIP library every product entry point raises."""
    import pytest
    from rscm_amd import RscmGpuUnavailable
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "librscm_gpu.so"))
    with pytest.raises(RscmGpuUnavailable, match="no CPU fallback"):
        _lib.load()
    import numpy as np
    import rscm_amd
    with pytest.raises(RscmGpuUnavailable):
        rscm_amd.Ensemble(rscm_amd.KIND_TWO_LAYER, 4, np.arange(1750.0, 1760.0))
    (tmp_path / "librscm_gpu.so").write_bytes(b"not an ELF file")
    with pytest.raises(RscmGpuUnavailable, match="cannot load"):
        _lib.load()
