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


def test_library_exports_every_declared_symbol():
    lib = _lib.load()  # binds each symbol; AttributeError if one is missing
    for name in _declared():
        assert hasattr(lib, name)
    assert lib.rscm_gpu_abi_version() == 1
    assert lib.rscm_gpu_abi_minor() >= 3
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
