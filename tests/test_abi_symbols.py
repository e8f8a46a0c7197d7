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
    assert lib.rscm_gpu_abi_minor() >= 5
    assert lib.rscm_gpu_experiments_build() == 0     # the shipped library reads no experiment knob from the environment
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


# ---- the Rust binding text of INTEGRATION.md ------------------------------------------------------------------------------------
# north_star puts the host in Rust; this image has no rustc, so the `extern "C"` blocks a maintainer would paste into
# crates/rscm-calibrate/src/model_runner.rs:38-85 (`impl ModelRunner`) and crates/rscm-core/src/component.rs:350-437 (`impl Component`)
# are checked the way the ctypes table is: declaration by declaration against the prototypes of include/rscm_gpu.h.

_RUST_BASE = {"i32": "i4", "i64": "i8", "u8": "u1", "u32": "u4", "u64": "u8", "f64": "f8", "f32": "f4", "c_int": "i4", "c_char": "i1",
              "c_void": "void", "RscmEns": "void", "RscmSampler": "void"}


def _rust_type(text):
    """`*mut *mut RscmEns` -> p:p:void, `*const f64` -> p:f8, `c_int` -> i4; anything this table does not know fails the test."""
    words = text.split()
    depth = 0
    while words and words[0] == "*":
        assert len(words) >= 3 and words[1] in ("const", "mut"), text
        depth += 1
        words = words[2:]
    assert len(words) == 1 and words[0] in _RUST_BASE, f"unknown Rust type {text!r}"
    return "p:" * depth + _RUST_BASE[words[0]]


def _rust_externs(path=os.path.join(ROOT, "INTEGRATION.md")):
    """{name: [(ret, [arg types]), ...]} over every `extern "C" { ... }` block inside a ```rust fence of INTEGRATION.md."""
    text = open(path).read()
    out = {}
    for fence in re.findall(r"```rust\n(.*?)```", text, flags=re.S):
        for block in re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', fence, flags=re.S):
            block = re.sub(r"/\*.*?\*/", " ", block, flags=re.S)
            block = re.sub(r"//[^\n]*", " ", block)
            for name, args, ret in re.findall(r"\bfn\s+(\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
                parsed = []
                for a in (x.strip() for x in args.split(",")):
                    if not a:
                        continue
                    assert ":" in a, (name, a)
                    parsed.append(_rust_type(a.split(":", 1)[1].replace("*", " * ")))
                out.setdefault(name, []).append((_rust_type(ret.replace("*", " * ")) if ret else "void", parsed))
            # nothing in a block may be left undeclared-looking: an elided argument list ("...") would not compile either
            assert "..." not in block, "an extern block elides arguments"
    return out


def test_rust_binding_text_agrees_with_the_header_by_type():
    protos = _prototypes(("rscm_gpu.h",))          # a Rust host binds the public boundary only
    rust = _rust_externs()
    assert len(rust) >= 25, sorted(rust)
    for must in ("rscm_ens_create", "rscm_ens_set_params_aos", "rscm_ens_set_forcing", "rscm_ens_set_initial", "rscm_ens_run",
                 "rscm_ens_get_series", "rscm_ens_status", "rscm_ens_loglik", "rscm_ens_link_input", "rscm_ens_run_lockstep",
                 "rscm_sampler_create", "rscm_sampler_create_sharded", "rscm_sampler_create_graph", "rscm_ens_create_windowed"):
        assert must in rust, must
    for name, decls in rust.items():
        assert name in protos, f"INTEGRATION.md declares {name}, which include/rscm_gpu.h does not export"
        want_ret, want_args = protos[name]
        for ret, args in decls:               # a function may be declared in more than one section: every copy must be right
            assert ret == want_ret, (name, "return", ret, want_ret)
            assert len(args) == len(want_args), (name, "arity", len(args), len(want_args))
            assert args == want_args, (name, args, want_args)


def test_the_rust_parser_refuses_what_it_does_not_know(tmp_path):
    p = tmp_path / "x.md"
    p.write_text('```rust\nextern "C" {\n    fn rscm_ens_run(h: *mut RscmEns, step_begin: i32, step_end: usize) -> c_int;\n}\n```\n')
    import pytest
    with pytest.raises(AssertionError, match="unknown Rust type"):
        _rust_externs(str(p))
    p.write_text('```rust\nextern "C" {\n    fn rscm_ens_run(h: *mut RscmEns, step_begin: i64, step_end: i32) -> c_int;\n}\n```\n')
    got = _rust_externs(str(p))["rscm_ens_run"][0]
    assert got == ("i4", ["p:void", "i8", "i4"]) and got[1] != _prototypes(("rscm_gpu.h",))["rscm_ens_run"][1]


def test_the_shipped_library_reads_only_the_documented_environment():
    """a Rust host linking librscm_gpu.so must not inherit undocumented environment-dependent launch plans.  The
    only getenv calls in csrc/ are RSCM_SPLIT_RUNS and RSCM_POISON_ALLOC -- both documented in include/rscm_gpu.h -- and the one
    inside `#ifdef RSCM_EXPERIMENTS` of experiment_env.hpp (compiled out of the shipped build: rscm_gpu_experiments_build() == 0)."""
    csrc = os.path.join(ROOT, "rscm_amd", "csrc")
    found = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".hpp")):
            for m in re.finditer(r"getenv\(([^)]*)\)", open(os.path.join(csrc, f)).read()):
                found.setdefault(f, []).append(m.group(1))
    assert found == {"ens.hpp": ['"RSCM_POISON_ALLOC"'], "experiment_env.hpp": ["name"], "rscm_gpu.cpp": ['"RSCM_SPLIT_RUNS"']}, found
    guard = open(os.path.join(csrc, "experiment_env.hpp")).read()
    assert guard.index("#ifdef RSCM_EXPERIMENTS") < guard.index("getenv(name)") < guard.index("#else")
    header = open(os.path.join(ROOT, "include", "rscm_gpu.h")).read()
    assert "RSCM_SPLIT_RUNS=0" in header and "RSCM_POISON_ALLOC=1" in header
