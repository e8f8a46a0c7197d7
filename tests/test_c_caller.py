"""A compiled, non-Python caller of the boundary: tests/c_abi/caller.c (strict C11, includes only include/rscm_gpu.h, links
rscm_amd/librscm_gpu.so) does ModelRunner::run_batch's job for the two-layer model
(crates/rscm-calibrate/src/model_runner.rs:161-266) the way a Rust / cgo / JNI host would: through the C prototypes, not through
ctypes.  CPU tier: it compiles and links with -Wall -Wextra -Werror -pedantic and fails loudly without a GPU.  GPU tier: its output
equals the oracle's bit for bit on every member and kept row."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOW = np.array([0.8, 0.0, 1.0, 0.5, 5.0, 50.0])
HIGH = np.array([1.5, 0.1, 1.8, 1.0, 15.0, 200.0])


def _build(tmp_path):
    exe = str(tmp_path / "caller")
    libdir = os.path.join(ROOT, "rscm_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "caller.c"), "-o", exe, "-L", libdir, "-lrscm_gpu", f"-Wl,-rpath,{libdir}",
           "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _gpus():
    import torch
    return torch.cuda.device_count()


def test_c_caller_builds_against_the_header_alone(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 64 and "usage" in r.stderr
    if _gpus() == 0:   # no CPU fallback behind the C-ABI either: the first device call reports, the caller exits non-zero
        r = subprocess.run([exe, str(tmp_path / "out.bin"), "8"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "rscm_gpu_device_count failed" in r.stderr and not (tmp_path / "out.bin").exists()


def caller_inputs(n):
    """The caller's inputs re-formed with the same integer and IEEE operations (no libm on either side)."""
    i = np.arange(n, dtype=np.uint64)[:, None]
    j = np.arange(6, dtype=np.uint64)[None, :]
    hashed = (i * np.uint64(2654435761) + j * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    u = hashed.astype(np.float64) / 4294967296.0
    params = LOW[None, :] + (HIGH - LOW)[None, :] * u          # [N][P]
    t = np.arange(1750, 2501, dtype=np.float64)
    x = (t - 1750.0) / 120.0
    return params, 4.0 * x / (1.0 + x), np.append(t, 2501.0)


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("n", [1000, 70_000])   # one launch; a run cut into two member blocks on two streams
def test_c_caller_matches_the_oracle_bit_for_bit(tmp_path, n):
    from oracle import cbind
    exe = _build(tmp_path)
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(out), str(n)], capture_output=True, text=True, timeout=600)   # a child process, never an exec
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    raw = out.read_bytes()
    n_file, kept = np.frombuffer(raw, dtype=np.int64, count=2)
    rows = np.frombuffer(raw, dtype=np.int32, count=int(kept), offset=16)
    off = 16 + 4 * int(kept)
    ts = np.frombuffer(raw, dtype=np.float64, count=int(kept) * n, offset=off).reshape(int(kept), n)
    td = np.frombuffer(raw, dtype=np.float64, count=int(kept) * n, offset=off + 8 * int(kept) * n).reshape(int(kept), n)
    status = np.frombuffer(raw, dtype=np.uint8, count=n, offset=off + 16 * int(kept) * n)
    assert n_file == n and list(rows) == list(range(0, 751, 50)) and len(raw) == off + 16 * int(kept) * n + n
    params, forcing, bounds = caller_inputs(n)
    want_ts, want_td = cbind.two_layer_run(bounds, np.ascontiguousarray(params.T), forcing, 0.0, 0.0, threads=8)
    assert np.array_equal(ts.view(np.uint64), want_ts[rows].view(np.uint64))
    assert np.array_equal(td.view(np.uint64), want_td[rows].view(np.uint64))
    bad = ~(np.isfinite(want_ts[-1]) & np.isfinite(want_td[-1]))
    assert np.array_equal(status != 0, bad)
    assert line["members"] == n and line["time_index"] == 750 and line["failed_members"] == int(bad.sum())
    assert (line["member_blocks"], line["step_chunks"]) == ((2, 12) if n > 65536 else (1, 1))   # a cut run above 65 536 members
    fin = np.isfinite(want_ts[270])
    assert line["ts_2020_count"] == fin.sum() and abs(line["ts_2020_mean"] - want_ts[270][fin].mean()) < 1e-9


# ---------------------------------------------------------------------------------------------------------------------------
# One process, one thread + one handle per device: tests/c_abi/two_devices.c


def _build_two_devices(tmp_path):
    exe = str(tmp_path / "two_devices")
    libdir = os.path.join(ROOT, "rscm_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "two_devices.c"), "-o", exe, "-L", libdir, "-lrscm_gpu",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lpthread", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_threaded_multi_device_caller_builds(tmp_path):
    exe = _build_two_devices(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 64 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("n_total,threads", [(100_001, 2), (70_000, 3)])
def test_one_process_one_thread_per_handle_equals_one_handle(tmp_path, n_total, threads):
    """The host the north-star names is one (Rust) process driving the GPUs of a node: one thread and one handle per device, member
    blocks of one global Latin hypercube drawn where they run, the per-member losses gathered from the library's device pointers --
    ncclAllGather when every thread has a device of its own, and on a one-GPU box (RCCL takes one rank per device) host copies plus
    a one-rank ncclAllGather of the same pointer.  Whatever the number of threads, the gathered vector is the one a single handle of
    all the members computes, bit for bit; handles driven from different threads of one process do not disturb each other."""
    import rscm_amd
    exe = _build_two_devices(tmp_path)
    out = tmp_path / "out.bin"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(out), str(n_total), str(threads)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["members"] == n_total and line["threads"] == threads and len(line["run_ms"]) == threads and min(line["run_ms"]) > 0
    if line["devices"] < threads:
        assert line["gather"] == "rscm_gpu_copy_to_host" and line["one_rank_rccl_ok"] is True
    else:
        assert line["gather"] == "ncclAllGather"
    raw = out.read_bytes()
    got = np.frombuffer(raw, dtype=np.float64, count=n_total)
    status = np.frombuffer(raw, dtype=np.uint8, count=n_total, offset=8 * n_total)
    _, forcing, bounds = caller_inputs(1)
    with rscm_amd.Ensemble(rscm_amd.KIND_TWO_LAYER, n_total, bounds) as e:
        e.sample_lhs(20260327, LOW, HIGH)
        e.set_forcing(forcing)
        e.set_initial("Surface Temperature", 0.0)
        e.set_initial("Deep Ocean Temperature", 0.0)
        e.run()
        tidx = 100 + 10 * np.arange(18)
        want = e.loglik(["Surface Temperature"] * 18, tidx, 1.0 + 0.004 * tidx, np.full(18, 0.1))
        assert np.array_equal(status, e.status())
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.isfinite(got).sum() > 0.9 * n_total


@pytest.mark.parametrize("header", ["rscm_gpu.h", "rscm_gpu_internal.h"])
def test_headers_compile_alone_as_strict_c11_and_as_cxx(tmp_path, header):
    """The boundary's headers are self-contained C (what a bindgen / cgo / JNI generator reads) and valid C++ (extern "C" guards)."""
    src = tmp_path / "use.c"
    src.write_text(f'#include "{header}"\nint main(void) {{ return RSCM_GPU_ABI_VERSION == 1 ? 0 : 1; }}\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                ["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)]):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------------------
# The component-graph API from C: tests/c_abi/caller_graph.c


def _build_graph_caller(tmp_path):
    exe = str(tmp_path / "caller_graph")
    libdir = os.path.join(ROOT, "rscm_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "caller_graph.c"), "-o", exe, "-L", libdir, "-lrscm_gpu", f"-Wl,-rpath,{libdir}",
           "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_graph_caller_builds_against_the_header_alone(tmp_path):
    exe = _build_graph_caller(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 64 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(_gpus() < 1, reason="needs a GPU")
def test_c_graph_caller_links_steps_and_scores(tmp_path):
    """The reference's coupled model (docs/notebooks/coupled_model.py: CarbonCycle -> CO2ERF -> Sum -> TwoLayer with the lagged
    temperature feedback) assembled FROM C as four linked ensembles with each consumer's VariableSource, stepped in lock-step on the
    caller's stream, equals the fused kind bit for bit on every series and in the per-member log-likelihood (checked inside the C
    program), refuses to destroy a source that is still linked, and matches the oracle within the coupled kind's 1e-11."""
    from oracle import cbind
    n = 3000
    exe = _build_graph_caller(tmp_path)
    out = tmp_path / "graph.bin"
    r = subprocess.run([exe, str(out), str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"members": n, "steps": 200, "linked_equals_fused": True, "destroy_of_a_linked_source_refused": True}
    raw = out.read_bytes()
    n_file, T = (int(x) for x in np.frombuffer(raw, dtype=np.int64, count=2))
    assert (n_file, T) == (n, 201)
    ts = np.frombuffer(raw, dtype=np.float64, count=T * n, offset=16).reshape(T, n)
    co2 = np.frombuffer(raw, dtype=np.float64, count=T * n, offset=16 + 8 * T * n).reshape(T, n)
    ll = np.frombuffer(raw, dtype=np.float64, count=n, offset=16 + 16 * T * n)
    lo = np.array([0.9, 0.0, 1.0, 0.5, 5.0, 50.0, 15.0, 278.0, 0.0, 3.7])
    hi = np.array([1.5, 0.05, 1.8, 1.0, 15.0, 200.0, 40.0, 278.0, 0.1, 3.7])
    i = np.arange(n, dtype=np.uint64)[None, :]
    j = np.arange(10, dtype=np.uint64)[:, None]
    hashed = (i * np.uint64(2246822519) + j * np.uint64(374761393) + np.uint64(7)) & np.uint64(0xFFFFFFFF)
    P = lo[:, None] + (hi - lo)[:, None] * (hashed.astype(np.float64) / 4294967296.0)
    t = np.arange(1750.0, 1951.0)
    want = cbind.coupled_run(np.append(t, 1951.0), np.ascontiguousarray(P), 0.02 * np.arange(201.0),
                             dict(ts=0.0, td=0.0, conc=278.0, cum_uptake=0.0, cum_emis=0.0), threads=8)
    close = lambda a, b: np.abs(a - b) <= 1e-11 * np.maximum(1.0, np.abs(b))   # noqa: E731
    assert close(ts, want["ts"]).all() and close(co2, want["conc"]).all()
    tidx = 30 * np.arange(1, 7)
    expect = (-0.5 * ((0.004 * tidx[:, None] - want["ts"][tidx]) ** 2 / 0.2 ** 2)).sum(axis=0)
    assert np.allclose(ll, expect, rtol=1e-9, atol=1e-9) and np.isfinite(ll).all()
