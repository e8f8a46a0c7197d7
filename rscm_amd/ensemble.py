"""``Ensemble``: one device-resident batch of model instances behind the C-ABI.

This is the host-side object the reference-shaped front-ends (``rscm_amd.core.Model``,
``rscm_amd.calibrate.ModelRunner``) drive.  It owns one ``rscm_ens`` handle = one GPU; all
arithmetic happens in the HIP kernels.  Semantics follow the reference's stepper
(crates/rscm-core/src/model/runtime.rs:504-527): ``step()`` solves the current step and writes
index ``time_index + 1``; ``run()`` steps to the end of the axis.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib as L


class DeviceVector:
    """A per-member vector left in device memory by an ensemble (``loglik(..., on_device=True)``,
    ``status_device()``): owned by the handle and valid until its next call of the same kind.  It
    carries ``__cuda_array_interface__``, so ``torch.as_tensor(v, device="cuda")`` is a zero-copy
    view -- what the RCCL all-gather of ``rscm_amd.distributed`` sends -- and ``to_host()`` copies it
    out through the library."""

    def __init__(self, ptr: int, n: int, dtype, owner: "Ensemble"):
        self.ptr, self.n, self.dtype, self.owner = int(ptr), int(n), np.dtype(dtype), owner

    def __len__(self) -> int:
        return self.n

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.n,), "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2,
                "strides": None}

    def to_host(self) -> np.ndarray:
        out = np.empty(self.n, dtype=self.dtype)
        L.check(L.load().rscm_gpu_copy_to_host(self.owner.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                               out.nbytes))
        return out


class Ensemble:
    def __init__(self, kind: int, n_members: int, time_bounds: Sequence[float], device: int = 0,
                 store_series: bool = True, window_rows: Optional[int] = None, output_stride: int = 0,
                 output_vars: Optional[Sequence] = None):
        """``window_rows``: keep only a sliding window of that many rows of every series
        (``RSCM_FLAG_WINDOWED``) plus, with ``output_stride`` > 0, every ``output_stride``-th row of
        ``output_vars`` (names or ids; None: all) -- for long axes stepped in lock-step."""
        self._lib = L.load()
        b = L.f64(time_bounds)
        if b.ndim != 1 or len(b) < 3:
            raise ValueError("time_bounds needs at least 3 entries (2 time points)")
        self.kind = kind
        self.n_members = int(n_members)
        self.n_times = len(b) - 1
        self.bounds = b
        self.device = device
        var_ids, self.n_params, input_rows = L.KIND_TABLE[kind]
        self.var_ids: Dict[str, int] = dict(var_ids)
        self.input_rows = input_rows  # names of the rows of the input block, or None
        self.n_inputs = len(input_rows) if input_rows else 1
        h = C.c_void_p()
        self.store_series = bool(store_series)
        self.window_rows = None if window_rows is None or window_rows >= self.n_times else int(window_rows)
        self.output_stride = int(output_stride) if self.window_rows else 0
        if self.window_rows:
            ov = None
            if output_vars is not None:
                ov = np.ascontiguousarray([self._var(v) for v in output_vars], dtype=np.int32)
            self.output_vars = None if ov is None else [int(v) for v in ov]
            L.check(self._lib.rscm_ens_create_windowed(kind, self.n_members, self.n_times, L.dptr(b), device, L.FLAG_WINDOWED,
                                                       self.window_rows, self.output_stride, -1 if ov is None else len(ov),
                                                       L.iptr(ov), C.byref(h)))
        else:
            L.check(self._lib.rscm_ens_create_ex(kind, self.n_members, self.n_times, L.dptr(b), device,
                                                 0 if store_series else L.FLAG_NO_SERIES, C.byref(h)))
        self._h = h
        # the library and the tables of this package must agree on the shape of the kind
        got = [C.c_int32() for _ in range(3)]
        for fn, x in zip((self._lib.rscm_ens_n_params, self._lib.rscm_ens_n_vars, self._lib.rscm_ens_n_inputs), got):
            L.check(fn(self._h, C.byref(x)))
        want = (self.n_params, max(self.var_ids.values()) + 1, self.n_inputs)
        if tuple(x.value for x in got) != want:
            self.close()
            raise RuntimeError(f"kind {kind}: library reports (params, vars, inputs) = {tuple(x.value for x in got)}, "
                               f"package tables say {want}")

    # -- lifecycle --------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            L.check(self._lib.rscm_ens_destroy(self._h))  # fails while other ensembles link to this one
            self._h = None
            self._linked = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _var(self, var) -> int:
        return self.var_ids[var] if isinstance(var, str) else int(var)

    # -- configuration ----------------------------------------------------------------------
    def set_mode(self, mode: int) -> None:
        L.check(self._lib.rscm_ens_set_mode(self._h, mode))

    def set_step_size(self, component: int, step: float) -> None:
        L.check(self._lib.rscm_ens_set_step_size(self._h, component, float(step)))

    def set_params(self, soa) -> None:
        p = L.f64(soa)
        if p.shape != (self.n_params, self.n_members):
            raise ValueError(f"params must be [{self.n_params}][{self.n_members}], got {p.shape}")
        L.check(self._lib.rscm_ens_set_params(self._h, L.dptr(p)))

    def set_params_aos(self, aos) -> None:
        p = L.f64(aos)
        if p.shape != (self.n_members, self.n_params):
            # mirrors model_runner.rs:225-231 "Expected {} parameters, got {}"
            raise ValueError(f"Expected {self.n_params} parameters per member "
                             f"([{self.n_members}][{self.n_params}]), got {p.shape}")
        L.check(self._lib.rscm_ens_set_params_aos(self._h, L.dptr(p)))

    def set_forcing(self, series, scenario_of_member=None, source: int = L.SRC_EXOGENOUS,
                    var=0) -> None:
        s = L.f64(series)
        if self.input_rows:  # a block of rows per scenario: [S][n_inputs][T] or [n_inputs][T]
            if s.ndim == 1 and self.n_inputs == 1:
                s = s[None, None]
            elif s.ndim == 2:
                # one input row: [S][T]; several: one scenario's [n_inputs][T]
                s = s[:, None] if self.n_inputs == 1 else s[None]
            if s.ndim != 3 or s.shape[1:] != (self.n_inputs, self.n_times):
                raise ValueError(f"input block must be [S][{self.n_inputs}][{self.n_times}], got {s.shape}")
            s = np.ascontiguousarray(s)
        else:
            s = np.atleast_2d(s)
            if s.shape[1] != self.n_times:
                raise ValueError(f"forcing must have {self.n_times} time points, got {s.shape[1]}")
        sc = None
        if scenario_of_member is not None:
            sc = np.ascontiguousarray(scenario_of_member, dtype=np.int32)
            if sc.shape != (self.n_members,):
                raise ValueError("scenario_of_member must have one entry per member")
        L.check(self._lib.rscm_ens_set_forcing(self._h, self._var(var), s.shape[0], L.dptr(s),
                                               L.iptr(sc), source))

    def link_input(self, input_row, src: "Ensemble", src_var, source: int = L.SRC_EXOGENOUS) -> None:
        """Read input row ``input_row`` (index or name) member by member from ``src``'s stored
        series ``src_var`` instead of the scenario table: an edge of a component graph kept on the
        device.  ``source`` is this consumer's VariableSource (``SRC_EXOGENOUS``: index n,
        ``SRC_UPSTREAM``: n+1).  Both ensembles must share a stream; ``src`` must outlive the link."""
        row = (self.input_rows.index(input_row) if self.input_rows else 0) if isinstance(input_row, str) else int(input_row)
        L.check(self._lib.rscm_ens_link_input(self._h, row, src._h, src._var(src_var), source))
        self._linked = getattr(self, "_linked", {})
        self._linked[row] = src  # keeps the producer alive as long as this consumer

    def unlink_input(self, input_row) -> None:
        row = (self.input_rows.index(input_row) if self.input_rows else 0) if isinstance(input_row, str) else int(input_row)
        L.check(self._lib.rscm_ens_unlink_input(self._h, row))
        getattr(self, "_linked", {}).pop(row, None)

    def set_initial(self, var, values) -> None:
        v = np.atleast_1d(L.f64(values))
        L.check(self._lib.rscm_ens_set_initial(self._h, self._var(var), L.dptr(v), v.size))

    def set_state(self, var, time_index: int, values) -> None:
        """Overwrite row ``time_index`` of a stored series (1 value = broadcast, or one per member)."""
        v = np.atleast_1d(L.f64(values))
        L.check(self._lib.rscm_ens_set_state(self._h, self._var(var), int(time_index), L.dptr(v), v.size))

    def set_time_index(self, time_index: int) -> None:
        L.check(self._lib.rscm_ens_set_time_index(self._h, int(time_index)))

    def set_stream(self, hip_stream: Optional[int]) -> None:
        L.check(self._lib.rscm_ens_set_stream(self._h, C.c_void_p(hip_stream)))

    def sample_lhs(self, seed: int, low, high, member_offset: int = 0,
                   n_total: Optional[int] = None) -> None:
        lo, hi = L.f64(low), L.f64(high)
        if lo.shape != (self.n_params,) or hi.shape != (self.n_params,):
            raise ValueError("low/high need one entry per parameter")
        n_total = self.n_members if n_total is None else n_total
        L.check(self._lib.rscm_ens_sample_lhs(self._h, C.c_uint64(seed), L.dptr(lo), L.dptr(hi),
                                              member_offset, n_total))

    def get_params(self) -> np.ndarray:
        out = np.empty((self.n_params, self.n_members))
        L.check(self._lib.rscm_ens_get_params(self._h, L.dptr(out)))
        return out

    # -- stepping ---------------------------------------------------------------------------
    @property
    def time_index(self) -> int:
        n = C.c_int32()
        L.check(self._lib.rscm_ens_time_index(self._h, C.byref(n)))
        return n.value

    def step(self) -> None:
        n = self.time_index
        L.check(self._lib.rscm_ens_run(self._h, n, n + 1))

    def run(self, step_end: Optional[int] = None, *, sync: bool = True) -> None:
        end = self.n_times - 1 if step_end is None else step_end
        fn = self._lib.rscm_ens_run if sync else self._lib.rscm_ens_run_async
        L.check(fn(self._h, self.time_index, end))

    def sync(self) -> None:
        L.check(self._lib.rscm_ens_sync(self._h))

    def rewind(self) -> None:
        L.check(self._lib.rscm_ens_rewind(self._h))

    def clear_series(self) -> None:
        """Rewind and make every stored row after index 0 NaN again (a fresh collection)."""
        L.check(self._lib.rscm_ens_clear_series(self._h))

    def finished(self) -> bool:
        return self.time_index == self.n_times - 1

    def last_run_ms(self) -> float:
        ms = C.c_float()
        L.check(self._lib.rscm_ens_last_run_ms(self._h, C.byref(ms)))
        return ms.value

    def last_run_plan(self):
        """(member blocks, step chunks) the most recent run was cut into: (2, k) when a whole-axis run of the two-layer or the
        coupled kind -- or an unlinked ClimateUDEB run over more than 65 536 members (two halves, each chunk reloading and storing
        the block's columns) -- was issued as two member blocks on two streams in k chunks of model steps
        (rscm_ens_last_run_plan), else (1, 1)."""
        mb, sc = C.c_int32(), C.c_int32()
        L.check(self._lib.rscm_ens_last_run_plan(self._h, C.byref(mb), C.byref(sc)))
        return mb.value, sc.value

    # -- checkpoint / resume ------------------------------------------------------------------
    def state_vars(self) -> Dict[str, int]:
        """The State variables of the kind (what the stepper reads back at the next step)."""
        ids = {L.KIND_TWO_LAYER: (1, 2), L.KIND_COUPLED: (1, 5), L.KIND_UDEB: (1, 4), L.KIND_CH4_CHEMISTRY: (1, 1),
               L.KIND_N2O_CHEMISTRY: (1, 1), L.KIND_CO2_BUDGET: (1, 1), L.KIND_TERRESTRIAL_CARBON: (1, 4),
               L.KIND_OCEAN_CARBON: (1, 2), L.KIND_HALOCARBON: (1, len(L.HC_SPECIES)), L.KIND_CARBON_CYCLE: (1, 3)}
        lo, hi = ids.get(self.kind, (1, 0))  # the other kinds are stateless
        return {k: v for k, v in self.var_ids.items() if lo <= v <= hi}

    def _history_depth(self) -> int:
        """Rows before the current one that the next step reads (previous() / at_offset(-k))."""
        if self.kind == L.KIND_CH4_CHEMISTRY:
            return 1
        if self.kind == L.KIND_N2O_CHEMISTRY:  # at_offset(-(strat_delay + 1)), n2o.rs:203-218
            return int(max(1.0, np.nanmax(self.get_params()[L.N2O_PARAM_NAMES.index("strat_delay")]))) + 1
        return 0

    def checkpoint(self, all_variables: bool = False) -> Dict[str, object]:
        """What is needed to resume: time index, parameters, row ``time_index`` of every state
        variable (``all_variables``: of every stored variable -- what linked consumers read), the
        earlier rows the chemistry kinds look back at, and the internal component state of
        ClimateUDEB / OceanCarbon.  The reference's checkpoint holds time_index, the whole collection
        and the component states (crates/rscm-core/src/model/runtime.rs:270-282)."""
        k = self.time_index
        names = {n: v for n, v in self.var_ids.items() if v > 0} if all_variables else self.state_vars()
        depth = min(self._history_depth(), k)
        history = {name: self.get_series(v, k - depth, k) for name, v in self.state_vars().items()} if depth else {}
        n = C.c_int64()
        L.check(self._lib.rscm_ens_internal_state_size(self._h, C.byref(n)))
        internal = None
        if n.value:
            internal = np.empty(n.value)
            L.check(self._lib.rscm_ens_get_internal_state(self._h, L.dptr(internal)))
        return {"kind": self.kind, "n_members": self.n_members, "bounds": self.bounds.copy(),
                "time_index": k, "params": self.get_params(),
                "state": {name: self.get_series(v, k, k + 1)[0] for name, v in names.items()},
                "history": history, "internal": internal}

    def restore(self, ck: Dict[str, object], clear_later_rows: bool = False) -> None:
        """``clear_later_rows``: make every stored row after the checkpoint's time index NaN again, as
        in the collection the checkpoint was taken from -- needed when an already advanced ensemble is
        rolled back and a linked consumer reads rows its producer has not rewritten yet."""
        if (ck["kind"] != self.kind or ck["n_members"] != self.n_members
                or not np.array_equal(ck["bounds"], self.bounds)):
            raise ValueError("checkpoint does not match this ensemble (kind, members or time axis)")
        self.set_params(ck["params"])
        k = int(ck["time_index"])
        # the stepper first: a windowed ensemble positions its window at k, then the rows go in
        internal = ck.get("internal")
        if internal is not None:
            blob = L.f64(internal)
            L.check(self._lib.rscm_ens_set_internal_state(self._h, L.dptr(blob), blob.size, k))
        else:
            L.check(self._lib.rscm_ens_set_time_index(self._h, k))
        for name, row in ck["state"].items():
            self.set_state(name, k, row)
        for name, rows in ck.get("history", {}).items():
            for d, row in enumerate(rows):
                self.set_state(name, k - len(rows) + d, row)
        if clear_later_rows and self.store_series and not self.window_rows:
            L.check(self._lib.rscm_ens_clear_rows_after(self._h, k))

    # -- outputs ----------------------------------------------------------------------------
    def get_series(self, var, t_begin: int = 0, t_end: Optional[int] = None, t_stride: int = 1,
                   m_begin: int = 0, m_end: Optional[int] = None,
                   out: Optional[np.ndarray] = None) -> np.ndarray:
        """``[n_t][m_end - m_begin]`` copy of a stored series.  Pass ``out`` (e.g. from
        ``pinned_empty``) to reuse a buffer; a page-locked one is filled by DMA at PCIe rate."""
        t_end = self.n_times if t_end is None else t_end
        m_end = self.n_members if m_end is None else m_end
        nt = len(range(t_begin, t_end, t_stride))
        if out is None:
            out = np.empty((nt, m_end - m_begin))
        elif out.shape != (nt, m_end - m_begin) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float64 array of shape {(nt, m_end - m_begin)}")
        L.check(self._lib.rscm_ens_get_series(self._h, self._var(var), t_begin, t_end, t_stride,
                                              m_begin, m_end, L.dptr(out)))
        return out

    def series_devptr(self, var) -> int:
        p = C.c_void_p()
        L.check(self._lib.rscm_ens_series_devptr(self._h, self._var(var), C.byref(p)))
        return p.value

    def status(self) -> np.ndarray:
        out = np.empty(self.n_members, dtype=np.uint8)
        L.check(self._lib.rscm_ens_status(self._h, L.bptr(out)))
        return out

    def status_device(self) -> DeviceVector:
        p = C.c_void_p()
        L.check(self._lib.rscm_ens_status_devptr(self._h, C.byref(p)))
        self.sync()
        return DeviceVector(p.value, self.n_members, np.uint8, self)

    def loglik(self, obs_var, obs_tidx, obs_value, obs_sigma, normalize: bool = False, on_device: bool = False):
        """Gaussian log-likelihood per member, ``[N]``; ``on_device`` leaves it in device memory (a
        ``DeviceVector``) for a reduction or all-gather without a host round trip."""
        ov = np.ascontiguousarray([self._var(v) for v in np.atleast_1d(obs_var)], dtype=np.int32)
        ot = np.ascontiguousarray(obs_tidx, dtype=np.int32)
        val, sig = L.f64(obs_value), L.f64(obs_sigma)
        if not (len(ov) == len(ot) == len(val) == len(sig)):
            raise ValueError("observation arrays differ in length")
        if on_device:
            p = C.c_void_p()
            L.check(self._lib.rscm_ens_loglik_device(self._h, len(ov), L.iptr(ov), L.iptr(ot), L.dptr(val),
                                                     L.dptr(sig), int(normalize), C.byref(p)))
            return DeviceVector(p.value, self.n_members, np.float64, self)
        out = np.empty(self.n_members)
        L.check(self._lib.rscm_ens_loglik(self._h, len(ov), L.iptr(ov), L.iptr(ot), L.dptr(val),
                                          L.dptr(sig), int(normalize), L.dptr(out)))
        return out

    def run_loglik(self, obs_var, obs_tidx, obs_value, obs_sigma, normalize: bool = False, on_device: bool = False):
        """Fused run + Gaussian log-likelihood: no series is written (see rscm_ens_run_loglik)."""
        ov = np.ascontiguousarray([self._var(v) for v in np.atleast_1d(obs_var)], dtype=np.int32)
        ot = np.ascontiguousarray(obs_tidx, dtype=np.int32)
        val, sig = L.f64(obs_value), L.f64(obs_sigma)
        if not (len(ov) == len(ot) == len(val) == len(sig)):
            raise ValueError("observation arrays differ in length")
        if on_device:
            p = C.c_void_p()
            L.check(self._lib.rscm_ens_run_loglik_device(self._h, len(ov), L.iptr(ov), L.iptr(ot), L.dptr(val),
                                                         L.dptr(sig), int(normalize), C.byref(p)))
            return DeviceVector(p.value, self.n_members, np.float64, self)
        out = np.empty(self.n_members)
        L.check(self._lib.rscm_ens_run_loglik(self._h, len(ov), L.iptr(ov), L.iptr(ot), L.dptr(val),
                                              L.dptr(sig), int(normalize), L.dptr(out)))
        return out

    def summary(self, var, tidx: int) -> Dict[str, float]:
        out = np.empty(4)
        L.check(self._lib.rscm_ens_summary(self._h, self._var(var), tidx, L.dptr(out)))
        cnt = out[0]
        return {"count": int(cnt), "mean": out[1] / cnt if cnt else float("nan"),
                "min": out[2], "max": out[3]}

    def summary_series(self, var, t_begin: int = 0, t_end: Optional[int] = None) -> Dict[str, np.ndarray]:
        """Ensemble count / mean / min / max over the finite members at every time index of
        ``[t_begin, t_end)`` -- the plume of a variable -- reduced on the device in two launches."""
        t_end = self.n_times if t_end is None else t_end
        out = np.empty((max(0, t_end - t_begin), 4))
        L.check(self._lib.rscm_ens_summary_series(self._h, self._var(var), t_begin, t_end, L.dptr(out)))
        cnt = out[:, 0]
        with np.errstate(all="ignore"):
            mean = np.where(cnt > 0, out[:, 1] / np.where(cnt > 0, cnt, 1.0), np.nan)
        return {"count": cnt.astype(np.int64), "mean": mean, "min": out[:, 2].copy(), "max": out[:, 3].copy()}

    def quantile_series(self, var, q, t_begin: int = 0, t_end: Optional[int] = None) -> Dict[str, np.ndarray]:
        """Ensemble quantiles at every time index of ``[t_begin, t_end)``, reduced on the device:
        ``numpy.nanquantile(series[t], q)`` (linear method) per row.  Returns ``{"count": [rows],
        "quantiles": [rows][len(q)]}``."""
        qq = np.atleast_1d(L.f64(q))
        t_end = self.n_times if t_end is None else t_end
        rows = max(0, t_end - t_begin)
        out, cnt = np.empty((rows, qq.size)), np.empty(rows)
        L.check(self._lib.rscm_ens_quantile_series(self._h, self._var(var), t_begin, t_end, qq.size, L.dptr(qq), L.dptr(out), L.dptr(cnt)))
        return {"count": cnt.astype(np.int64), "quantiles": out}


class _PinnedOwner:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            L.load().rscm_gpu_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def run_lockstep(ensembles: Sequence["Ensemble"], step_end: Optional[int] = None, *, sync: bool = True) -> None:
    """``Model::run`` over linked ensembles: every step, each ensemble in the order given advances by
    one step (``rscm_ens_run_lockstep``); all must stand at the same time index and share a stream."""
    first = ensembles[0]
    end = first.n_times - 1 if step_end is None else step_end
    arr = (C.c_void_p * len(ensembles))(*[e._h.value for e in ensembles])
    L.check(first._lib.rscm_ens_run_lockstep(arr, len(ensembles), first.time_index, end))
    if sync:
        first.sync()


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """numpy array over page-locked host memory (hipHostMalloc); freed with the array."""
    lib = L.load()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    L.check(lib.rscm_gpu_host_alloc(max(n, 1), C.byref(p)))
    owner = _PinnedOwner(p.value)
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    arr = arr.view(_PinnedArray)
    arr._owner = owner
    return arr


class _PinnedArray(np.ndarray):
    """ndarray that keeps its page-locked allocation alive."""

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def selftest_div(num, den, device: int = 0):
    """(ref, fast, used_fast) of num/den on the device; see rscm_gpu_selftest_div."""
    lib = L.load()
    a, b = L.f64(num).ravel(), L.f64(den).ravel()
    ref, fast = np.empty_like(a), np.empty_like(a)
    used = np.empty(a.size, dtype=np.uint8)
    L.check(lib.rscm_gpu_selftest_div(device, a.size, L.dptr(a), L.dptr(b), L.dptr(ref),
                                      L.dptr(fast), L.bptr(used)))
    return ref, fast, used
