"""Experiment (run on the GPU box): ClimateUDEB over the whole axis as G member groups on G streams in S chunks of years, against one
launch -- does the dispatcher even out 1563 one-per-SIMD wavefronts the way it does for the two-layer kernel?

    python scripts/multi_stream_udeb.py [members]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import rscm_amd  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
t = np.arange(1750.0, 2501.0)
b = np.append(t, 2501.0)
F = 4.0 * (1.0 - np.exp(-(t - 1750.0) / 120.0))
lo = np.array(L.UD_DEFAULTS, dtype=float)
hi = lo.copy()
for name, (x, y) in dict(ecs=(2.0, 5.0), kappa=(0.5, 1.5)).items():
    j = L.UD_PARAM_NAMES.index(name)
    lo[j], hi[j] = x, y
lib = L.load()


def build(members, offset, total, stream):
    e = rscm_amd.Ensemble(rscm_amd.KIND_UDEB, members, b)
    if stream is not None:
        e.set_stream(stream)
    e.sample_lhs(1, lo, hi, offset, total)
    e.set_forcing(F)
    for v in (1, 2, 3, 4):
        e.set_initial(v, 0.0)
    return e


def timed(fn, reps=3):
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


whole = build(N, 0, N, None)


def one():
    whole.rewind()
    whole.run()


one()
base = timed(one)
ref = whole.get_series(7, 750, 751)[0].copy()
print(f"one launch of {N} members: {base:.2f} ms")
whole.close()
splits = [[N // 2, N - N // 2]]
if N > 65536:
    splits.append([65536, N - 65536])
for sizes in splits:
    G = len(sizes)
    streams = []
    for _ in range(G):
        s = C.c_void_p()
        L.check(lib.rscm_gpu_stream_create(0, C.byref(s)))
        streams.append(s)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    groups = [build(sizes[g], int(offs[g]), N, streams[g].value) for g in range(G)]
    print(f"groups {sizes}")
    for S in (1, 3, 5, 8, 15):
        cuts = [int(round(750 * (k + 1) / S)) for k in range(S)]

        def run():
            for e in groups:
                e.rewind()
            for c in cuts:
                for e in groups:
                    e.run(c, sync=False)
            for e in groups:
                e.sync()

        run()
        ms = timed(run)
        got = np.concatenate([e.get_series(7, 750, 751)[0] for e in groups])
        same = np.array_equal(got.view(np.uint64), ref.view(np.uint64))
        print(f"G={G} groups x S={S:2d} chunks: {ms:.2f} ms ({base / ms:.2f}x), same bits: {same}", flush=True)
    for e in groups:
        e.close()
    for s in streams:
        L.check(lib.rscm_gpu_stream_destroy(0, s))
