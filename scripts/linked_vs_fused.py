"""The coupled chain as four linked components in one fused launch (group_seq_kernel) against the fused coupled kernel
on the same card (bench.py: extra.coupled_linked_1e6), on its own.
    python scripts/linked_vs_fused.py [members]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rscm_amd import _lib as L  # noqa: E402

members = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
stream = C.c_void_p()
L.check(L.load().rscm_gpu_stream_create(0, C.byref(stream)))
print(json.dumps(bench.linked_graph_extra(members, 0, stream.value, bench.T1 - bench.T0)))
