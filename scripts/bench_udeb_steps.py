"""ClimateUDEB stepped one model step per launch (as inside a linked graph) against whole-axis launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for members in [int(a) for a in sys.argv[1:]] or [65_536, 125_000]:
    e = bench.make_udeb_ensemble(members, 0)
    e.run()
    whole = e.last_run_ms() / 750
    e.rewind()
    e.run(100)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        e.run(e.time_index + 1, sync=False)
    e.sync()
    per = (time.perf_counter() - t0) / 200 * 1e6
    print(f"udeb {members}: whole-axis launch {whole * 1e3:.1f} us per year; one step per launch {per:.1f} us", flush=True)
    e.close()
