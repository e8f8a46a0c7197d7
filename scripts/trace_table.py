"""Per-(kernel, grid size) table of a rocprofv3 --kernel-trace run, plus the timeline of a few consecutive model steps of the
largest ensemble in it (start / end of every launch relative to the first: where the gaps between dependent kernels are).

    python3 scripts/trace_table.py <dir with *_kernel_trace.csv> [min grid of the 'big' run] > out.txt
"""
import collections
import csv
import glob
import sys


def short(name):
    # (nested template arguments of the anonymous namespace keep their names: group_split_seq_kernel<OpKinds<...>, ...>)
    head = name[5:] if name.startswith("void ") else name
    head = head.replace("rscm::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return head.split("(rscm::")[0][:60]


def main():
    d = sys.argv[1]
    big_min = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    rows = []
    for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    acc = collections.defaultdict(list)
    for r in rows:
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{'kernel':62s} {'grid':>9s} {'regs':>5s} {'calls':>6s} {'avg us':>9s} {'min':>8s} {'max':>8s} {'total ms':>9s}")
    for (n, g, v), t in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if sum(t) < 50.0:
            continue
        print(f"{n:62s} {g:9d} {v:5d} {len(t):6d} {sum(t) / len(t):9.1f} {min(t):8.1f} {max(t):8.1f} {sum(t) / 1e3:9.2f}")
    # launches of one kernel may overlap (two member blocks on two streams): per kernel name, the time during which at least one of
    # its launches was running (the union of the intervals) beside the sum of the durations
    spans = collections.defaultdict(list)
    for r in rows:
        spans[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print("\n# per kernel name: launches, sum of durations, time with at least one launch running (overlapping launches counted once)")
    for n, iv in sorted(spans.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        total = sum(b - a for a, b in iv)
        if total < 50_000:
            continue
        iv.sort()
        busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
        for a, b in iv[1:]:
            if a > cur_b:
                busy += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        busy += cur_b - cur_a
        print(f"{n:62s} {len(iv):6d} launches {total / 1e6:10.3f} ms summed {busy / 1e6:10.3f} ms busy")
    big = sorted((r for r in rows if int(r["Grid_Size_X"]) >= big_min), key=lambda r: int(r["Start_Timestamp"]))
    if len(big) > 400:
        s = big[len(big) // 2: len(big) // 2 + 14]
        t0 = int(s[0]["Start_Timestamp"])
        print("\n# consecutive launches of the big run, mid-run (us from the first one's start): start, end, gap to the previous end")
        prev = None
        for r in s:
            a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
            print(f"{short(r['Kernel_Name']):62s} {a:9.1f} {b:9.1f} {'' if prev is None else f'{a - prev:7.1f}'}")
            prev = b


if __name__ == "__main__":
    main()
