"""Point an entry of profiles/traffic.json at a summary written by scripts/summarize_profile.py:

    python3 scripts/update_traffic.py 'two_layer|1000000|exact' profiles/r4_exact_1e6.txt

reads the summary's "HBM traffic per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = R MB read + W MB written" line and, where the
summary has them, "effective shader clock = X GHz", "VALU issue utilisation = ... = U" and "per wavefront-year over Y years = V"
(bench.py puts them into the roofline objects).  `--refresh` re-reads every entry's own source summary."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def entry(summary, source_text=None):
    text = open(os.path.join(ROOT, summary)).read()
    m = re.search(r"HBM traffic per launch = .*? = ([0-9.]+) MB read \+ ([0-9.]+) MB written", text)
    if not m:
        raise SystemExit(f"{summary}: no 'HBM traffic per launch' line")
    read, written = float(m.group(1)) * 1e6, float(m.group(2)) * 1e6
    e = {"bytes": read + written, "read": read, "written": written, "source": source_text or summary}
    for field, pattern in (("clock_ghz", r"effective shader clock = .*? = ([0-9.]+) GHz"),
                           ("valu_issue_utilisation", r"VALU issue utilisation = .*? = ([0-9.]+)"),
                           ("valu_per_wavefront_year", r"per wavefront-year over \d+ years = ([0-9.]+)\)")):
        m = re.search(pattern, text)
        if m:
            e[field] = float(m.group(1))
    return e


def main():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path))
    if sys.argv[1] == "--refresh":
        for key, old in table.items():
            if isinstance(old, dict) and "source" in old and "bytes_per_member_step" not in old:   # (per-step entries: scripts/summarize_share_pmc.py)
                summary = old["source"].split(" ")[0]
                if os.path.exists(os.path.join(ROOT, summary)):
                    table[key] = entry(summary, old["source"])
                    print(key, {k: v for k, v in table[key].items() if k != "source"})
    else:
        key, summary = sys.argv[1], sys.argv[2]
        table[key] = e = entry(summary)
        print(f"{key}: {e['read'] / 1e6:.1f} MB read + {e['written'] / 1e6:.1f} MB written ({summary}); "
              f"clock {e.get('clock_ghz')} GHz, issue utilisation {e.get('valu_issue_utilisation')}")
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
