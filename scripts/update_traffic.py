"""Point an entry of profiles/traffic.json at a summary written by scripts/summarize_profile.py:

    python3 scripts/update_traffic.py 'two_layer|1000000|exact' profiles/r4_exact_1e6.txt

reads the summary's "HBM traffic per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = R MB read + W MB written" line."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    key, summary = sys.argv[1], sys.argv[2]
    text = open(os.path.join(ROOT, summary)).read()
    m = re.search(r"HBM traffic per launch = .*? = ([0-9.]+) MB read \+ ([0-9.]+) MB written", text)
    if not m:
        raise SystemExit(f"{summary}: no 'HBM traffic per launch' line")
    read, written = float(m.group(1)) * 1e6, float(m.group(2)) * 1e6
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path))
    table[key] = {"bytes": read + written, "read": read, "written": written, "source": summary}
    json.dump(table, open(path, "w"), indent=1)
    print(f"{key}: {read / 1e6:.1f} MB read + {written / 1e6:.1f} MB written ({summary})")


if __name__ == "__main__":
    main()
