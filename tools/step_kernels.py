"""Kernel sequence of ONE bench step from a rocprofv3 kernel trace: everything between the last two mask_decode launches, with
the kernels that are not this library's (framework kernels) marked and summed.
usage: python tools/step_kernels.py <dir or *_kernel_trace.csv> [--all]"""
import csv
import glob
import os
import re
import sys


def short(n):
    m = re.search(r"svps::(\w+::)?(\w+)(<[^>]*>)?", n)
    if m:
        return m.group(2) + (m.group(3) or ""), True
    m = re.search(r"_ZN4svps\d+(\w+?)(ILb|E)", n)
    if m:
        return m.group(1), True
    return n[:70], False


def main():
    src = sys.argv[1]
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for fn in files:
        with open(fn) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "mask_decode" in r["Kernel_Name"]]
    if len(idx) < 2:
        raise SystemExit("fewer than two mask_decode launches in the trace")
    a, b = idx[-2], idx[-1]
    own_us = other_us = 0.0
    others = {}
    for r in rows[a + 1:b + 1]:
        name, own = short(r["Kernel_Name"])
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if own:
            own_us += us
        else:
            other_us += us
            others[name] = others.get(name, 0) + 1
        if "--all" in sys.argv:
            print(f"{'' if own else 'FRAMEWORK '}{name:60s} {us:9.1f}")
    print(f"launches in the step: {b - a}; library kernels {own_us / 1e3:.3f} ms; framework kernels {other_us / 1e3:.3f} ms in {sum(others.values())} launches")
    for k, v in others.items():
        print(f"  FRAMEWORK x{v}: {k}")


if __name__ == "__main__":
    main()
