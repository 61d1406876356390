"""Print the kernel timeline of the last `n` dispatches in a rocprofv3 results .db (kernel-trace): start offset,
duration and the idle gap before each kernel.  Used to see where a hipGraph replay spends time outside our kernels."""
import sqlite3
import sys


def main(db, n=40, anchor=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = list(c.execute("select name, start, end from kernels order by start"))
    if anchor:
        idx = [i for i, r in enumerate(rows) if anchor in r[0]]
        i = idx[-1]
        rows = rows[max(0, i - 8):i - 8 + n]
    else:
        rows = rows[-n:]
    t0 = rows[0][1]
    prev = None
    for name, s, e in rows:
        short = name.split("(")[0].replace("void ", "")[:70]
        gap = 0 if prev is None else (s - prev) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {short}")
        prev = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else None)
