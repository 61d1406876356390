"""Turn a rocprofv3 results .db (rocpd sqlite) into a small text summary for profiles/ (kernel-trace --stats)."""
import sqlite3
import sys


def main(db, out, title):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats (view top_kernels); durations in microseconds\n")
        f.write("name,calls,total_us,avg_us,percent\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if short.startswith("Cijk_"):
                short = short[:40] + "...(rocBLAS/hipBLASLt, Net.lin)"
            if len(short) > 110:
                short = short[:107] + "..."
            f.write(f"\"{short}\",{calls},{tot:.1f},{avg:.2f},{pct:.2f}\n")
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
