"""Turn a rocprofv3 rocpd sqlite database (``*_results.db``) into the plain-text kernel summary kept
under profiles/ (top kernels: calls, total / average duration, share)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {db.rsplit('/', 1)[-1]} (durations in microseconds)",
             f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'share%':>7}  kernel"]
    for name, calls, total, avg, pct in rows:
        lines.append(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:7.2f}  {name}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
