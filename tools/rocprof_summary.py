"""Turn a rocprofv3 rocpd database (kernel trace) into a small markdown summary for profiles/.
usage: python tools/rocprof_summary.py <results.db> <out.md> "<title>" "<command>" """
import sqlite3
import sys


def main():
    db, out, title, cmd = sys.argv[1:5]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# {title}\n\nCommand (MI355X box, cwd /tmp): `{cmd}`\n\n")
        f.write("`top_kernels` view of the rocpd database written by rocprofv3 (durations in microseconds):\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for n, calls, tot, avg, pct in rows:
            if not (n.startswith("gsr::") or "rocclr" in n or pct > 0.5):
                continue
            n = n if len(n) <= 80 else n[:77] + "..."
            f.write(f"| `{n}` | {calls} | {tot:.1f} | {avg:.3f} | {pct:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
