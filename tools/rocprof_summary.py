"""Condense a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db, ROCm 7.2 default output)
into a small per-kernel table (calls, total us, avg us, %), the form kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/<run>/prof/kd_results.db profiles/<name>.md [steps]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)I(.*)E+v", name)
    if name.startswith("_Z"):
        # keep the mangled template arguments readable: kernel name + integer args
        base = re.search(r"N_1\d+([a-z0-9_]+?)(?:I|E)", name)
        ints = re.findall(r"Li(\d+)E", name)
        ty = "bf16" if "DF16b" in name else ("f32" if "If" in name else "")
        return "%s<%s%s>" % (base.group(1) if base else name[:40], ty, ("," + ",".join(ints)) if ints else "")
    name = re.sub(r"\(.*", "", name)
    return name[-90:]


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary", "",
             "source: `%s`; %d dispatches, %.3f ms total kernel time%s" %
             (db, n, tot / 1e6, (" over %d steps = %.3f ms/step, %.0f dispatches/step" %
                                 (steps, tot / 1e6 / steps, n / steps)) if steps else ""), "",
             "| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, cnt, s, a, mn, mx in rows:
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" %
                     (short(name), cnt, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main()
