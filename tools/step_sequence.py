"""The dispatch sequence of ONE replayed KD step out of a rocprofv3 --kernel-trace result (rocpd sqlite .db).

    python tools/step_sequence.py gpurun_out/<run>/prof/kd_results.db profiles/<name>.md [period]

period (default 1): with the grouped teacher pass (kd6d.graph.GroupedTeacherKDStep) the unit that repeats is `period` =
group consecutive steps -- one teacher pass over their batches, cut into `period` segments, plus `period` student steps;
the table then shows one whole period and the summary gives per-step figures (totals / period).

A step is delimited by its closing `clip_adamw_kernel` dispatch: everything that started after the previous step's
optimizer launch ended, up to and including this step's.  A bench invocation also holds eager warm-up, probe and
instrumented steps; the replayed ones are the majority and all have the same dispatch count, so the step shown is
the middle one of the most frequent count.  Columns: start offset inside the step (us), duration (us),
queue / stream id as the trace reports it, grid size in workgroups, kernel.  Under the tracer a hipGraph replays its
nodes one at a time, so the offsets are the serialised order, not the concurrent schedule (that one is
tools/step_timeline.py, device timestamps from inside the graph); what this table pins is WHICH dispatches a step is
made of -- the count VERDICT r1 asked to bring down -- and who launched them (kd6d kernels vs torch's).
"""
import sqlite3
import sys

from rocprof_summary import short


def main():
    db, out = sys.argv[1], sys.argv[2]
    period = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("PRAGMA table_info(kernels)")]
    pick = lambda *names: next((n for n in names if n in cols), None)      # noqa: E731
    c_start, c_end = pick("start", "start_ns"), pick("end", "end_ns")
    c_queue = pick("stream_id", "queue_id", "stream", "queue")
    c_grid = pick("grid_size", "grid_x", "grid_size_x")
    c_wg = pick("workgroup_size", "workgroup_x", "workgroup_size_x")
    sel = "name, %s, %s, %s, %s, %s" % (c_start, c_end, c_queue or "0", c_grid or "0", c_wg or "1")
    rows = list(c.execute("select %s from kernels order by %s" % (sel, c_start)))
    opt = [i for i, r in enumerate(rows) if "clip_adamw" in r[0]]
    if len(opt) < 3:
        raise SystemExit("need at least 3 optimizer launches in the trace, found %d" % len(opt))
    best = None
    for off in range(period):              # where a period starts is not known: take the phase whose spans repeat most
        spans = [(opt[i - period] + 1, opt[i]) for i in range(period + off, len(opt), period)]
        by_len = {}
        for sp in spans:
            by_len.setdefault(sp[1] - sp[0] + 1, []).append(sp)
        if not by_len:
            continue
        n_mode, group = max(by_len.items(), key=lambda kv: len(kv[1]))
        if best is None or len(group) > len(best[1]):
            best = (n_mode, group, spans)
    n_mode, group, spans = best
    lo, hi = group[len(group) // 2]
    back = "%d of %d %s with %d dispatches" % (len(group) // 2 + 1, len(group), "steps" if period == 1 else "periods of %d steps" % period, n_mode)
    step = rows[lo:hi + 1]
    t0 = step[0][1]
    ours = [r for r in step if not any(s in r[0] for s in ("at::", "rocclr", "ncclDevKernel"))]
    def n_wg(r):
        grid, wg = r[4], r[5]
        return (grid // wg) if (isinstance(grid, int) and isinstance(wg, int) and wg) else 0
    cu_time = sum((r[2] - r[1]) * min(1.0, (n_wg(r) or 256) / 256.0) for r in step) / 1e6
    per = "" if period == 1 else " = %.1f dispatches, %.3f ms of kernel time, %.3f ms of CU-time PER STEP" % (
        len(step) / period, sum(r[2] - r[1] for r in step) / 1e6 / period, cu_time / period)
    lines = ["# dispatches of one replayed KD %s (rocprofv3 --kernel-trace)%s" % ("step" if period == 1 else "period of %d steps" % period, per), "",
             "source: `%s`, step %s (%d steps in the trace); columns of the `kernels` view used: %s" %
             (db, back, len(spans), ", ".join(x for x in (c_start, c_end, c_queue, c_grid, c_wg) if x)), "",
             "%d dispatches: %d kd6d kernels, %d torch / runtime (fills, copies, RNG); serialised span %.3f ms, "
             "kernel time %.3f ms" % (len(step), len(ours), len(step) - len(ours), (step[-1][2] - t0) / 1e6,
                                      sum(r[2] - r[1] for r in step) / 1e6), "",
             "CU-time estimate: sum of duration x min(1, workgroups / 256) = %.3f ms (a kernel of fewer workgroups than "
             "CUs leaves the rest to whatever runs beside it; crude -- a workgroup need not fill its CU)" % cu_time, "",
             "| # | start us | dur us | queue | workgroups | kernel |", "|---|---|---|---|---|---|"]
    for i, (name, s, e, q, grid, wg) in enumerate(step):
        nwg = n_wg((name, s, e, q, grid, wg))
        lines.append("| %d | %.1f | %.2f | %s | %s | `%s` |" % (i, (s - t0) / 1e3, (e - s) / 1e3, q, nwg or "", short(name)))
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))
    foreign = {}
    for r in step:
        if r not in ours:
            foreign[short(r[0])] = foreign.get(short(r[0]), 0) + 1
    for k, v in sorted(foreign.items(), key=lambda kv: -kv[1]):
        print("%3d  %s" % (v, k))


if __name__ == "__main__":
    main()
