#!/usr/bin/env python3
"""Per-phase kernel statistics of a profiled bench.py run.

`rocprofv3 --kernel-trace --stats` averages a kernel over EVERY launch of the process; bench.py's default run launches
the headline kernel in several phases -- settle, warm-up, the timed launches, then the brackets (cold first launches after
1 s of idle, action rings larger than the Infinity Cache) -- so that one average mixes populations and cannot be compared
with the line's ms_per_step.  bench.py records how many launches each phase issued (`rank_times.phases`, in order); this
script cuts the kernel trace of the same run at those counts and writes one row per phase.

usage: phase_stats.py <kernel_trace.csv> <bench.json> [out.csv]
The `timed` row is the one to hold against the line: its average must not exceed `ms_per_step`."""
import csv
import json
import sys


def main():
    trace, bench = sys.argv[1], sys.argv[2]
    rec = json.loads([l for l in open(bench).read().splitlines() if l.startswith("{")][-1])
    kernel = rec["roofline"]["kernel"].replace(",", ", ")          # bench names it without blanks, the trace with
    phases = rec["rank_times"]["phases"]
    want_name = kernel.replace(" ", "").rstrip(">")            # (the trace spells out defaulted template arguments: match the prefix)
    rows = [r for r in csv.DictReader(open(trace)) if want_name in r["Kernel_Name"].replace(" ", "").replace("nig::", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    want = sum(p["launches"] for p in phases)
    out = [("phase", "kernel", "calls", "avg_us", "median_us", "min_us", "max_us", "first_start_ns", "span_ms")]
    if len(rows) != want:
        print(f"# WARNING: {len(rows)} launches of {kernel} in the trace, the line's phases add up to {want}", file=sys.stderr)
    i = 0
    for p in phases:
        seg = rows[i:i + p["launches"]]
        i += p["launches"]
        if not seg:
            continue
        d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg)
        span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
        out.append((p["name"], kernel, len(seg), f"{sum(d) / len(d):.2f}", f"{d[len(d) // 2]:.2f}", f"{d[0]:.2f}", f"{d[-1]:.2f}",
                    seg[0]["Start_Timestamp"], f"{span:.3f}"))
    text = "\n".join(",".join(f'"{x}"' if isinstance(x, str) and "," in x else str(x) for x in row) for row in out)
    timed = [r for r in out[1:] if r[0] == "timed"]
    note = ""
    if timed:
        ms = rec["ms_per_step"]
        note = (f"\n# timed phase: {timed[0][3]} us average kernel time vs ms_per_step {ms * 1e3:.2f} us (wall, incl. launch gaps); "
                f"roofline.launch_us {rec['roofline']['launch_us']:.2f} us (HIP events)")
    print(text + note)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + note + "\n")


if __name__ == "__main__":
    main()
