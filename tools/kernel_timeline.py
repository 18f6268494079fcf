"""Per-iteration kernel timeline from a rocprofv3 --kernel-trace CSV: for every launch of one iteration its
start offset, duration and the idle gap in front of it, averaged over the iterations of the trace.

usage: python tools/kernel_timeline.py <dir with *_kernel_trace.csv> <anchor kernel substring> [skip_iterations]
An iteration runs from one launch of the anchor kernel to the next."""
import csv
import glob
import sys


def main():
    d, anchor = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    iters = [rows[a:b] for a, b in zip(marks[:-1], marks[1:])][skip:]
    n = max(set(len(it) for it in iters), key=[len(it) for it in iters].count)
    iters = [it for it in iters if len(it) == n]
    print(f"{len(iters)} iterations of {n} launches; mean period {sum(it[-1][1] - it[0][0] for it in iters) / len(iters) / 1e3:.1f} us (first start to last end)")
    busy = gap_total = 0.0
    for k in range(n):
        start = sum(it[k][0] - it[0][0] for it in iters) / len(iters) / 1e3
        dur = sum(it[k][1] - it[k][0] for it in iters) / len(iters) / 1e3
        gap = sum(it[k][0] - max(x[1] for x in it[:k]) for it in iters) / len(iters) / 1e3 if k else 0.0
        busy += dur
        gap_total += max(gap, 0.0)
        print(f"{k:3d} +{start:8.1f} us  dur {dur:7.1f}  gap {gap:6.1f}  {iters[0][k][2][:90]}")
    period = sum(b[0][0] - a[0][0] for a, b in zip(iters[:-1], iters[1:])) / max(len(iters) - 1, 1) / 1e3
    print(f"busy {busy:.1f} us, gaps inside {gap_total:.1f} us, anchor-to-anchor {period:.1f} us")


if __name__ == "__main__":
    main()
