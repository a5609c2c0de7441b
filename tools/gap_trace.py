"""What runs between two fit launches: reads a `rocprofv3 --kernel-trace --output-format csv` kernel trace of a
bench.py run and prints, for every pair of consecutive launches, the idle stretch between the last fit kernel of one and
the first fit kernel of the next with the kernels that ran inside it (name, queue, start / end relative to the end of
the previous launch).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python3 bench.py --steps 4 --warmup 1 <light flags>
    python tools/gap_trace.py gpurun_out/gap
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % root)
    rows = []
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                             r.get("Stream_Id", "?")))
    rows.sort()
    fit = [r for r in rows if "k_svgp_fit" in r[2]]
    if not fit:
        sys.exit("no fit kernels in the trace")
    # launches: fit kernels whose starts are within 10 ms of an already-running fit kernel belong together
    launches, cur = [], [fit[0]]
    cur_end = fit[0][1]
    for r in fit[1:]:
        if r[0] > cur_end + 1_000:  # starts after every kernel of the current launch has ended
            launches.append(cur)
            cur = [r]
            cur_end = r[1]
        else:
            cur.append(r)
            cur_end = max(cur_end, r[1])
    launches.append(cur)
    print("%d fit launches" % len(launches))
    for a, b in zip(launches, launches[1:]):
        end_a = max(r[1] for r in a)
        start_b = min(r[0] for r in b)
        span_a = (end_a - min(r[0] for r in a)) / 1e6
        print("launch of %.1f ms (%d kernels), then %.2f ms without a fit kernel:" % (span_a, len(a), (start_b - end_a) / 1e6))
        # the tail of launch a: when its kernels ended
        for r in sorted(a, key=lambda r: r[1])[-4:]:
            print("      tail  %-40s q %-3s ends %+8.2f ms" % (r[2][:40], r[3], (r[1] - end_a) / 1e6))
        for r in rows:
            if r[1] > end_a - 5_000_000 and r[0] < start_b and "k_svgp_fit" not in r[2]:
                print("      %-46s q %-3s st %-3s %+8.3f .. %+8.3f ms" % (r[2][:46], r[3], r[4], (r[0] - end_a) / 1e6,
                                                                        (r[1] - end_a) / 1e6))
        for r in sorted(b)[:6]:
            print("      next  %-40s q %-3s starts %+8.2f ms" % (r[2][:40], r[3], (r[0] - end_a) / 1e6))


if __name__ == "__main__":
    main()
