"""Merged kernel / memory-copy timeline from a rocprofv3 --kernel-trace --memory-copy-trace run (csv output).

usage: python tools/timeline.py DIR PREFIX [t_from_ms t_to_ms]
Times are ms from the first copy longer than 0.2 ms; copies shorter than --min-ms are dropped from the listing."""
import csv
import sys


def load(d, prefix):
    cp = list(csv.DictReader(open(f"{d}/{prefix}_memory_copy_trace.csv")))
    kt = list(csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")))
    ev = []
    for c in cp:
        s, e = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
        ev.append((s, e, "COPY " + c["Direction"][12:] + " s" + c["Stream_Id"]))
    for k in kt:
        s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
        name = k["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        ev.append((s, e, name[:28] + " q" + k["Queue_Id"] + " s" + k.get("Stream_Id", "?")))
    ev.sort()
    return ev


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    lo, hi = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (-1e18, 1e18)
    ev = load(d, prefix)
    t0 = next(s for s, e, n in ev if n.startswith("COPY") and e - s > 200000)
    for s, e, n in ev:
        t = (s - t0) / 1e6
        if lo < t < hi:
            print(f"{t:9.2f} {(e - s) / 1e6:7.3f} {n}")


if __name__ == "__main__":
    main()
