"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace csv directory (the step between the last two
launches of the blend kernel): start / end / duration in microseconds relative to the previous blend kernel's end."""
import csv
import glob
import sys


def main(path, anchor="render_kernel"):
    files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for fn in files:
        rows += list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
    t0 = int(rows[a]["End_Timestamp"])
    for r in rows[a:b + 1]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:90]}")


if __name__ == "__main__":
    main(*sys.argv[1:])
