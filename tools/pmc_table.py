"""Per-kernel means of rocprofv3 --pmc counters (csv output).  usage: python tools/pmc_table.py <dir-with-*counter_collection.csv> ..."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "gsr::" not in k:
                    continue
                if r.get("Grid_Size_Y", "1") not in ("1", ""):
                    pass
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in acc for c in acc[k]})
    print("| kernel | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for k in sorted(acc):
        row = []
        for c in names:
            v = acc[k].get(c)
            row.append(f"{sum(v) / len(v):.3g}" if v else "")
        print(f"| `{k}` | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
