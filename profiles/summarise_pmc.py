"""Per-kernel means of every counter in the rocprofv3 counter_collection CSVs below a directory."""
import collections
import csv
import os
import sys


def main(root):
    for base, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            if not f.endswith("counter_collection.csv"):
                continue
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(os.path.join(base, f))):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if name.startswith("__amd"):
                    continue
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(agg.items()):
                print("%-28s %-32s n=%-3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
