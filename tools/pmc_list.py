"""Per-dispatch rocprofv3 --pmc counters of the kernels matching a regex (usage: pmc_list.py DIR REGEX)."""
import csv, glob, sys, re, collections
rows = collections.OrderedDict()
for path in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        m = re.search(sys.argv[2], row["Kernel_Name"])
        if m:
            rows.setdefault(int(row["Dispatch_Id"]), {"grid": row["Grid_Size"]})[row["Counter_Name"]] = float(row["Counter_Value"])
for d, r in rows.items():
    print(d, r)
