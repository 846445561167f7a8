"""Average rocprofv3 --pmc counters per kernel over the given output directories (usage: pmc_sum.py DIR...)."""
import csv, glob, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            m = re.search(r"(transcode_\w+|encode_\w+|filter_range)", row["Kernel_Name"])
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "dispatches", len(next(iter(cs.values()))))
