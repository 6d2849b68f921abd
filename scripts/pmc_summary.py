"""Per-dispatch averages of the SQ counters and kernel-trace averages of the transform kernels under a
scripts/profile_pipe.sh output directory (argv[1])."""
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == list(acc[k].keys())[0]: n[k] += 1
    for k, c in acc.items():
        if "k_p<" in k or "k_ph<" in k or "k_f<" in k or "k_fb" in k:
            print(sub, k, "dispatches", n[k])
            for name, v in c.items():
                print(f"    {name:28s} {v / max(n[k], 1):16.0f} per dispatch")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_p<" in r["Name"] or "k_ph<" in r["Name"] or "k_f<" in r["Name"] or "k_fb" in r["Name"]:
            print("trace", r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
