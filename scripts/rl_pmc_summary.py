"""Developer tool: per-dispatch averages of SQ counters for the Richardson-Lucy tile kernels in rocprofv3
--pmc output directories (argv[1:]); dispatches with the full grid only (the first launches of a call)."""
import collections, csv, glob, sys
for out in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rl_step" not in r["Kernel_Name"]: continue
            k = ("sep" if "k_rl_step_sep" in r["Kernel_Name"] else "wide" if "ILb1" in r["Kernel_Name"] else "narrow", r["Grid_Size"], r["Workgroup_Size"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k, c in sorted(acc.items()):
        print(out, k)
        for name, v in c.items():
            print(f"    {name:28s} {v / max(n[k][name], 1):14.0f} per dispatch ({n[k][name]} dispatches)")
