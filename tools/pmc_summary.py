"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_summary.py dir [dir ...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kname, cs in acc.items():
            if "rq_scan" not in kname: continue
            print(kname)
            for c, v in sorted(cs.items()):
                print(f"    {c:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
