"""aggregate rocprofv3 --pmc counter_collection csv files: mean counter value per kernel name (raw counter units, KiB)"""
import csv, glob, json, os, sys
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name") or r.get("Kernel Name"); c = r.get("Counter_Name"); v = float(r.get("Counter_Value") or 0)
            e = out.setdefault(k, {}).setdefault(c, {"sum": 0.0, "n": 0})
            e["sum"] += v; e["n"] += 1
res = {k: {c: {"mean": e["sum"] / e["n"], "n": e["n"]} for c, e in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
