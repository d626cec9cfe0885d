"""aggregate rocprofv3 --pmc counter_collection csv files: mean counter value per kernel name (raw counter units, KiB).
--last N: only the last N dispatches of every kernel (the load-time warm-up replays every greedy graph once -- flm_gpu.hip warm_up --: the timed legs of bench.py are what comes last)"""
import csv, glob, json, os, sys
args = sys.argv[1:]; last = 0
if "--last" in args: i = args.index("--last"); last = int(args[i + 1]); del args[i:i + 2]
out = {}
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name") or r.get("Kernel Name"); c = r.get("Counter_Name"); v = float(r.get("Counter_Value") or 0)
            out.setdefault(k, {}).setdefault(c, []).append((int(r.get("Dispatch_Id") or 0), v))
res = {}
for k, cs in out.items():
    for c, vals in cs.items():
        vals.sort(); vals = vals[-last:] if last else vals
        res.setdefault(k, {})[c] = {"mean": sum(v for _, v in vals) / len(vals), "n": len(vals)}
print(json.dumps(res, indent=1))
