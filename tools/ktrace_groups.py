"""rocprofv3 --kernel-trace csv -> the launches of the token's kernels (k_layers) as back-to-back GROUPS (a gap above 2 ms starts a new one): count, average / min duration and where the
group starts.  The load-time warm-up replays every greedy graph once (flm_gpu.hip warm_up), so the profiler's per-kernel average mixes those launches with the timed region's: this
lists them apart -- the timed region of `bench.py --steps K --warmup W` is the group of W + K launches.   python tools/ktrace_groups.py run_kernel_trace.csv [name-substring]"""
import csv, sys
path = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "k_layers"
rows = {}
for r in csv.DictReader(open(path)):
    k = r.get("Kernel_Name") or r.get("Kernel Name") or ""
    if want not in k: continue
    rows.setdefault(k, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0 = min(s for v in rows.values() for s, _ in v) if rows else 0
for k, v in sorted(rows.items()):
    v.sort(); groups = [[v[0]]]
    for a, b in zip(v, v[1:]):
        if b[0] - a[1] > 2_000_000: groups.append([])
        groups[-1].append(b)
    print(f"{k[:100]}   ({len(v)} launches, average {sum(e - s for s, e in v) / len(v) / 1000:.2f} us)")
    for g in groups:
        d = [(e - s) / 1000 for s, e in g]
        print(f"    group of {len(g):3d} launches starting {(g[0][0] - t0) / 1e9:8.3f} s after the first: average {sum(d) / len(d):8.2f} us   min {min(d):8.2f}   max {max(d):8.2f}")
