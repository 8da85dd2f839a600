"""Compact timeline of one steady step from a slim kernel trace (tools/prof.sh with KEEP_TRACE=1): per-family kernel
time, when the main / side chains end, and every kernel after `--from` us.  Marker = the fused update kernel."""
import csv, collections, sys
path = sys.argv[1]
t_from = float(sys.argv[2]) if len(sys.argv) > 2 else 1e18
marker = sys.argv[3] if len(sys.argv) > 3 else "k_masked_sgd"
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["start_ns"]), int(r["end_ns"])
idx = [i for i, r in enumerate(rows) if marker in r["name"]]
pairs = [(a, b) for a, b in zip(idx, idx[1:]) if b - a > 100]
a, b = pairs[-2]
seg = rows[a + 1:b + 1]
t0 = rows[a]["e"]
print(f"step wall {(rows[b]['e'] - rows[a]['e']) / 1e3:.1f} us, {len(seg)} kernels, {len(pairs)} steps in the trace")
fam = collections.defaultdict(lambda: [0.0, 0])
for r in seg:
    n = r["name"].replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("<")[0].split("(")[0][:40]
    fam[n][0] += (r["e"] - r["s"]) / 1e3
    fam[n][1] += 1
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {k:40s} {t:8.1f} us x{c}")
iv = sorted((r["s"], r["e"]) for r in seg)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"union busy {busy / 1e3:.1f} us")
for r in sorted(seg, key=lambda r: r["s"]):
    if (r["s"] - t0) / 1e3 >= t_from:
        print(f"   {(r['s'] - t0) / 1e3:8.1f} {(r['e'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f}  {r['name'][:70]}")
