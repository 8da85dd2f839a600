"""Per-hardware-queue view of the last full step of a rocprofv3 kernel trace (`*_kernel_trace.csv`): busy time, first / last
kernel and the longest idle gaps of the busiest queue, plus every collective kernel.  Marker = the fused optimizer kernel.
    python tools/queue_timeline.py trace.csv [marker]"""
import collections, csv, sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_masked_adam"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = marks[-2], marks[-1]
seg = rows[a + 1:b + 1]
t0 = rows[a][1]
print(f"step wall {(rows[b][1] - rows[a][1]) / 1e6:.2f} ms, {len(seg)} kernels")
byq = collections.defaultdict(list)
for s, e, n, q, st in seg:
    byq[(q, st)].append((s, e, n))
for key, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in ks) / 1e6
    print(f"  queue {key[0]} stream {key[1]}: {len(ks):6d} kernels, busy {busy:8.2f} ms, first {(ks[0][0] - t0) / 1e6:8.2f} ms, last end {(max(e for _, e, _ in ks) - t0) / 1e6:8.2f} ms")
main = max(byq.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))[1]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(main, main[1:]):
    if s1 - e0 > 50_000:
        gaps.append((s1 - e0, e0, n0, n1))
print(f"busiest queue: {len(gaps)} gaps > 50 us, {sum(g[0] for g in gaps) / 1e6:.2f} ms in total; the longest:")
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
for g, e0, n0, n1 in sorted(gaps, reverse=True)[:14]:
    print(f"   {g / 1e3:8.1f} us at {(e0 - t0) / 1e6:7.2f} ms   {short(n0)} -> {short(n1)}")
for s, e, n, q, st in seg:
    if "ccl" in n.lower():
        print(f"  collective {short(n)} queue {q}: {(s - t0) / 1e6:7.2f} .. {(e - t0) / 1e6:7.2f} ms ({(e - s) / 1e3:.0f} us)")
