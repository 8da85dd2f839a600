"""Per-step summary of a slim kernel trace (start_ns,end_ns,queue,name — written by tools/_run_trace*.sh): wall time
between optimizer kernels, union busy time, per-queue kernel time, main-queue gaps and the kernel families.
    python tools/trace_summary.py gpurun_out/r3_trace_ddpm_slim.csv [marker=k_masked_adam]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_masked_adam"
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["start_ns"]), int(r["end_ns"])
    idx = [i for i, r in enumerate(rows) if marker in r["name"]]
    a, b = idx[0], idx[-1]
    n = len(idx) - 1
    seg = rows[a + 1:b + 1]
    print(f"{n} steady steps, wall {(rows[b]['e'] - rows[a]['e']) / n / 1e6:.2f} ms/step, {len(seg) / n:.0f} kernels/step")
    iv = sorted((r["s"], r["e"]) for r in seg)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"union of all kernel intervals {busy / n / 1e6:.2f} ms/step")
    by_q = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        by_q[r["queue"]][0] += r["e"] - r["s"]
        by_q[r["queue"]][1] += 1
    for q, (t, c) in sorted(by_q.items(), key=lambda kv: -kv[1][0]):
        print(f"queue {q}: {t / n / 1e6:8.2f} ms/step in {c / n:7.0f} kernels/step")
    mainq = max(by_q.items(), key=lambda kv: kv[1][0])[0]
    fam = collections.defaultdict(float)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        if r["queue"] != mainq:
            continue
        nm = r["name"]
        key = ("convolution" if "conv_" in nm else "ATen" if "at::" in nm else "GroupNorm" if "k_gn" in nm else
               "library GEMM" if "Cijk" in nm else "other")
        fam[key] += r["e"] - r["s"]
        agg[nm[:64]][0] += r["e"] - r["s"]
        agg[nm[:64]][1] += 1
    print("main queue by family (ms/step):", {k: round(v / n / 1e6, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{t / n / 1e3:9.1f} us/step {c / n:7.1f} calls avg {t / c / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
