"""Turn the per-kernel PMC tables of tools/pmc.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, SEPARATE passes) into
HBM bytes per launch next to the algorithmic bytes.

    python tools/pmc_traffic.py <tag>:<N> [<tag>:<N> ...] [<tag>:<kernel substring>=<algorithmic bytes> ...] > profiles/rNN_pmc_traffic.json

(the second form: any kernel of a table by name, with the algorithmic bytes of one launch given by the caller)

reads gpurun_out/<tag>_FETCH_SIZE.csv and gpurun_out/<tag>_WRITE_SIZE.csv.  Units: rocprofv3 reports KiB; FETCH_SIZE
is doubled for wide coalesced reads on gfx950 (MI355X_MICROARCH.md §HBM).  The top-k is reported per CALL (sum over its
kernels x dispatches per call)."""
import csv, json, os, sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG = {"k_masked_sgd_vec": 21, "k_masked_adam": 29, "k_saliency_accumulate": 12, "k_sqnorm_partial": 4}
TOPK = ("k_sample", "k_bracket", "k_main", "k_hist_a", "k_resolve", "k_finish", "k_fullscan", "k_gather_sample",  # (k_hist_a: rounds 2-4)
        "k_bracket_from_hist")


def table(path):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        out[r["kernel"]] = (int(r["dispatches"]), float(list(r.values())[2]))
    return out


def main():
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc.sh), KiB units, FETCH_SIZE "
                   "doubled per MI355X_MICROARCH.md §HBM (gfx950 wide coalesced reads); top-k figures are per CALL of "
                   "salun_mask_topk (one threshold), summed over its kernels", "kernels": {}}
    for spec in sys.argv[1:]:
        tag, n = spec.split(":", 1)
        if "=" in n:
            kern, alg = n.split("=")
            alg = float(alg)
            f = table(os.path.join(HERE, "gpurun_out", f"{tag}_FETCH_SIZE.csv"))
            w = table(os.path.join(HERE, "gpurun_out", f"{tag}_WRITE_SIZE.csv"))
            fk = [k for k in f if kern in k]
            wk = [k for k in w if kern in k]
            if fk and wk:
                rd, wr = 2 * f[fk[0]][1] * 1024, w[wk[0]][1] * 1024
                res["kernels"][f"{kern}@{tag}"] = {"fetch_size_kib_raw": f[fk[0]][1], "write_size_kib_raw": w[wk[0]][1],
                                                   "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes": rd + wr,
                                                   "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr) / alg}
            continue
        n = int(n)
        f = table(os.path.join(HERE, "gpurun_out", f"{tag}_FETCH_SIZE.csv"))
        w = table(os.path.join(HERE, "gpurun_out", f"{tag}_WRITE_SIZE.csv"))
        for kern, bpe in ALG.items():
            fk = [k for k in f if kern in k]
            wk = [k for k in w if kern in k]
            if not fk or not wk:
                continue
            rd, wr = 2 * f[fk[0]][1] * 1024, w[wk[0]][1] * 1024
            res["kernels"][f"{kern}@{tag}"] = {"fetch_size_kib_raw": f[fk[0]][1], "write_size_kib_raw": w[wk[0]][1],
                                               "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes": rd + wr,
                                               "algorithmic_bytes": bpe * n,
                                               "traffic_over_algorithmic": (rd + wr) / (bpe * n)}
        # top-k: per call = sum over kernels of (mean per dispatch x dispatches) / calls; calls = dispatches of k_main
        # on the outer vector (the most expensive k_main row)
        mains = sorted(((v[1], v[0], k) for k, v in f.items() if "k_main" in k), reverse=True)
        if mains:
            calls = mains[0][1]
            rd = sum(2 * v[1] * 1024 * v[0] for k, v in f.items() if any(t in k for t in TOPK)) / calls
            wr = sum(v[1] * 1024 * v[0] for k, v in w.items() if any(t in k for t in TOPK)) / calls
            res["kernels"][f"salun_mask_topk(nk=1)@{tag}"] = {
                "calls": calls, "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes": rd + wr,
                "algorithmic_bytes": 5 * n, "algorithmic_read_bytes": 4 * n, "read_over_algorithmic_read": rd / (4 * n),
                "traffic_over_algorithmic": (rd + wr) / (5 * n)}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
