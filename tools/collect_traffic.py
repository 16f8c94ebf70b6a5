"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into profiles/<round>_traffic.json.

HBM bytes per launch (x launches_per_step = per step: the dS hand-over launches its kernels once per chunk of (b,h) units) = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced stream (16 B per lane, buffer_load ... lds
included) — MI355X_MICROARCH.md §HBM — which is how every tensor is read here.
usage: collect_traffic.py out.json fetch.csv write.csv
"""
import collections
import csv
import json
import sys

SHORT = {"fwd_mfma_kernel": "fwd_mfma", "fwd_mfma_stag_kernel": "fwd_mfma", "bwd_dkdv_mfma_kernel": "bwd_mfma", "bwd_mfma_kernel": "bwd_mfma",
         "bwd_dq_mfma_kernel": "bwd_dq_mfma", "bwd_dkdv_w4_kernel": "bwd_mfma", "bwd_dq_w4_kernel": "bwd_dq_mfma", "bwd_dq_ds_kernel": "bwd_dq_mfma", "bwd_prep_kernel": "bwd_delta", "dq_convert_kernel": "bwd_dq_cvt"}


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for k, v in SHORT.items():
                if "fa::" + k + "<" in name:
                    acc[v].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    out, fetch_csv, write_csv = sys.argv[1:4]
    fetch, nfetch = mean_by_kernel(fetch_csv, "FETCH_SIZE")
    write, _ = mean_by_kernel(write_csv, "WRITE_SIZE")
    steps = max(1, nfetch.get("fwd_mfma", 1))   # one forward launch per step
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fb = 2.0 * fetch.get(k, 0.0) * 1024.0
        wb = write.get(k, 0.0) * 1024.0
        res[k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "launches_per_step": nfetch.get(k, steps) / steps,
                  "note": "FETCH_SIZE KiB x2 (gfx950 half-count of wide coalesced reads) + WRITE_SIZE KiB"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
