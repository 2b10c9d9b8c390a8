#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: launches, mean duration, mean counter value.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so ``fetch_bytes_corrected`` doubles it.
"""
import collections
import csv
import json
import sys


def short(name):
    for key in ("pw_gemm_split_kernel", "pw_gemm_bf16x3_kernel", "pw_gemm_kernel", "dw_conv_kernel", "dw_pair_kernel", "dw_toeplitz_kernel",
                "dwpw_fused_kernel"):
        if key + "<" in name:
            return key + "<" + name.split(key + "<")[1].split(">")[0] + ">"
    for key in ("dw_conv_generic", "stft_logmel", "normalize_kernel", "logsoftmax_argmax", "ctc_collapse",
                "len_chain", "seq_len", "repad", "beam"):
        if key in name:
            return key
    return name[:60]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[(p, k)]:
                seen[(p, k)].add(r["Dispatch_Id"])
                agg[k]["_dur_ns:" + p] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                agg[k]["_n:" + p] += 1
    out = {}
    for k, c in sorted(agg.items()):
        ns = {kk[3:]: v for kk, v in c.items() if kk.startswith("_n:")}
        n = max(ns.values())
        d = {"launches": int(n)}
        for kk, v in c.items():
            if kk.startswith("_dur_ns:"):
                d.setdefault("avg_us", round(v / ns[kk[8:]] / 1e3, 2))
            elif not kk.startswith("_"):
                d[kk + "_per_launch"] = v / n
        if "FETCH_SIZE_per_launch" in d:
            d["fetch_bytes_corrected_per_launch"] = d["FETCH_SIZE_per_launch"] * 1024 * 2
        if "WRITE_SIZE_per_launch" in d:
            d["write_bytes_per_launch"] = d["WRITE_SIZE_per_launch"] * 1024
        out[k] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
