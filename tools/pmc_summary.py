#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel.
Usage: pmc_summary.py <counter_collection.csv> [...]"""
import csv
import collections
import sys


def short(name):
    for key, s in (("fc_gemm_phased_kernel<false, false, 2, 2", "fc1_gemm(128x128)"), ("fc_gemm_phased_kernel<false, false, 1, 1", "fc2_gemm(64x64)"),
                   ("fc_gemm_phased_kernel<true, true, 2, 2", "fc1_gemm_bf16"), ("fc_gemm_phased_kernel<true, false, 1, 1", "fc2_gemm_bf16"),
                   ("fc6_combine", "fc3_tail"), ("conv_h2_kernel", "conv_h2"), ("fc_gemm_h2k_kernel", "fc1_gemm_h2k"), ("fc_gemm_h2_kernel", "fc1_gemm_h2"), ("conv_x3_kernel", "conv_x3"), ("fc_gemm_x3_kernel", "fc1_gemm_x3"), ("conv_wino_kernel", "conv_stack"), ("conv_stack_kernel", "conv_stack_direct"), ("fc_gemm_kernelILi2ELi2", "fc1_gemm(128x128)"),
                   ("fc_gemm_kernelILi1ELi1", "fc2_gemm(64x64)"), ("fc_gemm_kernel<2, 2, false", "fc1_gemm(128x128)"), ("fc_gemm_kernel<2, 2, true", "fc1_gemm_bf16"),
                   ("fc_gemm_kernel<1, 1, false", "fc2_gemm(64x64)"), ("fc_gemm_kernel<1, 1, true", "fc2_gemm_bf16"), ("fc_gemm_small_kernel", "fc2_gemm(64x64)"), ("fc3_tail", "fc3_tail"),
                   ("zscore_windows", "zscore_windows"), ("fc_fused", "fc_fused")):
        if key in name:
            return s
    return name[:40]


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for row in csv.DictReader(open(p)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, ctrs in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in ctrs.items()}
        print(k, {c: f"{x:.4g}" for c, x in out[k].items()})
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
