"""Entry script mirroring the hot-path part of the reference's src/test.py:113-147.

    python -m deep_contact_estimator_amd.test --config_name config/test_params.yaml

Same YAML keys (data_folder, model_load_path, window_size, batch_size); runs compute_accuracy
over <data_folder>/test.npy + test_label.npy and prints the accuracy block.  Launched as
``python -m torch.distributed.run --nproc-per-node G -m deep_contact_estimator_amd.test ...`` the
windows are sharded over the G GPUs (one process each) and the counts summed with one all-reduce.  The
precision / Jaccard / confusion-matrix numbers the reference gets from scikit-learn
(src/test.py:19-70,137-139) are closed forms of one 16x16 integer matrix accumulated on the
device (dce_confusion_counts + metrics.py); the float64 arrays are still returned unchanged so the
reference's own functions can consume them.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import yaml

from .contact_cnn import contact_cnn, load_checkpoint
from .data_handler import contact_dataset, WindowLoader
from .inference import compute_accuracy


def main(argv=None):
    import torch
    if not torch.cuda.is_available():
        sys.exit("deep_contact_estimator_amd needs an MI355X (no CPU path)")
    from .distributed import init_from_env, confusion_sharded
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    if rank == 0:
        print("Using ", device, "" if world == 1 else f"(+{world - 1} more ranks)")
    parser = argparse.ArgumentParser(description="Test the contact network")
    parser.add_argument("--config_name", type=str,
                        default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "config",
                                             "test_params.yaml"))
    parser.add_argument("--precision", default=None, choices=["fp32", "bf16_fc", "fp32_split", "fp32_f16x2"],
                        help="arithmetic of the library (default fp32 = the reference's; the YAML may carry a `precision` key too): "
                             "bf16_fc = bf16 operands on fc.0 / fc.3; fp32_split = fp32 results on the bf16 matrix pipe (DESIGN.md 4.1x / 4.2x); "
                             "fp32_f16x2 = the fp32 tolerance from two fp16 terms per operand with per-window scales (DESIGN.md 4.6)")
    parser.add_argument("--latency", action="store_true",
                        help="the library's latency mode (fp32; option latency=1, a `latency: true` key in the YAML does the same): every batch of up to 32 windows "
                             "-- the reference ships batch_size 1 and 30 -- is ONE kernel (DESIGN.md 4.5); needs the whole device")
    args = parser.parse_args(argv)
    config = yaml.safe_load(open(args.config_name))
    tune = {"latency": 1} if (args.latency or config.get("latency")) else None

    test_data = contact_dataset(data_path=config["data_folder"] + "test.npy",
                                label_path=config["data_folder"] + "test_label.npy",
                                window_size=config["window_size"], device=device)
    test_dataloader = WindowLoader(test_data, batch_size=config["batch_size"])
    model = contact_cnn(device=local, max_batch=max(int(config["batch_size"]), 32768),
                        precision=args.precision or config.get("precision", "fp32"), tune=tune)
    model.load_state_dict(load_checkpoint(config["model_load_path"]))
    model = model.eval().to(device)

    from . import metrics
    if world > 1:
        # sharded evaluation: each rank one fused pass over its windows, one 2 KB all-reduce
        import torch.distributed as dist
        from .distributed import comm_bootstrap
        if dist.get_backend() == "nccl":      # the count all-reduce is then issued by libdce.so (dce_allreduce_counts)
            comm_bootstrap(model, rank, world)
        C = confusion_sharded(lambda rows: model.infer_sequence(rows), model.confusion_counts,
                              test_data.data, test_data.label, model=model)
        mt = metrics.metrics_from_confusion16(C.cpu().numpy())
        if rank == 0:
            print("\n".join(metrics.report_lines(mt)))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        return mt

    test_acc, acc_per_leg, bin_pred_arr, bin_gt_arr, pred_arr, gt_arr = compute_accuracy(test_dataloader, model)
    # precision / Jaccard / confusion matrices / FN / FP rates (src/test.py:137-139) are closed forms of ONE
    # 16x16 integer matrix -- built from the class ids the loop above already collected (no second pass,
    # no scikit-learn); the multi-GPU path accumulates the same matrix on the devices (dce_confusion_counts)
    mt = metrics.metrics_from_confusion16(metrics.confusion16(pred_arr, gt_arr))
    mt["acc"], mt["acc_per_leg"] = test_acc, acc_per_leg           # the loop's own numbers head the report
    print("\n".join(metrics.report_lines(mt)))
    return test_acc, acc_per_leg, bin_pred_arr, bin_gt_arr, pred_arr, gt_arr


if __name__ == "__main__":
    main()
