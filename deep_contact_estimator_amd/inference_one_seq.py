"""Entry script mirroring the reference's src/inference_one_seq.py:137-179.

    python -m deep_contact_estimator_amd.inference_one_seq --config_name config/inference_one_seq_params.yaml

Reads the same YAML keys (data_path, label_path, model_load_path, window_size, batch_size,
calculate_accuracy, save_mat, mat_save_path, save_lcm, lcm_save_path), builds
contact_dataset / loader / contact_cnn the same way and prints the same accuracy lines.
``--fused`` runs the whole sequence through one dce_infer_sequence call instead of the
per-batch loop (identical results).  Under ``python -m torch.distributed.run --nproc-per-node G``
the windows are sharded over the G GPUs (one fused pass each, 149-row halo) and the (N,4) estimates
gathered to rank 0 over RCCL, which alone writes the outputs.  save_mat / save_lcm
call export.save2mat / export.save2lcm (restated formats, see export.py) when the raw .mat named by
mat_data_path exists; synthetic runs without one get the (N,4) estimates as .npy instead.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import yaml

from .contact_cnn import contact_cnn, load_checkpoint
from .data_handler import contact_dataset, WindowLoader
from .inference import inference, inference_and_compute_acc, inference_sequence


def main(argv=None):
    import torch
    if not torch.cuda.is_available():
        sys.exit("deep_contact_estimator_amd needs an MI355X (no CPU path)")
    from .distributed import init_from_env, infer_sequence_sharded, confusion_sharded
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    if rank == 0:
        print("Using ", device, "" if world == 1 else f"(+{world - 1} more ranks)")

    parser = argparse.ArgumentParser(description="Run the contact network on one sequence")
    parser.add_argument("--config_name", type=str,
                        default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "config",
                                             "inference_one_seq_params.yaml"))
    parser.add_argument("--fused", action="store_true", help="one fused pass instead of the batch loop")
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

    dataset = contact_dataset(data_path=config["data_path"], label_path=config["label_path"],
                              window_size=config["window_size"], device=device)
    dataloader = WindowLoader(dataset, batch_size=config["batch_size"])

    model = contact_cnn(device=local, max_batch=max(int(config["batch_size"]), 32768),
                        precision=args.precision or config.get("precision", "fp32"), tune=tune)
    model.load_state_dict(load_checkpoint(config["model_load_path"]))
    model = model.eval().to(device)

    if world > 1:
        import torch.distributed as dist
        from . import metrics
        from .distributed import comm_bootstrap
        if dist.get_backend() == "nccl":      # the exchanges below are then issued by libdce.so itself (dce_gather_results)
            comm_bootstrap(model, rank, world)
        if config["calculate_accuracy"]:
            C = confusion_sharded(model.infer_sequence, model.confusion_counts, dataset.data, dataset.label, model=model)
            mt = metrics.metrics_from_confusion16(C.cpu().numpy())
            if rank == 0:
                print("Accuracy in terms of class: %.4f" % mt["acc"])
                for leg in range(4):
                    print("Accuracy of leg %d is: %.4f" % (leg, mt["acc_per_leg"][leg]))
                print("Accuracy is: %.4f" % (np.sum(mt["acc_per_leg"]) / 4.0))
        res = infer_sequence_sharded(model.infer_sequence, dataset.data, dst=0, model=model)
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return None
        pred = res["contacts"]
    elif config["calculate_accuracy"]:
        pred, acc, acc_per_leg = inference_and_compute_acc(dataloader, model, device)
        print("Accuracy in terms of class: %.4f" % acc)
        for leg in range(4):
            print("Accuracy of leg %d is: %.4f" % (leg, acc_per_leg[leg]))
        print("Accuracy is: %.4f" % (np.sum(acc_per_leg) / 4.0))
    elif args.fused:
        pred = inference_sequence(dataset, model)
    else:
        pred = inference(dataloader, model, device)

    from . import export
    have_mat = os.path.exists(str(config.get("mat_data_path", "")))
    if config.get("save_mat"):
        if have_mat:
            export.save2mat(pred, config)                      # src/inference_one_seq.py:64-89
        else:                                                  # synthetic runs have no raw .mat
            out = os.path.splitext(config["mat_save_path"])[0] + ".npy"
            np.save(out, pred.cpu().numpy())
            print("mat_data_path not found: saved the (N,4) contact estimates to", out)
    if config.get("save_lcm"):
        if have_mat:
            export.save2lcm(pred, config)                      # src/inference_one_seq.py:91-133
        else:
            print("save_lcm requested but mat_data_path not found: skipped")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return pred


if __name__ == "__main__":
    main()
