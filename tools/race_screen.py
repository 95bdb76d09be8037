#!/usr/bin/env python3
"""Race screen for the phased GEMMs (LDS-DMA ordered against fragment reads only by counted waits + barriers): many
repeated runs at several batch sizes, in both precisions and both schedules, must return the bits of the tile kernels
(option gemm_tile=1; the switches belong to a context, and dce_last_plan is recorded to prove which kernels each context ran), also while a second stream keeps the memory system busy (uneven load shifts LDS-DMA landing times)."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(precision, sched, sizes, reps):
    import torch
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    out, plans = {}, {}
    noise_src = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    noise_dst = torch.empty_like(noise_src)
    side = torch.cuda.Stream()
    for n in sizes:
        x = torch.randn((n, 150, 54), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
        ref_m = contact_cnn(device=0, max_batch=n, precision=precision, tune={"gemm_tile": 1}); ref_m.load_state_dict(sd)
        ref = ref_m.predict(x)["logits"].clone(); ref_plan = ref_m.last_plan(); ref_m.close()
        m = contact_cnn(device=0, max_batch=n, precision=precision, tune={"gemm_lockstep": int(sched == "lockstep")}); m.load_state_dict(sd)
        bad = 0
        for r in range(reps):
            if r % 2:                                   # every other run under a concurrent copy stream
                with torch.cuda.stream(side):
                    for _ in range(4):
                        noise_dst.copy_(noise_src, non_blocking=True)
            got = m.predict(x)["logits"]
            bad += int(not torch.equal(got, ref))
        torch.cuda.synchronize()
        plan = m.last_plan()
        assert plan != ref_plan and not any("tile" not in k and k.startswith("fc_") and "chain" not in k for k in ref_plan if "phased" in k or "lockstep" in k), (plan, ref_plan)
        m.close()
        out[n] = bad
        plans[n] = {"reference": ref_plan, "screened": plan}
    print(json.dumps({"precision": precision, "schedule": sched, "reps": reps, "mismatching_runs": out, "kernels": plans}))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2], [int(v) for v in sys.argv[3].split(",")], int(sys.argv[4]))
    else:
        t0 = time.time()
        for precision in ("fp32", "bf16_fc"):
            for sched in ("phased", "lockstep"):
                subprocess.run([sys.executable, __file__, precision, sched, "3072,4096,5003,12288,32768", "150"], check=True)
        print(f"done in {time.time() - t0:.0f} s")
