"""GPU: BASELINE.json configs[0] -- the reference's entry scripts' plumbing on the mirrored CLI:
unchanged YAML keys, .npy inputs, a checkpoint in the reference's torch schema, 16-class output."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entry_scripts_on_synthetic_config(tmp_path):
    from deep_contact_estimator_amd import synth
    from oracle import oracle as orc
    out = tmp_path / "synthetic_data"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_data.py"), "--out", str(out),
                    "--T", str(150 + 299)], check=True, capture_output=True)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "inference_one_seq_params.yaml")))
    for k in ("data_path", "label_path", "mat_data_path", "model_load_path", "mat_save_path", "lcm_save_path"):
        cfg[k] = cfg[k].replace("synthetic_data", str(out))
    cfg["save_mat"] = True
    cfg_path = tmp_path / "inference_one_seq_params.yaml"
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    for extra in ([], ["--fused"], ["--fused", "--precision", "fp32_split"], ["--precision", "bf16_fc"]):
        r = subprocess.run([sys.executable, "-m", "deep_contact_estimator_amd.inference_one_seq",
                            "--config_name", str(cfg_path), *extra], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        contacts = np.load(os.path.splitext(cfg["mat_save_path"])[0] + ".npy")
        ref = orc.Oracle(synth.make_state_dict(1, "uniform")).infer_sequence(
            synth.make_sequence(150 + 299, 0).astype(np.float32))
        assert contacts.shape == (300, 4) and contacts.dtype == np.uint8
        if "--precision" not in extra:
            assert np.array_equal(contacts, ref["contacts"])
        else:       # the other precisions of the library (--precision; a `precision` key in the YAML does the same): same states
            srt = np.sort(ref["logits"], axis=1)      # wherever the reference's decision is not a toss-up
            clear = (srt[:, -1] - srt[:, -2]) > (1e-3 if extra[-1] == "fp32_split" else 5e-2) * np.abs(ref["logits"]).max()
            assert clear.mean() > 0.5 and np.array_equal(contacts[clear], ref["contacts"][clear])
    tcfg = yaml.safe_load(open(os.path.join(ROOT, "config", "test_params.yaml")))
    tcfg["data_folder"] = str(out) + "/"
    tcfg["model_load_path"] = cfg["model_load_path"]
    tpath = tmp_path / "test_params.yaml"
    yaml.safe_dump(tcfg, open(tpath, "w"))
    r = subprocess.run([sys.executable, "-m", "deep_contact_estimator_amd.test", "--config_name", str(tpath)],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert "Test accuracy in terms of class is:" in r.stdout and "jaccard of class is:" in r.stdout
    i0 = next(k for k, l in enumerate(lines) if l.startswith("Test accuracy in terms of class is:"))
    assert len(lines) - i0 == 78 + 6                       # the reference's report: 46 labelled lines (6 matrices of 2 rows), 32 raw
    # the reference's shipped batch_size 30 in the library's latency mode (round 6: one kernel per batch, csrc/latency_mb.hip): the same report
    # (the loop's contact states are the reference's wherever its decision is not a toss-up; the synthetic set has none within fp32 noise)
    rl = subprocess.run([sys.executable, "-m", "deep_contact_estimator_amd.test", "--config_name", str(tpath), "--latency"],
                        env=env, capture_output=True, text=True)
    assert rl.returncode == 0, rl.stderr[-2000:]
    assert tcfg["batch_size"] == 30 and rl.stdout.splitlines()[i0:] == lines[i0:], "the latency mode's report differs from the batch path's"


@pytest.mark.parametrize("nproc", [2, 8])
def test_entry_scripts_sharded_over_two_processes(nproc, tmp_path):
    """The multi-GPU launch of the entry scripts (one process per rank under torch.distributed.run),
    on ONE GPU: two ranks share it and exchange over gloo (RCCL refuses two ranks per device), which
    exercises everything but the transport -- halo sharding, per-rank fused pass, gather of the (N,4)
    estimates to rank 0, all-reduce of the 16x16 counts -- against the single-process run."""
    from deep_contact_estimator_amd import synth
    from oracle import oracle as orc
    out = tmp_path / "synthetic_data"
    T = 150 + 300                                        # 301 windows: uneven split 151 / 150
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_data.py"), "--out", str(out),
                    "--T", str(T)], check=True, capture_output=True)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "inference_one_seq_params.yaml")))
    for k in ("data_path", "label_path", "mat_data_path", "model_load_path", "mat_save_path", "lcm_save_path"):
        cfg[k] = cfg[k].replace("synthetic_data", str(out))
    cfg["save_mat"] = True
    cfg["calculate_accuracy"] = True
    cfg_path = tmp_path / "inference_one_seq_params.yaml"
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    tcfg = yaml.safe_load(open(os.path.join(ROOT, "config", "test_params.yaml")))
    tcfg["data_folder"] = str(out) + "/"
    tcfg["model_load_path"] = cfg["model_load_path"]
    tpath = tmp_path / "test_params.yaml"
    yaml.safe_dump(tcfg, open(tpath, "w"))
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_DIST_BACKEND="gloo")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
              "--master-addr", "127.0.0.1", "--master-port", str(29571 + nproc)]      # (8: the ranks of an 8-GPU node, here sharing the one GPU over gloo; 301 windows = 38 / 37 per rank)

    r = subprocess.run(launch + ["-m", "deep_contact_estimator_amd.inference_one_seq", "--config_name", str(cfg_path)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    contacts = np.load(os.path.splitext(cfg["mat_save_path"])[0] + ".npy")
    ref = orc.Oracle(synth.make_state_dict(1, "uniform")).infer_sequence(synth.make_sequence(T, 0).astype(np.float32))
    assert contacts.shape == (T - 149, 4) and np.array_equal(contacts, ref["contacts"])
    single = subprocess.run([sys.executable, "-m", "deep_contact_estimator_amd.inference_one_seq", "--config_name", str(cfg_path)],
                            env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    acc = [l for l in r.stdout.splitlines() if l.startswith("Accuracy")]
    assert len(acc) == 6 and acc == [l for l in single.stdout.splitlines() if l.startswith("Accuracy")]

    r = subprocess.run(launch + ["-m", "deep_contact_estimator_amd.test", "--config_name", str(tpath)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    single = subprocess.run([sys.executable, "-m", "deep_contact_estimator_amd.test", "--config_name", str(tpath)],
                            env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
    keep = ("Test accuracy", "Accuracy", "Precision of", "jaccard of", "false ", "AVG false", "confusion matrix")
    pick = lambda txt: [l for l in txt.splitlines() if l.startswith(keep)]
    assert pick(r.stdout) == pick(single.stdout) and len(pick(r.stdout)) == 6 + 6 + 6 + 6 + 10   # the reference's labelled lines


def test_bench_two_rank_flow():
    """bench.py's N>1 flow (rank env from torch.distributed.run, per-step async gather, barrier +
    max-over-ranks timing, one JSON line from rank 0) with two ranks sharing the one GPU over gloo."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29573", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"]
    sh = j["extra"]["sharded_1e6"]                       # BASELINE configs[3] literally, measured in the same run
    assert sh["windows_per_s_incl_gather"] > 0 and abs(sh["gathered_MB"] - 2 * 1_000_000 * 68 / 1e6) < 1e-9
    assert [p["rank"] for p in sh["per_rank"]] == [0, 1] and sum(p["windows"] for p in sh["per_rank"]) == 2_000_000
    assert abs(j["value"] - 2 * 4096 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6


def test_bench_eight_rank_flow_rehearsal():
    """The driver's `bench.py --gpus 8` command line, rehearsed with the eight ranks sharing this box's one GPU over gloo
    (RCCL refuses two ranks on one device): eight per-rank records, the weak-scaling arithmetic of the line, and configs[3]'s
    sharded pass on 1,000,003 windows -- ragged shards of 125,001 / 125,000 -- gathered to rank 0."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_DIST_BACKEND="gloo", DCE_SHARDED_TOTAL="1000003", DCE_EXTRA_TIMEOUT="600")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29591", os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--steps", "3", "--warmup", "1", "--settle-s", "0.2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["global_batch"] == 8 * j["config"]["batch_per_gpu"]
    assert abs(j["value"] - 8 * 4096 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-6
    pr = j["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and 0 <= pr["rank_of_max"] < 8 and abs(pr["ms_per_step_max"] - j["ms_per_step"]) / j["ms_per_step"] < 0.05
    sh = j["extra"]["sharded_1e6"]
    assert sh["windows"] == 1000003 and sh["windows_per_s_incl_gather"] > 0 and abs(sh["gathered_MB"] - 1000003 * 68 / 1e6) < 1e-9, sh
    assert not j.get("partial")


def test_bench_self_launch_from_the_plain_command_line():
    """`python bench.py --gpus 2 --steps 5` started PLAINLY (no torch.distributed.run, no WORLD_SIZE): bench.py re-executes
    itself as two ranks on a free port and rank 0 prints the one JSON line with n_gpus == 2 -- the command the driver will
    run with an 8 on an 8-GPU node.  Here the two ranks share the one GPU, so the transport is gloo (RCCL refuses two ranks
    per device); without DCE_DIST_BACKEND=gloo the same command must refuse loudly instead of hanging."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(PYTHONPATH=ROOT, DCE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"] and "rccl" in j
    assert j["extra"]["sharded_1e6"]["windows_per_s_incl_gather"] > 0
    del env["DCE_DIST_BACKEND"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "RCCL takes one rank per device" in (r.stdout + r.stderr)


def test_config3_full_size_shards_equal_single_call():
    """BASELINE configs[3] at its full size (8e6 windows, 1.73 GB sequence): the 8 halo-sharded ranges
    of a node, computed one after the other on this GPU, equal the single call bit for bit, and the
    size-independent properties hold (tools/check_sharding_8e6.py)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sharding_8e6.py")],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("shards_equal_single_call_bitwise", "deterministic", "pred_is_argmax_of_logits",
              "contacts_are_bits_of_pred", "finite_logits"):
        assert j[k] is True, (k, j)
    assert j["classes_seen"] >= 4


def test_rccl_single_rank_degenerate(tmp_path):
    """RCCL itself (backend "nccl"), in the only form one GPU allows: a world of ONE rank.  Exercises what the
    8-GPU run will execute first -- init_process_group("nccl", device_id=...), the packed gather of
    infer_sequence_sharded, bench.py's AsyncRowGather (async gather handles on device tensors), the all-reduce of
    the confusion counts, barrier, destroy -- against the single-process results."""
    code = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DCE_ROOT"])
from deep_contact_estimator_amd import contact_cnn, synth
from deep_contact_estimator_amd.distributed import init_from_env, infer_sequence_sharded, AsyncRowGather, confusion_sharded
rank, world, local = init_from_env()
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
m = contact_cnn(device=local, max_batch=512); m.load_state_dict(synth.make_state_dict(1, "uniform"))
seq = torch.from_numpy(synth.make_sequence(150 + 700, 9).astype(np.float32)).to(dev)
one = m.infer_sequence(seq)
got = infer_sequence_sharded(m.infer_sequence, seq, dst=0)
ok = all(torch.equal(got[k], one[k]) for k in ("logits", "pred", "contacts"))
g = AsyncRowGather(701, 16, torch.float32, dev, dst=0, depth=2)
for _ in range(5):
    g.submit(one["logits"])
g.drain()
ok = ok and torch.equal(g.latest()[0], one["logits"])
labels = torch.from_numpy(synth.make_labels(150 + 700, 9).reshape(-1)).to(dev)
C = confusion_sharded(m.infer_sequence, m.confusion_counts, seq, labels)
ok = ok and int(C.sum().item()) == 701
# ---- the product's exchange: RCCL issued by libdce.so itself (dce_comm_init / dce_gather_results / dce_allreduce_counts)
from deep_contact_estimator_amd.distributed import comm_bootstrap, PackedStepGather
comm_bootstrap(m, rank, world)
info = m.comm_info()
ok = ok and info["world"] == 1 and info["rank"] == 0 and info["rccl_version"] > 0 and "rccl" in info["library"]
got2 = infer_sequence_sharded(m.infer_sequence, seq, dst=0, model=m)            # packed rows, ONE ncclGather (uniform sizes)
ok = ok and all(torch.equal(got2[k], one[k]) for k in ("logits", "pred", "contacts"))
packed = m.infer_sequence_packed(seq)
rag = m.gather_results(packed, [701], root=0)                                    # explicit sizes -> same result
ok = ok and torch.equal(rag, packed)
C2 = confusion_sharded(m.infer_sequence, m.confusion_counts, seq, labels, model=m)   # ncclAllReduce int64
ok = ok and torch.equal(C2.cpu(), C.cpu())
win = m.zscore_windows(seq, 0, 64)
psg = PackedStepGather(m, 64, dev, dst=0)
for _ in range(5):                                                               # two async gathers in flight, buffers alternate
    psg.step(win)
psg.drain()
ok = ok and torch.equal(psg.latest(), m.predict_packed(win)) and psg.bytes_per_step == 64 * 68
m.comm_sync(); m.comm_destroy()
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); dist.destroy_process_group()
print(json.dumps({"ok": bool(ok), "max": float(t.item())}))
'''
    script = tmp_path / "rccl1.py"
    script.write_text(code)
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_ROOT=ROOT, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    env.update(WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    # init_from_env only joins a group when WORLD_SIZE > 1; force the one-rank group through the same call
    env["DCE_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    import json
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["ok"] is True and j["max"] == 1.5


def test_bench_multi_gpu_flow_on_rccl_single_rank():
    """bench.py's N>1 flow on real RCCL in a world of one rank (DCE_FORCE_DIST=1): nccl process group with
    device_id, the asynchronous per-step gather of the logits, barrier + all-reduce(MAX) timing, and
    extra.sharded_1e6 (configs[3]: 1e6 windows, the one packed gather) -- the code the driver's --gpus 8 run executes."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_FORCE_DIST="1", MASTER_PORT="29583")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--settle-s", "0.2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and "RCCL gather" in j["config"]["sharding"] and j["value"] > 1e6
    rc = j["rccl"]                                         # the per-step exchange went through libdce.so's own communicator
    assert rc["world_size_ncclCommCount"] == 1 and rc["gathered_bytes_per_step"] == 4096 * 68 and "dce_gather_results" in rc["backend"]
    sh = j["extra"]["sharded_1e6"]
    assert sh["windows_per_s_incl_gather"] > 1e6 and abs(sh["gathered_MB"] - 68.0) < 1e-9 and "dce_gather_results" in sh["transport"]
    pr = sh["per_rank"]                                    # the record explains itself: every rank's compute and gather phase (round 5's review, item 8)
    assert len(pr) == 1 and pr[0]["windows"] == 1_000_000 and pr[0]["sent_bytes"] == 68_000_000 and 100 < pr[0]["compute_ms"] < 2000 and 0 < pr[0]["gather_ms"] < 500
    assert pr[0]["compute_ms"] + pr[0]["gather_ms"] <= sh["ms"] * 1.05


def test_bench_device_map_and_per_rank_record():
    """A launcher whose LOCAL_RANK is not a device index (DCE_DEVICE_MAP) and a permuted HIP_VISIBLE_DEVICES: the rank lands on
    the mapped visible device, and the N>1 line carries every rank's own ms_per_step, the slow rank and the exchange's own time."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_FORCE_DIST="1", MASTER_PORT="29587", LOCAL_RANK="2", DCE_DEVICE_MAP="7,7,0",
               HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0").split(",")[-1])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--settle-s", "0.2",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    pr = j["per_rank"]
    assert len(pr["ms_per_step"]) == 1 and pr["rank_of_max"] == 0 and pr["ms_per_step_min"] == pr["ms_per_step_max"] > 0
    assert abs(j["ms_per_step"] - pr["ms_per_step_max"]) < 1e-6 * j["ms_per_step"] + 1e-9
    assert 0 < j["rccl"]["gather_alone_us"] < 5000


def test_bench_reports_a_fallback_when_rccl_cannot_be_bound():
    """If libdce.so cannot bind RCCL (DCE_RCCL_LIB=none stands in for a box without a usable librccl) the ranks agree on
    it before any collective of the data path and the per-step gather runs over torch.distributed -- and the line SAYS so
    (`rccl.backend` starts with FALLBACK); nothing hangs, nothing is silent."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_FORCE_DIST="1", MASTER_PORT="29585", DCE_RCCL_LIB="none")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--settle-s", "0.2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["rccl"]["backend"].startswith("FALLBACK") and "DCE_RCCL_LIB=none" in j["rccl"]["backend"]
    assert j["value"] > 1e6 and "torch.distributed" in j["extra"]["sharded_1e6"]["transport"]


def test_bench_falls_back_when_the_first_exchange_fails():
    """The library's communicator is brought up under a watchdog (comm_bootstrap_checked: init + a first gather whose
    bytes rank 0 checks).  When that fails on a rank (DCE_COMM_SELFTEST=fail stands in for it) the ranks agree on the
    verdict, the communicator is dropped and the exchange runs over torch.distributed -- the line says FALLBACK and why,
    and the measurement itself is unharmed."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, DCE_FORCE_DIST="1", MASTER_PORT="29586", DCE_COMM_SELFTEST="fail")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--settle-s", "0.2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["rccl"]["backend"].startswith("FALLBACK") and "DCE_COMM_SELFTEST=fail" in j["rccl"]["backend"]
    assert j["value"] > 1e6 and "torch.distributed" in j["extra"]["sharded_1e6"]["transport"]


def test_bench_default_line_contract():
    """The line the driver records: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the
    contract's keys, the roofline and cpu_baseline objects, and every other BASELINE config measured under `extra`."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2"],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "windows/s" and j["dtype"] == "f32" and j["vs_baseline"] is None and "workload" in j["config"]
    assert abs(j["value"] - 4096 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-6 and j["value"] > 1e6
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0.5 < rf["frac"] < 1.0 and "traffic" in rf and "traffic_stale" in rf
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["sample"]
    ex = j["extra"]
    assert ex["streaming_1e6"]["hbm_resident_windows_per_s"] > 1e6 and ex["bf16_fc"]["windows_per_s"] > 1e6
    assert 10 < ex["online_push"]["us_per_push"] < 500 and ex["online_push"]["pushes"] >= 1000
    fs = ex["fp32_f16x2"]                                    # the opt-in precision: fp32-tolerance results, so (almost) no argmax change
    assert fs["windows_per_s"] > 1e6 and fs["vs_fp32_same_input"]["argmax_flips"] <= 2
    assert fs["vs_fp32_same_input"]["max_abs_dlogit"] < 1e-4 * fs["vs_fp32_same_input"]["max_abs_logit"]
    assert "fp16 MFMA" in fs["kernels"]["fc1_gemm"]["pipe"] and "fp16 MFMA" in fs["kernels"]["conv_stack"]["pipe"]
    assert "fp32_split" not in ex                             # (round 6: retired from the product library; the alias is tested in tests/test_round6_gpu.py)
    sb = ex["small_batches"]["batches"]
    assert set(sb) >= {"1", "30"} and 0 < sb["1"]["us_per_call"] < sb["30"]["us_per_call"] < 500
    lm = ex["latency_mode"]                                  # the reference's shipped batch sizes 1 and 30 in the latency mode
    assert lm["batches"]["per_batch"]["30"]["plan"] == ["latency_mb"] and lm["batches"]["per_batch"]["30"]["us_per_call"] < sb["30"]["us_per_call"]
    assert lm["cold_weights_30"]["cold"]["fc0_weight_stream_TBs"] > 2.0
    # every roofline fraction of the line is a fraction: a stage priced against the wrong pipe (round 5: fp32_f16x2's fc.3, an fp16-pipe kernel,
    # against the fp32 peak -> 1.65) must not come back.  bench.py checks itself; the walk below checks bench.py.
    assert j["self_check"]["roofline_fractions_above_one"] == [], j["self_check"]

    def fractions(o, path="line"):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ("frac", "path_frac_of_roof", "frac_of_roof") and isinstance(v, (int, float)):
                    yield f"{path}.{k}", v
                yield from fractions(v, f"{path}.{k}")
    fr = dict(fractions(j))
    assert len(fr) >= 12 and all(0.0 < v <= 1.0 for v in fr.values()), {k: v for k, v in fr.items() if not 0.0 < v <= 1.0}
    fx = ex["fp32_f16x2"]                                    # fc.3 of this precision runs on the fp16 pipe, three MFMAs per product
    assert "fp16 MFMA" in fx["kernels"]["fc2_gemm"]["pipe"] and fx["kernels"]["fc2_gemm"]["kernel_family"].startswith("fc23_fused_h2"), fx["kernels"]["fc2_gemm"]
    assert 0.3 < fx["path_frac_of_roof"] < 0.7
