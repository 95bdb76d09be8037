"""The C ABI from plain C: tests/c/abi_client.c is compiled with gcc against include/dce.h and
libdce.so (no Python, no torch in that process) with the CPU oracle linked in as the checker.
CPU: it must compile and link against every symbol it uses.  GPU: it must run green."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep_contact_estimator_amd")
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")


def _build_client(out):
    if not os.path.exists(os.path.join(PKG, "libdce.so")):
        from deep_contact_estimator_amd import build
        build.build()
    cmd = ["gcc", "-O2", "-std=c11", "-fopenmp", "-Wall", "-Werror",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tests", "c", "abi_client.c"), os.path.join(ROOT, "oracle", "dce_oracle.c"),
           "-L" + PKG, "-ldce", "-L" + ROCM_LIB, "-lamdhip64", "-lm",
           "-Wl,-rpath," + PKG, "-Wl,-rpath," + ROCM_LIB, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_client_compiles_and_links(tmp_path):
    exe = _build_client(str(tmp_path / "abi_client"))
    assert os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = _build_client(str(tmp_path / "abi_client"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_client: OK" in r.stdout
    print(r.stdout.strip())


def _build_ranks(out):
    if not os.path.exists(os.path.join(PKG, "libdce.so")):
        from deep_contact_estimator_amd import build
        build.build()
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_ranks.c"), "-L" + PKG, "-ldce", "-L" + ROCM_LIB, "-lamdhip64", "-lm",
           "-Wl,-rpath," + PKG, "-Wl,-rpath," + ROCM_LIB, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_n_rank_client_rendezvous_without_a_gpu(tmp_path):
    """SURVEY.md 8(e)'s torch-free bootstrap in its N-rank form (tests/c/abi_ranks.c + tools/launch_ranks.sh): three processes
    meet through the id file although a stale file of another job sits at the same path; a rank with another job's nonce is
    NOT fooled by the file and gives up with an error instead of taking a wrong id (--dry-run: no GPU, no RCCL)."""
    exe = _build_ranks(str(tmp_path / "abi_ranks"))
    idf = tmp_path / "id"
    idf.write_bytes(b"s" * 144)                                  # left behind by an earlier job
    env = dict(os.environ, DCE_COMM_ID_FILE=str(idf), DCE_COMM_TIMEOUT="20")
    r = subprocess.run([os.path.join(ROOT, "tools", "launch_ranks.sh"), "3", exe, "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK (dry run") == 3 and "all 3 ranks OK" in r.stdout
    env = dict(os.environ, DCE_COMM_TIMEOUT="1")
    r = subprocess.run([exe, "--rank", "1", "--world", "2", "--id-file", str(idf), "--nonce", "another-job", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no id with this job's nonce" in r.stderr


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_eight_rank_rendezvous_rehearsal_without_a_gpu(tmp_path):
    """The first contact with an 8-GPU node, rehearsed as far as a box without one allows: EIGHT torch-free ranks (tests/c/abi_ranks.c
    --dry-run under tools/launch_ranks.sh) meet through the id file with a stale file of another job in place and one rank 1.5 s
    late; every rank derives its shard of 1,000,003 windows by itself and the eight shards tile the range with sizes one apart; a ninth
    process with another job's nonce is refused."""
    import re
    exe = _build_ranks(str(tmp_path / "abi_ranks"))
    idf = tmp_path / "id"
    idf.write_bytes(b"s" * 144)                                  # left behind by an earlier job
    env = dict(os.environ, DCE_COMM_ID_FILE=str(idf), DCE_COMM_TIMEOUT="30", DCE_LAUNCH_LATE="5:1500")
    r = subprocess.run([os.path.join(ROOT, "tools", "launch_ranks.sh"), "8", exe, "--dry-run", "--windows", "1000003"], env=env,
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK (dry run") == 8 and "all 8 ranks OK" in r.stdout
    shards = sorted((int(m.group(1)), int(m.group(2)), int(m.group(3))) for m in re.finditer(r"abi_ranks\[(\d)\]: shard \[(\d+), (\d+)\) of 1000003", r.stdout))
    assert [s[0] for s in shards] == list(range(8)) and shards[0][1] == 0 and shards[-1][2] == 1000003
    assert all(a[2] == b[1] for a, b in zip(shards, shards[1:]))
    sizes = [s[2] - s[1] for s in shards]
    from deep_contact_estimator_amd.distributed import shard_sizes
    assert sizes == shard_sizes(1000003, 8) and max(sizes) - min(sizes) == 1
    env = dict(os.environ, DCE_COMM_TIMEOUT="1")
    r = subprocess.run([exe, "--rank", "7", "--world", "8", "--id-file", str(idf), "--nonce", "another-job", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no id with this job's nonce" in r.stderr


def test_python_id_file_rendezvous_rejects_a_stale_id(tmp_path, monkeypatch):
    """distributed.id_file_rendezvous (DCE_COMM_ID_FILE without a torch.distributed group): same file format, same rules."""
    import threading
    import time
    from deep_contact_estimator_amd import distributed as d
    p = str(tmp_path / "id")
    open(p, "wb").write(b"x" * 144)
    monkeypatch.setenv("DCE_COMM_NONCE", "job-a")
    got = {}
    ts = [threading.Thread(target=lambda r=r: got.__setitem__(r, d.id_file_rendezvous(p, r, None, timeout=20))) for r in (1, 2)]
    [t.start() for t in ts]
    time.sleep(0.2)
    got[0] = d.id_file_rendezvous(p, 0, lambda: bytes(range(128)))
    [t.join() for t in ts]
    assert got[0] == got[1] == got[2] == bytes(range(128))
    monkeypatch.setenv("DCE_COMM_NONCE", "job-b")
    with pytest.raises(RuntimeError, match="DCE_COMM_NONCE"):
        d.id_file_rendezvous(p, 1, None, timeout=0.3)


@pytest.mark.gpu
def test_n_rank_client_runs_in_a_world_of_one(tmp_path):
    """The whole torch-free flow on the one GPU of this box: launcher -> id file -> dce_comm_init -> shard ->
    dce_gather_results -> byte comparison on the root -> dce_allreduce_counts."""
    exe = _build_ranks(str(tmp_path / "abi_ranks"))
    env = dict(os.environ, DCE_COMM_ID_FILE=str(tmp_path / "id"))
    r = subprocess.run([os.path.join(ROOT, "tools", "launch_ranks.sh"), "1", exe, "--windows", "700"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_ranks[0]: OK" in r.stdout and "every gathered row equals the root's own" in r.stdout
    assert not os.path.exists(str(tmp_path / "id"))              # rank 0 removed the id file once everybody had joined
    print(r.stdout.strip())
