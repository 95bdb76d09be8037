"""GPU (-m gpu): a short randomised differential run of the product path against the ORACLE (tools/fuzz_parity.py: call sizes on both sides of every plan
threshold, precisions, windows | raw rows, host | device pointers, the latency option, random checkpoints and input scales, several sizes per context).
The long runs are profiles/r6p_fuzz_parity.json; this keeps 30 s of it in the driver's suite."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_calls_hold_the_contract(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "15", str(seed)], env=dict(os.environ, PYTHONPATH=ROOT),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.returncode, r.stderr[-2000:])
    j = json.loads(lines[-1])
    assert r.returncode == 0 and not j["violations"], j["violations"]
    assert j["calls"] >= 10 and all(v <= 1.0 for v in j["worst_err_over_bound"].values()), j["worst_err_over_bound"]
