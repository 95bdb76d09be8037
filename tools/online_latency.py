#!/usr/bin/env python3
"""Per-sample latency of the online mode (push one (54,) sample, get one estimate): default launches, hipGraph form."""
import os, sys, time, json, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    from deep_contact_estimator_amd import contact_cnn, synth
    m = contact_cnn(device=0, max_batch=64, precision=os.environ.get("DCE_LAT_PRECISION", "fp32")); m.load_state_dict(synth.make_state_dict(1, "uniform"))
    seq = synth.make_sequence(150 + 6000, 5).astype(np.float32)
    m.online_reset()
    for t in range(400): m.online_push(seq[t])
    t0 = time.perf_counter(); k = 0
    for t in range(400, len(seq)):
        if m.online_push(seq[t]) is not None: k += 1
    dt = (time.perf_counter() - t0) / k
    print("RESULT " + json.dumps({"us_per_push": dt * 1e6, "pushes": k}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(); sys.exit(0)
    out = {}
    for tag, env in (("launches", {}), ("graph", {"DCE_TUNE": "online_graph=1"})):
        p = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        out[tag] = json.loads(line[-1][7:]) if line else p.stderr[-500:]
    print(json.dumps(out))
