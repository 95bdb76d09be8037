"""CPU: result export against golden vectors made with the reference's own
generated LCM codecs and its save2mat (tests/golden/make_golden.py: export_case)."""
import os

import numpy as np
import pytest

from deep_contact_estimator_amd import export


def test_lcm_message_bytes_match_reference_codecs(golden):
    g = golden("lcm_messages")
    n = int(g["n"])
    for i in range(n):
        d = i + 149
        leg = export.encode_leg_control_data(g["mat_q"][d], g["mat_qd"][d], g["mat_p"][d], g["mat_v"][d], g["mat_tau_est"][d])
        con = export.encode_contact_t(4, float(g["mat_imu_time"][d]), g["contacts"][i])
        imu = export.encode_microstrain(g["mat_imu_quat"][d], g["mat_imu_rpy"][d], g["mat_imu_omega"][d], g["mat_imu_acc"][d])
        assert leg == g["msg_leg"][i].tobytes()
        assert con == g["msg_contact"][i].tobytes()
        assert imu == g["msg_imu"][i].tobytes()
    assert export.fingerprint("contact_t") == g["msg_contact"][0][:8].tobytes()


def test_vectorised_log_equals_per_message_encoding(golden):
    """build_log (numpy, whole log at once) == the three per-message encoders wrapped in the event
    container, in the reference's order and with its timestamps (src/inference_one_seq.py:101-131)."""
    g = golden("lcm_messages")
    n, w = int(g["n"]), 149
    utime = 1_700_000_000_000_000
    sl = lambda k: g["mat_" + k][w:w + n]
    blob = export.build_log(utime, sl("imu_time"), sl("q"), sl("qd"), sl("p"), sl("v"), sl("tau_est"), g["contacts"],
                            sl("imu_acc"), sl("imu_omega"), sl("imu_rpy"), sl("imu_quat"))
    ev = export.read_log(blob)
    assert len(ev) == 3 * n
    for i in range(n):
        ts = utime + int(10 ** 6 * g["mat_imu_time"][w + i])
        want = [("leg_control_data", g["msg_leg"][i]), ("contact", g["msg_contact"][i]), ("microstrain", g["msg_imu"][i])]
        for k, (chan, data) in enumerate(want):
            num, t, c, d = ev[3 * i + k]
            assert (num, t, c) == (3 * i + k, ts, chan) and d == data.tobytes()


def test_event_log_container_hand_derived_golden():
    """The event-log container, pinned to a byte string written out BY HAND from the published LCM log
    format (lcm-proj/lcm, lcm/eventlog.c: lcm_eventlog_write_event writes, big-endian, the int32 sync
    word 0xEDA1DA01, int64 event number, int64 timestamp [us], int32 channel length, int32 data
    length, the channel name without a terminator, the data -- unchanged from v1.0.0 through v1.5.x;
    the reference's Dockerfiles clone lcm master untagged: docker/cuda11_1/Dockerfile:23).  liblcm
    itself is not in this image, so this is the strongest pin available offline; EventLog.write_event
    numbers events 0,1,2,... in write order (the log's own counter), which build_log reproduces."""
    golden_event = bytes.fromhex(
        "eda1da01"                  # sync word
        "0000000000000007"          # event number 7
        "0005af3107a5e240"          # timestamp 1600000000123456 us
        "00000007"                  # channel length
        "00000015"                  # data length 21 = 8 fingerprint + 13 body
        "636f6e74616374"            # "contact"
        "25c625be5e3a8dec"          # contact_t fingerprint: 0x12e312df2f1d46f6 rotated left by one
        "04"                        # int8 num_legs
        "4029000000000000"          # double timestamp 12.5
        "01000001")                 # int8 contact[4]
    data = export.encode_contact_t(4, 12.5, [1, 0, 0, 1])
    assert data == golden_event[35:]
    assert export.read_log(golden_event) == [(7, 1600000000123456, "contact", data)]
    # build_log's rows: event k of sample i carries number 3*i + k, sync/lengths/channel exactly as above
    z = lambda k: np.zeros((3, k))
    blob = export.build_log(1600000000000000, np.array([0.0, 0.123456, 12.5]), z(12), z(12), z(12), z(12), z(12),
                            np.array([[1, 0, 0, 1]] * 3), z(3), z(3), z(3), z(4))
    ev = export.read_log(blob)
    assert [e[0] for e in ev] == list(range(9)) and [e[2] for e in ev] == ["leg_control_data", "contact", "microstrain"] * 3
    row = len(blob) // 3
    second = blob[row:2 * row]                         # sample 1: its 'contact' event sits after the leg event
    leg_len = 28 + len("leg_control_data") + 8 + 240
    got = second[leg_len:leg_len + 28 + 7 + 21]
    want = bytearray(golden_event)
    want[4:12] = (4).to_bytes(8, "big")                # event number 3*1 + 1
    want[44:52] = export.encode_contact_t(4, 0.123456, [1, 0, 0, 1])[9:17]    # the message body's own double timestamp
    assert got == bytes(want)


def test_save2mat_and_save2lcm_match_reference(golden, tmp_path):
    sio = pytest.importorskip("scipy.io")
    g = golden("lcm_messages")
    n, T = int(g["n"]), int(g["n"]) + 149
    mat = {k[4:]: g[k] for k in g.files if k.startswith("mat_")}
    data = np.concatenate([mat["q"], mat["qd"], mat["imu_acc"], mat["imu_omega"], mat["p"], mat["v"]], axis=1)
    cfg = {"mat_data_path": str(tmp_path / "in.mat"), "data_path": str(tmp_path / "d.npy"),
           "label_path": str(tmp_path / "l.npy"), "window_size": 150, "mat_save_path": str(tmp_path / "out.mat"),
           "lcm_save_path": str(tmp_path / "out.lcm")}
    sio.savemat(cfg["mat_data_path"], mat)
    np.save(cfg["data_path"], data)
    np.save(cfg["label_path"], g["labels"])
    export.save2mat(g["contacts"], cfg)
    res = sio.loadmat(cfg["mat_save_path"])
    keys = sorted(k for k in res if not k.startswith("__"))
    assert keys == sorted(k[9:] for k in g.files if k.startswith("save2mat_"))
    for k in keys:
        assert res[k].dtype == g["save2mat_" + k].dtype and np.array_equal(res[k], g["save2mat_" + k]), k
    export.save2lcm(g["contacts"], cfg, utime=42)
    ev = export.read_log(open(cfg["lcm_save_path"], "rb").read())
    assert len(ev) == 3 * n and ev[1][3] == g["msg_contact"][0].tobytes()


def test_ingest_matches_reference_mat2numpy(golden, tmp_path):
    """.mat -> (T,54) float64 + decimal labels, vs the reference's
    utils/mat2numpy.py:16-83,199-200 outputs (tests/golden/ingest.npz)."""
    sio = pytest.importorskip("scipy.io")
    from deep_contact_estimator_amd import ingest
    g = golden("ingest")
    assert np.array_equal(ingest.binary2decimal(g["bits"]), g["dec"]) and ingest.binary2decimal(np.array([[1, 0, 0, 1]])) == 9
    os.makedirs(tmp_path / "mat"); os.makedirs(tmp_path / "npy")
    sio.savemat(tmp_path / "mat" / "seq0.mat", {k[4:]: g[k] for k in g.files if k.startswith("mat_")})
    out = ingest.mat2numpy_one_seq(str(tmp_path / "mat") + "/", str(tmp_path / "npy") + "/")
    data, label = np.load(out[0]), np.load(out[0].replace(".npy", "_label.npy"))
    assert data.dtype == g["data"].dtype and np.array_equal(data, g["data"])
    assert label.shape == g["label"].shape and np.array_equal(label, g["label"])
