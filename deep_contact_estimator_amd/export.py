"""Result export after the inference path -- SURVEY.md 8(f) rank 2 (host-side, no GPU work):

  save2mat   reference src/inference_one_seq.py:64-89   contacts_est + column-sliced features -> .mat
  save2lcm   reference src/inference_one_seq.py:91-133  LCM event log with three channels per sample

The LCM wire formats are restated here from the reference's .lcm definitions and generated codecs
(lcm_types/contact_t.lcm:1-6, leg_control_data_lcmt.lcm:1-7, microstrain_lcmt.lcm:1-8;
lcm_types/python/contact_t.py:24-32,51-64): big-endian fields behind an 8-byte fingerprint, which
is the type's base hash rotated left by one bit.  Message bytes are pinned by golden vectors
produced with the reference's own generated Python codecs (tests/golden/lcm_messages.npz).
The event-log CONTAINER (liblcm's eventlog.c: sync word 0xEDA1DA01, event number, timestamp in us,
channel length, data length, channel, data -- all big-endian) is restated from the published LCM
log format (lcm-proj/lcm lcm/eventlog.c, identical in every release v1.0.0 - v1.5.x; the reference's
Dockerfiles clone lcm master without a tag).  liblcm is not installed here, so the container cannot be
pinned on liblcm's own output: it is pinned on a hand-derived golden event written out from that format
description (tests/test_export.py::test_event_log_container_hand_derived_golden) plus a round trip --
"parity pinned to the published format, not to liblcm".  The whole log is assembled with vectorised
numpy, not one Python call per message.
"""
from __future__ import annotations

import struct
import time

import numpy as np

LCM_SYNC = 0xEDA1DA01
_BASE_HASH = {                       # lcm-gen base hashes of the three types (from the generated codecs)
    "contact_t": 0x12E312DF2F1D46F6,
    "leg_control_data_lcmt": 0xA7D2775A407DECA7,
    "microstrain_lcmt": 0x710A98F509C97D55,
}


def fingerprint(type_name: str) -> bytes:
    h = _BASE_HASH[type_name]
    h = (((h << 1) & 0xFFFFFFFFFFFFFFFF) + (h >> 63)) & 0xFFFFFFFFFFFFFFFF
    return struct.pack(">Q", h)


def encode_contact_t(num_legs: int, timestamp: float, contact) -> bytes:
    c = [int(x) for x in contact][:num_legs]
    return fingerprint("contact_t") + struct.pack(">bd", num_legs, timestamp) + struct.pack(">%db" % num_legs, *c)


def encode_leg_control_data(q, qd, p, v, tau_est) -> bytes:
    vals = [*q[:12], *qd[:12], *p[:12], *v[:12], *tau_est[:12]]
    return fingerprint("leg_control_data_lcmt") + struct.pack(">60f", *[float(x) for x in vals])


def encode_microstrain(quat, rpy, omega, acc, good_packets: int = 0, bad_packets: int = 0) -> bytes:
    vals = [*quat[:4], *rpy[:3], *omega[:3], *acc[:3]]
    return (fingerprint("microstrain_lcmt") + struct.pack(">13f", *[float(x) for x in vals])
            + struct.pack(">qq", int(good_packets), int(bad_packets)))


def _event_dtype(channel: bytes, payload_len: int):
    return np.dtype([("sync", ">u4"), ("num", ">i8"), ("ts", ">i8"), ("clen", ">i4"), ("dlen", ">i4"),
                     ("chan", "S%d" % len(channel)), ("data", "V%d" % payload_len)])


def build_log(utime: int, imu_time, q, qd, p, v, tau_est, contacts, acc, omega, rpy, quat) -> bytes:
    """The byte image of the reference's log: for every sample i three events, in this order and
    with the same timestamp utime + int(1e6 * imu_time[i]): 'leg_control_data', 'contact',
    'microstrain' (src/inference_one_seq.py:101-131).  All arrays have one row per sample."""
    n = len(imu_time)
    ts = utime + (1e6 * np.asarray(imu_time, np.float64)).astype(np.int64)        # int() truncation
    f32 = lambda a, k: np.asarray(a, np.float64)[:, :k].astype(np.float32)
    be = lambda parts: np.ascontiguousarray(np.concatenate(parts, axis=1).astype(">f4"))   # big-endian image
    pay_leg = be([f32(q, 12), f32(qd, 12), f32(p, 12), f32(v, 12), f32(tau_est, 12)])
    pay_imu = be([f32(quat, 4), f32(rpy, 3), f32(omega, 3), f32(acc, 3)])
    specs = [
        (b"leg_control_data", fingerprint("leg_control_data_lcmt"), pay_leg.view(np.uint8).reshape(n, -1)),
        (b"contact", fingerprint("contact_t"), None),
        (b"microstrain", fingerprint("microstrain_lcmt"),
         np.concatenate([pay_imu.view(np.uint8).reshape(n, -1), np.zeros((n, 16), np.uint8)], axis=1)),
    ]
    # contact_t body: int8 num_legs(4), double timestamp (= imu_time), int8 contact[4]
    cbody = np.zeros(n, dtype=np.dtype([("n", "i1"), ("t", ">f8"), ("c", "i1", (4,))]))
    cbody["n"] = 4
    cbody["t"] = np.asarray(imu_time, np.float64)
    cbody["c"] = np.asarray(contacts).astype(np.int8)[:, :4]
    specs[1] = (specs[1][0], specs[1][1], cbody.view(np.uint8).reshape(n, -1))
    parts, sizes = [], []
    for k, (chan, fp, body) in enumerate(specs):
        dlen = 8 + body.shape[1]
        ev = np.zeros(n, dtype=_event_dtype(chan, dlen))
        ev["sync"] = LCM_SYNC
        ev["num"] = 3 * np.arange(n, dtype=np.int64) + k
        ev["ts"] = ts
        ev["clen"] = len(chan)
        ev["dlen"] = dlen
        ev["chan"] = chan
        data = np.concatenate([np.broadcast_to(np.frombuffer(fp, np.uint8), (n, 8)), body], axis=1)
        ev["data"] = np.ascontiguousarray(data).view("V%d" % dlen).reshape(n)
        parts.append(ev.view(np.uint8).reshape(n, -1))
        sizes.append(parts[-1].shape[1])
    return np.concatenate(parts, axis=1).tobytes()      # row i = its three events back to back


def read_log(buf: bytes):
    """Minimal event-log reader (round-trip tests): -> list of (event number, timestamp, channel, data)."""
    out, o = [], 0
    while o < len(buf):
        sync, num, ts, clen, dlen = struct.unpack_from(">IqqiI", buf, o)
        assert sync == LCM_SYNC
        o += 28
        out.append((num, ts, buf[o:o + clen].decode(), buf[o + clen:o + clen + dlen]))
        o += clen + dlen
    return out


def _np(pred):
    return pred.cpu().numpy() if hasattr(pred, "cpu") else np.asarray(pred)


def save2lcm(pred, config, utime: int | None = None):
    """Mirror of src/inference_one_seq.py:91-133: reads config['mat_data_path'], writes
    config['lcm_save_path'].  Row idx of pred belongs to data row idx + window_size - 1."""
    import scipy.io as sio
    mat = sio.loadmat(config["mat_data_path"])
    w = config["window_size"] - 1
    if utime is None:
        utime = int(time.time() * 10 ** 6)
    imu_time = mat["imu_time"].flatten()[w:]
    pred = _np(pred)
    sl = lambda k: np.asarray(mat[k])[w:w + len(imu_time)]
    blob = build_log(utime, imu_time, sl("q"), sl("qd"), sl("p"), sl("v"), sl("tau_est"), pred[:len(imu_time)],
                     sl("imu_acc"), sl("imu_omega"), sl("imu_rpy"), sl("imu_quat"))
    with open(config["lcm_save_path"], "wb") as f:
        f.write(blob)
    print("Saved data to lcm!")


def save2mat(pred, config):
    """Mirror of src/inference_one_seq.py:64-89 (same keys, same slicing)."""
    import scipy.io as sio
    from .inference import decimal2binary
    mat_raw = sio.loadmat(config["mat_data_path"])
    data = np.load(config["data_path"])
    label = decimal2binary(np.load(config["label_path"]).astype(np.int64)).reshape(-1, 4)
    w = config["window_size"] - 1
    out = {
        "contacts_est": _np(pred), "contacts_gt": label[w:, :],
        "q": data[w:, :12], "qd": data[w:, 12:24], "imu_acc": data[w:, 24:27], "imu_omega": data[w:, 27:30],
        "p": data[w:, 30:42], "v": data[w:, 42:54],
        "control_time": mat_raw["control_time"].flatten().tolist()[w:],
        "imu_time": mat_raw["imu_time"].flatten().tolist()[w:],
        "tau_est": mat_raw["tau_est"][w:], "F": mat_raw["F"][w:],
    }
    sio.savemat(config["mat_save_path"], out)
    print("Saved data to mat!")
