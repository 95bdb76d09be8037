#!/usr/bin/env python3
"""CPU emulation of the operand rounding of DCE_FP32_F16X2 (csrc/conv_h2.hip), isolated from fp32 accumulation: the net of reference
src/contact_cnn.py:60-66 in float64 with every operand of conv1..4 and fc.0 replaced by its two fp16 terms (h1 k1 + h1 k2 + h2 k1, the
dropped h2 k2 included in the error), against the same net on exact operands, as err / bound with bound = 1e-5 max|ref| + 1e-4 |ref|.
Swept over where the scale puts a layer's largest activation (2^14 .. 2^-6) and with fp16 subnormals flushed or honoured; a two-term
bf16 split for comparison.  Result (512 windows, N(0,1) and AR(1)): fp16 x 2 costs 0.014 of the bound wherever the largest activation is
scaled to 2^2 or above (subnormals honoured: down to 2^-2 at 0.1), bf16 x 2 0.74 - 1.02.   python tools/emulate_f16x2.py
"""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from deep_contact_estimator_amd import synth
torch.set_num_threads(16)
def split_f16(x, scale, flush=False):
    # x float64 tensor holding fp32 values; returns (a1, a2) float64 as exact values of fp16 terms (unscaled back)
    xs = (x * scale).to(torch.float32)
    a1 = xs.to(torch.float16)
    r = xs - a1.to(torch.float32)
    a2 = r.to(torch.float16)
    a1 = a1.to(torch.float64); a2 = a2.to(torch.float64)
    if flush:
        tiny = 2.0 ** -14
        a1 = torch.where(a1.abs() < tiny, torch.zeros_like(a1), a1)
        a2 = torch.where(a2.abs() < tiny, torch.zeros_like(a2), a2)
    return a1 / scale, a2 / scale
def split_bf16_2(x, scale, flush=False):
    xs = x.to(torch.float32)
    a1 = xs.to(torch.bfloat16); r = xs - a1.to(torch.float32); a2 = r.to(torch.bfloat16)
    return a1.to(torch.float64), a2.to(torch.float64)
def p2scale(t, target=2.0**14):
    m = float(t.abs().max())
    return 2.0 ** np.floor(np.log2(target / m)) if m > 0 else 1.0
def lin(op, a, w, b, mode, act_scale, flush):
    # op(a, w) bilinear
    if mode == "exact":
        return op(a, w) + b
    sp = split_f16 if mode == "f16" else split_bf16_2
    ws = p2scale(w)
    w1, w2 = sp(w, ws, flush)
    a1, a2 = sp(a, act_scale(a), flush)
    y = op(a1, w1) + op(a1, w2) + op(a2, w1)
    return (y + b).to(torch.float32).to(torch.float64)     # result rounded to fp32 as the kernel's accumulator would hold it
def forward(sd, x, mode, act_scale, flush=False, fc3_exact=True):
    c = lambda a, w: F.conv1d(a, w, None, padding=1)
    l = lambda a, w: F.linear(a, w)
    rb = lambda k: sd[k].view(1, -1, 1)
    x = x.permute(0, 2, 1)
    x = F.relu(lin(c, x, sd["block1.0.weight"], rb("block1.0.bias"), mode, act_scale, flush))
    x = F.relu(lin(c, x, sd["block1.2.weight"], rb("block1.2.bias"), mode, act_scale, flush))
    x = F.max_pool1d(x, 2, 2)
    x = F.relu(lin(c, x, sd["block2.0.weight"], rb("block2.0.bias"), mode, act_scale, flush))
    x = F.relu(lin(c, x, sd["block2.2.weight"], rb("block2.2.bias"), mode, act_scale, flush))
    x = F.max_pool1d(x, 2, 2)
    x = x.reshape(x.shape[0], -1)
    x = F.relu(lin(l, x, sd["fc.0.weight"], sd["fc.0.bias"], mode, act_scale, flush))
    m2 = "exact" if fc3_exact else mode
    x = F.relu(lin(l, x, sd["fc.3.weight"], sd["fc.3.bias"], m2, act_scale, flush))
    return lin(l, x, sd["fc.6.weight"], sd["fc.6.bias"], "exact", act_scale, flush)
if __name__ == "__main__":
    n = 512
    for kind in ("normal", "ar1"):
        sd = {k: torch.from_numpy(v).double() for k, v in synth.make_state_dict(1, "uniform").items()}
        seq = synth.make_sequence(n + 149, 3, kind)
        w = np.stack([seq[i:i+150] for i in range(n)])
        w = (w - w.mean(1, keepdims=True)) / w.std(1, ddof=1, keepdims=True)
        x = torch.from_numpy(w.astype(np.float32)).double()
        ref = forward(sd, x, "exact", None)
        bound = 1e-5 * ref.abs().max() + 1e-4 * ref.abs()
        for mode in ("f16", "bf16"):
            for tgt in (2.0**14, 2.0**8, 2.0**2, 2.0**-2, 2.0**-6):
                for flush in (False, True):
                    sc = lambda a, tgt=tgt: p2scale(a, tgt)
                    y = forward(sd, x, mode, sc, flush)
                    e = ((y - ref).abs() / bound).max().item()
                    fl = (y.argmax(1) != ref.argmax(1)).sum().item()
                    print(f"{kind:7s} {mode:5s} act max -> 2^{int(np.log2(tgt)):3d} flush={int(flush)}  err/bound {e:.4f}  max|d| {float((y-ref).abs().max()):.3e}  argmax diff {fl}")
                if mode == "bf16": break
