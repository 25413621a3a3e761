#!/usr/bin/env python3
"""Generate a synthetic, fully-integer-quantised EfficientDet-Lite model container.

Why synthetic: every models/*.tflite of the reference is missing (.MISSING_LARGE_BLOBS:2-6,8)
and there is no network.  SURVEY.md section 8d prescribes seeded random weights
(PCG64(seed=1), He-normal, BN folded).  The reference models are full-integer-quantised
(SURVEY.md section 0 item 4), so the stand-in is too:

  1. float weights: He-normal, then a data-dependent per-channel rescale on calibration frames
     (what a folded BatchNorm with real statistics does) so activations stay inside ReLU6's
     useful range through all ~60 layers;
  2. post-training quantisation exactly like the TFLite converter: per-tensor asymmetric int8
     activations from calibration min/max, per-output-channel symmetric int8 weights, int32 bias
     at scale s_x*s_w[c], float32 requantisation multiplier M[c] = s_x*s_w[c]/s_y.

torch (CPU) is used here only as a float conv engine for calibration; the output file is
committed under models/ so nothing in tests/bench depends on re-running this tool.

usage: python tools/make_model.py --arch 0 --out models/efficientdet_lite0_synth.vbtm
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vbt_amd import quant, spec, synth  # noqa: E402
from vbt_amd.container import (OP_DTYPE, TENSOR_DTYPE, BlobWriter, write_container)  # noqa: E402


def rne(x):
    return np.rint(x)


def qparams(lo, hi):
    """TFLite-style asymmetric int8 params with an exactly representable zero."""
    lo, hi = min(float(lo), 0.0), max(float(hi), 0.0)
    if hi - lo < 1e-6:
        hi = lo + 1e-6
    scale = np.float32((hi - lo) / 255.0)
    zp = int(np.clip(rne(-128 - lo / float(scale)), -128, 127))
    return scale, zp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", type=int, default=0)
    ap.add_argument("--out", type=str, required=True)
    ap.add_argument("--calib", type=int, default=8)
    ap.add_argument("--pos_frac", type=float, default=1e-4, help="fraction of anchors with score >= 0.5")
    ap.add_argument("--nms_score_threshold", type=float, default=1.0 / 256)
    ap.add_argument("--num_classes", type=int, default=spec.NUM_CLASSES, help="class columns per anchor (the reference's models: 2, vbt_amd/spec.py)")
    ap.add_argument("--tie_adds", action="store_true",
                    help="test model: force every ADD's input/output scale ratios to exactly 0.5 or 1.0, so that the integer "
                         "rounding of XNNPACK's qs8-vadd (half towards +infinity) is hit on about half of all elements")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    g = spec.build_graph(args.arch, args.num_classes)
    S = g.cfg.image_size
    rng = np.random.Generator(np.random.PCG64(1))
    rngn = np.random.Generator(np.random.PCG64(2))      # BN-like targets
    frames = np.stack([synth.render(synth.background(1000 + i, S), 17 * i) for i in range(args.calib)])
    x0 = (torch.from_numpy(frames).float() - 127.0) / 128.0
    val = {0: x0.permute(0, 3, 1, 2).contiguous()}
    rng_of = {0: (-127 / 128.0, 1.0)}

    fw, fb = {}, {}          # op index -> float weight (torch layout), bias
    base_w = {}              # share key -> base float weights
    dw_final = {}            # share key -> (w, b) reused verbatim (depthwise has no BN of its own)
    last_head = {"cls": [], "box": []}

    def pad_same(x, op, value):
        ih, iw = x.shape[2], x.shape[3]
        oh, ow = g.tensors[op.output].h, g.tensors[op.output].w
        pb = max((oh - 1) * op.stride + op.k - ih - op.pad_t, 0)
        pr = max((ow - 1) * op.stride + op.k - iw - op.pad_l, 0)
        return F.pad(x, (op.pad_l, pr, op.pad_t, pb), value=value)

    for oi, op in enumerate(g.ops):
        tin = [g.tensors[i] for i in op.inputs]
        tout = g.tensors[op.output]
        if op.type in (spec.OP_STEM, spec.OP_PW, spec.OP_DW):
            x = val[op.inputs[0]]
            cin, cout = tin[0].c, tout.c
            is_dw = op.type == spec.OP_DW
            key = op.share or op.name
            if key in dw_final:
                w, b = dw_final[key]
            else:
                if key in base_w:
                    w = base_w[key].clone()
                else:
                    if is_dw:
                        w = rng.normal(0.0, math.sqrt(2.0 / (op.k * op.k)), size=(cout, 1, op.k, op.k))
                    else:
                        w = rng.normal(0.0, math.sqrt(2.0 / (cin * op.k * op.k)), size=(cout, cin, op.k, op.k))
                    w = torch.from_numpy(w.astype(np.float32))
                    base_w[key] = w.clone()
                xp = pad_same(x, op, 0.0) if op.k > 1 else x
                y = F.conv2d(xp, w, None, stride=op.stride, groups=cin if is_dw else 1)
                mu = y.mean(dim=(0, 2, 3)).numpy().astype(np.float64)
                sd = y.std(dim=(0, 2, 3)).numpy().astype(np.float64) + 1e-3
                final_head = op.name.endswith(".out.pw")
                sep_dw = is_dw and op.act == spec.ACT_NONE          # depthwise half of a SeparableConv
                if final_head:
                    tgt_sd = np.full(cout, 1.0 if op.name.startswith("cls") else 0.25)
                    g_sd = float(np.sqrt((y.numpy().astype(np.float64) ** 2).mean())) + 1e-6
                    scale = tgt_sd / g_sd                       # one scale for all channels (shared conv)
                    b = np.zeros(cout)
                elif sep_dw:
                    scale = 1.0 / sd
                    b = np.zeros(cout)
                else:
                    if op.act == spec.ACT_RELU6:
                        tgt_sd = rngn.uniform(0.8, 1.4, cout)
                        tgt_mu = rngn.uniform(0.1, 1.2, cout)
                    else:
                        tgt_sd = rngn.uniform(0.7, 1.3, cout)
                        tgt_mu = rngn.normal(0.0, 0.25, cout)
                    scale = tgt_sd / sd
                    b = tgt_mu - mu * scale
                w = w * torch.from_numpy(np.asarray(scale, dtype=np.float32)).reshape(-1, 1, 1, 1)
                b = torch.from_numpy(np.asarray(b, dtype=np.float32))
                if is_dw and op.share:
                    dw_final[key] = (w, b)
            fw[oi], fb[oi] = w, b
            xp = pad_same(x, op, 0.0) if op.k > 1 else x
            y = F.conv2d(xp, w, b, stride=op.stride, groups=cin if is_dw else 1)
            if op.act == spec.ACT_RELU6:
                y = y.clamp(0.0, 6.0)
            val[op.output] = y
            if op.name.endswith(".out.pw"):
                last_head[op.name[:3]].append(oi)
        elif op.type == spec.OP_ADD:
            y = val[op.inputs[0]]
            for i in op.inputs[1:]:
                y = y + val[i]
            if op.act == spec.ACT_RELU6:
                y = y.clamp(0.0, 6.0)
            val[op.output] = y
        elif op.type == spec.OP_MAXPOOL:
            xp = pad_same(val[op.inputs[0]], op, float("-inf"))
            val[op.output] = F.max_pool2d(xp, 3, 2)
        elif op.type == spec.OP_RESIZE_NN:
            x = val[op.inputs[0]]
            ih, iw = x.shape[2], x.shape[3]
            iy = torch.tensor([(d * ih) // tout.h for d in range(tout.h)])
            ix = torch.tensor([(d * iw) // tout.w for d in range(tout.w)])
            val[op.output] = x[:, :, iy][:, :, :, ix]
        elif op.type == spec.OP_POSTPROCESS:
            pass

    # ---- class bias so that pos_frac of the anchors score >= 0.5 on the calibration frames
    # (per anchor the post-process scores the best of its class columns: channel = anchor_in_location * num_classes + class)
    cls_logits = np.concatenate([val[g.ops[oi].output].permute(0, 2, 3, 1).numpy().reshape(-1, g.num_classes).max(axis=1) for oi in last_head["cls"]])
    cls_bias = -float(np.quantile(cls_logits, 1.0 - args.pos_frac))
    for oi in last_head["cls"]:
        fb[oi] = fb[oi] + cls_bias
        val[g.ops[oi].output] = val[g.ops[oi].output] + cls_bias

    # ---- activation quantisation parameters
    tq = {}
    tq[0] = (np.float32(1.0 / 128.0), -1)
    for op in g.ops:
        if op.type in (spec.OP_MAXPOOL, spec.OP_RESIZE_NN):
            tq[op.output] = tq[op.inputs[0]]
        elif op.type == spec.OP_POSTPROCESS:
            tq[op.output] = (np.float32(1.0), 0)
        else:
            v = val[op.output]
            tq[op.output] = qparams(v.min().item(), v.max().item())
    for head in ("cls", "box"):                       # concat inputs share one scale
        outs = [g.ops[oi].output for oi in last_head[head]]
        lo = min(val[t].min().item() for t in outs)
        hi = max(val[t].max().item() for t in outs)
        for t in outs:
            tq[t] = qparams(lo, hi)

    if args.tie_adds:
        producer = {op.output: op for op in g.ops}

        def requ(t, scale):
            lo = min(val[t].min().item(), 0.0)
            tq[t] = (np.float32(scale), int(np.clip(rne(-128 - lo / float(scale)), -128, 127)))

        def source(t):          # the tensor whose quantisation a resize / max-pool output inherits
            while t in producer and producer[t].type in (spec.OP_MAXPOOL, spec.OP_RESIZE_NN):
                t = producer[t].inputs[0]
            return t

        adds = [op for op in g.ops if op.type == spec.OP_ADD]
        fpn = [op for op in adds if op.stage == "fpn"]
        sums = {op.output for op in fpn}
        feats = {source(i) for op in fpn for i in op.inputs if i not in sums}
        s_f = max(float(tq[t][0]) for t in feats)
        for t in feats:
            requ(t, s_f)                                     # every feature map entering a sum: one scale
        for op in fpn:
            requ(op.output, 2.0 * s_f)                       # partial sums: 0.5 / 0.5; final sums: 1.0 / 0.5 or 0.5 / 0.5
        for op in adds:
            if op.stage == "backbone":                       # ADD(project output, block input)
                s_skip = float(tq[op.inputs[1]][0])
                requ(op.inputs[0], s_skip / 2.0)             # 0.5 / 1.0
                requ(op.output, s_skip)
        for op in g.ops:
            if op.type in (spec.OP_MAXPOOL, spec.OP_RESIZE_NN):
                tq[op.output] = tq[op.inputs[0]]

    # ---- emit
    blob = BlobWriter()
    tensors = np.zeros(len(g.tensors), TENSOR_DTYPE)
    for t in g.tensors:
        s, z = tq[t.id]
        tensors[t.id] = (t.h, t.w, t.c, z, s, (0, 0, 0))
    ops = np.zeros(len(g.ops), OP_DTYPE)
    dedupe = {}
    for oi, op in enumerate(g.ops):
        r = ops[oi]
        r["type"] = op.type
        r["n_inputs"] = len(op.inputs)
        r["inputs"][:len(op.inputs)] = op.inputs
        r["output"] = op.output
        r["k"], r["stride"], r["pad_t"], r["pad_l"] = op.k, op.stride, op.pad_t, op.pad_l
        r["level"] = op.level
        so, zo = tq[op.output]
        amin, amax = -128, 127
        if op.act == spec.ACT_RELU6:
            amin = max(-128, zo)
            amax = min(127, zo + int(rne(6.0 / float(so))))
        r["act_min"], r["act_max"] = amin, amax
        if op.type in (spec.OP_STEM, spec.OP_PW, spec.OP_DW):
            sx, _ = tq[op.inputs[0]]
            w = fw[oi].numpy().astype(np.float64)
            b = fb[oi].numpy().astype(np.float64)
            cout = w.shape[0]
            # per-channel weight scales are float32 in a TFLite file; everything below derives from the stored values
            sw = (np.maximum(np.abs(w.reshape(cout, -1)).max(axis=1), 1e-9) / 127.0).astype(np.float32).astype(np.float64)
            wq = np.clip(rne(w / sw.reshape(-1, 1, 1, 1)), -127, 127).astype(np.int8)
            bq = rne(b / (float(sx) * sw)).astype(np.int64)
            assert np.abs(bq).max() < 2 ** 30
            mult = quant.conv_requant_scales(sx, sw, so)           # XNNPACK: (s_x * s_w[c]) / s_y in float32
            if op.type == spec.OP_DW:
                wl = np.ascontiguousarray(wq[:, 0].transpose(1, 2, 0))          # [ky][kx][C]
            elif op.type == spec.OP_STEM:
                wl = np.ascontiguousarray(wq.transpose(0, 2, 3, 1))             # [Cout][ky][kx][Cin]
            else:
                wl = np.ascontiguousarray(wq[:, :, 0, 0])                       # [Cout][Cin]
            kkey = (op.share, wl.tobytes()) if op.share else None
            if kkey is not None and kkey in dedupe:
                r["w_off"] = dedupe[kkey]
            else:
                r["w_off"] = blob.add(wl)
                if kkey is not None:
                    dedupe[kkey] = int(r["w_off"])
            r["b_off"] = blob.add(bq.astype("<i4"))
            r["m_off"] = blob.add(mult)
        elif op.type == spec.OP_ADD:
            assert len(op.inputs) == 2
            (sa, za), (sb, zb) = tq[op.inputs[0]], tq[op.inputs[1]]
            r["in_mult"][0], r["in_mult"][1] = np.float32(sa) / np.float32(so), np.float32(sb) / np.float32(so)
            r["add_q"][:] = quant.xnn_qs8_add_params(sa, sb, so, za, zb)
        elif op.type == spec.OP_POSTPROCESS:
            anchors = spec.make_anchors(args.arch)
            assert anchors.shape[0] == g.num_anchors()
            r["aux_off"] = blob.add(anchors)
            sc, zc = tq[op.inputs[0]]
            sb, zb = tq[op.inputs[5]]
            r["aux2_off"] = blob.add(quant.pack_postprocess_tables(sc, zc, sb, zb))     # LOGISTIC / DEQUANTIZE / decode tables
    header = dict(arch=args.arch, image_size=S, num_anchors=g.num_anchors(), max_detections=spec.MAX_DETECTIONS,
                  nms_iou_threshold=0.5, nms_score_threshold=args.nms_score_threshold, input_tensor=0, num_classes=g.num_classes)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    write_container(args.out, header, tensors, ops, blob.bytes())
    # report
    sat = []
    for op in g.ops:
        if op.act == spec.ACT_RELU6:
            v = val[op.output]
            sat.append(((v <= 0).float().mean().item(), (v >= 6).float().mean().item()))
    sat = np.array(sat)
    print(f"wrote {args.out}: {os.path.getsize(args.out)/1e6:.2f} MB, ops={len(g.ops)} tensors={len(g.tensors)}")
    print(f"relu6 layers: mean frac==0 {sat[:,0].mean():.3f}, mean frac==6 {sat[:,1].mean():.4f}; cls_bias {cls_bias:.3f}")


if __name__ == "__main__":
    main()
