#!/usr/bin/env python3
"""Turn the raw outputs of one measurement session (tools/profile_session.sh <tag>, under gpurun_out/<tag>/) into the
committed artefacts under profiles/:
  <tag>_bench.json, <tag>_bench_k20.json      the bench lines (defaults / the driver's --steps 20 --warmup 5)
  <tag>_kernel_stats_depth{1,3}.csv           rocprofv3 --kernel-trace --stats summaries
  <tag>_summary.md                            per-family kernel time, HBM traffic, instruction / wait counters
  <tag>_counters.json                         what bench.py's `roofline` block reads: HBM bytes per launch per family and the
                                              two counter-derived fractions BASELINE.json's north_star names
usage: python tools/make_profile_summary.py r02 [out_tag]
Peaks (MI355X_MICROARCH.md): HBM 8 TB/s spec; int8 MFMA 16x16x64 = 2x the bf16 rate = 16 cycles per instruction per SIMD,
1024 SIMDs; SQ_* wave counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE in cycles.
"""
import collections
import csv
import glob
import json
import shutil
import sys

FAM = (("expdw_image", "fused_expand_dw"), ("expdw2_kernel", "fused_expand_dw"), ("pw_d", "pw_conv_mfma_i8"), ("pw_e", "pw_conv_mfma_i8"), ("pw_multi", "pw_conv_mfma_i8"), ("sepconv_band", "fused_sepconv_band"), ("fused_block_multi", "fused_heads_multi"), ("stem_block", "fused_stem_block"),
       ("mbconv_image", "fused_mbconv"), ("dw_tile", "dw_conv"), ("dw_col", "dw_conv"), ("dw_kernel", "dw_conv"),
       ("pw_a", "pw_conv_mfma_i8"), ("pw_b", "pw_conv_mfma_i8"), ("pw_c", "pw_conv_mfma_i8"), ("stem_kernel", "stem_conv_mfma_i8"),
       ("add_kernel", "add_requant"), ("maxpool", "maxpool3x3s2"), ("resize_kernel", "resize_nn"), ("postprocess", "decode_nms"),
       ("tracker_from", "ocsort_step"), ("tracker_seq", "ocsort_walk"), ("gather_frames", "frame_gather"), ("analyze", "rep_analysis"), ("select_gather", "export_select"), ("pack_summary", "close_pack"))
FIRST = "stem_block_kernel"        # first kernel of every forward
HBM_PEAK = 8.0e12
SIMDS, CLOCK = 1024, 2.4e9


def fam(n):
    for k, v in FAM:
        if k in n:
            return v
    if "fused_block_kernel" in n:
        t = n.split("<")[1].split(">")[0].split(",")
        return "fused_mbconv" if t[3].strip() in ("true", "1") else "fused_sepconv+bifpn_node"
    return "runtime/other"


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    if not g:
        sys.exit(f"missing {pattern}")
    return g[0]


def steady(d, nf=40):
    tr = sorted(csv.DictReader(open(one(f"{d}/**/*kernel_trace.csv"))), key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(tr) if FIRST in r["Kernel_Name"]]
    nf = min(nf, len(idx) - 1)
    st = tr[idx[-(nf + 1)]:idx[-1]]
    agg = collections.OrderedDict()
    for r in st:
        a = agg.setdefault(fam(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t0 = min(int(r["Start_Timestamp"]) for r in st)
    t1 = max(int(r["End_Timestamp"]) for r in st)
    return agg, nf, (t1 - t0) / nf / 1e6


def counters(d):
    """{family: {counter: value per forward}} from the second-to-last complete forward of a --pmc pass"""
    rs = list(csv.DictReader(open(one(f"{d}/**/*counter_collection.csv"))))
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    ids = sorted({int(r["Dispatch_Id"]) for r in rs if FIRST in r["Kernel_Name"]})
    lo, hi = ids[-3], ids[-2]
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in rs:
        di = int(r["Dispatch_Id"])
        if lo <= di < hi:
            f = fam(r["Kernel_Name"])
            out[f][r["Counter_Name"]] += float(r["Counter_Value"])
            if di not in seen:
                seen.add(di)
                launches[f] += 1
    return out, launches


def summarize(src, pre, out_tag, title, plan, alg_mb_forward, alg_mb_frame, bench=None, k20=None):
    """src/<pre>trace_d1 ... -> profiles/<out_tag>_{kernel_stats_depth*.csv, summary.md, counters.json}"""
    shutil.copy(one(f"{src}/{pre}trace_d3/**/*kernel_stats.csv"), f"profiles/{out_tag}_kernel_stats_depth3.csv")
    shutil.copy(one(f"{src}/{pre}trace_d1/**/*kernel_stats.csv"), f"profiles/{out_tag}_kernel_stats_depth1.csv")
    a3, nf3, wall3 = steady(f"{src}/{pre}trace_d3")
    a1, nf1, wall1 = steady(f"{src}/{pre}trace_d1")
    cf, lf = counters(f"{src}/{pre}pmc_fetch")
    cw, _ = counters(f"{src}/{pre}pmc_write")
    ca, _ = counters(f"{src}/{pre}pmc_sqa")
    cb, _ = counters(f"{src}/{pre}pmc_sqb")
    fams = sorted(a1, key=lambda k: -a1[k][1])
    hbm = {}
    for k in fams:
        fetch = 2.0 * cf[k].get("FETCH_SIZE", 0.0) * 1024       # gfx950: FETCH_SIZE reports half of a wide streaming read; KB units
        write = cw[k].get("WRITE_SIZE", 0.0) * 1024
        hbm[k] = (fetch, write)
    us = {k: a1[k][1] / nf1 / 1e3 for k in fams}                  # kernel microseconds per forward (depth 1)
    with open(f"profiles/{out_tag}_summary.md", "w") as f:
        f.write(f"# {out_tag}: {title} (64 clips per step, plan = {plan}, tools/profile_session.sh)\n\n")
        if bench is not None:
            f.write(f"Un-profiled `python bench.py` (defaults: 1000 steps): **{bench['value']:.0f} frames/s end-to-end, {bench['ms_per_step']:.3f} ms/step** "
                    f"(pipeline depth {bench['config'].get('pipeline_depth', 3)}); H2D-inclusive (pinned host frames in, rows on the host out, same W / K) "
                    f"{bench.get('value_h2d_inclusive', 0):.0f} frames/s; "
                    f"detector only {bench['splits']['detect_only']['frames_per_s']:.0f} frames/s; OC-SORT step {bench['splits']['track_only']['us_per_step']:.1f} us per 64 clips.\n")
            cfgs = bench.get("configs") or {}
            if cfgs:
                f.write("Other configurations in the same line: " + "; ".join(f"{k} {v['frames_per_s']:.0f} frames/s" + (f" ({100 * v['roofline_frac_8d']:.1f} % of the 8d roofline)" if "roofline_frac_8d" in v else "")
                                                                              for k, v in cfgs.items() if "frames_per_s" in v) + ".\n")
        if k20 is not None:
            f.write(f"Driver-style `python bench.py --steps 20 --warmup 5`: **{k20['value']:.0f} frames/s, {k20['ms_per_step']:.3f} ms/step** "
                    f"(timed region: enqueue {k20['timed_region_ms']['enqueue']:.2f} ms, clip close incl. pipeline drain {k20['timed_region_ms']['clip_close']:.2f} ms); "
                    f"the same run repeated after a clock-settle phase {k20.get('value_settled', 0):.0f} frames/s; "
                    f"H2D-inclusive with the same W / K {k20.get('value_h2d_inclusive', 0):.0f} frames/s.\n")
        f.write("Under `rocprofv3 --kernel-trace` dispatches serialise, so the overlap between the forwards in flight is lost while profiling: "
                f"trace wall per step {wall3:.3f} ms (depth 3) / {wall1:.3f} ms (`VBT_PIPELINE_DEPTH=1`).\n\n")
        for ttl, agg, nf, wall in (("depth 1 (one forward at a time)", a1, nf1, wall1), ("depth 3 (bench.py default)", a3, nf3, wall3)):
            tot = sum(v[1] for v in agg.values())
            f.write(f"## kernel time, {ttl}\n\n| family | launches/step | avg us/launch | ms/step | % GPU time |\n|---|---|---|---|---|\n")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"| {k} | {v[0]/nf:.1f} | {v[1]/v[0]/1e3:.2f} | {v[1]/nf/1e6:.4f} | {100*v[1]/tot:.1f} |\n")
            f.write(f"\nSum of kernel durations per step: {tot/nf/1e6:.3f} ms; trace wall per step: {wall:.3f} ms.\n\n")
        f.write("## HBM-side traffic of ONE forward (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes at depth 1; FETCH_SIZE x2 per the gfx950 "
                "note of MI355X_MICROARCH.md, WRITE_SIZE as is, KB -> bytes)\n\n| family | launches | fetch MB | write MB | total MB | kernel us | achieved GB/s | % of 8 TB/s |\n|---|---|---|---|---|---|---|---|\n")
        tot_b = 0
        for k in fams:
            fb, wb = hbm[k]
            tot_b += fb + wb
            gbs = (fb + wb) / (us[k] * 1e-6) / 1e9 if us[k] else 0
            f.write(f"| {k} | {lf[k]} | {fb/1e6:.1f} | {wb/1e6:.1f} | {(fb+wb)/1e6:.1f} | {us[k]:.1f} | {gbs:.0f} | {100*gbs*1e9/HBM_PEAK:.1f} |\n")
        f.write(f"\nTotal {tot_b/1e6:.0f} MB per 64-frame forward = {tot_b/64/1e6:.1f} MB/frame, against {alg_mb_forward:.0f} MB ({alg_mb_frame:.1f} MB/frame, int8) of compulsory traffic "
                "of the unfused graph (SURVEY.md 8d scaled to 1 byte per element).\n\n")
        f.write("## SQ counters of ONE forward (two `--pmc` passes at depth 1; instruction counts in millions of wave-instructions; wave-state "
                "counters as a share of SQ_WAVE_CYCLES)\n\n| family | VALU | SALU | LDS | VMEM_RD | MFMA | MFMA busy % of kernel | LDS bank conflict % | waves parked (WAIT_ANY) % | issue stall (WAIT_INST_ANY) % | issuing (ACTIVE_INST_ANY) % |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        derived = {}
        for k in fams:
            b, a = cb[k], ca[k]
            wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            mfma_busy = b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (us[k] * 1e-6 * CLOCK * SIMDS) if us[k] else 0.0
            conf = b.get("SQ_LDS_BANK_CONFLICT", 0.0) / (b.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0)
            derived[k] = {"mfma_busy_frac": mfma_busy, "lds_bank_conflict_frac": conf, "valu_minsts": b.get("SQ_INSTS_VALU", 0.0) / 1e6,
                          "wait_any_frac": a.get("SQ_WAIT_ANY", 0.0) / wc, "active_inst_frac": a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
            f.write(f"| {k} | {b.get('SQ_INSTS_VALU',0)/1e6:.2f} | {b.get('SQ_INSTS_SALU',0)/1e6:.2f} | {b.get('SQ_INSTS_LDS',0)/1e6:.2f} | "
                    f"{b.get('SQ_INSTS_VMEM_RD',0)/1e6:.2f} | {b.get('SQ_INSTS_MFMA',0)/1e6:.2f} | {100*mfma_busy:.1f} | {100*conf:.1f} | "
                    f"{100*a.get('SQ_WAIT_ANY',0)/wc:.1f} | {100*a.get('SQ_WAIT_INST_ANY',0)/wc:.1f} | {100*a.get('SQ_ACTIVE_INST_ANY',0)/wc:.1f} |\n")
        valu = sum(cb[k].get("SQ_INSTS_VALU", 0.0) for k in fams)
        f.write(f"\nVALU issue floor of the forward: {valu/1e6:.0f} M wave-instructions x 2.8..3.5 cycles / 1024 SIMDs / 2.4 GHz = "
                f"{valu*2.8/SIMDS/CLOCK*1e3:.3f}..{valu*3.5/SIMDS/CLOCK*1e3:.3f} ms (issue cost per SIMD measured with tools/probes/valu_rate.hip for the "
                f"VOP3 / VOP1 / packed forms the requantisation is made of: 3.5 cycles per wave-instruction with two waves per SIMD, 2.8 with four; "
                f"plain VOP2 adds / multiplies 2.4 / 1.7; one wave alone 4.8-5.8).\n\n")
        # the two fractions BASELINE.json's north_star names
        dw_fams = [k for k in fams if k in ("dw_conv", "fused_expand_dw", "fused_sepconv_band")]
        pw = "pw_conv_mfma_i8"
        f.write("## The two counter-derived fractions named by the north_star\n\n")
        nstar = {}
        for k in dw_fams + [x for x in ("fused_mbconv",) if x in us]:
            fb, wb = hbm[k]
            gbs = (fb + wb) / (us[k] * 1e-6)
            nstar[f"{k}_hbm_frac"] = gbs / HBM_PEAK
            f.write(f"* depthwise-bearing family `{k}`: {(fb+wb)/1e6:.1f} MB over {us[k]:.1f} us = {gbs/1e9:.0f} GB/s = **{100*gbs/HBM_PEAK:.1f} % of the 8 TB/s HBM peak**\n")
        if pw in us:
            nstar["pw_mfma_util"] = derived[pw]["mfma_busy_frac"]
            f.write(f"* pointwise convs `{pw}` (16x16x64 int8 MFMA): SQ_VALU_MFMA_BUSY_CYCLES {cb[pw].get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1e6:.1f} M over "
                    f"{us[pw]:.1f} us x 2.4 GHz x 1024 SIMDs = **{100*derived[pw]['mfma_busy_frac']:.1f} % MFMA utilisation**\n")
    json.dump({"plan": plan, "batch": 64, "session": out_tag,
               "families": {k: {"launches": lf[k], "hbm_bytes_per_launch": (hbm[k][0] + hbm[k][1]) / lf[k], "kernel_us_per_forward": us[k]} for k in fams if lf[k]},
               "derived": {"per_family": derived, "north_star": nstar}},
              open(f"profiles/{out_tag}_counters.json", "w"), indent=1)
    print(open(f"profiles/{out_tag}_summary.md").read())


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out_tag = sys.argv[2] if len(sys.argv) > 2 else tag
    src = f"gpurun_out/{tag}"
    bench = json.loads(open(f"{src}/bench_default.json").read().strip().splitlines()[-1])
    k20 = json.loads(open(f"{src}/bench_k20.json").read().strip().splitlines()[-1])
    json.dump(bench, open(f"profiles/{out_tag}_bench.json", "w"), indent=1)
    json.dump(k20, open(f"profiles/{out_tag}_bench_k20.json", "w"), indent=1)
    summarize(src, "", out_tag, "measurement session on one MI355X, EfficientDet-Lite0 320x320", "profiles/plan_lite0.b64.f0", 2450.0, 38.3, bench, k20)
    if glob.glob(f"{src}/lite2_trace_d1"):                       # BASELINE config 4: EfficientDet-Lite2 448x448
        summarize(src, "lite2_", out_tag + "_lite2", "EfficientDet-Lite2 448x448 on one MI355X", "profiles/plan_lite2.b64.f0", 64 * 115.645, 115.6)


if __name__ == "__main__":
    main()
