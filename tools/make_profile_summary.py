#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs of one measurement session (under gpurun_out/) into the committed artefacts under
profiles/: r01_final_{summary.md,kernel_stats_depth1.csv,kernel_stats_depth3.csv,bench.json,traffic.json}.

Expected inputs (see DESIGN.md section 5 for the commands that produce them):
  gpurun_out/bench_final.json                        python bench.py
  gpurun_out/prof_final/**/kernel_{trace,stats}.csv  rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --cpu-frames 0
  gpurun_out/prof_final_d1/...                       same with VBT_PIPELINE_DEPTH=1
  gpurun_out/pmc_fetchF, gpurun_out/pmc_writeF       rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (depth 1, --no-roofline)
"""
import collections
import csv
import glob
import json
import shutil
import sys

FAM = (("fused_block_multi", "fused_heads_multi"), ("stem_block", "fused_stem_block"), ("mbconv_image", "fused_mbconv"),
       ("dw_tile", "dw_conv_f32acc"), ("dw_col", "dw_conv_f32acc"), ("dw_kernel", "dw_conv_f32acc"), ("pw_a", "pw_conv_mfma_i8"),
       ("pw_b", "pw_conv_mfma_i8"), ("pw_c", "pw_conv_mfma_i8"), ("stem_kernel", "stem_conv_mfma_i8"), ("add_kernel", "add_requant"),
       ("maxpool", "maxpool3x3s2"), ("resize_kernel", "resize_nn"), ("postprocess", "decode_nms"), ("tracker_from", "ocsort_step"),
       ("analyze", "rep_analysis"), ("select_gather", "export_select"))
FIRST = "stem_block_kernel"        # first kernel of every forward


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


def steady(dirn, nf=50, skip=5):
    tr = sorted(csv.DictReader(open(one(f"gpurun_out/{dirn}/**/*kernel_trace.csv"))), key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(tr) if FIRST in r["Kernel_Name"]]
    st = tr[idx[-(nf + 1)]:idx[-1]]
    agg = collections.OrderedDict()
    for r in st:
        a = agg.setdefault(fam(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t0 = min(int(r["Start_Timestamp"]) for r in st)
    t1 = max(int(r["End_Timestamp"]) for r in st)
    return agg, nf, (t1 - t0) / nf / 1e6


def counter_rows(d, counter):
    rs = [r for r in csv.DictReader(open(one(f"gpurun_out/{d}/**/*counter_collection.csv"))) if r["Counter_Name"] == counter]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    ix = [i for i, r in enumerate(rs) if FIRST in r["Kernel_Name"]]
    return rs[ix[-3]:ix[-2]]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
    bench = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
    json.dump(bench, open(f"profiles/{tag}_bench.json", "w"), indent=1)
    shutil.copy(one("gpurun_out/prof_final/**/*kernel_stats.csv"), f"profiles/{tag}_kernel_stats_depth3.csv")
    shutil.copy(one("gpurun_out/prof_final_d1/**/*kernel_stats.csv"), f"profiles/{tag}_kernel_stats_depth1.csv")
    tr_agg = collections.OrderedDict()
    for r in counter_rows("pmc_fetchF", "FETCH_SIZE"):
        a = tr_agg.setdefault(fam(r["Kernel_Name"]), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for r in counter_rows("pmc_writeF", "WRITE_SIZE"):
        tr_agg.setdefault(fam(r["Kernel_Name"]), [0, 0.0, 0.0])[2] += float(r["Counter_Value"])
    a3, nf, wall3 = steady("prof_final")
    a1, _, wall1 = steady("prof_final_d1")
    with open(f"profiles/{tag}_summary.md", "w") as f:
        f.write(f"# {tag}: rocprofv3 on `python bench.py --steps 50 --warmup 5 --cpu-frames 0` (MI355X, 64 clips/step, plan = profiles/plan_lite0.b64.f0)\n\n")
        f.write(f"Un-profiled bench.py (defaults): **{bench['value']:.0f} frames/s end-to-end, {bench['ms_per_step']:.3f} ms/step** (pipeline depth 3).\n")
        f.write("Under `rocprofv3 --kernel-trace --stats` dispatches serialise, so the overlap between the forwards in flight is lost while "
                f"profiling: trace wall per step {wall3:.3f} ms (depth 3) / {wall1:.3f} ms (`VBT_PIPELINE_DEPTH=1`).\n\n")
        for title, agg, wall in (("depth 1 (one forward at a time; comparable with bench.py's isolated HIP-event pass)", a1, wall1),
                                 ("depth 3 (bench.py default)", a3, wall3)):
            tot = sum(v[1] for v in agg.values())
            f.write(f"## kernel time, {title}\n\n| family | launches/step | avg us/launch | ms/step | % GPU time |\n|---|---|---|---|---|\n")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"| {k} | {v[0]/nf:.1f} | {v[1]/v[0]/1e3:.2f} | {v[1]/nf/1e6:.4f} | {100*v[1]/tot:.1f} |\n")
            f.write(f"\nSum of kernel durations per step: {tot/nf/1e6:.3f} ms; trace wall per step: {wall:.3f} ms.\n\n")
        rf = bench.get("roofline") or {}
        dom = a1.get(rf.get("kernel"))
        if dom:
            f.write(f"bench.py HIP-event figure for the dominant family ({rf['kernel']}): {rf['avg_launch_us']:.1f} us/launch over "
                    f"{rf['launches_per_step']} launches; rocprofv3 (depth 1): {dom[1]/dom[0]/1e3:.1f} us/launch.\n")
        f.write("\n## HBM-side traffic of ONE forward (separate --pmc passes at depth 1; FETCH_SIZE x2 per the gfx950 note of "
                "MI355X_MICROARCH.md, WRITE_SIZE as is; KB -> bytes x1024)\n\n| family | launches | fetch MB | write MB | total MB |\n|---|---|---|---|---|\n")
        t2 = 0
        for k, (n, fv, wv) in sorted(tr_agg.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2])):
            hb = (2 * fv + wv) * 1024
            t2 += hb
            f.write(f"| {k} | {n} | {2*fv*1024/1e6:.1f} | {wv*1024/1e6:.1f} | {hb/1e6:.1f} |\n")
        f.write(f"\nTotal {t2/1e6:.0f} MB per 64-frame forward = {t2/64/1e6:.1f} MB/frame, against 2450 MB (38.3 MB/frame) of "
                "compulsory traffic of the unfused graph.\n")
    json.dump({"plan": "profiles/plan_lite0.b64.f0", "batch": 64,
               "families": {k: {"launches": v[0], "hbm_bytes_per_launch": (2 * v[1] + v[2]) * 1024 / v[0]} for k, v in tr_agg.items() if v[0]}},
              open(f"profiles/{tag}_traffic.json", "w"), indent=1)
    print(open(f"profiles/{tag}_summary.md").read())


if __name__ == "__main__":
    main()
