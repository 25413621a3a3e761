"""Build the in-tree HIP extension (libvbt_hip.so) for gfx950 with hipcc.

No torch.utils.cpp_extension, no JIT cache: the .so lives next to the package so that it travels
to the GPU box with the snapshot (and is visible to the driver's "which .so was loaded" check).
Every .hip translation unit is compiled to its own object (in parallel, only when it or a header
changed) and the objects are linked into the one shared library.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libvbt_hip.so")
FLAGS = (["-DFB_MINW=" + os.environ["VBT_FB_MINW"]] if os.environ.get("VBT_FB_MINW") else []) + os.environ.get("VBT_EXTRA_CXXFLAGS", "").split() + ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-pass-failed",
         # MFMA results land in VGPRs (gfx90a+ unified register file): the requantisation epilogues read the
         # accumulators directly instead of through one v_accvgpr_read per element
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include")) if f.endswith(".h")]


# Kernel headers whose change cannot affect a translation unit's code (launchers.h shows every argument struct to every unit, but
# a unit only instantiates the kernels of its own headers).  Anything not listed here is a dependency of every unit.
KERNEL_HEADERS = {"op_kernels.h", "band_block.h", "expdw_block.h", "expdw2_block.h", "stem_block.h", "image_block.h", "fused_block.h"}
UNIT_KERNEL_HEADERS = {
    "k_band.hip": {"band_block.h", "expdw_block.h", "expdw2_block.h", "stem_block.h", "fused_block.h"},
    "k_image.hip": {"image_block.h", "fused_block.h"},
    "k_fused_mbconv.hip": {"fused_block.h"},
    "k_fused_sepconv.hip": {"fused_block.h"},
    "tracker.hip": set(),
    "frames.hip": set(),
}


def _unit_headers(unit, hdrs):
    mine = UNIT_KERNEL_HEADERS.get(unit)
    if mine is None:            # the planner (detector.hip) and anything new: every header
        return hdrs
    return [h for h in hdrs if os.path.basename(h) not in KERNEL_HEADERS or os.path.basename(h) in mine]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, s) for s in sources()] + _headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    flag_tag = os.path.join(OBJ, "flags.txt")                    # objects built with other flags are stale too
    flags_txt = " ".join(FLAGS)
    if not os.path.exists(flag_tag) or open(flag_tag).read() != flags_txt:
        force = True
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + _unit_headers(s, hdrs)):
            jobs.append([hipcc] + FLAGS + ["-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), int(os.environ.get("VBT_BUILD_JOBS", "6"))))) as ex:
        list(ex.map(run, jobs))
    with open(flag_tag, "w") as f:
        f.write(flags_txt)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [os.path.join(OBJ, s[:-4] + ".o") for s in sources()])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
