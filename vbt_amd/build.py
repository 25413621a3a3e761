"""Build the in-tree HIP extension (libvbt_hip.so) for gfx950 with hipcc.

No torch.utils.cpp_extension, no JIT cache: the .so lives next to the package so that it travels
to the GPU box with the snapshot (and is visible to the driver's "which .so was loaded" check).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvbt_hip.so")
SOURCES = ["detector.hip", "tracker.hip"]
FLAGS = (["-DFB_MINW=" + os.environ["VBT_FB_MINW"]] if os.environ.get("VBT_FB_MINW") else []) + os.environ.get("VBT_EXTRA_CXXFLAGS", "").split() + ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-pass-failed",
         # MFMA results land in VGPRs (gfx90a+ unified register file): the requantisation epilogues read the
         # accumulators directly instead of through one v_accvgpr_read per element
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vbt_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
