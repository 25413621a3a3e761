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


def _dep_file(obj):
    return obj[:-2] + ".d"


def _deps(obj, src):
    """Prerequisites of an object as the compiler saw them (hipcc -MD -MF obj/x.d): the source and every header it pulled in,
    directly or not.  No depfile yet (first build, or an object of an older build.py): every header of the tree."""
    d = _dep_file(obj)
    if not os.path.exists(d):
        return [src] + _headers()
    txt = open(d).read().replace("\\\n", " ")
    deps = []
    for rule in txt.split("\n"):
        if ":" in rule:
            deps += rule.split(":", 1)[1].split()
    return [p for p in deps if p.startswith(os.path.dirname(HERE))] or [src] + _headers()     # (system headers: covered by the compiler version tag)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    # a prerequisite that no longer exists (a header deleted or renamed, still listed in the depfile) makes the target stale too
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, s) for s in sources()] + _headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    flag_tag = os.path.join(OBJ, "flags.txt")                    # objects built with other flags or another compiler are stale too
    try:
        cc_version = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().replace("\n", " | ")
    except OSError:
        cc_version = "unknown"
    flags_txt = " ".join(FLAGS) + "\n" + cc_version
    if not os.path.exists(flag_tag) or open(flag_tag).read() != flags_txt:
        force = True
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, _deps(obj, src)):
            jobs.append([hipcc] + FLAGS + ["-MD", "-MF", _dep_file(obj), "-c", "-o", obj, src])

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
