"""Sanitizer run of the library's file parsers (SURVEY.md section 5 "sanitizers": ASan/UBSan on the host code; GPU ASan is not
available).  vbt_amd/csrc/container_parse.h - the reader + validator vbt_model_create uses and the plan-file reader - is built
host-only with gcc -fsanitize=address,undefined (tests/fuzz/parse_fuzz.cc) and fed truncated and bit-flipped .vbtm / VBTPLAN2
files: every one must come back as "ok" or "refused: <reason>", none may crash, throw or trip a sanitizer."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fuzz") / "parse_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "fuzz", "parse_fuzz.cc"), "-o", exe])
    return exe


def _run(cmd):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run(cmd, capture_output=True, text=True, errors="replace", env=env, timeout=600)   # (refusal texts may quote mangled bytes)
    assert p.returncode == 0 and "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln]
    assert all(ln.startswith(("ok ", "refused ")) for ln in lines)
    return lines


def test_truncated_and_bit_flipped_containers_are_refused_not_crashed(harness, model_path, tmp_path):
    raw = np.fromfile(model_path, dtype=np.uint8)
    hdr = np.frombuffer(raw[:128].tobytes(), dtype="<i4")
    nt, no = int(hdr[4]), int(hdr[5])
    recs = 128 + 32 * nt + 160 * no                       # header + tensor + op tables: where every index and offset lives
    rng = np.random.default_rng(11)
    d = tmp_path / "c"
    d.mkdir()
    raw.tofile(d / "000_intact.vbtm")
    n = 0
    for cut in [0, 7, 8, 127, 128, 129, 128 + 32 * nt - 1, recs - 1, recs, recs + 5, len(raw) // 2, len(raw) - 1]:
        raw[:cut].tofile(d / f"t{n:03d}.vbtm")
        n += 1
    for i in range(300):                                  # single bit flips in the record tables (a few in the blob)
        m = raw.copy()
        pos = int(rng.integers(0, recs)) if i < 270 else int(rng.integers(recs, len(raw)))
        m[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        m.tofile(d / f"b{i:03d}.vbtm")
    for i in range(60):                                   # a whole 32-bit field replaced by an extreme value
        m = raw.copy()
        pos = int(rng.integers(2, recs // 4)) * 4
        m[pos:pos + 4] = np.frombuffer(np.array([rng.choice([-1, 0x7fffffff, -0x80000000, 1 << 20, 65536, 255])], "<i4").tobytes(), np.uint8)
        m.tofile(d / f"w{i:03d}.vbtm")
    lines = _run([harness, "container", str(d)])
    assert len(lines) == 1 + 12 + 300 + 60
    verdict = {os.path.basename(ln.split()[1].rstrip(":")): ln.split()[0] for ln in lines}
    assert verdict["000_intact.vbtm"] == "ok"
    assert all(verdict[f"t{k:03d}.vbtm"] == "refused" for k in range(12))
    refused = sum(v == "refused" for v in verdict.values())
    assert refused > 80, refused                           # flips in scales, weights, in-range offsets, informational fields give legal (if different) models


def test_mangled_plan_files_are_refused_not_crashed(harness, tmp_path):
    good = os.path.join(ROOT, "profiles", "plan_lite0.b64.f0")
    txt = open(good).read()
    rng = np.random.default_rng(5)
    d = tmp_path / "p"
    d.mkdir()
    (d / "000_intact").write_text(txt)
    for i, cut in enumerate([0, 3, 8, 9, 12, 40, len(txt) // 2, len(txt) - 2]):
        (d / f"t{i:02d}").write_text(txt[:cut])
    raw = np.frombuffer(txt.encode(), np.uint8)
    for i in range(200):
        m = raw.copy()
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, len(m)))] = np.uint8(rng.choice([0, 10, 32, 45, 48, 57, 58, 65, 255]))
        (d / f"b{i:03d}").write_bytes(m.tobytes())
    (d / "huge_numbers").write_text(txt.replace(":25", ":99999999999999999999").replace("VBTPLAN2 53", "VBTPLAN2 53"))
    (d / "negative").write_text(txt.replace("2 1 fused_stem_block:-1", "-7 1 fused_stem_block:-1"))
    (d / "other_family").write_text(txt.replace("decode_nms", "something_else"))
    (d / "unknown_variant").write_text(txt.replace("fused_mbconv:9", "fused_mbconv:77"))
    lines = _run([harness, "plan", good, str(d)])
    verdict = {os.path.basename(ln.split()[1].rstrip(":")): ln.split()[0] for ln in lines}
    assert verdict["000_intact"] == "ok"
    for k in ("huge_numbers", "negative", "other_family", "unknown_variant", "t00", "t03", "t06"):
        assert verdict[k] == "refused", k
