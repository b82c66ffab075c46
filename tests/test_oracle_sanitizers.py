"""The C part of the oracle under AddressSanitizer + UBSan (sanitizers on the host-side checker).

oracle/oracle_c.c is compiled, with a small stdin driver (tests/golden/oracle_c_sanitize.c), by gcc with
-fsanitize=address,undefined -fno-sanitize-recover; hostile and boundary cases (no boxes, one box, NaN / inf scores,
exact ties, caps of 0 / 1 / n, boxes far outside the image, 1x1 feature maps) run through it with exactly-sized
buffers.  The instrumented results must equal the regular build's bit for bit, and no sanitizer report may appear."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sanitized(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("san") / "oracle_c_sanitize")
    cmd = ["gcc", "-O1", "-g", "-ffp-contract=off", "-fno-fast-math", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-o", exe, os.path.join(ROOT, "tests", "golden", "oracle_c_sanitize.c"),
           os.path.join(ROOT, "oracle", "oracle_c.c"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return exe


def _hex(a):
    return " ".join(float(x).hex() for x in np.asarray(a, np.float32).reshape(-1))


def _run(exe, text):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], input=text, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, "sanitizer report or crash:\n" + r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    return r.stdout.strip().splitlines()


def _nms_cases():
    rng = np.random.default_rng(5)
    cases = []
    for n in (0, 1, 2, 63, 64, 65, 300):
        for cap in (-1, 0, 1, 7, 10 ** 6):
            for thr in (0.0, 0.3, 1.0):
                xy = rng.uniform(0, 200, (n, 2)).astype(np.float32)
                wh = rng.uniform(1, 120, (n, 2)).astype(np.float32)
                s = rng.standard_normal(n).astype(np.float32)
                if n >= 4:
                    s[1] = s[0]                                   # exact tie
                    s[2] = np.nan; s[3] = np.inf
                cases.append((np.concatenate([xy, xy + wh, s[:, None]], 1).astype(np.float32), thr, cap))
    # degenerate geometry: zero-area, inverted and coincident boxes
    b = np.array([[5, 5, 5, 5, 1.0], [9, 9, 3, 3, 0.5], [5, 5, 5, 5, 1.0], [0, 0, 1e30, 1e30, 0.2]], np.float32)
    cases.append((b, 0.5, -1))
    return cases


def test_nms_under_sanitizers_equals_the_regular_build(sanitized):
    from oracle import densecap_oracle as O
    cases = _nms_cases()
    text = "".join("nms %d %s %d\n%s\n" % (len(b), float(np.float32(t)).hex(), cap, _hex(b)) for b, t, cap in cases)
    lines = _run(sanitized, text)
    assert len(lines) == len(cases)
    lib = O._clib()
    for (b, t, cap), line in zip(cases, lines):
        got = [int(x) for x in line.split()[1:]]
        n = len(b)
        pick = (C.c_int * max(n, 1))()
        cnt = lib.oracle_nms(np.ascontiguousarray(b).ctypes.data_as(C.POINTER(C.c_float)), n, C.c_float(t), cap, pick) if n else 0
        assert got[0] == cnt and got[1:] == list(pick[:cnt]), (n, t, cap)
        if cap >= 0:
            assert cnt <= cap


def test_roi_pool_under_sanitizers_equals_the_regular_build(sanitized):
    from oracle import densecap_oracle as O
    rng = np.random.default_rng(6)
    cases = []
    for (Cc, h, w, HH, WW) in ((3, 1, 1, 2, 2), (2, 5, 7, 7, 7), (4, 38, 45, 7, 7), (1, 2, 9, 3, 5)):
        feat = rng.standard_normal((Cc, h, w)).astype(np.float32)
        ih, iw = 16 * h, 16 * w
        boxes = np.array([[iw / 2, ih / 2, iw, ih],              # the whole image
                          [1, 1, 1, 1],                           # a one-pixel box in the corner
                          [-500, -500, 40, 40],                   # entirely outside: every tap contributes zero
                          [iw + 300, ih + 300, 1000, 1000],
                          [iw / 2, ih / 2, 1e6, 1e6],             # sampling positions far beyond the map
                          [iw - 1, ih - 1, 3, 3]], np.float32)
        cases.append((feat, boxes, ih, iw, HH, WW))
    text = "".join("roi %d %d %d %d %d %d %d %d\n%s\n%s\n" % (f.shape[0], f.shape[1], f.shape[2], len(b), ih, iw, HH, WW,
                                                               _hex(f), _hex(b)) for f, b, ih, iw, HH, WW in cases)
    lines = _run(sanitized, text)
    assert len(lines) == len(cases)
    for (f, b, ih, iw, HH, WW), line in zip(cases, lines):
        got = np.array([float.fromhex(x) for x in line.split()[1:]], np.float32).reshape(len(b), f.shape[0], HH, WW)
        ref = O.bilinear_roi_pool(f, b, ih, iw, HH, WW)
        np.testing.assert_array_equal(got, np.asarray(ref, np.float32))
