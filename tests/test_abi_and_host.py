"""CPU-only checks: the C-ABI library builds, loads and exports every symbol that
include/densecap.h declares; host-side logic (decodeSequence, synthetic weights) behaves."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="densecap.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", src)))


DEBUG_SYMBOLS = ["dc_debug_fetch", "dc_debug_plan_gemm", "dc_debug_set", "dc_mfma_profile"]


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "densecap_amd", "lib", "libdensecap_hip.so")):
        g.build()
    from densecap_amd import _lib
    lib = _lib.lib()
    boundary, debug = _declared_symbols(), _declared_symbols("densecap_debug.h")
    assert len(boundary) >= 30
    # the boundary header holds only what a reference maintainer binds; measurement / test hooks live in densecap_debug.h
    assert debug == DEBUG_SYMBOLS and not set(boundary) & set(debug)
    assert not [n for n in boundary if "debug" in n or "profile" in n]
    for name in boundary + debug:
        assert hasattr(lib, name), "missing export %s" % name
    assert sorted(_lib.EXPORTED_SYMBOLS) == sorted(boundary + debug)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from densecap_amd import Context
    from densecap_amd._lib import DenseCapError
    with pytest.raises(DenseCapError, match="no CPU fallback"):
        Context(0)


def test_product_path_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "densecap_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt, f


def test_synthetic_weights_shapes_and_determinism():
    from densecap_amd.weights import make_synthetic_weights
    a = make_synthetic_weights(seed=7, vocab_size=50, seq_length=6)
    b = make_synthetic_weights(seed=7, vocab_size=50, seq_length=6)
    assert a["fc6_w"].shape == (4096, 25088) and a["lstm_w"].shape == (1024, 2048)
    assert a["lm_emb"].shape == (52, 512) and a["lm_out_w"].shape == (51, 512)
    assert a["rpn_box_w"].shape == (48, 256, 1, 1) and a["rpn_score_w"].shape == (24, 256, 1, 1)
    assert all(np.array_equal(x.numpy(), y.numpy()) for x, y in zip(a["conv_w"], b["conv_w"]))


def test_oracle_forward_tiny_runs():
    # the oracle itself end to end on a tiny image (shape / ordering contract of forward_test)
    from densecap_amd.weights import make_synthetic_weights, make_synthetic_image
    from oracle import densecap_oracle as O
    W = make_synthetic_weights(seed=1, vocab_size=40, seq_length=5)
    img = make_synthetic_image(96, 128, 0)
    st = {}
    boxes, scores, seq = O.forward_test(img, W, 0.7, 0.3, 20, 5, stages=st)
    assert st["feat"].shape == (512, 6, 8)
    assert boxes.shape[1] == 4 and seq.shape[1] == 5 and len(boxes) == len(scores) == len(seq) > 0
    assert (np.diff(scores) <= 0).all()
    assert seq.min() >= 1 and seq.max() <= 41
    caps = O.decode_sequence(seq, W["idx_to_token"], 40)
    assert len(caps) == len(boxes)


def _build_c_harness(tmp_path):
    import subprocess
    exe = str(tmp_path / "dc_harness")
    lib = os.path.join(ROOT, "densecap_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "dc_harness.c"), "-o", exe, "-L" + lib, "-ldensecap_hip",
                           "-Wl,-rpath," + lib, "-lm"])
    return exe


def test_c_harness_links_against_the_header_and_fails_loudly_without_gpu(tmp_path):
    """A plain-C program (no Python, no torch types) builds against include/densecap.h and the .so; on a box
    without a HIP device dc_create reports the error through the return code (no abort, no CPU fallback)."""
    import subprocess
    import torch
    exe = _build_c_harness(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_c_harness_runs_on_gpu")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1
    assert "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_c_harness_runs_on_gpu(tmp_path):
    import subprocess
    exe = _build_c_harness(tmp_path)
    p = subprocess.run([exe, "224", "288", "100", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "HARNESS OK" in p.stdout and "expected error before load_weights" in p.stdout


def _prototypes(text):
    """{name: normalised 'ret name(args)'} of the dc_* prototypes in a C declaration block."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^[ \t]*#.*$", "", text, flags=re.M)          # preprocessor lines are not part of a prototype
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(dc_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        norm = lambda t: re.sub(r"\s*\*\s*", "* ", re.sub(r"\s+", " ", t)).strip()
        out[name] = "%s %s(%s)" % (norm(ret), name, norm(args))
    return out


def test_lua_ffi_cdef_matches_the_header():
    """lua/densecap_hip.lua cannot be executed here (no LuaJIT): at least every prototype in its ffi.cdef must be
    the header's, character for character after whitespace normalisation, and the structs must list the same fields."""
    hdr = open(os.path.join(ROOT, "include", "densecap.h")).read()
    lua = open(os.path.join(ROOT, "lua", "densecap_hip.lua")).read()
    cdef = re.search(r"ffi\.cdef\[\[(.*?)\]\]", lua, flags=re.S).group(1)
    hp, lp = _prototypes(hdr), _prototypes(cdef)
    assert len(lp) >= 10
    assert not set(lp) & set(DEBUG_SYMBOLS), "the LuaJIT binding must not bind the debug hooks"
    for name, proto in lp.items():
        assert name in hp, "%s is not declared in densecap.h" % name
        assert proto == hp[name], "cdef drifted:\n  lua: %s\n  hdr: %s" % (proto, hp[name])
    # struct fields (order matters for the ABI)
    def fields(text, struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), re.sub(r"/\*.*?\*/", "", text, flags=re.S),
                         flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.sub(r"\[.*?\]", "", first.split()[-1].lstrip("*")))
            names += [re.sub(r"\[.*?\]", "", r.strip().lstrip("*")) for r in rest]
        return names
    for st in ("dc_weights", "dc_result"):
        assert fields(cdef, st) == fields(hdr, st), st


def test_lua_run_model_substitutions_match_the_reference():
    """lua/run_model_hip.lua runs the reference's own run_model.lua with two statements replaced (so every flag of
    run_model.lua:26-61 is kept without restating the script).  No Lua here: check that each Lua pattern, read as the
    literal it escapes, occurs exactly once in the reference script -- where the reference tree is present."""
    ref = "/root/reference/run_model.lua"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this box")
    lua = open(os.path.join(ROOT, "lua", "run_model_hip.lua")).read()
    pats = re.findall(r'\{"((?:[^"\\]|\\.)*)",\s*\n\s*"((?:[^"\\]|\\.)*)"\}', lua)
    assert len(pats) == 2
    text = open(ref).read()
    for pat, repl in pats:
        literal = re.sub(r"%(.)", r"\1", pat)          # Lua escapes a magic character with %
        assert text.count(literal) == 1, literal
        assert "DenseCapModelHIP" in repl or "torch.FloatTensor" in repl
    # the flags a user expects are the reference's own (they are never restated in our file)
    for flag in ("-input_dir", "-max_images", "-output_vis_dir", "-input_split"):
        assert flag in text and ("'%s'" % flag) not in lua


def test_run_model_cli_accepts_every_reference_flag(tmp_path):
    """Round-4 verdict: eight of run_model.lua's flags (run_model.lua:26-61) were missing from the executed host, so a caller
    passing `-use_cudnn 1` got an argparse error.  Every `cmd:option` of the reference script must parse, with the
    reference's default."""
    from densecap_amd import run_model as R
    ref = "/root/reference/run_model.lua"
    flags = {"-checkpoint": "data/models/densecap/densecap-pretrained-vgg16.t7", "-image_size": 720, "-rpn_nms_thresh": 0.7,
             "-final_nms_thresh": 0.3, "-num_proposals": 1000, "-input_image": "", "-input_dir": "", "-input_split": "",
             "-splits_json": "info/densecap_splits.json", "-vg_img_root_dir": "", "-max_images": 100, "-output_dir": "",
             "-num_to_draw": 10, "-text_size": 2, "-box_width": 2, "-output_vis": 1, "-output_vis_dir": "vis/data", "-gpu": 0,
             "-use_cudnn": 1}
    if os.path.exists(ref):            # the list above IS the reference's (checked where the reference tree is present)
        src = open(ref).read()
        found = dict(re.findall(r"cmd:option\('(-\w+)',\s*\n?\s*('[^']*'|[\d.]+)", src))
        assert set(found) == set(flags), set(found) ^ set(flags)
        for k, v in found.items():
            assert str(flags[k]) == v.strip("'"), (k, v, flags[k])
    opt = R.build_parser().parse_args([])
    for k, v in flags.items():
        assert getattr(opt, k[1:]) == v, (k, getattr(opt, k[1:]), v)
    argv = []
    for k, v in flags.items():
        argv += [k, str(v if v != "" else "x")]
    R.build_parser().parse_args(argv)                      # every flag is accepted with a value
    # -input_split: ids of the split -> <vg_img_root_dir>/<id>.jpg (run_model.lua:128-137)
    sj = tmp_path / "splits.json"
    sj.write_text('{"train": [1, 2], "val": [7, 9, 11], "test": []}')
    opt = R.build_parser().parse_args(["-input_split", "val", "-splits_json", str(sj), "-vg_img_root_dir", "/vg"])
    assert R.get_input_images(opt) == ["/vg/7.jpg", "/vg/9.jpg", "/vg/11.jpg"]
    with pytest.raises(SystemExit):
        R.get_input_images(R.build_parser().parse_args([]))


def test_extract_features_and_daemon_clis_accept_every_reference_flag():
    """The other two scripts on the boundary (extract_features.lua:14-27, webcam/daemon.lua:14-26): every `cmd:option` parses,
    with the reference's default."""
    from densecap_amd import daemon as D, extract_features as E
    cases = (
        (E, "/root/reference/extract_features.lua",
         {"-checkpoint": "data/models/densecap/densecap-pretrained-vgg16.t7", "-image_size": 720, "-rpn_nms_thresh": 0.7,
          "-final_nms_thresh": 0.4, "-num_proposals": 1000, "-boxes_per_image": 100, "-input_txt": "", "-max_images": 0,
          "-output_h5": "", "-gpu": 0, "-use_cudnn": 1}),
        (D, "/root/reference/webcam/daemon.lua",
         {"-checkpoint": "data/models/densecap/densecap-pretrained-vgg16.t7", "-max_image_size": 720, "-input_dir": "webcam/inputs",
          "-input_ext": ".jpg", "-output_dir": "webcam/outputs", "-timing": 0, "-rpn_nms_thresh": 0.7, "-final_nms_thresh": 0.3,
          "-num_proposals": 1000, "-gpu": 0, "-use_cudnn": 1}))
    for mod, ref, flags in cases:
        if os.path.exists(ref):        # the lists above ARE the reference's (checked where the reference tree is present)
            found = dict(re.findall(r"cmd:option\('(-\w+)',\s*\n?\s*('[^']*'|[\d.]+)", open(ref).read()))
            assert set(found) == set(flags), (ref, set(found) ^ set(flags))
            for k, v in found.items():
                assert str(flags[k]) == v.strip("'"), (ref, k, v, flags[k])
        opt = mod.build_parser().parse_args([])
        for k, v in flags.items():
            assert getattr(opt, k[1:]) == v, (mod.__name__, k, getattr(opt, k[1:]), v)
        argv = []
        for k, v in flags.items():
            argv += [k, str(v if v != "" else "x")]
        mod.build_parser().parse_args(argv)


def test_oracle_transcendentals_are_double_then_cast():
    """docs/SEMANTICS.md (round 5): exp / sigmoid / tanh of a FloatTensor as TH computes them -- the C double function, result
    cast to float.  The float result is the correctly rounded one wherever the double is not within 2^-29 of a boundary; overflow
    sits where the FLOAT result overflows."""
    import math
    import torch
    from oracle import densecap_oracle as O
    x = np.array([-120, -88.8, -1.5, -1e-8, 0.0, 1e-8, 0.3, 1.0, 17.25, 88.72, 88.73, 120.0, np.inf, -np.inf], np.float32)
    got = O.th_exp(x)
    for xi, gi in zip(x, got):
        try:
            want = np.float32(math.exp(float(xi)))
        except OverflowError:
            want = np.float32(np.inf)
        assert gi == want or (np.isinf(gi) and np.isinf(want)), (xi, gi, want)
    assert np.isinf(O.th_exp(np.float32(88.73))) and np.isfinite(O.th_exp(np.float32(88.72)))
    assert np.isnan(O.th_exp(np.array([np.nan], np.float32)))[0]
    t = torch.tensor([-30.0, -2.5, 0.0, 0.1, 4.0, 30.0])
    np.testing.assert_array_equal(O.th_sigmoid(t).numpy(), np.array([1.0 / (1.0 + math.exp(-v)) for v in t.tolist()], np.float32))
    np.testing.assert_array_equal(O.th_tanh(t).numpy(), np.array([math.tanh(v) for v in t.tolist()], np.float32))


def test_oracle_forward_vector_pass_nms_equals_the_c_nms():
    """cpu_baseline times box_utils.nms in the reference's vector-pass-per-pick form (nms_impl = "vector"); the parity tests use
    the early-out C restatement.  Same picks, so the same forward_test outputs (small image, small language model)."""
    from densecap_amd.weights import make_synthetic_image, make_synthetic_weights
    from oracle import densecap_oracle as O
    W = make_synthetic_weights(seed=5, vocab_size=40, seq_length=4)
    img = make_synthetic_image(96, 128, 1)
    a = O.forward_test(img, W, 0.7, 0.3, 30, 4, nms_impl="c")
    b = O.forward_test(img, W, 0.7, 0.3, 30, 4, nms_impl="vector")
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    assert len(a[0]) > 0
