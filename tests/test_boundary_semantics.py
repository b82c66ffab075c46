"""Method BEHAVIOUR of the host shims against the reference's DenseCapModel / LocalizationLayer (CPU only).

test_abi_and_host.py pins prototypes and struct fields; this file pins what the methods DO (round-3 verdict: the one
parity-class defect was `setTestArgs` merging into a persistent table with a default of 300 proposals):

  DenseCapModel:setTestArgs      densecap/DenseCapModel.lua:185-191   every call re-derives rpn 0.7 / num_proposals 1000 /
                                                                      final 0.3 for absent keys; unknown keys ignored
  LocalizationLayer:setTestArgs  densecap/LocalizationLayer.lua:233-238   clip_boxes true / nms_thresh 0.7 / max_proposals 300
  callers                        evaluate_model.lua:39-43 (`max_proposals=` key), run_model.lua:149-153, train.lua:139-143

The Python shim runs against a recording stand-in for the C library (no GPU); the Lua shim cannot run here (no LuaJIT), so
its text is checked for the same constants and control flow.
"""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _RecordingLib:
    """Stands in for libdensecap_hip.so: every dc_* call is recorded and succeeds."""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("dc_"):
            raise AttributeError(name)

        def fn(*args):
            self.calls.append((name,) + tuple(a for a in args[1:] if isinstance(a, (int, float))))
            return 0
        return fn

    def last(self, name):
        hits = [c for c in self.calls if c[0] == name]
        return hits[-1] if hits else None


class _Ctx:
    def __init__(self):
        self.lib = _RecordingLib()
        self.h = None


def _model(test_args=None):
    from densecap_amd.model import DenseCapModel
    from densecap_amd.weights import make_synthetic_weights
    W = make_synthetic_weights(seed=3, vocab_size=30, seq_length=5, fc_dim=256)
    if test_args is not None:
        W["test_args"] = test_args
    ctx = _Ctx()
    return DenseCapModel(W, ctx=ctx), ctx.lib


def _state(lib):
    """(clip, rpn_thresh, final_thresh, num_proposals) as the library holds it after the recorded calls."""
    clip = rpn = fin = P = None
    for c in lib.calls:
        if c[0] == "dc_set_test_args":
            clip, rpn, fin, P = 1, c[1], c[2], c[3]
        elif c[0] == "dc_set_localization_test_args":
            clip, rpn, P = c[1], c[2], c[3]
    return clip, rpn, fin, P


def test_fresh_model_runs_with_the_constructor_defaults():
    # LocalizationLayer.lua:155 (its own setTestArgs(): true / 0.7 / 300) and DenseCapModel.lua:31 (0.3)
    m, lib = _model()
    assert _state(lib) == (1, pytest.approx(0.7), pytest.approx(0.3), 300)


def test_evaluate_model_call_gets_1000_proposals():
    """evaluate_model.lua:39-43 passes `max_proposals=opt.num_proposals`; DenseCapModel:setTestArgs reads `num_proposals`,
    so the reference runs that script with its default of 1000 whatever the flag says -- and must not raise."""
    m, lib = _model()
    m.setTestArgs(dict(rpn_nms_thresh=0.7, final_nms_thresh=0.3, max_proposals=300))
    assert _state(lib) == (1, pytest.approx(0.7), pytest.approx(0.3), 1000)
    assert m._capacity(600, 720) == 1000


def test_every_call_resets_the_keys_it_does_not_name():
    m, lib = _model()
    m.setTestArgs(rpn_nms_thresh=0.5, final_nms_thresh=0.4, num_proposals=50)
    assert _state(lib) == (1, pytest.approx(0.5), pytest.approx(0.4), 50)
    m.setTestArgs(num_proposals=70)                       # the other two go back to 0.7 / 0.3, they are not retained
    assert _state(lib) == (1, pytest.approx(0.7), pytest.approx(0.3), 70)
    m.setTestArgs()                                       # nil table: all defaults, 1000 proposals
    assert _state(lib) == (1, pytest.approx(0.7), pytest.approx(0.3), 1000)
    m.setTestArgs({"no_such_key": 1})                     # unknown keys are ignored
    assert _state(lib) == (1, pytest.approx(0.7), pytest.approx(0.3), 1000)


def test_checkpoint_object_state_is_the_initial_state():
    # train.lua:139-143 leaves 0.7 / 1000 / 0.3 in the objects it saves; whatever is stored is what a loaded model runs with
    m, lib = _model(dict(test_clip_boxes=False, test_nms_thresh=0.65, test_max_proposals=77, final_nms_thresh=0.45))
    assert _state(lib) == (0, pytest.approx(0.65), pytest.approx(0.45), 77)
    ll = m.nets.localization_layer
    assert (ll.test_clip_boxes, ll.test_nms_thresh, ll.test_max_proposals) == (False, 0.65, 77)
    assert m.opt["final_nms_thresh"] == 0.45
    m.setTestArgs(num_proposals=10)                       # the model's setTestArgs turns clipping back on (no clip_boxes key)
    assert _state(lib) == (1, pytest.approx(0.7), pytest.approx(0.3), 10)


def test_direct_field_writes_travel_at_forward_time():
    """train.lua:139-143 calls the LAYER's setTestArgs and writes model.opt.final_nms_thresh directly; the reference reads
    both when forward runs (LocalizationLayer.lua:250-256, DenseCapModel.lua:261)."""
    m, lib = _model()
    m.nets.localization_layer.setTestArgs(clip_boxes=False, nms_thresh=0.6, max_proposals=40)
    m.opt["final_nms_thresh"] = -1.0
    n0 = len(lib.calls)
    m.forward_raw(np.zeros((3, 64, 96), np.float32))
    names = [c[0] for c in lib.calls[n0:]]
    assert names.index("dc_set_localization_test_args") < names.index("dc_forward_test")
    assert _state(lib) == (0, pytest.approx(0.6), pytest.approx(-1.0), 40)
    # the layer's setTestArgs re-derives its three fields too: clip_boxes defaults back to true, max_proposals to 300
    m.nets.localization_layer.setTestArgs(nms_thresh=0.55)
    m.extractFeatures(np.zeros((3, 64, 96), np.float32))
    assert _state(lib) == (1, pytest.approx(0.55), pytest.approx(-1.0), 300)


def test_test_args_survive_the_t7_round_trip(tmp_path):
    from densecap_amd import t7
    from densecap_amd.weights import make_synthetic_weights
    from tests.golden.t7_assembler import assemble_densecap_checkpoint
    W = make_synthetic_weights(seed=5, vocab_size=40, seq_length=6, fc_dim=256)
    p = tmp_path / "a.t7"
    assemble_densecap_checkpoint(str(p), W)               # fields as train.lua:139-143 leaves them
    ta = t7.weights_from_checkpoint(t7.load(str(p)))["test_args"]
    assert ta == dict(test_clip_boxes=True, test_nms_thresh=0.7, test_max_proposals=1000, final_nms_thresh=0.3)
    assemble_densecap_checkpoint(str(p), W, test_args=dict(test_clip_boxes=False, test_nms_thresh=0.65,
                                                           test_max_proposals=77, final_nms_thresh=0.45))
    ta = t7.weights_from_checkpoint(t7.load(str(p)))["test_args"]
    assert ta == dict(test_clip_boxes=False, test_nms_thresh=0.65, test_max_proposals=77, final_nms_thresh=0.45)
    # the product's own writer keeps them as well
    q = tmp_path / "b.t7"
    W["test_args"] = ta
    t7.save(str(q), t7.checkpoint_from_weights(W))
    assert t7.weights_from_checkpoint(t7.load(str(q)))["test_args"] == ta


def test_a_refused_setTestArgs_leaves_the_previous_state_usable():
    """Advisor finding (round 4): setTestArgs wrote num_proposals into the object BEFORE the ABI validated it; a refused value
    (0, 2000000) then made every later forward re-raise in _push_test_args.  The previous values come back on failure."""
    from densecap_amd._lib import DenseCapError
    m, lib = _model()
    m.setTestArgs(rpn_nms_thresh=0.6, final_nms_thresh=0.2, num_proposals=77)
    before = _state(lib)

    class _Refusing(_RecordingLib):
        def __getattr__(self, name):
            fn = _RecordingLib.__getattr__(self, name)
            if name == "dc_set_test_args":
                return lambda *a: -5
            if name == "dc_last_error":
                return lambda *a: b"num_proposals must be -1 (uncapped) or in [1,1048576] (got 0)"
            return fn
    good = m.lib
    m.lib = m.ctx.lib = _Refusing()
    with pytest.raises(DenseCapError):
        m.setTestArgs(num_proposals=0)
    m.lib = m.ctx.lib = good
    ll = m.nets.localization_layer
    assert (ll.test_nms_thresh, ll.test_max_proposals, m.opt["final_nms_thresh"]) == (0.6, 77, 0.2)
    m._push_test_args()                                   # what every forward does first: must not raise
    assert _state(lib) == before


# ---- the Lua twin (text checks: no LuaJIT in this image) -----------------------------------------------------------------
def _lua():
    return open(os.path.join(ROOT, "lua", "DenseCapModelHIP.lua")).read()


def _lua_function(src, header):
    """Body of the Lua function that starts with `header` up to its closing `end` at the header's indentation."""
    i = src.index(header)
    indent = re.match(r"[ ]*", src[src.rfind("\n", 0, i) + 1:]).group(0)
    m = re.search(r"\n%send\b" % indent, src[i:])
    return src[i:i + m.start()]


def test_lua_setTestArgs_has_the_reference_defaults_and_no_persistent_merge():
    src = _lua()
    body = _lua_function(src, "function Model:setTestArgs(kwargs)")
    assert re.search(r"nms_thresh\s*=\s*getopt\(kwargs,\s*'rpn_nms_thresh',\s*0\.7\)", body)
    assert re.search(r"max_proposals\s*=\s*getopt\(kwargs,\s*'num_proposals',\s*1000\)", body)
    assert re.search(r"final_nms_thresh\s*=\s*getopt\(kwargs,\s*'final_nms_thresh',\s*0\.3\)", body)
    assert "pairs(kwargs" not in body                       # round 3 merged the table into persistent state
    call = re.search(r"localization_layer:setTestArgs\{(.*?)\}", body, flags=re.S)
    assert call and "clip_boxes" not in call.group(1)      # no clip_boxes key: the layer's default (true) comes back
    # a refused value does not stay behind (advisor finding, round 4): the previous state is restored before the error travels on
    assert "pcall(" in body and "unpack(saved)" in body and "error(err" in body
    layer = _lua_function(src, "function ll.setTestArgs(layer, args)")
    assert re.search(r"getopt\(args,\s*'clip_boxes',\s*true\)", layer)
    assert re.search(r"getopt\(args,\s*'nms_thresh',\s*0\.7\)", layer)
    assert re.search(r"getopt\(args,\s*'max_proposals',\s*300\)", layer)
    # the reference's own text has the same constants (where the reference tree is present)
    ref = "/root/reference/densecap/DenseCapModel.lua"
    if os.path.exists(ref):
        rbody = _lua_function(open(ref).read(), "function DenseCapModel:setTestArgs(kwargs)")
        for pat in (r"'rpn_nms_thresh',\s*0\.7", r"'num_proposals',\s*1000", r"'final_nms_thresh',\s*0\.3"):
            assert re.search(pat, rbody) and re.search(pat, body)
        rl = _lua_function(open("/root/reference/densecap/LocalizationLayer.lua").read(), "function layer:setTestArgs(args)")
        for pat in (r"'clip_boxes',\s*true", r"'nms_thresh',\s*0\.7", r"'max_proposals',\s*300"):
            assert re.search(pat, rl) and re.search(pat, layer)


def test_lua_initial_state_comes_from_the_checkpoint_object_and_travels_before_forward():
    src = _lua()
    ctor = _lua_function(src, "function Model.fromCheckpoint(ref, gpu)")
    for f in ("test_clip_boxes", "test_nms_thresh", "test_max_proposals"):
        assert re.search(r"if rll\.%s ~= nil then ll\.%s = rll\.%s end" % (f, f, f), ctor), f
    assert "local rll = ref.nets.localization_layer" in ctor
    assert re.search(r"final_nms_thresh\s*=\s*getopt\(ref\.opt,\s*'final_nms_thresh',\s*0\.3\)", ctor)
    assert "num_proposals = 300" not in src
    push = _lua_function(src, "function Model:_push_test_args()")
    assert "C.dc_set_test_args(self.ctx, ll.test_nms_thresh, self.opt.final_nms_thresh" in push
    assert "if not ll.test_clip_boxes then" in push and "C.dc_set_localization_test_args(self.ctx, 0," in push
    for fn in ("function Model:forward_test(input)", "function Model:extractFeatures(input)",
               "function Model:forward_raw(input)"):
        body = _lua_function(src, fn)
        assert body.index("self:_push_test_args()") < body.index("C.dc_")
    # capacity follows the layer's field, as LocalizationLayer.lua:322-324 reads it
    assert "self.nets.localization_layer.test_max_proposals" in _lua_function(src, "function Model:_capacity(H, W)")
    # getopt itself is utils.getopt (densecap/utils.lua:67-75): nil -> default, `false` is kept
    g = _lua_function(src, "local function getopt(opt, key, default_value)")
    assert "if v == nil then v = default_value end" in g
