"""Assembles tests/golden/handmade_checkpoint.t7 BYTE BY BYTE from the Torch7 serialization format
(torch7 File.lua: writeObject/readObject), independently of densecap_amd/t7.py's own writer, so that the
reader is tested against bytes it did not produce.  The file is a miniature densecap checkpoint
(train.lua:174-185: {model=nn.DenseCapModel, iter=..}) with every feature a real one has that the
round-trip tests never exercised:

  * serialized Lua closures: TYPE_FUNCTION (6), TYPE_LEGACY_RECUR_FUNCTION (7), TYPE_RECUR_FUNCTION (8)
    [int index][int size][bytecode][upvalue table] (tag 6: no index, File.lua)  -- must be skipped, not fatal;
  * object back-references (a second occurrence of an index carries no body), incl. modules shared between
    `nets.*` and the nn.gModule `recog_net` (DenseCapModel.lua:127-162) whose forwardnodes are graph.Node objects;
  * a legacy nn.SpatialConvolutionMM whose weight is 2-D (nOutputPlane x nInputPlane*kH*kW);
  * a tensor that is a strided, offset VIEW of a larger storage, and two tensors sharing one storage;
  * class names with and without the "V 1" version header; number keys stored as doubles.

Layout of the binary format (little endian):  int32 type tag; NUMBER: float64; STRING: int32 len + bytes;
BOOLEAN: int32; TABLE: int32 index, [int32 n, n x (key object, value object)]; TORCH: int32 index,
[string "V 1", string class, payload]; tensor payload: int32 ndim, int64 sizes, int64 strides, int64 offset(1-based),
storage object; storage payload: int64 n + raw data.  Run:  python tests/golden/make_t7_fixture.py
"""
import os
import struct

import numpy as np

NIL, NUMBER, STRING, TABLE, TORCH, BOOLEAN, FUNCTION, LEGACY_RECUR_FUNCTION, RECUR_FUNCTION = range(9)
out = bytearray()
_next = [1]


def i32(v): out.extend(struct.pack("<i", v))
def i64(v): out.extend(struct.pack("<q", v))
def f64(v): out.extend(struct.pack("<d", v))
def raw_string(s): b = s.encode("latin-1"); i32(len(b)); out.extend(b)
def new_index(): v = _next[0]; _next[0] += 1; return v
def num(v): i32(NUMBER); f64(float(v))
def string(s): i32(STRING); raw_string(s)
def boolean(v): i32(BOOLEAN); i32(1 if v else 0)
def nil(): i32(NIL)
def backref(tag, index): i32(tag); i32(index)


def table(pairs):
    """pairs: list of (emit_key, emit_value) callables.  Returns the table's index."""
    i32(TABLE); idx = new_index(); i32(idx); i32(len(pairs))
    for k, v in pairs:
        k(); v()
    return idx


def torch_header(cls, versioned=True):
    i32(TORCH); idx = new_index(); i32(idx)
    if versioned:
        raw_string("V 1")
    raw_string(cls)
    return idx


def storage(data, cls="torch.FloatStorage"):
    idx = torch_header(cls)
    i64(data.size); out.extend(np.ascontiguousarray(data).tobytes())
    return idx


def tensor(shape, strides, offset1, emit_storage, cls="torch.FloatTensor", versioned=True):
    idx = torch_header(cls, versioned)
    i32(len(shape))
    for s in shape: i64(s)
    for s in strides: i64(s)
    i64(offset1)
    emit_storage()
    return idx


def ftensor(a):
    a = np.ascontiguousarray(a, np.float32)
    strides = [s // 4 for s in a.strides]
    return tensor(a.shape, strides, 1, lambda: storage(a.reshape(-1)))


def K(s): return lambda: string(s)
def Kn(i): return lambda: num(i)


def obj(cls, pairs, versioned=True):
    idx = torch_header(cls, versioned)
    table(pairs)
    return idx


def lua_function(tag, upvalues):
    """tags 7/8: [int tag][int index][int size][bytecode][upvalue table]; tag 6 (pre-2015 TYPE_FUNCTION): no index"""
    i32(tag)
    if tag != FUNCTION:
        i32(new_index())
    code = b"\x1bLJ\x02 fake bytecode"
    i32(len(code)); out.extend(code)
    upvalues()


def arr(items):
    return lambda: table([(Kn(i + 1), it) for i, it in enumerate(items)])


rng = np.random.default_rng(7)
W = {}                      # what the reader must recover: name -> array


def rnd(name, *shape):
    W[name] = rng.standard_normal(shape).astype(np.float32)
    return W[name]


def conv(name, co, ci, legacy_mm=False, view=False):
    w = rnd(name + "_w", co, ci, 3, 3); b = rnd(name + "_b", co)
    def emit():
        def weight():
            if view:
                # the weight is rows 1.. of a bigger storage, every element 2 apart: offset 1+5, strides doubled
                big = np.full(5 + 2 * w.size, 777.0, np.float32)
                big[5::2][:w.size] = w.reshape(-1)
                tensor(w.shape, [2 * (s // 4) for s in w.strides], 6, lambda: storage(big))
            elif legacy_mm:
                ftensor(w.reshape(co, ci * 9))                    # SpatialConvolutionMM kept (nOut, nIn*kH*kW)
            else:
                ftensor(w)
        obj("nn.SpatialConvolutionMM" if legacy_mm else "nn.SpatialConvolution",
            [(K("weight"), weight), (K("bias"), lambda: ftensor(b)), (K("nOutputPlane"), lambda: num(co)),
             (K("nInputPlane"), lambda: num(ci)), (K("kH"), lambda: num(3)), (K("kW"), lambda: num(3)),
             (K("train"), lambda: boolean(False))])
    return emit


def simple(cls, pairs=(), versioned=True):
    return lambda: obj(cls, list(pairs), versioned)


def seq(items, cls="nn.Sequential"):
    return lambda: obj(cls, [(K("modules"), arr(items))])


def linear(name, no, ni, remember=None):
    w = rnd(name + "_w", no, ni); b = rnd(name + "_b", no)
    def emit():
        idx = obj("nn.Linear", [(K("weight"), lambda: ftensor(w)), (K("bias"), lambda: ftensor(b))])
        if remember is not None:
            remember[name] = idx
    return emit


# miniature dimensions: VGG widths 2,2,3,3,4,4,4,5,5,5,5,5,5 ; k=2 anchors ; R=3 ; D=6 ; E=Hd=4 ; V=5 ; T=3
chans = [(3, 2), (2, 2), (2, 3), (3, 3), (3, 4), (4, 4), (4, 4), (4, 5), (5, 5), (5, 5), (5, 5), (5, 5), (5, 5)]
relu = simple("nn.ReLU", versioned=False)            # old files carry class names without the "V 1" header
pool = simple("nn.SpatialMaxPooling", [(K("kW"), lambda: num(2)), (K("ceil_mode"), lambda: boolean(True))])
net1, net2 = [], []
for li, (ci, co) in enumerate(chans):
    c = conv("conv%d" % li, co, ci, legacy_mm=(li == 2), view=(li == 4))
    (net1 if li < 4 else net2).extend([c, relu])
    if li in (1, 3):
        net1.append(pool)
    if li in (6, 9):
        net2.append(pool)
anchors = rnd("anchors", 2, 2)
shared_idx = {}
make_anchors = simple("nn.MakeAnchors", [(K("x0"), lambda: num(8.5)), (K("y0"), lambda: num(8.5)), (K("sx"), lambda: num(16)),
                                         (K("sy"), lambda: num(16)), (K("anchors"), lambda: ftensor(anchors))])
reshape = simple("nn.ReshapeBoxFeatures", [(K("k"), lambda: num(2))])
box_branch = seq([conv("rpn_box", 8, 3), simple("nn.RegularizeLayer"),
                  seq([seq([make_anchors, reshape]), reshape], "nn.ConcatTable"),
                  seq([simple("nn.ApplyBoxTransform"), simple("nn.Identity")], "nn.ConcatTable")])
rpn = seq([conv("rpn_conv", 3, 5), relu, seq([box_branch, seq([conv("rpn_score", 4, 3), reshape])], "nn.ConcatTable"),
           simple("nn.FlattenTable")])


def localization_layer():
    obj("nn.LocalizationLayer", [
        (K("nets"), lambda: table([(K("rpn"), rpn)])),
        # a closure stored on the module (all three function encodings must be skipped)
        (K("timer_hook"), lambda: lua_function(RECUR_FUNCTION, lambda: table([(Kn(1), lambda: table([(K("name"), lambda: string("_ENV"))]))]))),
        (K("legacy_hook"), lambda: lua_function(LEGACY_RECUR_FUNCTION, lambda: table([(Kn(1), lambda: num(3))]))),
        (K("old_hook"), lambda: lua_function(FUNCTION, lambda: table([]))),
        (K("image_height"), nil)])


def recog_base():
    shared_idx["recog_base"] = obj("nn.Sequential", [(K("modules"), arr([
        simple("nn.View"), linear("fc6", 6, 5 * 49), relu, simple("nn.Dropout", [(K("p"), lambda: num(0.5))]),
        linear("fc7", 6, 6), relu, simple("nn.Dropout", [(K("p"), lambda: num(0.5))])]))])


lstm_w = rnd("lstm_w", 8, 16); lstm_b = rnd("lstm_b", 16)
emb = rnd("lm_emb", 7, 4)
# lookup-table weight and LSTM bias share ONE storage in this file (second tensor refers back to the storage index)
both = np.concatenate([emb.reshape(-1), lstm_b])
share = {}


def emb_tensor():
    def st():
        share["idx"] = storage(both)
    tensor(emb.shape, [4, 1], 1, st)


def lstm_bias_tensor():
    tensor(lstm_b.shape, [1], 1 + emb.size, lambda: backref(TORCH, share["idx"]))


def language_model():
    shared_idx["language_model"] = obj("nn.LanguageModel", [
        (K("vocab_size"), lambda: num(5)), (K("seq_length"), lambda: num(3)),
        (K("idx_to_token"), lambda: table([(Kn(i), (lambda i=i: string("tok%d" % i))) for i in range(1, 6)])),
        (K("image_encoder"), seq([linear("lm_enc", 4, 6), relu, simple("nn.View")])),
        (K("lookup_table"), lambda: obj("nn.LookupTable", [(K("weight"), emb_tensor)])),
        (K("rnn"), seq([lambda: obj("nn.LSTM", [(K("weight"), lambda: ftensor(lstm_w)), (K("bias"), lstm_bias_tensor)]),
                        simple("nn.View"), linear("lm_out", 6, 4), simple("nn.View")])),
        (K("sample_argmax"), lambda: boolean(True))])


def node(module_emit, name):
    return lambda: obj("graph.Node", [(K("data"), lambda: table([
        (K("module"), module_emit), (K("annotations"), lambda: table([(K("name"), lambda: string(name))])),
        (K("mapindex"), lambda: table([]))])), (K("children"), lambda: table([]))], versioned=False)


def recog_net():
    # nn.gModule (DenseCapModel.lua:159): its nodes hold the SAME module objects as nets.* -> back-references
    obj("nn.gModule", [
        (K("name"), lambda: string("recognition_network")),
        (K("forwardnodes"), arr([node(lambda: backref(TORCH, shared_idx["recog_base"]), "recog_base"),
                                 node(lambda: backref(TORCH, shared_idx["obj"]), "objectness_branch"),
                                 node(lambda: backref(TORCH, shared_idx["language_model"]), "language_model")])),
        (K("verbose"), lambda: boolean(False))])


def model():
    obj("nn.DenseCapModel", [
        (K("nets"), lambda: table([
            (K("conv_net1"), seq(net1)), (K("conv_net2"), seq(net2)), (K("localization_layer"), localization_layer),
            (K("recog_base"), recog_base), (K("objectness_branch"), linear("obj", 1, 6, shared_idx)),
            (K("box_reg_branch"), linear("boxreg", 4, 6)), (K("language_model"), language_model),
            (K("recog_net"), recog_net)])),
        (K("opt"), lambda: table([(K("final_nms_thresh"), lambda: num(0.3)), (K("rpn_hidden_dim"), lambda: num(512))])),
        (K("finetune_cnn"), lambda: boolean(False))])


table([(K("model"), model), (K("iter"), lambda: num(20000)),
       (K("loss_history"), lambda: table([(Kn(1), lambda: num(2.5)), (Kn(2), lambda: num(2.25))])),
       (K("opt"), lambda: table([(K("checkpoint_path"), lambda: string("checkpoint.t7"))]))])

here = os.path.dirname(os.path.abspath(__file__))
open(os.path.join(here, "handmade_checkpoint.t7"), "wb").write(bytes(out))
np.savez(os.path.join(here, "handmade_checkpoint_expected.npz"), **W)
print("wrote handmade_checkpoint.t7 (%d bytes), %d expected arrays" % (len(out), len(W)))
