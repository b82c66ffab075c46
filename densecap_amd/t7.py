"""Torch7 binary serialization (`.t7`) reader + the walk from a densecap checkpoint to a weights dict.

SURVEY.md 8(f) row 1: `run_model.lua:146-147` does `torch.load(checkpoint).model`; the checkpoint is the
whole `nn.DenseCapModel` object graph written by `train.lua:174-185` (float tensors).  This module reads
that format without Torch7 and extracts the tensors `dc_load_weights` needs.

Status: written from the published Torch7 `File.lua` format; no real checkpoint is available offline.  It is
exercised against (a) `tests/golden/handmade_checkpoint.t7`, a miniature checkpoint assembled byte by byte from the
format by an independent script (closures in all three encodings, back-references, an nn.gModule with graph.Node
objects, a legacy SpatialConvolutionMM 2-D weight, strided/offset tensor views, shared storages, unversioned class
names) and (b) round trips through the writer below (which exists for those tests and for exporting synthetic
checkpoints) -- still "parity unpinned" against a real `.t7` until one is at hand.
"""
from __future__ import annotations

import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5
TYPE_FUNCTION, TYPE_LEGACY_RECUR_FUNCTION, TYPE_RECUR_FUNCTION = 6, 7, 8

_TENSOR_DTYPES = {
    "torch.FloatTensor": np.float32, "torch.DoubleTensor": np.float64, "torch.LongTensor": np.int64,
    "torch.IntTensor": np.int32, "torch.ByteTensor": np.uint8, "torch.CharTensor": np.int8,
    "torch.ShortTensor": np.int16, "torch.CudaTensor": np.float32,
}
_STORAGE_DTYPES = {k.replace("Tensor", "Storage"): v for k, v in _TENSOR_DTYPES.items()}


class TorchObject:
    """A deserialised torch class instance that is not a tensor/storage: class name + field table."""

    def __init__(self, torch_type, fields=None):
        self.torch_type = torch_type
        self.fields = fields if fields is not None else {}

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default) if isinstance(self.fields, dict) else default

    def __repr__(self):
        return "TorchObject(%s)" % self.torch_type


class LuaFunction:
    """Placeholder for a serialized Lua closure (its bytecode is not kept)."""

    def __init__(self, size):
        self.bytecode_size = size
        self.upvalues = None

    def __repr__(self):
        return "LuaFunction(%d bytes)" % self.bytecode_size


def _lua_list(t):
    """Lua array part of a table dict {1:..,2:..} -> python list (or None if not array-like)."""
    if not isinstance(t, dict):
        return None
    n = len(t)
    out = []
    for i in range(1, n + 1):
        if i in t:
            out.append(t[i])
        elif float(i) in t:
            out.append(t[float(i)])
        else:
            return None
    return out


class T7FormatError(ValueError):
    """The bytes are not a well-formed Torch7 serialisation (or not one this reader understands): raised instead of
    returning a half-parsed object."""


class T7Reader:
    """Torch7 `File.lua` binary reader.  Every structural field is checked against what `torch.save` can write
    (docs/SEMANTICS.md, section `.t7`), so that a damaged or truncated file fails loudly:
      * lengths (strings, closures, storages, table counts) are non-negative and fit the bytes that are left;
      * a NEW object's index is the next one in sequence (File.lua numbers objects 1, 2, 3 ... in the order they are first
        written), anything else must be a back-reference to an object already read;
      * a tensor's view (size, stride, offset) lies inside its storage; at most MAX_DIMS dimensions; booleans are 0 / 1;
      * `load` requires the top-level object to end exactly at the end of the file."""
    MAX_DIMS = 16

    def __init__(self, f, strict=True):
        self.f = f
        self.memo = {}
        self.strict = strict          # False: the two checks that rest on torch.save's habits (index sequence, single object) are off
        pos = f.tell()
        f.seek(0, 2)
        self.size = f.tell()
        f.seek(pos)

    def _left(self):
        return self.size - self.f.tell()

    def _read(self, fmt):
        sz = struct.calcsize(fmt)
        b = self.f.read(sz)
        if len(b) != sz:
            raise EOFError("truncated t7 file")
        return struct.unpack(fmt, b)

    def _bytes(self, n, what):
        if n < 0 or n > self._left():
            raise T7FormatError("t7: %s of %d bytes at offset %d does not fit the file (%d bytes left)" % (what, n, self.f.tell(), self._left()))
        return self.f.read(n)

    def read_int(self):
        return self._read("<i")[0]

    def read_long(self):
        return self._read("<q")[0]

    def read_double(self):
        return self._read("<d")[0]

    def read_string(self):
        n = self.read_int()
        return self._bytes(n, "string").decode("latin-1")

    def read_bool(self):
        v = self.read_int()                                    # File.lua writes a boolean as the int 1 or 0
        if v not in (0, 1):
            raise T7FormatError("t7: boolean %d at offset %d" % (v, self.f.tell() - 4))
        return v == 1

    def _new_index(self, idx):
        """File.lua gives the objects of one file the indices 1, 2, 3 ... in the order of their first appearance."""
        if self.strict and idx != len(self.memo) + 1:
            raise T7FormatError("t7: object index %d at offset %d is neither a back-reference nor the next new index (%d)"
                                % (idx, self.f.tell() - 4, len(self.memo) + 1))

    def read_object(self):
        t = self.read_int()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = self.read_double()
            return int(v) if v == int(v) and abs(v) < 2 ** 53 else v
        if t == TYPE_BOOLEAN:
            return self.read_bool()
        if t == TYPE_STRING:
            return self.read_string()
        if t == TYPE_TABLE:
            idx = self.read_int()
            if idx in self.memo:
                return self.memo[idx]
            self._new_index(idx)
            n = self.read_int()
            if n < 0 or n > self._left() // 8:                # an entry is at least two 4-byte tags
                raise T7FormatError("t7: table of %d entries at offset %d does not fit the file" % (n, self.f.tell() - 4))
            tab = {}
            self.memo[idx] = tab
            for _ in range(n):
                k = self.read_object()
                v = self.read_object()
                try:
                    tab[k] = v
                except TypeError:
                    raise T7FormatError("t7: unhashable table key of type %s" % type(k).__name__)
            return tab
        if t == TYPE_TORCH:
            idx = self.read_int()
            if idx in self.memo:
                return self.memo[idx]
            self._new_index(idx)
            version = self.read_string()
            if version.startswith("V "):
                cls = self.read_string()
            else:
                cls = version
            if cls in _TENSOR_DTYPES:
                self.memo[idx] = None                          # the tensor's slot: its storage is numbered after it
                nd = self.read_int()
                if nd < 0 or nd > self.MAX_DIMS:
                    raise T7FormatError("t7: tensor with %d dimensions at offset %d" % (nd, self.f.tell() - 4))
                size = [self.read_long() for _ in range(nd)]
                stride = [self.read_long() for _ in range(nd)]
                offset = self.read_long() - 1
                storage = self.read_object()
                if storage is not None and not (isinstance(storage, np.ndarray) and storage.ndim == 1):
                    raise T7FormatError("t7: a tensor's storage field holds a %s" % type(storage).__name__)
                if storage is None or nd == 0:
                    arr = np.zeros((0,), _TENSOR_DTYPES[cls])
                else:
                    if storage.dtype != np.dtype(_TENSOR_DTYPES[cls]):
                        raise T7FormatError("t7: %s over a %s storage" % (cls, storage.dtype))
                    if min(size) < 0 or min(stride) < 0 or offset < 0:
                        raise T7FormatError("t7: tensor view with a negative size / stride / offset")
                    last = offset + sum((n - 1) * st for n, st in zip(size, stride)) if min(size) > 0 else offset - 1
                    if last >= storage.size:
                        raise T7FormatError("t7: tensor view (size %s, stride %s, offset %d) leaves its storage of %d elements"
                                            % (size, stride, offset + 1, storage.size))
                    stride = [st if n > 1 else 0 for n, st in zip(size, stride)]      # a length-1 axis never steps
                    arr = np.lib.stride_tricks.as_strided(
                        storage[offset:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
                self.memo[idx] = arr
                return arr
            if cls in _STORAGE_DTYPES:
                n = self.read_long()
                dt = np.dtype(_STORAGE_DTYPES[cls])
                if n < 0 or n > self._left() // dt.itemsize:
                    raise T7FormatError("t7: storage of %d elements at offset %d does not fit the file" % (n, self.f.tell() - 8))
                arr = np.frombuffer(self._bytes(n * dt.itemsize, "storage"), dtype=dt).copy()
                self.memo[idx] = arr
                return arr
            obj = TorchObject(cls)
            self.memo[idx] = obj
            obj.fields = self.read_object()     # default torch class read(): one table of fields
            if not isinstance(obj.fields, dict):
                raise T7FormatError("t7: %s is followed by a %s, not by its field table" % (cls, type(obj.fields).__name__))
            return obj
        if t in (TYPE_FUNCTION, TYPE_RECUR_FUNCTION, TYPE_LEGACY_RECUR_FUNCTION):
            # File.lua: RECUR_FUNCTION (8) / LEGACY_RECUR_FUNCTION (7) are memoised objects: [index][int size][string.dump
            # bytecode][upvalue table]; the pre-2015 TYPE_FUNCTION (6) is [int size][bytecode][upvalue table] with NO
            # index and no memo entry.  Closures carry no weights: the bytecode is skipped; the upvalue table is still
            # parsed so that object indices stay in step for later back-references.
            idx = None
            if t != TYPE_FUNCTION:
                idx = self.read_int()
                if idx in self.memo:
                    return self.memo[idx]
                self._new_index(idx)
            size = self.read_int()
            self._bytes(size, "function body")
            fn = LuaFunction(size)
            if idx is not None:
                self.memo[idx] = fn
            fn.upvalues = self.read_object()
            if fn.upvalues is not None and not isinstance(fn.upvalues, dict):
                raise T7FormatError("t7: a closure's upvalues are a %s, not a table" % type(fn.upvalues).__name__)
            return fn
        raise T7FormatError("t7: unknown type tag %d at offset %d" % (t, self.f.tell() - 4))


def load(path, strict=True):
    """torch.load(path) for the binary format.  strict: the file must hold exactly one object whose parts are numbered 1, 2, 3
    ... in order of first appearance (what torch.save writes); anything else means the structure was not what it claimed to
    be.  strict=False keeps every bounds check but drops those two habits-of-the-writer checks (a file appended to by hand, or
    written through an already-used File object, still loads)."""
    with open(path, "rb") as f:
        r = T7Reader(f, strict=strict)
        obj = r.read_object()
        if strict and r._left() != 0:
            raise T7FormatError("t7: %d bytes follow the top-level object" % r._left())
        return obj


# ---- minimal writer (tests / exporting synthetic checkpoints) ---------------------------------------
class T7Writer:
    def __init__(self, f):
        self.f = f
        self.next_index = 1
        self.memo = {}
        self.keep = []      # memoised objects stay alive so that id() keys cannot be recycled

    def _w(self, fmt, *v):
        self.f.write(struct.pack(fmt, *v))

    def write_string(self, s):
        b = s.encode("latin-1")
        self._w("<i", len(b))
        self.f.write(b)

    def _index(self, obj):
        key = id(obj)
        if key in self.memo:
            self._w("<i", self.memo[key])
            return True
        self.memo[key] = self.next_index
        self.keep.append(obj)
        self._w("<i", self.next_index)
        self.next_index += 1
        return False

    def write_object(self, o):
        if o is None:
            self._w("<i", TYPE_NIL)
        elif isinstance(o, bool):
            self._w("<i", TYPE_BOOLEAN)
            self._w("<i", 1 if o else 0)
        elif isinstance(o, (int, float)):
            self._w("<i", TYPE_NUMBER)
            self._w("<d", float(o))
        elif isinstance(o, str):
            self._w("<i", TYPE_STRING)
            self.write_string(o)
        elif isinstance(o, (list, tuple)):
            self.write_object({i + 1: v for i, v in enumerate(o)})
        elif isinstance(o, dict):
            self._w("<i", TYPE_TABLE)
            if self._index(o):
                return
            self._w("<i", len(o))
            for k, v in o.items():
                self.write_object(k)
                self.write_object(v)
        elif isinstance(o, np.ndarray):
            cls = {np.dtype(np.float32): "torch.FloatTensor", np.dtype(np.float64): "torch.DoubleTensor",
                   np.dtype(np.int64): "torch.LongTensor", np.dtype(np.int32): "torch.IntTensor",
                   np.dtype(np.uint8): "torch.ByteTensor"}[o.dtype]
            self._w("<i", TYPE_TORCH)
            if self._index(o):
                return
            self.write_string("V 1")
            self.write_string(cls)
            a = np.ascontiguousarray(o)
            self._w("<i", a.ndim)
            for s in a.shape:
                self._w("<q", s)
            for s in a.strides:
                self._w("<q", s // a.itemsize)
            self._w("<q", 1)
            self._w("<i", TYPE_TORCH)            # the storage object
            self._w("<i", self.next_index)
            self.next_index += 1
            self.write_string("V 1")
            self.write_string(cls.replace("Tensor", "Storage"))
            self._w("<q", a.size)
            self.f.write(a.tobytes())
        elif isinstance(o, TorchObject):
            self._w("<i", TYPE_TORCH)
            if self._index(o):
                return
            self.write_string("V 1")
            self.write_string(o.torch_type)
            self.write_object(o.fields)
        else:
            raise TypeError("t7 writer: unsupported %r" % type(o))


def save(path, obj):
    with open(path, "wb") as f:
        T7Writer(f).write_object(obj)


# ---- checkpoint -> weights dict -------------------------------------------------------------------
def _modules(seq):
    mods = _lua_list(seq["modules"])
    if mods is None:
        raise ValueError("expected an nn container with a `modules` array")
    return mods


def iter_modules(obj, _seen=None):
    """Depth-first walk over every torch object reachable from `obj`: container `modules` arrays, modules held in
    named fields (nn.LanguageModel's image_encoder / rnn / lookup_table) and nn.gModule graphs (`forwardnodes` of
    graph.Node objects whose data.module is the wrapped module, DenseCapModel.lua:127-162).  Shared objects are
    yielded once; tensors and closures are leaves."""
    seen = _seen if _seen is not None else set()
    if isinstance(obj, TorchObject):
        if id(obj) in seen:
            return
        seen.add(id(obj))
        yield obj
        yield from iter_modules(obj.fields, seen)
    elif isinstance(obj, dict):
        if id(obj) in seen:
            return
        seen.add(id(obj))
        for k in sorted(obj, key=lambda k: (not isinstance(k, (int, float)), str(type(k)), k if isinstance(k, (int, float)) else str(k))):
            yield from iter_modules(obj[k], seen)


def _is(obj, *names):
    return isinstance(obj, TorchObject) and any(obj.torch_type.endswith(n) for n in names)


def _conv_weight(m):
    w = np.asarray(m["weight"], np.float32)
    no, ni = int(m["nOutputPlane"]), int(m["nInputPlane"])
    kh, kw = int(m.get("kH", 3)), int(m.get("kW", 3))
    return np.ascontiguousarray(w.reshape(no, ni, kh, kw)), np.asarray(m["bias"], np.float32)


def weights_from_checkpoint(ckpt):
    """ckpt: object returned by load() for a densecap checkpoint (table with `.model`) or the model itself.
    Returns the dict DenseCapModel(...) takes (same keys as densecap_amd.weights.make_synthetic_weights)."""
    model = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    nets = model["nets"]
    W = {}
    convs = []
    for net in (nets["conv_net1"], nets["conv_net2"]):          # DenseCapModel.lua:73-76
        for m in _modules(net):
            if _is(m, "SpatialConvolution", "SpatialConvolutionMM"):
                convs.append(_conv_weight(m))
    if len(convs) != 13:
        raise ValueError("expected the 13 VGG-16 convolutions, found %d" % len(convs))
    W["conv_w"] = [c[0] for c in convs]
    W["conv_b"] = [c[1] for c in convs]
    rpn = _modules(nets["localization_layer"]["nets"]["rpn"])    # LocalizationLayer.lua:609-690
    W["rpn_conv_w"], W["rpn_conv_b"] = _conv_weight(rpn[0])
    branches = _modules(rpn[2])
    box_branch = _modules(branches[0])
    W["rpn_box_w"], W["rpn_box_b"] = _conv_weight(box_branch[0])
    W["rpn_score_w"], W["rpn_score_b"] = _conv_weight(_modules(branches[1])[0])
    make_anchors = None
    for m in box_branch:
        if _is(m, "ConcatTable"):
            for sub in _modules(m):
                if _is(sub, "Sequential"):
                    for leaf in _modules(sub):
                        if _is(leaf, "MakeAnchors"):
                            make_anchors = leaf
    if make_anchors is None:
        raise ValueError("nn.MakeAnchors not found in the RPN")
    W["anchors"] = np.asarray(make_anchors["anchors"], np.float32)
    W["field_centers"] = tuple(float(make_anchors[k]) for k in ("x0", "y0", "sx", "sy"))
    fcs = [m for m in _modules(nets["recog_base"]) if _is(m, "Linear")]   # VGG layers 32..38
    W["fc6_w"], W["fc6_b"] = np.asarray(fcs[0]["weight"], np.float32), np.asarray(fcs[0]["bias"], np.float32)
    W["fc7_w"], W["fc7_b"] = np.asarray(fcs[1]["weight"], np.float32), np.asarray(fcs[1]["bias"], np.float32)
    for key, name in (("obj", "objectness_branch"), ("boxreg", "box_reg_branch")):
        W[key + "_w"] = np.asarray(nets[name]["weight"], np.float32)
        W[key + "_b"] = np.asarray(nets[name]["bias"], np.float32)
    lm = nets["language_model"]                                   # LanguageModel.lua:10-74
    enc = _modules(lm["image_encoder"])[0]
    W["lm_enc_w"], W["lm_enc_b"] = np.asarray(enc["weight"], np.float32), np.asarray(enc["bias"], np.float32)
    W["lm_emb"] = np.asarray(lm["lookup_table"]["weight"], np.float32)
    for m in _modules(lm["rnn"]):
        if _is(m, "LSTM"):
            W["lstm_w"], W["lstm_b"] = np.asarray(m["weight"], np.float32), np.asarray(m["bias"], np.float32)
        elif _is(m, "Linear"):
            W["lm_out_w"], W["lm_out_b"] = np.asarray(m["weight"], np.float32), np.asarray(m["bias"], np.float32)
    W["vocab_size"] = int(lm["vocab_size"])
    W["seq_length"] = int(lm["seq_length"])
    itt = lm.get("idx_to_token") or {}
    W["idx_to_token"] = {int(k): v for k, v in itt.items()}
    # Test-time state the deserialised OBJECT carries (what the model runs with until setTestArgs is called):
    # LocalizationLayer.test_clip_boxes / test_nms_thresh / test_max_proposals (LocalizationLayer.lua:155,233-238) and
    # DenseCapModel.opt.final_nms_thresh (DenseCapModel.lua:31,261).  Absent fields stay None -> constructor defaults.
    ll = nets["localization_layer"]
    opt = model.get("opt") if hasattr(model, "get") else None
    ta = {}
    for k in ("test_clip_boxes", "test_nms_thresh", "test_max_proposals"):
        v = ll.get(k) if hasattr(ll, "get") else None
        if v is not None:
            ta[k] = bool(v) if k == "test_clip_boxes" else (int(v) if k == "test_max_proposals" else float(v))
    if isinstance(opt, dict) and opt.get("final_nms_thresh") is not None:
        ta["final_nms_thresh"] = float(opt["final_nms_thresh"])
    W["test_args"] = ta
    return W


def checkpoint_from_weights(W):
    """Inverse of weights_from_checkpoint: builds the nn.DenseCapModel object graph (only the fields the
    reader needs) so synthetic checkpoints can be written and round-tripped."""
    def T(a):
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else a
        return np.ascontiguousarray(a, dtype=np.float32)

    def conv(w, b):
        w = T(w)
        return TorchObject("nn.SpatialConvolution", dict(weight=w, bias=T(b), nOutputPlane=w.shape[0],
                                                         nInputPlane=w.shape[1], kH=w.shape[2], kW=w.shape[3]))

    def seq(mods, cls="nn.Sequential"):
        return TorchObject(cls, dict(modules=list(mods)))

    def lin(w, b):
        return TorchObject("nn.Linear", dict(weight=T(w), bias=T(b)))

    ta = dict(W.get("test_args") or {})
    relu = TorchObject("nn.ReLU", {})
    pool = TorchObject("nn.SpatialMaxPooling", dict(kW=2, kH=2, dW=2, dH=2, ceil_mode=True))
    cw, cb = W["conv_w"], W["conv_b"]
    net1 = [conv(cw[0], cb[0]), relu, conv(cw[1], cb[1]), relu, pool, conv(cw[2], cb[2]), relu, conv(cw[3], cb[3]),
            relu, pool]
    net2 = []
    for i in range(4, 13):
        net2 += [conv(cw[i], cb[i]), relu]
        if i in (6, 9):
            net2.append(pool)
    x0, y0, sx, sy = W["field_centers"]
    make_anchors = TorchObject("nn.MakeAnchors", dict(x0=x0, y0=y0, sx=sx, sy=sy, anchors=T(W["anchors"])))
    k = T(W["anchors"]).shape[1]
    reshape = TorchObject("nn.ReshapeBoxFeatures", dict(k=k))
    box_branch = seq([conv(W["rpn_box_w"], W["rpn_box_b"]), TorchObject("nn.RegularizeLayer", {}),
                      seq([seq([make_anchors, reshape]), reshape], "nn.ConcatTable"),
                      seq([TorchObject("nn.ApplyBoxTransform", {}), TorchObject("nn.Identity", {})], "nn.ConcatTable")])
    rpn_branch = seq([conv(W["rpn_score_w"], W["rpn_score_b"]), reshape])
    rpn = seq([conv(W["rpn_conv_w"], W["rpn_conv_b"]), relu, seq([box_branch, rpn_branch], "nn.ConcatTable"),
               TorchObject("nn.FlattenTable", {})])
    drop = TorchObject("nn.Dropout", dict(p=0.5))
    recog_base = seq([TorchObject("nn.View", {}), lin(W["fc6_w"], W["fc6_b"]), relu, drop,
                      lin(W["fc7_w"], W["fc7_b"]), relu, drop])
    lm = TorchObject("nn.LanguageModel", dict(
        vocab_size=int(W["vocab_size"]), seq_length=int(W["seq_length"]),
        idx_to_token={int(k_): v for k_, v in (W.get("idx_to_token") or {}).items()},
        image_encoder=seq([lin(W["lm_enc_w"], W["lm_enc_b"]), relu, TorchObject("nn.View", {})]),
        lookup_table=TorchObject("nn.LookupTable", dict(weight=T(W["lm_emb"]))),
        rnn=seq([TorchObject("nn.LSTM", dict(weight=T(W["lstm_w"]), bias=T(W["lstm_b"]))),
                 TorchObject("nn.View", {}), lin(W["lm_out_w"], W["lm_out_b"]), TorchObject("nn.View", {})])))
    nets = dict(conv_net1=seq(net1), conv_net2=seq(net2),
                localization_layer=TorchObject("nn.LocalizationLayer", dict(
                    nets=dict(rpn=rpn), test_clip_boxes=bool(ta.get("test_clip_boxes", True)),
                    test_nms_thresh=float(ta.get("test_nms_thresh", 0.7)),
                    test_max_proposals=int(ta.get("test_max_proposals", 300)))),
                recog_base=recog_base, objectness_branch=lin(W["obj_w"], W["obj_b"]),
                box_reg_branch=lin(W["boxreg_w"], W["boxreg_b"]), language_model=lm)
    model = TorchObject("nn.DenseCapModel", dict(nets=nets, opt=dict(final_nms_thresh=float(ta.get("final_nms_thresh", 0.3)))))
    return dict(model=model, iter=0)
