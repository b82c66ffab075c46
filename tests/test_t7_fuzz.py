"""Property fuzz of the `.t7` reader (densecap_amd/t7.py): a damaged checkpoint must fail LOUDLY.

No file written by Torch7 exists offline (f1 row of SURVEY.md 8: format parity stays "no real file available"); what can
be pinned is the reader's behaviour on bytes that are NOT a well-formed serialisation.  Every structural field of two
byte-assembled fixtures -- type tags, object indices, string / closure / storage / table lengths, tensor ranks, sizes,
strides and offsets, booleans -- is overwritten with hostile values; the reader must then either

  * raise (T7FormatError / EOFError / a decode error) -- never hang, never read outside a storage, never return a
    half-parsed object, or
  * return an object EQUAL to the clean parse (the mutation hit a field that carries no information), or
  * -- only for a back-reference rewritten to another EXISTING object index, and for a tensor's size / stride / offset
    rewritten to another view that still lies inside its storage -- return a different object graph: those bytes are a
    well-formed file that says something else, which no reader can tell from damage (a tensor's geometry carries no
    redundancy in the format; wrong SHAPES are caught one layer up, test_checkpoint_walk_rejects_inconsistent_shapes).

docs/SEMANTICS.md (section ".t7") lists every format assumption the checks encode with the File.lua behaviour behind it.
"""
import io
import os
import struct

import numpy as np
import pytest

from densecap_amd import t7

HERE = os.path.dirname(os.path.abspath(__file__))


class RecordingReader(t7.T7Reader):
    """Notes the offset and width of every structural integer the reader consumes, and which ones are object indices."""

    def __init__(self, f):
        super().__init__(f)
        self.fields = []          # (offset, width, value, kind)

    def read_int(self):
        off = self.f.tell()
        v = super().read_int()
        self.fields.append([off, 4, v, "int"])
        return v

    def read_long(self):
        off = self.f.tell()
        v = super().read_long()
        self.fields.append([off, 8, v, "long"])
        return v

    def _new_index(self, idx):
        self.fields[-1][3] = "index"
        return super()._new_index(idx)

    def read_bool(self):
        v = super().read_bool()
        self.fields[-1][3] = "bool"
        return v


def _equal(a, b, seen=None):
    seen = seen if seen is not None else set()
    key = (id(a), id(b))
    if key in seen:
        return True
    seen.add(key)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return (isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape
                and np.array_equal(a, b, equal_nan=a.dtype.kind == "f"))
    if isinstance(a, t7.TorchObject) or isinstance(b, t7.TorchObject):
        return (isinstance(a, t7.TorchObject) and isinstance(b, t7.TorchObject) and a.torch_type == b.torch_type
                and _equal(a.fields, b.fields, seen))
    if isinstance(a, t7.LuaFunction) or isinstance(b, t7.LuaFunction):
        return (isinstance(a, t7.LuaFunction) and isinstance(b, t7.LuaFunction) and a.bytecode_size == b.bytecode_size
                and _equal(a.upvalues, b.upvalues, seen))
    if isinstance(a, dict) or isinstance(b, dict):
        return (isinstance(a, dict) and isinstance(b, dict) and a.keys() == b.keys()
                and all(_equal(a[k], b[k], seen) for k in a))
    return type(a) is type(b) and a == b


def _parse(data):
    r = t7.T7Reader(io.BytesIO(data))
    obj = r.read_object()
    if r._left() != 0:
        raise t7.T7FormatError("trailing bytes")
    return obj


LOUD = (t7.T7FormatError, EOFError, UnicodeDecodeError, KeyError)


def _fixtures(tmp_path):
    from densecap_amd.weights import make_synthetic_weights
    from tests.golden.t7_assembler import assemble_densecap_checkpoint
    out = [open(os.path.join(HERE, "golden", "handmade_checkpoint.t7"), "rb").read()]
    W = make_synthetic_weights(seed=5, vocab_size=12, seq_length=3, fc_dim=256)
    # the reader does not care whether the tensors fit an architecture: cut every one down to a few elements per axis so that
    # the checkpoint-SHAPED file (module tree, flat-storage views, closures, gModule) is ~100 KB and parses in milliseconds
    small = lambda a: np.ascontiguousarray(np.asarray(a)[tuple(slice(0, 5) for _ in np.asarray(a).shape)])
    for k, v in list(W.items()):
        if k in ("conv_w", "conv_b"):
            W[k] = [small(x) for x in v]
        elif hasattr(v, "shape") and k != "anchors":
            W[k] = small(v)
    p = tmp_path / "shaped.t7"
    assemble_densecap_checkpoint(str(p), W)
    out.append(open(p, "rb").read())
    assert len(out[1]) < 400_000
    return out


def _mutations(width, value, rng):
    base = [0, -1, 1, 2, 3, 7, 9, value + 1, value - 1, value ^ 1, value << 1, 2 ** 31 - 1 if width == 4 else 2 ** 62,
            -(2 ** 31) if width == 4 else -(2 ** 62), int(rng.integers(0, 2 ** 20)), value + 256]
    lim = 2 ** 31 if width == 4 else 2 ** 63
    return sorted({m for m in base if m != value and -lim <= m < lim})


@pytest.mark.parametrize("which", [0, 1])
def test_structural_mutations_fail_loudly_or_change_nothing(tmp_path, which):
    data = _fixtures(tmp_path)[which]
    rr = RecordingReader(io.BytesIO(data))
    clean = rr.read_object()
    assert rr._left() == 0
    fields = rr.fields
    assert len(fields) > 300
    memo_indices = set(rr.memo)
    rng = np.random.default_rng(which)
    # every field of the small fixture; a spread sample of the checkpoint-shaped one (its 13 convs repeat the same layout)
    picks = range(len(fields)) if len(fields) < 1500 else sorted(set(rng.choice(len(fields), 500, replace=False)))
    loud = same = rewired = 0
    silent = []
    for fi in picks:
        off, width, value, kind = fields[fi]
        for m in _mutations(width, value, rng):
            mutated = bytearray(data)
            mutated[off:off + width] = struct.pack("<i" if width == 4 else "<q", m)
            try:
                obj = _parse(bytes(mutated))
            except LOUD:
                loud += 1
                continue
            except MemoryError:
                pytest.fail("mutation %d at offset %d made the reader ask for unbounded memory" % (m, off))
            if _equal(obj, clean):
                same += 1
            elif m in memo_indices and (kind == "index" or value in memo_indices):
                rewired += 1                     # a back-reference that now names another existing object: a valid other file
            elif kind == "long" or (kind == "bool" and m in (0, 1)):
                rewired += 1                     # another in-bounds view of the same storage / the other truth value: a valid other file
            else:
                silent.append((off, width, value, m, kind))
    assert not silent, "silent mis-parses (offset, width, clean value, mutated value, kind): %s" % silent[:8]
    assert loud > 3 * (same + rewired) and loud > 1000, (loud, same, rewired)


def test_truncation_at_every_structural_boundary_is_an_error(tmp_path):
    data = _fixtures(tmp_path)[0]
    rr = RecordingReader(io.BytesIO(data))
    rr.read_object()
    cuts = sorted({f[0] for f in rr.fields} | {f[0] + f[1] for f in rr.fields} | {len(data) - 1, len(data) // 2})
    for c in cuts:
        if c >= len(data):
            continue
        with pytest.raises(LOUD):
            _parse(data[:c])
    with pytest.raises(t7.T7FormatError, match="follow the top-level object"):
        p = tmp_path / "trailing.t7"
        p.write_bytes(data + b"\0\0\0\0")
        t7.load(str(p))
    assert t7.load(str(p), strict=False) is not None          # torch.load itself would stop after the first object


def test_hostile_tensor_views_never_leave_their_storage(tmp_path):
    """size / stride / offset combinations that would make a strided view read outside the storage (the reader builds the
    view with numpy's as_strided, which checks nothing by itself)."""
    def tensor_bytes(size, stride, offset, nelem):
        b = io.BytesIO()
        w = lambda fmt, *v: b.write(struct.pack(fmt, *v))
        def s(x): w("<i", len(x)); b.write(x.encode())
        w("<i", 4); w("<i", 1); s("V 1"); s("torch.FloatTensor")
        w("<i", len(size))
        for v in size: w("<q", v)
        for v in stride: w("<q", v)
        w("<q", offset)
        w("<i", 4); w("<i", 2); s("V 1"); s("torch.FloatStorage"); w("<q", nelem)
        b.write(np.arange(nelem, dtype=np.float32).tobytes())
        return b.getvalue()
    ok = _parse(tensor_bytes([2, 3], [3, 1], 1, 6))
    np.testing.assert_array_equal(ok, np.arange(6, dtype=np.float32).reshape(2, 3))
    np.testing.assert_array_equal(_parse(tensor_bytes([2, 2], [0, 1], 5, 6)), [[4, 5], [4, 5]])     # stride 0 = expand()
    assert _parse(tensor_bytes([0, 3], [3, 1], 1, 6)).size == 0
    for size, stride, offset, n in [([2, 3], [3, 1], 2, 6), ([2, 3], [4, 1], 1, 6), ([7], [1], 1, 6), ([2, 3], [3, 1], 0, 6),
                                    ([2, -3], [3, 1], 1, 6), ([2, 3], [-3, 1], 4, 6), ([2 ** 40], [1], 1, 6),
                                    ([3], [2 ** 40], 1, 6), ([1] * 17, [1] * 17, 1, 6)]:
        with pytest.raises(LOUD):
            _parse(tensor_bytes(size, stride, offset, n))


def test_checkpoint_walk_rejects_inconsistent_shapes():
    """One layer above the format: tensors that parse but do not fit together (a damaged size field, a checkpoint of another
    architecture) are refused by the host before any pointer reaches dc_load_weights -- which trusts the shapes it is told."""
    from densecap_amd.weights import check_weight_shapes, make_synthetic_weights
    W = make_synthetic_weights(seed=5, vocab_size=12, seq_length=3, fc_dim=256)
    W = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in W.items()}
    W["conv_w"] = [w.numpy() if hasattr(w, "numpy") else w for w in W["conv_w"]]
    W["conv_b"] = [b.numpy() if hasattr(b, "numpy") else b for b in W["conv_b"]]
    check_weight_shapes(W)                                   # the clean dict passes
    import copy
    for key, bad in [("fc6_w", lambda a: a[:, :-1]), ("fc7_w", lambda a: a[:-1]), ("lm_emb", lambda a: a[:-1]),
                     ("lstm_w", lambda a: a[:, :-4]), ("lm_out_b", lambda a: a[:-1]), ("rpn_box_w", lambda a: a[:-1]),
                     ("anchors", lambda a: a[:1]), ("obj_w", lambda a: a[:, :-1]), ("fc6_b", lambda a: a[:0])]:
        V = copy.copy(W)
        V[key] = bad(np.asarray(W[key]))
        with pytest.raises(ValueError, match=key.split("_")[0]):
            check_weight_shapes(V)
    V = copy.copy(W)
    V["conv_w"] = list(W["conv_w"]); V["conv_w"][3] = np.asarray(W["conv_w"][3])[:, :-1]
    with pytest.raises(ValueError, match="conv"):
        check_weight_shapes(V)
