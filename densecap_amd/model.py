"""Host-side mirror of the reference's `DenseCapModel` for the test-time path.

Same method names, argument meaning and error behaviour as
densecap/DenseCapModel.lua (setTestArgs :185-191, convert :198-208, evaluate,
forward_test :319-327, extractFeatures :285-304) and LanguageModel:decodeSequence
(LanguageModel.lua:86-103), so the callers `run_model.lua:145-164`,
`webcam/daemon.lua:46-85` and `eval/eval_utils.lua:62` change only their constructor
line.  All numerics run in libdensecap_hip.so (HIP, gfx950); this file only marshals
pointers.  The LuaJIT twin of this class is lua/DenseCapModelHIP.lua.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DcResult, DcWeights, check
from .ops import Context


def _np32(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def decode_sequence(seq, idx_to_token, vocab_size):
    """LanguageModel:decodeSequence (LanguageModel.lua:86-103): per row, join idx_to_token[tok] with ' ' until the END
    token (vocab_size + 1) or a 0.  idx_to_token: dict or list keyed by the 1-based token id (int or str keys, as the
    checkpoint's JSON-born table has them); None -> the ids themselves."""
    end = int(vocab_size) + 1
    caps = []
    for row in np.asarray(seq):
        words = []
        for tok in row:
            tok = int(tok)
            if tok == end or tok == 0:
                break
            if idx_to_token is None:
                words.append(str(tok))
            elif isinstance(idx_to_token, dict):
                words.append(idx_to_token[tok] if tok in idx_to_token else idx_to_token[str(tok)])
            else:
                words.append(idx_to_token[tok])
        caps.append(" ".join(words))
    return caps


def getopt(opt, key, default_value=None):
    """utils.getopt (densecap/utils.lua:67-75): opt[key], or the default when the key is absent / nil (None here);
    a missing key without a default is an error, as in the reference."""
    v = None if opt is None else opt.get(key)
    if v is None:
        if default_value is None:
            raise KeyError("error: required key %s was not provided in an opt." % key)
        return default_value
    return v


class LocalizationLayerTestArgs:
    """`model.nets.localization_layer` as far as the test-time path reads it: the three fields that
    LocalizationLayer:setTestArgs writes (LocalizationLayer.lua:233-238) and _forward_test reads (:250-256).  A call
    re-derives ALL three -- an omitted key goes back to its default (true / 0.7 / 300), it is not retained."""

    def __init__(self, stored=None):
        self.setTestArgs()                               # LocalizationLayer.lua:155: the constructor's own call
        for k in ("test_clip_boxes", "test_nms_thresh", "test_max_proposals"):
            if stored and stored.get(k) is not None:     # the deserialised object's fields (a checkpoint stores them)
                setattr(self, k, stored[k])

    def setTestArgs(self, args=None, **kw):
        args = dict(args or {}, **kw)
        self.test_clip_boxes = bool(getopt(args, "clip_boxes", True))
        self.test_nms_thresh = float(getopt(args, "nms_thresh", 0.7))
        self.test_max_proposals = int(getopt(args, "max_proposals", 300))
        return self


class _Nets:
    def __init__(self, localization_layer):
        self.localization_layer = localization_layer


class DenseCapModel:
    def __init__(self, weights, device=0, ctx=None):
        """weights: dict in checkpoint layouts (see densecap_amd/weights.py); device: HIP index
        (utils.setup_gpus(gpu) with gpu >= 0; there is no `-gpu -1` CPU mode here).
        weights["test_args"] (optional; t7.weights_from_checkpoint fills it): the test-time state the checkpoint OBJECT
        carries -- localization_layer.test_clip_boxes / test_nms_thresh / test_max_proposals and opt.final_nms_thresh --
        which is what the model runs with until somebody calls setTestArgs."""
        from .weights import check_weight_shapes
        check_weight_shapes(weights)                  # before any pointer crosses the ABI: dc_load_weights trusts the shapes
        self.ctx = ctx or Context(device)
        self.lib = self.ctx.lib
        stored = dict(weights.get("test_args") or {})
        self.nets = _Nets(LocalizationLayerTestArgs(stored))
        # DenseCapModel.lua:31: opt.final_nms_thresh defaults to 0.3 at construction; forward reads self.opt at call time
        self.opt = dict(final_nms_thresh=float(getopt(stored, "final_nms_thresh", 0.3)))
        self.vocab_size = int(weights["vocab_size"])
        self.seq_length = int(weights["seq_length"])
        self.idx_to_token = weights.get("idx_to_token")
        self.captions_after_final_nms = False
        self._keep = []  # host arrays referenced by the struct during dc_load_weights
        w = DcWeights()

        def ptr(a):
            a = _np32(a)
            self._keep.append(a)
            return a.ctypes.data_as(_lib.c_float_p)

        for i in range(_lib.DC_NUM_VGG_CONVS):
            w.conv_w[i] = ptr(weights["conv_w"][i])
            w.conv_b[i] = ptr(weights["conv_b"][i])
        for name in ("rpn_conv_w", "rpn_conv_b", "rpn_box_w", "rpn_box_b", "rpn_score_w", "rpn_score_b", "fc6_w",
                     "fc6_b", "fc7_w", "fc7_b", "obj_w", "obj_b", "boxreg_w", "boxreg_b", "lm_enc_w", "lm_enc_b",
                     "lm_emb", "lstm_w", "lstm_b", "lm_out_w", "lm_out_b", "anchors"):
            setattr(w, name, ptr(weights[name]))
        for i, v in enumerate(weights["field_centers"]):
            w.field_centers[i] = float(v)
        w.num_anchors = int(_np32(weights["anchors"]).shape[1])
        self.num_anchors = int(w.num_anchors)
        w.rpn_hidden = int(_np32(weights["rpn_conv_w"]).shape[0])
        w.vocab_size = self.vocab_size
        w.seq_length = self.seq_length
        w.enc_size = int(_np32(weights["lm_enc_w"]).shape[0])
        w.rnn_size = int(_np32(weights["lstm_w"]).shape[1] // 4)
        w.fc_dim = int(_np32(weights["fc7_w"]).shape[0])
        self.fc_dim = w.fc_dim
        check(self.ctx.h, self.lib.dc_load_weights(self.ctx.h, C.byref(w)), "dc_load_weights")
        self._keep = []
        self._push_test_args()

    # ---- reference API ---------------------------------------------------------------------
    def setTestArgs(self, args=None, **kw):
        """DenseCapModel:setTestArgs{rpn_nms_thresh=, final_nms_thresh=, num_proposals=} (DenseCapModel.lua:185-191), as
        written: EVERY call re-derives all three values -- rpn 0.7, num_proposals 1000, final 0.3 for the keys that are
        absent -- and, because it calls the layer's setTestArgs without a `clip_boxes` key, turns box clipping back on.
        Keys it does not know are ignored (evaluate_model.lua:39-43 passes `max_proposals=`, which the reference never
        reads: that caller runs with 1000 proposals)."""
        kwargs = dict(args or {}, **kw)
        ll = self.nets.localization_layer
        # a value the library refuses (num_proposals = 0, 2000000 ...) must not stay behind in the object: every later
        # forward would re-raise in _push_test_args (advisor finding, round 4).  The previous state comes back on failure.
        saved = (ll.test_clip_boxes, ll.test_nms_thresh, ll.test_max_proposals, self.opt["final_nms_thresh"])
        try:
            ll.setTestArgs(nms_thresh=getopt(kwargs, "rpn_nms_thresh", 0.7),
                           max_proposals=getopt(kwargs, "num_proposals", 1000))
            self.opt["final_nms_thresh"] = float(getopt(kwargs, "final_nms_thresh", 0.3))
            self._push_test_args()
        except Exception:
            ll.test_clip_boxes, ll.test_nms_thresh, ll.test_max_proposals, self.opt["final_nms_thresh"] = saved
            raise
        return self

    def _push_test_args(self):
        """The reference reads localization_layer.test_* and opt.final_nms_thresh when forward runs
        (LocalizationLayer.lua:250-256, DenseCapModel.lua:261) -- callers such as train.lua:139-143 write them directly --
        so the current values travel to the library before every forward."""
        ll = self.nets.localization_layer
        check(self.ctx.h, self.lib.dc_set_test_args(self.ctx.h, float(ll.test_nms_thresh),
                                                    float(self.opt["final_nms_thresh"]),
                                                    int(ll.test_max_proposals)), "dc_set_test_args")
        if not ll.test_clip_boxes:
            check(self.ctx.h, self.lib.dc_set_localization_test_args(self.ctx.h, 0, float(ll.test_nms_thresh),
                                                                     int(ll.test_max_proposals)),
                  "dc_set_localization_test_args")

    def setLanes(self, lanes):
        """Streams dc_forward_batch pipelines images over (1 = serial kernels)."""
        check(self.ctx.h, self.lib.dc_set_lanes(self.ctx.h, int(lanes)), "dc_set_lanes")
        return self

    def autotuneLanes(self, dev_ptr, n, H, W, candidates=(2, 3, 4), reps=2):
        """Pick the number of lanes (>= 2: all give bit-identical results, only the overlap of the images' kernels
        changes) that gives the best throughput on THIS device for n resident images; which count wins differs
        between otherwise identical GPUs (measured: 3 lanes 171 vs 2 lanes 162 images/s on one box, 160 vs 167 on
        another).  Returns {lanes: images/s}; the best one is left set."""
        import time
        rates = {}
        for lanes in candidates:
            self.setLanes(lanes)
            self.forward_batch_device(dev_ptr, min(n, lanes), H, W)          # lane workspaces exist before timing
            best = 0.0
            for _ in range(reps):
                t0 = time.perf_counter()
                self.forward_batch_device(dev_ptr, n, H, W)
                best = max(best, n / (time.perf_counter() - t0))
            rates[lanes] = best
        self.setLanes(max(rates, key=rates.get))
        return rates

    def autotuneSchedule(self, dev_ptr, n, H, W, lanes=(2, 3, 4), groups=(1, 2, 4, 8), reps=2):
        """Pick the lane count AND the images per group together (both are pure scheduling knobs: results are bit-identical
        for any lanes >= 2 and any group): the best lane count depends on the group size -- 300 proposals: 2 lanes x groups
        of four 312 images/s, 4 lanes x groups of four 299, 2 lanes x single images 285.  Returns {(lanes, group): images/s};
        the best pair is left set."""
        import time
        rates = {}
        for l in lanes:
            for g in groups:
                self.setLanes(l); self.setGroup(g)
                self.forward_batch_device(dev_ptr, min(n, l * g), H, W)      # workspaces of this shape exist before timing
                best = 0.0
                for _ in range(reps):
                    t0 = time.perf_counter()
                    self.forward_batch_device(dev_ptr, n, H, W)
                    best = max(best, n / (time.perf_counter() - t0))
                rates[(l, g)] = best
        l, g = max(rates, key=rates.get)
        self.setLanes(l); self.setGroup(g)
        return rates

    def autotuneGroup(self, dev_ptr, n, H, W, candidates=(1, 2, 4, 8), reps=2):
        """Pick the images-per-group setting (dc_set_group: a scheduling knob like the lane count -- results are
        bit-identical) at the current lane count.  With few proposals per image the RoI stages of one image leave the
        chip's tile rounds badly filled and a group of four shares them (300 proposals: +9 % images/s at two lanes); at
        1000 proposals groups change nothing.  Returns {group: images/s}; the best one is left set."""
        import time
        rates = {}
        for g in candidates:
            self.setGroup(g)
            self.forward_batch_device(dev_ptr, min(n, 2 * g), H, W)
            best = 0.0
            for _ in range(reps):
                t0 = time.perf_counter()
                self.forward_batch_device(dev_ptr, n, H, W)
                best = max(best, n / (time.perf_counter() - t0))
            rates[g] = best
        self.setGroup(max(rates, key=rates.get))
        return rates

    def setCaptionOrder(self, after_final_nms):
        """False (default): decode all proposals then NMS, as the reference does.  True: final NMS first,
        decode only the survivors (bit-identical outputs, less LSTM work)."""
        check(self.ctx.h, self.lib.dc_set_caption_order(self.ctx.h, int(bool(after_final_nms))), "dc_set_caption_order")
        self.captions_after_final_nms = bool(after_final_nms)
        return self

    def setMathMode(self, mode):
        """dc_set_math_mode: 0 = fp32 MFMA (default; the arithmetic results are bit-compared in), 1 = split-bf16 (every
        operand as three bf16 planes, six partial products on the bf16 matrix cores, fp32 accumulate: fp32-class error at
        2.67x the matrix rate).  Opt-in; may be switched between forwards."""
        check(self.ctx.h, self.lib.dc_set_math_mode(self.ctx.h, int(mode)), "dc_set_math_mode")
        self.math_mode = int(mode)
        return self

    def setGraphReplay(self, on):
        """dc_set_graph_replay: repeated forwards of one shape on a lane are captured once and relaunched as a hipGraph
        (bit-identical; pays with one image in flight -- run_model on single images, the webcam daemon)."""
        check(self.ctx.h, self.lib.dc_set_graph_replay(self.ctx.h, int(bool(on))), "dc_set_graph_replay")
        return self

    def setGroup(self, images):
        """Images per group inside a batch call: 0/1 = every image on its own (default), 2..8 = the images of a group share
        the launches of the dense stages (bit-identical results; 1000 proposals: +4.5 % images/s on one lane, nothing with
        two or more lanes; 300 proposals: +9 % at two lanes with groups of four)."""
        check(self.ctx.h, self.lib.dc_set_group(self.ctx.h, int(images)), "dc_set_group")
        return self

    def setBeamSize(self, beam_size):
        """language_model.beam_size (LanguageModel.lua:129-131): None/0 = greedy sample, n = beam search with n beams."""
        check(self.ctx.h, self.lib.dc_set_beam_size(self.ctx.h, int(beam_size or 0)), "dc_set_beam_size")
        return self

    def convert(self, dtype=None, use_cudnn=None):
        """model:convert(dtype, use_cudnn): the HIP path is always fp32 on the ctx's device."""
        return self

    def evaluate(self):
        return self

    def _check_input(self, img):
        img = np.asarray(img)
        if img.ndim == 4:
            assert img.shape[0] == 1 and img.shape[1] == 3, "input must be (1,3,H,W)"  # DenseCapModel.lua:244
            img = img[0]
        assert img.ndim == 3 and img.shape[0] == 3, "input must be (1,3,H,W)"
        return np.ascontiguousarray(img, dtype=np.float32)

    def _new_result(self, P):
        T = self.seq_length
        boxes = np.zeros((P, 4), np.float32); scores = np.zeros((P,), np.float32)
        tokens = np.zeros((P, T), np.int32)
        r = DcResult()
        r.capacity = P
        r.boxes = boxes.ctypes.data_as(_lib.c_float_p)
        r.scores = scores.ctypes.data_as(_lib.c_float_p)
        r.tokens = tokens.ctypes.data_as(_lib.c_int32_p)
        return r, boxes, scores, tokens

    def _capacity(self, H, W):
        P = int(self.nets.localization_layer.test_max_proposals)
        if P != -1:
            return P
        for _ in range(4):                       # four ceil-mode 2x2 pools (conv5_3 map)
            H, W = (H + 1) // 2, (W + 1) // 2
        return self.num_anchors * H * W

    def forward_raw(self, img):
        """forward_test without string decoding: (boxes (K,4) xcycwh, scores (K,), tokens (K,T))."""
        self._push_test_args()
        img = self._check_input(img)
        P = self._capacity(img.shape[1], img.shape[2])
        r, boxes, scores, tokens = self._new_result(P)
        check(self.ctx.h, self.lib.dc_forward_test(self.ctx.h, img.ctypes.data, img.shape[1], img.shape[2], 0,
                                                   C.byref(r)), "dc_forward_test")
        K = r.K
        return boxes[:K].copy(), scores[:K].copy(), tokens[:K].copy()

    def forward_test(self, img):
        """Returns final_boxes (K,4), objectness_scores (K,1), captions (list of K strings)."""
        boxes, scores, tokens = self.forward_raw(img)
        return boxes, scores[:, None], self.decodeSequence(tokens)

    def forward_batch_device(self, imgs_dev_ptr, n, H, W):
        """run_model.lua's image loop over n device-resident images of one size; returns a list of
        (boxes, scores, tokens).  imgs_dev_ptr: device pointer to (n,3,H,W) fp32."""
        self._push_test_args()
        P = self._capacity(H, W)
        arr = (DcResult * n)()
        keep = []
        for i in range(n):
            r, b, s, t = self._new_result(P)
            arr[i] = r
            keep.append((b, s, t))
        check(self.ctx.h, self.lib.dc_forward_batch(self.ctx.h, imgs_dev_ptr, n, H, W, 1, arr), "dc_forward_batch")
        return [(b[:arr[i].K], s[:arr[i].K], t[:arr[i].K]) for i, (b, s, t) in enumerate(keep)]

    def forward_batch(self, imgs):
        self._push_test_args()
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        n, c, H, W = imgs.shape
        assert c == 3
        P = self._capacity(H, W)
        arr = (DcResult * n)()
        keep = []
        for i in range(n):
            r, b, s, t = self._new_result(P)
            arr[i] = r
            keep.append((b, s, t))
        check(self.ctx.h, self.lib.dc_forward_batch(self.ctx.h, imgs.ctypes.data, n, H, W, 0, arr), "dc_forward_batch")
        return [(b[:arr[i].K].copy(), s[:arr[i].K].copy(), t[:arr[i].K].copy()) for i, (b, s, t) in enumerate(keep)]

    def forward_images(self, imgs):
        """run_model.lua's loop over a list of images of DIFFERENT sizes, pipelined over the lanes (dc_forward_images);
        imgs: sequence of (3,H,W) / (1,3,H,W) arrays.  Returns a list of (boxes, scores, tokens)."""
        self._push_test_args()
        arrs = [self._check_input(im) for im in imgs]
        n = len(arrs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        Hs = (C.c_int * n)(*[a.shape[1] for a in arrs])
        Ws = (C.c_int * n)(*[a.shape[2] for a in arrs])
        res = (DcResult * n)()
        keep = []
        for i, a in enumerate(arrs):
            r, b, s, t = self._new_result(self._capacity(a.shape[1], a.shape[2]))
            res[i] = r
            keep.append((b, s, t))
        check(self.ctx.h, self.lib.dc_forward_images(self.ctx.h, ptrs, Hs, Ws, n, 0, res), "dc_forward_images")
        return [(b[:res[i].K].copy(), s[:res[i].K].copy(), t[:res[i].K].copy()) for i, (b, s, t) in enumerate(keep)]

    def forward_images_device(self, dev_imgs):
        """forward_images on images that are ALREADY on the device: dev_imgs = sequence of ops.DeviceArray (3,H,W) float32
        (ops.preprocess_u8 makes them).  Runs of equal-sized images travel as groups (setGroup)."""
        self._push_test_args()
        n = len(dev_imgs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[a.ptr.value for a in dev_imgs])
        Hs = (C.c_int * n)(*[a.shape[1] for a in dev_imgs])
        Ws = (C.c_int * n)(*[a.shape[2] for a in dev_imgs])
        res = (DcResult * n)()
        keep = []
        for i, a in enumerate(dev_imgs):
            r, b, s, t = self._new_result(self._capacity(a.shape[1], a.shape[2]))
            res[i] = r
            keep.append((b, s, t))
        check(self.ctx.h, self.lib.dc_forward_images(self.ctx.h, ptrs, Hs, Ws, n, 1, res), "dc_forward_images")
        return [(b[:res[i].K].copy(), s[:res[i].K].copy(), t[:res[i].K].copy()) for i, (b, s, t) in enumerate(keep)]

    def extractFeatures_images_device(self, dev_imgs):
        """extractFeatures_images on device-resident images (ops.preprocess_u8): list of (boxes, feats)."""
        self._push_test_args()
        n = len(dev_imgs)
        if n == 0:
            return []
        cap = max(self._capacity(a.shape[1], a.shape[2]) for a in dev_imgs)
        ptrs = (C.c_void_p * n)(*[a.ptr.value for a in dev_imgs])
        Hs = (C.c_int * n)(*[a.shape[1] for a in dev_imgs])
        Ws = (C.c_int * n)(*[a.shape[2] for a in dev_imgs])
        boxes = np.zeros((n, cap, 4), np.float32); feats = np.zeros((n, cap, self.fc_dim), np.float32)
        K = (C.c_int32 * n)()
        check(self.ctx.h, self.lib.dc_extract_features_images(self.ctx.h, ptrs, Hs, Ws, n, 1, cap, boxes.ctypes.data,
                                                              feats.ctypes.data, K), "dc_extract_features_images")
        return [(boxes[i, :K[i]].copy(), feats[i, :K[i]].copy()) for i in range(n)]

    def extractFeatures(self, img):
        """DenseCapModel:extractFeatures -> (boxes_xcycwh (K,4), feats (K,fc_dim))."""
        self._push_test_args()
        img = self._check_input(img)
        P = self._capacity(img.shape[1], img.shape[2])
        boxes = np.zeros((P, 4), np.float32); feats = np.zeros((P, self.fc_dim), np.float32)
        K = C.c_int32(0)
        check(self.ctx.h, self.lib.dc_extract_features(self.ctx.h, img.ctypes.data, img.shape[1], img.shape[2], 0, P,
                                                       boxes.ctypes.data, feats.ctypes.data, C.byref(K)),
              "dc_extract_features")
        return boxes[:K.value].copy(), feats[:K.value].copy()

    def extractFeatures_images(self, imgs):
        """extract_features.lua's loop over images (any sizes), pipelined over the lanes: list of (boxes, feats)."""
        self._push_test_args()
        arrs = [self._check_input(im) for im in imgs]
        n = len(arrs)
        if n == 0:
            return []
        cap = max(self._capacity(a.shape[1], a.shape[2]) for a in arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        Hs = (C.c_int * n)(*[a.shape[1] for a in arrs])
        Ws = (C.c_int * n)(*[a.shape[2] for a in arrs])
        boxes = np.zeros((n, cap, 4), np.float32); feats = np.zeros((n, cap, self.fc_dim), np.float32)
        K = np.zeros((n,), np.int32)
        check(self.ctx.h, self.lib.dc_extract_features_images(self.ctx.h, ptrs, Hs, Ws, n, 0, cap, boxes.ctypes.data,
                                                              feats.ctypes.data, K.ctypes.data_as(_lib.c_int32_p)),
              "dc_extract_features_images")
        return [(boxes[i, :K[i]].copy(), feats[i, :K[i]].copy()) for i in range(n)]

    def decodeSequence(self, seq):
        """LanguageModel:decodeSequence (LanguageModel.lua:86-103)."""
        return decode_sequence(seq, self.idx_to_token, self.vocab_size)

    # ---- instrumentation ---------------------------------------------------------------------
    def stage_times(self):
        names = (C.c_char_p * 16)(); ms = (C.c_float * 16)()
        n = self.lib.dc_stage_times(self.ctx.h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(max(n, 0))}

    def mfma_profile(self, reset=0):
        l = C.c_int64(0); ms = C.c_double(0); fl = C.c_double(0)
        check(self.ctx.h, self.lib.dc_mfma_profile(self.ctx.h, reset, C.byref(l), C.byref(ms), C.byref(fl)))
        return dict(launches=l.value, ms=ms.value, flops=fl.value)

    def debug_fetch(self, name, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        n = self.lib.dc_debug_fetch(self.ctx.h, name.encode(), out.ctypes.data, out.nbytes)
        check(self.ctx.h, int(min(n, 0)), "dc_debug_fetch(%s)" % name)
        return out, int(n)
