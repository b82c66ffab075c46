"""Transcribes the reference's own known-answer tests into reference_vectors.json.

The reference (jcjohnson/densecap) is Lua/Torch7 and cannot be executed here, so
the vectors are TRANSCRIBED BY HAND from its test files (file:line cited per
entry); this script only serialises them.  Indices are converted to 0-based.
Run:  python tests/golden/make_reference_vectors.py
"""
import json, math, os

L = math.log
nms_boxes_12 = [[-12, 3, -7, 9, 1], [-9, 7, -4, 13, 2], [-8, 8, -3, 14, 3], [3.5, 4.5, 8.5, 12.5, 4],
                [-6, -6, -1, -1, 5], [4, 5, 9, 13, 6], [4.5, 5.5, 9.5, 13.5, 7]]
nms_boxes_3 = [[-12, 3, -7, 9, 2.5], [-9, 7, -4, 13, 2], [-8, 8, -3, 14, 3], [3.5, 4.5, 8.5, 12.5, 4],
               [-6, -6, -1, -1, 5], [4, 5, 9, 13, 10], [4.5, 5.5, 9.5, 13.5, 7]]

def mb(N, k, H, W, x0, y0, sx, sy, anchors, inputs, expected_nhwk):
    """MakeBoxes case: inputs {(n,k,y,x):[tx,ty,tw,th]} (0-based), expected {(n,y,x,k):[...]}"""
    return dict(N=N, k=k, H=H, W=W, x0=x0, y0=y0, sx=sx, sy=sy, anchors=anchors,
                inputs=[[list(key), v] for key, v in inputs.items()],
                expected=[[list(key), v] for key, v in expected_nhwk.items()])

log0_5, log1_1, log2, log1_5, log0_9 = (-0.69314718055995, 0.095310179804325, 0.69314718055995,
                                         0.40546510810816, -0.10536051565783)
vec = {
  "_source": "jcjohnson/densecap test/*.lua (hand transcription; see make_reference_vectors.py)",
  "nms": [  # test/nms_test.lua:9-95 ; expected picks converted to 0-based
    dict(cite="test/nms_test.lua:9-32", boxes=nms_boxes_12, thresh=0.7, expected=[6, 4, 3, 2, 1, 0]),
    dict(cite="test/nms_test.lua:39-62", boxes=nms_boxes_12, thresh=0.5, expected=[6, 4, 2, 0]),
    dict(cite="test/nms_test.lua:71-94", boxes=nms_boxes_3, thresh=0.7, expected=[5, 4, 2, 0, 1]),
  ],
  "apply_box_transform": dict(  # test/ApplyBoxTransform_test.lua:12-35 (tol 1e-5)
    cite="test/ApplyBoxTransform_test.lua:12-35", tol=1e-5,
    boxes=[[10, 20, 30, 40], [-30, 50, 100, 200]],
    trans=[[0.5, -0.2, 0, L(2)], [0, -0.5, L(0.5), L(1.1)]],
    expected=[[25, 12, 30, 80], [-30, -50, 50, 220]]),
  "box_to_affine": dict(  # test/BoxToAffine_test.lua:14-44 (tol 1e-6)
    cite="test/BoxToAffine_test.lua:14-44", tol=1e-6, H=20, W=30,
    boxes=[[15.5, 10.5, 30, 20], [10, 12, 4, 5], [15, 17, 2, 3], [15.5, 10.5, 15, 10]],
    expected=[[[1, 0, 0], [0, 1, 0]],
              [[1 / 4, 0, 3 / 19], [0, 2 / 15, -11 / 29]],
              [[3 / 20, 0, 13 / 19], [0, 1 / 15, -1 / 29]],
              [[0.5, 0, 0], [0, 0.5, 0]]]),
  "make_boxes": [  # test/MakeBoxes_test.lua:47-234 (tol 1e-4); MakeAnchors o Reshape o Apply == MakeBoxes (MakeAnchors_test.lua:17-44)
    mb(1, 1, 1, 1, 1.5, 2.5, 1.0, 1.0, [[10], [20]], {}, {(0, 0, 0, 0): [1.5, 2.5, 10, 20]}),
    mb(1, 3, 1, 1, 1.5, 2.5, 1.0, 1.0, [[10, 30, 100], [20, 40, 200]], {},
       {(0, 0, 0, 0): [1.5, 2.5, 10, 20], (0, 0, 0, 1): [1.5, 2.5, 30, 40], (0, 0, 0, 2): [1.5, 2.5, 100, 200]}),
    mb(1, 1, 2, 3, 1.5, 2.5, 1.0, 2.0, [[10], [20]], {},
       {(0, 0, 0, 0): [1.5, 2.5, 10, 20], (0, 0, 1, 0): [2.5, 2.5, 10, 20], (0, 0, 2, 0): [3.5, 2.5, 10, 20],
        (0, 1, 0, 0): [1.5, 4.5, 10, 20], (0, 1, 1, 0): [2.5, 4.5, 10, 20], (0, 1, 2, 0): [3.5, 4.5, 10, 20]}),
    mb(2, 1, 1, 1, 2.0, 3.0, 10.0, 20.0, [[100], [200]], {},
       {(0, 0, 0, 0): [2.0, 3.0, 100, 200], (1, 0, 0, 0): [2.0, 3.0, 100, 200]}),
    mb(2, 2, 1, 1, 2.0, 3.0, 1.0, 2.0, [[100, 10], [200, 20]],
       {(0, 0, 0, 0): [0.25, 0.1, 0, 0], (0, 1, 0, 0): [0.1, 0.25, 0, 0],
        (1, 0, 0, 0): [0, 0.05, 0, 0], (1, 1, 0, 0): [0.05, 0, 0, 0]},
       {(0, 0, 0, 0): [27.0, 23.0, 100, 200], (0, 0, 0, 1): [3.0, 8.0, 10, 20],
        (1, 0, 0, 0): [2, 13, 100, 200], (1, 0, 0, 1): [2.5, 3, 10, 20]}),
    mb(2, 2, 1, 1, 2.0, 3.0, 1.0, 2.0, [[100, 10], [200, 20]],
       {(0, 0, 0, 0): [0, 0, log1_1, 0], (0, 1, 0, 0): [0, 0, 0, log2],
        (1, 0, 0, 0): [0, 0, log1_5, log1_1], (1, 1, 0, 0): [0, 0, log2, log1_5]},
       {(0, 0, 0, 0): [2, 3, 110, 200], (0, 0, 0, 1): [2, 3, 10, 40],
        (1, 0, 0, 0): [2, 3, 150, 220], (1, 0, 0, 1): [2, 3, 20, 30]}),
    # bigTestForward (:165-234). NB the Lua expected tensor is filled as
    # [n][y][x][k] and the x index there is written as the 3rd subscript.
    mb(2, 2, 2, 2, 2.0, 3.0, 1.0, 2.0, [[100, 10], [200, 20]],
       {(0, 0, 0, 0): [0, 0, 0, 0], (0, 1, 0, 0): [0, 0, 0, log1_5],
        (0, 0, 0, 1): [0, 0, log0_5, 0], (0, 1, 0, 1): [0, 0, log2, log1_5],
        (0, 0, 1, 0): [0, -0.02, 0, 0], (0, 1, 1, 0): [0, 0.1, 0, log1_5],
        (0, 0, 1, 1): [0, 0.1, log1_1, 0], (0, 1, 1, 1): [0, 0.25, log1_5, log2],
        (1, 0, 0, 0): [-0.05, 0, 0, 0], (1, 1, 0, 0): [0.5, 0, 0, log0_5],
        (1, 0, 0, 1): [0.1, 0, log1_1, 0], (1, 1, 0, 1): [0.2, 0, log0_5, log1_5],
        (1, 0, 1, 0): [-0.01, 0.1, 0, 0], (1, 1, 1, 0): [1.1, 2.0, 0, log2],
        (1, 0, 1, 1): [-1, 1, log0_9, 0], (1, 1, 1, 1): [0.1, -0.2, log1_1, log0_9]},
       {(0, 0, 0, 0): [2, 3, 100, 200], (0, 0, 0, 1): [2, 3, 10, 30],
        (0, 0, 1, 0): [3, 3, 50, 200], (0, 0, 1, 1): [3, 3, 20, 30],
        (0, 1, 0, 0): [2, 1, 100, 200], (0, 1, 0, 1): [2, 7, 10, 30],
        (0, 1, 1, 0): [3, 25, 110, 200], (0, 1, 1, 1): [3, 10, 15, 40],
        (1, 0, 0, 0): [-3, 3, 100, 200], (1, 0, 0, 1): [7, 3, 10, 10],
        (1, 0, 1, 0): [13, 3, 110, 200], (1, 0, 1, 1): [5, 3, 5, 30],
        (1, 1, 0, 0): [1, 25, 100, 200], (1, 1, 0, 1): [13, 45, 10, 40],
        (1, 1, 1, 0): [-97, 205, 90, 200], (1, 1, 1, 1): [4, 1, 11, 18]}),
  ],
  "decode_sequence": dict(  # test/LanguageModel_test.lua:135-160
    cite="test/LanguageModel_test.lua:135-160", vocab_size=5,
    idx_to_token={"1": "a", "2": "cat", "3": "dog", "4": "eating", "5": "hungry"},
    seq=[[1, 5, 2, 4, 1, 3, 6], [1, 3, 6, 0, 0, 0, 0], [2, 3, 1, 3, 2, 6, 0]],
    expected=["a hungry cat eating a dog", "a dog", "cat dog a dog cat"]),
  # test/BoxIoU_test.lua:13-94 -- written for the module's ORIGINAL converter (x0 = xc - w/2, kept as a comment at
  # BoxIoU.lua:15-37); the live module calls box_utils.xcycwh_to_x1y1x2y2 ((w-1)/2) and no longer gives these values.
  # They pin the `legacy_half_w` convention.  Batches are listed one (boxes1, boxes2, expected) each.
  "box_iou_legacy_half_w": [
    dict(cite="test/BoxIoU_test.lua:13-24", tol=1e-10, boxes1=[[10, 10, 10, 10]], boxes2=[[15, 15, 10, 10]],
         expected=[[25 / 175]]),
    dict(cite="test/BoxIoU_test.lua:27-38", tol=1e-10, boxes1=[[10, 10, 5, 5]], boxes2=[[15, 15, 5, 5]], expected=[[0]]),
    dict(cite="test/BoxIoU_test.lua:41-61", tol=1e-8, boxes1=[[2, 4, 2, 6], [5, 7.5, 2, 5]],
         boxes2=[[5, 8, 4, 2], [4.5, 4.5, 5, 3], [4.5, 0, 5, 4]],
         expected=[[0, 3 / 24, 1 / 31], [4 / 14, 2 / 23, 0]]),
    dict(cite="test/BoxIoU_test.lua:64-94 (batch element 2)", tol=1e-8, boxes1=[[4, 2, 2, 6], [6, -2, 2, 2]],
         boxes2=[[4, 2, 4, 2], [4.5, -1, 3, 2], [6, -2, 4, 4]],
         expected=[[1 / 4, 1 / 8, 1 / 27], [0, 1 / 9, 1 / 4]]),
  ],
  "reshape_consistency": dict(  # test/ReshapeBoxFeatures_test.lua:33-58 (exact)
    cite="test/ReshapeBoxFeatures_test.lua:33-58", k=2, D=5, H=4, W=3,
    x0=1, y0=1, sx=2, sy=2, anchors=[[10, 20], [20, 10]],
    set_transform=dict(a=0, y=2, x=1, value=[10, 10, 0, 0]),
    set_feature=dict(a=0, y=2, x=1, value=100),
    expected_row=7, expected_box=[103, 205, 10, 20]),
}
here = os.path.dirname(os.path.abspath(__file__))
json.dump(vec, open(os.path.join(here, "reference_vectors.json"), "w"), indent=1)
print("wrote reference_vectors.json")
