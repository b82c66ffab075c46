/* CPU ORACLE (C part) -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * Scalar fp32 restatements of two hot loops of jcjohnson/densecap's test path:
 *   oracle_nms                -- densecap/box_utils.lua:154-256 (box_utils.nms)
 *   oracle_bilinear_roi_pool  -- densecap/modules/BilinearRoiPooling.lua:42-60 =
 *        BoxToAffine.lua:69-93 -> stnbhwd AffineGridGeneratorBHWD(7,7)
 *        -> BatchBilinearSamplerBHWD.lua:104-122 (stnbhwd BilinearSamplerBHWD_updateOutput)
 *        -> Transpose to (B,C,HH,WW)
 * Pinned by tests/test_oracle_golden.py against test/nms_test.lua:9-95 and
 * test/BoxToAffine_test.lua:14-44; the stnbhwd sampler itself is un-vendored
 * ("parity unpinned" for absolute sampler values).
 *
 * Build with -ffp-contract=off so every fp32 op rounds exactly once, in the
 * order the reference's TH vector ops apply them.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int i; } si_t;
static int cmp_desc(const void* a, const void* b) {
    const si_t* x = (const si_t*)a; const si_t* y = (const si_t*)b;
    const int xn = x->s != x->s, yn = y->s != y->s;   /* TH sort (GT_OR_NAN): NaN ranks above every number */
    if (xn != yn) return xn ? -1 : 1;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);  /* tie rule: lower original index first */
}

/* boxes5: (n,5) x1 y1 x2 y2 score.  Writes 0-based picks, returns their count.
 * max_boxes < 0 means uncapped.  NaN scores rank above every number (TH's sort puts NaN at the end of
 * the ascending list, box_utils.lua:185 takes from the tail), ties by index. */
int oracle_nms(const float* boxes5, int n, float overlap, int max_boxes, int* pick) {
    if (n <= 0) return 0;
    si_t* ord = (si_t*)malloc(sizeof(si_t) * (size_t)n);
    float* area = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    for (int j = 0; j < n; ++j) {
        const float* b = boxes5 + 5 * (size_t)j;
        ord[j].s = b[4]; ord[j].i = j;
        float aw = (b[2] - b[0]) + 1.0f;           /* box_utils.lua:178-181 */
        float ah = (b[3] - b[1]) + 1.0f;
        area[j] = aw * ah;
    }
    qsort(ord, (size_t)n, sizeof(si_t), cmp_desc);
    int cnt = 0;
    for (int r = 0; r < n; ++r) {
        if (max_boxes >= 0 && cnt >= max_boxes) break;
        if (dead[r]) continue;
        const int i = ord[r].i;
        pick[cnt++] = i;
        const float* bi = boxes5 + 5 * (size_t)i;
        for (int q = r + 1; q < n; ++q) {            /* candidates still in I */
            if (dead[q]) continue;
            const int j = ord[q].i;
            const float* bj = boxes5 + 5 * (size_t)j;
            float xx1 = bj[0] > bi[0] ? bj[0] : bi[0];   /* cmax(x1, x1[i]) */
            float xx2 = bj[2] < bi[2] ? bj[2] : bi[2];   /* cmin(x2, x2[i]) */
            float yy1 = bj[1] > bi[1] ? bj[1] : bi[1];
            float yy2 = bj[3] < bi[3] ? bj[3] : bi[3];
            float w = (xx2 - xx1) + 1.0f; if (!(w > 0.0f)) w = 0.0f;
            float h = (yy2 - yy1) + 1.0f; if (!(h > 0.0f)) h = 0.0f;
            float inter = w * h;
            float uni = (area[j] + area[i]) - inter;     /* :219-226 */
            float iou = inter / uni;
            if (!(iou <= overlap)) dead[q] = 1;          /* keep iff iou <= overlap (:241) */
        }
    }
    free(ord); free(area); free(dead);
    return cnt;
}

/* feat: (C,h,w) fp32; boxes: (B,4) xc,yc,w,h in image pixels; out: (B,C,HH,WW). */
void oracle_bilinear_roi_pool(const float* feat, int C, int h, int w, const float* boxes, int B,
                              int img_h, int img_w, int HH, int WW, float* out) {
    const float fH = (float)img_h, fW = (float)img_w;
    for (int b = 0; b < B; ++b) {
        const float xc = boxes[4 * b + 0], yc = boxes[4 * b + 1];
        const float bw = boxes[4 * b + 2], bh = boxes[4 * b + 3];
        /* BoxToAffine.lua:88-91 */
        const float th23 = (xc * 2.0f + (-1.0f - fW)) / (fW - 1.0f);
        const float th13 = (yc * 2.0f + (-1.0f - fH)) / (fH - 1.0f);
        const float th22 = bw / fW;
        const float th11 = bh / fH;
        for (int i = 0; i < HH; ++i) {
            const float yb = (float)(-1.0 + ((double)i / (double)(HH - 1)) * 2.0);
            for (int j = 0; j < WW; ++j) {
                const float xb = (float)(-1.0 + ((double)j / (double)(WW - 1)) * 2.0);
                /* AffineGridGeneratorBHWD: grid = base(y,x,1) . theta^T, k-ordered */
                const float gy = (yb * th11 + xb * 0.0f) + th13;
                const float gx = (yb * 0.0f + xb * th22) + th23;
                /* BilinearSamplerBHWD_updateOutput */
                const float xcoord = (gx + 1.0f) * (float)(w - 1) / 2.0f;
                const float ycoord = (gy + 1.0f) * (float)(h - 1) / 2.0f;
                const float xfl = floorf(xcoord), yfl = floorf(ycoord);
                const int x0 = (int)xfl, y0 = (int)yfl;
                const float wx = 1.0f - (xcoord - xfl);
                const float wy = 1.0f - (ycoord - yfl);
                const int tl = x0 >= 0 && x0 <= w - 1 && y0 >= 0 && y0 <= h - 1;
                const int tr = x0 + 1 >= 0 && x0 + 1 <= w - 1 && y0 >= 0 && y0 <= h - 1;
                const int bl = x0 >= 0 && x0 <= w - 1 && y0 + 1 >= 0 && y0 + 1 <= h - 1;
                const int br = x0 + 1 >= 0 && x0 + 1 <= w - 1 && y0 + 1 >= 0 && y0 + 1 <= h - 1;
                for (int c = 0; c < C; ++c) {
                    const float* f = feat + (size_t)c * h * w;
                    const float vtl = tl ? f[y0 * w + x0] : 0.0f;
                    const float vtr = tr ? f[y0 * w + x0 + 1] : 0.0f;
                    const float vbl = bl ? f[(y0 + 1) * w + x0] : 0.0f;
                    const float vbr = br ? f[(y0 + 1) * w + x0 + 1] : 0.0f;
                    const float v = wx * wy * vtl + (1.0f - wx) * wy * vtr
                                  + wx * (1.0f - wy) * vbl + (1.0f - wx) * (1.0f - wy) * vbr;
                    out[(((size_t)b * C + c) * HH + i) * WW + j] = v;
                }
            }
        }
    }
}
