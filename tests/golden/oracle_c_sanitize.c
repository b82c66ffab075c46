/* Driver for the AddressSanitizer / UBSan build of oracle/oracle_c.c (tests/test_oracle_sanitizers.py).
 * Reads cases from stdin, runs them through the instrumented functions and prints the results; any sanitizer report
 * aborts with a non-zero status.  Test infrastructure only.
 *   nms n overlap max_boxes  then n x 5 floats        -> "nms cnt p0 p1 ..."
 *   roi C h w B img_h img_w HH WW  then C*h*w feature floats and B*4 box floats -> "roi" + B*C*HH*WW floats (%a)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_nms(const float* boxes5, int n, float overlap, int max_boxes, int* pick);
void oracle_bilinear_roi_pool(const float* feat, int C, int h, int w, const float* boxes, int B, int img_h, int img_w,
                              int HH, int WW, float* out);

static float* read_floats(size_t n) {
  float* p = (float*)malloc((n ? n : 1) * sizeof(float));      /* exact size: ASan sees every overrun */
  for (size_t i = 0; i < n; ++i)
    if (scanf("%a", &p[i]) != 1) { fprintf(stderr, "short input\n"); exit(3); }
  return p;
}

int main(void) {
  char op[8];
  while (scanf("%7s", op) == 1) {
    if (strcmp(op, "nms") == 0) {
      int n, max_boxes; float overlap;
      if (scanf("%d %a %d", &n, &overlap, &max_boxes) != 3) return 3;
      float* b = read_floats((size_t)(n > 0 ? n : 0) * 5);
      const int cap = n > 0 ? (max_boxes >= 0 && max_boxes < n ? max_boxes : n) : 0;
      int* pick = (int*)malloc((cap ? cap : 1) * sizeof(int));  /* exactly what the contract promises to fill */
      const int cnt = oracle_nms(b, n, overlap, max_boxes, pick);
      printf("nms %d", cnt);
      for (int i = 0; i < cnt; ++i) printf(" %d", pick[i]);
      printf("\n");
      free(b); free(pick);
    } else if (strcmp(op, "roi") == 0) {
      int C, h, w, B, ih, iw, HH, WW;
      if (scanf("%d %d %d %d %d %d %d %d", &C, &h, &w, &B, &ih, &iw, &HH, &WW) != 8) return 3;
      float* f = read_floats((size_t)C * h * w);
      float* bx = read_floats((size_t)B * 4);
      const size_t no = (size_t)B * C * HH * WW;
      float* out = (float*)malloc((no ? no : 1) * sizeof(float));
      oracle_bilinear_roi_pool(f, C, h, w, bx, B, ih, iw, HH, WW, out);
      printf("roi");
      for (size_t i = 0; i < no; ++i) printf(" %a", out[i]);
      printf("\n");
      free(f); free(bx); free(out);
    } else {
      fprintf(stderr, "unknown op %s\n", op);
      return 3;
    }
  }
  return 0;
}
