/* dc_harness.c -- plain-C caller of libdensecap_hip.so (no Python, no torch).
 *
 * Shows the C ABI of include/densecap.h used the way run_model.lua:145-164 drives the reference:
 * create -> load weights -> setTestArgs -> forward_test -> read (boxes, scores, tokens).
 * Weights are random (a checkpoint-shaped dc_weights filled by a small LCG; no .t7 ships with the
 * reference), so the numbers mean nothing -- the harness checks the call protocol, error reporting and
 * run-to-run determinism, and prints a per-image time.
 *
 * build:  gcc -O2 -std=c11 -Iinclude tools/dc_harness.c -o tools/dc_harness \
 *             -Ldensecap_amd/lib -ldensecap_hip -Wl,-rpath,'$ORIGIN/../densecap_amd/lib' -lm
 * run:    tools/dc_harness [H W num_proposals repeats]
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "densecap.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static float urand(void) { /* xorshift64*, uniform in [0,1) */
  g_state ^= g_state >> 12; g_state ^= g_state << 25; g_state ^= g_state >> 27;
  return (float)((g_state * 0x2545F4914F6CDD1Dull) >> 40) / 16777216.0f;
}
static float nrand(void) { /* sum of 4 uniforms, unit variance */
  return (urand() + urand() + urand() + urand() - 2.0f) * 1.7320508f;
}
static float* filled(size_t n, float std, float mean) {
  float* p = (float*)malloc(n * sizeof(float));
  if (!p) { fprintf(stderr, "out of host memory\n"); exit(2); }
  for (size_t i = 0; i < n; ++i) p[i] = mean + std * nrand();
  return p;
}

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != DC_OK) {                                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dc_last_error(ctx));              \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 224, W = argc > 2 ? atoi(argv[2]) : 288;
  const int P = argc > 3 ? atoi(argv[3]) : 100, reps = argc > 4 ? atoi(argv[4]) : 3;
  static const int cin[DC_NUM_VGG_CONVS] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
  static const int cout[DC_NUM_VGG_CONVS] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
  static const float aw[12] = {45, 90, 64, 90, 180, 128, 181, 362, 256, 362, 724, 512};   /* LocalizationLayer.lua:613-619 */
  static const float ah[12] = {90, 45, 64, 180, 90, 128, 362, 181, 256, 724, 362, 512};
  const int k = 12, R = 256, V = 1000, T = 8, E = 512, Hd = 512, D = 4096;

  dc_ctx* ctx = NULL;
  if (dc_create(&ctx, 0) != DC_OK) { fprintf(stderr, "dc_create: %s\n", dc_last_error(NULL)); return 1; }

  /* call-order error is reported, not thrown: forward before load_weights */
  {
    float dummy[3 * 64 * 64] = {0};
    dc_result r; memset(&r, 0, sizeof r);
    int rc = dc_forward_test(ctx, dummy, 64, 64, 0, &r);
    if (rc == DC_OK) { fprintf(stderr, "forward_test without weights must fail\n"); return 1; }
    printf("expected error before load_weights: %d (%s)\n", rc, dc_last_error(ctx));
  }

  dc_weights w; memset(&w, 0, sizeof w);
  for (int i = 0; i < DC_NUM_VGG_CONVS; ++i) {
    w.conv_w[i] = filled((size_t)cout[i] * cin[i] * 9, sqrtf(2.0f / (9.0f * cin[i])), 0.f);
    w.conv_b[i] = filled((size_t)cout[i], 0.01f, 0.f);
  }
  w.rpn_conv_w = filled((size_t)R * 512 * 9, sqrtf(2.0f / (9.0f * 512)), 0.f); w.rpn_conv_b = filled(R, 0.01f, 0.f);
  w.rpn_box_w = filled((size_t)4 * k * R, 0.01f, 0.f);   w.rpn_box_b = filled(4 * k, 0.01f, 0.f);
  w.rpn_score_w = filled((size_t)2 * k * R, 0.05f, 0.f); w.rpn_score_b = filled(2 * k, 0.1f, 0.f);
  w.fc6_w = filled((size_t)D * 25088, sqrtf(2.0f / 25088), 0.f); w.fc6_b = filled(D, 0.01f, 0.f);
  w.fc7_w = filled((size_t)D * D, sqrtf(2.0f / D), 0.f);          w.fc7_b = filled(D, 0.01f, 0.f);
  w.obj_w = filled(D, 0.02f, 0.f);                  w.obj_b = filled(1, 0.f, 0.f);
  w.boxreg_w = filled((size_t)4 * D, 0.001f, 0.f);  w.boxreg_b = filled(4, 0.f, 0.f);
  w.lm_enc_w = filled((size_t)E * D, sqrtf(2.0f / D), 0.f); w.lm_enc_b = filled(E, 0.01f, 0.f);
  w.lm_emb = filled((size_t)(V + 2) * E, 1.0f, 0.f);
  w.lstm_w = filled((size_t)(E + Hd) * 4 * Hd, 1.0f / sqrtf((float)Hd), 0.f); w.lstm_b = filled((size_t)4 * Hd, 0.01f, 0.f);
  w.lm_out_w = filled((size_t)(V + 1) * Hd, 3.0f / sqrtf((float)Hd), 0.f);    w.lm_out_b = filled(V + 1, 0.1f, 0.f);
  float* anc = (float*)malloc(2 * k * sizeof(float));
  for (int a = 0; a < k; ++a) { anc[a] = aw[a]; anc[k + a] = ah[a]; }
  w.anchors = anc;
  w.field_centers[0] = 8.5f; w.field_centers[1] = 8.5f; w.field_centers[2] = 16.f; w.field_centers[3] = 16.f;
  w.num_anchors = k; w.rpn_hidden = R; w.vocab_size = V; w.seq_length = T; w.enc_size = E; w.rnn_size = Hd; w.fc_dim = D;
  CHECK(dc_load_weights(ctx, &w));

  /* bad argument is reported through the return code as well */
  if (dc_set_test_args(ctx, 0.7f, 0.3f, 0) == DC_OK) { fprintf(stderr, "num_proposals=0 must be rejected\n"); return 1; }
  CHECK(dc_set_test_args(ctx, 0.7f, 0.3f, P));

  float* img = (float*)malloc((size_t)3 * H * W * sizeof(float));
  static const float mean[3] = {103.939f, 116.779f, 123.68f};   /* run_model.lua:72-74 */
  for (int c = 0; c < 3; ++c)
    for (size_t i = 0; i < (size_t)H * W; ++i) img[(size_t)c * H * W + i] = urand() * 255.f - mean[c];

  dc_result res[2];
  for (int i = 0; i < 2; ++i) {
    res[i].capacity = P; res[i].K = 0; res[i].T = 0;
    res[i].boxes = (float*)malloc((size_t)P * 4 * sizeof(float));
    res[i].scores = (float*)malloc((size_t)P * sizeof(float));
    res[i].tokens = (int32_t*)malloc((size_t)P * T * sizeof(int32_t));
  }
  CHECK(dc_forward_test(ctx, img, H, W, 0, &res[0]));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < reps; ++r) CHECK(dc_forward_test(ctx, img, H, W, 0, &res[1]));
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double ms = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (reps > 0 ? reps : 1);

  const int K = res[0].K;
  if (K < 1 || K > P || res[0].T != T) { fprintf(stderr, "bad result header K=%d T=%d\n", K, res[0].T); return 1; }
  if (res[1].K != K || memcmp(res[0].boxes, res[1].boxes, (size_t)K * 16) || memcmp(res[0].scores, res[1].scores, (size_t)K * 4) ||
      memcmp(res[0].tokens, res[1].tokens, (size_t)K * T * 4)) {
    fprintf(stderr, "results differ between identical calls\n");
    return 1;
  }
  for (int i = 0; i < K; ++i) {
    if (i > 0 && res[0].scores[i] > res[0].scores[i - 1]) { fprintf(stderr, "scores not in decreasing order at %d\n", i); return 1; }
    for (int t = 0; t < T; ++t) {
      const int tok = res[0].tokens[i * T + t];
      if (tok < 1 || tok > V + 1) { fprintf(stderr, "token out of range\n"); return 1; }
    }
  }
  printf("image %dx%d  num_proposals %d  ->  K=%d T=%d  first box (%.2f, %.2f, %.2f, %.2f) score %.4f  %.3f ms/image\n",
         W, H, P, K, T, res[0].boxes[0], res[0].boxes[1], res[0].boxes[2], res[0].boxes[3], res[0].scores[0], ms);

  /* ---- round 5 entry points, from plain C -------------------------------------------------------------------------- */
  /* (a) run_model.lua:67-74 on the device: decoder bytes in, the device tensor forward_test takes out */
  {
    const int H0 = 2 * H + 3, W0 = 2 * W + 1;                       /* a "photograph" that has to shrink */
    uint8_t* rgb = (uint8_t*)malloc((size_t)H0 * W0 * 3);
    for (size_t i = 0; i < (size_t)H0 * W0 * 3; ++i) rgb[i] = (uint8_t)(urand() * 256.0f);
    int Hs = 0, Ws = 0;
    const int size = W0 > H0 ? W : H;                               /* longer side -> the network size used above */
    CHECK(dc_preprocess_size(H0, W0, size, &Hs, &Ws));
    void* dev = NULL;
    CHECK(dc_malloc(ctx, &dev, (size_t)3 * Hs * Ws * sizeof(float)));
    CHECK(dc_preprocess_u8(ctx, rgb, H0, W0, 0, size, (float*)dev, NULL));
    CHECK(dc_forward_test(ctx, (const float*)dev, Hs, Ws, 1, &res[1]));
    printf("dc_preprocess_u8: %dx%d bytes -> %dx%d device tensor -> K=%d\n", W0, H0, Ws, Hs, res[1].K);
    CHECK(dc_free(ctx, dev));
    free(rgb);
  }
  /* (b) the opt-in split-bf16 arithmetic: another arithmetic (bits may differ), the same protocol; mode 0 restores the bits */
  {
    CHECK(dc_set_math_mode(ctx, DC_MATH_SPLIT_BF16));
    CHECK(dc_forward_test(ctx, img, H, W, 0, &res[1]));
    const int Ks = res[1].K;
    CHECK(dc_set_math_mode(ctx, DC_MATH_FP32));
    CHECK(dc_forward_test(ctx, img, H, W, 0, &res[1]));
    if (res[1].K != K || memcmp(res[0].boxes, res[1].boxes, (size_t)K * 16)) { fprintf(stderr, "fp32 bits did not come back\n"); return 1; }
    if (dc_set_math_mode(ctx, 7) == DC_OK) { fprintf(stderr, "a bad math mode was accepted\n"); return 1; }
    printf("dc_set_math_mode(1): K=%d (fp32: %d)\n", Ks, K);
  }
  /* (c) the multi-GPU gather on one GPU: a one-rank RCCL communicator, the block travels through ncclSend / ncclRecv to self */
  {
    dc_comm* comm = NULL;
    if (dc_comm_create_ex(&comm, ctx, NULL, 0, 1, DC_COMM_SELF_TRANSPORT) != DC_OK) {
      fprintf(stderr, "dc_comm_create_ex: %s\n", dc_comm_last_error(NULL));
      return 1;
    }
    dc_result g = res[1];
    g.boxes = (float*)malloc((size_t)P * 16); g.scores = (float*)malloc((size_t)P * 4); g.tokens = (int32_t*)malloc((size_t)P * T * 4);
    g.K = -1;
    if (dc_gather_results(comm, &res[0], 1, &g) != DC_OK) { fprintf(stderr, "dc_gather_results: %s\n", dc_comm_last_error(comm)); return 1; }
    if (g.K != K || memcmp(g.boxes, res[0].boxes, (size_t)K * 16) || memcmp(g.scores, res[0].scores, (size_t)K * 4) ||
        memcmp(g.tokens, res[0].tokens, (size_t)K * T * 4)) { fprintf(stderr, "gathered record differs\n"); return 1; }
    printf("dc_gather_results over \"%s\": record intact\n", dc_comm_transport(comm));
    dc_comm_destroy(comm);
  }
  dc_destroy(ctx);
  printf("HARNESS OK\n");
  return 0;
}
