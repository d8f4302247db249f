/* fav_exact.c — CPU restatement of the convolution accumulator in the EXACT
 * summation order of the HIP kernel's fp32-MFMA validation mode.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/fav_oracle.py header): PARITY UNPINNED by
 * the reference, which has no convolution on this path; this file
 * restates failure_aware_vision_amd/csrc/fav_kernels.hpp, conv_igemm_kernel<MODE=1>:
 *
 *   acc[m][n] = fmaf-chain over k' of x[m][k'] * w[n][k'],   starting from 0,
 *   k' visiting K in 64-wide tiles, each tile in two 32-wide blocks, each block
 *   as  for j in 0..7: for g in 0..3: k = 8*g + j
 *   (v_mfma_f32_16x16x4_f32 sums its four k values in lane-group order, and
 *   lane group g holds elements 8g..8g+7 of the block).
 *
 * Operands are bf16 values held in fp32, so every product is exact in fp32 and
 * fmaf(a, b, acc) == (acc + a*b) rounded once; contraction on or off gives the
 * same bits.  k = (r*kw + s)*C + c over NHWC input, zero padding.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PB 8    /* output pixels per block */
#define NB 256  /* output channels per block */

/* position k' in the summation order -> k inside the row */
static void build_order(int K, int* order) {
    int p = 0;
    for (int t = 0; t < K; t += 64)
        for (int b = 0; b < 64; b += 32)
            for (int j = 0; j < 8; ++j)
                for (int g = 0; g < 4; ++g) order[p++] = t + b + 8 * g + j;
}

/* x: [B][H][W][C] fp32, w: [N][kh][kw][C] fp32, acc: [B*Ho*Wo][N] fp32.  C % 64 == 0. */
int fav_exact_conv_acc(const float* x, const float* w, float* acc, int B, int H, int W, int C, int N, int kh, int kw,
                       int stride, int pad) {
    const int K = kh * kw * C;
    if (C % 64 != 0) return 1;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const long M = (long)B * Ho * Wo;
    int* order = (int*)malloc(sizeof(int) * K);
    float* wt = (float*)malloc(sizeof(float) * (size_t)K * N); /* [k'][n] in summation order */
    if (!order || !wt) return 2;
    build_order(K, order);
    for (int p = 0; p < K; ++p)
        for (int n = 0; n < N; ++n) wt[(size_t)p * N + n] = w[(size_t)n * K + order[p]];
    const long nblocks = (M + PB - 1) / PB;
#pragma omp parallel
    {
        float* patch = (float*)malloc(sizeof(float) * (size_t)PB * K);
        float accb[PB][NB];
#pragma omp for schedule(dynamic, 16)
        for (long blk = 0; blk < nblocks; ++blk) {
            const long m0 = blk * PB;
            const int np = (int)((M - m0) < PB ? (M - m0) : PB);
            /* gather the PB patches in summation order */
            for (int p = 0; p < np; ++p) {
                const long m = m0 + p;
                const int ow = (int)(m % Wo), oh = (int)((m / Wo) % Ho);
                const long b = m / ((long)Wo * Ho);
                float* dst = patch + (size_t)p * K;
                for (int q = 0; q < K; ++q) {
                    const int k = order[q];
                    const int c = k % C, tap = k / C, s = tap % kw, r = tap / kw;
                    const int ih = oh * stride - pad + r, iw = ow * stride - pad + s;
                    dst[q] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((b * H + ih) * W + iw) * C + c] : 0.0f;
                }
            }
            for (int n0 = 0; n0 < N; n0 += NB) {
                const int nn = (N - n0) < NB ? (N - n0) : NB;
                for (int p = 0; p < np; ++p)
                    for (int n = 0; n < nn; ++n) accb[p][n] = 0.0f;
                for (int q = 0; q < K; ++q) {
                    const float* wr = wt + (size_t)q * N + n0;
                    for (int p = 0; p < np; ++p) {
                        const float a = patch[(size_t)p * K + q];
                        float* ar = accb[p];
#pragma omp simd
                        for (int n = 0; n < nn; ++n) ar[n] = ar[n] + a * wr[n];
                    }
                }
                for (int p = 0; p < np; ++p) memcpy(acc + (size_t)(m0 + p) * N + n0, accb[p], sizeof(float) * nn);
            }
        }
        free(patch);
    }
    free(order);
    free(wt);
    return 0;
}
