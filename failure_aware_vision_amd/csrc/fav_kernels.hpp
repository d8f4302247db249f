// fav_kernels.hpp — gfx950 (MI355X / CDNA4) kernels of the failure-aware
// classification path.  Written for wave64 + MFMA + 160 KiB LDS directly; there
// is no other target.
//
// None of these kernels has a counterpart in the reference ("no
// reference counterpart exists for any row"); the numerical contract they
// implement is the one stated in oracle/fav_oracle.py and DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Phase stamps (per-block clocks written through ConvParams / TailParams / GemmSkParams::dbg) exist in the experiments build only
// (make EXPERIMENTS=1): in the normal build the pointer is a compile-time null and every stamp folds away - left in as run-time
// checks they cost the layer-4 tails 4-5 % and the layer-3 tails 1-2 % (register allocation; profiles/r4q_dbg_stamps_ab.txt).
#ifdef FAV_EXPERIMENTS
#define FAV_DBG(P) ((P).dbg)
#else
#define FAV_DBG(P) ((unsigned long long*)nullptr)
#endif

namespace fav {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// round-to-nearest-even fp32 -> bf16 bits (same integer recipe as the oracle;
// no NaN can reach it on this path: pixels are sanitised where they enter (fav_sanitize_px), weights are checked finite when loaded)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// two fp32 -> packed bf16x2 with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, the
// same result as the integer recipe above for every finite input)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// ReLU on a PACKED pair of bf16 values: as signed 16-bit integers every negative float (and -0) is negative, every non-negative one
// keeps its pattern, so max(x, 0) per half is ONE v_pk_max_i16 for two elements.  Rounding first and clamping afterwards gives the same
// bits as clamping the fp32 value first: a negative value rounds to a negative bf16 (or -0) and becomes +0 either way.
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, v), (s16x2_t){0, 0}));
}

// fp32 frames come from the caller: a NaN pixel counts as 0, everything else is clamped to [-64, 64] (far outside any image range;
// pixels in [0, 1] pass unchanged), so that no garbage frame can put a NaN or an infinity into the network - the reference's seam
// answers a bad frame with a status, never an exception (signal_analyzer.py:145-171).  oracle/fav_oracle.py: sanitize_pixels().
__device__ __forceinline__ float fav_sanitize_px(float px) {
    px = px == px ? px : 0.f;
    return fminf(fmaxf(px, -64.f), 64.f);
}

// ---------------------------------------------------------------------------
// exp and GELU as fixed sequences of IEEE fp32 operations (the ViT path).  The CPU
// oracle (oracle/fav_exact.c: fav_expf_ref, fav_gelu_ref) executes the same sequence, so the
// results are bit-identical; libm's expf / erff would differ by an ulp here and there.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fav_expf(float x) {
    if (!(x >= -80.0f)) return 0.0f;   // also NaN -> 0; keeps every result a normal number
    if (x > 88.0f) x = 88.0f;
    const float k = rintf(__fmul_rn(x, 0x1.715476p+0f));
    float r = __fmaf_rn(-k, 0x1.62e400p-1f, x);
    r = __fmaf_rn(-k, 0x1.7f7d1cp-20f, r);
    float p = 0x1.6c16c2p-10f;
    p = __fmaf_rn(p, r, 0x1.111112p-7f);
    p = __fmaf_rn(p, r, 0x1.555556p-5f);
    p = __fmaf_rn(p, r, 0x1.555556p-3f);
    p = __fmaf_rn(p, r, 0.5f);
    p = __fmaf_rn(p, r, 1.0f);
    p = __fmaf_rn(p, r, 1.0f);
    return ldexpf(p, (int)k);
}
// Correctly rounded fp32 square root: __fsqrt_rn maps to the 1-ulp hardware approximation on this target
// (measured: 11 % of random inputs differ from IEEE), so go through the correctly rounded f64 root - rounding
// that to fp32 is exact-rounded as well (53 >= 2*24 + 2 bits).
__device__ __forceinline__ float fav_sqrtf(float x) { return (float)sqrt((double)x); }
// GELU(x) = x * Phi(x) with the normal CDF (the erf form, torch.nn.GELU's default) as a fixed polynomial:
// Phi(x) = 0.5 + u q(u^2 - 0.5), u = clamp(x, +-4.25) / 4.25, q of degree 7 (Horner, fused), fitted so that
// |x Phi(x) - GELU(x)| <= 8.6e-5 everywhere (a fifth of a bf16 step at |y| = 0.1; the output is rounded to bf16) and Phi(+-4.25) rounds
// to exactly 1 / 0.  12 vector instructions; the exp-and-divide tanh form it replaced cost ~40 and is 4.7e-4 away from the erf form.
__device__ __forceinline__ float fav_gelu(float x) {
    const float u = __fmul_rn(__builtin_amdgcn_fmed3f(x, -4.25f, 4.25f), 0x1.e1e1e2p-3f);
    const float s = __fmaf_rn(u, u, -0.5f);
    float q = -0x1.22be24p+1f;
    q = __fmaf_rn(q, s, 0x1.a50adap+1f);
    q = __fmaf_rn(q, s, -0x1.2a2ab2p+1f);
    q = __fmaf_rn(q, s, 0x1.a45d50p+0f);
    q = __fmaf_rn(q, s, -0x1.4cc5a6p+0f);
    q = __fmaf_rn(q, s, 0x1.e645eap-1f);
    q = __fmaf_rn(q, s, -0x1.5fe64ep-1f);
    q = __fmaf_rn(q, s, 0x1.691206p-1f);
    return __fmul_rn(x, __fmaf_rn(u, q, 0.5f));
}

// Philox4x32-10 (Random123).  counter = (chunk, frame, sample, site), key = seed.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_lo / mul_hi pair: the integer
        // multiplier runs at a quarter of the vector rate, and the two calls per 64-channel chunk were half the
        // epilogue's issue cycles
        const unsigned long long p0 = (unsigned long long)M0 * c.x, p1 = (unsigned long long)M1 * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // a ^ b ^ c as ONE v_bitop3_b32 (truth table 0x96): gfx950 has no v_xor3, and hipcc leaves the pair of v_xor_b32 alone -
        // 20 of a call's ~100 vector instructions (headline 75.44 -> 74.73 ms in a same-box A/B, profiles/r4n_philox_bitop3_ab.txt)
        c = make_uint4(__builtin_amdgcn_bitop3_b32(hi1, c.y, k0, 0x96), lo1, __builtin_amdgcn_bitop3_b32(hi0, c.w, k1, 0x96), lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Exact unsigned division by a launch constant (Granlund-Montgomery): for n < 2^31,
// n / d == (umulhi(n, magic) + n) >> shift.  Host side: fastdiv_make().
struct FastDiv {
    uint32_t magic, shift, d;
};
inline FastDiv fastdiv_make(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.magic = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.shift = l;
    f.d = d;
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.magic) + n) >> f.shift; }

struct DropParams {
    int site;                 // -1: none
    uint32_t thr;             // 8-bit threshold: drop iff draw < thr
    float scale;
    uint32_t seed_lo, seed_hi;
    long long v0;             // virtual frame index of row 0 (v = t * n_img + i)
    int n_img;                // frames per sample
    long long first_index;    // global index of frame 0
    FastDiv div_img;          // / n_img (virtual frame indices of one launch stay below 2^31)
};

// The 16 draws (bytes, little endian over the four words) of chunk `chunk` (= element_index / 16)
// of virtual frame v (< 2^31); element 16*chunk + j is KEPT iff byte j >= d.thr.
__device__ __forceinline__ uint4 drop_draws16(const DropParams& d, uint32_t v, uint32_t chunk) {
    const uint32_t t = fastdiv(v, d.div_img);
    const uint32_t img = (uint32_t)d.first_index + (v - t * (uint32_t)d.n_img);
    return philox4x32_10(make_uint4(chunk, img, t, (uint32_t)d.site), d.seed_lo, d.seed_hi);
}

// Same draws when the sample t and the frame's index inside the launch are already known.
__device__ __forceinline__ uint4 drop_draws16_ti(const DropParams& d, uint32_t t, uint32_t img_local, uint32_t chunk) {
    return philox4x32_10(make_uint4(chunk, (uint32_t)d.first_index + img_local, t, (uint32_t)d.site), d.seed_lo, d.seed_hi);
}
// v = keep(byte j of the draws) ? v * scale : 0
#define FAV_DROP_APPLY(V, DRAWS, J, D) \
    ((((DRAWS)[(J) >> 2] >> (8 * ((J) & 3))) & 0xFFu) >= (D).thr ? __fmul_rn((V), (D).scale) : 0.f)

// ---------------------------------------------------------------------------
// Stem: frames (u8 / fp32 NHWC3) -> normalised bf16 im2col rows [n*Ho*Wo][kpad]
// k = (r*kw + s)*3 + c ; zero for padding taps and for k >= kh*kw*3.
// One thread writes 8 consecutive k (one 16-B store).
// ---------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const void* __restrict__ images, uint4* __restrict__ out,
                                                          int n, int H, int W, int Ho, int Wo, int kh, int kw,
                                                          int stride, int pad, int kpad, float m0, float m1, float m2,
                                                          float i0, float i1, float i2) {
    const int chunks = kpad >> 3;
    const long long total = (long long)n * Ho * Wo * chunks;
    const int K = kh * kw * 3;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ch = (int)(idx % chunks);
        const long long row = idx / chunks;
        const int ow = (int)(row % Wo);
        const int oh = (int)((row / Wo) % Ho);
        const long long img = row / ((long long)Wo * Ho);
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ch * 8 + j;
            float val = 0.f;
            if (k < K) {
                const int c = k % 3, tap = k / 3;
                const int s = tap % kw, r = tap / kw;
                const int ih = ih0 + r, iw = iw0 + s;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    const long long off = ((img * H + ih) * W + iw) * 3 + c;
                    float px;
                    if (LAYOUT == 0) px = __fmul_rn((float)((const uint8_t*)images)[off], 1.0f / 255.0f);
                    else px = fav_sanitize_px(((const float*)images)[off]);
                    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
                    const float istd = c == 0 ? i0 : (c == 1 ? i1 : i2);
                    val = __fmul_rn(__fsub_rn(px, mean), istd);
                }
            }
            v[j] = val;
        }
        out[idx] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                              pack_bf16x2(v[6], v[7]));
    }
}

// Row-wise variant (the one the launcher uses): a thread owns one tap row r of one output pixel, i.e.
// kw*3 values that are CONTIGUOUS both in the NHWC3 image (row ih, columns iw0 .. iw0+kw-1) and in k,
// normalises them into an LDS row [pixel][kpad], and the block then writes its pixels' rows with coalesced
// 16-B stores.  No per-element division: 0.95 -> ~0.3 ms for the 7x7/2 stem of 256 frames (the element-wise
// kernel above spends ~50 integer instructions per value on index arithmetic).
template <int LAYOUT>
__global__ __launch_bounds__(256) void stem_im2col_rows_kernel(const void* __restrict__ images, uint4* __restrict__ out,
                                                               long long total_pix, int H, int W, int Ho, int Wo, int kh, int kw,
                                                               int stride, int pad, int kpad, int ppb, float m0, float m1, float m2,
                                                               float i0, float i1, float i2) {
    extern __shared__ __attribute__((aligned(16))) uint16_t rows_s[];   // [ppb][kpad]
    const int tid = threadIdx.x;
    const int K = kh * kw * 3;
    const long long pix0 = (long long)blockIdx.x * ppb;
    for (int i = tid; i < ppb * (kpad - K); i += 256) rows_s[(i / (kpad - K)) * kpad + K + i % (kpad - K)] = 0;
    // Patch-embedding shapes (no padding, a kernel row a whole number of 16-byte pieces, rows 16-byte aligned: ViT's 16 x 16 / 16): one
    // 16-byte load per work item, items ordered piece -> pixel -> kernel row so that a wave reads contiguous runs of an image row
    // (the scalar path below reads 4 bytes per instruction at a 2.7 KB lane pitch: 0.8 TB/s at 512 frames).  Same arithmetic per value.
    constexpr int PER = LAYOUT == 0 ? 16 : 4;                 // values per 16-byte piece
    const bool vec = pad == 0 && (kw * 3) % PER == 0 && (stride * 3) % PER == 0 && (W * 3) % PER == 0 &&
                     ((uintptr_t)images & 15) == 0 && ((long long)H * W * 3) % PER == 0;
    if (vec) {
        const int qn = kw * 3 / PER;
        for (int i = tid; i < qn * ppb * kh; i += 256) {
            const int q = i % qn, pl = (i / qn) % ppb, r = i / (qn * ppb);
            const long long pixel = pix0 + pl;
            if (pixel >= total_pix) continue;
            const long long img = pixel / ((long long)Ho * Wo);
            const int rem = (int)(pixel - img * (long long)Ho * Wo);
            const int oh = rem / Wo, ow = rem - oh * Wo;
            const long long base = ((img * H + (oh * stride + r)) * W + ow * stride) * 3 + q * PER;
            uint16_t* dst = rows_s + pl * kpad + r * kw * 3 + q * PER;
            float px[PER];
            if (LAYOUT == 0) {
                const uint4 w = *(const uint4*)((const uint8_t*)images + base);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < PER; ++j) px[j] = __fmul_rn((float)((ww[j >> 2] >> (8 * (j & 3))) & 0xFFu), 1.0f / 255.0f);
            } else {
                const float4 w = *(const float4*)((const float*)images + base);
                const float wf[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) px[j & (PER - 1)] = fav_sanitize_px(wf[j]);
            }
            const int c0 = (q * PER) % 3;
            uint32_t o[PER / 2];
#pragma unroll
            for (int j = 0; j < PER; j += 2) {
                float v2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int c = (c0 + j + e) % 3;
                    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
                    const float istd = c == 0 ? i0 : (c == 1 ? i1 : i2);
                    v2[e] = __fmul_rn(__fsub_rn(px[j + e], mean), istd);
                }
                o[j / 2] = (uint32_t)f32_to_bf16_bits(v2[0]) | ((uint32_t)f32_to_bf16_bits(v2[1]) << 16);
            }
            if (PER == 4) *(uint2*)dst = make_uint2(o[0], o[1]);
            else { ((uint4*)dst)[0] = make_uint4(o[0], o[1], o[2], o[3]); ((uint4*)dst)[1] = make_uint4(o[4 % (PER / 2)], o[5 % (PER / 2)], o[6 % (PER / 2)], o[7 % (PER / 2)]); }
        }
    }
    const int pl = tid / kh, r = tid - pl * kh;
    const long long pixel = pix0 + pl;
    if (!vec && pl < ppb && pixel < total_pix) {
        const long long img = pixel / ((long long)Ho * Wo);
        const int rem = (int)(pixel - img * (long long)Ho * Wo);
        const int oh = rem / Wo, ow = rem - oh * Wo;
        const int ih = oh * stride - pad + r, iw0 = ow * stride - pad;
        const bool rowok = (unsigned)ih < (unsigned)H;
        const long long base = ((img * H + ih) * W + iw0) * 3;
        uint16_t* dst = rows_s + pl * kpad + r * kw * 3;
        for (int sx = 0; sx < kw; ++sx) {
            const bool ok = rowok && (unsigned)(iw0 + sx) < (unsigned)W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float val = 0.f;
                if (ok) {
                    float px;
                    if (LAYOUT == 0) px = __fmul_rn((float)((const uint8_t*)images)[base + sx * 3 + c], 1.0f / 255.0f);
                    else px = fav_sanitize_px(((const float*)images)[base + sx * 3 + c]);
                    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
                    const float istd = c == 0 ? i0 : (c == 1 ? i1 : i2);
                    val = __fmul_rn(__fsub_rn(px, mean), istd);
                }
                dst[sx * 3 + c] = (uint16_t)f32_to_bf16_bits(val);
            }
        }
    }
    __syncthreads();
    const int cpr = kpad >> 3;   // 16-B chunks per row
    for (int i = tid; i < ppb * cpr; i += 256) {
        const long long pixel2 = pix0 + i / cpr;
        if (pixel2 < total_pix) out[pixel2 * cpr + (i % cpr)] = *(const uint4*)(rows_s + (size_t)i * 8);
    }
}

// ---------------------------------------------------------------------------
// The ImageNet stem in one launch: normalise -> 7x7 / 2 convolution to 64 channels -> bias -> ReLU -> bf16 ->
// 3x3 / 2 max pool.  The three-launch form (im2col rows, GEMM, pool) moves 384 B + 128 B + 128 B per conv pixel
// through HBM; here a block owns an 8 x 8 tile of POOL pixels, i.e. a 17 x 17 tile of conv pixels on a 39 x 39 patch
// of input pixels, and nothing but the frames and the pooled tile crosses HBM:
//   1. the patch is normalised and rounded ONCE, into 17 column strips in LDS: strip lx holds, for each of the 39 patch
//      rows y, the 21 values (7 pixels x 3 channels) under conv column lx.  k = (r*7 + s)*3 + c then makes the 147-long
//      im2col row of conv pixel (ly, lx) the CONTIGUOUS run strip[lx][2*ly .. 2*ly + 6][0 .. 20], so the B fragment of
//      k-step s is one 16-byte LDS read at 1640*lx + 84*ly + 64*s + 16*(lane / 16) - no im2col matrix, no gather
//      arithmetic (k >= 147 is masked to zero, as the padded matrix has it).  A patch value lies in up to four strips;
//      the thread that loaded it writes each copy at (per-thread base) + (compile-time row offset);
//   2. the whole [64][192] weight matrix sits in registers as A fragments (96 VGPRs), loaded once per block; a wave
//      takes 16 conv pixels x 64 channels x 6 k-steps per round - the same MFMA and the same k order as the GEMM over
//      the im2col matrix, so the results are bit-identical to the three-launch form;
//   3. (acc + bias) -> ReLU -> bf16 into an LDS image of the conv tile, then the max over the 3 x 3 windows (taps
//      outside the conv map re-read the centre tap; bf16 values >= +0 order like their bit patterns: packed u16 max);
//   4. the raw pixels of the block's NEXT tile are requested into registers before the convolution of this one.
// ---------------------------------------------------------------------------
struct StemPoolParams {
    const void* images;
    const uint16_t* w;       // [64][192] bf16, k = (r*7 + s)*3 + c, zero for k >= 147
    const float* bias;       // [64]
    uint16_t* out;           // [n][Hp][Wp][64] bf16
    int n, H, W, Hc, Wc, Hp, Wp;
    int tiles_y, tiles_x;
    long long tiles;         // n * tiles_y * tiles_x
    float m0, m1, m2, i0, i1, i2;
    long long g_w, g_bias, g_out;   // grouped launch: byte strides per member (blockIdx.y); the frames are shared
};

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(4))) StemFrag { uint32_t x, y, z, w; };     // a 16-byte LDS read at a 4-byte aligned address

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void stem7_pool_kernel(StemPoolParams p) {
    p.w = (const uint16_t*)((const char*)p.w + (long long)blockIdx.y * p.g_w);
    p.bias = (const float*)((const char*)p.bias + (long long)blockIdx.y * p.g_bias);
    p.out = (uint16_t*)((char*)p.out + (long long)blockIdx.y * p.g_out);
    constexpr int PT = 8, CT = 2 * PT + 1, NPIX = CT * CT, NGRP = (NPIX + 15) / 16;
    constexpr int PROWS = 2 * CT + 5, PVALS = PROWS * 3;          // 39 patch rows of 39 pixels = 117 values
    constexpr int SPITCH = PROWS * 21 + 1;                        // 820 elements per strip: every window 4-byte aligned
    constexpr int ESZ = LAYOUT == 0 ? 1 : 4;
    __shared__ __attribute__((aligned(16))) uint16_t strip_s[CT * SPITCH + 44];   // the last pixel's k = 147..159 reads run 11 past
    __shared__ __attribute__((aligned(16))) uint16_t conv_s[NGRP * 16 * 64];      // [conv pixel][64], 16-B units XOR-swizzled by pixel
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    uint4 wf[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int s = 0; s < 6; ++s) wf[a][s] = *(const uint4*)(p.w + (16 * a + fr) * 192 + 32 * s + 8 * fq);
    if (tid < 64) bias_s[tid] = p.bias[tid];
    // k-step 4 holds k = 128 .. 159: the lanes of quarter 2 keep k = 144, 145, 146, quarter 3 nothing
    const uint32_t km1 = fq < 2 ? 0xFFFFFFFFu : (fq == 2 ? 0x0000FFFFu : 0u), km0 = fq < 3 ? 0xFFFFFFFFu : 0u, km23 = fq < 2 ? 0xFFFFFFFFu : 0u;

    // patch fill: a thread owns patch column fe = x*3 + c (128 threads per row, 117 used) of rows frow0, frow0 + 2, ...
    const int fe = tid & 127, frow0 = tid >> 7;
    const int fx = fe / 3, fc = fe - 3 * fx;
    const float fmean = fc == 0 ? p.m0 : (fc == 1 ? p.m1 : p.m2);
    const float fistd = fc == 0 ? p.i0 : (fc == 1 ? p.i1 : p.i2);
    // the strips this column lies in: lx with 0 <= x - 2*lx <= 6; element (y, fe) of strip lx sits at 814*lx + 21*y + fe
    const int lx_lo = fx < 6 ? 0 : (fx - 5) >> 1, lx_hi = (fx >> 1) < CT - 1 ? (fx >> 1) : CT - 1;
    uint32_t sdst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lxj = lx_lo + j <= lx_hi ? lx_lo + j : lx_hi;       // fewer than four strips: the last one is written again
        sdst[j] = (uint32_t)(2 * ((SPITCH - 6) * lxj + 21 * frow0 + fe));
    }
    const long long tiles_img = (long long)p.tiles_y * p.tiles_x;
    const long long frame_bytes = (long long)p.H * p.W * 3 * ESZ;

    constexpr int NFILL = (PROWS + 1) / 2;
    uint32_t pv[NFILL];      // raw pixels (u8 value or fp32 bits) of the block's next tile
    uint32_t pok = 0;        // bit i: pv[i] lies inside the frame
#define FAV_STEM_REQUEST(TILE)                                                                           \
    do {                                                                                                \
        const long long tq_ = (TILE);                                                                   \
        const long long t_ = tq_ < p.tiles ? tq_ : p.tiles - 1;                                         \
        const long long img_ = t_ / tiles_img;                                                          \
        const int trem_ = (int)(t_ - img_ * tiles_img);                                                 \
        const int ty_ = trem_ / p.tiles_x, tx_ = trem_ - ty_ * p.tiles_x;                               \
        const int iy0_ = 4 * ty_ * PT - 5 + frow0, ix_ = 4 * tx_ * PT - 5 + fx;                         \
        const bool colok_ = fe < PVALS && (unsigned)ix_ < (unsigned)p.W;                                \
        const __amdgpu_buffer_rsrc_t srd_ = __builtin_amdgcn_make_buffer_rsrc(                          \
            (void*)((const char*)p.images + img_ * frame_bytes), 0, (int)frame_bytes, 0x00020000);      \
        const int col_ = (ix_ * 3 + fc) * ESZ, rowb_ = p.W * 3 * ESZ;                                   \
        pok = 0;                                                                                        \
        _Pragma("unroll") for (int i = 0; i < NFILL; ++i) {                                             \
            const int iy_ = iy0_ + 2 * i;                                                               \
            const bool ok_ = colok_ && (unsigned)iy_ < (unsigned)p.H && (2 * i + 1 < PROWS || frow0 == 0); \
            const uint32_t off_ = ok_ ? (uint32_t)(iy_ * rowb_ + col_) : 0x80000000u;                   \
            if (LAYOUT == 0) pv[i] = __builtin_amdgcn_raw_buffer_load_b8(srd_, off_, 0, 0);             \
            else pv[i] = __builtin_amdgcn_raw_buffer_load_b32(srd_, off_, 0, 0);                        \
            pok |= ok_ ? 1u << i : 0u;                                                                  \
        }                                                                                               \
    } while (0)
    FAV_STEM_REQUEST(blockIdx.x);

    for (long long tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const long long img = tile / tiles_img;
        const int trem = (int)(tile - img * tiles_img);
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int py0 = ty * PT, px0 = tx * PT;
        if (fe < PVALS) {
#pragma unroll
            for (int i = 0; i < NFILL; ++i) {
                if (2 * i + 1 >= PROWS && frow0 != 0) break;     // 39 rows: the odd rows stop one short
                const float px = LAYOUT == 0 ? __fmul_rn((float)pv[i], 1.0f / 255.0f) : fav_sanitize_px(__uint_as_float(pv[i]));
                const float val = (pok >> i & 1) ? __fmul_rn(__fsub_rn(px, fmean), fistd) : 0.f;
                const uint16_t hb = (uint16_t)pack_bf16x2(val, 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) *(uint16_t*)((char*)strip_s + sdst[j] + 84 * i) = hb;
            }
        }
        __syncthreads();
        FAV_STEM_REQUEST(tile + gridDim.x);

        for (int grp = wave; grp < NGRP; grp += 4) {
            const int q = grp * 16 + fr;
            const int qc = q < NPIX ? q : NPIX - 1;              // the last group's spare pixels recompute pixel 288; never pooled
            const int ly = qc / CT, lx = qc - ly * CT;
            const char* win = (const char*)strip_s + 2 * (SPITCH * lx + 42 * ly) + 16 * fq;
            f32x4_t acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                union { uint4 u; bf16x8_t v; } ub;
                ub.u = make_uint4(0, 0, 0, 0);
                if (s < 5) {
                    const StemFrag f = *(const StemFrag*)(win + 64 * s);
                    ub.u = s < 4 ? make_uint4(f.x, f.y, f.z, f.w) : make_uint4(f.x & km0, f.y & km1, f.z & km23, f.w & km23);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    union { uint4 u; bf16x8_t v; } ua;
                    ua.u = wf[a][s];
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a], 0, 0, 0);
                }
            }
            char* crow = (char*)conv_s + q * 128 + ((fq & 1) << 3);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 bq = *(const float4*)(bias_s + 16 * a + 4 * fq);
                const float v0 = __fadd_rn(acc[a][0], bq.x), v1 = __fadd_rn(acc[a][1], bq.y);
                const float v2 = __fadd_rn(acc[a][2], bq.z), v3 = __fadd_rn(acc[a][3], bq.w);
                *(uint2*)(crow + (((2 * a + (fq >> 1)) ^ (q & 7)) << 4)) = make_uint2(relu_bf16x2(pack_bf16x2(v0, v1)), relu_bf16x2(pack_bf16x2(v2, v3)));
            }
        }
        __syncthreads();

#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + 256 * it;
            const int u = item & 7, pp = item >> 3;
            const int ppy = pp >> 3, ppx = pp & 7;
            const int gpy = py0 + ppy, gpx = px0 + ppx;
            if (gpy < p.Hp && gpx < p.Wp) {
                const int qm = (2 * ppy + 1) * CT + 2 * ppx + 1;        // the centre tap: always inside the conv map
                u16x2_t best[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const bool rok = (unsigned)(2 * gpy - 1 + r) < (unsigned)p.Hc;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const bool ok = rok && (unsigned)(2 * gpx - 1 + s) < (unsigned)p.Wc;
                        const int q = ok ? qm + (r - 1) * CT + (s - 1) : qm;
                        const uint4 v = *(const uint4*)((const char*)conv_s + q * 128 + ((u ^ (q & 7)) << 4));
                        best[0] = __builtin_elementwise_max(best[0], __builtin_bit_cast(u16x2_t, v.x));
                        best[1] = __builtin_elementwise_max(best[1], __builtin_bit_cast(u16x2_t, v.y));
                        best[2] = __builtin_elementwise_max(best[2], __builtin_bit_cast(u16x2_t, v.z));
                        best[3] = __builtin_elementwise_max(best[3], __builtin_bit_cast(u16x2_t, v.w));
                    }
                }
                *(uint4*)(p.out + (((img * p.Hp + gpy) * p.Wp + gpx) << 6) + 8 * u) =
                    make_uint4(__builtin_bit_cast(uint32_t, best[0]), __builtin_bit_cast(uint32_t, best[1]),
                               __builtin_bit_cast(uint32_t, best[2]), __builtin_bit_cast(uint32_t, best[3]));
            }
        }
        // the next tile's fill only writes strip_s (its readers are past the barrier above); conv_s is rewritten behind the
        // next barrier, which every pool reader reaches first
    }
#undef FAV_STEM_REQUEST
}

// ---------------------------------------------------------------------------
// Convolution as implicit-im2col GEMM on MFMA.
//   Y[m, n] = sum_k A[m, k] * Wt[n, k],  m = (frame, oh, ow), k = (r, s, c)
// NHWC activations make every 64-wide K tile one contiguous 128-B run of one
// input pixel (Cin % 64 == 0), so the A tile is gathered with per-row base
// addresses and zero fill for padding taps; weights are [Cout][K] so both MFMA
// operands are K-contiguous.
//
// Block: (BM/64) x WN waves (4 for the 128-row tiles, 8 for 256 x 256), tile BM x BN x BK with BK = 32
// or 64 channels of one tap, an NS-stage LDS ring filled by LDS-DMA, and an XOR swizzle on the staged rows
// so the ds_read_b128 fragment reads are bank-conflict free.  MFMA orientation: A-operand = weights
// (rows n), B-operand = activations (cols m), so a lane's 4 accumulator registers are 4 consecutive
// output channels of one output pixel -> one 16-B LDS write in the epilogue.
// Epilogue: fp32 tile staged through LDS 64 rows at a time, then 16 consecutive channels per thread:
// ((acc + bias) + residual) -> ReLU / GELU -> dropout -> one bf16 rounding -> two 16-B stores; every
// global load of the epilogue (bias, residual) is issued BEFORE the K loop (see the comments there).
// ---------------------------------------------------------------------------
// One LDS-DMA piece: buffer_load_dwordx4 ... lds moves 64 lanes x 16 B from
// (descriptor base + per-lane voffset + uniform soffset) to 1 KiB of LDS at the
// wave-uniform address in M0.  A lane whose voffset fails the descriptor's range
// check (we use 0x80000000 for padding taps and rows beyond M) contributes zeros,
// which is exactly the im2col zero padding.  Inline asm on purpose: hipcc orders a
// builtin LDS-DMA against every later ds_read with s_waitcnt vmcnt(0), which
// serialises the prefetch of tile k+1 with the MFMAs of tile k; an asm statement is
// not counted, and the kernel retires it itself (vmcnt(0) + barrier before the
// stage is read).  M0 is saved/restored in the same statement (the compiler owns it).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t srd, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
    // M0 is written and read inside one statement and declared clobbered; the kernel uses
    // no other M0 consumer (no builtin LDS-DMA, no s_movrel), so nothing needs restoring.
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
                 : "memory", "m0");
}

struct ConvParams {
    const uint16_t* x;
    const uint16_t* w;
    const float* bias;
    const uint16_t* res;
    void* y;
    int H, W, Cin, Ho, Wo, HWo;
    int Cout;       // real output channels (stores masked beyond)
    int ldy;        // output row stride in elements
    int kw, stride, pad;
    int M;          // n_frames * Ho * Wo
    int K, nk;      // K = kh*kw*Cin, nk = K / BK
    int relu, out_f32;   // relu: 0 none, 1 ReLU, 2 GELU
    int tiles_m, tiles_n;
    int stage_mid;  // issue a 64-deep step's DMA after its first MFMA group (3x3) or in front (1x1)
    DropParams drop;
    FastDiv div_hwo;  // / HWo
    FastDiv div_w;    // / Wo (the staged 3x3 kernel)
    unsigned long long* dbg;  // FAV_CONV_DBG: per-block phase timestamps (null in normal runs)
    // grouped launch (the members of a deep ensemble in one launch): block row blockIdx.y works on the tensors
    // g_* BYTES behind these, per member (0: shared by all members)
    long long g_x, g_w, g_bias, g_res, g_y;
};

// the parameters of group member blockIdx.y
__device__ __forceinline__ ConvParams conv_group_params(const ConvParams& q) {
    ConvParams p = q;
    const long long g = blockIdx.y;
    p.x = (const uint16_t*)((const char*)q.x + g * q.g_x);
    p.w = (const uint16_t*)((const char*)q.w + g * q.g_w);
    p.bias = (const float*)((const char*)q.bias + g * q.g_bias);
    if (q.res) p.res = (const uint16_t*)((const char*)q.res + g * q.g_res);
    p.y = (void*)((char*)q.y + g * q.g_y);
    return p;
}

// LDS swizzle of a staged K tile: rows are BK*2 bytes, a row holds BK/8 16-B chunks.
// Physical chunk = logical chunk ^ swz(row), chosen so the four 16-lane groups a
// ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) hit 16
// distinct 16-B slots of the 256-B bank row:
//   BK = 64 (128-B rows, 2 rows per bank row): swz = row & 7
//   BK = 32 ( 64-B rows, 4 rows per bank row): swz = (-(row >> 2)) & 3
template <int BK>
__device__ __forceinline__ int lds_swz(int row) {
    return BK == 64 ? (row & 7) : ((-(row >> 2)) & 3);
}

// waves per SIMD the LDS footprint allows (what __launch_bounds__ should ask for)
constexpr int conv_waves_per_simd(int BM, int BN, int BK, int NS) {
    const int stage = NS * (BM + BN) * BK * 2, out = 64 * (BN + 4) * 4;
    const int lds = (stage > out ? stage : out) + BN * 5;
    int blocks = 163840 / lds;
    if (blocks > 4) blocks = 4;
    if (BN >= 128 && blocks > 3) blocks = 3;  // 64 accumulators + the prefetched residual need > 128 VGPRs
    if (blocks < 1) blocks = 1;
    return blocks * (BM * 2 / 64) / 4;
}

// EPI = 1 (production): the epilogue runs in registers.  The weight rows of the B tile are staged in the order
// tail_row_perm() gives (within every 64-row group LDS row 16a + i holds weight row 16*(i/4) + 4a + i%4), so lane
// (pixel frow, quad fq) ends up with 16 CONSECUTIVE output channels of its pixel per group: bias, residual, ReLU /
// GELU, Philox dropout (one call = exactly its 16 draws) and the bf16 rounding need no fp32 staging through LDS and no
// barrier.  (Tiles with 32 columns per wave use 32-row groups and 8 channels per lane.)  The sums are unchanged - an
// output element still meets its products in ascending k.  EPI = 0 is the round-1 epilogue through an fp32 LDS stage.
// GELU: the GELU epilogue (ViT MLP) is compiled in.  The ResNet launches use the instantiations without it: ~40 vector
// instructions per element that never run still cost the 250-register kernels their schedule.
template <int BM, int BN, int BK, int NS, int MODE, int WN = 2, int OCCW = 0, int EPI = 1, int PP = 0, bool GELU = false>
__global__ __launch_bounds__((BM / 64) * WN * 64, OCCW ? OCCW : conv_waves_per_simd(BM, BN, BK, NS)) void conv_igemm_kernel(const ConvParams p_launch) {
    const ConvParams p = conv_group_params(p_launch);
    constexpr int NT = (BM / 64) * WN * 64;     // threads: (BM/64) x WN waves, each a 64 x BN/WN sub-tile
    constexpr int NWAVES = NT / 64;
    constexpr int ROWB = BK * 2;                // bytes per staged row
    constexpr int CPR = BK / 8;                 // 16-B chunks per row
    constexpr int PROWS = 1024 / ROWB;          // rows per 1-KiB LDS-DMA piece
    constexpr int WTM = 64, WTN = BN / WN;      // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16; // 16x16 MFMA tiles per wave
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int OUT_LD = BN + 4;              // fp32 staging row stride (floats)
    constexpr int OUT_BYTES = 64 * OUT_LD * 4;  // the epilogue stages 64 rows (one wave row) at a time
    constexpr int LDS_BYTES = (EPI == 1 || NS * STAGE_BYTES > OUT_BYTES) ? NS * STAGE_BYTES : OUT_BYTES;
    constexpr int GS = WTN >= 64 ? 64 : 32;     // EPI 1: rows per permuted group of the B tile
    constexpr int CPL = GS / 4;                 // ... and consecutive channels a lane holds per group (16 or 8)
    constexpr int NG = WTN / GS;                // ... groups per wave
    constexpr int AR = A_BYTES / 1024 / NWAVES, BR = B_BYTES / 1024 / NWAVES;  // LDS-DMA pieces per wave per tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + BN * 5];  // + this tile's bias, 16-channel chunks 20 floats apart

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 4ull] = wall_clock64();

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (L2); give each XCD a
    // contiguous run of tiles, n fastest, so the A tile of one m is re-read from
    // that XCD's L2 by its n neighbours.
    const int nwg = gridDim.x;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_n = tile % p.tiles_n;
    const int tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // The tile's bias goes to LDS now: on gfx9 stores count in vmcnt like loads, so ANY global
    // load inside the epilogue would wait for every store issued before it (a full write
    // round trip per pass).  The epilogue therefore issues all its loads before its first store.
    // Chunk c of 16 channels lives at float offset 20*c: the epilogue's lanes read 8 (BN = 128) or 16 (BN = 256)
    // different chunks with one ds_read_b128, and at a stride of 16 floats chunks c and c+4 share their banks
    // (2- and 4-way conflicts: the 13 % / 18 % LDS conflict cycles of profiles/r1e); 5*c mod 16 is distinct for c < 16.
    if (tid < BN) ((float*)(smem + LDS_BYTES))[(tid >> 4) * 20 + (tid & 15)] = p.bias[n0 + tid];

    // ---- epilogue geometry, and the residual prefetch -------------------------------
    // A thread finishes 16 consecutive channels of one output pixel per pass.  The residual
    // rows of the WHOLE tile are requested here, in front of the K loop: their latency hides
    // under it and no load is left to issue once the stores begin.  Tiles whose residual
    // would not fit in registers (256 x 256) fetch it one 64-row group at a time instead.
    constexpr int NCH = BN / 16;           // 16-channel chunks per row
    constexpr int RPP = (NT / NCH > 64) ? 64 : NT / NCH;  // rows per pass (threads beyond 64 rows idle)
    constexpr int NPASS = 64 / RPP;        // passes per 64-row group
    constexpr int NGROUPS = BM / 64;
    const int ec = tid % NCH, er = tid / NCH;
    const int n = n0 + ec * 16;
    const float* bias_s = (const float*)(smem + LDS_BYTES) + ec * 20;
    constexpr bool RES_ALL = NGROUPS * NPASS * 2 <= 8;
    constexpr int RG = RES_ALL ? NGROUPS : 1;
    u32x4_t rres[RG][NPASS][2];
    // EPI 1: lane (frow, fq) of wave (wm, wn) finishes, for b < TM and g < NG, the CPL channels
    // n0 + wn*WTN + g*GS + CPL*fq .. of pixel m0 + wm*64 + b*16 + frow.  Residual: all of the tile's rows before the
    // K loop when they fit in 8 registers quads, else one pixel tile b ahead of the stores (two sets).
    constexpr int RQ = CPL / 8;                              // 16-B quads per (b, g)
    constexpr bool RES_ALL1 = TM * NG * RQ <= 8;
    constexpr int RB = RES_ALL1 ? TM : 2;
    u32x4_t rr1[RB][NG][RQ];
    const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
    const int esz = p.out_f32 ? 4 : 2;
    const __amdgpu_buffer_rsrc_t srd_res1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.res ? p.res + (long long)m0 * p.ldy : p.x), 0, p.res ? rows_valid * p.ldy * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_y1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.y + (long long)m0 * p.ldy * esz), 0, rows_valid * p.ldy * esz, 0x00020000);
#define FAV_LOAD_RES1(SLOT, B)                                                                   \
    _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_)                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < RQ; ++q_)                                        \
            rr1[SLOT][g_][q_] = __builtin_amdgcn_raw_buffer_load_b128(                           \
                srd_res1, ((wm * 64 + (B) * 16 + (lane & 15)) * p.ldy + n0 + wn * WTN + g_ * GS + CPL * (lane >> 4) + 8 * q_) * 2, 0, 0)
#define FAV_RES1_LANDED(SLOT)                                                                    \
    _Pragma("unroll") for (int g_ = 0; g_ < NG; ++g_)                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < RQ; ++q_) asm volatile("" : "+v"(rr1[SLOT][g_][q_]))
    if (EPI == 1) {
#pragma unroll
        for (int b_ = 0; b_ < RB; ++b_)
#pragma unroll
            for (int g_ = 0; g_ < NG; ++g_)
#pragma unroll
                for (int q_ = 0; q_ < RQ; ++q_) rr1[b_][g_][q_] = (u32x4_t){0u, 0u, 0u, 0u};
        if (p.res) {
            if (RES_ALL1) {
#pragma unroll
                for (int b_ = 0; b_ < TM; ++b_) FAV_LOAD_RES1(b_, b_);
            } else {
                FAV_LOAD_RES1(0, 0);
            }
        }
    }
#define FAV_LOAD_RES(G, HALF)                                                              \
    _Pragma("unroll") for (int pass = 0; pass < NPASS; ++pass) {                           \
        const int m = m0 + (HALF) * 64 + er + pass * RPP;                                  \
        if (er < 64 && m < p.M && n < p.Cout) {                                            \
            const u32x4_t* rp = (const u32x4_t*)(p.res + (long long)m * p.ldy + n);        \
            rres[G][pass][0] = rp[0];                                                      \
            rres[G][pass][1] = rp[1];                                                      \
        }                                                                                  \
    }
// Empty asm that "redefines" the residual registers: the compiler puts its one wait for the
// loads in front of it and forgets them afterwards, so later uses (after stores have been
// issued) do not turn into waits for those stores.
#define FAV_RES_LANDED()                                                                   \
    _Pragma("unroll") for (int g_ = 0; g_ < RG; ++g_)                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < NPASS; ++q_)                               \
            asm volatile("" : "+v"(rres[g_][q_][0]), "+v"(rres[g_][q_][1]))
#pragma unroll
    for (int g_ = 0; g_ < RG; ++g_)
#pragma unroll
        for (int q_ = 0; q_ < NPASS; ++q_) rres[g_][q_][0] = rres[g_][q_][1] = (u32x4_t){0u, 0u, 0u, 0u};
    if (EPI == 0 && RES_ALL && p.res) {
#pragma unroll
        for (int g = 0; g < NGROUPS; ++g) FAV_LOAD_RES(g, g)
    }

    // ---- per-lane gather descriptors ------------------------------------------
    // Tiles are staged with LDS-DMA: one wave instruction writes 1 KiB = PROWS rows x
    // ROWB bytes linearly into LDS, lane l -> row l / CPR, 16-B slot l % CPR.  The
    // bank-conflict swizzle therefore goes on the SOURCE: the lane that owns physical
    // slot s of row r fetches logical chunk s ^ swz(r) (pieces start at multiples of
    // PROWS rows, so swz(r) depends on the lane only), and fragment reads apply the
    // same XOR.
    //
    // Addressing is split so the K loop does almost no vector arithmetic:
    //   address = [descriptor base: first frame of the tile, shifted back by the
    //              padding so every offset is >= 0]                      (uniform)
    //           + voffset: this lane's row (frame, oh*stride, ow*stride, chunk)  (per lane, fixed)
    //           + soffset: the K tile's tap (r, s) and channel offset     (uniform, per tile)
    // and a padding tap / row beyond M just swaps voffset for an out-of-range value.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane / CPR;
    const int lch = (lane % CPR) ^ lds_swz<BK>(lrow);
    const long long frame_elems = (long long)p.H * p.W * p.Cin;
    const int vimg0 = m0 / p.HWo;
    const int pad_shift = (p.pad * p.W + p.pad) * p.Cin;
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + vimg0 * frame_elems - pad_shift), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long long)n0 * p.K), 0, BN * p.K * 2, 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;
    uint32_t a_voff[AR];
    int a_ih0[AR], a_iw0[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + (wave * AR + i) * PROWS + lrow;
        if (m < p.M) {
            // magic-number divisions (two per staged row; as plain `/` they were ~300 of the ~440 instructions in front of
            // the first LDS-DMA request)
            const int vimg = (int)fastdiv((uint32_t)m, p.div_hwo);
            const int pix = m - vimg * p.HWo;
            const int oh = (int)fastdiv((uint32_t)pix, p.div_w), ow = pix - oh * p.Wo;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_voff[i] = (uint32_t)(((long long)(vimg - vimg0) * frame_elems +
                                    (long long)(a_ih0[i] * p.W + a_iw0[i]) * p.Cin + pad_shift + lch * 8) * 2);
        } else {
            a_ih0[i] = -0x40000000;  // never in range
            a_iw0[i] = 0;
            a_voff[i] = OOB;
        }
    }
    uint32_t b_voff[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        int r = (wave * BR + i) * PROWS + lrow;      // LDS row of the B tile this lane fills
        if (EPI == 1) r = (r & ~(GS - 1)) + CPL * ((r & 15) >> 2) + 4 * ((r >> 4) & (GS / 16 - 1)) + (r & 3);   // weight row it holds
        b_voff[i] = (uint32_t)((r * p.K + lch * 8) * 2);
    }
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)smem);
    const uint32_t lds_a = lds_base + wave_u * (AR * 1024);
    const uint32_t lds_b = lds_base + A_BYTES + wave_u * (BR * 1024);
    int tap_r = 0, tap_s = 0, c0 = 0;  // K-tile position: tap (r, s), channel offset
    const bool has_taps = p.K != p.Cin || p.pad != 0 || p.stride != 1;  // uniform

// Issue the LDS-DMA of K tile KT into stage BUF (asynchronous; retired by the
// explicit vmcnt(0) in front of the barrier that ends the K step).
#define FAV_STAGE(BUF, KT)                                                                              \
    do {                                                                                                \
        const uint32_t soff_a = (uint32_t)(((tap_r * p.W + tap_s) * p.Cin + c0) * 2);                   \
        if (has_taps) {                                                                                 \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                            \
                const bool ok = (unsigned)(a_ih0[i] + tap_r) < (unsigned)p.H &&                         \
                                (unsigned)(a_iw0[i] + tap_s) < (unsigned)p.W;                           \
                lds_dma16(srd_a, ok ? a_voff[i] : OOB, soff_a, lds_a + (BUF) * STAGE_BYTES + i * 1024); \
            }                                                                                           \
        } else { /* 1x1, no padding: a row is either always valid or always beyond M */               \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                              \
                lds_dma16(srd_a, a_voff[i], soff_a, lds_a + (BUF) * STAGE_BYTES + i * 1024);            \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < BR; ++i)                                                  \
            lds_dma16(srd_b, b_voff[i], (uint32_t)((KT) * ROWB), lds_b + (BUF) * STAGE_BYTES + i * 1024); \
        c0 += BK;                                                                                       \
        if (c0 == p.Cin) {                                                                              \
            c0 = 0;                                                                                     \
            if (++tap_s == p.kw) { tap_s = 0; ++tap_r; }                                                \
        }                                                                                               \
    } while (0)

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;  // row of the 16-row fragment this lane reads
    const int fq = lane >> 4;    // which 8-element k chunk of a 32-wide k step

    // NS-stage ring: tiles kt+1 .. kt+NS-1 are in flight while tile kt is multiplied.
    // vmcnt counts this wave's LDS-DMA pieces in issue order, so "all but the newest
    // (NS-2) tiles have landed" is the counted wait below; the barrier then makes
    // every wave's share of tile kt+1 visible.
    constexpr int PIECES = AR + BR;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < p.nk) FAV_STAGE(t, t);
    if (p.nk >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 4ull + 1] = wall_clock64();

    int cur = 0, nxt = NS - 1;  // stage being read / stage being filled
    if constexpr (PP == 1) {
        // ---- ping-pong K loop (256 x 256 x 64 tile, 8 waves, two stages) ---------------------------------------------
        // Waves w and w + 4 share a SIMD.  Re-aligned by the per-step barrier they would both read fragments, then both
        // multiply: LDS and the matrix pipe take turns.  Here waves 4..7 run half a step behind - they carry the fragments
        // of a tile's second 32-deep half across the barrier and multiply them while waves 0..3 read - so one wave of a
        // SIMD feeds the matrix pipe while the other one reads.  Every accumulator still meets its products in ascending k.
        static_assert(BK == 64 && NS == 2 && MODE == 0 && NT == 512, "ping-pong loop: 8 waves, 64-deep steps, two stages");
#define FAV_PP_READ(FX, FW, KK)                                                                          \
    do {                                                                                                \
        _Pragma("unroll") for (int b = 0; b < TM; ++b) {                                                \
            const int row = wm * WTM + b * 16 + frow;                                                   \
            FX[b] = *(const uint4*)(As + row * ROWB + ((((KK) * 4 + fq) ^ lds_swz<BK>(row)) << 4));     \
        }                                                                                               \
        _Pragma("unroll") for (int a = 0; a < TN; ++a) {                                                \
            const int row = wn * WTN + a * 16 + frow;                                                   \
            FW[a] = *(const uint4*)(Bs + row * ROWB + ((((KK) * 4 + fq) ^ lds_swz<BK>(row)) << 4));     \
        }                                                                                               \
    } while (0)
#define FAV_PP_MMA(FX, FW)                                                                               \
    do {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int a = 0; a < TN; ++a)                                                  \
            _Pragma("unroll") for (int b = 0; b < TM; ++b) {                                            \
                union { uint4 u; bf16x8_t v; } ua, ub;                                                  \
                ua.u = FW[a]; ub.u = FX[b];                                                             \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a][b], 0, 0, 0);    \
            }                                                                                           \
        __builtin_amdgcn_s_setprio(0);                                                                  \
    } while (0)
#define FAV_PP_STEP_END()                                                                                \
    do {                                                                                                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        __syncthreads();                                                                                \
        cur ^= 1; nxt ^= 1;                                                                             \
    } while (0)
        if (wave_u < NWAVES / 2) {
            for (int kt = 0; kt < p.nk; ++kt) {
                const unsigned char* As = smem + cur * STAGE_BYTES;
                const unsigned char* Bs = As + A_BYTES;
                uint4 fx[TM], fw[TN];
                FAV_PP_READ(fx, fw, 0);
                if (kt + 1 < p.nk) FAV_STAGE(nxt, kt + 1);
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_MMA(fx, fw);
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_READ(fx, fw, 1);
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_MMA(fx, fw);
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_STEP_END();
            }
        } else {
            uint4 hx[TM], hw[TN];        // second half of the previous tile, carried across the barrier
            for (int kt = 0; kt < p.nk; ++kt) {
                const unsigned char* As = smem + cur * STAGE_BYTES;
                const unsigned char* Bs = As + A_BYTES;
                if (kt > 0) FAV_PP_MMA(hx, hw);
                if (kt + 1 < p.nk) FAV_STAGE(nxt, kt + 1);      // (issuing the DMA first in the step measured 3-6 % slower)
                __builtin_amdgcn_sched_barrier(0);
                uint4 fx[TM], fw[TN];
                FAV_PP_READ(fx, fw, 0);
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_MMA(fx, fw);
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_READ(hx, hw, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // in registers before the stage may be overwritten
                __builtin_amdgcn_sched_barrier(0);
                FAV_PP_STEP_END();
            }
            if (p.nk > 0) FAV_PP_MMA(hx, hw);
        }
#undef FAV_PP_READ
#undef FAV_PP_MMA
#undef FAV_PP_STEP_END
    } else
    for (int kt = 0; kt < p.nk; ++kt) {
        if ((BK == 32 || !p.stage_mid) && kt + NS - 1 < p.nk) FAV_STAGE(nxt, kt + NS - 1);
        const unsigned char* As = smem + cur * STAGE_BYTES;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            // 64-deep steps: the next tile's DMA is issued after the first MFMA group so its
            // address arithmetic runs under matrix work instead of in front of it
            if (BK == 64 && kk == 1 && p.stage_mid && kt + NS - 1 < p.nk) FAV_STAGE(nxt, kt + NS - 1);
            uint4 fx[TM], fw[TN];
            const int ch = kk * 4 + fq;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int row = wm * WTM + b * 16 + frow;
                fx[b] = *(const uint4*)(As + row * ROWB + ((ch ^ lds_swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int row = wn * WTN + a * 16 + frow;
                fw[a] = *(const uint4*)(Bs + row * ROWB + ((ch ^ lds_swz<BK>(row)) << 4));
            }
            if (MODE == 0) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        union { uint4 u; bf16x8_t v; } ua, ub;
                        ua.u = fw[a];
                        ub.u = fx[b];
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a][b], 0, 0, 0);
                    }
            } else {
                // exact mode: the same bf16 operands through the fp32-input MFMA
                // (k-ordered fmaf chain).  Step j of a 32-wide block multiplies
                // k = 8*g + j for lane groups g = 0..3, in that order.
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int a = 0; a < TN; ++a) {
                        const uint32_t wa = ((const uint32_t*)&fw[a])[j >> 1];
                        const float wf = bf16_bits_to_f32((j & 1) ? (wa >> 16) : (wa & 0xFFFFu));
#pragma unroll
                        for (int b = 0; b < TM; ++b) {
                            const uint32_t xa = ((const uint32_t*)&fx[b])[j >> 1];
                            const float xf = bf16_bits_to_f32((j & 1) ? (xa >> 16) : (xa & 0xFFFFu));
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xf, acc[a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // tile kt+1 has landed (its successors may still be in flight)
        if (kt + NS - 1 < p.nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = (cur + 1 == NS) ? 0 : cur + 1;
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }

    // ---- epilogue ----------------------------------------------------------------
    // 64 rows (= the rows of one wave row wm) go through LDS at a time so the staging
    // area stays under the K-loop stages (more blocks per CU).  A thread
    // finishes 16 consecutive channels of one output pixel per pass (two 16-B stores;
    // one Philox call covers exactly its 16 dropout draws).  Residual rows are requested
    // BEFORE the accumulators go through LDS so their latency hides behind the staging
    // barrier.  (The barrier that ended the K loop already separates the last fragment
    // reads from the first staging writes.)
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 4ull + 2] = wall_clock64();
    if (EPI == 1) {
        // ---- epilogue in registers (see the comment above the kernel) -----------------------------------------
        // LSTORE (the ping-pong tile): bf16 results go through a wave-private image in the K stages, which nobody reads after the loop's
        // last barrier (waves 4..7 finish on registers), and leave as whole lines
        // (not with the GELU epilogue: its vector work hides the direct stores, fc1 639 -> 661 us with the image; profiles/r4y_lstore_ab.txt)
        constexpr bool LSTORE = PP == 1 && !GELU && WTN == 128 && NS * STAGE_BYTES >= NWAVES * 64 * WTN * 2;
        unsigned char* const ytile = smem + wave * (64 * WTN * 2);
        const int pr = wm * 64 + frow;                       // pixel row of b = 0 inside the tile
        float bia[NG][CPL];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                const int c = wn * WTN + g * GS + CPL * fq + 4 * q;          // column inside the tile
                const float4 bq = *(const float4*)((const float*)(smem + LDS_BYTES) + (c >> 4) * 20 + (c & 15));
                bia[g][4 * q] = bq.x; bia[g][4 * q + 1] = bq.y; bia[g][4 * q + 2] = bq.z; bia[g][4 * q + 3] = bq.w;
            }
        if (RES_ALL1) {
#pragma unroll
            for (int b_ = 0; b_ < TM; ++b_) FAV_RES1_LANDED(b_);
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int slot = RES_ALL1 ? b : (b & 1);
            if (!RES_ALL1) {
                FAV_RES1_LANDED(b & 1);
                if (p.res && b + 1 < TM) FAV_LOAD_RES1((b + 1) & 1, b + 1);   // one pixel tile ahead of this one's stores
            }
            const int ml = pr + b * 16;
            const int m = m0 + ml;
            uint32_t vl = 0, pix = 0;
            if (!p.out_f32 && p.drop.site >= 0) {
                vl = fastdiv((uint32_t)m, p.div_hwo);
                pix = (uint32_t)m - vl * (uint32_t)p.HWo;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int n = n0 + wn * WTN + g * GS + CPL * fq;
                uint32_t draws[4] = {~0u, ~0u, ~0u, ~0u};
                if (!p.out_f32 && p.drop.site >= 0) {
                    const uint32_t chunk = (uint32_t)(((long long)pix * p.Cout + n) >> 4);
                    const uint4 w = drop_draws16(p.drop, (uint32_t)p.drop.v0 + vl, chunk);
                    if (CPL == 16 || !(fq & 1)) { draws[0] = w.x; draws[1] = w.y; draws[2] = w.z; draws[3] = w.w; }
                    else { draws[0] = w.z; draws[1] = w.w; }    // 8 channels per lane: the odd quad owns draws 8..15
                }
#pragma unroll
                for (int h8 = 0; h8 < CPL / 8; ++h8) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = __fadd_rn(acc[g * (GS / 16) + 2 * h8 + (k >> 2)][b][k & 3], bia[g][8 * h8 + k]);
                    if (p.res) {
                        const uint32_t rw[4] = {rr1[slot][g][h8][0], rr1[slot][g][h8][1], rr1[slot][g][h8][2], rr1[slot][g][h8][3]};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[2 * k] = __fadd_rn(v[2 * k], bf16_bits_to_f32(rw[k] & 0xFFFFu));
                            v[2 * k + 1] = __fadd_rn(v[2 * k + 1], bf16_bits_to_f32(rw[k] >> 16));
                        }
                    }
                    if (p.relu == 1 && p.out_f32) {        // (bf16 output: ReLU on the packed pairs below)
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                    } else if (GELU && p.relu == 2) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = fav_gelu(v[k]);
                    }
                    if (p.out_f32) {
                        if (m < p.M) {
                            float* yo = (float*)p.y + (long long)m * p.ldy + n + 8 * h8;
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                if (n + 8 * h8 + 4 * q < p.Cout)
                                    *(float4*)(yo + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                        }
                    } else {
                        if (p.drop.site >= 0) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] = FAV_DROP_APPLY(v[k], draws, 8 * h8 + k, p.drop);
                        }
                        u32x4_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                        if (p.relu == 1) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) o[k] = relu_bf16x2(o[k]);
                        }
                        if (LSTORE) {
                            const int r = b * 16 + frow, c16 = (g * GS + CPL * fq + 8 * h8) >> 3;   // row / 16-byte chunk inside the wave tile
                            *(u32x4_t*)(ytile + r * (WTN * 2) + ((c16 ^ (r & 15)) << 4)) = o;
                        } else {
                            // unconditional: rows beyond M fail the descriptor's range check (the store count stays a constant)
                            __builtin_amdgcn_raw_buffer_store_b128(o, srd_y1, (ml * p.ldy + n + 8 * h8) * 2, 0, 0);
                        }
                    }
                }
            }
        }
        if (LSTORE && !p.out_f32) {
            // the wave's 64 x WTN sub-tile leaves in WHOLE 128-byte lines: read back from its private LDS image in line order, one store
            // instruction = 4 rows x 256 contiguous bytes, instead of 16 rows x four 16-byte pieces at a 32-byte pitch (two instructions
            // per line) straight from the accumulator layout - the store path is priced per line touched (the tail kernel's LINEST)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int CH = WTN / 8, RPI = 64 / CH;
#pragma unroll
            for (int i = 0; i < 64 / RPI; ++i) {
                const int r = i * RPI + lane / CH, c = lane % CH;
                const u32x4_t o = *(const u32x4_t*)(ytile + r * (WTN * 2) + ((c ^ (r & 15)) << 4));
                __builtin_amdgcn_raw_buffer_store_b128(o, srd_y1, ((wm * 64 + r) * p.ldy + n0 + wn * WTN + c * 8) * 2, 0, 0);
            }
        }
    } else {
    float* outs = (float*)smem;
    FAV_RES_LANDED();
#pragma unroll
    for (int half = 0; half < NGROUPS; ++half) {
        const int rg = RES_ALL ? half : 0;
        if (!RES_ALL && p.res) FAV_LOAD_RES(0, half)
        if (half > 0) __syncthreads();  // everyone is done reading the previous group
        if (wm == half) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int ml = b * 16 + frow;  // row inside the 64-row group
                    // BN = 256: a row is 16 chunks and the reading lanes of one ds_read_b128 group span chunks
                    // {0-3, 12-15 | 4-11 of the next row}; chunks c and c+8 (>= 8: float4 slot ^ 2) must not share banks
                    const int cw = (wn * WTN + a * 16) >> 4;
                    const int nl = cw * 16 + ((fq ^ (BN == 256 ? ((cw >> 3) & 1) * 2 : 0)) << 2);
                    *(f32x4_t*)(outs + ml * OUT_LD + nl) = acc[a][b];
                }
        }
        __syncthreads();
        if (!RES_ALL) FAV_RES_LANDED();
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int ml = er + pass * RPP;
            const int m = m0 + half * 64 + ml;
            if (er >= 64 || m >= p.M || n >= p.Cout) continue;
            uint32_t draws[4] = {~0u, ~0u, ~0u, ~0u};  // all-ones bytes: always kept
            if (!p.out_f32 && p.drop.site >= 0) {
                const uint32_t vl = fastdiv((uint32_t)m, p.div_hwo);
                const uint32_t pix = (uint32_t)m - vl * (uint32_t)p.HWo;
                const uint32_t chunk = (uint32_t)(((long long)pix * p.Cout + n) >> 4);
                const uint4 w = drop_draws16(p.drop, (uint32_t)p.drop.v0 + vl, chunk);
                draws[0] = w.x; draws[1] = w.y; draws[2] = w.z; draws[3] = w.w;
            }
            // the 16 channels in two groups of 8 (keeps the live register set small)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int gs = BN == 256 ? (g ^ ((ec >> 3) & 1)) : g;   // the staging swizzle (see the writes)
                    const float4 t4 = *(const float4*)(outs + ml * OUT_LD + ec * 16 + 8 * gs + 4 * q);
                    const float4 bq = *(const float4*)(bias_s + 8 * g + 4 * q);
                    v[4 * q] = __fadd_rn(t4.x, bq.x); v[4 * q + 1] = __fadd_rn(t4.y, bq.y);
                    v[4 * q + 2] = __fadd_rn(t4.z, bq.z); v[4 * q + 3] = __fadd_rn(t4.w, bq.w);
                }
                if (p.res) {
                    const uint32_t rw[4] = {rres[rg][pass][g][0], rres[rg][pass][g][1], rres[rg][pass][g][2], rres[rg][pass][g][3]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[2 * j] = __fadd_rn(v[2 * j], bf16_bits_to_f32(rw[j] & 0xFFFFu));
                        v[2 * j + 1] = __fadd_rn(v[2 * j + 1], bf16_bits_to_f32(rw[j] >> 16));
                    }
                }
                if (p.relu == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                } else if (GELU && p.relu == 2) {   // GELU (ViT MLP)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fav_gelu(v[j]);
                }
                if (p.out_f32) {
                    float* yo = (float*)p.y + (long long)m * p.ldy + n + 8 * g;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (n + 8 * g + 4 * q < p.Cout)
                            *(float4*)(yo + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                } else {
                    if (p.drop.site >= 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            v[j] = FAV_DROP_APPLY(v[j], draws, 8 * g + j, p.drop);
                    }
                    *(uint4*)((uint16_t*)p.y + (long long)m * p.ldy + n + 8 * g) =
                        make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                   pack_bf16x2(v[6], v[7]));
                }
            }
        }
    }
    }
    if (FAV_DBG(p)) {
        __syncthreads();
        if (tid == 0) FAV_DBG(p)[blockIdx.x * 4ull + 3] = wall_clock64();
    }
}

#undef FAV_STAGE
#undef FAV_LOAD_RES
#undef FAV_RES_LANDED
#undef FAV_LOAD_RES1
#undef FAV_RES1_LANDED

// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with the input staged ONCE per tile.
//
// The generic kernel above re-gathers the A tile for every tap: 9 x the input through
// L2 -> LDS, and for Cin <= 128 that path (measured 15 TB/s of the ~17-19 TB/s the
// L2 -> LDS gather sustains, profiles/r1d_pmc_3x3.txt) is what bounds the 3x3
// convolutions of layers 1 and 2, not the MFMA.  Here a block of BM consecutive
// output pixels m (flattened frame, y, x) stages the flattened input pixels
// [m0 - W - 1, m0 + BM + W + 1), all Cin channels, once: tap (r, s) of output pixel m
// is LDS row (m - m0) + r*W + s.  Where that tap falls outside the frame (which in the
// flattened order is some neighbouring row or frame) the lane reads a 16-B zero slot
// instead - the zero padding.  Weights are consumed in 64-channel K tiles (tap, channel
// block) in the SAME k order as the generic kernel (k = (r*3 + s)*Cin + c ascending), so
// the sums are bit-identical to it.  Two shapes:
//   * Cin = Cout = 64 : BM = 512, eight waves of 64 x 64; all 9 K tiles of the weights
//     (72 KB) are staged with the patch (79 KB for W = 56) and the K loop has no wait
//     and no barrier at all;
//   * Cin = Cout = 128: BM = 256, 4 x 2 waves of 64 x 64; the weights stream through a
//     double buffer, one tap (two K tiles, 32 KB) per step and barrier.
//
// LDS rows are Cin*2 bytes with chunk ^= (row & 7) | ((row & 1) << 3): conflict-free
// ds_read_b128 for any tap shift.  Epilogue: bias -> ReLU -> one bf16 rounding (these
// convolutions carry no residual and no dropout site; the launcher sends every other
// case to the generic kernel).
// ---------------------------------------------------------------------------
template <int CIN, int BN, int BM, int NS, int SUB, int OCC, int MODE>
__global__ __launch_bounds__(BM * 2, OCC) void conv3x3_halo_kernel(const ConvParams p_launch, int patch_bytes) {
    const ConvParams p = conv_group_params(p_launch);
    constexpr int NT = BM * 2, NWAVES = NT / 64;   // (BM/64) x 2 waves
    constexpr int ROWB = CIN * 2;               // bytes per staged input pixel
    constexpr int CPR = ROWB / 16;              // 16-B chunks per pixel (8 or 16)
    constexpr int PROWS = 1024 / ROWB;          // pixels per 1-KiB LDS-DMA piece
    constexpr int CB = CIN / 64;                // 64-channel K tiles per tap
    constexpr int NKT = 9 * CB;                 // K tiles
    constexpr int NSTEPS = NKT / SUB;           // a step = SUB K tiles = one wait + barrier
    constexpr bool RESIDENT = NS >= NSTEPS;     // every K tile staged up front
    constexpr int WAVES_M = BM / 64, WAVES_N = 2;
    constexpr int WTM = 64, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int B_BYTES = BN * 128;           // one K tile of weights: BN rows x 64 k
    constexpr int BR = B_BYTES / 1024 / NWAVES; // LDS-DMA pieces per wave per K tile
    constexpr int OUT_LD = BN + 4;
    static_assert(NKT % SUB == 0 && BR >= 1 && WAVES_M * WAVES_N == NWAVES, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    // [patch | NS*SUB weight K tiles | bias (16-channel chunks, 20 floats apart) | 16 zero bytes]; the fp32 staging of the epilogue overlays the patch
    unsigned char* const Bring = hsm + patch_bytes;
    float* const bias_s = (float*)(Bring + NS * SUB * B_BYTES);
    // 256 zero bytes on a 256-byte boundary: a lane whose tap leaves the frame reads the zero region at the SAME 16-byte
    // slot of the bank row its patch address has (zero_off | (address & 0xF0)), so the 16 lanes a ds_read_b128 is served in
    // keep their 16 distinct slots - one shared 16-byte zero slot collided with a valid lane's bank at every frame border
    // (8-12 % of the LDS cycles of these kernels were bank-conflict cycles, profiles/r3g_lds_zero_region_ab.txt)
    const uint32_t zero_off = ((uint32_t)(patch_bytes + NS * SUB * B_BYTES + BN * 5) + 255u) & ~255u;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nwg = gridDim.x;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = tile * BM;
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 4ull] = wall_clock64();
    if (tid < BN) bias_s[(tid >> 4) * 20 + (tid & 15)] = p.bias[tid];
    if (tid < 64) ((uint32_t*)(hsm + zero_off))[tid] = 0u;

    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)hsm);
    // ---- stage the patch ------------------------------------------------------
    const int W = p.W;
    const long long g0 = (long long)m0 - W - 1;              // global pixel of patch row 0
    const long long gbase = g0 > 0 ? g0 : 0;
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + gbase * CIN), 0, 0x7FFFFFFF, 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;
    {
        const int npieces = patch_bytes >> 10;
        const int lrow = lane / CPR, lslot = lane % CPR;
        for (int j = wave_u; j < npieces; j += NWAVES) {
            const int q = j * PROWS + lrow;                  // patch row of this lane
            const int sw = (CPR == 8) ? (q & 7) : ((q & 7) | ((q & 1) << 3));
            const long long g = g0 + q;
            const bool ok = g >= 0 && g < (long long)p.M;
            const uint32_t voff = ok ? (uint32_t)((g - gbase) * ROWB + ((lslot ^ sw) << 4)) : OOB;
            lds_dma16(srd_a, voff, 0u, __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)j * 1024u));
        }
    }
    // ---- weights -------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t srd_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, BN * p.K * 2, 0x00020000);
    uint32_t b_voff[BR];
    {
        const int lrow = lane >> 3, lch = (lane & 7) ^ (lrow & 7);
#pragma unroll
        for (int i = 0; i < BR; ++i) b_voff[i] = (uint32_t)((((wave * BR + i) * 8 + lrow) * p.K + lch * 8) * 2);
    }
    const uint32_t lds_b = lds_base + (uint32_t)patch_bytes + wave_u * (BR * 1024);
// K tiles [STEP*SUB, STEP*SUB + SUB) into ring slot BUF
#define FAV_HSTAGE(BUF, STEP)                                                                          \
    do {                                                                                               \
        _Pragma("unroll") for (int u = 0; u < SUB; ++u)                                                \
            _Pragma("unroll") for (int i = 0; i < BR; ++i)                                             \
                lds_dma16(srd_b, b_voff[i], (uint32_t)(((STEP) * SUB + u) * 128),                      \
                          __builtin_amdgcn_readfirstlane(lds_b + ((BUF) * SUB + u) * B_BYTES + i * 1024)); \
    } while (0)
    constexpr int PIECES = BR * SUB;
    if (RESIDENT) {
#pragma unroll
        for (int t = 0; t < NSTEPS; ++t) FAV_HSTAGE(t, t);
    } else {
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) FAV_HSTAGE(t, t);
    }

    // ---- per-lane row geometry ---------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4;
    uint32_t tapmask[TM];                                    // bit (r*3+s): the tap lies inside the frame
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * WTM + b * 16 + frow;
        uint32_t mk = 0;
        if (m < p.M) {
            const uint32_t vl = fastdiv((uint32_t)m, p.div_hwo);
            const uint32_t pix = (uint32_t)m - vl * (uint32_t)p.HWo;
            const uint32_t y = fastdiv(pix, p.div_w), x = pix - y * (uint32_t)W;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if ((unsigned)((int)y + r - 1) < (unsigned)p.H && (unsigned)((int)x + s - 1) < (unsigned)W) mk |= 1u << (r * 3 + s);
        }
        tapmask[b] = mk;
    }

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // the patch and the first weight step (RESIDENT: everything) have landed
    if (RESIDENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
    __syncthreads();
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 4ull + 1] = wall_clock64();

    int cur = 0, nxt = NS - 1;
    int tap = 0, cb = 0, tapoff = 0, tr = 0, ts = 0;        // tapoff = r*W + s
    for (int step = 0; step < NSTEPS; ++step) {
        if (!RESIDENT && step + NS - 1 < NSTEPS) FAV_HSTAGE(nxt, step + NS - 1);
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const unsigned char* Bs = Bring + (cur * SUB + u) * B_BYTES;
            uint32_t a_addr[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int q = wm * WTM + b * 16 + frow + tapoff;
                const int sw = (CPR == 8) ? (q & 7) : ((q & 7) | ((q & 1) << 3));
                const uint32_t ad = (uint32_t)q * ROWB + (uint32_t)(((cb * 8 + fq) ^ sw) << 4);
                a_addr[b] = ((tapmask[b] >> tap) & 1u) ? ad : (zero_off | (ad & 0xF0u));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fx[TM], fw[TN];
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    // chunk + 4 flips bit 2 of the (swizzled) chunk index: address ^ 64 (inside the zero region too)
                    fx[b] = *(const uint4*)(hsm + (a_addr[b] ^ (uint32_t)(kk << 6)));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int row = wn * WTN + a * 16 + frow;
                    fw[a] = *(const uint4*)(Bs + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
                }
                if (MODE == 0) {
#pragma unroll
                    for (int a = 0; a < TN; ++a)
#pragma unroll
                        for (int b = 0; b < TM; ++b) {
                            union { uint4 u; bf16x8_t v; } ua, ub;
                            ua.u = fw[a];
                            ub.u = fx[b];
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a][b], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                        for (int a = 0; a < TN; ++a) {
                            const uint32_t wa = ((const uint32_t*)&fw[a])[j >> 1];
                            const float wf = bf16_bits_to_f32((j & 1) ? (wa >> 16) : (wa & 0xFFFFu));
#pragma unroll
                            for (int b = 0; b < TM; ++b) {
                                const uint32_t xa = ((const uint32_t*)&fx[b])[j >> 1];
                                const float xf = bf16_bits_to_f32((j & 1) ? (xa >> 16) : (xa & 0xFFFFu));
                                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xf, acc[a][b], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (++cb == CB) {
                cb = 0;
                ++tap;
                if (++ts == 3) { ts = 0; ++tr; }
                tapoff = tr * W + ts;
            }
        }
        if (RESIDENT) {
            ++cur;
        } else {
            if (step + NS - 1 < NSTEPS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    }
#undef FAV_HSTAGE
    if (RESIDENT) __syncthreads();   // every wave is done with the patch before the staging overwrites it
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 4ull + 2] = wall_clock64();

    // ---- epilogue: GROWS rows at a time through LDS (over the patch), 16 channels per thread ----
    constexpr int NCH = BN / 16;
    constexpr int GROWS = (NT / NCH >= 128 && BM >= 128) ? 128 : 64;   // rows staged per round: every thread busy when it can be
    constexpr int RPP = (NT / NCH > GROWS) ? GROWS : NT / NCH;
    constexpr int NPASS = GROWS / RPP;
    constexpr int WPG = GROWS / 64;                                     // wave rows per round
    const int ec = tid % NCH, er = tid / NCH;
    const int n = ec * 16;
    float* outs = (float*)hsm;
#pragma unroll 1
    for (int grp = 0; grp < BM / GROWS; ++grp) {
        if (grp > 0) __syncthreads();
        if (wm / WPG == grp) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    *(f32x4_t*)(outs + ((wm % WPG) * 64 + b * 16 + frow) * OUT_LD + wn * WTN + a * 16 + fq * 4) = acc[a][b];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int ml = er + pass * RPP;
            const int m = m0 + grp * GROWS + ml;
            if (er >= GROWS || m >= p.M) continue;
            uint32_t o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *(const float4*)(outs + ml * OUT_LD + n + 4 * q);
                const float4 bq = *(const float4*)(bias_s + ec * 20 + 4 * q);
                float v0 = __fadd_rn(t4.x, bq.x), v1 = __fadd_rn(t4.y, bq.y), v2 = __fadd_rn(t4.z, bq.z), v3 = __fadd_rn(t4.w, bq.w);
                if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                o[2 * q] = pack_bf16x2(v0, v1);
                o[2 * q + 1] = pack_bf16x2(v2, v3);
            }
            uint4* yo = (uint4*)((uint16_t*)p.y + (long long)m * p.ldy + n);
            yo[0] = make_uint4(o[0], o[1], o[2], o[3]);
            yo[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
    if (FAV_DBG(p)) {
        __syncthreads();
        if (tid == 0) FAV_DBG(p)[blockIdx.x * 4ull + 3] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------
// Bottleneck tail: three convolutions of the residual chain in ONE launch, with the two intermediate
// tensors kept on chip:
//   conv_b : 3x3 / 1 / 1, CMID -> CMID, bias, ReLU                    (block b, its conv2)      [HAS3X3]
//   conv_c : 1x1, CMID -> COUT = 4*CMID, bias + residual, ReLU, dropout site, Y written to HBM  (block b, conv3)
//   conv_a : 1x1, COUT -> NRED, bias, ReLU, written to HBM            (block b+1, its conv1)    [NRED > 0]
// Layer by layer the block boundary moves 6.4 MB per frame through HBM in layer 1 (read t1, write t2, read t2,
// read X, write Y, read Y, write t1'); fused it moves 4.0 MB (read t1 and X, write Y and t1'): conv_b's output
// never leaves the CU, and Y is consumed by conv_a while its 64-channel chunk is still in LDS.
//
// A block of NW waves owns BM = 32*NW consecutive output pixels m (flattened frame, y, x):
//   P1  conv_b exactly as conv3x3_halo_kernel (patch staged once, weights through a ring, same k order), waves
//       BM/64 (pixels) x 2 (channels);
//   P1e T2 = bf16(relu(acc + bias_b)) -> LDS in the MFMA operand layout; from here on wave w OWNS the 32 pixel
//       rows [32w, 32w + 32) with all their channels, and keeps their T2 fragments (all CMID k) in registers;
//   P2  for each 64-channel chunk j of Y:  acc2 = T2 x Wc[j]^T  (K = CMID).  The rows of Wc[j] are staged in the
//       order 16*(i/4) + 4*a + i%4 (a = MFMA tile, i = its row), so lane (pixel frow, quad fq) ends up with the 16
//       CONSECUTIVE channels 16*fq .. 16*fq+15 of its pixel in registers: bias + residual + ReLU + Philox dropout
//       (one call = exactly its 16 draws) + one bf16 rounding happen in registers, two 16-B stores send the chunk
//       to HBM and two ds_write_b128 put it into the wave's own rows of the Y-chunk LDS image; then
//       acc3 += Ychunk x Wa[:, 64j..64j+63]^T on the wave's own rows - no workgroup barrier, no fp32 staging.
//       (k ascending, 64 per chunk: the order conv_igemm_kernel uses, so all sums are bit-identical to it.)
//   P3  t1' = bf16(relu(acc3 + bias_a)) from registers (same row order trick for Wa) -> HBM.
// The weights of chunk j+1 (Wc, Wa: double-buffered) and its residual are requested at the start of chunk j; the
// only workgroup barrier of a chunk is the one that publishes those weight pieces.  Every load is an LDS-DMA or
// a raw buffer load whose descriptor ends at the tile's last valid row (rows beyond M read zeros / are not
// written), and every store is an unconditional raw buffer store, so the NUMBER of vector memory operations per
// chunk is a compile-time constant: the one wait per chunk is a counted `s_waitcnt vmcnt(N)` that leaves the
// residual loads of chunk j+1 and the stores of chunk j in flight, and no store is ever waited for.
// ---------------------------------------------------------------------------
struct TailParams {
    const uint16_t* t1;       // [M][CMID]: input of conv_b (HAS3X3) or of conv_c (!HAS3X3)
    const uint16_t* wb; const float* bias_b;     // [CMID][9*CMID]
    const uint16_t* wc; const float* bias_c;     // [COUT][CMID]
    const uint16_t* res;      // [M][COUT]
    uint16_t* y;              // [M][COUT]
    const uint16_t* wa; const float* bias_a;     // [NRED][COUT]
    uint16_t* t1n;            // [M][NRED]
    int H, W, HW, M;           // OUTPUT frame geometry (= input geometry unless in_stride > 1)
    int in_W, in_HW, in_stride; // conv_c alone as a strided 1x1 (a downsample convolution): input row pitch, frame size, stride
    int rega_bytes;           // LDS region A: patch | T2 tile | Y chunk
    // The LDS plan is the launcher's (fav.hip: tail_geometry); byte offsets from the block's LDS base: bias_b, then bias_c and bias_a
    // (16-channel chunks 20 floats apart), the two Wa chunk buffers (chunk j reads buffer (j + 1) & 1), 256 zero bytes on a 256-byte boundary
    int bias_b_off, bias_ca_off, wa_off0, wa_off1, zero_off;
    DropParams drop;
    FastDiv div_hw, div_w;
    unsigned long long* dbg;  // FAV_CONV_DBG: per-block phase timestamps (null in normal runs)
    // grouped launch (see ConvParams): byte strides per member of the activations (t1, res, y, t1n) and of each weight / bias set
    long long g_t1, g_res, g_y, g_t1n, g_wb, g_bb, g_wc, g_bc, g_wa, g_ba;
    // RESE (the first block behind the entry dropout of an MC-Dropout suffix): `res` is the CACHED prefix output
    // [drop.n_img][HW][COUT]; the residual of virtual frame v is dropout_{site_e}(res[v % n_img]), computed here, so the
    // T dropped copies are never written to HBM nor read back
    int site_e;
    int rs_T, rs_tps;         // RESE tile order: samples in the launch, tiles per sample (rs_T <= 1: row order)
};

__device__ __forceinline__ TailParams tail_group_params(const TailParams& q) {
    TailParams p = q;
    const long long g = blockIdx.y;
    p.t1 = (const uint16_t*)((const char*)q.t1 + g * q.g_t1);
    if (q.wb) { p.wb = (const uint16_t*)((const char*)q.wb + g * q.g_wb); p.bias_b = (const float*)((const char*)q.bias_b + g * q.g_bb); }
    p.wc = (const uint16_t*)((const char*)q.wc + g * q.g_wc);
    p.bias_c = (const float*)((const char*)q.bias_c + g * q.g_bc);
    if (q.res) p.res = (const uint16_t*)((const char*)q.res + g * q.g_res);
    p.y = (uint16_t*)((char*)q.y + g * q.g_y);
    if (q.wa) { p.wa = (const uint16_t*)((const char*)q.wa + g * q.g_wa); p.bias_a = (const float*)((const char*)q.bias_a + g * q.g_ba); }
    if (q.t1n) p.t1n = (uint16_t*)((char*)q.t1n + g * q.g_t1n);
    return p;
}

// raw workgroup barrier: this wave's LDS operations retired, no vmcnt drain (stores and prefetches stay in flight)
#define FAV_BAR()                                               \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
    } while (0)

// LDS row r (tile a = r/16 of a 64-row group, MFMA row i = r%16) holds weight row 16*(i/4) + 4*a + i%4 of the group
__device__ __forceinline__ int tail_row_perm(int r) {
    const int g = r & ~63, a = (r >> 4) & 3, i = r & 15;
    return g + 16 * (i >> 2) + 4 * a + (i & 3);
}

// XOR applied to the 16-B chunk index of an activation row of CPR chunks (128, 256 or 512 B), chosen so that the 16
// lanes one ds_read_b128 group serves - 8 rows with quad f and the other 8 with quad f + 1 - hit 16 different slots
template <int CPR>
__device__ __forceinline__ int tail_sw(int row) {
    return CPR == 8 ? (row & 7) : (CPR == 16 ? ((row & 7) | ((row & 1) << 3)) : (row & 15));
}

// COUT_ = 0: the bottleneck's 4*CMID.  HAS_RES / RELU = false and in_stride = 2 turn conv_c alone into the projection
// shortcut of a stage's first block (1x1 / 2, no residual, no ReLU, no dropout).
template <int CMID, int NRED, bool HAS3X3, int NS, int NW, bool WC2, int RP = 32, int COUT_ = 0, bool HAS_RES = true, bool RELU = true, bool RESE = false>
__global__ __launch_bounds__(NW * 64, 2) void bottleneck_tail_kernel(const TailParams p_launch, int patch_bytes) {
    const TailParams p = tail_group_params(p_launch);
    // Y is stored in whole 128-byte lines from an LDS image of the chunk: the one conv_a reads (NRED > 0), or - behind a staged-patch
    // conv_b, whose region A is free in P2 - an image kept for that purpose alone
    constexpr bool LINEST = NRED > 0 || (HAS3X3 && !(HAS3X3 && CMID == 256));
    constexpr int COUT = COUT_ > 0 ? COUT_ : 4 * CMID;
    static_assert(HAS_RES || (!HAS3X3 && NRED == 0), "only conv_c alone runs without a residual");
    constexpr int BM = RP * NW, NT = NW * 64;      // every wave owns RP pixel rows in P2 (16 only without conv_b)
    static_assert(RP == 32 || (RP == 16 && !HAS3X3), "rows per wave");
    constexpr int ROWB = CMID * 2;               // bytes per pixel of t1 / T2
    constexpr int CPR = ROWB / 16;               // 16-B chunks per pixel (8 or 16)
    constexpr int PROWS = 1024 / ROWB;           // pixels per 1-KiB LDS-DMA piece
    constexpr int CB = CMID / 64;                // 64-channel K tiles per tap
    constexpr int NKT = 9 * CB;
    constexpr int TM = 4;                        // conv_b: BM/64 x 2 waves, wave tile 64 px x CMID/2 channels
    constexpr int WN1 = 2;
    constexpr int TN1 = CMID / WN1 / 16;
    constexpr int SLOT = CMID * 128;             // one K tile of Wb: CMID rows x 64 k
    constexpr int BR = SLOT / 1024 / NW;         // LDS-DMA pieces per wave per K tile of Wb
    constexpr int NCHUNK = COUT / 64;
    constexpr int KS2 = CMID / 32;               // 32-deep k steps of conv_c
    constexpr int TM2 = RP / 16;
    constexpr int G3 = NRED / 64;                // 64-channel groups of conv_a's output
    constexpr int WC_BYTES = 64 * ROWB;          // Wc chunk: 64 rows x CMID k, as CB sub tiles of 64 x 64
    constexpr int WA_BYTES = NRED * 128;         // Wa chunk: NRED rows x 64 k
    // Wc buffers: 2 = requested a chunk ahead; 1 = requested behind a barrier after step A; 4 (generic conv_b, whose 128 KB
    // of stages are free in P2) = requested two chunks at a time, ONE workgroup barrier per two chunks (BARQ)
    constexpr int WCN = (HAS3X3 && CMID == 256) ? 4 : (WC2 ? 2 : 1);
    constexpr int BARQ = WCN == 4 ? 2 : 1;
    constexpr int P2W = WCN * WC_BYTES + 2 * WA_BYTES;
    constexpr int RING = HAS3X3 ? NS * SLOT : 0;
    constexpr int REGB = RING > P2W ? RING : P2W;
    constexpr int WC_PIECES = WC_BYTES / 1024, WA_PIECES = WA_BYTES / 1024;
    constexpr int WC_PW = (WC_PIECES + NW - 1) / NW, WA_PW = (WA_PIECES + NW - 1) / NW;   // pieces per wave (the last may be skipped)
    static_assert(BR >= 1 && TN1 >= 1 && TM2 >= 1 && WC_PIECES % NW == 0 && (NRED == 0 || WA_PIECES % NW == 0), "tile shape");
    static_assert(!HAS3X3 || (NKT % NS == 0 && WC_BYTES <= SLOT), "Wc buffer 0 = ring slot 0, free during the last NS - 1 steps");
    const bool WA0_EARLY = !HAS3X3 || p.wa_off1 >= p.rega_bytes + RING;        // chunk 0's Wa buffer lies beyond the P1 ring (uniform)
    // 256 mid channels: the patch of a 256-pixel tile would not fit, so conv_b runs as the generic 256 x 256 x 64 loop
    // (BOTH operands DMA'd per K tile, two stages = 128 KB); T2 then takes that same 128 KB, and once its fragments
    // sit in registers the Wc buffers of P2 reuse it from offset 0 (no region B, no Y chunk: NRED = 0)
    constexpr bool P1G = HAS3X3 && CMID == 256;
    static_assert(!P1G || (NW == 8 && NS == 2 && RP == 32 && NRED == 0 && WC2), "generic conv_b: 8 waves, two stages, conv_c alone behind it");
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    // [region A | region B: weight ring (P1) / Wc x2, Wa x2 (P2) | biases (16-channel chunks 20 floats apart) | 16 zero bytes]
    unsigned char* const ych = tsm;                                // P2: Y chunk [BM][64] bf16, chunk ^= row & 7
    unsigned char* const regb = P1G ? tsm : tsm + p.rega_bytes;
    float* const bias_b_s = (float*)(tsm + p.bias_b_off);
    float* const bias_c_s = (float*)(tsm + p.bias_ca_off);
    float* const bias_a_s = bias_c_s + (COUT / 16) * 20;
    const uint32_t zero_off = (uint32_t)p.zero_off;                // 256 zero bytes, see conv3x3_halo_kernel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN1, wn = wave % WN1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int frow = lane & 15, fq = lane >> 4;
    const int nwg = gridDim.x;
    int tile;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    if constexpr (RESE) {
        // sample-minor order: an XCD's consecutive tiles are the rs_T samples of ONE pixel tile of the cached tensor, so the
        // residual of that tile comes from HBM once and from the XCD's L2 for the other samples
        if (p.rs_T > 1) {
            const int pt = tile / p.rs_T, t = tile - pt * p.rs_T;
            tile = t * p.rs_tps + pt;
        }
    }
    const int m0 = tile * BM;
    const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull] = wall_clock64();

    // ---- P0: biases, descriptors, the first requests ----------------------------------------------------------------
    // The bias values are REQUESTED here, all at once, and written to LDS only after the first LDS-DMA requests are out:
    // written where they are loaded, each `global_load` waited for its own data before the next one - and before the
    // patch - was even requested (three L2 round trips, ~2 us, in front of every tile).
    constexpr int NBB = HAS3X3 ? (CMID + NT - 1) / NT : 0, NBC = (COUT + NT - 1) / NT, NBA = NRED > 0 ? (NRED + NT - 1) / NT : 0;
    float bias_v[NBB + NBC + NBA];
#pragma unroll
    for (int k = 0; k < NBB; ++k) bias_v[k] = (tid + k * NT < CMID) ? p.bias_b[tid + k * NT] : 0.f;
#pragma unroll
    for (int k = 0; k < NBC; ++k) bias_v[NBB + k] = (tid + k * NT < COUT) ? p.bias_c[tid + k * NT] : 0.f;
#pragma unroll
    for (int k = 0; k < NBA; ++k) bias_v[NBB + NBC + k] = (tid + k * NT < NRED) ? p.bias_a[tid + k * NT] : 0.f;
#define FAV_T_BIAS_TO_LDS(DO_B, DO_CA)                                                                            \
    do {                                                                                                          \
        if (DO_B) _Pragma("unroll") for (int k = 0; k < NBB; ++k) {                                               \
            const int i = tid + k * NT;                                                                           \
            if (i < CMID) bias_b_s[(i >> 4) * 20 + (i & 15)] = bias_v[k];                                         \
        }                                                                                                         \
        if (DO_CA) {                                                                                              \
            _Pragma("unroll") for (int k = 0; k < NBC; ++k) {                                                     \
                const int i = tid + k * NT;                                                                       \
                if (i < COUT) bias_c_s[(i >> 4) * 20 + (i & 15)] = bias_v[NBB + k];                               \
            }                                                                                                     \
            _Pragma("unroll") for (int k = 0; k < NBA; ++k) {                                                     \
                const int i = tid + k * NT;                                                                       \
                if (i < NRED) bias_a_s[(i >> 4) * 20 + (i & 15)] = bias_v[NBB + NBC + k];                         \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
    if (tid < 64) ((uint32_t*)(tsm + zero_off))[tid] = 0u;
    // the residual of chunk 0 comes from HBM: requested now, it lands under conv_b instead of in front of P2's first epilogue
    const __amdgpu_buffer_rsrc_t srd_res = RESE
        ? __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)((long long)p.drop.n_img * p.HW * COUT * 2), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)((HAS_RES ? p.res : p.y) + (long long)m0 * COUT), 0, rows_valid * COUT * 2, 0x00020000);
    uint32_t eoff[TM2];          // RESE: byte offset of this lane's pixel rows in the cached tensor (frame v % n_img)
    if constexpr (RESE) {
#pragma unroll
        for (int b = 0; b < TM2; ++b) {
            const uint32_t m = (uint32_t)(m0 + wave * RP + b * 16 + frow);
            const uint32_t vl = fastdiv(m, p.div_hw);
            const uint32_t v = (uint32_t)p.drop.v0 + vl;
            const uint32_t il = v - fastdiv(v, p.drop.div_img) * (uint32_t)p.drop.n_img;
            eoff[b] = m < (uint32_t)p.M ? (il * (uint32_t)p.HW + (m - vl * (uint32_t)p.HW)) * (uint32_t)(COUT * 2) : 0x80000000u;
        }
    }
    u32x4_t rnext[TM2][2];
#define FAV_T_LOAD_RES(J)                                                                                        \
    if (HAS_RES) _Pragma("unroll") for (int b = 0; b < TM2; ++b) {                                               \
        const int off = RESE ? (int)(eoff[b] + (uint32_t)(((J) * 64 + fq * 16) * 2))                             \
                             : ((wave * RP + b * 16 + frow) * COUT + (J) * 64 + fq * 16) * 2;                    \
        rnext[b][0] = __builtin_amdgcn_raw_buffer_load_b128(srd_res, off, 0, 0);                                 \
        rnext[b][1] = __builtin_amdgcn_raw_buffer_load_b128(srd_res, off + 16, 0, 0);                            \
    }
    FAV_T_LOAD_RES(0)
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)tsm);
    const uint32_t lds_regb = lds_base + (P1G ? 0u : (uint32_t)p.rega_bytes);
    constexpr uint32_t OOB = 0x80000000u;
    // weights: rows of 128 B (64 k), a piece = 8 rows; lane -> row l>>3, physical slot l&7 holds chunk (l&7)^(row&7)
    const int wrow = lane >> 3, wch = (lane & 7) ^ (wrow & 7);
    const __amdgpu_buffer_rsrc_t srd_wc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wc, 0, COUT * CMID * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_wa = __builtin_amdgcn_make_buffer_rsrc((void*)p.wa, 0, NRED > 0 ? NRED * COUT * 2 : 16, 0x00020000);
// Wc chunk J (64 rows x CMID k) -> buffer J % WCN: CB sub tiles [64][64], LDS row r <- weight row tail_row_perm(r)
#define FAV_T_STAGE_WC(J)                                                                                        \
    do {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WC_PW; ++i) {                                                      \
            const int pc = wave_u + i * NW;                /* piece: sub tile pc / 8, rows (pc % 8) * 8 .. + 7 */ \
            const int sub = pc >> 3, r0 = (pc & 7) * 8;                                                          \
            lds_dma16(srd_wc, (uint32_t)((((J) * 64 + tail_row_perm(r0 + wrow)) * CMID + sub * 64 + wch * 8) * 2), 0u, \
                      __builtin_amdgcn_readfirstlane(lds_regb + (uint32_t)(((J) % WCN) * WC_BYTES + pc * 1024)));  \
        }                                                                                                        \
    } while (0)
// Wa chunk J (NRED rows x 64 k) -> buffer (J + 1) & 1: chunk 0's lies beyond the P1 weight ring, so it is requested in P0
#define FAV_T_STAGE_WA(J)                                                                                        \
    do {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < WA_PW; ++i) {                                                      \
            const int pc = wave_u + i * NW;                                                                      \
            lds_dma16(srd_wa, (uint32_t)((tail_row_perm(pc * 8 + wrow) * COUT + wch * 8) * 2), (uint32_t)((J) * 128), \
                      __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(((((J) + 1) & 1) ? p.wa_off1 : p.wa_off0) + pc * 1024))); \
        }                                                                                                        \
    } while (0)

    const int W = p.W;
    if (NRED > 0 && WA0_EARLY) FAV_T_STAGE_WA(0);   // oldest request of the block (its buffer is not part of the P1 ring)
    if (HAS3X3) {
        f32x4_t acc[TN1][TM];
#pragma unroll
        for (int a = 0; a < TN1; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (P1G) {
            // ---- P1 (generic): K tile kt = tap kt / CB, channels 64 * (kt % CB); A tile [BM][64 k] gathered per tap
            //      (pixel m + (r-1) W + (s-1) of the flattened tensor, or zeros where the tap leaves the frame), B tile
            //      [CMID][64 k]; the same k order and the same loop as conv_igemm_kernel<256, 256, 64, 2> ----------------
            constexpr int STG = 2 * 32768, AR = 32768 / 1024 / NW;
            static_assert(BM == 256 && AR == 4 && BR == 4, "generic conv_b tile");
            const __amdgpu_buffer_rsrc_t srd_a =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1 + ((long long)m0 - W - 1) * CMID), 0, 0x7FFFFFFF, 0x00020000);
            const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wb, 0, CMID * 9 * CMID * 2, 0x00020000);
            uint32_t a_voff[AR], a_mask[AR], b_voff[BR];
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int row = (wave * AR + i) * 8 + wrow, m = m0 + row;
                uint32_t mk = 0;
                if (m < p.M) {
                    const uint32_t vl = fastdiv((uint32_t)m, p.div_hw);
                    const uint32_t pix = (uint32_t)m - vl * (uint32_t)p.HW;
                    const uint32_t y = fastdiv(pix, p.div_w), x = pix - y * (uint32_t)W;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int s2 = 0; s2 < 3; ++s2)
                            if ((unsigned)((int)y + r - 1) < (unsigned)p.H && (unsigned)((int)x + s2 - 1) < (unsigned)W) mk |= 1u << (r * 3 + s2);
                }
                a_mask[i] = mk;
                a_voff[i] = (uint32_t)(row * ROWB + wch * 16);
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) b_voff[i] = (uint32_t)((((wave * BR + i) * 8 + wrow) * (9 * CMID) + wch * 8) * 2);
            const uint32_t lds_a = lds_base + wave_u * (AR * 1024), lds_b = lds_base + 32768 + wave_u * (BR * 1024);
            int s_tap = 0, s_cb = 0, s_r = 0, s_s = 0;     // position of the NEXT K tile to stage
#define FAV_T_GSTAGE(BUF, KT)                                                                                    \
    do {                                                                                                         \
        const uint32_t soff_a = (uint32_t)((s_r * W + s_s) * ROWB + s_cb * 128);                                 \
        _Pragma("unroll") for (int i = 0; i < AR; ++i)                                                           \
            lds_dma16(srd_a, ((a_mask[i] >> s_tap) & 1u) ? a_voff[i] : OOB, soff_a,                              \
                      __builtin_amdgcn_readfirstlane(lds_a + (BUF) * STG + i * 1024));                           \
        _Pragma("unroll") for (int i = 0; i < BR; ++i)                                                           \
            lds_dma16(srd_b, b_voff[i], (uint32_t)((KT) * 128),                                                  \
                      __builtin_amdgcn_readfirstlane(lds_b + (BUF) * STG + i * 1024));                           \
        if (++s_cb == CB) {                                                                                      \
            s_cb = 0; ++s_tap;                                                                                   \
            if (++s_s == 3) { s_s = 0; ++s_r; }                                                                  \
        }                                                                                                        \
    } while (0)
            FAV_T_GSTAGE(0, 0);
            FAV_T_BIAS_TO_LDS(true, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 1] = wall_clock64();
            int cur = 0;
            for (int kt = 0; kt < NKT; ++kt) {
                const unsigned char* As = tsm + cur * STG;
                const unsigned char* Bs = As + 32768;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    // the next tile's DMA goes out after the first MFMA group: its address arithmetic runs under matrix work
                    if (kk == 1 && kt + 1 < NKT) FAV_T_GSTAGE(cur ^ 1, kt + 1);
                    uint4 fx[TM], fw[TN1];
                    const int ch = kk * 4 + fq;
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        const int row = wm * 64 + b * 16 + frow;
                        fx[b] = *(const uint4*)(As + row * 128 + ((ch ^ (row & 7)) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < TN1; ++a) {
                        const int row = wn * (CMID / WN1) + a * 16 + frow;
                        fw[a] = *(const uint4*)(Bs + row * 128 + ((ch ^ (row & 7)) << 4));
                    }
#pragma unroll
                    for (int a = 0; a < TN1; ++a)
#pragma unroll
                        for (int b = 0; b < TM; ++b) {
                            union { uint4 u; bf16x8_t v; } ua, ub;
                            ua.u = fw[a];
                            ub.u = fx[b];
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a][b], 0, 0, 0);
                        }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
#undef FAV_T_GSTAGE
        } else {
        // ---- patch: flattened input pixels [m0 - W - 1, m0 + BM + W + 1), all CMID channels --------------------
        const long long g0 = (long long)m0 - W - 1;
        const long long gbase = g0 > 0 ? g0 : 0;
        const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1 + gbase * CMID), 0, 0x7FFFFFFF, 0x00020000);
        {
            const int npieces = patch_bytes >> 10;
            const int lrow = lane / CPR, lslot = lane % CPR;
            for (int j = wave_u; j < npieces; j += NW) {
                const int q = j * PROWS + lrow;
                const int sw = tail_sw<CPR>(q);
                const long long g = g0 + q;
                const bool ok = g >= 0 && g < (long long)p.M;
                const uint32_t voff = ok ? (uint32_t)((g - gbase) * ROWB + ((lslot ^ sw) << 4)) : OOB;
                lds_dma16(srd_a, voff, 0u, __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)j * 1024u));
            }
        }
        const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wb, 0, CMID * 9 * CMID * 2, 0x00020000);
        uint32_t b_voff[BR];
#pragma unroll
        for (int i = 0; i < BR; ++i) b_voff[i] = (uint32_t)((((wave * BR + i) * 8 + wrow) * (9 * CMID) + wch * 8) * 2);
        const uint32_t lds_b = lds_regb + wave_u * (BR * 1024);
#define FAV_T_HSTAGE(BUF, KT)                                                                         \
    do {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < BR; ++i)                                                \
            lds_dma16(srd_b, b_voff[i], (uint32_t)((KT) * 128),                                       \
                      __builtin_amdgcn_readfirstlane(lds_b + (BUF) * SLOT + i * 1024));              \
    } while (0)
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) FAV_T_HSTAGE(t, t);
        uint32_t tapmask[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = m0 + wm * 64 + b * 16 + frow;
            uint32_t mk = 0;
            if (m < p.M) {
                const uint32_t vl = fastdiv((uint32_t)m, p.div_hw);
                const uint32_t pix = (uint32_t)m - vl * (uint32_t)p.HW;
                const uint32_t y = fastdiv(pix, p.div_w), x = pix - y * (uint32_t)W;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2)
                        if ((unsigned)((int)y + r - 1) < (unsigned)p.H && (unsigned)((int)x + s2 - 1) < (unsigned)W) mk |= 1u << (r * 3 + s2);
            }
            tapmask[b] = mk;
        }
        FAV_T_BIAS_TO_LDS(true, true);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * BR) : "memory");
        __syncthreads();
        if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 1] = wall_clock64();
        int cur = 0, nxt = NS - 1;
        int tap = 0, cb = 0, tapoff = 0, tr = 0, ts = 0;
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt + NS - 1 < NKT) FAV_T_HSTAGE(nxt, kt + NS - 1);
            else if (kt + NS - 1 == NKT) FAV_T_STAGE_WC(0);   // ring slot NKT % NS = 0 is free: conv_c's first weights ride under the last steps
            const unsigned char* Bs = regb + cur * SLOT;
            uint32_t a_addr[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int q = wm * 64 + b * 16 + frow + tapoff;
                const int sw = tail_sw<CPR>(q);
                const uint32_t ad = (uint32_t)q * ROWB + (uint32_t)(((cb * 8 + fq) ^ sw) << 4);
                a_addr[b] = ((tapmask[b] >> tap) & 1u) ? ad : (zero_off | (ad & 0xF0u));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fx[TM], fw[TN1];
#pragma unroll
                for (int b = 0; b < TM; ++b) fx[b] = *(const uint4*)(tsm + (a_addr[b] ^ (uint32_t)(kk << 6)));
#pragma unroll
                for (int a = 0; a < TN1; ++a) {
                    const int row = wn * (CMID / WN1) + a * 16 + frow;
                    fw[a] = *(const uint4*)(Bs + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < TN1; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        union { uint4 u; bf16x8_t v; } ua, ub;
                        ua.u = fw[a];
                        ub.u = fx[b];
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a][b], 0, 0, 0);
                    }
            }
            if (++cb == CB) {
                cb = 0;
                ++tap;
                if (++ts == 3) { ts = 0; ++tr; }
                tapoff = tr * W + ts;
            }
            // the next step's K tile has landed; behind it only younger ring tiles / the Wc pieces may still be in flight
            if (kt + NS - 1 < NKT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * BR) : "memory");
            else if (kt + 1 < NKT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WC_PW + (NS - 3) * BR) : "memory");
            __syncthreads();
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
#undef FAV_T_HSTAGE
        }
        if (NRED > 0 && !WA0_EARLY) FAV_T_STAGE_WA(0);
        if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 2] = wall_clock64();
        // ---- P1e: T2 = bf16(relu(acc + bias_b)) -> LDS [BM][CMID], operand layout (every wave is past the patch) ----
#pragma unroll
        for (int a = 0; a < TN1; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int row = wm * 64 + b * 16 + frow;
                const int c0 = wn * (CMID / WN1) + a * 16 + fq * 4;
                const float4 bq = *(const float4*)(bias_b_s + (c0 >> 4) * 20 + (c0 & 15));
                const float v0 = __fadd_rn(acc[a][b][0], bq.x), v1 = __fadd_rn(acc[a][b][1], bq.y);
                const float v2 = __fadd_rn(acc[a][b][2], bq.z), v3 = __fadd_rn(acc[a][b][3], bq.w);
                const int sw = tail_sw<CPR>(row);
                *(uint2*)(tsm + row * ROWB + (((c0 >> 3) ^ sw) << 4) + ((c0 >> 2) & 1) * 8) =
                    make_uint2(relu_bf16x2(pack_bf16x2(v0, v1)), relu_bf16x2(pack_bf16x2(v2, v3)));
            }
    }
    // wave w owns pixel rows [w*RP, w*RP + RP): their T2 fragments (all CMID k) live in registers from here on
    uint4 t2f[TM2][KS2];
    if (HAS3X3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first chunk's weights
        __syncthreads();                                      // T2 tile complete, weight pieces of every wave visible
#pragma unroll
        for (int b = 0; b < TM2; ++b) {
            const int row = wave * RP + b * 16 + frow;
            const int sw = tail_sw<CPR>(row);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) t2f[b][ks] = *(const uint4*)(tsm + row * ROWB + (((ks * 4 + fq) ^ sw) << 4));
        }
        __syncthreads();                                      // region A becomes the Y-chunk image
        if constexpr (P1G) {                                  // ... and, for the generic conv_b, the home of the Wc buffers
            FAV_T_STAGE_WC(0);
            FAV_T_STAGE_WC(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // ---- conv_c alone: every wave reads the fragments of ITS rows straight from global memory (each byte of the
        //      tile is needed by one wave only, so an LDS round trip and its two barriers would buy nothing) ----------
        FAV_T_STAGE_WC(0);
        // row b*16 + frow of this wave: output pixel m -> its input pixel (the same one unless the 1x1 is strided);
        // offsets are relative to the first frame the tile touches, rows beyond M fail the range check (zeros)
        const int v0 = (int)fastdiv((uint32_t)m0, p.div_hw);
        const __amdgpu_buffer_rsrc_t srd_a =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1 + (long long)v0 * p.in_HW * CMID), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int b = 0; b < TM2; ++b) {
            const int m = m0 + wave * RP + b * 16 + frow;
            const uint32_t vl = fastdiv((uint32_t)m, p.div_hw);
            const uint32_t pix = (uint32_t)m - vl * (uint32_t)p.HW;
            const uint32_t oy = fastdiv(pix, p.div_w), ox = pix - oy * (uint32_t)p.W;
            const uint32_t roff = m < p.M ? (((vl - (uint32_t)v0) * (uint32_t)p.in_HW + oy * (uint32_t)(p.in_stride * p.in_W) + ox * (uint32_t)p.in_stride) * ROWB)
                                          : 0x80000000u;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(srd_a, (int)(roff + (ks * 4 + fq) * 16), 0, 0);
                t2f[b][ks] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
        FAV_T_BIAS_TO_LDS(true, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first chunk's weights (and the fragments)
        __syncthreads();
    }
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 3] = wall_clock64();

    // ---- P2 ---------------------------------------------------------------------------------------------------
    // lane (frow, fq) finishes channels 64j + 16fq .. + 15 of pixel rows wave*RP + b*16 + frow
    const __amdgpu_buffer_rsrc_t srd_y =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)m0 * COUT), 0, rows_valid * COUT * 2, 0x00020000);
    u32x4_t rcur[TM2][2];
    uint32_t drop_v[TM2], drop_pix[TM2];
#pragma unroll
    for (int b = 0; b < TM2; ++b) {
        const uint32_t m = (uint32_t)(m0 + wave * RP + b * 16 + frow);
        const uint32_t vl = fastdiv(m, p.div_hw);
        drop_v[b] = (uint32_t)p.drop.v0 + vl;
        drop_pix[b] = m - vl * (uint32_t)p.HW;
    }
    constexpr int NA3 = NRED > 0 ? NRED / 16 : 1;
    f32x4_t acc3[NA3][TM2];
#pragma unroll
    for (int a = 0; a < NA3; ++a)
#pragma unroll
        for (int b = 0; b < TM2; ++b) acc3[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Fully unrolled: in straight-line code hipcc counts its own loads and stores exactly, so the wait it places in
    // front of the residual pin below leaves the previous chunk's stores in flight; as a loop it merges the first
    // iteration's state and falls back to vmcnt(0), i.e. a write round trip per chunk.
#pragma unroll
    for (int j = 0; j < NCHUNK; ++j) {
        const unsigned char* const wcb = regb + (j % WCN) * WC_BYTES;
        const unsigned char* const wab = tsm + (((j + 1) & 1) ? p.wa_off1 : p.wa_off0);
        // residual of THIS chunk has landed; then the requests of the NEXT chunk (the other weight buffers: every
        // wave is past chunk j-1, the barrier at its end says so)
        if ((j == 1 || j == 2) && FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + (j == 1 ? 12 : 14)] = __builtin_amdgcn_s_memtime();
        if (HAS_RES) {
#pragma unroll
            for (int b = 0; b < TM2; ++b) {
                asm volatile("" : "+v"(rnext[b][0]), "+v"(rnext[b][1]));
                rcur[b][0] = rnext[b][0];
                rcur[b][1] = rnext[b][1];
            }
        }
        if (j == 1 && FAV_DBG(p) && tid == 0) { asm volatile("s_nop 0" :: "v"(rcur[TM2 - 1][1])); FAV_DBG(p)[blockIdx.x * 16ull + 13] = __builtin_amdgcn_s_memtime(); }
        if (j + 1 < NCHUNK) {
            if (BARQ == 2) {
                if ((j & 1) == 0 && j + 2 < NCHUNK) { FAV_T_STAGE_WC(j + 2); FAV_T_STAGE_WC(j + 3); }
            } else if (WC2) FAV_T_STAGE_WC(j + 1);
            if (NRED > 0) FAV_T_STAGE_WA(j + 1);
            FAV_T_LOAD_RES(j + 1)
        }
        if (j == 1 && FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 6] = __builtin_amdgcn_s_memtime();
        // -- A: acc2 = T2 x Wc[j]^T : RP pixels x 64 channels
        f32x4_t acc2[4][TM2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < TM2; ++b) acc2[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            uint4 fw[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = a * 16 + frow;
                fw[a] = *(const uint4*)(wcb + (ks >> 1) * 8192 + row * 128 + ((((ks & 1) * 4 + fq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < TM2; ++b) {
                    union { uint4 u; bf16x8_t v; } ua, ub;
                    ua.u = fw[a];
                    ub.u = t2f[b][ks];
                    acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc2[a][b], 0, 0, 0);
                }
        }
        if (j == 1 && FAV_DBG(p) && tid == 0) { asm volatile("s_nop 0" :: "v"(acc2[3][TM2 - 1])); FAV_DBG(p)[blockIdx.x * 16ull + 7] = __builtin_amdgcn_s_memtime(); }
        if (!WC2 && j + 1 < NCHUNK) {      // single Wc buffer: every wave has read Wc[j], the next chunk's may overwrite it
            FAV_BAR();
            FAV_T_STAGE_WC(j + 1);
        }
        // -- B: epilogue in registers: acc2[a][b][r] is channel 64j + 16fq + 4a + r of pixel row b*16 + frow
        {
            float bia[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *(const float4*)(bias_c_s + (j * 4 + fq) * 20 + 4 * q);
                bia[4 * q] = bq.x; bia[4 * q + 1] = bq.y; bia[4 * q + 2] = bq.z; bia[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int b = 0; b < TM2; ++b) {
                const int row = wave * RP + b * 16 + frow;
                const int n = j * 64 + fq * 16;
                uint32_t draws[4] = {~0u, ~0u, ~0u, ~0u};
                uint32_t edraws[4] = {~0u, ~0u, ~0u, ~0u};
                if (p.drop.site >= 0) {
                    const uint32_t chunk = (uint32_t)(((long long)drop_pix[b] * COUT + n) >> 4);
                    const uint4 w4 = drop_draws16(p.drop, drop_v[b], chunk);
                    draws[0] = w4.x; draws[1] = w4.y; draws[2] = w4.z; draws[3] = w4.w;
                    if constexpr (RESE) {         // the entry site's draws for the same 16 elements of the same virtual frame
                        DropParams de = p.drop;
                        de.site = p.site_e;
                        const uint4 e4 = drop_draws16(de, drop_v[b], chunk);
                        edraws[0] = e4.x; edraws[1] = e4.y; edraws[2] = e4.z; edraws[3] = e4.w;
                    }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = __fadd_rn(acc2[2 * g + (k >> 2)][b][k & 3], bia[8 * g + k]);
                    if (HAS_RES) {
                        uint32_t rw[4] = {rcur[b][g][0], rcur[b][g][1], rcur[b][g][2], rcur[b][g][3]};
                        if constexpr (RESE) {     // cached pixel -> the bf16 the entry dropout would have stored for this sample
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float lo = FAV_DROP_APPLY(bf16_bits_to_f32(rw[k] & 0xFFFFu), edraws, 8 * g + 2 * k, p.drop);
                                const float hi = FAV_DROP_APPLY(bf16_bits_to_f32(rw[k] >> 16), edraws, 8 * g + 2 * k + 1, p.drop);
                                rw[k] = pack_bf16x2(lo, hi);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[2 * k] = __fadd_rn(v[2 * k], bf16_bits_to_f32(rw[k] & 0xFFFFu));
                            v[2 * k + 1] = __fadd_rn(v[2 * k + 1], bf16_bits_to_f32(rw[k] >> 16));
                        }
                    }
                    if (p.drop.site >= 0) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = FAV_DROP_APPLY(v[k], draws, 8 * g + k, p.drop);
                    }
                    u32x4_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                    if (RELU) {      // on the packed pairs: half the instructions of eight v_max_f32 (scale > 0: the sign is the value's own)
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = relu_bf16x2(o[k]);
                    }
                    if (!LINEST) __builtin_amdgcn_raw_buffer_store_b128(o, srd_y, (row * COUT + n + 8 * g) * 2, 0, 0);
                    if (LINEST) *(u32x4_t*)(ych + row * 128 + (((2 * fq + g) ^ (row & 7)) << 4)) = o;
                }
            }
        }
        if (j == 1 && FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 8] = __builtin_amdgcn_s_memtime();
        // -- C: acc3 += Ychunk (this wave's own rows: its LDS writes are in order, no barrier) x Wa[:, 64j .. 64j+63]^T
        if constexpr (LINEST) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                // Y leaves the CU in WHOLE 128-byte lines: the wave reads its own rows of the Y-chunk image back in line order
                // (lane l: 16-byte piece l % 8 of row 8 i + l / 8) and one store instruction covers 8 full lines, instead of 16
                // rows x four 16-byte pieces at a 32-byte pitch (two instructions per line) straight from the accumulator layout:
                // P2 is bound by the CU's own memory path, and that path is priced per line touched (profiles/r3r_linest_ab.txt:
                // -2.0 % / -2.5 % on the layer-1 / layer-2 tails; same number of stores per chunk, so the counted waits stand)
#pragma unroll
                for (int i = 0; i < TM2 * 2; ++i) {
                    const int row = wave * RP + i * 8 + (lane >> 3);
                    const u32x4_t o = *(const u32x4_t*)(ych + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(o, srd_y, (row * COUT + j * 64 + (lane & 7) * 8) * 2, 0, 0);
                }
            }
        }
        if constexpr (NRED > 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 fy[TM2];
#pragma unroll
                for (int b = 0; b < TM2; ++b) {
                    const int row = wave * RP + b * 16 + frow;
                    fy[b] = *(const uint4*)(ych + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < NA3; ++a) {
                    const int row = a * 16 + frow;
                    union { uint4 u; bf16x8_t v; } ua;
                    ua.u = *(const uint4*)(wab + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
#pragma unroll
                    for (int b = 0; b < TM2; ++b) {
                        union { uint4 u; bf16x8_t v; } ub;
                        ub.u = fy[b];
                        acc3[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc3[a][b], 0, 0, 0);
                    }
                }
            }
        }
        if (j == 1 && FAV_DBG(p) && tid == 0) { asm volatile("s_nop 0" :: "v"(acc3[NA3 - 1][TM2 - 1])); FAV_DBG(p)[blockIdx.x * 16ull + 9] = __builtin_amdgcn_s_memtime(); }
        // the next chunk's weight pieces of this wave have landed (its residual loads and this chunk's stores stay in
        // flight), then the barrier publishes every wave's pieces and retires this chunk's reads of the current buffers
        if (j + 1 < NCHUNK && (BARQ == 1 || (j & 1))) {
            // WC2: the youngest operations are the next residual loads and this chunk's stores; otherwise the Wc pieces
            // were issued after the residual loads, so only the stores may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WC2 && HAS_RES) ? 4 * TM2 : 2 * TM2) : "memory");
            if (j == 1 && FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 10] = __builtin_amdgcn_s_memtime();
            FAV_BAR();
            if (j == 1 && FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 11] = __builtin_amdgcn_s_memtime();
        }
    }
#undef FAV_T_LOAD_RES
#undef FAV_T_STAGE_WC
#undef FAV_T_STAGE_WA
#undef FAV_T_BIAS_TO_LDS

    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 4] = wall_clock64();
    // ---- P3: t1' = bf16(relu(acc3 + bias_a)): lane holds channels 64*g3 + 16fq .. + 15 of its pixel rows ----------
    if constexpr (NRED > 0) {
        const __amdgpu_buffer_rsrc_t srd_t =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1n + (long long)m0 * NRED), 0, rows_valid * NRED * 2, 0x00020000);
#pragma unroll
        for (int g3 = 0; g3 < G3; ++g3) {
            float bia[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *(const float4*)(bias_a_s + (g3 * 4 + fq) * 20 + 4 * q);
                bia[4 * q] = bq.x; bia[4 * q + 1] = bq.y; bia[4 * q + 2] = bq.z; bia[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int b = 0; b < TM2; ++b) {
                const int row = wave * RP + b * 16 + frow;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = __fadd_rn(acc3[g3 * 4 + 2 * g + (k >> 2)][b][k & 3], bia[8 * g + k]);
                    const u32x4_t o = {relu_bf16x2(pack_bf16x2(v[0], v[1])), relu_bf16x2(pack_bf16x2(v[2], v[3])), relu_bf16x2(pack_bf16x2(v[4], v[5])),
                                       relu_bf16x2(pack_bf16x2(v[6], v[7]))};
                    __builtin_amdgcn_raw_buffer_store_b128(o, srd_t, (row * NRED + g3 * 64 + fq * 16 + 8 * g) * 2, 0, 0);
                }
            }
        }
    }
    if (FAV_DBG(p)) {
        __syncthreads();
        if (tid == 0) FAV_DBG(p)[blockIdx.x * 16ull + 5] = wall_clock64();
    }
}
#undef FAV_BAR

// ---------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max pool, NHWC bf16; one thread = 8 channels of one output
// pixel (16-B loads/stores).  HBM-bound.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int n,
                                                           int H, int W, int C, int Ho, int Wo) {
    const int cch = C >> 3;
    const long long total = (long long)n * Ho * Wo * cch;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % cch);
        const long long pix = idx / cch;
        const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho);
        const long long img = pix / ((long long)Wo * Ho);
        float best[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 - 1 + r;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = ow * 2 - 1 + s;
                if ((unsigned)iw >= (unsigned)W) continue;
                const uint4 v = x[((img * H + ih) * W + iw) * cch + c];
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    best[2 * j] = fmaxf(best[2 * j], bf16_bits_to_f32(vw[j] & 0xFFFFu));
                    best[2 * j + 1] = fmaxf(best[2 * j + 1], bf16_bits_to_f32(vw[j] >> 16));
                }
            }
        }
        y[idx] = make_uint4(pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]),
                            pack_bf16x2(best[4], best[5]), pack_bf16x2(best[6], best[7]));
    }
}

// ---------------------------------------------------------------------------
// Global average pool [n][HW][C] bf16 -> [n][C] bf16: sequential fp32 sum over
// HW (the oracle's order), * fp32(1/HW), optional dropout, one bf16 rounding.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const uint4* __restrict__ x_, uint4* __restrict__ y_, int n,
                                                      int HW, int C, float inv_hw, DropParams drop, long long g_x, long long g_y) {
    // grouped launch: member blockIdx.y reads / writes g_x / g_y bytes further on
    const uint4* __restrict__ x = (const uint4*)((const char*)x_ + (long long)blockIdx.y * g_x);
    uint4* __restrict__ y = (uint4*)((char*)y_ + (long long)blockIdx.y * g_y);
    const int cch = C >> 4;  // 16-channel chunks = pairs of uint4
    const long long total = (long long)n * cch;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % cch);
        const long long img = idx / cch;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const uint4* src = x + (img * HW * cch + c) * 2;
        for (int i = 0; i < HW; ++i) {
            const uint4 v0 = src[(long long)i * cch * 2], v1 = src[(long long)i * cch * 2 + 1];
            const uint32_t vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[2 * j] = __fadd_rn(acc[2 * j], bf16_bits_to_f32(vw[j] & 0xFFFFu));
                acc[2 * j + 1] = __fadd_rn(acc[2 * j + 1], bf16_bits_to_f32(vw[j] >> 16));
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __fmul_rn(acc[j], inv_hw);
        if (drop.site >= 0) {
            const uint4 w = drop_draws16(drop, (uint32_t)(drop.v0 + img), (uint32_t)c);
            const uint32_t draws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = FAV_DROP_APPLY(acc[j], draws, j, drop);
        }
        y[idx * 2] = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                                pack_bf16x2(acc[6], acc[7]));
        y[idx * 2 + 1] = make_uint4(pack_bf16x2(acc[8], acc[9]), pack_bf16x2(acc[10], acc[11]),
                                    pack_bf16x2(acc[12], acc[13]), pack_bf16x2(acc[14], acc[15]));
    }
}

// ---------------------------------------------------------------------------
// Entry dropout of the MC-Dropout suffix: the cached prefix output x[n_img][E]
// is expanded to out[v - v0][E] = dropout_{t,frame}(x[v % n_img]) for the chunk
// of virtual frames [v0, v0 + n_out).  HBM-bound.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void entry_dropout_kernel(const uint4* __restrict__ x, uint4* __restrict__ out,
                                                            long long chunks_per_frame, int n_out, DropParams drop) {
    // A thread owns one 16-element chunk of one cached frame and writes every sample t of it
    // that falls in [v0, v0 + n_out): the cached tensor is read once, not once per sample.
    const long long total = chunks_per_frame * drop.n_img;
    const long long v_end = drop.v0 + n_out;
    const int t_lo = (int)(drop.v0 / drop.n_img), t_hi = (int)((v_end - 1) / drop.n_img);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long img = idx / chunks_per_frame;
        const uint32_t chunk = (uint32_t)(idx - img * chunks_per_frame);
        const uint4* src = x + idx * 2;
        const uint4 a0 = src[0], a1 = src[1];
        const uint32_t vw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        for (int t = t_lo; t <= t_hi; ++t) {
            const long long v = (long long)t * drop.n_img + img;
            if (v < drop.v0 || v >= v_end) continue;
            const uint4 w = drop_draws16_ti(drop, (uint32_t)t, (uint32_t)img, chunk);
            const uint32_t draws[4] = {w.x, w.y, w.z, w.w};
            uint32_t o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo = FAV_DROP_APPLY(bf16_bits_to_f32(vw[j] & 0xFFFFu), draws, 2 * j, drop);
                const float hi = FAV_DROP_APPLY(bf16_bits_to_f32(vw[j] >> 16), draws, 2 * j + 1, drop);
                o[j] = pack_bf16x2(lo, hi);
            }
            uint4* dst = out + ((v - drop.v0) * chunks_per_frame + chunk) * 2;
            dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
            dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
}

// ---------------------------------------------------------------------------
// Entry dropout + the first 1x1 reduce behind it, one launch:
//   y[v - v0]  = dropout_{t, frame}(x[v % n_img])                      (what entry_dropout_kernel writes)
//   t1[v - v0] = bf16(relu(conv1x1(y[v - v0], Wa) + bias_a))           (the next block's conv1, C -> NRED)
// Layer by layer the reduce reads all of y back (12.3 GB per step of the headline); here a block keeps 128 pixels
// of the CACHED tensor in registers - in the epilogue layout of bottleneck_tail_kernel, lane (pixel frow, quad fq)
// holds channels 64j + 16fq .. + 15 of its pixels - and walks over the samples t of the chunk: one Philox call per
// 16 channels, y stored straight from registers, the same 16 channels written to the wave's own rows of a
// 64-channel LDS image and multiplied with the resident Wa (k ascending in 64-channel chunks, two 32-deep MFMA
// steps each: the order conv_igemm_kernel uses, so t1 is bit-identical to the separate launches).  No global load
// and no workgroup barrier inside the sample loop; the cached tensor is read once, not once per sample.
// ---------------------------------------------------------------------------
struct EntryReduceParams {
    const uint16_t* x;        // cached prefix output [n_img][HW][C]
    uint16_t* y;              // [n_out][HW][C]; null: the dropped copies are not stored (the next tail recomputes them, RESE)
    const uint16_t* wa; const float* bias_a;     // [NRED][C]
    uint16_t* t1;             // [n_out][HW][NRED]
    int HW, M, n_out;         // M = n_img * HW pixels of the cached tensor
    DropParams drop;
    FastDiv div_hw;
};

// Two blocks per CU: under __launch_bounds__(256, 3) the 64 cached-pixel registers + 32 accumulators + one Philox call spilled 21
// registers into the sample loop (1.93 -> 1.72 ms per headline step without them, profiles/r4f_entry_occupancy_ab.txt).
template <int C, int NRED>
__global__ __launch_bounds__(256, 2) void entry_reduce_kernel(const EntryReduceParams p) {
    constexpr int NW = 4, BM = 32 * NW, TM2 = 2, NJ = C / 64, NA3 = NRED / 16, G3 = NRED / 64;
    constexpr int WA_BYTES = NRED * 128;                  // one 64-k chunk of Wa: NRED rows x 128 B
    constexpr int WA_PW = WA_BYTES / 1024 / NW;           // LDS-DMA pieces per wave per chunk
    static_assert(WA_BYTES % (1024 * NW) == 0 && NRED % 64 == 0 && C % 64 == 0, "tile shape");
    // [Wa: NJ chunks | Y chunk image [BM][64] | bias (16-channel chunks 20 floats apart)]
    __shared__ __attribute__((aligned(16))) unsigned char esm[NJ * WA_BYTES + BM * 128 + NRED * 5];
    unsigned char* const ych = esm + NJ * WA_BYTES;
    float* const bias_s = (float*)(esm + NJ * WA_BYTES + BM * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int frow = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    for (int i = tid; i < NRED; i += 256) bias_s[(i >> 4) * 20 + (i & 15)] = p.bias_a[i];
    // ---- Wa -> LDS (rows in the order tail_row_perm gives, 16-B chunks XOR (row & 7)), once per block ----
    {
        const uint32_t lds_base =
            __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)esm);
        const __amdgpu_buffer_rsrc_t srd_wa = __builtin_amdgcn_make_buffer_rsrc((void*)p.wa, 0, NRED * C * 2, 0x00020000);
        const int wrow = lane >> 3, wch = (lane & 7) ^ (wrow & 7);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < WA_PW; ++i) {
                const int pc = wave_u + i * NW;
                lds_dma16(srd_wa, (uint32_t)((tail_row_perm(pc * 8 + wrow) * C + wch * 8) * 2), (uint32_t)(j * 128),
                          __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(j * WA_BYTES + pc * 1024)));
            }
    }
    // ---- this wave's 32 pixels of the cached tensor, epilogue layout ----
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)p.M * (uint32_t)(C * 2), 0x00020000);
    u32x4_t xr[TM2][NJ][2];
    uint32_t img[TM2], pix[TM2], moff[TM2];
#pragma unroll
    for (int b = 0; b < TM2; ++b) {
        const uint32_t m = (uint32_t)(m0 + wave * 32 + b * 16 + frow);
        img[b] = fastdiv(m, p.div_hw);
        pix[b] = m - img[b] * (uint32_t)p.HW;
        moff[b] = m < (uint32_t)p.M ? m : 0x7fffffffu / (uint32_t)(C * 2);      // beyond the tensor: loads read zeros, stores are dropped
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int off = (int)(moff[b] * (uint32_t)(C * 2)) + (j * 64 + fq * 16) * 2;
            xr[b][j][0] = __builtin_amdgcn_raw_buffer_load_b128(srd_x, off, 0, 0);
            xr[b][j][1] = __builtin_amdgcn_raw_buffer_load_b128(srd_x, off + 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const long long v_end = p.drop.v0 + p.n_out;
    const int t_lo = (int)(p.drop.v0 / p.drop.n_img), t_hi = (int)((v_end - 1) / p.drop.n_img);
    for (int t = t_lo; t <= t_hi; ++t) {
        // rows of sample t: virtual frame v = t * n_img + img; its output row block starts (v - v0) * HW pixels into y / t1
        const long long sample0 = ((long long)t * p.drop.n_img - p.drop.v0) * p.HW;      // may be negative for a chunk's first sample
        const __amdgpu_buffer_rsrc_t srd_y =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + sample0 * C), 0, (uint32_t)p.M * (uint32_t)(C * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t srd_t =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1 + sample0 * NRED), 0, (uint32_t)p.M * (uint32_t)(NRED * 2), 0x00020000);
        uint32_t orow[TM2];
#pragma unroll
        for (int b = 0; b < TM2; ++b) {
            const long long v = (long long)t * p.drop.n_img + img[b];
            orow[b] = (v >= p.drop.v0 && v < v_end) ? moff[b] : 0x7fffffffu / (uint32_t)(C * 2);
        }
        f32x4_t acc3[NA3][TM2];
#pragma unroll
        for (int a = 0; a < NA3; ++a)
#pragma unroll
            for (int b = 0; b < TM2; ++b) acc3[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int b = 0; b < TM2; ++b) {
                const int row = wave * 32 + b * 16 + frow;
                const uint32_t chunk = pix[b] * (uint32_t)(C / 16) + (uint32_t)(4 * j + fq);
                const uint4 w4 = drop_draws16_ti(p.drop, (uint32_t)t, img[b], chunk);
                const uint32_t draws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint32_t o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t w = xr[b][j][g][k];
                        const float lo = FAV_DROP_APPLY(bf16_bits_to_f32(w & 0xFFFFu), draws, 8 * g + 2 * k, p.drop);
                        const float hi = FAV_DROP_APPLY(bf16_bits_to_f32(w >> 16), draws, 8 * g + 2 * k + 1, p.drop);
                        o[k] = pack_bf16x2(lo, hi);
                    }
                    const u32x4_t ov = {o[0], o[1], o[2], o[3]};
                    if (p.y) __builtin_amdgcn_raw_buffer_store_b128(ov, srd_y, (int)(orow[b] * (uint32_t)(C * 2)) + (j * 64 + fq * 16 + 8 * g) * 2, 0, 0);
                    *(u32x4_t*)(ych + row * 128 + (((2 * fq + g) ^ (row & 7)) << 4)) = ov;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned char* const wab = esm + j * WA_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 fy[TM2];
#pragma unroll
                for (int b = 0; b < TM2; ++b) {
                    const int row = wave * 32 + b * 16 + frow;
                    fy[b] = *(const uint4*)(ych + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < NA3; ++a) {
                    const int row = a * 16 + frow;
                    union { uint4 u; bf16x8_t v; } ua;
                    ua.u = *(const uint4*)(wab + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
#pragma unroll
                    for (int b = 0; b < TM2; ++b) {
                        union { uint4 u; bf16x8_t v; } ub;
                        ub.u = fy[b];
                        acc3[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc3[a][b], 0, 0, 0);
                    }
                }
            }
        }
        // t1 = bf16(relu(acc3 + bias)): lane holds channels 64*g3 + 16fq .. + 15 of its pixel rows
#pragma unroll
        for (int g3 = 0; g3 < G3; ++g3) {
            float bia[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *(const float4*)(bias_s + (g3 * 4 + fq) * 20 + 4 * q);
                bia[4 * q] = bq.x; bia[4 * q + 1] = bq.y; bia[4 * q + 2] = bq.z; bia[4 * q + 3] = bq.w;
            }
#pragma unroll
            for (int b = 0; b < TM2; ++b)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = __fadd_rn(acc3[g3 * 4 + 2 * g + (k >> 2)][b][k & 3], bia[8 * g + k]);
                    const u32x4_t o = {relu_bf16x2(pack_bf16x2(v[0], v[1])), relu_bf16x2(pack_bf16x2(v[2], v[3])), relu_bf16x2(pack_bf16x2(v[4], v[5])),
                                       relu_bf16x2(pack_bf16x2(v[6], v[7]))};
                    __builtin_amdgcn_raw_buffer_store_b128(o, srd_t, (int)(orow[b] * (uint32_t)(NRED * 2)) + (g3 * 64 + fq * 16 + 8 * g) * 2, 0, 0);
                }
        }
    }
}

// ---------------------------------------------------------------------------
// Chained stream-K GEMM: the ViT encoder's linear layers, Y[m, n] = act((sum_k A[m, k] W[n, k] + bias[n]) + res[m, n]).
//
// At the per-GPU share of BASELINE configs[4] (64 frames: M = 12 608 rows) a linear layer is 594 - 2 376 tiles of 128 x 128 on
// 256 CUs: 2.3 - 9.3 tiles per CU, and the launch ends when the CUs with one tile more than the others finish - 77 % of the chip
// at 2.32 tiles per CU.  Here the unit of work is one 32-deep K step of one tile, and a persistent grid deals the steps out in
// equal contiguous shares, so every workgroup does the same number of steps (+- 1) whatever the tile count is.
//
// A share that ends inside a tile leaves an fp32 partial accumulator in a workspace slot; the workgroup whose share starts inside
// that tile LOADS it as its initial accumulator and continues the chain of MFMAs - the sum an output element sees is the one
// uninterrupted ascending-k chain the tile-per-block kernel runs, so the results are BIT-IDENTICAL to conv_igemm_kernel's (no
// split-K reordering, no dependence on the grid or on M: every fixture stays).  The chain costs no waiting because a workgroup
// walks its share BACKWARDS by tile: first the head part of its last tile (published at once), then its whole tiles, and only at
// the very end the tail part of its first tile, whose head its predecessor published at ITS very start.
//
// Placement: blocks b, b + 8, ... share an XCD (observed; speed only).  The tiles are cut into 8 contiguous runs, one per such
// group, n fastest, and a group's Q = grid / 8 workgroups share the run's steps: neighbours in a group work on the same rows of A
// (one L2), and a workgroup only ever waits for block b - 8, which is dispatched before it and needs nobody to publish.  The
// launcher sizes the grid to what is co-resident anyway and keeps shares >= one tile (no workgroup both waits and publishes).
// Hand-off: plain 16-byte stores, every wave's vmcnt(0), barrier, one lane's agent-scope release + vmcnt(0) + relaxed flag store;
// the consumer polls relaxed (bounded: gives up into *err), acquires at agent scope, vmcnt(0), barrier, plain loads - the
// MI355X guide's valid form.  Flags carry the launch's epoch, so nothing is reset between launches.
//
// Tile 128 x 128, 4 waves of 64 x 64, 32-deep steps through a three-slot LDS-DMA ring that never drains: the steps of the next
// tile are in flight while this tile's epilogue runs.  Epilogue in registers as conv_igemm_kernel's EPI = 1 (weight rows staged
// in tail_row_perm-like order: a lane ends up with 16 consecutive channels), the tile's bias arrives by LDS-DMA with its first step.
// ---------------------------------------------------------------------------
struct GemmSkParams {
    const uint16_t* a;        // [M][K] bf16
    const uint16_t* w;        // [N][K] bf16
    const float* bias;        // [N]
    const uint16_t* res;      // [M][N] bf16 or null; may be y (in-place residual: a tile reads its own rows, then writes them)
    uint16_t* y;              // [M][N] bf16
    int M, N, K, ksteps;      // ksteps = K / 32
    int tiles_n, tiles;       // tiles = ceil(M / 128) * (N / 128)
    int act;                  // 0 none, 2 GELU
    int Q;                    // workgroups per XCD group; grid = 8 Q
    float* ws;                // [grid][16][256] float4: partial accumulators
    uint32_t* flags;          // [grid]: == epoch once the slot is published
    uint32_t epoch;
    uint32_t* err;            // set to 1 if a consumer gave up waiting
    FastDiv div_tn;           // / tiles_n
    unsigned long long* dbg;  // FAV_SK_DBG: per-workgroup sums of s_memtime ticks per loop phase (null in normal runs)
};

// wait until at most n vector-memory operations of this wave are outstanding (n uniform; one of the counts the kernel produces)
__device__ __forceinline__ void sk_wait_vmcnt(int n) {
#define FAV_SKW(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
    switch (n) {
        FAV_SKW(4) FAV_SKW(5) FAV_SKW(12) FAV_SKW(13) FAV_SKW(20) FAV_SKW(21) FAV_SKW(28) FAV_SKW(29) FAV_SKW(36) FAV_SKW(37)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef FAV_SKW
}

__global__ __launch_bounds__(256, 3) void gemm_streamk_kernel(const GemmSkParams p) {
    constexpr int BM = 128, BN = 128, ROWB = 64, STAGE = (BM + BN) * ROWB, A_BYTES = BM * ROWB, NS = 3;
    constexpr int TM = 4, TN = 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE + 4 * 1024];   // ring | four bias buffers (128 floats + the piece's zero half):
                                                                                          // the staging side runs up to two SEGMENTS ahead (a head part may be one step long)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- this workgroup's schedule inside its group's run of tiles [tx0, tx0 + ntx) ------------------------------------------------
    // The run is walked in rounds of Q consecutive tiles, one tile per workgroup - all workgroups of the group on the same rows of A
    // and the same K slice at the same time, as a tile-per-block launch has them (dealing the WHOLE run out in contiguous shares
    // was 2x slower: a share of several tiles walks along n alone, 43 % of its L2 requests missed) - and only the last Q + r tiles
    // (one round plus the remainder) are dealt out by K steps: shares of 1 .. 2 tiles, [u0, u1) in steps from the region's start.
    const int wg = blockIdx.x, xg = wg & 7, qg = wg >> 3;
    const int tq = p.tiles >> 3, tr = p.tiles & 7;
    const int tx0 = xg * tq + (xg < tr ? xg : tr), ntx = tq + (xg < tr ? 1 : 0);
    const int ndp = ntx / p.Q - 1;                // whole rounds in front of the stream-K region (the launcher keeps ntx >= Q)
    if (ndp < 0) return;
    const int sk0 = ndp * p.Q, nsk = ntx - sk0;   // the region: tiles [sk0, ntx) of the run, Q <= nsk < 2 Q
    const long long Ug = (long long)nsk * p.ksteps;
    // (64-bit divisions run on the vector ALUs: back into scalar registers, the LDS-DMA statements take their offsets from there)
    const int u0 = __builtin_amdgcn_readfirstlane((int)(Ug * qg / p.Q)), u1 = __builtin_amdgcn_readfirstlane((int)(Ug * (qg + 1) / p.Q));
    const int nsteps = ndp * p.ksteps + (u1 - u0);
    const int tf = u0 / p.ksteps, k0 = u0 - tf * p.ksteps;                  // first tile of the share (region-local), its first step
    const int tl = (u1 - 1) / p.ksteps, k1 = u1 - tl * p.ksteps;            // last tile, the step behind its last one (<= ksteps)
    // segments in processing order: [the rounds' tiles] [head part of the share's last tile] [its whole tiles] [tail part of its
    // first tile] - the share is walked backwards, so that a partial accumulator is published long before it is wanted
    const bool one = tf == tl;
    const int hasH = (one || k1 < p.ksteps) ? 1 : 0;                       // a single-tile share is one segment, filed as "head"
    const int hasT = (!one && k0 > 0) ? 1 : 0;
    const int tfull0 = tf + hasT, nfull = one ? 0 : (tl - (k1 < p.ksteps ? 1 : 0)) - tfull0 + 1;
    const int nseg = ndp + hasH + nfull + hasT;
    // segment j: run-local tile, steps [ka, kb)
#define FAV_SK_SEG(J, TILE, KA, KB)                                                   \
    do {                                                                              \
        const int j_ = (J) - ndp;                                                     \
        if (j_ < 0) { TILE = (J) * p.Q + qg; KA = 0; KB = p.ksteps; }                 \
        else if (hasH && j_ == 0) { TILE = sk0 + tl; KA = one ? k0 : 0; KB = k1; }    \
        else if (j_ - hasH < nfull) { TILE = sk0 + tfull0 + (j_ - hasH); KA = 0; KB = p.ksteps; } \
        else { TILE = sk0 + tf; KA = k0; KB = p.ksteps; }                             \
    } while (0)

    // ---- staging side ----------------------------------------------------------------------------------------------------------
    // rows of 64 B (32 k), a piece = 16 rows; lane -> row l >> 2, physical 16-B slot l & 3 holds chunk (l & 3) ^ swz(row)
    const int lrow = lane >> 2, lch = (lane & 3) ^ lds_swz<32>(lrow);
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)((long long)p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_bias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.N * 4, 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;
    const uint32_t lds_base =
        __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)smem);
    const uint32_t lds_a = lds_base + wave_u * 2048, lds_b = lds_base + A_BYTES + wave_u * 2048;
    uint32_t a_voff[2], b_voff[2], bias_voff = OOB;
    int pj = 0, ptile = 0, pk = 0, pkb = 0;         // producer cursor: segment, tile, next step to stage, end of the segment
    bool pfirst = true;                             // the next staged step is the first of its segment
#define FAV_SK_PTILE()                                                                                           \
    do {                                                                                                         \
        const int tg_ = tx0 + ptile;                                                                             \
        const int tm_ = (int)fastdiv((uint32_t)tg_, p.div_tn), tn_ = tg_ - tm_ * p.tiles_n;                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                          \
            const int ra_ = tm_ * BM + (wave * 2 + i) * 16 + lrow;                                               \
            a_voff[i] = ra_ < p.M ? (uint32_t)(((long long)ra_ * p.K + lch * 8) * 2) : OOB;                      \
            int rb_ = (wave * 2 + i) * 16 + lrow;        /* LDS row of the B tile this lane fills -> the weight row it holds */ \
            rb_ = (rb_ & ~63) + 16 * ((rb_ & 15) >> 2) + 4 * ((rb_ >> 4) & 3) + (rb_ & 3);                       \
            b_voff[i] = (uint32_t)(((long long)(tn_ * BN + rb_) * p.K + lch * 8) * 2);                           \
        }                                                                                                        \
        bias_voff = lane < 32 ? (uint32_t)((tn_ * BN + lane * 4) * 4) : OOB;                                     \
    } while (0)
    // stage the producer cursor's step into ring slot SLOT; returns the pieces issued in NP
#define FAV_SK_STAGE(SLOT, NP)                                                                                   \
    do {                                                                                                         \
        const uint32_t soff_ = (uint32_t)(pk * ROWB);                                                            \
        NP = 4;                                                                                                  \
        if (pfirst) {                                                                                            \
            lds_dma16(srd_bias, bias_voff, 0u, __builtin_amdgcn_readfirstlane(lds_base + NS * STAGE + (uint32_t)(pj & 3) * 1024u)); \
            NP = 5;                                                                                              \
            pfirst = false;                                                                                      \
        }                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            lds_dma16(srd_a, a_voff[i], soff_, __builtin_amdgcn_readfirstlane(lds_a + (uint32_t)(SLOT) * STAGE + i * 1024)); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            lds_dma16(srd_b, b_voff[i], soff_, __builtin_amdgcn_readfirstlane(lds_b + (uint32_t)(SLOT) * STAGE + i * 1024)); \
        if (++pk == pkb) {                                                                                       \
            if (++pj < nseg) { FAV_SK_SEG(pj, ptile, pk, pkb); FAV_SK_PTILE(); pfirst = true; }                  \
        }                                                                                                        \
    } while (0)

    FAV_SK_SEG(0, ptile, pk, pkb);
    FAV_SK_PTILE();

    // ---- consumer side ---------------------------------------------------------------------------------------------------------
    int cj = 0, ctile = 0, ck = 0, ckb = 0;
    FAV_SK_SEG(0, ctile, ck, ckb);
    bool cfirst = true;
    f32x4_t acc[TN][TM];
    u32x4_t rr[TM][2];                               // residual of the current tile: 16 channels of 4 pixel rows
    const __amdgpu_buffer_rsrc_t srd_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.y), 0, p.res ? (int)((long long)p.M * p.N * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((long long)p.M * p.N * 2), 0x00020000);
#pragma unroll
    for (int b = 0; b < TM; ++b) rr[b][0] = rr[b][1] = (u32x4_t){0u, 0u, 0u, 0u};

    int np0, np1;
    FAV_SK_STAGE(0, np0);
    np1 = 0;
    if (nsteps > 1) FAV_SK_STAGE(1, np1);
    sk_wait_vmcnt(np1);                              // step 0 has landed (np1 = 0: everything)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int e_prev = 0;                                  // vector-memory operations issued in the previous iteration behind its stage
    int slot = 0;
    unsigned long long dsum[5] = {0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0;
    if (FAV_DBG(p) && tid == 0) FAV_DBG(p)[wg * 8ull + 6] = wall_clock64();
    for (int i = 0; i < nsteps; ++i) {
        int np = 0;
        if (FAV_DBG(p)) dt0 = __builtin_amdgcn_s_memtime();
        if (i + 2 < nsteps) { const int s2 = slot == 0 ? 2 : slot - 1; FAV_SK_STAGE(s2, np); }
        if (FAV_DBG(p)) { dt1 = __builtin_amdgcn_s_memtime(); dsum[0] += dt1 - dt0; dt0 = dt1; }
        int e_cur = 0;
        const int tg = tx0 + ctile;
        const int tm = (int)fastdiv((uint32_t)tg, p.div_tn), tn = tg - tm * p.tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        if (cfirst) {
            cfirst = false;
            if (ck > 0 && !(p.act & 0x100)) {
                // ---- the tile's head was computed by block wg - 8: wait for it, take its accumulator over -----------------------
                if (tid == 0) {
                    const uint32_t* fl = p.flags + (wg - 8);
                    int spins = 0;
                    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++spins > (1 << 22)) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // (descriptor + one per-lane offset: as sixteen 64-bit pointers the slot's addresses were hoisted out of the loop into 64 registers)
                const __amdgpu_buffer_rsrc_t srd_in = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.ws + (size_t)(wg - 8) * 65536), 0, 65536, 0x00020000);
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(srd_in, tid * 16, (a * TM + b) * 4096, 0);
                        acc[a][b] = (f32x4_t){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
                    }
                // the compiler's own wait for these loads goes HERE (an empty statement that redefines the registers): left to itself it
                // waits in front of the loop's MFMAs - vmcnt(0) in every step, which drains the LDS-DMA ring it does not know about
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) asm volatile("" : "+v"(acc[a][b]));
            } else {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
            if (p.res && ckb == p.ksteps) {          // this segment ends in the epilogue: its residual rows, requested a whole tile ahead
#pragma unroll
                for (int b = 0; b < TM; ++b)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        rr[b][q] = __builtin_amdgcn_raw_buffer_load_b128(
                            srd_res, (int)((((long long)(m0 + wm * 64 + b * 16 + frow)) * p.N + n0 + wn * 64 + 16 * fq + 8 * q) * 2), 0, 0);
                e_cur += 2 * TM;
            }
        }
        // ---- one 32-deep step ------------------------------------------------------------------------------------------------
        {
            const unsigned char* As = smem + slot * STAGE;
            const unsigned char* Bs = As + A_BYTES;
            uint4 fx[TM], fw[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int row = wm * 64 + b * 16 + frow;
                fx[b] = *(const uint4*)(As + row * ROWB + ((fq ^ lds_swz<32>(row)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int row = wn * 64 + a * 16 + frow;
                fw[a] = *(const uint4*)(Bs + row * ROWB + ((fq ^ lds_swz<32>(row)) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    union { uint4 u; bf16x8_t v; } ua, ub;
                    ua.u = fw[a];
                    ub.u = fx[b];
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[a][b], 0, 0, 0);
                }
        }
        if (FAV_DBG(p)) { asm volatile("s_nop 0" :: "v"(acc[TN - 1][TM - 1])); dt1 = __builtin_amdgcn_s_memtime(); dsum[1] += dt1 - dt0; dt0 = dt1; }
        // ---- end of a segment: epilogue, or publish the partial accumulator ----------------------------------------------------
        if (++ck == ckb) {
            if (ckb == p.ksteps) {
                const float* bias_s = (const float*)(smem + NS * STAGE + (cj & 3) * 1024);
                float bia[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *(const float4*)(bias_s + wn * 64 + 16 * fq + 4 * q);
                    bia[4 * q] = bq.x; bia[4 * q + 1] = bq.y; bia[4 * q + 2] = bq.z; bia[4 * q + 3] = bq.w;
                }
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int m = m0 + wm * 64 + b * 16 + frow;
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8) {
                        __builtin_amdgcn_sched_barrier(0);       // one 8-channel group at a time: the GELU's temporaries are not worth a fourth of the register file
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = __fadd_rn(acc[2 * h8 + (k >> 2)][b][k & 3], bia[8 * h8 + k]);
                        if (p.res) {
                            const uint32_t rw[4] = {rr[b][h8][0], rr[b][h8][1], rr[b][h8][2], rr[b][h8][3]};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                v[2 * k] = __fadd_rn(v[2 * k], bf16_bits_to_f32(rw[k] & 0xFFFFu));
                                v[2 * k + 1] = __fadd_rn(v[2 * k + 1], bf16_bits_to_f32(rw[k] >> 16));
                            }
                        }
                        if ((p.act & 0xff) == 1) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                        } else if ((p.act & 0xff) == 2) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] = fav_gelu(v[k]);
                        }
                        const u32x4_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                        // rows beyond M fail the descriptor's range check
                        __builtin_amdgcn_raw_buffer_store_b128(o, srd_y, m < p.M ? (int)(((long long)m * p.N + n0 + wn * 64 + 16 * fq + 8 * h8) * 2) : (int)OOB, 0, 0);
                    }
                }
                e_cur += 2 * TM;
            } else if (!(p.act & 0x100)) {
                const __amdgpu_buffer_rsrc_t srd_out = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.ws + (size_t)wg * 65536), 0, 65536, 0x00020000);
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        const u32x4_t v = {__float_as_uint(acc[a][b][0]), __float_as_uint(acc[a][b][1]), __float_as_uint(acc[a][b][2]), __float_as_uint(acc[a][b][3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, srd_out, tid * 16, (a * TM + b) * 4096, 0);
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's part of the slot has left
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(p.flags + wg, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (++cj < nseg) { FAV_SK_SEG(cj, ctile, ck, ckb); cfirst = true; }
        }
        // ---- step i + 1 has landed: everything issued behind its stage may stay in flight ----------------------------------------
        if (FAV_DBG(p)) { dt1 = __builtin_amdgcn_s_memtime(); dsum[2] += dt1 - dt0; dt0 = dt1; }
        if (i + 1 < nsteps) {
            sk_wait_vmcnt(e_prev + np + e_cur);
            if (FAV_DBG(p)) { dt1 = __builtin_amdgcn_s_memtime(); dsum[3] += dt1 - dt0; dt0 = dt1; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (FAV_DBG(p)) { dt1 = __builtin_amdgcn_s_memtime(); dsum[4] += dt1 - dt0; dt0 = dt1; }
        }
        e_prev = e_cur;
        slot = slot == 2 ? 0 : slot + 1;
    }
    if (FAV_DBG(p) && tid == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) FAV_DBG(p)[wg * 8ull + k] = dsum[k];
        FAV_DBG(p)[wg * 8ull + 5] = (unsigned long long)nsteps;
        FAV_DBG(p)[wg * 8ull + 7] = wall_clock64();
    }
#undef FAV_SK_STAGE
#undef FAV_SK_PTILE
#undef FAV_SK_SEG
}

// ---------------------------------------------------------------------------
// ViT kernels (BASELINE configs[4]).
// ---------------------------------------------------------------------------
// Token assembly: x[f][0] = pos[0] (class token folded in), x[f][1+p] = bf16(emb[f*np + p] + pos[1+p]).
__global__ __launch_bounds__(256) void vit_assemble_kernel(const uint16_t* __restrict__ emb, const float* __restrict__ pos,
                                                           uint16_t* __restrict__ x, int n, int ntok, int D) {
    const long long total = (long long)n * ntok * (D / 4);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c4 = (int)(idx % (D / 4));
        const long long row = idx / (D / 4);
        const int t = (int)(row % ntok);
        const long long f = row / ntok;
        const float4 pv = *(const float4*)(pos + (long long)t * D + c4 * 4);
        float v0 = pv.x, v1 = pv.y, v2 = pv.z, v3 = pv.w;
        if (t > 0) {
            const uint2 e = *(const uint2*)(emb + ((f * (ntok - 1) + (t - 1)) * D + c4 * 4));
            v0 = __fadd_rn(bf16_bits_to_f32(e.x & 0xFFFFu), v0); v1 = __fadd_rn(bf16_bits_to_f32(e.x >> 16), v1);
            v2 = __fadd_rn(bf16_bits_to_f32(e.y & 0xFFFFu), v2); v3 = __fadd_rn(bf16_bits_to_f32(e.y >> 16), v3);
        }
        *(uint2*)(x + row * D + c4 * 4) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
    }
}

// LayerNorm of bf16 rows (row r at x + r*ldx; D % 4 == 0, D <= 1024): a wave takes R consecutive rows, one after the other in
// arithmetic but with all their loads requested up front (one row per wave kept 1.5 KB in flight per wave: 2.7 TB/s).  Lane l owns
// the 4-element groups l, l+64, ...; sums are per-lane sequential, then a 6-level xor butterfly - the order
// oracle/fav_exact.c: fav_layernorm_rows restates.  Statistics, scale and shift in fp32, one bf16 rounding.
template <int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ x, long long ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ y, long long rows,
                                                        int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int ng = D >> 2;
    uint2 e[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = lane + 64 * j;
            e[r][j] = make_uint2(0u, 0u);
            if (g < ng && row0 + r < rows) e[r][j] = *(const uint2*)(x + (row0 + r) * ldx + 4 * g);
        }
    float4 gm[4], bt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = lane + 64 * j;
        gm[j] = bt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < ng) { gm[j] = *(const float4*)(gamma + 4 * g); bt[j] = *(const float4*)(beta + 4 * g); }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long row = row0 + r;
        if (row >= rows) break;
        float v[4][4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (lane + 64 * j < ng) {
                v[j][0] = bf16_bits_to_f32(e[r][j].x & 0xFFFFu); v[j][1] = bf16_bits_to_f32(e[r][j].x >> 16);
                v[j][2] = bf16_bits_to_f32(e[r][j].y & 0xFFFFu); v[j][3] = bf16_bits_to_f32(e[r][j].y >> 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) s = __fadd_rn(s, v[j][k]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = __fadd_rn(s, __shfl_xor(s, o, 64));
        const float mean = __fdiv_rn(s, (float)D);
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < ng) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[j][k] = __fsub_rn(v[j][k], mean);
                    s2 = __fadd_rn(s2, __fmul_rn(v[j][k], v[j][k]));
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 = __fadd_rn(s2, __shfl_xor(s2, o, 64));
        const float var = __fdiv_rn(s2, (float)D);
        const float rstd = __fdiv_rn(1.0f, fav_sqrtf(__fadd_rn(var, eps)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = lane + 64 * j;
            if (g < ng) {
                const float o0 = __fadd_rn(__fmul_rn(__fmul_rn(v[j][0], rstd), gm[j].x), bt[j].x);
                const float o1 = __fadd_rn(__fmul_rn(__fmul_rn(v[j][1], rstd), gm[j].y), bt[j].y);
                const float o2 = __fadd_rn(__fmul_rn(__fmul_rn(v[j][2], rstd), gm[j].z), bt[j].z);
                const float o3 = __fadd_rn(__fmul_rn(__fmul_rn(v[j][3], rstd), gm[j].w), bt[j].w);
                *(uint2*)(y + row * D + 4 * g) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            }
        }
    }
}

// Multi-head attention, head width 64, up to 256 tokens: one block per (frame, head), two blocks per CU.
//   qkv [n][T][3D] bf16 (Q | K | V, head h = columns 64h..64h+63 of each)  ->  out [n][T][D] bf16
// K and V live in LDS, both row-major (whole 16-byte chunks, XOR-swizzled), 55 KB for 197 tokens - nothing else does, so two
// blocks share a CU and one stages its K / V while the other computes.  A wave takes 16 queries at a time (197 tokens: 7 waves,
// two tiles each):
//   S = K Q^T on MFMA (lane: 4 consecutive keys of one query per key tile), keys >= T masked (last key tile only),
//   row max of the raw scores by the lane and two xor-shuffles, e = 2^fma(s, log2(e) / 8, -max log2(e) / 8) two elements per
//   instruction (fav_attn_exp2: 14 issue slots per score, was 20), row sum = two interleaved partial sums per lane, then the shuffles,
//   p = e * (1 / sum) rounded to bf16 IN REGISTERS: the S accumulator layout (4 consecutive keys per lane and key tile) is used
//   directly as the B operand of O = V^T P^T, whose k slot 8g + j of 32-key block b therefore holds key 32b + 16 (j >> 2) + 4g +
//   (j & 3) - a fixed permutation of the keys inside each block that the A operand follows: V^T comes out of LDS through gfx950's
//   transposing ds_read_b64_tr_b16, two per fragment, at rows 4g.. and 16 + 4g...  No P strip in LDS, no round trip.
// The operation order is the one oracle/fav_oracle.py: attention() + fav_attn_softmax_rows restate (attn_key_order there).
typedef short attn_v4s __attribute__((ext_vector_type(4)));
// XOR on the 16-byte chunk index of V's 128-byte rows: the 8 consecutive rows x 32 bytes a 32-lane half of a transposed read
// touches fall on 64 different banks (rows two apart would share them); even, so a chunk pair stays a pair
__device__ __forceinline__ int attn_vsw(int row) { return 2 * ((row >> 1) & 3); }

// 2^z for two z <= ~0 (clamped at -125), the attention softmax's exponential: k = rint(z) by the 1.5 * 2^23 addition, r = z - k, a
// degree-4 polynomial for 2^r on [-0.5, 0.5] (relative error 2.9e-6: the probabilities are rounded to bf16), 2^k added into the exponent
// field.  7 packed + 1 scalar instruction per pair; oracle/fav_exact.c: fav_attn_exp2_ref is the same sequence.
__device__ __forceinline__ f32x2_t fav_attn_exp2(f32x2_t z) {
    z.x = fmaxf(z.x, -125.0f); z.y = fmaxf(z.y, -125.0f);
    const f32x2_t M = {12582912.0f, 12582912.0f};
    const f32x2_t zk = z + M;
    const f32x2_t r = z - (zk - M);
    f32x2_t p = __builtin_elementwise_fma((f32x2_t){0x1.3a02c2p-7f, 0x1.3a02c2p-7f}, r, (f32x2_t){0x1.c9fc46p-5f, 0x1.c9fc46p-5f});
    p = __builtin_elementwise_fma(p, r, (f32x2_t){0x1.ec0378p-3f, 0x1.ec0378p-3f});
    p = __builtin_elementwise_fma(p, r, (f32x2_t){0x1.62e12cp-1f, 0x1.62e12cp-1f});
    p = __builtin_elementwise_fma(p, r, (f32x2_t){1.0f, 1.0f});
    f32x2_t e;
    e.x = __uint_as_float(__float_as_uint(p.x) + (__float_as_uint(zk.x) << 23));
    e.y = __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(zk.y) << 23));
    return e;
}

// NKT = key tiles the kernel is built for (T <= 16 NKT): 13 for ViT-B/16's 197 tokens, 12 score registers fewer than 16.
// FULL: T needs exactly NKT tiles, so every tile guard and the position of the masked tile are compile-time (the production shape).
template <int MODE, int NKT = 16, bool FULL = false>
__global__ __launch_bounds__(512, 4) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int T, int D,
                                                           int heads) {
    extern __shared__ __attribute__((aligned(16))) unsigned char asm_[];
    const int nkt = FULL ? NKT : (T + 15) >> 4;                    // key tiles of 16
    const int Tp2 = FULL ? (NKT + 1) / 2 * 32 : ((T + 31) >> 5) << 5;   // keys padded for the 32-deep second product
    unsigned char* const Ks = asm_;                                   // [nkt*16][128 B], chunk ^= row & 7
    unsigned char* const Vs = Ks + nkt * 16 * 128;                    // [Tp2][128 B], chunk ^= attn_vsw(row)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nwaves = nthr >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int h = blockIdx.x % heads;
    const long long f = blockIdx.x / heads;
    const uint16_t* base = qkv + f * (long long)T * 3 * D;
    const int nqt = nkt;
    // the Q fragments of this wave's first query tile: requested before anything else
    uint4 fqn[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    if (wave < nqt && wave * 16 + frow < T) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fqn[kk] = *(const uint4*)(base + (long long)(wave * 16 + frow) * 3 * D + h * 64 + kk * 32 + fq * 8);
    }
    // ---- stage K and V (zero beyond T): four (K, V) chunk pairs per thread in flight, then their LDS writes
    for (int c0 = 0; c0 < Tp2 * 8; c0 += 4 * nthr) {
        uint4 kv[4], vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = c0 + tid + r * nthr, row = i >> 3, ch = i & 7;
            kv[r] = make_uint4(0u, 0u, 0u, 0u);
            vv[r] = make_uint4(0u, 0u, 0u, 0u);
            if (row < T) {
                kv[r] = *(const uint4*)(base + (long long)row * 3 * D + D + h * 64 + ch * 8);
                vv[r] = *(const uint4*)(base + (long long)row * 3 * D + 2 * D + h * 64 + ch * 8);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = c0 + tid + r * nthr, row = i >> 3, ch = i & 7;
            if (row < nkt * 16) *(uint4*)(Ks + row * 128 + ((ch ^ (row & 7)) << 4)) = kv[r];
            if (row < Tp2) *(uint4*)(Vs + row * 128 + ((ch ^ attn_vsw(row)) << 4)) = vv[r];
        }
    }
    __syncthreads();

    // transposed reads of V: lane 4q + p of a 16-lane group addresses row (key) r0 + q, channels 4p .. 4p + 3 of the block
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int vsw_l = attn_vsw(4 * fq + tq);       // = attn_vsw of every row this lane addresses (rows 32*ks + 4*fq + tq (+ 16))
    const unsigned char* const vbase = Vs + (4 * fq + tq) * 128 + ((tp >> 1) << 4) + 8 * (tp & 1);

    for (int qt = wave; qt < nqt; qt += nwaves) {
        const int q = qt * 16 + frow;          // this lane's query (as MFMA column)
        const uint4 fqv[2] = {fqn[0], fqn[1]};
        {   // the next tile's Q fragments travel while this one is computed
            const int q2 = q + nwaves * 16;
            fqn[0] = make_uint4(0u, 0u, 0u, 0u);
            fqn[1] = make_uint4(0u, 0u, 0u, 0u);
            if (qt + nwaves < nqt && q2 < T) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fqn[kk] = *(const uint4*)(base + (long long)q2 * 3 * D + h * 64 + kk * 32 + fq * 8);
            }
        }
        f32x4_t sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            sc[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int row = kt * 16 + frow;
                    const uint4 fk = *(const uint4*)(Ks + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
                    if (MODE == 0) {
                        union { uint4 u; bf16x8_t v; } ua, ub;
                        ua.u = fk; ub.u = fqv[kk];
                        sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, sc[kt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t wa = ((const uint32_t*)&fk)[j >> 1], xa = ((const uint32_t*)&fqv[kk])[j >> 1];
                            const float wf = bf16_bits_to_f32((j & 1) ? (wa >> 16) : (wa & 0xFFFFu));
                            const float xf = bf16_bits_to_f32((j & 1) ? (xa >> 16) : (xa & 0xFFFFu));
                            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xf, sc[kt], 0, 0, 0);
                        }
                    }
                }
            }
            // straight-line code (FULL): without a fence the scheduler hoists every K fragment read to the top - 104 registers
            if (FULL && (kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // row max of the raw scores (only the last key tile reaches past T: its padded keys are -inf for the max and 0 afterwards)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            if (kt < nkt) {
                if (FULL ? kt == NKT - 1 : kt == nkt - 1) {
                    const int key = kt * 16 + fq * 4;
                    sc[kt][0] = key < T ? sc[kt][0] : -INFINITY;
                    sc[kt][1] = key + 1 < T ? sc[kt][1] : -INFINITY;
                    sc[kt][2] = key + 2 < T ? sc[kt][2] : -INFINITY;
                    sc[kt][3] = key + 3 < T ? sc[kt][3] : -INFINITY;
                }
                mx = fmaxf(fmaxf(mx, sc[kt][0]), sc[kt][1]);     // v_max3_f32
                mx = fmaxf(fmaxf(mx, sc[kt][2]), sc[kt][3]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // e = 2^((s - max) / 8 * log2 e) = 2^fma(s, c, -(max * c)), c = log2(e) / 8
        const float C2 = 0x1.715476p-3f;
        const float nmc = -__fmul_rn(mx, C2);
        const f32x2_t c2v = {C2, C2}, nmcv = {nmc, nmc};
        f32x2_t sum2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            if (kt < nkt) {
                f32x2_t lo = fav_attn_exp2(__builtin_elementwise_fma((f32x2_t){sc[kt][0], sc[kt][1]}, c2v, nmcv));
                f32x2_t hi = fav_attn_exp2(__builtin_elementwise_fma((f32x2_t){sc[kt][2], sc[kt][3]}, c2v, nmcv));
                if (FULL ? kt == NKT - 1 : kt == nkt - 1) {
                    const int key = kt * 16 + fq * 4;
                    lo.x = key < T ? lo.x : 0.f;
                    lo.y = key + 1 < T ? lo.y : 0.f;
                    hi.x = key + 2 < T ? hi.x : 0.f;
                    hi.y = key + 3 < T ? hi.y : 0.f;
                }
                sum2 = sum2 + lo;
                sum2 = sum2 + hi;
                sc[kt] = (f32x4_t){lo.x, lo.y, hi.x, hi.y};
            }
        float sum = __fadd_rn(sum2.x, sum2.y);
        sum = __fadd_rn(sum, __shfl_xor(sum, 16, 64));
        sum = __fadd_rn(sum, __shfl_xor(sum, 32, 64));
        const float inv_sum = __fdiv_rn(1.0f, sum);   // one IEEE division per query row, then multiplications
        const f32x2_t inv2 = {inv_sum, inv_sum};
        uint2 pk[NKT + 1];
#pragma unroll
        for (int kt = 0; kt <= NKT; ++kt) {
            pk[kt] = make_uint2(0u, 0u);
            if (kt < NKT && kt < nkt) {
                const f32x2_t lo = (f32x2_t){sc[kt][0], sc[kt][1]} * inv2, hi = (f32x2_t){sc[kt][2], sc[kt][3]} * inv2;
                pk[kt] = make_uint2(pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y));
            }
        }
        f32x4_t oc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < (NKT + 1) / 2; ++ks)
            if (ks < (Tp2 >> 5)) {
                // P^T fragment: query frow, k slots 8 fq .. + 7 = keys 32 ks + 4 fq .. + 3 and 32 ks + 16 + 4 fq .. + 3
                const uint4 fp = make_uint4(pk[2 * ks].x, pk[2 * ks].y, pk[2 * ks + 1].x, pk[2 * ks + 1].y);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    // V^T fragment: channel dt*16 + frow at the same keys = two transposed 4-key x 16-channel blocks, 16 rows apart
                    const unsigned char* va = vbase + ks * 4096 + (((2 * dt) ^ vsw_l) << 4);
                    const attn_v4s t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_v4s*)(va));
                    const attn_v4s t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_v4s*)(va + 2048));
                    const uint2 t0u = __builtin_bit_cast(uint2, t0), t1u = __builtin_bit_cast(uint2, t1);
                    const uint4 fv = make_uint4(t0u.x, t0u.y, t1u.x, t1u.y);
                    if (MODE == 0) {
                        union { uint4 u; bf16x8_t v; } ua, ub;
                        ua.u = fv; ub.u = fp;
                        oc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, oc[dt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t wa = ((const uint32_t*)&fv)[j >> 1], xa = ((const uint32_t*)&fp)[j >> 1];
                            const float wf = bf16_bits_to_f32((j & 1) ? (wa >> 16) : (wa & 0xFFFFu));
                            const float xf = bf16_bits_to_f32((j & 1) ? (xa >> 16) : (xa & 0xFFFFu));
                            oc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, xf, oc[dt], 0, 0, 0);
                        }
                    }
                }
                if (FULL && (ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        if (q < T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(uint2*)(out + (f * T + q) * (long long)D + h * 64 + dt * 16 + fq * 4) =
                    make_uint2(pack_bf16x2(oc[dt][0], oc[dt][1]), pack_bf16x2(oc[dt][2], oc[dt][3]));
        }
    }
}

// ---------------------------------------------------------------------------
// Confidence head: logits [T][n][ld] fp32 -> pbar = mean_t softmax(z_t * inv_temp),
// label = argmax (lowest index on ties), conf = max pbar or 1 - H(pbar)/ln C,
// fail = conf < tau, score = clamp(1 - conf).
// One block (4 waves) per frame; wave w takes samples t = w, w+4, ...; a lane
// owns classes {4*lane + 256*i + (0..3)} (coalesced float4 loads); all
// reductions are wavefront shuffles, waves meet once through LDS.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NV>  // float4 groups per lane: supports num_classes <= 256 * NV
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ logits, int T, int n, int C, int ld,
                                                   float inv_temp, int conf_kind, float tau, float inv_lnC,
                                                   int* __restrict__ labels, float* __restrict__ conf,
                                                   uint8_t* __restrict__ fail, float* __restrict__ score, int out_stride) {
    // out_stride: element pitch of labels / conf (1: two arrays; 2: one packed (label, confidence) record per frame,
    // labels = record base, conf = record base + 1 - the 8-byte unit the multi-GPU all-gather moves)
    __shared__ __attribute__((aligned(16))) float part[4][NV * 256];
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ float red_h[4];
    const int img = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float p[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) p[i][j] = 0.f;
    for (int t = wave; t < T; t += 4) {
        const float* row = logits + ((long long)t * n + img) * ld;
        float z[NV][4];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) z[i][j] = -INFINITY;
            if (c + 3 < C) {
                const float4 v = *(const float4*)(row + c);
                z[i][0] = v.x * inv_temp; z[i][1] = v.y * inv_temp; z[i][2] = v.z * inv_temp; z[i][3] = v.w * inv_temp;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < C) z[i][j] = row[c + j] * inv_temp;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, z[i][j]);
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[i][j] = expf(z[i][j] - mx);  // exp(-inf) = 0 for padded classes
                s += z[i][j];
            }
        s = wave_sum(s);
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[i][j] += z[i][j] * inv;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        *(float4*)&part[wave][4 * lane + 256 * i] = make_float4(p[i][0], p[i][1], p[i][2], p[i][3]);
    __syncthreads();
    // thread tid owns classes 4*tid + 1024*i'  (NV*256 floats per wave row = NV*64 float4)
    const float inv_T = 1.0f / (float)T;
    float best = -1.f;
    int besti = 0x7fffffff;
    float h = 0.f;
    for (int q = tid; q < NV * 64; q += 256) {
        const float4 a = *(const float4*)&part[0][4 * q], b = *(const float4*)&part[1][4 * q];
        const float4 c4 = *(const float4*)&part[2][4 * q], d = *(const float4*)&part[3][4 * q];
        const float pb[4] = {(((a.x + b.x) + c4.x) + d.x) * inv_T, (((a.y + b.y) + c4.y) + d.y) * inv_T,
                             (((a.z + b.z) + c4.z) + d.z) * inv_T, (((a.w + b.w) + c4.w) + d.w) * inv_T};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cls = 4 * q + j;
            if (cls < C) {
                if (pb[j] > best) { best = pb[j]; besti = cls; }
                if (pb[j] > 0.f) h -= pb[j] * logf(pb[j]);
            }
        }
    }
    // wave argmax (ties -> lowest index), wave entropy sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    h = wave_sum(h);
    if (lane == 0) { red_v[wave] = best; red_i[wave] = besti; red_h[wave] = h; }
    __syncthreads();
    if (tid == 0) {
        float bv = red_v[0]; int bi = red_i[0]; float hh = red_h[0];
        for (int w = 1; w < 4; ++w) {
            if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
            hh += red_h[w];
        }
        const float cf = conf_kind == 0 ? bv : 1.0f - hh * inv_lnC;
        labels[(long long)img * out_stride] = bi;
        conf[(long long)img * out_stride] = cf;
        if (fail) fail[img] = cf < tau ? 1 : 0;
        if (score) score[img] = fminf(fmaxf(1.0f - cf, 0.f), 1.f);
    }
}

}  // namespace fav

// ===========================================================================
// SignalAnalyzer.analyze_frame as ONE fused pass per frame (SURVEY.md §8f row 2;
// reference: platform/backend/signal_analyzer.py:62-112).  Per uint8 BGR frame:
//   gray     = (1868*B + 9617*G + 4899*R + 8192) >> 14            (cv2.COLOR_BGR2GRAY, 8-bit)
//   lap      = 4-neighbour Laplacian of gray, BORDER_REFLECT_101   (cv2.Laplacian ksize=1)
//   sums     = sum gray, sum |gray - prev_gray|, sum lap, sum lap^2, 256-bin histogram
// One block per frame keeps the whole gray plane in LDS (H*W <= 150 KB), so the frame
// is read from HBM exactly once (plus once more as the predecessor of the next frame:
// inside a batch the previous gray is recomputed from frame i-1 rather than exchanged
// between blocks).  All accumulations are integers, hence exact; the derived floats are
// computed from them at the end.
// ===========================================================================
namespace fav {

struct SignalStats {       // mirrors fav_signal_stats in include/fav.h
    double lap_var, mean, mean_diff;
    float entropy;
    int has_prev;
    long long sum_lap, sum_lap2;
    unsigned int sum_gray, sum_absdiff;
    unsigned int hist[256];
};

__device__ __forceinline__ uint32_t bgr_quad_to_gray(const uint32_t* p3) {
    // 4 pixels = 12 bytes = 3 dwords: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
    const uint32_t w0 = p3[0], w1 = p3[1], w2 = p3[2];
    const uint32_t b0 = w0 & 255, g0 = (w0 >> 8) & 255, r0 = (w0 >> 16) & 255, b1 = w0 >> 24;
    const uint32_t g1 = w1 & 255, r1 = (w1 >> 8) & 255, b2 = (w1 >> 16) & 255, g2 = w1 >> 24;
    const uint32_t r2 = w2 & 255, b3 = (w2 >> 8) & 255, g3 = (w2 >> 16) & 255, r3 = w2 >> 24;
    const uint32_t y0 = (1868u * b0 + 9617u * g0 + 4899u * r0 + 8192u) >> 14;
    const uint32_t y1 = (1868u * b1 + 9617u * g1 + 4899u * r1 + 8192u) >> 14;
    const uint32_t y2 = (1868u * b2 + 9617u * g2 + 4899u * r2 + 8192u) >> 14;
    const uint32_t y3 = (1868u * b3 + 9617u * g3 + 4899u * r3 + 8192u) >> 14;
    return y0 | (y1 << 8) | (y2 << 16) | (y3 << 24);
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void signal_stats_kernel(const uint8_t* __restrict__ frames, int n, int H, int W,
                                                           const uint8_t* __restrict__ prev_gray,
                                                           uint8_t* __restrict__ last_gray, SignalStats* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int npx = H * W, nq = npx >> 2;             // W % 4 == 0
    uint32_t* gq = (uint32_t*)dyn;                    // gray plane, 4 pixels per dword
    unsigned int* hist = (unsigned int*)(dyn + ((npx + 15) & ~15));   // [4][256] per-wave histograms
    long long* red = (long long*)(hist + 1024);       // [4][4] wave partials
    const int f = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t* cur = (const uint32_t*)(frames + (size_t)f * npx * 3);
    const bool has_prev = f > 0 || prev_gray != nullptr;
    const uint32_t* prv_bgr = f > 0 ? (const uint32_t*)(frames + (size_t)(f - 1) * npx * 3) : nullptr;
    const uint32_t* prv_gray = (const uint32_t*)prev_gray;
    for (int i = tid; i < 1024; i += 256) hist[i] = 0;
    __syncthreads();
    unsigned int s_gray = 0, s_diff = 0;
    for (int q = tid; q < nq; q += 256) {
        const uint32_t g = bgr_quad_to_gray(cur + 3 * q);
        gq[q] = g;
        uint32_t pg = 0;
        if (f > 0) pg = bgr_quad_to_gray(prv_bgr + 3 * q);
        else if (prv_gray) pg = prv_gray[q];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int y = (g >> (8 * b)) & 255, py = (pg >> (8 * b)) & 255;
            s_gray += y;
            s_diff += has_prev ? (unsigned)abs(y - py) : 0u;
            atomicAdd(&hist[wave * 256 + y], 1u);
        }
        if (last_gray && f == n - 1) ((uint32_t*)last_gray)[q] = g;
    }
    __syncthreads();
    // Laplacian from the LDS plane, reflect-101 borders: index -1 -> 1, H -> H-2
    const uint8_t* g8 = (const uint8_t*)gq;
    long long s_lap = 0, s_lap2 = 0;
    for (int px = tid; px < npx; px += 256) {
        const int y = px / W, x = px - y * W;
        const int yu = y == 0 ? 1 : y - 1, yd = y == H - 1 ? H - 2 : y + 1;
        const int xl = x == 0 ? 1 : x - 1, xr = x == W - 1 ? W - 2 : x + 1;
        const int lap = (int)g8[yu * W + x] + (int)g8[yd * W + x] + (int)g8[y * W + xl] + (int)g8[y * W + xr] - 4 * (int)g8[px];
        s_lap += lap;
        s_lap2 += (long long)lap * lap;
    }
    const long long r0 = wave_sum_i64(s_gray), r1 = wave_sum_i64(s_diff), r2 = wave_sum_i64(s_lap), r3 = wave_sum_i64(s_lap2);
    if (lane == 0) { red[wave * 4] = r0; red[wave * 4 + 1] = r1; red[wave * 4 + 2] = r2; red[wave * 4 + 3] = r3; }
    __syncthreads();
    // merge histograms; one bin per thread; entropy in fp32 (as the fp32 cv2.calcHist output is used)
    const unsigned int cnt = hist[tid] + hist[256 + tid] + hist[512 + tid] + hist[768 + tid];
    const float p = (float)cnt / (float)npx;
    float term = cnt ? -p * log2f(p) : 0.f;
    term = wave_sum(term);
    __shared__ float ent_part[4];
    if (lane == 0) ent_part[wave] = term;
    SignalStats* o = out + f;
    o->hist[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        const long long sg = red[0] + red[4] + red[8] + red[12], sd = red[1] + red[5] + red[9] + red[13];
        const long long sl = red[2] + red[6] + red[10] + red[14], sl2 = red[3] + red[7] + red[11] + red[15];
        const double N = (double)npx, m = (double)sl / N;
        o->lap_var = (double)sl2 / N - m * m;
        o->mean = (double)sg / N;
        o->mean_diff = has_prev ? (double)sd / N : 0.0;
        o->entropy = ((ent_part[0] + ent_part[1]) + ent_part[2]) + ent_part[3];
        o->has_prev = has_prev ? 1 : 0;
        o->sum_lap = sl; o->sum_lap2 = sl2;
        o->sum_gray = (unsigned int)sg; o->sum_absdiff = (unsigned int)sd;
    }
}

}  // namespace fav

// ===========================================================================
// On-device corruption generator.  The reference corrupts frames in
// the browser with Math.random (platform/frontend/js/app.js:782-857; modes
// platform/backend/vision_simulator.py:15); here the same four modes plus ImageNet-C style
// Gaussian noise run on the device with a counter-based generator, so a corrupted test set is
// a pure function of (seed, global frame index) and needs no host round trip:
//   normal    : p' = clamp(rne(p * gain + n)),  n = (u - 0.5) * 255 * level, ONE u per pixel   (app.js:789-798)
//   blank     : (2, 2, 4)                                                                      (app.js:820-822)
//   corrupted : with probability 0.2 a pixel becomes (u1*255, 0, u2*255); six horizontal
//               bars of height 2..14 blended towards (255, 0, 170) with alpha 0.4..0.9        (app.js:835-850)
//   gaussian  : fp32 out = clamp(p/255 + sigma * N(0,1)), Box-Muller on Philox uniforms
// Philox4x32-10, key = seed, counter = (pixel or pair index, global frame, stream, 0).
// ===========================================================================
namespace fav {

struct CorruptParams {
    int mode;            // 0 normal, 1 blank, 2 corrupted, 3 gaussian (fp32 out)
    float level, gain;   // uniform-noise level [0,1]; brightness gain
    float sigma;         // gaussian sigma on [0,1] pixels
    uint32_t seed_lo, seed_hi;
    long long first_index;
};

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // 24-bit, exact
__device__ __forceinline__ uint32_t clamp_u8_rne(float v) { return (uint32_t)__float2int_rn(fminf(fmaxf(v, 0.f), 255.f)); }

__global__ __launch_bounds__(256) void corrupt_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, int n, int H,
                                                      int W, CorruptParams cp) {
    const long long npx = (long long)H * W, total = npx * n;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long f = idx / npx;
        const uint32_t px = (uint32_t)(idx - f * npx);
        const uint32_t frame = (uint32_t)(cp.first_index + f);
        const uint8_t* p = in + idx * 3;
        const float c0 = p[0], c1 = p[1], c2 = p[2];
        if (cp.mode == 3) {
            float* o = (float*)out + idx * 3;
            const uint4 a = philox4x32_10(make_uint4(px, frame, 3u, 0u), cp.seed_lo, cp.seed_hi);
            const uint4 b = philox4x32_10(make_uint4(px, frame, 4u, 0u), cp.seed_lo, cp.seed_hi);
            // Box-Muller: radius from (1 - u) in (0,1], angle 2*pi*u'
            const float r0 = sqrtf(-2.0f * logf(1.0f - u01(a.x))), r1 = sqrtf(-2.0f * logf(1.0f - u01(a.z)));
            const float t0 = 6.2831853071795864f * u01(a.y), t1 = 6.2831853071795864f * u01(a.w);
            const float n0 = r0 * cosf(t0), n1 = r0 * sinf(t0), n2 = r1 * cosf(t1);
            (void)b;
            o[0] = fminf(fmaxf(c0 * (1.0f / 255.0f) + cp.sigma * n0, 0.f), 1.f);
            o[1] = fminf(fmaxf(c1 * (1.0f / 255.0f) + cp.sigma * n1, 0.f), 1.f);
            o[2] = fminf(fmaxf(c2 * (1.0f / 255.0f) + cp.sigma * n2, 0.f), 1.f);
            continue;
        }
        uint8_t* o = (uint8_t*)out + idx * 3;
        if (cp.mode == 1) { o[0] = 2; o[1] = 2; o[2] = 4; continue; }
        const uint4 u = philox4x32_10(make_uint4(px, frame, (uint32_t)cp.mode, 0u), cp.seed_lo, cp.seed_hi);
        float v0 = c0 * cp.gain, v1 = c1 * cp.gain, v2 = c2 * cp.gain;
        if (cp.mode == 0) {
            const float nz = (u01(u.x) - 0.5f) * 255.0f * cp.level;
            v0 += nz; v1 += nz; v2 += nz;
        } else {
            if (u01(u.x) > 0.8f) { v0 = u01(u.y) * 255.0f; v1 = 0.f; v2 = u01(u.z) * 255.0f; }
            const int y = (int)(px / (uint32_t)W);
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const uint4 q = philox4x32_10(make_uint4((uint32_t)b, frame, 7u, 0u), cp.seed_lo, cp.seed_hi);
                const float by = u01(q.x) * (float)H, bh = 2.0f + u01(q.y) * 12.0f, al = 0.4f + u01(q.z) * 0.5f;
                if ((float)y >= floorf(by) && (float)y < floorf(by) + floorf(bh)) {
                    v0 = v0 + (255.0f - v0) * al; v1 = v1 + (0.0f - v1) * al; v2 = v2 + (170.0f - v2) * al;
                }
            }
        }
        o[0] = (uint8_t)clamp_u8_rne(v0); o[1] = (uint8_t)clamp_u8_rne(v1); o[2] = (uint8_t)clamp_u8_rne(v2);
    }
}

}  // namespace fav
