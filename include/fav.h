/* fav.h — C ABI of the MI355X-native failure-aware classification path.
 *
 * Drop-in boundary.  The reference has NO plugin / FFI API for
 * this path: its seam is a plain Python method call whose result goes straight
 * into the trust engine,
 *
 *     last_analysis = analyzer.analyze_frame(frame)         platform/backend/main.py:160
 *     anomaly_score = anomaly.compute_anomaly(...)          platform/backend/main.py:141-143,347
 *     state = engine.update(vision_status, anomaly_score, dt)   main.py:145,168,348
 *
 * so the entry points below are what a binding for that seam needs: create a
 * per-connection scorer (main.py:110-118 constructs one per WebSocket), load a
 * checkpoint, classify a batch of frames into (label, confidence), derive the
 * failure flag / anomaly score the engine consumes, destroy on disconnect
 * (main.py:310-317).  Conventions follow the reference's own: status codes and
 * sentinels instead of exceptions (video_source.py:76-78,117; main.py:233-236),
 * single caller per handle (one scorer per connection, main.py:117), caller
 * owns every buffer it passes (video_source.py:114-117 hands out copies).
 *
 * Plain C types only; nothing of torch or HIP appears in a signature (a HIP
 * stream travels as void*).  All *_dev pointers are device (HBM) addresses on
 * the handle's device.
 */
#ifndef FAV_H
#define FAV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAV_ABI_VERSION 2

typedef struct fav_handle fav_handle;

typedef enum fav_status {
    FAV_OK = 0,
    FAV_ERR_INVALID_ARG = 1,
    FAV_ERR_BAD_BLOB = 2,
    FAV_ERR_NO_WEIGHTS = 3,
    FAV_ERR_HIP = 4,        /* a HIP runtime call failed; see fav_last_error() */
    FAV_ERR_NO_DEVICE = 5,  /* no usable gfx950 device: the path has no CPU fallback */
    FAV_ERR_UNSUPPORTED = 6
} fav_status;

/* Frame layout handed to fav_classify.  NHWC_U8 is the reference's frame
 * format (np.uint8 HxWx3, signal_analyzer.py:47-58; video_source.py:144-148);
 * NHWC_F32 carries [0,1] pixels (corrupted frames that are not 8-bit). */
typedef enum fav_layout { FAV_LAYOUT_NHWC_U8 = 0, FAV_LAYOUT_NHWC_F32 = 1 } fav_layout;
/* FAV_ARCH_VIT_B16: ViT-B/16 (BASELINE configs[4]: attention path + temperature-scaled entropy; single pass,
 * no dropout sites, no ensemble; input a multiple of 16 with at most 256 tokens).  FAV_ARCH_VIT_TINY: a
 * two-layer, 128-wide miniature of it for the parity tests. */
typedef enum fav_arch { FAV_ARCH_RESNET18_CIFAR = 0, FAV_ARCH_RESNET50 = 1, FAV_ARCH_VIT_B16 = 2, FAV_ARCH_VIT_TINY = 3 } fav_arch;
typedef enum fav_conf_kind { FAV_CONF_MAX_SOFTMAX = 0, FAV_CONF_ENTROPY = 1 } fav_conf_kind;
/* FAV_MATH_BF16: bf16 MFMA, fp32 accumulate (production).
 * FAV_MATH_F32_EXACT: same bf16 operands fed to the fp32-input MFMA, whose
 * result is a k-ordered fmaf chain; bit-reproducible against oracle/ (validation). */
typedef enum fav_math_mode { FAV_MATH_BF16 = 0, FAV_MATH_F32_EXACT = 1 } fav_math_mode;

typedef struct fav_config {
    uint32_t struct_size;   /* sizeof(fav_config), for ABI growth */
    int32_t device;         /* HIP device ordinal */
    int32_t arch;           /* fav_arch */
    int32_t num_classes;
    int32_t in_h, in_w;     /* frame size */
    int32_t max_batch;      /* largest n passed to fav_classify */
    float mean[3];          /* per-channel normalisation on [0,1] pixels */
    float stdev[3];
    int32_t n_samples;      /* MC-Dropout T; 1 = single deterministic pass */
    uint32_t site_mask;     /* bit s<n_blocks: output of residual block s; bit n_blocks: pooled features */
    float dropout_p;
    uint64_t seed;          /* Philox key */
    float temperature;      /* softmax(z / temperature) */
    int32_t conf_kind;      /* fav_conf_kind */
    float tau;              /* failure threshold: fail = conf < tau */
    int32_t math_mode;      /* fav_math_mode */
    int32_t chunk_a;        /* frames per pass through the high-resolution stages (0 = auto) */
    int32_t chunk_b;        /* frames per pass through the low-resolution stages (0 = auto) */
    int32_t regroup_block;  /* first residual block of the low-resolution group (-1 = auto) */
    int32_t n_members;      /* deep ensemble: independently trained checkpoints whose softmax is averaged
                               (1 = single model; > 1 excludes MC-Dropout) */
    /* Schedule choices (ABI 2).  0 = this build's measured default; they never change a result, only which launches compute it
     * (the library reads no environment variable). */
    int32_t tail_min_rows;  /* layers 3-4 run their one-block-per-CU fused tails when the planned launch (max_batch x T x H x W
                               rows, x members when grouped) has at least this many rows; 0 = 131 072; -1 = at any size */
    int32_t ens_grouped_max;/* deep ensemble: calls of up to this many frames run every op as ONE launch over all members;
                               0 = every call; -1 = never (one stream per member) */
    int32_t vit_streams;    /* ViT: parts of a call's batch run side by side on as many streams, 1 .. 4; 0 = 2 */
    int32_t stem_fused;     /* ImageNet stem as one launch (normalise + 7x7/2 + ReLU + max pool); 0 = yes; -1 = three launches */
} fav_config;

/* Fills *cfg with the defaults (ImageNet mean/std, T=1, no dropout, tau=0.5). */
void fav_default_config(fav_config* cfg, int32_t arch);

/* Lifecycle: create -> load_weights -> classify xN -> destroy
 * (reference lifecycle: construct on accept main.py:110-118, reset main.py:284-291,
 * drop on disconnect main.py:310-317). */
fav_status fav_create(const fav_config* cfg, fav_handle** out);
fav_status fav_load_weights(fav_handle* h, const void* blob_host, size_t size);   /* member 0 */
fav_status fav_load_member_weights(fav_handle* h, int32_t member, const void* blob_host, size_t size);
void fav_destroy(fav_handle* h);
/* Structural validation of a checkpoint blob (magic, version, layer table, every data range and its
 * alignment), without a device or a handle; fav_load_*_weights runs it first.  err (may be NULL)
 * receives a message.  The blob is file-supplied, hence untrusted. */
fav_status fav_check_blob(const void* blob_host, size_t size, char* err, size_t err_cap);
/* The static launch schedule of a configuration (ResNet archs) as text, one line per op; needs no device.
 * flags bit 0: the layer-by-layer schedule (no fused bottleneck tails).  A test / inspection hook. */
fav_status fav_plan_schedule(const fav_config* cfg, int32_t flags, char* out, size_t cap);
const char* fav_last_error(const fav_handle* h); /* h may be NULL: error of the last failed fav_create */
int32_t fav_abi_version(void);

/* The hot path: n frames -> labels[n] (int32), conf[n] (fp32), both device
 * pointers.  Asynchronous on `hip_stream` (NULL = the default stream); results
 * are complete once the stream is synchronised.  Replaces the scorer call at
 * main.py:160 / main.py:141. */
fav_status fav_classify(fav_handle* h, const void* images_dev, int32_t n, int32_t layout,
                        int32_t* labels_dev, float* conf_dev, void* hip_stream);

/* Same, plus: first_image_index (global index of frame 0 of this call — dropout
 * masks are keyed by global frame index so a sharded batch reproduces the
 * unsharded result), fail_dev[n] (uint8: conf < tau) and
 * score_dev[n] (fp32 anomaly_score = clamp(1 - conf, 0, 1), the value fed to
 * TrustEngine.update, trust_engine.py:139).  fail_dev / score_dev may be NULL. */
fav_status fav_classify_ex(fav_handle* h, const void* images_dev, int32_t n, int32_t layout,
                           int64_t first_image_index, int32_t* labels_dev, float* conf_dev,
                           uint8_t* fail_dev, float* score_dev, void* hip_stream);

/* Same as fav_classify_ex with the two result arrays packed: records_dev[n] holds one 8-byte record per frame,
 * { int32 label, fp32 confidence } (8-byte aligned device pointer).  This is the unit the multi-GPU path exchanges
 * (one all-gather of 8 B per frame): the confidence head writes each rank's records straight
 * into its slot of the all-gather send buffer, no pack / concatenate launches in between.  The reference is a
 * single process and has no counterpart ("Distributed communication backend: None"). */
fav_status fav_classify_records(fav_handle* h, const void* images_dev, int32_t n, int32_t layout,
                                int64_t first_image_index, void* records_dev, uint8_t* fail_dev,
                                float* score_dev, void* hip_stream);

/* Host-buffer convenience (frames and results in host memory; synchronous). */
fav_status fav_classify_host(fav_handle* h, const void* images_host, int32_t n, int32_t layout,
                             int64_t first_image_index, int32_t* labels_host, float* conf_host,
                             uint8_t* fail_host, float* score_host);

/* Logits of the last fav_classify call, copied to logits_dev as fp32
 * [T][n][num_classes] (T = 1 without dropout).  Parity tests use it. */
fav_status fav_get_logits(fav_handle* h, float* logits_dev, int32_t* t_out, int32_t* n_out, void* hip_stream);

/* Per-kernel-class timing with HIP events on the launch stream (bench.py's
 * roofline leg).  Classes: see fav_kernel_class. */
typedef enum fav_kernel_class {
    FAV_K_STEM = 0, FAV_K_CONV = 1, FAV_K_MAXPOOL = 2, FAV_K_AVGPOOL = 3,
    FAV_K_DROPOUT = 4, FAV_K_HEAD = 5, FAV_K_COUNT = 6
} fav_kernel_class;
typedef struct fav_profile {
    double ms[FAV_K_COUNT];        /* summed event-measured duration per class */
    double flops[FAV_K_COUNT];     /* algorithmic FLOPs launched */
    double bytes[FAV_K_COUNT];     /* algorithmic HBM bytes launched */
    int64_t launches[FAV_K_COUNT];
} fav_profile;
/* One row per op of the static schedule (valid after fav_get_profile). */
typedef struct fav_op_profile {
    int32_t op_index, kind;          /* kind: 0 stem im2col, 1 conv/fc, 2 maxpool, 3 avgpool, 4 entry dropout, 5 fused bottleneck tail, 6 entry dropout + reduce, 7 fused stem (conv + max pool) */
    int32_t H, W, Cin, Ho, Wo, Cout, kh, kw, stride;
    int32_t reserved;
    double ms, flops, bytes;
    int64_t launches;
} fav_op_profile;
fav_status fav_set_profiling(fav_handle* h, int32_t enable);
fav_status fav_get_op_profile(fav_handle* h, fav_op_profile* out, int32_t cap, int32_t* n_out);
fav_status fav_get_profile(fav_handle* h, fav_profile* out, int32_t reset); /* synchronises the device */

/* ---- Operator level (one launch each; used by the executor and by the
 * per-kernel parity tests).  All tensors NHWC, bf16 unless noted. ---- */
typedef struct fav_dropout_desc {
    int32_t site;            /* -1 = no dropout */
    uint32_t threshold;      /* drop iff 8-bit draw < threshold (= round(p * 256)) */
    float scale;             /* 1 / (1 - threshold/256) */
    uint64_t seed;
    int64_t v0;              /* virtual frame index of row 0: v = t * n_img + i */
    int32_t n_img;           /* frames per sample */
    int64_t first_image_index;
} fav_dropout_desc;

typedef struct fav_conv_desc {
    const void* x;           /* [n_frames][H][W][Cin] bf16, Cin % 64 == 0 */
    const void* w;           /* [Cout][kh][kw][Cin] bf16, Cout % 64 == 0 */
    const float* bias;       /* [Cout] */
    const void* res;         /* optional [n_frames][Ho][Wo][Cout] bf16 */
    void* y;                 /* [n_frames][Ho][Wo][Cout] bf16, or fp32 if out_f32 */
    int32_t n_frames, H, W, Cin, Cout, kh, kw, stride, pad;
    int32_t relu, out_f32, math_mode;   /* relu: 0 = none, 1 = ReLU, 2 = GELU (ViT MLP) */
    fav_dropout_desc drop;
} fav_conv_desc;
fav_status fav_op_conv2d(const fav_conv_desc* d, void* hip_stream);

/* Bottleneck tail (one launch): conv_b 3x3/1/1 Cmid->Cmid + ReLU (skipped when wb == NULL: x is then conv_c's
 * input), conv_c 1x1 Cmid->4*Cmid + bias + residual + ReLU + dropout site -> y, and the NEXT block's conv_a
 * 1x1 4*Cmid->Nred + ReLU -> t1n (skipped when wa == NULL).  Same arithmetic, k order and rounding points as the
 * three fav_op_conv2d launches it replaces (bit-identical results); production math mode only.
 * Cmid in {64, 128}; Nred in {0, Cmid, 128}. */
typedef struct fav_tail_desc {
    const void* x;                          /* [n][H][W][Cmid] bf16 */
    const void* wb; const float* bias_b;    /* [Cmid][3][3][Cmid] bf16, [Cmid] */
    const void* wc; const float* bias_c;    /* [4*Cmid][Cmid] bf16, [4*Cmid] */
    const void* res;                        /* [n][H][W][4*Cmid] bf16 */
    void* y;                                /* [n][H][W][4*Cmid] bf16 */
    const void* wa; const float* bias_a;    /* [Nred][4*Cmid] bf16, [Nred] */
    void* t1n;                              /* [n][H][W][Nred] bf16 */
    int32_t n_frames, H, W, Cmid, Nred;
    fav_dropout_desc drop;
    /* res_entry != 0 (Cmid = Nred = 64 with the 3x3, drop.site >= 0): `res` is the CACHED prefix output
     * [drop.n_img][H][W][4*Cmid] and the residual of virtual frame v is dropout_{entry_site}(res[v % n_img]) - what
     * fav_op_entry_dropout / fav_op_entry_reduce would have stored for it (same bits), computed in the epilogue instead */
    int32_t res_entry, entry_site;
} fav_tail_desc;
fav_status fav_op_bottleneck_tail(const fav_tail_desc* d, void* hip_stream);
/* frames (u8 or fp32 NHWC3) -> normalised bf16 im2col matrix [n*Ho*Wo][kpad] */
fav_status fav_op_stem_im2col(const void* images, int32_t layout, int32_t n, int32_t H, int32_t W,
                              int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t kpad,
                              const float* mean3, const float* inv_std3, void* out, void* hip_stream);
/* the ImageNet stem in one launch: frames (u8 or fp32 NHWC3) -> normalise -> 7x7/2 conv to 64 channels (w [64][192] bf16,
 * k = (r*7 + s)*3 + c, zero for k >= 147) -> + bias -> ReLU -> bf16 -> 3x3/2 max pool -> out [n][Hp][Wp][64] bf16;
 * bit-identical to fav_op_stem_im2col + fav_op_conv2d + fav_op_maxpool3x3s2 */
fav_status fav_op_stem_pool(const void* images, int32_t layout, int32_t n, int32_t H, int32_t W, const void* w, const float* bias,
                            const float* mean3, const float* inv_std3, void* out, void* hip_stream);
fav_status fav_op_maxpool3x3s2(const void* x, void* y, int32_t n, int32_t H, int32_t W, int32_t C, void* hip_stream);
fav_status fav_op_avgpool(const void* x, void* y, int32_t n, int32_t HW, int32_t C,
                          const fav_dropout_desc* drop, void* hip_stream);
/* out[v - v0][e] = dropout(x[v % n_img][e]) for v in [v0, v0 + n_out) */
fav_status fav_op_entry_dropout(const void* x, void* out, int64_t elems_per_frame, int32_t n_out,
                                const fav_dropout_desc* drop, void* hip_stream);
/* entry dropout and the 1x1 reduce behind it in one launch (C = 256, Nred = 64):
 * y[v - v0] = dropout(x[v % n_img]) as above, t1[v - v0] = bf16(relu(conv1x1(y[v - v0], wa) + bias_a)); x [n_img][HW][C],
 * wa [Nred][C] bf16.  The executor's fusion of the MC-Dropout suffix's first two launches (no reference counterpart).
 * y may be NULL: the dropped copies are then not stored (a tail with res_entry recomputes them). */
fav_status fav_op_entry_reduce(const void* x, void* y, const void* wa, const float* bias_a, void* t1, int32_t C, int32_t Nred,
                               int32_t HW, int32_t n_out, const fav_dropout_desc* drop, void* hip_stream);
/* logits fp32 [T][n][ld] -> labels, conf (and fail/score if non-NULL) */
fav_status fav_op_head(const float* logits, int32_t T, int32_t n, int32_t num_classes, int32_t ld,
                       float temperature, int32_t conf_kind, float tau,
                       int32_t* labels, float* conf, uint8_t* fail, float* score, void* hip_stream);

/* ---- ViT building blocks (BASELINE configs[4]); linear layers go through fav_op_conv2d with kh = kw = 1.
 * LayerNorm over rows of D bf16 values (row r at x + r*ldx elements; D % 4 == 0, D <= 1024), fp32 statistics,
 * y[rows][D] bf16. */
fav_status fav_op_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t rows,
                            int32_t D, float eps, void* hip_stream);
/* Multi-head attention with 64-wide heads: qkv [n][T][3D] bf16 (Q | K | V) -> out [n][T][D] bf16,
 * softmax(Q K^T / 8) V per head, T <= 256, D = 64 * heads. */
fav_status fav_op_attention(const void* qkv, void* out, int32_t n, int32_t T, int32_t D, int32_t heads,
                            int32_t math_mode, void* hip_stream);
/* One linear layer of the encoder as the chained stream-K GEMM (gemm_streamk_kernel): y[rows][N] = act((x[rows][K] w[N][K]^T + bias) + res),
 * bf16 in and out, fp32 bias; res may be NULL or y itself (in-place residual); act 0 none, 1 ReLU, 2 GELU.  Bit-identical to
 * fav_op_conv2d with kh = kw = 1 on the same operands (the K steps are dealt out evenly over a persistent grid and a tile's partial
 * accumulator is handed on, never re-associated).  K % 32 == 0, N % 128 == 0, at least 256 tiles of 128 x 128; production math only.
 * No reference counterpart (the slot is platform/backend/main.py:160). */
typedef struct fav_linear_desc {
    const void* x; const void* w; const float* bias; const void* res; void* y;
    int64_t rows;
    int32_t K, N, act;
} fav_linear_desc;
fav_status fav_op_linear_streamk(const fav_linear_desc* d, void* hip_stream);
/* Token assembly: x[f][0] = pos[0], x[f][1 + p] = bf16(emb[f][p] + pos[1 + p]); emb [n][ntok-1][D] bf16, pos fp32. */
fav_status fav_op_vit_assemble(const void* emb, const float* pos, void* x, int32_t n, int32_t ntok, int32_t D,
                               void* hip_stream);

/* ---- SignalAnalyzer.analyze_frame as one fused pass per frame (SURVEY.md §8f row 2;
 * reference platform/backend/signal_analyzer.py:62-112): cv2.COLOR_BGR2GRAY, cv2.Laplacian
 * (ksize 1, reflect-101) variance, mean brightness, mean |gray - previous gray|, 256-bin
 * histogram entropy.  frames_bgr: [n][H][W][3] uint8 on the device, consecutive frames of
 * one stream; prev_gray: gray plane preceding frame 0 (NULL = none); last_gray_out: receives
 * the gray plane of frame n-1 (NULL = not wanted).  W % 4 == 0, H*W <= 150000. */
typedef struct fav_signal_stats {
    double lap_var, mean, mean_diff;
    float entropy;
    int32_t has_prev;
    int64_t sum_lap, sum_lap2;         /* exact integer sums the doubles are derived from */
    uint32_t sum_gray, sum_absdiff;
    uint32_t hist[256];
} fav_signal_stats;
fav_status fav_op_signal_stats(const uint8_t* frames_bgr, int32_t n, int32_t H, int32_t W, const uint8_t* prev_gray,
                               uint8_t* last_gray_out, fav_signal_stats* stats_dev, void* hip_stream);

/* ---- On-device corruption generator: the reference's four vision modes
 * (platform/backend/vision_simulator.py:15, painted at platform/frontend/js/app.js:782-857) and
 * ImageNet-C style Gaussian noise, as a pure function of (seed, global frame index).
 * frames: uint8 [n][H][W][3] on the device.  out: uint8 [n][H][W][3] for modes 0..2, fp32
 * [n][H][W][3] in [0,1] for FAV_CORRUPT_GAUSSIAN. */
typedef enum fav_corrupt_mode { FAV_CORRUPT_NORMAL = 0, FAV_CORRUPT_BLANK = 1, FAV_CORRUPT_GLITCH = 2,
                                FAV_CORRUPT_GAUSSIAN = 3 } fav_corrupt_mode;
fav_status fav_op_corrupt(const uint8_t* frames, void* out, int32_t n, int32_t H, int32_t W, int32_t mode,
                          float noise_level, float brightness_gain, float gaussian_sigma, uint64_t seed,
                          int64_t first_frame_index, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FAV_H */
