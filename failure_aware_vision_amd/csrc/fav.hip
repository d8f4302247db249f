// fav.hip — executor and C ABI (include/fav.h) of the MI355X-native
// failure-aware classification path.
//
// Host side of the hot path: a static schedule of kernel launches on ONE HIP
// stream (handle is single-caller, like the reference's per-connection scorer,
// platform/backend/main.py:110-118), a workspace arena allocated once, and
// Infinity-Cache-sized passes: frames go through the high-resolution stages in
// chunks small enough that producer->consumer activations stay in the 256 MiB
// L3, then through the low-resolution stages in larger chunks that fill the
// 256 CUs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fav.h"
#include "fav_kernels.hpp"

namespace {

thread_local std::string g_create_error;

std::string fmt(const char* f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

// inside a fork/join region: remember the first failure but keep going, so that every forked stream is joined
#define HIP_KEEP(h, st, expr)                                                                    \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess && (st) == FAV_OK) {                                                \
            (h)->err = fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (st) = FAV_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FAV_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

struct ArchDef {
    bool bottleneck;
    int depths[4];
    int planes[4];
    bool imagenet_stem;
};
const ArchDef kArch[2] = {
    {false, {2, 2, 2, 2}, {64, 128, 256, 512}, false},
    {true, {3, 4, 6, 3}, {64, 128, 256, 512}, true},
};

struct VitDef { int dim, depth, heads, mlp, patch; };
// arch 2 = ViT-B/16 (BASELINE configs[4]); arch 3 = a two-layer miniature of it for the parity tests
const VitDef kVit[2] = {{768, 12, 12, 3072, 16}, {128, 2, 2, 256, 16}};
inline bool is_vit_arch(int arch) { return arch == 2 || arch == 3; }

struct Layer {  // one convolution / fc
    int cout, cin, kh, kw, stride, pad;
    int cout_pad, k;         // device layout: w[cout_pad][k], bias[cout_pad]
    uint16_t* w = nullptr;          // weights / bias of the member being run
    float* b = nullptr;
    std::vector<uint16_t*> w_m;     // per ensemble member: w_slab + member * w_stride (one allocation, so that a grouped
    std::vector<float*> b_m;        // launch reaches member g's weights at a constant stride)
    void *w_slab = nullptr, *b_slab = nullptr;
    size_t w_stride = 0, b_stride = 0;
};

enum OpKind { OP_STEM_IM2COL, OP_CONV, OP_MAXPOOL, OP_AVGPOOL, OP_ENTRY_DROPOUT, OP_TAIL, OP_ENTRY_REDUCE, OP_STEM_POOL };
enum BufId { B_INPUT = -1, B_PHASE_IN = -2, B_PHASE_OUT = -3, B_NONE = -4, B_A1 = 5 };  // 0..4 rotating

struct Op {
    OpKind kind;
    int layer = -1;          // conv layer index (OP_TAIL: the 3x3, or -1 when the tail starts at the expanding 1x1)
    int layer_c = -1, layer_a = -1;   // OP_TAIL: the expanding 1x1 and the NEXT block's reducing 1x1 (-1: none)
    int in = B_NONE, out = B_NONE, res = B_NONE;
    int out2 = B_NONE;       // OP_TAIL: the next block's conv1 output
    int Co2 = 0;
    int H = 0, W = 0, C = 0;           // input dims per frame
    int Ho = 0, Wo = 0, Co = 0;        // output dims per frame
    int relu = 0, out_f32 = 0;
    int site = -1;                     // dropout site fused into this op
    int res_entry = 0;                 // OP_TAIL: the residual is the cached prefix output under the entry dropout (RESE)
    int skip_y = 0;                    // OP_ENTRY_REDUCE: the dropped copies are not stored (the next tail recomputes them)
    long long in_elems = 0, out_elems = 0;  // per frame
};

struct Phase {
    int op_begin, op_end;
    bool suffix;             // operates on virtual frames (t, i)
    long long in_elems, out_elems;   // per (virtual) frame
    int out_bytes_per_elem;
    bool low_res;            // belongs to the low-resolution group (chunk_b)
    int chunk;
};

}  // namespace

struct fav_handle {
    fav_config cfg;
    std::string err;
    int nblocks = 0;
    std::vector<Layer> layers;
    std::vector<Op> ops;
    std::vector<Phase> phases;
    bool weights_loaded = false;
    std::vector<char> member_loaded;   // deep ensemble (BASELINE configs[3]): one checkpoint per member
    int n_members = 1;
    // workspace
    void* act[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t act_bytes = 0;
    // two-stream pipeline over the last two phases: the low-resolution,
    // MFMA-bound phase of chunk c runs on stream_b while the high-resolution, HBM-bound
    // phase of chunk c+1 runs on stream_a.  The second phase has its own rotating set.
    void* act2[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t act2_bytes = 0;
    int pipe_first = -1;            // index of the first phase of the pipelined pair (-1: none)
    hipStream_t stream_a = nullptr, stream_b = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join_a = nullptr, ev_join_b = nullptr;
    std::vector<hipEvent_t> ev_chunk;
    void* a1 = nullptr;
    size_t a1_bytes = 0;
    // deep ensemble (BASELINE configs[3]): the members are independent networks over the same frames, so each runs on
    // its own stream with its own rotating buffers (member 0: the handle's) and the head waits for all of them - at
    // the per-GPU share of 32 frames a single member leaves most CUs idle in layers 3-4.  FAV_ENS_STREAMS=0: serial.
    struct MemberWs { void* act[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; void* a1 = nullptr; std::vector<void*> phase_out;
                      hipStream_t stream = nullptr; hipEvent_t done = nullptr; };
    std::vector<MemberWs> mws;
    hipEvent_t ev_members = nullptr;
    // Grouped launches (default: every call; fav_config.ens_grouped_max limits the frames per call or switches them off): every op of
    // the schedule is ONE launch over all members - block row blockIdx.y is member y, whose tensors lie at a constant byte
    // stride behind member 0's (the workspaces and the weights are slabs).  `grp` is what the launchers add to a launch.
    struct Group { int n = 1; long long x = 0, w = 0, b = 0, res = 0, y = 0, wb = 0, bb = 0, wa = 0, ba = 0, y2 = 0; };
    Group grp;
    bool can_group = false;
    int group_max_frames = 0;
    std::vector<hipStream_t> vit_streams;   // ViT: parts of the batch side by side
    std::vector<hipEvent_t> vit_done;
    // chained stream-K GEMM (gemm_streamk_kernel): one workspace per stream the encoder may run on (slot 0: the caller's)
    struct SkWs { float* ws = nullptr; uint32_t* flags = nullptr; uint32_t* err = nullptr; uint32_t epoch = 0; int grid_cap = 0; };
    SkWs sk[5];
    int sk_slot = 0;                        // the slot run_vit's launches use
    bool plan_no_fuse = false;      // fav_plan_schedule: build the layer-by-layer schedule (the fused one's reference)
    std::vector<void*> phase_out;   // output tensor of each phase
    float* logits = nullptr;        // [T][max_batch][cpad]
    int cpad = 0;
    int last_T = 0, last_n = 0;
    int T_eff = 1;                  // samples actually run
    int first_site = -1;
    void* host_stage = nullptr;     // for fav_classify_host
    hipStream_t host_stream = nullptr;
    // Every call that touches the handle's buffers (act[], a1, phase outputs, logits) records ev_last on its stream when it has
    // queued its work, and the next call makes ITS stream wait for that event first: two calls on different streams (a torch
    // stream and host_stream, or two torch streams) are then ordered on the device instead of racing on the activations.
    hipEvent_t ev_last = nullptr;
    bool ev_last_set = false;
    // ViT path (arch 2, 3): layers in blob order (kh == 0: a pair of fp32 vectors kept in w / b), fixed buffers
    bool vit = false;
    int vit_ntok = 0;
    void *v_patches = nullptr, *v_emb = nullptr, *v_x = nullptr, *v_y = nullptr, *v_qkv = nullptr, *v_hid = nullptr, *v_cls = nullptr;
    // profiling
    bool profiling = false;
    struct Ev { hipEvent_t a, b; int cls; int op; };
    std::vector<fav_op_profile> op_prof;   // one row per op of the static schedule
    int cur_op = -1;
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    fav_profile prof{};
};

namespace {

using namespace fav;

// Experiment knobs.  A normal build has NONE: every FAV_KNOB below is its measured default, a compile-time constant, and the
// library reads no environment variable.  `make EXPERIMENTS=1` (-DFAV_EXPERIMENTS) turns them back into environment variables -
// that build is what tools/*_bench.py, the phase-clock dumps and the A/B records under profiles/ use.  Decided schedule choices
// a caller may want to override are fav_config fields (tail_min_rows, ens_grouped_max, vit_streams, stem_fused), not knobs.
#ifdef FAV_EXPERIMENTS
long long fav_knob_read(const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; }
#define FAV_KNOB(NAME, DFLT) ([] { static const long long v_ = fav_knob_read(NAME, (DFLT)); return v_; }())
const char* fav_knob_str(const char* name) { return getenv(name); }
#else
#define FAV_KNOB(NAME, DFLT) ((long long)(DFLT))
const char* fav_knob_str(const char*) { return nullptr; }
#endif

int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

// ------------------------------------------------------------------ launchers
struct Prof {
    fav_handle* h;
    hipStream_t s;
    int idx = -1;
    Prof(fav_handle* h_, hipStream_t s_, int cls, double flops, double bytes) : h(h_), s(s_) {
        if (!h || !h->profiling) return;
        if (h->ev_used == h->ev_pool.size()) {
            fav_handle::Ev e;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
            h->ev_pool.push_back(e);
        }
        idx = (int)h->ev_used++;
        h->ev_pool[idx].cls = cls;
        h->ev_pool[idx].op = h->cur_op;
        if (h->cur_op >= 0 && h->cur_op < (int)h->op_prof.size()) {
            h->op_prof[h->cur_op].flops += flops;
            h->op_prof[h->cur_op].bytes += bytes;
            h->op_prof[h->cur_op].launches += 1;
        }
        h->prof.flops[cls] += flops;
        h->prof.bytes[cls] += bytes;
        h->prof.launches[cls] += 1;
        (void)hipEventRecord(h->ev_pool[idx].a, s);
    }
    ~Prof() {
        if (idx >= 0) (void)hipEventRecord(h->ev_pool[idx].b, s);
    }
};

// Per-device "done once" flags: a process may hold handles on several devices (fav_config.device), and a
// function attribute such as the dynamic-LDS limit is a property of (kernel, device).
struct DeviceFlags {
    std::atomic<unsigned long long> bits[2] = {};   // up to 128 device ordinals
    static int current() { int d = 0; (void)hipGetDevice(&d); return (d < 0 || d >= 128) ? 0 : d; }
    bool test_current() const { const int d = current(); return (bits[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1ull; }
    void set_current() { const int d = current(); bits[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release); }
};

DropParams make_drop(const fav_dropout_desc* d) {
    DropParams p;
    if (!d || d->site < 0) {
        p.site = -1; p.thr = 0; p.scale = 1.f; p.seed_lo = p.seed_hi = 0; p.v0 = 0; p.n_img = 1; p.first_index = 0;
        p.div_img = fastdiv_make(1);
        return p;
    }
    p.site = d->site;
    p.thr = d->threshold;
    p.scale = d->scale;
    p.seed_lo = (uint32_t)(d->seed & 0xFFFFFFFFull);
    p.seed_hi = (uint32_t)(d->seed >> 32);
    p.v0 = d->v0;
    p.n_img = d->n_img > 0 ? d->n_img : 1;
    p.first_index = d->first_image_index;
    p.div_img = fastdiv_make((uint32_t)p.n_img);
    return p;
}

// K-tile depth and ring stages of the 128-row tiles.  Measured on MI355X (profiles/r1d_conv_sweep.txt):
//  * 3x3: MFMA-bound, 64-deep tiles (half the barriers per FLOP), double buffer;
//  * 1x1 with a residual (the expanding convolution of a bottleneck): bound by HBM and by the
//    epilogue; 32-deep tiles with a 3-stage ring (50 KB of LDS -> 3 blocks per CU, which is also
//    what the 143 VGPRs allow);
//  * 1x1 without residual: the same up to K = 256; 64-deep tiles and a double buffer from K = 512.
// FAV_CONV_BK=32|64 forces a value (experiments build).
int conv_bk(int kh, int kw, int K, bool has_res) {
    const int forced = (int)FAV_KNOB("FAV_CONV_BK", 0);
    if (forced == 32 || forced == 64) return forced;
    return (kh * kw > 1 || (!has_res && K >= 512) || K >= 1024) ? 64 : 32;   // K >= 1024 with a residual: the ViT MLP's second GEMM
}

// ring depth: three 32-deep stages or two 64-deep ones (the other depths measured no better; their
// instantiations were dropped in round 3) - fixed in the launch table of launch_conv

// 256 x 256 x 64 tile (8 waves, 128 KB of LDS, one block per CU): twice the FLOPs per
// byte staged from L2, which is what bounds the MFMA-heavy shapes.
// Measured on MI355X it wins on the residual-free 3x3 convolutions and on the 1x1
// convolutions with K >= 512 (+6..25 %), and loses on the shallow 1x1 (K <= 256), whose
// time is the epilogue (nothing overlaps it at one block per CU).  FAV_CONV_BIG: 0 never, 1 always
// when Cout % 256 == 0, unset = the measured rule.
bool conv_big(int kh, int kw, long long M, int cout_pad, int K, bool has_res) {
    const int mode = (int)FAV_KNOB("FAV_CONV_BIG", 2);
    const long long min_m = FAV_KNOB("FAV_CONV_BIG_MINM", 8192);
    if (mode == 0 || cout_pad % 256 != 0 || M < min_m) return false;
    if (mode == 1) return true;
    if ((M / 256) * (cout_pad / 256) < 512) return false;   // fewer than two 256x256 tiles per CU: 128-row tiles fill the chip better
    return kh * kw > 1 ? !has_res : K >= 512;
}

// ---- projection shortcut (1x1 / stride s, 256 -> 512, no residual, no ReLU) on the row-owning structure of the tail
//      kernel: every wave gathers the fragments of its 32 output pixels straight from global memory (a strided gather
//      costs nothing there), the 512 output channels stream through as 8 weight chunks, register epilogue.  Measured
//      against the generic kernel on layer 2's shortcut (56x56x256 -> 28x28x512): see profiles/r2e_*.  FAV_PROJ=0 disables.
bool proj_enabled() {
    return FAV_KNOB("FAV_PROJ", 1) != 0;
}

bool launch_proj(fav_handle* h, const fav_conv_desc& d, hipStream_t s) {
    const bool wide = d.Cin == 512 && d.Cout == 1024;       // layer 3's shortcut: 8 waves x 32 pixels, one block per CU
    if (!proj_enabled() || d.kh != 1 || d.kw != 1 || d.pad != 0 || !((d.Cin == 256 && d.Cout == 512) || wide) || d.res || d.drop.site >= 0 ||
        d.out_f32 || d.relu != 0 || d.math_mode != FAV_MATH_BF16 || (d.stride != 1 && d.stride != 2)) return false;
    const int Ho = conv_out(d.H, 1, d.stride, 0), Wo = conv_out(d.W, 1, d.stride, 0);
    const long long M = (long long)d.n_frames * Ho * Wo;
    const int groups = h ? h->grp.n : 1;               // a grouped launch is as large as all its members together
    if (M * groups < (wide ? 512 * 256 : 4096) || M > 0x7fffffffLL || (long long)d.H * d.W * d.Cin * 2 * 4 >= 0x40000000LL) return false;
    TailParams p;
    memset(&p, 0, sizeof p);
    p.t1 = (const uint16_t*)d.x; p.wc = (const uint16_t*)d.w; p.bias_c = d.bias; p.y = (uint16_t*)d.y;
    p.H = Ho; p.W = Wo; p.HW = Ho * Wo; p.M = (int)M;
    p.in_W = d.W; p.in_HW = d.H * d.W; p.in_stride = d.stride;
    p.rega_bytes = 0;
    // [Wc x 2 | bias_b (unused) | bias_c | 256 zero bytes on a 256-byte boundary]
    p.bias_b_off = 2 * 64 * d.Cin * 2; p.bias_ca_off = p.bias_b_off + d.Cin * 5; p.wa_off0 = p.wa_off1 = p.bias_b_off;
    p.zero_off = (p.bias_b_off + (d.Cin + d.Cout) * 5 + 255) & ~255;
    p.drop = make_drop(nullptr);
    p.div_hw = fastdiv_make((uint32_t)p.HW);
    p.div_w = fastdiv_make((uint32_t)Wo);
    p.dbg = nullptr;
    const fav_handle::Group G = h ? h->grp : fav_handle::Group{};
    p.g_t1 = G.x; p.g_wc = G.w; p.g_bc = G.b; p.g_y = G.y;
    const int lds = p.zero_off + 256;
    auto kern = bottleneck_tail_kernel<256, 0, false, 2, 4, true, 32, 512, false, false>;
    auto kern_w = bottleneck_tail_kernel<512, 0, false, 2, 8, true, 32, 1024, false, false>;
    static DeviceFlags attr_set;
    if (!attr_set.test_current()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)kern_w, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        attr_set.set_current();
    }
    const double flops = 2.0 * (double)M * d.Cin * d.Cout * G.n;
    const double bytes = 2.0 * ((double)M * (d.Cin + d.Cout) + (double)d.Cin * d.Cout) * G.n;
    Prof pr(h, s, FAV_K_CONV, flops, bytes);
    if (wide) hipLaunchKernelGGL(kern_w, dim3((unsigned)((M + 255) / 256), G.n), dim3(512), lds, s, p, 0);
    else hipLaunchKernelGGL(kern, dim3((unsigned)((M + 127) / 128), G.n), dim3(256), lds, s, p, 0);
    return true;
}


const char* launch_conv(fav_handle* h, const fav_conv_desc& d, int cout_pad, int ldy, hipStream_t s) {
    if (d.Cin % 64 != 0) return "conv: Cin must be a multiple of 64";
    if (cout_pad % 64 != 0) return "conv: padded Cout must be a multiple of 64";
    // the bf16 epilogues store whole 16-byte groups of channels and range-check rows only: padded columns would land in
    // the next pixel's first channels.  Only the fp32 (logit) output, whose row pitch is the padded width, may be padded.
    if (!d.out_f32 && cout_pad != d.Cout) return "conv: a bf16 output needs Cout to be a multiple of 64 (no column padding)";
    if (d.out_f32 && ldy < cout_pad) return "conv: the fp32 output's row pitch must cover the padded Cout";
    if (cout_pad == d.Cout && ldy == d.Cout && launch_proj(h, d, s)) return nullptr;
    ConvParams p;
    p.x = (const uint16_t*)d.x; p.w = (const uint16_t*)d.w; p.bias = d.bias; p.res = (const uint16_t*)d.res; p.y = d.y;
    p.H = d.H; p.W = d.W; p.Cin = d.Cin;
    p.Ho = conv_out(d.H, d.kh, d.stride, d.pad);
    p.Wo = conv_out(d.W, d.kw, d.stride, d.pad);
    p.HWo = p.Ho * p.Wo;
    p.Cout = d.Cout; p.ldy = ldy;
    p.kw = d.kw; p.stride = d.stride; p.pad = d.pad;
    const long long M = (long long)d.n_frames * p.HWo;
    if (M <= 0 || M > 0x7fffffffLL) return "conv: row count out of range";
    p.M = (int)M;
    p.K = d.kh * d.kw * d.Cin; p.nk = p.K / 64;
    p.relu = d.relu; p.out_f32 = d.out_f32;
    p.drop = make_drop(&d.drop);
    p.div_hwo = fastdiv_make((uint32_t)p.HWo);
    p.div_w = fastdiv_make((uint32_t)p.Wo);
    if (p.drop.site >= 0 && (p.drop.v0 < 0 || p.drop.v0 + d.n_frames > 0x7fffffffLL)) return "conv: virtual frame index out of range";
    p.dbg = nullptr;
    const fav_handle::Group G = h ? h->grp : fav_handle::Group{};
    p.g_x = G.x; p.g_w = G.w; p.g_bias = G.b; p.g_res = G.res; p.g_y = G.y;
    if (d.out_f32 && p.drop.site >= 0) return "conv: dropout on fp32 output unsupported";
    // the ViT encoder's GEMMs (M = 197 rows per frame, K = 768 / 3072): the 256 x 256 tile from 50 of them up (measured, round 4,
    // tools/experiments/r4_vit_tiles.sh: +18 % at 128 frames on two streams, +1.5 % at the 64-frame share; since the GELU epilogue
    // shrank to 14 instructions per element it no longer needs a second block per CU to hide behind); below that 128-row tiles with
    // 32-deep steps at three blocks per CU
    const bool vit = h && h->vit;
    const long long vit_big_tiles = FAV_KNOB("FAV_VIT_BIG_TILES", 50);
    const bool vit_big = vit && cout_pad % 256 == 0 && d.kh * d.kw * d.Cin >= 512 && (M / 256) * (cout_pad / 256) >= vit_big_tiles &&
                         FAV_KNOB("FAV_CONV_BIG", 2) != 0;
    const bool big = vit ? vit_big
                         : conv_big(d.kh, d.kw, M * G.n, cout_pad, d.kh * d.kw * d.Cin, d.res != nullptr);   // 256 x 256 x 64 tile, 8 waves, 128 KB of LDS
    const int BN = big ? 256 : ((cout_pad % 128 == 0) ? 128 : 64);
    const int BK = big ? 64 : (vit ? 32 : conv_bk(d.kh, d.kw, d.kh * d.kw * d.Cin, d.res != nullptr));
    const int BM = big ? 256 : 128;   // (256-row tiles with 128 columns lose to two blocks per CU of 128-row tiles on every 3x3 shape)
    // measured: issuing the DMA after the first MFMA group gains ~7 % on the 256x256 3x3 launches and
    // loses 3-5 % on the 128-row tiles and on every 1x1
    p.stage_mid = (big && d.kh * d.kw > 1) ? 1 : 0;
    {   // LDS-DMA offsets are 32-bit from the tile's first frame; out-of-range lanes use 0x80000000
        const double frame_bytes = 2.0 * d.H * d.W * d.Cin;
        const double span = (BM / (double)p.HWo + 2.0) * frame_bytes + 2.0 * ((double)d.pad * d.W + d.pad) * d.Cin +
                            2.0 * (((double)d.kh * d.W + d.kw) * d.Cin);
        if (span >= 2147483647.0 || 2.0 * BN * (double)p.K >= 2147483647.0) return "conv: frame too large for 32-bit tile offsets";
    }
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = cout_pad / BN;
    const long long tiles = (long long)p.tiles_m * p.tiles_n;
    if (tiles > 0x7fffffffLL) return "conv: too many tiles";
    const double flops = 2.0 * (double)M * d.Cout * p.K * G.n;
    const double bytes = 2.0 * ((double)d.n_frames * d.H * d.W * d.Cin + (double)M * d.Cout * (d.res ? 2 : 1) * (d.out_f32 ? 2 : 1)
                                + (double)d.Cout * p.K) * G.n;
    const bool dbg_on = FAV_KNOB("FAV_CONV_DBG", 0) != 0;   // experiments build only: per-block phase clocks
    if (dbg_on && !h) { (void)hipMalloc((void**)&p.dbg, (size_t)tiles * 32); (void)hipMemset(p.dbg, 0, (size_t)tiles * 32); }
    auto dbg_report = [&](long long nblocks, int bm, int bn, int bk) {
        if (!p.dbg) return;
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> t((size_t)nblocks * 4);
        (void)hipMemcpy(t.data(), p.dbg, t.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(p.dbg);
        unsigned long long lo = ~0ull, hi = 0;
        double ph[3] = {0, 0, 0};
        for (long long i = 0; i < nblocks; ++i) {
            lo = std::min(lo, t[i * 4]); hi = std::max(hi, t[i * 4 + 3]);
            for (int j = 0; j < 3; ++j) ph[j] += (double)(t[i * 4 + j + 1] - t[i * 4 + j]);
        }
        const double span = (double)(hi - lo), life = ph[0] + ph[1] + ph[2];
        fprintf(stderr, "[conv dbg] blocks %lld BMxBNxBK %dx%dx%d span %.1f us; per block: prologue %.2f us, k-loop %.2f us, epilogue %.2f us; "
                        "resident blocks/CU %.2f\n", nblocks, bm, bn, bk, span / 100.0, ph[0] / nblocks / 100.0, ph[1] / nblocks / 100.0,
                ph[2] / nblocks / 100.0, life / span / 256.0);
    };
    Prof pr(h, s, FAV_K_CONV, flops, bytes);
    // 3x3 / stride 1 / pad 1 with Cin <= 128 and the whole Cout in one tile: the input patch is staged once
    // per 256 output pixels instead of once per tap (conv3x3_halo_kernel).  FAV_CONV_HALO=0 disables.
    const int halo_mode = (int)FAV_KNOB("FAV_CONV_HALO", 1);
    if (halo_mode && d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad == 1 && !d.res && p.drop.site < 0 && !d.out_f32 &&
        (d.Cin == 64 || d.Cin == 128) && d.Cout == cout_pad && d.Cout == d.Cin && M * G.n >= 2048) {
        // Cin 64: 512-pixel tiles, all 9 K tiles of the weights resident; Cin 128: 256-pixel tiles, weights double-buffered per tap
        // 256-pixel tiles, 8 waves (measured best on both shapes); FAV_HALO_CFG=0 selects 128-pixel tiles with 4 waves and
        // several blocks per CU for experiments
        const int halo_cfg = (int)FAV_KNOB("FAV_HALO_CFG", 1);
        const int HBM = halo_cfg == 0 ? 128 : 256;
        const int wstages = d.Cin == 64 ? (halo_cfg == 0 ? 2 : 3) : (halo_cfg == 0 ? 2 : 4);   // K tiles of weights held in LDS
        const int patch_bytes = (int)((((long long)(HBM + 2 * d.W + 2) * d.Cin * 2) + 1023) / 1024 * 1024);
        const int lds = patch_bytes + wstages * d.Cout * 128 + d.Cout * 5 + 512;   // + 256 zero bytes on a 256-byte boundary
        if (lds <= 160 * 1024) {
            p.nk = 9 * d.Cin / 64;
            dim3 hgrid((unsigned)((p.M + HBM - 1) / HBM), G.n);
#define FAV_HALO(KERNEL_)                                                                                            \
    do {                                                                                                             \
        static DeviceFlags attr_set;   /* hipFuncSetAttribute applies to the CURRENT device only */                  \
        if (!attr_set.test_current()) {                                                                              \
            if (hipFuncSetAttribute((const void*)KERNEL_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
                return "conv: cannot reserve LDS for the staged 3x3 kernel";                                         \
            attr_set.set_current();                                                                                  \
        }                                                                                                            \
        hipLaunchKernelGGL(KERNEL_, hgrid, dim3(HBM * 2), lds, s, p, patch_bytes);                                   \
    } while (0)
            const bool bf = d.math_mode == FAV_MATH_BF16;
            if (d.Cin == 64 && halo_cfg == 0) {
                if (bf) FAV_HALO((conv3x3_halo_kernel<64, 64, 128, 2, 1, 3, 0>)); else FAV_HALO((conv3x3_halo_kernel<64, 64, 128, 2, 1, 3, 1>));
            } else if (d.Cin == 64) {
                if (bf) FAV_HALO((conv3x3_halo_kernel<64, 64, 256, 3, 1, 4, 0>)); else FAV_HALO((conv3x3_halo_kernel<64, 64, 256, 3, 1, 4, 1>));
            } else if (halo_cfg == 0) {
                if (bf) FAV_HALO((conv3x3_halo_kernel<128, 128, 128, 2, 1, 2, 0>)); else FAV_HALO((conv3x3_halo_kernel<128, 128, 128, 2, 1, 2, 1>));
            } else {
                if (bf) FAV_HALO((conv3x3_halo_kernel<128, 128, 256, 2, 2, 2, 0>)); else FAV_HALO((conv3x3_halo_kernel<128, 128, 256, 2, 2, 2, 1>));
            }
#undef FAV_HALO
            dbg_report((p.M + HBM - 1) / HBM, HBM, d.Cout, 64);
            return nullptr;
        }
    }
    dim3 grid((unsigned)tiles, G.n);
    p.nk = p.K / BK;
    // FAV_CONV_EPI=0 selects the round-1 epilogue (fp32 staging through LDS) for A/B measurements
    // measured (profiles/r2b_conv_epilogue_ab.txt): the register epilogue wins 2-4 % on the 3x3 and K >= 512 launches
    // (also with a residual on the 256 x 256 tile: layer 4's expand 1.24 vs 1.29 ms) and loses ~3 % on the 128-row
    // tiles with a residual, so those keep the staged one.  FAV_CONV_EPI=0|1 forces.
    const int epi_forced = (int)FAV_KNOB("FAV_CONV_EPI", -1);
    const int epi = epi_forced >= 0 ? epi_forced : ((d.res && !big) ? 0 : 1);
#define FAV_LAUNCH(BN_, BK_, NS_, MODE_)                                                                          \
    do {                                                                                                          \
        if (d.relu == 2) {      /* GELU (ViT MLP): the instantiations that carry it */                            \
            if (epi) hipLaunchKernelGGL((conv_igemm_kernel<128, BN_, BK_, NS_, MODE_, 2, 0, 1, 0, true>), grid, dim3(256), 0, s, p); \
            else hipLaunchKernelGGL((conv_igemm_kernel<128, BN_, BK_, NS_, MODE_, 2, 0, 0, 0, true>), grid, dim3(256), 0, s, p);     \
        } else if (epi) hipLaunchKernelGGL((conv_igemm_kernel<128, BN_, BK_, NS_, MODE_, 2, 0, 1>), grid, dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((conv_igemm_kernel<128, BN_, BK_, NS_, MODE_, 2, 0, 0>), grid, dim3(256), 0, s, p);     \
    } while (0)
#define FAV_LAUNCH_MODE(MODE_)                                                                    \
    do {                                                                                          \
        if (BN == 128) { if (BK == 32) FAV_LAUNCH(128, 32, 3, MODE_); else FAV_LAUNCH(128, 64, 2, MODE_); } \
        else { if (BK == 32) FAV_LAUNCH(64, 32, 3, MODE_); else FAV_LAUNCH(64, 64, 2, MODE_); }   \
    } while (0)
#define FAV_LAUNCH_BIG(MODE_)                                                                                     \
    do {                                                                                                          \
        const int pp = (int)FAV_KNOB("FAV_CONV_PP", 1);                                                           \
        if (d.relu == 2) {                                                                                        \
            if (epi && pp && MODE_ == 0) hipLaunchKernelGGL((conv_igemm_kernel<256, 256, 64, 2, 0, 2, 0, 1, 1, true>), grid, dim3(512), 0, s, p); \
            else if (epi) hipLaunchKernelGGL((conv_igemm_kernel<256, 256, 64, 2, MODE_, 2, 0, 1, 0, true>), grid, dim3(512), 0, s, p);    \
            else hipLaunchKernelGGL((conv_igemm_kernel<256, 256, 64, 2, MODE_, 2, 0, 0, 0, true>), grid, dim3(512), 0, s, p);        \
        } else if (epi && pp && MODE_ == 0) hipLaunchKernelGGL((conv_igemm_kernel<256, 256, 64, 2, 0, 2, 0, 1, 1>), grid, dim3(512), 0, s, p); \
        else if (epi) hipLaunchKernelGGL((conv_igemm_kernel<256, 256, 64, 2, MODE_, 2, 0, 1>), grid, dim3(512), 0, s, p);    \
        else hipLaunchKernelGGL((conv_igemm_kernel<256, 256, 64, 2, MODE_, 2, 0, 0>), grid, dim3(512), 0, s, p);        \
    } while (0)
    if (big) {
        if (d.math_mode == FAV_MATH_BF16) FAV_LAUNCH_BIG(0); else FAV_LAUNCH_BIG(1);
    } else if (d.math_mode == FAV_MATH_BF16) FAV_LAUNCH_MODE(0); else FAV_LAUNCH_MODE(1);
#undef FAV_LAUNCH_BIG
#undef FAV_LAUNCH_MODE
#undef FAV_LAUNCH
    dbg_report(tiles, BM, BN, BK);
    return nullptr;
}


// ---- bottleneck tail (conv_b 3x3 -> conv_c 1x1 + residual + dropout -> next block's conv_a 1x1), one launch ----
struct TailGeom { int patch_bytes, rega_bytes, lds_bytes, nw, wc2, rp, bias_b_off, bias_ca_off, wa_off0, wa_off1, zero_off; };
// 512 mid channels (layer 4): the expanding 1x1 + residual + dropout on the row-owning structure.  FAV_TAIL_L4=0 disables.
bool tail_l4() {
    return FAV_KNOB("FAV_TAIL_L4", 1) != 0;
}
// LDS plan of bottleneck_tail_kernel<CMID, NRED, HAS3X3, NS, NW, WC2> (must match the kernel's own layout)
bool tail_geometry(int cmid, int nred, bool has3x3, int W, TailGeom* g) {
    const int cout = 4 * cmid;
    // default plan behind the weight buffers: bias_b | bias_c | bias_a | 256 zero bytes on a 256-byte boundary
    auto finish = [&](int bias_b_off, int wa_off0, int wa_bytes) {
        g->bias_b_off = bias_b_off; g->bias_ca_off = bias_b_off + cmid * 5;
        g->wa_off0 = wa_off0; g->wa_off1 = wa_off0 + wa_bytes;
        g->zero_off = (bias_b_off + (cmid + cout + nred) * 5 + 255) & ~255;
        g->lds_bytes = g->zero_off + 256;
    };
    // 256 mid channels (layer 3): the expanding 1x1 alone, or with the next block's reduce (then 8 waves x 16 rows)
    if (cmid == 512) {      // layer 4: the expanding 1x1 alone
        if (has3x3 || nred != 0) return false;
        // 8 waves x 32 pixels, Wc double-buffered (2 x 64 KB), one block per CU: 1.14 ms against 1.24 ms for 4 waves with a
        // single Wc buffer at two blocks per CU and 1.27 ms for the generic kernel (profiles/r2j_tail_l4.txt)
        g->patch_bytes = 0; g->rega_bytes = 0; g->nw = 8; g->rp = 32; g->wc2 = 1;
        finish(2 * 65536, 2 * 65536, 0);
        return true;
    }
    else if (cmid == 256) { if ((has3x3 && nred != 0) || !(nred == 0 || nred == 256)) return false; }
    else if ((cmid != 64 && cmid != 128) || !(nred == 0 || nred == cmid || nred == 128)) return false;
    if (cmid == 256 && has3x3) {
        // conv_b as the generic 256 x 256 x 64 loop (two 64 KB stages); T2, then the Wc buffers, reuse those 128 KB
        g->patch_bytes = 0; g->rega_bytes = 128 * 1024; g->nw = 8; g->rp = 32; g->wc2 = 1;
        finish(g->rega_bytes, g->rega_bytes, 0);
        return true;
    }
    const int rp = (cmid == 256 && nred == 256) ? 16 : 32;
    const int rowb = cmid * 2, ns = cmid == 64 ? 3 : 2;
    const int wc_bytes = 64 * rowb, wa_bytes = nred * 128;
    auto plan = [&](int nw) {
        const int bm = rp * nw;
        const int patch = has3x3 ? (int)((((long long)(bm + 2 * W + 2) * rowb) + 1023) / 1024 * 1024) : 0;
        // region A: patch | T2 tile | Y chunk; without conv_b the T2 fragments come straight from global memory
        int rega = has3x3 ? std::max(std::max(patch, bm * 128), bm * rowb) : (nred > 0 ? bm * 128 : 0);
        rega = (rega + 1023) / 1024 * 1024;
        const int ring = has3x3 ? ns * cmid * 128 : 0;
        const int budget = nw == 4 ? 80 * 1024 : 160 * 1024;
        g->patch_bytes = patch; g->rega_bytes = rega; g->nw = nw; g->rp = rp;
        // Wc double-buffered when two blocks still fit a CU (4-wave blocks) / the block fits at all (8-wave blocks)
        g->wc2 = 1;
        int regb = std::max(ring, 2 * wc_bytes + 2 * wa_bytes);
        finish(rega + regb, rega + 2 * wc_bytes, wa_bytes);
        if (g->lds_bytes > budget) {
            g->wc2 = 0;
            regb = std::max(ring, wc_bytes + 2 * wa_bytes);
            finish(rega + regb, rega + wc_bytes, wa_bytes);
        }
        return g->lds_bytes <= budget;
    };
    // Waves per block (pixels per block = 32 * waves).  64 mid channels: 4 waves (128 pixels, two blocks per CU; an 8-wave / 256-pixel
    // variant measured the same, profiles/r2a_tail_bench.txt); 128 mid channels: 8 waves (256 pixels, one block per CU; 4-wave blocks
    // at two per CU - 81 920 B each with a compact LDS plan - are bit-identical and within 1 %: profiles/r4d_l2_tail_two_blocks_per_cu.txt);
    // 256 (conv_c alone): 4 waves, two blocks per CU.
    return plan((cmid == 128 || (cmid == 256 && nred == 256)) ? 8 : 4);
}

bool tail_wide() {
    return FAV_KNOB("FAV_TAIL_WIDE", 1) != 0;
}

// 256 mid channels: conv_b (generic loop) + conv_c in one launch.  FAV_TAIL_WIDE3X3=0 keeps the 3x3 as its own launch.
bool tail_wide3x3() {
    return FAV_KNOB("FAV_TAIL_WIDE3X3", 1) != 0;
}

// 512 mid channels (layer 4): the expanding 1x1 + residual + dropout on the row-owning structure.  FAV_TAIL_L4=0 disables.
bool tail_enabled() {
    return FAV_KNOB("FAV_FUSE", 1) != 0;
}

const char* launch_tail(fav_handle* h, const fav_tail_desc& d, hipStream_t s) {
    const bool has3x3 = d.wb != nullptr;
    const int nred = d.wa ? d.Nred : 0;
    TailGeom g;
    if (!tail_geometry(d.Cmid, nred, has3x3, d.W, &g)) return "bottleneck tail: unsupported shape";
    const long long M = (long long)d.n_frames * d.H * d.W;
    if (M <= 0 || M > 0x7fffffffLL) return "bottleneck tail: row count out of range";
    TailParams p;
    p.t1 = (const uint16_t*)d.x; p.wb = (const uint16_t*)d.wb; p.bias_b = d.bias_b;
    p.wc = (const uint16_t*)d.wc; p.bias_c = d.bias_c; p.res = (const uint16_t*)d.res; p.y = (uint16_t*)d.y;
    p.wa = (const uint16_t*)d.wa; p.bias_a = d.bias_a; p.t1n = (uint16_t*)d.t1n;
    p.H = d.H; p.W = d.W; p.HW = d.H * d.W; p.M = (int)M;
    p.in_W = d.W; p.in_HW = p.HW; p.in_stride = 1;
    p.rega_bytes = g.rega_bytes;
    p.bias_b_off = g.bias_b_off; p.bias_ca_off = g.bias_ca_off; p.wa_off0 = g.wa_off0; p.wa_off1 = g.wa_off1; p.zero_off = g.zero_off;
    p.drop = make_drop(&d.drop);
    p.div_hw = fastdiv_make((uint32_t)p.HW);
    p.div_w = fastdiv_make((uint32_t)d.W);
    if (p.drop.site >= 0 && (p.drop.v0 < 0 || p.drop.v0 + d.n_frames > 0x7fffffffLL)) return "bottleneck tail: virtual frame index out of range";
    const fav_handle::Group G = h ? h->grp : fav_handle::Group{};
    p.g_t1 = G.x; p.g_res = G.res; p.g_y = G.y; p.g_t1n = G.y2;
    p.g_wb = G.wb; p.g_bb = G.bb; p.g_wc = G.w; p.g_bc = G.b; p.g_wa = G.wa; p.g_ba = G.ba;
    p.site_e = d.entry_site;
    p.rs_T = 0; p.rs_tps = 0;
    if (d.res_entry) {
        if (!(d.Cmid == 64 && nred == 64 && has3x3) || p.drop.site < 0 || d.entry_site < 0 || !d.res)
            return "bottleneck tail: res_entry needs Cmid = Nred = 64 with the 3x3 and both dropout sites";
        if ((double)p.drop.n_img * p.HW * 4.0 * d.Cmid * 2.0 >= 2147483647.0) return "bottleneck tail: cached tensor too large for 32-bit offsets";
    }
    const int cmid = d.Cmid, cout = 4 * cmid;
    const double flops = 2.0 * (double)M * ((has3x3 ? 9.0 * cmid * cmid : 0.0) + (double)cmid * cout + (double)cout * nred) * G.n;
    // res_entry: the residual is the cached tensor [n_img][HW][cout], counted ONCE (every sample's tile re-reads it, from L2), not once per row it serves
    const double res_rows = d.res_entry ? (double)std::min<long long>(d.n_frames, p.drop.n_img) * p.HW : (double)M;
    const double bytes = 2.0 * ((double)M * (cmid + 1.0 * cout + nred) + res_rows * cout + (has3x3 ? 9.0 * cmid * cmid : 0.0) + (double)cmid * cout + (double)cout * nred) * G.n;
    Prof pr(h, s, FAV_K_CONV, flops, bytes);
    { const int lds_pad = (int)FAV_KNOB("FAV_TAIL_LDS_PAD", 0); if (g.lds_bytes + lds_pad <= 160 * 1024) g.lds_bytes += lds_pad; }   // experiments build: fewer blocks per CU
    const int bm = g.rp * g.nw;
    const long long nblocks = (M + bm - 1) / bm;
    dim3 grid((unsigned)nblocks, G.n);
    p.dbg = nullptr;
    const bool dbg_on = FAV_KNOB("FAV_CONV_DBG", 0) != 0;   // experiments build only: per-block phase clocks
    if (dbg_on && !h) { (void)hipMalloc((void**)&p.dbg, (size_t)nblocks * 128); (void)hipMemset(p.dbg, 0, (size_t)nblocks * 128); }
    auto dbg_report = [&]() {
        if (!p.dbg) return;
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> t((size_t)nblocks * 16);
        (void)hipMemcpy(t.data(), p.dbg, t.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(p.dbg);
        if (const char* dump = fav_knob_str("FAV_CONV_DBG_DUMP")) {     // raw stamps, [block][16] u64, appended: tools/phase_overlap.py
            if (FILE* f = fopen(dump, "ab")) { const long long hdr[2] = {nblocks, cmid * 1000 + nred}; fwrite(hdr, 8, 2, f); fwrite(t.data(), 8, t.size(), f); fclose(f); }
        }
        unsigned long long lo = ~0ull, hi = 0;
        double ph[5] = {0, 0, 0, 0, 0};
        double cy[5] = {0, 0, 0, 0, 0};   // chunk 1 of wave 0, shader clocks: step A | epilogue | step C | DMA wait | barrier
        double cz[3] = {0, 0, 0};         // ... residual wait at the top of the chunk | requests for the next chunk | the whole chunk (top of 1 to top of 2)
        for (long long i = 0; i < nblocks; ++i) {
            lo = std::min(lo, t[i * 16]); hi = std::max(hi, t[i * 16 + 5]);
            unsigned long long prev = t[i * 16];
            for (int j = 0; j < 5; ++j) { const unsigned long long c = t[i * 16 + j + 1] ? t[i * 16 + j + 1] : prev; ph[j] += (double)(c - prev); prev = c; }
            for (int j = 0; j < 5; ++j) cy[j] += (double)(t[i * 16 + 7 + j] - t[i * 16 + 6 + j]);
            cz[0] += (double)(t[i * 16 + 13] - t[i * 16 + 12]); cz[1] += (double)(t[i * 16 + 6] - t[i * 16 + 13]); cz[2] += (double)(t[i * 16 + 14] - t[i * 16 + 12]);
        }
        const double span = (double)(hi - lo), life = ph[0] + ph[1] + ph[2] + ph[3] + ph[4];
        fprintf(stderr, "[tail dbg] blocks %lld x %d px, Cmid %d Nred %d 3x3 %d: span %.1f us; per block: patch wait %.2f us, 3x3 loop %.2f us, "
                        "T2 + first weights %.2f us, chunks %.2f us, tail %.2f us; resident blocks/CU %.2f\n", nblocks, bm, cmid, nred, (int)has3x3,
                span / 100.0, ph[0] / nblocks / 100.0, ph[1] / nblocks / 100.0, ph[2] / nblocks / 100.0, ph[3] / nblocks / 100.0,
                ph[4] / nblocks / 100.0, life / span / 256.0);
        fprintf(stderr, "[tail dbg]   chunk 1, wave 0, shader clocks: step A %.0f, epilogue %.0f, step C %.0f, DMA wait %.0f, barrier %.0f\n",
                cy[0] / nblocks, cy[1] / nblocks, cy[2] / nblocks, cy[3] / nblocks, cy[4] / nblocks);
        fprintf(stderr, "[tail dbg]   ... residual wait %.0f, next chunk's requests %.0f, whole chunk 1 %.0f\n", cz[0] / nblocks, cz[1] / nblocks, cz[2] / nblocks);
    };
#define FAV_TAIL(CMID_, NRED_, H3_, NS_, NW_, WC2_)                                                                   \
    do {                                                                                                              \
        static DeviceFlags attr_set;                                                                                  \
        if (!attr_set.test_current()) {                                                                               \
            if (hipFuncSetAttribute((const void*)bottleneck_tail_kernel<CMID_, NRED_, H3_, NS_, NW_, WC2_>,           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)            \
                return "bottleneck tail: cannot reserve LDS";                                                         \
            attr_set.set_current();                                                                                   \
        }                                                                                                             \
        hipLaunchKernelGGL((bottleneck_tail_kernel<CMID_, NRED_, H3_, NS_, NW_, WC2_>), grid, dim3(NW_ * 64), g.lds_bytes, s, p, g.patch_bytes); \
        dbg_report();                                                                                                 \
        return nullptr;                                                                                               \
    } while (0)
#define FAV_TAIL_W(CMID_, NRED_, H3_, NS_, NW_) do { if (g.wc2) FAV_TAIL(CMID_, NRED_, H3_, NS_, NW_, true); else FAV_TAIL(CMID_, NRED_, H3_, NS_, NW_, false); } while (0)
#define FAV_TAIL_N(CMID_, NRED_, H3_, NS_) FAV_TAIL_W(CMID_, NRED_, H3_, NS_, 4)
    if (d.res_entry) {
        // whole samples, tiles that do not straddle them: the T tiles over one pixel tile of the cached tensor run back to back
        const bool sample_minor = FAV_KNOB("FAV_ENTRY_RES_ORDER", 1) != 0;
        const long long sample_rows = (long long)p.drop.n_img * p.HW;
        if (sample_minor && p.drop.v0 % p.drop.n_img == 0 && d.n_frames % p.drop.n_img == 0 && sample_rows % bm == 0) {
            p.rs_T = d.n_frames / p.drop.n_img;
            p.rs_tps = (int)(sample_rows / bm);
        }
        static DeviceFlags attr_set;
        auto k0 = bottleneck_tail_kernel<64, 64, true, 3, 4, false, 32, 0, true, true, true>;
        auto k1 = bottleneck_tail_kernel<64, 64, true, 3, 4, true, 32, 0, true, true, true>;
        if (!attr_set.test_current()) {
            if (hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return "bottleneck tail: cannot reserve LDS";
            attr_set.set_current();
        }
        if (g.wc2) hipLaunchKernelGGL(k1, grid, dim3(256), g.lds_bytes, s, p, g.patch_bytes);
        else hipLaunchKernelGGL(k0, grid, dim3(256), g.lds_bytes, s, p, g.patch_bytes);
        dbg_report();
        return nullptr;
    }
    if (cmid == 64) {
        if (has3x3) { if (nred == 0) FAV_TAIL_N(64, 0, true, 3); if (nred == 64) FAV_TAIL_N(64, 64, true, 3); if (nred == 128) FAV_TAIL_N(64, 128, true, 3); }
        else { if (nred == 0) FAV_TAIL_N(64, 0, false, 3); if (nred == 64) FAV_TAIL_N(64, 64, false, 3); if (nred == 128) FAV_TAIL_N(64, 128, false, 3); }
    } else if (cmid == 128) {
        if (has3x3) { if (nred == 0) FAV_TAIL_W(128, 0, true, 2, 8); if (nred == 128) FAV_TAIL_W(128, 128, true, 2, 8); }
        else { if (nred == 0) FAV_TAIL_W(128, 0, false, 2, 8); if (nred == 128) FAV_TAIL_W(128, 128, false, 2, 8); }
    } else if (cmid == 512) {
        FAV_TAIL(512, 0, false, 2, 8, true);
    } else if (nred == 0 && has3x3) {
        FAV_TAIL(256, 0, true, 2, 8, true);
    } else if (nred == 0) {
        FAV_TAIL_W(256, 0, false, 2, 4);
    } else {
        static DeviceFlags attr_set;
        if (!attr_set.test_current()) {
            if (hipFuncSetAttribute((const void*)bottleneck_tail_kernel<256, 256, false, 2, 8, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)bottleneck_tail_kernel<256, 256, false, 2, 8, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return "bottleneck tail: cannot reserve LDS";
            attr_set.set_current();
        }
        if (g.wc2) hipLaunchKernelGGL((bottleneck_tail_kernel<256, 256, false, 2, 8, true, 16>), grid, dim3(512), g.lds_bytes, s, p, g.patch_bytes);
        else hipLaunchKernelGGL((bottleneck_tail_kernel<256, 256, false, 2, 8, false, 16>), grid, dim3(512), g.lds_bytes, s, p, g.patch_bytes);
        dbg_report();
        return nullptr;
    }
#undef FAV_TAIL_N
#undef FAV_TAIL_W
#undef FAV_TAIL
    return "bottleneck tail: unsupported shape";
}

// One work item per thread whenever the grid allows (grid-stride loops only catch the rest): on gfx9
// stores count in vmcnt, so a second iteration's loads would wait for the first iteration's stores.
unsigned grid_for(long long work_items) {
    long long g = (work_items + 255) / 256;
    return (unsigned)std::max<long long>(1, std::min<long long>(g, (1ll << 24) - 1));
}

void launch_stem(fav_handle* h, const void* images, int layout, int n, int H, int W, int kh, int kw, int stride, int pad,
                 int kpad, const float* mean, const float* istd, void* out, hipStream_t s) {
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    const long long total = (long long)n * Ho * Wo * (kpad / 8);
    Prof pr(h, s, FAV_K_STEM, 0.0, (double)n * H * W * 3 * (layout == 0 ? 1 : 4) + (double)total * 16);
    const long long total_pix = (long long)n * Ho * Wo;
    const int ppb = kh <= 64 ? 256 / kh : 0;                    // pixels per block of the row-wise kernel
    const size_t lds = (size_t)ppb * kpad * 2;
    if (ppb >= 4 && lds <= 64 * 1024 && total_pix / ppb < (1ll << 31) - 1) {
        const unsigned blocks = (unsigned)((total_pix + ppb - 1) / ppb);
        if (layout == FAV_LAYOUT_NHWC_U8)
            hipLaunchKernelGGL((stem_im2col_rows_kernel<0>), dim3(blocks), dim3(256), lds, s, images, (uint4*)out, total_pix, H, W, Ho, Wo,
                               kh, kw, stride, pad, kpad, ppb, mean[0], mean[1], mean[2], istd[0], istd[1], istd[2]);
        else
            hipLaunchKernelGGL((stem_im2col_rows_kernel<1>), dim3(blocks), dim3(256), lds, s, images, (uint4*)out, total_pix, H, W, Ho, Wo,
                               kh, kw, stride, pad, kpad, ppb, mean[0], mean[1], mean[2], istd[0], istd[1], istd[2]);
        return;
    }
    if (layout == FAV_LAYOUT_NHWC_U8)
        hipLaunchKernelGGL((stem_im2col_kernel<0>), dim3(grid_for(total)), dim3(256), 0, s, images, (uint4*)out, n, H, W,
                           Ho, Wo, kh, kw, stride, pad, kpad, mean[0], mean[1], mean[2], istd[0], istd[1], istd[2]);
    else
        hipLaunchKernelGGL((stem_im2col_kernel<1>), dim3(grid_for(total)), dim3(256), 0, s, images, (uint4*)out, n, H, W,
                           Ho, Wo, kh, kw, stride, pad, kpad, mean[0], mean[1], mean[2], istd[0], istd[1], istd[2]);
}

void launch_maxpool(fav_handle* h, const void* x, void* y, int n, int H, int W, int C, hipStream_t s) {
    const int Ho = conv_out(H, 3, 2, 1), Wo = conv_out(W, 3, 2, 1);
    const long long total = (long long)n * Ho * Wo * (C / 8);
    Prof pr(h, s, FAV_K_MAXPOOL, 0.0, 2.0 * ((double)n * H * W * C + (double)n * Ho * Wo * C));
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const uint4*)x, (uint4*)y, n, H, W,
                       C, Ho, Wo);
}

// normalise + 7x7/2 conv (64 channels, weights [64][192]) + bias + ReLU + 3x3/2 max pool, frames -> [n][Hp][Wp][64] bf16
const char* launch_stem_pool(fav_handle* h, const void* images, int layout, int n, int H, int W, const void* w, const float* bias,
                             const float* mean, const float* istd, void* out, hipStream_t s) {
    if (n < 1 || H < 1 || W < 1) return "stem: empty input";
    if (layout != FAV_LAYOUT_NHWC_U8 && layout != FAV_LAYOUT_NHWC_F32) return "stem: unknown layout";
    StemPoolParams p;
    p.images = images; p.w = (const uint16_t*)w; p.bias = bias; p.out = (uint16_t*)out;
    p.n = n; p.H = H; p.W = W;
    p.Hc = conv_out(H, 7, 2, 3); p.Wc = conv_out(W, 7, 2, 3);
    if (p.Hc < 1 || p.Wc < 1) return "stem: input too small";
    if ((double)H * W * 12.0 >= 2147483647.0) return "stem: frame too large for 32-bit offsets";
    p.Hp = conv_out(p.Hc, 3, 2, 1); p.Wp = conv_out(p.Wc, 3, 2, 1);
    p.tiles_y = (p.Hp + 7) / 8; p.tiles_x = (p.Wp + 7) / 8;
    p.tiles = (long long)n * p.tiles_y * p.tiles_x;
    p.m0 = mean[0]; p.m1 = mean[1]; p.m2 = mean[2]; p.i0 = istd[0]; p.i1 = istd[1]; p.i2 = istd[2];
    const fav_handle::Group G = h ? h->grp : fav_handle::Group{};
    p.g_w = G.w; p.g_bias = G.b; p.g_out = G.y;
    const double M = (double)n * p.Hc * p.Wc;
    Prof pr(h, s, FAV_K_CONV, 2.0 * M * 64 * 192 * G.n,
            ((double)n * H * W * 3 * (layout == FAV_LAYOUT_NHWC_U8 ? 1 : 4) + 2.0 * n * p.Hp * p.Wp * 64 + 2.0 * 64 * 192) * G.n);
    // two blocks per CU (232 VGPRs, 66 KB of LDS); blocks loop over the tiles with the weights in registers
    const unsigned blocks = (unsigned)std::min<long long>(p.tiles, std::max(1, 256 * 2 / G.n));
    if (layout == FAV_LAYOUT_NHWC_U8) hipLaunchKernelGGL(stem7_pool_kernel<0>, dim3(blocks, G.n), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(stem7_pool_kernel<1>, dim3(blocks, G.n), dim3(256), 0, s, p);
    return nullptr;
}

void launch_avgpool(fav_handle* h, const void* x, void* y, int n, int HW, int C, const DropParams& dp, hipStream_t s) {
    const long long total = (long long)n * (C / 16);
    Prof pr(h, s, FAV_K_AVGPOOL, 0.0, 2.0 * ((double)n * HW * C + (double)n * C));
    const float inv = 1.0f / (float)HW;
    const fav_handle::Group G = h ? h->grp : fav_handle::Group{};
    hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for(total), G.n), dim3(256), 0, s, (const uint4*)x, (uint4*)y, n, HW, C, inv,
                       dp, G.x, G.y);
}

void launch_entry_dropout(fav_handle* h, const void* x, void* out, long long elems, int n_out, const DropParams& dp,
                          hipStream_t s) {
    const long long total = (elems / 16) * std::min<long long>(dp.n_img, n_out);   // threads: one per cached chunk
    Prof pr(h, s, FAV_K_DROPOUT, 0.0, 2.0 * (double)elems * (n_out + std::min<long long>(dp.n_img, n_out)));
    hipLaunchKernelGGL(entry_dropout_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const uint4*)x, (uint4*)out,
                       elems / 16, n_out, dp);
}

// Entry dropout + the 1x1 reduce behind it (entry_reduce_kernel): 256 -> 64 channels, the first dropout site of the
// all_blocks policy.  FAV_ENTRY_FUSE=0 keeps the two launches.
bool entry_reduce_enabled() {
    return FAV_KNOB("FAV_ENTRY_FUSE", 1) != 0;
}
bool entry_reduce_supported(int C, int nred, long long n_img, long long HW) {
    return C == 256 && nred == 64 && n_img * HW * C * 2 < 0x70000000LL;     // 32-bit byte offsets into the cached tensor
}
const char* launch_entry_reduce(fav_handle* h, const void* x, void* y, const void* wa, const float* bias_a, void* t1, int C, int nred,
                                int HW, int n_out, const DropParams& dp, hipStream_t s) {
    if (!entry_reduce_supported(C, nred, dp.n_img, HW)) return "entry reduce: unsupported shape";
    if (dp.site < 0 || n_out < 1 || dp.v0 < 0 || dp.v0 + n_out > 0x7fffffffLL) return "entry reduce: bad dropout descriptor";
    EntryReduceParams p;
    p.x = (const uint16_t*)x; p.y = (uint16_t*)y; p.wa = (const uint16_t*)wa; p.bias_a = bias_a; p.t1 = (uint16_t*)t1;
    p.HW = HW; p.M = (int)((long long)dp.n_img * HW); p.n_out = n_out;
    p.drop = dp;
    p.div_hw = fastdiv_make((uint32_t)HW);
    const long long cached = std::min<long long>(dp.n_img, n_out);
    const double rows = (double)n_out * HW;
    Prof pr(h, s, FAV_K_CONV, 2.0 * rows * C * nred, 2.0 * ((double)cached * HW * C + rows * ((y ? C : 0) + nred) + (double)C * nred));
    hipLaunchKernelGGL((entry_reduce_kernel<256, 64>), dim3((unsigned)((p.M + 127) / 128)), dim3(256), 0, s, p);
    return nullptr;
}

const char* launch_head(fav_handle* h, const float* logits, int T, int n, int C, int ld, float temperature, int kind,
                        float tau, int* labels, float* conf, uint8_t* fail, float* score, hipStream_t s, int out_stride = 1) {
    if (C > 1024 || C < 1) return "head: num_classes must be in [1, 1024]";
    if (ld % 4 != 0 || ld < C) return "head: bad row stride";
    const float inv_temp = 1.0f / temperature;
    const float inv_lnC = C > 1 ? (float)(1.0 / std::log((double)C)) : 0.f;
    Prof pr(h, s, FAV_K_HEAD, 0.0, 4.0 * (double)T * n * C + 8.0 * n);
    if (C <= 256)
        hipLaunchKernelGGL((head_kernel<1>), dim3(n), dim3(256), 0, s, logits, T, n, C, ld, inv_temp, kind, tau, inv_lnC,
                           labels, conf, fail, score, out_stride);
    else
        hipLaunchKernelGGL((head_kernel<4>), dim3(n), dim3(256), 0, s, logits, T, n, C, ld, inv_temp, kind, tau, inv_lnC,
                           labels, conf, fail, score, out_stride);
    return nullptr;
}

// ------------------------------------------------------------------ graph build
// ------------------------------------------------------------------ ViT launchers
const char* launch_layernorm(fav_handle* h, const void* x, long long ldx, const float* gamma, const float* beta, void* y, long long rows,
                             int D, float eps, hipStream_t s) {
    if (D % 4 != 0 || D > 1024 || rows < 1) return "layernorm: need D % 4 == 0, D <= 1024";
    Prof pr(h, s, FAV_K_AVGPOOL, 0.0, (double)rows * D * 4);
    const int lnr = (int)FAV_KNOB("FAV_LN_ROWS", 4);    // rows per wave (experiments build: 1 / 2 / 4)
    if (rows >= 4096 && lnr >= 4)
        hipLaunchKernelGGL(layernorm_kernel<4>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, (const uint16_t*)x, ldx, gamma, beta,
                           (uint16_t*)y, rows, D, eps);
    else if (rows >= 4096 && lnr >= 2)
        hipLaunchKernelGGL(layernorm_kernel<2>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, s, (const uint16_t*)x, ldx, gamma, beta,
                           (uint16_t*)y, rows, D, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const uint16_t*)x, ldx, gamma, beta,
                           (uint16_t*)y, rows, D, eps);
    return nullptr;
}

// ---- chained stream-K GEMM (gemm_streamk_kernel): the ViT encoder's linear layers -----------------------------------------------
// Returns false when the shape is not the kernel's (the caller then takes the tile-per-block kernel: same bits).
// The grid is k x CUs workgroups, k = 3, 2 or 1: the largest for which every XCD group's share of the tiles is >= its workgroups,
// i.e. every workgroup's share of K steps is at least one whole tile (a workgroup then publishes at most one partial accumulator,
// at its start, and takes over at most one, at its end).
struct SkDevice { fav_handle::SkWs ws; int device = -1; };
bool sk_prepare(fav_handle::SkWs* w, int grid) {
    if (w->grid_cap >= grid) return true;
    if (w->ws) { (void)hipFree(w->ws); (void)hipFree(w->flags); (void)hipFree(w->err); w->ws = nullptr; }
    const int cap = std::max(grid, 768);
    if (hipMalloc((void**)&w->ws, (size_t)cap * 65536) != hipSuccess || hipMalloc((void**)&w->flags, (size_t)cap * 4) != hipSuccess ||
        hipMalloc((void**)&w->err, 4) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemset(w->flags, 0, (size_t)cap * 4) != hipSuccess || hipMemset(w->err, 0, 4) != hipSuccess) return false;
    w->grid_cap = cap;
    w->epoch = 0;
    return true;
}
// Off by default: bit-identical to the tile-per-block kernel but slower at the encoder's shapes on this part (profiles/r4g_streamk_*:
// the hand-offs' agent-scope fences and 2 x 64 KB per workgroup cost ~25 % of a launch, and shares that start at different K steps
// lose the lock step that keeps the tile kernel's LDS-DMA stream in the XCD's 4 MB L2).  FAV_STREAMK=1 routes the encoder through it.
bool streamk_enabled() {
    return FAV_KNOB("FAV_STREAMK", 0) != 0;
}
bool launch_gemm_streamk(fav_handle* h, const void* a, const void* w, const float* bias, const void* res, void* y, long long M, int K, int N,
                         int act, hipStream_t s) {
    if ((h && !streamk_enabled()) || M < 1 || K % 32 != 0 || K < 128 || N % 128 != 0 || act < 0 || act > 2) return false;
    const long long tiles_m = (M + 127) / 128;
    const long long tiles = tiles_m * (N / 128);
    if ((double)(M + 128) * std::max(N, K) * 2.0 >= 2147483647.0 || (double)N * K * 2.0 >= 2147483647.0 || tiles > 0x3fffffffLL) return false;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false; n_cu = prop.multiProcessorCount; }
    int per_cu = 0;
    for (int k = 3; k >= 1; --k)
        if (tiles / 8 >= (long long)(n_cu * k) / 8) { per_cu = k; break; }
    if (!per_cu || (n_cu * per_cu) % 8 != 0) return false;
    const int grid = n_cu * per_cu;
    static SkDevice op_ws[16];               // op-level calls (no handle): one workspace per device, kept for the life of the process
    fav_handle::SkWs* W;
    if (h) W = &h->sk[h->sk_slot];
    else { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false; W = &op_ws[dev].ws; }
    if (!sk_prepare(W, grid)) return false;
    GemmSkParams p;
    memset(&p, 0, sizeof p);
    p.a = (const uint16_t*)a; p.w = (const uint16_t*)w; p.bias = bias; p.res = (const uint16_t*)res; p.y = (uint16_t*)y;
    p.M = (int)M; p.N = N; p.K = K; p.ksteps = K / 32;
    p.tiles_n = N / 128; p.tiles = (int)tiles;
    p.act = act;
    if (FAV_KNOB("FAV_SK_NOHANDOFF", 0) != 0) p.act |= 0x100;   // experiments build, TIMING ONLY (wrong results): what the hand-offs cost
    p.Q = grid / 8;
    p.ws = W->ws; p.flags = W->flags; p.err = W->err;
    p.epoch = ++W->epoch;
    if (p.epoch == 0) { (void)hipMemsetAsync(W->flags, 0, (size_t)W->grid_cap * 4, s); p.epoch = W->epoch = 1; }
    p.div_tn = fastdiv_make((uint32_t)p.tiles_n);
    const double flops = 2.0 * (double)M * N * K;
    const double bytes = 2.0 * ((double)M * K + (double)M * N * (res ? 2 : 1) + (double)N * K);
    Prof pr(h, s, FAV_K_CONV, flops, bytes);
    // dynamic LDS the kernel never touches: it makes per_cu workgroups - not more - fit a CU beside the kernel's own 53 248 B, so the
    // smaller grids sit two / one per CU instead of three on some CUs and none on others
    const int pad_lds = per_cu == 3 ? 0 : (per_cu == 2 ? 26 * 1024 : 104 * 1024);
    static DeviceFlags attr_set;
    if (!attr_set.test_current()) {
        if (hipFuncSetAttribute((const void*)gemm_streamk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024) != hipSuccess) { (void)hipGetLastError(); return false; }
        attr_set.set_current();
    }
    const bool dbg_on = FAV_KNOB("FAV_SK_DBG", 0) != 0;   // experiments build only: where a step's ticks go
    if (dbg_on && !h) { (void)hipMalloc((void**)&p.dbg, (size_t)grid * 64); (void)hipMemset(p.dbg, 0, (size_t)grid * 64); }
    hipLaunchKernelGGL(gemm_streamk_kernel, dim3((unsigned)grid), dim3(256), pad_lds, s, p);
    if (p.dbg) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> t((size_t)grid * 8);
        (void)hipMemcpy(t.data(), p.dbg, t.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(p.dbg);
        double ph[5] = {0, 0, 0, 0, 0}, steps = 0;
        unsigned long long lo = ~0ull, hi = 0; double life = 0;
        for (int i = 0; i < grid; ++i) {
            for (int k = 0; k < 5; ++k) ph[k] += (double)t[i * 8 + k];
            steps += (double)t[i * 8 + 5];
            lo = std::min(lo, t[i * 8 + 6]); hi = std::max(hi, t[i * 8 + 7]); life += (double)(t[i * 8 + 7] - t[i * 8 + 6]);
        }
        fprintf(stderr, "[sk dbg] span %.1f us, mean workgroup life %.1f us, resident workgroups per CU %.2f\n", (hi - lo) / 100.0, life / grid / 100.0, life / (double)(hi - lo) / n_cu);
        fprintf(stderr, "[sk dbg] M %lld K %d N %d grid %d: %.1f steps per workgroup; ticks per step (wave 0): stage %.0f, reads + MFMAs %.0f, segment end %.0f, "
                        "vmcnt wait %.0f, barrier %.0f\n", M, K, N, grid, steps / grid, ph[0] / steps, ph[1] / steps, ph[2] / steps, ph[3] / steps, ph[4] / steps);
    }
    return true;
}

const char* launch_attention(fav_handle* h, const void* qkv, void* out, int n, int T, int D, int heads, int math_mode, hipStream_t s) {
    if (T < 1 || T > 256 || heads * 64 != D || n < 1) return "attention: need 1 <= tokens <= 256 and 64-wide heads";
    const int nkt = (T + 15) / 16, Tp2 = (T + 31) / 32 * 32;
    // as few rounds of query tiles as 8 waves allow, then as few waves as those rounds need (197 tokens: 13 tiles = 2 rounds of 7
    // waves); K and V are all the LDS a block holds (55 KB), so two blocks share a CU
    const int rounds = (nkt + 7) / 8;
    int nw = (nkt + rounds - 1) / rounds;
    const int attn_nw = (int)FAV_KNOB("FAV_ATTN_WAVES", 0);   // experiments build: another block shape
    if (attn_nw >= 1 && attn_nw <= 8) nw = attn_nw;
    const int lds = nkt * 16 * 128 + Tp2 * 128;
    static DeviceFlags attr_set;
    if (!attr_set.test_current()) {
        if (hipFuncSetAttribute((const void*)attention_kernel<0, 13, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)attention_kernel<0, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)attention_kernel<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)attention_kernel<1, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)attention_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return "attention: cannot reserve LDS";
        attr_set.set_current();
    }
    const double flops = 4.0 * n * heads * (double)T * T * 64;
    Prof pr(h, s, FAV_K_CONV, flops, (double)n * T * D * 2 * 4);
    const dim3 grid((unsigned)(n * heads)), block(nw * 64);
#define FAV_ATTN(MODE_, ...) hipLaunchKernelGGL((attention_kernel<MODE_, __VA_ARGS__>), grid, block, lds, s, (const uint16_t*)qkv, (uint16_t*)out, T, D, heads)
    if (math_mode == FAV_MATH_BF16) { if (nkt == 13) FAV_ATTN(0, 13, true); else if (nkt < 13) FAV_ATTN(0, 13); else FAV_ATTN(0, 16); }
    else { if (nkt <= 13) FAV_ATTN(1, 13); else FAV_ATTN(1, 16); }
#undef FAV_ATTN
    return nullptr;
}

void launch_vit_assemble(fav_handle* h, const void* emb, const float* pos, void* x, int n, int ntok, int D, hipStream_t s) {
    const long long total = (long long)n * ntok * (D / 4);
    Prof pr(h, s, FAV_K_STEM, 0.0, (double)n * ntok * D * 4);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)emb, pos, (uint16_t*)x, n, ntok, D);
}

// A deep ensemble runs every op as ONE launch over its members (classify_on_stream) when: there is no MC-Dropout suffix, the
// production math mode, the one-launch stem, the members' workspaces side by side, fav_config.ens_grouped_max >= 0 - and the call
// has at most ens_grouped_max frames.  build_graph plans the size-dependent tail kernels of layers 3-4 for max_batch frames, so it
// asks with the same predicate whether a call of max_batch frames will be grouped.
bool will_group(const fav_handle* h, bool has_mc, bool stem_fused) {
    const fav_config& c = h->cfg;
    if (h->n_members <= 1 || has_mc || c.math_mode != FAV_MATH_BF16 || !stem_fused || c.ens_grouped_max < 0) return false;
    if (FAV_KNOB("FAV_ENS_STREAMS", 1) == 0) return false;
    return c.ens_grouped_max == 0 || c.max_batch <= c.ens_grouped_max;
}

fav_status build_graph(fav_handle* h) {
    const fav_config& c = h->cfg;
    const ArchDef& A = kArch[c.arch];
    h->layers.clear();
    h->ops.clear();
    auto add_layer = [&](int cout, int cin, int kh, int kw, int stride, int pad) {
        Layer L;
        L.cout = cout; L.cin = cin; L.kh = kh; L.kw = kw; L.stride = stride; L.pad = pad;
        L.cout_pad = (cout + 63) / 64 * 64;
        L.k = kh * kw * cin;
        h->layers.push_back(L);
        return (int)h->layers.size() - 1;
    };
    int H = c.in_h, W = c.in_w;
    // stem: im2col (normalise fused) + dense GEMM over the padded patch matrix
    int sk = A.imagenet_stem ? 7 : 3, ss = A.imagenet_stem ? 2 : 1, sp = A.imagenet_stem ? 3 : 1;
    int li = add_layer(64, 3, sk, sk, ss, sp);
    {
        Layer& L = h->layers[li];
        L.k = (sk * sk * 3 + 63) / 64 * 64;  // device K is the padded patch length
    }
    int Ho = conv_out(H, sk, ss, sp), Wo = conv_out(W, sk, ss, sp);
    if (Ho < 1 || Wo < 1) { h->err = "input too small"; return FAV_ERR_INVALID_ARG; }
    int cur = 0;  // rotating buffer holding the current activation
    auto other = [&](std::initializer_list<int> used) {
        for (int i = 0; i < 5; ++i) {
            bool u = false;
            for (int x : used) u |= (x == i);
            if (!u) return i;
        }
        return -1;
    };
    // production math mode: the whole ImageNet stem (normalise, 7x7/2, ReLU, max pool) is one launch
    const bool stem_fused = A.imagenet_stem && c.math_mode == FAV_MATH_BF16 && c.stem_fused >= 0 && !h->plan_no_fuse &&
                            conv_out(Ho, 3, 2, 1) >= 1 && conv_out(Wo, 3, 2, 1) >= 1;
    int stem_ops = 0;
    if (stem_fused) {
        Op o; o.kind = OP_STEM_POOL; o.layer = li; o.in = B_INPUT; o.out = cur; o.relu = 1;
        o.H = H; o.W = W; o.C = 3; o.Ho = conv_out(Ho, 3, 2, 1); o.Wo = conv_out(Wo, 3, 2, 1); o.Co = 64;
        o.in_elems = (long long)H * W * 3; o.out_elems = (long long)o.Ho * o.Wo * 64;
        h->ops.push_back(o);
        H = o.Ho; W = o.Wo;
        stem_ops = 1;
    } else {
        Op o; o.kind = OP_STEM_IM2COL; o.layer = li; o.in = B_INPUT; o.out = B_A1;
        o.H = H; o.W = W; o.C = 3; o.Ho = Ho; o.Wo = Wo; o.Co = h->layers[li].k;
        o.in_elems = (long long)H * W * 3; o.out_elems = (long long)Ho * Wo * o.Co;
        h->ops.push_back(o);
        Op g; g.kind = OP_CONV; g.layer = li; g.in = B_A1; g.out = cur; g.relu = 1;
        g.H = Ho; g.W = Wo; g.C = h->layers[li].k; g.Ho = Ho; g.Wo = Wo; g.Co = 64;
        g.in_elems = o.out_elems; g.out_elems = (long long)Ho * Wo * 64;
        h->ops.push_back(g);
        H = Ho; W = Wo;
        stem_ops = 2;
    }
    int C = 64;
    if (A.imagenet_stem && !stem_fused) {
        ++stem_ops;
        Op o; o.kind = OP_MAXPOOL; o.in = cur; o.out = other({cur});
        o.H = H; o.W = W; o.C = C; o.Ho = conv_out(H, 3, 2, 1); o.Wo = conv_out(W, 3, 2, 1); o.Co = C;
        o.in_elems = (long long)H * W * C; o.out_elems = (long long)o.Ho * o.Wo * C;
        h->ops.push_back(o);
        cur = o.out; H = o.Ho; W = o.Wo;
    }
    std::vector<int> block_last_op;
    const int exp = A.bottleneck ? 4 : 1;
    int inpl = 64, bidx = 0;
    // phase boundaries in block units (known before the ops exist; the fused tails must not straddle them)
    const int nblocks_total = A.depths[0] + A.depths[1] + A.depths[2] + A.depths[3];
    int mc_first_site = -1;
    {
        const uint32_t thr0 = (uint32_t)std::lround((double)c.dropout_p * 256.0);
        if (c.site_mask != 0 && thr0 > 0)
            for (int sb = 0; sb <= nblocks_total; ++sb)
                if (c.site_mask >> sb & 1) { mc_first_site = sb; break; }
    }
    int regroup_blk = c.regroup_block;
    if (regroup_blk < 0) regroup_blk = A.depths[0] + A.depths[1];
    regroup_blk = std::min(regroup_blk, nblocks_total);
    int pre_t1 = -1;   // rotating buffer already holding the coming block's conv1 output
    for (int st = 0; st < 4; ++st) {
        for (int bi = 0; bi < A.depths[st]; ++bi, ++bidx) {
            const int pl = A.planes[st];
            const int s = (bi == 0 && st > 0) ? 2 : 1;
            const bool ds = (bi == 0) && (s != 1 || inpl != pl * exp);
            const int xin = cur;
            int t1 = pre_t1 >= 0 ? pre_t1 : other({xin}), t2 = other({xin, t1});
            auto conv_op = [&](int layer, int in, int out, int res, int relu, int Hi, int Wi) {
                const Layer& L = h->layers[layer];
                Op o; o.kind = OP_CONV; o.layer = layer; o.in = in; o.out = out; o.res = res; o.relu = relu;
                o.H = Hi; o.W = Wi; o.C = L.cin;
                o.Ho = conv_out(Hi, L.kh, L.stride, L.pad); o.Wo = conv_out(Wi, L.kw, L.stride, L.pad); o.Co = L.cout;
                o.in_elems = (long long)Hi * Wi * L.cin; o.out_elems = (long long)o.Ho * o.Wo * L.cout;
                h->ops.push_back(o);
                return o;
            };
            int Hn, Wn;
            if (A.bottleneck) {
                int l1 = add_layer(pl, inpl, 1, 1, 1, 0), l2 = add_layer(pl, pl, 3, 3, s, 1), l3 = add_layer(pl * 4, pl, 1, 1, 1, 0);
                if (pre_t1 < 0) conv_op(l1, xin, t1, B_NONE, 1, H, W);   // else: written by the previous block's fused tail
                pre_t1 = -1;
                Hn = conv_out(H, 3, s, 1); Wn = conv_out(W, 3, s, 1);
                int ld = -1;
                if (ds) ld = add_layer(pl * exp, inpl, 1, 1, s, 0);
                // Fused tail (bottleneck_tail_kernel): conv2 (when 3x3/1) + conv3 (+ the NEXT block's conv1 unless a
                // phase boundary or the end of the network lies between the two blocks).  Production math mode only.
                const bool last_block = (st == 3 && bi + 1 == A.depths[3]);
                const bool boundary_after = last_block || (mc_first_site == bidx) || (bidx + 1 == regroup_blk);
                const int next_pl = (bi + 1 < A.depths[st]) ? pl : (st < 3 ? A.planes[st + 1] : 0);
                int nred = boundary_after ? 0 : next_pl;
                TailGeom tg;
                // the 256-pixel / 8-wave kernels of layers 3-4 run one block per CU: they pay only when the planned launch
                // (max_batch frames, x T samples behind the first dropout site) brings two blocks per CU
                // (an ensemble without MC-Dropout runs every op as one launch over its members, see classify_on_stream)
                const int plan_groups = will_group(h, mc_first_site >= 0, stem_fused) ? h->n_members : 1;
                const long long plan_rows = (long long)c.max_batch * ((mc_first_site >= 0 && bidx > mc_first_site) ? c.n_samples : 1) * Hn * Wn * plan_groups;
                const bool big_launch = c.tail_min_rows < 0 || plan_rows >= (c.tail_min_rows > 0 ? (long long)c.tail_min_rows : 512ll * 256);
                const bool tail_3x3 = (s == 1) && (pl <= 128 || (pl == 256 && tail_wide3x3() && big_launch));
                if (pl > 128) nred = 0;   // wide blocks: the expanding 1x1 alone (with the next block's reduce in the launch it measured 3.01 ms against 1.77 + 0.95: only fav_op_bottleneck_tail still reaches that kernel)
                bool fuse = tail_enabled() && !h->plan_no_fuse && c.math_mode == FAV_MATH_BF16 && (pl == 64 || pl == 128 || (pl == 256 && tail_wide()) || (pl == 512 && tail_l4() && big_launch));
                if (fuse && !tail_geometry(pl, nred, tail_3x3, Wn, &tg)) {
                    nred = 0;
                    fuse = tail_geometry(pl, 0, tail_3x3, Wn, &tg);
                }
                int t2v = t1;
                if (!(fuse && tail_3x3)) {                          // the 3x3 as its own launch
                    conv_op(l2, t1, t2, B_NONE, 1, H, W);
                    t2v = t2;
                }
                int idn = xin;
                // blob order is conv1, conv2, conv3, downsample; launch order: downsample before conv3
                if (ds) { idn = other({xin, t1, t2}); conv_op(ld, xin, idn, B_NONE, 0, H, W); }
                if (fuse) {
                    const int yout = other({xin, t1, t2v, idn});    // t2 is free when the 3x3 is fused, xin when it is not the residual
                    Op o; o.kind = OP_TAIL; o.layer = tail_3x3 ? l2 : -1; o.layer_c = l3;
                    o.layer_a = nred > 0 ? (int)h->layers.size() : -1;      // the next add_layer() is the next block's conv1
                    o.in = t2v; o.res = idn; o.out = yout; o.relu = 1;
                    o.H = Hn; o.W = Wn; o.C = pl; o.Ho = Hn; o.Wo = Wn; o.Co = pl * 4; o.Co2 = nred;
                    o.in_elems = (long long)Hn * Wn * pl; o.out_elems = (long long)Hn * Wn * pl * 4;
                    if (nred > 0) { o.out2 = other({t2v, idn, yout}); pre_t1 = o.out2; }
                    h->ops.push_back(o);
                    cur = yout;
                } else {
                    int yout = ds ? other({xin, t1, t2, idn}) : t1;  // t1 is dead after conv2
                    conv_op(l3, t2, yout, idn, 1, Hn, Wn);
                    cur = yout;
                }
            } else {
                int l1 = add_layer(pl, inpl, 3, 3, s, 1), l2 = add_layer(pl, pl, 3, 3, 1, 1);
                Op o1 = conv_op(l1, xin, t1, B_NONE, 1, H, W);
                Hn = o1.Ho; Wn = o1.Wo;
                int idn = xin, ld = -1;
                if (ds) ld = add_layer(pl * exp, inpl, 1, 1, s, 0);
                if (ds) { idn = t2; conv_op(ld, xin, idn, B_NONE, 0, H, W); }
                int yout = other({xin, t1, idn});
                conv_op(l2, t1, yout, idn, 1, Hn, Wn);
                cur = yout;
            }
            block_last_op.push_back((int)h->ops.size() - 1);
            H = Hn; W = Wn; inpl = pl * exp; C = inpl;
        }
    }
    h->nblocks = bidx;
    {
        Op o; o.kind = OP_AVGPOOL; o.in = cur; o.out = other({cur});
        o.H = H; o.W = W; o.C = C; o.Ho = 1; o.Wo = 1; o.Co = C;
        o.in_elems = (long long)H * W * C; o.out_elems = C;
        h->ops.push_back(o);
        cur = o.out;
    }
    const int pool_op = (int)h->ops.size() - 1;
    int lfc = add_layer(c.num_classes, C, 1, 1, 1, 0);
    h->cpad = h->layers[lfc].cout_pad;
    {
        Op o; o.kind = OP_CONV; o.layer = lfc; o.in = cur; o.out = other({cur}); o.out_f32 = 1;
        o.H = 1; o.W = 1; o.C = C; o.Ho = 1; o.Wo = 1; o.Co = c.num_classes;
        o.in_elems = C; o.out_elems = h->cpad;
        h->ops.push_back(o);
    }

    // ---- dropout sites and prefix / suffix split -----------------------------
    const uint32_t valid_mask = (h->nblocks + 1 >= 32) ? 0xFFFFFFFFu : ((1u << (h->nblocks + 1)) - 1);
    if (c.site_mask & ~valid_mask) { h->err = "site_mask has bits beyond the pooled-feature site"; return FAV_ERR_INVALID_ARG; }
    const uint32_t thr = (uint32_t)std::lround((double)c.dropout_p * 256.0);
    const bool mc = c.site_mask != 0 && thr > 0;
    h->T_eff = mc ? c.n_samples : 1;
    h->first_site = -1;
    int split = (int)h->ops.size();  // ops [0, split) are the prefix
    if (mc) {
        for (int s = 0; s <= h->nblocks; ++s)
            if (c.site_mask >> s & 1) { h->first_site = s; break; }
        const int first_op = h->first_site < h->nblocks ? block_last_op[h->first_site] : pool_op;
        split = first_op + 1;
        for (int s = h->first_site + 1; s <= h->nblocks; ++s)
            if (c.site_mask >> s & 1) h->ops[s < h->nblocks ? block_last_op[s] : pool_op].site = s;
    }
    // op index where the low-resolution group starts
    int regroup = c.regroup_block;
    if (regroup < 0) regroup = A.bottleneck ? A.depths[0] + A.depths[1] : A.depths[0] + A.depths[1];
    regroup = std::min(regroup, h->nblocks);
    const int regroup_op = regroup == 0 ? stem_ops
                                        : (regroup >= h->nblocks ? pool_op : block_last_op[regroup - 1] + 1);

    // ---- phases ---------------------------------------------------------------
    h->phases.clear();
    int regroup_now = regroup_op;  // op index of the low-resolution group in the CURRENT op list
    auto add_phase = [&](int b, int e, bool suffix) {
        if (b >= e) return;
        Phase p; p.op_begin = b; p.op_end = e; p.suffix = suffix;
        p.low_res = b >= regroup_now;
        p.in_elems = h->ops[b].in_elems;
        p.out_elems = h->ops[e - 1].out_elems;
        p.out_bytes_per_elem = h->ops[e - 1].out_f32 ? 4 : 2;
        p.chunk = 0;
        h->phases.push_back(p);
    };
    auto add_range = [&](int b, int e, bool suffix) {
        if (b < regroup_op && regroup_op < e) { add_phase(b, regroup_op, suffix); add_phase(regroup_op, e, suffix); }
        else add_phase(b, e, suffix);
    };
    add_range(0, split, false);
    if (mc) {
        // the suffix starts with the entry dropout of the cached prefix output
        Op ed; ed.kind = OP_ENTRY_DROPOUT; ed.in = B_PHASE_IN; ed.site = h->first_site;
        ed.in_elems = ed.out_elems = h->ops[split - 1].out_elems;
        ed.out = 0;  // patched below
        // insert before ops[split]; the op list after `split` reads its input from
        // whatever buffer the prefix's last op wrote, so write the dropout there.
        ed.out = h->ops[split - 1].out;
        h->ops.insert(h->ops.begin() + split, ed);
        regroup_now = regroup_op >= split ? regroup_op + 1 : regroup_op;
        // the 1x1 reduce that follows (the next block's conv1) joins the dropout launch when both lie in one phase
        if (split + 2 < (int)h->ops.size() && split + 1 != regroup_now && tail_enabled() && entry_reduce_enabled() && !h->plan_no_fuse &&
            c.math_mode == FAV_MATH_BF16) {
            const Op& cv = h->ops[split + 1];
            const Op& e0 = h->ops[split];
            if (cv.kind == OP_CONV && cv.in == e0.out && cv.res == B_NONE && cv.relu == 1 && !cv.out_f32 && cv.out != e0.out && cv.site < 0) {
                const Layer& L = h->layers[cv.layer];
                if (L.kh == 1 && L.kw == 1 && L.stride == 1 && L.pad == 0 && L.cout == L.cout_pad &&
                    entry_reduce_supported(L.cin, L.cout, c.max_batch, (long long)cv.H * cv.W)) {
                    Op f = e0;
                    f.kind = OP_ENTRY_REDUCE; f.layer_a = cv.layer; f.out2 = cv.out; f.Co2 = L.cout;
                    f.H = cv.H; f.W = cv.W; f.C = L.cin; f.relu = 1;
                    h->ops[split] = f;
                    h->ops.erase(h->ops.begin() + split + 1);
                    if (regroup_now > split + 1) --regroup_now;
                    // The block behind the entry: its tail can take its residual - the dropped copy of the cached prefix output -
                    // from the cached tensor itself and apply the entry mask in its epilogue; the T copies are then neither
                    // written (12 GB per step at the headline shape) nor read back.  FAV_ENTRY_RES=0 keeps them.
                    const bool entry_res = FAV_KNOB("FAV_ENTRY_RES", 1) != 0;
                    const int y0 = f.out;
                    if (entry_res && split + 1 < (int)h->ops.size() && regroup_now != split + 1) {
                        Op& tl = h->ops[split + 1];
                        bool ok = tl.kind == OP_TAIL && tl.res == y0 && tl.in == f.out2 && tl.layer >= 0 && tl.C == 64 && tl.Co2 == 64 &&
                                  tl.layer_a >= 0 && tl.site >= 0 && tl.out != y0 && tl.out2 != y0 &&
                                  (double)c.max_batch * f.H * f.W * f.C * 2.0 < 2147483647.0;
                        const int phase_end = regroup_now > split + 1 ? regroup_now : (int)h->ops.size();
                        for (int k = split + 2; ok && k < phase_end; ++k) {     // nobody else reads the copies before their buffer is reused
                            const Op& o = h->ops[k];
                            if (o.in == y0 || o.res == y0) ok = false;
                            if (o.out == y0 || o.out2 == y0) break;
                        }
                        if (ok) { tl.res_entry = 1; h->ops[split].skip_y = 1; }
                    }
                }
            }
        }
        const int nops = (int)h->ops.size();
        // split the suffix only if at least one real op lies on each side
        if (split + 1 < regroup_now && regroup_now < nops) { add_phase(split, regroup_now, true); add_phase(regroup_now, nops, true); }
        else add_phase(split, nops, true);
    }
    // mark phase boundaries in the ops' buffers
    for (size_t i = 0; i < h->phases.size(); ++i) {
        Phase& p = h->phases[i];
        Op& first = h->ops[p.op_begin];
        if (first.kind != OP_STEM_IM2COL && first.kind != OP_STEM_POOL) {
            // consumers of the phase input: every op in the phase reading the buffer the
            // previous phase's last op wrote, until that rotating buffer is overwritten
            const bool entry = first.kind == OP_ENTRY_DROPOUT || first.kind == OP_ENTRY_REDUCE;
            const int src = entry ? B_PHASE_IN : first.in;
            if (!entry) {
                for (int k = p.op_begin; k < p.op_end; ++k) {
                    Op& o = h->ops[k];
                    if (o.in == src) o.in = B_PHASE_IN;
                    if (o.res == src) o.res = B_PHASE_IN;
                    if (o.out == src || o.out2 == src) break;
                }
            }
        }
        h->ops[p.op_end - 1].out = B_PHASE_OUT;
    }
    return FAV_OK;
}

// ------------------------------------------------------------------ memory plan
fav_status plan_memory(fav_handle* h) {
    const fav_config& c = h->cfg;
    const long long nv_max = (long long)c.max_batch * h->T_eff;
    // chunk sizes
    long long max_elems = 1, max_a1 = 1;
    for (const Op& o : h->ops) {
        if (o.out == B_A1) max_a1 = std::max(max_a1, o.out_elems);
        else if (o.out >= 0) max_elems = std::max(max_elems, o.out_elems);
    }
    for (size_t i = 0; i < h->phases.size(); ++i) {
        Phase& p = h->phases[i];
        long long pe = 1;
        for (int k = p.op_begin; k < p.op_end; ++k) {
            const Op& o = h->ops[k];
            pe = std::max(pe, o.out == B_A1 ? o.out_elems / 2 : o.out_elems);
        }
        // Pass size: measured on MI355X, fewer and larger launches beat
        // keeping producer->consumer tensors inside the 256 MiB Infinity Cache at every
        // size tried (launch ramp/tail cost more than the HBM round trip), so by default a
        // phase runs all its frames in one pass, bounded by a 16 GiB-per-tensor arena
        // budget (5 rotating tensors; 288 GB of HBM makes that a non-issue).
        const long long target = 16ll << 30;
        long long auto_chunk = std::max<long long>(1, target / (pe * 2));
        const int want = p.low_res ? c.chunk_b : c.chunk_a;
        long long chunk = want > 0 ? want : auto_chunk;
        const long long dom = p.suffix ? nv_max : c.max_batch;
        p.chunk = (int)std::max<long long>(1, std::min(chunk, dom));
    }
    // pipeline the last two phases when they cover the same frames (both suffix, or both
    // prefix when there is no MC-Dropout): FAV_PIPE = number of chunks.  Default 1 = off:
    // measured on MI355X the two kernels only time-share the CUs (each already fills every
    // CU's LDS), 118.5 ms/step off vs 119.0-121.2 with 2..16 chunks.
    {
        const int npipe = (int)std::max<long long>(1, FAV_KNOB("FAV_PIPE", 1));
        const size_t np = h->phases.size();
        h->pipe_first = -1;
        if (npipe > 1 && np >= 2 && h->phases[np - 1].suffix == h->phases[np - 2].suffix) {
            const long long dom = h->phases[np - 1].suffix ? nv_max : c.max_batch;
            if (dom >= 2 * npipe) {
                long long step = (dom + npipe - 1) / npipe;
                if (c.chunk_a > 0) step = std::min<long long>(step, c.chunk_a);
                step = std::min<long long>(step, std::min(h->phases[np - 2].chunk, h->phases[np - 1].chunk));
                h->phases[np - 2].chunk = h->phases[np - 1].chunk = (int)step;
                h->pipe_first = (int)np - 2;
            }
        }
    }
    int max_chunk = 1;
    for (const Phase& p : h->phases) max_chunk = std::max(max_chunk, p.chunk);
    // rotating buffers sized for the largest (chunk x tensor) in any phase
    size_t act_bytes = 0, act2_bytes = 0, a1_bytes = 0;
    for (size_t pi = 0; pi < h->phases.size(); ++pi) {
        const Phase& p = h->phases[pi];
        const bool second = h->pipe_first >= 0 && (int)pi == h->pipe_first + 1;
        for (int k = p.op_begin; k < p.op_end; ++k) {
            const Op& o = h->ops[k];
            const size_t b = (size_t)o.out_elems * (o.out_f32 ? 4 : 2) * p.chunk;
            if (o.out == B_A1) a1_bytes = std::max(a1_bytes, b);
            else if (o.out >= 0) (second ? act2_bytes : act_bytes) = std::max(second ? act2_bytes : act_bytes, b);
        }
    }
    act_bytes = (act_bytes + 255) / 256 * 256 + 256;
    a1_bytes = (a1_bytes + 255) / 256 * 256 + 256;
    // deep ensemble with its members side by side: every workspace tensor is a slab of n_members equal parts (member 0's part
    // is the handle's own buffer), so that a grouped launch finds member g's tensors at a constant stride
    const int ens_streams = (int)FAV_KNOB("FAV_ENS_STREAMS", 1);
    const bool side_by_side = h->n_members > 1 && ens_streams && h->pipe_first < 0;
    const size_t parts = side_by_side ? (size_t)h->n_members : 1;
    for (int i = 0; i < 5; ++i) HIP_TRY(h, hipMalloc(&h->act[i], act_bytes * parts));
    h->act_bytes = act_bytes;
    if (h->pipe_first >= 0) {
        act2_bytes = (act2_bytes + 255) / 256 * 256 + 256;
        for (int i = 0; i < 5; ++i) HIP_TRY(h, hipMalloc(&h->act2[i], act2_bytes));
        h->act2_bytes = act2_bytes;
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream_a, hipStreamNonBlocking));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join_a, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join_b, hipEventDisableTiming));
    }
    HIP_TRY(h, hipMalloc(&h->a1, a1_bytes * parts));
    h->a1_bytes = a1_bytes;
    h->phase_out.assign(h->phases.size(), nullptr);
    std::vector<size_t> phase_bytes(h->phases.size(), 0);
    for (size_t i = 0; i + 1 < h->phases.size(); ++i) {
        const Phase& p = h->phases[i];
        const long long dom = p.suffix ? nv_max : c.max_batch;
        phase_bytes[i] = ((size_t)dom * p.out_elems * p.out_bytes_per_elem + 255) / 256 * 256 + 256;
        HIP_TRY(h, hipMalloc(&h->phase_out[i], phase_bytes[i] * parts));
    }
    HIP_TRY(h, hipMalloc((void**)&h->logits, (size_t)nv_max * h->n_members * h->cpad * 4 + 256));
    h->phase_out.back() = h->logits;
    if (side_by_side) {
        h->mws.resize(h->n_members);
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_members, hipEventDisableTiming));
        for (int m = 0; m < h->n_members; ++m) {
            fav_handle::MemberWs& w = h->mws[m];
            HIP_TRY(h, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
            HIP_TRY(h, hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
            w.phase_out.assign(h->phases.size(), nullptr);
            for (int i = 0; i < 5; ++i) w.act[i] = (char*)h->act[i] + (size_t)m * act_bytes;
            w.a1 = (char*)h->a1 + (size_t)m * a1_bytes;
            for (size_t i = 0; i + 1 < h->phases.size(); ++i) w.phase_out[i] = (char*)h->phase_out[i] + (size_t)m * phase_bytes[i];
        }
        // grouped launches: every op must be one the grouped kernels cover (no MC-Dropout suffix, fused stem)
        h->can_group = h->T_eff == 1 && c.math_mode == FAV_MATH_BF16;
        for (const Op& o : h->ops)
            if (o.kind != OP_STEM_POOL && o.kind != OP_CONV && o.kind != OP_TAIL && o.kind != OP_AVGPOOL) h->can_group = false;
        for (const Op& o : h->ops) if (o.site >= 0) h->can_group = false;
        if (c.ens_grouped_max < 0) h->can_group = false;      // (build_graph planned with will_group(): the same conditions, asked for max_batch frames)
        h->group_max_frames = c.ens_grouped_max > 0 ? c.ens_grouped_max : 0x7fffffff;
    }
    return FAV_OK;
}

// ------------------------------------------------------------------ execution
// member >= 0: that ensemble member's weights (instead of the handle's current L.w / L.b); [k_lo, k_hi): restrict the
// launches to these ops of the phase (-1: all) - the ensemble enqueues op by op across its members' streams
fav_status run_chunks(fav_handle* h, size_t pi, const void* images, int layout, int n, long long first_index,
                      hipStream_t s, long long v_begin, long long v_end, void** act_set, const fav_handle::MemberWs* ws = nullptr,
                      int member = -1, int k_lo = -1, int k_hi = -1, int group_n = 1) {
    const fav_config& c = h->cfg;
    const Phase& p = h->phases[pi];
    auto LW = [&](int li) -> void* { return member >= 0 ? h->layers[li].w_m[member] : h->layers[li].w; };
    auto LB = [&](int li) -> float* { return member >= 0 ? h->layers[li].b_m[member] : h->layers[li].b; };
    const int op_lo = k_lo >= 0 ? std::max(k_lo, p.op_begin) : p.op_begin, op_hi = k_hi >= 0 ? std::min(k_hi, p.op_end) : p.op_end;
    const long long dom = p.suffix ? (long long)n * h->T_eff : n;
    const std::vector<void*>& phase_out = ws ? ws->phase_out : h->phase_out;
    void* const a1 = ws ? ws->a1 : h->a1;
    const char* pin_base = pi == 0 ? (const char*)images : (const char*)phase_out[pi - 1];
    const int in_bpe = pi == 0 ? (layout == FAV_LAYOUT_NHWC_U8 ? 1 : 4) : h->phases[pi - 1].out_bytes_per_elem;
    // a suffix phase that follows the prefix reads frame (v % n); later phases read virtual frame v
    const bool in_is_virtual = pi > 0 && h->phases[pi - 1].suffix;
    char* pout_base = (char*)phase_out[pi];
    const uint32_t thr = (uint32_t)std::lround((double)c.dropout_p * 256.0);
    const float scale = thr > 0 ? (float)(1.0 / (1.0 - thr / 256.0)) : 1.0f;
    float istd[3] = {1.0f / c.stdev[0], 1.0f / c.stdev[1], 1.0f / c.stdev[2]};

    (void)dom;
    for (long long v0 = v_begin; v0 < v_end; v0 += p.chunk) {
        const int cn = (int)std::min<long long>(p.chunk, v_end - v0);
        auto buf = [&](int id, bool is_out, const Op& o) -> void* {
            switch (id) {
                case B_INPUT: return (void*)(pin_base + (size_t)v0 * o.in_elems * in_bpe);
                case B_PHASE_IN:
                    if (!in_is_virtual && p.suffix) return (void*)pin_base;  // entry dropout indexes v % n itself
                    return (void*)(pin_base + (size_t)v0 * p.in_elems * in_bpe);
                case B_PHASE_OUT: return (void*)(pout_base + (size_t)v0 * p.out_elems * p.out_bytes_per_elem);
                case B_A1: return a1;
                case B_NONE: return nullptr;
                default: return act_set[id];
            }
        };
        // grouped launch (ws = member 0's workspace): byte stride from member 0's tensor to member 1's, per buffer and per layer
        auto gbuf = [&](int id) -> long long {
            if (group_n <= 1) return 0;
            const fav_handle::MemberWs &w0 = h->mws[0], &w1 = h->mws[1];
            switch (id) {
                case B_INPUT: case B_NONE: return 0;
                case B_PHASE_IN: return pi == 0 ? 0 : (char*)w1.phase_out[pi - 1] - (char*)w0.phase_out[pi - 1];
                case B_PHASE_OUT: return (char*)w1.phase_out[pi] - (char*)w0.phase_out[pi];
                case B_A1: return (char*)w1.a1 - (char*)w0.a1;
                default: return (char*)w1.act[id] - (char*)w0.act[id];
            }
        };
        auto gw = [&](int li) -> long long { return group_n > 1 && li >= 0 ? (long long)h->layers[li].w_stride : 0; };
        auto gb = [&](int li) -> long long { return group_n > 1 && li >= 0 ? (long long)h->layers[li].b_stride : 0; };
        struct GroupReset { fav_handle* h; ~GroupReset() { h->grp = fav_handle::Group{}; } } group_reset{h};
        for (int k = op_lo; k < op_hi; ++k) {
            const Op& o = h->ops[k];
            h->cur_op = k;
            if (group_n > 1) {
                fav_handle::Group G;
                G.n = group_n;
                G.x = gbuf(o.in); G.res = gbuf(o.res); G.y = gbuf(o.out); G.y2 = gbuf(o.out2);
                if (o.kind == OP_TAIL) { G.w = gw(o.layer_c); G.b = gb(o.layer_c); G.wb = gw(o.layer); G.bb = gb(o.layer); G.wa = gw(o.layer_a); G.ba = gb(o.layer_a); }
                else { G.w = gw(o.layer); G.b = gb(o.layer); }
                h->grp = G;
            }
            fav_dropout_desc dd;
            dd.site = o.site; dd.threshold = thr; dd.scale = scale; dd.seed = c.seed;
            dd.v0 = p.suffix ? v0 : 0; dd.n_img = n; dd.first_image_index = first_index;
            switch (o.kind) {
                case OP_STEM_IM2COL: {
                    const Layer& L = h->layers[o.layer];
                    launch_stem(h, buf(o.in, false, o), layout, cn, o.H, o.W, L.kh, L.kw, L.stride, L.pad, L.k, c.mean,
                                istd, buf(o.out, true, o), s);
                    break;
                }
                case OP_CONV: {
                    const Layer& L = h->layers[o.layer];
                    fav_conv_desc d;
                    d.x = buf(o.in, false, o); d.w = LW(o.layer); d.bias = LB(o.layer); d.res = buf(o.res, false, o); d.y = buf(o.out, true, o);
                    d.n_frames = cn; d.H = o.H; d.W = o.W; d.Cin = o.C; d.Cout = L.cout;
                    // the stem GEMM runs as a 1x1 conv over the im2col matrix
                    const bool stem = (o.in == B_A1);
                    d.kh = stem ? 1 : L.kh; d.kw = stem ? 1 : L.kw; d.stride = stem ? 1 : L.stride; d.pad = stem ? 0 : L.pad;
                    d.relu = o.relu; d.out_f32 = o.out_f32; d.math_mode = c.math_mode;
                    d.drop = dd;
                    const int ldy = o.out_f32 ? L.cout_pad : L.cout;
                    if (const char* e = launch_conv(h, d, L.cout_pad, ldy, s)) { h->err = e; return FAV_ERR_INVALID_ARG; }
                    break;
                }
                case OP_TAIL: {
                    const Layer& Lc = h->layers[o.layer_c];
                    fav_tail_desc d;
                    memset(&d, 0, sizeof d);
                    d.x = buf(o.in, false, o);
                    (void)Lc;
                    if (o.layer >= 0) { d.wb = LW(o.layer); d.bias_b = LB(o.layer); }
                    d.wc = LW(o.layer_c); d.bias_c = LB(o.layer_c); d.res = buf(o.res, false, o); d.y = buf(o.out, true, o);
                    if (o.layer_a >= 0) { d.wa = LW(o.layer_a); d.bias_a = LB(o.layer_a); d.t1n = buf(o.out2, true, o); }
                    d.n_frames = cn; d.H = o.H; d.W = o.W; d.Cmid = o.C; d.Nred = o.Co2;
                    d.drop = dd;
                    if (o.res_entry) { d.res = pin_base; d.res_entry = 1; d.entry_site = h->first_site; }   // the cached prefix output
                    if (const char* e = launch_tail(h, d, s)) { h->err = e; return FAV_ERR_INVALID_ARG; }
                    break;
                }
                case OP_STEM_POOL:
                    if (const char* e = launch_stem_pool(h, buf(o.in, false, o), layout, cn, o.H, o.W, LW(o.layer), LB(o.layer), c.mean, istd,
                                                         buf(o.out, true, o), s)) { h->err = e; return FAV_ERR_INVALID_ARG; }
                    break;
                case OP_MAXPOOL:
                    launch_maxpool(h, buf(o.in, false, o), buf(o.out, true, o), cn, o.H, o.W, o.C, s);
                    break;
                case OP_AVGPOOL: {
                    DropParams dp = make_drop(&dd);
                    launch_avgpool(h, buf(o.in, false, o), buf(o.out, true, o), cn, o.H * o.W, o.C, dp, s);
                    break;
                }
                case OP_ENTRY_DROPOUT: {
                    DropParams dp = make_drop(&dd);
                    launch_entry_dropout(h, pin_base, buf(o.out, true, o), o.in_elems, cn, dp, s);
                    break;
                }
                case OP_ENTRY_REDUCE: {
                    DropParams dp = make_drop(&dd);
                    if (const char* e = launch_entry_reduce(h, pin_base, o.skip_y ? nullptr : buf(o.out, true, o), LW(o.layer_a), LB(o.layer_a), buf(o.out2, true, o), o.C, o.Co2,
                                                            o.H * o.W, cn, dp, s)) { h->err = e; return FAV_ERR_INVALID_ARG; }
                    break;
                }
            }
        }
    }
    HIP_TRY(h, hipGetLastError());
    return FAV_OK;
}

void free_all(fav_handle* h) {
    for (auto& L : h->layers) {
        if (L.w_slab) (void)hipFree(L.w_slab);
        if (L.b_slab) (void)hipFree(L.b_slab);
        L.w_slab = L.b_slab = nullptr;
        L.w_m.clear(); L.b_m.clear(); L.w = nullptr; L.b = nullptr;
    }
    for (int i = 0; i < 5; ++i) if (h->act[i]) (void)hipFree(h->act[i]);
    for (int i = 0; i < 5; ++i) if (h->act2[i]) (void)hipFree(h->act2[i]);
    if (h->stream_a) (void)hipStreamDestroy(h->stream_a);
    if (h->stream_b) (void)hipStreamDestroy(h->stream_b);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join_a) (void)hipEventDestroy(h->ev_join_a);
    if (h->ev_join_b) (void)hipEventDestroy(h->ev_join_b);
    for (auto e : h->ev_chunk) (void)hipEventDestroy(e);
    if (h->a1) (void)hipFree(h->a1);
    for (size_t m = 0; m < h->mws.size(); ++m) {
        fav_handle::MemberWs& w = h->mws[m];
        if (w.stream) (void)hipStreamDestroy(w.stream);
        if (w.done) (void)hipEventDestroy(w.done);
    }
    if (h->ev_members) (void)hipEventDestroy(h->ev_members);
    for (auto& w_ : h->sk) if (w_.ws) { (void)hipFree(w_.ws); (void)hipFree(w_.flags); (void)hipFree(w_.err); }
    for (auto st_ : h->vit_streams) (void)hipStreamDestroy(st_);
    for (auto ev_ : h->vit_done) (void)hipEventDestroy(ev_);
    for (size_t i = 0; i + 1 < h->phase_out.size(); ++i) if (h->phase_out[i]) (void)hipFree(h->phase_out[i]);
    if (h->logits) (void)hipFree(h->logits);
    if (h->host_stage) (void)hipFree(h->host_stage);
    if (h->host_stream) (void)hipStreamDestroy(h->host_stream);
    if (h->ev_last) (void)hipEventDestroy(h->ev_last);
    for (void* q : {h->v_patches, h->v_emb, h->v_x, h->v_y, h->v_qkv, h->v_hid, h->v_cls}) if (q) (void)hipFree(q);
    for (auto& e : h->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
}


// ViT: layers in blob order, fixed buffers, no phases (single deterministic pass).
fav_status build_vit(fav_handle* h) {
    const fav_config& c = h->cfg;
    const VitDef& V = kVit[c.arch - 2];
    if (c.in_h % V.patch || c.in_w % V.patch) { h->err = "ViT input must be a multiple of the patch size"; return FAV_ERR_INVALID_ARG; }
    const int np = (c.in_h / V.patch) * (c.in_w / V.patch), ntok = np + 1;
    if (ntok > 256) { h->err = "ViT path supports at most 256 tokens"; return FAV_ERR_UNSUPPORTED; }
    if (c.site_mask != 0 && c.dropout_p > 0.f) { h->err = "the ViT path has no dropout sites"; return FAV_ERR_UNSUPPORTED; }
    if (c.n_members > 1) { h->err = "the ViT path has no ensemble mode"; return FAV_ERR_UNSUPPORTED; }
    h->vit = true;
    h->vit_ntok = ntok;
    h->T_eff = 1;
    h->layers.clear();
    auto lin = [&](int cout, int cin, int kh, int kw, int stride) {
        Layer L; L.cout = cout; L.cin = cin; L.kh = kh; L.kw = kw; L.stride = stride; L.pad = 0;
        L.cout_pad = (cout + 63) / 64 * 64; L.k = kh * kw * cin;
        h->layers.push_back(L);
    };
    auto vec = [&](int len) {
        Layer L; L.cout = len; L.cin = 0; L.kh = 0; L.kw = 0; L.stride = 0; L.pad = 0; L.cout_pad = len; L.k = 0;
        h->layers.push_back(L);
    };
    lin(V.dim, 3, V.patch, V.patch, V.patch);
    vec(ntok * V.dim);
    for (int i = 0; i < V.depth; ++i) {
        vec(V.dim); lin(3 * V.dim, V.dim, 1, 1, 1); lin(V.dim, V.dim, 1, 1, 1);
        vec(V.dim); lin(V.mlp, V.dim, 1, 1, 1); lin(V.dim, V.mlp, 1, 1, 1);
    }
    vec(V.dim);
    lin(c.num_classes, V.dim, 1, 1, 1);
    h->cpad = h->layers.back().cout_pad;
    HIP_TRY(h, hipSetDevice(c.device));
    const size_t B = (size_t)c.max_batch, D = (size_t)V.dim;
    HIP_TRY(h, hipMalloc(&h->v_patches, B * np * (size_t)(V.patch * V.patch * 3) * 2 + 256));
    HIP_TRY(h, hipMalloc(&h->v_emb, B * np * D * 2 + 256));
    HIP_TRY(h, hipMalloc(&h->v_x, B * ntok * D * 2 + 256));
    HIP_TRY(h, hipMalloc(&h->v_y, B * ntok * D * 2 + 256));
    HIP_TRY(h, hipMalloc(&h->v_qkv, B * ntok * 3 * D * 2 + 256));
    HIP_TRY(h, hipMalloc(&h->v_hid, B * ntok * (size_t)V.mlp * 2 + 256));
    HIP_TRY(h, hipMalloc(&h->v_cls, B * D * 2 + 256));
    HIP_TRY(h, hipMalloc((void**)&h->logits, B * h->cpad * 4 + 256));
    return FAV_OK;
}

// One forward pass of the ViT encoder over n frames -> fp32 logits [n][cpad].
// Frames [f0, f0 + n) of the call: every buffer is indexed by frame, so two halves of a batch can run side by side
// on two streams (fav_classify_ex).
fav_status run_vit(fav_handle* h, const void* images_all, int layout, int f0, int n, hipStream_t s) {
    const fav_config& c = h->cfg;
    const VitDef& V = kVit[c.arch - 2];
    const int ntok = h->vit_ntok, D = V.dim, gh = c.in_h / V.patch, gw = c.in_w / V.patch;
    const float inv_std[3] = {1.0f / c.stdev[0], 1.0f / c.stdev[1], 1.0f / c.stdev[2]};
    const size_t F = (size_t)f0, np = (size_t)gh * gw;
    const void* images = (const char*)images_all + F * c.in_h * c.in_w * 3 * (layout == FAV_LAYOUT_NHWC_U8 ? 1 : 4);
    struct { void *v_patches, *v_emb, *v_x, *v_y, *v_qkv, *v_hid, *v_cls; float* logits; } B;
    B.v_patches = (char*)h->v_patches + F * np * (size_t)(V.patch * V.patch * 3) * 2;
    B.v_emb = (char*)h->v_emb + F * np * D * 2;
    B.v_x = (char*)h->v_x + F * ntok * D * 2;
    B.v_y = (char*)h->v_y + F * ntok * D * 2;
    B.v_qkv = (char*)h->v_qkv + F * ntok * 3 * D * 2;
    B.v_hid = (char*)h->v_hid + F * ntok * (size_t)V.mlp * 2;
    B.v_cls = (char*)h->v_cls + F * D * 2;
    B.logits = h->logits + F * h->cpad;
    auto gemm = [&](int layer, const void* x, int rows_per_frame, const void* res, int act, void* y, int out_f32) -> const char* {
        const Layer& L = h->layers[layer];
        fav_conv_desc d;
        memset(&d, 0, sizeof d);
        d.x = x; d.w = L.w; d.bias = L.b; d.res = res; d.y = y;
        d.n_frames = n; d.H = rows_per_frame; d.W = 1; d.Cin = L.k; d.Cout = L.cout;
        d.kh = 1; d.kw = 1; d.stride = 1; d.pad = 0; d.relu = act; d.out_f32 = out_f32; d.math_mode = c.math_mode;
        d.drop.site = -1;
        if (!out_f32 && c.math_mode == FAV_MATH_BF16 && L.cout == L.cout_pad &&
            launch_gemm_streamk(h, x, L.w, L.b, res, y, (long long)n * rows_per_frame, L.k, L.cout, act, s)) return nullptr;
        return launch_conv(h, d, L.cout_pad, out_f32 ? L.cout_pad : L.cout, s);
    };
#define FAV_VIT_TRY(expr)                                            \
    do {                                                             \
        if (const char* e_ = (expr)) { h->err = e_; return FAV_ERR_INVALID_ARG; } \
    } while (0)
    h->cur_op = -1;
    // patch embedding: normalise + im2col (k = (r*P + s)*3 + c), GEMM, add positions / class token
    launch_stem(h, images, layout, n, c.in_h, c.in_w, V.patch, V.patch, V.patch, 0, V.patch * V.patch * 3, c.mean, inv_std, B.v_patches, s);
    FAV_VIT_TRY(gemm(0, B.v_patches, gh * gw, nullptr, 0, B.v_emb, 0));
    launch_vit_assemble(h, B.v_emb, (const float*)h->layers[1].w, B.v_x, n, ntok, D, s);
    int li = 2;
    for (int blk = 0; blk < V.depth; ++blk, li += 6) {
        const Layer &ln1 = h->layers[li], &ln2 = h->layers[li + 3];
        FAV_VIT_TRY(launch_layernorm(h, B.v_x, D, (const float*)ln1.w, ln1.b, B.v_y, (long long)n * ntok, D, 1e-6f, s));
        FAV_VIT_TRY(gemm(li + 1, B.v_y, ntok, nullptr, 0, B.v_qkv, 0));
        FAV_VIT_TRY(launch_attention(h, B.v_qkv, B.v_y, n, ntok, D, V.heads, c.math_mode, s));
        FAV_VIT_TRY(gemm(li + 2, B.v_y, ntok, B.v_x, 0, B.v_x, 0));                 // x = x + proj(attn), in place tile by tile
        FAV_VIT_TRY(launch_layernorm(h, B.v_x, D, (const float*)ln2.w, ln2.b, B.v_y, (long long)n * ntok, D, 1e-6f, s));
        FAV_VIT_TRY(gemm(li + 4, B.v_y, ntok, nullptr, 2, B.v_hid, 0));              // GELU fused
        FAV_VIT_TRY(gemm(li + 5, B.v_hid, ntok, B.v_x, 0, B.v_x, 0));
    }
    const Layer& lnf = h->layers[li];
    FAV_VIT_TRY(launch_layernorm(h, B.v_x, (long long)ntok * D, (const float*)lnf.w, lnf.b, B.v_cls, n, D, 1e-6f, s));   // class tokens only
    FAV_VIT_TRY(gemm(li + 1, B.v_cls, 1, nullptr, 0, B.logits, 1));
#undef FAV_VIT_TRY
    return FAV_OK;
}

}  // namespace

// =============================================================================
// C ABI
// =============================================================================
extern "C" {

int32_t fav_abi_version(void) { return FAV_ABI_VERSION; }

void fav_default_config(fav_config* c, int32_t arch) {
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->struct_size = sizeof *c;
    c->device = 0;
    c->arch = arch;
    c->num_classes = arch == FAV_ARCH_RESNET18_CIFAR ? 10 : 1000;
    c->in_h = c->in_w = (arch == FAV_ARCH_RESNET50 || arch == FAV_ARCH_VIT_B16) ? 224 : (arch == FAV_ARCH_VIT_TINY ? 64 : 32);
    c->max_batch = 256;
    c->mean[0] = 0.485f; c->mean[1] = 0.456f; c->mean[2] = 0.406f;
    c->stdev[0] = 0.229f; c->stdev[1] = 0.224f; c->stdev[2] = 0.225f;
    c->n_samples = 1;
    c->site_mask = 0;
    c->dropout_p = 0.f;
    c->seed = 0;
    c->temperature = 1.f;
    c->conf_kind = FAV_CONF_MAX_SOFTMAX;
    c->tau = 0.5f;
    c->math_mode = FAV_MATH_BF16;
    c->chunk_a = 0; c->chunk_b = 0; c->regroup_block = -1;
    c->n_members = 1;
}

const char* fav_last_error(const fav_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

fav_status fav_create(const fav_config* cfg, fav_handle** out) {
    if (out) *out = nullptr;
    if (!cfg || !out) { g_create_error = "fav_create: null argument"; return FAV_ERR_INVALID_ARG; }
    if (cfg->struct_size != sizeof(fav_config)) { g_create_error = "fav_create: fav_config.struct_size mismatch"; return FAV_ERR_INVALID_ARG; }
    if (cfg->arch < 0 || cfg->arch > 3) { g_create_error = "fav_create: unknown arch"; return FAV_ERR_UNSUPPORTED; }
    if (cfg->num_classes < 1 || cfg->num_classes > 1024 || cfg->max_batch < 1 || cfg->in_h < 8 || cfg->in_w < 8 ||
        cfg->n_samples < 1 || cfg->n_samples > 4096 || !(cfg->temperature > 0.f) || cfg->dropout_p < 0.f || cfg->dropout_p >= 1.f ||
        !(cfg->stdev[0] > 0.f && cfg->stdev[1] > 0.f && cfg->stdev[2] > 0.f) || cfg->math_mode < 0 || cfg->math_mode > 1 ||
        cfg->conf_kind < 0 || cfg->conf_kind > 1 || cfg->n_members < 0 || cfg->n_members > 64 ||
        cfg->tail_min_rows < -1 || cfg->ens_grouped_max < -1 || cfg->vit_streams < 0 || cfg->vit_streams > 4 || cfg->stem_fused < -1 || cfg->stem_fused > 0 ||
        (cfg->n_members > 1 && cfg->site_mask != 0 && cfg->dropout_p > 0.f)) {   // ensemble members are deterministic
        g_create_error = "fav_create: config value out of range";
        return FAV_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        (void)hipGetLastError();
        g_create_error = fmt("fav_create: no usable HIP device (count=%d, requested=%d); this path has no CPU fallback", ndev, cfg->device);
        return FAV_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = fmt("fav_create: device %d is '%s', kernels are built for gfx950 only", cfg->device, prop.gcnArchName);
        return FAV_ERR_NO_DEVICE;
    }
    fav_handle* h = new fav_handle();
    h->cfg = *cfg;
    h->n_members = cfg->n_members > 1 ? cfg->n_members : 1;
    h->member_loaded.assign(h->n_members, 0);
    if (hipSetDevice(cfg->device) != hipSuccess) { g_create_error = "hipSetDevice failed"; delete h; return FAV_ERR_HIP; }
    fav_status st = is_vit_arch(cfg->arch) ? build_vit(h) : build_graph(h);
    if (st == FAV_OK && !h->vit) st = plan_memory(h);
    if (st != FAV_OK) { g_create_error = h->err; free_all(h); delete h; return st; }
    *out = h;
    return FAV_OK;
}

// The static schedule of a configuration as text, one line per op - no device needed (tests/test_host.py replays it
// symbolically and checks that the fused schedule computes the same dataflow as the layer-by-layer one).
//   "op <i> kind=<k> phase=<p> layer=<l> lc=<l> la=<l> in=<b> res=<b> out=<b> out2=<b> site=<s> relu=<r>"
// buffers: 0..4 rotating, 5 im2col matrix, -1 frames, -2 phase input, -3 phase output, -4 none.
fav_status fav_plan_schedule(const fav_config* cfg, int32_t flags, char* out, size_t cap) {
    if (!cfg || !out || cap < 2 || cfg->struct_size != sizeof(fav_config) || cfg->arch < 0 || cfg->arch > 1) return FAV_ERR_INVALID_ARG;
    fav_handle h;
    h.cfg = *cfg;
    h.n_members = cfg->n_members > 1 ? cfg->n_members : 1;
    h.plan_no_fuse = (flags & 1) != 0;
    fav_status st = build_graph(&h);
    if (st != FAV_OK) { snprintf(out, cap, "%s", h.err.c_str()); return st; }
    std::string txt;
    for (size_t i = 0; i < h.ops.size(); ++i) {
        const Op& o = h.ops[i];
        int phase = -1;
        for (size_t pi = 0; pi < h.phases.size(); ++pi)
            if ((int)i >= h.phases[pi].op_begin && (int)i < h.phases[pi].op_end) phase = (int)pi;
        txt += fmt("op %zu kind=%d phase=%d layer=%d lc=%d la=%d in=%d res=%d out=%d out2=%d site=%d relu=%d suffix=%d rese=%d skipy=%d esite=%d\n", i, (int)o.kind, phase, o.layer,
                   o.layer_c, o.layer_a, o.in, o.res, o.out, o.out2, o.site, o.relu, phase >= 0 ? (int)h.phases[phase].suffix : 0, o.res_entry, o.skip_y,
                   h.first_site);
    }
    if (txt.size() + 1 > cap) return FAV_ERR_INVALID_ARG;
    memcpy(out, txt.c_str(), txt.size() + 1);
    return FAV_OK;
}

void fav_destroy(fav_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
}

fav_status fav_load_weights(fav_handle* h, const void* blob, size_t size) { return fav_load_member_weights(h, 0, blob, size); }

// Structural validation of a FAVW v1 blob; needs no device and no handle.  Every offset is taken from the
// (untrusted) file, so the range checks are written overflow-safe (off > size || bytes > size - off) and
// the data must be aligned for its element type (pack_blob emits 64-byte aligned offsets).
fav_status fav_check_blob(const void* blob, size_t size, char* err, size_t err_cap) {
    auto fail = [&](const std::string& m) {
        if (err && err_cap) { snprintf(err, err_cap, "%s", m.c_str()); }
        return FAV_ERR_BAD_BLOB;
    };
    if (err && err_cap) err[0] = 0;
    if (!blob || size < 32) return fail("blob too small");
    const uint8_t* p = (const uint8_t*)blob;
    uint32_t hdr[8];
    memcpy(hdr, p, 32);
    if (hdr[0] != 0x57564146u || hdr[1] != 1u) return fail("not a FAVW v1 blob");
    const size_t nl = hdr[4];
    if (nl == 0 || nl > 4096) return fail("implausible layer count");
    if ((size - 32) / 48 < nl) return fail("truncated layer table");
    const size_t data0 = 32 + 48 * nl;
    auto in_range = [&](uint64_t off, uint64_t bytes, uint64_t align) {
        return off >= data0 && off <= size && bytes <= size - off && off % align == 0;
    };
    for (size_t i = 0; i < nl; ++i) {
        uint32_t t[8];
        uint64_t off[2];
        memcpy(t, p + 32 + 48 * i, 32);
        memcpy(off, p + 32 + 48 * i + 32, 16);
        const uint64_t cout = t[0], cin = t[1], kh = t[2], kw = t[3];
        if (cout == 0 || cout > (1u << 24) || cin > (1u << 20) || kh > 64 || kw > 64) return fail(fmt("layer %zu: implausible shape", i));
        // every value must be finite: a NaN weight would defeat the pixel sanitiser and the bf16 rounding recipe (exponent all ones = Inf / NaN)
        auto f32_finite = [&](uint64_t o, uint64_t n) { for (uint64_t j = 0; j < n; ++j) { uint32_t u; memcpy(&u, p + o + 4 * j, 4); if ((u & 0x7F800000u) == 0x7F800000u) return false; } return true; };
        auto bf16_finite = [&](uint64_t o, uint64_t n) { for (uint64_t j = 0; j < n; ++j) { uint16_t u; memcpy(&u, p + o + 2 * j, 2); if ((u & 0x7F80u) == 0x7F80u) return false; } return true; };
        if (kh == 0) {   // a pair of fp32 vectors
            if (!in_range(off[0], cout * 4, 4) || !in_range(off[1], cout * 4, 4)) return fail(fmt("layer %zu data out of range", i));
            if (!f32_finite(off[0], cout) || !f32_finite(off[1], cout)) return fail(fmt("layer %zu holds a non-finite value", i));
            continue;
        }
        if (kw == 0 || cin == 0) return fail(fmt("layer %zu: implausible shape", i));
        const uint64_t k = kh * kw * cin;   // < 2^32
        if (k > (1ull << 32) / cout) return fail(fmt("layer %zu: implausible shape", i));
        if (!in_range(off[0], cout * k * 2, 2) || !in_range(off[1], cout * 4, 4)) return fail(fmt("layer %zu data out of range", i));
        if (!bf16_finite(off[0], cout * k) || !f32_finite(off[1], cout)) return fail(fmt("layer %zu holds a non-finite value", i));
    }
    return FAV_OK;
}

namespace {
// one allocation per layer for the weights of all members (and one for the biases): member m at slab + m * stride
fav_status alloc_layer_slabs(fav_handle* h, Layer& L, size_t wbytes, size_t bbytes) {
    if (L.w_slab) return FAV_OK;
    L.w_stride = (wbytes + 255) / 256 * 256;
    L.b_stride = (bbytes + 255) / 256 * 256;
    HIP_TRY(h, hipMalloc(&L.w_slab, L.w_stride * h->n_members));
    HIP_TRY(h, hipMalloc(&L.b_slab, L.b_stride * h->n_members));
    L.w_m.assign(h->n_members, nullptr);
    L.b_m.assign(h->n_members, nullptr);
    for (int m = 0; m < h->n_members; ++m) {
        L.w_m[m] = (uint16_t*)((char*)L.w_slab + (size_t)m * L.w_stride);
        L.b_m[m] = (float*)((char*)L.b_slab + (size_t)m * L.b_stride);
    }
    return FAV_OK;
}
}  // namespace

fav_status fav_load_member_weights(fav_handle* h, int32_t member, const void* blob, size_t size) {
    if (!h) return FAV_ERR_INVALID_ARG;
    if (member < 0 || member >= h->n_members) { h->err = fmt("fav_load_member_weights: member %d outside [0, %d)", member, h->n_members); return FAV_ERR_INVALID_ARG; }
    {
        char msg[200];
        if (fav_check_blob(blob, size, msg, sizeof msg) != FAV_OK) { h->err = std::string("fav_load_weights: ") + msg; return FAV_ERR_BAD_BLOB; }
    }
    const uint8_t* p = (const uint8_t*)blob;
    uint32_t hdr[8];
    memcpy(hdr, p, 32);
    if ((int)hdr[2] != h->cfg.arch || (int)hdr[3] != h->cfg.num_classes || hdr[4] != h->layers.size()) {
        h->err = fmt("fav_load_weights: blob is arch %u / %u classes / %u layers, handle expects %d / %d / %zu", hdr[2], hdr[3],
                     hdr[4], h->cfg.arch, h->cfg.num_classes, h->layers.size());
        return FAV_ERR_BAD_BLOB;
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    std::vector<uint16_t> wtmp;
    std::vector<float> btmp;
    for (size_t i = 0; i < h->layers.size(); ++i) {
        Layer& L = h->layers[i];
        uint32_t t[8];
        uint64_t off[2];
        memcpy(t, p + 32 + 48 * i, 32);
        memcpy(off, p + 32 + 48 * i + 32, 16);
        if ((int)t[0] != L.cout || (int)t[1] != L.cin || (int)t[2] != L.kh || (int)t[3] != L.kw || (int)t[4] != L.stride ||
            (int)t[5] != L.pad) {
            h->err = fmt("fav_load_weights: layer %zu shape mismatch", i);
            return FAV_ERR_BAD_BLOB;
        }
        if (L.kh == 0) {   // a pair of fp32 vectors (LayerNorm gamma / beta, ViT position table)
            const size_t vb = (size_t)L.cout * 4;
            // (ranges and alignment already validated by fav_check_blob against these very table entries)
            if (fav_status st = alloc_layer_slabs(h, L, vb, vb)) return st;
            HIP_TRY(h, hipMemcpy(L.w_m[member], p + off[0], vb, hipMemcpyHostToDevice));
            HIP_TRY(h, hipMemcpy(L.b_m[member], p + off[1], vb, hipMemcpyHostToDevice));
            continue;
        }
        const size_t kreal = (size_t)L.kh * L.kw * L.cin;
        const size_t wbytes = (size_t)L.cout * kreal * 2, bbytes = (size_t)L.cout * 4;
        (void)wbytes;
        // device layout: [cout_pad][L.k] bf16, zero padded in both dimensions
        wtmp.assign((size_t)L.cout_pad * L.k, 0);
        const uint16_t* src = (const uint16_t*)(p + off[0]);
        for (int n = 0; n < L.cout; ++n) memcpy(&wtmp[(size_t)n * L.k], src + (size_t)n * kreal, kreal * 2);
        btmp.assign(L.cout_pad, 0.f);
        memcpy(btmp.data(), p + off[1], bbytes);
        if (fav_status st = alloc_layer_slabs(h, L, wtmp.size() * 2, btmp.size() * 4)) return st;
        HIP_TRY(h, hipMemcpy(L.w_m[member], wtmp.data(), wtmp.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(L.b_m[member], btmp.data(), btmp.size() * 4, hipMemcpyHostToDevice));
    }
    h->member_loaded[member] = 1;
    h->weights_loaded = true;
    for (char c : h->member_loaded) h->weights_loaded = h->weights_loaded && c;
    return FAV_OK;
}

namespace {
// orders this call's stream behind the previous user of the handle's buffers (see fav_handle::ev_last)
fav_status wait_last_use(fav_handle* h, hipStream_t s) {
    if (!h->ev_last) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming));
    if (h->ev_last_set) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_last, 0));
    return FAV_OK;
}
void mark_last_use(fav_handle* h, hipStream_t s) {
    if (h->ev_last && hipEventRecord(h->ev_last, s) == hipSuccess) h->ev_last_set = true;
}
fav_status classify_on_stream(fav_handle* h, const void* images, int32_t n, int32_t layout, int64_t first_index,
                              int32_t* labels, float* conf, uint8_t* fail, float* score, hipStream_t s, int out_stride);
}  // namespace

fav_status fav_classify_ex(fav_handle* h, const void* images, int32_t n, int32_t layout, int64_t first_index,
                           int32_t* labels, float* conf, uint8_t* fail, float* score, void* stream) {
    if (!h) return FAV_ERR_INVALID_ARG;
    if (!h->weights_loaded) { h->err = "fav_classify: no weights loaded"; return FAV_ERR_NO_WEIGHTS; }
    if (!images || !labels || !conf) { h->err = "fav_classify: null buffer"; return FAV_ERR_INVALID_ARG; }
    if (n < 1 || n > h->cfg.max_batch) { h->err = fmt("fav_classify: n=%d outside [1, max_batch=%d]", n, h->cfg.max_batch); return FAV_ERR_INVALID_ARG; }
    if (layout != FAV_LAYOUT_NHWC_U8 && layout != FAV_LAYOUT_NHWC_F32) { h->err = "fav_classify: unknown layout"; return FAV_ERR_INVALID_ARG; }
    if (first_index < 0 || first_index + n > 0xFFFFFFFFll) { h->err = "fav_classify: first_image_index out of range"; return FAV_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (fav_status st = wait_last_use(h, s)) return st;
    const fav_status st = classify_on_stream(h, images, n, layout, first_index, labels, conf, fail, score, s, 1);
    mark_last_use(h, s);      // also after a failure: whatever was queued before it still uses the buffers
    return st;
}

fav_status fav_classify_records(fav_handle* h, const void* images, int32_t n, int32_t layout, int64_t first_index,
                                void* records, uint8_t* fail, float* score, void* stream) {
    if (!h) return FAV_ERR_INVALID_ARG;
    if (!h->weights_loaded) { h->err = "fav_classify_records: no weights loaded"; return FAV_ERR_NO_WEIGHTS; }
    if (!images || !records || ((uintptr_t)records & 7)) { h->err = "fav_classify_records: null or misaligned buffer"; return FAV_ERR_INVALID_ARG; }
    if (n < 1 || n > h->cfg.max_batch) { h->err = fmt("fav_classify_records: n=%d outside [1, max_batch=%d]", n, h->cfg.max_batch); return FAV_ERR_INVALID_ARG; }
    if (layout != FAV_LAYOUT_NHWC_U8 && layout != FAV_LAYOUT_NHWC_F32) { h->err = "fav_classify_records: unknown layout"; return FAV_ERR_INVALID_ARG; }
    if (first_index < 0 || first_index + n > 0xFFFFFFFFll) { h->err = "fav_classify_records: first_image_index out of range"; return FAV_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (fav_status st = wait_last_use(h, s)) return st;
    const fav_status st = classify_on_stream(h, images, n, layout, first_index, (int32_t*)records, (float*)records + 1, fail, score, s, 2);
    mark_last_use(h, s);
    return st;
}

namespace {
fav_status classify_on_stream(fav_handle* h, const void* images, int32_t n, int32_t layout, int64_t first_index,
                              int32_t* labels, float* conf, uint8_t* fail, float* score, hipStream_t s, int out_stride) {
    h->ev_used = h->profiling ? h->ev_used : 0;
    if (h->vit) {
        for (auto& L : h->layers) { L.w = L.w_m[0]; L.b = L.b_m[0]; }
        // the batch in fav_config.vit_streams (default 2) parts on as many streams: at 197 rows per frame every GEMM of the encoder is a
        // few hundred tiles, and the partial last round of one part's launch is filled by another part's (1: one stream)
        const int vit_streams = h->cfg.vit_streams <= 0 ? 2 : std::min(4, (int)h->cfg.vit_streams);
        if (vit_streams > 1 && n >= 8 * vit_streams) {
            if (!h->ev_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            while ((int)h->vit_streams.size() < vit_streams) {
                hipStream_t st_; hipEvent_t ev_;
                HIP_TRY(h, hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
                HIP_TRY(h, hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
                h->vit_streams.push_back(st_); h->vit_done.push_back(ev_);
            }
            HIP_TRY(h, hipEventRecord(h->ev_fork, s));
            fav_status st = FAV_OK;     // from here to the join nothing returns: a forked stream is always joined into s
            for (int part = 0; part < vit_streams; ++part) {
                const int f0 = (int)((long long)n * part / vit_streams), f1 = (int)((long long)n * (part + 1) / vit_streams);
                HIP_KEEP(h, st, hipStreamWaitEvent(h->vit_streams[part], h->ev_fork, 0));
                h->sk_slot = 1 + part;
                if (st == FAV_OK) st = run_vit(h, images, layout, f0, f1 - f0, h->vit_streams[part]);
                HIP_KEEP(h, st, hipEventRecord(h->vit_done[part], h->vit_streams[part]));
                HIP_KEEP(h, st, hipStreamWaitEvent(s, h->vit_done[part], 0));
            }
            if (st != FAV_OK) return st;
        } else {
            h->sk_slot = 0;
            fav_status st = run_vit(h, images, layout, 0, n, s);
            if (st != FAV_OK) return st;
        }
    } else if (!h->mws.empty() && h->can_group && n <= h->group_max_frames) {
        // small calls: every op ONE launch over all members (block row = member), on the caller's stream.  At the 8-GPU share
        // of configs[3] (32 frames) a member's launches are a few dozen tiles each; five of them in one grid fill the chip
        // where five streams only interleave (profiles/r3aa_ens_grouped_ab.txt)
        for (int member = 0; member < h->n_members; ++member)
            h->mws[member].phase_out.back() = (char*)h->logits + (size_t)member * n * h->cpad * 4;
        for (size_t pi = 0; pi < h->phases.size(); ++pi) {
            fav_status st = run_chunks(h, pi, images, layout, n, first_index, s, 0, n, h->mws[0].act, &h->mws[0], 0, -1, -1, h->n_members);
            if (st != FAV_OK) return st;
        }
    } else if (!h->mws.empty()) {
        // members side by side: fork from the caller's stream, one stream per member, join before the head
        HIP_TRY(h, hipEventRecord(h->ev_members, s));
        fav_status st = FAV_OK;         // nothing returns between the fork and the join
        for (int member = 0; member < h->n_members; ++member) {
            fav_handle::MemberWs& w = h->mws[member];
            w.phase_out.back() = (char*)h->logits + (size_t)member * n * h->cpad * 4;
            HIP_KEEP(h, st, hipStreamWaitEvent(w.stream, h->ev_members, 0));
        }
        // member by member.  (Enqueuing op by op ACROSS the members - so that all five streams start together and the launches
        // sharing the chip are the same op of different members - measured 5 % slower at 32 frames per call, 7 205 vs 7 585
        // frames/s, and 1 % slower at 256: profiles/r3j_ens_interleave_ab.txt.)
        for (int member = 0; member < h->n_members && st == FAV_OK; ++member) {
            fav_handle::MemberWs& w = h->mws[member];
            for (size_t pi = 0; pi < h->phases.size() && st == FAV_OK; ++pi) {
                const long long dom = h->phases[pi].suffix ? (long long)n * h->T_eff : n;
                st = run_chunks(h, pi, images, layout, n, first_index, w.stream, 0, dom, w.act, &w, member);
            }
        }
        for (int member = 0; member < h->n_members; ++member) {
            fav_handle::MemberWs& w = h->mws[member];
            HIP_KEEP(h, st, hipEventRecord(w.done, w.stream));
            HIP_KEEP(h, st, hipStreamWaitEvent(s, w.done, 0));
        }
        if (st != FAV_OK) return st;
    } else
    for (int member = 0; member < h->n_members; ++member) {
    for (auto& L : h->layers) { L.w = L.w_m[member]; L.b = L.b_m[member]; }
    // member m writes logits[m][n][cpad]: the head then averages members exactly as it averages samples
    h->phase_out.back() = (char*)h->logits + (size_t)member * n * h->cpad * 4;
    const size_t nph = h->phases.size();
    const size_t serial_end = h->pipe_first >= 0 ? (size_t)h->pipe_first : nph;
    for (size_t pi = 0; pi < serial_end; ++pi) {
        const long long dom = h->phases[pi].suffix ? (long long)n * h->T_eff : n;
        fav_status st = run_chunks(h, pi, images, layout, n, first_index, s, 0, dom, h->act);
        if (st != FAV_OK) return st;
    }
    if (h->pipe_first >= 0) {
        const size_t pa = (size_t)h->pipe_first, pb = pa + 1;
        const long long dom = h->phases[pa].suffix ? (long long)n * h->T_eff : n;
        const long long step = h->phases[pa].chunk;  // == phases[pb].chunk
        HIP_TRY(h, hipEventRecord(h->ev_fork, s));
        HIP_TRY(h, hipStreamWaitEvent(h->stream_a, h->ev_fork, 0));
        HIP_TRY(h, hipStreamWaitEvent(h->stream_b, h->ev_fork, 0));
        size_t ci = 0;
        fav_status st = FAV_OK;         // nothing returns between the fork and the join
        for (long long v0 = 0; v0 < dom && st == FAV_OK; v0 += step, ++ci) {
            const long long v1 = std::min(dom, v0 + step);
            st = run_chunks(h, pa, images, layout, n, first_index, h->stream_a, v0, v1, h->act);
            if (st != FAV_OK) break;
            if (ci >= h->ev_chunk.size()) {
                hipEvent_t e = nullptr;
                HIP_KEEP(h, st, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                if (st != FAV_OK) break;
                h->ev_chunk.push_back(e);
            }
            HIP_KEEP(h, st, hipEventRecord(h->ev_chunk[ci], h->stream_a));
            HIP_KEEP(h, st, hipStreamWaitEvent(h->stream_b, h->ev_chunk[ci], 0));
            if (st == FAV_OK) st = run_chunks(h, pb, images, layout, n, first_index, h->stream_b, v0, v1, h->act2);
        }
        HIP_KEEP(h, st, hipEventRecord(h->ev_join_a, h->stream_a));
        HIP_KEEP(h, st, hipEventRecord(h->ev_join_b, h->stream_b));
        HIP_KEEP(h, st, hipStreamWaitEvent(s, h->ev_join_a, 0));
        HIP_KEEP(h, st, hipStreamWaitEvent(s, h->ev_join_b, 0));
        if (st != FAV_OK) return st;
    }
    }
    if (!h->vit) h->phase_out.back() = h->logits;
    const int T_head = h->n_members > 1 ? h->n_members : h->T_eff;
    if (const char* e = launch_head(h, h->logits, T_head, n, h->cfg.num_classes, h->cpad, h->cfg.temperature,
                                    h->cfg.conf_kind, h->cfg.tau, labels, conf, fail, score, s, out_stride)) {
        h->err = e;
        return FAV_ERR_INVALID_ARG;
    }
    HIP_TRY(h, hipGetLastError());
    h->last_T = T_head;
    h->last_n = n;
    return FAV_OK;
}
}  // namespace

fav_status fav_classify(fav_handle* h, const void* images, int32_t n, int32_t layout, int32_t* labels, float* conf,
                        void* stream) {
    return fav_classify_ex(h, images, n, layout, 0, labels, conf, nullptr, nullptr, stream);
}

fav_status fav_classify_host(fav_handle* h, const void* images, int32_t n, int32_t layout, int64_t first_index,
                             int32_t* labels, float* conf, uint8_t* fail, float* score) {
    if (!h) return FAV_ERR_INVALID_ARG;
    if (!images || !labels || !conf || n < 1 || n > h->cfg.max_batch) { h->err = "fav_classify_host: bad argument"; return FAV_ERR_INVALID_ARG; }
    // validate everything that sizes the copy BEFORE touching the caller's buffer
    if (layout != FAV_LAYOUT_NHWC_U8 && layout != FAV_LAYOUT_NHWC_F32) { h->err = "fav_classify_host: unknown layout"; return FAV_ERR_INVALID_ARG; }
    if (!h->weights_loaded) { h->err = "fav_classify_host: no weights loaded"; return FAV_ERR_NO_WEIGHTS; }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->host_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking));
    hipStream_t hs = h->host_stream;
    const size_t bpe = layout == FAV_LAYOUT_NHWC_U8 ? 1 : 4;
    const size_t img_bytes = (size_t)h->cfg.max_batch * h->cfg.in_h * h->cfg.in_w * 3 * 4;
    const size_t res_off = (img_bytes + 255) / 256 * 256;
    if (!h->host_stage) HIP_TRY(h, hipMalloc(&h->host_stage, res_off + (size_t)h->cfg.max_batch * 16 + 256));
    char* base = (char*)h->host_stage;
    int32_t* dl = (int32_t*)(base + res_off);
    float* dc = (float*)(dl + h->cfg.max_batch);
    float* ds = dc + h->cfg.max_batch;
    uint8_t* df = (uint8_t*)(ds + h->cfg.max_batch);
    // everything on the handle's own stream; only that stream is synchronised (not the device)
    HIP_TRY(h, hipMemcpyAsync(base, images, (size_t)n * h->cfg.in_h * h->cfg.in_w * 3 * bpe, hipMemcpyHostToDevice, hs));
    fav_status st = fav_classify_ex(h, base, n, layout, first_index, dl, dc, df, ds, hs);
    if (st != FAV_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(labels, dl, (size_t)n * 4, hipMemcpyDeviceToHost, hs));
    HIP_TRY(h, hipMemcpyAsync(conf, dc, (size_t)n * 4, hipMemcpyDeviceToHost, hs));
    if (score) HIP_TRY(h, hipMemcpyAsync(score, ds, (size_t)n * 4, hipMemcpyDeviceToHost, hs));
    if (fail) HIP_TRY(h, hipMemcpyAsync(fail, df, (size_t)n, hipMemcpyDeviceToHost, hs));
    HIP_TRY(h, hipStreamSynchronize(hs));
    return FAV_OK;
}

fav_status fav_get_logits(fav_handle* h, float* out, int32_t* t_out, int32_t* n_out, void* stream) {
    if (!h) return FAV_ERR_INVALID_ARG;
    if (h->last_n == 0) { h->err = "fav_get_logits: no classify call yet"; return FAV_ERR_INVALID_ARG; }
    if (t_out) *t_out = h->last_T;
    if (n_out) *n_out = h->last_n;
    if (out) {
        if (fav_status st = wait_last_use(h, (hipStream_t)stream)) return st;
        HIP_TRY(h, hipMemcpy2DAsync(out, (size_t)h->cfg.num_classes * 4, h->logits, (size_t)h->cpad * 4,
                                    (size_t)h->cfg.num_classes * 4, (size_t)h->last_T * h->last_n,
                                    hipMemcpyDeviceToDevice, (hipStream_t)stream));
        mark_last_use(h, (hipStream_t)stream);
    }
    return FAV_OK;
}

fav_status fav_set_profiling(fav_handle* h, int32_t enable) {
    if (!h) return FAV_ERR_INVALID_ARG;
    h->profiling = enable != 0;
    h->ev_used = 0;
    memset(&h->prof, 0, sizeof h->prof);
    h->op_prof.assign(h->ops.size(), fav_op_profile{});
    for (size_t i = 0; i < h->ops.size(); ++i) {
        const Op& o = h->ops[i];
        fav_op_profile& r = h->op_prof[i];
        r.op_index = (int)i; r.kind = (int)o.kind;
        r.H = o.H; r.W = o.W; r.Cin = o.C; r.Ho = o.Ho; r.Wo = o.Wo; r.Cout = o.Co;
        r.kh = r.kw = r.stride = 0;
        if (o.layer >= 0) { const Layer& L = h->layers[o.layer]; r.kh = L.kh; r.kw = L.kw; r.stride = L.stride; }
        if (o.kind == OP_ENTRY_REDUCE) { r.reserved = o.Co2; r.kh = r.kw = r.stride = 1; r.Ho = o.H; r.Wo = o.W; r.Cout = o.C; }
        if (o.kind == OP_TAIL) { r.reserved = o.Co2; if (o.layer < 0) { r.kh = r.kw = r.stride = 1; } }   // reserved: channels of the fused next-block conv1
    }
    return FAV_OK;
}

fav_status fav_get_profile(fav_handle* h, fav_profile* out, int32_t reset) {
    if (!h || !out) return FAV_ERR_INVALID_ARG;
    HIP_TRY(h, hipDeviceSynchronize());
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b) == hipSuccess) {
            h->prof.ms[h->ev_pool[i].cls] += ms;
            const int op = h->ev_pool[i].op;
            if (op >= 0 && op < (int)h->op_prof.size()) h->op_prof[op].ms += ms;
        }
    }
    h->ev_used = 0;
    *out = h->prof;
    if (reset) memset(&h->prof, 0, sizeof h->prof);
    return FAV_OK;
}

fav_status fav_get_op_profile(fav_handle* h, fav_op_profile* out, int32_t cap, int32_t* n_out) {
    if (!h || !n_out) return FAV_ERR_INVALID_ARG;
    *n_out = (int32_t)h->op_prof.size();
    if (out) for (int i = 0; i < cap && i < (int)h->op_prof.size(); ++i) out[i] = h->op_prof[i];
    return FAV_OK;
}

// ---- operator level --------------------------------------------------------
static fav_status op_done(const char* e) {
    if (e) { g_create_error = e; return FAV_ERR_INVALID_ARG; }
    if (hipGetLastError() != hipSuccess) { g_create_error = "kernel launch failed"; return FAV_ERR_HIP; }
    return FAV_OK;
}

fav_status fav_op_conv2d(const fav_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->bias || !d->y) return op_done("fav_op_conv2d: null pointer");
    if (d->Cout % 64 != 0) return op_done("fav_op_conv2d: Cout must be a multiple of 64");
    return op_done(launch_conv(nullptr, *d, d->Cout, d->Cout, (hipStream_t)stream));
}

fav_status fav_op_bottleneck_tail(const fav_tail_desc* d, void* stream) {
    if (!d || !d->x || !d->wc || !d->bias_c || !d->res || !d->y) return op_done("fav_op_bottleneck_tail: null pointer");
    if ((d->wb && !d->bias_b) || (d->wa && (!d->bias_a || !d->t1n))) return op_done("fav_op_bottleneck_tail: null pointer");
    return op_done(launch_tail(nullptr, *d, (hipStream_t)stream));
}

fav_status fav_op_stem_im2col(const void* images, int32_t layout, int32_t n, int32_t H, int32_t W, int32_t kh, int32_t kw,
                              int32_t stride, int32_t pad, int32_t kpad, const float* mean3, const float* inv_std3,
                              void* out, void* stream) {
    if (!images || !out || !mean3 || !inv_std3 || kpad % 64 != 0 || kpad < kh * kw * 3) return op_done("fav_op_stem_im2col: bad argument");
    launch_stem(nullptr, images, layout, n, H, W, kh, kw, stride, pad, kpad, mean3, inv_std3, out, (hipStream_t)stream);
    return op_done(nullptr);
}

fav_status fav_op_stem_pool(const void* images, int32_t layout, int32_t n, int32_t H, int32_t W, const void* w, const float* bias,
                            const float* mean3, const float* inv_std3, void* out, void* stream) {
    if (!images || !w || !bias || !mean3 || !inv_std3 || !out) return op_done("fav_op_stem_pool: null argument");
    return op_done(launch_stem_pool(nullptr, images, layout, n, H, W, w, bias, mean3, inv_std3, out, (hipStream_t)stream));
}

fav_status fav_op_maxpool3x3s2(const void* x, void* y, int32_t n, int32_t H, int32_t W, int32_t C, void* stream) {
    if (!x || !y || C % 8 != 0) return op_done("fav_op_maxpool3x3s2: bad argument");
    launch_maxpool(nullptr, x, y, n, H, W, C, (hipStream_t)stream);
    return op_done(nullptr);
}

fav_status fav_op_avgpool(const void* x, void* y, int32_t n, int32_t HW, int32_t C, const fav_dropout_desc* drop, void* stream) {
    if (!x || !y || C % 16 != 0 || HW < 1) return op_done("fav_op_avgpool: bad argument");
    launch_avgpool(nullptr, x, y, n, HW, C, make_drop(drop), (hipStream_t)stream);
    return op_done(nullptr);
}

fav_status fav_op_entry_reduce(const void* x, void* y, const void* wa, const float* bias_a, void* t1, int32_t C, int32_t Nred,
                               int32_t HW, int32_t n_out, const fav_dropout_desc* drop, void* stream) {
    if (!x || !wa || !bias_a || !t1 || !drop || drop->site < 0 || HW < 1) return op_done("fav_op_entry_reduce: bad argument");
    return op_done(launch_entry_reduce(nullptr, x, y, wa, bias_a, t1, C, Nred, HW, n_out, make_drop(drop), (hipStream_t)stream));
}

fav_status fav_op_entry_dropout(const void* x, void* out, int64_t elems, int32_t n_out, const fav_dropout_desc* drop, void* stream) {
    if (!x || !out || !drop || drop->site < 0 || elems % 16 != 0) return op_done("fav_op_entry_dropout: bad argument");
    launch_entry_dropout(nullptr, x, out, elems, n_out, make_drop(drop), (hipStream_t)stream);
    return op_done(nullptr);
}

fav_status fav_op_head(const float* logits, int32_t T, int32_t n, int32_t C, int32_t ld, float temperature, int32_t kind,
                       float tau, int32_t* labels, float* conf, uint8_t* fail, float* score, void* stream) {
    if (!logits || !labels || !conf || T < 1 || n < 1 || !(temperature > 0.f)) return op_done("fav_op_head: bad argument");
    return op_done(launch_head(nullptr, logits, T, n, C, ld, temperature, kind, tau, labels, conf, fail, score, (hipStream_t)stream));
}

fav_status fav_op_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t rows, int32_t D,
                            float eps, void* stream) {
    if (!x || !gamma || !beta || !y) return op_done("fav_op_layernorm: null pointer");
    return op_done(launch_layernorm(nullptr, x, ldx, gamma, beta, y, rows, D, eps, (hipStream_t)stream));
}

fav_status fav_op_attention(const void* qkv, void* out, int32_t n, int32_t T, int32_t D, int32_t heads, int32_t math_mode,
                            void* stream) {
    if (!qkv || !out) return op_done("fav_op_attention: null pointer");
    return op_done(launch_attention(nullptr, qkv, out, n, T, D, heads, math_mode, (hipStream_t)stream));
}

fav_status fav_op_linear_streamk(const fav_linear_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->bias || !d->y) return op_done("fav_op_linear_streamk: null pointer");
    if (!launch_gemm_streamk(nullptr, d->x, d->w, d->bias, d->res, d->y, d->rows, d->K, d->N, d->act, (hipStream_t)stream))
        return op_done("fav_op_linear_streamk: shape not supported (K % 32, N % 128, >= 256 tiles of 128 x 128, 32-bit offsets)");
    return op_done(nullptr);
}

fav_status fav_op_vit_assemble(const void* emb, const float* pos, void* x, int32_t n, int32_t ntok, int32_t D, void* stream) {
    if (!emb || !pos || !x || n < 1 || ntok < 2 || D % 4 != 0) return op_done("fav_op_vit_assemble: bad argument");
    launch_vit_assemble(nullptr, emb, pos, x, n, ntok, D, (hipStream_t)stream);
    return op_done(nullptr);
}

fav_status fav_op_signal_stats(const uint8_t* frames, int32_t n, int32_t H, int32_t W, const uint8_t* prev_gray,
                               uint8_t* last_gray, fav_signal_stats* stats, void* stream) {
    static_assert(sizeof(fav_signal_stats) == sizeof(SignalStats), "fav_signal_stats layout");
    if (!frames || !stats || n < 1 || H < 3 || W < 4 || W % 4 != 0 || (long long)H * W > 150000)
        return op_done("fav_op_signal_stats: bad argument (need W % 4 == 0, 3 <= H, H*W <= 150000)");
    const size_t lds = (((size_t)H * W + 15) & ~(size_t)15) + 1024 * 4 + 16 * 8;
    if (hipFuncSetAttribute((const void*)signal_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return op_done("fav_op_signal_stats: cannot reserve LDS for the gray plane");
    hipLaunchKernelGGL(signal_stats_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, frames, n, H, W, prev_gray,
                       last_gray, (SignalStats*)stats);
    return op_done(nullptr);
}

fav_status fav_op_corrupt(const uint8_t* frames, void* out, int32_t n, int32_t H, int32_t W, int32_t mode, float level,
                          float gain, float sigma, uint64_t seed, int64_t first_index, void* stream) {
    if (!frames || !out || n < 1 || H < 1 || W < 1 || mode < 0 || mode > 3) return op_done("fav_op_corrupt: bad argument");
    CorruptParams cp;
    cp.mode = mode; cp.level = level; cp.gain = gain; cp.sigma = sigma;
    cp.seed_lo = (uint32_t)(seed & 0xFFFFFFFFull); cp.seed_hi = (uint32_t)(seed >> 32); cp.first_index = first_index;
    hipLaunchKernelGGL(corrupt_kernel, dim3(grid_for((long long)n * H * W)), dim3(256), 0, (hipStream_t)stream, frames, out,
                       n, H, W, cp);
    return op_done(nullptr);
}

}  // extern "C"
