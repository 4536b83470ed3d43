// Host-side declarations shared by the translation units behind the C ABI (include/ssd_hip.h):
//   abi.hip      lifetime, options, ssd_forward, retained tensors, profiling
//   weights.hip  ssd_finalize: TF variables -> packed device weights (channel order, batch-norm scale factors)
//   plan.hip     the layer plan of one (B,H,W): ops, streams, dependencies; enqueue
//   stages.hip   the stage entry points the parity tests call, anchors, resize arithmetic, diagnostics (-DSSD_DIAG)
// Kernels and their launch wrappers: ssd_internal.h.
#pragma once
#include "../../include/ssd_hip.h"
#include "ssd_internal.h"

#include <climits>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// ----------------------------------------------------------------------------- errors
int ssd_fail(int code, const std::string &msg);        // records the text for ssd_last_error() of this thread, returns code
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return ssd_fail(SSD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define SSDCHK(expr)                                                                         \
    do {                                                                                     \
        int r_ = (expr);                                                                     \
        if (r_ != SSD_OK) return r_;                                                         \
    } while (0)

// ----------------------------------------------------------------------------- options
// Tuning / test switches behind ssd_set_option (include/ssd_hip.h lists them).  None is read from the
// environment; a handle's own value wins over the process-wide one (handle == NULL), which wins over the default.
enum SsdOpt {
    // selectors a caller may want
    OPT_STREAMS = 0,        // 0 auto | 1: every op of a plan on the caller's stream (measurement aid; also the form a caller's own graph capture takes)
    OPT_H2D_CHUNKS,         // 2 (default) | 1 .. 16: pieces of ssd_forward_host's staging copy + upload (piece k uploads under the host copy of k + 1)
    OPT_FRONT_FUSE,         // -1 auto | 0 | 1: the backbone's first layers as one launch (front.hip)
    OPT_FUSE_DW,            // -1 default | mask: depthwise+pointwise pairs that run as one launch
    OPT_BACKBONE_SPLIT,     // 0 auto | 1 | 2: backbone chains (two half-batch chains from 4 images on)
    OPT_EVENT_FENCE,        // 0 (default): the library's ordering events carry no system-scope fence (hipEventDisableSystemFence: they order
                            // streams of ONE device) | 1: default event flags (a cache writeback per record: ~8-12 us per cross-stream edge)
    OPT_PLAN_CACHE_MB,      // 0 auto (a quarter of the device's memory) | n: the layer plans a handle keeps (one per network shape it has
                            // served) may hold up to n MiB of arena; the least recently used ones go first, the running one always stays
    // test hooks: pin a kernel variant / a plan shape so that the parity tests see every shape on it (none changes a result bit in mode f32)
    OPT_IGEMM_TILE,         // 0 auto | 128 | 64: pins the 128x128-vs-64x64 choice of the implicit-GEMM kernel | 20 .. 27: a wave tile of igemm_lat.hip
    OPT_IGEMM16,            // -1 auto | 0 | 1: f16x3 launches on the 256x256-tile kernel
    OPT_IGEMM_96,           // 1 (default) | 0: 128x96 tiles for widths 96 divides and 128 does not
    OPT_IGEMM_LAT,          // 1 (default) | 0: tiny exact-fp32 launches on the latency form (igemm_lat.hip)
    OPT_IGEMM_DEEP64,       // -1 auto | 0 | 1: 64x64 tiles with loads three K-steps ahead
    OPT_LATERAL_SPLIT,      // 1 (default) | 0: f16x3 laterals split fp32 rows while staging them
    OPT_FPN_GROUP,          // -1 auto | 0 | 1: fpn p3 + p4 + p5 as one grouped launch (exact fp32, batch <= 2)
    OPT_FPN_P7_GROUP,       // 1 (default) | 0: fpn p7 as a fourth level of that launch
    OPT_FPN_EARLY_LAT,      // -1 auto (batch <= 2, one backbone chain, exact fp32) | 0 | 1: lateral3 / lateral4 early beside the backbone's last
                            // layers, their top-down sums as one elementwise launch behind lateral5
    OPT_NSUB,               // 0 auto | 1..8: at least this many consecutive sub-batch plans (the split a > 2 GiB batch takes)
    OPT_NMS_FAST_MAX,       // -1 default | n >= 0: candidate lists up to n run in one wave's registers
    OPT_FIRST_CONV_PX,      // 1 (default) | 0: resized frames' first convolution on the lane-per-pixel kernel (elementwise.hip K1d) / on K1, K1c
    OPT_DEBUG_SYNC,         // 0 | 1: announce every op, run it alone, wait for it (fault localisation)
    OPT_COUNT
};
#define SSD_OPT_UNSET INT_MIN
struct Options { int v[OPT_COUNT]; Options() { for (int &x : v) x = SSD_OPT_UNSET; } };
int ssd_opt(const struct ssd_handle *h, int key, int dflt);      // handle value, else process value, else dflt
int ssd_opt_index(const char *key);                              // -1: unknown key
unsigned ssd_sync_event_flags(const struct ssd_handle *h);       // flags of the events that order streams inside the library

// ----------------------------------------------------------------------------- helpers
struct Tensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

struct DevPool {
    std::vector<void *> ptrs;
    size_t bytes = 0;           // what the pool holds (the plan cache's budget counts these)
    int alloc(void **out, size_t nbytes)
    {
        // 256 bytes of slack behind every tensor: igemm_lat.hip's 4-byte-shifted 16-byte loads touch (and never use)
        // the dword behind a tensor's last chunk
        HIPCHK(hipMalloc(out, nbytes + 256));
        ptrs.push_back(*out);
        bytes += nbytes + 256;
        return SSD_OK;
    }
    template <class T> int upload(T **out, const std::vector<T> &v)
    {
        void *p = nullptr;
        SSDCHK(alloc(&p, v.size() * sizeof(T)));
        if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        *out = (T *)p;
        return SSD_OK;
    }
    void free_all()
    {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
        bytes = 0;
    }
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// physical position p -> logical channel (or -1 for a pad channel)
std::vector<int> phys_map(int C, int Cp);
std::vector<int> ident_map(int C, int Cp);
// a ShuffleNet stage output [x half | y half] (D logical channels each, each half in its own standard layout over Dp)
std::vector<int> twopart_map(int D, int Dp);
static inline int twopart_phys(int c, int D, int Dp) { return c < D ? ssd_phys_of_logical(c) : Dp + ssd_phys_of_logical(c - D); }

struct BnHost { std::vector<float> mean, sf, beta; };

struct ConvW {
    float *wt = nullptr, *mean = nullptr, *sf = nullptr, *beta = nullptr, *bias = nullptr;
    float *wlat = nullptr;     // the same kernel in igemm_lat.hip's lane-order pieces (tiny launches, exact fp32)
    float *wt16 = nullptr;     // the same rows in split-fp16 form, scaled by 2^s (precision mode f16x3)
    float *wt16w = nullptr;    // wide outputs whose width 256 does not divide (480 class logits): the S16 rows again,
    int CoutPad16 = 0;         //   padded to CoutPad16 = a multiple of 256 rows per tap for the 256x256-tile kernel
    float scale16 = 1.0f;      // 2^-s
    int CinP = 0, CoutP = 0, CoutPad = 0, taps = 1, tile = IGEMM_128x128;
    int Cin_l = 0, Cout_l = 0;
};

struct DwW {
    float *w = nullptr, *mean = nullptr, *sf = nullptr, *beta = nullptr;
    float *pack = nullptr;      // [Cp/32][12][32]: per 32-channel slice 9 taps, mean, sf, beta (dwpw_stream.hip); Cp % 32 == 0 only
    int Cp = 0;
};

#define SSD_NCLS 8                   // profile classes (include/ssd_hip.h)
// Conv2d_1..4 as one depthwise+pointwise launch: 751.9 -> 760.6 img/s at B=32 (masks 0x3 / 0x5 / 0x7 / 0xf:
// 754.6 / 757.2 / 760.2 / 760.6); from Conv2d_5 on (K >= 256) the two-kernel pair is faster.
#define SSD_FUSE_DW_DEFAULT 0xfu
#define SSD_FUSE_SHUFFLE_DEFAULT true    // ShuffleNet B=64 640x640: depthwise + pointwise 4.19 -> 3.76 ms per step
#ifdef SSD_DIAG                      // libssd_hip_diag.so (scripts/): tile override and phase-stamp buffer of ssd_bench_conv
extern int g_force_tile;
extern long long *g_dbg_ts;
#else                                // the shipped library: compile-time constants, no override exists
static constexpr int g_force_tile = -1;
static constexpr long long *g_dbg_ts = nullptr;
#endif

// weights.hip: w HWIO [k,k,Cin_l,Cout_l] -> wt [taps][CoutPad][CinP]
int pack_conv(const struct ssd_handle *h, DevPool &pool, const float *w, int k, int Cin_l, int Cout_l, const std::vector<int> &inmap,
              const std::vector<int> &outmap, ConvW &cw);
int upload_bn(DevPool &pool, const BnHost &b, ConvW &cw);
int pack_dw(DevPool &pool, const std::vector<float> &w9, const std::vector<float> &mean, const std::vector<float> &sf,
            const std::vector<float> &beta, DwW &d);
int finalize_weights(struct ssd_handle *h);

// ----------------------------------------------------------------------------- ops
struct Op {
    int cls;            // profile class
    int stream = 0;     // 0: the plan's main stream, 1: its second stream, 2 / 3: further chains
    std::vector<int> deps;          // indices of ops (on the other stream) that must have finished
    hipEvent_t done = nullptr;      // recorded after the op when another op depends on it
    double flops, bytes;
    std::function<hipError_t(hipStream_t)> run;
};

struct LevelDesc {
    int H, W, OH, OW;
    long long in_off, out_off, out_bstride;
    int out_rstride, param_off;
    long long res_off;
    long long wt_off = 0;           // float offset of this level's kernel inside the ConvW (grouped launches; 0 = shared)
    int stride = 0, pad = -1;       // > 0 / >= 0: this level's own stride / pad_beg (else the launch's)
};

// in_fmt / out_fmt / res_fmt: 0 fp32 rows, 1 split-fp16 rows (ssd_internal.h); flags: the handle's status word
Op make_conv_op(const struct ssd_handle *h, const ConvW &cw, const float *in, float *out, float *out2, const float *res, int B,
                int stride, int pad, int act, const std::vector<LevelDesc> &lv, bool dense, int in_fmt = 0, int out_fmt = 0,
                int res_fmt = 0, int *flags = nullptr, unsigned *scan_bits = nullptr, float scan_lo = 0.0f,
                bool *scan_marked = nullptr);
Op make_dw_op(const DwW &d, const float *in, int B, int H, int W, int stride, int act, float *out, int Cl, int out16 = 0,
              int *flags = nullptr);
bool dwpws_eligible(const DwW &d, const ConvW &cw, int B, int H, int W, int stride);
Op make_dwpws_op(const DwW &d, const ConvW &cw, const float *in, int B, int H, int W, int stride, int dact, int act,
                 float *out, int out_rs = 0 /* floats between output rows; 0: cw.CoutP */);
// sn_pw.hip: 1x1 + batch norm + activation on rows gathered through `src` (device table, CinP entries) from `base`
Op make_pw_gather_op(const ConvW &cw, const float *base, long long base_bytes, const int *src, int rs, long long M, int act, float *out);
// front.hip: first convolution + Conv2d_1 in one launch; the frame pointer is the handle's cur_images + img_index frames at run time
Op make_front_op(struct ssd_handle *h, int img_index, const DwW &f, int act0, const DwW &d, const ConvW &cw, int B, int H, int W, int dact, int act, float *out);
LevelDesc dense_level(int H, int W, int OH, int OW, int CoutP, long long in_off = 0, long long out_off = 0,
                      int param_off = 0, long long res_off = 0);
float conservative_logit_bound(float thr);
int nms_fast_max(const struct ssd_handle *h);        // PostArgs encoding: 0 = default, -1 = "0"

// resize_keeping_aspect_ratio (pipeline.py:138-194), the size arithmetic of the TF graph
struct ResizeDims { int nh, nw, ph, pw; float box_scaler[4]; };
ResizeDims resize_dims(int height, int width, int min_dimension, int divisor);
extern const int A_STRIDES[5];
extern const int MB_STRIDE[13];          // mobilenet_v1.py:52-58 (weights.hip)

// ----------------------------------------------------------------------------- handle
struct Retained { const float *dev; int B, H, W, C, Cp; bool permuted; int fmt = 0; /* 1: split-fp16 rows */
                  int split = 0; /* > 0: two-part rows (a ShuffleNet stage output): the first `split` logical channels in their own
                                    standard layout over Cp / 2 physical channels, the rest likewise behind them */ };

struct EvPair { hipEvent_t a, b; int cls; int fwd; };

// The layer plan of one SUB-BATCH (normally the whole batch: consecutive sub-batch plans exist for batches whose tensors would
// pass the 2 GiB that a launch addresses with 32-bit byte offsets).
struct Plan {
    int B = 0, img0 = 0, N = 0;
    DevPool pool;                       // activations / workspace
    std::vector<Op> ops;
    PostArgs post;
    std::map<std::string, Retained> retained;
    hipStream_t s_aux = nullptr;        // Op::stream 1: class tower beside the box tower (box_predictor.py:47-59), second backbone chain
    hipStream_t s_bb[2] = {nullptr, nullptr};   // Op::stream 2, 3 (fpn p6 -> p7; the batch-1 lateral chain); process-wide, not owned
    hipEvent_t ev_join = nullptr, ev_begin = nullptr;
    bool need_begin = false;            // some chain starts on an internal stream without a dependency: it waits for ev_begin
};

// What a forward takes from the SOURCE frames (resize_keeping_aspect_ratio, pipeline.py:138-194; model.py:67-68): launch arguments
// of the first kernel and of the pack kernel, set per call -- NOT part of a plan, which is keyed on what the network sees.
struct SrcGeom { int srcH = 0, srcW = 0, nh = 0, nw = 0; float box_scaler[4] = {1, 1, 1, 1}; };

// A batch of frames of DIFFERENT sizes that resize to one network shape (ssd_forward_mixed): per frame its geometry for the first
// kernel and its box_scaler {y, x} for the pack kernel; valid while the forward is being enqueued (the handle's mutex is held).
struct MixedCall { MixedGeom geom; float scaler[SSD_MIXED_MAX][2]; };

// The plans of one network shape: key = (batch, the resized + padded size the network sees, whether the source already has that
// size -- then the first layers are the fused front launch --, and the number of consecutive sub-batch plans).
struct PlanKey {
    int B = 0, netH = 0, netW = 0, ident = 0, nsub = 0;
    bool operator==(const PlanKey &o) const { return B == o.B && netH == o.netH && netW == o.netW && ident == o.ident && nsub == o.nsub; }
};
struct PlanSet {
    PlanKey key;
    std::vector<Plan *> plans;          // one, or consecutive sub-batch plans
    size_t bytes = 0;                   // arena of all of them
    unsigned long long last_use = 0;    // the handle's use clock at the last forward that ran this set
};

struct ssd_handle {
    ssd_config cfg;
    std::mutex mu;                      // every entry point that takes the handle holds it: calls on one handle are serialised
    Options opts;
    std::map<std::string, Tensor> vars;
    bool finalized = false;
    DevPool wpool;      // weights
    // packed weights
    DwW first;                          // first conv (w = [27][CoutP])
    int firstCp = 0, firstAct = SSD_ACT_RELU6;
    std::vector<DwW> dw;                // depthwise layers in execution order
    std::vector<ConvW> pw;              // backbone pointwise layers in execution order
    ConvW lat[3], pconv[5];             // fpn lateral3..5, p3..p7
    ConvW pgroup;                       // fpn p3 | p4 | p5 | p7 kernels and batch norms behind one pointer each: one grouped launch at batch 1
    ConvW tower[2][4], final_[2];       // [box, class]
    int c_ch[3] = {0, 0, 0};            // logical channels of c3, c4, c5
    int c_split[3] = {0, 0, 0};         // > 0: c3 / c4 is a ShuffleNet stage output in two-part rows [x half | y half], this many channels each
    int precision = SSD_PRECISION_F32;  // ssd_set_precision
    int *flags_dev = nullptr;           // status word (bit 0: an S16 tensor was clamped to the fp16 range)
    // Layer plans: one set per network shape this handle has served, each with an arena of its own, kept until the budget
    // (option plan_cache_mb) or an option / precision change evicts it.  The reference's graph takes any image size in one
    // session (detector/ssd.py:27-31, create_pb.py:24,40); a mix of sizes must not re-plan per call.
    std::vector<PlanSet *> cache;
    PlanSet *cur = nullptr;             // the set of the last forward (ssd_get_tensor reads its retained tensors)
    unsigned long long use_clock = 0;
    long long cache_hits = 0, cache_misses = 0, cache_evictions = 0;
    const uint8_t *cur_images = nullptr;
    SrcGeom src;                        // of the forward being enqueued (read by the ops' launch closures, like cur_images)
    const MixedCall *mixed = nullptr;   // non-null while a mixed-size batch is being enqueued: per-frame geometry instead of `src`
    MixedCall mixed_store;
    // the plans share the process's internal streams: a forward enqueued on another stream than the previous one waits for it
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    bool multi_stream = false;          // forwards have arrived on more than one stream: ev_last is recorded behind every forward
    hipEvent_t ev_last = nullptr;
    const void *detect_rec_ok[8] = {nullptr};     // ssd_detect_host: record pointers verified to be host-visible memory
    int n_rec_ok = 0, rec_ok_next = 0;
    // ssd_forward_host: pinned staging + device image, grown on demand; the stream whose copy last read the staging buffer
    uint8_t *stage_pin = nullptr, *stage_dev = nullptr;
    size_t stage_bytes = 0;
    hipStream_t stage_stream = nullptr;
    bool stage_busy = false, stage_stream_set = false;      // (the NULL stream is a stream too)
    bool stage_by_event = false;         // the busy staging buffer is released by ev_stage (else: by draining stage_stream, ssd_detect_host)
    hipEvent_t ev_stage = nullptr;       // recorded behind the last upload of a call: the PINNED buffer is free again once it has passed
                                         // (the device image is protected by stream order: the next upload queues behind the forward that reads it)
    // profiling
    bool profiling = false;
    std::vector<EvPair> evs;
    std::vector<hipEvent_t> ref_evs;     // one reference event per profiled forward
    std::vector<hipEvent_t> ev_pool;     // timing events, created once and reused (none is created inside a timed region
                                         // after the first profiled forward)
    double acc_ms[SSD_NCLS] = {0}, acc_flops[SSD_NCLS] = {0}, acc_bytes[SSD_NCLS] = {0};
    long long acc_n[SSD_NCLS] = {0};
};

// plan.hip
void free_plans(ssd_handle *h);                         // every cached set (the device must be idle)
int select_plans(ssd_handle *h, int B, int H, int W);   // h->cur = the set of this source shape: cache hit, or build (+ evict)
// ... of a network shape directly (a mixed-size batch: ident = 0, max_src_bytes = its largest frame)
int select_plans_net(ssd_handle *h, int B, int netH, int netW, int ident, long long max_src_bytes);
int trim_plan_cache(ssd_handle *h, const PlanSet *keep);
size_t plan_cache_limit_bytes(const ssd_handle *h);
int enqueue_forward(ssd_handle *h, const uint8_t *images_dev, float *boxes_dev, int32_t *labels_dev, float *scores_dev,
                    int32_t *num_boxes_dev, long long out_stride, hipStream_t s);
