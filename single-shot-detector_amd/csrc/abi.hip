// The C ABI of include/ssd_hip.h, handle side: lifetime, options, ssd_forward (= the frozen graph's sess.run),
// retained tensors, per-class profiling.  Every entry point that takes a handle holds the handle's mutex.
// No CPU fallback exists: every entry point either launches HIP kernels or fails.
#include "host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

// ----------------------------------------------------------------------------- errors
static thread_local std::string g_err;
int ssd_fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *ssd_last_error(void) { return g_err.c_str(); }

// ----------------------------------------------------------------------------- options
static Options g_opts;                 // process-wide values (ssd_set_option with a NULL handle)
static std::mutex g_opts_mu;
static const char *const OPT_NAMES[OPT_COUNT] = {"streams", "h2d_chunks", "front_fuse", "fuse_dw", "backbone_split", "event_fence", "plan_cache_mb",
                                                 "igemm_tile", "igemm16", "igemm_96", "igemm_lat", "igemm_deep64", "lateral_split", "fpn_group", "fpn_p7_group",
                                                 "fpn_early_lat", "nsub", "nms_fast_max", "first_conv_px", "debug_sync"};
int ssd_opt_index(const char *key)
{
    for (int i = 0; i < OPT_COUNT; ++i)
        if (key && !strcmp(key, OPT_NAMES[i])) return i;
    return -1;
}
int ssd_opt(const ssd_handle *h, int key, int dflt)
{
    if (h && h->opts.v[key] != SSD_OPT_UNSET) return h->opts.v[key];
    const int g = g_opts.v[key];
    return g != SSD_OPT_UNSET ? g : dflt;
}
// Events that order the library's streams against each other live on ONE device: no system-scope fence (the default flags
// make every record write the caches back -- measured as 8-12 us between the kernels on either side of a record or a wait,
// three of them on the critical path of a batch-1 forward).  Host visibility of the results is the caller's stream
// synchronisation, as before.
unsigned ssd_sync_event_flags(const ssd_handle *h)
{
    return hipEventDisableTiming | (ssd_opt(h, OPT_EVENT_FENCE, 0) ? 0u : (unsigned)hipEventDisableSystemFence);
}
int nms_fast_max(const ssd_handle *h)
{
    const int v = ssd_opt(h, OPT_NMS_FAST_MAX, -1);
    return v < 0 ? 0 : (v == 0 ? -1 : v);
}

// The handle's own ordering event (previous forward on another stream) with the flags the handle's options ask for; called by
// ssd_create and again when option event_fence changes on the handle (device idle, plans freed).
static int make_handle_events(ssd_handle *h)
{
    if (h->ev_last) { (void)hipEventDestroy(h->ev_last); h->ev_last = nullptr; }
    HIPCHK(hipEventCreateWithFlags(&h->ev_last, ssd_sync_event_flags(h)));
    h->have_last = false;           // (ev_last is new: nothing is recorded in it; the device is idle)
    return SSD_OK;
}

extern "C" int ssd_set_option(ssd_handle *h, const char *key, int32_t value)
{
    const int k = ssd_opt_index(key);
    if (k < 0) return ssd_fail(SSD_ERR_INVALID, std::string("ssd_set_option: unknown option ") + (key ? key : "(null)"));
    // (SSD_OPT_UNSET = INT32_MIN puts an option back to "not set")
    if (k == OPT_IGEMM_TILE && !(value == SSD_OPT_UNSET || value == 0 || value == 64 || value == 128 || igemm_is_lat(value)))
        return ssd_fail(SSD_ERR_INVALID, "ssd_set_option: igemm_tile must be 0 (auto), 128, 64 or a wave tile of igemm_lat.hip (20 .. 27, 30)");
    if (k == OPT_PLAN_CACHE_MB && value < 0 && value != SSD_OPT_UNSET) return ssd_fail(SSD_ERR_INVALID, "ssd_set_option: plan_cache_mb must be >= 0");
    if (!h) {
        std::lock_guard<std::mutex> g(g_opts_mu);
        g_opts.v[k] = value;
        return SSD_OK;
    }
    std::lock_guard<std::mutex> g(h->mu);
    if (h->opts.v[k] == value) return SSD_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (k == OPT_PLAN_CACHE_MB) {   // the budget of the plan cache: nothing a plan depends on -- set it and evict down to it
        h->opts.v[k] = value;
        return trim_plan_cache(h, h->cur);
    }
    HIPCHK(hipDeviceSynchronize());
    free_plans(h);                  // the layer plans depend on the options: every cached one goes
    h->opts.v[k] = value;
    if (k == OPT_EVENT_FENCE) SSDCHK(make_handle_events(h));      // the handle's own ordering events follow the option too
    return SSD_OK;
}

extern "C" int ssd_get_option(ssd_handle *h, const char *key, int32_t *value)
{
    const int k = ssd_opt_index(key);
    if (k < 0 || !value) return ssd_fail(SSD_ERR_INVALID, "ssd_get_option: unknown option or null argument");
    *value = ssd_opt(h, k, SSD_OPT_UNSET);
    return SSD_OK;
}

// ----------------------------------------------------------------------------- lifetime
extern "C" int ssd_create(const ssd_config *cfg, ssd_handle **out)
{
    if (!cfg || !out) return ssd_fail(SSD_ERR_INVALID, "ssd_create: null argument");
    if (cfg->backbone != SSD_BACKBONE_MOBILENET && cfg->backbone != SSD_BACKBONE_SHUFFLENET)
        return ssd_fail(SSD_ERR_INVALID, "ssd_create: unknown backbone");
    if (cfg->num_classes < 1 || cfg->max_boxes_per_class < 1)
        return ssd_fail(SSD_ERR_INVALID, "ssd_create: num_classes and max_boxes_per_class must be >= 1");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return ssd_fail(SSD_ERR_INVALID, "ssd_create: no such HIP device");
    HIPCHK(hipSetDevice(cfg->device));
    ssd_handle *h = new ssd_handle();
    h->cfg = *cfg;
    if (make_handle_events(h) != SSD_OK) {
        if (h->ev_last) (void)hipEventDestroy(h->ev_last);
        delete h;
        return ssd_fail(SSD_ERR_HIP, "ssd_create: cannot create events");
    }
    if (hipMalloc((void **)&h->flags_dev, sizeof(int)) != hipSuccess || hipMemset(h->flags_dev, 0, sizeof(int)) != hipSuccess) {
        delete h;
        return ssd_fail(SSD_ERR_HIP, "ssd_create: cannot allocate the status word");
    }
    if (const char *e = getenv("SSD_PRECISION")) {     // default for handles that never call ssd_set_precision
        if (!strcmp(e, "f16x3")) h->precision = SSD_PRECISION_F16X3;
        else if (!strcmp(e, "f32")) h->precision = SSD_PRECISION_F32;
        else { (void)hipFree(h->flags_dev); delete h; return ssd_fail(SSD_ERR_INVALID, "SSD_PRECISION must be f32 or f16x3"); }
    }
    *out = h;
    return SSD_OK;
}

extern "C" int ssd_set_precision(ssd_handle *h, int32_t mode)
{
    if (!h || (mode != SSD_PRECISION_F32 && mode != SSD_PRECISION_F16X3)) return ssd_fail(SSD_ERR_INVALID, "ssd_set_precision: bad arguments");
    std::lock_guard<std::mutex> g(h->mu);
    if (mode == h->precision) return SSD_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    free_plans(h);                  // the layer plans (tensor formats, kernels) depend on the mode: every cached one goes
    h->precision = mode;
    return SSD_OK;
}

extern "C" int ssd_get_precision(ssd_handle *h)
{
    if (!h) return SSD_ERR_INVALID;
    std::lock_guard<std::mutex> g(h->mu);
    return h->precision;
}

extern "C" int ssd_status(ssd_handle *h, int32_t *flags_out)
{
    if (!h || !flags_out) return ssd_fail(SSD_ERR_INVALID, "ssd_status: null argument");
    std::lock_guard<std::mutex> g(h->mu);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    int v = 0;
    HIPCHK(hipMemcpy(&v, h->flags_dev, sizeof(int), hipMemcpyDeviceToHost));
    if (v) HIPCHK(hipMemset(h->flags_dev, 0, sizeof(int)));
    *flags_out = v;
    return SSD_OK;
}

extern "C" void ssd_destroy(ssd_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto &e : h->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto r : h->ref_evs) (void)hipEventDestroy(r);
    for (auto r : h->ev_pool) (void)hipEventDestroy(r);
    free_plans(h);
    if (h->ev_last) (void)hipEventDestroy(h->ev_last);
    if (h->ev_stage) (void)hipEventDestroy(h->ev_stage);
    if (h->stage_pin) (void)hipHostFree(h->stage_pin);
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    if (h->flags_dev) (void)hipFree(h->flags_dev);
    h->wpool.free_all();
    delete h;
}

extern "C" int ssd_load_weight(ssd_handle *h, const char *name, const float *host, const int64_t *shape, int32_t ndim)
{
    if (!h || !name || !host || !shape || ndim < 1 || ndim > 4) return ssd_fail(SSD_ERR_INVALID, "ssd_load_weight: bad arguments");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return ssd_fail(SSD_ERR_STATE, "ssd_load_weight after ssd_finalize");
    Tensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 1) return ssd_fail(SSD_ERR_INVALID, "ssd_load_weight: non-positive dimension");
        t.shape.push_back(shape[i]);
        n *= shape[i];
    }
    t.data.assign(host, host + n);
    h->vars[name] = std::move(t);
    return SSD_OK;
}

extern "C" int ssd_finalize(ssd_handle *h)
{
    if (!h) return ssd_fail(SSD_ERR_INVALID, "ssd_finalize: null handle");
    std::lock_guard<std::mutex> g(h->mu);
    if (h->finalized) return ssd_fail(SSD_ERR_STATE, "ssd_finalize called twice");
    HIPCHK(hipSetDevice(h->cfg.device));
    SSDCHK(finalize_weights(h));
    h->vars.clear();   // host copies no longer needed
    h->finalized = true;
    return SSD_OK;
}

// One forward on stream `s` with the handle's mutex held.  The plans of a handle share the process's internal streams (and a
// shape's arena is one): when this forward is enqueued on another stream than the previous one it first waits for that one's
// last kernel (two host threads sharing a Detector on their own streams, as tf.Session.run allows, inference/detector.py:34,52).
// (a mixed-size batch: `mx` = its per-frame geometry, the network shape all frames resize to, its largest frame's bytes)
struct MixedSel { const MixedCall *mx; int netH, netW; long long max_bytes; };

static int forward_locked(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, float *boxes_dev,
                          int32_t *labels_dev, float *scores_dev, int32_t *num_boxes_dev, long long out_stride, hipStream_t s,
                          const MixedSel *ms = nullptr)
{
    HIPCHK(hipSetDevice(h->cfg.device));
    // the plans of this shape: kept from an earlier call (no HIP call at all), or built now beside the others
    if (ms) SSDCHK(select_plans_net(h, B, ms->netH, ms->netW, 0, ms->max_bytes));
    else SSDCHK(select_plans(h, B, H, W));
    // a forward on another stream than the previous one waits for it.  The event is recorded HERE, at the
    // current tail of the previous stream (everything the previous forward enqueued there precedes it), not at the end of
    // every forward: a record behind the last kernel is one more packet the caller's synchronisation waits for.
    // Single-stream steady state (the usual caller): nothing is ever recorded.  The FIRST time a forward arrives on another
    // stream the event is recorded lazily at the tail of the previous stream -- unless that stream is being captured by the
    // caller or is gone (then the device is drained instead) -- and from then on the handle is `multi_stream`: every forward
    // records ev_last right behind its own last kernel (forward_checked_locked), so a later forward on another stream never
    // touches a stream handle it does not own any more and never waits for unrelated work queued there afterwards.
    if (h->have_last && h->last_stream != s) {
        if (!h->multi_stream) {
            h->multi_stream = true;
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool known = hipStreamIsCapturing(h->last_stream, &cs) == hipSuccess;
            if (known && cs == hipStreamCaptureStatusNone && hipEventRecord(h->ev_last, h->last_stream) == hipSuccess)
                HIPCHK(hipStreamWaitEvent(s, h->ev_last, 0));
            else { (void)hipGetLastError(); HIPCHK(hipDeviceSynchronize()); }      // (captured, destroyed or unknown stream)
        } else {
            // (ev_last was recorded outside any capture: a stream the caller is capturing into a graph cannot wait for it -- and a
            //  graph must not depend on work outside it.  Refused with a message instead of a HIP capture error mid-enqueue.)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
            if (cs != hipStreamCaptureStatusNone)
                return ssd_fail(SSD_ERR_STATE, "ssd_forward: this handle's previous forward ran on another stream; a capturing stream cannot wait "
                                               "for it -- synchronise that stream (or use one handle per captured stream) before capturing");
            HIPCHK(hipStreamWaitEvent(s, h->ev_last, 0));
        }
        h->have_last = false;
    }
    // (hipGraph replay of a repeating forward was measured in rounds 1-2 -- batch 1: replay 2.49 ms against eager 2.33, the forward
    //  is GPU-bound -- and is not part of the library.)
    h->mixed = ms ? ms->mx : nullptr;
    const int rc = enqueue_forward(h, images_dev, boxes_dev, labels_dev, scores_dev, num_boxes_dev, out_stride, s);
    h->mixed = nullptr;
    return rc;
}

// (the caller holds h->mu)
static int forward_checked_locked(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, float *boxes_dev,
                                  int32_t *labels_dev, float *scores_dev, int32_t *num_boxes_dev, long long out_stride, void *stream,
                                  const MixedSel *ms = nullptr)
{
    if (!h->finalized) return ssd_fail(SSD_ERR_STATE, "ssd_forward before ssd_finalize");
    if (B < 1 || H < 1 || W < 1 || h->cfg.min_dimension < 128 || h->cfg.min_dimension % 128)
        return ssd_fail(SSD_ERR_INVALID, "ssd_forward: B, H, W must be positive and min_dimension a multiple of 128 (pipeline.py:152)");
    {
        const ResizeDims rd = resize_dims(H, W, h->cfg.min_dimension, 128);
        if ((long long)(rd.nh + rd.ph) * (rd.nw + rd.pw) > (1LL << 26))
            return ssd_fail(SSD_ERR_INVALID, "ssd_forward: aspect ratio too extreme (resized image exceeds 64 Mpixel)");
    }
    hipStream_t s = (hipStream_t)stream;
    const int rc = forward_locked(h, images_dev, B, H, W, boxes_dev, labels_dev, scores_dev, num_boxes_dev, out_stride, s, ms);
    if (rc == SSD_OK) {
        // (not while a caller captures `s` into a graph of its own: the stream then is not a queue of the device's)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            h->last_stream = s;
            h->have_last = true;
            if (h->multi_stream) HIPCHK(hipEventRecord(h->ev_last, s));       // (a handle that has seen two streams: eager record)
        }
    }
    return rc;
}

static int forward_checked(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, float *boxes_dev,
                           int32_t *labels_dev, float *scores_dev, int32_t *num_boxes_dev, long long out_stride, void *stream)
{
    if (!h || !images_dev || !boxes_dev || !labels_dev || !scores_dev || !num_boxes_dev)
        return ssd_fail(SSD_ERR_INVALID, "ssd_forward: null argument");
    std::lock_guard<std::mutex> g(h->mu);
    return forward_checked_locked(h, images_dev, B, H, W, boxes_dev, labels_dev, scores_dev, num_boxes_dev, out_stride, stream);
}

extern "C" int ssd_forward(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, float *boxes_dev,
                           int32_t *labels_dev, float *scores_dev, int32_t *num_boxes_dev, void *stream)
{
    return forward_checked(h, images_dev, B, H, W, boxes_dev, labels_dev, scores_dev, num_boxes_dev, 0, stream);
}

// The same graph with its outputs as B fixed RECORDS (what the all-gather of a data-parallel step moves), record b
// at records_dev + b * ssd_record_words(h) 32-bit words:
//     boxes [T,4] f32 | scores [T] f32 | labels [T] i32 | num_boxes i32        T = num_classes * max_boxes_per_class
// (48 004 bytes at T = 2 000).  One block instead of four tensors: a rank's output pointer can be its slice of the all-gather's
// receive buffer (the collective runs in place) and one copy moves a batch's results to the host.
extern "C" int32_t ssd_record_words(const ssd_handle *h)
{
    return h ? 6 * h->cfg.num_classes * h->cfg.max_boxes_per_class + 1 : 0;
}

extern "C" int ssd_forward_records(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W, void *records_dev, void *stream)
{
    if (!h || !records_dev) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_records: null argument");
    if ((reinterpret_cast<uintptr_t>(records_dev) & 3) != 0) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_records: records must be 4-byte aligned");
    const long long T = (long long)h->cfg.num_classes * h->cfg.max_boxes_per_class;
    int32_t *r = (int32_t *)records_dev;
    return forward_checked(h, images_dev, B, H, W, (float *)r, r + 5 * T, (float *)(r + 4 * T), r + 6 * T, 6 * T + 1, stream);
}

// Before the host writes the pinned staging buffer again: the previous call's uploads must have read it.  Only the uploads -- not the
// forward behind them (round 6: a caller that stages batch k + 1 while batch k computes, Detector.detect_many, is not held up; the
// wait used to be for the whole stream).  A new stream's uploads queue behind nothing of the old one: drain the old one's event first.
static int stage_acquire(ssd_handle *h)
{
    if (!h->stage_busy) return SSD_OK;
    const hipError_t e = h->stage_by_event && h->ev_stage ? hipEventSynchronize(h->ev_stage) : hipStreamSynchronize(h->stage_stream);
    if (e != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipDeviceSynchronize()); }   // (the caller destroyed that stream)
    h->stage_busy = false;
    return SSD_OK;
}
// by_event = false: the caller drains `s` itself right after the forward (ssd_detect_host): nothing is recorded -- an event record
// between the upload and the first kernel is one more packet on the critical path of a batch-1 call.
static int stage_release(ssd_handle *h, hipStream_t s, bool by_event)
{
    if (by_event) {
        if (!h->ev_stage) HIPCHK(hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) HIPCHK(hipEventRecord(h->ev_stage, s));
        else by_event = false;
    }
    h->stage_stream = s;
    h->stage_stream_set = true;
    h->stage_by_event = by_event;
    h->stage_busy = true;
    return SSD_OK;
}

// The handle's staging pair (pinned host + device image): grow-only, in powers of two from 2 MiB (a mix of image sizes settles after
// a few calls: a 640 x 480 frame is 0.9 MB, the largest COCO frame 1.2 MB); the device drains because the previous upload's buffers
// are freed.
static int grow_stage(ssd_handle *h, size_t bytes)
{
    if (bytes <= h->stage_bytes) return SSD_OK;
    size_t cap = (size_t)2 << 20;
    while (cap < bytes) cap <<= 1;
    HIPCHK(hipDeviceSynchronize());
    if (h->stage_pin) { (void)hipHostFree(h->stage_pin); h->stage_pin = nullptr; }
    if (h->stage_dev) { (void)hipFree(h->stage_dev); h->stage_dev = nullptr; }
    h->stage_bytes = 0;
    HIPCHK(hipHostMalloc((void **)&h->stage_pin, cap + 256, hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&h->stage_dev, cap + 256));
    h->stage_bytes = cap;
    return SSD_OK;
}

// The boundary's own form (inference/detector.py:51-52: a HOST image in on every call): pageable host memory -> the handle's
// pinned staging buffer -> its device image, in `h2d_chunks` pieces so that piece k crosses the bus under the host copy of
// piece k + 1 -- one C loop instead of a Python one (two numpy / torch calls per piece cost more than a piece's copy) --
// then ssd_forward_records on the same stream.  The host copy is synchronous (on return `images_host` may be reused), the rest
// asynchronous on `stream`.
static int forward_host_impl(ssd_handle *h, const uint8_t *images_host, int32_t B, int32_t H, int32_t W, void *records, void *stream, bool stage_event)
{
    if (!h || !images_host || !records) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_host: null argument");
    if (B < 1 || H < 1 || W < 1) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_host: B, H, W must be positive");
    if ((reinterpret_cast<uintptr_t>(records) & 3) != 0) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_host: records must be 4-byte aligned");
    std::lock_guard<std::mutex> g(h->mu);
    HIPCHK(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const size_t bytes = (size_t)B * H * W * 3;
    // the previous call's upload may still be reading the staging buffer when the caller did not wait for it
    SSDCHK(stage_acquire(h));
    if (h->stage_stream_set && h->stage_stream != s) {
        // (another stream than last time: its uploads into the DEVICE image would not queue behind the forward that still reads it)
        if (hipStreamSynchronize(h->stage_stream) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipDeviceSynchronize()); }
    }
    SSDCHK(grow_stage(h, bytes));
    int nchunk = ssd_opt(h, OPT_H2D_CHUNKS, 2);      // (measured: 1 / 2 / 3 / 4 / 6 / 8 pieces -> Detector p50 1.695 / 1.678 / 1.691 / 1.698 / 1.717 / 1.735 ms: a hipMemcpyAsync costs the host ~10 us)
    if (nchunk < 1) nchunk = 1;
    if (nchunk > 16) nchunk = 16;
    if (bytes < ((size_t)1 << 20)) nchunk = 1;
    const size_t step = ((bytes + nchunk - 1) / nchunk + 4095) & ~(size_t)4095;
    for (size_t lo = 0; lo < bytes; lo += step) {
        const size_t n = bytes - lo < step ? bytes - lo : step;
        memcpy(h->stage_pin + lo, images_host + lo, n);
        HIPCHK(hipMemcpyAsync(h->stage_dev + lo, h->stage_pin + lo, n, hipMemcpyHostToDevice, s));
    }
    SSDCHK(stage_release(h, s, stage_event));
    const long long T = (long long)h->cfg.num_classes * h->cfg.max_boxes_per_class;
    int32_t *r = (int32_t *)records;
    return forward_checked_locked(h, h->stage_dev, B, H, W, (float *)r, r + 5 * T, (float *)(r + 4 * T), r + 6 * T, 6 * T + 1, stream);
}

extern "C" int ssd_forward_host(ssd_handle *h, const uint8_t *images_host, int32_t B, int32_t H, int32_t W, void *records, void *stream)
{
    return forward_host_impl(h, images_host, B, H, W, records, stream, true);
}

// Frames of DIFFERENT sizes as ONE batch (round 6).  The reference's graph is fed one image at a time because a tensor has one
// height and width (create_pb.py:40: [None, None, None, 3]); what the network sees, though, is the size AFTER
// resize_keeping_aspect_ratio (pipeline.py:138-194), and frames of many source sizes share it (480 x 640, 375 x 500, 333 x 500 ->
// 640 x 896): their first kernel reads every frame through its own geometry, everything behind it is the ordinary batched plan of
// that network shape, and the pack kernel divides every image's boxes by its own box_scaler (model.py:67-68).  Image b of the
// result is bit for bit what the frame gives alone.
// Fills h->mixed_store from hw [B][2] (+ offsets, NULL = frames back to back) and the selector; every frame must resize to one shape.
static int prepare_mixed(ssd_handle *h, int32_t B, const int32_t *hw, const int64_t *offsets, MixedSel *ms, const char *who)
{
    if (B < 1 || B > SSD_MIXED_MAX)
        return ssd_fail(SSD_ERR_INVALID, std::string(who) + ": 1 .. " + std::to_string(SSD_MIXED_MAX) + " frames per mixed-size batch");
    if (h->cfg.min_dimension < 128 || h->cfg.min_dimension % 128) return ssd_fail(SSD_ERR_INVALID, std::string(who) + ": min_dimension must be a multiple of 128");
    MixedCall &mc = h->mixed_store;
    long long off = 0, max_bytes = 0;
    ms->mx = &mc; ms->netH = ms->netW = 0;
    for (int b = 0; b < B; ++b) {
        const int H = hw[2 * b], W = hw[2 * b + 1];
        if (H < 1 || W < 1) return ssd_fail(SSD_ERR_INVALID, std::string(who) + ": frame sizes must be positive");
        const ResizeDims rd = resize_dims(H, W, h->cfg.min_dimension, 128);
        const int nH = rd.nh + rd.ph, nW = rd.nw + rd.pw;
        if ((long long)nH * nW > (1LL << 26)) return ssd_fail(SSD_ERR_INVALID, std::string(who) + ": aspect ratio too extreme (resized image exceeds 64 Mpixel)");
        if (b == 0) { ms->netH = nH; ms->netW = nW; }
        else if (nH != ms->netH || nW != ms->netW)
            return ssd_fail(SSD_ERR_INVALID, std::string(who) + ": frame " + std::to_string(b) + " (" + std::to_string(H) + " x " + std::to_string(W) + ") resizes to " +
                                             std::to_string(nH) + " x " + std::to_string(nW) + ", frame 0 to " + std::to_string(ms->netH) + " x " +
                                             std::to_string(ms->netW) + ": the frames of one batch must share the network shape (group them by ssd_network_shape)");
        const long long bytes = (long long)H * W * 3;
        const long long o = offsets ? offsets[b] : off;
        if (o < 0 || o + bytes >= (1LL << 31)) return ssd_fail(SSD_ERR_INVALID, std::string(who) + ": the frames of a batch must lie within 2 GiB of the base pointer");
        FrameGeom &g = mc.geom.f[b];
        g.off = (unsigned)o; g.srcH = H; g.srcW = W; g.nh = rd.nh; g.nw = rd.nw;
        g.hs = (float)H / (float)rd.nh; g.ws = (float)W / (float)rd.nw;       // (launch_first_conv's own expressions)
        mc.scaler[b][0] = rd.box_scaler[0]; mc.scaler[b][1] = rd.box_scaler[1];
        off = o + ((bytes + 15) & ~15LL);                                    // back to back: every frame on a 16-byte boundary
        max_bytes = bytes > max_bytes ? bytes : max_bytes;
    }
    ms->max_bytes = max_bytes;
    return SSD_OK;
}

// The network's input size of a [height, width] frame (resize_keeping_aspect_ratio with this handle's min_dimension, padded to
// multiples of 128): what a caller groups frames by before ssd_forward_mixed.
extern "C" int ssd_network_shape(ssd_handle *h, int32_t height, int32_t width, int32_t *net_hw)
{
    if (!h || !net_hw || height < 1 || width < 1) return ssd_fail(SSD_ERR_INVALID, "ssd_network_shape: bad arguments");
    if (h->cfg.min_dimension < 128 || h->cfg.min_dimension % 128) return ssd_fail(SSD_ERR_INVALID, "ssd_network_shape: min_dimension must be a multiple of 128");
    const ResizeDims rd = resize_dims(height, width, h->cfg.min_dimension, 128);
    net_hw[0] = rd.nh + rd.ph; net_hw[1] = rd.nw + rd.pw;
    return SSD_OK;
}

extern "C" int ssd_forward_mixed(ssd_handle *h, const uint8_t *images_dev, int32_t B, const int32_t *hw_host, const int64_t *offsets_host,
                                 void *records_dev, void *stream)
{
    if (!h || !images_dev || !hw_host || !records_dev) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_mixed: null argument");
    if ((reinterpret_cast<uintptr_t>(records_dev) & 3) != 0) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_mixed: records must be 4-byte aligned");
    std::lock_guard<std::mutex> g(h->mu);
    MixedSel ms;
    SSDCHK(prepare_mixed(h, B, hw_host, offsets_host, &ms, "ssd_forward_mixed"));
    const long long T = (long long)h->cfg.num_classes * h->cfg.max_boxes_per_class;
    int32_t *r = (int32_t *)records_dev;
    return forward_checked_locked(h, images_dev, B, hw_host[0], hw_host[1], (float *)r, r + 5 * T, (float *)(r + 4 * T), r + 6 * T, 6 * T + 1, stream, &ms);
}

// ... fed from host memory: frames_host[b] = frame b, uint8 [hw[b][0], hw[b][1], 3].  The frames are staged back to back (16-byte
// aligned) through the handle's pinned buffer and uploaded in one piece per frame, then ssd_forward_mixed on the same stream.
extern "C" int ssd_forward_mixed_host(ssd_handle *h, const uint8_t *const *frames_host, int32_t B, const int32_t *hw_host, void *records, void *stream)
{
    if (!h || !frames_host || !hw_host || !records) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_mixed_host: null argument");
    if ((reinterpret_cast<uintptr_t>(records) & 3) != 0) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_mixed_host: records must be 4-byte aligned");
    std::lock_guard<std::mutex> g(h->mu);
    MixedSel ms;
    SSDCHK(prepare_mixed(h, B, hw_host, nullptr, &ms, "ssd_forward_mixed_host"));
    for (int b = 0; b < B; ++b)
        if (!frames_host[b]) return ssd_fail(SSD_ERR_INVALID, "ssd_forward_mixed_host: null frame pointer");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    const MixedCall &mc = h->mixed_store;
    const size_t bytes = (size_t)mc.geom.f[B - 1].off + (size_t)hw_host[2 * (B - 1)] * hw_host[2 * (B - 1) + 1] * 3;
    SSDCHK(stage_acquire(h));
    if (h->stage_stream_set && h->stage_stream != s) {
        if (hipStreamSynchronize(h->stage_stream) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipDeviceSynchronize()); }
    }
    SSDCHK(grow_stage(h, bytes));
    for (int b = 0; b < B; ++b) {           // frame b crosses the bus under the host copy of frame b + 1
        const size_t n = (size_t)hw_host[2 * b] * hw_host[2 * b + 1] * 3, lo = mc.geom.f[b].off;
        memcpy(h->stage_pin + lo, frames_host[b], n);
        HIPCHK(hipMemcpyAsync(h->stage_dev + lo, h->stage_pin + lo, n, hipMemcpyHostToDevice, s));
    }
    SSDCHK(stage_release(h, s, true));
    const long long T = (long long)h->cfg.num_classes * h->cfg.max_boxes_per_class;
    int32_t *r = (int32_t *)records;
    return forward_checked_locked(h, h->stage_dev, B, hw_host[0], hw_host[1], (float *)r, r + 5 * T, (float *)(r + 4 * T), r + 6 * T, 6 * T + 1, stream, &ms);
}

// inference/detector.py:33-58 as ONE call for one frame: ssd_forward_host, the wait for `stream`, and the score filter
// (`scores > score_threshold` over the frame's num_boxes rows, order kept) from the record into the caller's arrays.
extern "C" int ssd_detect_host(ssd_handle *h, const uint8_t *image_host, int32_t H, int32_t W, float score_threshold, void *record,
                               float *boxes_out, int32_t *labels_out, float *scores_out, int32_t capacity, int32_t *n_out, void *stream)
{
    if (!h || !record || !boxes_out || !labels_out || !scores_out || !n_out || capacity < 0) return ssd_fail(SSD_ERR_INVALID, "ssd_detect_host: bad arguments");
    {   // `record` is written by the GPU and read HERE, on the host: it must be pinned (or managed) host memory.  Checked once per
        // pointer (hipPointerGetAttributes costs microseconds; a serving loop reuses its blocks -- the last 8 verified ones are
        // remembered), so a device pointer comes back as an error code instead of a fault inside the library.  A block that is
        // freed and later re-allocated as another kind of memory at the same address must be announced: ssd_plan_cache_clear
        // forgets the verified pointers too.
        std::lock_guard<std::mutex> g(h->mu);
        bool known = false;
        for (int i = 0; i < h->n_rec_ok; ++i) known = known || h->detect_rec_ok[i] == record;
        if (!known) {
            HIPCHK(hipSetDevice(h->cfg.device));
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            const hipError_t pe = hipPointerGetAttributes(&at, record);
            if (pe != hipSuccess) (void)hipGetLastError();
            const bool host_visible = pe == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged || at.isManaged);
            if (!host_visible)
                return ssd_fail(SSD_ERR_INVALID, "ssd_detect_host: `record` must be pinned (hipHostMalloc / hipHostRegister) or managed host memory -- "
                                                 "a device pointer or pageable memory cannot be written by the GPU and read by this call");
            h->detect_rec_ok[h->n_rec_ok < 8 ? h->n_rec_ok++ : (h->rec_ok_next++ & 7)] = record;
        }
    }
    int rc = forward_host_impl(h, image_host, 1, H, W, record, stream, false);       // (this call drains `stream` itself, next line)
    if (rc != SSD_OK) return rc;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    const long long T = (long long)h->cfg.num_classes * h->cfg.max_boxes_per_class;
    const int32_t *r = (const int32_t *)record;
    const float *bx = (const float *)r, *sc = (const float *)(r + 4 * T);
    const int32_t *lb = r + 5 * T;
    int32_t n = r[6 * T];
    if (n < 0 || n > T) return ssd_fail(SSD_ERR_STATE, "ssd_detect_host: the record's num_boxes is out of range (is `record` host-visible memory?)");
    int32_t k = 0;
    for (int32_t i = 0; i < n; ++i)
        if (sc[i] > score_threshold) {
            if (k >= capacity) return ssd_fail(SSD_ERR_INVALID, "ssd_detect_host: capacity too small");
            memcpy(boxes_out + 4 * (size_t)k, bx + 4 * (size_t)i, 16);
            labels_out[k] = lb[i];
            scores_out[k] = sc[i];
            ++k;
        }
    *n_out = k;
    return SSD_OK;
}

extern "C" int ssd_get_tensor(ssd_handle *h, const char *name, float *dst, int64_t cap, int32_t *dims)
{
    if (!h || !name || !dst || !dims) return ssd_fail(SSD_ERR_INVALID, "ssd_get_tensor: null argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->cur) return ssd_fail(SSD_ERR_STATE, "ssd_get_tensor before ssd_forward");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    long long done = 0;
    int Btot = 0;
    for (Plan *pl : h->cur->plans) {             // sub-batches are consecutive images
        auto it = pl->retained.find(name);
        if (it == pl->retained.end()) return ssd_fail(SSD_ERR_INVALID, std::string("ssd_get_tensor: unknown tensor ") + name);
        const Retained &r = it->second;
        const long long rows = (long long)r.B * r.H * r.W;
        if (cap < done + rows * r.C) return ssd_fail(SSD_ERR_INVALID, "ssd_get_tensor: destination too small");
        std::vector<float> tmp((size_t)rows * r.Cp);
        HIPCHK(hipMemcpy(tmp.data(), r.dev, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (long long q = 0; q < rows; ++q)
            for (int c = 0; c < r.C; ++c) {
                const int pc = !r.permuted ? c : (r.split > 0 ? twopart_phys(c, r.split, r.Cp / 2) : ssd_phys_of_logical(c));
                if (r.fmt) {          // split-fp16 row: per octet 8 halves h, 8 halves l
                    const _Float16 *row = (const _Float16 *)&tmp[q * r.Cp];
                    dst[done + q * r.C + c] = (float)row[(pc >> 3) * 16 + (pc & 7)] + (float)row[(pc >> 3) * 16 + 8 + (pc & 7)];
                } else {
                    dst[done + q * r.C + c] = tmp[q * r.Cp + pc];
                }
            }
        done += rows * r.C;
        Btot += r.B;
        dims[1] = r.H; dims[2] = r.W; dims[3] = r.C;
    }
    dims[0] = Btot;
    return SSD_OK;
}

// Device-to-device variant of ssd_get_tensor: dst_dev receives the tensor in logical
// channel order; enqueued on `stream`, no synchronisation.
extern "C" int ssd_get_tensor_dev(ssd_handle *h, const char *name, float *dst_dev, int64_t cap, int32_t *dims, void *stream)
{
    if (!h || !name || !dst_dev || !dims) return ssd_fail(SSD_ERR_INVALID, "ssd_get_tensor_dev: null argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (!h->cur) return ssd_fail(SSD_ERR_STATE, "ssd_get_tensor_dev before ssd_forward");
    HIPCHK(hipSetDevice(h->cfg.device));
    long long done = 0;
    int Btot = 0;
    for (Plan *pl : h->cur->plans) {
        auto it = pl->retained.find(name);
        if (it == pl->retained.end()) return ssd_fail(SSD_ERR_INVALID, std::string("ssd_get_tensor_dev: unknown tensor ") + name);
        const Retained &r = it->second;
        const long long rows = (long long)r.B * r.H * r.W;
        if (cap < done + rows * r.C) return ssd_fail(SSD_ERR_INVALID, "ssd_get_tensor_dev: destination too small");
        if (r.permuted) HIPCHK(launch_permute_channels(r.dev, rows, r.C, r.Cp, r.fmt ? 2 : 0, dst_dev + done, (hipStream_t)stream, r.fmt ? 0 : r.split));
        else HIPCHK(hipMemcpyAsync(dst_dev + done, r.dev, (size_t)rows * r.C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        done += rows * r.C;
        Btot += r.B;
        dims[1] = r.H; dims[2] = r.W; dims[3] = r.C;
    }
    dims[0] = Btot;
    return SSD_OK;
}

// ----------------------------------------------------------------------------- plan cache
// out[0..7] = cached plan sets, their arena bytes, the budget in bytes, hits, misses, evictions, and the network shape
// (height, width) of the set the last forward ran.
extern "C" int ssd_plan_cache_stats(ssd_handle *h, int64_t *out)
{
    if (!h || !out) return ssd_fail(SSD_ERR_INVALID, "ssd_plan_cache_stats: null argument");
    std::lock_guard<std::mutex> g(h->mu);
    HIPCHK(hipSetDevice(h->cfg.device));
    size_t total = 0;
    for (const PlanSet *ps : h->cache) total += ps->bytes;
    out[0] = (int64_t)h->cache.size();
    out[1] = (int64_t)total;
    out[2] = (int64_t)plan_cache_limit_bytes(h);
    out[3] = h->cache_hits; out[4] = h->cache_misses; out[5] = h->cache_evictions;
    out[6] = h->cur ? h->cur->key.netH : 0;
    out[7] = h->cur ? h->cur->key.netW : 0;
    return SSD_OK;
}

// Drops every cached plan (and the verified record pointers of ssd_detect_host) after draining the device; the counters stay.
extern "C" int ssd_plan_cache_clear(ssd_handle *h)
{
    if (!h) return ssd_fail(SSD_ERR_INVALID, "ssd_plan_cache_clear: null handle");
    std::lock_guard<std::mutex> g(h->mu);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    h->cache_evictions += (long long)h->cache.size();
    free_plans(h);
    h->n_rec_ok = 0;
    return SSD_OK;
}

// ----------------------------------------------------------------------------- profiling
extern "C" int ssd_profile_enable(ssd_handle *h, int32_t on)
{
    if (!h) return ssd_fail(SSD_ERR_INVALID, "null handle");
    std::lock_guard<std::mutex> g(h->mu);
    h->profiling = on != 0;
    return SSD_OK;
}

// Per class, the time is the UNION of its kernels' [start, end] intervals inside each forward
// (the two head towers run concurrently on two streams: their kernels overlap, and a sum of
// durations would count the shared GPU twice).
static int drain_events(ssd_handle *h)
{
    std::vector<std::vector<std::pair<float, float>>> iv(SSD_NCLS);
    int cur_fwd = -1;
    auto flush = [&]() {
        for (int c = 0; c < SSD_NCLS; ++c) {
            auto &v = iv[c];
            std::sort(v.begin(), v.end());
            float lo = 0, hi = -1;
            for (auto &p : v) {
                if (hi < 0) { lo = p.first; hi = p.second; }
                else if (p.first <= hi) { if (p.second > hi) hi = p.second; }
                else { h->acc_ms[c] += hi - lo; lo = p.first; hi = p.second; }
            }
            if (hi >= 0) h->acc_ms[c] += hi - lo;
            v.clear();
        }
    };
    for (auto &e : h->evs) {
        HIPCHK(hipEventSynchronize(e.b));
        if (e.fwd != cur_fwd) { flush(); cur_fwd = e.fwd; }
        float t0 = 0, t1 = 0;
        if (e.fwd >= 0 && e.fwd < (int)h->ref_evs.size()) {
            HIPCHK(hipEventElapsedTime(&t0, h->ref_evs[e.fwd], e.a));
            HIPCHK(hipEventElapsedTime(&t1, h->ref_evs[e.fwd], e.b));
        } else {
            HIPCHK(hipEventElapsedTime(&t1, e.a, e.b));
        }
        iv[e.cls].push_back({t0, t1});
        h->ev_pool.push_back(e.a);
        h->ev_pool.push_back(e.b);
    }
    flush();
    h->evs.clear();
    for (auto r : h->ref_evs) h->ev_pool.push_back(r);
    h->ref_evs.clear();
    return SSD_OK;
}

extern "C" int ssd_profile_read(ssd_handle *h, int32_t cls, double *total_ms, int64_t *launches, double *flops, double *bytes)
{
    if (!h || cls < 0 || cls >= SSD_NCLS) return ssd_fail(SSD_ERR_INVALID, "ssd_profile_read: bad arguments");
    std::lock_guard<std::mutex> g(h->mu);
    SSDCHK(drain_events(h));
    if (total_ms) *total_ms = h->acc_ms[cls];
    if (launches) *launches = h->acc_n[cls];
    if (flops) *flops = h->acc_flops[cls];
    if (bytes) *bytes = h->acc_bytes[cls];
    return SSD_OK;
}

extern "C" int ssd_profile_reset(ssd_handle *h)
{
    if (!h) return ssd_fail(SSD_ERR_INVALID, "null handle");
    std::lock_guard<std::mutex> g(h->mu);
    SSDCHK(drain_events(h));
    for (int i = 0; i < SSD_NCLS; ++i) { h->acc_ms[i] = h->acc_flops[i] = h->acc_bytes[i] = 0; h->acc_n[i] = 0; }
    return SSD_OK;
}
