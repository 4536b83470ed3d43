/*
 * ssd_hip.h -- C ABI of libssd_hip.so: the MI355X (gfx950) RetinaNet inference path
 * that replaces the TensorFlow session behind TropComplique/single-shot-detector's
 * `Detector` (reference paths below are relative to the reference repository).
 *
 * The reference has no FFI/plugin layer of its own (it is Python on TF 1.12); the
 * boundary it does have is the frozen graph's tensor contract
 *     images:0 uint8 [1,H,W,3]  ->  boxes:0 [1,2000,4] f32, labels:0 [1,2000] i32,
 *                                   scores:0 [1,2000] f32, num_boxes:0 [1] i32
 * (create_pb.py:40,72; inference/detector.py:21-27,51-52; model.py:70-73).  ssd_forward()
 * is that `sess.run`, generalised over the batch dimension.  The other entry points are
 * the pieces the graph is assembled from, exported so that each can be parity-tested
 * against the CPU oracle exactly where the reference defines it.
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is device (HIP) memory owned by the
 *     caller, every pointer named *_host is host memory; the library owns its weights
 *     and workspace arena.
 *   - layout NHWC fp32, conv kernels HWIO, exactly the reference's TF variable layout.
 *   - return value: 0 (SSD_OK) or a negative code; ssd_last_error() gives the text for
 *     the calling thread.  Nothing throws across the ABI.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  ssd_postprocess only
 *     enqueues work.  ssd_forward only enqueues work for a shape it has a layer plan for.  Plans
 *     are keyed on what the NETWORK sees -- batch, the resized + padded height and width -- and a
 *     handle KEEPS the plans of every shape it has served (the reference's graph takes any image
 *     size in one session, detector/ssd.py:27-31, create_pb.py:24,40; its accuracy harness feeds
 *     val2017's mix of sizes through one Detector, inference/evaluate_on_COCO.ipynb:125-150): the
 *     FIRST call that lands on a new network shape -- and the first after ssd_set_precision or a
 *     handle option change -- builds that shape's plan beside the others (allocates its arena,
 *     uploads the anchor table; no device-wide wait), later calls of any source size that resizes
 *     to it call no HIP API but kernel launches.  The plans' arenas are bounded by option
 *     "plan_cache_mb"; passing it evicts the least recently used plans, which drains the device
 *     (only then).  ssd_plan_cache_stats reports the cache.
 *     ssd_status, ssd_get_tensor and ssd_set_precision synchronise (documented at each).  The
 *     stage entry points that take host weights (ssd_conv2d, ssd_depthwise3x3, ssd_dw_pw,
 *     ssd_first_conv, ssd_concat_shuffle_split, ssd_shuffle_conv1x1) are test conveniences and synchronise before
 *     returning.
 *   - one handle per device.  Every entry point that takes a handle holds the handle's mutex, so concurrent calls on
 *     one handle are serialised by the library, and a forward enqueued on another stream than the previous one first
 *     waits for that one's last kernel (a shape's arena is one, and the plans share the library's internal streams): two host
 *     threads may share a handle the way they
 *     may share a tf.Session (inference/detector.py:34,52).  The OUTPUT buffers belong to the caller, who must not let
 *     two in-flight forwards write the same ones.
 */
#ifndef SSD_HIP_H
#define SSD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSD_OK 0
#define SSD_ERR_INVALID (-1)   /* bad argument / unsupported shape           */
#define SSD_ERR_HIP (-2)       /* a HIP runtime call failed                  */
#define SSD_ERR_STATE (-3)     /* call out of order (e.g. forward before finalize) */
#define SSD_ERR_WEIGHT (-4)    /* missing / mis-shaped / unknown variable    */

#define SSD_ACT_NONE 0
#define SSD_ACT_RELU 1         /* tf.nn.relu  */
#define SSD_ACT_RELU6 2        /* tf.nn.relu6 */

#define SSD_BACKBONE_MOBILENET 0   /* detector/backbones/mobilenet_v1.py  */
#define SSD_BACKBONE_SHUFFLENET 1  /* detector/backbones/shufflenet_v2.py */

/* Arithmetic of the dense convolutions (ssd_set_precision): FPN laterals and 3x3 outputs, head
 * towers, class / box heads, and the MobileNet pointwise layers that are not fused with their
 * depthwise convolution (Conv2d_5..13).  Depthwise layers, the first convolution, the fused
 * depthwise+pointwise blocks and the ShuffleNet backbone are exact fp32 in both modes.  The
 * reference graph is fp32 end to end (tf.float32 everywhere, model.py:13-77): F32 is the default
 * and the mode every parity statement and the benchmark's headline refer to; F16X3 is opt-in.
 *   F32    every product on the exact-fp32 matrix instruction, one k-ordered fmaf chain per
 *          output: bit-identical to the CPU oracle.
 *   F16X3  each fp32 operand is carried as two halves x = h + l (22 significand bits) and a
 *          product is evaluated as xh*wh + xh*wl + xl*wh on the fp16 matrix instruction with
 *          fp32 accumulation: the error of one convolution equals that of an fp32 accumulation
 *          chain (measured), but the summation ORDER differs from the oracle's, so
 *          results agree within the north-star tolerance (1e-4), not bit for bit. */
#define SSD_PRECISION_F32 0
#define SSD_PRECISION_F16X3 1

typedef struct ssd_handle ssd_handle;

/* The inference keys of config_mobilenet.json / config_shufflenet.json (:7-12,21),
 * consumed by model.py:22-30,46,57-61. */
typedef struct ssd_config {
    int32_t backbone;            /* SSD_BACKBONE_*                       */
    float depth_multiplier;      /* "depth_multiplier"                   */
    int32_t num_classes;         /* "num_classes"                        */
    float score_threshold;       /* "score_threshold"                    */
    float iou_threshold;         /* "iou_threshold"                      */
    int32_t max_boxes_per_class; /* "max_boxes_per_class"                */
    int32_t min_dimension;       /* "min_dimension" (create_pb.py:24)    */
    int32_t device;              /* HIP device index (visible_device_list, inference/detector.py:6) */
} ssd_config;

/* ---- lifetime: replaces Detector.__init__ (inference/detector.py:6-34) ------------- */
int ssd_create(const ssd_config *cfg, ssd_handle **out);
void ssd_destroy(ssd_handle *h);
const char *ssd_last_error(void);

/* Selects the arithmetic of the dense convolutions (list above) for the following ssd_forward
 * calls (synchronises and drops the cached layer plan when the mode changes).  A new handle
 * takes its mode from the environment variable SSD_PRECISION ("f32" | "f16x3"), default f32. */
int ssd_set_precision(ssd_handle *h, int32_t mode /* SSD_PRECISION_* */);
int ssd_get_precision(ssd_handle *h);
/* Options.  None changes a result bit in mode F32 (each selects among kernels or schedules that are bit-identical by
 * construction and by test); the library reads NO environment variable for them (SSD_PRECISION above is the only one it
 * reads).  `h` == NULL sets the process-wide value, which the handle-less stage entry points below use and which a handle
 * falls back to for an option it has not been given itself; with a handle the call synchronises and drops the cached layer
 * plan.  Keys (value; default).
 * Selectors a caller may want:
 *   "streams"         0 auto | 1: every kernel of a forward on the caller's stream, in plan order            (0)
 *   "h2d_chunks"      2 | 1..16: pieces of ssd_forward_host's staging copy + upload                           (2)
 *   "front_fuse"      -1 auto | 0 | 1: the backbone's first layers as one launch (front.hip): MobileNet's first
 *                     convolution + Conv2d_1, ShuffleNet's first convolution + max pool                       (-1)
 *   "fuse_dw"         -1 default | bit mask of depthwise+pointwise pairs that run as one launch (MobileNet: bit i =
 *                     Conv2d_{i+1}; ShuffleNet: non-zero = every unit)                                        (-1)
 *   "backbone_split"  0 auto | 1 .. 4 (ShuffleNet: 1 | 2): backbone chains (two half-batch chains on two streams from 4 images on) (0)
 *   "event_fence"     0 | 1: the library's stream-ordering events with the default flags (system-scope fence per record) (0)
 *   "plan_cache_mb"   0 auto (a quarter of the device's memory) | n: MiB of arena the handle's cached layer plans may hold;
 *                     the least recently used go first, the plan of the running shape always stays.  Setting it on a handle
 *                     evicts down to the new budget and does NOT drop the other plans                          (0)
 * Test hooks -- they pin a kernel variant or a plan shape so that the parity tests see every shape on it:
 *   "igemm_tile"      0 auto | 128 | 64 pin the tile of the 128x128-class launches | 20..27, 30 pin a tile of the latency form
 *                     wherever that form applies: 20..23 one wave per block (1x1, 1x2, 2x1, 2x2 sixteen-wide units), 24..27
 *                     two / four waves per block sharing the positions through LDS, 30 the one-wave tile with 16 K-steps of
 *                     operands in flight (the auto choice for the smallest launches).  Any other value is refused  (0)
 *   "igemm_lat"       1 | 0: small exact-fp32 launches on the latency form (v_mfma_f32_16x16x4_f32) | 2: ... and 1x1 launches
 *                     up to 1 280 tiles | 3: as 1 without fpn p6 / p7 of the serving batches                  (1)
 *   "igemm_deep64"    -1 auto | 0 | 1: 64x64 tiles with operand loads three K-steps ahead                     (-1)
 *   "igemm16"         -1 auto | 0 | 1: F16X3 launches on the 256x256-tile kernel                              (-1)
 *   "igemm_96"        1 | 0: 128x96 tiles for widths 96 divides and 128 does not (read by ssd_finalize)       (1)
 *   "lateral_split"   1 | 0: F16X3 laterals split fp32 rows while staging them                                (1)
 *   "fpn_group"       -1 auto | 0 | 1: fpn p3 + p4 + p5 as one grouped launch (batch <= 2, F32)               (-1)
 *   "fpn_p7_group"    1 | 0: fpn p7 as a fourth level of that launch                                          (1)
 *   "fpn_early_lat"   -1 auto | 0 | 1: lateral3 / lateral4 early, their top-down sums as one elementwise launch (batch <= 2) (-1)
 *   "nsub"            0 auto | 1..8: at least this many consecutive sub-batch plans (the split a batch whose tensors would
 *                     pass 2 GiB takes)                                                                        (0)
 *   "nms_fast_max"    -1 default | n >= 0: candidate lists up to n stay in one wave's registers               (-1)
 *   "first_conv_px"   1 | 0: the first convolution of RESIZED frames (any size that is not the network's own) on the
 *                     lane-per-pixel kernel / on the thread-per-4-channels kernel of rounds 1-5                (1)
 *   "debug_sync"      0 | 1: announce every op on stderr, run it alone, wait for it, print its time           (0)
 * (Rounds 1-4 carried more switches -- schedule experiments that measured equal or slower: tower_group, head_serial,
 *  side_priority, level_split, fpn_p6_first, lat_one, dwpw_lat, graph, staggered sub-batch plans.  They are out of the library;
 *  scripts/experiments/README.md has what each measured and the commit that holds its source.)
 * ssd_get_option returns the value in effect (handle, else process), INT32_MIN when neither was set. */
int ssd_set_option(ssd_handle *h, const char *key, int32_t value);
int ssd_get_option(ssd_handle *h, const char *key, int32_t *value);
/* Synchronises the device and returns (and clears) the handle's status word.  Bit 0: in
 * F16X3 mode an activation left the fp16 range (|x| > 65504) and was clamped -- the results
 * of the forwards since the last call are not trustworthy; re-run them in F32 mode. */
int ssd_status(ssd_handle *h, int32_t *flags_out);

/* One call per TF variable of the frozen graph, by the reference's variable name
 * (e.g. "MobilenetV1/Conv2d_3_pointwise/weights", "fpn/p6/kernel",
 * "class_net/batch_norm_2_for_level_5/moving_variance"; SURVEY.md 8a "Weights").
 * Replaces tf.import_graph_def of the Const nodes (inference/detector.py:13-19). */
int ssd_load_weight(ssd_handle *h, const char *name, const float *host, const int64_t *shape,
                    int32_t ndim);
/* Checks that every variable of the configured architecture was loaded, computes the
 * batch-norm scale factors, re-lays the kernels out for the HIP kernels and uploads. */
int ssd_finalize(ssd_handle *h);

/* ---- the hot path: replaces sess.run(output_ops, {images: ...})
 *      (inference/detector.py:51-52) = create_pb.py:42-47 + model.py:13-77 ------------ */
/* images_dev uint8 [B,H,W,3], any H and W: the serving graph's preprocessing
 * (create_pb.py:42-47 -> resize_keeping_aspect_ratio(min_dimension, 128), pipeline.py:138-194:
 * nearest-neighbour resize so that the short side is min_dimension, zero pad bottom/right to
 * multiples of 128, box_scaler) is fused into the first kernel; for H, W multiples of 128 with
 * min(H,W) == min_dimension it is the identity.
 * Outputs, T = num_classes*max_boxes_per_class (2000): boxes_dev f32 [B,T,4]
 * (ymin,xmin,ymax,xmax, normalised, already divided by box_scaler, model.py:67-68),
 * labels_dev i32 [B,T], scores_dev f32 [B,T], num_boxes_dev i32 [B]; zero padded. */
int ssd_forward(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W,
                float *boxes_dev, int32_t *labels_dev, float *scores_dev,
                int32_t *num_boxes_dev, void *stream);

/* The same graph with the outputs as B fixed RECORDS -- the unit the all-gather of a data-parallel step moves (one process
 * per GPU, images sharded, detections exchanged once per batch; the reference maps over images, nms.py:96-101, and has no
 * multi-GPU code).  Record b starts at records_dev + b * ssd_record_words(h) 32-bit words:
 *     boxes [T,4] f32 | scores [T] f32 | labels [T] i32 | num_boxes i32      T = num_classes * max_boxes_per_class
 * (48 004 bytes at T = 2000).  records_dev: device memory, or pinned host memory (device-accessible at its own address).
 * Asynchronous on `stream` like ssd_forward. */
int32_t ssd_record_words(const ssd_handle *h);
int ssd_forward_records(ssd_handle *h, const uint8_t *images_dev, int32_t B, int32_t H, int32_t W,
                        void *records_dev, void *stream);
/* Detector.__call__'s own form (inference/detector.py:51-52 feeds a HOST array on every call): images_host is ordinary
 * (pageable) host memory; the library copies it through its pinned staging buffer into device memory in option
 * "h2d_chunks" pieces (piece k crosses the bus under the host copy of piece k + 1) and runs ssd_forward_records on
 * `stream`.  On return images_host may be reused; the upload and the forward are asynchronous on `stream`.
 * `records`: device memory or pinned host memory, as for ssd_forward_records. */
int ssd_forward_host(ssd_handle *h, const uint8_t *images_host, int32_t B, int32_t H, int32_t W,
                     void *records, void *stream);

/* Frames of DIFFERENT sizes as ONE batch.  The reference's graph is fed one image per call because a tensor has one height and
 * width (create_pb.py:40); what the network sees, though, is the size AFTER resize_keeping_aspect_ratio (pipeline.py:138-194), and
 * frames of many source sizes share it (480x640, 375x500, 333x500 -> 640x896).  ssd_forward_mixed runs B <= 64 such frames through the
 * ordinary batched plan of that network shape: the first kernel reads every frame through its own geometry (passed in its
 * arguments: nothing is uploaded), the pack kernel divides every image's boxes by its own box_scaler (model.py:67-68).  Record b
 * is bit for bit what frame b gives alone.
 *   images_dev   base pointer of the frames, uint8; frame b = [hw_host[2b], hw_host[2b+1], 3] at byte offset offsets_host[b]
 *                (offsets_host == NULL: back to back, every frame on the next 16-byte boundary); all within 2 GiB of the base
 *   records_dev  B records as for ssd_forward_records (device or pinned host memory)
 * Every frame must resize to the same network shape (ssd_network_shape tells a caller which: group by it), else SSD_ERR_INVALID.
 * Asynchronous on `stream`; hw_host / offsets_host may be reused on return. */
int ssd_network_shape(ssd_handle *h, int32_t height, int32_t width, int32_t *net_hw_out /* [2] */);
int ssd_forward_mixed(ssd_handle *h, const uint8_t *images_dev, int32_t B, const int32_t *hw_host, const int64_t *offsets_host,
                      void *records_dev, void *stream);
/* ... fed from host memory: frames_host[b] points to frame b (pageable memory is fine); staged through the handle's pinned buffer,
 * one upload per frame (frame b crosses the bus under the host copy of frame b + 1).  On return the frames may be reused. */
int ssd_forward_mixed_host(ssd_handle *h, const uint8_t *const *frames_host, int32_t B, const int32_t *hw_host, void *records,
                           void *stream);

/* inference/detector.py:33-58 (Detector.__call__) for ONE frame as one call: ssd_forward_host with B = 1, the wait for
 * `stream`, and the score filter `scores > score_threshold` over the frame's num_boxes rows (order kept) from the record
 * into the caller's host arrays boxes_out [capacity,4], labels_out / scores_out [capacity]; *n_out = rows kept.
 * `record` must be HOST-VISIBLE device-accessible memory of ssd_record_words(h) words (pinned: hipHostMalloc / a pinned
 * torch tensor): the last kernel writes the frame's record there and this call reads it.  A pointer that is not pinned or
 * managed host memory (device memory, pageable memory) is refused with SSD_ERR_INVALID -- checked once per pointer. */
int ssd_detect_host(ssd_handle *h, const uint8_t *image_host, int32_t H, int32_t W, float score_threshold,
                    void *record, float *boxes_out, int32_t *labels_out, float *scores_out, int32_t capacity,
                    int32_t *n_out, void *stream);

/* Copy a retained intermediate of the last ssd_forward to the host in the reference's
 * logical NHWC channel order (synchronises).  Names: "c3","c4","c5" (backbone outputs),
 * "p3".."p7" (feature_extractor.py:71-76), "encoded_boxes" [B,N,4] and
 * "class_predictions" [B,N,C] (box_predictor.py:102-104).  dims_out receives 4 ints. */
int ssd_get_tensor(ssd_handle *h, const char *name, float *host_dst, int64_t capacity_floats,
                   int32_t *dims_out);

/* Same, device to device: dst_dev receives the tensor in logical channel order; enqueued
 * on `stream` without synchronising (SSD.raw_predictions, ssd.py:37-40). */
int ssd_get_tensor_dev(ssd_handle *h, const char *name, float *dst_dev, int64_t capacity_floats,
                       int32_t *dims_out, void *stream);

/* The handle's cache of layer plans (conventions at the top of this file).  out[0..7] = cached plan sets, bytes of arena they
 * hold, the budget in bytes (option "plan_cache_mb"), hits, misses, evictions since ssd_create, and the network shape (height,
 * width) the last forward ran at.  A hit = a forward that found its shape's plan; a miss = one that built it. */
int ssd_plan_cache_stats(ssd_handle *h, int64_t *out8);
/* Drains the device and drops every cached plan (the next forward of each shape rebuilds it); also forgets the record
 * pointers ssd_detect_host has verified. */
int ssd_plan_cache_clear(ssd_handle *h);

/* Per-kernel-class timing with HIP events on the forward's stream (bench.py roofline).
 * classes: 0 conv3x3 MFMA, 1 pointwise MFMA, 2 depthwise, 3 first conv, 4 postprocess,
 * 5 other, 6 fused depthwise+pointwise, 7 conv3x3 on the 256x256-tile f16x3 kernel (igemm16.hip;
 * class 0 then holds the remaining 3x3 launches).  total_ms of a class is the union of its kernels' intervals (the two head towers
 * overlap on two streams).  ssd_profile_read synchronises the recorded events. */
int ssd_profile_enable(ssd_handle *h, int32_t on);
int ssd_profile_read(ssd_handle *h, int32_t cls, double *total_ms, int64_t *launches,
                     double *flops, double *bytes);
int ssd_profile_reset(ssd_handle *h);

/* ---- pieces of the graph, each where the reference defines it ---------------------- */

/* AnchorGenerator.__call__ (anchor_generator.py:40-120) with model.py:37-42 constants. */
int32_t ssd_num_anchors(int32_t H, int32_t W);
int ssd_anchors(int32_t H, int32_t W, float *anchors_host /* [N,4] */);
/* AnchorGenerator(strides, scales, scale_multipliers, aspect_ratios).__call__ (anchor_generator.py:13-120) for any
 * hyper-parameters: n_levels strides / scales, anchors per location = n_mult * n_ratios in itertools.product order.
 * Returns the number of anchors N (>= 0) or a negative SSD_ERR_*; writes [N,4] when anchors_host != NULL and
 * capacity_rows >= N (call once with NULL to size the buffer). */
int64_t ssd_anchors_ex(int32_t H, int32_t W, int32_t n_levels, const int32_t *strides, const double *scales,
                       int32_t n_mult, const double *multipliers, int32_t n_ratios, const double *ratios,
                       float *anchors_host, int64_t capacity_rows);

/* Dense k x k convolution (k = 1 or 3) on the MFMA implicit-GEMM kernel:
 * slim.conv2d / tf.layers.conv2d / conv2d_same (mobilenet_v1.py:49,66;
 * layer_utils.py:15-43; box_predictor.py:117-130,144-154; feature_extractor.py:57-69).
 * out = act( bn( conv(in) ) + bias + upsample2(up) ), each term optional (NULL).
 * pad_beg: 1 for 'same' stride 1 and for conv2d_same stride 2 (explicit pad),
 * 0 for TF 'SAME' stride 2 on even sizes and for k = 1.
 * bn_*: moving_mean, gamma*rsqrt(var+eps), beta (batch_norm_relu, layer_utils.py:5-12).
 * up_dev: coarser map [B,OH/2,OW/2,Cout] added after nearest x2 upsampling
 * (feature_extractor.py:67,79-100).  Cin must be a multiple of 8.  Batch norm, bias and
 * up_dev are mutually exclusive (the reference's graph has no layer combining them). */
int ssd_conv2d(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t Cin,
               const float *w_host /* [k,k,Cin,Cout] */, int32_t k, int32_t Cout,
               int32_t stride, int32_t pad_beg, int32_t OH, int32_t OW,
               const float *bn_mean_host, const float *bn_sf_host, const float *bn_beta_host,
               const float *bias_host, const float *up_dev, int32_t act, float *out_dev,
               void *stream);

/* The same convolution in F16X3 arithmetic (see SSD_PRECISION_F16X3): fp32 tensors in and
 * out, the split-fp16 rows exist only inside the call.  Test entry point for the kernel the
 * F16X3 forward runs. */
int ssd_conv2d_f16x3(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t Cin,
                     const float *w_host /* [k,k,Cin,Cout] */, int32_t k, int32_t Cout,
                     int32_t stride, int32_t pad_beg, int32_t OH, int32_t OW,
                     const float *bn_mean_host, const float *bn_sf_host, const float *bn_beta_host,
                     const float *bias_host, const float *up_dev, int32_t act, float *out_dev,
                     void *stream);

/* depthwise_conv (depthwise_conv.py:5-26): 3x3, weights [3,3,C,1], optional BN + act. */
int ssd_depthwise3x3(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t C,
                     const float *w_host, int32_t stride, int32_t pad_beg, int32_t OH,
                     int32_t OW, const float *bn_mean_host, const float *bn_sf_host,
                     const float *bn_beta_host, int32_t act, float *out_dev, void *stream);

/* depthwise_conv + BN + act followed by the 1x1 conv + BN + act that consumes it, as ONE kernel
 * (the MobileNet block, mobilenet_v1.py:59-67; shufflenet_v2.py:118-137 depthwise -> conv1x1_after):
 * the depthwise result stays in LDS (input patches staged by LDS-DMA, channels streamed in slices of
 * 32: any C, any H x W).  'SAME' padding; stride 2 needs even H, W; every tensor below 2 GiB
 * (SSD_ERR_INVALID otherwise). */
int ssd_dw_pw(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t C,
              const float *dw_w_host /* [3,3,C,1] */, int32_t stride, const float *dw_mean_host,
              const float *dw_sf_host, const float *dw_beta_host, int32_t dw_act,
              const float *pw_w_host /* [1,1,C,Cout] */, int32_t Cout, const float *pw_mean_host,
              const float *pw_sf_host, const float *pw_beta_host, int32_t pw_act,
              float *out_dev, void *stream);

/* uint8 image -> /255 -> 2x-1 -> 3x3 stride-2 'SAME' conv -> BN -> act, fused
 * (create_pb.py:42-47; mobilenet_v1.py:34,49; shufflenet_v2.py:37,50). */
int ssd_first_conv(const uint8_t *images_dev, int32_t B, int32_t H, int32_t W,
                   const float *w_host /* [3,3,3,Cout] */, int32_t Cout,
                   const float *bn_mean_host, const float *bn_sf_host,
                   const float *bn_beta_host, int32_t act, float *out_dev, void *stream);

/* MobileNet's first three layers as ONE launch -- what the layer plan runs for frames that arrive at the network's input
 * size (option "front_fuse"): ssd_first_conv (create_pb.py:42-47; mobilenet_v1.py:34,49: 3 -> 32) followed by ssd_dw_pw at
 * stride 1 (mobilenet_v1.py:59-67: depthwise 3x3, pointwise 32 -> 64), the 32-channel tensor kept in LDS.  Only these
 * widths (C0 == 32, Cout == 64; SSD_ERR_INVALID otherwise); bit-identical to the two calls it replaces.
 * images_dev [B,H,W,3] uint8, H and W even; out_dev [B,H/2,W/2,Cout]. */
int ssd_front_block(const uint8_t *images_dev, int32_t B, int32_t H, int32_t W,
                    const float *w0_host /* [3,3,3,C0] */, int32_t C0, const float *bn0_mean_host,
                    const float *bn0_sf_host, const float *bn0_beta_host, int32_t act0,
                    const float *dw_w_host /* [3,3,C0,1] */, const float *dw_mean_host,
                    const float *dw_sf_host, const float *dw_beta_host, int32_t dw_act,
                    const float *pw_w_host /* [1,1,C0,Cout] */, int32_t Cout, const float *pw_mean_host,
                    const float *pw_sf_host, const float *pw_beta_host, int32_t pw_act,
                    float *out_dev, void *stream);

/* ShuffleNet's first two layers as ONE launch -- what the layer plan runs for frames that arrive at the network's input size
 * (option "front_fuse"): ssd_first_conv (shufflenet_v2.py:37,50: 3 -> 24) followed by ssd_maxpool3x3s2 (:51-54), the
 * half-resolution tensor kept in LDS.  Only Cout == 24, H and W multiples of 4 (SSD_ERR_INVALID otherwise); bit-identical
 * to the two calls it replaces.  out_dev [B,H/4,W/4,Cout]. */
int ssd_first_conv_maxpool(const uint8_t *images_dev, int32_t B, int32_t H, int32_t W,
                           const float *w_host /* [3,3,3,Cout] */, int32_t Cout,
                           const float *bn_mean_host, const float *bn_sf_host,
                           const float *bn_beta_host, int32_t act, float *out_dev, void *stream);

/* slim.max_pool2d 3x3 stride 2 'SAME' (shufflenet_v2.py:51-54). */
int ssd_maxpool3x3s2(const float *in_dev, int32_t B, int32_t H, int32_t W, int32_t C,
                     float *out_dev, void *stream);

/* concat_shuffle_split (shufflenet_v2.py:94-115) on [rows, D] tensors. */
int ssd_concat_shuffle_split(const float *x_dev, const float *y_dev, int64_t rows, int32_t D,
                             float *xo_dev, float *yo_dev, void *stream);

/* concat_shuffle_split (shufflenet_v2.py:94-115) followed by the unit's conv1x1_before + batch norm + activation (:119) on the
 * new x half, as the ONE kernel the layer plan runs for it (sn_pw.hip): the shuffle is the kernel's per-channel source table.
 * x_dev, y_dev [rows, D] (D even), w_host [1,1,D,Cout] -> out_dev [rows, Cout]; bit-identical to ssd_concat_shuffle_split
 * followed by ssd_conv2d on its first output. */
int ssd_shuffle_conv1x1(const float *x_dev, const float *y_dev, int64_t rows, int32_t D,
                        const float *w_host, int32_t Cout, const float *bn_mean_host, const float *bn_sf_host,
                        const float *bn_beta_host, int32_t act, float *out_dev, void *stream);

/* SSD.get_predictions (ssd.py:42-69) = sigmoid + batch_multiclass_non_max_suppression
 * (nms.py:48-102, decode box_utils.py:114-142, tf.image.non_max_suppression of TF r1.12)
 * + boxes /= box_scaler (model.py:67-68).
 * logits_dev [B,N,C], codes_dev [B,N,4], anchors_dev [N,4]; outputs as ssd_forward.
 * workspace_dev: ssd_postprocess_workspace_bytes(B,N,C,max_boxes_per_class) bytes. */
size_t ssd_postprocess_workspace_bytes(int32_t B, int32_t N, int32_t C,
                                       int32_t max_boxes_per_class);
int ssd_postprocess(const float *logits_dev, const float *codes_dev, const float *anchors_dev,
                    int32_t B, int32_t N, int32_t C, float score_threshold, float iou_threshold,
                    int32_t max_boxes_per_class, const float *box_scaler_host /* [4] or NULL */,
                    float *boxes_dev, int32_t *labels_dev, float *scores_dev,
                    int32_t *num_boxes_dev, void *workspace_dev, size_t workspace_bytes,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSD_HIP_H */
