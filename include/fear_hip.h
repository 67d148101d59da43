/*
 * fear_hip.h — C ABI of the MI355X (gfx950) FEAR per-frame inference engine.
 *
 * This is the drop-in boundary for the reference's hot path: the two methods of
 * `FEARNet` that `FEARTracker.initialize()/update()` call
 *     model_training/model/fear_net.py:63-66   get_features(crop)
 *     model_training/model/fear_net.py:90-96   track(search, template_features)
 * plus the arg-max box decode of `FEARBoxCoder.decode`
 *     model_training/dataset/box_coder.py:75-107
 * Everything else of the reference (Hydra config, Lightning training, datasets, CoreML export,
 * iOS apps) is out of scope.
 *
 * Conventions
 *   - plain C, no torch/HIP types in signatures; `stream` is a `hipStream_t` passed as void*
 *     (NULL = the null stream).  Calls are asynchronous on that stream.
 *   - every tensor pointer is a DEVICE pointer owned by the caller, contiguous, in the
 *     reference's layouts: fp32 NCHW in and out.  Internal NHWC buffers are private.
 *   - no exceptions cross the ABI: 0 on success, negative FEAR_ERR_* otherwise;
 *     `fear_strerror` maps a status to text.
 *   - a handle is bound to one device, owns the packed weights and ONE workspace, and is not
 *     thread-safe; use one handle per device (one process per GPU for multi-GPU).  Calls on a
 *     handle are ordered: a call given another stream than the previous call first waits (on the
 *     device) for that call's work, because both use the same workspace.
 */
#ifndef FEAR_HIP_H
#define FEAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEAR_OK 0
#define FEAR_ERR_NULL (-1)      /* null handle / pointer                                   */
#define FEAR_ERR_SHAPE (-2)     /* unsupported n / hw / option value                       */
#define FEAR_ERR_FORMAT (-3)    /* malformed or unsupported .fearw model blob              */
#define FEAR_ERR_HIP (-4)       /* a HIP runtime call failed (see fear_last_hip_error)     */
#define FEAR_ERR_ALLOC (-5)     /* device/host allocation failed                           */
#define FEAR_ERR_NOHEAD (-6)    /* model blob has no correlation head (trunk-only file)    */

typedef struct fear_handle fear_handle;

/* Parse a `.fearw` model blob (host memory, layout include/fearw_format.h), upload the weights
 * (fp16 -> fp32, re-laid-out for the kernels) to `device` and build the launch plans.
 * Replaces: FEARNet.__init__ + load_from_lighting (model/fear_net.py:15-56, utils/torch.py:11-24). */
int fear_create(const void* fearw_blob, size_t nbytes, int device, fear_handle** out);

int fear_destroy(fear_handle* h);

/* FEARNet.get_features (fear_net.py:63-66): trunk + 1x1 neck.
 *   img   : (n, 3, hw, hw) fp32 NCHW, already normalised; hw a multiple of 32 (128 template / 256 search)
 *   out   : (n, 256, hw/16, hw/16) fp32 NCHW                                                  */
int fear_features(fear_handle* h, const float* img, int n, int hw, float* out, void* stream);

/* FEARNet.track (fear_net.py:90-96) == get_features(search) + BoxTower (model/blocks.py:174-194).
 *   search   : (n, 3, 256, 256) fp32 NCHW, normalised
 *   tmpl     : (n, 256, 8, 8)   template features of each crop (as returned by fear_features)
 *   tmpl_cls : optional second template used by the classification branch only
 *              (`update` argument of BoxTower.forward, blocks.py:174-179); NULL = tmpl
 *   bbox     : (n, 4, 16, 16) ltrb distances in search-crop pixels (exp already applied)
 *   cls      : (n, 1, 16, 16) classification logits (0.1 factor already applied)              */
int fear_track(fear_handle* h, const float* search, const float* tmpl, const float* tmpl_cls, int n,
               float* bbox, float* cls, void* stream);

/* fear_track writing both maps into ONE packed tensor, the payload of the multi-GPU all-gather:
 *   maps : (n, 5, 16, 16) fp32 — channels 0..3 = bbox (ltrb), channel 4 = cls.  Same arithmetic, same kernels; only the
 *   per-crop output stride differs (replaces a torch.cat of FEARNet.track's two outputs, fear_net.py:90-96).        */
int fear_track_packed(fear_handle* h, const float* search, const float* tmpl, const float* tmpl_cls, int n,
                      float* maps, void* stream);

/* FEARBoxCoder.decode (box_coder.py:75-107) with use_sigmoid=True on device, one wavefront per crop:
 *   rc    : (n, 2) int32   arg-max cell (row, col), first maximum
 *   xywh  : (n, 4) float64 box in search-crop pixels (float64 like the reference's grids)
 *   score : (n)    fp32    sigmoid(cls) at the arg-max cell
 * score_size / total_stride / instance_size are the tracker config values (16 / 16 / 256).      */
int fear_decode(fear_handle* h, const float* cls, const float* bbox, int n, int score_size, int total_stride,
                int instance_size, int32_t* rc, double* xywh, float* score, void* stream);

/* Tracker._postprocess with tracking_config["smooth"] = True (base_tracker.py:149-205: _confidence_postprocess,
 * box_coder.decode on the blended score, _postprocess_bbox, _smooth_size) on device, batched.
 *   prev_size : (n, 2) float64  previous box size in search-crop pixels (TrackingState.prev_size)
 *   window    : (score_size^2) float64  the tracker's window (np.hanning outer product for "cosine")
 *   penalty_k, window_influence, lr : the tracking_config values (siam_tracker.yaml)
 *   rc / xywh / score as fear_decode; xywh[2:4] is the smoothed size                             */
int fear_decode_smooth(fear_handle* h, const float* cls, const float* bbox, int n, int score_size, int total_stride,
                       int instance_size, const double* prev_size, const double* window, double penalty_k,
                       double window_influence, double lr, int32_t* rc, double* xywh, float* score, void* stream);

/* Tracker._preprocess_image (base_tracker.py:97-103) on device: uint8 HWC RGB crops ->
 * normalised fp32 NCHW, (px - 255*mean) * (1/(255*std)).
 *   u8  : (n, hw, hw, 3) uint8     out : (n, 3, hw, hw) fp32                                    */
int fear_normalize_u8(fear_handle* h, const uint8_t* u8, int n, int hw, float* out, void* stream);

/* get_extended_crop + _preprocess_image on device (utils/utils.py:215-253, base_tracker.py:97-103):
 * for each of n context boxes (x, y, w, h in frame pixels, as returned by extend_bbox; may leave the frame) cut the
 * box out of ONE uint8 RGB frame, fill what lies outside the frame with pad_rgb (the saturate-cast mean colour),
 * resize to out_hw x out_hw like cv2.INTER_LINEAR on uint8 and normalise -> (n, 3, out_hw, out_hw) fp32 NCHW.
 *   frame_u8 : (frame_h, frame_w, 3) uint8, device     ctx_xywh : (n, 4) int32, device     pad_rgb : (n, 3) uint8, device */
int fear_crop_normalize(fear_handle* h, const uint8_t* frame_u8, int frame_h, int frame_w, const int32_t* ctx_xywh,
                        const uint8_t* pad_rgb, int n, int out_hw, float* out, void* stream);

/* ---- engine options ------------------------------------------------------------------------- */
#define FEAR_OPT_MAX_BATCH 1   /* crops processed per internal pass (workspace is sized for it)  */
#define FEAR_OPT_PROFILE 2     /* 1: bracket kernel launches with hipEvents (fear_profile_*)     */
#define FEAR_OPT_PROFILE_OP 3  /* -1: every op of a plan; i >= 0: op i and all ops sharing its name */
#define FEAR_OPT_FUSE 4        /* 1 (default): fused block kernels; 0: one kernel per conv layer  */
#define FEAR_OPT_MATH 5        /* pointwise-conv arithmetic of the fused blocks:                   */
                               /*   0 (default) fp32 operands, v_mfma_f32_16x16x4_f32 (exact fp32) */
                               /*   1 fp32 activations split into fp16 hi+lo, exact-fp16 weights,  */
                               /*     v_mfma_f32_16x16x32_f16 on the matrix pipe, fp32 accumulate  */
                               /*     (fp32-grade for an fp16 payload — every shipped model; the    */
                               /*     weights of an fp32 payload, fearw_format.h, are ROUNDED to    */
                               /*     fp16 in this mode)                                            */
                               /*   2 activations and weights rounded to bf16, v_mfma_f32_16x16x32_bf16,*/
                               /*     fp32 accumulate: reduced precision (BASELINE config "bf16")  */
                               /*   Modes 1 / 2 also cover the plain GEMMs of the TRACK plan: the   */
                               /*   search branch's neck (AdjustLayer) and the two pixel-wise       */
                               /*   correlations (both operands activations: three MFMAs in mode 1).*/
                               /*   fear_features (template branch) keeps its neck in fp32 in every */
                               /*   mode, so template and search features of one net differ in      */
                               /*   precision in modes 1 / 2 (tolerances: tests/test_gpu_parity.py  */
                               /*   test_matrix_pipe_split_mode_matches_fp32 / ..._all_math_modes). */
#define FEAR_OPT_CHAIN 6       /* 1 (default): stride-16 trunk stage + neck as one register-resident chain kernel */
                               /*   (fp32 mode); 0: one fused kernel per block                                 */
#define FEAR_OPT_SMALL_PASS 7  /* passes of at most this many crops (default 96; 0 = never) run the small-batch plan:      */
                               /*   several workgroups per crop in the 16x16 kernels (split over channel chunks, partial   */
                               /*   sums reduced afterwards), the head's two branches on two streams; passes of <= 16 crops */
                               /*   run a third plan (one chunk per workgroup, the head's SepConvs as 16-channel output     */
                               /*   slices without partial sums, 16x8 tiles split over their expansion chunks)             */
#define FEAR_OPT_PLAN_CROPS 8  /* crop count whose launch plan fear_plan_size / fear_plan_op / fear_profile_read describe (a    */
                               /*   pass of <= FEAR_OPT_SMALL_PASS crops runs another plan than a full one); 0 (default) =     */
                               /*   FEAR_OPT_MAX_BATCH, i.e. the plan of a full pass                                           */
#define FEAR_OPT_DUAL_HEAD 9   /* 1: the throughput plan runs the head's cls and bbox branches on two streams (one workgroup   */
                               /*   of each fits on a CU); 0 (default): one stream — measured equal, the kernels are ALU-bound */
#define FEAR_OPT_HEAD_STAGGER 10 /* with two head streams (FEAR_OPT_DUAL_HEAD, or a small pass): microseconds (0..1000, default 0) the  */
                               /*   second branch's first kernel is held back, so that the co-resident kernels of the two branches  */
                               /*   run out of phase and one's prologue / output burst overlaps the other's MFMA stretch            */
#define FEAR_OPT_TILE_V4 11    /* 1 (default): the throughput plan runs the blocks listed in the engine's kFusedTileV4 (stage 6) on the   */
                               /*   phase-overlapped tile kernel (depthwise taps of chunk c interleaved with the expansion MFMAs of      */
                               /*   chunk c + 1); 0: every tiled block on ir_tile_v2 (A/B)                                              */
#define FEAR_OPT_TINY_SEP 12   /* 1 (default): in the plan of a handful of crops (<= 16) the head's 16-channel SepConv slices and the      */
                               /*   prediction convs run sep16_tiny_kernel (the map cut into row groups as well, one row per wave: 4x the */
                               /*   workgroups, a quarter of the instruction issue per workgroup); 0: sep16_kernel<CIN, 16, KS> (A/B)     */
#define FEAR_OPT_HEAD_CHAIN 13 /* 1 (default): the throughput plan runs the whole BoxTower (model/blocks.py:129-194) — both branches,       */
                               /*   their eight SepConvs, the two pixel-wise correlations and the two prediction heads — as ONE launch   */
                               /*   whose activations stay on the CU between layers: headchain_kernel in fp32 mode (maps bit-identical   */
                               /*   to the eight sep16 launches), headchain_b_kernel with FEAR_OPT_MATH = 2 (same bf16 rounding points   */
                               /*   as the `*_h` launches, another summation order); FEAR_OPT_MATH = 1 keeps the launches.  0: launches  */
#define FEAR_OPT_BF16_STORE 14 /* 1 (default): with FEAR_OPT_MATH = 2 the throughput plan keeps the activations of the trunk's HBM-bound      */
                               /*   front (stem output ... input of the 64 -> 32 block; maps of 64 x 64 and larger) in bf16 BETWEEN kernels — */
                               /*   half the traffic of the kernels that are bound by it; 0: fp32 storage (A/B).  No effect in modes 0 / 1.   */
#define FEAR_OPT_E1_PAIR 15    /* 1 (default): fp32 mode, throughput plan — two consecutive 24-channel e1 blocks (depthwise 3x3 + pointwise  */
                               /*   + input, no expansion; model/blocks.py:8-42) run as ONE launch with the map between them in LDS          */
                               /*   (e1pair_kernel: half the HBM traffic of these two memory-bound blocks); 0: one launch per block (A/B).    */
#define FEAR_OPT_SPLIT_STREAMS 16 /* 1: a throughput pass of fear_track / fear_track_packed is issued as two half-batches, the second on a  */
                               /*   stream of the handle's own, joined before the call returns to the caller's stream: one half's launch    */
                               /*   boundaries and memory-bound kernels overlap the other's ALU-bound ones (+3 % crops/s at 256 crops).     */
                               /*   Same plans, same kernels per crop: bit-identical maps.  0 (default): one stream — every per-kernel      */
                               /*   figure of bench.py (the roofline object first of all) is then a full-grid launch.  Only when both        */
                               /*   halves still exceed FEAR_OPT_SMALL_PASS; ignored while FEAR_OPT_PROFILE is on.                           */
#define FEAR_OPT_CHAIN32 17    /* fp32 mode, throughput plan — the 32 x 32 trunk stage (three inverted-residual blocks of 32 channels + the     */
                               /*   stride-2 block down to the 16 x 16 map; model/blocks.py:8-42) as a register-resident chain:                  */
                               /*   2 (default): in ONE launch with the stride-16 stage + neck when FEAR_OPT_CHAIN is on (chain32_16_kernel);    */
                               /*   1: a launch of its own (chain32_kernel); 0: one tile kernel per block (A/B).                                 */
                               /*   With the automatic plan selection on (FEAR_OPT_SMALL_PASS > 0) a pass of at most 128 crops keeps the tile    */
                               /*   kernels for this stage: one workgroup per crop on half the CUs is slower than tiles there.                    */
int fear_set_option(fear_handle* h, int option, int64_t value);
int64_t fear_get_option(fear_handle* h, int option);

/* ---- introspection / measurement ------------------------------------------------------------- */
/* Number of kernel launches ("ops") in the plan for input size hw (with_head: track vs features). */
int fear_plan_size(fear_handle* h, int hw, int with_head);
/* Describe op `i` of that plan: name (<=63 chars), algorithmic FLOPs and compulsory bytes per crop. */
int fear_plan_op(fear_handle* h, int hw, int with_head, int i, char* name64, double* flops_per_crop,
                 double* bytes_per_crop);
/* With FEAR_OPT_PROFILE=1: synchronise and return the accumulated device time (ms) and launch count of
 * op `i` since the last fear_profile_reset; events are recorded on the stream the op was launched on. */
int fear_profile_read(fear_handle* h, int hw, int with_head, int i, double* total_ms, int64_t* launches);
int fear_profile_reset(fear_handle* h);

size_t fear_workspace_bytes(fear_handle* h);
const char* fear_strerror(int status);
/* hipError_t (as int) of the last failing HIP call on this handle, 0 if none. */
int fear_last_hip_error(fear_handle* h);
const char* fear_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FEAR_HIP_H */
