/* einx.h -- C ABI of libeinx_hip.so: the MI355X (gfx950) native implementation of the
 * EI-Nexus event<->image feature extraction + matching hot path.
 *
 * The reference (ZhonghuaYi/EI-Nexus_official) is pure Python/PyTorch and owns no native
 * layer, so the "FFI" a maintainer would bind is the set of torch op groups its nn.Modules
 * dispatch.  Each entry point below names the reference code it replaces (file:line relative
 * to the reference root).  Conventions:
 *   - plain pointers + sizes; every pointer is DEVICE memory unless it says "host";
 *   - tensors are dense, fp32, NCHW / row-major, exactly the reference's logical layouts;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue;
 *   - the caller makes the device of `stream` and of the buffers the current HIP device (hipSetDevice) before a call;
 *   - return 0 on success, negative on error (einx_last_error() gives the text);
 *   - no device-memory allocation inside: callers pass workspaces sized by the *_ws_bytes helpers.  Two documented pieces of
 *     library-owned state: (a) a pool of EINX_FORK_STREAM_POOL streams per device, created at the first einx_extract that
 *     forks on that device and never destroyed, and two events per (device, caller stream) that forks (see einx_extract),
 *     shared by every handle of the process; at most EINX_FORK_STREAMS_MAX such pairs exist at a time (least recently used
 *     first out; einx_fork_stream_release drops one explicitly); (b) einx_voxel_grid /
 *     einx_events_mask keep a few hundred bytes of pinned staging per host thread for the host offsets array;
 *   - einx_build_flags() tells a shipped library from a timing-only experiment build (see below).
 * No torch types cross this boundary.  INTEGRATION.md shows the ctypes binding.
 */
#ifndef EINX_H
#define EINX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EINX_OK 0
#define EINX_ERR_ARG (-1)
#define EINX_ERR_LAUNCH (-2)
#define EINX_ERR_NO_DEVICE (-3)

/* ABI revision of this header.  The public structs change layout between revisions (6: struct_size members, the weight watch out
 * of einx_extract_out, einx_lg_layer::Wqk_v); a host compares einx_abi_version() with the EINX_ABI_VERSION it was compiled
 * against before anything else, and every struct that is read by the library starts with `struct_size` = sizeof(the struct),
 * which the library checks (EINX_ERR_ARG on a mismatch instead of reading garbage from a shorter / longer struct). */
#define EINX_ABI_VERSION 6
int einx_abi_version(void);
const char* einx_version(void);
const char* einx_last_error(void);
/* Compile-time switches of this binary as a space-separated string ("" for the shipped build).  Experiment builds
 * (tools/build_variant.sh) that drop work to time what is left produce WRONG results; every such switch only takes effect
 * under -DEINX_TIMING_ONLY_BUILD, which this string then reports as "timing-only" -- the Python package refuses to load such
 * a library unless EINX_ALLOW_TIMING_ONLY=1. */
const char* einx_build_flags(void);
/* number of visible HIP devices (0 without a GPU); never initialises a context beyond that */
int einx_device_count(void);

/* Content watch of a module's weights (the reference's nn.Modules see an in-place edit through `p.data` at the next forward,
 * torch/nn/modules/module.py semantics; here weights are repacked / folded at load time).  table: device [n][2] int64 =
 * (device pointer, number of 32-bit words) per row.  Every word of a row is hashed (position-dependent, order-independent sum:
 * any edit of any word changes the hash) by ONE 64-lane wave, so callers cut large tensors into rows of
 * EINX_WATCH_CHUNK_WORDS words (any row length works; short rows keep a row's loads in flight together).
 * ref == stale == NULL: store the hashes in hash[n]; otherwise hash[n] is scratch and *stale (device int32) is OR-ed with 1
 * when any row's hash differs from ref. */
#define EINX_WATCH_CHUNK_WORDS 4096
int einx_params_hash(const int64_t* table, int n, uint64_t* hash, const uint64_t* ref, int32_t* stale, void* stream);

/* The numeric contract on the device (test aid): y[i] = f(x[i]) for the functions of include/einx_math.h as compiled for gfx950
 * (fn: 0 exp, 1 log, 2 sin, 3 cos, 4 erf, 5 sigmoid, 6 logsigmoid, 7 gelu, 8 acos -- what torch.exp / log / sin / cos / erf /
 * sigmoid / F.logsigmoid / F.gelu / acos compute in the reference's modules).  The oracle compiles the same header with gcc;
 * tests hold the two bit-equal and the header within 2 ulp of float64 libm. */
int einx_math_eval(int fn, const float* x, long long n, float* y, void* stream);

/* Do two streams of the current device run side by side (no reference counterpart: the reference has one stream)?  HIP deals
 * streams onto a few hardware queues; two that share one serialise, and which share depends on everything the process created
 * before (a process group's streams, a loader's copy streams).  Holds one wave spinning for spin_us (1..5000) on each stream
 * between a common start and end and returns the elapsed time in *elapsed_us: about spin_us when they overlap, about twice that
 * when they do not (or when a == b).  Synchronises with both streams; not capturable.  The Python package uses it to choose the
 * event extractor's side stream, einx_fork_stream_prepare to choose the fork streams. */
int einx_stream_overlap_us(void* stream_a, void* stream_b, int spin_us, float* elapsed_us);

/* Measurement aid (no reference counterpart; the reference's scripts time with wall clocks around
 * whole forwards, test_events-image_same-time.py:196-208): while enabled, every kernel launch of
 * this library is bracketed by HIP events recorded on the launch stream.  einx_profile_report
 * waits for them and writes one text line per kernel class, "name calls total_ms\n", into `buf`
 * (host memory).  einx_profile_enable(0|1) also clears the collected records.  While enabled einx_extract keeps its two head
 * branches in line (no fork), so that every number is a kernel alone on the chip. */
int einx_profile_enable(int on);
int einx_profile_report(char* buf, size_t cap);

/* ------------------------------------------------------------------------------------------
 * Convolution blocks  (K1/K2 of SURVEY.md 2.3)
 * replaces: nn.Conv2d(3x3,pad 1 | 1x1) + ReLU + BatchNorm2d(eval) + MaxPool2d(2,2)
 *   core/modules/net/vgg.py:34-38, core/modules/net/backbone.py:105-128,
 *   core/modules/net/detector_head.py:42-48, core/modules/net/descriptor_head.py:40-43,
 *   core/modules/image_extractors/superpoint_extractor.py:388-406,
 *   core/modules/image_extractors/silk/backbones/superpoint/vgg.py:284-290
 * and Padder.pad (replicate) folded into the first layer's addressing
 *   core/modules/utils/util.py:17-32.
 * ---------------------------------------------------------------------------------------- */

/* floats needed for the kernel-native weight image of one conv (k-major, cout padded to 64) */
size_t einx_conv_weight_elems(int cin, int cout, int ks);
/* OIHW fp32 -> kernel-native [K][CoutPad], K ordered (ci>>1, tap, ci&1); runs on `stream` */
int einx_conv_repack(const float* w_oihw, int cin, int cout, int ks, float* w_native, void* stream);

/* BatchNorm2d(eval) running statistics -> (scale, shift): scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale (torch.nn.BatchNorm2d in eval mode; core/modules/net/vgg.py:37) */
int einx_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int n, float* scale,
                 float* shift, void* stream);

typedef struct einx_conv_desc {
  const float* w_native; /* from einx_conv_repack */
  const float* bias;     /* [cout] or NULL */
  const float* scale;    /* [cout] BN(eval) gamma/sqrt(var+eps), or NULL (no BN) */
  const float* shift;    /* [cout] beta - mean*scale, or NULL */
  int32_t cin, cout, ks; /* ks in {1,3}; padding ks/2, stride 1 */
  int32_t relu;          /* apply ReLU after bias (before BN, as the reference orders it) */
  int32_t pool;          /* fuse MaxPool2d(2,2) after BN (H, W must be even) */
} einx_conv_desc;

/* in:  [B,cin,Hs,Ws] source tensor.  The layer sees a logical [B,cin,H,W] input where
 *      logical (y,x) = source (clamp(y-h0,0,Hs-1), clamp(x-w0,0,Ws-1))  (replicate padding);
 *      pass Hs=H, Ws=W, h0=w0=0 for an ordinary layer.
 * out: [B,cout,H,W] or [B,cout,H/2,W/2] when pool. */
int einx_conv_block(const float* in, int B, int Hs, int Ws, int h0, int w0, int H, int W, const einx_conv_desc* d, float* out,
                    void* stream);

/* The first two layers of a network with a thin first layer as ONE launch (round 6): SuperPointv1's conv1a (1 -> 64, 3x3, ReLU) and conv1b
 * (64 -> 64, 3x3, ReLU, MaxPool 2x2), superpoint_extractor.py:388-390 -- the first layer's output (761 MB at B = 32) is
 * recomputed per tile on the matrix cores and never touches HBM.  Bit-identical to einx_conv_block(d0) followed by
 * einx_conv_block(d1).  in / Hs / Ws / h0 / w0 / H / W: as einx_conv_block for the FIRST layer; out: the second layer's output.
 * einx_conv_first_two_fused_ok: 1 when the pair of layers and the launch size are what the kernel covers AND the one launch is the
 * faster form (64-channel 3x3 layers on ONE input channel, a launch of at least eight rounds of workgroups) -- what einx_extract
 * uses; 2 when the kernel covers the pair but the two launches are faster (5 input channels: the event voxel grid of the bench shape;
 * einx_conv_first_two_fused still takes it); 0 otherwise: run the two layers one by one. */
int einx_conv_first_two_fused_ok(const einx_conv_desc* d0, const einx_conv_desc* d1, int B, int H, int W);
int einx_conv_first_two_fused(const float* in, int B, int Hs, int Ws, int h0, int w0, int H, int W, const einx_conv_desc* d0,
                              const einx_conv_desc* d1, float* out, void* stream);

/* name of the kernel instantiation the calling thread's last einx_conv_block launched, e.g.
 * "conv_block_kernel<3,8,32,2,4,1,2,8,true,true>" (tile shape, wave layout, chunk, pool, reload path): provenance for
 * profiles and bench lines, so that a reported kernel name cannot go stale against the dispatcher */
const char* einx_conv_last_kernel(void);

/* x /= divisor in place (SuperPointv1.forward `image /= 255.0`, superpoint_extractor.py:372) */
int einx_div_inplace(float* x, size_t n, float divisor, void* stream);

/* SuperPointv1.forward's input handling for RGB and / or non-contiguous images (superpoint_extractor.py:372-376):
 *   image /= divisor in place on the caller's [B,C,H,W] tensor THROUGH ITS ELEMENT STRIDES (the caller sees the scaled tensor
 *   afterwards, as with the reference), and the network's contiguous [B,1,H,W] input written to `gray`:
 *   C == 1: the scaled value; C == 3: kornia.color.rgb_to_grayscale (kornia 0.7.1) = (0.299 r + 0.587 g) + 0.114 b in fp32,
 *   every product and sum rounded separately.  Overlapping strides (expanded tensors) are the caller's problem, as in torch. */
int einx_image_prepare(float* image, int B, int C, int H, int W, long long stride_b, long long stride_c, long long stride_h,
                       long long stride_w, float divisor, float* gray, void* stream);

/* ------------------------------------------------------------------------------------------
 * Detector head post-processing  (K3, K4, K5)
 * ---------------------------------------------------------------------------------------- */

/* logits_to_prob + depth_to_space + score[~dilate3x3(mask)] = 0 + remove_border_points
 *   core/modules/utils/detector_util.py:18-77,138-164,
 *   core/modules/event_extractors/EventExtractors.py:544-550,561-562
 * logits [B,C,hc,wc], C in {65,1}.  prob [B,C,hc,wc] (C==1: written with the mask/border zeros,
 * because the reference's score aliases probability there).  score [B,Hp,Wp], Hp=hc*cell.
 * mask: uint8 [B,H,W] (bool) or NULL; (h0,w0) = top/left padding of the padded map; dilate:
 * apply the 3x3 dilation (event extractors) or use the mask as is (image extractors). */
int einx_score_map(const float* logits, int B, int C, int hc, int wc, const uint8_t* mask, int H, int W, int h0, int w0,
                   int dilate, int border, float* prob, float* score, void* stream);

/* remove_border_points alone, in place on [B,Hp,Wp] (detector_util.py:138-164) */
int einx_remove_border(float* score, int B, int Hp, int Wp, int border, void* stream);

/* get_dense_positions (core/modules/utils/detector_util.py:504-519) for an unpadded score map
 * [B,1,H,W]: out [B,H*W,3] = (y+0.5, x+0.5, score) ("yx") or (x+0.5, y+0.5, score) ("xy") */
int einx_dense_positions(const float* score, int B, int H, int W, int ordering_xy, float* out, void* stream);

typedef struct einx_detect_params {
  int32_t B, Hp, Wp;     /* padded map */
  int32_t H, W, h0, w0;  /* unpadded size and top/left padding (Padder) */
  int32_t radius;        /* nms_radius (0 = no NMS) */
  int32_t top_k;         /* detection_top_k (0 = none) */
  float det_thr;         /* detection_threshold */
  int32_t ordering_xy;   /* 0: (y,x,p)  1: (x,y,p) */
  int32_t cap;           /* rows available per image in positions/indices */
  int32_t nms_iters;     /* suppression passes enqueued (>= 1); see not_converged */
} einx_detect_params;

size_t einx_detect_ws_bytes(const einx_detect_params* p);

/* prob_map_to_points_map (fast_nms fix-point + top-k quantile threshold) +
 * prob_map_to_positions_with_prob + Padder.unpad_positions + filter_sparse_feats
 *   core/modules/utils/detector_util.py:80-135,243-337,451-484, core/modules/utils/util.py:52-66,
 *   core/modules/event_extractors/EventExtractors.py:496-515
 * score     [B,Hp,Wp]  input (already masked / border-zeroed), not modified
 * nms_out   [B,H,W]    thresholded NMS map cropped to the unpadded window, or NULL
 * positions [B,cap,3]  (y+.5-h0, x+.5-w0, p) in raster order
 * indices   [B,cap]    int32 flat index y*Wp+x in the padded map
 * counts    [B]        int32 number of keypoints (may exceed cap: only cap rows are written)
 * thr       [B]        threshold used
 * not_converged [B]    int32, nonzero if the fix-point needed more than nms_iters passes (caller re-runs with more
 *                      passes).  Radius 4 (every shipped configuration) never raises it: images that are still changing
 *                      after the nms_iters wide passes are completed on the device by a per-image finisher
 *                      (nms4_finish_kernel), e.g. quantised maps that need 14-25 passes */
int einx_detect(const float* score, const einx_detect_params* p, void* ws, float* nms_out, float* positions, int32_t* indices,
                int32_t* counts, float* thr, int32_t* not_converged, void* stream);

/* ------------------------------------------------------------------------------------------
 * Descriptor heads post-processing  (K6, K10)
 * ---------------------------------------------------------------------------------------- */

/* sparsify_low_resolution_descriptors (bilinear=1; descriptor_util.py:74-128) or
 * sparsify_full_resolution_descriptors (bilinear=0; descriptor_util.py:50-71), then
 * F.normalize * scale.  raw [B,D,hc,wc], or with channels_last=1 (bilinear only) the [B,hc*wc,D]
 * copy written by einx_normalize_map; indices/counts from einx_detect; out [B,cap,D]. */
int einx_desc_sample(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int bilinear, int channels_last,
                     const int32_t* indices, const int32_t* counts, int cap, float scale, float* out, void* stream);

/* normalize_descriptors over channels of a dense map (descriptor_util.py:21-28). raw/out [B,D,P].
 * raw_cl: optional [B,P,D] channels-last copy of `raw` (un-normalised) for einx_desc_sample, or NULL. */
int einx_normalize_map(const float* raw, int B, int D, int P, float scale, float* out, float* raw_cl, void* stream);

/* F.normalize(x, dim=1) * scale on a row-major [R,C] matrix: the random padding descriptors of the
 * trainable matcher branch (core/modules/Matchers.py:114-131) */
int einx_normalize_rows(const float* x, int R, int C, float scale, float* out, void* stream);

/* random padding keypoints of the same branch (Matchers.py:80-91): u [R,2] uniform draws ->
 * out [R,3] = (u0*size0, u1*size1, 0) */
int einx_random_positions(const float* u, int R, float size0, float size1, float* out, void* stream);

/* upsample_descriptors: bilinear resize to (Hp,Wp) + normalize, written cropped to the
 * unpadded window [B,D,H,W] (descriptor_util.py:131-138 + Padder.unpad util.py:34-50).
 * ws: device scratch of at least einx_upsample_ws_bytes(B,H,W) bytes (the per-pixel norms between the two
 * kernels); required when W <= 384 and wc <= 63 (every shipped geometry), unused (may be NULL) otherwise. */
size_t einx_upsample_ws_bytes(int B, int H, int W);
int einx_upsample_normalize(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, int h0, int w0, int H, int W,
                            float scale, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mutual-nearest-neighbour matcher  (K7)   core/modules/matchers/MNN.py:43-140
 * desc0 [B,cap0,D], desc1 [B,cap1,D]; n[b], m[b] device int32 counts (<= cap).
 * matches0 [B,cap0] int64 (-1 = none), matches1 [B,cap1]; scores fp32 0/1.
 * la: optional [B,cap0+1,cap1+1] log_assignment written at [b, :n+1, :m+1] with row pitch
 * (cap1+1), or NULL.  ws: einx_mnn_ws_bytes.
 * ---------------------------------------------------------------------------------------- */
size_t einx_mnn_ws_bytes(int B, int cap0, int cap1);
int einx_mnn(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
             void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la, void* stream);
/* einx_mnn followed by einx_gather_matches (below) with the mutual check and the matched-keypoint compaction as ONE launch
 * (MNN.py:98-129: the matches and the per-pair matched_kpts lists); same outputs as the two calls. */
int einx_mnn_gather(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
                    void* ws, int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la, const float* kpts0,
                    const float* kpts1, int cols, float* mk0, float* mk1, int32_t* nmatch, void* stream);
/* the same with find_nn's optional thresholds (MNN.py:12-22): dist = 2*(1-sim) of the best / second-best
 * neighbour per row (per column for matches1); a match is kept when dist0 <= ratio_sq*dist1 (use_ratio) and
 * dist0 <= dist_sq (use_dist); the mutual check (:25-32) runs on the masked matches.  ratio_sq / dist_sq are the
 * squared thresholds rounded to fp32 (what torch computes for `tensor <= python_float`).  Rows / columns with a
 * single candidate have no second neighbour: the ratio test passes there (torch.topk(2) raises; the Python
 * wrapper raises the same error when it knows the counts). */
int einx_mnn_thresh(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
                    int use_ratio, float ratio_sq, int use_dist, float dist_sq, void* ws, int64_t* matches0, int64_t* matches1,
                    float* scores0, float* scores1, float* la, void* stream);

/* similarity = einsum("bnd,bmd->bnm") (MNN.py:88), returned by the reference's un-frozen Matcher
 * branch (Matchers.py:204-222).  sim [B,cap0,cap1]; entries outside [n[b], m[b]] are zero. */
int einx_similarity(const float* desc0, const int32_t* n, int cap0, const float* desc1, const int32_t* m, int cap1, int B, int D,
                    float* sim, void* stream);

/* matched keypoint gather in ascending keypoint-0 order (MNN.py:119-129, lightglue.py:690-698)
 * kpts0 [B,cap0,3], kpts1 [B,cap1,3]; cols = 3 (MNN) or 2 (LightGlue);
 * out0/out1 [B,cap0,cols]; nmatch [B] int32. */
int einx_gather_matches(const float* kpts0, const float* kpts1, const int64_t* matches0, const int32_t* n, int cap0, int cap1,
                        int B, int cols, float* out0, float* out1, int32_t* nmatch, void* stream);

/* Packs the first counts[b] rows of two padded [B,cap,width] arrays (the matched keypoints of
 * einx_gather_matches) into two flat [sum(counts),width] arrays, pair after pair, so that the
 * per-pair lists the reference returns (Matchers.py:177-201) are views cut by one split call.
 * dst0/dst1 need room for B*cap rows. */
int einx_compact_rows(const float* src0, const float* src1, const int32_t* counts, int B, int cap, int width, float* dst0, float* dst1,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * LightGlue  (K8, K9)   core/modules/matchers/lightglue.py:522-716
 * ---------------------------------------------------------------------------------------- */
/* y[M,N] (+)= x[M,K] w[N,K]^T + bias[N]   (torch.nn.functional.linear; also used at load time to
 * fold LightGlue's out_proj / to_out into the following FFN Linear).  accumulate != 0: y += ... */
int einx_linear(const float* x, int M, int K, const float* w, const float* bias, int N, float* y, int accumulate, void* stream);

/* Wo / Wco may be NULL: the message projection has then been folded into sf0_w / cf0_w
 * (x' = ffn(cat[x, context]) with W0' = [W0a | W0b Wo], b0' = b0 + W0b bo). */
typedef struct einx_lg_layer {
  /* SelfBlock (:240-272) */
  const float *Wqkv, *bqkv, *Wo, *bo, *sf0_w, *sf0_b, *sln_g, *sln_b, *sf3_w, *sf3_b;
  /* CrossBlock (:275-330) */
  const float *Wqk, *bqk, *Wv, *bv, *Wco, *bco, *cf0_w, *cf0_b, *cln_g, *cln_b, *cf3_w, *cf3_b;
  /* optional (NULL: two launches): the rows of Wqk followed by the rows of Wv as ONE [2d, d] matrix, and [bqk | bv] -- to_qk and
   * to_v then run as one launch (d = 256 with 64-wide heads; same results) */
  const float *Wqk_v, *bqk_v;
} einx_lg_layer;

typedef struct einx_lg_weights {
  size_t struct_size; /* sizeof(einx_lg_weights) of the caller's header (checked) */
  size_t layer_size;  /* sizeof(einx_lg_layer): the stride of `layers` (checked) */
  const float* in_w; /* input_proj [d,input_dim] or NULL (Identity) */
  const float* in_b;
  const float* Wr;   /* posenc.Wr [head_dim/2,2] */
  const float* proj_w; /* log_assignment[last].final_proj [d,d] */
  const float* proj_b;
  const float* match_w; /* log_assignment[last].matchability [1,d] */
  const float* match_b;
  int32_t n_layers, heads, d, input_dim;
  float filter_threshold;
  const einx_lg_layer* layers; /* host array of n_layers */
} einx_lg_weights;

/* normalize_keypoints (lightglue.py:137-148): kpts [rows,cols>=2] -> out [rows,out_cols>=2] with
 * out[:, :2] = (kpt - (h,w)/2) / (max(h,w)/2) and zeros in further columns.  The batched (B>1)
 * LightGlue call returns matched keypoints in these coordinates (lightglue.py:677-687). */
int einx_normalize_keypoints(const float* kpts, int rows, int cols, float h, float w, float* out, int out_cols, void* stream);

/* Model widths (lightglue.py:456-461): d = descriptor_dim = heads x head_dim, head_dim any multiple of 4 up to 256 (the attention
 * kernel is instantiated for 32 / 64 / 128 / 256; other widths run the next larger one on zero-padded heads; anything else is refused);
 * Wr is [head_dim/2, 2].  d = 256 with 4 heads of 64 -- every EI-Nexus configuration -- runs kernels instantiated for those widths.
 * einx_lg_ws_bytes assumes 64-wide heads (heads = d / 64); 0 = unsupported widths. */
size_t einx_lg_ws_bytes(int B, int cap0, int cap1, int d, int input_dim);
size_t einx_lg_ws_bytes_heads(int B, int cap0, int cap1, int d, int heads, int input_dim);
/* kpts [B,cap,3] (first two columns used), desc [B,cap,input_dim], counts device int32.
 * size0/size1: image sizes (H,W) used by normalize_keypoints (:137-148).
 * outputs as einx_mnn; scores are exp(max log-assignment) for mutual matches (:402-418);
 * ref0/ref1: optional ref_descriptors, or NULL.  ref_layers 0/1: [B,cap,d], the last layer
 * (eval, lightglue.py:626-629); ref_layers == n_layers: [B,n_layers,cap,d], every layer
 * (what the reference stacks when self.training, lightglue.py:626-629,709-710).
 * Schedule (same results either way): with cap0 == cap1 the two sides are stacked in the workspace and every layer is one
 * launch over 2B entries; grids of fewer than 256 128x128 tiles (up to 7 pairs) run their linears on 64x64 tiles.
 * ws: device scratch of einx_lg_ws_bytes_heads(B, cap0, cap1, d, heads, input_dim) bytes; rows of the outputs past an entry's
 * count are left unwritten. */
int einx_lightglue(const einx_lg_weights* w, const float* kpts0, const float* desc0, const int32_t* n, int cap0, const float* kpts1,
                   const float* desc1, const int32_t* m, int cap1, int B, float h0, float w0, float h1, float w1, void* ws,
                   int64_t* matches0, int64_t* matches1, float* scores0, float* scores1, float* la, float* ref0, float* ref1,
                   int ref_layers, void* stream);

/* ------------------------------------------------------------------------------------------
 * Event representation (the step before the path)
 * events of B samples are concatenated; offsets_host[B+1] (HOST array) delimits the samples.
 * x, y, p: float32 [N]; t: float64 [N] raw timestamps (normalised inside exactly like
 * datasets/representations.py:8-21).
 * ---------------------------------------------------------------------------------------- */
size_t einx_events_ws_bytes(int B, int H, int W);                                        /* workspace of einx_events_mask */
size_t einx_voxel_ws_bytes(int B, int bins, int H, int W, int64_t total_events);         /* workspace of einx_voxel_grid */
/* events_to_voxel_grid (datasets/representations.py:67-124): grid [B,bins,H,W].  Deterministic: per voxel the contributions
 * are added in the order of the reference's serial path (corner (dx,dy,dt) major, then event order, :94-114), so two calls give
 * the same bits and the un-normalised grid is bit-equal to a sequential restatement.  offsets_host[0] must be 0;
 * ws_bytes >= einx_voxel_ws_bytes(B, bins, H, W, offsets_host[B]) is checked. */
int einx_voxel_grid(const float* x, const float* y, const double* t, const float* p, const int64_t* offsets_host, int B, int bins, int H,
                    int W, int normalize, float* grid, void* ws, size_t ws_bytes, void* stream);
/* draw_events_accumulation_image(...) > 0 (datasets/visualize.py:23-50,
 * test_events-image_same-time.py:137): mask uint8 [B,H,W] */
int einx_events_mask(const float* x, const float* y, const int64_t* offsets_host, int B, int H, int W, void* ws, uint8_t* mask,
                     void* stream);

/* Host-side helper (no kernel, no device access): concatenates the B per-sample event arrays of a batch into the flat
 * x / y / p (fp32) and t (fp64) HOST arrays einx_voxel_grid / einx_events_mask read after an upload, converting element types
 * (C casts: the rounding of numpy's astype), and writes offsets[B + 1].  `threads` host threads share the copy (page-locked
 * destinations make the upload one asynchronous copy per array).  Replaces the per-sample tensor building of
 * datasets/representations.py:67-80 + a Python host's np.concatenate passes. */
enum { EINX_EV_F32 = 0, EINX_EV_F64, EINX_EV_I64, EINX_EV_I32, EINX_EV_I16, EINX_EV_U16, EINX_EV_I8, EINX_EV_U8, EINX_EV_U32, EINX_EV_U64 };
typedef struct einx_event_arrays {
  const void *x, *y, *t, *p;                 /* host arrays of n elements each */
  int32_t x_type, y_type, t_type, p_type;    /* EINX_EV_* */
  int64_t n;
} einx_event_arrays;
int einx_events_pack(const einx_event_arrays* samples, int B, float* x, float* y, double* t, float* p, int64_t* offsets, int threads);

/* ------------------------------------------------------------------------------------------
 * Handle-level extractor: ONE call enqueues a whole network (SURVEY.md 8b's coarse ABI)
 * replaces: VGGExtractor.forward / VGGExtractorNP.forward
 *             core/modules/event_extractors/EventExtractors.py:517-624, :331-434
 *           SuperPointv1.forward  core/modules/image_extractors/superpoint_extractor.py:345-480
 *           SiLKModel.forward     core/modules/image_extractors/silk_extractor.py:177-257
 * up to (not including) the Python output dict.  The handle owns copies of the layer descriptors only;
 * weights stay in caller-owned device memory (einx_conv_repack / einx_bn_fold images).  einx_extract makes no
 * allocation and no host synchronisation; it is bit-identical to the same sequence of op-level calls.
 * ---------------------------------------------------------------------------------------- */
typedef struct einx_extractor einx_extractor; /* opaque */

typedef struct einx_extractor_desc {
  size_t struct_size;  /* sizeof(einx_extractor_desc) of the caller's header (checked by einx_extractor_create) */
  int32_t cell;        /* 8: SuperPoint-shaped (65-channel detector head, coarse descriptors); 1: SiLK-shaped */
  int32_t n_backbone, n_det, n_desc;
  const einx_conv_desc* backbone; /* host arrays, copied by einx_extractor_create */
  const einx_conv_desc* det_head;
  const einx_conv_desc* desc_head;
  int32_t dilate_mask; /* event extractors dilate the events mask 3x3 (EventExtractors.py:544-550) */
  int32_t border;      /* remove_borders */
  int32_t nms_radius;
  int32_t top_k;       /* detection_top_k (0 = none) */
  float det_thr;       /* detection_threshold */
  int32_t ordering_xy; /* 0: (y,x,p)  1: (x,y,p) */
  float desc_scale;    /* descriptor_scale_factor */
  float input_div;     /* SuperPointv1 scales its input in place (`image /= 255.0`, superpoint_extractor.py:372); 0 = off */
  /* optional (NULL = off): det_head[0] and desc_head[0] -- two 3x3 layers that read the same backbone features -- as ONE layer whose
   * output channels are det_head[0]'s followed by desc_head[0]'s (weights / bias / BatchNorm concatenated by the caller).  Used for
   * single images (B == 1, two-layer heads): one launch instead of two on the latency-bound chain of a single-pair forward; every
   * output channel is the same k-ordered chain, so results are bit-identical.  Larger batches keep the two launches. */
  const einx_conv_desc* merged_head0;
} einx_extractor_desc;

typedef struct einx_extract_shapes_t {
  int32_t Hp, Wp, h0, w0; /* padded size and top/left padding (Padder, utils/util.py:6-15) */
  int32_t hc, wc;         /* head resolution (Hp/cell, Wp/cell) */
  int32_t feat_channels, det_channels, desc_dim;
  int32_t cap;            /* keypoint rows per image the detector can emit (top-k quantile arithmetic) */
} einx_extract_shapes_t;

typedef struct einx_extract_out {
  float* feats;     /* [B,feat_channels,hc,wc]   backbone_feats */
  float* logits;    /* [B,det_channels,hc,wc] */
  float* raw;       /* [B,desc_dim,hc,wc]        raw_descriptors */
  float* prob;      /* [B,det_channels,hc,wc]    probability */
  float* score;     /* [B,1,Hp,Wp]               padded score map (masked, border-zeroed) */
  float* coarse;    /* [B,desc_dim,hc,wc]        coarse_descriptors (cell 8) or NULL */
  float* raw_cl;    /* [B,hc*wc,desc_dim]        channels-last raw copy for the sparse sampler (cell 8) or NULL */
  float* nms;       /* [B,H,W]                   thresholded NMS map, cropped, or NULL */
  float* positions; /* [B,cap,3] */
  int32_t* indices; /* [B,cap] */
  int32_t* counts;  /* [B] */
  float* thr;       /* [B] */
  int32_t* not_converged; /* [B] 1: the NMS fix-point of this image needs more passes than were enqueued (see einx_detect) */
  float* sparse_desc;     /* [B,cap,desc_dim] */
  int32_t cap;
  float* score_crop;      /* optional (NULL: off): [B,1,H,W] the un-padded score map of the output dict (Padder.unpad, utils/util.py:42-50),
                             written by the score kernel instead of by a crop + clone afterwards */
} einx_extract_out;

/* Optional content watch of a network's weights riding on an einx_extract_watch call (what einx_params_hash does as a launch
 * of its own): the rows of `table` are hashed by spare workgroups of the call's last kernel and compared with `ref`; a
 * difference ORs 1 into *stale.  The library never clears *stale: it belongs to whoever recorded `ref` (a stale watch stays
 * stale until its owner re-records the hashes and zeroes the word).  A torch-free host that never edits weights behind the
 * library's back has no use for it and calls einx_extract. */
typedef struct einx_weight_watch {
  size_t struct_size;     /* sizeof(einx_weight_watch) */
  int32_t n;              /* rows (0: off) */
  const int64_t* table;   /* device [n][2]: (pointer, number of 32-bit words) per row, as einx_params_hash */
  const uint64_t* ref;    /* device [n] hashes stored when the native weight images were built */
  uint64_t* scratch;      /* device [n] */
  int32_t* stale;         /* device int32, OR-ed with 1 on a mismatch */
} einx_weight_watch;

einx_extractor* einx_extractor_create(const einx_extractor_desc* d); /* NULL on error (einx_last_error) */
void einx_extractor_destroy(einx_extractor* e);
/* Optional: choose the library's side stream and create the fork / join events of `stream` NOW instead of at the first
 * einx_extract that forks on it.  HIP deals streams onto a few hardware queues on the GPU's compute pipes (GPU_MAX_HW_QUEUES, 4
 * by default), and which one a stream got depends on everything the process created before it: a side stream on its caller's
 * queue serialises the fork (single pair: 0.77 -> 1.06 ms).  Since round 6 the choice is PROBED (einx_stream_overlap_us): the
 * library keeps EINX_FORK_STREAM_POOL streams per device (created together at the first use, never destroyed) and lends
 * `stream` the one that runs beside it and beside the side streams of the two most recently used other sides of the device.
 * The probe synchronises with `stream` (a few hundred microseconds per pool stream tried, once per (device, stream)); it is
 * skipped while `stream` is capturing.  A host that wants that cost at start-up calls this once per caller stream; the Python
 * package does so when a model is first used on a device. */
int einx_fork_stream_prepare(void* stream);
/* The same with up to 8 further streams the new side stream should stay clear of (a host that runs two extractors on two streams
 * names the other extractor's stream and its side stream, einx_fork_stream_of).  No effect when `stream` has a side already. */
int einx_fork_stream_prepare_beside(void* stream, void* const* beside, int n_beside);
/* The library keeps at most EINX_FORK_STREAMS_MAX (device, caller stream) sides; a call on a further stream evicts the least
 * recently used one that no call is using at that moment (its events are destroyed once their enqueued work has drained; its
 * stream goes back to the pool), so a server that creates a stream per request does not grow HIP streams / events without
 * bound: EINX_FORK_STREAM_POOL streams per device for the life of the process, whatever the host does.
 * einx_fork_stream_release drops the side of `stream` now (a host that destroys a stream it has made calls on; optional).
 * einx_fork_stream_count: sides alive at the moment (tests, diagnostics). */
#define EINX_FORK_STREAMS_MAX 16
#define EINX_FORK_STREAM_POOL 8
int einx_fork_stream_release(void* stream);
int einx_fork_stream_count(void);
/* the side stream (a hipStream_t) the library forks `stream` onto, NULL when none exists yet (diagnostics: einx_stream_overlap_us) */
void* einx_fork_stream_of(void* stream);
int einx_extract_shapes(const einx_extractor* e, int H, int W, einx_extract_shapes_t* shapes);
/* nms_iters: NMS pass budget per call (see einx_detect); <= 0 selects the default (8) in BOTH functions below.  The
 * workspace size depends on it (B x nms_iters convergence flags), so query and call must pass the same value. */
size_t einx_extract_ws_bytes(const einx_extractor* e, int B, int H, int W, int cap, int nms_iters);
/* in [B,cin,H,W] (modified in place only when input_div is set); mask [B,1,H,W] uint8 or NULL;
 * ws: device scratch, ws_bytes >= einx_extract_ws_bytes(e, B, H, W, out->cap, nms_iters) (checked).
 * Networks with 1/8-resolution heads (cell == 8; full-resolution networks while B x head pixels <= 8192) enqueue the descriptor
 * branch on a library-owned side stream between a fork and a join event of `stream`; every return path has `stream` wait for the
 * join.  The side stream and its two events are created on the FIRST such call for a (device, stream) pair (or by
 * einx_fork_stream_prepare), shared by every handle of the process and bounded in number (einx_fork_stream_release) -- so make one un-captured call (or the prepare call) per
 * stream before capturing einx_extract into a hipGraph (stream / event creation is not capturable).  While `stream` is being
 * captured the branches are enqueued in line (no fork: a fork nested in a caller's own fork / join makes hipStreamEndCapture of
 * ROCm 7.2 crash).  Calls that fork from one stream are serialised on a mutex. */
int einx_extract(const einx_extractor* e, float* in, const uint8_t* mask, int B, int H, int W, int nms_iters, void* ws,
                 size_t ws_bytes, const einx_extract_out* out, void* stream);
/* the same call with the weight watch riding on it (watch == NULL or watch->n == 0: exactly einx_extract) */
int einx_extract_watch(const einx_extractor* e, float* in, const uint8_t* mask, int B, int H, int W, int nms_iters, void* ws,
                       size_t ws_bytes, const einx_extract_out* out, const einx_weight_watch* watch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics of the reference's test harness (the step after the path)
 *   MatchingRatio            core/metrics/matching_metrics.py:30-51
 *   MeanMatchingAccuracy@t   core/metrics/matching_metrics.py:84-156   (points ordering "yx")
 *   ValidDescriptorsDistance core/metrics/keypoints_metrics.py:160-290 (repeat./distance/angle @t)
 * Inputs are the device-side batch results of the path: keypoints [B,cap,3], descriptors
 * [B,cap,D], counts, matched keypoints [B,cap0,cols] + nmatch (einx_gather_matches), optional
 * homography [B,9] (NULL = identity, as test_events-image_same-time.py:146 uses).
 * out: [B, 1 + n_mma + 3*n_vdd] float64 = MR, MMA@t..., (Repeatability, ValidDistance, Angle)@t...
 * ---------------------------------------------------------------------------------------- */
typedef struct einx_metric_params {
  int32_t B, cap0, cap1, D, cols;
  int32_t H0, W0, H1, W1; /* img1_shape, img2_shape */
  int32_t kp_yx;          /* 1: keypoints are (y,x,..) (the extractors' "yx" ordering) */
  int32_t n_mma, n_vdd;
  float mma_thr[4];
  float vdd_thr[4];
  int32_t rep_nan_if_empty; /* 1: Repeatability@t = NaN when NEITHER image keeps a keypoint after the visibility filter
                               (class Repeatability emits no entry then, keypoints_metrics.py:126-128; VDD reports 0) */
} einx_metric_params;
size_t einx_metrics_ws_bytes(const einx_metric_params* p);
int einx_pair_metrics(const einx_metric_params* p, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                      const int32_t* n, const int32_t* m, const float* mk0, const float* mk1, const int32_t* nmatch,
                      const float* homography, void* ws, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EINX_H */
