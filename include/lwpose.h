/*
 * lwpose.h — C ABI of the MI355X-native Lightweight-OpenPose inference path.
 *
 * The reference (vivek87799/lightweight-human-pose-estimation.pytorch) is pure Python and has no
 * FFI/plugin registry; its drop-in boundary is duck typing at a few Python call sites.  This header
 * is what a binding for those call sites talks to (the ctypes stub is in INTEGRATION.md and in
 * lwpose_amd/_lib.py).  Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = LWP_OK, negative = error class; nothing throws across the ABI;
 *     lwp_last_error(h) returns a human-readable message for the last failing call on that handle
 *     (h == NULL: last failure of a call that had no handle).
 *   - plain pointers and sizes only; the caller owns every buffer it passes, the handle owns all
 *     device memory it allocates.  Pointer arguments tagged "mem" accept host or device memory
 *     according to the LWP_MEM_* flag passed with them (device pointers = zero-copy from
 *     torch-ROCm tensors via data_ptr()).
 *   - one handle per (device, stream); a handle is not thread-safe; distinct handles are independent (handles on
 *     different devices may live in one process; the usual deployment is one process per GPU).
 *   - all maps are float32.  Network tensors crossing the ABI are NCHW (as the reference's
 *     nn.Module sees them); post-processing maps are HWC (as numpy sees them after demo.py:71-76).
 */
#ifndef LWPOSE_H
#define LWPOSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lwp_context* lwp_handle;

enum {
    LWP_OK = 0,
    LWP_ERR_ARG = -1,       /* bad argument / shape                               */
    LWP_ERR_HIP = -2,       /* HIP runtime error (message has the hipError string) */
    LWP_ERR_STATE = -3,     /* call order (e.g. forward before load_weights)       */
    LWP_ERR_CAPACITY = -4,  /* a peak / key-point / connection list overflowed its configured capacity */
    LWP_ERR_NOGPU = -5,     /* no usable HIP device                                */
    LWP_ERR_UNBOUND = -6    /* the reference would raise UnboundLocalError here (keypoints.py:116,137) */
};

enum { LWP_MEM_HOST = 0, LWP_MEM_DEVICE = 1 };
enum { LWP_LAYOUT_NCHW = 0, LWP_LAYOUT_NHWC = 1 };
enum { LWP_F32 = 0, LWP_BF16 = 1 };   /* storage/MFMA dtype of the conv stack; accumulation, bias,
                                         activations and all post-processing are always f32/f64 */

/* role codes returned by lwp_param_spec */
enum { LWP_ROLE_CONV_W = 0, LWP_ROLE_CONV_B = 1, LWP_ROLE_BN_W = 2, LWP_ROLE_BN_B = 3,
       LWP_ROLE_BN_MEAN = 4, LWP_ROLE_BN_VAR = 5, LWP_ROLE_BN_NBT = 6 };

int lwp_version(void);

/* ---- parameter table: replaces nn.Module.state_dict() key/shape enumeration
 *      (models/with_mobilenet.py:89-112 + modules/conv.py:4-32; consumed by modules/load_state.py:4-15).
 *      No GPU needed. */
int lwp_param_count(int num_refinement_stages, int num_channels, int num_heatmaps, int num_pafs);
int lwp_param_spec(int num_refinement_stages, int num_channels, int num_heatmaps, int num_pafs,
                   int index, char* name, int name_cap, int64_t shape[4], int* ndim, int* role);

/* ---- lifetime: replaces PoseEstimationWithMobileNet(...) + net.cuda() (demo.py:156, demo.py:82-84) */
int lwp_create(int device_id, int num_refinement_stages, int num_channels, int num_heatmaps,
               int num_pafs, int dtype, lwp_handle* out);
int lwp_destroy(lwp_handle h);
const char* lwp_last_error(lwp_handle h);

/* ---- stream ordering: the reference's net(tensor_img) runs on torch's CURRENT stream (demo.py:64-68), so a caller never
 *      synchronises by hand.  The handle computes on its own non-blocking stream; with lwp_set_stream(h, s, 1) every later
 *      entry point that takes or returns DEVICE memory (a) makes its stream wait — by event, without blocking the host — for
 *      the work queued on `s` so far (the producers of its inputs and the last users of the buffers it overwrites), and (b) makes
 *      `s` wait for the results it leaves on the device.  `caller_stream` is a hipStream_t (NULL = the legacy default stream,
 *      which is torch's default current stream).  enable = 0 restores the initial state: no ordering, the caller synchronises
 *      (lwp_synchronize) around device-memory hand-overs.  enable = 2: (a) only — device results stay on the handle's stream for
 *      its own next call (the frame tensor between lwp_preprocess_u8 and lwp_infer_poses inside infer_fast, demo.py:59-68, never
 *      meets the caller's stream), which saves the cross-queue wait.  An idle caller stream costs nothing in any mode (it is
 *      queried first).  Host-memory results are complete on return in all modes. */
int lwp_set_stream(lwp_handle h, void* caller_stream, int enable);

/* capacities of the post-processing lists (defaults 2048 / 128 / 4096 / 256); overflow => LWP_ERR_CAPACITY */
int lwp_set_capacity(lwp_handle h, int max_peaks_per_channel, int max_kpts_per_type,
                     int max_connections_per_limb, int max_pose_entries);

/* ---- weights: replaces net.load_state_dict(...) at the end of load_state (modules/load_state.py:15).
 *      names[i] = state_dict key, ptrs[i] = host float32 data (int64 for num_batches_tracked, ignored),
 *      shapes = n x 4 (unused dims 1), ndims[i].  Conv weights OIHW.  Every key of lwp_param_spec must
 *      be present with the right shape.  Folds eval-mode BatchNorm (eps 1e-5, modules/conv.py:7) into
 *      the preceding conv, repacks to the kernels' layouts and uploads one blob. */
int lwp_load_weights(lwp_handle h, const char* const* names, const void* const* ptrs,
                     const int64_t* shapes, const int* ndims, int n);

/* packed weight blob, for one-shot replication to other GPUs (RCCL broadcast done by the host side
 * on a device buffer; replaces nn.DataParallel's per-iteration replicate, train.py:74). */
int lwp_weights_blob_bytes(lwp_handle h, size_t* bytes);
int lwp_weights_blob_export(lwp_handle h, void* dst_device, size_t bytes);
int lwp_weights_blob_import(lwp_handle h, const void* src_device, size_t bytes);

/* ---- network forward: replaces net(tensor_img) (demo.py:68, val.py:94;
 *      PoseEstimationWithMobileNet.forward, models/with_mobilenet.py:114-123).
 *      in: N x 3 x H x W float32 (mem).  outs: 2*(1+nref) pointers (mem, same kind as out_mem) to
 *      N x {num_heatmaps | num_pafs} x h x w float32 in the order [heat0, paf0, heat1, paf1, ...].
 *      Any H, W >= 8: the map size is that of three stride-2 convs, h = ((H-1)/2+1 -> ... ) (the reference pads to
 *      the stride, val.py:36-49, but does not require it).
 *      Runs on the handle's stream and synchronises it before returning when out_mem is host; device outputs are handed
 *      to the caller's stream by an event when lwp_set_stream is in effect (else call lwp_synchronize before reading them).
 *      Any N: the kernels address a tensor with 32-bit byte offsets, so a batch whose tensors would reach 2 GiB (at 368 x 656:
 *      N > 138 in fp32, N > 277 in bf16) is processed in equal chunks inside the call — same results as separate calls.
 *      fp32 results depend on the batch size at the 1e-6 level only: kernels (tile shapes, split-K, the fused head pair up to
 *      4096 pixels) are chosen by problem size, and their summation orders differ. */
int lwp_forward(lwp_handle h, const float* in, int in_mem, int N, int H, int W,
                float* const* outs, int out_mem);

/* ---- frame pre-processing on the device: replaces demo.py:55-64 (scale = net_input_height / H; cv2.resize of the
 *      uint8 frame with fx = fy = scale, INTER_CUBIC; normalize, val.py:30-33; pad_width, val.py:36-49; HWC -> 1x3xHxW
 *      float32).  lwp_preprocess_dims is pure host arithmetic: the scaled size (round half to even of H*scale, W*scale),
 *      the padded size out_h x out_w, pad = [top, left, bottom, right] and scale, exactly as infer_fast returns them.
 *      lwp_preprocess_u8: img is H x W x 3 uint8 (mem), out is a DEVICE buffer of 3*out_h*out_w float32.
 *      pad_value / img_mean: 3 doubles each (the reference's defaults are (0,0,0) and (128,128,128)); img_scale 1/256.
 *      The pad value is written as is (the reference pads AFTER normalising).  Runs on the handle's stream. */
int lwp_preprocess_dims(int H, int W, int net_input_height, int stride, int* scaled_h, int* scaled_w,
                        int* out_h, int* out_w, int* pad, double* scale);
int lwp_preprocess_u8(lwp_handle h, const unsigned char* img, int img_mem, int H, int W, int net_input_height, int stride,
                      const double* pad_value, const double* img_mean, double img_scale, float* out_device);

/* ---- image side of ONE scale of the multi-scale driver: replaces val.py:84-93 for N same-sized uint8 frames
 *      (normalize, val.py:30-33: float64 (u8 - mean) * scale;  cv2.resize(normed_img, (0,0), fx=fy=ratio, INTER_CUBIC) on
 *      the float64 image: float32 cubic coefficients (A = -0.75), float64 left-to-right sums, horizontal pass then vertical
 *      pass;  pad_width(scaled, stride, pad_value, [base_height, max(scaled_w, base_height)]), val.py:36-49;  HWC -> NCHW
 *      float32, val.py:93).  lwp_scale_dims is pure host arithmetic: scaled size = round-half-even(H*ratio, W*ratio), the
 *      padded size out_h x out_w and pad = [top, left, bottom, right] exactly as pad_width returns them.
 *      lwp_preprocess_scaled_u8: imgs is N x H x W x 3 uint8 (mem), out a DEVICE buffer of N*3*out_h*out_w float32.
 *      Runs on the handle's stream (the resize tables of a geometry are built once and kept on the device). */
int lwp_scale_dims(int H, int W, double ratio, int base_height, int stride, int* scaled_h, int* scaled_w,
                   int* out_h, int* out_w, int* pad);
int lwp_preprocess_scaled_u8(lwp_handle h, const unsigned char* imgs, int img_mem, int N, int H, int W, double ratio,
                             int base_height, int stride, const double* pad_value, const double* img_mean, double img_scale,
                             float* out_device);
/* the same for float32 frames: val.normalize (val.py:30-33) starts with np.array(img, dtype=np.float32), so an image of any
 * other dtype than uint8 is the float32 case after that cast (done by the caller) */
int lwp_preprocess_scaled_f32(lwp_handle h, const float* imgs, int img_mem, int N, int H, int W, double ratio,
                              int base_height, int stride, const double* pad_value, const double* img_mean, double img_scale,
                              float* out_device);

/* ---- bicubic up-sampling: replaces cv2.resize(map, (0,0), fx=r, fy=r, INTER_CUBIC) on float maps
 *      (demo.py:72,76; val.py:98,105).  src: N x C x h x w (mem);  dst: N x (h*r) x (w*r) x C (mem). */
int lwp_upsample(lwp_handle h, const float* src, int src_mem, int N, int C, int hs, int ws, int ratio,
                 float* dst, int dst_mem);

/* ---- one scale of the multi-scale average: replaces val.py:96-101 / 103-108 (x`up_ratio` cubic up-sampling of one
 *      stage output, crop of the padding pad = [top, left, bottom, right], cubic resize to (dst_w, dst_h),
 *      accum = accum + maps / n_scales).  maps: N x C x hs x ws float32 (mem); accum: N x dst_h x dst_w x C float32 HWC (mem);
 *      the N frames share one geometry (same pad, same destination size).  init != 0: accum is taken as zero (the first
 *      scale of val.py:86-87) and need not be initialised by the caller. */
int lwp_multiscale_accumulate(lwp_handle h, const float* maps, int maps_mem, int N, int C, int hs, int ws, int up_ratio,
                              const int* pad, int dst_h, int dst_w, int n_scales, float* accum, int accum_mem, int init);

/* ---- extract_keypoints: replaces modules/keypoints.py:16-48 for one heat-map channel.
 *      heatmap: H x W float32 with row stride `row_stride` and pixel stride `pix_stride` (elements), host.
 *      It is thresholded IN PLACE (values < 0.1 -> 0) like the reference.  Outputs (host, capacity `cap`):
 *      xs, ys (int64), scores (float32) in the reference's order (x ascending, then y).  *count = number
 *      kept after the radius-6 suppression; ids are total_keypoint_num + index (caller side). */
int lwp_extract_keypoints(lwp_handle h, float* heatmap, int H, int W, int64_t row_stride, int64_t pix_stride,
                          int64_t* xs, int64_t* ys, float* scores, int cap, int* count);

/* ---- group_keypoints: replaces modules/keypoints.py:51-201.
 *      kpts: K x 4 float64 rows (x, y, score, id) concatenated by type; type_counts[18].
 *      pafs: H x W x num_pafs float32 HWC (mem).  demo != 0 -> int() truncation, else round-half-even.
 *      pose_entries: cap_entries x 20 float64 out; *n_entries out. */
int lwp_group_keypoints(lwp_handle h, const double* kpts, const int* type_counts,
                        const float* pafs, int pafs_mem, int H, int W, int demo,
                        double* pose_entries, int cap_entries, int* n_entries);

/* ---- fused frame pipeline: replaces the body of run_demo's loop up to group_keypoints
 *      (demo.py:93-100: infer_fast -> 18 x extract_keypoints -> group_keypoints) for a batch.
 *      in: N x 3 x H x W float32, already normalised and padded (mem).  The up-sampled maps are never
 *      materialised: peaks and PAF samples are interpolated on the fly with the same arithmetic.
 *      Outputs (host): for frame f, kpt_counts[f*18 + t] key-points of type t; kpts rows
 *      (x, y, score, id) float64 at kpts + f*kpt_cap*4; entries at entries + f*entry_cap*20;
 *      n_entries[f]. */
int lwp_infer_poses(lwp_handle h, const float* in, int in_mem, int N, int H, int W,
                    int upsample_ratio, int demo,
                    int* kpt_counts, double* kpts, int kpt_cap,
                    double* entries, int entry_cap, int* n_entries);

/* same post-processing from already computed maps: the low-resolution stage outputs net(x) returns (demo.py:70,74),
 * heat N x num_heatmaps x h x w, paf N x num_pafs x h x w, LWP_LAYOUT_NCHW; or the full-resolution averaged maps of the
 * multi-scale path (val.py:129-134), N x h x w x C, LWP_LAYOUT_NHWC with upsample_ratio = 1.  float32 (mem). */
int lwp_poses_from_maps(lwp_handle h, const float* heat, const float* paf, int mem, int layout, int N, int hs, int ws,
                        int upsample_ratio, int demo,
                        int* kpt_counts, double* kpts, int kpt_cap,
                        double* entries, int entry_cap, int* n_entries);

/* enqueue-only variant for benchmarking / pipelining: same work, results stay on the device until
 * lwp_fetch_poses; does not synchronise.  `in` must be device memory. */
int lwp_infer_poses_async(lwp_handle h, const float* in_device, int N, int H, int W,
                          int upsample_ratio, int demo);
int lwp_fetch_poses(lwp_handle h, int* kpt_counts, double* kpts, int kpt_cap,
                    double* entries, int entry_cap, int* n_entries);

/* ---- pipelined streaming (video): two result slots.  lwp_pipeline_submit enqueues the network of one batch on the
 *      handle's main stream and its post-processing + result copy on a second stream, and returns at once;
 *      lwp_pipeline_fetch(slot) waits for that slot only.  With submit(k) issued before fetch(k-1), the
 *      post-processing and host fetch of batch k-1 overlap the network of batch k (replaces the strictly serial
 *      frame loop of run_demo, demo.py:91-114; results are identical).  A slot must be fetched before it is reused. */
int lwp_pipeline_submit(lwp_handle h, const float* in_device, int N, int H, int W, int upsample_ratio, int demo, int slot);
int lwp_pipeline_fetch(lwp_handle h, int slot, int* kpt_counts, double* kpts, int kpt_cap,
                       double* entries, int entry_cap, int* n_entries);

/* ---- measurement helpers (bench.py): time `iters` back-to-back enqueues with HIP events on the
 *      handle's own stream.  what: 0 = forward only, 1 = full infer_poses.  ms_total out. */
int lwp_time_pipeline(lwp_handle h, const float* in_device, int N, int H, int W, int upsample_ratio,
                      int demo, int what, int iters, float* ms_total);
/* per-kernel-class device time of ONE pass, measured with HIP events around each launch on the
 * handle's stream: classes 0=stem 1=depthwise 2=pointwise-1x1 3=dense-3x3 4=post 5=other.
 * ms[6], launches[6] out. */
int lwp_profile_classes(lwp_handle h, const float* in_device, int N, int H, int W, int upsample_ratio,
                        int demo, int reps, float* ms, int* launches);
/* per-launch device time of one pass (averaged over reps): ms[i], kclass[i] for launch i in issue order.
 * kclass[i] & 0xff = kernel class as above; kclass[i] >> 8 = 1 + index of the first layer the launch covers
 * (a fused head pair is ONE launch covering two layers), 0 for a post-processing kernel. */
int lwp_profile_launches(lwp_handle h, const float* in_device, int N, int H, int W, int upsample_ratio,
                         int demo, int reps, float* ms, int* kclass, int cap, int* n_launches);
int lwp_synchronize(lwp_handle h);

/* ---- introspection / per-layer parity (tests): the layer list mirrors the module tree of
 *      models/with_mobilenet.py:92-112.  lwp_debug_layer_output runs the first layer_index+1 layers on
 *      `in` (host, N x 3 x H x W) and copies that layer's output to dst (host) as NCHW float32
 *      N x cout x h x w; dims are returned in out_dims[4]. */
int lwp_layer_count(lwp_handle h);
int lwp_layer_info(lwp_handle h, int layer_index, char* name, int name_cap, int* kind, int* cin, int* cout,
                   int* ksize, int* stride, int* dilation, int64_t* macs_per_pixel /* algorithmic multiply-adds */);
/* average device time (ms) of `iters` back-to-back launches of one layer on the current buffers */
int lwp_debug_time_layer(lwp_handle h, int layer_index, int N, int H, int W, int iters, float* ms_avg);
int lwp_debug_layer_output(lwp_handle h, const float* in, int N, int H, int W, int layer_index,
                           float* dst, size_t dst_floats, int out_dims[4]);
/* name of the kernel variant the LAST lwp_debug_layer_output / lwp_profile_launches pass picked for a layer, e.g.
 * "dw_tiled<cc=64,s=1,d=1,ph=8>" or "gemm_bf16_ar<256,4,2,3>" ("" before any such pass; the second layer of a fused head
 * pair reports the pair's kernel).  Lets a test prove that the kernel it means to cover is the one that ran: the launchers
 * choose by problem size, and the A/B switches (environment, read once per handle in lwp_create) only override that choice. */
int lwp_debug_layer_variant(lwp_handle h, int layer_index, char* name, int name_cap);
/* frames one launch sequence of an N x 3 x H x W call takes (N unless a tensor of the pass would reach the kernels' 2 GiB
 * addressing range: lwp_forward / lwp_infer_poses* / lwp_pipeline_submit then walk the batch in equal chunks of this size) */
int lwp_debug_frames_per_pass(lwp_handle h, int N, int H, int W);
/* intermediate counts of the grouping kernels for frame `frame` of the handle's LAST lwp_infer_poses / lwp_infer_poses_async /
 * lwp_poses_from_maps call (after its results were fetched): candidate peaks per key-point type before the NMS [18], key-points per
 * type after it [18], scored connection candidates per limb [19], connections picked per limb [19].  Tests / tools: which
 * form of nms_kernel / match_kernel (register form up to 64 candidates, LDS form beyond) a workload exercises. */
int lwp_debug_post_counts(lwp_handle h, int frame, int* peaks18, int* kpts18, int* candidates19, int* picked19);

#ifdef __cplusplus
}
#endif
#endif /* LWPOSE_H */
