/*
 * poserisk_hip.h -- C ABI of libposerisk_hip.so, the MI355X (gfx950) implementation of
 * PoseRisk's per-frame hot path.
 *
 * The reference (hygenie1228/PoseRisk_RELEASE) is pure Python and has no FFI of its own:
 * its "plugin surface" is the set of Python callables main/run.py -> lib/core/base.py use
 * (SURVEY.md 8b).  Each entry point below replaces the arithmetic behind one of those
 * callables and cites it; the Python drop-in modules under poserisk_release_amd/dropin/
 * bind these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - Every function returns 0 on success or a negative pr_status; pr_last_error() gives
 *     the message of the calling thread's last failure.  No exception crosses the boundary.
 *   - The CALLER owns every I/O buffer.  Pointers named *_dev are device (HBM) pointers
 *     (e.g. torch.Tensor.data_ptr()); pointers named *_host are host pointers.  The
 *     library never frees caller memory.  Handles own packed weights / model constants /
 *     workspaces and release them in *_destroy.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Compute calls
 *     are asynchronous on that stream and perform no host synchronisation, allocation or
 *     blocking copy, so they may be captured into a hipGraph.
 *   - One handle per device; a handle is not thread-safe; distinct handles are independent.
 *   - Layouts are row-major with the last index fastest; float = IEEE binary32.
 */
#ifndef POSERISK_HIP_H
#define POSERISK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pr_status {
  PR_OK = 0,
  PR_ERR_INVALID = -1,   /* bad argument (shape, null pointer, unsupported size)      */
  PR_ERR_HIP = -2,       /* a HIP runtime call failed (message has the HIP error)     */
  PR_ERR_NO_DEVICE = -3, /* no gfx950 device visible                                  */
  PR_ERR_CAPACITY = -4   /* batch larger than the handle was created for              */
} pr_status;

const char* pr_last_error(void);
int pr_abi_version(void); /* bumps on any signature change */
/* "gfx950 release" for the shipped build; an ablation / experiment build names its macros (PR_TIMING_HOOKS builds compute
 * wrong results on purpose).  bench.py prints it as `library`. */
const char* pr_build_info(void);

/* Capture guard (ABI 10).  pr_hmr_create, pr_smpl_create, pr_hmr_set_streams, pr_hmr_destroy and pr_smpl_destroy allocate,
 * copy and synchronise: reached while the caller is capturing a stream into a hipGraph they would invalidate the capture (and
 * the process aborts at its next synchronisation).  They take no stream, so the caller declares the stream it enqueues on --
 * thread-local, declared != 0 sets it (NULL = the null stream), 0 clears it -- and those five entry points return
 * PR_ERR_INVALID, having touched nothing, while hipStreamIsCapturing() reports a capture on it; the handle passed to a refused
 * destroy stays valid.  The stand-alone test entries (which allocate and take a stream) check their own stream argument.  The
 * Python binding declares torch's current stream before each of those calls (poserisk_release_amd/_lib.py::declare_stream).
 * No reference counterpart: the reference never captures (lib/core/base.py:81-84 builds its models once, eagerly). */
int pr_declare_stream(void* stream, int declared);

/* The fp32 encoder's stem as ONE kernel (csrc/stem_pool_f32.hip; same arithmetic as above in fp32, weights in registers,
 * pooling in registers): x_dev f32 [B,112,112,12] (the 2x2 space-to-depth image), w_host f32[64,12,4,4] OIHW, bias_host f32[64]
 * -> y_dev f32 [B,56,56,64].  w_host must be a 7x7 kernel laid into the 4x4 taps' 8x8 window with a zero row and a zero
 * column in front -- W2[o][(2 di + dj) 3 + c][th][tw] = W[o][c][2 th + di - 1][2 tw + dj - 1], zero outside 0..6, as
 * pr_hmr_create builds it from conv1.weight -- the kernel packs the half-empty taps together; anything else is refused
 * (PR_ERR_INVALID).  Exported for parity tests and timing (allocates, synchronises). */
int pr_stem_pool_f32_nhwc(int device, const float* x_dev, const float* w_host, const float* bias_host, float* y_dev, int B,
                          int repeats, float* ms_out, void* stream);

/* ------------------------------------------------------------------------------------ */
/* a1-a3  HMR: ResNet-50 encoder + iterative regressor + rot6d->rotmat                   */
/* replaces: models.hmr(...) / spin_model(batch)   lib/core/base.py:81-84, :220          */
/* ------------------------------------------------------------------------------------ */
typedef struct pr_hmr pr_hmr_t;

/* Number of floats in the canonical weight blob (order below). */
size_t pr_hmr_weight_floats(void);

/*
 * weights_host: the SPIN state dict flattened in this order (all float32, PyTorch layouts):
 *   conv1.weight[64,3,7,7], bn1.{weight,bias,running_mean,running_var}[64];
 *   for L in 1..4, for i in block(L):  conv1.weight, bn1.*4, conv2.weight, bn2.*4,
 *       conv3.weight, bn3.*4, and for i==0: downsample.0.weight, downsample.1.*4;
 *   fc1.weight[1024,2205], fc1.bias, fc2.weight[1024,1024], fc2.bias,
 *   decpose.weight[144,1024], decpose.bias, decshape.weight[10,1024], decshape.bias,
 *   deccam.weight[3,1024], deccam.bias, init_pose[144], init_shape[10], init_cam[3].
 * BatchNorm (eval, eps 1e-5) is folded into the conv weights in double precision at
 * create time.  max_batch sizes the activation workspace (frames per forward call).
 * precision: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32; products and sums are exact-fp32 fmaf chains),
 * 1 = bf16 MFMA encoder with fp32 accumulate (regressor stays fp32).
 * conv_form (fp32 encoder only; a property of the handle, so one process may hold several): how the ten 3x3 /
 * stride-1 layers with >= 128 channels are computed on that kernel -- PR_CONV_FORM_DIRECT (implicit GEMM, the
 * reference's arithmetic up to summation order), PR_CONV_FORM_WINOGRAD_2X2 / _4X4 (F(2x2,3x3) / F(4x4,3x3) on Lavin &
 * Gray's points 0, +-1, +-2: 2.25x / 4x fewer multiplies, a different rounding pattern), PR_CONV_FORM_WINOGRAD_4X4_B
 * (F(4x4,3x3) on the points 0, +-11/16, +-3/2: the same cost as _4X4 with half its per-layer rounding error, every
 * transform constant exact in fp32), PR_CONV_FORM_DEFAULT (= PR_CONV_FORM_BUILTIN_DEFAULT; the environment
 * variable POSERISK_WINOGRAD moves this default only).  A three-digit value selects the form per ResNet stage, layer2 /
 * layer3 / layer4 (e.g. 244 = F(2x2) in layer2, F(4x4) in layer3 and layer4).  The built-in default is 5, chosen on error
 * DISTRIBUTIONS over 768 frames of trained-like stress weights, all joints, against an fp64 run of the network
 * (profiles/r04_wino_stats.txt; rms of the 6-D pose / 99th percentile of the rotation matrices / relative rms of the pooled
 * features, as multiples of the direct form's): form 5 1.02 / 1.05 / 1.03, form 4 1.19 / 1.20 / 1.19, form 244 (round 3's
 * default) 1.04 / 1.04 / 1.04, F(2x2) 0.99 / 1.00 / 0.99 -- at 4.29 / 4.31 / 4.45 / 4.53 ms of conv per 64 frames (direct:
 * 4.90).  The MAXIMUM over those 166 k samples (direct 4.8e-4, the fp32 oracle itself 1.1e-3) is set by the ~13 % of joints
 * whose 6-D vector is nearly degenerate under that random decoder and does not rank the forms.
 */
enum { PR_CONV_FORM_DEFAULT = -1, PR_CONV_FORM_DIRECT = 0, PR_CONV_FORM_WINOGRAD_2X2 = 2, PR_CONV_FORM_WINOGRAD_4X4 = 4,
       PR_CONV_FORM_WINOGRAD_4X4_B = 5, PR_CONV_FORM_BUILTIN_DEFAULT = 5 };
int pr_hmr_create(int device, const float* weights_host, size_t n_floats, int max_batch,
                  int precision, int conv_form, pr_hmr_t** out);
int pr_hmr_destroy(pr_hmr_t* h);

/* x_dev f32[B,3,224,224] NCHW in [0,1] (no mean/std normalisation: _img_utils.py:259-266)
 * -> rotmat_dev f32[B,24,3,3], betas_dev f32[B,10], cam_dev f32[B,3].
 * Any of the three outputs may be NULL.  xf_dev (optional, may be NULL) receives the
 * pooled encoder features f32[B,2048]; pose6d_dev (optional) the 6-D pose f32[B,144]. */
int pr_hmr_forward(pr_hmr_t* h, const float* x_dev, int B, float* rotmat_dev, float* betas_dev,
                   float* cam_dev, float* xf_dev, float* pose6d_dev, void* stream);

/* The encoder cuts a batch into n_streams contiguous sub-batches that run concurrently on internal
 * HIP streams (forked from / joined to the caller's stream with events; still asynchronous, still no
 * host synchronisation in pr_hmr_forward).  Frames are independent, so results are bit-identical for any
 * n_streams (1..8).  Default 1 (or the environment variable POSERISK_HMR_STREAMS at create time): on
 * MI355X at B=64 the lock-step sub-batches measured slower than one stream; overlapping WHOLE batches on
 * caller-side streams (one handle per stream) is what fills the tails.  This call reallocates workspaces
 * and synchronises the device: configuration time only. */
int pr_hmr_set_streams(pr_hmr_t* h, int n_streams);

/* A hint, not a mode: n_in_flight = how many handles' forward calls overlap on this device (one per caller-side stream /
 * pipeline lane).  The persistent kernels size their grids by it -- conv1x1_regw_f32 runs two workgroups per CU when a
 * batch has the GPU to itself and ONE when other batches' kernels should find room beside it (measured, B=64 fp32: one
 * batch in flight +2.0 % with two, three in flight +1.8 % with one; profiles/r04_experiments.txt section 5).  Results do not
 * depend on it (the assignment of work units to workgroups changes, no sum's order does).  Default 1. */
int pr_hmr_set_concurrency(pr_hmr_t* h, int n_in_flight);

/* Per-kernel timing of the conv launches (for bench.py's roofline): when enabled, every
 * conv launch of the NEXT forward calls is bracketed by hipEvents on `stream`.
 * pr_hmr_profile_read synchronises those events and returns, per conv layer (53 entries,
 * execution order), the accumulated milliseconds and the launch count since enable (a Winograd layer's three
 * kernels are one bracket; a convolution fused into another's launch -- a downsample branch in its conv3, a layer1
 * conv2 + conv3 pair -- reports zero launches and its work under the launch that carries it), the ALGORITHMIC FLOP per
 * frame (direct convolution) and, optionally, the FLOP the matrix pipes execute (K padding included,
 * (m+2)^2 products per Winograd tile).  While enabled the
 * encoder runs serially on the caller's stream (one sub-batch at a time) so each bracket is one kernel. */
int pr_hmr_profile_enable(pr_hmr_t* h, int on);
int pr_hmr_profile_read(pr_hmr_t* h, float* ms_per_layer_host, int* launches_per_layer_host,
                        double* flops_per_layer_per_frame_host, double* mfma_flops_per_layer_per_frame_host,
                        int n_layers);
int pr_hmr_num_conv_layers(void);
/* What one forward of B frames launches for the encoder's 53 convolutions: conv_launches = event brackets of the
 * profile above (a Winograd layer counts once), winograd_layers = how many of them are Winograd layers, each of which
 * is three kernels (transform, grouped GEMMs, transform).  Kernel launches between the layout change and the global
 * average pool = conv_launches + 2 * winograd_layers: scripts/pmc_summary.py refuses to summarise counter passes whose
 * dispatch count differs (round 4's summary silently missed a new kernel).  No reference counterpart (measurement). */
int pr_hmr_plan_counts(pr_hmr_t* h, int B, int* conv_launches, int* winograd_layers);
/* The conv form this handle really runs (what PR_CONV_FORM_DEFAULT resolved to at create time: the built-in default,
 * or POSERISK_WINOGRAD's value): 0, 2, 4, 5 or three digits.  scripts/validate_checkpoint.py bases its exit status on it. */
int pr_hmr_conv_form(pr_hmr_t* h);

/* Stand-alone conv + folded-BN bias + optional residual + optional ReLU on NHWC tensors:
 * the building block of the encoder, exported for per-shape parity tests and tuning.
 *   x_dev f32[B,H,W,Cin] (Cin % 4 == 0), w_host f32[Cout,Cin_real,KH,KW] (PyTorch OIHW;
 *   Cin_real <= Cin, extra input channels are treated as zero), bias_host f32[Cout] or NULL,
 *   res_dev f32[B,Ho,Wo,Cout] or NULL, y_dev f32[B,Ho,Wo,Cout].  Cout % 64 == 0.
 * tile_cfg -1 selects the built-in heuristic, >= 6 a tile configuration index (pr_conv_num_tile_cfgs(); 0..5 are reserved:
 * the first-generation kernel they selected was retired and they return PR_ERR_INVALID),
 * -2 / -4 the Winograd F(2x2,3x3) / F(4x4,3x3) form (the encoder runs its 3x3 / stride-1 layers with >= 128
 * channels as F(4x4,3x3); fp32, pad 1, no residual, Cin % 32 == 0: input transform, 16 / 36 grouped GEMMs in one
 * launch, output transform), 100 the row-panel form of a short-K 1x1 convolution, 300 (bf16, 1x1, Cin 128 -> Cout 512, with bias
 * and residual) the persistent kernel that keeps the weights in registers (csrc/expand_res_bf16.hip), 301 / 302 (bf16, 1x1 or
 * 3x3, Cin % 64 == 0, Cout % 128 == 0, no residual) the persistent kernel that deals the pixels evenly to one workgroup per
 * CU (csrc/conv_bal_bf16.hip; 302 forces channel blocks of 128), 200 + S (S = 2..8) the 64x64 tile with
 * every tile's K-steps dealt to S workgroups (split-K, fp32).  precision 1: x_dev, res_dev and y_dev hold bfloat16 (Cin % 8 == 0), the
 * weights are rounded to bfloat16, accumulation and bias stay fp32; only the LDS-DMA tile configs apply.
 * This call packs the weights on every invocation (it allocates and synchronises): test/tuning use only. */
int pr_conv_num_tile_cfgs(void);
int pr_conv2d_nhwc(int device, const void* x_dev, const float* w_host, const float* bias_host,
                   const void* res_dev, void* y_dev, int B, int H, int W, int Cin, int Cin_real,
                   int Cout, int KH, int KW, int stride, int pad, int relu, int tile_cfg,
                   int precision, int repeats, float* ms_out, void* stream);

/* The fused form of a first Bottleneck's tail, relu(bn3(conv3(t)) + bn_d(conv_d(x))) (SPIN models/hmr.py Bottleneck
 * with a downsample branch), as ONE GEMM whose K loop runs over t's channels and then over x's: exported for parity
 * tests (allocates, synchronises).  x1_dev [B,Ho,Wo,Cin1], w1_host f32[Cout,Cin1], x2_dev [B,H2,W2,Cin2] sampled at
 * (ho*stride2, wo*stride2), w2_host f32[Cout,Cin2], bias_host f32[Cout] or NULL -> y_dev [B,Ho,Wo,Cout].  precision as
 * pr_conv2d_nhwc (1: tensors hold bfloat16).  Channel counts are multiples of 32 (fp32) / 64 (bf16). */
int pr_conv1x1_dual_nhwc(int device, const void* x1_dev, const float* w1_host, const void* x2_dev,
                         const float* w2_host, const float* bias_host, void* y_dev, int B, int Ho, int Wo,
                         int Cin1, int H2, int W2, int Cin2, int stride2, int Cout, int relu, int tile_cfg,
                         int precision, void* stream);

/* A layer1 Bottleneck's conv2 + conv3 as one kernel (the 64-channel map between them stays in LDS): exported for
 * parity tests (allocates, synchronises).  x_dev [B,H,W,Cin] (Cin a power of two >= 32; bf16: >= 64), w2_host
 * f32[64,Cin,3,3] OIHW, b2_host f32[64], w3_host f32[N3,64], b3_host f32[N3], res_dev [B,H,W,N3] or NULL
 * -> y_dev [B,H,W,N3] = act(relu(conv3x3(x, w2) + b2) * w3^T + b3 + res), act = ReLU if relu3.  precision as
 * pr_conv2d_nhwc (1: the three tensors hold bfloat16, the map between the convolutions is rounded to bfloat16). */
int pr_conv3x3_conv1x1_nhwc(int device, const void* x_dev, const float* w2_host, const float* b2_host,
                            const float* w3_host, const float* b3_host, const void* res_dev, void* y_dev,
                            int B, int H, int W, int Cin, int N3, int relu3, int precision, void* stream);

/* A whole layer1 Bottleneck (SPIN models/hmr.py Bottleneck.forward: conv1 1x1 -> conv2 3x3 64->64 -> conv3 1x1 64->256,
 * BatchNorm folded by the caller, + identity, ReLU) as ONE persistent bf16 kernel (csrc/bottleneck_bf16.hip): exported for
 * parity tests and timing (allocates, synchronises).  y_dev bf16 [B,H,W,256] (W <= 63), w2_host f32[64,64,3,3] OIHW,
 * w3_host f32[256,64], biases f32.
 *   wd_host == NULL: a block without a downsample branch -- x_dev bf16 [B,H,W,256], w1_host f32[64,256], identity = x;
 *   wd_host != NULL: the stage's first block -- x_dev bf16 [B,H,W,64], w1_host f32[64,64], identity = the downsample
 *   branch wd_host f32[256,64] (+ bd_host f32[256]) of x, summed into conv3's K loop.
 * repeats > 0 and ms_out != NULL: the mean time of `repeats` further launches in milliseconds. */
int pr_bottleneck_nhwc(int device, const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host,
                       const float* b2_host, const float* w3_host, const float* b3_host, const float* wd_host,
                       const float* bd_host, void* y_dev, int B, int H, int W, int repeats, float* ms_out, void* stream);

/* A whole layer2 Bottleneck (plain block, 512 -> 128 -> 128 -> 512 channels, W <= 31; SPIN models/hmr.py Bottleneck.forward) as
 * ONE persistent bf16 kernel (csrc/bottleneck128_bf16.hip): exported for parity tests and timing (allocates, synchronises).
 * x_dev, y_dev bf16 [B,H,W,512]; w1_host f32[128,512], w2_host f32[128,128,3,3] OIHW, w3_host f32[512,128], biases f32
 * (BatchNorm folded by the caller); identity = x.  repeats / ms_out as pr_bottleneck_nhwc. */
int pr_bottleneck128_nhwc(int device, const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host,
                          const float* b2_host, const float* w3_host, const float* b3_host, void* y_dev, int B, int H, int W,
                          int repeats, float* ms_out, void* stream);

/* A whole layer3 Bottleneck (plain block, 1024 -> 256 -> 256 -> 1024 channels, H W <= 224; SPIN models/hmr.py Bottleneck.forward)
 * as ONE bf16 kernel, one frame per workgroup (csrc/bottleneck256_bf16.hip): exported for parity tests and timing (allocates,
 * synchronises).  x_dev, y_dev bf16 [B,H,W,1024]; w1_host f32[256,1024], w2_host f32[256,256,3,3] OIHW, w3_host f32[1024,256],
 * biases f32 (BatchNorm folded by the caller); identity = x.  repeats / ms_out as pr_bottleneck_nhwc. */
int pr_bottleneck256_nhwc(int device, const void* x_dev, const float* w1_host, const float* b1_host, const float* w2_host,
                          const float* b2_host, const float* w3_host, const float* b3_host, void* y_dev, int B, int H, int W,
                          int repeats, float* ms_out, void* stream);

/* The bf16 encoder's stem as ONE kernel (csrc/stem_pool_bf16.hip): the 7x7 / stride-2 conv1 in its 4x4 / stride-1 form on
 * the 2x2 space-to-depth image (window rows y-2 .. y+1) + bias (folded bn1) + ReLU + MaxPool2d(3, 2, 1)  (SPIN models/hmr.py
 * conv1 / bn1 / relu / maxpool): exported for parity tests and timing (allocates, synchronises).
 * x_dev bf16 [B,H,H,16] (H even, <= 112), w_host f32[64,16,4,4] OIHW, bias_host f32[64] -> y_dev bf16 [B,H/2,H/2,64]. */
int pr_stem_pool_nhwc(int device, const void* x_dev, const float* w_host, const float* bias_host, void* y_dev, int B, int H,
                      int repeats, float* ms_out, void* stream);

/* ------------------------------------------------------------------------------------ */
/* f-1  crop front-end                                                  */
/* replaces: CropDataset.__getitem__ data/demo_dataset.py:58-74 ->                       */
/*           get_single_image_crop_demo lib/utils/_img_utils.py:219-252 (affine :53-101, */
/*           cv2.warpAffine INTER_LINEAR / BORDER_CONSTANT) -> ToTensor :259-266          */
/* ------------------------------------------------------------------------------------ */
/* frames_dev uint8[F,H,W,3] decoded video frames (bgr != 0: channel order of cv2.imread, swapped to RGB
 * like demo_dataset.py:59), bboxes_dev f32[N,4] (cx,cy,w,h) one per crop, frame_idx_dev int32[N] or NULL
 * (crop n comes from frame n), scale = cfg.DATASET.bbox_scale (1.2) -> crops_dev f32[N,3,224,224] in [0,1].
 * OpenCV's fixed-point bilinear warp is reproduced (integer weights, round-half-even), so crops are bit-exact
 * against the restated algorithm.  A frame index outside [0, F) never becomes an out-of-range read: that crop
 * is zero-filled and status_dev[n] (int32[N], may be NULL) is set to 1 (0 otherwise); the reference would raise
 * from cv2.imread on the missing file. */
int pr_crop_frames(const uint8_t* frames_dev, int F, int H, int W, int bgr, const int32_t* frame_idx_dev,
                   const float* bboxes_dev, int N, float scale, float* crops_dev, int32_t* status_dev,
                   void* stream);

/* ------------------------------------------------------------------------------------ */
/* a3-a5  rotation conversions                                                           */
/* ------------------------------------------------------------------------------------ */
/* SPIN utils/geometry.py rot6d_to_rotmat: pose6d_dev f32[N,144] -> rotmat_dev f32[N,24,3,3] */
int pr_rot6d_to_rotmat(const float* pose6d_dev, int N, float* rotmat_dev, void* stream);

/* replaces: rot_to_angle + axis_angle_to_euler_angle   lib/utils/coord_utils.py:24-30, 83-95
 * rotmat_dev f32[N,24,3,3] -> axis_angle_dev f32[N,24,3] (OpenCV Rodrigues semantics,
 * double arithmetic, float32 result) and euler_deg_dev f64[N,24,3] (x,y,z degrees).
 * status_dev int32[N] (may be NULL): bit0 = isRotationMatrix failed (coord_utils.py:70),
 * bit1 = Euler round-trip check failed (coord_utils.py:90-91); the reference aborts there. */
int pr_pose_to_euler(const float* rotmat_dev, int N, float* axis_angle_dev, double* euler_deg_dev,
                     int32_t* status_dev, void* stream);
/* replaces: axis_angle_to_euler_angle called on its own   lib/utils/coord_utils.py:83-95
 * axis_angle_dev f32[N,24,3] (read only) -> euler_deg_dev f64[N,24,3]; status bits as above. */
int pr_axis_angle_to_euler(const float* axis_angle_dev, int N, double* euler_deg_dev, int32_t* status_dev,
                           void* stream);

/* ------------------------------------------------------------------------------------ */
/* a6-a11  SMPL: Rodrigues, shape blend, joint regression, pose blend, chain, skinning   */
/* replaces: SMPL_Layer.forward  lib/smplpytorch/smplpytorch/pytorch/smpl_layer.py:65-158 */
/* ------------------------------------------------------------------------------------ */
typedef struct pr_smpl pr_smpl_t;

/* Host arrays as SMPL_Layer registers them (smpl_layer.py:40-63):
 *   v_template f32[V,3], shapedirs f32[V,3,NB], posedirs f32[V,3,(J-1)*9],
 *   J_regressor f32[J,V] (dense), weights f32[V,J], parents int32[J] (parents[0] < 0),
 *   model_betas f32[NB] or NULL (the pkl's own betas, used when the caller's betas are
 *   all zero: smpl_layer.py:87-91).  J must be 24, NB <= 16. */
int pr_smpl_create(int device, const float* v_template_host, const float* shapedirs_host,
                   const float* posedirs_host, const float* J_regressor_host,
                   const float* weights_host, const int32_t* parents_host,
                   const float* model_betas_host, int V, int J, int NB, int max_batch,
                   pr_smpl_t** out);
int pr_smpl_destroy(pr_smpl_t* h);

/* pose_dev f32[B,72] axis-angle, betas_dev f32[B,NB] or NULL, trans_dev f32[B,3] or NULL
 * -> verts_dev f32[B,V,3] (may be NULL: joints only), joints_dev f32[B,J,3]; metres.
 * center_idx < 0 = no centring (smpl_layer.py:148-152). */
int pr_smpl_forward(pr_smpl_t* h, const float* pose_dev, const float* betas_dev,
                    const float* trans_dev, int B, int center_idx, float* verts_dev,
                    float* joints_dev, void* stream);

/* replaces: get_joint_cam   lib/utils/coord_utils.py:7-21
 * axis_angle_dev f32[N,24,3] is MUTATED: every root row becomes (3.14, 0, 0) as in the
 * reference; joint_cam_dev f32[N,24,3] = joints(mm) - root joint, zero betas.
 * verts_dev (optional) f32[N,V,3] receives the mesh the reference computes and drops. */
int pr_smpl_joint_cam(pr_smpl_t* h, float* axis_angle_dev, int N, float* joint_cam_dev,
                      float* verts_dev, void* stream);

/* ------------------------------------------------------------------------------------ */
/* a12-a13  REBA / RULA                                                                  */
/* replaces: REBA.__call__ lib/utils/reba.py:50-81, RULA.__call__ lib/utils/rula.py:66-98 */
/* ------------------------------------------------------------------------------------ */
typedef struct pr_reba_info { /* add_info["REBA"], example/additional_information.json:2-11 */
  int32_t legs_bilateral;     /* "Legs_bilateral_weight_bearing/walking" */
  int32_t sitting;            /* "Sitting" */
  int32_t load_force;         /* "Load/Force Score" */
  int32_t arm_supported_l;    /* "Arm_supported_leaning_L" */
  int32_t arm_supported_r;    /* "Arm_supported_leaning_R" */
  int32_t coupling;           /* "Coupling" */
  int32_t activity;           /* "Activity_Score" */
} pr_reba_info;

typedef struct pr_rula_info { /* add_info["RULA"], example/additional_information.json:13-24 */
  int32_t arm_supported_l, arm_supported_r;
  int32_t a_muscle_l, a_muscle_r;
  int32_t a_load_l, a_load_r;
  int32_t legs_bilateral;
  int32_t b_muscle, b_load;
} pr_rula_info;

/* euler_deg_dev f64[N,24,3] -> out_dev int32[N,10]:
 *   score, trunk, neck, leg, upper_arm L,R, lower_arm L,R, wrist L,R  (reba.py:71-75 log_score) */
int pr_reba(const double* euler_deg_dev, int N, const pr_reba_info* info, int32_t* out_dev,
            void* stream);
/* -> out_dev int32[N,12]:
 *   score, upper_arm L,R, lower_arm L,R, wrist L,R, wrist_twist L,R, neck, trunk, leg */
int pr_rula(const double* euler_deg_dev, int N, const pr_rula_info* info, int32_t* out_dev,
            void* stream);

/* ------------------------------------------------------------------------------------ */
/* a15  the per-batch driver: crops -> everything the reference's loop produces          */
/* replaces: Predictor.get_pose_estimation_results lib/core/base.py:211-240 + :151,168   */
/* ------------------------------------------------------------------------------------ */
typedef struct pr_frames_out { /* all device pointers; any may be NULL except where noted */
  float* rotmat;        /* f32[B,24,3,3] (required) */
  float* betas;         /* f32[B,10] */
  float* cam;           /* f32[B,3]  */
  float* axis_angle;    /* f32[B,24,3] (required)  root rows overwritten with 3.14,0,0 (Q5) */
  double* euler_deg;    /* f64[B,24,3] (required) */
  float* joint_cam;     /* f32[B,24,3] (required) */
  float* verts;         /* f32[B,V,3]  optional mesh */
  int32_t* reba;        /* int32[B,10] */
  int32_t* rula;        /* int32[B,12] */
  int32_t* status;      /* int32[B]    */
} pr_frames_out;

int pr_frames_forward(pr_hmr_t* hmr, pr_smpl_t* smpl, const float* x_dev, int B,
                      const pr_reba_info* reba_info, const pr_rula_info* rula_info,
                      const pr_frames_out* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POSERISK_HIP_H */
