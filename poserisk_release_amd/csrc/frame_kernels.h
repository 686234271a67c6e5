// Launchers of the per-frame non-GEMM kernels (frame_kernels.hip).
#pragma once
#include "common.h"
#include "host_plan.h"

namespace pr {

// kStateStride (the regressor state row: pose6d 144 | betas 10 | cam 3 | zero pad to 192): host_plan.h

int launch_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, hipStream_t s);
int launch_nchw3_to_s2d12(const float* x, float* y, int B, int H, int W, hipStream_t s);
int launch_nchw3_to_s2d16_bf16(const float* x, void* y, int B, int H, int W, hipStream_t s);
int launch_maxpool(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);
int launch_avgpool(const float* x, float* y, int B, int HW, int C, hipStream_t s);
// bf16 encoder plumbing (precision = 1): bf16 buffers are passed as void*
int launch_nchw3_to_nhwc8_bf16(const float* x, void* y, int B, int H, int W, hipStream_t s);
int launch_maxpool_bf16(const void* x, void* y, int B, int H, int W, int C, hipStream_t s);
int launch_avgpool_bf16(const void* x, float* y, int B, int HW, int C, hipStream_t s);
int launch_f32_to_bf16(const float* x, void* y, long n, hipStream_t s);
int launch_bf16_to_f32(const void* x, float* y, long n, hipStream_t s);
int launch_crop_frames(const unsigned char* frames, int F, int H, int W, int bgr, const int* frame_idx,
                       const float* bboxes, int N, float scale, float* crops, int* status, hipStream_t s);
int launch_state_init(const float* init157, float* state, int B, hipStream_t s);
int launch_regressor_finalize(const float* state, float* rotmat, float* betas, float* cam, float* pose6d,
                              int B, hipStream_t s);
// fully connected layer of the regressor (fc_regressor.hip): y[M][N] = x[M][K] W[N][K]^T + bias + res (res may alias y)
// shape: 0 = chosen from M; else 10 MT + NT output tiles of 16x16 per workgroup (same bits for every shape)
int launch_fc_rows16(const float* x, const float* w, const float* bias, const float* res, float* y, int M, int N, int K,
                     hipStream_t s, int shape = 0);
int launch_rot6d(const float* pose6d, float* rotmat, long n_joints, hipStream_t s);
int launch_pose_to_euler(const float* rotmat, int N, float* axis_angle, double* euler, int32_t* status,
                         hipStream_t s);
int launch_reba(const double* euler, int N, const pr_reba_info& info, int32_t* out, hipStream_t s);
int launch_rula(const double* euler, int N, const pr_rula_info& info, int32_t* out, hipStream_t s);

}  // namespace pr
