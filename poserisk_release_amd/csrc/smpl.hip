// pr_smpl: SMPL forward (axis-angle -> Rodrigues, shape blend, joint regression, pose blend,
// kinematic chain, linear-blend skinning) on gfx950.
// Replaces SMPL_Layer.forward (lib/smplpytorch/smplpytorch/pytorch/smpl_layer.py:65-158) and
// get_joint_cam (lib/utils/coord_utils.py:7-21).
//
// HBM layout (all float32, resident in the handle):
//   posedirs_T [207][R]   R = 3V rows (row = 3*vertex + component), padded to a multiple of 64
//   shapedirs_T[NB][R], v_template[R]
//   ell_idx/ell_w [NNZ][V]  skinning weights, nonzeros only, ascending joint order
//   J_template[24][3], J_dirs[24][3][NB]  = J_regressor applied to v_template / shapedirs (double)
// Per call workspaces: A[B][24][12] skinning transforms, pose_map_T[207][Bs], betas_T[NB][Bs]
// (frame index fastest so that one wave's 16 frames are one scalar-load of 64 bytes).
//
// Kernels:
//   smpl_flags   : the reference's two host-synchronising tests (norm(betas)==0, norm(trans)==0)
//                  evaluated on device instead (smpl_layer.py:87,148).
//   smpl_pose    : one wave per frame; lanes = joints.  Rodrigues via half-angle quaternion with
//                  the reference's norm(v+1e-8) quirk, rest joints, level-synchronous kinematic
//                  chain in LDS, A_i = G_i - pack(G_i [j_i;0]).
//   smpl_skin_tile: handles up to 128 frames per call, SMPL's own shape (<= 4 weights per vertex, <= 10 shape
//                  coefficients).  252 rows x 16 frames per workgroup of 8 waves; pose map, transforms, shape
//                  coefficients and offsets staged in LDS by LDS-DMA, coefficients read back as broadcasts,
//                  two rows x 16 frames of accumulators per lane.  Same bits as smpl_skin.
//   smpl_skin    : the same arithmetic for any weight density.  A workgroup owns 63 rows (21 vertices x 3 components)
//                  x 16 frames, its 4 waves each take a quarter of the 207 coefficients;
//                  the model rows are read once per workgroup with coalesced dword loads,
//                  the per-frame coefficients come through the scalar cache (wave-uniform), the
//                  16 frames' transforms are staged in LDS and gathered per nonzero weight,
//                  x/y/z of a vertex are exchanged with wavefront shuffles.  No MFMA.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <memory>
#include <vector>

#include "common.h"

namespace pr {
namespace {

constexpr int kJ = 24;
constexpr int kFB = 16;        // frames per wave in the skinning kernel
constexpr int kRowsPerWave = 63;  // 21 vertices x 3 components
constexpr int kMaxNB = 16;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct Tree {  // host copy; the kernels read parent/depth from a 48-int device array
  int parent[kJ];
  int depth[kJ];
  int max_depth;
};

__global__ void smpl_flags(const float* __restrict__ betas, long nb, const float* __restrict__ trans,
                           long nt, int* __restrict__ flags) {
  __shared__ int any_b, any_t;
  if (threadIdx.x == 0) {
    any_b = 0;
    any_t = 0;
  }
  __syncthreads();
  int lb = 0, lt = 0;
  if (betas)
    for (long i = threadIdx.x; i < nb; i += blockDim.x) lb |= (betas[i] != 0.f);
  if (trans)
    for (long i = threadIdx.x; i < nt; i += blockDim.x) lt |= (trans[i] != 0.f);
  if (lb) atomicOr(&any_b, 1);
  if (lt) atomicOr(&any_t, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    flags[0] = any_b ? 0 : 1;  // 1 = betas missing or all zero -> use the model's own betas
    flags[1] = any_t ? 0 : 1;  // 1 = trans missing or all zero
  }
}

struct PoseArgs {
  float* pose;              // [B,72] (root row rewritten when overwrite_root)
  const float* betas;       // [B,NB] or null
  const float* trans;       // [B,3] or null
  const int* flags;         // device flags from smpl_flags, or null (=> use model betas, no trans)
  const float* J_template;  // [24,3]
  const float* J_dirs;      // [24,3,NB]
  const float* model_betas; // [NB]
  float* A;                 // [B,24,12]
  float* pm_T;              // [207][Bs]
  float* betas_T;           // [NB][Bs]
  float* voff;              // [B,3] offset added to the vertices
  float* joints;            // [B,24,3]
  int B, Bs, NB, b0;        // b0: frame offset of this chunk inside pm_T/betas_T/A (always 0 here)
  int overwrite_root, joint_cam_mode, center_idx;
  const int* tree;  // device: parent[24], depth[24]
  int max_depth;
};

__global__ __launch_bounds__(64) void smpl_pose(const PoseArgs a) {
  __shared__ float Rl[kJ][9];
  __shared__ float Jr[kJ][3];
  __shared__ float G[kJ][12];
  __shared__ float beta[kMaxNB];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int my_parent = lane < kJ ? a.tree[lane] : 0;
  const int my_depth = lane < kJ ? a.tree[kJ + lane] : -1;
  const bool use_model = a.flags ? (a.flags[0] != 0) : true;
  const bool trans_zero = a.flags ? (a.flags[1] != 0) : true;

  if (lane < kJ) {
    float* pv = a.pose + (long)b * 72 + lane * 3;
    float vx = pv[0], vy = pv[1], vz = pv[2];
    if (a.overwrite_root && lane == 0) {  // coord_utils.py:10,13  (3.14, not pi)
      vx = 3.14f; vy = 0.f; vz = 0.f;
      pv[0] = vx; pv[1] = vy; pv[2] = vz;
    }
    // rodrigues_layer.py:41-52 batch_rodrigues + :13-38 quat2mat
    const float ex = vx + 1e-8f, ey = vy + 1e-8f, ez = vz + 1e-8f;
    const float n = sqrtf(ex * ex + ey * ey + ez * ez);
    const float ax = vx / n, ay = vy / n, az = vz / n;
    const float half = n * 0.5f;
    const float cs = cosf(half), sn = sinf(half);
    float qw = cs, qx = sn * ax, qy = sn * ay, qz = sn * az;
    const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    const float w2 = qw * qw, x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
    const float wx = qw * qx, wy = qw * qy, wz = qw * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
    float R[9];
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;     R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;     R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rl[lane][e] = R[e];
    if (lane > 0) {  // tensutils.py:41-48 subtract_flat_id
#pragma unroll
      for (int e = 0; e < 9; ++e)
        a.pm_T[(long)((lane - 1) * 9 + e) * a.Bs + b] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
  }
  if (lane < a.NB) {
    const float be = use_model ? a.model_betas[lane] : a.betas[(long)b * a.NB + lane];
    beta[lane] = be;
    a.betas_T[(long)lane * a.Bs + b] = be;
  }
  __syncthreads();
  for (int i = lane; i < kJ * 3; i += 64) {  // smpl_layer.py:88-95 joint regression (pre-contracted)
    float s = a.J_template[i];
    for (int l = 0; l < a.NB; ++l) s += a.J_dirs[i * a.NB + l] * beta[l];
    Jr[i / 3][i % 3] = s;
  }
  __syncthreads();
  // smpl_layer.py:102-119 kinematic chain, one tree level per step
  for (int d = 0; d <= a.max_depth; ++d) {
    if (my_depth == d) {
      const float* R = Rl[lane];
      if (d == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          G[lane][r * 4 + 0] = R[r * 3 + 0];
          G[lane][r * 4 + 1] = R[r * 3 + 1];
          G[lane][r * 4 + 2] = R[r * 3 + 2];
          G[lane][r * 4 + 3] = Jr[lane][r];
        }
      } else {
        const int p = my_parent;
        const float tx = Jr[lane][0] - Jr[p][0], ty = Jr[lane][1] - Jr[p][1], tz = Jr[lane][2] - Jr[p][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float g0 = G[p][r * 4 + 0], g1 = G[p][r * 4 + 1], g2 = G[p][r * 4 + 2], g3 = G[p][r * 4 + 3];
          G[lane][r * 4 + 0] = g0 * R[0] + g1 * R[3] + g2 * R[6];
          G[lane][r * 4 + 1] = g0 * R[1] + g1 * R[4] + g2 * R[7];
          G[lane][r * 4 + 2] = g0 * R[2] + g1 * R[5] + g2 * R[8];
          G[lane][r * 4 + 3] = g0 * tx + g1 * ty + g2 * tz + g3;
        }
      }
    }
    __syncthreads();
  }
  // offsets (smpl_layer.py:147-155): trans given and non-zero -> +trans; else centre on center_idx
  float ox = 0.f, oy = 0.f, oz = 0.f;
  if (a.trans && !trans_zero) {
    ox = a.trans[(long)b * 3 + 0]; oy = a.trans[(long)b * 3 + 1]; oz = a.trans[(long)b * 3 + 2];
  } else if (a.center_idx >= 0) {
    ox = -G[a.center_idx][3]; oy = -G[a.center_idx][7]; oz = -G[a.center_idx][11];
  }
  if (lane == 0 && a.voff) {
    a.voff[(long)b * 3 + 0] = ox; a.voff[(long)b * 3 + 1] = oy; a.voff[(long)b * 3 + 2] = oz;
  }
  if (lane < kJ) {
    // smpl_layer.py:122-132  A = G - pack(G [j;0])
    float* Ao = a.A + ((long)b * kJ + lane) * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float g0 = G[lane][r * 4 + 0], g1 = G[lane][r * 4 + 1], g2 = G[lane][r * 4 + 2];
      Ao[r * 4 + 0] = g0; Ao[r * 4 + 1] = g1; Ao[r * 4 + 2] = g2;
      Ao[r * 4 + 3] = G[lane][r * 4 + 3] - (g0 * Jr[lane][0] + g1 * Jr[lane][1] + g2 * Jr[lane][2]);
    }
    if (a.joints) {
      float* jo = a.joints + ((long)b * kJ + lane) * 3;
      if (a.joint_cam_mode) {  // coord_utils.py:16-17: (joints * 1000) - root
        jo[0] = G[lane][3] * 1000.f - G[0][3] * 1000.f;
        jo[1] = G[lane][7] * 1000.f - G[0][7] * 1000.f;
        jo[2] = G[lane][11] * 1000.f - G[0][11] * 1000.f;
      } else {
        jo[0] = G[lane][3] + ox; jo[1] = G[lane][7] + oy; jo[2] = G[lane][11] + oz;
      }
    }
  }
}

struct SkinArgs {
  const float* posedirs_T;   // [NP][R]
  const float* shapedirs_T;  // [NB][R]
  const float* v_template;   // [R]
  const int* ell_idx;        // [NNZ][V]
  const float* ell_w;        // [NNZ][V]
  const float* A;            // [Bs][24][12]
  const float* pm_T;         // [NP][Bs]
  const float* betas_T;      // [NB][Bs]
  const float* voff;         // [Bs][3]
  float* verts;              // [B][V][3]
  int V, R, NP, NPpad, NB, NNZ, B, Bs;
  int n_rt, n_fg;  // row tiles (padded to a multiple of 8), frame groups
  unsigned long long* stamps;  // diagnosis only (scripts/micro/t_smpl_tile.hip): 8 clock stamps per workgroup, or null
};

// 1-D grid -> (row tile, frame group).  Workgroups b and b+8 share an XCD, so the frame groups of one
// row tile are placed 8 ids apart: they re-read that tile's model rows from the same L2 instead of
// fetching them once per frame group (measured: 98 MB of L2 misses per launch at B=64 for 24.7 MB
// of algorithmic bytes before this mapping).  Placement is a speed matter only.
__device__ inline void skin_block_map(const SkinArgs& a, int& rt, int& fg) {
  const int id = blockIdx.x, lo = id & 7, k = id >> 3;
  fg = k % a.n_fg;
  rt = (k / a.n_fg) * 8 + lo;
}

template <int NNZ_MAX>
__global__ __launch_bounds__(256) void smpl_skin(const SkinArgs a) {
  // One workgroup = 63 rows (21 vertices x xyz) x 16 frames.  Its four waves split the 207 pose-blend
  // coefficients (52 each, zero-padded to 208) so that the dependent load->FMA chain per wave is four
  // times shorter and the grid has four times more waves in flight (this kernel is latency-bound at
  // small B: a wave alone needs ~26 dependent HBM round trips); partial sums meet in LDS and are added
  // in wave order, then wave w finishes frames 4w..4w+3 (shape blend, skinning, store).
  __shared__ __attribute__((aligned(16))) float As[kFB * kJ * 12];
  __shared__ float Off[kFB * 3];
  __shared__ float Red[4][kFB][64];
  int rt, fg;
  skin_block_map(a, rt, fg);
  const int fb0 = fg * kFB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane % 3;
  const int v = rt * 21 + lane / 3;
  const bool active = lane < kRowsPerWave && v < a.V;
  const int row = active ? v * 3 + c : 0;

  // Everything that does not depend on anything else is requested first, so that one memory round trip
  // covers it all (the kernel is latency-bound: 336 workgroups took 14 us with these loads issued one
  // group after another, ~6 dependent HBM round trips).
  constexpr int ASN = kFB * kJ * 12 / 256;  // 18 staged floats per thread
  float a_stage[ASN];
  {
    const float* src = a.A + (long)fb0 * kJ * 12 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < ASN; ++i) a_stage[i] = src[i * 256];
  }
  const float off_v = threadIdx.x < kFB * 3 ? a.voff[(long)fb0 * 3 + threadIdx.x] : 0.f;
  float sdv[kMaxNB];
#pragma unroll
  for (int l = 0; l < kMaxNB; ++l) sdv[l] = l < a.NB ? a.shapedirs_T[(long)l * a.R + row] : 0.f;
  const float vtmp = a.v_template[row];
  int jidx[NNZ_MAX];
  float jw[NNZ_MAX];
#pragma unroll
  for (int k = 0; k < NNZ_MAX; ++k) {
    const bool ok = active && k < a.NNZ;
    jidx[k] = ok ? a.ell_idx[(long)k * a.V + v] : 0;
    jw[k] = ok ? a.ell_w[(long)k * a.V + v] : 0.f;
  }

  // pose blend (smpl_layer.py:97-99): this wave's quarter of the coefficients, all 16 frames
  const int pq = a.NPpad / 4;  // 52
  const float* __restrict__ pT = a.pm_T + (long)wave * pq * a.Bs + fb0;
  const float* __restrict__ pd = a.posedirs_T + (long)wave * pq * a.R + row;
  float acc[kFB];
#pragma unroll
  for (int f = 0; f < kFB; ++f) acc[f] = 0.f;
  // 52 coefficients per wave in 4 batches of 13, the next batch's 13 model values requested before the
  // current batch is consumed: at B=64 there are only ~5 workgroups per CU, so memory latency must be
  // covered by loads in flight per wave, not by occupancy (prefetching 4 values ahead left 13 exposed
  // round trips per wave and 32 us per launch).
  constexpr int PB = 13;
  float nxt[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) nxt[j] = pd[(long)j * a.R];
  for (int p0 = 0; p0 < pq; p0 += PB) {
    float cur[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) cur[j] = nxt[j];
    if (p0 + PB < pq) {
#pragma unroll
      for (int j = 0; j < PB; ++j) nxt[j] = pd[(long)(p0 + PB + j) * a.R];
    }
    // fused multiply-adds are spelled out so that all 16 frame slots of a lane get the same
    // instruction selection: results are bit-identical wherever a frame lands in a batch
#pragma unroll
    for (int j = 0; j < PB; ++j)
#pragma unroll
      for (int f = 0; f < kFB; ++f) acc[f] = __builtin_fmaf(cur[j], pT[(long)(p0 + j) * a.Bs + f], acc[f]);
  }
#pragma unroll
  for (int f = 0; f < kFB; ++f) Red[wave][f][lane] = acc[f];
  // the 16 frames' transforms (contiguous in A) and vertex offsets go to LDS
#pragma unroll
  for (int i = 0; i < ASN; ++i) As[threadIdx.x + i * 256] = a_stage[i];
  if (threadIdx.x < kFB * 3) Off[threadIdx.x] = off_v;

  // this wave's 4 frames: shape blend (smpl_layer.py:88-95), summed on its own like the reference
  const int f0 = wave * 4;
  const float* __restrict__ bT = a.betas_T + fb0 + f0;
  float sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < kMaxNB; ++l) {
    if (l < a.NB) {
#pragma unroll
      for (int f = 0; f < 4; ++f) sb[f] = __builtin_fmaf(sdv[l], bT[(long)l * a.Bs + f], sb[f]);
    }
  }
  __syncthreads();

  const int l0 = lane - c;
#pragma unroll
  for (int ff = 0; ff < 4; ++ff) {
    const int f = f0 + ff;
    // (v_template + S) + P in the reference's order; P = partial sums in coefficient order
    const float P = ((Red[0][f][lane] + Red[1][f][lane]) + Red[2][f][lane]) + Red[3][f][lane];
    const float vp = (vtmp + sb[ff]) + P;
    const float x = __shfl(vp, l0, 64), y = __shfl(vp, l0 + 1, 64), z = __shfl(vp, l0 + 2, 64);
    // smpl_layer.py:134 T = A . W^T (row c of the blended transform), nonzero weights only
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NNZ_MAX; ++k) {
      if (k < a.NNZ) {
        const f32x4 ar = *reinterpret_cast<const f32x4*>(&As[(f * kJ + jidx[k]) * 12 + c * 4]);
        t[0] = __builtin_fmaf(jw[k], ar[0], t[0]);
        t[1] = __builtin_fmaf(jw[k], ar[1], t[1]);
        t[2] = __builtin_fmaf(jw[k], ar[2], t[2]);
        t[3] = __builtin_fmaf(jw[k], ar[3], t[3]);
      }
    }
    // smpl_layer.py:143 (T * [v;1]).sum over the 4 columns, then the centring / translation offset
    const float o = __builtin_fmaf(t[2], z, __builtin_fmaf(t[1], y, t[0] * x)) + t[3] + Off[f * 3 + c];
    if (active && fb0 + f < a.B) a.verts[((long)(fb0 + f) * a.V + v) * 3 + c] = o;
  }
}

// Large-batch variant: a WAVE owns 63 rows x 16 frames and walks all 207 coefficients itself (no LDS
// reduction, a quarter of the workgroups); with thousands of waves in flight latency is hidden by
// occupancy and this form does less work per frame.  Used for B > 128.
template <int NNZ_MAX>
__global__ __launch_bounds__(256) void smpl_skin_rows(const SkinArgs a) {
  __shared__ __attribute__((aligned(16))) float As[kFB * kJ * 12];
  __shared__ float Off[kFB * 3];
  int rt, fg;
  skin_block_map(a, rt, fg);
  const int fb0 = fg * kFB;
  {  // stage the 16 frames' transforms (contiguous in A) and vertex offsets
    const float* src = a.A + (long)fb0 * kJ * 12;
    for (int i = threadIdx.x; i < kFB * kJ * 12; i += 256) As[i] = src[i];
    if (threadIdx.x < kFB * 3) Off[threadIdx.x] = a.voff[(long)fb0 * 3 + threadIdx.x];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vt = rt * 4 + wave;
  const int c = lane % 3;
  const int v = vt * 21 + lane / 3;
  const bool active = lane < kRowsPerWave && v < a.V;
  const int row = active ? v * 3 + c : 0;

  // shape blend (smpl_layer.py:88-95), then pose blend (:97-99), each summed on its own and
  // added in the reference's order: (v_template + S) + P.  Fused multiply-adds are spelled out so
  // that all 16 frame slots of a lane get the same instruction selection (bit-identical results
  // wherever a frame lands in a batch).
  const float* __restrict__ bT = a.betas_T + fb0;
  const float* __restrict__ pT = a.pm_T + fb0;
  float acc[kFB];
#pragma unroll
  for (int f = 0; f < kFB; ++f) acc[f] = 0.f;
  for (int l = 0; l < a.NB; ++l) {
    const float sd = a.shapedirs_T[(long)l * a.R + row];
#pragma unroll
    for (int f = 0; f < kFB; ++f) acc[f] = __builtin_fmaf(sd, bT[(long)l * a.Bs + f], acc[f]);
  }
  const float vtmp = a.v_template[row];
  float vs[kFB];
#pragma unroll
  for (int f = 0; f < kFB; ++f) {
    vs[f] = vtmp + acc[f];
    acc[f] = 0.f;
  }
#pragma unroll 4
  for (int p = 0; p < a.NP; ++p) {
    const float pd = a.posedirs_T[(long)p * a.R + row];
#pragma unroll
    for (int f = 0; f < kFB; ++f) acc[f] = __builtin_fmaf(pd, pT[(long)p * a.Bs + f], acc[f]);
  }

  int jidx[NNZ_MAX];
  float jw[NNZ_MAX];
#pragma unroll
  for (int k = 0; k < NNZ_MAX; ++k) {
    const bool ok = active && k < a.NNZ;
    jidx[k] = ok ? a.ell_idx[(long)k * a.V + v] : 0;
    jw[k] = ok ? a.ell_w[(long)k * a.V + v] : 0.f;
  }
  const int l0 = lane - c;
#pragma unroll
  for (int f = 0; f < kFB; ++f) {
    const float vp = vs[f] + acc[f];
    const float x = __shfl(vp, l0, 64), y = __shfl(vp, l0 + 1, 64), z = __shfl(vp, l0 + 2, 64);
    // smpl_layer.py:134 T = A . W^T (row c of the blended transform), nonzero weights only
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NNZ_MAX; ++k) {
      if (k < a.NNZ) {
        const f32x4 ar = *reinterpret_cast<const f32x4*>(&As[(f * kJ + jidx[k]) * 12 + c * 4]);
        t[0] = __builtin_fmaf(jw[k], ar[0], t[0]);
        t[1] = __builtin_fmaf(jw[k], ar[1], t[1]);
        t[2] = __builtin_fmaf(jw[k], ar[2], t[2]);
        t[3] = __builtin_fmaf(jw[k], ar[3], t[3]);
      }
    }
    // smpl_layer.py:143 (T * [v;1]).sum over the 4 columns, then the centring / translation offset
    // explicit fused ops: every frame slot must round identically (frames are sharded across GPUs)
    const float o = __builtin_fmaf(t[2], z, __builtin_fmaf(t[1], y, t[0] * x)) + t[3] + Off[f * 3 + c];
    if (active && fb0 + f < a.B) a.verts[((long)(fb0 + f) * a.V + v) * 3 + c] = o;
  }
}

// Register-tiled variant, 8 waves: a workgroup owns 252 rows (four row sets of 63: 84 vertices) x 16 frames.  Wave
// (q, h) takes quarter q of the 207 coefficients for row sets 2h and 2h+1: a lane holds two rows x 16 frames = 32
// accumulators as 16 v_pk_fma_f32 pairs, one model value feeds 16 multiply-adds, one coefficient two.  The quarter of
// the pose map (52 coefficients x 16 frames = 3.3 KB) is staged in LDS once and read back as wave-wide broadcasts (one
// ds_read_b128 = 4 frames' coefficients, identical addresses: 4 LDS cycles), which takes the per-iteration scalar
// loads -- every one of them a scalar-cache miss in smpl_skin -- off the dependent chain.  Two waves per SIMD come from
// the same workgroup (a single wave issues v_pk_fma_f32 at half rate; measured with scripts/micro/t_smpl_tile.hip).
// Wave (q, h) walks the frames in the order 4q, 4q+1, ... (mod 16): the four it finishes itself are then always its
// first four accumulators and only the other twelve go through LDS.
// The arithmetic per (row, frame) is smpl_skin's, operation for operation (quarter sums in coefficient order, quarters
// added in wave order, (v_template + S) + P, skinning with explicit fused multiply-adds): the two produce the same bits.
constexpr int kRS = 4;                                 // row sets per workgroup
constexpr int kTileThreads = 512;
constexpr int kTileNB = 10;                            // shape coefficients held in registers (SMPL has 10)
constexpr int kTilePM = 208 * kFB;                     // floats: the pose-map quarters [208][16]
constexpr int kTileAs = kFB * kJ * 12;                 // floats: the 16 frames' transforms
constexpr int kTileRed = 2 * 4 * 3 * 4 * 2 * 64;       // floats: [half][destination quarter][source slot][frame][row set][lane]
constexpr size_t kTileLds = (size_t)(kTilePM + kTileAs + kTileRed + kFB * 3 + kTileNB * kFB) * 4;  // 81 728 B: two workgroups per CU

template <int NNZ_MAX>
__global__ __launch_bounds__(kTileThreads, 4) void smpl_skin_tile(const SkinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
#define PR_STAMP(i) do { if (a.stamps && threadIdx.x == 0) a.stamps[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  PR_STAMP(0);
  float* PM = tile_lds;
  float* As = tile_lds + kTilePM;
  float* Red = As + kTileAs;
  float* Off = Red + kTileRed;
  float* Bt = Off + kFB * 3;
  int rt, fg;
  skin_block_map(a, rt, fg);
  const int fb0 = fg * kFB;
  const int lane = threadIdx.x & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = w8 & 3, hrs = w8 >> 2;   // coefficient quarter, row-set half
  const int c = lane % 3;
  const int pq = a.NPpad / 4;  // 52
  int v[2], row[2];
  bool active[2];
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
    v[rs] = (rt * kRS + 2 * hrs + rs) * 21 + lane / 3;
    active[rs] = lane < kRowsPerWave && v[rs] < a.V;
    row[rs] = active[rs] ? v[rs] * 3 + c : 0;
  }
  // Everything the workgroup shares goes from global memory straight into LDS (16 bytes per lane, 1 KB per wave
  // instruction, no registers and nothing to wait for here): quarter q of the pose map (16 frames: 52 rows of 64 bytes =
  // 208 pieces dealt to the two waves that read it), the 16 frames' transforms (18 KB, contiguous in A), shape
  // coefficients and vertex offsets.
  {
    typedef __attribute__((address_space(3))) void lds_void;
    char* lds = reinterpret_cast<char*>(tile_lds);
    const auto pmsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pm_T), 0, (int)((long)a.NPpad * a.Bs * 4), 0x00020000);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int uu = 2 * hrs + u, i = lane + 64 * uu;
      if (i < pq * 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pmsrc, (lds_void*)(lds + ((q * pq * kFB) * 4 + uu * 1024)), 16,
                                                 (unsigned)((((q * pq + (i >> 2)) * a.Bs) + fb0 + (i & 3) * 4) * 4), 0, 0, 0);
    }
    const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)((long)a.Bs * kJ * 12 * 4), 0x00020000);
    for (int pc = w8; pc < kTileAs / 256; pc += 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(asrc, (lds_void*)(lds + (kTilePM + pc * 256) * 4), 16,
                                               (unsigned)((fb0 * kJ * 12 + pc * 256) * 4 + lane * 16), 0, 0, 0);
    if (w8 == 0 && lane < kTileNB * 4) {
      const auto bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.betas_T), 0, (int)((long)kMaxNB * a.Bs * 4), 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bsrc, (lds_void*)(lds + (kTilePM + kTileAs + kTileRed + kFB * 3) * 4), 16,
                                               (unsigned)((((lane >> 2) * a.Bs) + fb0 + (lane & 3) * 4) * 4), 0, 0, 0);
    }
    if (w8 == 1 && lane < kFB * 3 / 4) {
      const auto osrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.voff), 0, (int)((long)a.Bs * 3 * 4), 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(osrc, (lds_void*)(lds + (kTilePM + kTileAs + kTileRed) * 4), 16,
                                               (unsigned)(fb0 * 3 * 4 + lane * 16), 0, 0, 0);
    }
  }
  // Quarter q of the model rows as a buffer.  The reloads of the last 13 coefficients lie past it: they get the
  // out-of-range sentinel as their VECTOR offset (zero, no memory access) instead of relying on the scalar offset,
  // which carries the coefficient, being range-checked (gfx950 does check it: scripts/micro/t_soffset.hip; LLVM's
  // description of the intrinsic says it need not be).
  const auto pdsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.posedirs_T + (long)q * pq * a.R), 0,
                                                       (int)((long)pq * a.R * 4), 0x00020000);
  const unsigned roff0 = (unsigned)row[0] * 4u, roff1 = (unsigned)row[1] * 4u;
  auto model = [&](int k) {
    const unsigned so = (unsigned)k * (unsigned)a.R * 4u;
    const bool in = k < pq;      // wave-uniform
    return f32x2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pdsrc, in ? roff0 : 0x80000000u, in ? so : 0u, 0)),
                 __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pdsrc, in ? roff1 : 0x80000000u, in ? so : 0u, 0))};
  };
  // One v_pk_fma_f32 multiplies the two rows' model values by one coefficient (the same element of the coefficient pair
  // for both halves).  acc[i] belongs to frame (4q + i) mod 16.
  f32x2 acc[kFB];
#pragma unroll
  for (int f = 0; f < kFB; ++f) acc[f] = f32x2{0.f, 0.f};
  // 52 coefficients per wave; the model values of 13 are in flight at any time: a slot is reloaded (13 coefficients
  // ahead) as soon as its value has been used, so the prefetch costs no second set of registers
  constexpr int PB = 13;
  f32x2 md[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    md[j] = model(j);
    __builtin_amdgcn_sched_barrier(0);   // issue order = the loop's reload order: its vmcnt waits are then exact
  }
  // the finishing pass's model values are requested now and arrive under the loop
  const int f0 = q * 4;
  float sdv[2][kTileNB], vtmp[2];
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
#pragma unroll
    for (int l = 0; l < kTileNB; ++l) sdv[rs][l] = l < a.NB ? a.shapedirs_T[(long)l * a.R + row[rs]] : 0.f;
    vtmp[rs] = a.v_template[row[rs]];
  }
  PR_STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS fills above have landed (the compiler does not track them)
  __syncthreads();   // the staged coefficients, transforms and offsets are visible
  PR_STAMP(2);
  // (four waves per SIMD are resident: the LDS latency of a step's coefficients is covered by the other waves)
  const f32x4* c4 = reinterpret_cast<const f32x4*>(&PM[q * pq * kFB]);
  int cq[kFB / 4];       // float4 index of the g-th group of four frames in this wave's order
#pragma unroll
  for (int g = 0; g < kFB / 4; ++g) cq[g] = (q + g) & 3;
#pragma unroll 1
  for (int p0 = 0; p0 < pq; p0 += PB) {
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      f32x4 cf[kFB / 4];
#pragma unroll
      for (int g = 0; g < kFB / 4; ++g) cf[g] = c4[(p0 + j) * (kFB / 4) + cq[g]];
#pragma unroll
      for (int g = 0; g < kFB / 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[g * 4 + e] = __builtin_elementwise_fma(md[j], f32x2{cf[g][e], cf[g][e]}, acc[g * 4 + e]);
      md[j] = model(p0 + PB + j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  PR_STAMP(3);
  // skinning weights: requested here, used after the barrier
  float jw[2][NNZ_MAX];
  int jidx[2][NNZ_MAX];
#pragma unroll
  for (int rs = 0; rs < 2; ++rs)
#pragma unroll
    for (int k = 0; k < NNZ_MAX; ++k) {
      const bool ok = active[rs] && k < a.NNZ;
      jidx[rs][k] = ok ? a.ell_idx[(long)k * a.V + v[rs]] : 0;
      jw[rs][k] = ok ? a.ell_w[(long)k * a.V + v[rs]] : 0.f;
    }
  // partial sums of the twelve frames other quarters finish: group g (frames 4(q+g) .. +3) goes to quarter (q+g) mod 4,
  // which finds it in slot g-1
  float* red_h = Red + hrs * (4 * 3 * 4 * 2 * 64);
#pragma unroll
  for (int g = 1; g < 4; ++g) {
    float* dst = red_h + (((((q + g) & 3) * 3 + (g - 1)) * 4) * 2) * 64 + lane;
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
      dst[(ff * 2 + 0) * 64] = acc[g * 4 + ff][0];
      dst[(ff * 2 + 1) * 64] = acc[g * 4 + ff][1];
    }
  }
  // this wave's four frames (4q .. 4q+3): shape blend (smpl_layer.py:88-95), summed on its own like the reference
  // (rows of betas_T past NB are zero: those terms add 0 * 0)
  float vs[4][2];   // v_template + S
  {
    f32x4 bt[kTileNB];
#pragma unroll
    for (int l = 0; l < kTileNB; ++l) bt[l] = *reinterpret_cast<const f32x4*>(&Bt[l * kFB + f0]);
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      float sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int l = 0; l < kTileNB; ++l)
#pragma unroll
        for (int ff = 0; ff < 4; ++ff) sb[ff] = __builtin_fmaf(sdv[rs][l], bt[l][ff], sb[ff]);
#pragma unroll
      for (int ff = 0; ff < 4; ++ff) vs[ff][rs] = vtmp[rs] + sb[ff];
    }
  }
  PR_STAMP(4);
  __syncthreads();   // partial sums are in LDS
  PR_STAMP(5);
  PR_STAMP(6);

  const int l0 = lane - c;
  // the other quarters' partial sums of this wave's frames, by how far behind the source quarter is: slot g-1 came from
  // quarter (q - g) mod 4
  const float* rd = red_h + (q * 3 * 4 * 2) * 64 + lane;
#pragma unroll
  for (int ff = 0; ff < 4; ++ff) {
    const int f = f0 + ff;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const float own = acc[ff][rs];
      const float o1 = rd[((0 * 4 + ff) * 2 + rs) * 64];   // from quarter q-1
      const float o2 = rd[((1 * 4 + ff) * 2 + rs) * 64];   // from quarter q-2
      const float o3 = rd[((2 * 4 + ff) * 2 + rs) * 64];   // from quarter q-3
      // quarters added in coefficient order (0, 1, 2, 3), as smpl_skin does
      float P;
      if (q == 0) P = ((own + o3) + o2) + o1;
      else if (q == 1) P = ((o1 + own) + o3) + o2;
      else if (q == 2) P = ((o2 + o1) + own) + o3;
      else P = ((o3 + o2) + o1) + own;
      const float vp = vs[ff][rs] + P;   // (v_template + S) + P in the reference's order
      const float x = __shfl(vp, l0, 64), y = __shfl(vp, l0 + 1, 64), z = __shfl(vp, l0 + 2, 64);
      // smpl_layer.py:134 T = A . W^T (row c of the blended transform), nonzero weights only
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NNZ_MAX; ++k) {
        if (k < a.NNZ) {
          const f32x4 ar = *reinterpret_cast<const f32x4*>(&As[(f * kJ + jidx[rs][k]) * 12 + c * 4]);
          t[0] = __builtin_fmaf(jw[rs][k], ar[0], t[0]);
          t[1] = __builtin_fmaf(jw[rs][k], ar[1], t[1]);
          t[2] = __builtin_fmaf(jw[rs][k], ar[2], t[2]);
          t[3] = __builtin_fmaf(jw[rs][k], ar[3], t[3]);
        }
      }
      // smpl_layer.py:143 (T * [v;1]).sum over the 4 columns, then the centring / translation offset
      const float o = __builtin_fmaf(t[2], z, __builtin_fmaf(t[1], y, t[0] * x)) + t[3] + Off[f * 3 + c];
      if (active[rs] && fb0 + f < a.B) a.verts[((long)(fb0 + f) * a.V + v[rs]) * 3 + c] = o;
    }
  }
  PR_STAMP(7);
#undef PR_STAMP
}

}  // namespace
}  // namespace pr

struct pr_smpl {
  int device = 0, V = 0, R = 0, NB = 0, NP = 0, NNZ = 0, max_batch = 0, Bs = 0;
  pr::Tree tree;
  std::vector<void*> allocs;
  float *posedirs_T = nullptr, *shapedirs_T = nullptr, *v_template = nullptr, *ell_w = nullptr;
  int* ell_idx = nullptr;
  float *J_template = nullptr, *J_dirs = nullptr, *model_betas = nullptr;
  float *A = nullptr, *pm_T = nullptr, *betas_T = nullptr, *voff = nullptr, *joints_tmp = nullptr;
  int* flags = nullptr;
  int* tree_dev = nullptr;
  int tile = 1;  // register-tiled skinning kernel (smpl_skin_tile) where it applies; POSERISK_SMPL_TILE=0: A/B timing
};

namespace pr {
namespace {

template <typename T>
int smpl_upload(pr_smpl* h, const std::vector<T>& host, T** out) {
  void* d = nullptr;
  const size_t bytes = std::max<size_t>(host.size() * sizeof(T), 16);
  PR_HIP(hipMalloc(&d, bytes));
  h->allocs.push_back(d);
  PR_HIP(hipMemset(d, 0, bytes));
  if (!host.empty()) PR_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)d;
  return PR_OK;
}

int smpl_build(pr_smpl* h, const float* vt, const float* sd, const float* pd, const float* jr,
               const float* w, const int32_t* parents, const float* mb) {
  const int V = h->V, NB = h->NB, NP = h->NP;
  const int R = ceil_div(3 * V, 64) * 64 + 64;  // padded so inactive lanes may read row 0..R-1 safely
  h->R = R;
  const int NPpad = ceil_div(NP, 8) * 8;  // zero rows up to a multiple of 8 (smpl_skin's unroll)
  std::vector<float> pT((size_t)NPpad * R, 0.f), sT((size_t)std::max(NB, 1) * R, 0.f), vtp(R, 0.f);
  for (int r = 0; r < 3 * V; ++r) {
    vtp[r] = vt[r];
    for (int p = 0; p < NP; ++p) pT[(size_t)p * R + r] = pd[(size_t)r * NP + p];
    for (int l = 0; l < NB; ++l) sT[(size_t)l * R + r] = sd[(size_t)r * NB + l];
  }
  PR_TRY(smpl_upload(h, pT, &h->posedirs_T));
  PR_TRY(smpl_upload(h, sT, &h->shapedirs_T));
  PR_TRY(smpl_upload(h, vtp, &h->v_template));
  // joint regressor contracted with the template / shape directions in double
  std::vector<float> Jt(kJ * 3), Jd((size_t)kJ * 3 * std::max(NB, 1), 0.f);
  for (int j = 0; j < kJ; ++j)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int v = 0; v < V; ++v) s += (double)jr[(size_t)j * V + v] * vt[v * 3 + c];
      Jt[j * 3 + c] = (float)s;
      for (int l = 0; l < NB; ++l) {
        double sl = 0;
        for (int v = 0; v < V; ++v) sl += (double)jr[(size_t)j * V + v] * sd[((size_t)v * 3 + c) * NB + l];
        Jd[(size_t)(j * 3 + c) * NB + l] = (float)sl;
      }
    }
  PR_TRY(smpl_upload(h, Jt, &h->J_template));
  PR_TRY(smpl_upload(h, Jd, &h->J_dirs));
  std::vector<float> mbv(kMaxNB, 0.f);
  if (mb) std::copy(mb, mb + NB, mbv.begin());
  PR_TRY(smpl_upload(h, mbv, &h->model_betas));
  // ELL skinning weights
  int nnz = 1;
  for (int v = 0; v < V; ++v) {
    int n = 0;
    for (int j = 0; j < kJ; ++j) n += (w[(size_t)v * kJ + j] != 0.f);
    nnz = std::max(nnz, n);
  }
  h->NNZ = nnz;
  std::vector<int> eidx((size_t)nnz * V, 0);
  std::vector<float> ew((size_t)nnz * V, 0.f);
  for (int v = 0; v < V; ++v) {
    int k = 0;
    for (int j = 0; j < kJ; ++j)
      if (w[(size_t)v * kJ + j] != 0.f) {
        eidx[(size_t)k * V + v] = j;
        ew[(size_t)k * V + v] = w[(size_t)v * kJ + j];
        ++k;
      }
  }
  PR_TRY(smpl_upload(h, eidx, &h->ell_idx));
  PR_TRY(smpl_upload(h, ew, &h->ell_w));
  // kinematic tree
  for (int j = 0; j < kJ; ++j) h->tree.parent[j] = parents[j];
  h->tree.max_depth = 0;
  for (int j = 0; j < kJ; ++j) {
    int d = 0, p = j;
    while (p > 0 && parents[p] >= 0 && d < kJ) {
      p = parents[p];
      ++d;
    }
    h->tree.depth[j] = d;
    h->tree.max_depth = std::max(h->tree.max_depth, d);
  }
  // workspaces
  const int Bs = ceil_div(h->max_batch, kFB) * kFB;
  h->Bs = Bs;
  PR_TRY(smpl_upload(h, std::vector<float>((size_t)Bs * kJ * 12, 0.f), &h->A));
  PR_TRY(smpl_upload(h, std::vector<float>((size_t)ceil_div(NP, 8) * 8 * Bs, 0.f), &h->pm_T));
  PR_TRY(smpl_upload(h, std::vector<float>((size_t)kMaxNB * Bs, 0.f), &h->betas_T));
  PR_TRY(smpl_upload(h, std::vector<float>((size_t)Bs * 3, 0.f), &h->voff));
  PR_TRY(smpl_upload(h, std::vector<float>((size_t)Bs * kJ * 3, 0.f), &h->joints_tmp));
  PR_TRY(smpl_upload(h, std::vector<int>(4, 0), &h->flags));
  std::vector<int> tr(2 * kJ);
  for (int j = 0; j < kJ; ++j) {
    tr[j] = h->tree.parent[j];
    tr[kJ + j] = h->tree.depth[j];
  }
  PR_TRY(smpl_upload(h, tr, &h->tree_dev));
  return PR_OK;
}

// One chunk (B <= max_batch) of the forward.
int smpl_run_chunk(pr_smpl* h, float* pose, const float* betas, const float* trans, const int* flags,
                   int B, int center_idx, int overwrite_root, int joint_cam_mode, float* verts,
                   float* joints, hipStream_t s) {
  PoseArgs pa;
  pa.pose = pose; pa.betas = betas; pa.trans = trans; pa.flags = flags;
  pa.J_template = h->J_template; pa.J_dirs = h->J_dirs; pa.model_betas = h->model_betas;
  pa.A = h->A; pa.pm_T = h->pm_T; pa.betas_T = h->betas_T; pa.voff = h->voff;
  pa.joints = joints ? joints : h->joints_tmp;
  pa.B = B; pa.Bs = h->Bs; pa.NB = h->NB; pa.b0 = 0;
  pa.overwrite_root = overwrite_root; pa.joint_cam_mode = joint_cam_mode; pa.center_idx = center_idx;
  pa.tree = h->tree_dev;
  pa.max_depth = h->tree.max_depth;
  hipLaunchKernelGGL(smpl_pose, dim3(B), dim3(64), 0, s, pa);
  PR_TRY(check_launch("smpl_pose"));
  if (verts) {
    SkinArgs sa;
    sa.stamps = nullptr;
    sa.posedirs_T = h->posedirs_T; sa.shapedirs_T = h->shapedirs_T; sa.v_template = h->v_template;
    sa.ell_idx = h->ell_idx; sa.ell_w = h->ell_w; sa.A = h->A; sa.pm_T = h->pm_T;
    sa.betas_T = h->betas_T; sa.voff = h->voff; sa.verts = verts;
    sa.V = h->V; sa.R = h->R; sa.NP = h->NP; sa.NPpad = ceil_div(h->NP, 8) * 8; sa.NB = h->NB; sa.NNZ = h->NNZ; sa.B = B; sa.Bs = h->Bs;
    // The variant is fixed per handle (by its max_batch, not by this call's B) so that a frame's bits
    // never depend on how the caller partitions its frames into calls.
    if (h->tile && h->max_batch <= 128 && h->NNZ <= 4 && h->NB <= kTileNB && sa.NPpad == 208) {
      // SMPL's own shape (4 weights per vertex, 10 shape coefficients): same bits as smpl_skin, 12 % faster at B = 64
      // (21.6 against 24.6 us), the same at B <= 16; smpl_skin stays for denser weights or more shape coefficients
      sa.n_fg = ceil_div(B, kFB);
      sa.n_rt = ceil_div(ceil_div(h->V, 21 * kRS), 8) * 8;
      static std::atomic<uint64_t> done{0};
      PR_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(smpl_skin_tile<4>), kTileLds, done));
      hipLaunchKernelGGL(smpl_skin_tile<4>, dim3(sa.n_rt * sa.n_fg), dim3(kTileThreads), kTileLds, s, sa);
    } else if (h->max_batch <= 128) {  // latency-bound regime: coefficient range split over the 4 waves
      sa.n_fg = ceil_div(B, kFB);
      sa.n_rt = ceil_div(ceil_div(h->V, 21), 8) * 8;
      const dim3 grid(sa.n_rt * sa.n_fg);
      if (h->NNZ <= 4) hipLaunchKernelGGL(smpl_skin<4>, grid, dim3(256), 0, s, sa);
      else if (h->NNZ <= 8) hipLaunchKernelGGL(smpl_skin<8>, grid, dim3(256), 0, s, sa);
      else hipLaunchKernelGGL(smpl_skin<kJ>, grid, dim3(256), 0, s, sa);
    } else {         // throughput regime: one wave per 63 rows x 16 frames
      sa.n_fg = ceil_div(B, kFB);
      sa.n_rt = ceil_div(ceil_div(ceil_div(h->V, 21), 4), 8) * 8;
      const dim3 grid(sa.n_rt * sa.n_fg);
      if (h->NNZ <= 4) hipLaunchKernelGGL(smpl_skin_rows<4>, grid, dim3(256), 0, s, sa);
      else if (h->NNZ <= 8) hipLaunchKernelGGL(smpl_skin_rows<8>, grid, dim3(256), 0, s, sa);
      else hipLaunchKernelGGL(smpl_skin_rows<kJ>, grid, dim3(256), 0, s, sa);
    }
    PR_TRY(check_launch("smpl_skin"));
  }
  return PR_OK;
}

}  // namespace
}  // namespace pr

extern "C" {

int pr_smpl_create(int device, const float* v_template_host, const float* shapedirs_host,
                   const float* posedirs_host, const float* J_regressor_host, const float* weights_host,
                   const int32_t* parents_host, const float* model_betas_host, int V, int J, int NB,
                   int max_batch, pr_smpl_t** out) {
  using namespace pr;
  PR_REQUIRE(out && v_template_host && shapedirs_host && posedirs_host && J_regressor_host &&
                 weights_host && parents_host,
             "pr_smpl_create: null argument");
  PR_REQUIRE(J == kJ, "pr_smpl_create: J must be 24 (got %d)", J);
  PR_REQUIRE(V > 0 && V <= (1 << 20), "pr_smpl_create: V %d out of range", V);
  PR_REQUIRE(NB >= 0 && NB <= kMaxNB, "pr_smpl_create: NB %d out of range", NB);
  PR_REQUIRE(max_batch > 0 && max_batch <= 65536, "pr_smpl_create: max_batch %d out of range", max_batch);
  PR_REQUIRE(parents_host[0] < 0, "pr_smpl_create: parents[0] must be negative (root)");
  for (int j = 1; j < kJ; ++j)
    PR_REQUIRE(parents_host[j] >= 0 && parents_host[j] < j, "pr_smpl_create: parents[%d]=%d not topological", j,
               parents_host[j]);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("pr_smpl_create: no HIP device visible");
    return PR_ERR_NO_DEVICE;
  }
  PR_REQUIRE(device >= 0 && device < ndev, "pr_smpl_create: device %d of %d", device, ndev);
  DeviceGuard g(device);
  PR_TRY(refuse_under_declared_capture("pr_smpl_create"));
  std::unique_ptr<pr_smpl> h(new pr_smpl);
  h->device = device; h->V = V; h->NB = NB; h->NP = (kJ - 1) * 9; h->max_batch = max_batch;
  if (const char* e = getenv("POSERISK_SMPL_TILE")) h->tile = atoi(e) != 0;
  int st = smpl_build(h.get(), v_template_host, shapedirs_host, posedirs_host, J_regressor_host,
                      weights_host, parents_host, model_betas_host);
  if (st != PR_OK) {
    for (void* p : h->allocs) (void)hipFree(p);
    return st;
  }
  *out = h.release();
  return PR_OK;
}

int pr_smpl_destroy(pr_smpl_t* h) {
  if (!h) return PR_OK;
  pr::DeviceGuard g(h->device);
  PR_TRY(pr::refuse_under_declared_capture("pr_smpl_destroy"));   // the handle stays valid: destroy it after the capture
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return PR_OK;
}

int pr_smpl_forward(pr_smpl_t* h, const float* pose_dev, const float* betas_dev, const float* trans_dev,
                    int B, int center_idx, float* verts_dev, float* joints_dev, void* stream) {
  using namespace pr;
  PR_REQUIRE(h && pose_dev, "pr_smpl_forward: null argument");
  PR_REQUIRE(B >= 0 && center_idx < kJ, "pr_smpl_forward: bad B/center_idx");
  if (B == 0) return PR_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(smpl_flags, dim3(1), dim3(256), 0, s, betas_dev, (long)B * h->NB, trans_dev, (long)B * 3,
                     h->flags);
  PR_TRY(check_launch("smpl_flags"));
  for (int b0 = 0; b0 < B; b0 += h->max_batch) {
    const int nb = std::min(h->max_batch, B - b0);
    PR_TRY(smpl_run_chunk(h, const_cast<float*>(pose_dev) + (long)b0 * 72,
                          betas_dev ? betas_dev + (long)b0 * h->NB : nullptr,
                          trans_dev ? trans_dev + (long)b0 * 3 : nullptr, h->flags, nb, center_idx, 0, 0,
                          verts_dev ? verts_dev + (long)b0 * h->V * 3 : nullptr,
                          joints_dev ? joints_dev + (long)b0 * kJ * 3 : nullptr, s));
  }
  return PR_OK;
}

int pr_smpl_joint_cam(pr_smpl_t* h, float* axis_angle_dev, int N, float* joint_cam_dev, float* verts_dev,
                      void* stream) {
  using namespace pr;
  PR_REQUIRE(h && axis_angle_dev && joint_cam_dev, "pr_smpl_joint_cam: null argument");
  PR_REQUIRE(N >= 0, "pr_smpl_joint_cam: negative N");
  hipStream_t s = (hipStream_t)stream;
  for (int b0 = 0; b0 < N; b0 += h->max_batch) {
    const int nb = std::min(h->max_batch, N - b0);
    PR_TRY(smpl_run_chunk(h, axis_angle_dev + (long)b0 * 72, nullptr, nullptr, nullptr, nb, -1, 1, 1,
                          verts_dev ? verts_dev + (long)b0 * h->V * 3 : nullptr,
                          joint_cam_dev + (long)b0 * kJ * 3, s));
  }
  return PR_OK;
}

}  // extern "C"
