// Per-hypothesis pose arithmetic shared by the stand-alone kernels (crop_window_tf_kernel, pose_update_kernel) and the fused tail of
// a refinement pass (refine_tail_kernel): ONE definition each, so the fused pass computes what the building blocks compute bit for bit
// (the library is built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

// pose update (predict_pose_refine.py:195-231): float32, mirrors oracle/predict.py:pose_update
struct DeepimArgs {        // trans_rep='deepim' (predict_pose_refine.py:201-215): the crop transforms of this pass, the intrinsics, input_resize[0]
  const float *tf;         // N x 9
  float K[9];
  float resize;
};

__device__ __forceinline__ void inv3x3(const float *m, float *o) {     // adjugate / determinant
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float id = 1.f / (m[0] * c00 + m[1] * c01 + m[2] * c02);
  o[0] = c00 * id, o[1] = (m[2] * m[7] - m[1] * m[8]) * id, o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id, o[4] = (m[0] * m[8] - m[2] * m[6]) * id, o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id, o[7] = (m[1] * m[6] - m[0] * m[7]) * id, o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// One hypothesis `b` of predict_pose_refine.py:195-231; `trans` / `rot` are THIS hypothesis' rows (global memory or LDS).  `outp` may be
// `poseA` (in-place update): the pose is read completely first.
__device__ __forceinline__ void pose_update_one(int b, const float *__restrict__ poseA, const float *trans, const float *rot,
                                                int rot_dim, int trans_tanh, float tn0, float tn1, float tn2, float rot_normalizer,
                                                float trans_scale, const DeepimArgs &dp, float *__restrict__ outp) {
  float A[12];                 // the pose is read completely before anything is written: outp may be poseA (in-place update)
#pragma unroll
  for (int i = 0; i < 12; ++i) A[i] = poseA[(size_t)b * 16 + i];
  float td[3] = {trans[0], trans[1], trans[2]};
  if (trans_tanh == 1) {
    td[0] = tanhf(td[0]) * tn0;
    td[1] = tanhf(td[1]) * tn1;
    td[2] = tanhf(td[2]) * tn2;
  } else if (trans_tanh == 2) {
    // deepim: the network predicts the shift of the projected centre in the crop (in units of the crop size) and the depth ratio
    const float *tf = dp.tf + (size_t)b * 9, *K = dp.K;
    const float cx = A[3], cy = A[7], cz = A[11];
    const float z_pred = td[2] * cz;
    float uvw[3];
    for (int r = 0; r < 3; ++r) uvw[r] = K[r * 3] * cx + K[r * 3 + 1] * cy + K[r * 3 + 2] * cz;
    const float u = uvw[0] / uvw[2], v = uvw[1] / uvw[2], w1 = uvw[2] / uvw[2];
    const float uc = tf[0] * u + tf[1] * v + tf[2] * w1 + td[0] * dp.resize;      // uvA_crop + trans[:2] * input_resize[0]
    const float vc = tf[3] * u + tf[4] * v + tf[5] * w1 + td[1] * dp.resize;
    float ti[9], Ki[9];
    inv3x3(tf, ti);
    inv3x3(K, Ki);
    const float up = ti[0] * uc + ti[1] * vc + ti[2], vp = ti[3] * uc + ti[4] * vc + ti[5];      // transform_pts(uv_pred_crop, tf^-1)
    td[0] = (Ki[0] * up + Ki[1] * vp + Ki[2]) * z_pred - cx;
    td[1] = (Ki[3] * up + Ki[4] * vp + Ki[5]) * z_pred - cy;
    td[2] = (Ki[6] * up + Ki[7] * vp + Ki[8]) * z_pred - cz;
  }
  td[0] *= trans_scale;
  td[1] *= trans_scale;
  td[2] *= trans_scale;
  float R[9];  // rot_mat_delta (already transposed as in the reference)
  if (rot_dim == 3) {
    const float x = tanhf(rot[0]) * rot_normalizer, y = tanhf(rot[1]) * rot_normalizer, z = tanhf(rot[2]) * rot_normalizer;
    const float nrm = fmaxf(x * x + y * y + z * z, 1e-4f);
    const float th = sqrtf(nrm), ith = 1.f / th;
    const float f1 = ith * sinf(th), f2 = ith * ith * (1.f - cosf(th));
    const float Kx[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float K2[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K2[r * 3 + c] = Kx[r * 3] * Kx[c] + Kx[r * 3 + 1] * Kx[3 + c] + Kx[r * 3 + 2] * Kx[6 + c];
    float E[9];
    for (int i = 0; i < 9; ++i) E[i] = f1 * Kx[i] + f2 * K2[i] + ((i % 4 == 0) ? 1.f : 0.f);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = E[c * 3 + r];  // .permute(0,2,1)
  } else {
    const float *d6 = rot;
    float a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
    float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float dp = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float b2[3] = {a2[0] - dp * b1[0], a2[1] - dp * b1[1], a2[2] - dp * b1[2]};
    float n2 = fmaxf(sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]), 1e-12f);
    b2[0] /= n2;
    b2[1] /= n2;
    b2[2] /= n2;
    float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    // rows (b1,b2,b3) then transposed
    for (int c = 0; c < 3; ++c) {
      R[c * 3 + 0] = b1[c];
      R[c * 3 + 1] = b2[c];
      R[c * 3 + 2] = b3[c];
    }
  }
  float *O = outp + (size_t)b * 16;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) O[r * 4 + c] = R[r * 3] * A[c] + R[r * 3 + 1] * A[4 + c] + R[r * 3 + 2] * A[8 + c];
    O[r * 4 + 3] = A[r * 4 + 3] + td[r];
  }
  O[12] = 0.f;
  O[13] = 0.f;
  O[14] = 0.f;
  O[15] = 1.f;
}

// compute_crop_window_tf_batch(method='box_3d') (src/Utils.py:577-621) + bbox2d_ori (predict_pose_refine.py:44-45) of hypothesis `b`:
// float32, left to right, no FMA contraction (mirrors oracle/geometry.py:compute_crop_window_tf_batch).  `bbox` may be null.
__device__ __forceinline__ void crop_window_tf_one(int b, const float *poses, const CropWindowK &c, float *tf, float *bbox) {
  const float k00 = c.k00, k01 = c.k01, k02 = c.k02, k10 = c.k10, k11 = c.k11, k12 = c.k12, k20 = c.k20, k21 = c.k21, k22 = c.k22;
  const float radius = c.radius, ow = c.ow, oh = c.oh;
  const float *p = poses + (size_t)b * 16;
  float tx = p[3], ty = p[7], tz = p[11];
  float offx[5] = {0.f, radius, -radius, 0.f, 0.f};
  float offy[5] = {0.f, 0.f, 0.f, radius, -radius};
  float u[5], v[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float x = __fadd_rn(tx, offx[k]), y = __fadd_rn(ty, offy[k]), z = __fadd_rn(tz, 0.f);
    float pu = __fadd_rn(__fadd_rn(__fmul_rn(k00, x), __fmul_rn(k01, y)), __fmul_rn(k02, z));
    float pv = __fadd_rn(__fadd_rn(__fmul_rn(k10, x), __fmul_rn(k11, y)), __fmul_rn(k12, z));
    float pw = __fadd_rn(__fadd_rn(__fmul_rn(k20, x), __fmul_rn(k21, y)), __fmul_rn(k22, z));
    u[k] = __fdiv_rn(pu, pw);
    v[k] = __fdiv_rn(pv, pw);
  }
  float rad = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    rad = fmaxf(rad, fabsf(__fsub_rn(u[k], u[0])));
    rad = fmaxf(rad, fabsf(__fsub_rn(v[k], v[0])));
  }
  float left = rintf(__fsub_rn(u[0], rad)), right = rintf(__fadd_rn(u[0], rad));
  float top = rintf(__fsub_rn(v[0], rad)), bottom = rintf(__fadd_rn(v[0], rad));
  // `out_size[0]/(right-left)` is int / Tensor in the reference -> Tensor.__rtruediv__ = reciprocal() * scalar
  float sx = __fmul_rn(__fdiv_rn(1.f, __fsub_rn(right, left)), ow), sy = __fmul_rn(__fdiv_rn(1.f, __fsub_rn(bottom, top)), oh);
  float t02 = __fmul_rn(sx, -left), t12 = __fmul_rn(sy, -top);
  float *T = tf + (size_t)b * 9;
  T[0] = sx; T[1] = 0.f; T[2] = t02;
  T[3] = 0.f; T[4] = sy; T[5] = t12;
  T[6] = 0.f; T[7] = 0.f; T[8] = 1.f;
  if (bbox) {
    // tf^-1 applied to (0,0) and (ow-1,oh-1): inverse of [[sx,0,t02],[0,sy,t12],[0,0,1]]
    float i00 = __fdiv_rn(1.f, sx), i11 = __fdiv_rn(1.f, sy);
    float i02 = -__fdiv_rn(t02, sx), i12 = -__fdiv_rn(t12, sy);
    float *B = bbox + (size_t)b * 4;
    B[0] = i02;
    B[1] = i12;
    B[2] = __fadd_rn(__fmul_rn(i00, ow - 1.f), i02);
    B[3] = __fadd_rn(__fmul_rn(i11, oh - 1.f), i12);
  }
}

// pose @ get_tf_to_centered_mesh() (src/estimater.py:82-86,268): that matrix is the identity with the translation column cneg = -model_center,
// so the rotation block is copied and column 3 is row . (cneg, 1), summed left to right in float32.
__device__ __forceinline__ void pose_of_mesh_one(const float *pose, const float *cneg, float *out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float p0 = pose[i * 4 + 0], p1 = pose[i * 4 + 1], p2 = pose[i * 4 + 2], p3 = pose[i * 4 + 3];
    out[i * 4 + 0] = p0, out[i * 4 + 1] = p1, out[i * 4 + 2] = p2;
    out[i * 4 + 3] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(p0, cneg[0]), __fmul_rn(p1, cneg[1])), __fmul_rn(p2, cneg[2])), p3);
  }
}
