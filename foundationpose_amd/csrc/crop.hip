// Observed-frame crops (side B) and depth pre-processing for gfx950.
//
// crop_observed: one gather kernel replaces kornia.warp_perspective x2..x5 + the dataset batch
// transform (predict_pose_refine.py:63,72; predict_score.py:89-90; h5_dataset.py:101-112,158-170).
// kornia 0.7.2's map (normalise with (size-1), invert, grid_sample(align_corners=False)) collapses,
// for the axis-aligned crop transform tf = [[sx,0,tx],[0,sy,ty],[0,0,1]], to
//     x_src = ((i - tx)/sx) * W/(W-1) - 0.5,   y_src = ((j - ty)/sy) * H/(H-1) - 0.5
// evaluated here in float64 per pixel (bilinear taps then use the float32-rounded coordinate, as
// F.grid_sample does; nearest lookups use round_snapped below).
// The frame (480x640x(3+3) floats = 7 MB) is L2/MALL resident; every hypothesis re-reads it from
// cache and the only HBM traffic is the 16 B/pixel network-ready store.
#include "common.h"

// nearbyint in float64 with coordinates within 1e-4 px of a half-integer treated as exact ties
// (round half to even) - same rule as oracle/warp.py:round_half_even_snapped, which explains why.
__device__ __forceinline__ int round_snapped(double x) {
  double f = floor(x), t = x - f;
  if (fabs(t - 0.5) < 1e-4) {
    long long fi = (long long)f;
    return (int)((fi & 1) ? fi + 1 : fi);
  }
  return (int)floor(x + 0.5);
}

__device__ __forceinline__ float frame_at(const float *img, int H, int W, int C, int y, int x, int c) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? img[((size_t)y * W + x) * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void crop_observed_kernel(CropArgs a) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const float *T = a.tf + (size_t)b * 9;
  const float *pose = a.poses + (size_t)b * 16;
  const int H = a.H, W = a.W;
  // the per-hypothesis affine coefficients (eight float64 divisions) once per workgroup, not once per pixel
  __shared__ double s_aff[4];
  if (threadIdx.x == 0) {
    const double sx0 = T[0], tx0 = T[2], sy0 = T[4], ty0 = T[5];
    s_aff[0] = (1.0 / sx0) * W / (W - 1.0);
    s_aff[1] = (-tx0 / sx0) * W / (W - 1.0) - 0.5;
    s_aff[2] = (1.0 / sy0) * H / (H - 1.0);
    s_aff[3] = (-ty0 / sy0) * H / (H - 1.0) - 0.5;
  }
  __syncthreads();
  if (p >= a.Ho * a.Wo) return;
  const int j = p / a.Wo, i = p - j * a.Wo;
  const double sx = T[0], tx = T[2], sy = T[4], ty = T[5];
  const double ax = s_aff[0], bx = s_aff[1], ay = s_aff[2], by = s_aff[3];
  const double xd = ax * i + bx, yd = ay * j + by;
  const float x = (float)xd, y = (float)yd;

  // ---- rgb: bilinear, zeros padding (F.grid_sample arithmetic, float32) ----
  float rgb[3];
  {
    float x0f = floorf(x), y0f = floorf(y);
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    float x1f = x0f + 1.f, y1f = y0f + 1.f;
    float nw = (x1f - x) * (y1f - y), ne = (x - x0f) * (y1f - y), sw = (x1f - x) * (y - y0f), se = (x - x0f) * (y - y0f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      acc = __fadd_rn(acc, __fmul_rn(frame_at(a.rgb, H, W, 3, y0, x0, c), nw));
      acc = __fadd_rn(acc, __fmul_rn(frame_at(a.rgb, H, W, 3, y0, x1, c), ne));
      acc = __fadd_rn(acc, __fmul_rn(frame_at(a.rgb, H, W, 3, y1, x0, c), sw));
      acc = __fadd_rn(acc, __fmul_rn(frame_at(a.rgb, H, W, 3, y1, x1, c), se));
      rgb[c] = __fdiv_rn(acc, 255.f);   // h5_dataset.py:123-124
    }
  }
  // ---- geometry: nearest (round half to even), zeros padding ----
  const int qx = round_snapped(xd), qy = round_snapped(yd);
  const bool q_in = (qx >= 0 && qx < W && qy >= 0 && qy < H);
  float xyz[3] = {0.f, 0.f, 0.f};
  float invalid_thres;
  if (a.mode == 0) {
    invalid_thres = 0.001f;
    if (q_in) {
#pragma unroll
      for (int c = 0; c < 3; ++c) xyz[c] = a.geom[((size_t)qy * W + qx) * 3 + c];
    }
  } else {
    // scorer: depthB crop -> full-res (nearest) -> back-project at full-res pixel (qx,qy) -> crop (nearest)
    invalid_thres = 0.1f;
    if (q_in) {
      // full-res pixel q samples the CROP image at p' (warp with M = tf^-1, src = crop (Ho,Wo), dst = (H,W)):
      //   p~ = tf * q ;  x' = p~ * Wo/(Wo-1) - 0.5
      const double cxp = (sx * qx + tx) * a.Wo / (a.Wo - 1.0) - 0.5;
      const double cyp = (sy * qy + ty) * a.Ho / (a.Ho - 1.0) - 0.5;
      const int pi = round_snapped(cxp), pj = round_snapped(cyp);
      float z = 0.f;
      if (pi >= 0 && pi < a.Wo && pj >= 0 && pj < a.Ho) {
        const int rx = round_snapped(ax * pi + bx), ry = round_snapped(ay * pj + by);
        if (rx >= 0 && rx < W && ry >= 0 && ry < H) z = a.geom[(size_t)ry * W + rx];
      }
      // depth2xyzmap_batch (src/Utils.py:420-438), zfar = inf, float32
      const float fx = (float)a.K[0], fy = (float)a.K[4], cx = (float)a.K[2], cy = (float)a.K[5];
      if (!(z < 0.001f)) {
        xyz[0] = __fdiv_rn(__fmul_rn(__fsub_rn((float)qx, cx), z), fx);
        xyz[1] = __fdiv_rn(__fmul_rn(__fsub_rn((float)qy, cy), z), fy);
        xyz[2] = z;
      }
    }
  }
  // ---- batch transform (h5_dataset.py:104-112 | :162-170) ----
  const bool invalid = xyz[2] < invalid_thres;
  const float radius = __fdiv_rn(a.mesh_diameter, 2.f);
  const float inv_r = __fdiv_rn(1.f, radius);
  float o6[6] = {rgb[0], rgb[1], rgb[2], 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = __fsub_rn(xyz[c], pose[c * 4 + 3]);
    if (a.normalize_xyz) {
      v = __fmul_rn(v, inv_r);
      if (invalid || fabsf(v) >= 2.f) v = 0.f;
    }
    o6[3 + c] = v;
  }
  if (a.out_fmt == 0) {
    float *out = (float *)a.out + (size_t)b * 6 * a.Ho * a.Wo;
#pragma unroll
    for (int c = 0; c < 6; ++c) out[(size_t)c * a.Ho * a.Wo + p] = o6[c];
  } else {
    half8 hv;
#pragma unroll
    for (int c = 0; c < 6; ++c) hv[c] = (f16)o6[c];
    hv[6] = (f16)0.f;
    hv[7] = (f16)0.f;
    *reinterpret_cast<half8 *>((f16 *)a.out + ((size_t)b * a.Ho * a.Wo + p) * 8) = hv;
  }
}

int launch_crop_observed(const CropArgs &a, hipStream_t s) {
  FP_REQUIRE(a.mode == 0 || a.mode == 1, "crop_observed: mode must be 0 (refiner) or 1 (scorer)");
  FP_REQUIRE(a.H > 1 && a.W > 1 && a.Ho > 1 && a.Wo > 1, "crop_observed: degenerate image size");
  if (a.N == 0) return FP_OK;
  dim3 grid((a.Ho * a.Wo + 255) / 256, a.N);
  hipLaunchKernelGGL(crop_observed_kernel, grid, dim3(256), 0, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// ----------------------------------------------------------------------------------------------
// kornia.warp_perspective(mode='nearest', align_corners=False, zeros padding) of a channel-last image
// batch with the axis-aligned crop transforms: the warps of the use_normal branch
// (predict_pose_refine.py:74-76: normalAs = warp(rendered normals (B,h,w,3), tf_to_crops) and
// normalBs = warp(the frame's normal map, broadcast over B, tf_to_crops)).  Same source coordinate
// and tie rule as crop_observed_kernel's nearest lookups; output planar (B,C,Ho,Wo) float32.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_nearest_kernel(const float *__restrict__ src, size_t src_batch_stride, int Hs, int Ws, int C,
                                                           const float *__restrict__ tf, int Ho, int Wo, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Ho * Wo) return;
  const float *T = tf + (size_t)b * 9;
  const double sx = T[0], tx = T[2], sy = T[4], ty = T[5];
  const double ax = (1.0 / sx) * Ws / (Ws - 1.0), bx = (-tx / sx) * Ws / (Ws - 1.0) - 0.5;
  const double ay = (1.0 / sy) * Hs / (Hs - 1.0), by = (-ty / sy) * Hs / (Hs - 1.0) - 0.5;
  const int j = p / Wo, i = p - j * Wo;
  const int qx = round_snapped(ax * i + bx), qy = round_snapped(ay * j + by);
  const bool in = qx >= 0 && qx < Ws && qy >= 0 && qy < Hs;
  const float *s = src + (size_t)b * src_batch_stride + ((size_t)(in ? qy : 0) * Ws + (in ? qx : 0)) * C;
  float *o = out + (size_t)b * C * Ho * Wo + p;
  for (int c = 0; c < C; ++c) o[(size_t)c * Ho * Wo] = in ? s[c] : 0.f;
}

int launch_warp_nearest(const float *src, int src_batch, int Hs, int Ws, int C, const float *tf, int N, int Ho, int Wo, float *out,
                        hipStream_t s) {
  FP_REQUIRE(Hs > 1 && Ws > 1 && Ho > 0 && Wo > 0 && C > 0, "warp_nearest: degenerate image size");
  FP_REQUIRE(src_batch == 1 || src_batch == N, "warp_nearest: the source batch is 1 (broadcast) or N");
  if (N == 0) return FP_OK;
  dim3 grid((Ho * Wo + 255) / 256, N);
  hipLaunchKernelGGL(warp_nearest_kernel, grid, dim3(256), 0, s, src, src_batch == 1 ? (size_t)0 : (size_t)Hs * Ws * C, Hs, Ws, C, tf, Ho,
                     Wo, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// ----------------------------------------------------------------------------------------------
// depth pre-processing: the reference's NVIDIA Warp kernels (src/Utils.py:304-395), one thread/pixel
// ----------------------------------------------------------------------------------------------
__global__ void erode_depth_kernel(const float *__restrict__ depth, int H, int W, int radius, float diff_thres, float ratio_thres,
                                   float zfar, float *__restrict__ out) {
  int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  float d_ori = depth[(size_t)h * W + w];
  float bad = 0.f, total = 0.f;
  for (int u = w - radius; u <= w + radius; ++u) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; ++v) {
      if (v < 0 || v >= H) continue;
      float cur = depth[(size_t)v * W + u];
      total += 1.f;
      if (cur < 0.001f || cur >= zfar || fabsf(cur - d_ori) > diff_thres) bad += 1.f;
    }
  }
  out[(size_t)h * W + w] = (__fdiv_rn(bad, total) > ratio_thres) ? 0.f : d_ori;
}

__global__ void bilateral_depth_kernel(const float *__restrict__ depth, int H, int W, int radius, float zfar, float sigmaD,
                                       float sigmaR, float *__restrict__ out) {
  int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  float mean_depth = 0.f;
  int num_valid = 0;
  for (int u = w - radius; u <= w + radius; ++u) {
    if (u < 0 || u >= W) continue;
    for (int v = h - radius; v <= h + radius; ++v) {
      if (v < 0 || v >= H) continue;
      float cur = depth[(size_t)v * W + u];
      if (cur >= 0.001f && cur < zfar) {
        num_valid += 1;
        mean_depth = __fadd_rn(mean_depth, cur);
      }
    }
  }
  float res = 0.f;
  if (num_valid > 0) {
    mean_depth = __fdiv_rn(mean_depth, (float)num_valid);
    float dc = depth[(size_t)h * W + w];
    float sum_w = 0.f, sum = 0.f;
    const float den_d = __fmul_rn(__fmul_rn(2.f, sigmaD), sigmaD), den_r = __fmul_rn(__fmul_rn(2.f, sigmaR), sigmaR);
    for (int u = w - radius; u <= w + radius; ++u) {
      if (u < 0 || u >= W) continue;
      for (int v = h - radius; v <= h + radius; ++v) {
        if (v < 0 || v >= H) continue;
        float cur = depth[(size_t)v * W + u];
        if (cur >= 0.001f && cur < zfar && fabsf(cur - mean_depth) < 0.01f) {
          float sp = __fdiv_rn(-(float)((u - w) * (u - w) + (h - v) * (h - v)), den_d);
          float df = __fsub_rn(dc, cur);
          float rg = __fdiv_rn(__fmul_rn(df, df), den_r);
          float wt = expf(__fsub_rn(sp, rg));
          sum_w = __fadd_rn(sum_w, wt);
          sum = __fadd_rn(sum, __fmul_rn(wt, cur));
        }
      }
    }
    if (sum_w > 0.f) res = __fdiv_rn(sum, sum_w);
  }
  out[(size_t)h * W + w] = res;
}

__global__ void depth2xyz_kernel(const float *__restrict__ depth, int H, int W, float fx, float fy, float cx, float cy, float zfar,
                                 float *__restrict__ xyz) {
  int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  float z = depth[(size_t)h * W + w];
  float o[3] = {0.f, 0.f, 0.f};
  if (!(z < 0.001f) && !(z > zfar)) {
    o[0] = __fdiv_rn(__fmul_rn(__fsub_rn((float)w, cx), z), fx);
    o[1] = __fdiv_rn(__fmul_rn(__fsub_rn((float)h, cy), z), fy);
    o[2] = z;
  }
  size_t p = ((size_t)h * W + w) * 3;
  xyz[p] = o[0];
  xyz[p + 1] = o[1];
  xyz[p + 2] = o[2];
}

static dim3 grid2d(int H, int W) { return dim3((W + 31) / 32, (H + 7) / 8); }

int launch_erode(const float *d, int H, int W, int radius, float diff_thres, float ratio_thres, float zfar, float *out, hipStream_t s) {
  hipLaunchKernelGGL(erode_depth_kernel, grid2d(H, W), dim3(32, 8), 0, s, d, H, W, radius, diff_thres, ratio_thres, zfar, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
int launch_bilateral(const float *d, int H, int W, int radius, float zfar, float sigmaD, float sigmaR, float *out, hipStream_t s) {
  hipLaunchKernelGGL(bilateral_depth_kernel, grid2d(H, W), dim3(32, 8), 0, s, d, H, W, radius, zfar, sigmaD, sigmaR, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
int launch_depth2xyz(const float *d, int H, int W, const double *K, float zfar, float *xyz, hipStream_t s) {
  hipLaunchKernelGGL(depth2xyz_kernel, grid2d(H, W), dim3(32, 8), 0, s, d, H, W, (float)K[0], (float)K[4], (float)K[2], (float)K[5],
                     zfar, xyz);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// The tracking prelude of a frame in ONE launch: erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch (radius 2 both,
// src/estimater.py:256-260).  As three launches a 640x480 frame costs 10 + 23 + 3 us, nearly all of it the latency of 25 + 50
// dependent global loads per pixel behind data-dependent branches.  Here a workgroup stages the raw depth of its 32x8 tile plus a
// 4-pixel border in LDS, writes the eroded depth of the tile plus a 2-pixel border to LDS, and filters from there.  The arithmetic
// and its order (columns outer, rows inner) are those of the three kernels above: the outputs are theirs bit for bit.
#define PF_R 2
#define PF_TW 32
#define PF_TH 8
__global__ __launch_bounds__(PF_TW *PF_TH) void depth_prefilter_kernel(const float *__restrict__ depth, int H, int W, float diff_thres,
                                                                       float ratio_thres, float zfar_e, float zfar_b, float sigmaD, float sigmaR,
                                                                       float fx, float fy, float cx, float cy, float zfar_x,
                                                                       float *__restrict__ out, float *__restrict__ xyz,
                                                                       const unsigned char *__restrict__ rgb_u8, float *__restrict__ rgb_f) {
  constexpr int R = PF_R, RW = PF_TW + 4 * R, RH = PF_TH + 4 * R, EW = PF_TW + 2 * R, EH = PF_TH + 2 * R, NTAP = (2 * R + 1) * (2 * R + 1);
  __shared__ float raw[RH][RW + 1], er[EH][EW + 1], sp_tab[NTAP];
  const float OUTSIDE = -__builtin_inff();            // marks pixels beyond the image (the loops of the three kernels skip them)
  const int tid = threadIdx.y * PF_TW + threadIdx.x, w0 = blockIdx.x * PF_TW, h0 = blockIdx.y * PF_TH;
  // the spatial term of the bilateral weight depends on the tap only: 25 IEEE divisions per workgroup instead of per pixel
  if (tid < NTAP) {
    const int du = tid / (2 * R + 1), dv = tid % (2 * R + 1);
    sp_tab[tid] = __fdiv_rn(-(float)((du - R) * (du - R) + (dv - R) * (dv - R)), __fmul_rn(__fmul_rn(2.f, sigmaD), sigmaD));
  }
  for (int i = tid; i < RH * RW; i += PF_TW * PF_TH) {
    const int r = i / RW, c = i - r * RW, h = h0 - 2 * R + r, w = w0 - 2 * R + c;
    raw[r][c] = (h >= 0 && h < H && w >= 0 && w < W) ? depth[(size_t)h * W + w] : OUTSIDE;
  }
  __syncthreads();
  for (int i = tid; i < EH * EW; i += PF_TW * PF_TH) {
    const int r = i / EW, c = i - r * EW;             // eroded pixel (h0 - R + r, w0 - R + c) = raw[r + R][c + R]
    const float d_ori = raw[r + R][c + R];
    float v = OUTSIDE;
    if (d_ori != OUTSIDE) {
      float bad = 0.f, total = 0.f;
#pragma unroll
      for (int du = 0; du <= 2 * R; ++du)
#pragma unroll
        for (int dv = 0; dv <= 2 * R; ++dv) {
          const float cur = raw[r + dv][c + du];
          if (cur == OUTSIDE) continue;
          total += 1.f;
          if (cur < 0.001f || cur >= zfar_e || fabsf(cur - d_ori) > diff_thres) bad += 1.f;
        }
      v = (__fdiv_rn(bad, total) > ratio_thres) ? 0.f : d_ori;
    }
    er[r][c] = v;
  }
  __syncthreads();
  const int w = w0 + threadIdx.x, h = h0 + threadIdx.y;
  if (w >= W || h >= H) return;
  const int r = threadIdx.y, c = threadIdx.x;          // this pixel = er[r + R][c + R]
  float mean_depth = 0.f;
  int num_valid = 0;
#pragma unroll
  for (int du = 0; du <= 2 * R; ++du)
#pragma unroll
    for (int dv = 0; dv <= 2 * R; ++dv) {
      const float cur = er[r + dv][c + du];             // (OUTSIDE fails cur >= 0.001 like a skipped pixel)
      if (cur >= 0.001f && cur < zfar_b) {
        num_valid += 1;
        mean_depth = __fadd_rn(mean_depth, cur);
      }
    }
  float res = 0.f;
  if (num_valid > 0) {
    mean_depth = __fdiv_rn(mean_depth, (float)num_valid);
    const float dc = er[r + R][c + R];
    float sum_w = 0.f, sum = 0.f;
    const float den_r = __fmul_rn(__fmul_rn(2.f, sigmaR), sigmaR);
#pragma unroll
    for (int du = 0; du <= 2 * R; ++du)
#pragma unroll
      for (int dv = 0; dv <= 2 * R; ++dv) {
        const float cur = er[r + dv][c + du];
        if (cur >= 0.001f && cur < zfar_b && fabsf(cur - mean_depth) < 0.01f) {
          const float sp = sp_tab[du * (2 * R + 1) + dv];
          const float df = __fsub_rn(dc, cur);
          const float rg = __fdiv_rn(__fmul_rn(df, df), den_r);
          const float wt = expf(__fsub_rn(sp, rg));
          sum_w = __fadd_rn(sum_w, wt);
          sum = __fadd_rn(sum, __fmul_rn(wt, cur));
        }
      }
    if (sum_w > 0.f) res = __fdiv_rn(sum, sum_w);
  }
  out[(size_t)h * W + w] = res;
  float o[3] = {0.f, 0.f, 0.f};
  if (!(res < 0.001f) && !(res > zfar_x)) {
    o[0] = __fdiv_rn(__fmul_rn(__fsub_rn((float)w, cx), res), fx);
    o[1] = __fdiv_rn(__fmul_rn(__fsub_rn((float)h, cy), res), fy);
    o[2] = res;
  }
  const size_t q = ((size_t)h * W + w) * 3;
  xyz[q] = o[0];
  xyz[q + 1] = o[1];
  xyz[q + 2] = o[2];
  if (rgb_u8) {                  // the frame's colours as float (what the crop kernels read): exact
    rgb_f[q] = (float)rgb_u8[q];
    rgb_f[q + 1] = (float)rgb_u8[q + 1];
    rgb_f[q + 2] = (float)rgb_u8[q + 2];
  }
}

int launch_depth_prefilter(const float *d, int H, int W, float diff_thres, float ratio_thres, float zfar_e, float zfar_b, float sigmaD,
                           float sigmaR, const double *K, float zfar_x, float *out, float *xyz, const unsigned char *rgb_u8, float *rgb_f, hipStream_t s) {
  hipLaunchKernelGGL(depth_prefilter_kernel, dim3((W + PF_TW - 1) / PF_TW, (H + PF_TH - 1) / PF_TH), dim3(PF_TW, PF_TH), 0, s, d, H, W, diff_thres,
                     ratio_thres, zfar_e, zfar_b, sigmaD, sigmaR, (float)K[0], (float)K[4], (float)K[2], (float)K[5], zfar_x, out, xyz, rgb_u8, rgb_f);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// depth2xyzmap (src/Utils.py:399-417): the host function of the reference works in float64 (numpy promotes the float32 depth
// against the float64 K) and rounds once to float32; depth < 0.001 -> 0.
__global__ void depth2xyz_f64_kernel(const float *__restrict__ depth, int H, int W, double fx, double fy, double cx, double cy,
                                     float *__restrict__ xyz) {
  int w = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y * blockDim.y + threadIdx.y;
  if (w >= W || h >= H) return;
  const float z = depth[(size_t)h * W + w];
  float o[3] = {0.f, 0.f, 0.f};
  if (!(z < 0.001f)) {
    o[0] = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)w, cx), (double)z), fx);
    o[1] = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)h, cy), (double)z), fy);
    o[2] = z;
  }
  size_t p = ((size_t)h * W + w) * 3;
  xyz[p] = o[0];
  xyz[p + 1] = o[1];
  xyz[p + 2] = o[2];
}

// Statistics behind FoundationPose.guess_translation / the "valid too small" test of register() (src/estimater.py:137-156,
// 173-177), one workgroup: bounding box of mask > 0, count of mask > 0, count of usable pixels (mask > 0 and depth >= min_depth)
// and np.median of the usable depths (exact: bitwise radix select on the float pattern, two order statistics for an even
// count, averaged in float32 like numpy's mean of the two middle float32 values).
// out: [0] cmin [1] cmax [2] rmin [3] rmax [4] n_mask [5] n_usable, median
__global__ __launch_bounds__(1024) void mask_depth_stats_kernel(const float *__restrict__ depth, const unsigned char *__restrict__ mask, int H,
                                                                int W, float min_depth, int *__restrict__ out, float *__restrict__ median) {
  __shared__ int s_red[6];
  const int n = H * W, tid = threadIdx.x;
  if (tid == 0) {
    s_red[0] = 0x7fffffff; s_red[1] = -1; s_red[2] = 0x7fffffff; s_red[3] = -1; s_red[4] = 0; s_red[5] = 0;
  }
  __syncthreads();
  int cmin = 0x7fffffff, cmax = -1, rmin = 0x7fffffff, rmax = -1, nm = 0, nu = 0;
  for (int i = tid; i < n; i += blockDim.x) {
    if (mask[i]) {
      const int r = i / W, c = i - r * W;
      cmin = min(cmin, c); cmax = max(cmax, c); rmin = min(rmin, r); rmax = max(rmax, r);
      ++nm;
      if (depth[i] >= min_depth) ++nu;
    }
  }
  atomicMin(&s_red[0], cmin); atomicMax(&s_red[1], cmax); atomicMin(&s_red[2], rmin); atomicMax(&s_red[3], rmax);
  atomicAdd(&s_red[4], nm); atomicAdd(&s_red[5], nu);
  __syncthreads();
  const int n_us = s_red[5];
  if (tid < 6) out[tid] = s_red[tid];
  if (n_us == 0) {
    if (tid == 0) *median = 0.f;
    return;
  }
  // usable depths are >= min_depth > 0: their bit patterns order like the values.  8-bit radix select, both order
  // statistics at once: 4 passes over the image, a 256-bin LDS histogram per statistic and pass.
  __shared__ unsigned s_hist[2][256];
  __shared__ unsigned s_prefix[2], s_rank[2];
  if (tid < 2) {
    s_prefix[tid] = 0;
    s_rank[tid] = tid == 0 ? (unsigned)((n_us - 1) / 2) : (unsigned)(n_us / 2);
  }
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 512; i += blockDim.x) s_hist[i >> 8][i & 255] = 0;
    __syncthreads();
    const unsigned p0 = s_prefix[0], p1 = s_prefix[1];
    const unsigned hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < n; i += blockDim.x) {
      if (mask[i] && depth[i] >= min_depth) {
        const unsigned u = __float_as_uint(depth[i]);
        const unsigned d = (u >> shift) & 255u;
        if ((u & hi_mask) == (p0 & hi_mask)) atomicAdd(&s_hist[0][d], 1u);
        if ((u & hi_mask) == (p1 & hi_mask)) atomicAdd(&s_hist[1][d], 1u);
      }
    }
    __syncthreads();
    if (tid < 2) {
      unsigned r = s_rank[tid], cum = 0;
      int d = 0;
      for (; d < 256; ++d) {
        const unsigned c = s_hist[tid][d];
        if (r < cum + c) break;
        cum += c;
      }
      s_rank[tid] = r - cum;
      s_prefix[tid] |= (unsigned)d << shift;
    }
    __syncthreads();
  }
  float vals[2] = {__uint_as_float(s_prefix[0]), __uint_as_float(s_prefix[1])};
  if (tid == 0) *median = __fmul_rn(__fadd_rn(vals[0], vals[1]), 0.5f);
}

int launch_depth2xyz_f64(const float *d, int H, int W, const double *K, float *xyz, hipStream_t s) {
  hipLaunchKernelGGL(depth2xyz_f64_kernel, grid2d(H, W), dim3(32, 8), 0, s, d, H, W, K[0], K[4], K[2], K[5], xyz);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
int launch_mask_depth_stats(const float *d, const unsigned char *mask, int H, int W, float min_depth, int *out6, float *median, hipStream_t s) {
  hipLaunchKernelGGL(mask_depth_stats_kernel, dim3(1), dim3(1024), 0, s, d, mask, H, W, min_depth, out6, median);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
