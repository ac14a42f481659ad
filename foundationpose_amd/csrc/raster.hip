// Batched pose-hypothesis rasteriser for gfx950 (replaces nvdiffrast in src/Utils.py:133-219) and the
// crop-window transform (src/Utils.py:577-621).
//
// Regime: a 160x160 crop of a ~16k-face mesh => triangles are 1-3 pixels ("micro-polygons"), and about half of them cover
// no pixel centre at all.  Three launches per batch of hypotheses:
//   1. xform_vertices_kernel: every (hypothesis, vertex) transformed ONCE (clip matrix in float64, 1/16-px snapping) -> 16 B;
//   2. bin_faces_kernel: one thread per (hypothesis, face) computes the face's pixel range with the rasteriser's own integer
//      arithmetic, drops faces that cover no pixel centre of the crop (and zero-area ones), and appends the face id to the
//      list of every horizontal STRIP of the crop its rows touch (wave-aggregated atomic append; the order inside a list is
//      not deterministic and does not matter: visibility below is a minimum over (depth, face id) keys);
//   3. render_kernel: one workgroup = one hypothesis x one strip (40 rows x 160 px x 8 B = 50 KiB of LDS at >= 64 hypotheses,
//      thinner strips for fewer; several workgroups per CU).  The strip's FRAMEBUFFER lives in LDS; the lanes walk the strip's
//      face list (whole triangles per lane: 64-bit / 32-bit integer edge functions, top-left rule) and resolve visibility with
//      one ds_min_u64 per covered pixel on the packed key (ordered z/w : 32 | face id : 32) - nearest wins, ties go to the
//      lower face id, independent of lane scheduling and list order => bit-reproducible.
//      (Until round 3 a workgroup walked ALL faces of the mesh for each of its two 80-row strips: 200 us per 252-hypothesis
//      launch, 65 us of a workgroup's 110 in the triangle loop.)
//   A second pass resolves each pixel: perspective-correct barycentrics, attribute interpolation,
//   shading, and either fp32 channels-last maps (API parity with nvdiffrast_render) or the fused
//   network-ready fp16 NHWC8 tensor (rgb, (xyz-t)*2/diam with invalid masking; h5_dataset.py:92-99),
//   written as one coalesced 16-byte store per pixel.
// Triangles that straddle the camera plane are rasterised in homogeneous coordinates (below).
// The arithmetic is mirrored 1:1 by oracle/raster_c.c.
#include "common.h"

#define RB_THREADS 512
#ifndef RB_MINW
#define RB_MINW 4     // waves per SIMD the register budget is held to: two 8-wave workgroups per CU
#endif
#ifndef RB_FLY
#define RB_FLY 2      // triangles in flight per thread in the list walk (4 need 159 VGPRs: one workgroup per CU instead of two)
#endif

__device__ __forceinline__ unsigned ordered_key(float z) {
  unsigned b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct Vtx {
  int X, Y;
  float w, zn;
  bool ok;
};

__device__ __forceinline__ Vtx xform_vertex(const float *__restrict__ pos, int v, const float *M, float hw, float hh) {
  float px = pos[v * 3], py = pos[v * 3 + 1], pz = pos[v * 3 + 2];
  float c0 = fmaf(M[0], px, fmaf(M[1], py, fmaf(M[2], pz, M[3])));
  float c1 = fmaf(M[4], px, fmaf(M[5], py, fmaf(M[6], pz, M[7])));
  float c2 = fmaf(M[8], px, fmaf(M[9], py, fmaf(M[10], pz, M[11])));
  float c3 = fmaf(M[12], px, fmaf(M[13], py, fmaf(M[14], pz, M[15])));
  Vtx o;
  o.w = c3;
  o.ok = false;
  o.X = o.Y = 0;
  o.zn = 0.f;
  if (c3 > 0.f) {
    float xn = c0 / c3, yn = c1 / c3;
    o.zn = c2 / c3;
    float sx = fmaf(xn, hw, hw), sy = fmaf(yn, hh, hh);
    if (fabsf(sx) <= 1e6f && fabsf(sy) <= 1e6f) {
      o.X = (int)rintf(sx * 16.f);
      o.Y = (int)rintf(sy * 16.f);
      o.ok = true;
    }
  }
  return o;
}

// ---- triangles that straddle the camera plane (a vertex with w <= 0 or projected out of range, another in front) ----------
// nvdiffrast clips such triangles against the frustum; dropping them (round 1) leaves holes in an object that reaches behind
// the camera.  They are rasterised in 2-D homogeneous coordinates instead (no clipping, no new vertices): with the
// pixel-homogeneous vertices v_k = (cx hw + cw hw, cy hh + cw hh, cw) the barycentric weights of the surface point seen at
// pixel centre p = (i + .5, j + .5, 1) are b_k = l_k / (l_0 + l_1 + l_2), l_k = sign(D) n_k . p, n_0 = v_1 x v_2 (cyclic),
// D = v_0 . n_0; the point is on the triangle and in front of the camera iff every l_k > 0.  z/w = sum l_k cz_k / sum l_k cw_k
// (the same projective depth the screen-affine interpolation gives) must lie in [-1, 1]: that IS the near / far clip, per
// pixel.  All float32, fixed operation order, no contraction - mirrored by oracle/raster_c.c.
struct ClipTri {
  float n[3][3];     // sign-normalised edge normals
  float cz[3], cw[3];
  bool valid;
};

__device__ __forceinline__ void clip_coords(const float *__restrict__ pos, int v, const float *M, float *c) {
  const float px = pos[v * 3], py = pos[v * 3 + 1], pz = pos[v * 3 + 2];
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = fmaf(M[r * 4 + 0], px, fmaf(M[r * 4 + 1], py, fmaf(M[r * 4 + 2], pz, M[r * 4 + 3])));
}

__device__ __forceinline__ ClipTri clip_setup(const float c[3][4], float hw, float hh) {
  ClipTri T;
  float v[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v[k][0] = __fadd_rn(__fmul_rn(c[k][0], hw), __fmul_rn(c[k][3], hw));
    v[k][1] = __fadd_rn(__fmul_rn(c[k][1], hh), __fmul_rn(c[k][3], hh));
    v[k][2] = c[k][3];
    T.cz[k] = c[k][2];
    T.cw[k] = c[k][3];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float *a = v[(k + 1) % 3], *b = v[(k + 2) % 3];
    T.n[k][0] = __fsub_rn(__fmul_rn(a[1], b[2]), __fmul_rn(a[2], b[1]));
    T.n[k][1] = __fsub_rn(__fmul_rn(a[2], b[0]), __fmul_rn(a[0], b[2]));
    T.n[k][2] = __fsub_rn(__fmul_rn(a[0], b[1]), __fmul_rn(a[1], b[0]));
  }
  const float D = __fadd_rn(__fadd_rn(__fmul_rn(v[0][0], T.n[0][0]), __fmul_rn(v[0][1], T.n[0][1])), __fmul_rn(v[0][2], T.n[0][2]));
  T.valid = D != 0.f && isfinite(D);
  const float sg = D > 0.f ? 1.f : -1.f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int e = 0; e < 3; ++e) T.n[k][e] = __fmul_rn(T.n[k][e], sg);
  return T;
}

// weights l_k at pixel (i, j) and z/w; false if the pixel does not see the triangle
__device__ __forceinline__ bool clip_eval(const ClipTri &T, int i, int j, float *l, float *zp) {
  const float Px = (float)i + 0.5f, Py = (float)j + 0.5f;
#pragma unroll
  for (int k = 0; k < 3; ++k) l[k] = __fadd_rn(__fadd_rn(__fmul_rn(T.n[k][0], Px), __fmul_rn(T.n[k][1], Py)), T.n[k][2]);
  if (!(l[0] > 0.f && l[1] > 0.f && l[2] > 0.f)) return false;
  const float num = __fadd_rn(__fadd_rn(__fmul_rn(l[0], T.cz[0]), __fmul_rn(l[1], T.cz[1])), __fmul_rn(l[2], T.cz[2]));
  const float den = __fadd_rn(__fadd_rn(__fmul_rn(l[0], T.cw[0]), __fmul_rn(l[1], T.cw[1])), __fmul_rn(l[2], T.cw[2]));
  if (!(den > 0.f)) return false;
  *zp = __fdiv_rn(num, den);
  return *zp >= -1.f && *zp <= 1.f;
}

// clip matrix of hypothesis b in float64, rounded once (oracle/render.py: clip_matrices; src/Utils.py:155-181); one thread
__device__ void clip_matrix(const RenderArgs &a, int b, const float *pose, float *sM) {
  const double W = a.W, H = a.H, zn = 0.001, zf = 100.0;
  double P[16] = {2 * a.K[0] / W, -2 * a.K[1] / W, (-2 * a.K[2] + W) / W, 0,
                  0, 2 * a.K[4] / H, (2 * a.K[5] - H) / H, 0,
                  0, 0, -(zf + zn) / (zf - zn), -2 * (zf * zn) / (zf - zn),
                  0, 0, -1, 0};
  if (a.has_proj)                       // projection_mat given by the caller (src/Utils.py:159-161)
    for (int i = 0; i < 16; ++i) P[i] = a.proj[i];
  double G[16];  // glcam_in_cvcam @ ob_in_cam : negate rows 1,2
  for (int c = 0; c < 4; ++c) {
    G[c] = pose[c];
    G[4 + c] = -(double)pose[4 + c];
    G[8 + c] = -(double)pose[8 + c];
    G[12 + c] = pose[12 + c];
  }
  double Mx[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += P[r * 4 + k] * G[k * 4 + c];
      Mx[r * 4 + c] = s;
    }
  if (a.bbox2d) {
    const float *bb = a.bbox2d + (size_t)b * 4;
    double l = bb[0], t = H - (double)bb[1], r = bb[2], bt = H - (double)bb[3];
    double t00 = W / (r - l), t11 = H / (t - bt), t30 = (W - r - l) / (r - l), t31 = (H - t - bt) / (t - bt);
    for (int c = 0; c < 4; ++c) {
      double r3 = Mx[12 + c];
      Mx[c] = t00 * Mx[c] + t30 * r3;
      Mx[4 + c] = t11 * Mx[4 + c] + t31 * r3;
    }
  }
  for (int i = 0; i < 16; ++i) sM[i] = (float)Mx[i];
}

// Vertex pre-pass for the fused network path: every hypothesis' vertices transformed ONCE (the triangle loop of render_kernel
// otherwise re-transforms a vertex for each of its ~6 triangles, in each strip, behind two dependent global round trips).
// One int4 per vertex: X (INT_MIN = behind the camera / off range), Y, bits of z/w, bits of w - exactly xform_vertex's values.
__global__ __launch_bounds__(256) void xform_vertices_kernel(RenderArgs a, int4 *__restrict__ vout, int *__restrict__ count, int S) {
  __shared__ float sM[16];
  const int b = blockIdx.y;
  if (threadIdx.x == 0) clip_matrix(a, b, a.poses + (size_t)b * 16, sM);
  if (blockIdx.x == 0)                  // the strip lists of this hypothesis start empty (bin_faces_kernel runs behind this launch)
    for (int q = threadIdx.x; q < S; q += 256) count[(size_t)b * S + q] = 0;
  __syncthreads();
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= a.mesh.V) return;
  float M[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) M[i] = sM[i];
  const Vtx o = xform_vertex(a.mesh.pos, v, M, 0.5f * (float)a.Wo, 0.5f * (float)a.Ho);
  vout[(size_t)b * a.mesh.V + v] = make_int4(o.ok ? o.X : (int)0x80000000, o.Y, __float_as_int(o.zn), __float_as_int(o.w));
}


// Face lists per (hypothesis, strip).  A face is dropped here exactly when raster_tri() below would return without touching a
// pixel for EVERY strip (pixel range empty after clamping to the crop, or zero area) - the same integers, so the images are
// bit-identical to the unbinned walk; a face that straddles the camera plane cannot be bounded cheaply and goes to every strip.
__global__ __launch_bounds__(256) void bin_faces_kernel(RenderArgs a, const int4 *__restrict__ vbuf, int *__restrict__ count,
                                                        int *__restrict__ list, int S, int strip_rows) {
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const MeshDev &m = a.mesh;
  int s0 = 1, s1 = 0;                                   // strips [s0, s1] (empty)
  if (t < m.F) {
    const int4 *vb = vbuf + (size_t)b * m.V;
    const int4 q0 = vb[m.faces[t * 3]], q1 = vb[m.faces[t * 3 + 1]], q2 = vb[m.faces[t * 3 + 2]];
    const bool ok0 = q0.x != (int)0x80000000, ok1 = q1.x != (int)0x80000000, ok2 = q2.x != (int)0x80000000;
    if (!(ok0 && ok1 && ok2)) {
      if (__int_as_float(q0.w) > 0.f || __int_as_float(q1.w) > 0.f || __int_as_float(q2.w) > 0.f) s0 = 0, s1 = S - 1;
    } else {
      const long long X0 = q0.x, Y0 = q0.y, X1 = q1.x, Y1 = q1.y, X2 = q2.x, Y2 = q2.y;
      const long long area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
      const int xmin = min(q0.x, min(q1.x, q2.x)), xmax = max(q0.x, max(q1.x, q2.x));
      const int ymin = min(q0.y, min(q1.y, q2.y)), ymax = max(q0.y, max(q1.y, q2.y));
      const int ia = max((xmin - 8 + 15) >> 4, 0), ib = min((xmax - 8) >> 4, a.Wo - 1);
      const int ja = max((ymin - 8 + 15) >> 4, 0), jb = min((ymax - 8) >> 4, a.Ho - 1);
      if (area != 0 && ia <= ib && ja <= jb) s0 = ja / strip_rows, s1 = jb / strip_rows;
    }
  }
  if (S <= 16) {                                        // one atomic per wave and strip
    for (int s = 0; s < S; ++s) {
      const bool in = s >= s0 && s <= s1;
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(in);
      if (mk == 0) continue;
      int base = 0;
      if (lane == __builtin_ctzll(mk)) base = atomicAdd(&count[(size_t)b * S + s], __builtin_popcountll(mk));
      base = __shfl(base, __builtin_ctzll(mk));
      if (in) list[((size_t)b * S + s) * m.F + base + __builtin_popcountll(mk & ((1ull << lane) - 1))] = t;
    }
  } else {
    for (int s = s0; s <= s1; ++s) list[((size_t)b * S + s) * m.F + atomicAdd(&count[(size_t)b * S + s], 1)] = t;
  }
}

template <int MODE>
__global__ __launch_bounds__(RB_THREADS, RB_MINW) void render_kernel(RenderArgs a, int strip_rows, int n_strips, const int4 *__restrict__ vbuf,
                                                                 const int *__restrict__ count, const int *__restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long zbuf[];
  __shared__ float sM[16];
  __shared__ float sP[12];
  const int L = xcd_remap(blockIdx.x, gridDim.x);          // the strips of one hypothesis share an XCD's L2 (vertices, lists)
  const int b = L / n_strips, strip = L % n_strips;
  const int Ho = a.Ho, Wo = a.Wo;
  const int row0 = strip * strip_rows;  // GL (bottom-up) rows [row0, row1)
  const int row1 = min(Ho, row0 + strip_rows);
  const int npix = (row1 - row0) * Wo;
  const float *pose = a.poses + (size_t)b * 16;

  if (threadIdx.x == 0) {
    clip_matrix(a, b, pose, sM);
    for (int i = 0; i < 12; ++i) sP[i] = pose[i];
  }
  for (int i = threadIdx.x; i < npix; i += RB_THREADS) zbuf[i] = ~0ull;
  __syncthreads();

  // (the clip matrix stays in LDS: only triangles that straddle the camera plane read it)
  const float *M = sM;
  const float hw = 0.5f * (float)Wo, hh = 0.5f * (float)Ho;
  const MeshDev &m = a.mesh;
  const int4 *vb = vbuf + (size_t)b * m.V;
  auto vertex = [&](int i) -> Vtx {          // the pre-pass' record: exactly xform_vertex's values
    const int4 q = vb[i];
    Vtx o;
    o.X = q.x;
    o.Y = q.y;
    o.zn = __int_as_float(q.z);
    o.w = __int_as_float(q.w);
    o.ok = q.x != (int)0x80000000;
    return o;
  };

  // ---- pass 1: walk the strip's face list, resolve visibility in LDS.  Four triangles per thread are in flight at a time:
  // their list, index and vertex loads (three dependent memory round trips) are issued together; coverage and keys do not
  // depend on the order.
  auto raster_tri = [&](const int t, const int *fi3, const Vtx &v0, const Vtx &v1, const Vtx &v2) __attribute__((always_inline)) {
    if (!(v0.ok && v1.ok && v2.ok)) {
      if (!(v0.w > 0.f || v1.w > 0.f || v2.w > 0.f)) return;      // entirely behind the camera
      float c[3][4];
#pragma unroll
      for (int k = 0; k < 3; ++k) clip_coords(m.pos, fi3[k], M, c[k]);
      const ClipTri T = clip_setup(c, hw, hh);
      if (!T.valid) return;
      for (int j = row0; j < row1; ++j)
        for (int i = 0; i < Wo; ++i) {
          float l[3], zp;
          if (!clip_eval(T, i, j, l, &zp)) continue;
          const unsigned long long key = ((unsigned long long)ordered_key(zp) << 32) | (unsigned)t;
          atomicMin(&zbuf[(j - row0) * Wo + i], key);
        }
      return;
    }
    // Triangles whose snapped coordinates stay within +-1024 px (all but the ones far outside the crop) take the same
    // integer edge functions in 32-bit arithmetic: |X|,|Y| < 2^14 -> differences < 2^15, products < 2^30, sums < 2^31.
    // The values are the same integers as in the 64-bit path, so coverage, barycentrics and depth keys are bit-identical.
    const int amax = max(max(max(abs(v0.X), abs(v0.Y)), max(abs(v1.X), abs(v1.Y))), max(abs(v2.X), abs(v2.Y)));
    if (amax < 16384) {
      const int X0 = v0.X, Y0 = v0.Y, X1 = v1.X, Y1 = v1.Y, X2 = v2.X, Y2 = v2.Y;
      int area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
      if (area == 0) return;
      const int sg = area > 0 ? 1 : -1;
      area *= sg;
      const int xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
      const int ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
      int ia = (xmin - 8 + 15) >> 4, ib = (xmax - 8) >> 4, ja = (ymin - 8 + 15) >> 4, jb = (ymax - 8) >> 4;
      ia = max(ia, 0);
      ja = max(ja, row0);
      ib = min(ib, Wo - 1);
      jb = min(jb, row1 - 1);
      if (ia > ib || ja > jb) return;
      const int dx0 = sg * (X2 - X1), dy0 = sg * (Y2 - Y1);
      const int dx1 = sg * (X0 - X2), dy1 = sg * (Y0 - Y2);
      const int dx2 = sg * (X1 - X0), dy2 = sg * (Y1 - Y0);
      const bool tl0 = (dy0 > 0) || (dy0 == 0 && dx0 < 0);
      const bool tl1 = (dy1 > 0) || (dy1 == 0 && dx1 < 0);
      const bool tl2 = (dy2 > 0) || (dy2 == 0 && dx2 < 0);
      const float fa = (float)area;
      for (int j = ja; j <= jb; ++j) {
        const int Py = 16 * j + 8;
        for (int i = ia; i <= ib; ++i) {
          const int Px = 16 * i + 8;
          const int e0 = dx0 * (Py - Y1) - dy0 * (Px - X1);
          const int e1 = dx1 * (Py - Y2) - dy1 * (Px - X2);
          const int e2 = dx2 * (Py - Y0) - dy2 * (Px - X0);
          if (!((e0 > 0 || (e0 == 0 && tl0)) && (e1 > 0 || (e1 == 0 && tl1)) && (e2 > 0 || (e2 == 0 && tl2)))) continue;
          const float b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
          const float zp = fmaf(b2, v2.zn, fmaf(b1, v1.zn, b0 * v0.zn));
          if (!(zp >= -1.f && zp <= 1.f)) continue;
          const unsigned long long key = ((unsigned long long)ordered_key(zp) << 32) | (unsigned)t;
          atomicMin(&zbuf[(j - row0) * Wo + i], key);
        }
      }
      return;
    }
    long long X0 = v0.X, Y0 = v0.Y, X1 = v1.X, Y1 = v1.Y, X2 = v2.X, Y2 = v2.Y;
    long long area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
    if (area == 0) return;
    long long sg = area > 0 ? 1 : -1;
    area *= sg;
    long long xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
    long long ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
    long long ia = (xmin - 8 + 15) >> 4, ib = (xmax - 8) >> 4, ja = (ymin - 8 + 15) >> 4, jb = (ymax - 8) >> 4;
    ia = max(ia, 0ll);
    ja = max(ja, (long long)row0);
    ib = min(ib, (long long)Wo - 1);
    jb = min(jb, (long long)row1 - 1);
    if (ia > ib || ja > jb) return;
    long long dx0 = sg * (X2 - X1), dy0 = sg * (Y2 - Y1);
    long long dx1 = sg * (X0 - X2), dy1 = sg * (Y0 - Y2);
    long long dx2 = sg * (X1 - X0), dy2 = sg * (Y1 - Y0);
    bool tl0 = (dy0 > 0) || (dy0 == 0 && dx0 < 0);
    bool tl1 = (dy1 > 0) || (dy1 == 0 && dx1 < 0);
    bool tl2 = (dy2 > 0) || (dy2 == 0 && dx2 < 0);
    float fa = (float)area;
    for (long long j = ja; j <= jb; ++j) {
      long long Py = 16 * j + 8;
      for (long long i = ia; i <= ib; ++i) {
        long long Px = 16 * i + 8;
        long long e0 = dx0 * (Py - Y1) - dy0 * (Px - X1);
        long long e1 = dx1 * (Py - Y2) - dy1 * (Px - X2);
        long long e2 = dx2 * (Py - Y0) - dy2 * (Px - X0);
        if (!((e0 > 0 || (e0 == 0 && tl0)) && (e1 > 0 || (e1 == 0 && tl1)) && (e2 > 0 || (e2 == 0 && tl2)))) continue;
        float b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
        float zp = fmaf(b2, v2.zn, fmaf(b1, v1.zn, b0 * v0.zn));
        if (!(zp >= -1.f && zp <= 1.f)) continue;
        unsigned long long key = ((unsigned long long)ordered_key(zp) << 32) | (unsigned)t;
        atomicMin(&zbuf[(int)(j - row0) * Wo + (int)i], key);
      }
    }
  };
  const int nlist = count[(size_t)b * n_strips + strip];
  const int *mylist = list + ((size_t)b * n_strips + strip) * m.F;
  for (int e0 = threadIdx.x; e0 < nlist; e0 += RB_FLY * RB_THREADS) {
    int ft[RB_FLY], fi[RB_FLY][3];
    Vtx fv[RB_FLY][3];
#pragma unroll
    for (int u = 0; u < RB_FLY; ++u) ft[u] = mylist[min(e0 + u * RB_THREADS, nlist - 1)];
#pragma unroll
    for (int u = 0; u < RB_FLY; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) fi[u][k] = m.faces[ft[u] * 3 + k];
#pragma unroll
    for (int u = 0; u < RB_FLY; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) fv[u][k] = vertex(fi[u][k]);
#pragma unroll
    for (int u = 0; u < RB_FLY; ++u)
      if (e0 + u * RB_THREADS < nlist) raster_tri(ft[u], fi[u], fv[u][0], fv[u][1], fv[u][2]);
  }
  __syncthreads();

  // ---- pass 2: per-pixel resolve + attribute interpolation + fused epilogue ----
  const float P0 = sP[0], P1 = sP[1], P2 = sP[2], P3 = sP[3], P4 = sP[4], P5 = sP[5], P6 = sP[6], P7 = sP[7], P8 = sP[8],
              P9 = sP[9], P10 = sP[10], P11 = sP[11];
  for (int p = threadIdx.x; p < npix; p += RB_THREADS) {
    const int jl = p / Wo, i = p - jl * Wo;
    const int j = row0 + jl;
    const int jo = Ho - 1 - j;  // flipped output row (src/Utils.py:216-218)
    const size_t o = ((size_t)b * Ho + jo) * Wo + i;
    unsigned long long key = zbuf[p];
    float col[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, p3[3] = {0, 0, 0};
    if (key != ~0ull) {
      int t = (int)(unsigned)(key & 0xffffffffull);
      int i0 = m.faces[t * 3], i1 = m.faces[t * 3 + 1], i2 = m.faces[t * 3 + 2];
      Vtx v0 = vertex(i0), v1 = vertex(i1), v2 = vertex(i2);
      float fa, b0, b1, b2;
      float u, v, w2;
      const bool clipped = !(v0.ok && v1.ok && v2.ok);
      const int amax = clipped ? 0 : max(max(max(abs(v0.X), abs(v0.Y)), max(abs(v1.X), abs(v1.Y))), max(abs(v2.X), abs(v2.Y)));
      if (clipped) {                              // straddles the camera plane: weights from the homogeneous edge functions
        float c[3][4];
        const int id3[3] = {i0, i1, i2};
#pragma unroll
        for (int k = 0; k < 3; ++k) clip_coords(m.pos, id3[k], M, c[k]);
        const ClipTri T = clip_setup(c, hw, hh);
        float l[3], zp;
        (void)clip_eval(T, i, j, l, &zp);
        const float ls = __fadd_rn(__fadd_rn(l[0], l[1]), l[2]);
        u = __fdiv_rn(l[0], ls), v = __fdiv_rn(l[1], ls), w2 = (1.f - u) - v;
        fa = b0 = b1 = b2 = 0.f;
      } else if (amax < 16384) {                  // the same integers in 32-bit arithmetic (see pass 1)
        const int X0 = v0.X, Y0 = v0.Y, X1 = v1.X, Y1 = v1.Y, X2 = v2.X, Y2 = v2.Y;
        int area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
        const int sg = area > 0 ? 1 : -1;
        area *= sg;
        const int Px = 16 * i + 8, Py = 16 * j + 8;
        const int e0 = sg * ((X2 - X1) * (Py - Y1) - (Y2 - Y1) * (Px - X1));
        const int e1 = sg * ((X0 - X2) * (Py - Y2) - (Y0 - Y2) * (Px - X2));
        const int e2 = sg * ((X1 - X0) * (Py - Y0) - (Y1 - Y0) * (Px - X0));
        fa = (float)area;
        b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
      } else {
        long long X0 = v0.X, Y0 = v0.Y, X1 = v1.X, Y1 = v1.Y, X2 = v2.X, Y2 = v2.Y;
        long long area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
        long long sg = area > 0 ? 1 : -1;
        area *= sg;
        long long Px = 16ll * i + 8, Py = 16ll * j + 8;
        long long e0 = sg * ((X2 - X1) * (Py - Y1) - (Y2 - Y1) * (Px - X1));
        long long e1 = sg * ((X0 - X2) * (Py - Y2) - (Y0 - Y2) * (Px - X2));
        long long e2 = sg * ((X1 - X0) * (Py - Y0) - (Y1 - Y0) * (Px - X0));
        fa = (float)area;
        b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
      }
      if (!clipped) {
        float q0 = b0 / v0.w, q1 = b1 / v1.w, q2 = b2 / v2.w;
        float qs = (q0 + q1) + q2;
        u = q0 / qs, v = q1 / qs, w2 = (1.f - u) - v;
      }
      const int idx[3] = {i0, i1, i2};
      float pc[3][3], nc[3][3], dv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float px = m.pos[idx[k] * 3], py = m.pos[idx[k] * 3 + 1], pz = m.pos[idx[k] * 3 + 2];
        pc[k][0] = fmaf(P0, px, fmaf(P1, py, fmaf(P2, pz, P3)));
        pc[k][1] = fmaf(P4, px, fmaf(P5, py, fmaf(P6, pz, P7)));
        pc[k][2] = fmaf(P8, px, fmaf(P9, py, fmaf(P10, pz, P11)));
        float nx = m.vnormals[idx[k] * 3], ny = m.vnormals[idx[k] * 3 + 1], nz = m.vnormals[idx[k] * 3 + 2];
        nc[k][0] = fmaf(P0, nx, fmaf(P1, ny, P2 * nz));
        nc[k][1] = fmaf(P4, nx, fmaf(P5, ny, P6 * nz));
        nc[k][2] = fmaf(P8, nx, fmaf(P9, ny, P10 * nz));
        float nn = sqrtf(fmaf(nc[k][0], nc[k][0], fmaf(nc[k][1], nc[k][1], nc[k][2] * nc[k][2])));
        nn = nn > 1e-12f ? nn : 1e-12f;
        if (a.light_mode == 0) {
          dv[k] = fminf(fmaxf(-(nc[k][2] / nn), 0.f), 1.f);
        } else {                        // src/Utils.py:200-205: normalize(vnormals_cam) . normalize(-light_dir | light_pos - pts_cam), clipped to [0,1]
          float L[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) L[c] = a.light_mode == 1 ? a.light_vec[c] : a.light_vec[c] - pc[k][c];
          float ln = sqrtf(fmaf(L[0], L[0], fmaf(L[1], L[1], L[2] * L[2])));
          ln = ln > 1e-12f ? ln : 1e-12f;
          const float dt = fmaf(nc[k][0] / nn, L[0] / ln, fmaf(nc[k][1] / nn, L[1] / ln, (nc[k][2] / nn) * (L[2] / ln)));
          dv[k] = fminf(fmaxf(dt, 0.f), 1.f);
        }
      }
      float base[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        p3[c] = fmaf(u, pc[0][c], fmaf(v, pc[1][c], w2 * pc[2][c]));
        nrm[c] = fmaf(u, nc[0][c], fmaf(v, nc[1][c], w2 * nc[2][c]));
      }
      if (m.tex) {
        int a0 = m.uv_idx[t * 3], a1 = m.uv_idx[t * 3 + 1], a2 = m.uv_idx[t * 3 + 2];
        float tu = fmaf(u, m.uv[a0 * 2], fmaf(v, m.uv[a1 * 2], w2 * m.uv[a2 * 2]));
        float tv = fmaf(u, m.uv[a0 * 2 + 1], fmaf(v, m.uv[a1 * 2 + 1], w2 * m.uv[a2 * 2 + 1]));
        float x = tu * (float)m.texW - 0.5f, y = tv * (float)m.texH - 0.5f;
        float fx0 = floorf(x), fy0 = floorf(y);
        float fx = x - fx0, fy = y - fy0;
        int x0 = (int)fx0 % m.texW;
        if (x0 < 0) x0 += m.texW;
        int y0 = (int)fy0 % m.texH;
        if (y0 < 0) y0 += m.texH;
        int x1 = (x0 + 1) % m.texW, y1 = (y0 + 1) % m.texH;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float t00 = m.tex[(y0 * m.texW + x0) * 3 + c], t10 = m.tex[(y0 * m.texW + x1) * 3 + c];
          float t01 = m.tex[(y1 * m.texW + x0) * 3 + c], t11 = m.tex[(y1 * m.texW + x1) * 3 + c];
          float ta = __fadd_rn(t00, __fmul_rn(fx, (t10 - t00)));
          float tb = __fadd_rn(t01, __fmul_rn(fx, (t11 - t01)));
          base[c] = __fadd_rn(ta, __fmul_rn(fy, (tb - ta)));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          base[c] = fmaf(u, m.vcolor[i0 * 3 + c], fmaf(v, m.vcolor[i1 * 3 + c], w2 * m.vcolor[i2 * 3 + c]));
      }
      if (a.use_light) {
        float d = fmaf(u, dv[0], fmaf(v, dv[1], w2 * dv[2]));
#pragma unroll
        for (int c = 0; c < 3; ++c)
          base[c] = __fadd_rn(__fmul_rn(base[c], a.w_ambient), __fmul_rn(__fmul_rn(d, a.has_light_color ? a.light_color[c] : base[c]), a.w_diffuse));
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = fminf(fmaxf(base[c], 0.f), 1.f);
      float nn = sqrtf(fmaf(nrm[0], nrm[0], fmaf(nrm[1], nrm[1], nrm[2] * nrm[2])));
      nn = nn > 1e-12f ? nn : 1e-12f;
#pragma unroll
      for (int c = 0; c < 3; ++c) nrm[c] = nrm[c] / nn;
    }
    if (MODE == 0) {
      if (a.color) {
        a.color[o * 3] = col[0];
        a.color[o * 3 + 1] = col[1];
        a.color[o * 3 + 2] = col[2];
      }
      if (a.normal) {
        a.normal[o * 3] = nrm[0];
        a.normal[o * 3 + 1] = nrm[1];
        a.normal[o * 3 + 2] = nrm[2];
      }
      if (a.xyz) {
        a.xyz[o * 3] = p3[0];
        a.xyz[o * 3 + 1] = p3[1];
        a.xyz[o * 3 + 2] = p3[2];
      }
      if (a.depth) a.depth[o] = p3[2];
    } else {
      // rgbAs = (color*255)/255 (predict_pose_refine.py:57 + h5_dataset.py:123); xyz transform h5_dataset.py:92-99 | 151-156
      float r[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) r[c] = __fdiv_rn(__fmul_rn(col[c], 255.f), 255.f);
      bool invalid = p3[2] < a.invalid_thres;
      float radius = __fdiv_rn(a.mesh_diameter, 2.f);
      float inv_r = __fdiv_rn(1.f, radius);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float x = __fsub_rn(p3[c], pose[c * 4 + 3]);
        if (a.normalize_xyz) {
          x = __fmul_rn(x, inv_r);
          if (invalid || fabsf(x) >= 2.f) x = 0.f;
        }
        r[3 + c] = x;
      }
      half8 hv;
#pragma unroll
      for (int c = 0; c < 6; ++c) hv[c] = (f16)r[c];
      hv[6] = (f16)0.f;
      hv[7] = (f16)0.f;
      *reinterpret_cast<half8 *>(a.net_out + o * 8) = hv;
    }
  }
}

// Strips per hypothesis and the scratch a launch needs: transformed vertices (16 B each), list counters, face lists (worst
// case: every face in every strip).
RenderPlan render_plan(int N, int V, int F, int Ho, int Wo, int num_cu) {
  RenderPlan p;
  const size_t max_lds = 150 * 1024;
  int rows_max = (int)(max_lds / ((size_t)Wo * 8));
  if (rows_max > Ho) rows_max = Ho;
  if (rows_max < 1) rows_max = 1;
  int S = (Ho + rows_max - 1) / rows_max;
  // 40-row strips of a 160-row crop (50 KB: three workgroups per CU); thinner ones while the launch is still under two
  // workgroups per CU - a workgroup's resolve pass shrinks with its strip, and the face lists keep the triangle pass from
  // being repeated per strip (a pixel's result does not depend on the strip it is in)
  while (S < 4 && Ho / (S * 2) >= 8) S *= 2;
  while ((size_t)N * S < (size_t)2 * num_cu && S < 16 && Ho / (S * 2) >= 8) S *= 2;
  p.S = S;
  p.strip_rows = (Ho + S - 1) / S;
  p.S = (Ho + p.strip_rows - 1) / p.strip_rows;
  p.vbuf_bytes = ((size_t)N * V * 16 + 255) & ~(size_t)255;
  p.count_bytes = ((size_t)N * p.S * 4 + 255) & ~(size_t)255;
  p.list_bytes = (size_t)N * p.S * F * 4;
  p.total = p.vbuf_bytes + p.count_bytes + p.list_bytes;
  return p;
}

int launch_render(fp_ctx *ctx, const RenderArgs &a, hipStream_t s) {
  FP_REQUIRE(a.N >= 0 && a.Ho > 0 && a.Wo > 0, "render: bad shape N=%d out=%dx%d", a.N, a.Ho, a.Wo);
  if (a.N == 0) return FP_OK;
  FP_REQUIRE((size_t)a.Wo * 8 <= 150 * 1024, "render: output width %d too large for one LDS strip", a.Wo);
  const RenderPlan pl = render_plan(a.N, a.mesh.V, a.mesh.F, a.Ho, a.Wo, ctx->num_cu);
  FP_REQUIRE(a.scratch && a.scratch_bytes >= pl.total, "render: scratch of %zu bytes needed, %zu given", pl.total, a.scratch_bytes);
  int4 *vbuf = (int4 *)a.scratch;
  int *count = (int *)((char *)a.scratch + pl.vbuf_bytes);
  int *list = (int *)((char *)a.scratch + pl.vbuf_bytes + pl.count_bytes);
  const size_t lds = (size_t)pl.strip_rows * a.Wo * 8;
  ProfScope ps(ctx, s, "render", 0);
  hipLaunchKernelGGL(xform_vertices_kernel, dim3((a.mesh.V + 255) / 256, a.N), dim3(256), 0, s, a, vbuf, count, pl.S);
  hipLaunchKernelGGL(bin_faces_kernel, dim3((a.mesh.F + 255) / 256, a.N), dim3(256), 0, s, a, (const int4 *)vbuf, count, list, pl.S, pl.strip_rows);
  FP_CHECK_HIP(hipGetLastError());
  auto go = [&](auto kern, bool *attr_set) -> int {
    if (!*attr_set) {              // once per instantiation, for the largest strip: not a stream operation
      FP_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      *attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * pl.S)), dim3(RB_THREADS), lds, s, a, pl.strip_rows, pl.S, (const int4 *)vbuf, (const int *)count,
                       (const int *)list);
    return FP_OK;
  };
  static bool set1 = false, set0 = false;
  if (a.net_out) FP_TRY(go(render_kernel<1>, &set1));
  else FP_TRY(go(render_kernel<0>, &set0));
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// ----------------------------------------------------------------------------------------------
// compute_crop_window_tf_batch(method='box_3d') (src/Utils.py:577-621) + bbox2d_ori
// (predict_pose_refine.py:44-45).  float32, left-to-right, no FMA contraction: mirrors
// oracle/geometry.py:compute_crop_window_tf_batch.
// ----------------------------------------------------------------------------------------------
__global__ void crop_window_tf_kernel(const float *__restrict__ poses, int N, float k00, float k01, float k02, float k10, float k11,
                                      float k12, float k20, float k21, float k22, float radius, float ow, float oh, float *tf,
                                      float *bbox) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
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

int launch_crop_window_tf(const float *poses, int N, const double *K, double crop_ratio, double diameter, int ow, int oh, float *tf,
                          float *bbox, hipStream_t s) {
  if (N == 0) return FP_OK;
  float radius = (float)(diameter * crop_ratio / 2.0);
  hipLaunchKernelGGL(crop_window_tf_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poses, N, (float)K[0], (float)K[1], (float)K[2],
                     (float)K[3], (float)K[4], (float)K[5], (float)K[6], (float)K[7], (float)K[8], radius, (float)ow, (float)oh, tf,
                     bbox);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
