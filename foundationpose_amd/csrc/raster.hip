// Batched pose-hypothesis rasteriser for gfx950 (replaces nvdiffrast in src/Utils.py:133-219) and the
// crop-window transform (src/Utils.py:577-621).
//
// Regime: a 160x160 crop of a ~16k-face mesh => triangles are 1-3 pixels ("micro-polygons"), about half of them cover no pixel
// centre at all, and a few (silhouette slivers, cap fans) span dozens.  What the rasteriser of rounds 1-2 spent its time on was not
// memory but LANE UTILISATION: every strip workgroup walked all faces, so its divergent per-triangle pixel loop ran for the few lanes
// of a wave whose face touched the strip, as long as the largest triangle among them needed (measured by switching the phases off one
// by one).  Three launches per batch of hypotheses now:
//   1. xform_vertices_kernel: every (hypothesis, vertex) ONCE -> three records: A (8 B) the snapped window position as two
//      int16 + z/w, for the triangle pass; B (32 B) camera-space position, Lambert term, vertex colour and w, everything the
//      shading interpolates; C (16 B) the full-precision position for the rare triangles A cannot describe;
//   2. classify_faces_kernel: one workgroup per hypothesis keeps its A records in LDS, walks the faces ONCE and appends each
//      face that can cover a pixel centre to the list of every horizontal STRIP of the crop it touches, by the number of candidate
//      pixels it has there: small (<= 4) and medium (<= 32) in one list from either end, large ones and those the 32-bit edge
//      functions cannot take (a vertex beyond +-1024 px, or on / behind the camera plane) in another.  The counters are in LDS;
//      the order inside a list is not deterministic and does not matter (visibility below is a minimum over keys);
//   3. render_kernel: one workgroup (16 waves) = one hypothesis x one strip (40 rows x 160 px x 8 B = 50 KiB at >= 64 hypotheses,
//      thinner strips for fewer).  The strip's FRAMEBUFFER and the hypothesis' A records (64 KiB for 8k vertices) live in LDS.  A
//      lane takes one small or medium triangle from the dense list (neighbours in a wave have similar pixel counts), reads its
//      vertices from LDS and rasterises it (32-bit integer edge functions, top-left rule), resolving visibility with one
//      ds_min_u64 per covered pixel on the packed key (ordered z/w : 32 | face id : 32) - nearest wins, ties go to the lower
//      face id, independent of lane scheduling and list order => bit-reproducible.  The second list is worked off one triangle
//      per WAVE, one candidate pixel per lane (64-bit edge functions / homogeneous form).  The covered pixels of the strip are
//      then compacted into a queue and resolved one per lane.
//   A second pass resolves each pixel: perspective-correct barycentrics, attribute interpolation,
//   shading, and either fp32 channels-last maps (API parity with nvdiffrast_render) or the fused
//   network-ready fp16 NHWC8 tensor (rgb, (xyz-t)*2/diam with invalid masking; h5_dataset.py:92-99),
//   written as one coalesced 16-byte store per pixel.
// Triangles that straddle the camera plane are rasterised in homogeneous coordinates (below).
// The arithmetic is mirrored 1:1 by oracle/raster_c.c.
#include "common.h"
#include "pose_math.h"
#include <cstdlib>

#define RB_THREADS 1024

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

// Vertex pre-pass: every hypothesis' vertices ONCE.
//   C (int4):   X (INT_MIN = behind the camera / off range), Y, bits of z/w, bits of w - exactly xform_vertex's values;
//   A (uint2):  X | Y << 16 as int16 where |X|, |Y| < 16384 (the range of the 32-bit edge functions) and bits of z/w; else RB_A_NONE
//               and bits of w;
//   B (2 x float4): camera-space position (pts_cam, src/Utils.py:168) and the vertex' Lambert term (src/Utils.py:200-206) - the
//               expressions the per-pixel resolve evaluated until round 3, moved here verbatim -, then colour and w.
#define RB_A_NONE 0x7fff7fffu
struct VtxRecords {
  int4 c;
  uint2 a;
  float4 b0, b1;
};
__device__ __forceinline__ VtxRecords vertex_records(const RenderArgs &a, int v, const float *sM, const float *sP) {
  float M[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) M[i] = sM[i];
  const MeshDev &m = a.mesh;
  const Vtx o = xform_vertex(m.pos, v, M, 0.5f * (float)a.Wo, 0.5f * (float)a.Ho);
  const float P0 = sP[0], P1 = sP[1], P2 = sP[2], P3 = sP[3], P4 = sP[4], P5 = sP[5], P6 = sP[6], P7 = sP[7], P8 = sP[8],
              P9 = sP[9], P10 = sP[10], P11 = sP[11];
  const float px = m.pos[v * 3], py = m.pos[v * 3 + 1], pz = m.pos[v * 3 + 2];
  float pc[3], nc[3];
  pc[0] = fmaf(P0, px, fmaf(P1, py, fmaf(P2, pz, P3)));
  pc[1] = fmaf(P4, px, fmaf(P5, py, fmaf(P6, pz, P7)));
  pc[2] = fmaf(P8, px, fmaf(P9, py, fmaf(P10, pz, P11)));
  const float nx = m.vnormals[v * 3], ny = m.vnormals[v * 3 + 1], nz = m.vnormals[v * 3 + 2];
  nc[0] = fmaf(P0, nx, fmaf(P1, ny, P2 * nz));
  nc[1] = fmaf(P4, nx, fmaf(P5, ny, P6 * nz));
  nc[2] = fmaf(P8, nx, fmaf(P9, ny, P10 * nz));
  float nn = sqrtf(fmaf(nc[0], nc[0], fmaf(nc[1], nc[1], nc[2] * nc[2])));
  nn = nn > 1e-12f ? nn : 1e-12f;
  float dv;
  if (a.light_mode == 0) {
    dv = fminf(fmaxf(-(nc[2] / nn), 0.f), 1.f);
  } else {                        // src/Utils.py:200-205: normalize(vnormals_cam) . normalize(-light_dir | light_pos - pts_cam), clipped to [0,1]
    float L[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) L[c] = a.light_mode == 1 ? a.light_vec[c] : a.light_vec[c] - pc[c];
    float ln = sqrtf(fmaf(L[0], L[0], fmaf(L[1], L[1], L[2] * L[2])));
    ln = ln > 1e-12f ? ln : 1e-12f;
    const float dt = fmaf(nc[0] / nn, L[0] / ln, fmaf(nc[1] / nn, L[1] / ln, (nc[2] / nn) * (L[2] / ln)));
    dv = fminf(fmaxf(dt, 0.f), 1.f);
  }
  VtxRecords r;
  r.c = make_int4(o.ok ? o.X : (int)0x80000000, o.Y, __float_as_int(o.zn), __float_as_int(o.w));
  const bool small = o.ok && abs(o.X) < 16384 && abs(o.Y) < 16384;
  // (where A cannot describe the vertex its second word carries w instead of z/w: the classification drops faces whose three
  // vertices are all on or behind the camera plane - a drifted tracking pose puts the whole mesh there - without reading C)
  r.a = small ? make_uint2(((unsigned)o.X & 0xffffu) | ((unsigned)o.Y << 16), __float_as_uint(o.zn)) : make_uint2(RB_A_NONE, __float_as_uint(o.w));
  r.b0 = make_float4(pc[0], pc[1], pc[2], dv);
  r.b1 = make_float4(0.f, 0.f, 0.f, o.w);
  if (m.vcolor) r.b1.x = m.vcolor[v * 3], r.b1.y = m.vcolor[v * 3 + 1], r.b1.z = m.vcolor[v * 3 + 2];
  return r;
}

__global__ __launch_bounds__(256) void xform_vertices_kernel(RenderArgs a, int4 *__restrict__ recC, float4 *__restrict__ recB, uint2 *__restrict__ recA) {
  __shared__ float sM[16];
  __shared__ float sP[12];
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    clip_matrix(a, b, a.poses + (size_t)b * 16, sM);
    for (int i = 0; i < 12; ++i) sP[i] = a.poses[(size_t)b * 16 + i];
  }
  __syncthreads();
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= a.mesh.V) return;
  const VtxRecords q = vertex_records(a, v, sM, sP);
  const size_t r = (size_t)b * a.mesh.V + v;
  recC[r] = q.c;
  recA[r] = q.a;
  recB[2 * r] = q.b0;
  recB[2 * r + 1] = q.b1;
}

#define RB_SMALL 4                      // candidate pixels of a "small" triangle
#define RB_MEDIUM 32                    // ... of a "medium" one; larger triangles are rasterised by a whole wave
#define RB_SLOW 0x80000000u             // list B entry flag: needs the C records (64-bit edge functions or the homogeneous form)
#define RB_MAXS 255                     // strips per hypothesis

// Per (hypothesis, strip) face lists.  A face is dropped here exactly when it cannot touch a pixel of the crop (pixel range empty
// after clamping to the crop, or zero area): the same integers as in the triangle pass, so the images are bit-identical to a walk
// over all faces.  The faces of a hypothesis are cut into G contiguous ranges of Fg, one workgroup each (G = 1 from 64 hypotheses on;
// a handful of hypotheses - tracking - would otherwise leave the classification to a handful of workgroups).  Per (hypothesis b, strip s,
// range g): count[((b*S + s)*G + g)*4 + {0,1,2}] = small, medium, list-B entries; listA[(b*S + s)*G*Fg + g*Fg ..]: small from the
// front of the range's segment, medium from its back; listB likewise: large 32-bit triangles and RB_SLOW ones.
// FUSED (the A records fit LDS: lds_verts): the vertex pre-pass runs HERE - every workgroup of a hypothesis transforms all its vertices
// (it needs their A records in LDS for its face range anyway; 8 vertices per thread) and writes the records of its own share to global
// memory for the strip kernel: one launch and one LDS fill from global memory less than xform_vertices_kernel + this kernel.
template <bool FUSED>
__global__ __launch_bounds__(RB_THREADS) void classify_faces_kernel(RenderArgs a, uint2 *__restrict__ recA, int4 *__restrict__ recC, float4 *__restrict__ recB,
                                                                   int *__restrict__ count, unsigned *__restrict__ listA, unsigned *__restrict__ listB, int S,
                                                                   int strip_rows, int lds_verts, int G, int Fg) {
  extern __shared__ __attribute__((aligned(16))) uint2 cl_ldsA[];
  __shared__ int cs[RB_MAXS + 1][3];
  __shared__ float sM[16];
  __shared__ float sP[12];
  const int b = blockIdx.x, grp = blockIdx.y;
  const MeshDev &m = a.mesh;
  const uint2 *gA = recA + (size_t)b * m.V;
  const int f0 = grp * Fg, f1 = min(m.F, f0 + Fg);
  for (int i = threadIdx.x; i < S * 3; i += RB_THREADS) cs[i / 3][i % 3] = 0;
  if constexpr (FUSED) {
    if (threadIdx.x == 0) {
      clip_matrix(a, b, a.poses + (size_t)b * 16, sM);
      for (int i = 0; i < 12; ++i) sP[i] = a.poses[(size_t)b * 16 + i];
    }
    __syncthreads();
    for (int v = threadIdx.x; v < m.V; v += RB_THREADS) {
      const VtxRecords q = vertex_records(a, v, sM, sP);
      cl_ldsA[v] = q.a;
      if ((v / RB_THREADS) % G == grp) {               // this workgroup's share of the hypothesis' records
        const size_t r = (size_t)b * m.V + v;
        recC[r] = q.c;
        recA[r] = q.a;
        recB[2 * r] = q.b0;
        recB[2 * r + 1] = q.b1;
      }
    }
  } else if (lds_verts) {
    for (int i = threadIdx.x; i < m.V; i += RB_THREADS) cl_ldsA[i] = gA[i];
  }
  __syncthreads();
  auto getA = [&](int i) -> uint2 { return (FUSED || lds_verts) ? cl_ldsA[i] : gA[i]; };
  const size_t seg = (size_t)G * Fg;                     // list entries per (hypothesis, strip)
  unsigned *lA = listA + (size_t)b * S * seg + (size_t)grp * Fg, *lB = listB + (size_t)b * S * seg + (size_t)grp * Fg;
  int4 f_n = make_int4(0, 0, 0, 0);
  if (f0 + (int)threadIdx.x < f1) f_n = m.faces4[f0 + threadIdx.x];
  for (int t = f0 + threadIdx.x; t < f1; t += RB_THREADS) {
    const int4 f = f_n;
    if (t + RB_THREADS < f1) f_n = m.faces4[t + RB_THREADS];
    const uint2 a0 = getA(f.x), a1 = getA(f.y), a2 = getA(f.z);
    if (a0.x == RB_A_NONE || a1.x == RB_A_NONE || a2.x == RB_A_NONE) {
      // a vertex A cannot describe.  Entirely on or behind the camera plane (w <= 0 for all three; a describable vertex has w > 0):
      // dropped, as the triangle pass would.  Otherwise the face cannot be bounded here: every strip looks at it.
      const bool front = (a0.x != RB_A_NONE || __uint_as_float(a0.y) > 0.f) || (a1.x != RB_A_NONE || __uint_as_float(a1.y) > 0.f) ||
                         (a2.x != RB_A_NONE || __uint_as_float(a2.y) > 0.f);
      if (!front) continue;
      for (int sI = 0; sI < S; ++sI) lB[(size_t)sI * seg + atomicAdd(&cs[sI][2], 1)] = (unsigned)t | RB_SLOW;
      continue;
    }
    const int X0 = (short)(a0.x & 0xffffu), Y0 = (int)a0.x >> 16, X1 = (short)(a1.x & 0xffffu), Y1 = (int)a1.x >> 16,
              X2 = (short)(a2.x & 0xffffu), Y2 = (int)a2.x >> 16;
    if ((X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0) == 0) continue;
    const int xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
    const int ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
    const int ia = max((xmin - 8 + 15) >> 4, 0), ib = min((xmax - 8) >> 4, a.Wo - 1);
    const int ja = max((ymin - 8 + 15) >> 4, 0), jb = min((ymax - 8) >> 4, a.Ho - 1);
    if (ia > ib || ja > jb) continue;
    const int bw = ib - ia + 1;
    for (int sI = ja / strip_rows; sI <= jb / strip_rows; ++sI) {
      const int rows = min(jb, (sI + 1) * strip_rows - 1) - max(ja, sI * strip_rows) + 1;
      const int nc = bw * rows;
      if (nc <= RB_SMALL) lA[(size_t)sI * seg + atomicAdd(&cs[sI][0], 1)] = (unsigned)t;
      else if (nc <= RB_MEDIUM) lA[(size_t)sI * seg + (Fg - 1 - atomicAdd(&cs[sI][1], 1))] = (unsigned)t;
      else lB[(size_t)sI * seg + atomicAdd(&cs[sI][2], 1)] = (unsigned)t;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S * 3; i += RB_THREADS) count[(((size_t)b * S + i / 3) * G + grp) * 4 + i % 3] = cs[i / 3][i % 3];
}

// MODE 0: the API form (fp32 channels-last maps), 1: the fused network tensor, 2: dr.rasterize's own output only (u, v, z/w, triangle id + 1: parity tests)
// SOLO: the whole render of a hypothesis' strip in THIS launch - for one or two hypotheses (a tracking frame), where the vertex pass,
// the classification and this kernel are three dependent launches of a few workgroups each (4.5 + 7.8 + 31 us and two 5-us gaps at
// one hypothesis, 37 us of it with nothing to draw).  Every strip's workgroup transforms all vertices itself (A records straight
// into LDS), files the faces that touch ITS strip into lists in LDS (16-bit entries; the same tests and size classes as
// classify_faces_kernel), and takes the B / C records of the few vertices it needs from vertex_records() directly: the same
// functions on the same inputs, and a z-buffer of (depth, face) keys under atomicMin does not depend on the order of the lists,
// so the images are bit-identical to the three-launch form (test_render_solo_equals_three_launches).
template <int MODE, bool SOLO>
__global__ __launch_bounds__(RB_THREADS) void render_kernel(RenderArgs a, int strip_rows, int n_strips, const int4 *__restrict__ recC,
                                                            const float4 *__restrict__ recB, const uint2 *__restrict__ recA, const int *__restrict__ count,
                                                            const unsigned *__restrict__ listA, const unsigned *__restrict__ listB, int lds_verts, int G, int Fg) {
  // dynamic LDS: the strip (8 B per pixel), the queue of its covered pixels (2 B per pixel), then (lds_verts) the hypothesis' A records,
  // then (SOLO) the strip's two face lists, 2 B per face each
  extern __shared__ __attribute__((aligned(16))) unsigned long long zbuf[];
  __shared__ float sM[16];
  __shared__ float sP[12];
  __shared__ int covn;
  __shared__ int scnt[3];
  const int L = xcd_remap(blockIdx.x, gridDim.x);          // the strips of one hypothesis share an XCD's L2
  const int b = L / n_strips, strip = L % n_strips;
  const int Ho = a.Ho, Wo = a.Wo;
  const int row0 = strip * strip_rows;  // GL (bottom-up) rows [row0, row1)
  const int row1 = min(Ho, row0 + strip_rows);
  const int npix = (row1 - row0) * Wo;
  const float *pose = a.poses + (size_t)b * 16;
  const MeshDev &m = a.mesh;
  const uint2 *gA = recA + (size_t)b * m.V;
  unsigned short *covq = reinterpret_cast<unsigned short *>(zbuf + (size_t)strip_rows * Wo);
  uint2 *ldsA = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(covq) + ((((size_t)strip_rows * Wo * 2) + 15) & ~(size_t)15));
  unsigned short *soloA = reinterpret_cast<unsigned short *>(ldsA + ((m.V + 1) & ~1));
  unsigned short *soloB = soloA + ((m.F + 7) & ~7);

  if (threadIdx.x == 0) {
    clip_matrix(a, b, pose, sM);
    covn = 0;
    if constexpr (SOLO) {
      for (int i = 0; i < 12; ++i) sP[i] = pose[i];
      scnt[0] = scnt[1] = scnt[2] = 0;
    }
  }
  for (int i = threadIdx.x; i < npix; i += RB_THREADS) zbuf[i] = ~0ull;
  if constexpr (SOLO) {
    __syncthreads();
    for (int v = threadIdx.x; v < m.V; v += RB_THREADS) ldsA[v] = (a.dbg & 64) ? make_uint2(RB_A_NONE, 0u) : vertex_records(a, v, sM, sP).a;
    __syncthreads();
    // the faces of this strip (classify_faces_kernel's tests, for one strip): small from the front of soloA, medium from its back
    // (four faces per thread and round: their index quads are requested together - at one hypothesis a load's latency is all there is to hide)
    auto file_face = [&](int t, const int4 &f) {
      const uint2 a0 = ldsA[f.x], a1 = ldsA[f.y], a2 = ldsA[f.z];
      if (a0.x == RB_A_NONE || a1.x == RB_A_NONE || a2.x == RB_A_NONE) {
        const bool front = (a0.x != RB_A_NONE || __uint_as_float(a0.y) > 0.f) || (a1.x != RB_A_NONE || __uint_as_float(a1.y) > 0.f) ||
                           (a2.x != RB_A_NONE || __uint_as_float(a2.y) > 0.f);
        if (front) soloB[atomicAdd(&scnt[2], 1)] = (unsigned short)t;
        return;
      }
      const int X0 = (short)(a0.x & 0xffffu), Y0 = (int)a0.x >> 16, X1 = (short)(a1.x & 0xffffu), Y1 = (int)a1.x >> 16,
                X2 = (short)(a2.x & 0xffffu), Y2 = (int)a2.x >> 16;
      if ((X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0) == 0) return;
      const int xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
      const int ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
      const int ia = max((xmin - 8 + 15) >> 4, 0), ib = min((xmax - 8) >> 4, Wo - 1);
      const int ja = max((ymin - 8 + 15) >> 4, row0), jb = min((ymax - 8) >> 4, row1 - 1);
      if (ia > ib || ja > jb) return;
      const int nc = (ib - ia + 1) * (jb - ja + 1);
      if (nc <= RB_SMALL) soloA[atomicAdd(&scnt[0], 1)] = (unsigned short)t;
      else if (nc <= RB_MEDIUM) soloA[m.F - 1 - atomicAdd(&scnt[1], 1)] = (unsigned short)t;
      else soloB[atomicAdd(&scnt[2], 1)] = (unsigned short)t;
    };
    const int F_do = (a.dbg & 32) ? 0 : m.F;
    for (int t0 = threadIdx.x; t0 < F_do; t0 += 4 * RB_THREADS) {
      int4 fq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) fq[u] = m.faces4[min(t0 + u * RB_THREADS, m.F - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t0 + u * RB_THREADS < F_do) file_face(t0 + u * RB_THREADS, fq[u]);
    }
  } else if (lds_verts) {
    for (int i = threadIdx.x; i < m.V; i += RB_THREADS) ldsA[i] = gA[i];
  }
  __syncthreads();

  // (the clip matrix stays in LDS: only triangles that straddle the camera plane read it)
  const float *M = sM;
  const float hw = 0.5f * (float)Wo, hh = 0.5f * (float)Ho;
  const int4 *vc = recC + (size_t)b * m.V;
  const float4 *vbB = recB + 2 * (size_t)b * m.V;
  auto getA = [&](int i) -> uint2 { return (SOLO || lds_verts) ? ldsA[i] : gA[i]; };
  auto getC = [&](int i) -> int4 {
    if constexpr (SOLO) return vertex_records(a, i, sM, sP).c;
    else return vc[i];
  };
  auto unpack = [](const int4 &q) -> Vtx {      // record C: exactly xform_vertex's values
    Vtx o;
    o.X = q.x;
    o.Y = q.y;
    o.zn = __int_as_float(q.z);
    o.w = __int_as_float(q.w);
    o.ok = q.x != (int)0x80000000;
    return o;
  };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (SOLO) G = 1, Fg = m.F;
  for (int grp = 0; grp < G; ++grp) {             // the face ranges the classification was cut into (one from 64 hypotheses on)
  const int *cnt = SOLO ? scnt : count + (((size_t)b * n_strips + strip) * G + grp) * 4;
  const int n_small = (a.dbg & 4) ? 0 : cnt[0], n_med = (a.dbg & 4) ? 0 : cnt[1], n_b = (a.dbg & 20) ? 0 : cnt[2];
  const unsigned *lA = listA + ((size_t)b * n_strips + strip) * G * Fg + (size_t)grp * Fg, *lB = listB + ((size_t)b * n_strips + strip) * G * Fg + (size_t)grp * Fg;
  auto entryA = [&](int i) -> int { return SOLO ? (int)soloA[i] : (int)lA[i]; };
  auto entryB = [&](int i) -> int { return SOLO ? (int)soloB[i] : (int)(lB[i] & ~RB_SLOW); };

  // ---- pass 1a, per lane: the small and medium triangles of this strip, 32-bit edge functions.  |X|,|Y| < 2^14 -> differences
  // < 2^15, products < 2^30, sums < 2^31: the same integers as the 64-bit form below, so coverage, barycentrics and depth keys
  // are bit-identical.  The list is dense and ordered small -> medium, so the lanes of a wave run similar pixel loops; the next
  // entry's index quad is requested one iteration ahead.
  {
    const int n_a = n_small + n_med;
    auto entry = [&](int e) -> int { return e < n_small ? entryA(e) : entryA(Fg - 1 - (e - n_small)); };
    int t_n = 0;
    int4 f_n = make_int4(0, 0, 0, 0);
    if ((int)threadIdx.x < n_a) {
      t_n = entry(threadIdx.x);
      f_n = m.faces4[t_n];
    }
    for (int e = threadIdx.x; e < n_a; e += RB_THREADS) {
      const int t = t_n;
      const int4 f = f_n;
      if (e + RB_THREADS < n_a) {
        t_n = entry(e + RB_THREADS);
        f_n = m.faces4[t_n];
      }
      if (a.dbg & 1) continue;
      const uint2 a0 = getA(f.x), a1 = getA(f.y), a2 = getA(f.z);
      const int X0 = (short)(a0.x & 0xffffu), Y0 = (int)a0.x >> 16, X1 = (short)(a1.x & 0xffffu), Y1 = (int)a1.x >> 16,
                X2 = (short)(a2.x & 0xffffu), Y2 = (int)a2.x >> 16;
      int area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
      const int xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
      const int ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
      int ia = (xmin - 8 + 15) >> 4, ib = (xmax - 8) >> 4, ja = (ymin - 8 + 15) >> 4, jb = (ymax - 8) >> 4;
      ia = max(ia, 0);
      ja = max(ja, row0);
      ib = min(ib, Wo - 1);
      jb = min(jb, row1 - 1);
      const int sg = area > 0 ? 1 : -1;
      area *= sg;
      const int dx0 = sg * (X2 - X1), dy0 = sg * (Y2 - Y1);
      const int dx1 = sg * (X0 - X2), dy1 = sg * (Y0 - Y2);
      const int dx2 = sg * (X1 - X0), dy2 = sg * (Y1 - Y0);
      const bool tl0 = (dy0 > 0) || (dy0 == 0 && dx0 < 0);
      const bool tl1 = (dy1 > 0) || (dy1 == 0 && dx1 < 0);
      const bool tl2 = (dy2 > 0) || (dy2 == 0 && dx2 < 0);
      const float fa = (float)area;
      const float z0 = __uint_as_float(a0.y), z1 = __uint_as_float(a1.y), z2 = __uint_as_float(a2.y);
      for (int j = ja; j <= jb; ++j) {
        const int Py = 16 * j + 8;
        for (int i = ia; i <= ib; ++i) {
          const int Px = 16 * i + 8;
          const int e0 = dx0 * (Py - Y1) - dy0 * (Px - X1);
          const int e1 = dx1 * (Py - Y2) - dy1 * (Px - X2);
          const int e2 = dx2 * (Py - Y0) - dy2 * (Px - X0);
          if (!((e0 > 0 || (e0 == 0 && tl0)) && (e1 > 0 || (e1 == 0 && tl1)) && (e2 > 0 || (e2 == 0 && tl2)))) continue;
          const float b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
          const float zp = fmaf(b2, z2, fmaf(b1, z1, b0 * z0));
          if (!(zp >= -1.f && zp <= 1.f)) continue;
          const unsigned long long key = ((unsigned long long)ordered_key(zp) << 32) | (unsigned)t;
          atomicMin(&zbuf[(j - row0) * Wo + i], key);
        }
      }
    }
  }
  // ---- pass 1b, per wave: list B, one candidate pixel per lane.  A wave takes CHB entries at a time: lane l fetches entry l's
  // index quad and its three C records (the triangles' loads in flight together), then the wave walks them with the records
  // broadcast from the owning lane; everything but the pixel is wave-uniform.
  constexpr int CHB = 4;        // entries a wave takes at a time (a strip has some tens of them: spread over all 16 waves)
  for (int base = wave * CHB; base < n_b; base += (RB_THREADS / 64) * CHB) {
    int my_t = 0;
    int4 my_f = make_int4(0, 0, 0, 0), my_c0 = my_f, my_c1 = my_f, my_c2 = my_f;
    if (lane < CHB && base + lane < n_b) {
      my_t = entryB(base + lane);
      my_f = m.faces4[my_t];
      my_c0 = getC(my_f.x), my_c1 = getC(my_f.y), my_c2 = getC(my_f.z);
    }
    const int cntw = min(CHB, n_b - base);
    for (int q = 0; q < cntw; ++q) {
      auto bc4 = [&](const int4 &x) -> int4 {
        return make_int4(__builtin_amdgcn_readlane(x.x, q), __builtin_amdgcn_readlane(x.y, q), __builtin_amdgcn_readlane(x.z, q),
                         __builtin_amdgcn_readlane(x.w, q));
      };
      const int t = __builtin_amdgcn_readlane(my_t, q);
      const int4 f = bc4(my_f);
      const Vtx v0 = unpack(bc4(my_c0)), v1 = unpack(bc4(my_c1)), v2 = unpack(bc4(my_c2));
      if (!(v0.ok && v1.ok && v2.ok)) {
        // a vertex on or behind the camera plane (or projected out of range): homogeneous edge functions, every pixel of the strip
        if (!(v0.w > 0.f || v1.w > 0.f || v2.w > 0.f)) continue;      // entirely behind the camera
        float c[3][4];
        const int fi3[3] = {f.x, f.y, f.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) clip_coords(m.pos, fi3[k], M, c[k]);
        const ClipTri T = clip_setup(c, hw, hh);
        if (!T.valid) continue;
        for (int p = lane; p < npix; p += 64) {
          const int jl = p / Wo, i = p - jl * Wo;
          float l[3], zp;
          if (!clip_eval(T, i, row0 + jl, l, &zp)) continue;
          const unsigned long long key = ((unsigned long long)ordered_key(zp) << 32) | (unsigned)t;
          atomicMin(&zbuf[p], key);
        }
        continue;
      }
      const long long X0 = v0.X, Y0 = v0.Y, X1 = v1.X, Y1 = v1.Y, X2 = v2.X, Y2 = v2.Y;
      long long area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
      if (area == 0) continue;
      const long long sg = area > 0 ? 1 : -1;
      area *= sg;
      const long long xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
      const long long ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
      long long ia = (xmin - 8 + 15) >> 4, ib = (xmax - 8) >> 4, ja = (ymin - 8 + 15) >> 4, jb = (ymax - 8) >> 4;
      ia = max(ia, 0ll);
      ja = max(ja, (long long)row0);
      ib = min(ib, (long long)Wo - 1);
      jb = min(jb, (long long)row1 - 1);
      if (ia > ib || ja > jb) continue;
      const long long dx0 = sg * (X2 - X1), dy0 = sg * (Y2 - Y1);
      const long long dx1 = sg * (X0 - X2), dy1 = sg * (Y0 - Y2);
      const long long dx2 = sg * (X1 - X0), dy2 = sg * (Y1 - Y0);
      const bool tl0 = (dy0 > 0) || (dy0 == 0 && dx0 < 0);
      const bool tl1 = (dy1 > 0) || (dy1 == 0 && dx1 < 0);
      const bool tl2 = (dy2 > 0) || (dy2 == 0 && dx2 < 0);
      const float fa = (float)area;
      const int bw = (int)(ib - ia + 1), ncand = bw * (int)(jb - ja + 1);
      for (int k = lane; k < ncand; k += 64) {
        const int jj = k / bw;
        const long long i = ia + (k - jj * bw), j = ja + jj;
        const long long Px = 16 * i + 8, Py = 16 * j + 8;
        const long long e0 = dx0 * (Py - Y1) - dy0 * (Px - X1);
        const long long e1 = dx1 * (Py - Y2) - dy1 * (Px - X2);
        const long long e2 = dx2 * (Py - Y0) - dy2 * (Px - X0);
        if (!((e0 > 0 || (e0 == 0 && tl0)) && (e1 > 0 || (e1 == 0 && tl1)) && (e2 > 0 || (e2 == 0 && tl2)))) continue;
        const float b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
        const float zp = fmaf(b2, v2.zn, fmaf(b1, v1.zn, b0 * v0.zn));
        if (!(zp >= -1.f && zp <= 1.f)) continue;
        const unsigned long long key = ((unsigned long long)ordered_key(zp) << 32) | (unsigned)t;
        atomicMin(&zbuf[(int)(j - row0) * Wo + (int)i], key);
      }
    }
  }
  }
  __syncthreads();

  // ---- pass 2: resolve.  (a) the covered pixels of the strip are queued (ballot + one LDS atomic per wave), the others written as
  // zeros at once; (b) one covered pixel per lane: attribute interpolation, shading, fused epilogue.  Per pixel: the face's index quad,
  // the three A records (LDS) and the three 32-byte B records - 7 scattered 16-byte loads on 4 cache lines (27 + 9 scattered 4-byte
  // loads until round 3, with the position / normal transform and Lambert term of all three vertices recomputed per pixel).
  float P0 = 0.f, P1 = 0.f, P2 = 0.f, P4 = 0.f, P5 = 0.f, P6 = 0.f, P8 = 0.f, P9 = 0.f, P10 = 0.f;
  constexpr bool API = MODE == 0, RAST = MODE == 2;
  if (API) {                            // the API form also interpolates the camera-space normals
    P0 = pose[0], P1 = pose[1], P2 = pose[2], P4 = pose[4], P5 = pose[5], P6 = pose[6], P8 = pose[8], P9 = pose[9], P10 = pose[10];
  }
  auto emit = [&](int p, const float *col, const float *nrm, const float *p3, const float *r4) __attribute__((always_inline)) {
    const int jl = p / Wo, i = p - jl * Wo;
    const int jo = Ho - 1 - (row0 + jl);  // flipped output row (src/Utils.py:216-218)
    const size_t o = ((size_t)b * Ho + jo) * Wo + i;
    if (API) {
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
    } else if (RAST) {
      *reinterpret_cast<float4 *>(a.rast + o * 4) = make_float4(r4[0], r4[1], r4[2], r4[3]);
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
  };
  for (int p0 = 0; p0 < npix; p0 += RB_THREADS) {
    const int p = p0 + threadIdx.x;
    const bool covered = p < npix && !(a.dbg & 2) && zbuf[p] != ~0ull;
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(covered);
    if (mk != 0) {
      int base = 0;
      if (lane == __builtin_ctzll(mk)) base = atomicAdd(&covn, __builtin_popcountll(mk));
      base = __shfl(base, __builtin_ctzll(mk));
      if (covered) covq[base + __builtin_popcountll(mk & ((1ull << lane) - 1))] = (unsigned short)p;
    }
    if (p < npix && !covered) {
      const float z3[4] = {0.f, 0.f, 0.f, 0.f};
      emit(p, z3, z3, z3, z3);
    }
  }
  __syncthreads();
  const int ncov = covn;
  for (int e = threadIdx.x; e < ncov; e += RB_THREADS) {
    const int p = covq[e];
    const int jl = p / Wo, i = p - jl * Wo;
    const int j = row0 + jl;
    const unsigned long long key = zbuf[p];
    float col[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, p3[3] = {0, 0, 0}, r4[4] = {0, 0, 0, 0};
    {
      int t = (int)(unsigned)(key & 0xffffffffull);
      const int4 f4 = m.faces4[t];
      const int i0 = f4.x, i1 = f4.y, i2 = f4.z;
      float4 rb0, rb1, rb2, rc0, rc1, rc2;
      [[maybe_unused]] int4 cc0, cc1, cc2;
      if constexpr (SOLO) {
        const VtxRecords q0 = vertex_records(a, i0, sM, sP), q1 = vertex_records(a, i1, sM, sP), q2 = vertex_records(a, i2, sM, sP);
        rb0 = q0.b0, rb1 = q1.b0, rb2 = q2.b0, rc0 = q0.b1, rc1 = q1.b1, rc2 = q2.b1;
        cc0 = q0.c, cc1 = q1.c, cc2 = q2.c;
      } else {
        rb0 = vbB[2 * i0], rb1 = vbB[2 * i1], rb2 = vbB[2 * i2];
        rc0 = vbB[2 * i0 + 1], rc1 = vbB[2 * i1 + 1], rc2 = vbB[2 * i2 + 1];
      }
      const uint2 a0 = getA(i0), a1 = getA(i1), a2 = getA(i2);
      float fa, b0, b1, b2;
      float u, v, w2;
      if (a0.x != RB_A_NONE && a1.x != RB_A_NONE && a2.x != RB_A_NONE) {      // the same integers in 32-bit arithmetic (see pass 1)
        const int X0 = (short)(a0.x & 0xffffu), Y0 = (int)a0.x >> 16, X1 = (short)(a1.x & 0xffffu), Y1 = (int)a1.x >> 16,
                  X2 = (short)(a2.x & 0xffffu), Y2 = (int)a2.x >> 16;
        int area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
        const int sg = area > 0 ? 1 : -1;
        area *= sg;
        const int Px = 16 * i + 8, Py = 16 * j + 8;
        const int e0 = sg * ((X2 - X1) * (Py - Y1) - (Y2 - Y1) * (Px - X1));
        const int e1 = sg * ((X0 - X2) * (Py - Y2) - (Y0 - Y2) * (Px - X2));
        const int e2 = sg * ((X1 - X0) * (Py - Y0) - (Y1 - Y0) * (Px - X0));
        fa = (float)area;
        b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
        float q0 = b0 / rc0.w, q1 = b1 / rc1.w, q2 = b2 / rc2.w;
        float qs = (q0 + q1) + q2;
        u = q0 / qs, v = q1 / qs, w2 = (1.f - u) - v;
        if (RAST) r4[2] = fmaf(b2, __uint_as_float(a2.y), fmaf(b1, __uint_as_float(a1.y), b0 * __uint_as_float(a0.y)));
      } else {
        const Vtx v0 = unpack(SOLO ? cc0 : vc[i0]), v1 = unpack(SOLO ? cc1 : vc[i1]), v2 = unpack(SOLO ? cc2 : vc[i2]);
        if (!(v0.ok && v1.ok && v2.ok)) {         // straddles the camera plane: weights from the homogeneous edge functions
          float c[3][4];
          const int id3[3] = {i0, i1, i2};
#pragma unroll
          for (int k = 0; k < 3; ++k) clip_coords(m.pos, id3[k], M, c[k]);
          const ClipTri T = clip_setup(c, hw, hh);
          float l[3], zp;
          (void)clip_eval(T, i, j, l, &zp);
          const float ls = __fadd_rn(__fadd_rn(l[0], l[1]), l[2]);
          u = __fdiv_rn(l[0], ls), v = __fdiv_rn(l[1], ls), w2 = (1.f - u) - v;
          if (RAST) r4[2] = zp;
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
          float q0 = b0 / v0.w, q1 = b1 / v1.w, q2 = b2 / v2.w;
          float qs = (q0 + q1) + q2;
          u = q0 / qs, v = q1 / qs, w2 = (1.f - u) - v;
          if (RAST) r4[2] = fmaf(b2, v2.zn, fmaf(b1, v1.zn, b0 * v0.zn));
        }
      }
      if (RAST) r4[0] = u, r4[1] = v, r4[3] = (float)(t + 1);
      if (!RAST) {
      // camera-space position and Lambert term of the three vertices: the pre-pass' records
      const float pc[3][3] = {{rb0.x, rb0.y, rb0.z}, {rb1.x, rb1.y, rb1.z}, {rb2.x, rb2.y, rb2.z}};
      const float dv[3] = {rb0.w, rb1.w, rb2.w};
      float nc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      if (API) {
        const int idx[3] = {i0, i1, i2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float nx = m.vnormals[idx[k] * 3], ny = m.vnormals[idx[k] * 3 + 1], nz = m.vnormals[idx[k] * 3 + 2];
          nc[k][0] = fmaf(P0, nx, fmaf(P1, ny, P2 * nz));
          nc[k][1] = fmaf(P4, nx, fmaf(P5, ny, P6 * nz));
          nc[k][2] = fmaf(P8, nx, fmaf(P9, ny, P10 * nz));
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
        base[0] = fmaf(u, rc0.x, fmaf(v, rc1.x, w2 * rc2.x));
        base[1] = fmaf(u, rc0.y, fmaf(v, rc1.y, w2 * rc2.y));
        base[2] = fmaf(u, rc0.z, fmaf(v, rc1.z, w2 * rc2.z));
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
    }
    emit(p, col, nrm, p3, r4);
  }
}

// Strips per hypothesis and the scratch a launch needs: the vertex records C (16 B), B (32 B) and A (8 B), the list counters and
// the two face lists of every (hypothesis, strip) (worst case: every face in every strip).
RenderPlan render_plan(int N, int V, int F, int Ho, int Wo, int num_cu) {
  RenderPlan p;
  const size_t budget = 148 * 1024;               // dynamic LDS of a workgroup
  p.lds_verts = (size_t)V * 8 <= 64 * 1024 ? 1 : 0;                          // the hypothesis' A records beside the strip
  p.a_lds = p.lds_verts ? (((size_t)V * 8 + 15) & ~(size_t)15) : 0;
  int rows_max = (int)((budget - p.a_lds - 16) / ((size_t)Wo * 10));        // 8 B framebuffer + 2 B covered-pixel queue per pixel
  if (rows_max > Ho) rows_max = Ho;
  if (rows_max < 1) rows_max = 1;
  int S = (Ho + rows_max - 1) / rows_max;
  // 27-row strips of a 160-row crop; thinner ones while the launch is under ~0.8 workgroups per CU (a workgroup's resolve pass shrinks
  // with its strip; twice the strips are then still under two rounds)
  while (S < 4 && Ho / (S * 2) >= 8) S *= 2;
  // (measured, 160x160 crops of the 16k-face mesh, strips 6 -> 12: 24 hypotheses 58 -> 49 us, 32: 72 -> 62, 40: 72 -> 78, 63: 92 -> 100)
  while ((size_t)N * S * 5 <= (size_t)num_cu * 4 && S < 16 && Ho / (S * 2) >= 8) S *= 2;
  static const int force_s = getenv("FP_RENDER_S") ? atoi(getenv("FP_RENDER_S")) : 0;      // experiments: strips per hypothesis
  if (force_s > 0 && (Ho + force_s - 1) / force_s <= rows_max) S = force_s;
  p.strip_rows = (Ho + S - 1) / S;
  p.S = (Ho + p.strip_rows - 1) / p.strip_rows;
  p.lds_bytes = (size_t)p.strip_rows * Wo * 8 + ((((size_t)p.strip_rows * Wo * 2) + 15) & ~(size_t)15) + p.a_lds;
  p.c_bytes = ((size_t)N * V * 16 + 255) & ~(size_t)255;
  p.b_bytes = ((size_t)N * V * 32 + 255) & ~(size_t)255;
  p.a_bytes = ((size_t)N * V * 8 + 255) & ~(size_t)255;
  p.G = 1;                                        // face ranges per hypothesis in the classification: more while it would fill < 1/4 of the chip
  while ((size_t)N * p.G * 4 <= (size_t)num_cu && p.G < 8 && F / (p.G * 2) >= 1024) p.G *= 2;
  p.Fg = (F + p.G - 1) / p.G;
  p.count_bytes = ((size_t)N * p.S * p.G * 16 + 255) & ~(size_t)255;
  p.list_bytes = ((size_t)N * p.S * p.G * p.Fg * 4 + 255) & ~(size_t)255;
  p.total = p.c_bytes + p.b_bytes + p.a_bytes + p.count_bytes + 2 * p.list_bytes;
  // one or two hypotheses (a tracking frame): vertex pass, classification and triangle pass as ONE launch (render_kernel<.., true>) when a
  // strip, the A records and the strip's two 16-bit face lists fit a workgroup's LDS.  FP_RENDER_SOLO = largest batch that takes it (0: off).
  static const int solo_max = getenv("FP_RENDER_SOLO") ? atoi(getenv("FP_RENDER_SOLO")) : 2;
  p.solo_lds = (((size_t)p.strip_rows * Wo * 8 + ((((size_t)p.strip_rows * Wo * 2) + 15) & ~(size_t)15) + (size_t)((V + 1) & ~1) * 8 + 15) & ~(size_t)15) +
               2 * (size_t)((F + 7) & ~7) * 2;
  p.solo = (N <= solo_max && p.lds_verts && F <= 65535 && p.solo_lds <= budget) ? 1 : 0;
  return p;
}

void raster_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)classify_faces_kernel<true>, 64 * 1024});
  v.push_back({(const void *)classify_faces_kernel<false>, 64 * 1024});
  v.push_back({(const void *)render_kernel<1, false>, 148 * 1024});        // the largest strip
  v.push_back({(const void *)render_kernel<0, false>, 148 * 1024});
  v.push_back({(const void *)render_kernel<2, false>, 148 * 1024});
  v.push_back({(const void *)render_kernel<1, true>, 148 * 1024});
  v.push_back({(const void *)render_kernel<0, true>, 148 * 1024});
  v.push_back({(const void *)render_kernel<2, true>, 148 * 1024});
}

int render_chunk(int N, int V, int F, int Ho, int Wo, int num_cu) {
  static const size_t cap = getenv("FP_RENDER_SCRATCH_MAX") ? (size_t)atoll(getenv("FP_RENDER_SCRATCH_MAX")) : ((size_t)1 << 30);
  int chunk = N < 1 ? 1 : N;
  while (chunk > 1 && render_plan(chunk, V, F, Ho, Wo, num_cu).total > cap) chunk = (chunk + 1) / 2;
  return chunk;
}

size_t render_scratch_bytes(int N, int V, int F, int Ho, int Wo, int num_cu) {
  return render_plan(render_chunk(N, V, F, Ho, Wo, num_cu), V, F, Ho, Wo, num_cu).total;
}

// one sub-batch; plan_n: the batch size the plan (strips, face ranges, list strides) is made for (>= a.N: a smaller last sub-batch runs on
// the plan of the full ones, so the scratch of a full one always holds it)
static int launch_render_one(fp_ctx *ctx, const RenderArgs &a_in, int plan_n, hipStream_t s) {
  RenderArgs a = a_in;
  static const int dbg_env = getenv("FP_RENDER_DBG") ? atoi(getenv("FP_RENDER_DBG")) : 0;      // timing experiments only (wrong images): 1 no per-lane rasterisation, 2 no resolve, 4 no triangle pass, 16 no per-wave rasterisation
  a.dbg = dbg_env;
  FP_REQUIRE(a.N >= 0 && a.Ho > 0 && a.Wo > 0, "render: bad shape N=%d out=%dx%d", a.N, a.Ho, a.Wo);
  if (a.N == 0) return FP_OK;
  FP_REQUIRE((size_t)a.Wo * 10 <= 64 * 1024, "render: output width %d too large for one LDS strip", a.Wo);
  FP_REQUIRE(a.mesh.F < (1 << 30), "render: %d faces", a.mesh.F);
  const RenderPlan pl = render_plan(plan_n, a.mesh.V, a.mesh.F, a.Ho, a.Wo, ctx->num_cu);
  FP_REQUIRE(pl.S <= RB_MAXS && pl.strip_rows * a.Wo <= 65535, "render: output %dx%d needs %d strips of %d pixels", a.Ho, a.Wo, pl.S, pl.strip_rows * a.Wo);
  FP_REQUIRE(a.scratch && a.scratch_bytes >= pl.total, "render: scratch of %zu bytes needed, %zu given", pl.total, a.scratch_bytes);
  char *sc = (char *)a.scratch;
  int4 *recC = (int4 *)sc;
  float4 *recB = (float4 *)(sc + pl.c_bytes);
  uint2 *recA = (uint2 *)(sc + pl.c_bytes + pl.b_bytes);
  int *count = (int *)(sc + pl.c_bytes + pl.b_bytes + pl.a_bytes);
  unsigned *listA = (unsigned *)(sc + pl.c_bytes + pl.b_bytes + pl.a_bytes + pl.count_bytes);
  unsigned *listB = (unsigned *)(sc + pl.c_bytes + pl.b_bytes + pl.a_bytes + pl.count_bytes + pl.list_bytes);
  // (profiling: the class' work figure is BYTES WRITTEN - the fused fp16 net tensor, or the API's fp32 maps - for the HBM-stage line of bench.py)
  ProfScope ps(ctx, s, "render", a.net_out ? (double)a.N * a.Ho * a.Wo * 16.0 : (double)a.N * a.Ho * a.Wo * 40.0);
  static const bool two_launches = getenv("FP_RENDER_PREPASS2") != nullptr;       // A/B knob: vertex pre-pass and classification as two launches (identical images)
  // fused where one workgroup per hypothesis classifies (G == 1: from 64 hypotheses on): 220 -> 210 us at 252 hypotheses, 144 -> 136 at 126; with
  // the faces of a hypothesis cut into G ranges every range's workgroup would redo the vertex pass (32 hypotheses: 70 -> 73 us, 1: 32 -> 34)
  if (pl.solo) {
    // (nothing in front of the strip kernel)
  } else if (pl.lds_verts && pl.G == 1 && !two_launches) {
    hipLaunchKernelGGL(classify_faces_kernel<true>, dim3(a.N, pl.G), dim3(RB_THREADS), pl.a_lds, s, a, recA, recC, recB, count, listA, listB, pl.S,
                       pl.strip_rows, pl.lds_verts, pl.G, pl.Fg);
  } else {
    hipLaunchKernelGGL(xform_vertices_kernel, dim3((a.mesh.V + 255) / 256, a.N), dim3(256), 0, s, a, recC, recB, recA);
    hipLaunchKernelGGL(classify_faces_kernel<false>, dim3(a.N, pl.G), dim3(RB_THREADS), pl.a_lds, s, a, recA, recC, recB, count, listA, listB, pl.S,
                       pl.strip_rows, pl.lds_verts, pl.G, pl.Fg);
  }
  FP_CHECK_HIP(hipGetLastError());
  auto go = [&](auto kern) -> int {
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * pl.S)), dim3(RB_THREADS), pl.solo ? pl.solo_lds : pl.lds_bytes, s, a, pl.strip_rows, pl.S, (const int4 *)recC,
                       (const float4 *)recB, (const uint2 *)recA, (const int *)count, (const unsigned *)listA, (const unsigned *)listB, pl.lds_verts, pl.G,
                       pl.Fg);
    return FP_OK;
  };
  if (a.net_out) {
    FP_TRY(pl.solo ? go(render_kernel<1, true>) : go(render_kernel<1, false>));
  } else {
    if (a.color || a.depth || a.normal || a.xyz) FP_TRY(pl.solo ? go(render_kernel<0, true>) : go(render_kernel<0, false>));
    if (a.rast) FP_TRY(pl.solo ? go(render_kernel<2, true>) : go(render_kernel<2, false>));        // (a second pass over the same lists: the parity tests' output)
  }
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int launch_render(fp_ctx *ctx, const RenderArgs &a, hipStream_t s) {
  FP_REQUIRE(a.N >= 0 && a.Ho > 0 && a.Wo > 0, "render: bad shape N=%d out=%dx%d", a.N, a.Ho, a.Wo);
  if (a.N == 0) return FP_OK;
  const int chunk = render_chunk(a.N, a.mesh.V, a.mesh.F, a.Ho, a.Wo, ctx->num_cu);
  const size_t px = (size_t)a.Ho * a.Wo;
  for (int b0 = 0; b0 < a.N; b0 += chunk) {
    RenderArgs c = a;
    c.N = a.N - b0 < chunk ? a.N - b0 : chunk;
    c.poses = a.poses + (size_t)b0 * 16;
    if (a.bbox2d) c.bbox2d = a.bbox2d + (size_t)b0 * 4;
    if (a.color) c.color = a.color + b0 * px * 3;
    if (a.depth) c.depth = a.depth + b0 * px;
    if (a.normal) c.normal = a.normal + b0 * px * 3;
    if (a.xyz) c.xyz = a.xyz + b0 * px * 3;
    if (a.rast) c.rast = a.rast + b0 * px * 4;
    if (a.net_out) c.net_out = a.net_out + b0 * px * 8;
    FP_TRY(launch_render_one(ctx, c, chunk, s));
  }
  return FP_OK;
}

// ----------------------------------------------------------------------------------------------
// compute_crop_window_tf_batch(method='box_3d') (src/Utils.py:577-621) + bbox2d_ori
// (predict_pose_refine.py:44-45).  float32, left-to-right, no FMA contraction: mirrors
// oracle/geometry.py:compute_crop_window_tf_batch.
// ----------------------------------------------------------------------------------------------
__global__ void crop_window_tf_kernel(const float *__restrict__ poses, int N, CropWindowK c, float *tf, float *bbox) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  crop_window_tf_one(b, poses, c, tf, bbox);
}

CropWindowK crop_window_k(const double *K, double crop_ratio, double diameter, int ow, int oh) {
  return CropWindowK{(float)K[0], (float)K[1], (float)K[2], (float)K[3], (float)K[4], (float)K[5], (float)K[6], (float)K[7], (float)K[8],
                     (float)(diameter * crop_ratio / 2.0), (float)ow, (float)oh};
}

int launch_crop_window_tf(const float *poses, int N, const double *K, double crop_ratio, double diameter, int ow, int oh, float *tf,
                          float *bbox, hipStream_t s) {
  if (N == 0) return FP_OK;
  hipLaunchKernelGGL(crop_window_tf_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poses, N, crop_window_k(K, crop_ratio, diameter, ow, oh), tf, bbox);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
