/* Oracle software rasteriser (CPU, plain C + OpenMP).  TEST INFRASTRUCTURE - see oracle/__init__.py.
 *
 * Restates what the reference asks of nvdiffrast in src/Utils.py:133-219 (nvdiffrast_render):
 *   dr.rasterize (:182)  -> coverage + nearest z/w, rows bottom-up, pixel centres at half-integers
 *   dr.interpolate (:183,:186/:189,:195,:207) -> perspective-correct barycentric interpolation
 *   dr.texture (:187)    -> bilinear, wrap
 *   shading (:199-212), mask (:215), row flips (:216-218)
 * nvdiffrast itself is not in the reference checkout nor in this image: PARITY UNPINNED.  The
 * rules fixed here (and mirrored 1:1 by foundationpose_amd/csrc/raster.hip) are:
 *   - clip = M * (p,1) evaluated as an fmaf chain; a triangle with a vertex at w<=0 or |screen|>1e6 (it straddles the camera
 *     plane) is rasterised in 2-D homogeneous coordinates instead of being clipped: clip_setup / clip_eval below
 *   - screen coordinates snapped to 1/16 pixel (rintf), integer edge functions, top-left rule
 *   - both windings are rasterised (nvdiffrast does not cull back faces)
 *   - depth = screen-affine interpolation of z/w; -1<=z/w<=1; nearest wins, ties -> lower face id
 *   - attributes: u = (b0/w0)/S, v = (b1/w1)/S, third weight 1-u-v
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  const float *pos;      /* V*3 */
  const int32_t *faces;  /* F*3 */
  const float *vnormals; /* V*3 */
  const float *vcolor;   /* V*3 or NULL */
  const float *uv;       /* Vt*2 or NULL (v already flipped) */
  const int32_t *uv_idx; /* F*3 or NULL */
  const float *tex;      /* Ht*Wt*3 or NULL */
  int V, F, texH, texW;
} oracle_mesh;

static inline uint32_t ordered_key(float z) {
  uint32_t b; memcpy(&b, &z, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void tex_fetch(const oracle_mesh *m, float u, float v, float *rgb) {
  float x = u * (float)m->texW - 0.5f, y = v * (float)m->texH - 0.5f;
  float fx0 = floorf(x), fy0 = floorf(y);
  float fx = x - fx0, fy = y - fy0;
  int W = m->texW, H = m->texH;
  int x0 = (int)fx0 % W; if (x0 < 0) x0 += W;
  int y0 = (int)fy0 % H; if (y0 < 0) y0 += H;
  int x1 = (x0 + 1) % W, y1 = (y0 + 1) % H;
  for (int c = 0; c < 3; ++c) {
    float t00 = m->tex[(y0 * W + x0) * 3 + c], t10 = m->tex[(y0 * W + x1) * 3 + c];
    float t01 = m->tex[(y1 * W + x0) * 3 + c], t11 = m->tex[(y1 * W + x1) * 3 + c];
    float a = t00 + fx * (t10 - t00);
    float b = t01 + fx * (t11 - t01);
    rgb[c] = a + fy * (b - a);
  }
}

/* Homogeneous rasterisation of a triangle that straddles the camera plane (nvdiffrast clips it; see raster.hip): pixel-homogeneous
 * vertices v_k = (cx hw + cw hw, cy hh + cw hh, cw), edge normals n_0 = v_1 x v_2 (cyclic) times sign(D), D = v_0 . n_0; weights
 * l_k = n_k . (i+.5, j+.5, 1); the pixel sees the triangle iff all l_k > 0; z/w = sum l_k cz_k / sum l_k cw_k in [-1,1]. */
typedef struct { float n[3][3], cz[3], cw[3]; int valid; } clip_tri;

static clip_tri clip_setup(const float *c0, const float *c1, const float *c2, float hw, float hh) {
  clip_tri T;
  const float *c[3] = {c0, c1, c2};
  float v[3][3];
  for (int k = 0; k < 3; ++k) {
    v[k][0] = c[k][0] * hw + c[k][3] * hw;
    v[k][1] = c[k][1] * hh + c[k][3] * hh;
    v[k][2] = c[k][3];
    T.cz[k] = c[k][2];
    T.cw[k] = c[k][3];
  }
  for (int k = 0; k < 3; ++k) {
    const float *a = v[(k + 1) % 3], *b = v[(k + 2) % 3];
    T.n[k][0] = a[1] * b[2] - a[2] * b[1];
    T.n[k][1] = a[2] * b[0] - a[0] * b[2];
    T.n[k][2] = a[0] * b[1] - a[1] * b[0];
  }
  float D = (v[0][0] * T.n[0][0] + v[0][1] * T.n[0][1]) + v[0][2] * T.n[0][2];
  T.valid = D != 0.f && isfinite(D);
  float sg = D > 0.f ? 1.f : -1.f;
  for (int k = 0; k < 3; ++k) for (int e = 0; e < 3; ++e) T.n[k][e] = T.n[k][e] * sg;
  return T;
}

static int clip_eval(const clip_tri *T, int i, int j, float *l, float *zp) {
  float Px = (float)i + 0.5f, Py = (float)j + 0.5f;
  for (int k = 0; k < 3; ++k) l[k] = (T->n[k][0] * Px + T->n[k][1] * Py) + T->n[k][2];
  if (!(l[0] > 0.f && l[1] > 0.f && l[2] > 0.f)) return 0;
  float num = (l[0] * T->cz[0] + l[1] * T->cz[1]) + l[2] * T->cz[2];
  float den = (l[0] * T->cw[0] + l[1] * T->cw[1]) + l[2] * T->cw[2];
  if (!(den > 0.f)) return 0;
  *zp = num / den;
  return *zp >= -1.f && *zp <= 1.f;
}

/* M: B*16 float (row-major clip matrix incl. bbox window transform), pose: B*16 float (ob_in_cam).
 * outputs (each may be NULL): color B*Ho*Wo*3, depth B*Ho*Wo, normal B*Ho*Wo*3, xyz B*Ho*Wo*3,
 * rast B*Ho*Wo*4 = (u, v, z/w, face_id+1) already flipped top-down. */
/* light_mode 0: light_dir = (0,0,1) (default); 1: light_vec = -light_dir; 2: light_vec = light_pos (light_dir=None);
 * light_color NULL: the diffuse term takes the surface colour (src/Utils.py:200-211) */
int oracle_render_lit(const oracle_mesh *m, int B, const float *Mclip, const float *pose, int Ho, int Wo,
                      int use_light, float w_ambient, float w_diffuse, int light_mode, const float *light_vec, const float *light_color,
                      float *color, float *depth, float *normal, float *xyz, float *rast);

int oracle_render(const oracle_mesh *m, int B, const float *Mclip, const float *pose, int Ho, int Wo,
                  int use_light, float w_ambient, float w_diffuse,
                  float *color, float *depth, float *normal, float *xyz, float *rast) {
  return oracle_render_lit(m, B, Mclip, pose, Ho, Wo, use_light, w_ambient, w_diffuse, 0, 0, 0, color, depth, normal, xyz, rast);
}

int oracle_render_lit(const oracle_mesh *m, int B, const float *Mclip, const float *pose, int Ho, int Wo,
                      int use_light, float w_ambient, float w_diffuse, int light_mode, const float *light_vec, const float *light_color,
                      float *color, float *depth, float *normal, float *xyz, float *rast) {
  const int V = m->V, F = m->F;
  int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    const float *M = Mclip + (size_t)b * 16, *P = pose + (size_t)b * 16;
    float *clipc = (float *)malloc(sizeof(float) * V * 4);      /* clip coordinates */
    float *clipw = (float *)malloc(sizeof(float) * V);          /* w */
    float *zn = (float *)malloc(sizeof(float) * V);             /* z/w */
    int32_t *fx = (int32_t *)malloc(sizeof(int32_t) * V * 2);   /* fixed-point screen */
    uint8_t *ok = (uint8_t *)malloc(V);
    float *pc = (float *)malloc(sizeof(float) * V * 3);         /* camera-space xyz */
    float *nc = (float *)malloc(sizeof(float) * V * 3);         /* camera-space normal */
    float *dv = (float *)malloc(sizeof(float) * V);             /* per-vertex diffuse */
    uint64_t *zbuf = (uint64_t *)malloc(sizeof(uint64_t) * Ho * Wo);
    if (!clipc || !clipw || !zn || !fx || !ok || !pc || !nc || !dv || !zbuf) { err = 1; }
    else {
    for (int i = 0; i < Ho * Wo; ++i) zbuf[i] = ~(uint64_t)0;
    for (int v = 0; v < V; ++v) {
      float px = m->pos[v * 3], py = m->pos[v * 3 + 1], pz = m->pos[v * 3 + 2];
      float c[4];
      for (int r = 0; r < 4; ++r)
        c[r] = fmaf(M[r * 4 + 0], px, fmaf(M[r * 4 + 1], py, fmaf(M[r * 4 + 2], pz, M[r * 4 + 3])));
      for (int r = 0; r < 3; ++r)
        pc[v * 3 + r] = fmaf(P[r * 4 + 0], px, fmaf(P[r * 4 + 1], py, fmaf(P[r * 4 + 2], pz, P[r * 4 + 3])));
      float nx = m->vnormals[v * 3], ny = m->vnormals[v * 3 + 1], nz = m->vnormals[v * 3 + 2];
      for (int r = 0; r < 3; ++r)
        nc[v * 3 + r] = fmaf(P[r * 4 + 0], nx, fmaf(P[r * 4 + 1], ny, P[r * 4 + 2] * nz));
      float nn = sqrtf(fmaf(nc[v * 3], nc[v * 3], fmaf(nc[v * 3 + 1], nc[v * 3 + 1], nc[v * 3 + 2] * nc[v * 3 + 2])));
      nn = nn > 1e-12f ? nn : 1e-12f;
      if (light_mode == 0) {
        dv[v] = clampf(-(nc[v * 3 + 2] / nn), 0.f, 1.f);
      } else {
        float L[3];
        for (int r = 0; r < 3; ++r) L[r] = light_mode == 1 ? light_vec[r] : light_vec[r] - pc[v * 3 + r];
        float ln = sqrtf(fmaf(L[0], L[0], fmaf(L[1], L[1], L[2] * L[2])));
        ln = ln > 1e-12f ? ln : 1e-12f;
        float dt = fmaf(nc[v * 3] / nn, L[0] / ln, fmaf(nc[v * 3 + 1] / nn, L[1] / ln, (nc[v * 3 + 2] / nn) * (L[2] / ln)));
        dv[v] = clampf(dt, 0.f, 1.f);
      }
      float w = c[3];
      for (int r = 0; r < 4; ++r) clipc[v * 4 + r] = c[r];
      clipw[v] = w;
      ok[v] = 0;
      if (w > 0.f) {
        float xn = c[0] / w, yn = c[1] / w;
        zn[v] = c[2] / w;
        float sx = fmaf(xn, 0.5f * (float)Wo, 0.5f * (float)Wo);
        float sy = fmaf(yn, 0.5f * (float)Ho, 0.5f * (float)Ho);
        if (fabsf(sx) <= 1e6f && fabsf(sy) <= 1e6f) {
          fx[v * 2] = (int32_t)rintf(sx * 16.f);
          fx[v * 2 + 1] = (int32_t)rintf(sy * 16.f);
          ok[v] = 1;
        }
      }
    }
    for (int t = 0; t < F; ++t) {
      int i0 = m->faces[t * 3], i1 = m->faces[t * 3 + 1], i2 = m->faces[t * 3 + 2];
      if (!(ok[i0] && ok[i1] && ok[i2])) {
        if (!(clipw[i0] > 0.f || clipw[i1] > 0.f || clipw[i2] > 0.f)) continue;   /* entirely behind the camera */
        clip_tri T = clip_setup(clipc + i0 * 4, clipc + i1 * 4, clipc + i2 * 4, 0.5f * (float)Wo, 0.5f * (float)Ho);
        if (!T.valid) continue;
        for (int j = 0; j < Ho; ++j)
          for (int i = 0; i < Wo; ++i) {
            float l[3], zp;
            if (!clip_eval(&T, i, j, l, &zp)) continue;
            uint64_t key = ((uint64_t)ordered_key(zp) << 32) | (uint32_t)t;
            uint64_t *zb = &zbuf[j * Wo + i];
            if (key < *zb) *zb = key;
          }
        continue;
      }
      int64_t X0 = fx[i0 * 2], Y0 = fx[i0 * 2 + 1], X1 = fx[i1 * 2], Y1 = fx[i1 * 2 + 1], X2 = fx[i2 * 2], Y2 = fx[i2 * 2 + 1];
      int64_t area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
      if (area == 0) continue;
      int64_t s = area > 0 ? 1 : -1;
      area *= s;
      int64_t xmin = X0 < X1 ? X0 : X1; if (X2 < xmin) xmin = X2;
      int64_t xmax = X0 > X1 ? X0 : X1; if (X2 > xmax) xmax = X2;
      int64_t ymin = Y0 < Y1 ? Y0 : Y1; if (Y2 < ymin) ymin = Y2;
      int64_t ymax = Y0 > Y1 ? Y0 : Y1; if (Y2 > ymax) ymax = Y2;
      int64_t ia = (xmin - 8 + 15) >> 4, ib = (xmax - 8) >> 4, ja = (ymin - 8 + 15) >> 4, jb = (ymax - 8) >> 4;
      if (ia < 0) ia = 0; if (ja < 0) ja = 0; if (ib > Wo - 1) ib = Wo - 1; if (jb > Ho - 1) jb = Ho - 1;
      if (ia > ib || ja > jb) continue;
      /* edge i is opposite vertex i; (dx,dy) sign-normalised */
      int64_t dx0 = s * (X2 - X1), dy0 = s * (Y2 - Y1);
      int64_t dx1 = s * (X0 - X2), dy1 = s * (Y0 - Y2);
      int64_t dx2 = s * (X1 - X0), dy2 = s * (Y1 - Y0);
      int tl0 = (dy0 > 0) || (dy0 == 0 && dx0 < 0);
      int tl1 = (dy1 > 0) || (dy1 == 0 && dx1 < 0);
      int tl2 = (dy2 > 0) || (dy2 == 0 && dx2 < 0);
      float fa = (float)area;
      for (int64_t j = ja; j <= jb; ++j) {
        int64_t Py = 16 * j + 8;
        for (int64_t i = ia; i <= ib; ++i) {
          int64_t Px = 16 * i + 8;
          int64_t e0 = dx0 * (Py - Y1) - dy0 * (Px - X1);
          int64_t e1 = dx1 * (Py - Y2) - dy1 * (Px - X2);
          int64_t e2 = dx2 * (Py - Y0) - dy2 * (Px - X0);
          if (!((e0 > 0 || (e0 == 0 && tl0)) && (e1 > 0 || (e1 == 0 && tl1)) && (e2 > 0 || (e2 == 0 && tl2)))) continue;
          float b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
          float zp = fmaf(b2, zn[i2], fmaf(b1, zn[i1], b0 * zn[i0]));
          if (!(zp >= -1.f && zp <= 1.f)) continue;
          uint64_t key = ((uint64_t)ordered_key(zp) << 32) | (uint32_t)t;
          uint64_t *zb = &zbuf[j * Wo + i];
          if (key < *zb) *zb = key;
        }
      }
    }
    /* resolve */
    for (int j = 0; j < Ho; ++j) {
      int jo = Ho - 1 - j;   /* flipped output row (src/Utils.py:216-218) */
      for (int i = 0; i < Wo; ++i) {
        size_t o = ((size_t)b * Ho + jo) * Wo + i;
        uint64_t key = zbuf[j * Wo + i];
        float col[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, p3[3] = {0, 0, 0}, r4[4] = {0, 0, 0, 0};
        if (key != ~(uint64_t)0) {
          int t = (int)(uint32_t)(key & 0xffffffffu);
          int i0 = m->faces[t * 3], i1 = m->faces[t * 3 + 1], i2 = m->faces[t * 3 + 2];
          float u, v, w2, zp;
          if (!(ok[i0] && ok[i1] && ok[i2])) {     /* straddles the camera plane: weights from the homogeneous edge functions */
            clip_tri T = clip_setup(clipc + i0 * 4, clipc + i1 * 4, clipc + i2 * 4, 0.5f * (float)Wo, 0.5f * (float)Ho);
            float l[3];
            (void)clip_eval(&T, i, j, l, &zp);
            float ls = (l[0] + l[1]) + l[2];
            u = l[0] / ls; v = l[1] / ls; w2 = (1.f - u) - v;
          } else {
          int64_t X0 = fx[i0 * 2], Y0 = fx[i0 * 2 + 1], X1 = fx[i1 * 2], Y1 = fx[i1 * 2 + 1], X2 = fx[i2 * 2], Y2 = fx[i2 * 2 + 1];
          int64_t area = (X1 - X0) * (Y2 - Y0) - (X2 - X0) * (Y1 - Y0);
          int64_t s = area > 0 ? 1 : -1;
          area *= s;
          int64_t Px = 16 * (int64_t)i + 8, Py = 16 * (int64_t)j + 8;
          int64_t e0 = s * ((X2 - X1) * (Py - Y1) - (Y2 - Y1) * (Px - X1));
          int64_t e1 = s * ((X0 - X2) * (Py - Y2) - (Y0 - Y2) * (Px - X2));
          int64_t e2 = s * ((X1 - X0) * (Py - Y0) - (Y1 - Y0) * (Px - X0));
          float fa = (float)area;
          float b0 = (float)e0 / fa, b1 = (float)e1 / fa, b2 = (float)e2 / fa;
          zp = fmaf(b2, zn[i2], fmaf(b1, zn[i1], b0 * zn[i0]));
          float q0 = b0 / clipw[i0], q1 = b1 / clipw[i1], q2 = b2 / clipw[i2];
          float qs = (q0 + q1) + q2;
          u = q0 / qs; v = q1 / qs; w2 = (1.f - u) - v;
          }
          r4[0] = u; r4[1] = v; r4[2] = zp; r4[3] = (float)(t + 1);
          for (int c = 0; c < 3; ++c) {
            p3[c] = fmaf(u, pc[i0 * 3 + c], fmaf(v, pc[i1 * 3 + c], w2 * pc[i2 * 3 + c]));
            nrm[c] = fmaf(u, nc[i0 * 3 + c], fmaf(v, nc[i1 * 3 + c], w2 * nc[i2 * 3 + c]));
          }
          float base[3];
          if (m->tex) {
            int a0 = m->uv_idx[t * 3], a1 = m->uv_idx[t * 3 + 1], a2 = m->uv_idx[t * 3 + 2];
            float tu = fmaf(u, m->uv[a0 * 2], fmaf(v, m->uv[a1 * 2], w2 * m->uv[a2 * 2]));
            float tv = fmaf(u, m->uv[a0 * 2 + 1], fmaf(v, m->uv[a1 * 2 + 1], w2 * m->uv[a2 * 2 + 1]));
            tex_fetch(m, tu, tv, base);
          } else {
            for (int c = 0; c < 3; ++c)
              base[c] = fmaf(u, m->vcolor[i0 * 3 + c], fmaf(v, m->vcolor[i1 * 3 + c], w2 * m->vcolor[i2 * 3 + c]));
          }
          if (use_light) {
            float d = fmaf(u, dv[i0], fmaf(v, dv[i1], w2 * dv[i2]));
            for (int c = 0; c < 3; ++c) base[c] = base[c] * w_ambient + (d * (light_color ? light_color[c] : base[c])) * w_diffuse;
          }
          for (int c = 0; c < 3; ++c) col[c] = clampf(base[c], 0.f, 1.f);
          float nn = sqrtf(fmaf(nrm[0], nrm[0], fmaf(nrm[1], nrm[1], nrm[2] * nrm[2])));
          nn = nn > 1e-12f ? nn : 1e-12f;
          for (int c = 0; c < 3; ++c) nrm[c] = nrm[c] / nn;
        }
        if (color) for (int c = 0; c < 3; ++c) color[o * 3 + c] = col[c];
        if (normal) for (int c = 0; c < 3; ++c) normal[o * 3 + c] = nrm[c];
        if (xyz) for (int c = 0; c < 3; ++c) xyz[o * 3 + c] = p3[c];
        if (depth) depth[o] = p3[2];
        if (rast) for (int c = 0; c < 4; ++c) rast[o * 4 + c] = r4[c];
      }
    }
    }
    free(clipc); free(clipw); free(zn); free(fx); free(ok); free(pc); free(nc); free(dv); free(zbuf);
  }
  return err;
}
