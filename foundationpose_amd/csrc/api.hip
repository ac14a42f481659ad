// extern "C" entry points of libfoundationpose_amd (see include/foundationpose_amd.h).
#include "common.h"
#include <memory>
#include <cstring>

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <algorithm>

static thread_local char g_err[1024] = "";

void fp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *fp_last_error(void) { return g_err; }
extern "C" int fp_version(void) { return 100; }

// ---- context / arena ---------------------------------------------------------------------------
size_t fp_arena_bytes_for(int n_hyp) {
  // per hypothesis: net input 0.82 MB (x2 sides) + the forward's buffers (fp_arena_inner_bytes)
  return (size_t)n_hyp * (size_t)(17u << 20) + ((size_t)64 << 20);
}

size_t fp_arena_inner_bytes(int n_hyp) {
  // exact sum of the forward's buffers is 11,485,184 B per hypothesis (DESIGN.md "HBM layout")
  // + 2.9 MB per hypothesis for the second transformer head's buffers (the two heads of RefineNet run side by side)
  return (size_t)n_hyp * (size_t)(15u << 20) + ((size_t)8 << 20) + (n_hyp <= 4 ? ((size_t)40 << 20) : 0);   // + split-K scratch (1 .. 4 hypotheses)
}

int fp_arena_ensure(fp_ctx *ctx, size_t bytes) {
  Arena &a = ctx->arena;
  if (a.cap - a.off >= bytes && a.base) return FP_OK;
  if (a.off != 0) {
    fp_set_error("arena too small for a nested request of %zu bytes (cap %zu, used %zu); call fp_ctx_reserve first", bytes, a.cap, a.off);
    return FP_ENOMEM;
  }
  FP_CHECK_HIP(hipSetDevice(ctx->device));
  if (a.base) {
    FP_CHECK_HIP(hipDeviceSynchronize());
    FP_CHECK_HIP(hipFree(a.base));
    a.base = nullptr;
    a.cap = 0;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    fp_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return FP_ENOMEM;
  }
  a.base = (char *)p;
  a.cap = bytes;
  a.off = 0;
  ++a.generation;
  return FP_OK;
}

extern int g_halo_tail, g_halo_npw, g_halo_form;
static int g_one_chain = 0;      // FP_ONE_CHAIN=1: the two sides of encodeA as one chain (A/B timing; identical results)

int fp_set_kernel_attributes(fp_ctx *ctx) {
  std::vector<KernelLds> v;
  conv_kernel_lds(v), conv_halo_kernel_lds(v), conv_s1b_kernel_lds(v), conv_small_kernel_lds(v), conv_wino_kernel_lds(v), conv_s2_kernel_lds(v), stem_kernel_lds(v);
  tok_gemm_kernel_lds(v), tok_qkv_kernel_lds(v), head_mlp_kernel_lds(v), attn_kernel_lds(v), raster_kernel_lds(v);
  FP_CHECK_HIP(hipSetDevice(ctx->device));
  for (const KernelLds &k : v)
    if (k.bytes > 48 * 1024) FP_CHECK_HIP(hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes));
  return FP_OK;
}

extern "C" int fp_ctx_create(int device, fp_ctx **out) {
  FP_REQUIRE(out, "fp_ctx_create: null out");
  if (const char *e = getenv("FP_HALO_TAIL")) g_halo_tail = atoi(e) != 0;
  if (const char *e = getenv("FP_ONE_CHAIN")) g_one_chain = atoi(e) != 0;
  if (const char *e = getenv("FP_HALO_NPW")) g_halo_npw = atoi(e) == 2 ? 2 : 4;
  if (const char *e = getenv("FP_HALO_FORM")) g_halo_form = atoi(e);
  int n = 0;
  FP_CHECK_HIP(hipGetDeviceCount(&n));
  FP_REQUIRE(device >= 0 && device < n, "fp_ctx_create: device %d out of range (%d visible)", device, n);
  FP_CHECK_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  FP_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  fp_ctx *c = new fp_ctx;
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  // one allocation: the 4-KB zero page, then the score tail's arrival counters (zeros as well)
  if (hipMalloc(&c->zero_page, 4096 + FP_TAIL_MAX_GROUPS * sizeof(int)) != hipSuccess ||
      hipMemset(c->zero_page, 0, 4096 + FP_TAIL_MAX_GROUPS * sizeof(int)) != hipSuccess) {
    delete c;
    fp_set_error("fp_ctx_create: zero page allocation failed");
    return FP_ENOMEM;
  }
  c->tail_counter = (int *)((char *)c->zero_page + 4096);
  const int rc = fp_set_kernel_attributes(c);      // per device: every kernel of the library that needs more than 64 KB of LDS
  if (rc != FP_OK) {
    (void)hipFree(c->zero_page);
    delete c;
    return rc;
  }
  *out = c;
  return FP_OK;
}

extern "C" int fp_ctx_destroy(fp_ctx *ctx) {
  if (!ctx) return FP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto &e : ctx->pending) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->ev_ref) (void)hipEventDestroy(ctx->ev_ref);
  if (ctx->side_ready) {
    for (int i = 0; i < fp_ctx::NSIDE; ++i) {
      (void)hipStreamDestroy(ctx->side[i]);
      (void)hipEventDestroy(ctx->ev_join[i]);
    }
    (void)hipEventDestroy(ctx->ev_fork);
  }
  if (ctx->arena.base) (void)hipFree(ctx->arena.base);
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  delete ctx;
  return FP_OK;
}

extern "C" int fp_ctx_arena_generation(const fp_ctx *ctx) { return ctx ? ctx->arena.generation : -1; }

extern "C" int fp_ctx_reserve(fp_ctx *ctx, int max_hyp) {
  FP_REQUIRE(ctx && max_hyp >= 1, "fp_ctx_reserve: bad argument");
  FP_REQUIRE(ctx->arena.off == 0, "fp_ctx_reserve: arena in use");
  size_t need = fp_arena_bytes_for(max_hyp);
  if (ctx->arena.cap >= need) return FP_OK;
  ctx->arena.off = 0;
  Arena &a = ctx->arena;
  if (a.base) {
    FP_CHECK_HIP(hipDeviceSynchronize());
    FP_CHECK_HIP(hipFree(a.base));
    a.base = nullptr;
    a.cap = 0;
  }
  FP_TRY(fp_arena_ensure(ctx, need));
  ctx->reserved_hyp = max_hyp;
  return FP_OK;
}

// ---- profiling ---------------------------------------------------------------------------------
static int prof_drain(fp_ctx *ctx);

extern "C" int fp_prof_enable(fp_ctx *ctx, int on) {
  FP_REQUIRE(ctx, "fp_prof_enable: null ctx");
  ctx->prof = on < 0 ? 0 : (on > 2 ? 2 : on);
  if (ctx->prof) {                            // the time origin of the launch spans (fp_prof_read_busy), renewed whenever profiling is
    FP_TRY(prof_drain(ctx));                  // switched on: the spans are float32 milliseconds since then
    if (!ctx->ev_ref) FP_CHECK_HIP(hipEventCreate(&ctx->ev_ref));
    FP_CHECK_HIP(hipEventRecord(ctx->ev_ref, nullptr));
    FP_CHECK_HIP(hipEventSynchronize(ctx->ev_ref));
  }
  return FP_OK;
}

static int prof_drain(fp_ctx *ctx) {
  for (auto &e : ctx->pending) {
    FP_CHECK_HIP(hipEventSynchronize(e.b));
    float ms = 0.f, t0 = 0.f;
    FP_CHECK_HIP(hipEventElapsedTime(&ms, e.a, e.b));
    ProfEntry &p = ctx->prof_tab[e.cls];
    if (ctx->ev_ref && hipEventElapsedTime(&t0, ctx->ev_ref, e.a) == hipSuccess) p.spans.emplace_back(t0, t0 + ms);
    p.total_ms += ms;
    p.flops += e.flops;
    p.launches += 1;
    ctx->ev_pool.push_back(e.a);
    ctx->ev_pool.push_back(e.b);
  }
  ctx->pending.clear();
  return FP_OK;
}

extern "C" int fp_prof_read(fp_ctx *ctx, const char *cls, double *total_ms, int64_t *launches, double *flops) {
  FP_REQUIRE(ctx && cls, "fp_prof_read: null argument");
  FP_TRY(prof_drain(ctx));
  auto it = ctx->prof_tab.find(cls);
  ProfEntry e = (it == ctx->prof_tab.end()) ? ProfEntry() : it->second;
  if (total_ms) *total_ms = e.total_ms;
  if (launches) *launches = e.launches;
  if (flops) *flops = e.flops;
  return FP_OK;
}

// Time during which AT LEAST ONE launch of the class was executing (the union of the launch spans): equal to total_ms when the
// launches follow one another, smaller when launches of the class overlap - the two half-batch trunks on two streams, the two
// RefineNet heads.  FLOPs / busy time is the rate the chip sustains on the class.
extern "C" int fp_prof_read_busy(fp_ctx *ctx, const char *cls, double *busy_ms) {
  FP_REQUIRE(ctx && cls && busy_ms, "fp_prof_read_busy: null argument");
  FP_TRY(prof_drain(ctx));
  *busy_ms = 0.0;
  auto it = ctx->prof_tab.find(cls);
  if (it == ctx->prof_tab.end()) return FP_OK;
  std::vector<std::pair<float, float>> v = it->second.spans;
  std::sort(v.begin(), v.end());
  double busy = 0.0;
  float cur_a = 0.f, cur_b = -1.f;
  for (const auto &sp : v) {
    if (cur_b < cur_a || sp.first > cur_b) {
      if (cur_b >= cur_a) busy += cur_b - cur_a;
      cur_a = sp.first;
      cur_b = sp.second;
    } else if (sp.second > cur_b) {
      cur_b = sp.second;
    }
  }
  if (cur_b >= cur_a) busy += cur_b - cur_a;
  *busy_ms = busy;
  return FP_OK;
}

extern "C" int fp_prof_reset(fp_ctx *ctx) {
  FP_REQUIRE(ctx, "fp_prof_reset: null ctx");
  FP_TRY(prof_drain(ctx));
  ctx->prof_tab.clear();
  return FP_OK;
}

// ---- mesh --------------------------------------------------------------------------------------
template <typename T>
static int up(fp_mesh *m, const T *h, size_t n, const T **d) {
  void *p = nullptr;
  FP_CHECK_HIP(hipMalloc(&p, n * sizeof(T)));
  m->allocs.push_back(p);
  FP_CHECK_HIP(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice));
  *d = (const T *)p;
  return FP_OK;
}

extern "C" int fp_mesh_create(fp_ctx *ctx, const float *h_pos, int V, const int32_t *h_faces, int F, const float *h_vnormals,
                              const float *h_vertex_color, const float *h_uv, int n_uv, const int32_t *h_uv_idx, const float *h_tex,
                              int texH, int texW, fp_mesh **out) {
  FP_REQUIRE(ctx && h_pos && h_faces && h_vnormals && out, "fp_mesh_create: null argument");
  FP_REQUIRE(V > 0 && F > 0, "fp_mesh_create: empty mesh (V=%d F=%d)", V, F);
  FP_REQUIRE(h_vertex_color || (h_uv && h_uv_idx && h_tex && texH > 0 && texW > 0 && n_uv > 0),
             "fp_mesh_create: need vertex colours or (uv, uv_idx, tex)");
  for (int i = 0; i < F * 3; ++i) FP_REQUIRE(h_faces[i] >= 0 && h_faces[i] < V, "fp_mesh_create: face index %d out of range", h_faces[i]);
  const bool textured = (h_tex != nullptr && h_vertex_color == nullptr);
  if (textured)
    for (int i = 0; i < F * 3; ++i) FP_REQUIRE(h_uv_idx[i] >= 0 && h_uv_idx[i] < n_uv, "fp_mesh_create: uv index out of range");
  FP_CHECK_HIP(hipSetDevice(ctx->device));
  fp_mesh *m = new fp_mesh;
  memset(&m->d, 0, sizeof(MeshDev));
  m->d.V = V;
  m->d.F = F;
  int rc = FP_OK;
  auto run = [&]() -> int {
    FP_TRY(up(m, h_pos, (size_t)V * 3, &m->d.pos));
    FP_TRY(up(m, h_faces, (size_t)F * 3, &m->d.faces));
    {
      std::vector<int4> f4(F);
      for (int t = 0; t < F; ++t) f4[t] = make_int4(h_faces[t * 3], h_faces[t * 3 + 1], h_faces[t * 3 + 2], 0);
      FP_TRY(up(m, f4.data(), (size_t)F, &m->d.faces4));
    }
    FP_TRY(up(m, h_vnormals, (size_t)V * 3, &m->d.vnormals));
    if (textured) {
      FP_TRY(up(m, h_uv, (size_t)n_uv * 2, &m->d.uv));
      FP_TRY(up(m, h_uv_idx, (size_t)F * 3, &m->d.uv_idx));
      FP_TRY(up(m, h_tex, (size_t)texH * texW * 3, &m->d.tex));
      m->d.texH = texH;
      m->d.texW = texW;
    } else {
      FP_TRY(up(m, h_vertex_color, (size_t)V * 3, &m->d.vcolor));
    }
    return FP_OK;
  };
  rc = run();
  if (rc != FP_OK) {
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
    return rc;
  }
  *out = m;
  return FP_OK;
}

extern "C" int fp_mesh_destroy(fp_mesh *m) {
  if (!m) return FP_OK;
  for (void *p : m->allocs) (void)hipFree(p);
  delete m;
  return FP_OK;
}

// ---- render / crops ------------------------------------------------------------------------------
extern "C" int fp_crop_window_tf(fp_ctx *ctx, const float *d_poses, int N, const double *K, double crop_ratio, double mesh_diameter,
                                 int out_w, int out_h, float *d_tf, float *d_bbox2d, void *stream) {
  FP_REQUIRE(ctx && d_poses && K && d_tf, "fp_crop_window_tf: null argument");
  FP_REQUIRE(N >= 0 && out_w > 0 && out_h > 0 && mesh_diameter > 0 && crop_ratio > 0, "fp_crop_window_tf: bad argument");
  return launch_crop_window_tf(d_poses, N, K, crop_ratio, mesh_diameter, out_w, out_h, d_tf, d_bbox2d, (hipStream_t)stream);
}

// A render called on its own (the nvdiffrast_render API, fp_render_net): its scratch (transformed vertices + strip face lists) comes
// from the context's arena and is released when the launches are queued - the next taker runs behind them on the same stream.
static int render_with_arena_scratch(fp_ctx *ctx, RenderArgs &a, hipStream_t s) {
  if (a.N == 0) return FP_OK;
  const size_t bytes = render_scratch_bytes(a.N, a.mesh.V, a.mesh.F, a.Ho, a.Wo, ctx->num_cu);     // (sub-batches above 1 GiB: launch_render)
  FP_TRY(fp_arena_ensure(ctx, bytes + 4096));
  const size_t mark = ctx->arena.off;
  a.scratch = ctx->arena.take(bytes);
  a.scratch_bytes = bytes;
  const int rc = a.scratch ? launch_render(ctx, a, s) : FP_ENOMEM;
  ctx->arena.off = mark;
  return rc;
}

static int fill_render(RenderArgs &a, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                       const float *d_bbox2d, int out_h, int out_w) {
  FP_REQUIRE(mesh && K && (d_poses || N == 0), "render: null argument");
  FP_REQUIRE(N >= 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0, "render: bad shape");
  memset(&a, 0, sizeof(a));
  a.mesh = mesh->d;
  a.poses = d_poses;
  a.bbox2d = d_bbox2d;
  for (int i = 0; i < 9; ++i) a.K[i] = K[i];
  a.N = N;
  a.H = H;
  a.W = W;
  a.Ho = out_h;
  a.Wo = out_w;
  return FP_OK;
}

extern "C" int fp_render(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                         const float *d_bbox2d, int out_h, int out_w, int use_light, float w_ambient, float w_diffuse, float *d_color,
                         float *d_depth, float *d_normal, float *d_xyz, void *stream) {
  FP_REQUIRE(ctx, "fp_render: null ctx");
  RenderArgs a;
  FP_TRY(fill_render(a, mesh, d_poses, N, K, H, W, d_bbox2d, out_h, out_w));
  a.use_light = use_light;
  a.w_ambient = w_ambient;
  a.w_diffuse = w_diffuse;
  a.color = d_color;
  a.depth = d_depth;
  a.normal = d_normal;
  a.xyz = d_xyz;
  return render_with_arena_scratch(ctx, a, (hipStream_t)stream);
}

extern "C" int fp_render_ex(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                            const float *d_bbox2d, int out_h, int out_w, const fp_render_opts *opts_in, float *d_color, float *d_depth,
                            float *d_normal, float *d_xyz, void *stream) {
  FP_REQUIRE(ctx, "fp_render_ex: null ctx");
  if (!opts_in) return fp_render(ctx, mesh, d_poses, N, K, H, W, d_bbox2d, out_h, out_w, 0, 0.8f, 0.5f, d_color, d_depth, d_normal, d_xyz, stream);
  // versioned by size: a caller built against the header before `d_rast` passes a shorter struct - never read past what it holds
  fp_render_opts o;
  memset(&o, 0, sizeof(o));
  {
    const size_t sz = opts_in->struct_size, sz_r4 = offsetof(fp_render_opts, d_rast);
    FP_REQUIRE(sz == sizeof(fp_render_opts) || sz == sz_r4, "fp_render_ex: fp_render_opts.struct_size = %zu (this library knows %zu and %zu)", sz, sz_r4, sizeof(fp_render_opts));
    memcpy(&o, opts_in, sz);
  }
  const fp_render_opts *opts = &o;
  FP_REQUIRE(opts->light_mode >= 0 && opts->light_mode <= 2, "fp_render_ex: light_mode %d unknown", opts->light_mode);
  static const double unit_k[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  FP_REQUIRE(K || opts->has_projection, "fp_render_ex: neither K nor a projection matrix");
  RenderArgs a;
  FP_TRY(fill_render(a, mesh, d_poses, N, K ? K : unit_k, H, W, d_bbox2d, out_h, out_w));
  a.use_light = opts->use_light;
  a.w_ambient = opts->w_ambient;
  a.w_diffuse = opts->w_diffuse;
  a.light_mode = opts->light_mode;
  a.has_light_color = opts->has_light_color;
  for (int c = 0; c < 3; ++c) a.light_vec[c] = opts->light_vec[c], a.light_color[c] = opts->light_color[c];
  a.has_proj = opts->has_projection;
  for (int i = 0; i < 16; ++i) a.proj[i] = opts->projection[i];
  a.color = d_color;
  a.depth = d_depth;
  a.normal = d_normal;
  a.xyz = d_xyz;
  a.rast = opts->d_rast;
  return render_with_arena_scratch(ctx, a, (hipStream_t)stream);
}

// scratch: render_scratch_bytes(N, ...) bytes for the vertex pre-pass and the strip face lists (nullptr: taken from the arena here)
static int render_net_impl(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                           const float *d_bbox2d, int out_h, int out_w, double mesh_diameter, int normalize_xyz, float invalid_thres,
                           void *d_net_out, void *scratch, size_t scratch_bytes, void *stream) {
  FP_REQUIRE(ctx && d_net_out, "fp_render_net: null argument");
  RenderArgs a;
  FP_TRY(fill_render(a, mesh, d_poses, N, K, H, W, d_bbox2d, out_h, out_w));
  a.use_light = 1;  // make_crop_data_batch renders with use_light=True (predict_pose_refine.py:49)
  a.w_ambient = 0.8f;
  a.w_diffuse = 0.5f;
  a.net_out = (f16 *)d_net_out;
  a.mesh_diameter = (float)mesh_diameter;
  a.invalid_thres = invalid_thres;
  a.normalize_xyz = normalize_xyz;
  if (!scratch) return render_with_arena_scratch(ctx, a, (hipStream_t)stream);
  a.scratch = scratch;
  a.scratch_bytes = scratch_bytes;
  return launch_render(ctx, a, (hipStream_t)stream);
}

extern "C" int fp_render_net(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                             const float *d_bbox2d, int out_h, int out_w, double mesh_diameter, int normalize_xyz, float invalid_thres,
                             void *d_net_out, void *stream) {
  return render_net_impl(ctx, mesh, d_poses, N, K, H, W, d_bbox2d, out_h, out_w, mesh_diameter, normalize_xyz, invalid_thres, d_net_out,
                         nullptr, 0, stream);
}

extern "C" int fp_crop_observed(fp_ctx *ctx, const float *d_rgb, const float *d_geom, int H, int W, const double *K, const float *d_tf,
                                const float *d_poses, int N, int out_h, int out_w, int mode, double mesh_diameter, int normalize_xyz,
                                int out_fmt, void *d_out, void *stream) {
  FP_REQUIRE(ctx && d_rgb && d_geom && K && d_tf && d_poses && d_out, "fp_crop_observed: null argument");
  FP_REQUIRE(out_fmt == 0 || out_fmt == 1, "fp_crop_observed: out_fmt must be 0 or 1");
  CropArgs a;
  a.rgb = d_rgb;
  a.geom = d_geom;
  a.tf = d_tf;
  a.poses = d_poses;
  for (int i = 0; i < 9; ++i) a.K[i] = K[i];
  a.H = H;
  a.W = W;
  a.N = N;
  a.Ho = out_h;
  a.Wo = out_w;
  a.mode = mode;
  a.normalize_xyz = normalize_xyz;
  a.out_fmt = out_fmt;
  a.mesh_diameter = (float)mesh_diameter;
  a.out = d_out;
  ProfScope ps(ctx, (hipStream_t)stream, "crop", (double)N * out_h * out_w * (out_fmt == 1 ? 16.0 : 24.0));      // bytes written
  return launch_crop_observed(a, (hipStream_t)stream);
}

extern "C" int fp_warp_nearest(fp_ctx *ctx, const float *d_src, int src_batch, int src_h, int src_w, int channels, const float *d_tf, int N,
                               int out_h, int out_w, float *d_out, void *stream) {
  FP_REQUIRE(ctx && d_src && d_tf && d_out, "fp_warp_nearest: null argument");
  return launch_warp_nearest(d_src, src_batch, src_h, src_w, channels, d_tf, N, out_h, out_w, d_out, (hipStream_t)stream);
}

extern "C" int fp_erode_depth(fp_ctx *ctx, const float *d_depth, int H, int W, int radius, float depth_diff_thres, float ratio_thres,
                              float zfar, float *d_out, void *stream) {
  FP_REQUIRE(ctx && d_depth && d_out && H > 0 && W > 0 && radius >= 0, "fp_erode_depth: bad argument");
  return launch_erode(d_depth, H, W, radius, depth_diff_thres, ratio_thres, zfar, d_out, (hipStream_t)stream);
}
extern "C" int fp_bilateral_filter_depth(fp_ctx *ctx, const float *d_depth, int H, int W, int radius, float zfar, float sigmaD,
                                         float sigmaR, float *d_out, void *stream) {
  FP_REQUIRE(ctx && d_depth && d_out && H > 0 && W > 0 && radius >= 0, "fp_bilateral_filter_depth: bad argument");
  return launch_bilateral(d_depth, H, W, radius, zfar, sigmaD, sigmaR, d_out, (hipStream_t)stream);
}
extern "C" int fp_depth2xyzmap(fp_ctx *ctx, const float *d_depth, int H, int W, const double *K, float zfar, float *d_xyz, void *stream) {
  FP_REQUIRE(ctx && d_depth && d_xyz && K && H > 0 && W > 0, "fp_depth2xyzmap: bad argument");
  return launch_depth2xyz(d_depth, H, W, K, zfar, d_xyz, (hipStream_t)stream);
}

extern "C" int fp_depth_prefilter(fp_ctx *ctx, const float *d_depth, int H, int W, int radius, float depth_diff_thres, float ratio_thres,
                                  float zfar_erode, float zfar_bilateral, float sigmaD, float sigmaR, const double *K, float zfar_xyz,
                                  float *d_depth_out, float *d_xyz, const uint8_t *d_rgb_u8, float *d_rgb_f32, void *stream) {
  FP_REQUIRE(ctx && d_depth && d_depth_out && d_xyz && K && H > 0 && W > 0, "fp_depth_prefilter: bad argument");
  FP_REQUIRE((d_rgb_u8 == nullptr) == (d_rgb_f32 == nullptr), "fp_depth_prefilter: d_rgb_u8 and d_rgb_f32 go together");
  FP_REQUIRE(radius == 2, "fp_depth_prefilter: radius %d (the fused prelude is built for the radius 2 of src/estimater.py:256-257; "
                          "other radii: fp_erode_depth, fp_bilateral_filter_depth, fp_depth2xyzmap)", radius);
  FP_REQUIRE(d_depth != d_depth_out, "fp_depth_prefilter: in-place filtering is not possible (neighbouring workgroups read the input)");
  return launch_depth_prefilter(d_depth, H, W, depth_diff_thres, ratio_thres, zfar_erode, zfar_bilateral, sigmaD, sigmaR, K, zfar_xyz, d_depth_out,
                                d_xyz, d_rgb_u8, d_rgb_f32, (hipStream_t)stream);
}

extern "C" int fp_depth2xyzmap_f64(fp_ctx *ctx, const float *d_depth, int H, int W, const double *K, float *d_xyz, void *stream) {
  FP_REQUIRE(ctx && d_depth && d_xyz && K && H > 0 && W > 0, "fp_depth2xyzmap_f64: bad argument");
  return launch_depth2xyz_f64(d_depth, H, W, K, d_xyz, (hipStream_t)stream);
}

extern "C" int fp_mask_depth_stats(fp_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int H, int W, float min_depth, int32_t *h_stats6,
                                   float *h_median, void *stream) {
  FP_REQUIRE(ctx && d_depth && d_mask && h_stats6 && h_median && H > 0 && W > 0, "fp_mask_depth_stats: bad argument");
  hipStream_t s = (hipStream_t)stream;
  FP_TRY(fp_arena_ensure(ctx, 4096));
  const size_t mark = ctx->arena.off;
  int *d_out = (int *)ctx->arena.take(8 * sizeof(int));
  FP_REQUIRE(d_out, "fp_mask_depth_stats: arena exhausted");
  int rc = launch_mask_depth_stats(d_depth, d_mask, H, W, min_depth, d_out, (float *)(d_out + 6), s);
  ctx->arena.off = mark;
  if (rc != FP_OK) return rc;
  int host[8];
  FP_CHECK_HIP(hipMemcpyAsync(host, d_out, sizeof(host), hipMemcpyDeviceToHost, s));
  FP_CHECK_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < 6; ++i) h_stats6[i] = host[i];
  memcpy(h_median, &host[6], sizeof(float));
  return FP_OK;
}

extern "C" int fp_pose_update(fp_ctx *ctx, const float *d_poseA, const float *d_trans, const float *d_rot, int N, int rot_dim,
                              int trans_rep_tanh, const float *tn, float rot_normalizer, float trans_scale, float *d_pose_out,
                              void *stream) {
  FP_REQUIRE(ctx && d_poseA && d_trans && d_rot && d_pose_out, "fp_pose_update: null argument");
  float t0 = tn ? tn[0] : 1.f, t1 = tn ? tn[1] : 1.f, t2 = tn ? tn[2] : 1.f;
  FP_REQUIRE(trans_rep_tanh == 0 || trans_rep_tanh == 1, "fp_pose_update: trans_rep_tanh must be 0 or 1 (trans_rep='deepim': fp_pose_update_deepim)");
  return launch_pose_update(d_poseA, d_trans, d_rot, N, rot_dim, trans_rep_tanh, t0, t1, t2, rot_normalizer, trans_scale, d_pose_out,
                            (hipStream_t)stream);
}

extern "C" int fp_pose_update_deepim(fp_ctx *ctx, const float *d_poseA, const float *d_trans, const float *d_rot, int N, int rot_dim,
                                     const float *d_tf_to_crops, const double *K, float input_resize, float rot_normalizer, float trans_scale,
                                     float *d_pose_out, void *stream) {
  FP_REQUIRE(ctx && d_poseA && d_trans && d_rot && d_pose_out && d_tf_to_crops && K, "fp_pose_update_deepim: null argument");
  return launch_pose_update(d_poseA, d_trans, d_rot, N, rot_dim, 2, 1.f, 1.f, 1.f, rot_normalizer, trans_scale, d_pose_out, (hipStream_t)stream,
                            d_tf_to_crops, K, input_resize);
}

// ---- composed loops --------------------------------------------------------------------------------
#define TAKE(ptr, type, count)                                                             \
  type *ptr = (type *)ctx->arena.take((size_t)(count) * sizeof(type));                     \
  if (!ptr) {                                                                              \
    fp_set_error("arena exhausted (%s); call fp_ctx_reserve with a larger max_hyp", #ptr); \
    return FP_ENOMEM;                                                                      \
  }

static int check_objs(const fp_object_batch *objs, int n_obj, int *total) {
  FP_REQUIRE(objs && n_obj >= 1, "multi: need at least one object");
  int N = 0;
  for (int o = 0; o < n_obj; ++o) {
    FP_REQUIRE(objs[o].mesh && objs[o].d_rgb && objs[o].d_geom && objs[o].K, "multi: object %d has a null field", o);
    FP_REQUIRE(objs[o].n >= 0 && objs[o].H > 1 && objs[o].W > 1 && objs[o].mesh_diameter > 0, "multi: object %d has a bad shape", o);
    N += objs[o].n;
  }
  *total = N;
  return FP_OK;
}

// Objects whose crop windows and renders can share one launch: same mesh, camera and frame size (their hypotheses are
// contiguous in d_poses).  A rank of the sharded multi-GPU job holds a slice of several objects of ONE mesh: one 252-hypothesis
// render instead of eight 32-hypothesis ones that each fill a quarter of the chip.
static bool same_render_key(const fp_object_batch &a, const fp_object_batch &b) {
  return a.mesh == b.mesh && a.H == b.H && a.W == b.W && a.mesh_diameter == b.mesh_diameter && memcmp(a.K, b.K, 9 * sizeof(double)) == 0;
}

// Scratch of the renders of a pass: one render_scratch_bytes(...) per run of like objects (the renders of the runs may overlap on side
// streams, so each has its own block; a run whose worst case exceeds 1 GiB is rendered in sub-batches: launch_render).
static size_t render_scratch_total(fp_ctx *ctx, const fp_object_batch *objs, int n_obj) {
  size_t bytes = 0;
  for (int o = 0; o < n_obj;) {
    int e = o + 1, cnt = objs[o].n;
    while (e < n_obj && same_render_key(objs[o], objs[e])) cnt += objs[e++].n;
    if (cnt > 0) bytes += (render_scratch_bytes(cnt, objs[o].mesh->d.V, objs[o].mesh->d.F, 160, 160, ctx->num_cu) + 255) & ~(size_t)255;
    o = e;
  }
  return bytes;
}

// arena bytes of a fused pass over N hypotheses: crop transforms, the two net-input sides (0.82 MB per hypothesis), the render
// scratch, the network forward
static size_t pass_arena_bytes(int N, size_t render_scratch) {
  return (size_t)N * ((size_t)1 << 20) + ((size_t)1 << 20) + render_scratch + fp_arena_inner_bytes(N);
}

static int count_runs(const fp_object_batch *objs, int n_obj) {
  int k = 0;
  for (int o = 0; o < n_obj;) {
    int e = o + 1, cnt = objs[o].n;
    while (e < n_obj && same_render_key(objs[o], objs[e])) cnt += objs[e++].n;
    k += cnt > 0;
    o = e;
  }
  return k;
}

// `centered` (optional, one run of like objects): poses @ get_tf_to_centered_mesh() of the LAST iteration, written by that pass' tail launch
struct RefineFinal {
  float *centered;
  float cneg[3];
};

static int refine_predict_impl(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, const fp_refine_cfg *cfg,
                               float *d_poses, int iteration, float *d_trans, float *d_rot, void *stream, const RefineFinal *fin, unsigned flags = 0) {
  FP_REQUIRE(ctx && net && cfg && d_poses, "fp_refine_predict_multi: null argument");
  FP_REQUIRE(iteration >= 0, "fp_refine_predict_multi: bad iteration");
  int N = 0;
  FP_TRY(check_objs(objs, n_obj, &N));
  if (N == 0 || iteration == 0) return FP_OK;
  hipStream_t s = (hipStream_t)stream;
  const int rot_dim = fp_net_rot_dim(net);
  const size_t rs_total = render_scratch_total(ctx, objs, n_obj);
  FP_TRY(fp_arena_ensure(ctx, pass_arena_bytes(N, rs_total) + (size_t)n_obj * ((size_t)4 << 20)));
  const size_t mark = ctx->arena.off;
  const size_t img = (size_t)160 * 160 * 8;
  static const bool no_shared = getenv("FP_NO_SHARED_B") != nullptr;       // A/B knob: ignore FP_REFINE_SHARED_TRANSLATION (identical results)
  auto body = [&]() -> int {
    TAKE(tf, float, (size_t)N * 9);
    TAKE(bbox, float, (size_t)N * 4);
    TAKE(trans, float, (size_t)N * 3);
    TAKE(rot, float, (size_t)N * 6);
    TAKE(net_in, f16, (size_t)2 * N * img);
    float *tr = d_trans ? d_trans : trans, *ro = d_rot ? d_rot : rot;
    const int n_runs = count_runs(objs, n_obj);
    TAKE(rscratch, char, rs_total);                              // reused by every iteration
    // One run of like objects in a batch that the trunk does not cut in two by hypotheses: the rendered side (crop window -> rasteriser ->
    // encodeA) and the observed side (crop window -> observed crop -> encodeA) are two chains on two streams up to the channel concat
    // (run_trunk): side B does not wait for the rasteriser.  Bit-identical to one chain.
    const bool two_sides = n_runs == 1 && N < fp_trunk_split_min() && !g_one_chain;
    // One run of like objects (one camera, one mesh, one diameter): the heads' token means, the pose update and the crop windows of
    // the next iteration are ONE launch behind the heads (refine_tail_kernel) instead of four.  Bit-identical (FP_TAIL_SPLIT=1: the four).
    static const bool tail_split = getenv("FP_TAIL_SPLIT") != nullptr;
    int first = 0;
    while (first < n_obj && objs[first].n == 0) ++first;
    const bool fused_tail = n_runs == 1 && !tail_split && fp_hyp_chunk(N) == N;
    // FP_REFINE_SHARED_TRANSLATION: in the FIRST iteration every hypothesis of an object has the crop window of the object's first one, so
    // side B - the observed crop and its way through encodeA - is ONE image per object: cropped and encoded once on the side stream
    // (fp_encode_side_b), copied into the B half of the channel concat by run_trunk.  One run of like objects, at most 8 of them, no
    // hypothesis chunks; otherwise the plain pass runs.
    int n_live = 0;
    for (int o = 0; o < n_obj; ++o) n_live += objs[o].n > 0;
    const bool shared0 = (flags & FP_REFINE_SHARED_TRANSLATION) && !no_shared && n_runs == 1 && n_live <= 8 && fp_hyp_chunk(N) == N;
    f16 *xB1 = nullptr, *featB = nullptr;
    if (shared0) {
      TAKE(xB1_, f16, (size_t)n_live * img);
      TAKE(featB_, f16, (size_t)n_live * 1600 * 128);
      xB1 = xB1_, featB = featB_;
    }
    for (int it = 0; it < iteration; ++it) {
      size_t voff = 0;
      int off = 0, k = 0;
      StreamFanout fo(ctx, s, two_sides ? 1 : n_runs);
      std::unique_ptr<StreamFanout> ab;
      const bool shared = shared0 && it == 0;
      SharedB sb;
      sb.feat = featB, sb.n_groups = 0;
      for (int o = 0; o < n_obj;) {       // per run of like objects: crop windows + render (side A); per object: observed crop (side B)
        int e = o + 1, cnt = objs[o].n;
        while (e < n_obj && same_render_key(objs[o], objs[e])) cnt += objs[e++].n;
        if (cnt > 0) {
          const fp_object_batch &ob = objs[o];
          hipStream_t so = fo.stream_for(k++);
          float *p = d_poses + (size_t)off * 16;
          if (!(fused_tail && it > 0))       // (from the second iteration on the previous pass' tail has written the windows)
            FP_TRY(launch_crop_window_tf(p, cnt, ob.K, cfg->crop_ratio, ob.mesh_diameter, 160, 160, tf + (size_t)off * 9, bbox + (size_t)off * 4, so));
          if (two_sides || shared) ab.reset(new StreamFanout(ctx, s, 2));          // forks behind the crop windows
          const size_t rsb = (render_scratch_bytes(cnt, ob.mesh->d.V, ob.mesh->d.F, 160, 160, ctx->num_cu) + 255) & ~(size_t)255;
          int rc = render_net_impl(ctx, ob.mesh, p, cnt, ob.K, ob.H, ob.W, bbox + (size_t)off * 4, 160, 160, ob.mesh_diameter,
                                   cfg->normalize_xyz, 0.001f, net_in + (size_t)off * img, rscratch + voff, rsb, so);
          voff += rsb;
          hipStream_t sb_stream = ab ? ab->stream_for(0) : so;
          for (int q = o; q < e && rc == FP_OK; ++q) {
            const fp_object_batch &oq = objs[q];
            if (oq.n == 0) continue;
            if (shared) {       // the object's ONE observed crop (window and translation of its first hypothesis = of all of them)
              sb.start[sb.n_groups] = off;
              rc = fp_crop_observed(ctx, oq.d_rgb, oq.d_geom, oq.H, oq.W, oq.K, tf + (size_t)off * 9, d_poses + (size_t)off * 16, 1, 160, 160, 0,
                                    oq.mesh_diameter, cfg->normalize_xyz, 1, xB1 + (size_t)sb.n_groups * img, sb_stream);
              ++sb.n_groups;
            } else {
              rc = fp_crop_observed(ctx, oq.d_rgb, oq.d_geom, oq.H, oq.W, oq.K, tf + (size_t)off * 9, d_poses + (size_t)off * 16, oq.n, 160, 160, 0,
                                    oq.mesh_diameter, cfg->normalize_xyz, 1, net_in + ((size_t)N + off) * img, sb_stream);
            }
            off += oq.n;
          }
          if (shared && rc == FP_OK) {
            sb.start[sb.n_groups] = off;
            rc = fp_encode_side_b(ctx, net, xB1, sb.n_groups, N, featB, sb_stream);
          }
          if (rc != FP_OK) {
            if (ab) (void)ab->join();
            (void)fo.join();
            return rc;
          }
        }
        o = e;
      }
      FP_TRY(fo.join());
      if (fused_tail) {
        const fp_object_batch &ob = objs[first];
        RefineTailArgs t;
        memset(&t, 0, sizeof(t));
        t.poses = d_poses;
        t.trans_tanh = cfg->trans_rep_tanh;
        t.tn0 = cfg->trans_normalizer[0], t.tn1 = cfg->trans_normalizer[1], t.tn2 = cfg->trans_normalizer[2];
        t.rot_normalizer = cfg->rot_normalizer;
        t.trans_scale = cfg->normalize_xyz ? (float)(ob.mesh_diameter / 2) : 1.f;
        for (int i = 0; i < 9; ++i) t.K[i] = (float)ob.K[i];
        t.resize = 160.f;
        t.tf = tf, t.bbox = bbox;
        t.next_window = it + 1 < iteration;
        t.win = crop_window_k(ob.K, cfg->crop_ratio, ob.mesh_diameter, 160, 160);
        if (fin && it + 1 == iteration) {
          t.centered = fin->centered;
          for (int c = 0; c < 3; ++c) t.cneg[c] = fin->cneg[c];
        }
        FP_TRY(fp_refine_forward_ab(ctx, net, net_in, N, tr, ro, s, ab.get(), &t, shared ? &sb : nullptr));
        continue;
      }
      FP_REQUIRE(!fin, "refine pass: the centred poses come from the fused tail launch (one run of like objects, FP_TAIL_SPLIT unset)");
      FP_TRY(fp_refine_forward_ab(ctx, net, net_in, N, tr, ro, s, ab.get(), nullptr, shared ? &sb : nullptr));     // ONE network pass for every object (joins `ab`)
      // pose update: one launch per run of objects with the same translation scale (one launch when they share a mesh)
      off = 0;
      for (int o = 0; o < n_obj;) {
        const float trans_scale = cfg->normalize_xyz ? (float)(objs[o].mesh_diameter / 2) : 1.f;
        int cnt = 0, e = o;
        // (trans_rep='deepim' also needs the object's intrinsics: one launch per object)
        while (e < n_obj && (cfg->normalize_xyz ? (float)(objs[e].mesh_diameter / 2) : 1.f) == trans_scale && (cfg->trans_rep_tanh != 2 || e == o))
          cnt += objs[e++].n;
        if (cnt > 0)
          FP_TRY(launch_pose_update(d_poses + (size_t)off * 16, tr + (size_t)off * 3, ro + (size_t)off * rot_dim, cnt, rot_dim,
                                    cfg->trans_rep_tanh, cfg->trans_normalizer[0], cfg->trans_normalizer[1], cfg->trans_normalizer[2],
                                    cfg->rot_normalizer, trans_scale, d_poses + (size_t)off * 16, s, tf + (size_t)off * 9, objs[o].K, 160.f));      // in place
        off += cnt;
        o = e;
      }
    }
    return FP_OK;
  };
  int rc = body();
  ctx->arena.off = mark;
  return rc;
}

extern "C" int fp_refine_predict_multi(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, const fp_refine_cfg *cfg,
                                       float *d_poses, int iteration, float *d_trans, float *d_rot, void *stream) {
  return refine_predict_impl(ctx, net, objs, n_obj, cfg, d_poses, iteration, d_trans, d_rot, stream, nullptr);
}

extern "C" int fp_refine_predict_multi_flags(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, const fp_refine_cfg *cfg,
                                             float *d_poses, int iteration, float *d_trans, float *d_rot, unsigned flags, void *stream) {
  FP_REQUIRE((flags & ~(unsigned)FP_REFINE_SHARED_TRANSLATION) == 0, "fp_refine_predict_multi_flags: unknown flag bits 0x%x", flags);
  return refine_predict_impl(ctx, net, objs, n_obj, cfg, d_poses, iteration, d_trans, d_rot, stream, nullptr, flags);
}

extern "C" int fp_refine_predict(fp_ctx *ctx, const fp_net *net, const fp_mesh *mesh, const float *d_rgb, const float *d_xyz_map,
                                 int H, int W, const double *K, double mesh_diameter, const fp_refine_cfg *cfg, float *d_poses, int N,
                                 int iteration, float *d_trans, float *d_rot, void *stream) {
  FP_REQUIRE(N >= 0, "fp_refine_predict: bad N");
  fp_object_batch ob = {mesh, d_rgb, d_xyz_map, H, W, K, mesh_diameter, N};
  return fp_refine_predict_multi(ctx, net, &ob, 1, cfg, d_poses, iteration, d_trans, d_rot, stream);
}

static int score_features_impl(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, double crop_ratio,
                               int normalize_xyz, const float *d_poses, float *d_feats, int feat_ld, bool with_pose, void *stream) {
  FP_REQUIRE(ctx && net && d_poses && d_feats, "fp_score_predict_features_multi: null argument");
  int N = 0;
  FP_TRY(check_objs(objs, n_obj, &N));
  if (N == 0) return FP_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t rs_total = render_scratch_total(ctx, objs, n_obj);
  FP_TRY(fp_arena_ensure(ctx, pass_arena_bytes(N, rs_total)));
  const size_t mark = ctx->arena.off;
  const size_t img = (size_t)160 * 160 * 8;
  auto body = [&]() -> int {
    TAKE(tf, float, (size_t)N * 9);
    TAKE(bbox, float, (size_t)N * 4);
    TAKE(net_in, f16, (size_t)2 * N * img);
    int off = 0, k = 0;
    TAKE(rscratch, char, rs_total);
    size_t voff = 0;
    const int n_runs = count_runs(objs, n_obj);
    const bool two_sides = n_runs == 1 && N < fp_trunk_split_min() && !g_one_chain;       // as in fp_refine_predict_multi
    StreamFanout fo(ctx, s, two_sides ? 1 : n_runs);
    std::unique_ptr<StreamFanout> ab;
    for (int o = 0; o < n_obj;) {
      int e = o + 1, cnt = objs[o].n;
      while (e < n_obj && same_render_key(objs[o], objs[e])) cnt += objs[e++].n;
      if (cnt > 0) {
        const fp_object_batch &ob = objs[o];
        hipStream_t so = fo.stream_for(k++);
        const float *p = d_poses + (size_t)off * 16;
        FP_TRY(launch_crop_window_tf(p, cnt, ob.K, crop_ratio, ob.mesh_diameter, 160, 160, tf + (size_t)off * 9, bbox + (size_t)off * 4, so));
        if (two_sides) ab.reset(new StreamFanout(ctx, s, 2));
        const size_t rsb = (render_scratch_bytes(cnt, ob.mesh->d.V, ob.mesh->d.F, 160, 160, ctx->num_cu) + 255) & ~(size_t)255;
        int rc = render_net_impl(ctx, ob.mesh, p, cnt, ob.K, ob.H, ob.W, bbox + (size_t)off * 4, 160, 160, ob.mesh_diameter, normalize_xyz, 0.1f,
                                 net_in + (size_t)off * img, rscratch + voff, rsb, so);
        voff += rsb;
        hipStream_t sb = ab ? ab->stream_for(0) : so;
        for (int q = o; q < e && rc == FP_OK; ++q) {
          const fp_object_batch &oq = objs[q];
          if (oq.n == 0) continue;
          rc = fp_crop_observed(ctx, oq.d_rgb, oq.d_geom, oq.H, oq.W, oq.K, tf + (size_t)off * 9, d_poses + (size_t)off * 16, oq.n, 160, 160, 1,
                                oq.mesh_diameter, normalize_xyz, 1, net_in + ((size_t)N + off) * img, sb);
          off += oq.n;
        }
        if (rc != FP_OK) {
          if (ab) (void)ab->join();
          (void)fo.join();
          return rc;
        }
      }
      o = e;
    }
    FP_TRY(fo.join());
    FP_TRY(fp_score_features_ab(ctx, net, net_in, N, d_feats, s, ab.get(), feat_ld, with_pose ? d_poses : nullptr));
    return FP_OK;
  };
  int rc = body();
  ctx->arena.off = mark;
  return rc;
}

extern "C" int fp_score_predict_features_multi(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, double crop_ratio,
                                               int normalize_xyz, const float *d_poses, float *d_feats, void *stream) {
  return score_features_impl(ctx, net, objs, n_obj, crop_ratio, normalize_xyz, d_poses, d_feats, 512, false, stream);
}

extern "C" int fp_score_predict_rows_multi(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, double crop_ratio,
                                           int normalize_xyz, const float *d_poses, float *d_rows, void *stream) {
  return score_features_impl(ctx, net, objs, n_obj, crop_ratio, normalize_xyz, d_poses, d_rows, 528, true, stream);
}

extern "C" int fp_score_predict_features(fp_ctx *ctx, const fp_net *net, const fp_mesh *mesh, const float *d_rgb, const float *d_depth,
                                         int H, int W, const double *K, double mesh_diameter, double crop_ratio, int normalize_xyz,
                                         const float *d_poses, int N, float *d_feats, void *stream) {
  FP_REQUIRE(N >= 0, "fp_score_predict_features: N<0");
  fp_object_batch ob = {mesh, d_rgb, d_depth, H, W, K, mesh_diameter, N};
  return fp_score_predict_features_multi(ctx, net, &ob, 1, crop_ratio, normalize_xyz, d_poses, d_feats, stream);
}

// ---- one tracking frame, every launch of it (src/estimater.py:250-268; n_hyp > 1: the multi-hypothesis mode of BASELINE configs[4]) ----
// hypotheses of the multi-hypothesis mode: R_i = dR_i R, t_i = t + dt_i (tracking.py), hypothesis 0 = the pose itself
__global__ void track_hypotheses_kernel(const float *__restrict__ P, const float *__restrict__ pose, int n, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = P + (size_t)i * 16;
  float *o = out + (size_t)i * 16;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      float acc = __fmul_rn(p[r * 4 + 0], pose[0 * 4 + c]);
      acc = __fadd_rn(acc, __fmul_rn(p[r * 4 + 1], pose[1 * 4 + c]));
      acc = __fadd_rn(acc, __fmul_rn(p[r * 4 + 2], pose[2 * 4 + c]));
      o[r * 4 + c] = acc;
    }
    o[r * 4 + 3] = __fadd_rn(pose[r * 4 + 3], p[r * 4 + 3]);
  }
  o[12] = 0.f, o[13] = 0.f, o[14] = 0.f, o[15] = 1.f;
}

extern "C" int fp_track_frame(fp_ctx *ctx, const fp_track_args *a, void *stream) {
  FP_REQUIRE(ctx && a, "fp_track_frame: null argument");
  FP_REQUIRE(a->struct_size == sizeof(fp_track_args), "fp_track_frame: fp_track_args.struct_size = %zu (this library knows %zu)", a->struct_size, sizeof(fp_track_args));
  FP_REQUIRE(a->refine_net && a->mesh && a->d_rgb && a->d_depth && a->K && a->refine_cfg && a->d_pose && a->d_pose_of_mesh && a->d_depth_f && a->d_xyz,
             "fp_track_frame: null field");
  FP_REQUIRE(a->H > 0 && a->W > 0 && a->iteration >= 1 && a->n_hyp >= 1, "fp_track_frame: bad H / W / iteration / n_hyp");
  FP_REQUIRE(!a->rgb_is_u8 || a->d_rgb_f, "fp_track_frame: a uint8 frame needs the float workspace d_rgb_f");
  const bool multi = a->n_hyp > 1;
  FP_REQUIRE(!multi || (a->score_net && a->d_perturb && a->d_poses && a->d_scores && a->d_best), "fp_track_frame: n_hyp > 1 needs score_net, d_perturb, d_poses, d_scores, d_best");
  hipStream_t s = (hipStream_t)stream;
  // depth prelude: erode -> bilateral -> back-projection with the float32 camera matrix (src/estimater.py:256-260), + uint8 -> float colours
  double K32[9];
  for (int i = 0; i < 9; ++i) K32[i] = (double)(float)a->K[i];
  FP_TRY(launch_depth_prefilter(a->d_depth, a->H, a->W, 0.001f, 0.8f, 100.f, 100.f, 2.f, 100000.f, K32, 3.0e38f, a->d_depth_f, a->d_xyz,
                                a->rgb_is_u8 ? (const uint8_t *)a->d_rgb : nullptr, a->rgb_is_u8 ? a->d_rgb_f : nullptr, s));
  const float *rgb_f = a->rgb_is_u8 ? a->d_rgb_f : (const float *)a->d_rgb;
  RefineFinal fin;
  for (int c = 0; c < 3; ++c) fin.cneg[c] = -a->model_center[c];
  if (!multi) {
    // track_one: the pose is refined IN PLACE; its last tail launch also writes pose @ get_tf_to_centered_mesh()
    fin.centered = a->d_pose_of_mesh;
    fp_object_batch ob = {a->mesh, rgb_f, a->d_xyz, a->H, a->W, a->K, a->mesh_diameter, 1};
    return refine_predict_impl(ctx, a->refine_net, &ob, 1, a->refine_cfg, a->d_pose, a->iteration, nullptr, nullptr, stream, &fin);
  }
  hipLaunchKernelGGL(track_hypotheses_kernel, dim3((a->n_hyp + 63) / 64), dim3(64), 0, s, a->d_perturb, a->d_pose, a->n_hyp, a->d_poses);
  FP_CHECK_HIP(hipGetLastError());
  fp_object_batch ob = {a->mesh, rgb_f, a->d_xyz, a->H, a->W, a->K, a->mesh_diameter, a->n_hyp};
  FP_TRY(refine_predict_impl(ctx, a->refine_net, &ob, 1, a->refine_cfg, a->d_poses, a->iteration, nullptr, nullptr, stream, nullptr));
  FP_TRY(fp_arena_ensure(ctx, (size_t)a->n_hyp * (512 + 1) * 4 + 4096));
  const size_t mark = ctx->arena.off;
  float *feats = (float *)ctx->arena.take((size_t)a->n_hyp * 512 * sizeof(float));
  float *logits = (float *)ctx->arena.take((size_t)a->n_hyp * sizeof(float));
  int rc = (feats && logits) ? FP_OK : FP_ENOMEM;
  if (rc != FP_OK) fp_set_error("fp_track_frame: arena exhausted");
  if (rc == FP_OK) {
    fp_object_batch od = {a->mesh, rgb_f, a->d_depth_f, a->H, a->W, a->K, a->mesh_diameter, a->n_hyp};
    rc = fp_score_predict_features_multi(ctx, a->score_net, &od, 1, a->score_crop_ratio, a->score_normalize_xyz, a->d_poses, feats, stream);
  }
  if (rc == FP_OK) {
    ScoreTailOut o;
    o.logits = logits, o.scores = a->d_scores, o.score_offset = 100.f, o.argmax = a->d_best;
    o.poses = a->d_poses, o.best_pose = a->d_pose, o.best_centered = a->d_pose_of_mesh;
    for (int c = 0; c < 3; ++c) o.cneg[c] = fin.cneg[c];
    rc = fp_score_tail_impl(ctx, a->score_net, feats, 512, 1, a->n_hyp, o, s);
  }
  ctx->arena.off = mark;
  return rc;
}

int conv_ksplit(const ConvArgs &a, int num_cu);      // conv.hip

// ---- building blocks ---------------------------------------------------------------------------------
extern "C" int fp_conv2d_f16(fp_ctx *ctx, const void *d_in, int Nimg, int H, int W, int Cin, const void *d_w_packed, const float *d_bias,
                             int Cout, int KH, int KW, int stride, int pad, const void *d_res, int relu, void *d_out, int out_f32,
                             void *stream) {
  FP_REQUIRE(ctx && d_in && d_w_packed && d_bias && d_out, "fp_conv2d_f16: null argument");
  FP_REQUIRE(stride >= 1 && pad >= 0 && Nimg >= 0, "fp_conv2d_f16: bad stride/pad/N");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = (const f16 *)d_in;
  a.w = (const f16 *)d_w_packed;
  a.bias = d_bias;
  a.res = (const f16 *)d_res;
  a.out = d_out;
  a.Nimg = Nimg;
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.KH = KH;
  a.KW = KW;
  a.stride = stride;
  a.pad = pad;
  a.Ho = (H + 2 * pad - KH) / stride + 1;
  a.Wo = (W + 2 * pad - KW) / stride + 1;
  a.Cout = Cout;
  a.Kpad = (KH * KW * Cin + 31) / 32 * 32;
  a.M = Nimg * a.Ho * a.Wo;
  a.relu = relu;
  a.out_mode = out_f32 ? 1 : 0;
  a.out_ld = Cout;
  a.split_m = 0x7fffffff;
  a.post_period = 1;
  a.tokens = 400;
  // a few images of a 3x3 stride-1 trunk layer: conv_small.hip on weights packed here, as inside the networks (a tracking frame)
  if (!out_f32 && conv_small_shape(a, ctx->num_cu)) {
    const size_t bytes = small_packed_halfs(Cout, Cin) * sizeof(f16);
    FP_TRY(fp_arena_ensure(ctx, bytes + 4096));
    const size_t mark = ctx->arena.off;
    f16 *pk = (f16 *)ctx->arena.take(bytes);
    int rc = pk ? small_pack_weights(a.w, Cout, Cin, a.Kpad, pk, (hipStream_t)stream) : FP_ENOMEM;
    a.wsm = pk;
    if (rc == FP_OK) rc = launch_conv(ctx, a, (hipStream_t)stream);
    ctx->arena.off = mark;
    return rc;
  }
  // launches of a few workgroups take the split-K form of the 3x3 stride-1 kernel, as inside the networks (2 .. 4 hypotheses)
  a.ksplit = out_f32 ? 0 : conv_ksplit(a, ctx->num_cu);
  if (a.ksplit > 1) {
    const size_t bytes = (size_t)a.ksplit * a.M * a.Cout * sizeof(float);
    FP_TRY(fp_arena_ensure(ctx, bytes + 4096));
    const size_t mark = ctx->arena.off;
    a.splitk = (float *)ctx->arena.take(bytes);
    int rc = a.splitk ? launch_conv(ctx, a, (hipStream_t)stream) : FP_ENOMEM;
    ctx->arena.off = mark;        // the scratch is consumed by the finishing pass on the same stream before anything else takes it
    return rc;
  }
  a.ksplit = 0;
  if (!out_f32 && a.M >= S2_MIN_PIXELS && s2_supported(a)) {       // the band-in-LDS form of the 3x3 stride-2 layers, as inside the networks
    const size_t bytes = s2_packed_halfs(Cout, Cin) * sizeof(f16);
    FP_TRY(fp_arena_ensure(ctx, bytes + 4096));
    const size_t mark = ctx->arena.off;
    f16 *pk = (f16 *)ctx->arena.take(bytes);
    int rc = pk ? s2_pack_weights(a.w, Cout, Cin, a.Kpad, pk, (hipStream_t)stream) : FP_ENOMEM;
    a.wpk = pk;
    if (rc == FP_OK) rc = launch_conv(ctx, a, (hipStream_t)stream);
    ctx->arena.off = mark;
    return rc;
  }
  return launch_conv(ctx, a, (hipStream_t)stream);
}

// Building block for the parity tests: the band-in-LDS form of the C -> C 3x3 stride-1 layers on 40x40 maps (conv_s1b.hip; C = 128 or 256),
// which the networks use for batches of more than 40 hypotheses (more than one round of the general kernel's tiles); same operands as fp_conv2d_f16 (w_packed [C][9 C] fp16).
extern "C" int fp_conv3x3_band_f16(fp_ctx *ctx, const void *d_in, int Nimg, int C, const void *d_w_packed, const float *d_bias, const void *d_res,
                                   int relu, void *d_out, void *stream) {
  FP_REQUIRE(ctx && d_in && d_w_packed && d_bias && d_out && Nimg >= 0, "fp_conv3x3_band_f16: bad argument");
  FP_REQUIRE(C == 128 || C == 256, "fp_conv3x3_band_f16: C=%d (128 or 256)", C);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = (const f16 *)d_in;
  a.w = (const f16 *)d_w_packed;
  a.bias = d_bias;
  a.res = (const f16 *)d_res;
  a.out = d_out;
  a.Nimg = Nimg;
  a.H = a.W = a.Ho = a.Wo = 40;
  a.Cin = a.Cout = C;
  a.KH = a.KW = 3;
  a.stride = 1;
  a.pad = 1;
  a.Kpad = 9 * C;
  a.M = Nimg * 1600;
  a.relu = relu;
  a.out_ld = C;
  a.split_m = 0x7fffffff;
  a.post_period = 1;
  a.tokens = 400;
  if (a.M == 0) return FP_OK;
  const size_t bytes = s2_packed_halfs(C, C) * sizeof(f16);
  FP_TRY(fp_arena_ensure(ctx, bytes + 4096));
  const size_t mark = ctx->arena.off;
  f16 *pk = (f16 *)ctx->arena.take(bytes);
  int rc = pk ? s2_pack_weights(a.w, C, C, a.Kpad, pk, (hipStream_t)stream, 2, 1) : FP_ENOMEM;
  a.wpk = pk;
  if (rc == FP_OK) rc = launch_conv_s1b(ctx, a, (hipStream_t)stream);
  ctx->arena.off = mark;
  return rc;
}

extern "C" int fp_conv3x3_wino_f16(fp_ctx *ctx, const void *d_in, int Nimg, int HW, int Cin, int Cout, const float *h_weight, const float *d_bias,
                                   const void *d_res, int relu, void *d_out, void *stream) {
  FP_REQUIRE(ctx && d_in && h_weight && d_bias && d_out && Nimg >= 0, "fp_conv3x3_wino_f16: bad argument");
  FP_REQUIRE((HW == 40 || HW == 20) && Cin % 32 == 0 && Cin >= 64 && Cout % 64 == 0, "fp_conv3x3_wino_f16: HW=%d Cin=%d Cout=%d unsupported", HW, Cin, Cout);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = (const f16 *)d_in;
  a.bias = d_bias;
  a.res = (const f16 *)d_res;
  a.out = d_out;
  a.Nimg = Nimg;
  a.H = a.W = a.Ho = a.Wo = HW;
  a.Cin = Cin, a.Cout = Cout;
  a.KH = a.KW = 3;
  a.stride = 1;
  a.pad = 1;
  a.Kpad = 9 * Cin;
  a.M = Nimg * HW * HW;
  a.relu = relu;
  a.out_ld = Cout;
  a.split_m = 0x7fffffff;
  a.post_period = 1;
  a.tokens = 400;
  if (a.M == 0) return FP_OK;
  std::vector<f16> hu(wino_packed_halfs(Cout, Cin));
  wino_pack_weights(h_weight, nullptr, Cout, Cin, hu.data());
  const size_t bytes = hu.size() * sizeof(f16);
  FP_TRY(fp_arena_ensure(ctx, bytes + 4096));
  const size_t mark = ctx->arena.off;
  f16 *pk = (f16 *)ctx->arena.take(bytes);
  int rc = pk ? FP_OK : FP_ENOMEM;
  if (rc == FP_OK && hipMemcpyAsync(pk, hu.data(), bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) rc = FP_EHIP;
  a.wwino = pk;
  if (rc == FP_OK) rc = launch_conv_wino(ctx, a, (hipStream_t)stream);
  if (rc == FP_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = FP_EHIP;      // (hu is a host temporary)
  ctx->arena.off = mark;
  return rc;
}

extern "C" int fp_attention_f16(fp_ctx *ctx, const void *d_qk, const void *d_vt, int B, int T, void *d_out, void *stream) {
  FP_REQUIRE(ctx && d_qk && d_vt && d_out, "fp_attention_f16: null argument");
  return launch_attention(ctx, (const f16 *)d_qk, (const f16 *)d_vt, B, T, (f16 *)d_out, (hipStream_t)stream);
}

// ---- host: mycpp.cluster_poses ------------------------------------------------------------------------
extern "C" int fp_cluster_poses(float angle_diff_deg, float dist_diff_m, const float *in, int n_in, const float *sym, int n_sym,
                                float *out) {
  FP_REQUIRE(in && sym && out && n_in >= 1 && n_sym >= 1, "fp_cluster_poses: bad argument");
  const float radian_thres = angle_diff_deg / 180.0f * (float)M_PI;
  int n_out = 0;
  auto keep = [&](const float *p) {
    memcpy(out + (size_t)n_out * 16, p, 16 * sizeof(float));
    ++n_out;
  };
  keep(in);
  for (int i = 1; i < n_in; ++i) {
    const float *cur = in + (size_t)i * 16;
    bool isnew = true;
    for (int c = 0; c < n_out && isnew; ++c) {
      const float *cl = out + (size_t)c * 16;
      float dx = cl[3] - cur[3], dy = cl[7] - cur[7], dz = cl[11] - cur[11];
      if (std::sqrt(dx * dx + dy * dy + dz * dz) >= dist_diff_m) continue;
      for (int k = 0; k < n_sym; ++k) {
        const float *tf = sym + (size_t)k * 16;
        float R[9];  // rotation block of cur @ tf
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) R[r * 3 + cc] = cur[r * 4] * tf[cc] + cur[r * 4 + 1] * tf[4 + cc] + cur[r * 4 + 2] * tf[8 + cc] + cur[r * 4 + 3] * tf[12 + cc];
        float tr = 0.f;  // trace(R * cl_R^T)
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) tr += R[r * 3 + cc] * cl[r * 4 + cc];
        float cs = (tr - 1.f) / 2.0f;
        cs = std::fmax(std::fmin(cs, 1.0f), -1.0f);
        if (std::acos(cs) < radian_thres) {
          isnew = false;
          break;
        }
      }
    }
    if (isnew) keep(cur);
  }
  return n_out;
}
