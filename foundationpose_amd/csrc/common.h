// Shared host-side plumbing for libfoundationpose_amd (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/foundationpose_amd.h"

typedef _Float16 f16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

void fp_set_error(const char *fmt, ...);

#define FP_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      fp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
      return FP_EHIP;                                                                        \
    }                                                                                        \
  } while (0)

#define FP_REQUIRE(cond, ...)                                                                \
  do {                                                                                       \
    if (!(cond)) {                                                                           \
      fp_set_error(__VA_ARGS__);                                                             \
      return FP_EINVAL;                                                                      \
    }                                                                                        \
  } while (0)

#define FP_TRY(expr)                 \
  do {                               \
    int _rc = (expr);                \
    if (_rc != FP_OK) return _rc;    \
  } while (0)

struct MeshDev {
  const float *pos, *vnormals, *vcolor, *uv, *tex;
  const int32_t *faces, *uv_idx;
  int V, F, texH, texW;
  const int4 *faces4;                   // the face indices as 16-byte records {i0, i1, i2, 0}: one load per face in the rasteriser
};

struct fp_mesh {
  MeshDev d;
  std::vector<void *> allocs;
};

struct ProfEntry {
  double total_ms = 0, flops = 0;
  int64_t launches = 0;
  std::vector<std::pair<float, float>> spans;   // (start, end) of every launch in ms since fp_ctx::ev_ref: launches of one class may overlap (two streams)
};

struct PendingEvent {
  hipEvent_t a, b;
  std::string cls;
  double flops;
};

// Bump arena over one hipMalloc; reset at the start of every forward.
struct Arena {
  char *base = nullptr;
  size_t cap = 0, off = 0;
  int generation = 0;      // bumped whenever the arena is (re)allocated: captured hipGraphs hold its addresses
  void *take(size_t bytes) {
    size_t o = (off + 255) & ~(size_t)255;
    if (o + bytes > cap) return nullptr;
    off = o + bytes;
    return base + o;
  }
};

struct fp_ctx {
  int device = 0;
  Arena arena;
  int reserved_hyp = 0;
  int prof = 0;            // 0 off, 1 events around the dominant kernel class only (3x3 stride-1 convolutions), 2 around every class
  std::map<std::string, ProfEntry> prof_tab;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> ev_pool;   // recycled timing events: a profiled launch costs two hipEventRecord, no create / destroy
  hipEvent_t ev_ref = nullptr;       // time origin of the launch spans, recorded when profiling is switched on
  int num_cu = 256;
  void *zero_page = nullptr;   // 4 KB of zeros: DMA source for out-of-image taps
  int *tail_counter = nullptr; // FP_TAIL_MAX_GROUPS zeros: arrival counters of the score tail's last launch (left at zero by every call)
  // side streams for the per-object stages (crop window, render, observed crop) of a multi-object pass: with 8 objects
  // x 32 hypotheses a render launch fills a quarter of the chip, so the objects' stages run side by side and join the
  // launch stream before the (single) network pass
  static constexpr int NSIDE = 8;
  hipStream_t side[NSIDE] = {};
  hipEvent_t ev_fork = nullptr, ev_join[NSIDE] = {};
  bool side_ready = false;
};

int fp_arena_ensure(fp_ctx *ctx, size_t bytes);
size_t fp_arena_bytes_for(int n_hyp);     // whole refine/score pass (outer buffers + network)
size_t fp_arena_inner_bytes(int n_hyp);   // network forward only

// Fork / join of independent kernel chains onto the context's side streams: the per-object stages of a multi-object pass,
// the two transformer heads of RefineNet.  With fewer than two chains everything stays on the launch stream.
struct StreamFanout {
  fp_ctx *ctx;
  hipStream_t main;
  bool fan;
  bool used[fp_ctx::NSIDE] = {};
  int rc = FP_OK;
  StreamFanout(fp_ctx *c, hipStream_t s, int n_active) : ctx(c), main(s), fan(n_active > 1) {
    if (!fan) return;
    if (!ctx->side_ready) {
      bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
      for (int i = 0; ok && i < fp_ctx::NSIDE; ++i)
        ok = hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        fan = false;          // no side streams: run the objects one after the other
        return;
      }
      ctx->side_ready = true;
    }
    if (hipEventRecord(ctx->ev_fork, main) != hipSuccess) fan = false;
  }
  hipStream_t stream_for(int k) {
    if (!fan) return main;
    const int i = k % fp_ctx::NSIDE;
    if (!used[i]) {
      used[i] = true;
      if (hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0) != hipSuccess) rc = FP_EHIP;
    }
    return ctx->side[i];
  }
  int join() {
    if (fan)
      for (int i = 0; i < fp_ctx::NSIDE; ++i)
        if (used[i]) {
          if (hipEventRecord(ctx->ev_join[i], ctx->side[i]) != hipSuccess || hipStreamWaitEvent(main, ctx->ev_join[i], 0) != hipSuccess) rc = FP_EHIP;
          used[i] = false;
        }
    if (rc != FP_OK) fp_set_error("stream fork/join failed");
    return rc;
  }
  // every way out of a scope that forked - an early FP_REQUIRE / FP_TRY return included - joins the side streams before the caller
  // resets the arena they may still be writing (join() is idempotent: a joined fan-out has no used stream left)
  ~StreamFanout() { (void)join(); }
  StreamFanout(const StreamFanout &) = delete;
  StreamFanout &operator=(const StreamFanout &) = delete;
};


// net.hip: the network passes with the two sides of encodeA as two chains (`ab` forked by the caller: side B's input is produced on
// ab->stream_for(0), side A's on `s`; nullptr: one chain); batches below fp_trunk_split_min() hypotheses only (larger ones are cut in two by hypotheses)
struct fp_net;
int fp_trunk_split_min();
struct RefineTailArgs;
int fp_hyp_chunk(int n_total);             // hypotheses per network pass (FP_CHUNK; default: the whole batch)
// `tail` (optional; pose / window fields filled by the caller, head fields by the forward pass): the heads' token means, the pose update
// and the next crop windows as ONE launch behind the join of the heads (refine_tail_kernel) instead of two mean_head launches here
// `sb` (optional): side B of this pass is ONE observed crop per object - every hypothesis of an object has the same crop window (the first
// iteration of a registration: the rotation grid around one guessed centre, src/estimater.py:126-135) - already encoded by fp_encode_side_b
// (on ab->stream_for(0) when `ab` is given): encodeA runs on side A only and the objects' features are copied into the B half of the channel
// concat.  Same kernels on the same inputs as the batch: the tokens are those of the plain pass bit for bit.
struct SharedB {
  const f16 *feat;          // [n_groups][40 x 40][128] fp16: encodeA of each object's observed crop
  int n_groups;
  int start[9];             // hypothesis at which object g starts; start[n_groups] = N
};
int fp_encode_side_b(fp_ctx *ctx, const fp_net *net, const f16 *xB, int n_groups, int hyp, f16 *feat, hipStream_t s);
int fp_refine_forward_ab(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_trans, float *d_rot, hipStream_t s, StreamFanout *ab,
                         RefineTailArgs *tail = nullptr, const SharedB *sb = nullptr);
int fp_score_features_ab(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_feats, hipStream_t s, StreamFanout *ab, int feat_ld = 512,
                         const float *d_poses = nullptr);      // feat_ld / d_poses: [feature | pose] rows of the all-gather (score_tail.hip)

// profiling hooks (events on the launch stream)
struct ProfScope {
  fp_ctx *ctx;
  hipStream_t s;
  PendingEvent ev;
  bool on;
  ProfScope(fp_ctx *c, hipStream_t st, const char *cls, double flops)
      : ctx(c), s(st), on(c && (c->prof == 2 || (c->prof == 1 && strcmp(cls, "conv3x3_halo") == 0))) {
    if (on) {
      ev.a = take_event();
      ev.b = take_event();
      ev.cls = cls;
      ev.flops = flops;
      (void)hipEventRecord(ev.a, s);
    }
  }
  hipEvent_t take_event() {
    hipEvent_t e = nullptr;
    if (!ctx->ev_pool.empty()) {
      e = ctx->ev_pool.back();
      ctx->ev_pool.pop_back();
    } else {
      (void)hipEventCreate(&e);
    }
    return e;
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(ev.b, s);
      ctx->pending.push_back(ev);
    }
  }
};

// XCD-aware tile order (MI355X: 8 XCDs with private L2s, workgroups dealt round-robin): workgroup `id` of `nwg`
// gets logical tile index L such that each XCD walks a CONTIGUOUS range of L; neighbouring tiles (which share
// halo rows / the same pixels for another cout tile) then hit the same L2.  Bijective for any nwg.  Speed only.
#ifdef __HIPCC__
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int xcd = id & 7, slot = id >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
#endif

// Kernels that need more dynamic LDS than the 64-KB default carry hipFuncAttributeMaxDynamicSharedMemorySize.  The attribute belongs to
// the function ON ONE DEVICE, so it is set for every such kernel of the library when a context is created on its device
// (fp_ctx_create -> fp_set_kernel_attributes), not lazily behind process-wide flags: a second device in one process, a graph capture
// and concurrent first launches from two threads all find the attributes in place.  Each .hip file lists its own kernels.
struct KernelLds {
  const void *fn;
  int bytes;
};
void conv_kernel_lds(std::vector<KernelLds> &v);        // conv.hip
void conv_halo_kernel_lds(std::vector<KernelLds> &v);   // conv_halo.hip
void conv_s1b_kernel_lds(std::vector<KernelLds> &v);    // conv_s1b.hip
void conv_small_kernel_lds(std::vector<KernelLds> &v);  // conv_small.hip
void conv_wino_kernel_lds(std::vector<KernelLds> &v);   // conv_wino.hip
void conv_s2_kernel_lds(std::vector<KernelLds> &v);     // conv_s2.hip
void stem_kernel_lds(std::vector<KernelLds> &v);        // stem.hip
void tok_gemm_kernel_lds(std::vector<KernelLds> &v);    // tok_gemm.hip
void tok_qkv_kernel_lds(std::vector<KernelLds> &v);     // tok_qkv.hip
void head_mlp_kernel_lds(std::vector<KernelLds> &v);    // head_mlp.hip
void attn_kernel_lds(std::vector<KernelLds> &v);        // attn.hip
void raster_kernel_lds(std::vector<KernelLds> &v);      // raster.hip
int fp_set_kernel_attributes(fp_ctx *ctx);              // api.hip: all of the above on ctx->device

// ---- kernel launchers implemented in the .hip files ----
struct ConvArgs {
  const f16 *in;
  const f16 *w;
  const float *bias;
  const f16 *res;
  const float *post_add;
  void *out;
  int Nimg, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, Kpad, M;
  int relu, out_mode;  // 0 fp16 NHWC, 1 fp32 NHWC, 2 fp16 V-transposed [b][4][128][416], token t in column vt_col(t)
  int out_ld, split_m, coff_hi, post_period, tokens;
  // split-K of the 3x3 stride-1 kernel for launches of a few workgroups (3 .. 4 hypotheses; 1 .. 2, a tracking frame, run conv_small.hip): `ksplit` workgroups share
  // the input-channel chunks of a tile and leave fp32 partial sums in `splitk` [ksplit][M][Cout]; a finishing pass adds them in a
  // fixed order and applies bias-free epilogue (the bias rides in split 0).  0 / nullptr: off.
  float *splitk = nullptr;
  int ksplit = 0;
  // 3x3 stride-2 layers: the weights in the fragment order of conv_s2.hip (s2_pack_weights); nullptr: the implicit-GEMM kernel runs
  const f16 *wpk = nullptr;
  // 3x3 stride-1 layers: the Winograd F(2,3)-along-rows image of the fp32 weights (conv_wino.hip: wino_pack_weights); nullptr: the direct kernels run
  const f16 *wwino = nullptr;
  // 3x3 stride-1 layers: the weights in the fragment order of conv_small.hip (small_pack_weights); nullptr: that form is not used
  const f16 *wsm = nullptr;
  // hypotheses of the network pass this launch belongs to (0: a stand-alone call): the few-image form is chosen by it, so that the two sides
  // of encodeA as one chain or two (twice the images per launch) take the same kernel
  int hyp = 0;
};
int launch_conv(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);
// conv_s2.hip: band-in-LDS form of the 3x3 stride-2 layers
#define S2_MIN_PIXELS 2000      // below (1 .. 4 hypotheses at 20x20 outputs: the sizes whose 3x3 stride-1 layers run split-K) the implicit-GEMM kernel with its 64-pixel tail tiles runs
bool s2_supported(const ConvArgs &a);
int s2_ct_for(int Cout);
size_t s2_packed_halfs(int Cout, int Cin);
int s2_pack_weights(const f16 *d_w, int Cout, int Cin, int Kpad, f16 *d_out, hipStream_t s, int ct_force = 0, int order = 0);   // ct_force: co tiles of 32 per wave group (0: s2_ct_for); order 1: conv_s1b.hip
int launch_conv_s2(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);
// conv_s1b.hip: band-in-LDS form of the 128 -> 128 stride-1 layers on 40x40 maps (bit-identical to the halo kernel; FP_C128_BAND=0: off)
bool s1b_supported(const ConvArgs &a);
int launch_conv_s1b(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);

// conv_small.hip: the 3x3 layers of the trunk in the passes of one or two hypotheses (a tracking frame): 32 x 32 tiles, K split over the waves of a workgroup
bool conv_small_shape(const ConvArgs &a, int num_cu);     // the layer shapes and launch sizes it runs
bool conv_small_use(const ConvArgs &a, int num_cu);       // ... and the caller holds the packed weights (ConvArgs::wsm)
size_t small_packed_halfs(int Cout, int Cin);
int small_pack_weights(const f16 *d_w, int Cout, int Cin, int Kpad, f16 *d_out, hipStream_t s);
int launch_conv_small(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);

// conv_wino.hip: Winograd F(2,3) along rows for the 3x3 stride-1 layers (its own numerics: transformed weights and inputs in fp16)
bool conv_wino_supported(const ConvArgs &a);
size_t wino_packed_halfs(int Cout, int Cin);
void wino_pack_weights(const float *w_oihw, const float *scale_per_cout, int Cout, int Cin, f16 *out_host);
int launch_conv_wino(fp_ctx *ctx, const ConvArgs &a, hipStream_t s);
int fp_wino_mode();            // conv.hip: FP_WINO (0: off, the default - measured slower or equal, profiles/r05_experiments.md; 1: every supported 3x3 stride-1 launch; 2: the 512-channel layers)

// Column of token t inside the transposed V image [b][4][128][416].  Within each group of 16 tokens the order is
// {0-3, 8-11, 4-7, 12-15}: the 8 keys that one lane half of the attention kernel's P^T operand carries (the S^T
// accumulator registers of a k-step) are then 16 contiguous bytes - one ds_read_b128 per V^T fragment.
__host__ __device__ __forceinline__ int vt_col(int t) { return (t & ~15) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 2) | (t & 3); }
int launch_attention(fp_ctx *ctx, const f16 *qk, const f16 *vt, int B, int T, f16 *out, hipStream_t s);

// Token GEMM (tok_gemm.hip): out = x (M x 512) @ W^T + b for up to TG_MAXBLK blocks of 512 output columns in one launch.
#define TG_MAXBLK 6
struct TokGemmBlock {
  const f16 *w;        // 512 x 512 weights in MFMA-fragment order (pack_tok_weights)
  const float *bias;   // 512
  void *out;           // fp16 rows (EPI_ROWS / EPI_LN) or the transposed V image (EPI_VT)
  int ld, coff;        // row stride / column offset of this block in `out` (halfs)
  int relu;
  int vt = 0;          // tok_qkv.hip: 1 = `out` is the transposed V image (tok_gemm.hip takes this from its epilogue argument)
};
struct TokGemmArgs {
  const f16 *in;       // [M][512] fp16
  int M, nblk, tokens; // tokens per hypothesis (EPI_VT, EPI_LNSUM)
  TokGemmBlock blk[TG_MAXBLK];
  const f16 *res;      // [M][512] residual (EPI_LN, EPI_LNSUM)
  const float *gamma, *beta;
  float *gsum;         // EPI_LNSUM: [M/16][512] sums of the normalised rows over groups of 16 tokens
};
enum { TG_EPI_ROWS = 0, TG_EPI_VT = 1, TG_EPI_LN = 2, TG_EPI_LNSUM = 3 };
int launch_tok_gemm(fp_ctx *ctx, const TokGemmArgs &a, int epi, hipStream_t s);
// tok_qkv.hip: the in-projections - all blocks over a resident 128-token tile (fp16 rows or the transposed V image per block)
int tok_qkv_small_max();        // passes of up to this many hypotheses run tok_qkv_small_kernel (FP_QKV_SMALL)
int launch_tok_qkv(fp_ctx *ctx, const TokGemmArgs &a, hipStream_t s, int hyp = 0);      // hyp: hypotheses of the network pass (1 .. 2: tok_qkv_small_kernel; 0: a stand-alone call)
// head_mlp.hip: out-projection + LayerNorm1 + linear1 + ReLU + linear2 + LayerNorm2 statistics of one transformer head in one launch
struct HeadMlpArgs {
  const f16 *att, *tok;        // [M][512]: attention output, residual tokens
  int M;
  const f16 *w_out, *w1, *w2;  // 512 x 512 in MFMA-fragment order (pack_tok_weights)
  const float *b_out, *b1, *b2, *g1, *be1;
  float *gsum;                 // [M/16][512] sums of the LayerNorm2-normalised rows over groups of 16 tokens
};
int launch_head_mlp(fp_ctx *ctx, const HeadMlpArgs &a, hipStream_t s);
// Sum of `nparts` consecutive partial rows per hypothesis (fixed order) / T, gamma, beta, Linear(512 -> out_dim)
int launch_mean_head(const float *partial, int nparts, const float *g, const float *b, int Bn, int T, const float *hw, const float *hb,
                     int out_dim, float *out, hipStream_t s);
// score_tail.hip: token mean + att.out_proj of ScoreNet in one launch; att_cross + linear + argmax in two
int launch_score_feat(const f16 *att, int Bn, int T, const float *wt, const float *bias, float *out, int ld, const float *poses, hipStream_t s);
struct ScoreTailOut {
  float *logits = nullptr;             // (groups, L), required
  int32_t *argmax = nullptr;           // (groups), optional
  float *scores = nullptr;             // (groups, L) = logits + score_offset, optional (ScorePredictor.predict: + 100, predict_score.py:209)
  float score_offset = 0.f;
  // optional, with argmax (tracking): the winning hypothesis' pose (group g: poses + g L 16) -> best_pose (groups, 16) and
  // best_pose @ get_tf_to_centered_mesh() -> best_centered (groups, 16)
  const float *poses = nullptr;
  float *best_pose = nullptr, *best_centered = nullptr;
  float cneg[3] = {0.f, 0.f, 0.f};
};
int launch_score_tail(const float *feats, int feat_ld, const float *wqk_t, const float *bqk, const double *u, const double *c, double b_eff, int groups, int L,
                      float *qk, double *sv, int *counter, const ScoreTailOut &o, hipStream_t s);
#define FP_TAIL_MAX_GROUPS 4096
int fp_score_tail_impl(fp_ctx *ctx, const fp_net *net, const float *d_feats, int feat_ld, int groups, int L, const ScoreTailOut &o, hipStream_t s);      // net.hip
// trans_tanh: 0 raw, 1 tanh * trans_normalizer, 2 trans_rep='deepim' (needs tf N x 9, K, resize = input_resize[0])
// The tail of a refinement pass as one launch (attn.hip: refine_tail_kernel): both heads' token mean + output Linear, the pose update in
// place and, with `next_window`, the crop windows of the next iteration (tf / bbox overwritten; the deepim branch reads tf first)
struct CropWindowK {             // intrinsics as float, radius = diameter * crop_ratio / 2, output size (crop_window_tf_one, pose_math.h)
  float k00, k01, k02, k10, k11, k12, k20, k21, k22, radius, ow, oh;
};
CropWindowK crop_window_k(const double *K, double crop_ratio, double diameter, int ow, int oh);
struct RefineTailArgs {
  const float *partial[2], *gam[2], *bet[2], *hw[2], *hb[2];      // per head: group sums of linear2's LayerNorm rows, ln2 gamma / beta, head Linear
  int nparts, T, rot_dim;
  float *trans, *rot;            // (N,3), (N,rot_dim): the heads' outputs (API outputs of the pass)
  float *poses;                  // (N,4,4) updated in place
  int trans_tanh;
  float tn0, tn1, tn2, rot_normalizer, trans_scale;
  float K[9], resize;            // deepim
  float *tf, *bbox;              // (N,9), (N,4)
  int next_window;
  CropWindowK win;
  // optional (the last pass of a tracking frame): pose @ get_tf_to_centered_mesh() of every hypothesis, (N,4,4) - src/estimater.py:268;
  // cneg = -model_center (the translation column of that matrix)
  float *centered = nullptr;
  float cneg[3] = {0.f, 0.f, 0.f};
};
int launch_refine_tail(const RefineTailArgs &a, int N, hipStream_t s);
int launch_pose_update(const float *poseA, const float *trans, const float *rot, int N, int rot_dim, int trans_tanh,
                       float tn0, float tn1, float tn2, float rot_normalizer, float trans_scale, float *out, hipStream_t s,
                       const float *tf = nullptr, const double *K = nullptr, float resize = 160.f);

struct RenderArgs {
  MeshDev mesh;
  const float *poses, *bbox2d;
  double K[9];
  int N, H, W, Ho, Wo;
  int use_light;
  float w_ambient, w_diffuse;
  float *color, *depth, *normal, *xyz;  // mode 0
  float *rast = nullptr;                // mode 0, optional: dr.rasterize's output per pixel (u, v, z/w, triangle id + 1) - src/Utils.py:182; parity tests
  f16 *net_out;                         // mode 1
  float mesh_diameter, invalid_thres;
  int normalize_xyz;
  void *scratch = nullptr;              // render_plan(...).total bytes: the per-(hypothesis, vertex) records of the pre-pass (required)
  size_t scratch_bytes = 0;
  int dbg = 0;                          // FP_RENDER_DBG (timing experiments: phases switched off)
  // non-default lighting / projection of nvdiffrast_render (src/Utils.py:159-162,200-211)
  int light_mode = 0;                   // 0: light_dir = (0,0,1), the default; 1: direction light_vec = -light_dir; 2: point light at light_vec (light_dir=None)
  float light_vec[3] = {0.f, 0.f, -1.f};
  int has_light_color = 0;              // 0: the diffuse term takes the surface colour (light_color=None)
  float light_color[3] = {1.f, 1.f, 1.f};
  int has_proj = 0;                     // 1: proj replaces projection_matrix_from_intrinsics(K, H, W, 0.001, 100)
  double proj[16] = {0};
};
struct RenderPlan {
  int S, strip_rows, lds_verts;         // strips per hypothesis, rows per strip, whether the triangle pass keeps the vertex records in LDS
  int G, Fg;                            // face ranges per hypothesis in the classification, faces per range
  size_t lds_bytes, a_lds, c_bytes, b_bytes, a_bytes, count_bytes, list_bytes, total;
  int solo;                             // one or two hypotheses: the whole render in the strip kernel's launch (raster.hip: render_kernel<.., true>)
  size_t solo_lds;
};
RenderPlan render_plan(int N, int V, int F, int Ho, int Wo, int num_cu);
// A render of N hypotheses goes out in sub-batches of render_chunk(...) when the worst-case scratch of all N at once (two face lists of
// every face in every strip: N * S * F * 8 B) would exceed 1 GiB (FP_RENDER_SCRATCH_MAX; tests lower it): the sub-batches run behind
// each other on the stream, over ONE scratch of render_scratch_bytes(...), with one plan (strips, face ranges) for all of them.
int render_chunk(int N, int V, int F, int Ho, int Wo, int num_cu);
size_t render_scratch_bytes(int N, int V, int F, int Ho, int Wo, int num_cu);
int launch_render(fp_ctx *ctx, const RenderArgs &a, hipStream_t s);      // a.scratch: render_scratch_bytes(a.N, ...) bytes
int launch_crop_window_tf(const float *poses, int N, const double *K, double crop_ratio, double diameter, int ow, int oh, float *tf,
                          float *bbox, hipStream_t s);

struct CropArgs {
  const float *rgb, *geom, *tf, *poses;
  double K[9];
  int H, W, N, Ho, Wo, mode, normalize_xyz, out_fmt;
  float mesh_diameter;
  void *out;
};
int launch_crop_observed(const CropArgs &a, hipStream_t s);
int launch_warp_nearest(const float *src, int src_batch, int Hs, int Ws, int C, const float *tf, int N, int Ho, int Wo, float *out,
                        hipStream_t s);

int launch_erode(const float *d, int H, int W, int radius, float diff_thres, float ratio_thres, float zfar, float *out, hipStream_t s);
int launch_bilateral(const float *d, int H, int W, int radius, float zfar, float sigmaD, float sigmaR, float *out, hipStream_t s);
int launch_depth2xyz(const float *d, int H, int W, const double *K, float zfar, float *xyz, hipStream_t s);
int launch_depth_prefilter(const float *d, int H, int W, float diff_thres, float ratio_thres, float zfar_e, float zfar_b, float sigmaD,
                           float sigmaR, const double *K, float zfar_x, float *out, float *xyz, const unsigned char *rgb_u8, float *rgb_f, hipStream_t s);
int launch_depth2xyz_f64(const float *d, int H, int W, const double *K, float *xyz, hipStream_t s);
int launch_mask_depth_stats(const float *d, const unsigned char *mask, int H, int W, float min_depth, int *out6, float *median, hipStream_t s);
