/* foundationpose_amd - C ABI of the MI355X (gfx950) render-and-compare hot path.
 *
 * The reference (SavaRobotics/FoundationPose) has no FFI for this path: its boundary is a set of
 * Python call signatures backed by third-party CUDA libraries.  Each entry point
 * below names the reference interface it replaces (file:line relative to the reference repo).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative FP_E* code otherwise; fp_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - `d_*` pointers are DEVICE pointers owned by the caller; `h_*` pointers are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Nothing synchronises
 *     the stream except where stated; all launch functions are hipGraph-capturable.
 *   - poses are row-major 4x4 float32 (ob_in_cam, OpenCV camera), K is row-major 3x3 float64.
 *   - "net tensor" = fp16 NHWC with C padded to 8: [n][160][160][8] = (r,g,b,x,y,z,0,0), the
 *     fused, network-ready form of the reference's (N,6,160,160) fp32 A/B tensors.
 */
#ifndef FOUNDATIONPOSE_AMD_H
#define FOUNDATIONPOSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the declarations of this header are its ONLY dynamic symbols. */
#pragma GCC visibility push(default)

#define FP_OK 0
#define FP_EINVAL (-1)   /* bad argument / shape */
#define FP_EHIP (-2)     /* HIP runtime error */
#define FP_ENOMEM (-3)
#define FP_EKEY (-4)     /* state_dict key missing / wrong shape */

typedef struct fp_ctx fp_ctx;    /* per-device context: workspace arena (replaces dr.RasterizeCudaContext, src/estimater.py:102,168) */
typedef struct fp_mesh fp_mesh;  /* device-resident mesh_tensors (src/Utils.py:104-130) */
typedef struct fp_net fp_net;    /* folded + packed network weights (load_state_dict, predict_pose_refine.py:138-141 / predict_score.py:151-154) */

const char *fp_last_error(void);
int fp_version(void);

/* ---- context ------------------------------------------------------------------------------- */
int fp_ctx_create(int device, fp_ctx **out);
int fp_ctx_destroy(fp_ctx *ctx);
/* Pre-size the activation arena for batches of up to `max_hyp` hypotheses so that no allocation
 * happens inside a timed / captured region. */
int fp_ctx_reserve(fp_ctx *ctx, int max_hyp);
/* Counts the (re)allocations of the arena.  A captured hipGraph of the launch functions holds arena addresses: it stays valid
 * while this number does not change (reserve enough before capturing). */
int fp_ctx_arena_generation(const fp_ctx *ctx);

/* ---- mesh: make_mesh_tensors (src/Utils.py:104-130); host pointers, copied to the device ---- */
int fp_mesh_create(fp_ctx *ctx, const float *h_pos, int V, const int32_t *h_faces, int F, const float *h_vnormals,
                   const float *h_vertex_color /* V*3 in [0,1] or NULL */,
                   const float *h_uv /* n_uv*2, v already flipped, or NULL */, int n_uv, const int32_t *h_uv_idx /* F*3 */,
                   const float *h_tex /* texH*texW*3 in [0,1] */, int texH, int texW, fp_mesh **out);
int fp_mesh_destroy(fp_mesh *mesh);

/* ---- a9: compute_crop_window_tf_batch(method='box_3d') (src/Utils.py:577-621) + bbox2d_ori
 *      (predict_pose_refine.py:44-45).  d_tf: N*9, d_bbox2d: N*4 (umin,vmin,umax,vmax). -------- */
int fp_crop_window_tf(fp_ctx *ctx, const float *d_poses, int N, const double *K, double crop_ratio, double mesh_diameter,
                      int out_w, int out_h, float *d_tf, float *d_bbox2d, void *stream);

/* ---- a11: nvdiffrast_render (src/Utils.py:133-219): fp32 channels-last maps, rows top-down,
 *      background 0.  Any output pointer may be NULL.  d_bbox2d may be NULL (full frame). ------- */
int fp_render(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
              const float *d_bbox2d, int out_h, int out_w, int use_light, float w_ambient, float w_diffuse,
              float *d_color /* N*h*w*3 */, float *d_depth /* N*h*w */, float *d_normal /* N*h*w*3 */,
              float *d_xyz /* N*h*w*3 */, void *stream);

/* The same with the non-default arguments of nvdiffrast_render: light_dir / light_pos / light_color (src/Utils.py:200-211)
 * and projection_mat (src/Utils.py:159-161).  opts == NULL is fp_render with use_light = 0. */
typedef struct fp_render_opts {
  size_t struct_size;      /* = sizeof(fp_render_opts) of the header the CALLER was built against: fields beyond it are read as 0 / NULL,
                            * a size the library does not know is refused (FP_EINVAL) */
  int use_light;
  float w_ambient, w_diffuse;
  int light_mode;          /* 0: light_dir = (0,0,1) (the default); 1: light_vec = -light_dir; 2: light_vec = light_pos (light_dir=None) */
  float light_vec[3];
  int has_light_color;     /* 0: light_color=None (the diffuse term takes the surface colour) */
  float light_color[3];
  int has_projection;      /* 1: projection (row-major 4x4, the reference's projection_mat) replaces the matrix derived from K */
  double projection[16];
  float *d_rast;           /* optional device output N*h*w*4: dr.rasterize's (u, v, z/w, triangle_id + 1) per pixel (src/Utils.py:182), rows flipped like
                            * the other outputs; NULL: not written.  What the parity tests compare coverage and the winning face on. */
} fp_render_opts;
int fp_render_ex(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                 const float *d_bbox2d, int out_h, int out_w, const fp_render_opts *opts,
                 float *d_color, float *d_depth, float *d_normal, float *d_xyz, void *stream);

/* ---- a10/a13/a18 side A fused: render + rgb scaling + xyz centring / normalising / invalidating
 *      (h5_dataset.py:92-99 | :151-156) straight into a net tensor.  invalid_thres = 0.001
 *      (refiner) or 0.1 (scorer). ------------------------------------------------------------- */
int fp_render_net(fp_ctx *ctx, const fp_mesh *mesh, const float *d_poses, int N, const double *K, int H, int W,
                  const float *d_bbox2d, int out_h, int out_w, double mesh_diameter, int normalize_xyz, float invalid_thres,
                  void *d_net_out /* fp16 N*h*w*8 */, void *stream);

/* ---- a12/a13/a14 side B fused: kornia.warp_perspective crops of the observed frame + the batch
 *      transform.  mode 0 = refiner (rgb bilinear + xyz_map nearest, predict_pose_refine.py:63,72;
 *      h5_dataset.py:101-112); mode 1 = scorer (rgb bilinear + depth nearest + the full-resolution
 *      depth->xyz round trip of h5_dataset.py:158-161, composed per pixel, never materialised).
 *      d_rgb: H*W*3 float [0,255]; d_geom: H*W*3 xyz_map (mode 0) or H*W depth (mode 1).
 *      out_fmt 0: fp32 planar N*6*h*w (the reference's cat([rgbB, xyz_mapB],1)); 1: fp16 net tensor. */
int fp_crop_observed(fp_ctx *ctx, const float *d_rgb, const float *d_geom, int H, int W, const double *K,
                     const float *d_tf, const float *d_poses, int N, int out_h, int out_w, int mode,
                     double mesh_diameter, int normalize_xyz, int out_fmt, void *d_out, void *stream);

/* ---- a12, the use_normal branch: kornia.warp_perspective(mode='nearest', align_corners=False, zeros padding) of a
 *      channel-last float image batch by the axis-aligned crop transforms (predict_pose_refine.py:74-76: normalAs from
 *      the rendered normals, normalBs from the frame's normal map).  d_src: src_batch x src_h x src_w x channels with
 *      src_batch == N, or 1 = one image for every transform (the reference's .expand(B,-1,-1,-1)); d_tf: N x 3x3;
 *      d_out: N x channels x out_h x out_w (planar, as the reference's BatchPoseData holds it). */
int fp_warp_nearest(fp_ctx *ctx, const float *d_src, int src_batch, int src_h, int src_w, int channels, const float *d_tf, int N,
                    int out_h, int out_w, float *d_out, void *stream);

/* ---- a6/a7/a8: depth pre-processing (src/Utils.py:304-438) ----------------------------------- */
int fp_erode_depth(fp_ctx *ctx, const float *d_depth, int H, int W, int radius, float depth_diff_thres, float ratio_thres,
                   float zfar, float *d_out, void *stream);
int fp_bilateral_filter_depth(fp_ctx *ctx, const float *d_depth, int H, int W, int radius, float zfar, float sigmaD,
                              float sigmaR, float *d_out, void *stream);
int fp_depth2xyzmap(fp_ctx *ctx, const float *d_depth, int H, int W, const double *K, float zfar, float *d_xyz, void *stream);
/* The depth prelude of a tracking frame (src/estimater.py:256-260: erode_depth -> bilateral_filter_depth -> depth2xyzmap_batch) in
   one launch; `radius` must be 2 (both filters, as the reference calls them).  d_depth_out (H,W) and d_xyz (H,W,3) are bit-identical
   to the three calls above chained; d_depth_out must not alias d_depth.  Optionally the frame's colours ride along: d_rgb_u8 (H,W,3)
   uint8 -> d_rgb_f32 (H,W,3) float (what fp_crop_observed and the fused passes read); both null: not done. */
int fp_depth_prefilter(fp_ctx *ctx, const float *d_depth, int H, int W, int radius, float depth_diff_thres, float ratio_thres,
                       float zfar_erode, float zfar_bilateral, float sigmaD, float sigmaR, const double *K, float zfar_xyz,
                       float *d_depth_out, float *d_xyz, const uint8_t *d_rgb_u8, float *d_rgb_f32, void *stream);
/* depth2xyzmap of the registration path (src/Utils.py:399-417): float64 arithmetic, one rounding to float32, depth < 0.001 -> 0. */
int fp_depth2xyzmap_f64(fp_ctx *ctx, const float *d_depth, int H, int W, const double *K, float *d_xyz, void *stream);
/* The reductions behind FoundationPose.guess_translation and the "valid too small" test of register()
 * (src/estimater.py:137-156,173-177) without a host copy of the depth image.  d_mask: H*W bytes, non-zero = object.
 * h_stats6 (host): cmin, cmax, rmin, rmax of mask > 0 (cmax = -1 if the mask is empty), count of mask > 0, count of usable
 * pixels (mask > 0 and depth >= min_depth); h_median (host): np.median of the usable depths (0 if none).  Synchronises. */
int fp_mask_depth_stats(fp_ctx *ctx, const float *d_depth, const uint8_t *d_mask, int H, int W, float min_depth, int32_t *h_stats6,
                        float *h_median, void *stream);

/* ---- networks -------------------------------------------------------------------------------- */
typedef struct {
  const char *name;    /* reference state_dict key, e.g. "encodeA.0.net.0.weight" */
  const float *data;   /* host fp32, contiguous */
  int ndim;
  int64_t shape[4];
} fp_tensor;

#define FP_NET_REFINE 0  /* RefineNet (learning/models/refine_network.py:27-93) */
#define FP_NET_SCORE 1   /* ScoreNetMultiPair (learning/models/score_network.py:28-90) */
/* Builds device weights from a reference-layout state_dict: BatchNorm (eval) folded into the
 * preceding conv, fp16 [Cout][tap][Cin] packing, attention / linear weights.  use_bn mirrors cfg.use_BN. */
int fp_net_create(fp_ctx *ctx, int kind, const fp_tensor *tensors, int n_tensors, int use_bn, fp_net **out);
int fp_net_destroy(fp_net *net);
int fp_net_rot_dim(const fp_net *net);   /* 3 (axis_angle) or 6 (6d), from rot_head.1.weight */

/* a15: RefineNet.forward.  d_net_in: fp16 net tensor [2N][160][160][8], A = first N images, B = last N. */
int fp_refine_forward(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_trans /* N*3 */,
                      float *d_rot /* N*rot_dim */, void *stream);
/* The shared trunk alone (encodeA/encodeAB | encoderA/encoderAB + pos_embed: refine_network.py:79-88, score_network.py:66-72):
 * d_tokens receives the (N,400,512) fp16 token tensor both heads read.  Exported for the parity tests, which compare it with
 * the reference module's encodeAB output. */
int fp_net_tokens(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, void *d_tokens /* fp16 N*400*512 */, void *stream);
/* a19: ScoreNetMultiPair.extract_feat -> d_feats N*512 fp32 */
int fp_score_features(fp_ctx *ctx, const fp_net *net, const void *d_net_in, int N, float *d_feats, void *stream);
/* a19/a20: att_cross + linear over `groups` objects of L hypotheses each (score_network.py:82-88),
 * logits groups*L; d_argmax (groups, may be NULL) = per-object argmax (predict_score.py:196). */
int fp_score_tail(fp_ctx *ctx, const fp_net *net, const float *d_feats, int groups, int L, float *d_logits,
                  int32_t *d_argmax, void *stream);

/* a16: pose update (predict_pose_refine.py:195-231 + so3_exp_map + egocentric_delta_pose_to_pose).
 * trans_rep_tanh: 1 -> tanh(trans)*trans_normalizer (normalize_xyz False), 0 -> raw.  rot_dim 3|6.
 * (fp_refine_cfg.trans_rep_tanh also takes 2 = trans_rep 'deepim', see fp_pose_update_deepim.) */
int fp_pose_update(fp_ctx *ctx, const float *d_poseA, const float *d_trans, const float *d_rot, int N, int rot_dim,
                   int trans_rep_tanh, const float *trans_normalizer3, float rot_normalizer, float trans_scale,
                   float *d_pose_out, void *stream);

/* a16, trans_rep='deepim' (predict_pose_refine.py:201-215): trans[:, :2] shifts the projected centre inside the crop (units of
 * input_resize), trans[:, 2] scales its depth; d_tf_to_crops N*9 are the crop transforms of the pass, K the intrinsics. */
int fp_pose_update_deepim(fp_ctx *ctx, const float *d_poseA, const float *d_trans, const float *d_rot, int N, int rot_dim,
                          const float *d_tf_to_crops, const double *K, float input_resize, float rot_normalizer, float trans_scale,
                          float *d_pose_out, void *stream);

/* a17: PoseRefinePredictor.predict inner loop (predict_pose_refine.py:182-234), `iteration` rounds of
 * crop-window -> render -> observed crop -> RefineNet -> pose update, entirely on the device.
 * d_poses is updated in place.  d_trans/d_rot (may be NULL) receive the last raw network outputs. */
typedef struct {
  double crop_ratio;
  int normalize_xyz;
  int trans_rep_tanh;
  float trans_normalizer[3];
  float rot_normalizer;
} fp_refine_cfg;
int fp_refine_predict(fp_ctx *ctx, const fp_net *net, const fp_mesh *mesh, const float *d_rgb, const float *d_xyz_map,
                      int H, int W, const double *K, double mesh_diameter, const fp_refine_cfg *cfg, float *d_poses, int N,
                      int iteration, float *d_trans, float *d_rot, void *stream);

/* Several objects (frames / meshes) in ONE pass: the render + crop stages run per object, the networks run once on the
 * concatenated hypotheses (they are object-agnostic).  This is BASELINE configs[3] (4 concurrent objects x 252) and what
 * a rank of the sharded multi-GPU job executes (its slice of every object).  d_poses / d_trans / d_rot / d_feats are the
 * concatenation over objects in order; objs[i].n hypotheses belong to object i. */
typedef struct {
  const fp_mesh *mesh;
  const float *d_rgb;      /* H*W*3 float [0,255] */
  const float *d_geom;     /* refine: xyz_map H*W*3; score: depth H*W */
  int H, W;
  const double *K;         /* host, 3x3 row-major */
  double mesh_diameter;
  int n;                   /* hypotheses of this object */
} fp_object_batch;
int fp_refine_predict_multi(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, const fp_refine_cfg *cfg,
                            float *d_poses, int iteration, float *d_trans, float *d_rot, void *stream);
/* The same with flags.  FP_REFINE_SHARED_TRANSLATION: the caller states that, on entry, every hypothesis of an object has the SAME
 * translation - what FoundationPose.register builds (src/estimater.py:126-135,196-199: the rotation grid around ONE guessed centre).  The
 * crop window depends on the translation only (src/Utils.py:577-621), so in the first iteration the observed side (predict_pose_refine.py:63,72
 * and its half of RefineNet.encodeA, refine_network.py:74-78) is one crop per object: it is cropped and encoded once instead of once per
 * hypothesis, by the same kernels - the refined poses are those of fp_refine_predict_multi bit for bit.  A false statement gives the
 * hypotheses of an object the first one's observed crop in iteration 1. */
#define FP_REFINE_SHARED_TRANSLATION 1u
int fp_refine_predict_multi_flags(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, const fp_refine_cfg *cfg,
                                  float *d_poses, int iteration, float *d_trans, float *d_rot, unsigned flags, void *stream);
int fp_score_predict_features_multi(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, double crop_ratio,
                                    int normalize_xyz, const float *d_poses, float *d_feats, void *stream);

/* a18-a20: ScorePredictor.predict up to per-hypothesis features (shardable), then the tail. */
int fp_score_predict_features(fp_ctx *ctx, const fp_net *net, const fp_mesh *mesh, const float *d_rgb, const float *d_depth,
                              int H, int W, const double *K, double mesh_diameter, double crop_ratio, int normalize_xyz,
                              const float *d_poses, int N, float *d_feats, void *stream);

/* ---- FoundationPose.track_one (src/estimater.py:250-268) as ONE call: depth prelude (erode -> bilateral -> back-projection, :256-260),
 *      refinement of the previous pose IN PLACE (:263), pose @ get_tf_to_centered_mesh() (:268) - every launch hand-written, hipGraph-
 *      capturable, no host synchronisation.  n_hyp > 1 (BASELINE configs[4], a build extension): d_perturb[i] applied to the previous pose
 *      (R_i = dR_i R, t_i = t + dt_i), all refined, scored (logits + 100, predict_score.py:209); the winner becomes d_pose. ------------- */
typedef struct fp_track_args {
  size_t struct_size;              /* = sizeof(fp_track_args) */
  const fp_net *refine_net, *score_net /* NULL when n_hyp == 1 */;
  const fp_mesh *mesh;
  const void *d_rgb;               /* H*W*3: uint8 (rgb_is_u8) or float [0,255] */
  int rgb_is_u8;
  const float *d_depth;            /* H*W raw depth, metres */
  int H, W;
  const double *K;                 /* host, 3x3 row-major */
  double mesh_diameter;
  const fp_refine_cfg *refine_cfg;
  double score_crop_ratio;
  int score_normalize_xyz;
  int iteration;
  int n_hyp;
  const float *d_perturb;          /* n_hyp*16 rigid perturbations, the first the identity; NULL when n_hyp == 1 */
  float model_center[3];           /* get_tf_to_centered_mesh() = translation by -model_center (src/estimater.py:82-86) */
  float *d_pose;                   /* 16: in = the previous frame's pose (centred mesh), out = this frame's */
  float *d_pose_of_mesh;           /* 16 out: d_pose @ get_tf_to_centered_mesh(), what track_one returns; device memory, or pinned host memory
                                      (hipHostMalloc: the last launch writes it there and the caller only waits for the stream) */
  float *d_poses, *d_scores;       /* n_hyp > 1: the refined hypotheses (n_hyp*16) and their scores (n_hyp) */
  int32_t *d_best;                 /* n_hyp > 1: index of the winner */
  float *d_depth_f, *d_xyz, *d_rgb_f; /* workspace: filtered depth H*W, xyz_map H*W*3, float colours H*W*3 (uint8 frames only) */
} fp_track_args;
int fp_track_frame(fp_ctx *ctx, const fp_track_args *args, void *stream);

/* fp_score_tail with a feature row stride (feat_ld >= 512 floats: 528 reads the [feature | pose] rows below in place) and, optionally
 * (d_scores != NULL), scores = logits + score_offset from the same launch (ScorePredictor.predict: + 100, predict_score.py:209) */
int fp_score_tail_scores(fp_ctx *ctx, const fp_net *net, const float *d_feats, int feat_ld, int groups, int L, float score_offset, float *d_logits,
                         float *d_scores, int32_t *d_argmax, void *stream);
/* fp_score_predict_features_multi writing the hypothesis-parallel job's all-gather records directly: d_rows (sum n) x 528 floats =
 * [feature 512 | pose 16] per hypothesis (ONE all-gather of these rows precedes the cross-hypothesis tail) */
int fp_score_predict_rows_multi(fp_ctx *ctx, const fp_net *net, const fp_object_batch *objs, int n_obj, double crop_ratio,
                                int normalize_xyz, const float *d_poses, float *d_rows, void *stream);

/* ---- building blocks exported for parity tests and profiling ---------------------------------- */
/* fp16 NHWC implicit-GEMM convolution on MFMA: out = act(conv(in, w) + bias [+ res]).  w_packed is
 * [Cout][Kpad] fp16 with k = (ky*KW+kx)*Cin + ci, Kpad = roundup(KH*KW*Cin, 32), zero padded. */
int fp_conv2d_f16(fp_ctx *ctx, const void *d_in, int Nimg, int H, int W, int Cin, const void *d_w_packed, const float *d_bias,
                  int Cout, int KH, int KW, int stride, int pad, const void *d_res, int relu, void *d_out, int out_f32,
                  void *stream);

/* The band-in-LDS form of the C -> C (C = 128 | 256) 3x3 stride-1 / pad-1 convolutions on 40x40 maps (csrc/conv_s1b.hip; the
 * ResnetBasicBlock convolutions of encodeA and of encodeAB's first stage, learning/models/network_modules.py:73-111): what the networks
 * run for batches of more than 40 hypotheses (more than one round of the general kernel's tiles).  Same operands as fp_conv2d_f16 (w_packed [C][9 C]); bit-identical to the general 3x3 stride-1
 * kernel behind fp_conv2d_f16. */
int fp_conv3x3_band_f16(fp_ctx *ctx, const void *d_in, int Nimg, int C, const void *d_w_packed, const float *d_bias, const void *d_res,
                        int relu, void *d_out, void *stream);
/* The Winograd F(2,3)-along-rows form of a 3x3 stride-1 / pad-1 convolution on HW x HW maps (HW = 40 | 20; csrc/conv_wino.hip), the same
 * ResnetBasicBlock convolutions: 2/3 of the matrix work.  h_weight: HOST fp32 (Cout, Cin, 3, 3) - the transformed weights u = G g are
 * rounded to fp16 once from fp32; in NHWC fp16; out = act(conv + bias [+ res]).  Its own numerics (u and v = B^T d in fp16): compared with
 * the fp32 convolution at a tolerance, not bit for bit with fp_conv2d_f16.  Synchronises the stream. */
int fp_conv3x3_wino_f16(fp_ctx *ctx, const void *d_in, int Nimg, int HW, int Cin, int Cout, const float *h_weight, const float *d_bias,
                        const void *d_res, int relu, void *d_out, void *stream);
/* fused multi-head self-attention core, 4 heads x 128: qk [M][1024] fp16 (q|k), vt [B][4][128][416] fp16 -> out [M][512] fp16.
 * vt is V transposed, token t of a hypothesis in column (t & ~15) | ((t>>2 & 1) << 3) | ((t>>3 & 1) << 2) | (t & 3)
 * (tokens of a group of 16 in the order 0-3, 8-11, 4-7, 12-15); columns of tokens >= T must hold zeros. */
int fp_attention_f16(fp_ctx *ctx, const void *d_qk, const void *d_vt, int B, int T, void *d_out, void *stream);

/* One nn.Linear(512, 512) on M tokens with the epilogue the transformer heads fuse behind it (csrc/tok_gemm.hip;
 * nn.TransformerEncoderLayer / nn.MultiheadAttention, refine_network.py:56-70, score_network.py:53-54):
 *   epilogue 0: out = [relu](x W^T + b), fp16 [M][512]
 *            1: the same values as the transposed V image [M/tokens][4][128][416] (layout: fp_attention_f16)
 *            2: out = LayerNorm(res + x W^T + b) * gamma + beta, fp16 [M][512] (statistics and residual in fp32)
 *            3: sums over groups of 16 tokens of the normalised rows (before gamma / beta), fp32 [M/16][512]
 *            4 / 5: epilogues 0 / 1 through the kernel the networks' in-projections run (csrc/tok_qkv.hip: a resident 128-token
 *               tile, every column block in one launch); bit-identical to 0 / 1
 *            6 / 7: epilogues 0 / 1 through the few-image form of that kernel (one or two hypotheses: 32-token tiles, K in four
 *               quarters added in order - its own fp32 summation order)
 * h_weight (512x512 row-major) / h_bias / h_gamma / h_beta are host fp32; synchronises the stream. */
int fp_token_linear_f16(fp_ctx *ctx, const void *d_in, int M, const float *h_weight, const float *h_bias, int epilogue, int relu,
                        const void *d_res, const float *h_gamma, const float *h_beta, int tokens, void *d_out, void *stream);

/* Building block: the part of one nn.TransformerEncoderLayer behind its attention core as the RefineNet heads run it, in one launch
 * (refine_network.py:56-70,88-91): x1 = LayerNorm1(tok + att W_out^T + b_out); ff = relu(x1 W1^T + b1); y = LayerNorm2(x1 + ff W2^T + b2)
 * WITHOUT its gamma / beta; d_gsum [M/16][512] fp32 = sums of y over groups of 16 tokens (the token mean, gamma2 / beta2 and the
 * output Linear follow in fp_refine_forward).  d_att / d_tok fp16 [M][512], M a multiple of 16; weights host fp32 row-major
 * (512x512) / (512); synchronises the stream. */
int fp_head_mlp_f16(fp_ctx *ctx, const void *d_att, const void *d_tok, int M, const float *h_w_out, const float *h_b_out,
                    const float *h_gamma1, const float *h_beta1, const float *h_w1, const float *h_b1, const float *h_w2,
                    const float *h_b2, float *d_gsum, void *stream);

/* mycpp.cluster_poses (mycpp/src/app/pybind_api.cpp:24-68); host function, float32 row-major 4x4.
 * h_out must hold n_in*16 floats; returns the number of kept poses (>=1) or a negative error. */
int fp_cluster_poses(float angle_diff_deg, float dist_diff_m, const float *h_poses_in, int n_in, const float *h_symmetry_tfs,
                     int n_sym, float *h_out);

/* timing helper: device time of a kernel class ("conv3x3_halo", "conv3x3_s2", "conv7x7", "linear", "attention", "render") over
 * the launches since the last reset, measured with (pooled) HIP events on the launch stream.  on = 0: off; 1: events around the
 * dominant class only (the 3x3 stride-1 convolutions: what a timed benchmark run carries); 2: around every class. */
int fp_prof_enable(fp_ctx *ctx, int on);
int fp_prof_read(fp_ctx *ctx, const char *kernel_class, double *total_ms, int64_t *launches, double *flops);
/* time during which at least one launch of the class was executing (union of the launch spans; = total_ms unless launches of the
 * class overlap on two streams): FLOPs / busy time is the rate the chip sustains on the class */
int fp_prof_read_busy(fp_ctx *ctx, const char *kernel_class, double *busy_ms);
int fp_prof_reset(fp_ctx *ctx);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif
