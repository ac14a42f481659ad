// Transformer-head kernels for gfx950: fused 400-token multi-head attention on MFMA, LayerNorm,
// token-mean + output heads, the cross-hypothesis score tail, argmax and the pose update.
//
// Reference: nn.TransformerEncoderLayer / nn.MultiheadAttention as used by
// learning/models/refine_network.py:56-70,88-91 and score_network.py:53-54,73-88;
// pose update: learning/training/predict_pose_refine.py:195-231 + pytorch3d so3_exp_map +
// src/Utils.py:848-855.
#include "common.h"
#include "pose_math.h"

#define AT_DH 128
#define AT_TP 416    // padded token count of the transposed V image [b][4][128][416]
#define AT_MAXT 400
#define FA_WAVES 7
#define FA_NSTAGE 3
#define FA_AHEAD 2      // key blocks the DMA runs ahead: block p + 2 goes to the ring slot of block p - 1 once every wave is past barrier p
#define FA_DMA_PER_WAVE 5   // ceil(32 DMA instructions per stage / 7 waves); surplus slots repeat an earlier one
#define FA_THREADS (FA_WAVES * 64)
#define FA_QB (FA_WAVES * 32)   // queries per workgroup
#define FA_KB 64                // keys per pipeline stage
#define FA_STAGE_HALFS (FA_KB * AT_DH * 2)   // K block [64][128] + V^T block [128][64]
#define FA_QW_HALFS (32 * AT_DH)             // a wave's Q^T staging area: 8 fragments x 64 lanes x 8 halfs (8 KB)
#define FA_LDS_BYTES ((FA_NSTAGE * FA_STAGE_HALFS + FA_WAVES * FA_QW_HALFS) * 2)

// LDS-DMA from inline asm (see conv_halo.hip: through the builtin, hipcc turns every later LDS-read wait into lgkmcnt(0));
// completion is waited for by the explicit s_waitcnt vmcnt(n) in front of the barriers of the key loop.
__device__ __forceinline__ void at_glds16(const f16 *g, f16 *l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)l);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory");
}

// Flash-style multi-head self-attention core (400 tokens, 4 heads x 128) on v_mfma_f32_32x32x16_f16.
// A work item = one (hypothesis, head, 224-query block): 13 query tiles of 32 split 7 + 6 over two items; each of the
// 7 waves owns 32 queries and keeps Q^T (32 regs), the running max / sum and O^T (4 x 32x32 accumulators) in registers.
// Keys stream through LDS in blocks of 64 (K block + V^T block = 32 KB) in a 3-deep ring filled by LDS-DMA TWO blocks
// ahead: counted s_waitcnt vmcnt(N) + raw s_barrier keep the later blocks in flight across the barrier (a __syncthreads()
// would drain them).
// PERSISTENT workgroups (one per CU, 152 KB LDS): workgroup j walks the items j, j + grid, j + 2 grid, ... and the DMA ring
// does not stop at an item boundary - the last two iterations of item i stage the first two key blocks of item i + 1, and the
// Q^T fragments of item i + 1 come by LDS-DMA into a per-wave 8-KB area (in fragment order: lane-linear, one conflict-free
// ds_read_b128 per fragment) right after the first S^T phase of item i has consumed its own.  As one item per workgroup a
// workgroup spent 17 k of its 47 k cycles on a start-up nothing covered (stamps: Q 7.5 k, DMA issue 2.9 k, block-0 wait 6.8 k).
// vmcnt accounting (loads, stores and LDS-DMA retire in issue order): per iteration a wave issues 5 DMA instructions (the
// block two ahead), in iteration 0 of an item 8 more (the next item's Q^T) after its S^T phase, and 8 stores at the end of
// an item (none if all 32 queries of the wave are past T); the wait in front of barrier p allows exactly the operations
// younger than block p that need not have landed (see `allowed` below).
//   S^T = K Q^T       : A = K rows from LDS (swizzled: chunk ^ (key&15), conflict-free b128), B = Q^T in registers;
//                       the lane that owns query column q sees 16 of each 32 keys, its partner lane+32 the rest.
//   online softmax     : per key block, max / sum finish with one xor-32 shuffle; exp((s-m)/sqrt(128)).
//   O^T += V^T P^T     : the S^T accumulator registers 8s..8s+7 ARE the B operand of k-step s (key order
//                       16s + {4h+0..3, 8+4h+0..3}); the transposed V image stores its tokens in that order
//                       (vt_col(), common.h), so a V^T fragment is ONE conflict-free ds_read_b128 like a K fragment.
//                       O^T keeps the query on the lane, so the rescale is lane-local.
// The loop is bound by vector-instruction ISSUE, not by the matrix pipe (SQ counters: 643 vector instructions per wave per
// key block beside 32 MFMAs, two waves sharing one SIMD's issue port), so the body is written to issue few of them:
// exp2 with the scale folded into one packed fma per two scores, v_max3, packed adds, tail masking only in the last key
// block, O^T rescaled only when some lane's running max moved, LDS fragment offsets and DMA gather offsets precomputed
// per lane (the DMA of a slot is scalar base + min(key, limit) * stride + lane constant - no branches in the loop).
// (Measured and dropped: waves 4-6 running the PV product of block k - 1 at the start of iteration k: the same time.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool V> struct FaBool { static constexpr bool value = V; };

#ifdef HALO_STAMP
// diagnostic build only (make -B EXTRA=-DHALO_STAMP): per-wave s_memtime stamps of the attention kernel, first 1024 workgroups:
// [0] entry, [31] exit; of the workgroup's SECOND item (its first when it has only one): [1] item start, then per key block
// {barrier passed, S done, softmax done, PV done}, [2] stores issued
__device__ unsigned long long g_attn_stamps[1024 * 7 * 32];
extern "C" __attribute__((visibility("default"))) int fp_dbg_attn_stamps(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)) == hipSuccess ? 0 : -1;
}
#define ASTAMP(i) do { if (blockIdx.x < 1024 && stamp_on) st[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

// fmaxf() canonicalises each MFMA output first (one extra v_max per score); v_max3 from asm does not.  The hazard
// recognizer does not look inside asm: an asm instruction must never be the FIRST reader of an MFMA result (the required
// wait states would be missing and it would read a half-written accumulator) - the callers chain every fa_max3 behind a
// compiler-visible instruction that reads the same accumulators (operand `a`).
__device__ __forceinline__ float fa_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ void at_glds16s(const f16 *sbase, unsigned voff_bytes, f16 *l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)l);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_bytes), "s"(sbase), "s"(la) : "memory");
}

__global__ __launch_bounds__(FA_THREADS, 2) void attention_kernel(const f16 *__restrict__ qk, const f16 *__restrict__ vt, int T,
                                                                  f16 *__restrict__ out, int n_items) {
  extern __shared__ __attribute__((aligned(16))) f16 smem[];
  const int nqb = (T + FA_QB - 1) / FA_QB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nkb = (T + FA_KB - 1) / FA_KB;
  const int G = gridDim.x;
  f16 *qlds = smem + FA_NSTAGE * FA_STAGE_HALFS + wave * FA_QW_HALFS;
#ifdef HALO_STAMP
  unsigned long long st[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) st[i] = 0;
  bool stamp_on = true;
#endif
  ASTAMP(0);

  // item v (virtual workgroup id: v & 7 = the XCD of this workgroup, the query blocks of one (hypothesis, head) share an XCD L2)
  auto item_of = [&](int v, int &b, int &h, int &qb) __attribute__((always_inline)) {
    const int L = xcd_remap(v, n_items);
    qb = L % nqb;
    h = (L / nqb) & 3;
    b = L / (nqb * 4);
  };

  // ---- DMA slots: instruction i = wave + 7u (mod 32) of a stage; i < 16 -> 64 lanes of the K block (64 keys x 16 chunks),
  // else of the V^T block (128 dims x 8 chunks of 8 keys).  Byte offset from the scalar base of a slot for key block kb:
  //   min(kb*64 + sx, limit) * stride + sadd     K: sx = key in block, limit T-1 (rows past T repeat the last one: their
  //   scores are masked in the tail block), stride 2048;  V^T: sx = first key of the chunk, limit 408 (the last chunk of
  //   the zero pad - P is exactly 0 past T, the operand must only be finite), stride 2.
  unsigned sx[FA_DMA_PER_WAVE], sadd[FA_DMA_PER_WAVE];
#pragma unroll
  for (int u = 0; u < FA_DMA_PER_WAVE; ++u) {
    int i = wave + u * FA_WAVES;
    if (i >= 32) i -= 32;                                    // surplus slot: repeat an earlier instruction (same bytes)
    if (i < 16) {
      const int c = i * 64 + lane, kl = c >> 4, chp = c & 15;
      sx[u] = kl;
      sadd[u] = ((chp ^ (kl & 15)) * 8) * 2;
    } else {
      const int c = (i - 16) * 64 + lane, d = c >> 3, chp = c & 7;
      sx[u] = (chp ^ ((d >> 1) & 7)) * 8;
      sadd[u] = d * AT_TP * 2;
    }
  }
  // staging cursor: the next key block the DMA brings in - (item s_v, block s_kb) into ring slot s_slot
  int s_v = blockIdx.x, s_kb = 0, s_slot = 0;
  const f16 *s_kbase, *s_vsrc;
  auto stage_bases = [&]() __attribute__((always_inline)) {
    int b, h, qb;
    item_of(s_v, b, h, qb);
    s_kbase = qk + (size_t)b * T * 1024 + 512 + h * AT_DH;              // K row `key` starts at kbase + key*1024
    s_vsrc = vt + ((size_t)b * 4 + h) * AT_DH * AT_TP;
  };
  stage_bases();
  auto stage_next = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < FA_DMA_PER_WAVE; ++u) {              // exactly FA_DMA_PER_WAVE per wave (vmcnt accounting)
      int i = wave + u * FA_WAVES;
      if (i >= 32) i -= 32;
      const bool is_k = i < 16;                              // wave-uniform
      const unsigned t = min((unsigned)(s_kb * FA_KB) + sx[u], is_k ? (unsigned)(T - 1) : 408u);
      const unsigned voff = t * (is_k ? 2048u : 2u) + sadd[u];
      at_glds16s(is_k ? s_kbase : s_vsrc, voff, smem + s_slot * FA_STAGE_HALFS + i * 512);
    }
    s_slot = s_slot + 1 == FA_NSTAGE ? 0 : s_slot + 1;
    if (++s_kb == nkb) {
      s_kb = 0;
      s_v += G;
      if (s_v < n_items) stage_bases();
    }
  };
  // Q^T of item v -> this wave's staging area, in fragment order: instruction s, lane (lr, lh) = dims 16 s + 8 lh .. + 8 of query lr.
  // Query rows past T repeat row T - 1 (finite operands; their columns are never stored).
  const unsigned qrow = wave * 32 + lr;
  auto stage_q = [&](int v) __attribute__((always_inline)) {
    int b, h, qb;
    item_of(v, b, h, qb);
    const f16 *qbase = qk + ((size_t)b * T + qb * FA_QB) * 1024 + h * AT_DH;
    const unsigned voff = min(qrow, (unsigned)(T - 1 - qb * FA_QB)) * 2048u + lh * 16;
#pragma unroll
    for (int s = 0; s < 8; ++s) at_glds16s(qbase + s * 16, voff, qlds + s * 512);
  };

  // ---- LDS fragment offsets (bytes, within a stage) ----
  //   K: row (kt*32 + lr), chunk (2*s + lh) ^ (lr & 15), s = 0..7; kt adds 8192 B
  //   V^T: row (dt*32 + lr), chunk (2*s + lh) ^ ((lr>>1)&7), s = 0..3 (k-steps of 16 keys); dt adds 4096 B
  unsigned koffb[8], voffb[4];
#pragma unroll
  for (int s = 0; s < 8; ++s) koffb[s] = (lr * AT_DH + (((2 * s + lh) ^ (lr & 15)) * 8)) * 2;
#pragma unroll
  for (int s = 0; s < 4; ++s) voffb[s] = FA_KB * AT_DH * 2 + (lr * FA_KB + (((2 * s + lh) ^ ((lr >> 1) & 7)) * 8)) * 2;

  const float c2 = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128): exp((s-m)/sqrt(128)) = exp2(s*c2 - m*c2)
  const f32x2 c2v = {c2, c2};

  // Prologue of the workgroup (the only start-up nothing covers): two key blocks, then Q^T of the first item.
  int inflight = 0;                                            // key blocks staged and not consumed yet (the current one included)
#pragma unroll
  for (int p = 0; p < FA_AHEAD; ++p)
    if (s_v < n_items) {
      stage_next();
      ++inflight;
    }
  stage_q((int)blockIdx.x);

  half8 qf[8];
  floatx16 oacc[4];
  float m_run, l_run;
  int c_slot = 0;                                              // ring slot of the block being consumed
  int st_prev = 0;                                             // stores this wave issued at the end of the previous item
  bool first = true;

  for (int v = blockIdx.x; v < n_items; v += G) {
    int b, h, qb;
    item_of(v, b, h, qb);
    const size_t rowbase = (size_t)b * T;
    const int q = qb * FA_QB + wave * 32 + lr;
    const bool wave_valid = qb * FA_QB + wave * 32 < T;         // wave-uniform: some query of this wave exists
    const bool has_next = v + G < n_items;
#ifdef HALO_STAMP
    stamp_on = v == (int)blockIdx.x + G || n_items <= G;
#endif
    ASTAMP(1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
    m_run = -1.0e30f;
    l_run = 0.f;

    auto body = [&](const int kb, auto first_c, auto tail_c) __attribute__((always_inline)) {
      constexpr bool KB0 = decltype(first_c)::value;           // kb == 0: this item's Q^T comes out of LDS, the next item's goes in
      constexpr bool TAIL = decltype(tail_c)::value;           // this block may hold keys >= T
      // Block (v, kb) must have landed - and in iteration 0 this item's Q^T.  `allowed` = operations issued after it that may stay in
      // flight: the next block (5), in iteration 0 the stores of the previous item (issued after that block's DMA), in iteration 1 the
      // next item's Q^T (issued in iteration 0 behind the block two ahead).  The first item's Q^T is the youngest operation: drain.
      int allowed = inflight > 1 ? 5 : 0;
      if (KB0) allowed = first ? 0 : (nkb == 1 ? st_prev : allowed + st_prev);
      else if (kb == 1 && has_next) allowed += 8;
      if (allowed == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (allowed == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
      else if (allowed == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      ASTAMP(3 + 4 * kb);
      // ring slot of block p - 1: every wave is past it
      if (s_v < n_items) {
        stage_next();
        ++inflight;
      }
      if constexpr (KB0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const half8 *>(qlds + s * 512 + lane * 8);
      }
      const char *sb = reinterpret_cast<const char *>(smem) + c_slot * (FA_STAGE_HALFS * 2);
      floatx16 sacc[2];
      // two independent accumulation chains (key tiles 0/1), fragments fetched four k-steps at a time so that
      // eight LDS reads are in flight before the first MFMA of the group issues
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        half8 kf[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) kf[kt][s] = *reinterpret_cast<const half8 *>(sb + koffb[sh * 4 + s] + kt * (32 * AT_DH * 2));
        __builtin_amdgcn_sched_barrier(0);                     // all eight reads issue before the first MFMA waits for one
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            if (sh == 0 && s == 0) {
              const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][s], qf[0], zero, 0, 0, 0);
            } else {
              sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][s], qf[sh * 4 + s], sacc[kt], 0, 0, 0);
            }
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      ASTAMP(4 + 4 * kb);
      // the S^T MFMAs have consumed all eight Q^T fragments (their ds_reads are complete): the staging area is free for the next item's
      if constexpr (KB0) {
        if (has_next) stage_q(v + G);
      }
      // ---- online softmax for query column lr ----
      if constexpr (TAIL) {                                   // keys >= T: score -1e30 -> out of the max, exp2 -> exactly 0
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * FA_KB + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sacc[kt][r] = key < T ? sacc[kt][r] : -1.0e30f;
          }
      }
      float bm = fmaxf(sacc[0][15], sacc[1][15]);              // compiler-visible first reader of both accumulators
      bm = fa_max3(bm, sacc[0][0], sacc[1][0]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) bm = fa_max3(bm, sacc[0][r], sacc[0][r + 1]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) bm = fa_max3(bm, sacc[1][r], sacc[1][r + 1]);
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float m_new = fmaxf(m_run, bm);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
      const float nmc = -m_new * c2;
      const f32x2 nmcv = {nmc, nmc};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 sv = {sacc[kt][r], sacc[kt][r + 1]};
          const f32x2 ev = __builtin_elementwise_fma(sv, c2v, nmcv);
          const f32x2 e = {__builtin_amdgcn_exp2f(ev.x), __builtin_amdgcn_exp2f(ev.y)};
          sacc[kt][r] = e.x;
          sacc[kt][r + 1] = e.y;
          ps2 += e;
        }
      float ps = ps2.x + ps2.y;
      ps += __shfl_xor(ps, 32);
      l_run = l_run * alpha + ps;
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {    // some lane's maximum moved: rescale O^T (alpha == 1 elsewhere)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
      }
      m_run = m_new;
      // ---- P^T, and O^T += V^T P^T ----
      half8 pf[2][2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[kt][s2][j] = (f16)sacc[kt][8 * s2 + j];
      __builtin_amdgcn_sched_barrier(0);
      ASTAMP(5 + 4 * kb);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        half8 vf[2][4];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) vf[s2][dt] = *reinterpret_cast<const half8 *>(sb + voffb[kt * 2 + s2] + dt * (32 * FA_KB * 2));
        __builtin_amdgcn_sched_barrier(0);                     // eight V^T reads in flight before the first MFMA waits
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s2][dt], pf[kt][s2], oacc[dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      ASTAMP(6 + 4 * kb);
      c_slot = c_slot + 1 == FA_NSTAGE ? 0 : c_slot + 1;
      --inflight;
    };
    // the last block always runs the tail form (the mask is the identity when T is a multiple of 64)
    if (nkb == 1) {
      body(0, FaBool<true>{}, FaBool<true>{});
    } else {
      body(0, FaBool<true>{}, FaBool<false>{});
      for (int kb = 1; kb < nkb - 1; ++kb) body(kb, FaBool<false>{}, FaBool<false>{});
      body(nkb - 1, FaBool<false>{}, FaBool<true>{});
    }

    // O^T accumulator: col = query lr, row = dim (r&3) + 8*(r>>2) + 4*lh of each 32-dim tile: a query's output row is split across
    // the two halves of the wave (lane: dims 8g .. 8g+3, lane + 32: 8g+4 .. 8g+7).  One v_permlane32_swap per dword and pair of
    // dim groups (g, g+1) leaves lanes 0-31 with dims 8g .. 8g+7 and lanes 32-63 with 8g+8 .. 8g+15: 8 16-byte stores per lane instead
    // of 16 8-byte ones (the store tail is issue-bound: cdna_hip_programming.md T21).  The swaps need every lane: only the store is masked.
    if (wave_valid) {                                            // wave-uniform: exactly 8 store instructions, or none (vmcnt accounting)
      const float inv = 1.f / l_run;
      f16 *orow = out + (rowbase + min(q, T - 1)) * 512 + h * AT_DH + lh * 8;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; gq += 2) {
          half4 ha, hb;
#pragma unroll
          for (int j = 0; j < 4; ++j) ha[j] = (f16)(oacc[dt][gq * 4 + j] * inv), hb[j] = (f16)(oacc[dt][(gq + 1) * 4 + j] * inv);
          const uint2 ua = __builtin_bit_cast(uint2, ha), ub = __builtin_bit_cast(uint2, hb);
          const auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
          const uint4 ov = {r0[0], r1[0], r0[1], r1[1]};
          if (q < T) *reinterpret_cast<uint4 *>(orow + dt * 32 + gq * 8) = ov;
        }
    }
    ASTAMP(2);
    st_prev = wave_valid ? 8 : 0;
    first = false;
  }
#ifdef HALO_STAMP
  if (blockIdx.x < 1024 && lane == 0) {
    st[31] = __builtin_amdgcn_s_memtime();
    unsigned long long *o = g_attn_stamps + ((size_t)blockIdx.x * 7 + wave) * 32;
#pragma unroll
    for (int i = 0; i < 32; ++i) o[i] = st[i];
  }
#endif
}

// The same attention for ONE or TWO hypotheses (a tracking frame, src/estimater.py:250-268).  There the kernel above is 8 workgroups whose
// waves each walk all seven key blocks of their 32 queries one after the other behind a 17 k-cycle start-up: 19 us for 0.33 GFLOP.
// Here a workgroup is one (hypothesis, head, 32-query tile) - 52 per hypothesis - and its seven waves take ONE key block of 64 each
// (split-K attention): S^T = K Q^T, a block-local softmax (maximum m_w, sum l_w, P in fp16 relative to m_w), O_w^T = V^T P^T, all operands
// through a wave-private LDS area (coalesced loads, fragments by ds_read_b128), then the seven partial results are merged through LDS:
// m = max m_w, O = sum_w 2^((m_w - m) c) O_w / sum_w 2^((m_w - m) c) l_w.  Same products and the same fp16 rounding of P as above, relative
// to the block's maximum instead of the running one: a summation order of its own (the few-image size class, like conv_small.hip).
#define FS_WAVES 7
#define FS_KP 136                                  // halfs per key row of a wave's K block in LDS (128 dims + 8: conflict-free fragment reads)
#define FS_VP 72                                   // halfs per dim row of its V^T block (64 keys + 8)
#define FS_AREA_BYTES (AT_DH * FS_VP * 2)          // a wave's LDS area: K block (64 x 136 halfs), then V^T block (128 x 72), then its partial O^T (128 x 32 floats)
#define FS_LDS_BYTES (FS_WAVES * FS_AREA_BYTES + 2 * FS_WAVES * 32 * 4)
static_assert(64 * FS_KP * 2 <= FS_AREA_BYTES && AT_DH * 32 * 4 <= FS_AREA_BYTES, "attention_small: wave area");
__global__ __launch_bounds__(FS_WAVES * 64) void attention_small_kernel(const f16 *__restrict__ qk, const f16 *__restrict__ vt, int T, f16 *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char fs_lds[];     // per wave: K block -> V^T block -> partial O^T [dim][query]; then [wave][query] m and l
  float *fs_m = reinterpret_cast<float *>(fs_lds + FS_WAVES * FS_AREA_BYTES), *fs_l = fs_m + FS_WAVES * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 31, lh = lane >> 5;
  const int nqt = (T + 31) >> 5, nkb = (T + 63) >> 6;
  const int qt = blockIdx.x % nqt, h = (blockIdx.x / nqt) & 3, b = blockIdx.x / (nqt * 4);
  const float c2 = 0.08838834764831845f * 1.4426950408889634f;      // log2(e) / sqrt(128)
  f16 *area = reinterpret_cast<f16 *>(fs_lds + (size_t)w * FS_AREA_BYTES);
  if (w < nkb) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // K block and V^T block of this wave: coalesced 16-byte loads (a key row is 256 contiguous bytes, a dim row of the block 128), all
    // requested up front; (first form, measured: the fragments as per-lane gathers, 40 per wave at the L1's tag rate: 12.3 us)
    u32x4 kq[16], vq[16];
    {
      const f16 *kbase = qk + (size_t)b * T * 1024 + 512 + h * AT_DH;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = lane + 64 * u, key = min(w * 64 + (i >> 4), T - 1);       // rows past T repeat the last one (masked below)
        kq[u] = *reinterpret_cast<const u32x4 *>(kbase + (size_t)key * 1024 + (i & 15) * 8);
      }
      const f16 *vbase = vt + ((size_t)b * 4 + h) * AT_DH * AT_TP;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = lane + 64 * u, col = min(w * 64 + (i & 7) * 8, 408);      // chunks past the image's zero pad read its last one (P = 0 there)
        vq[u] = *reinterpret_cast<const u32x4 *>(vbase + (size_t)(i >> 3) * AT_TP + col);
      }
    }
    half8 qf[8];
    {
      const f16 *qrow = qk + ((size_t)b * T + min(qt * 32 + lr, T - 1)) * 1024 + h * AT_DH + lh * 8;
#pragma unroll
      for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + s * 16);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = lane + 64 * u;
      *reinterpret_cast<u32x4 *>(area + (i >> 4) * FS_KP + (i & 15) * 8) = kq[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (the wave reads what it wrote itself: no barrier)
    half8 kf[2][8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 8; ++s) kf[kt][s] = *reinterpret_cast<const half8 *>(area + (kt * 32 + lr) * FS_KP + s * 16 + lh * 8);
    floatx16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][s], qf[s], sacc[kt], 0, 0, 0);
    }
    // keys >= T: score -1e30 -> out of the maximum, exp2 -> exactly 0
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = w * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        sacc[kt][r] = key < T ? sacc[kt][r] : -1.0e30f;
      }
    float bm = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) bm = fmaxf(bm, fmaxf(sacc[0][r], sacc[1][r]));
    bm = fmaxf(bm, __shfl_xor(bm, 32));
    const float nmc = -bm * c2;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], c2, nmc));
        sacc[kt][r] = e;
        ps += e;
      }
    ps += __shfl_xor(ps, 32);
    // the K fragments are in registers: the area takes the V^T block (rows = dims, the block's 64 keys in the image's token order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = lane + 64 * u;
      *reinterpret_cast<u32x4 *>(area + (i >> 3) * FS_VP + (i & 7) * 8) = vq[u];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    floatx16 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (w * 64 + kt * 32 >= T) continue;                    // (wave-uniform) a key tile past T: P is exactly 0
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        half8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (f16)sacc[kt][8 * s2 + j];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8 *>(area + (dt * 32 + lr) * FS_VP + kt * 32 + s2 * 16 + lh * 8), pf,
                                                            oacc[dt], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the V^T block is read: the area takes the partial O^T
    float *po = reinterpret_cast<float *>(area) + lr;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) po[(dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32] = oacc[dt][r];
    if (lh == 0) fs_m[w * 32 + lr] = bm, fs_l[w * 32 + lr] = ps;
  }
  __syncthreads();
  if (tid >= 256) return;
  const int ql = tid & 31, dg = tid >> 5, q = qt * 32 + ql;        // query ql, dims 16 dg .. + 16
  if (q >= T) return;
  float m = fs_m[ql];
  for (int k = 1; k < nkb; ++k) m = fmaxf(m, fs_m[k * 32 + ql]);
  float l = 0.f, o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  for (int k = 0; k < nkb; ++k) {
    const float sc = __builtin_amdgcn_exp2f((fs_m[k * 32 + ql] - m) * c2);
    l += fs_l[k * 32 + ql] * sc;
    const float *pk = reinterpret_cast<const float *>(fs_lds + (size_t)k * FS_AREA_BYTES) + dg * 16 * 32 + ql;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = fmaf(pk[i * 32], sc, o[i]);
  }
  const float inv = 1.f / l;
  half8 h0, h1;
#pragma unroll
  for (int i = 0; i < 8; ++i) h0[i] = (f16)(o[i] * inv), h1[i] = (f16)(o[8 + i] * inv);
  f16 *orow = out + ((size_t)b * T + q) * 512 + h * AT_DH + dg * 16;
  *reinterpret_cast<half8 *>(orow) = h0;
  *reinterpret_cast<half8 *>(orow + 8) = h1;
}

void attn_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)attention_kernel, (int)FA_LDS_BYTES});
  v.push_back({(const void *)attention_small_kernel, (int)FS_LDS_BYTES});
}

int launch_attention(fp_ctx *ctx, const f16 *qk, const f16 *vt, int B, int T, f16 *out, hipStream_t s) {
  FP_REQUIRE(T > 0 && T <= AT_MAXT, "attention: T=%d must be in [1,%d]", T, AT_MAXT);
  if (B == 0) return FP_OK;
  static const int small_max = getenv("FP_ATTN_SMALL") ? atoi(getenv("FP_ATTN_SMALL")) : 2;      // hypotheses up to which the split-K form runs (0: off)
  if (B <= small_max) {
    ProfScope ps(ctx, s, "attention", 4.0 * B * 4 * (double)T * T * AT_DH);
    hipLaunchKernelGGL(attention_small_kernel, dim3(((T + 31) / 32) * 4 * B), dim3(FS_WAVES * 64), FS_LDS_BYTES, s, qk, vt, T, out);
    FP_CHECK_HIP(hipGetLastError());
    return FP_OK;
  }
  const int n_items = ((T + FA_QB - 1) / FA_QB) * 4 * B;
  const int grid = n_items < ctx->num_cu ? n_items : ctx->num_cu;        // one persistent workgroup per CU (152 KB of LDS each)
  ProfScope ps(ctx, s, "attention", 4.0 * B * 4 * (double)T * T * AT_DH);
  hipLaunchKernelGGL(attention_kernel, dim3(grid), dim3(FA_THREADS), FA_LDS_BYTES, s, qk, vt, T, out, n_items);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Token mean + output Linear behind the final LayerNorm of a RefineNet head:
// mean_t(Linear(LN(x_t))) == Linear(mean_t LN(x_t))  (refine_network.py:90-91).  The LayerNorm itself runs in the epilogue of
// linear2 (tok_gemm.hip, EPI_LNSUM), which leaves the sums of the normalised rows over groups of 16 tokens; one small workgroup
// per hypothesis adds these partial rows in a fixed order (deterministic), applies gamma / beta and the output Linear.
// (a) the 512 features of hypothesis b: partial rows added in a fixed order, gamma / beta applied -> meanv.  NH heads at once: every
// load of every head is in flight before the first add (one memory round trip; a loop that waits per load was 25 of them)
#define MH_MAXPARTS 32
template <int NH>
__device__ __forceinline__ void mean_head_rows(float (*meanv)[512], int b, const float *const *partial, int nparts, const float *const *gam,
                                               const float *const *bet, int T) {
  const int f = threadIdx.x;
  float v[NH][MH_MAXPARTS];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int u = 0; u < MH_MAXPARTS; ++u) v[h][u] = partial[h][((size_t)b * nparts + min(u, nparts - 1)) * 512 + f];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < MH_MAXPARTS; ++u)
      if (u < nparts) m += v[h][u];
    meanv[h][f] = m * (1.f / (float)T) * gam[h][f] + bet[h][f];
  }
}
// (b) output `o` of the head Linear by one wave
__device__ __forceinline__ float mean_head_out(const float *meanv, const float *__restrict__ hw, const float *__restrict__ hb, int o, int lane) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += meanv[lane * 8 + i] * hw[(size_t)o * 512 + lane * 8 + i];
  s = wave_sum(s);
  return s + hb[o];
}

__global__ __launch_bounds__(512) void mean_head_kernel(const float *__restrict__ partial, int nparts, const float *__restrict__ gam,
                                                        const float *__restrict__ bet, int T, const float *__restrict__ hw,
                                                        const float *__restrict__ hb, int out_dim, float *__restrict__ out) {
  __shared__ float meanv[1][512];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *p1[1] = {partial}, *g1[1] = {gam}, *b1[1] = {bet};
  mean_head_rows<1>(meanv, b, p1, nparts, g1, b1, T);
  __syncthreads();
  for (int o = wave; o < out_dim; o += 8) {
    const float r = mean_head_out(meanv[0], hw, hb, o, lane);
    if (lane == 0) out[(size_t)b * out_dim + o] = r;
  }
}

// The tail of a refinement pass in ONE launch behind the join of the two heads: token mean + output Linear of the translation head
// and of the rotation head (two mean_head_kernel launches), the pose update (pose_update_kernel) and - when another iteration
// follows - its crop windows (crop_window_tf_kernel): four dependent launches of a few microseconds each, a workgroup per hypothesis.
// Every piece is the building block's own device function: the results are those of the four launches bit for bit.
__global__ __launch_bounds__(512) void refine_tail_kernel(RefineTailArgs a) {
  __shared__ float meanv[2][512], delta[2][8];      // delta: this hypothesis' head outputs (also written to a.trans / a.rot)
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  mean_head_rows<2>(meanv, b, a.partial, a.nparts, a.gam, a.bet, a.T);
  __syncthreads();
  for (int o = wave; o < 3 + a.rot_dim; o += 8) {
    const int h = o >= 3, oh = h ? o - 3 : o;
    const float r = mean_head_out(meanv[h], a.hw[h], a.hb[h], oh, lane);
    if (lane == 0) {
      (h ? a.rot + (size_t)b * a.rot_dim : a.trans + (size_t)b * 3)[oh] = r;
      delta[h][oh] = r;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  DeepimArgs dp;
  dp.tf = a.tf;
  dp.resize = a.resize;
  for (int i = 0; i < 9; ++i) dp.K[i] = a.K[i];
  pose_update_one(b, a.poses, delta[0], delta[1], a.rot_dim, a.trans_tanh, a.tn0, a.tn1, a.tn2, a.rot_normalizer, a.trans_scale, dp, a.poses);
  if (a.next_window) crop_window_tf_one(b, a.poses, a.win, a.tf, a.bbox);
  if (a.centered) pose_of_mesh_one(a.poses + (size_t)b * 16, a.cneg, a.centered + (size_t)b * 16);
}

int launch_refine_tail(const RefineTailArgs &a, int N, hipStream_t s) {
  FP_REQUIRE(a.rot_dim == 3 || a.rot_dim == 6, "refine tail: rot_dim must be 3 or 6");
  FP_REQUIRE(a.nparts >= 1 && a.nparts <= MH_MAXPARTS, "refine tail: %d partial rows per hypothesis (at most %d)", a.nparts, MH_MAXPARTS);
  if (N == 0) return FP_OK;
  hipLaunchKernelGGL(refine_tail_kernel, dim3(N), dim3(512), 0, s, a);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

int launch_mean_head(const float *partial, int nparts, const float *g, const float *b, int Bn, int T, const float *hw, const float *hb,
                     int out_dim, float *out, hipStream_t s) {
  FP_REQUIRE(nparts >= 1 && nparts <= MH_MAXPARTS, "mean_head: %d partial rows per hypothesis (at most %d)", nparts, MH_MAXPARTS);
  if (Bn == 0) return FP_OK;
  hipLaunchKernelGGL(mean_head_kernel, dim3(Bn), dim3(512), 0, s, partial, nparts, g, b, T, hw, hb, out_dim, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// pose update (predict_pose_refine.py:195-231): float32, mirrors oracle/predict.py:pose_update (pose_math.h)
__global__ void pose_update_kernel(const float *__restrict__ poseA, const float *__restrict__ trans, const float *__restrict__ rot,
                                   int N, int rot_dim, int trans_tanh, float tn0, float tn1, float tn2, float rot_normalizer,
                                   float trans_scale, DeepimArgs dp, float *__restrict__ outp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  pose_update_one(b, poseA, trans + (size_t)b * 3, rot + (size_t)b * rot_dim, rot_dim, trans_tanh, tn0, tn1, tn2, rot_normalizer, trans_scale, dp, outp);
}

int launch_pose_update(const float *poseA, const float *trans, const float *rot, int N, int rot_dim, int trans_tanh, float tn0,
                       float tn1, float tn2, float rot_normalizer, float trans_scale, float *out, hipStream_t s, const float *tf,
                       const double *K, float resize) {
  FP_REQUIRE(rot_dim == 3 || rot_dim == 6, "pose_update: rot_dim must be 3 or 6");
  FP_REQUIRE(trans_tanh >= 0 && trans_tanh <= 2, "pose_update: trans mode %d unknown (0 raw, 1 tanh, 2 deepim)", trans_tanh);
  FP_REQUIRE(trans_tanh != 2 || (tf && K), "pose_update: trans_rep='deepim' needs the crop transforms and K");
  if (N == 0) return FP_OK;
  DeepimArgs dp;
  dp.tf = tf;
  dp.resize = resize;
  for (int i = 0; i < 9; ++i) dp.K[i] = K ? (float)K[i] : 0.f;
  hipLaunchKernelGGL(pose_update_kernel, dim3((N + 63) / 64), dim3(64), 0, s, poseA, trans, rot, N, rot_dim, trans_tanh, tn0, tn1, tn2,
                     rot_normalizer, trans_scale, dp, out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
