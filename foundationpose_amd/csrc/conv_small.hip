// 3x3 stride-1 convolution (+ folded BN + residual + ReLU, network_modules.py:73-111) for launches of a FEW images: a tracking
// frame (src/estimater.py:250-268: one hypothesis = one rendered and one observed crop per pass) and the 1 .. 2-hypothesis calls.
//
// At one hypothesis a layer is 2 GFLOP - 2 us of the matrix pipes - and what a launch costs is its latency chain, not its work:
// the large-batch kernel (conv_halo.hip: 128 .. 512 pixels x 128 couts per workgroup, 3 k-cycle prologue, 17 k-cycle epilogue)
// fills 13 .. 26 CUs, and its split-K form pays a second launch that adds the shares (11 + 5 us per layer, 5 us of gap between
// them in a frame's dependent chain).  This kernel is built the other way round:
//   * a workgroup owns 32 pixels x 32 couts - ONE accumulator tile - so that a layer is 200 .. 400 workgroups (40x40 x 128 / 256
//     channels) or 208 (20x20 x 512): every CU of the chip works on every layer;
//   * its 8 waves split the INPUT CHANNELS (wave w: channels [w Cin/8, (w+1) Cin/8) of all nine taps) and add their partial
//     tiles through LDS in wave order - split-K inside the workgroup, no second launch, no fp32 round trip through HBM;
//   * weights come from a copy packed in this kernel's fragment order (small_pack_weights: per cout tile, wave, tap and 16-channel
//     step one 1-KB block, lane = (cout, k-block)), so every weight load is one fully coalesced 1-KB read, L2 -> registers; all of a
//     wave's weight loads are issued before anything else;
//   * the pixels a tile's taps touch are ONE contiguous range of the NHWC input (32 + 2 W + 2 pixels): copied to LDS with
//     coalesced 16-byte loads (row pitch Cin + 8 halfs: conflict-free fragment reads), one barrier, then fragments by ds_read_b128;
//     taps outside the image and the pixels behind the last one read a zero row (an address select, no branch);
//     (first form, measured: both operands as per-lane 16-byte gathers from L2 - 32 .. 64 cache lines per load instruction - ran
//     at 16 B/clk per CU: 18 us for the 256- and 512-channel layers against 15 for split-K + finishing pass)
//   * workgroup id -> (pixel tile, cout tile) such that the workgroups of one XCD share cout tiles: an XCD's L2 holds 1/8 .. 1/4
//     of the weights and the (small) input.
// fp32 accumulation; one summation order of its own (per wave: taps in order, 16-channel steps in order; then the waves in
// order), so results differ in the last fp32 bits from the large-batch kernels' - like the split-K form it replaces.
#include "common.h"

namespace {

constexpr int SM_NW = 8;       // waves per workgroup = shares of the input channels (4 for the 64-channel layer: 16 channels per wave)
constexpr int SM_LD = 36;      // floats per pixel row of a partial tile in LDS (32 couts + 4: 16-byte rows, spread over the banks)

template <int KS, int HW, int ST, int NW = SM_NW>
struct SmallCfg {
  static constexpr int CIN = NW * 16 * KS;
  static constexpr int PITCH = CIN + 8;                 // halfs per staged pixel
  // input pixels (flat NHWC index) from the first tap of output pixel m0 to the last tap of m0 + 31.  Stride 1: m0 - HW - 1 .. m0 + 31 + HW + 1.
  // Stride 2 (40x40 in, 20x20 out): 31 output pixels on are one output row and 11 columns, or two rows on and 9 columns back (a step to the
  // next row - or into the next image - is 2 HW input pixels): at most 4 HW - 18 input pixels on, + the taps' HW + 1 on either side
  // (80x80 in, 40x40 out: one output row and 9 columns back at most: 2 HW - 18 input pixels on)
  static constexpr int HO = HW / ST;
  static constexpr int DMAX = ST == 1 ? 31 : (31 >= HO ? 4 * HW + 2 * (31 - 2 * HO) : 2 * HW + 2 * (31 - HO));
  static constexpr int SPAN = DMAX + 1 + 2 * HW + 2;
  static_assert(ST == 1 || (ST == 2 && HW % 2 == 0 && 2 * HO >= 31), "stride 2: at most two output rows per tile");
  static constexpr int BAND_BYTES = (SPAN + 1) * PITCH * 2;      // + the zero row
  static constexpr int PART_BYTES = NW * 32 * SM_LD * 4;
  static constexpr int LDS_BYTES = BAND_BYTES > PART_BYTES ? BAND_BYTES : PART_BYTES;
};

// KS 16-channel steps per tap and wave: Cin = 128 KS; HW x HW input maps, stride ST (1 or 2: HW/2 x HW/2 output maps)
template <int KS, int HW, int ST, int NW = SM_NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_small_kernel(ConvArgs p, const f16 *__restrict__ wsm) {
  using C = SmallCfg<KS, HW, ST, NW>;
  constexpr int HO = HW / ST;
  extern __shared__ __attribute__((aligned(16))) f16 band[];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const int nct = p.Cout >> 5;
  const int ct = blockIdx.x % nct, pt = blockIdx.x / nct;
  const int co0 = ct * 32, m0 = pt * 32;
  // ---- the pixels the tile's taps touch: ONE contiguous range of input pixels from pb on -> LDS rows 0 .. SPAN - 1; row SPAN = zeros
  half8 af[9][KS];
  auto in_index = [&](int mm) -> int {                  // flat input pixel under the centre tap of output pixel mm
    if constexpr (ST == 1) return mm;
    const int im = mm / (HO * HO), r = mm - im * (HO * HO), yy = r / HO, xx = r - yy * HO;
    return im * (HW * HW) + (yy * ST) * HW + xx * ST;
  };
  const int pb = in_index(m0) - HW - 1, n_in = p.Nimg * HW * HW;
  {
    constexpr int PPR = C::CIN / 8;                     // 16-byte pieces per pixel
    constexpr int NP = C::SPAN * PPR, IT = (NP + NW * 64 - 1) / (NW * 64);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * (NW * 64), pl = i / PPR, c8 = i - pl * PPR, pix = pb + pl;
      // (rows of the range in front of the first image or behind the last hold whatever pixel 0 holds: no tap reads them - a tap outside its
      // image reads the zero row -, and a select on the loaded value would make the wave wait for the load before it requests the next)
      const bool ok = i < NP && pix >= 0 && pix < n_in;
      v[u] = *reinterpret_cast<const u32x4 *>(p.in + (size_t)(ok ? pix : 0) * C::CIN + c8 * 8);
    }
    // ---- every weight fragment of this wave: 9 KS coalesced 1-KB loads, requested BEHIND the band's loads (which the first MFMA needs
    // first) and in flight across the band's way through LDS and the barrier; the K loop consumes them in the order they arrive
    {
      const f16 *wb = wsm + ((size_t)(ct * NW + w) * 9 * KS) * 512 + lane * 8;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < KS; ++j) af[t][j] = *reinterpret_cast<const half8 *>(wb + (t * KS + j) * 512);
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * (NW * 64), pl = i / PPR, c8 = i - pl * PPR;
      if (i < NP) *reinterpret_cast<u32x4 *>(&band[pl * C::PITCH + c8 * 8]) = v[u];
    }
    if (tid < PPR) *reinterpret_cast<u32x4 *>(&band[C::SPAN * C::PITCH + tid * 8]) = u32x4{0u, 0u, 0u, 0u};
  }
  // the band is in LDS for every wave (a __syncthreads() would also wait for the weight loads: vmcnt(0))
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // ---- K loop: this lane's pixel (operand B) m0 + lr; its k-block of step j: channels cofs + 8 j .. + 8
  const int m = m0 + lr;
  const bool mv = m < p.M;
  const int rem = m % (HO * HO), yo = rem / HO, y = yo * ST, x = (rem - yo * HO) * ST;      // centre tap in the input map
  const int l0 = in_index(mv ? m : m0) - pb - HW - 1;   // LDS row of this lane's tap (0, 0)
  const int cofs = w * (16 * KS) + lh * (8 * KS);
  floatx16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = y + ky - 1;
    const bool yok = mv && yy >= 0 && yy < HW;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = x + kx - 1;
      const int row = (yok && xx >= 0 && xx < HW) ? l0 + ky * HW + kx : C::SPAN;
      const f16 *src = band + row * C::PITCH + cofs;
#pragma unroll
      for (int j = 0; j < KS; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ky * 3 + kx][j], *reinterpret_cast<const half8 *>(src + j * 8), acc, 0, 0, 0);
    }
  }
  __syncthreads();            // the band is read: its LDS now takes the partial tiles
  // partial tile of this wave -> LDS [pixel][cout]: acc[rg*4 + e] = (cout rg*8 + lh*4 + e, pixel lr)
  float (*part)[32][SM_LD] = reinterpret_cast<float (*)[32][SM_LD]>(band);
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
    *reinterpret_cast<float4 *>(&part[w][lr][rg * 8 + lh * 4]) = make_float4(acc[rg * 4 + 0], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3]);
  __syncthreads();
  if (NW * 64 > 256 && tid >= 256) return;
  // 256 threads: pixel tid / 8, four couts each; the waves' shares in wave order, then bias, residual, ReLU, positional embedding
  const int px = tid >> 3, c4 = (tid & 7) * 4;
  const int mo = m0 + px;
  if (mo >= p.M) return;
  float4 v = *reinterpret_cast<const float4 *>(&part[0][px][c4]);
#pragma unroll
  for (int s = 1; s < NW; ++s) {
    const float4 q = *reinterpret_cast<const float4 *>(&part[s][px][c4]);
    v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
  }
  const int c = co0 + c4;
  const float4 bv = *reinterpret_cast<const float4 *>(p.bias + c);
  v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
  if (p.res) {
    const half4 r = *reinterpret_cast<const half4 *>(p.res + (size_t)mo * p.Cout + c);
    v.x += (float)r[0], v.y += (float)r[1], v.z += (float)r[2], v.w += (float)r[3];
  }
  const float lo = p.relu ? 0.f : -__builtin_inff();
  v.x = fmaxf(v.x, lo), v.y = fmaxf(v.y, lo), v.z = fmaxf(v.z, lo), v.w = fmaxf(v.w, lo);
  if (p.post_add) {
    const float4 pv = *reinterpret_cast<const float4 *>(p.post_add + (size_t)(mo % p.post_period) * p.Cout + c);
    v.x += pv.x, v.y += pv.y, v.z += pv.z, v.w += pv.w;
  }
  half4 hv;
  hv[0] = (f16)v.x, hv[1] = (f16)v.y, hv[2] = (f16)v.z, hv[3] = (f16)v.w;
  const bool hi = mo >= p.split_m;
  const long long orow = hi ? (long long)(mo - p.split_m) : (long long)mo;
  *reinterpret_cast<half4 *>((f16 *)p.out + orow * p.out_ld + (hi ? p.coff_hi : 0) + c) = hv;
}

// [Cout][9][Cin] -> per (cout tile of 32, wave, tap, 16-channel step) one block of 64 lanes x 8 halfs: lane (lh, lr) holds
// w[cout tile * 32 + lr][tap][wave * Cin/nw + lh * Cin/(2 nw) + 8 step .. + 8]  (nw waves: 8, or 4 for 64 input channels)
__global__ __launch_bounds__(256) void small_pack_kernel(const f16 *__restrict__ w, int Cout, int Cin, int Kpad, int nw, f16 *__restrict__ out) {
  const int KS = Cin / (16 * nw);
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)Cout * 9 * Cin / 8;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  long long blk = idx >> 6;
  const int j = (int)(blk % KS);
  blk /= KS;
  const int tap = (int)(blk % 9);
  blk /= 9;
  const int wv = (int)(blk % nw), ct = (int)(blk / nw);
  const int lr = lane & 31, lh = lane >> 5;
  const f16 *src = w + (size_t)(ct * 32 + lr) * Kpad + tap * Cin + wv * (16 * KS) + lh * (8 * KS) + j * 8;
  *reinterpret_cast<half8 *>(out + idx * 8) = *reinterpret_cast<const half8 *>(src);
}

}  // namespace

void conv_small_kernel_lds(std::vector<KernelLds> &v) {
  v.push_back({(const void *)conv3x3_small_kernel<1, 40, 1>, SmallCfg<1, 40, 1>::LDS_BYTES});
  v.push_back({(const void *)conv3x3_small_kernel<2, 40, 1>, SmallCfg<2, 40, 1>::LDS_BYTES});
  v.push_back({(const void *)conv3x3_small_kernel<4, 20, 1>, SmallCfg<4, 20, 1>::LDS_BYTES});
  v.push_back({(const void *)conv3x3_small_kernel<2, 40, 2>, SmallCfg<2, 40, 2>::LDS_BYTES});
  v.push_back({(const void *)conv3x3_small_kernel<1, 80, 2, 4>, SmallCfg<1, 80, 2, 4>::LDS_BYTES});
}

size_t small_packed_halfs(int Cout, int Cin) { return (size_t)Cout * 9 * Cin; }

int small_pack_weights(const f16 *d_w, int Cout, int Cin, int Kpad, f16 *d_out, hipStream_t s) {
  FP_REQUIRE((Cin == 64 || Cin == 128 || Cin == 256 || Cin == 512) && Cout % 32 == 0 && Kpad >= 9 * Cin, "small_pack_weights: Cout=%d Cin=%d unsupported", Cout, Cin);
  const long long total = (long long)Cout * 9 * Cin / 8;
  hipLaunchKernelGGL(small_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_w, Cout, Cin, Kpad, Cin == 64 ? 4 : SM_NW, d_out);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}

// The layers this form runs: the 3x3 stride-1 layers of the trunks (128 / 256 channels on 40x40 maps, 512 on 20x20) and the two
// stride-2 layers (64 -> 128 on 80x80, 256 -> 512 on 40x40) in the network passes
// of one or two hypotheses, and stand-alone launches of at most FP_SMALL_MAX_WG workgroups (default 2 per CU), when the caller holds the
// packed weights (ConvArgs::wsm).  FP_SMALL=0: off (A/B timing).
bool conv_small_shape(const ConvArgs &a, int num_cu) {
  static const int on = getenv("FP_SMALL") ? atoi(getenv("FP_SMALL")) : 1;
  static const int max_wg = getenv("FP_SMALL_MAX_WG") ? atoi(getenv("FP_SMALL_MAX_WG")) : 0;
  if (!on) return false;
  const bool s1 = a.stride == 1 && a.Ho == a.H && a.Wo == a.W && ((a.W == 40 && (a.Cin == 128 || a.Cin == 256)) || (a.W == 20 && a.Cin == 512));
  const bool s2 = a.stride == 2 && ((a.W == 40 && a.Ho == 20 && a.Wo == 20 && a.Cin == 256) ||   // the 256 -> 512 stride-2 layer between the two halves of encodeAB
                                    (a.W == 80 && a.Ho == 40 && a.Wo == 40 && a.Cin == 64));     // the 64 -> 128 stride-2 layer behind the stem
  if (!(a.KH == 3 && a.KW == 3 && a.pad == 1 && a.H == a.W && a.out_mode == 0 && (s1 || s2) && a.Cout % 32 == 0 && a.Kpad == 9 * a.Cin &&
        a.out_ld % 4 == 0 && a.coff_hi % 4 == 0 && a.M > 0))
    return false;
  if (a.hyp > 0 && max_wg == 0) return a.hyp <= 2;           // inside a network pass: by its hypotheses, whatever the launch's share of them
  const long long wgs = (long long)((a.M + 31) / 32) * (a.Cout / 32);
  return wgs <= (max_wg > 0 ? max_wg : 2 * num_cu);
}
bool conv_small_use(const ConvArgs &a, int num_cu) { return a.wsm != nullptr && conv_small_shape(a, num_cu); }

int launch_conv_small(fp_ctx *ctx, const ConvArgs &a, hipStream_t s) {
  FP_REQUIRE(conv_small_use(a, ctx->num_cu), "conv3x3 small: unsupported layer");
  FP_REQUIRE((double)a.M * a.Cin * 2.0 < 2147483648.0, "conv3x3 small: input tensor too large");
  const int grid = ((a.M + 31) / 32) * (a.Cout / 32);
  constexpr int l1 = SmallCfg<1, 40, 1>::LDS_BYTES, l2 = SmallCfg<2, 40, 1>::LDS_BYTES, l4 = SmallCfg<4, 20, 1>::LDS_BYTES, l2s = SmallCfg<2, 40, 2>::LDS_BYTES;
  static_assert(l2s <= 160 * 1024 - 1024, "stride-2 band fits LDS");
  const dim3 g(grid), b(SM_NW * 64);
  constexpr int l1s = SmallCfg<1, 80, 2, 4>::LDS_BYTES;
  if (a.stride == 2 && a.Cin == 64) hipLaunchKernelGGL((conv3x3_small_kernel<1, 80, 2, 4>), g, dim3(4 * 64), l1s, s, a, a.wsm);
  else if (a.stride == 2) hipLaunchKernelGGL((conv3x3_small_kernel<2, 40, 2>), g, b, l2s, s, a, a.wsm);
  else if (a.Cin == 128) hipLaunchKernelGGL((conv3x3_small_kernel<1, 40, 1>), g, b, l1, s, a, a.wsm);
  else if (a.Cin == 256) hipLaunchKernelGGL((conv3x3_small_kernel<2, 40, 1>), g, b, l2, s, a, a.wsm);
  else hipLaunchKernelGGL((conv3x3_small_kernel<4, 20, 1>), g, b, l4, s, a, a.wsm);
  FP_CHECK_HIP(hipGetLastError());
  return FP_OK;
}
