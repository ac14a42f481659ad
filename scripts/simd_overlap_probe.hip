// Probe (not part of the library): do the two waves of a SIMD overlap one wave's MFMA phase with the other's VALU phase?
// One workgroup per CU, 8 waves (two per SIMD).  Every iteration has two phases separated by s_barrier:
//   mode 0 (aligned)      : phase 1 = all waves MFMA (16 x 32x32x16 f16, 4 accumulators), phase 2 = all waves VALU (33 v_exp + ~150 others)
//   mode 1 (complementary): phase 1 = waves 0-3 MFMA / waves 4-7 VALU, phase 2 = the other way round
//   mode 2: MFMA phases only, mode 3: VALU phases only (what each costs alone for the pair)
//   mode 4: both phases = every wave runs MFMA and VALU interleaved 1 : 1/16 (the work of modes 0 / 1 per iteration x 2); mode 5: the same
//   with only one wave of each SIMD active per phase (the work of modes 0 / 1)
// build: hipcc -O3 --offload-arch=gfx950 scripts/simd_overlap_probe.hip -o gpurun_out/simd_probe ; run: gpurun_out/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_phase(floatx16 (&acc)[4], const half8 &a, const half8 &b) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
}
__device__ __forceinline__ void valu_phase(float (&v)[32], float c) {
  // a softmax-like body: max chain, 32 exps, packed fma / adds, converts
  float m = v[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) m = fmaxf(m, v[i]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    v[i] = __builtin_amdgcn_exp2f(v[i] * c - m * c);
    s += v[i];
  }
#pragma unroll
  for (int i = 0; i < 32; i += 2) {
    const _Float16 h0 = (_Float16)v[i], h1 = (_Float16)v[i + 1];
    v[i] = (float)h0 + s * 1e-9f;
    v[i + 1] = (float)h1 - s * 1e-9f;
  }
}

// 16 x [1 MFMA + 10 vector instructions (2 of them v_exp_f32), pinned in place as volatile asm]: what a wave gets when its vector work
// sits in the shadows of its own MFMAs.  valu_asm_phase: the same 160 vector instructions without the MFMAs.
__device__ __forceinline__ void valu_piece(float (&v)[32], int g, float c, float &s) {
  float x0 = v[2 * g], x1 = v[2 * g + 1], t0, t1, t2, t3;
  asm volatile("v_fma_f32 %0, %6, %8, %8\n\tv_exp_f32 %0, %0\n\tv_fma_f32 %1, %7, %8, %8\n\tv_exp_f32 %1, %1\n\t"
               "v_add_f32 %2, %0, %1\n\tv_add_f32 %3, %2, %0\n\tv_add_f32 %4, %3, %1\n\tv_add_f32 %5, %4, %2\n\t"
               "v_mul_f32 %4, %5, %8\n\tv_add_f32 %5, %5, %4"
               : "=&v"(x0), "=&v"(x1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
               : "v"(v[2 * g]), "v"(v[2 * g + 1]), "v"(c));
  v[2 * g] = x0;
  v[2 * g + 1] = x1;
  s += t3;
}
__device__ __forceinline__ void mixed_phase(floatx16 (&acc)[4], const half8 &a, const half8 &b, float (&v)[32], float c) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[g & 3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    valu_piece(v, g, c, s);
    __builtin_amdgcn_sched_barrier(0);
  }
  v[0] += s * 1e-9f;
}
__device__ __forceinline__ void valu_asm_phase(float (&v)[32], float c) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    valu_piece(v, g, c, s);
    __builtin_amdgcn_sched_barrier(0);
  }
  v[0] += s * 1e-9f;
}

__global__ __launch_bounds__(512, 1) void probe(int mode, int iters, float *out, unsigned long long *cyc) {
  const int wave = threadIdx.x >> 6;
  floatx16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  half8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.01f * (threadIdx.x % 7 + e)), b[e] = (_Float16)(0.02f * (threadIdx.x % 5 + e));
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 0.001f * (threadIdx.x + i);
  const bool late = wave >= 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    if (mode == 4 || (mode == 5 && !late)) mixed_phase(acc, a, b, v, 0.7f);
    if ((mode == 6 && !late) || (mode == 8 && late) || mode == 9) valu_asm_phase(v, 0.7f);
    if ((mode == 7 && !late) || (mode == 8 && !late)) mfma_phase(acc, a, b);
    if (mode >= 10) {        // complementary with the dense (compiler) vector body; 10: the MFMA wave at priority 1, 11: the vector wave at priority 1
      if (!late) {
        if (mode == 10) __builtin_amdgcn_s_setprio(1);
        mfma_phase(acc, a, b);
        __builtin_amdgcn_s_setprio(0);
      } else {
        if (mode == 11) __builtin_amdgcn_s_setprio(1);
        valu_phase(v, 0.7f);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    if (mode == 0 || mode == 2 || (mode == 1 && !late)) mfma_phase(acc, a, b);
    if ((mode == 1 && late) || mode == 3) valu_phase(v, 0.7f);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (mode == 4 || (mode == 5 && late)) mixed_phase(acc, a, b, v, 0.7f);
    if ((mode == 6 && late) || (mode == 8 && !late) || mode == 9) valu_asm_phase(v, 0.7f);
    if ((mode == 7 && late) || (mode == 8 && late)) mfma_phase(acc, a, b);
    if (mode >= 10) {
      if (late) {
        if (mode == 10) __builtin_amdgcn_s_setprio(1);
        mfma_phase(acc, a, b);
        __builtin_amdgcn_s_setprio(0);
      } else {
        if (mode == 11) __builtin_amdgcn_s_setprio(1);
        valu_phase(v, 0.7f);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    if (mode == 0 || mode == 3 || (mode == 1 && !late)) valu_phase(v, 0.7f);
    if ((mode == 1 && late) || mode == 2) mfma_phase(acc, a, b);
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
#pragma unroll
  for (int i = 0; i < 32; ++i) r += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float *out;
  unsigned long long *cyc, h[256];
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  const char *names[12] = {"aligned (all MFMA | all VALU)", "complementary (half MFMA, half VALU | swapped)", "MFMA in both phases", "VALU in both phases", "each phase: every wave 16 x [1 MFMA + 1/16 of the VALU body]", "the same, one wave of each SIMD per phase (the other idle)", "160 pinned vector instr., one wave of each SIMD per phase", "16 MFMAs, one wave of each SIMD per phase", "complementary: 16 MFMAs beside the partner's 160 pinned vector instr.", "160 pinned vector instr., every wave, both phases", "complementary (dense vector body), the MFMA wave at s_setprio 1", "complementary (dense vector body), the vector wave at s_setprio 1"};
  for (int rep = 0; rep < 1; ++rep)
    for (int mode = 0; mode < 12; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double c = 0;
      for (int i = 0; i < 256; ++i) c += h[i];
      printf("mode %d  %-50s  %7.1f cycles per iteration (two phases)   %6.3f ms\n", mode, names[mode], c / 256 / iters, ms);
    }
  return 0;
}
