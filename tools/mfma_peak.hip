// Sustained MFMA rate of this chip under load: register-only loops of v_mfma_f32_16x16x4_f32 (fp32) and
// v_mfma_f32_16x16x32_bf16 on random data, one or two waves per SIMD, ~0.2 s each.  Build: hipcc --offload-arch=gfx950 -O3
// tools/mfma_peak.hip -o tools/mfma_peak.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ __launch_bounds__(256) void f32_loop(const float* in, float* out, long iters, unsigned long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (long it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void bf16_loop(const float* in, float* out, long iters, unsigned long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[threadIdx.x + j]; b[j] = (__bf16)in[threadIdx.x + 300 + j]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (long it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  float *in, *out;
  unsigned long long* clk;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    for (int kind = 0; kind < 2; ++kind) {
      const int blocks = 256 * wps;
      const long iters = kind == 0 ? 40000 : 80000;
      float best = 0, ms = 0;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(f32_loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        else hipLaunchKernelGGL(bf16_loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 16 * (kind == 0 ? 2048.0 : 16384.0);
        float tf = flop / (ms * 1e-3) / 1e12;
        if (rep >= 2 && tf > best) best = tf;
      }
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      printf("%s  %d wave(s)/SIMD: %.1f TFLOP/s (last launch %.1f ms), in-kernel clock %.2f GHz\n", kind == 0 ? "fp32 16x16x4 " : "bf16 16x16x32",
             wps, best, ms, (double)c[0] / (double)c[1] * 0.1);
    }
  }
  return 0;
}
