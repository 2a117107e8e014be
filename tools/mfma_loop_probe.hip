// What slows an LDS-fed fp32 MFMA loop?  Variants of the 64x64-tile inner loop of gemm_f32.hip with pieces removed:
//   0: 32 MFMAs per phase on register operands (no LDS)          1: + fragment reads from LDS (8 ds_read_b128 per phase)
//   2: + one __syncthreads per phase                              3: + 4 ds_write_b128 per phase (no global loads)
//   4: + 4 global_load_dwordx4 per phase feeding the writes (L2-resident 2 MB buffer)
// 512 workgroups of 256 threads (2 per CU), 2000 phases each.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V>
__global__ __launch_bounds__(256, 2) void loop(const float* __restrict__ in, float* out, int phases) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][64 * 36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  for (int i = t; i < 2 * 2 * 64 * 36; i += 256) (&lds[0][0][0])[i] = in[i & 4095];
  __syncthreads();
  f32x4 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[2], rb[2];
  ra[0] = ra[1] = rb[0] = rb[1] = make_float4(in[t], in[t + 1], in[t + 2], in[t + 3]);
  float a0[2][4], b0[2][4];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) { a0[i][e] = in[t + i + e]; b0[i][e] = in[t + 9 + i + e]; }
  const float* gp = in + (blockIdx.x % 64) * 8192 + t * 4;
  for (int ph = 0; ph < phases; ++ph) {
    const int buf = ph & 1;
    if (V >= 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(gp + ((ph * 2 + i) & 7) * 1024);
        rb[i] = *reinterpret_cast<const float4*>(gp + ((ph * 2 + i + 4) & 7) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float af[2][4], bf[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (V >= 1) {
          const float4 v = *reinterpret_cast<const float4*>(&lds[buf][0][(wm * 32 + i * 16 + fr) * 36 + kk * 16 + fq * 4]);
          const float4 w = *reinterpret_cast<const float4*>(&lds[buf][1][(wn * 32 + i * 16 + fr) * 36 + kk * 16 + fq * 4]);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
          bf[i][0] = w.x; bf[i][1] = w.y; bf[i][2] = w.z; bf[i][3] = w.w;
        } else {
          for (int e = 0; e < 4; ++e) { af[i][e] = a0[i][e]; bf[i][e] = b0[i][e]; }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
    if (V >= 3) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<float4*>(&lds[buf ^ 1][0][(t / 8 + 32 * i) * 36 + (t % 8) * 4]) = ra[i];
        *reinterpret_cast<float4*>(&lds[buf ^ 1][1][(t / 8 + 32 * i) * 36 + (t % 8) * 4]) = rb[i];
      }
    }
    if (V >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + t] = s;
}

template <int V>
void run(const float* in, float* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int phases = 2000;
  float best = 1e9, ms;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop<V>, dim3(blocks), dim3(256), 0, 0, in, out, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * phases * 32 * 2048.0;
  printf("variant %d, %4d workgroups: %.3f ms  %.1f TFLOP/s\n", V, blocks, best, flop / (best * 1e-3) / 1e12);
}

// Software-pipelined forms of variant 4:
//   5: LDS stores in the middle of the phase (after the first 16 MFMAs), barrier at the end
//   6: 5 + fragments double-buffered in registers: barrier | read next tile's first half | last 16 MFMAs (pinned by asm)
//   7: 6 with the odd workgroups delayed by half a phase at start (stagger)
template <int V>
__global__ __launch_bounds__(256, 2) void loop_p(const float* __restrict__ in, float* out, int phases) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][64 * 36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  for (int i = t; i < 2 * 2 * 64 * 36; i += 256) (&lds[0][0][0])[i] = in[i & 4095];
  __syncthreads();
  f32x4 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[2][2], rb[2][2];
  for (int q = 0; q < 2; ++q) for (int i = 0; i < 2; ++i) ra[q][i] = rb[q][i] = make_float4(in[t], in[t + 1], in[t + 2], in[t + 3]);
  const float* gp = in + (blockIdx.x % 64) * 8192 + t * 4;
  float fa[2][2][4], fb[2][2][4];
  auto rd = [&](int buf, int kk, float (&af)[2][4], float (&bf)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(&lds[buf][0][(wm * 32 + i * 16 + fr) * 36 + kk * 16 + fq * 4]);
      const float4 w = *reinterpret_cast<const float4*>(&lds[buf][1][(wn * 32 + i * 16 + fr) * 36 + kk * 16 + fq * 4]);
      af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
      bf[i][0] = w.x; bf[i][1] = w.y; bf[i][2] = w.z; bf[i][3] = w.w;
    }
  };
  auto mm = [&](const float (&af)[2][4], const float (&bf)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
  };
  if (V >= 7 && (blockIdx.x & 256)) __builtin_amdgcn_s_sleep(16);
  rd(0, 0, fa[0], fb[0]);
  auto body = [&](auto bufc, int ph) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[buf][i] = *reinterpret_cast<const float4*>(gp + ((ph * 2 + i) & 7) * 1024);
      rb[buf][i] = *reinterpret_cast<const float4*>(gp + ((ph * 2 + i + 4) & 7) * 1024);
    }
    if (V == 5) rd(buf, 0, fa[0], fb[0]);
    rd(buf, 1, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mm(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {     // the tile loaded one phase ago
      *reinterpret_cast<float4*>(&lds[buf ^ 1][0][(t / 8 + 32 * i) * 36 + (t % 8) * 4]) = ra[buf ^ 1][i];
      *reinterpret_cast<float4*>(&lds[buf ^ 1][1][(t / 8 + 32 * i) * 36 + (t % 8) * 4]) = rb[buf ^ 1][i];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (V == 5) {
      mm(fa[1], fb[1]);
      __syncthreads();
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1][i][e], fb[1][j][e], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      rd(buf ^ 1, 0, fa[0], fb[0]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 2; e < 4; ++e) { asm volatile("" : "+v"(fa[1][i][e])); asm volatile("" : "+v"(fb[1][i][e])); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 2; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1][i][e], fb[1][j][e], acc[i][j], 0, 0, 0);
    }
  };
  for (int ph = 0; ph < phases; ph += 2) {
    body(std::integral_constant<int, 0>{}, ph);
    body(std::integral_constant<int, 1>{}, ph + 1);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + t] = s;
}
template <int V>
void run_p(const float* in, float* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int phases = 2000;
  float best = 1e9, ms;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop_p<V>, dim3(blocks), dim3(256), 0, 0, in, out, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * phases * 32 * 2048.0;
  printf("variant %d, %4d workgroups: %.3f ms  %.1f TFLOP/s\n", V, blocks, best, flop / (best * 1e-3) / 1e12);
}

//   8: LDS-DMA staging (global_load_lds_dwordx4 into the other LDS buffer at the top of the phase, no VGPR hop, no
//      ds_write), unpadded [row][32] image (bank conflicts not swizzled away here), __syncthreads at the end
template <int V>
__global__ __launch_bounds__(256, 2) void loop_d(const float* __restrict__ in, float* out, int phases) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][64 * 32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  for (int i = t; i < 2 * 2 * 64 * 32; i += 256) (&lds[0][0][0])[i] = in[i & 4095];
  __syncthreads();
  f32x4 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* gp = in + (blockIdx.x % 64) * 8192 + t * 4;
  auto body = [&](auto bufc, int ph) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    // each wave fills 2 x 1 KiB of A and 2 x 1 KiB of B of the OTHER buffer: wave w covers rows 16w..16w+15
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds(gp + ((ph * 2 + i) & 7) * 1024, (__attribute__((address_space(3))) void*)&lds[buf ^ 1][0][(wave * 16 + i * 8) * 32], 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gp + ((ph * 2 + i + 4) & 7) * 1024, (__attribute__((address_space(3))) void*)&lds[buf ^ 1][1][(wave * 16 + i * 8) * 32], 16, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      float af[2][4], bf[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra_ = wm * 32 + i * 16 + fr, rb_ = wn * 32 + i * 16 + fr;
        const int ca = ((kk * 4 + fq) ^ ((ra_ >> 1) & 7)) * 4, cb = ((kk * 4 + fq) ^ ((rb_ >> 1) & 7)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(&lds[buf][0][ra_ * 32 + ca]);
        const float4 w = *reinterpret_cast<const float4*>(&lds[buf][1][rb_ * 32 + cb]);
        af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        bf[i][0] = w.x; bf[i][1] = w.y; bf[i][2] = w.z; bf[i][3] = w.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int ph = 0; ph < phases; ph += 2) {
    body(std::integral_constant<int, 0>{}, ph);
    body(std::integral_constant<int, 1>{}, ph + 1);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + t] = s;
}
template <int V>
void run_d(const float* in, float* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int phases = 2000;
  float best = 1e9, ms;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop_d<V>, dim3(blocks), dim3(256), 0, 0, in, out, phases);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * phases * 32 * 2048.0;
  printf("variant %d, %4d workgroups: %.3f ms  %.1f TFLOP/s\n", V, blocks, best, flop / (best * 1e-3) / 1e12);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 64 * 8192 * 4 + 65536); hipMalloc(&out, 2048 * 256 * 4);
  std::vector<float> h(64 * 8192 + 16384);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int blocks : {512, 768}) {
    run<0>(in, out, blocks); run<2>(in, out, blocks); run<3>(in, out, blocks); run<4>(in, out, blocks);
    run_p<5>(in, out, blocks); run_p<6>(in, out, blocks); run_p<7>(in, out, blocks); run_d<8>(in, out, blocks);
  }
  return 0;
}
