// Shared device helpers for the espnet_amd HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EAMD_OK 0
#define EAMD_EINVAL (-1)
#define EAMD_EUNSUPPORTED (-2)

#define EAMD_WAVE 64

#define EAMD_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

// activation ids shared by host and device
enum { EAMD_ACT_NONE = 0, EAMD_ACT_RELU = 1, EAMD_ACT_SWISH = 2, EAMD_ACT_TANH = 3 };

__device__ __forceinline__ float eamd_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float eamd_swish(float x) { return x * eamd_sigmoid(x); }
// d/dx [x*sigmoid(x)] = s + x*s*(1-s)
__device__ __forceinline__ float eamd_dswish(float x) {
  float s = eamd_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float eamd_act(float x, int act) {
  if (act == EAMD_ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == EAMD_ACT_SWISH) return eamd_swish(x);
  if (act == EAMD_ACT_TANH) return tanhf(x);
  return x;
}
// derivative of eamd_act at pre-activation x
__device__ __forceinline__ float eamd_dact(float x, int act) {
  if (act == EAMD_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == EAMD_ACT_SWISH) return eamd_dswish(x);
  if (act == EAMD_ACT_TANH) { float t = tanhf(x); return 1.f - t * t; }
  return 1.f;
}

// Counter-based dropout bits shared by eamd_dropout and the fused GEMM epilogues: a per-launch 32-bit seed from
// (device step counter, site salt) and a 32-bit avalanche (lowbias32) of the element index.  The 64-bit
// mix runs once per thread, the per-element cost is ~10 VALU instructions (a 64-bit splitmix per element made the
// FFN up-projection epilogue ALU-bound).
__device__ __forceinline__ unsigned eamd_drop_seed(const unsigned long long* step, unsigned long long salt) {
  unsigned long long x = (step ? step[0] : 0ULL) * 0x9E3779B97F4A7C15ULL + salt * 0xD1B54A32D192ED03ULL;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x ^ (unsigned)(x >> 32);
}
__device__ __forceinline__ unsigned eamd_drop_bits(unsigned seed, unsigned long long idx) {
  unsigned h = ((unsigned)idx ^ seed) + (unsigned)(idx >> 32) * 0x9E3779B1u;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

// fp32 -> bf16 round-to-nearest-even (plain cast keeps NaN a NaN on gfx950).
__device__ __forceinline__ unsigned short eamd_f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// Wave-wide reductions (result in every lane).  The four steps inside a 16-lane row are DPP modifiers on the
// VALU operand (quad_perm, row_half_mirror, row_mirror: no LDS crossbar trip), only the two cross-row steps go
// through ds_bpermute; the plain six-step __shfl_xor butterfly spends ~6 dependent LDS round trips per reduction.
template <int CTRL>
__device__ __forceinline__ float eamd_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += eamd_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += eamd_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += eamd_dpp<0x141>(v);     // row_half_mirror
  v += eamd_dpp<0x140>(v);     // row_mirror  -> every lane holds its 16-lane row sum
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, eamd_dpp<0xB1>(v));
  v = fmaxf(v, eamd_dpp<0x4E>(v));
  v = fmaxf(v, eamd_dpp<0x141>(v));
  v = fmaxf(v, eamd_dpp<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// Block-wide reductions for blockDim.x <= 1024 (multiple of 64). `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}
