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

// Zero n bytes (n % 4 == 0, p 4-byte aligned) with a KERNEL.  Not hipMemsetAsync: captured into a hipGraph by torch
// (ROCm 7.2) the memset node filled the block with 16-byte garbage from the second replay on (tools/lstm_seq_graph_min.py).
template <typename W>
static __global__ __launch_bounds__(256) void eamd_zero_kernel(W* p, long n) {
  W z;
  __builtin_memset(&z, 0, sizeof(W));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = z;
}
static inline int eamd_zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return EAMD_OK;
  if ((bytes & 3) || ((uintptr_t)p & 3)) return EAMD_EINVAL;
  const bool wide = !(bytes & 15) && !((uintptr_t)p & 15);
  const long n = (long)(bytes / (wide ? 16 : 4));
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  if (wide) hipLaunchKernelGGL(eamd_zero_kernel<uint4>, dim3(blocks), dim3(256), 0, s, (uint4*)p, n);
  else hipLaunchKernelGGL(eamd_zero_kernel<unsigned>, dim3(blocks), dim3(256), 0, s, (unsigned*)p, n);
  return hipGetLastError() == hipSuccess ? EAMD_OK : EAMD_EINVAL;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

// activation ids shared by host and device
// (reference: nets_utils.py:485-498 get_activation - hardtanh, tanh, relu, selu, swish)
enum { EAMD_ACT_NONE = 0, EAMD_ACT_RELU = 1, EAMD_ACT_SWISH = 2, EAMD_ACT_TANH = 3, EAMD_ACT_HARDTANH = 4, EAMD_ACT_SELU = 5 };
#define EAMD_SELU_ALPHA 1.6732632423543772848170429916717f
#define EAMD_SELU_SCALE 1.0507009873554804934193349852946f

// v_exp_f32 + v_rcp_f32 (1 ulp each); an IEEE division here costs ten more VALU instructions per element, which the
// GEMM epilogues (Swish / its derivative on every FFN hidden unit) cannot hide
__device__ __forceinline__ float eamd_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
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
  if (act == EAMD_ACT_HARDTANH) return fminf(fmaxf(x, -1.f), 1.f);
  if (act == EAMD_ACT_SELU) return EAMD_SELU_SCALE * (x > 0.f ? x : EAMD_SELU_ALPHA * expm1f(x));
  return x;
}
// activation and its derivative from ONE sigmoid / tanh (the FFN forward's factor epilogue needs both per element)
__device__ __forceinline__ void eamd_act_dact(float x, int act, float& a, float& d) {
  if (act == EAMD_ACT_SWISH) {
    const float s = eamd_sigmoid(x);
    a = x * s;
    d = s * (1.0f + x * (1.0f - s));
  } else if (act == EAMD_ACT_RELU) {
    a = x > 0.f ? x : 0.f;
    d = x > 0.f ? 1.f : 0.f;
  } else if (act == EAMD_ACT_TANH) {
    a = tanhf(x);
    d = 1.f - a * a;
  } else if (act == EAMD_ACT_HARDTANH) {
    a = fminf(fmaxf(x, -1.f), 1.f);
    d = (x > -1.f && x < 1.f) ? 1.f : 0.f;
  } else if (act == EAMD_ACT_SELU) {
    const float ex = EAMD_SELU_ALPHA * expm1f(x);
    a = EAMD_SELU_SCALE * (x > 0.f ? x : ex);
    d = EAMD_SELU_SCALE * (x > 0.f ? 1.f : ex + EAMD_SELU_ALPHA);
  } else {
    a = x;
    d = 1.f;
  }
}
// derivative of eamd_act at pre-activation x
__device__ __forceinline__ float eamd_dact(float x, int act) {
  if (act == EAMD_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == EAMD_ACT_SWISH) return eamd_dswish(x);
  if (act == EAMD_ACT_TANH) { float t = tanhf(x); return 1.f - t * t; }
  if (act == EAMD_ACT_HARDTANH) return (x > -1.f && x < 1.f) ? 1.f : 0.f;
  if (act == EAMD_ACT_SELU) return EAMD_SELU_SCALE * (x > 0.f ? 1.f : EAMD_SELU_ALPHA * __expf(x));
  return 1.f;
}

// Counter-based dropout bits shared by eamd_dropout and the fused GEMM / LayerNorm epilogues: a per-launch 32-bit
// seed from (device step counter, site salt) and a 32-bit avalanche (lowbias32) of the element-PAIR index; element
// 2q takes the low 16 bits, element 2q+1 the high 16 bits, keep <=> bits >= round(p * 2^16).  One hash (two
// quarter-rate v_mul_lo_u32) serves two elements: a hash per element made the FFN epilogues VALU-bound.  The
// survivors are scaled by 2^16 / (2^16 - threshold), the exact inverse of the keep probability drawn.
__device__ __forceinline__ unsigned eamd_drop_seed(const unsigned long long* step, unsigned long long salt) {
  unsigned long long x = (step ? step[0] : 0ULL) * 0x9E3779B97F4A7C15ULL + salt * 0xD1B54A32D192ED03ULL;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x ^ (unsigned)(x >> 32);
}
__device__ __forceinline__ unsigned eamd_drop_thr16(float p) {
  return p >= 1.f ? 65536u : (unsigned)fminf(fmaxf(p, 0.f) * 65536.0f + 0.5f, 65535.0f);
}
__device__ __forceinline__ float eamd_drop_inv(unsigned thr16) {
  return thr16 >= 65536u ? 0.f : 65536.0f / (float)(65536u - thr16);
}
__device__ __forceinline__ unsigned eamd_drop_pair(unsigned seed, unsigned long long pair_idx) {
  unsigned h = ((unsigned)pair_idx ^ seed) + (unsigned)(pair_idx >> 32) * 0x9E3779B1u;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool eamd_drop_keep(unsigned seed, unsigned long long idx, unsigned thr16) {
  const unsigned h = eamd_drop_pair(seed, idx >> 1);
  return ((idx & 1ULL) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}
// four consecutive elements starting at an EVEN index: two hashes
__device__ __forceinline__ void eamd_drop_keep4(unsigned seed, unsigned long long base, unsigned thr16, bool (&k)[4]) {
  const unsigned h0 = eamd_drop_pair(seed, base >> 1), h1 = eamd_drop_pair(seed, (base >> 1) + 1ULL);
  k[0] = (h0 & 0xffffu) >= thr16; k[1] = (h0 >> 16) >= thr16;
  k[2] = (h1 & 0xffffu) >= thr16; k[3] = (h1 >> 16) >= thr16;
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
