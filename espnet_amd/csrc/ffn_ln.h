// LayerNorm of the 32 input rows of a fused feed-forward workgroup while they are staged (eamd_ffn_t.ln_x).
// reference: transformer/layer_norm.py:12-38 in front of positionwise_feed_forward.py:12-32 (encoder_layer.py:96-103).
// 512 threads: thread t owns row t >> 4, columns 4 (l + 16 j) .. + 3 (l = t & 15, j = 0 .. 3) - a row is 16 lanes of one wave,
// so both row sums are four DPP exchanges.  Same arithmetic as layernorm_fwd_vec_kernel (rowops.hip): mean, then the
// centred sum of squares of the values held in registers, rsqrt(var + eps).  The normalised rows go to LDS through
// put(row, col, y4), to p.x in the operand dtype (backward's weight gradient reads them) and mean / rstd to ln_mean / ln_rstd.
#pragma once
#include "common.h"
#include "../../include/espnet_amd.h"

template <typename Put>
__device__ __forceinline__ void ffn_ln_stage(const eamd_ffn_t& p, const int m0, const int t, Put&& put) {
  constexpr int D = 256;
  const int row = t >> 4, l = t & 15;
  const bool live = m0 + row < p.M;
  const long gr = (long)min(m0 + row, p.M - 1);
  const float4* __restrict__ xr = reinterpret_cast<const float4*>(p.ln_x + gr * D);
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = xr[l + 16 * j]; s += v[j].x + v[j].y + v[j].z + v[j].w; }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
  const float mean = s / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += a * a + b * b + c * c + d * d;
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
  const float rstd = rsqrtf(q / D + p.ln_eps);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(p.ln_w);
  const float4* __restrict__ b4 = reinterpret_cast<const float4*>(p.ln_b);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 g = g4[l + 16 * j], b = b4[l + 16 * j];
    float4 o;
    o.x = (v[j].x - mean) * rstd * g.x + b.x;
    o.y = (v[j].y - mean) * rstd * g.y + b.y;
    o.z = (v[j].z - mean) * rstd * g.z + b.z;
    o.w = (v[j].w - mean) * rstd * g.w + b.w;
    const int col = (l + 16 * j) * 4;
    put(row, col, o);
    if (live) {
      if (p.dtype == 1) {
        uint2 h;
        h.x = eamd_f2bf(o.x) | ((unsigned)eamd_f2bf(o.y) << 16);
        h.y = eamd_f2bf(o.z) | ((unsigned)eamd_f2bf(o.w) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(const_cast<float*>(p.x)) + gr * D + col) = h;
      } else {
        *reinterpret_cast<float4*>(const_cast<float*>(p.x) + gr * D + col) = o;
      }
    }
  }
  if (live && l == 0) { p.ln_mean[gr] = mean; p.ln_rstd[gr] = rstd; }
}
