// LayerNorm backward over the 32 finished rows of a row-block kernel (rowproj_f32.hip: dxn = dy W; ffn_f32.hip: dxn = dz W1):
// the rows never leave the workgroup between the product and the normalisation's backward.
// reference: transformer/layer_norm.py:12-38 (nn.LayerNorm, eps 1e-12) behind conformer/encoder_layer.py:96-146's pre-norm blocks.
// 512 threads; thread t owns row t >> 4, columns 4 (l + 16 j) .. + 3 (l = t & 15, j = 0 .. 3): a row is 16 lanes of one wave, both
// row sums are four DPP / shuffle exchanges (the arithmetic of layernorm_bwd_kernel<1> in rowops.hip):
//   dx = rstd (dy gamma - mean_c(dy gamma) - xhat mean_c(dy gamma xhat)) + dres,   optional dropped copy (eamd_dropout's mask of
//   the contiguous [M, 256] tensor),   ws[workgroup][2][256] = column sums of dy xhat (d gamma) and dy (d beta) over the 32 rows.
// `tile` = the rows in LDS ([32][TLD], complete: the caller has synchronised); gs / bsum = 2 x [32][256] floats of LDS scratch.
#pragma once
#include "common.h"

struct EamdLnbArgs {
  const float* x; const float* gamma; const float* mean; const float* rstd; const float* dres;
  float* ws; float* drop_out; float drop_p; unsigned long long drop_salt; const void* drop_step;
  float* out; long ldo; int M;
};

template <int TLD>
__device__ __forceinline__ void eamd_ln_bwd_rows32(const EamdLnbArgs& a, const float* tile, float* gs, float* bsum, const int m0,
                                                   const int t, const int blk) {
  constexpr int D = 256;
  const int row = t >> 4, l = t & 15;
  const bool live = m0 + row < a.M;
  const long gr = (long)min(m0 + row, a.M - 1);
  const float mu = a.mean[gr], rs = a.rstd[gr];
  float4 dq[4], h4[4], g4[4], rv[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = (l + 16 * j) * 4;
    dq[j] = *reinterpret_cast<const float4*>(&tile[row * TLD + col]);
    const float4 xv = *reinterpret_cast<const float4*>(a.x + gr * D + col);
    g4[j] = *reinterpret_cast<const float4*>(a.gamma + col);
    rv[j] = a.dres ? *reinterpret_cast<const float4*>(a.dres + gr * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    h4[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
    const float p0 = dq[j].x * g4[j].x, p1 = dq[j].y * g4[j].y, p2 = dq[j].z * g4[j].z, p3 = dq[j].w * g4[j].w;
    s1 += (p0 + p1) + (p2 + p3);
    s2 += (p0 * h4[j].x + p1 * h4[j].y) + (p2 * h4[j].z + p3 * h4[j].w);
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
  s1 /= D; s2 /= D;
  const unsigned thr_d = eamd_drop_thr16(a.drop_p);
  const float inv_d = eamd_drop_inv(thr_d);
  const unsigned seed_d = a.drop_out ? eamd_drop_seed((const unsigned long long*)a.drop_step, a.drop_salt) : 0u;
  const float z = live ? 1.f : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = (l + 16 * j) * 4;
    float4 o;
    o.x = rs * (dq[j].x * g4[j].x - s1 - h4[j].x * s2) + rv[j].x;
    o.y = rs * (dq[j].y * g4[j].y - s1 - h4[j].y * s2) + rv[j].y;
    o.z = rs * (dq[j].z * g4[j].z - s1 - h4[j].z * s2) + rv[j].z;
    o.w = rs * (dq[j].w * g4[j].w - s1 - h4[j].w * s2) + rv[j].w;
    if (live) {
      *reinterpret_cast<float4*>(a.out + gr * a.ldo + col) = o;
      if (a.drop_out) {
        bool keep[4];
        eamd_drop_keep4(seed_d, (unsigned long long)(gr * D + col), thr_d, keep);
        *reinterpret_cast<float4*>(a.drop_out + gr * D + col) =
            make_float4(keep[0] ? o.x * inv_d : 0.f, keep[1] ? o.y * inv_d : 0.f, keep[2] ? o.z * inv_d : 0.f,
                        keep[3] ? o.w * inv_d : 0.f);
      }
    }
    *reinterpret_cast<float4*>(&gs[row * D + col]) =
        make_float4(z * dq[j].x * h4[j].x, z * dq[j].y * h4[j].y, z * dq[j].z * h4[j].z, z * dq[j].w * h4[j].w);
    *reinterpret_cast<float4*>(&bsum[row * D + col]) = make_float4(z * dq[j].x, z * dq[j].y, z * dq[j].z, z * dq[j].w);
  }
  __syncthreads();
  // thread t sums one column of d gamma (t < 256) or d beta over the 32 rows: the workgroup's partial for the batched second
  // stage (eamd_layernorm_bwd_reduce; ws[workgroup][2 D])
  const float* src = (t < D ? gs : bsum) + (t & (D - 1));
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) s += src[r * D];
  a.ws[(long)blk * 2 * D + t] = s;
}
