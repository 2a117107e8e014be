// Row kernels (HBM-bound): LayerNorm fwd/bwd, masked softmax with the legacy rel-shift fused in,
// label-smoothing KL loss (+gradient, +accuracy), row argmax.  One 64-lane wave owns one row
// unless stated; rows are staged in registers / LDS so each element is read from HBM once.
#include <stdlib.h>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm  (reference: transformer/layer_norm.py:12-38 = nn.LayerNorm(d, eps=1e-12))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ y, unsigned short* __restrict__ y16,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int D,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mean = wave_sum(s) / D;
  float v = 0.f;
  for (int i = lane; i < D; i += 64) { float d = xr[i] - mean; v += d * d; }
  const float rstd = rsqrtf(wave_sum(v) / D + eps);
  for (int i = lane; i < D; i += 64) {
    const float o = (xr[i] - mean) * rstd * gamma[i] + beta[i];
    if (y) y[(long)row * D + i] = o;
    if (y16) y16[(long)row * D + i] = eamd_f2bf(o);
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// D % 256 == 0, D <= 1024: each lane keeps its float4 slices in registers.
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ y,
                                                                unsigned short* __restrict__ y16,
                                                                float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, int rows,
                                                                float eps) {
  constexpr int D = VEC * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
  float4 v[VEC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) { v[j] = xr[lane + 64 * j]; s += v[j].x + v[j].y + v[j].z + v[j].w; }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += a * a + b * b + c * c + d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
  float4* yr = y ? reinterpret_cast<float4*>(y + (long)row * D) : nullptr;
  uint2* yr16 = y16 ? reinterpret_cast<uint2*>(y16 + (long)row * D) : nullptr;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float4 g = g4[lane + 64 * j], b = b4[lane + 64 * j], o;
    o.x = (v[j].x - mean) * rstd * g.x + b.x;
    o.y = (v[j].y - mean) * rstd * g.y + b.y;
    o.z = (v[j].z - mean) * rstd * g.z + b.z;
    o.w = (v[j].w - mean) * rstd * g.w + b.w;
    if (yr) yr[lane + 64 * j] = o;
    if (yr16) {
      uint2 h;
      h.x = eamd_f2bf(o.x) | ((unsigned)eamd_f2bf(o.y) << 16);
      h.y = eamd_f2bf(o.z) | ((unsigned)eamd_f2bf(o.w) << 16);
      yr16[lane + 64 * j] = h;
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// Backward: dx per row; dgamma/dbeta accumulated per lane-column over the block's rows, combined
// across the 4 waves through LDS and added to global with one atomic per column per block.
// Lane l owns columns {4l..4l+3} + 256j (float4 accesses) when D % 256 == 0, else {l + 64c}.
template <int VEC>   // VEC = D/256 for the vector form, 0 = generic (D <= 1024)
__global__ __launch_bounds__(1024) void layernorm_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* dres, float* dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ws, int rows, int D,
    int rows_per_block, unsigned short* __restrict__ drop16, float drop_p, const unsigned long long* drop_step,
    unsigned long long drop_salt, int drop_f32) {
  // drop16 (vector form only): a second, bf16 output dropout(dx; p, salt) with the mask eamd_dropout would draw for
  // the contiguous [rows, D] tensor - the incoming-gradient dropout + cast of the PREVIOUS block, fused into this
  // kernel's store instead of a separate pass over dx
  extern __shared__ float lds[];  // [waves][2][D]: one slot per wave, summed after the barrier (plain stores:
                                  // ds_add_f32 runs at a fraction of a lane per clock on gfx950)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* slot = lds + (long)wave * 2 * D;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  if constexpr (VEC > 0) {
    const unsigned drop_thr = eamd_drop_thr16(drop_p);
    const float drop_inv = eamd_drop_inv(drop_thr);
    const unsigned drop_seed = drop16 ? eamd_drop_seed(drop_step, drop_salt) : 0u;
    float4 ag[VEC], ab[VEC], g4[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      ag[j] = make_float4(0.f, 0.f, 0.f, 0.f); ab[j] = ag[j];
      g4[j] = reinterpret_cast<const float4*>(gamma)[lane + 64 * j];
    }
    // two rows per trip, every operand of both rows (dy, x, residual gradient) requested before the first use:
    // one memory round trip per pair instead of two dependent ones per row
    const int nw = blockDim.x >> 6;
    for (int row = r0 + wave; row < r1; row += 2 * nw) {
      const int row2 = row + nw;
      const bool two = row2 < r1;
      const int rb = two ? row2 : row;          // second row falls back to the first (results discarded)
      float4 d4[2][VEC], xv[2][VEC], rv[2][VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        d4[0][j] = reinterpret_cast<const float4*>(dy + (long)row * D)[lane + 64 * j];
        xv[0][j] = reinterpret_cast<const float4*>(x + (long)row * D)[lane + 64 * j];
        d4[1][j] = reinterpret_cast<const float4*>(dy + (long)rb * D)[lane + 64 * j];
        xv[1][j] = reinterpret_cast<const float4*>(x + (long)rb * D)[lane + 64 * j];
        if (dres) {
          rv[0][j] = reinterpret_cast<const float4*>(dres + (long)row * D)[lane + 64 * j];
          rv[1][j] = reinterpret_cast<const float4*>(dres + (long)rb * D)[lane + 64 * j];
        }
      }
      const float mu[2] = {mean[row], mean[rb]}, rs[2] = {rstd[row], rstd[rb]};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q == 1 && !two) break;
        float4 h4[VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float4 xq = xv[q][j], dq = d4[q][j];
          h4[j] = make_float4((xq.x - mu[q]) * rs[q], (xq.y - mu[q]) * rs[q], (xq.z - mu[q]) * rs[q], (xq.w - mu[q]) * rs[q]);
          float a = dq.x * g4[j].x, b = dq.y * g4[j].y, c = dq.z * g4[j].z, d = dq.w * g4[j].w;
          s1 += (a + b) + (c + d);
          s2 += (a * h4[j].x + b * h4[j].y) + (c * h4[j].z + d * h4[j].w);
          ag[j].x += dq.x * h4[j].x; ag[j].y += dq.y * h4[j].y; ag[j].z += dq.z * h4[j].z; ag[j].w += dq.w * h4[j].w;
          ab[j].x += dq.x; ab[j].y += dq.y; ab[j].z += dq.z; ab[j].w += dq.w;
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
        float4* dxr = reinterpret_cast<float4*>(dx + (long)(q ? row2 : row) * D);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float4 dq = d4[q][j];
          float4 o;
          o.x = rs[q] * (dq.x * g4[j].x - s1 - h4[j].x * s2);
          o.y = rs[q] * (dq.y * g4[j].y - s1 - h4[j].y * s2);
          o.z = rs[q] * (dq.z * g4[j].z - s1 - h4[j].z * s2);
          o.w = rs[q] * (dq.w * g4[j].w - s1 - h4[j].w * s2);
          if (dres) { o.x += rv[q][j].x; o.y += rv[q][j].y; o.z += rv[q][j].z; o.w += rv[q][j].w; }
          dxr[lane + 64 * j] = o;
          if (drop16) {
            const long e0 = (long)(q ? row2 : row) * D + 4 * (lane + 64 * j);
            const float ov[4] = {o.x, o.y, o.z, o.w};
            bool keep[4];
            eamd_drop_keep4(drop_seed, (unsigned long long)e0, drop_thr, keep);     // e0 is a multiple of 4
            if (drop_f32) {          // fp32 copy (reference-precision mode); wave-uniform
              *reinterpret_cast<float4*>(reinterpret_cast<float*>(drop16) + e0) =
                  make_float4(keep[0] ? ov[0] * drop_inv : 0.f, keep[1] ? ov[1] * drop_inv : 0.f,
                              keep[2] ? ov[2] * drop_inv : 0.f, keep[3] ? ov[3] * drop_inv : 0.f);
            } else {
              unsigned short h16[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) h16[e] = eamd_f2bf(keep[e] ? ov[e] * drop_inv : 0.f);
              uint2 pk;
              pk.x = (unsigned)h16[0] | ((unsigned)h16[1] << 16);
              pk.y = (unsigned)h16[2] | ((unsigned)h16[3] << 16);
              *reinterpret_cast<uint2*>(drop16 + e0) = pk;
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int i = 4 * (lane + 64 * j);
      *reinterpret_cast<float4*>(slot + i) = ag[j];
      *reinterpret_cast<float4*>(slot + D + i) = ab[j];
    }
  } else {
    constexpr int MAXC = 16;
    float ag[MAXC], ab[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { ag[c] = 0.f; ab[c] = 0.f; }
    for (int row = r0 + wave; row < r1; row += (int)(blockDim.x >> 6)) {
      const float* dyr = dy + (long)row * D;
      const float* xr = x + (long)row * D;
      const float mu = mean[row], rs = rstd[row];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        int i = lane + 64 * c;
        if (i < D) {
          float xh = (xr[i] - mu) * rs;
          float dg = dyr[i] * gamma[i];
          s1 += dg; s2 += dg * xh;
          ag[c] += dyr[i] * xh; ab[c] += dyr[i];
        }
      }
      s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
      float* dxr = dx + (long)row * D;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        int i = lane + 64 * c;
        if (i < D) {
          float xh = (xr[i] - mu) * rs;
          float g = rs * (dyr[i] * gamma[i] - s1 - xh * s2);
          if (dres) g += dres[(long)row * D + i];
          dxr[i] = g;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int i = lane + 64 * c;
      if (i < D) { slot[i] = ag[c]; slot[D + i] = ab[c]; }
    }
  }
  __syncthreads();
  const int nwv = blockDim.x >> 6;
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) {
    float t = 0.f;
    for (int wv = 0; wv < nwv; ++wv) t += lds[(long)wv * 2 * D + i];
    if (ws) ws[(long)blockIdx.x * 2 * D + i] = t;   // per-block partials, combined by layernorm_bwd_reduce_kernel
    else if (i < D) atomicAdd(&dgamma[i], t);
    else atomicAdd(&dbeta[i - D], t);
  }
}

// ws[nblk][2D] -> dgamma[D] += , dbeta[D] +=.  grid (ceil(2D/64), row slices): each block sums its slice of
// partial rows for 64 columns (4 row-subgroups x 64 lanes, combined in LDS) and issues one atomic per column.
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ ws, int nblk, int D,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nblk, r0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < 2 * D) {
    int r = r0 + sub;
    for (; r + 12 < r1; r += 16) {
      s0 += ws[(long)r * 2 * D + col];        s1 += ws[(long)(r + 4) * 2 * D + col];
      s2 += ws[(long)(r + 8) * 2 * D + col];  s3 += ws[(long)(r + 12) * 2 * D + col];
    }
    for (; r < r1; r += 4) s0 += ws[(long)r * 2 * D + col];
  }
  red[sub][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && col < 2 * D) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (col < D) atomicAdd(&dgamma[col], v);
    else atomicAdd(&dbeta[col - D], v);
  }
}

// Deferred second stage: the partials of up to LN_JOBS LayerNorm backward passes (job table in the kernel argument
// segment) are summed by ONE launch at the end of backward instead of one launch per LayerNorm (80 per training
// step of the 12-layer Conformer).  grid (ceil(2D/64), njobs); one block owns 64 columns of one job.
constexpr int LN_JOBS = 64;
struct LnJobTable { eamd_ln_reduce_job_t j[LN_JOBS]; };
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_batched_kernel(const LnJobTable tab) {
  __shared__ float red[4][64];
  const eamd_ln_reduce_job_t jb = tab.j[blockIdx.y];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int D2 = 2 * jb.D;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < D2) {
    const float* __restrict__ ws = jb.ws;
    int r = sub;
    for (; r + 12 < jb.nblk; r += 16) {
      s0 += ws[(long)r * D2 + col];        s1 += ws[(long)(r + 4) * D2 + col];
      s2 += ws[(long)(r + 8) * D2 + col];  s3 += ws[(long)(r + 12) * D2 + col];
    }
    for (; r < jb.nblk; r += 4) s0 += ws[(long)r * D2 + col];
  }
  red[sub][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && col < D2) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (col < jb.D) atomicAdd(&jb.dgamma[col], v);
    else atomicAdd(&jb.dbeta[col - jb.D], v);
  }
}

// ---------------------------------------------------------------------------------------------
// Masked softmax over attention scores with the legacy Transformer-XL rel-shift folded in.
// reference: transformer/attention.py:63-90 (masked_fill(min) -> softmax -> masked_fill(0)),
//            attention.py:141-162 (rel_shift), attention.py:200-204 ((ac + bd) / sqrt(d_k)).
// scores layout: [nb][T1][ld] rows, nb = H*B blocks (block index z -> batch b = z % B).
//   s[i,j] = scale * (ac[i,j] + (bd ? shift(bd)[i,j] : 0))
//   shift(x)[i,j] = P_flat[T1 + i*T2 + j] with P = [0 | x] of shape (T1, T2+1)
// mask: uint8, element (b,i,j) at mask[b*mb + i*mi + j] (mi = 0 broadcasts over queries); 0 = masked.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float shifted_bd(const float* __restrict__ bd, int T1, int T2, long ld, int i,
                                            int j) {
  int f = T1 + i * T2 + j;
  int r = f / (T2 + 1), c = f % (T2 + 1);
  return c == 0 ? 0.f : bd[(long)r * ld + (c - 1)];
}

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* ac,
                                                          const float* __restrict__ bd,
                                                          const unsigned char* __restrict__ mask, long mb,
                                                          long mi, float* P, unsigned short* P16, int nb, int B,
                                                          int T1, int T2, long ld, float scale) {
  extern __shared__ float lds[];  // [4][T2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rowid = (long)blockIdx.x * 4 + wave;
  if (rowid >= (long)nb * T1) return;
  const int z = rowid / T1, i = rowid % T1;
  const int b = z % B;
  const float* acr = ac + ((long)z * T1 + i) * ld;
  const float* bdz = bd ? bd + (long)z * T1 * ld : nullptr;
  const unsigned char* mr = mask ? mask + b * mb + i * mi : nullptr;
  float* buf = lds + wave * T2;
  float mx = -INFINITY;
  for (int j = lane; j < T2; j += 64) {
    float v = acr[j];
    if (bdz) v += shifted_bd(bdz, T1, T2, ld, i, j);
    v *= scale;
    if (mr && mr[j] == 0) v = -INFINITY;
    buf[j] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  const long ro = ((long)z * T1 + i) * ld;
  if (mx == -INFINITY) {  // every key masked: softmax(min,...)=uniform, then masked_fill(0) -> zeros
    for (int j = lane; j < ld; j += 64) { if (P16) P16[ro + j] = 0; else P[ro + j] = 0.f; }
    return;
  }
  float s = 0.f;
  for (int j = lane; j < T2; j += 64) { float e = __expf(buf[j] - mx); buf[j] = e; s += e; }
  const float inv = 1.f / wave_sum(s);
  for (int j = lane; j < ld; j += 64) {
    const float o = j < T2 ? buf[j] * inv : 0.f;
    if (P16) P16[ro + j] = eamd_f2bf(o); else P[ro + j] = o;
  }
}

// Vector form of the forward for bf16 probabilities and rows of at most 4 * 256 columns (ld % 4 == 0, 16-byte aligned
// rows): a lane owns 4 adjacent columns per 256-column chunk, the whole row lives in registers (no LDS pass), the
// scores are read as float4 and the probabilities leave as one 8-byte store per chunk - 2-byte stores run at a
// fraction of the dword rate on gfx950.
template <int NCH>
__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(const float* __restrict__ ac, const float* __restrict__ bd,
                                                              const unsigned char* __restrict__ mask, long mb, long mi,
                                                              unsigned short* __restrict__ P16, int nb, int B, int T1,
                                                              int T2, long ld, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rowid = (long)blockIdx.x * 4 + wave;
  if (rowid >= (long)nb * T1) return;
  const int z = rowid / T1, i = rowid % T1;
  const int b = z % B;
  const long ro = ((long)z * T1 + i) * ld;
  const float* bdz = bd ? bd + (long)z * T1 * ld : nullptr;
  const unsigned char* mr = mask ? mask + b * mb + i * mi : nullptr;
  float v[NCH][4];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int j0 = 4 * lane + 256 * k;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 < ld) a = *reinterpret_cast<const float4*>(ac + ro + j0);
    const float av[4] = {a.x, a.y, a.z, a.w};
    // rel-shift source of column j0 (one division per chunk; the next columns step through the (T2+1)-wide rows)
    int sr = 0, sc = 0;
    if (bdz) { const int f = T1 + i * T2 + j0; sr = f / (T2 + 1); sc = f % (T2 + 1); }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e;
      float x = -INFINITY;
      if (j < T2) {
        x = av[e];
        if (bdz && sc != 0) x += bdz[(long)sr * ld + (sc - 1)];
        x *= scale;
        if (mr && mr[j] == 0) x = -INFINITY;
      }
      if (++sc > T2) { sc = 0; ++sr; }
      v[k][e] = x;
      mx = fmaxf(mx, x);
    }
  }
  mx = wave_max(mx);
  const bool dead = mx == -INFINITY;      // every key masked: softmax(min,...) = uniform, then masked_fill(0) -> zeros
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float ex = dead ? 0.f : __expf(v[k][e] - mx); v[k][e] = ex; s += ex; }
  const float inv = dead ? 0.f : 1.f / wave_sum(s);
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int j0 = 4 * lane + 256 * k;
    if (j0 < ld) {
      uint2 o;
      o.x = (unsigned)eamd_f2bf(v[k][0] * inv) | ((unsigned)eamd_f2bf(v[k][1] * inv) << 16);
      o.y = (unsigned)eamd_f2bf(v[k][2] * inv) | ((unsigned)eamd_f2bf(v[k][3] * inv) << 16);
      *reinterpret_cast<uint2*>(P16 + ro + j0) = o;
    }
  }
}

// Vector form of the backward (bf16 P in, bf16 dS out): 8-byte P loads, float4 gradient loads, 8-byte dS stores; the
// inverse rel-shift scatter into dbd keeps its 2-byte stores (its destination is shifted by one element per row).
template <int NCH>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const unsigned short* __restrict__ P16,
                                                              const float* __restrict__ dP,
                                                              unsigned short* __restrict__ dS16,
                                                              unsigned short* __restrict__ dbd16, int nb, int T1, int T2,
                                                              long ld, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rowid = (long)blockIdx.x * 4 + wave;
  if (rowid >= (long)nb * T1) return;
  const int z = rowid / T1, i = rowid % T1;
  const long ro = ((long)z * T1 + i) * ld;
  const long zo = (long)z * T1 * ld;
  float pv[NCH][4], dv[NCH][4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int j0 = 4 * lane + 256 * k;
    uint2 pr = make_uint2(0u, 0u);
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 < ld) { pr = *reinterpret_cast<const uint2*>(P16 + ro + j0); d = *reinterpret_cast<const float4*>(dP + ro + j0); }
    pv[k][0] = __uint_as_float(pr.x << 16); pv[k][1] = __uint_as_float(pr.x & 0xffff0000u);
    pv[k][2] = __uint_as_float(pr.y << 16); pv[k][3] = __uint_as_float(pr.y & 0xffff0000u);
    dv[k][0] = d.x; dv[k][1] = d.y; dv[k][2] = d.z; dv[k][3] = d.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (j0 + e < T2) s += pv[k][e] * dv[k][e];
  }
  s = wave_sum(s);
  if (dbd16) {      // dbd is fully defined here: pad columns of this row, and (row 0) the head the scatter never reaches
    for (int j = T2 + lane; j < ld; j += 64) dbd16[ro + j] = 0;
    if (i == 0)
      for (int f = 1 + lane; f < T1; f += 64) {
        const int r = f / (T2 + 1), c = f % (T2 + 1);
        if (c != 0) dbd16[zo + (long)r * ld + (c - 1)] = 0;
      }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int j0 = 4 * lane + 256 * k;
    if (j0 >= ld) continue;
    unsigned short g16[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e;
      const float g = j < T2 ? pv[k][e] * (dv[k][e] - s) * scale : 0.f;
      g16[e] = eamd_f2bf(g);
      if (dbd16 && j < T2) {
        const int f = T1 + i * T2 + j;
        const int r = f / (T2 + 1), c = f % (T2 + 1);
        if (c != 0) dbd16[zo + (long)r * ld + (c - 1)] = g16[e];
      }
    }
    uint2 o;
    o.x = (unsigned)g16[0] | ((unsigned)g16[1] << 16);
    o.y = (unsigned)g16[2] | ((unsigned)g16[3] << 16);
    *reinterpret_cast<uint2*>(dS16 + ro + j0) = o;
  }
}

// dS = P * (dP - sum_j dP*P) * scale, written over dP (d_ac); optional scatter of dS through the
// inverse rel-shift into dbd (every element of dbd is written: no pre-zeroing needed).
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P,
                                                          const unsigned short* __restrict__ P16, float* dP,
                                                          float* __restrict__ dbd, unsigned short* __restrict__ dS16,
                                                          unsigned short* __restrict__ dbd16, int nb, int T1, int T2,
                                                          long ld, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rowid = (long)blockIdx.x * 4 + wave;
  if (rowid >= (long)nb * T1) return;
  const int z = rowid / T1, i = rowid % T1;
  const long ro = ((long)z * T1 + i) * ld;
  float* dr = dP + ro;
  float s = 0.f;
  for (int j = lane; j < T2; j += 64) {
    const float pv = P16 ? __uint_as_float(((unsigned)P16[ro + j]) << 16) : P[ro + j];
    s += pv * dr[j];
  }
  s = wave_sum(s);
  const long zo = (long)z * T1 * ld;
  // dbd is fully defined here (no pre-zeroing by the caller): the scatter below covers every (r, c-1) whose padded
  // index r*(T2+1)+c lies in [T1, T1*(T2+1)); the wave of row 0 zeroes the head [1, T1) it never reaches, and every
  // wave zeroes the pad columns [T2, ld) of its own row
  if (dbd || dbd16) {
    for (int j = T2 + lane; j < ld; j += 64) { if (dbd16) dbd16[ro + j] = 0; else dbd[ro + j] = 0.f; }
    if (i == 0)
      for (int f = 1 + lane; f < T1; f += 64) {
        const int r = f / (T2 + 1), c = f % (T2 + 1);
        if (c != 0) { if (dbd16) dbd16[zo + (long)r * ld + (c - 1)] = 0; else dbd[zo + (long)r * ld + (c - 1)] = 0.f; }
      }
  }
  for (int j = lane; j < ld; j += 64) {
    if (j >= T2) { if (dS16) dS16[ro + j] = 0; continue; }
    const float pv = P16 ? __uint_as_float(((unsigned)P16[ro + j]) << 16) : P[ro + j];
    float g = pv * (dr[j] - s) * scale;
    if (dS16) dS16[ro + j] = eamd_f2bf(g); else dr[j] = g;
    if (dbd || dbd16) {
      int f = T1 + i * T2 + j;
      int r = f / (T2 + 1), c = f % (T2 + 1);
      if (c != 0) {
        if (dbd16) dbd16[zo + (long)r * ld + (c - 1)] = eamd_f2bf(g);
        else dbd[zo + (long)r * ld + (c - 1)] = g;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Label-smoothing KL loss + gradient + accuracy
// reference: transformer/label_smoothing_loss.py:44-63, nets_utils.py:299-319 (th_accuracy)
//   true_dist = eps/(V-1) everywhere, 1-eps at target; loss_row = sum_v td*(log td - logp_v)
// One 256-thread block per row; grad (unscaled by upstream) = (softmax - td) * inv_denom.
// ---------------------------------------------------------------------------------------------
// Register-resident form for rows of up to 8192 logits (V % 4 == 0, 16-byte aligned rows): the row is read ONCE as
// float4s (<= 8 per thread), max / argmax, sum-exp, sum and the gradient all come from registers, the gradient leaves
// as float4 stores - the three-pass form below re-reads the row twice through L1 with scalar loads.
template <int NV>
__global__ __launch_bounds__(256) void lsm_loss_reg_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                           float* __restrict__ loss_rows, float* __restrict__ correct_rows,
                                                           float* __restrict__ grad, int V, int ignore_id, float smoothing,
                                                           float inv_denom) {
  __shared__ float red[16];
  __shared__ int redi[16];
  const int row = blockIdx.x;
  const float4* xr = reinterpret_cast<const float4*>(x + (long)row * V);
  float4* gr = grad ? reinterpret_cast<float4*>(grad + (long)row * V) : nullptr;
  const long long tg = target[row];
  const int nq = V >> 2;
  if (tg == ignore_id) {
    if (gr) for (int q = threadIdx.x; q < nq; q += 256) gr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x == 0) { loss_rows[row] = 0.f; correct_rows[row] = 0.f; }
    return;
  }
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = threadIdx.x + 256 * i;
    v[i] = q < nq ? xr[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  }
  float mx = -INFINITY; int am = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NV; ++i) {          // ascending index order within the thread: first maximum wins
    const int base = (threadIdx.x + 256 * i) * 4;
    if (v[i].x > mx) { mx = v[i].x; am = base; }
    if (v[i].y > mx) { mx = v[i].y; am = base + 1; }
    if (v[i].z > mx) { mx = v[i].z; am = base + 2; }
    if (v[i].w > mx) { mx = v[i].w; am = base + 3; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(mx, o, 64); int oi = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oi < am)) { mx = om; am = oi; }
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { red[w] = mx; redi[w] = am; }
  __syncthreads();
  mx = red[0]; am = redi[0];
  for (int k = 1; k < 4; ++k)
    if (red[k] > mx || (red[k] == mx && redi[k] < am)) { mx = red[k]; am = redi[k]; }
  __syncthreads();
  float se = 0.f, sx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (threadIdx.x + 256 * i < nq) {
      se += (__expf(v[i].x - mx) + __expf(v[i].y - mx)) + (__expf(v[i].z - mx) + __expf(v[i].w - mx));
      sx += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  se = block_sum(se, red);
  sx = block_sum(sx, red);
  const float lse = mx + __logf(se);
  const float conf = 1.f - smoothing;
  const float low = smoothing / (V - 1);
  const float logp_t = x[(long)row * V + tg] - lse;
  const float sum_logp = sx - V * lse;
  float loss = -conf * logp_t - low * (sum_logp - logp_t);
  if (conf > 0.f) loss += conf * __logf(conf);
  if (low > 0.f) loss += (V - 1) * low * __logf(low);
  if (threadIdx.x == 0) {
    loss_rows[row] = loss;
    correct_rows[row] = (am == (int)tg) ? 1.f : 0.f;
  }
  if (gr) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = threadIdx.x + 256 * i;
      if (q < nq) {
        const int base = q * 4;
        float4 g;
        g.x = (__expf(v[i].x - lse) - ((base == (int)tg) ? conf : low)) * inv_denom;
        g.y = (__expf(v[i].y - lse) - ((base + 1 == (int)tg) ? conf : low)) * inv_denom;
        g.z = (__expf(v[i].z - lse) - ((base + 2 == (int)tg) ? conf : low)) * inv_denom;
        g.w = (__expf(v[i].w - lse) - ((base + 3 == (int)tg) ? conf : low)) * inv_denom;
        gr[q] = g;
      }
    }
  }
}

__global__ __launch_bounds__(256) void lsm_loss_kernel(const float* __restrict__ x,
                                                       const long long* __restrict__ target,
                                                       float* __restrict__ loss_rows,
                                                       float* __restrict__ correct_rows,
                                                       float* __restrict__ grad, int V, int ignore_id,
                                                       float smoothing, float inv_denom) {
  __shared__ float red[16];
  __shared__ int redi[16];
  const int row = blockIdx.x;
  const float* xr = x + (long)row * V;
  float* gr = grad ? grad + (long)row * V : nullptr;
  const long long tg = target[row];
  if (tg == ignore_id) {
    if (gr) for (int v = threadIdx.x; v < V; v += blockDim.x) gr[v] = 0.f;
    if (threadIdx.x == 0) { loss_rows[row] = 0.f; correct_rows[row] = 0.f; }
    return;
  }
  // max + first argmax
  float mx = -INFINITY; int am = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float a = xr[v];
    if (a > mx) { mx = a; am = v; }
  }
  // wave reduce (value, index) with lowest-index tie-break (torch.argmax semantics)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(mx, o, 64); int oi = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oi < am)) { mx = om; am = oi; }
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { red[w] = mx; redi[w] = am; }
  __syncthreads();
  mx = red[0]; am = redi[0];
  for (int k = 1; k < (blockDim.x >> 6); ++k)
    if (red[k] > mx || (red[k] == mx && redi[k] < am)) { mx = red[k]; am = redi[k]; }
  __syncthreads();
  float se = 0.f, sx = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) { float a = xr[v]; se += __expf(a - mx); sx += a; }
  se = block_sum(se, red);
  sx = block_sum(sx, red);
  const float lse = mx + __logf(se);
  const float conf = 1.f - smoothing;
  const float low = smoothing / (V - 1);
  const float logp_t = xr[tg] - lse;
  const float sum_logp = sx - V * lse;
  float loss = -conf * logp_t - low * (sum_logp - logp_t);
  if (conf > 0.f) loss += conf * __logf(conf);
  if (low > 0.f) loss += (V - 1) * low * __logf(low);
  if (threadIdx.x == 0) {
    loss_rows[row] = loss;
    correct_rows[row] = (am == (int)tg) ? 1.f : 0.f;
  }
  if (gr) {
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      float p = __expf(xr[v] - lse);
      float td = (v == (int)tg) ? conf : low;
      gr[v] = (p - td) * inv_denom;
    }
  }
}

// Row argmax (first maximal index), one block per row.  reference: ctc.py:144-151 (CTC.argmax),
// e2e_asr_transformer.py:274-284 (greedy CTC decode).
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ld,
                                                          int* __restrict__ out, int V) {
  __shared__ float red[16];
  __shared__ int redi[16];
  const float* xr = x + (long)blockIdx.x * ld;
  float mx = -INFINITY; int am = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float a = xr[v];
    if (a > mx || (a == mx && v < am)) { mx = a; am = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(mx, o, 64); int oi = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oi < am)) { mx = om; am = oi; }
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { red[w] = mx; redi[w] = am; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = red[0]; am = redi[0];
    for (int k = 1; k < (blockDim.x >> 6); ++k)
      if (red[k] > mx || (red[k] == mx && redi[k] < am)) { mx = red[k]; am = redi[k]; }
    out[blockIdx.x] = am == 0x7fffffff ? 0 : am;
  }
}

// Deterministic single-block sum: out[0] = scale * sum(in[0..n)).
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ in, long n,
                                                          float* __restrict__ out, float scale) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// Row log-softmax (decoder scoring).  reference: decoder.py:318 (log_softmax in forward_one_step),
// ctc.py:134-142 (CTC.log_softmax).
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x,
                                                               float* __restrict__ y, int V) {
  // rows of up to 6144 elements are read ONCE and stay in registers over the three passes (maximum, sum of exponentials, write);
  // a thread's elements and the order of its sums are those of the re-reading loops: the same bits
  __shared__ float red[16];
  const float* xr = x + (long)blockIdx.x * V;
  float* yr = y + (long)blockIdx.x * V;
  constexpr int RMAX = 24;
  const int t = threadIdx.x;
  if (V <= RMAX * 256) {
    float reg[RMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      const int v = t + 256 * q;
      reg[q] = v < V ? xr[v] : -INFINITY;
      mx = fmaxf(mx, reg[q]);
    }
    mx = block_max(mx, red);
    float se = 0.f;
#pragma unroll
    for (int q = 0; q < RMAX; ++q)
      if (t + 256 * q < V) se += __expf(reg[q] - mx);
    se = block_sum(se, red);
    const float lse = mx + __logf(se);
#pragma unroll
    for (int q = 0; q < RMAX; ++q)
      if (t + 256 * q < V) yr[t + 256 * q] = reg[q] - lse;
    return;
  }
  float mx = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, xr[v]);
  mx = block_max(mx, red);
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) se += __expf(xr[v] - mx);
  se = block_sum(se, red);
  const float lse = mx + __logf(se);
  for (int v = threadIdx.x; v < V; v += blockDim.x) yr[v] = xr[v] - lse;
}

}  // namespace

// y[M, N] = alpha * act(a_act(x)[M, K] W[N, K]^T + bias) + R for a HANDFUL of rows (M <= 16: the hypotheses of one utterance in a
// beam step - query / output projections, both feed-forward products, the output layer; decoder_layer.py:77-134).  As a tile
// GEMM these are 1 x N/64 workgroups walking K alone (10.8 us at K = 256, 58 us at K = 2048); here a WAVE owns one output
// column: lanes split K in 16-byte pieces (a weight row is read once, coalesced; the M input rows come from L1), the M dot
// products meet in a wave reduction, lane m finishes row m.  Exact fp32 arithmetic, summation order (lane-strided partial sums,
// then the wave tree) differs from the MFMA kernels'.
namespace {
// (round 4) the M input rows are staged ONCE per workgroup in LDS, every load of the staging pass in flight together: with each
// wave fetching row after row from global memory the launch was ten dependent round trips long (12 us for 10 x 256 x 256).
constexpr int ROWS_KC = 1024;          // reduction chunk staged at a time: 16 rows x 1024 floats = 64 KB of LDS
__global__ __launch_bounds__(256) void linear_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const float* __restrict__ R,
                                                              float* __restrict__ y, int M, int N, int K, int a_act, int act,
                                                              float alpha, long ldx, long ldr) {
  extern __shared__ __attribute__((aligned(16))) float xs_rows[];      // [M][kc]
  const int lane = threadIdx.x & 63;
  const int n = min(blockIdx.x * 4 + (int)(threadIdx.x >> 6), N - 1);  // (surplus waves of the last workgroup redo column N - 1)
  const bool store = blockIdx.x * 4 + (int)(threadIdx.x >> 6) < N;
  float acc[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = 0.f;
  const float* wr = W + (long)n * K;
  for (int kb = 0; kb < K; kb += ROWS_KC) {
    const int kc = min(ROWS_KC, K - kb);
    float4 w4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lane * 4 + 256 * q;
      w4[q] = k < kc ? *reinterpret_cast<const float4*>(wr + kb + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (kb > 0) __syncthreads();
    const int c4 = kc >> 2;
    for (int idx = threadIdx.x; idx < M * c4; idx += 256) {
      const int m = idx / c4, c = idx - m * c4;
      float4 v = *reinterpret_cast<const float4*>(x + (long)m * ldx + kb + c * 4);
      if (a_act != EAMD_ACT_NONE) {
        v.x = eamd_act(v.x, a_act); v.y = eamd_act(v.y, a_act); v.z = eamd_act(v.z, a_act); v.w = eamd_act(v.w, a_act);
      }
      *reinterpret_cast<float4*>(&xs_rows[m * kc + c * 4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = lane * 4 + 256 * q;
          if (k < kc) {
            const float4 v = *reinterpret_cast<const float4*>(&xs_rows[m * kc + k]);
            acc[m] = fmaf(v.x, w4[q].x, fmaf(v.y, w4[q].y, fmaf(v.z, w4[q].z, fmaf(v.w, w4[q].w, acc[m]))));
          }
        }
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (m < M) {
      const float r = wave_sum(acc[m]);
      if (lane == m) mine = r;
    }
  }
  if (lane < M && store) {
    float v = mine + (bias ? bias[n] : 0.f);
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = eamd_swish(v);
    v *= alpha;
    if (R) v += R[(long)lane * ldr + n];
    y[(long)lane * N + n] = v;
  }
}

// K >= 1024 (the feed-forward down-product, K = 2048: 16 us as above - two staged chunks of the rows, every wave walking all of K):
// a WORKGROUP owns CB output columns and its four waves split K; a lane's pieces of the weight rows and of the input rows go
// straight to registers, all requested together (no staging pass, no barrier in front of the products), the waves' partial sums
// meet in LDS and are added in a fixed order.  Same exact fp32 products; the summation order differs from the kernel above.
template <int QN, int CB>
__global__ __launch_bounds__(256) void linear_rows_wide_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                   const float* __restrict__ bias, const float* __restrict__ R,
                                                                   float* __restrict__ y, int M, int N, int K, int a_act, int act,
                                                                   float alpha, long ldx, long ldr) {
  __shared__ float part[4][CB * 16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = blockIdx.x * CB;
  const int k0 = w * QN * 256 + lane * 4;                  // this lane's pieces: k0 + 256 q, q < QN   (4 * QN * 256 >= K)
  float4 w4[CB][QN];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const float* wr = W + (long)min(c0 + c, N - 1) * K;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int k = k0 + 256 * q;
      w4[c][q] = k < K ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float acc[CB][16];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[c][m] = 0.f;
  // rows in two halves of eight; both halves' loads are requested before the first product (QN <= 2: 32 float4 in flight)
  constexpr bool BOTH = QN <= 2;
  float4 xa[8][QN], xb[BOTH ? 8 : 1][QN];
  auto load8 = [&](float4 (*dst)[QN], int mb) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float* xr = x + (long)min(mb + m, M - 1) * ldx;
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int k = k0 + 256 * q;
        dst[m][q] = k < K ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto fma8 = [&](float4 (*src)[QN], int mb) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        float4 v = src[m][q];
        if (a_act != EAMD_ACT_NONE) {
          v.x = eamd_act(v.x, a_act); v.y = eamd_act(v.y, a_act); v.z = eamd_act(v.z, a_act); v.w = eamd_act(v.w, a_act);
        }
#pragma unroll
        for (int c = 0; c < CB; ++c)
          acc[c][mb + m] = fmaf(v.x, w4[c][q].x, fmaf(v.y, w4[c][q].y, fmaf(v.z, w4[c][q].z, fmaf(v.w, w4[c][q].w, acc[c][mb + m]))));
      }
    }
  };
  load8(xa, 0);
  if constexpr (BOTH) {
    if (M > 8) load8(xb, 8);
    fma8(xa, 0);
    if (M > 8) fma8(xb, 8);
  } else {
    fma8(xa, 0);
    if (M > 8) { load8(xa, 8); fma8(xa, 8); }
  }
  float mine = 0.f;
#pragma unroll
  for (int c = 0; c < CB; ++c) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < M) {
        const float r = wave_sum(acc[c][m]);
        if (lane == c * 16 + m) mine = r;
      }
    }
  }
  if (lane < CB * 16) part[w][lane] = mine;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < CB * 16) {
    const int c = t >> 4, m = t & 15, n = c0 + c;
    if (m < M && n < N) {
      float v = ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t])) + (bias ? bias[n] : 0.f);
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = eamd_swish(v);
      v *= alpha;
      if (R) v += R[(long)m * ldr + n];
      y[(long)m * N + n] = v;
    }
  }
}
}  // namespace

// Bookkeeping of a beam step after the selection, one thread per surviving slot (reference: beam_search.py:177-203 post_process /
// batch_beam_search.py:249-284 on the host; here the state stays on the device): which hypothesis and token a winner is, the
// per-scorer scores carried along, the prefix copied and extended, the length cap / <eos> test, the next step's running score
// (-inf for ended or empty slots) and the row of the step log the host reads every few steps.  ~30 element-wise torch launches.
namespace {
struct BeamFinishArgs {
  const float* logp[4];
  const float* top_s; const int64_t* top_i; const int64_t* maxlen; const float* sc_in; const float* c_local; const int64_t* ids;
  const int64_t* yseq_in;
  float* sc_out; int64_t* yseq_out; float* hyp_out; int64_t* hyp_i; int64_t* tok_i; int64_t* pos; float* rec;
  int64_t ldc;
  int n, beam, V, W, L, step, eos, ns, nf, full_mode, ncand;
};
__global__ __launch_bounds__(64) void beam_finish_kernel(const BeamFinishArgs a) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.n) return;
  const int u = s / a.beam;
  long ti = a.top_i[s];
  float ts = a.top_s[s];
  // a selection index outside the utterance's beam x V continuations is never turned into an address: the slot dies (-inf)
  if (ti < 0 || ti >= (long)a.beam * a.V) { ti = 0; ts = -INFINITY; }
  const long h = (long)u * a.beam + ti / a.V;
  const long tok = ti % a.V;
  a.hyp_i[s] = h; a.tok_i[s] = tok;
  const int RW = 3 + a.ns + a.W;
  float* rec = a.rec + (long)s * RW;
  rec[0] = (float)a.step; rec[1] = ts; rec[2] = (float)tok;
  for (int j = 0; j < a.nf; ++j) {
    const float v = a.sc_in[(long)j * a.n + h] + a.logp[j][h * a.V + tok];
    a.sc_out[(long)j * a.n + s] = v;
    rec[3 + j] = v;
  }
  long p = tok;                                          // position of the token among the hypothesis's candidates
  if (a.ids) {
    p = 0;                                               // (argmax of an all-false row, as the tensor code had it)
    for (int q = 0; q < a.ncand; ++q)
      if (a.ids[h * a.ncand + q] == tok) { p = q; break; }
  }
  a.pos[s] = p;
  if (a.ns > a.nf) {
    const float v = a.sc_in[(long)a.nf * a.n + h] + a.c_local[h * a.ldc + (a.full_mode ? tok : p)];
    a.sc_out[(long)a.nf * a.n + s] = v;
    rec[3 + a.nf] = v;
  }
  const int64_t* yi = a.yseq_in + h * a.W;
  int64_t* yo = a.yseq_out + (long)s * a.W;
  for (int w = 0; w < a.W; ++w) {
    const int64_t t = w == a.L ? tok : yi[w];
    yo[w] = t;
    rec[3 + a.ns + w] = (float)t;
  }
  const bool finite = isfinite(ts);
  const bool at_cap = a.maxlen[u] <= a.step + 1;
  const bool done = finite && (tok == a.eos || at_cap);
  a.hyp_out[s] = (done || !finite) ? -INFINITY : ts;
}
}  // namespace

// The same product for a few hundred rows (a batched beam search steps B x beam hypotheses: M = 320).  As 64-wide tiles these
// launches are 5 x 4 workgroups walking K alone (15.6 us at K = 256, 60 us at K = 2048).  Here every WAVE owns one 16 x 16 output
// tile and feeds v_mfma_f32_16x16x4_f32 straight from global memory: a lane's 16 bytes of an input row and of a weight row are
// four k-steps of both operands (k-group g of step e holds k = 4 g + e: every k once, both operands alike); no LDS, no barrier,
// (M / 16) x (N / 16) independent waves.  Exact fp32 products; the summation order over k differs from the tile kernels'.
namespace {
__global__ __launch_bounds__(256) void linear_mfma16_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, const float* __restrict__ R,
                                                                float* __restrict__ y, int M, int N, int K, int a_act, int act,
                                                                float alpha, long ldx, long ldr) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  const int m0 = blockIdx.y * 16;
  if (n0 >= N) return;
  const float* xr = x + (long)min(m0 + fr, M - 1) * ldx + fq * 4;        // clamped rows / columns are never stored
  const float* wr = W + (long)min(n0 + fr, N - 1) * K + fq * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;                                                    // 16-k chunks requested together
  for (int k0 = 0; k0 < K; k0 += 16 * U) {
    float4 xa[U], wb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + 16 * u, K - 16);                             // (K % 16 == 0: checked on the host)
      xa[u] = *reinterpret_cast<const float4*>(xr + k);
      wb[u] = *reinterpret_cast<const float4*>(wr + k);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + 16 * u < K) {
        float4 v = xa[u];
        if (a_act != EAMD_ACT_NONE) {
          v.x = eamd_act(v.x, a_act); v.y = eamd_act(v.y, a_act); v.z = eamd_act(v.z, a_act); v.w = eamd_act(v.w, a_act);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, wb[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, wb[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, wb[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, wb[u].w, acc, 0, 0, 0);
      }
    }
  }
  const int n = n0 + fr;
  if (n < N) {
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + fq * 4 + r;
      if (m < M) {
        float v = acc[r] + b;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = eamd_swish(v);
        v *= alpha;
        if (R) v += R[(long)m * ldr + n];
        y[(long)m * N + n] = v;
      }
    }
  }
}
}  // namespace

// ... and for a LONG reduction (the feed-forward down-product of a batched beam step: 320 x 256 x K = 2048): one 16 x 16 output tile
// per WORKGROUP, its four waves split K (each walks K / 4 straight from global memory as above), the four partial tiles meet in LDS
// and are added in a fixed order.  As one wave per tile that product took 33 us (every wave alone on a 2048-long chain), as
// split-K 64 x 64 tiles with atomics 21 us + a 5 us zero fill.
namespace {
__global__ __launch_bounds__(256) void linear_mfma16_ksplit_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                       const float* __restrict__ bias, const float* __restrict__ R,
                                                                       float* __restrict__ y, int M, int N, int K, int a_act, int act,
                                                                       float alpha, long ldx, long ldr) {
  __shared__ float part[3][64][4];
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int kq = ((K / 4 + 15) / 16) * 16;                                 // a wave's share of K (a multiple of 16)
  const int kb = w * kq, ke = min(K, kb + kq);
  const float* xr = x + (long)min(m0 + fr, M - 1) * ldx + fq * 4;          // clamped rows / columns are never stored
  const float* wr = W + (long)min(n0 + fr, N - 1) * K + fq * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;                                                      // 16-k chunks requested together
  for (int k0 = kb; k0 < ke; k0 += 16 * U) {
    float4 xa[U], wb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + 16 * u, K - 16);                               // (K % 16 == 0: checked on the host)
      xa[u] = *reinterpret_cast<const float4*>(xr + k);
      wb[u] = *reinterpret_cast<const float4*>(wr + k);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k0 + 16 * u < ke) {
        float4 v = xa[u];
        if (a_act != EAMD_ACT_NONE) {
          v.x = eamd_act(v.x, a_act); v.y = eamd_act(v.y, a_act); v.z = eamd_act(v.z, a_act); v.w = eamd_act(v.w, a_act);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, wb[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, wb[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, wb[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, wb[u].w, acc, 0, 0, 0);
      }
    }
  }
  if (w > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w - 1][lane][r] = acc[r];
  }
  __syncthreads();
  if (w != 0) return;
  const int n = n0 + fr;
  if (n < N) {
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + fq * 4 + r;
      if (m < M) {
        float v = ((acc[r] + part[0][lane][r]) + (part[1][lane][r] + part[2][lane][r])) + b;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = eamd_swish(v);
        v *= alpha;
        if (R) v += R[(long)m * ldr + n];
        y[(long)m * N + n] = v;
      }
    }
  }
}
}  // namespace

// k largest of every row, sorted: value descending, equal values by ascending index (a total order: the selection is the same
// whatever the grid or the replay).  NaN counts as -inf, -0 as +0.  One workgroup per row; an element is the 64-bit key
// (order-preserving bits of the value, ~index): "ranks before" is one unsigned compare.
// For the beam search's selections (beam_search.py:143-176: top-k over V and over beam x V per utterance, k <= 1.5 beam):
// n = 5000 .. 50000, k = 10 .. 15.  Rows of up to 6144 elements stay in registers (24 per thread) and are cut down first: every
// element of the answer is >= the k-th largest of the 256 per-thread maxima (those k maxima alone are k elements that large), so
// the elements above that threshold - a few dozen in general - are gathered in LDS and ranked by counting (35 us as k rounds of
// "largest element below the previous winner" over the whole row, which remains the path for longer rows and for rows with more
// than TOPK_CAP elements at the threshold).
// torch.topk's multi-block path for these sizes (6 launches + a sort) is also what faults under hipGraph replay on this ROCm.
namespace {
constexpr int TOPK_CAP = 1024;
__device__ __forceinline__ unsigned topk_bits(float v) {
  v = (v != v) ? -INFINITY : v;
  if (v == 0.f) v = 0.f;
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float topk_value(unsigned b) { return __uint_as_float((b & 0x80000000u) ? (b ^ 0x80000000u) : ~b); }
__device__ __forceinline__ unsigned long long topk_key(unsigned bits, unsigned i) { return ((unsigned long long)bits << 32) | (0xFFFFFFFFu - i); }

// FUSED: the row is the weighted sum of up to four rows (the scorers' log-probabilities of a beam step, weighted_sum_kernel's
// arithmetic: separately rounded products and sums), formed while it is read and written out as `pre` for the selection.
// extra >= 0: one more output column per row (row stride k + 1) = that token, or -1 when it is already among the k (the <eos> a
// "full"-mode partial scorer always scores besides the pre-beam: batch_beam_search.py:221-231 with scorers/ctc.py:82-96).
struct TopkSum { const float* l[4]; float w[4]; float* pre; int extra; };
template <bool FUSED>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, long ld, int n, int k,
                                                        float* __restrict__ vals, int64_t* __restrict__ idx, int32_t* __restrict__ idx32,
                                                        const TopkSum ws) {
  auto element = [&](long i) -> float {
    if constexpr (!FUSED) return x[(long)blockIdx.x * ld + i];
    const long e = (long)blockIdx.x * n + i;
    float v = ws.w[0] * ws.l[0][e];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      if (ws.l[j]) {
        float p = ws.w[j] * ws.l[j][e];
        asm volatile("" : "+v"(p));                    // (keeps hipcc from fusing a * b + c: weighted_sum_kernel)
        v += p;
      }
    }
    ws.pre[e] = v;
    return v;
  };
  __shared__ __attribute__((aligned(16))) unsigned tmax[256];
  __shared__ unsigned long long cand[TOPK_CAP];
  __shared__ unsigned long long sk[4];
  __shared__ unsigned long long wk;
  __shared__ unsigned tau;
  __shared__ int cnt;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ko = k + (ws.extra >= 0 ? 1 : 0);    // output row stride
  if (t == 0 && ws.extra >= 0) {
    idx[(long)blockIdx.x * ko + k] = ws.extra;
    if (idx32) idx32[(long)blockIdx.x * ko + k] = ws.extra;
    vals[(long)blockIdx.x * ko + k] = 0.f;
  }
  constexpr int RMAX = 24;                       // register-resident elements per thread (n <= 6144); longer rows re-read memory
  unsigned reg[RMAX];                            // order-preserving bits; 0 (below every real element) past the end of the row
  const bool inreg = n <= RMAX * 256;
  const float* xr = FUSED ? ws.pre + (long)blockIdx.x * n : x + (long)blockIdx.x * ld;      // what the re-reading rounds walk
  if (FUSED && !inreg) {                         // a long row: formed once, then read back like any other
    for (long i = t; i < n; i += 256) element(i);
    __syncthreads();
  }
  if (inreg) {
    unsigned tm = 0;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      const int i = t + 256 * q;
      reg[q] = i < n ? topk_bits(element(i)) : 0u;
      tm = max(tm, reg[q]);
    }
    tmax[t] = tm;
    if (t == 0) { tau = 0xFFFFFFFFu; cnt = 0; }
    __syncthreads();
    int above = 0;                               // thread maxima strictly above mine
#pragma unroll 8
    for (int j = 0; j < 256; j += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(&tmax[j]);
      above += (v.x > tm) + (v.y > tm) + (v.z > tm) + (v.w > tm);
    }
    if (above < k && tm != 0u) atomicMin(&tau, tm);
    __syncthreads();
    const unsigned th = tau;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      if (reg[q] >= th && reg[q] != 0u) {
        const int p = atomicAdd(&cnt, 1);
        if (p < TOPK_CAP) cand[p] = topk_key(reg[q], (unsigned)(t + 256 * q));
      }
    }
    __syncthreads();
    const int c = cnt;
    if (c <= TOPK_CAP) {
      for (int j = t; j < c; j += 256) {
        const unsigned long long my = cand[j];
        int r = 0;
        for (int q = 0; q < c; ++q) r += cand[q] > my;
        if (r < k) {
          const long o = (long)blockIdx.x * ko + r;
          const unsigned i = 0xFFFFFFFFu - (unsigned)my;
          vals[o] = topk_value((unsigned)(my >> 32));
          idx[o] = (int64_t)i;
          if (idx32) idx32[o] = (int32_t)i;
          if ((int)i == ws.extra) {              // (after the barriers above: ordered behind thread 0's store)
            idx[(long)blockIdx.x * ko + k] = -1;
            if (idx32) idx32[(long)blockIdx.x * ko + k] = -1;
          }
        }
      }
      return;
    }
    __syncthreads();
  }
  unsigned long long prev = 0xFFFFFFFFFFFFFFFFull;      // previous winner: every element ranks after it
  for (int r = 0; r < k; ++r) {
    unsigned long long best = 0;
    if (inreg) {
#pragma unroll
      for (int q = 0; q < RMAX; ++q) {
        const unsigned long long key = topk_key(reg[q], (unsigned)(t + 256 * q));
        if (reg[q] != 0u && key < prev && key > best) best = key;
      }
    } else {
      for (long i = t; i < n; i += 256) {
        const unsigned long long key = topk_key(topk_bits(xr[i]), (unsigned)i);
        if (key < prev && key > best) best = key;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const unsigned long long o = __shfl_xor(best, m);
      best = o > best ? o : best;
    }
    if (lane == 0) sk[w] = best;
    __syncthreads();
    if (t == 0) {
      unsigned long long f = sk[0];
#pragma unroll
      for (int q = 1; q < 4; ++q) f = sk[q] > f ? sk[q] : f;
      wk = f;
      const long o = (long)blockIdx.x * ko + r;
      const unsigned i = 0xFFFFFFFFu - (unsigned)f;
      vals[o] = topk_value((unsigned)(f >> 32));
      idx[o] = (int64_t)i;
      if (idx32) idx32[o] = (int32_t)i;
      if ((int)i == ws.extra) {
        idx[(long)blockIdx.x * ko + k] = -1;
        if (idx32) idx32[(long)blockIdx.x * ko + k] = -1;
      }
    }
    __syncthreads();
    prev = wk;
  }
}
}  // namespace

extern "C" {

int eamd_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean,
                       float* rstd, int rows, int D, float eps, void* stream) {
  if (!x || !gamma || !beta || (!y && !y_bf16) || !mean || !rstd || rows <= 0 || D <= 0) return EAMD_EINVAL;
  unsigned short* y16 = (unsigned short*)y_bf16;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
  const bool al = ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
                  ((uintptr_t)beta % 16 == 0) && ((uintptr_t)y16 % 8 == 0);
  if (al && D == 256) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<1>, grid, block, 0, s, x, gamma, beta, y, y16, mean, rstd, rows, eps);
  else if (al && D == 512) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<2>, grid, block, 0, s, x, gamma, beta, y, y16, mean, rstd, rows, eps);
  else hipLaunchKernelGGL(layernorm_fwd_kernel, grid, block, 0, s, x, gamma, beta, y, y16, mean, rstd, rows, D, eps);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

static void ln_bwd_grid(int rows, int* nblk, int* rpb) {
  static const int per = [] { const char* e = getenv("EAMD_LNB_RPB"); return e ? atoi(e) : 16; }();
  int n = min(4096, (rows + per - 1) / per);
  *rpb = (rows + n - 1) / n;
  *nblk = (rows + *rpb - 1) / *rpb;
}

int64_t eamd_layernorm_bwd_workspace(int rows, int D) {
  if (rows <= 0 || D <= 0) return 0;
  int nblk, rpb;
  ln_bwd_grid(rows, &nblk, &rpb);
  return (int64_t)nblk * 2 * D;
}

static int layernorm_bwd_impl(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace, int rows, int D,
                              void* drop16, float drop_p, const uint64_t* drop_step, uint64_t drop_salt, void* stream,
                              int drop_f32 = 0) {
  const bool deferred = !dgamma && !dbeta && workspace;      // partials stay in `workspace` (eamd_layernorm_bwd_reduce)
  if (!dy || !x || !gamma || !mean || !rstd || !dx || (!deferred && (!dgamma || !dbeta)) || rows <= 0 || D <= 0)
    return EAMD_EINVAL;
  if (D > 1024) return EAMD_EUNSUPPORTED;
  if (drop16 && (drop_p < 0.f || drop_p >= 1.f || !drop_step || ((uintptr_t)drop16 & (drop_f32 ? 15 : 7)))) return EAMD_EINVAL;
  int nblk, rpb;
  ln_bwd_grid(rows, &nblk, &rpb);
  static const int ws_min = [] { const char* e = getenv("EAMD_LNB_WS_MIN"); return e ? atoi(e) : 32; }();
  float* ws = (deferred || nblk >= ws_min) ? workspace : nullptr;   // few blocks: direct atomics are cheaper than a second launch
  static const int nthr = [] { const char* e = getenv("EAMD_LNB_THREADS"); return e ? atoi(e) : 512; }();      // config 2, in the step: 256 threads 9.7 us, 512 -> 8.8 us, 1024 (32-row blocks) 9.0 us
  const bool al = (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0;
  hipStream_t s = (hipStream_t)stream;
  size_t sm = (size_t)(nthr / 64) * 2 * D * sizeof(float);
  unsigned short* d16 = (unsigned short*)drop16;
  const unsigned long long* dst = (const unsigned long long*)drop_step;
  if (al && D == 256)
    hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(nblk), dim3(nthr), sm, s, dy, x, gamma, mean, rstd, dres, dx,
                       dgamma, dbeta, ws, rows, D, rpb, d16, drop_p, dst, (unsigned long long)drop_salt, drop_f32);
  else if (al && D == 512)
    hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(nblk), dim3(nthr), sm, s, dy, x, gamma, mean, rstd, dres, dx,
                       dgamma, dbeta, ws, rows, D, rpb, d16, drop_p, dst, (unsigned long long)drop_salt, drop_f32);
  else {
    if (drop16) return EAMD_EUNSUPPORTED;       // the fused dropout output exists in the vector form only
    hipLaunchKernelGGL(layernorm_bwd_kernel<0>, dim3(nblk), dim3(nthr), sm, s, dy, x, gamma, mean, rstd, dres, dx,
                       dgamma, dbeta, ws, rows, D, rpb, d16, drop_p, dst, (unsigned long long)drop_salt, drop_f32);
  }
  EAMD_LAUNCH_CHECK();
  if (ws && !deferred) {
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * D + 63) / 64, min(16, (nblk + 31) / 32)), dim3(256), 0,
                       s, ws, nblk, D, dgamma, dbeta);
    EAMD_LAUNCH_CHECK();
  }
  return EAMD_OK;
}

int eamd_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta,
                       float* workspace, int rows, int D, void* stream) {
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, workspace, rows, D, nullptr, 0.f, nullptr, 0,
                            stream);
}

int eamd_layernorm_bwd_reduce(const eamd_ln_reduce_job_t* jobs, int njobs, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return EAMD_EINVAL;
  for (int i = 0; i < njobs; ++i)
    if (!jobs[i].ws || !jobs[i].dgamma || !jobs[i].dbeta || jobs[i].nblk <= 0 || jobs[i].D <= 0 || jobs[i].D > 1024)
      return EAMD_EINVAL;
  for (int j0 = 0; j0 < njobs; j0 += LN_JOBS) {
    const int n = min(LN_JOBS, njobs - j0);
    LnJobTable tab;
    int dmax = 0;
    for (int i = 0; i < LN_JOBS; ++i) {
      tab.j[i] = jobs[j0 + (i < n ? i : 0)];
      if (i < n) dmax = max(dmax, tab.j[i].D);
    }
    hipLaunchKernelGGL(layernorm_bwd_reduce_batched_kernel, dim3((2 * dmax + 63) / 64, n), dim3(256), 0,
                       (hipStream_t)stream, tab);
    EAMD_LAUNCH_CHECK();
  }
  return EAMD_OK;
}

int eamd_layernorm_bwd_drop(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                            const float* dres, float* dx, void* dx_drop_bf16, float drop_p, const uint64_t* step_dev,
                            uint64_t salt, float* dgamma, float* dbeta, float* workspace, int rows, int D, void* stream) {
  if (!dx_drop_bf16) return EAMD_EINVAL;
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, workspace, rows, D, dx_drop_bf16, drop_p,
                            step_dev, salt, stream);
}

int eamd_layernorm_bwd_drop_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                const float* dres, float* dx, float* dx_drop, float drop_p, const uint64_t* step_dev,
                                uint64_t salt, float* dgamma, float* dbeta, float* workspace, int rows, int D, void* stream) {
  if (!dx_drop) return EAMD_EINVAL;
  return layernorm_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, workspace, rows, D, dx_drop, drop_p,
                            step_dev, salt, stream, 1);
}

int eamd_softmax_fwd(const float* ac, const float* bd, const unsigned char* mask, int64_t mask_bstride,
                     int64_t mask_qstride, float* P, void* P_bf16, int nblocks, int B, int T1, int T2, int64_t ld,
                     float scale, void* stream) {
  if (!ac || (!P && !P_bf16) || nblocks <= 0 || B <= 0 || T1 <= 0 || T2 <= 0 || ld < T2) return EAMD_EINVAL;
  if ((size_t)T2 * 16 > 160 * 1024) return EAMD_EUNSUPPORTED;
  long rows = (long)nblocks * T1;
  size_t smem = (size_t)4 * T2 * sizeof(float);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&softmax_fwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  // bf16 probabilities in their own buffer, rows of <= 1024 columns: register-resident vector form
  const bool vec = P_bf16 && !P && ld % 4 == 0 && ld <= 1024 && (((uintptr_t)ac | (uintptr_t)P_bf16) & 15) == 0;
  if (vec) {
    const dim3 g((rows + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned short* p16 = (unsigned short*)P_bf16;
    const long mbs = (long)mask_bstride, mqs = (long)mask_qstride, l = (long)ld;
    if (ld <= 256) hipLaunchKernelGGL(softmax_fwd_vec_kernel<1>, g, blk, 0, s, ac, bd, mask, mbs, mqs, p16, nblocks, B, T1, T2, l, scale);
    else if (ld <= 512) hipLaunchKernelGGL(softmax_fwd_vec_kernel<2>, g, blk, 0, s, ac, bd, mask, mbs, mqs, p16, nblocks, B, T1, T2, l, scale);
    else hipLaunchKernelGGL(softmax_fwd_vec_kernel<4>, g, blk, 0, s, ac, bd, mask, mbs, mqs, p16, nblocks, B, T1, T2, l, scale);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), smem, (hipStream_t)stream, ac, bd,
                     mask, (long)mask_bstride, (long)mask_qstride, P, (unsigned short*)P_bf16, nblocks, B, T1, T2,
                     (long)ld, scale);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_softmax_bwd(const float* P, const void* P_bf16, float* dP, float* dbd, void* dS_bf16, void* dbd_bf16,
                     int nblocks, int T1, int T2, int64_t ld, float scale, void* stream) {
  if ((!P && !P_bf16) || !dP || nblocks <= 0 || T1 <= 0 || T2 <= 0 || ld < T2) return EAMD_EINVAL;
  long rows = (long)nblocks * T1;
  const bool vec = P_bf16 && !P && dS_bf16 && !dbd && ld % 4 == 0 && ld <= 1024 &&
                   (((uintptr_t)dP | (uintptr_t)P_bf16 | (uintptr_t)dS_bf16) & 15) == 0;
  if (vec) {
    const dim3 g((rows + 3) / 4), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* p16 = (const unsigned short*)P_bf16;
    unsigned short* ds = (unsigned short*)dS_bf16; unsigned short* db = (unsigned short*)dbd_bf16;
    const long l = (long)ld;
    if (ld <= 256) hipLaunchKernelGGL(softmax_bwd_vec_kernel<1>, g, blk, 0, s, p16, dP, ds, db, nblocks, T1, T2, l, scale);
    else if (ld <= 512) hipLaunchKernelGGL(softmax_bwd_vec_kernel<2>, g, blk, 0, s, p16, dP, ds, db, nblocks, T1, T2, l, scale);
    else hipLaunchKernelGGL(softmax_bwd_vec_kernel<4>, g, blk, 0, s, p16, dP, ds, db, nblocks, T1, T2, l, scale);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, P,
                     (const unsigned short*)P_bf16, dP, dbd, (unsigned short*)dS_bf16, (unsigned short*)dbd_bf16,
                     nblocks, T1, T2, (long)ld, scale);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_lsm_loss(const float* logits, const int64_t* target, float* loss_rows, float* correct_rows,
                  float* grad, int rows, int V, int ignore_id, float smoothing, float inv_denom,
                  void* stream) {
  if (!logits || !target || !loss_rows || !correct_rows || rows <= 0 || V <= 1) return EAMD_EINVAL;
  const bool vec = V % 4 == 0 && V <= 8192 && (((uintptr_t)logits | (uintptr_t)grad) & 15) == 0;
  if (vec && V <= 5120)
    hipLaunchKernelGGL(lsm_loss_reg_kernel<5>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                       (const long long*)target, loss_rows, correct_rows, grad, V, ignore_id, smoothing, inv_denom);
  else if (vec)
    hipLaunchKernelGGL(lsm_loss_reg_kernel<8>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                       (const long long*)target, loss_rows, correct_rows, grad, V, ignore_id, smoothing, inv_denom);
  else
    hipLaunchKernelGGL(lsm_loss_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits,
                       (const long long*)target, loss_rows, correct_rows, grad, V, ignore_id, smoothing,
                       inv_denom);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_argmax_rows(const float* x, int64_t ld, int32_t* out, int rows, int V, void* stream) {
  if (!x || !out || rows <= 0 || V <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (long)ld, out, V);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_reduce_sum(const float* in, int64_t n, float* out, float scale, void* stream) {
  if (!in || !out || n < 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, (long)n, out, scale);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_beam_finish(const float* top_s, const int64_t* top_i, int n, int beam, int V, int W, int L, int step, int eos,
                     const int64_t* maxlen, int ns, int nf, const float* sc_in, const float* const* logps, const float* c_local,
                     int64_t ldc, int full_mode, const int64_t* ids, int ncand, const int64_t* yseq_in, float* sc_out,
                     int64_t* yseq_out, float* hyp_out, int64_t* hyp_i, int64_t* tok_i, int64_t* pos, float* rec, void* stream) {
  if (!top_s || !top_i || !maxlen || !sc_in || !yseq_in || !sc_out || !yseq_out || !hyp_out || !hyp_i || !tok_i || !pos || !rec)
    return EAMD_EINVAL;
  if (n <= 0 || beam <= 0 || n % beam != 0 || V <= 0 || W <= 1 || L < 1 || L >= W || ns < 0 || nf < 0 || nf > 4 || ns < nf || ns > nf + 1)
    return EAMD_EINVAL;
  if ((nf > 0 && !logps) || (ns > nf && !c_local) || (ids && ncand <= 0)) return EAMD_EINVAL;
  BeamFinishArgs a;
  for (int j = 0; j < 4; ++j) a.logp[j] = j < nf ? logps[j] : nullptr;
  for (int j = 0; j < nf; ++j) if (!a.logp[j]) return EAMD_EINVAL;
  a.top_s = top_s; a.top_i = top_i; a.n = n; a.beam = beam; a.V = V; a.W = W; a.L = L; a.step = step; a.eos = eos; a.maxlen = maxlen;
  a.ns = ns; a.nf = nf; a.sc_in = sc_in; a.c_local = c_local; a.ldc = ldc; a.full_mode = full_mode; a.ids = ids; a.ncand = ncand;
  a.yseq_in = yseq_in; a.sc_out = sc_out; a.yseq_out = yseq_out; a.hyp_out = hyp_out; a.hyp_i = hyp_i; a.tok_i = tok_i; a.pos = pos;
  a.rec = rec;
  hipLaunchKernelGGL(beam_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_linear_rows_f32(const float* x, const float* W, const float* bias, const float* R, float* y, int M, int N, int K,
                         int a_act, int act, float alpha, int64_t ldx, int64_t ldr, void* stream) {
  if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0 || a_act < 0 || act < 0 || ldx < 0 || ldr < 0) return EAMD_EINVAL;
  if (ldx == 0) ldx = K;
  if (ldr == 0) ldr = N;
  if (ldx < K || ldr < N) return EAMD_EINVAL;
  if (M > 1024 || K % 4 != 0 || ldx % 4 != 0 || a_act > EAMD_ACT_SWISH || act > 2) return EAMD_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)W) & 15) return EAMD_EUNSUPPORTED;
  static const int rows_mfma = getenv("EAMD_ROWS_MFMA") ? atoi(getenv("EAMD_ROWS_MFMA")) : 0;      // A/B knob: the 16 x 16 tiles for M <= 16 too
  if (M > 16 || (rows_mfma && K % 16 == 0)) {
    if (K % 16 != 0) return EAMD_EUNSUPPORTED;
    if (K >= 1024 && M > 16) {                               // a long reduction: the waves of a workgroup split K
      hipLaunchKernelGGL(linear_mfma16_ksplit_f32_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, W,
                         bias, R, y, M, N, K, a_act, act, alpha, (long)ldx, (long)ldr);
      EAMD_LAUNCH_CHECK();
      return EAMD_OK;
    }
    hipLaunchKernelGGL(linear_mfma16_f32_kernel, dim3((N + 63) / 64, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, W, bias,
                       R, y, M, N, K, a_act, act, alpha, (long)ldx, (long)ldr);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  static const int rows_wide = getenv("EAMD_ROWS_WIDE") ? atoi(getenv("EAMD_ROWS_WIDE")) : 1;      // A/B knob: 0 = the staged kernel for every K
  if (rows_wide && K >= 1024 && K <= 4096) {               // the waves of a workgroup split K
    constexpr int CB = 2;
    const dim3 grid((N + CB - 1) / CB);
#define EAMD_ROWS_WIDE_(QN) hipLaunchKernelGGL((linear_rows_wide_f32_kernel<QN, CB>), grid, dim3(256), 0, (hipStream_t)stream, x, W, bias, R, y, M, \
                                               N, K, a_act, act, alpha, (long)ldx, (long)ldr)
    if (K <= 1024) EAMD_ROWS_WIDE_(1);
    else if (K <= 2048) EAMD_ROWS_WIDE_(2);
    else EAMD_ROWS_WIDE_(4);
#undef EAMD_ROWS_WIDE_
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(linear_rows_f32_kernel, dim3((N + 3) / 4), dim3(256), (size_t)M * (K < ROWS_KC ? K : ROWS_KC) * sizeof(float),
                     (hipStream_t)stream, x, W, bias, R, y, M, N, K, a_act, act, alpha, (long)ldx, (long)ldr);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_topk_rows(const float* x, int64_t ld, int rows, int n, int k, float* vals, int64_t* idx, void* stream) {
  return eamd_topk_rows_i32(x, ld, rows, n, k, vals, idx, nullptr, stream);
}

int eamd_weighted_topk_rows(const float* const* logps, const float* weights, int nf, int rows, int n, int k, int extra, float* pre,
                            float* vals, int64_t* idx, int32_t* idx32, void* stream) {
  if (!logps || !weights || !pre || !vals || !idx || nf < 1 || nf > 4 || rows <= 0 || n <= 0 || k <= 0 || k > n || extra >= n) return EAMD_EINVAL;
  if (k > 64) return EAMD_EUNSUPPORTED;
  TopkSum ws;
  ws.extra = extra < 0 ? -1 : extra;
  for (int j = 0; j < 4; ++j) { ws.l[j] = j < nf ? logps[j] : nullptr; ws.w[j] = j < nf ? weights[j] : 0.f; }
  for (int j = 0; j < nf; ++j) if (!ws.l[j]) return EAMD_EINVAL;
  ws.pre = pre;
  hipLaunchKernelGGL(topk_rows_kernel<true>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (long)n, n, k, vals, idx,
                     idx32, ws);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_topk_rows_i32(const float* x, int64_t ld, int rows, int n, int k, float* vals, int64_t* idx, int32_t* idx32, void* stream) {
  if (!x || !vals || !idx || rows <= 0 || n <= 0 || k <= 0 || k > n || ld < n) return EAMD_EINVAL;
  if (k > 64) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(topk_rows_kernel<false>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, n, k, vals, idx, idx32, TopkSum{{nullptr, nullptr, nullptr, nullptr}, {0.f, 0.f, 0.f, 0.f}, nullptr, -1});
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_log_softmax_rows(const float* x, float* y, int rows, int V, void* stream) {
  if (!x || !y || rows <= 0 || V <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, y, V);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
