// Element-wise / small reduction kernels (HBM-bound, float4-vectorised, grid-stride capped at
// 2048 blocks): GLU, Swish, bias broadcasts, axpby, column sums, embedding + positional encoding,
// weight-layout permutations for the implicit-GEMM convolutions, dropout.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

inline int grid_for(long n_threads_needed) {
  long b = (n_threads_needed + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// out = a*x + b*y (y optional)
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                             long n, long n4, float a, float b) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 xv = reinterpret_cast<const float4*>(x)[i], o;
    if (y) {
      float4 yv = reinterpret_cast<const float4*>(y)[i];
      o.x = a * xv.x + b * yv.x; o.y = a * xv.y + b * yv.y; o.z = a * xv.z + b * yv.z; o.w = a * xv.w + b * yv.w;
    } else {
      o.x = a * xv.x; o.y = a * xv.y; o.z = a * xv.z; o.w = a * xv.w;
    }
    reinterpret_cast<float4*>(out)[i] = o;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// fp32 -> bf16 (round-to-nearest-even), 8 elements per thread
__global__ void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n, long n8) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    uint4 o;
    o.x = eamd_f2bf(a.x) | ((unsigned)eamd_f2bf(a.y) << 16);
    o.y = eamd_f2bf(a.z) | ((unsigned)eamd_f2bf(a.w) << 16);
    o.z = eamd_f2bf(b.x) | ((unsigned)eamd_f2bf(b.y) << 16);
    o.w = eamd_f2bf(b.z) | ((unsigned)eamd_f2bf(b.w) << 16);
    reinterpret_cast<uint4*>(y)[i] = o;
  }
  for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = eamd_f2bf(x[i]);
}

// out = scale_dev[0] * x  (scale read from device memory: keeps upstream loss scaling sync-free)
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ scale_dev,
                                 float* __restrict__ out, long n, float extra) {
  const float s = scale_dev[0] * extra;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = s * x[i];
}

// Activation forward / backward on flat arrays.
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = eamd_act(x[i], act);
}
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                               long n, int act) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    dx[i] = dy[i] * eamd_dact(x[i], act);
  }
}

// y[r, :] = keep[r] ? x[r, :] : 0   (masked_fill of padded frames, rnn/encoders.py:323-325)
__global__ void mask_rows_kernel(const float* __restrict__ x, const unsigned char* __restrict__ keep,
                                 float* __restrict__ y, long rows, int D) {
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = keep[i / D] ? x[i] : 0.f;
}

// GLU over the channel dim of a [rows, 2C] matrix: y[r,c] = x[r,c] * sigmoid(x[r,C+c]).
// reference: conformer/convolution.py:72 (glu(dim=1) on (B,2C,T) == per-row halves in (B,T,2C)).
__global__ void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C) {
  const long n = rows * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / C; int c = i % C;
    float a = x[r * 2 * C + c], g = x[r * 2 * C + C + c];
    y[i] = a * eamd_sigmoid(g);
  }
}
__global__ void glu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                               unsigned short* __restrict__ dx16, long rows, int C) {
  const long n = rows * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / C; int c = i % C;
    float a = x[r * 2 * C + c], g = x[r * 2 * C + C + c];
    float s = eamd_sigmoid(g), d = dy[i];
    const float g1 = d * s, g2 = d * a * s * (1.f - s);
    if (dx16) { dx16[r * 2 * C + c] = eamd_f2bf(g1); dx16[r * 2 * C + C + c] = eamd_f2bf(g2); }
    else { dx[r * 2 * C + c] = g1; dx[r * 2 * C + C + c] = g2; }
  }
}

// qu = q + u, qv = q + v with u,v broadcast over rows ([D] each).
// reference: transformer/attention.py:186-190 (q_with_bias_u / q_with_bias_v).
__global__ void add_bias2_kernel(const float* __restrict__ q, const float* __restrict__ u,
                                 const float* __restrict__ v, float* __restrict__ qu, float* __restrict__ qv,
                                 long rows, int D, int bf16, long ldq) {
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  const unsigned short* q16 = reinterpret_cast<const unsigned short*>(q);
  unsigned short* qu16 = reinterpret_cast<unsigned short*>(qu);
  unsigned short* qv16 = reinterpret_cast<unsigned short*>(qv);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int d = i % D;
    const long qi = (i / D) * ldq + d;        // q may be a column block of a wider matrix (fused QKV projection)
    if (bf16) {
      float x = __uint_as_float(((unsigned)q16[qi]) << 16);
      qu16[i] = eamd_f2bf(x + u[d]);
      qv16[i] = eamd_f2bf(x + v[d]);
    } else {
      float x = q[qi];
      qu[i] = x + u[d];
      qv[i] = x + v[d];
    }
  }
}

// out_bf16 = a + b  (fp32 inputs): the query gradient dq = dq_u + dq_v feeding the bf16 GEMMs
__global__ void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b, unsigned short* __restrict__ o,
                                long n, int cols, long ld_out) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    o[(i / cols) * ld_out + i % cols] = eamd_f2bf(a[i] + (b ? b[i] : 0.f));
}

// ---- two adjacent columns per thread for the bf16-storage variants (dword stores; all widths even) ----------------
__device__ __forceinline__ unsigned eamd_pack2(float a, float b) {
  return (unsigned)eamd_f2bf(a) | ((unsigned)eamd_f2bf(b) << 16);
}
__global__ void glu_bwd_x2_kernel(const float* __restrict__ dy, const float* __restrict__ x, unsigned int* __restrict__ dx32,
                                  long rows, int C) {
  const int CW = C / 2;
  const long n = rows * CW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long r = i / CW; const int c = 2 * (int)(i % CW);
    const float2 a = *reinterpret_cast<const float2*>(x + r * 2 * C + c);
    const float2 g = *reinterpret_cast<const float2*>(x + r * 2 * C + C + c);
    const float2 d = *reinterpret_cast<const float2*>(dy + r * C + c);
    const float s0 = eamd_sigmoid(g.x), s1 = eamd_sigmoid(g.y);
    dx32[(r * 2 * C + c) >> 1] = eamd_pack2(d.x * s0, d.y * s1);
    dx32[(r * 2 * C + C + c) >> 1] = eamd_pack2(d.x * a.x * s0 * (1.f - s0), d.y * a.y * s1 * (1.f - s1));
  }
}
__global__ void add_bias2_x2_kernel(const unsigned int* __restrict__ q32, const float* __restrict__ u,
                                    const float* __restrict__ v, unsigned int* __restrict__ qu32,
                                    unsigned int* __restrict__ qv32, long rows, int D, long ldq) {
  const int DW = D / 2;
  const long n = rows * DW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int d = 2 * (int)(i % DW);
    const unsigned w = q32[((i / DW) * ldq + d) >> 1];
    const float x0 = __uint_as_float(w << 16), x1 = __uint_as_float(w & 0xffff0000u);
    qu32[i] = eamd_pack2(x0 + u[d], x1 + u[d + 1]);
    qv32[i] = eamd_pack2(x0 + v[d], x1 + v[d + 1]);
  }
}
__global__ void add_cast_x2_kernel(const float* __restrict__ a, const float* __restrict__ b, unsigned int* __restrict__ o32,
                                   long n2, int cols, long ld_out) {
  const int CW = cols / 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const long r = i / CW; const int c = 2 * (int)(i % CW);
    const float2 av = *reinterpret_cast<const float2*>(a + r * cols + c);
    float2 bv = make_float2(0.f, 0.f);
    if (b) bv = *reinterpret_cast<const float2*>(b + r * cols + c);
    o32[(r * ld_out + c) >> 1] = eamd_pack2(av.x + bv.x, av.y + bv.y);
  }
}

// Column sums of a [rows, D] matrix, ADDED into out[D] (bias gradients).  Each block reduces a slab
// of rows with threads along columns (coalesced) and issues one atomic per column.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, float* __restrict__ out,
                                                     long rows, int D, int rows_per_block, float scale, int bf16) {
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(rows, r0 + (long)rows_per_block);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  // 8 rows per trip, loads issued together (a serial row walk is latency-bound: one HBM round trip per row)
  float s = 0.f;
  if (bf16) {
    const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
    long r = r0;
    for (; r + 8 <= r1; r += 8) {
      unsigned short v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x16[(r + u) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += __uint_as_float(((unsigned)v[u]) << 16);
    }
    for (; r < r1; ++r) s += __uint_as_float(((unsigned)x16[r * ld + c]) << 16);
  } else {
    long r = r0;
    for (; r + 8 <= r1; r += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * ld + c];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < r1; ++r) s += x[r * ld + c];
  }
  atomicAdd(&out[c], s * scale);
}

// bf16 input, two adjacent columns per thread read as one dword (D, ld even; 4-byte aligned base)
__global__ __launch_bounds__(256) void colsum_bf16x2_kernel(const unsigned int* __restrict__ x, long ldw,
                                                            float* __restrict__ out, long rows, int D,
                                                            int rows_per_block, float scale) {
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(rows, r0 + (long)rows_per_block);
  const int cw = blockIdx.x * blockDim.x + threadIdx.x;       // column pair
  if (2 * cw >= D) return;
  float s0 = 0.f, s1 = 0.f;
  long r = r0;
  for (; r + 8 <= r1; r += 8) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * ldw + cw];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += __uint_as_float(v[u] << 16); s1 += __uint_as_float(v[u] & 0xffff0000u); }
  }
  for (; r < r1; ++r) { const unsigned v = x[r * ldw + cw]; s0 += __uint_as_float(v << 16); s1 += __uint_as_float(v & 0xffff0000u); }
  atomicAdd(&out[2 * cw], s0 * scale);
  atomicAdd(&out[2 * cw + 1], s1 * scale);
}

// Embedding lookup * scale + absolute positional encoding.
// reference: decoder.py:251 (embed = Embedding + PositionalEncoding), embedding.py:80-91.
__global__ void embed_pe_kernel(const long long* __restrict__ tok, long ldt, const float* __restrict__ table,
                                const float* __restrict__ pe, float* __restrict__ out, long rows, int U, int D,
                                float scale, int pos_offset, const int* __restrict__ pos_dev) {
  if (pos_dev) pos_offset += pos_dev[0];                 // (the step index of a replayed beam step lives on the device)
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / D; int d = i % D;
    long id = tok[r * ldt];
    int pos = (int)(r % U) + pos_offset;
    out[i] = table[id * D + d] * scale + (pe ? pe[(long)pos * D + d] : 0.f);
  }
}
__global__ void embed_bwd_kernel(const long long* __restrict__ tok, const float* __restrict__ dout,
                                 float* __restrict__ dtable, long rows, int D, float scale, long pad_idx) {
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / D; int d = i % D;
    if (tok[r] == pad_idx) continue;      // nn.Embedding(padding_idx): that row never receives gradient
    atomicAdd(&dtable[tok[r] * D + d], dout[i] * scale);
  }
}

// x*scale + pe[t] for [B, T, D] (absolute positional encoding on float inputs).
__global__ void posenc_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ out,
                              long rows, int T, int D, float scale) {
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / D; int d = i % D;
    out[i] = x[i] * scale + pe[(r % T) * D + d];
  }
}

// ScaledPositionalEncoding (embedding.py:95-128): out = x * scale + alpha[0] * pe[t]; alpha is read on the device.
__global__ void posenc_scaled_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                     const float* __restrict__ alpha, float* __restrict__ out, long rows, int T, int D,
                                     float scale) {
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  const float a = alpha[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / D; int d = i % D;
    out[i] = x[i] * scale + a * pe[(r % T) * D + d];
  }
}
// dalpha += sum_i dout[i] * pe[t(i), d(i)]: per-thread partial sums, wave shuffle, one LDS slot per wave, one atomic
// per workgroup.
__global__ __launch_bounds__(256) void posenc_scaled_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pe,
                                                                float* __restrict__ dalpha, long rows, int T, int D) {
  __shared__ float red[4];
  const long n = rows * D;
  const long stride = (long)gridDim.x * blockDim.x;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    long r = i / D; int d = i % D;
    s += dout[i] * pe[(r % T) * D + d];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dalpha, red[0] + red[1] + red[2] + red[3]);
}

// out[r * ld + c] = a[r, c] (+ b[r, c]): fp32 twin of add_cast (a column block of a wider fp32 matrix as destination)
__global__ void add_block_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                     long n, int cols, long ld) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long r = i / cols; const int c = i % cols;
    out[r * ld + c] = a[i] + (b ? b[i] : 0.f);
  }
}

// Generic 4-D permutation copy with optional accumulate: dst[perm(idx)] (+)= src[idx].
// Used for conv weight layouts: Conv2d weight [Co][Ci][kh][kw] <-> tap-major GEMM operands.
__global__ void permute4_kernel(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int d2,
                                int d3, long s0, long s1, long s2, long s3, int accumulate) {
  const long n = (long)d0 * d1 * d2 * d3;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int i3 = i % d3; long t = i / d3;
    int i2 = t % d2; t /= d2;
    int i1 = t % d1; int i0 = t / d1;
    long o = i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3;
    if (accumulate) dst[o] += src[i]; else dst[o] = src[i];
  }
}

// Counter-based dropout: keep(i) = 16 hash bits of (step_counter, salt, i) >= p * 2^16, y = x * keep / P(keep).
// The step counter lives in DEVICE memory (advanced once per training step by eamd_rng_advance), so a
// captured hipGraph draws fresh masks on every replay; backward re-derives the same mask by calling the
// same kernel on the gradient.  Optional fused activation (FFN inner dropout: drop(act(z))).
// reference: torch.nn.Dropout call sites (conformer/encoder_layer.py:55, positionwise_feed_forward.py:27,
// embedding.py:56, ctc.py:85); RNG streams cannot match torch's generator - parity runs use p = 0.
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                               const unsigned long long* __restrict__ step, unsigned long long salt, int act,
                               int in_bf16, int out_bf16) {
  const unsigned thr = eamd_drop_thr16(p);
  const float inv = eamd_drop_inv(thr);
  const unsigned seed = eamd_drop_seed(step, salt);
  const long stride = (long)gridDim.x * blockDim.x;
  const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
  unsigned short* y16 = reinterpret_cast<unsigned short*>(y);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = in_bf16 ? __uint_as_float(((unsigned)x16[i]) << 16) : x[i];
    v = eamd_act(v, act);
    v = eamd_drop_keep(seed, (unsigned long long)i, thr) ? v * inv : 0.f;
    if (out_bf16) y16[i] = eamd_f2bf(v); else y[i] = v;
  }
}
// four elements per thread: float4 / 8-byte bf16 accesses (2-byte loads and stores run at a fraction of the dword
// rate on gfx950); the mask of element i is the same hash as in the scalar kernel
__global__ void dropout_vec4_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, float p,
                                    const unsigned long long* __restrict__ step, unsigned long long salt, int act,
                                    int in_bf16, int out_bf16) {
  const unsigned thr = eamd_drop_thr16(p);
  const float inv = eamd_drop_inv(thr);
  const unsigned seed = eamd_drop_seed(step, salt);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    float v[4];
    if (in_bf16) {
      const uint2 r = reinterpret_cast<const uint2*>(x)[q];
      v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
      v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    } else {
      const float4 r = reinterpret_cast<const float4*>(x)[q];
      v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    bool keep[4];
    eamd_drop_keep4(seed, (unsigned long long)(4 * q), thr, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? eamd_act(v[e], act) * inv : 0.f;
    if (out_bf16) {
      uint2 o;
      o.x = (unsigned)eamd_f2bf(v[0]) | ((unsigned)eamd_f2bf(v[1]) << 16);
      o.y = (unsigned)eamd_f2bf(v[2]) | ((unsigned)eamd_f2bf(v[3]) << 16);
      reinterpret_cast<uint2*>(y)[q] = o;
    } else {
      reinterpret_cast<float4*>(y)[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
__global__ void rng_advance_kernel(unsigned long long* step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1ULL;
}

// ---- conv1d positionwise layers: im2col along time ------------------------------------------------------------
// col[(b,t), kk*C + c] = x[b, t + kk - p, c] (0 outside the sequence), p = (k-1)/2; 4-byte words, so bf16 rows are
// copied as channel pairs (C even) and fp32 rows as they are.  reference: transformer/multi_layer_conv.py:13-105
// (torch.nn.Conv1d(padding=(k-1)//2) over the padded batch).
__global__ void unfold1d_kernel(const unsigned int* __restrict__ x, unsigned int* __restrict__ col, int B, int T, int CW,
                                int k) {
  const int p = (k - 1) / 2;
  const long n = (long)B * T * k * CW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = i % CW; long r = i / CW;
    const int kk = r % k; r /= k;
    const int t = r % T; const int b = r / T;
    const int ts = t + kk - p;
    col[i] = (ts >= 0 && ts < T) ? x[((long)b * T + ts) * CW + c] : 0u;
  }
}
// dx[b,t,c] = sum_kk dcol[(b, t - kk + p), kk*C + c] over the rows inside the sequence
__global__ void fold1d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int T, int C, int k) {
  const int p = (k - 1) / 2;
  const long n = (long)B * T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = i % C; long r = i / C;
    const int t = r % T; const int b = r / T;
    float acc = 0.f;
    for (int kk = 0; kk < k; ++kk) {
      const int ts = t - kk + p;
      if (ts >= 0 && ts < T) acc += dcol[(((long)b * T + ts) * k + kk) * C + c];
    }
    dx[i] = acc;
  }
}

}  // namespace

extern "C" {

int eamd_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, void* stream) {
  if (!x || !out || n < 0) return EAMD_EINVAL;
  if (n == 0) return EAMD_OK;
  const bool vec = ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)y) & 15) == 0);
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(vec ? n / 4 + 1 : n)), dim3(256), 0, (hipStream_t)stream, x, y,
                     out, (long)n, vec ? (long)(n / 4) : 0L, a, b);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_cast_bf16(const float* x, void* y, int64_t n, void* stream) {
  if (!x || !y || n < 0) return EAMD_EINVAL;
  if (n == 0) return EAMD_OK;
  const bool vec = ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)y) & 15) == 0);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(vec ? n / 8 + 1 : n)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y, (long)n, vec ? (long)(n / 8) : 0L);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_scale_dev(const float* x, const float* scale_dev, float* out, int64_t n, float extra, void* stream) {
  if (!x || !out || !scale_dev || n < 0) return EAMD_EINVAL;
  if (n == 0) return EAMD_OK;
  hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, scale_dev, out,
                     (long)n, extra);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_act_fwd(const float* x, float* y, int64_t n, int act, void* stream) {
  if (!x || !y || n < 0) return EAMD_EINVAL;
  if (n == 0) return EAMD_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, act);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, void* stream) {
  if (!dy || !x || !dx || n < 0) return EAMD_EINVAL;
  if (n == 0) return EAMD_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, (long)n, act);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

// rows whose time index (row % T) is not below bound[0] <- 0, in place (fp32, or bf16 with 2-byte elements)
__global__ __launch_bounds__(256) void mask_time_f32_kernel(float* __restrict__ x, long rows, int C, int T, const int* __restrict__ bound) {
  const int tb = bound[0];
  const long n = rows * C, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if ((int)((i / C) % T) >= tb) x[i] = 0.f;
}
__global__ __launch_bounds__(256) void mask_time_b16_kernel(unsigned short* __restrict__ x, long rows, int C, int T, const int* __restrict__ bound) {
  const int tb = bound[0];
  const long n = rows * C, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if ((int)((i / C) % T) >= tb) x[i] = 0;
}
int eamd_mask_time(void* x, int64_t rows, int C, int T, const int32_t* bound, int is_bf16, void* stream) {
  if (!x || !bound || rows <= 0 || C <= 0 || T <= 0) return EAMD_EINVAL;
  if (is_bf16) hipLaunchKernelGGL(mask_time_b16_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream,
                                  (unsigned short*)x, (long)rows, C, T, (const int*)bound);
  else hipLaunchKernelGGL(mask_time_f32_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, (float*)x,
                          (long)rows, C, T, (const int*)bound);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_mask_rows(const float* x, const uint8_t* keep, float* y, int64_t rows, int D, void* stream) {
  if (!x || !keep || !y || rows <= 0 || D <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, x, keep, y,
                     (long)rows, D);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_unfold1d(const void* x, void* col, int B, int T, int C, int k, int bf16, void* stream) {
  if (!x || !col || x == col || B <= 0 || T <= 0 || C <= 0 || k <= 0 || (k & 1) == 0) return EAMD_EINVAL;
  if (bf16 && (C & 1)) return EAMD_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)col) & 3) return EAMD_EINVAL;
  const int CW = bf16 ? C / 2 : C;
  hipLaunchKernelGGL(unfold1d_kernel, dim3(grid_for((long)B * T * k * CW)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned int*)x, (unsigned int*)col, B, T, CW, k);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_fold1d(const float* dcol, float* dx, int B, int T, int C, int k, void* stream) {
  if (!dcol || !dx || dcol == dx || B <= 0 || T <= 0 || C <= 0 || k <= 0 || (k & 1) == 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(fold1d_kernel, dim3(grid_for((long)B * T * C)), dim3(256), 0, (hipStream_t)stream, dcol, dx, B, T, C, k);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_glu_fwd(const float* x, float* y, int64_t rows, int C, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows, C);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_glu_bwd(const float* dy, const float* x, float* dx, void* dx_bf16, int64_t rows, int C, void* stream) {
  if (!dy || !x || (!dx && !dx_bf16) || rows <= 0 || C <= 0) return EAMD_EINVAL;
  if (dx_bf16 && C % 2 == 0 && (((uintptr_t)dy | (uintptr_t)x) & 7) == 0 && ((uintptr_t)dx_bf16 & 3) == 0) {
    hipLaunchKernelGGL(glu_bwd_x2_kernel, dim3(grid_for(rows * C / 2)), dim3(256), 0, (hipStream_t)stream, dy, x,
                       (unsigned int*)dx_bf16, (long)rows, C);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, x, dx,
                     (unsigned short*)dx_bf16, (long)rows, C);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

// out = bf16(a + b) together with the column sums of a and of b (ADDED into suma / sumb): the relative-position
}  // extern "C"

// attention backward needs dq = dqu + dqv as a GEMM operand and the column sums of both terms as the gradients of
// pos_bias_u / pos_bias_v - one pass over the two [rows, D] tensors instead of three.  Thread = (column pair, row
// group); a block owns `rpb` rows, sums in registers, combines its row groups in LDS, one atomic per column.
// F32OUT: the sum is stored as fp32 (reference-precision mode) instead of a bf16 pair.
template <bool F32OUT>
__global__ __launch_bounds__(256) void add_cast_colsum2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               unsigned int* __restrict__ o32, long ld_out,
                                                               float* __restrict__ suma, float* __restrict__ sumb,
                                                               long rows, int D, int rpb) {
  __shared__ float red[256][4];
  const int CW = D / 2, ngrp = 256 / CW;
  const int cp = threadIdx.x % CW, grp = threadIdx.x / CW;
  const long r0 = (long)blockIdx.x * rpb, r1 = min(rows, r0 + (long)rpb);
  float2 sa = make_float2(0.f, 0.f), sb = sa;
  if (grp < ngrp) {
    long r = r0 + grp;
    for (; r + 3L * ngrp < r1; r += 4L * ngrp) {       // four rows per trip, loads issued together
      float2 av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = *reinterpret_cast<const float2*>(a + (r + (long)u * ngrp) * D + 2 * cp);
        bv[u] = *reinterpret_cast<const float2*>(b + (r + (long)u * ngrp) * D + 2 * cp);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sa.x += av[u].x; sa.y += av[u].y; sb.x += bv[u].x; sb.y += bv[u].y;
        if (F32OUT) *reinterpret_cast<float2*>(reinterpret_cast<float*>(o32) + (r + (long)u * ngrp) * ld_out + 2 * cp) =
            make_float2(av[u].x + bv[u].x, av[u].y + bv[u].y);
        else o32[((r + (long)u * ngrp) * ld_out + 2 * cp) >> 1] = eamd_pack2(av[u].x + bv[u].x, av[u].y + bv[u].y);
      }
    }
    for (; r < r1; r += ngrp) {
      const float2 av = *reinterpret_cast<const float2*>(a + r * D + 2 * cp);
      const float2 bv = *reinterpret_cast<const float2*>(b + r * D + 2 * cp);
      sa.x += av.x; sa.y += av.y; sb.x += bv.x; sb.y += bv.y;
      if (F32OUT) *reinterpret_cast<float2*>(reinterpret_cast<float*>(o32) + r * ld_out + 2 * cp) = make_float2(av.x + bv.x, av.y + bv.y);
      else o32[(r * ld_out + 2 * cp) >> 1] = eamd_pack2(av.x + bv.x, av.y + bv.y);
    }
  }
  red[threadIdx.x][0] = sa.x; red[threadIdx.x][1] = sa.y; red[threadIdx.x][2] = sb.x; red[threadIdx.x][3] = sb.y;
  __syncthreads();
  if (threadIdx.x < CW) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int g = 0; g < ngrp; ++g) {
      const float* q = red[g * CW + threadIdx.x];
      t0 += q[0]; t1 += q[1]; t2 += q[2]; t3 += q[3];
    }
    atomicAdd(&suma[2 * cp], t0); atomicAdd(&suma[2 * cp + 1], t1);
    atomicAdd(&sumb[2 * cp], t2); atomicAdd(&sumb[2 * cp + 1], t3);
  }
}

extern "C" {

int eamd_add_block_f32(const float* a, const float* b, float* out, int64_t rows, int cols, int64_t ld_out, void* stream) {
  if (!a || !out || rows <= 0 || cols <= 0 || ld_out < cols) return EAMD_EINVAL;
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(add_block_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, cols,
                     (long)ld_out);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_add_cast_bf16(const float* a, const float* b, void* out_bf16, int64_t rows, int cols, int64_t ld_out,
                       void* stream) {
  if (!a || !out_bf16 || rows <= 0 || cols <= 0 || ld_out < cols) return EAMD_EINVAL;
  const long n = (long)rows * cols;
  if (cols % 2 == 0 && ld_out % 2 == 0 && (((uintptr_t)a | (uintptr_t)b) & 7) == 0 && ((uintptr_t)out_bf16 & 3) == 0) {
    hipLaunchKernelGGL(add_cast_x2_kernel, dim3(grid_for(n / 2)), dim3(256), 0, (hipStream_t)stream, a, b,
                       (unsigned int*)out_bf16, n / 2, cols, (long)ld_out);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(add_cast_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b,
                     (unsigned short*)out_bf16, n, cols, (long)ld_out);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_add_cast_colsum2(const float* a, const float* b, void* out_bf16, int64_t ld_out, float* suma, float* sumb,
                            int64_t rows, int D, void* stream) {
  if (!a || !b || !out_bf16 || !suma || !sumb || rows <= 0 || D <= 0 || ld_out < D) return EAMD_EINVAL;
  if (D % 2 || D > 512 || ld_out % 2 || (((uintptr_t)a | (uintptr_t)b) & 7) || ((uintptr_t)out_bf16 & 3)) return EAMD_EUNSUPPORTED;
  static const int rpb = [] { const char* e = getenv("EAMD_ACC_RPB"); return e ? atoi(e) : 64; }();
  hipLaunchKernelGGL(add_cast_colsum2_kernel<false>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, a, b,
                     (unsigned int*)out_bf16, (long)ld_out, suma, sumb, (long)rows, D, rpb);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_add_colsum2_f32(const float* a, const float* b, float* out, int64_t ld_out, float* suma, float* sumb, int64_t rows,
                         int D, void* stream) {
  if (!a || !b || !out || !suma || !sumb || rows <= 0 || D <= 0 || ld_out < D) return EAMD_EINVAL;
  if (D % 2 || D > 512 || ld_out % 2 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 7)) return EAMD_EUNSUPPORTED;
  static const int rpb = [] { const char* e = getenv("EAMD_ACC_RPB"); return e ? atoi(e) : 64; }();
  hipLaunchKernelGGL(add_cast_colsum2_kernel<true>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, a, b,
                     (unsigned int*)out, (long)ld_out, suma, sumb, (long)rows, D, rpb);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_add_bias2(const void* q, int64_t ldq, const float* u, const float* v, void* qu, void* qv, int64_t rows, int D,
                   int bf16, void* stream) {
  if (!q || !u || !v || !qu || !qv || rows <= 0 || D <= 0 || ldq < D) return EAMD_EINVAL;
  if (bf16 && D % 2 == 0 && ldq % 2 == 0 && (((uintptr_t)q | (uintptr_t)qu | (uintptr_t)qv) & 3) == 0) {
    hipLaunchKernelGGL(add_bias2_x2_kernel, dim3(grid_for(rows * D / 2)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned int*)q, u, v, (unsigned int*)qu, (unsigned int*)qv, (long)rows, D, (long)ldq);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(add_bias2_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, (const float*)q, u,
                     v, (float*)qu, (float*)qv, (long)rows, D, bf16, (long)ldq);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_colsum(const void* x, int64_t ld, float* out, int64_t rows, int D, float scale, int bf16, void* stream) {
  if (!x || !out || rows <= 0 || D <= 0) return EAMD_EINVAL;
  const bool pair = bf16 && D % 2 == 0 && ld % 2 == 0 && ((uintptr_t)x & 3) == 0;
  const int ncol = pair ? D / 2 : D;                      // threads needed along the columns
  const int nthr = 256;
  const int gx = (ncol + nthr - 1) / nthr;
  static const int min_rpb = [] { const char* e = getenv("EAMD_COLSUM_RPB"); return e ? atoi(e) : 128; }();
  // ~1024 workgroups: more of them only lengthen the same-address atomic chains (measured: 2048 blocks on a
  // [151k, 256] bf16 matrix take 109 us, 1024 take 61 us)
  long want = 1024 / gx; if (want < 1) want = 1;
  long rpb = (rows + want - 1) / want; if (rpb < (pair ? min_rpb / 2 : min_rpb)) rpb = pair ? min_rpb / 2 : min_rpb;
  if (rpb < 8) rpb = 8;
  const int gy = (int)((rows + rpb - 1) / rpb);
  if (pair) {
    hipLaunchKernelGGL(colsum_bf16x2_kernel, dim3(gx, gy), dim3(nthr), 0, (hipStream_t)stream, (const unsigned int*)x,
                       (long)(ld / 2), out, (long)rows, D, (int)rpb, scale);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(nthr), 0, (hipStream_t)stream, (const float*)x, (long)ld, out,
                     (long)rows, D, (int)rpb, scale, bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_embed_pe(const int64_t* tok, const float* table, const float* pe, float* out, int64_t rows, int U,
                  int D, float scale, int pos_offset, void* stream) {
  return eamd_embed_pe_ld(tok, 1, table, pe, out, rows, U, D, scale, pos_offset, stream);
}

int eamd_embed_pe_ld(const int64_t* tok, int64_t ldt, const float* table, const float* pe, float* out, int64_t rows, int U,
                     int D, float scale, int pos_offset, void* stream) {
  return eamd_embed_pe_dyn(tok, ldt, table, pe, out, rows, U, D, scale, pos_offset, nullptr, stream);
}

int eamd_embed_pe_dyn(const int64_t* tok, int64_t ldt, const float* table, const float* pe, float* out, int64_t rows, int U,
                      int D, float scale, int pos_offset, const int32_t* pos_dev, void* stream) {
  if (!tok || !table || !out || rows <= 0 || U <= 0 || D <= 0 || ldt < 1) return EAMD_EINVAL;
  hipLaunchKernelGGL(embed_pe_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)tok, (long)ldt, table, pe, out, (long)rows, U, D, scale, pos_offset, pos_dev);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_embed_bwd(const int64_t* tok, const float* dout, float* dtable, int64_t rows, int D, float scale,
                   int64_t pad_idx, void* stream) {
  if (!tok || !dout || !dtable || rows <= 0 || D <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)tok, dout, dtable, (long)rows, D, scale, (long)pad_idx);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_posenc(const float* x, const float* pe, float* out, int64_t rows, int T, int D, float scale,
                void* stream) {
  if (!x || !pe || !out || rows <= 0 || T <= 0 || D <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(posenc_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, x, pe, out,
                     (long)rows, T, D, scale);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_posenc_scaled(const float* x, const float* pe, const float* alpha, float* out, int64_t rows, int T, int D,
                       float scale, void* stream) {
  if (!x || !pe || !alpha || !out || rows <= 0 || T <= 0 || D <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(posenc_scaled_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, x, pe, alpha,
                     out, (long)rows, T, D, scale);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_posenc_scaled_bwd(const float* dout, const float* pe, float* dalpha, int64_t rows, int T, int D,
                           void* stream) {
  if (!dout || !pe || !dalpha || rows <= 0 || T <= 0 || D <= 0) return EAMD_EINVAL;
  const long n = (long)rows * D;
  const int blocks = (int)std::min<long>(512, (n + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(posenc_scaled_bwd_kernel, dim3(std::max(1, blocks)), dim3(256), 0, (hipStream_t)stream, dout, pe,
                     dalpha, (long)rows, T, D);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_permute4(const float* src, float* dst, int d0, int d1, int d2, int d3, int64_t s0, int64_t s1,
                  int64_t s2, int64_t s3, int accumulate, void* stream) {
  if (!src || !dst || d0 <= 0 || d1 <= 0 || d2 <= 0 || d3 <= 0) return EAMD_EINVAL;
  long n = (long)d0 * d1 * d2 * d3;
  hipLaunchKernelGGL(permute4_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, d0, d1, d2,
                     d3, (long)s0, (long)s1, (long)s2, (long)s3, accumulate);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dropout(const void* x, void* y, int64_t n, float p, const uint64_t* step_dev, uint64_t salt, int act,
                 int in_bf16, int out_bf16, void* stream) {
  if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return EAMD_EINVAL;
  if (n == 0) return EAMD_OK;
  if (n % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    hipLaunchKernelGGL(dropout_vec4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (float*)y, (long)(n / 4), p, (const unsigned long long*)step_dev, (unsigned long long)salt, act,
                       in_bf16, out_bf16);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y,
                     (long)n, p, (const unsigned long long*)step_dev, (unsigned long long)salt, act, in_bf16, out_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_rng_advance(uint64_t* step_dev, void* stream) {
  if (!step_dev) return EAMD_EINVAL;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)step_dev);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
