// Conformer convolution-module kernels on channels-last [B, T, C] activations (C contiguous, so a
// wave's 64 lanes read 256 contiguous bytes): depthwise Conv1d fwd/bwd, BatchNorm1d (training batch
// statistics over all B*T positions, padded frames included - reference behaviour) fused with the
// following activation, and the first Conv2dSubsampling convolution (C_in = 1) as a direct kernel.
// reference: conformer/convolution.py:13-79, transformer/subsampling.py:28-33.
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

constexpr int KMAX = 32;

// y[b,t,c] = bias[c] + sum_k w[c,k] * x[b, t + k - pad, c]         (flip=0, forward)
// y[b,t,c] =           sum_k w[c,k] * x[b, t - k + pad, c]         (flip=1, input gradient)
// One thread = one channel x TT consecutive frames: the K taps stay in registers and the input
// window (TT + K - 1 values) is read once, lanes along C (coalesced 256-B rows).
constexpr int TT = 8;
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     int B, int T, int C, int K, int pad, int flip) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int nchunk = (T + TT - 1) / TT;
  const int b = blockIdx.y / nchunk;
  const int t0 = (blockIdx.y % nchunk) * TT;
  float wr[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int kk = flip ? K - 1 - k : k;          // flipped taps turn the gradient into the same correlation
    wr[k] = (k < K) ? w[(long)c * K + kk] : 0.f;
  }
  const float bv = (bias && !flip) ? bias[c] : 0.f;
  float acc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) acc[i] = bv;
  const float* xb = x + (long)b * T * C + c;
  // window position j covers frame t0 - pad + j, j in [0, TT + K - 1)
#pragma unroll
  for (int j = 0; j < TT + KMAX - 1; ++j) {
    if (j < TT + K - 1) {
      const int ts = t0 - pad + j;
      const float xv = (ts >= 0 && ts < T) ? xb[(long)ts * C] : 0.f;
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        const int k = j - i;                 // tap index feeding output t0 + i
        if (k >= 0 && k < KMAX) acc[i] += wr[k] * xv;
      }
    }
  }
  float* yb = y + (long)b * T * C + c;
#pragma unroll
  for (int i = 0; i < TT; ++i)
    if (t0 + i < T) yb[(long)(t0 + i) * C] = acc[i];
}

// dw[c,k] += sum_{b,t} dy[b,t,c] * x[b,t+k-pad,c];  db[c] += sum dy.
// One thread = one channel; a block walks `chunks_per_block` chunks of TT frames.  Per chunk the
// TT + K - 1 input window and the TT output gradients are loaded up front (independent loads, all
// in flight together) and combined with fully unrolled FMAs into K register accumulators.
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ db, int B,
                                                           int T, int C, int K, int pad, int chunks_per_block) {
  const int c_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = c_raw < C;
  const int c = live ? c_raw : C - 1;        // dead lanes shadow a valid channel and contribute nothing
  const int nchunk = (T + TT - 1) / TT;
  const long total = (long)B * nchunk;
  const long q0 = (long)blockIdx.y * chunks_per_block;
  const long q1 = live ? min(total, q0 + (long)chunks_per_block) : q0;
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  float accb = 0.f;
  for (long q = q0; q < q1; ++q) {
    const int b = q / nchunk;
    const int t0 = (int)(q % nchunk) * TT;
    const float* xb = x + (long)b * T * C + c;
    const float* gb = dy + (long)b * T * C + c;
    float g[TT];
#pragma unroll
    for (int i = 0; i < TT; ++i) { g[i] = (t0 + i < T) ? gb[(long)(t0 + i) * C] : 0.f; accb += g[i]; }
#pragma unroll
    for (int j = 0; j < TT + KMAX - 1; ++j) {
      if (j < TT + K - 1) {
        const int ts = t0 - pad + j;
        const float xv = (ts >= 0 && ts < T) ? xb[(long)ts * C] : 0.f;
#pragma unroll
        for (int i = 0; i < TT; ++i) {
          const int k = j - i;
          if (k >= 0 && k < KMAX) acc[k] += g[i] * xv;
        }
      }
    }
  }
  // transpose through LDS so that a wave's atomics hit 256 contiguous bytes (lanes along k within a channel
  // row), not 64 addresses K floats apart (uncoalesced f32 atomics run ~17x slower on gfx950)
  __shared__ float tr[256 * (KMAX + 1)];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) tr[threadIdx.x * (KMAX + 1) + k] = acc[k];
  __syncthreads();
  const int c0 = blockIdx.x * blockDim.x;
  const int nch = min(256, C - c0);
  for (int i = threadIdx.x; i < nch * K; i += blockDim.x) {
    const int cc = i / K, k = i % K;
    atomicAdd(&dw[(long)c0 * K + i], tr[cc * (KMAX + 1) + k]);
  }
  if (db && live) atomicAdd(&db[c], accb);
}

// ---- BatchNorm1d over [M, C] -----------------------------------------------------------------
// stage 1: per (slab, channel) count / mean / M2 (two passes over the slab, cache resident)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                         long M, int C, int rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + (long)rows_per_block);
  float s = 0.f;
  for (long r = r0; r < r1; ++r) s += x[r * C + c];
  const float n = (float)(r1 - r0);
  const float mean = s / n;
  float m2 = 0.f;
  for (long r = r0; r < r1; ++r) { float d = x[r * C + c] - mean; m2 += d * d; }
  float* p = part + ((long)blockIdx.y * 3) * C;
  p[c] = n; p[C + c] = mean; p[2 * C + c] = m2;
}
// stage 2: Chan combination in slab order (deterministic); writes mean, rstd, updates running stats
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nslab, int C, float eps, float momentum,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                   float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int s = 0; s < nslab; ++s) {
    const float* p = part + (long)s * 3 * C;
    float nb = p[c], mb = p[C + c], m2b = p[2 * C + c];
    float nt = n + nb;
    float d = mb - mean;
    mean += d * nb / nt;
    m2 += m2b + d * d * n * nb / nt;
    n = nt;
  }
  const float var = m2 / n;
  mean_out[c] = mean;
  rstd_out[c] = rsqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
  }
}
// y = act((x - mean) * rstd * gamma + beta)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long M, int C, int act,
                                int bf16) {
  const long n = M * C;
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned short* y16 = reinterpret_cast<unsigned short*>(y);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c = i % C;
    float z = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    z = eamd_act(z, act);
    if (bf16) y16[i] = eamd_f2bf(z); else y[i] = z;
  }
}
// backward stage 1: dz = dy * act'(z); partial sums of dz and dz*xhat per (slab, channel)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ part,
                                                             long M, int C, int rows_per_block, int act) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + (long)rows_per_block);
  const float mu = mean[c], rs = rstd[c], g = gamma[c], be = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (long r = r0; r < r1; ++r) {
    float xh = (x[r * C + c] - mu) * rs;
    float z = xh * g + be;
    float d = dy[r * C + c];
    if (act == EAMD_ACT_SWISH) d *= eamd_dswish(z);
    else if (act == EAMD_ACT_RELU) d = z > 0.f ? d : 0.f;
    s1 += d; s2 += d * xh;
  }
  float* p = part + (long)blockIdx.y * 2 * C;
  p[c] = s1; p[C + c] = s2;
}
// backward stage 2: reduce partials in slab order -> sums[2][C]; accumulate dgamma/dbeta
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nslab, int C, float* __restrict__ sums,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int s = 0; s < nslab; ++s) { s1 += part[(long)s * 2 * C + c]; s2 += part[(long)s * 2 * C + C + c]; }
  sums[c] = s1; sums[C + c] = s2;
  dbeta[c] += s1; dgamma[c] += s2;
}
// backward stage 3: dx = gamma*rstd*(dz - s1/M - xhat*s2/M)   (training-mode statistics)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ sums, float* __restrict__ dx, long M, int C, int act,
                                    int training) {
  const long n = M * C;
  const long stride = (long)gridDim.x * blockDim.x;
  const float invM = 1.f / (float)M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c = i % C;
    float rs = rstd[c], g = gamma[c];
    float xh = (x[i] - mean[c]) * rs;
    float z = xh * g + beta[c];
    float d = dy[i];
    if (act == EAMD_ACT_SWISH) d *= eamd_dswish(z);
    else if (act == EAMD_ACT_RELU) d = z > 0.f ? d : 0.f;
    if (training) dx[i] = g * rs * (d - sums[c] * invM - xh * sums[C + c] * invM);
    else dx[i] = g * rs * d;
  }
}

// ---- Conv2dSubsampling first convolution: 1 -> C channels, 3x3, stride 2, + ReLU, NHWC output ----
// x [B, T, F]; w [C, 1, 3, 3]; y [B, H, W, C] with H=(T-3)/2+1, W=(F-3)/2+1
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B,
                                                        int T, int F, int H, int W, int C, int bf16) {
  const long n = (long)B * H * W * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = i % C;
    long p = i / C;
    const int ww = p % W; p /= W;
    const int hh = p % H; const long b = p / H;
    const float* xp = x + (b * T + 2 * hh) * F + 2 * ww;
    const float* wc = w + (long)c * 9;
    float acc = bias[c];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc += wc[kh * 3 + kw] * xp[kh * F + kw];
    acc = acc > 0.f ? acc : 0.f;
    if (bf16) reinterpret_cast<unsigned short*>(y)[i] = eamd_f2bf(acc); else y[i] = acc;
  }
}
// dW[c, kh, kw] += sum_pos dy[pos, c] * x[pos shifted]; db[c] += sum dy   (dy already ReLU-masked)
__global__ __launch_bounds__(256) void conv1_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dw, float* __restrict__ db, int B,
                                                          int T, int F, int H, int W, int C, int pos_per_block,
                                                          int bf16) {
  const int c_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = c_raw < C;
  const int c = live ? c_raw : C - 1;
  const long npos = (long)B * H * W;
  const long p0 = (long)blockIdx.y * pos_per_block;
  const long p1 = live ? min(npos, p0 + (long)pos_per_block) : p0;
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  float accb = 0.f;
  for (long p = p0; p < p1; ++p) {
    const int ww = p % W; long q = p / W;
    const int hh = q % H; const long b = q / H;
    const float g = bf16 ? __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(dy)[p * C + c]) << 16)
                         : dy[p * C + c];
    accb += g;
    const float* xp = x + (b * T + 2 * hh) * F + 2 * ww;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] += g * xp[kh * F + kw];
  }
  __shared__ float tr9[256 * 10];
#pragma unroll
  for (int k = 0; k < 9; ++k) tr9[threadIdx.x * 10 + k] = acc[k];
  __syncthreads();
  const int c0 = blockIdx.x * blockDim.x;
  const int nch = min(256, C - c0);
  for (int i = threadIdx.x; i < nch * 9; i += blockDim.x) atomicAdd(&dw[(long)c0 * 9 + i], tr9[(i / 9) * 10 + (i % 9)]);
  if (live) atomicAdd(&db[c], accb);
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

}  // namespace

extern "C" {

int eamd_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int K,
                    void* stream) {
  if (!x || !w || !y || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(dwconv_kernel, dim3((C + 255) / 256, B * ((T + TT - 1) / TT)), dim3(256), 0, (hipStream_t)stream,
                     x, w, bias, y, B, T, C, K, (K - 1) / 2, 0);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dwconv_bwd_x(const float* dy, const float* w, float* dx, int B, int T, int C, int K, void* stream) {
  if (!dy || !w || !dx || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(dwconv_kernel, dim3((C + 255) / 256, B * ((T + TT - 1) / TT)), dim3(256), 0, (hipStream_t)stream,
                     dy, w, (const float*)nullptr, dx, B, T, C, K, (K - 1) / 2, 1);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dwconv_bwd_w(const float* dy, const float* x, float* dw, float* db, int B, int T, int C, int K,
                      void* stream) {
  if (!dy || !x || !dw || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  long total = (long)B * ((T + TT - 1) / TT);
  int gx = (C + 255) / 256;
  long want = 512 / gx; if (want < 1) want = 1;
  long cpb = (total + want - 1) / want; if (cpb < 1) cpb = 1;
  int gy = (int)((total + cpb - 1) / cpb);
  hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, dy, x, dw, db, B, T, C,
                     K, (K - 1) / 2, (int)cpb);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* workspace: 3*C*nslab floats where nslab = eamd_bn_nslab(M, C) */
int eamd_bn_nslab(int64_t M, int C) {
  int gx = (C + 255) / 256;
  long want = 96 / gx; if (want < 1) want = 1;
  long rpb = (M + want - 1) / want; if (rpb < 16) rpb = 16;
  return (int)((M + rpb - 1) / rpb);
}
static long bn_rpb(long M, int C) {
  int gx = (C + 255) / 256;
  long want = 96 / gx; if (want < 1) want = 1;
  long rpb = (M + want - 1) / want; if (rpb < 16) rpb = 16;
  return rpb;
}

int eamd_bn_stats(const float* x, float* workspace, float* mean, float* rstd, float* running_mean,
                  float* running_var, int64_t M, int C, float eps, float momentum, void* stream) {
  if (!x || !workspace || !mean || !rstd || M <= 0 || C <= 0) return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int gx = (C + 255) / 256;
  long rpb = bn_rpb(M, C);
  int nslab = (int)((M + rpb - 1) / rpb);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(gx, nslab), dim3(256), 0, s, x, workspace, (long)M, C, (int)rpb);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(gx), dim3(256), 0, s, workspace, nslab, C, eps, momentum, mean, rstd,
                     running_mean, running_var);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  void* y, int64_t M, int C, int act, int y_bf16, void* stream) {
  if (!x || !mean || !rstd || !gamma || !beta || !y || M <= 0 || C <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * C)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma,
                     beta, (float*)y, (long)M, C, act, y_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* workspace: (2*nslab + 2)*C floats */
int eamd_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                const float* beta, float* workspace, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                int act, int training, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !beta || !workspace || !dx || !dgamma || !dbeta || M <= 0 || C <= 0)
    return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int gx = (C + 255) / 256;
  long rpb = bn_rpb(M, C);
  int nslab = (int)((M + rpb - 1) / rpb);
  float* sums = workspace + (long)nslab * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(gx, nslab), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta,
                     workspace, (long)M, C, (int)rpb, act);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(gx), dim3(256), 0, s, workspace, nslab, C, sums, dgamma, dbeta);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * C)), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta,
                     sums, dx, (long)M, C, act, training);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int T, int F, int C,
                   int y_bf16, void* stream) {
  if (!x || !w || !bias || !y || B <= 0 || T < 3 || F < 3 || C <= 0) return EAMD_EINVAL;
  int H = (T - 3) / 2 + 1, W = (F - 3) / 2 + 1;
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, w,
                     bias, (float*)y, B, T, F, H, W, C, y_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_conv1_bwd_w(const void* dy, const float* x, float* dw, float* db, int B, int T, int F, int C,
                     int dy_bf16, void* stream) {
  if (!dy || !x || !dw || !db || B <= 0 || T < 3 || F < 3 || C <= 0) return EAMD_EINVAL;
  int H = (T - 3) / 2 + 1, W = (F - 3) / 2 + 1;
  long npos = (long)B * H * W;
  int gx = (C + 255) / 256;
  long want = 2048 / gx; if (want < 1) want = 1;
  long ppb = (npos + want - 1) / want; if (ppb < 16) ppb = 16;
  int gy = (int)((npos + ppb - 1) / ppb);
  hipLaunchKernelGGL(conv1_bwd_w_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const float*)dy, x, dw, db,
                     B, T, F, H, W, C, (int)ppb, dy_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
