// RNN-Transducer: joint network pointwise parts and the transducer loss (Graves 2012) on the
// (T, U+1) lattice.  reference: transducer/joint_network.py:34-48 (lin_out(act(lin_enc(h_enc) +
// lin_dec(h_dec)))), transducer/loss.py:8-79 (warp-transducer RNNTLoss(blank), raw logits in,
// log-softmax inside, mean over the batch), transducer/utils.py:9-53 (targets / lengths).
// The three Linear layers run through eamd_gemm; the kernels here are
//   joint_fwd / joint_bwd_* : broadcast add + activation over [B,T,U,J] and its two reductions,
//   rnnt_lse_gather         : per lattice node log-sum-exp over V and the two log-probs the lattice uses,
//   rnnt_alpha_beta         : forward/backward variables, one workgroup per (utterance, direction),
//                             anti-diagonal wavefront with the previous diagonal held in LDS,
//   rnnt_grad               : d loss / d logits written in place over the logits.
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

__global__ void joint_fwd_kernel(const float* __restrict__ e, const float* __restrict__ d, float* __restrict__ out,
                                 unsigned short* __restrict__ out16, int B, int T, int U, int J, int act) {
  const long n = (long)B * T * U * J;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = i % J; long p = i / J;
    const int u = p % U; p /= U;
    const int t = p % T; const int b = p / T;
    const float v = eamd_act(e[((long)b * T + t) * J + j] + d[((long)b * U + u) * J + j], act);
    if (out) out[i] = v;
    if (out16) out16[i] = eamd_f2bf(v);
  }
}

// d_enc[b,t,j] = sum_u dh[b,t,u,j] * act'(e[b,t,j] + d[b,u,j])      grid (B*T), threads over j
__global__ __launch_bounds__(256) void joint_bwd_enc_kernel(const float* __restrict__ dh, const float* __restrict__ e,
                                                            const float* __restrict__ d, float* __restrict__ de,
                                                            int B, int T, int U, int J, int act) {
  const long bt = blockIdx.x;
  const int b = bt / T;
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    const float ev = e[bt * J + j];
    const float* dhp = dh + bt * U * J + j;
    const float* dp = d + (long)b * U * J + j;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // eight label positions in flight (a chain of L2 trips otherwise)
    int u = 0;
    for (; u + 8 <= U; u += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s8[k] += dhp[(long)(u + k) * J] * eamd_dact(ev + dp[(long)(u + k) * J], act);
    }
    for (; u < U; ++u) s8[0] += dhp[(long)u * J] * eamd_dact(ev + dp[(long)u * J], act);
    de[bt * J + j] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  }
}
// d_dec[b,u,j] = sum_t dh[b,t,u,j] * act'(e[b,t,j] + d[b,u,j])      grid (B*U, frame slices); slices > 1: atomics into zeroed dd
__global__ __launch_bounds__(256) void joint_bwd_dec_kernel(const float* __restrict__ dh, const float* __restrict__ e,
                                                            const float* __restrict__ d, float* __restrict__ dd,
                                                            int B, int T, int U, int J, int act) {
  const long bu = blockIdx.x;
  const int b = bu / U, u = bu % U;
  const int per = (T + gridDim.y - 1) / gridDim.y;
  const int ta = blockIdx.y * per, tb = min(T, ta + per);
  for (int j = threadIdx.x; j < J; j += blockDim.x) {
    const float dv = d[bu * J + j];
    const float* dhp = dh + ((long)b * T * U + u) * J + j;
    const float* ep = e + (long)b * T * J + j;
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int t = ta;
    for (; t + 8 <= tb; t += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s8[k] += dhp[(long)(t + k) * U * J] * eamd_dact(ep[(long)(t + k) * J] + dv, act);
    }
    for (; t < tb; ++t) s8[0] += dhp[(long)t * U * J] * eamd_dact(ep[(long)t * J] + dv, act);
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    if (gridDim.y > 1) atomicAdd(&dd[bu * J + j], s);
    else dd[bu * J + j] = s;
  }
}

// one workgroup per lattice node (b,t,u): lse over V; log p(blank), log p(next label)
__global__ __launch_bounds__(256) void rnnt_lse_gather_kernel(const float* __restrict__ z, const int* __restrict__ labels,
                                                              float* __restrict__ lse, float* __restrict__ lpb,
                                                              float* __restrict__ lpl, int T, int U, int V,
                                                              int blank, long node0) {
  __shared__ float red[16];
  // z holds the rows of the lattice nodes node0, node0+1, ... (the whole [B,T,U] lattice when node0 = 0)
  const long row = node0 + blockIdx.x;
  const float* zr = z + (long)blockIdx.x * V;
  float m = -INFINITY;
  for (int v = threadIdx.x; v < V; v += blockDim.x) m = fmaxf(m, zr[v]);
  m = block_max(m, red);
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += expf(zr[v] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float l = m + logf(s);
    const int u = row % U;
    const long b = row / ((long)T * U);
    lse[row] = l;
    lpb[row] = zr[blank] - l;
    lpl[row] = (u < U - 1) ? zr[labels[b * (U - 1) + u]] - l : -INFINITY;
  }
}

// The same with one WAVE per node and the row read once (V % 4 == 0, 16-byte aligned rows): every lane holds its 16-byte
// pieces of the row in registers (20 KB per row at V = 5000: 20 pieces a lane), takes the maximum and the sum of
// exponentials from them and the wave reduces both - no second pass over the row, no block barriers.  A workgroup per
// row read it twice through scalar loads: 1.9 TB/s on the 378 MB logits chunk of config 5.
constexpr int RNNT_LSE_MAXP = 24;           // 16-byte pieces per lane: V <= 24 * 256
__global__ __launch_bounds__(256) void rnnt_lse_gather_wave_kernel(const float* __restrict__ z, const int* __restrict__ labels,
                                                                   float* __restrict__ lse, float* __restrict__ lpb,
                                                                   float* __restrict__ lpl, int T, int U, int V, int blank,
                                                                   long node0, long nrows) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;                                   // whole wave
  const long row = node0 + r;
  const float* zr = z + r * V;
  const int np = V >> 2;                                    // pieces of the row
  float4 p[RNNT_LSE_MAXP];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < RNNT_LSE_MAXP; ++i) {
    const int q = lane + 64 * i;
    if (q < np) {
      p[i] = *reinterpret_cast<const float4*>(zr + 4 * q);
      m = fmaxf(m, fmaxf(fmaxf(p[i].x, p[i].y), fmaxf(p[i].z, p[i].w)));
    }
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RNNT_LSE_MAXP; ++i) {
    const int q = lane + 64 * i;
    if (q < np) s += (expf(p[i].x - m) + expf(p[i].y - m)) + (expf(p[i].z - m) + expf(p[i].w - m));
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float l = m + logf(s);
    const int u = (int)(row % U);
    const long b = row / ((long)T * U);
    lse[row] = l;
    lpb[row] = zr[blank] - l;
    lpl[row] = (u < U - 1) ? zr[labels[b * (U - 1) + u]] - l : -INFINITY;
  }
}
// lse / log p(blank) / log p(label) of lattice nodes from the per-column-tile (max, sum exp) partials and the two gathered
// logits that the logits GEMM's row-statistics epilogue left behind (eamd_gemm_t.stats): the logits are never stored
__global__ __launch_bounds__(256) void rnnt_stats_combine_kernel(const float* __restrict__ part, const float* __restrict__ zlab,
                                                                 const float* __restrict__ zblank, float* __restrict__ lse,
                                                                 float* __restrict__ lpb, float* __restrict__ lpl,
                                                                 int tiles_n, int U, long node0, long nrows) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  const float2* pp = reinterpret_cast<const float2*>(part) + r * tiles_n;
  float m = -INFINITY;
  for (int j = 0; j < tiles_n; ++j) m = fmaxf(m, pp[j].x);
  float sm = 0.f;
  for (int j = 0; j < tiles_n; ++j) sm += pp[j].y * expf(pp[j].x - m);
  const float l = m + logf(sm);
  const long row = node0 + r;
  const int u = (int)(row % U);
  lse[row] = l;
  lpb[row] = zblank[r] - l;
  lpl[row] = (u < U - 1) ? zlab[r] - l : -INFINITY;
}
// the per-node part of rnnt_grad_kernel: (tot, gb, gl) and the node's next label, for the gradient epilogue of the GEMM
__global__ __launch_bounds__(256) void rnnt_row_coef_kernel(const int* __restrict__ labels, const float* __restrict__ lse,
                                                            const float* __restrict__ lpb, const float* __restrict__ lpl,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta,
                                                            const int* __restrict__ tlens, const int* __restrict__ ulens,
                                                            float* __restrict__ rowc, int* __restrict__ col, int T, int U,
                                                            long node0, long nrows) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  const long row = node0 + r;
  const int u = (int)(row % U);
  const int t = (int)((row / U) % T);
  const long b = row / ((long)T * U);
  const int Tb = tlens[b], Ub = ulens[b] + 1;
  const float logZ = beta[b * (long)T * U];
  const float a = alpha[row];
  const bool valid = t < Tb && u < Ub && a > -INFINITY && beta[row] > -INFINITY && isfinite(logZ);
  float tot = -INFINITY, gb = 0.f, gl = 0.f;
  int lab = -1;
  if (valid) {
    tot = a + beta[row] - logZ - lse[row];
    if (t < Tb - 1) gb = expf(a + lpb[row] + beta[row + U] - logZ);
    else gb = (u == Ub - 1) ? expf(a + lpb[row] - logZ) : 0.f;
    if (u < Ub - 1) { lab = labels[b * (U - 1) + u]; gl = expf(a + lpl[row] + beta[row + 1] - logZ); }
  }
  rowc[r * 3] = tot; rowc[r * 3 + 1] = gb; rowc[r * 3 + 2] = gl;
  col[r] = lab;
}
static void launch_lse_gather(const float* z, const int* labels, float* lse, float* lpb, float* lpl, int T, int U, int V,
                              int blank, long node0, long nrows, hipStream_t s) {
  if (V % 4 == 0 && V <= RNNT_LSE_MAXP * 256 && (reinterpret_cast<uintptr_t>(z) & 15) == 0)
    hipLaunchKernelGGL(rnnt_lse_gather_wave_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, z, labels, lse, lpb,
                       lpl, T, U, V, blank, node0, nrows);
  else
    hipLaunchKernelGGL(rnnt_lse_gather_kernel, dim3((unsigned)nrows), dim3(256), 0, s, z, labels, lse, lpb, lpl, T, U, V,
                       blank, node0);
}

__device__ __forceinline__ float lae(float a, float b) {   // log(exp(a) + exp(b)), -inf safe
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + log1pf(expf(fminf(a, b) - m));
}

// blocks [0,B): alpha ; blocks [B,2B): beta.  alpha/beta [B,T,U]; nodes outside (t < tlen, u <= ulen) = -inf.
// loss[b] = -beta(0,0)
__global__ __launch_bounds__(256) void rnnt_alpha_beta_kernel(const float* __restrict__ lpb,
                                                              const float* __restrict__ lpl,
                                                              const int* __restrict__ tlens,
                                                              const int* __restrict__ ulens, float* __restrict__ alpha,
                                                              float* __restrict__ beta, float* __restrict__ loss,
                                                              int B, int T, int U) {
  extern __shared__ float sh[];   // [2][U]
  const bool is_beta = blockIdx.x >= B;
  const int b = is_beta ? blockIdx.x - B : blockIdx.x;
  const int Tb = tlens[b], Ub = ulens[b] + 1;   // lattice of this utterance: Tb x Ub
  const long base = (long)b * T * U;
  float* out = (is_beta ? beta : alpha) + base;
  const float* pb = lpb + base;
  const float* pl = lpl + base;
  for (int i = threadIdx.x; i < T * U; i += blockDim.x) out[i] = -INFINITY;
  if (Tb <= 0 || Tb > T || Ub <= 0 || Ub > U) { if (is_beta && threadIdx.x == 0) loss[b] = INFINITY; return; }
  float* prev = sh;
  float* cur = sh + U;
  __syncthreads();
  const int ndiag = Tb + Ub - 1;
  if (!is_beta) {
    for (int dg = 0; dg < ndiag; ++dg) {
      const int u0 = max(0, dg - (Tb - 1)), u1 = min(dg, Ub - 1);
      for (int u = u0 + threadIdx.x; u <= u1; u += blockDim.x) {
        const int t = dg - u;
        float a;
        if (dg == 0) a = 0.f;
        else {
          const float from_t = t > 0 ? prev[u] + pb[(long)(t - 1) * U + u] : -INFINITY;       // blank: (t-1,u)->(t,u)
          const float from_u = u > 0 ? prev[u - 1] + pl[(long)t * U + (u - 1)] : -INFINITY;   // label: (t,u-1)->(t,u)
          a = lae(from_t, from_u);
        }
        cur[u] = a;
        out[(long)t * U + u] = a;
      }
      __syncthreads();
      float* tmp = prev; prev = cur; cur = tmp;
    }
  } else {
    for (int dg = ndiag - 1; dg >= 0; --dg) {
      const int u0 = max(0, dg - (Tb - 1)), u1 = min(dg, Ub - 1);
      for (int u = u0 + threadIdx.x; u <= u1; u += blockDim.x) {
        const int t = dg - u;
        float v;
        if (t == Tb - 1 && u == Ub - 1) v = pb[(long)t * U + u];
        else {
          const float to_t = t < Tb - 1 ? prev[u] + pb[(long)t * U + u] : -INFINITY;          // (t,u)->(t+1,u)
          const float to_u = u < Ub - 1 ? prev[u + 1] + pl[(long)t * U + u] : -INFINITY;      // (t,u)->(t,u+1)
          v = lae(to_t, to_u);
        }
        cur[u] = v;
        out[(long)t * U + u] = v;
      }
      __syncthreads();
      float* tmp = prev; prev = cur; cur = tmp;
    }
    if (threadIdx.x == 0) loss[b] = -prev[0];
  }
}

// g[row, v] <- scale * d(-log P(y|x)) / d z[row, v]   (g may alias z: every element is read before it is written)
__global__ __launch_bounds__(256) void rnnt_grad_kernel(const float* z, float* g_out, const int* __restrict__ labels,
                                                        const float* __restrict__ lse, const float* __restrict__ lpb,
                                                        const float* __restrict__ lpl, const float* __restrict__ alpha,
                                                        const float* __restrict__ beta, const int* __restrict__ tlens,
                                                        const int* __restrict__ ulens, const float* __restrict__ gscale,
                                                        float scale, int T, int U, int V, int blank, long node0,
                                                        unsigned short* g16) {
  const long row = node0 + blockIdx.x;            // lattice node; z / g_out hold rows node0.. only
  const int u = row % U;
  const int t = (row / U) % T;
  const long b = row / ((long)T * U);
  const float* zr = z + (long)blockIdx.x * V;
  float* gr = g_out ? g_out + (long)blockIdx.x * V : nullptr;
  unsigned short* gr16 = g16 ? g16 + (long)blockIdx.x * V : nullptr;
  const int Tb = tlens[b], Ub = ulens[b] + 1;
  const float logZ = beta[b * (long)T * U];
  const float a = alpha[row];
  const bool valid = t < Tb && u < Ub && a > -INFINITY && beta[row] > -INFINITY && isfinite(logZ);
  if (!valid) {
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      if (gr) gr[v] = 0.f;
      if (gr16) gr16[v] = 0;
    }
    return;
  }
  const float sc = scale * (gscale ? gscale[0] : 1.f);
  const float tot = a + beta[row] - logZ - lse[row];       // log of (node occupancy / softmax normaliser)
  float gb, gl = 0.f;
  if (t < Tb - 1) gb = expf(a + lpb[row] + beta[row + U] - logZ);
  else gb = (u == Ub - 1) ? expf(a + lpb[row] - logZ) : 0.f;
  int lab = -1;
  if (u < Ub - 1) { lab = labels[b * (U - 1) + u]; gl = expf(a + lpl[row] + beta[row + 1] - logZ); }
  if ((V & 3) == 0 && ((reinterpret_cast<uintptr_t>(zr) | reinterpret_cast<uintptr_t>(gr)) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(gr16) & 7) == 0) {
    // 16 bytes of the row per lane and trip (8-byte groups for the bf16 gradient)
    for (int q = threadIdx.x; q < (V >> 2); q += blockDim.x) {
      const int v = 4 * q;
      const float4 zv = *reinterpret_cast<const float4*>(zr + v);
      float g[4] = {expf(zv.x + tot), expf(zv.y + tot), expf(zv.z + tot), expf(zv.w + tot)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (v + k == blank) g[k] -= gb;
        if (v + k == lab) g[k] -= gl;
        g[k] *= sc;
      }
      if (gr) *reinterpret_cast<float4*>(gr + v) = make_float4(g[0], g[1], g[2], g[3]);
      if (gr16) {
        uint2 o;
        o.x = (unsigned)eamd_f2bf(g[0]) | ((unsigned)eamd_f2bf(g[1]) << 16);
        o.y = (unsigned)eamd_f2bf(g[2]) | ((unsigned)eamd_f2bf(g[3]) << 16);
        *reinterpret_cast<uint2*>(gr16 + v) = o;
      }
    }
    return;
  }
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float g = expf(zr[v] + tot);
    if (v == blank) g -= gb;
    if (v == lab) g -= gl;
    if (gr) gr[v] = sc * g;
    if (gr16) gr16[v] = eamd_f2bf(sc * g);
  }
}

}  // namespace

extern "C" {

int eamd_joint_fwd(const float* enc, const float* dec, float* out, void* out_bf16, int B, int T, int U, int J, int act,
                   void* stream) {
  if (!enc || !dec || (!out && !out_bf16) || B <= 0 || T <= 0 || U <= 0 || J <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(joint_fwd_kernel, dim3(grid_for((long)B * T * U * J)), dim3(256), 0, (hipStream_t)stream, enc, dec,
                     out, (unsigned short*)out_bf16, B, T, U, J, act);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_joint_bwd(const float* dh, const float* enc, const float* dec, float* d_enc, float* d_dec, int B, int T, int U,
                   int J, int act, void* stream) {
  if (!dh || !enc || !dec || !d_enc || !d_dec || B <= 0 || T <= 0 || U <= 0 || J <= 0) return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(joint_bwd_enc_kernel, dim3(B * T), dim3(256), 0, s, dh, enc, dec, d_enc, B, T, U, J, act);
  EAMD_LAUNCH_CHECK();
  // few (b, u) rows (one utterance's chunk): the frames are cut into slices so that ~512 workgroups share the pass
  int slices = 1;
  while ((long)B * U * slices < 512 && T / (slices * 2) >= 16) slices *= 2;
  if (slices > 1 && eamd_zero_async(d_dec, (size_t)B * U * J * sizeof(float), s) != EAMD_OK) return EAMD_EINVAL;
  hipLaunchKernelGGL(joint_bwd_dec_kernel, dim3(B * U, slices), dim3(256), 0, s, dh, enc, dec, d_dec, B, T, U, J, act);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* workspace floats: 5 * B*T*U  (lse, lp_blank, lp_label, alpha, beta) */
int64_t eamd_rnnt_workspace(int B, int T, int U) { return (B <= 0 || T <= 0 || U <= 0) ? 0 : (int64_t)5 * B * T * U; }

int eamd_rnnt_loss(const float* logits, const int32_t* labels, const int32_t* tlens, const int32_t* ulens,
                   float* workspace, float* loss, float* grad, int B, int T, int U, int V, int blank,
                   const float* gscale_dev, float scale, void* stream) {
  if (!logits || !labels || !tlens || !ulens || !workspace || !loss || B <= 0 || T <= 0 || U <= 0 || V <= 1)
    return EAMD_EINVAL;
  if (blank < 0 || blank >= V) return EAMD_EINVAL;
  if ((size_t)2 * U * sizeof(float) > 64 * 1024) return EAMD_EUNSUPPORTED;
  if ((long)B * T * U > 2147483647L) return EAMD_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)B * T * U;
  float* lse = workspace;
  float* lpb = lse + n;
  float* lpl = lpb + n;
  float* alpha = lpl + n;
  float* beta = alpha + n;
  launch_lse_gather(logits, labels, lse, lpb, lpl, T, U, V, blank, 0L, n, s);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(rnnt_alpha_beta_kernel, dim3(2 * B), dim3(256), 2 * U * sizeof(float), s, lpb, lpl, tlens, ulens,
                     alpha, beta, loss, B, T, U);
  EAMD_LAUNCH_CHECK();
  if (grad) {
    hipLaunchKernelGGL(rnnt_grad_kernel, dim3((unsigned)n), dim3(256), 0, s, logits, grad, labels, lse, lpb, lpl, alpha, beta,
                       tlens, ulens, gscale_dev, scale, T, U, V, blank, 0L, (unsigned short*)nullptr);
    EAMD_LAUNCH_CHECK();
  }
  return EAMD_OK;
}

/* gradient pass alone, from the workspace a previous eamd_rnnt_loss call on the same logits filled (lets the caller
 * fold the upstream scalar, read from device memory, into this pass instead of rescaling 4*B*T*U*V bytes later) */
int eamd_rnnt_grad(const float* logits, const int32_t* labels, const int32_t* tlens, const int32_t* ulens,
                   const float* workspace, float* grad, int B, int T, int U, int V, int blank, const float* gscale_dev,
                   float scale, void* stream) {
  if (!logits || !labels || !tlens || !ulens || !workspace || !grad || B <= 0 || T <= 0 || U <= 0 || V <= 1)
    return EAMD_EINVAL;
  if (blank < 0 || blank >= V || (long)B * T * U > 2147483647L) return EAMD_EINVAL;
  const long n = (long)B * T * U;
  const float* lse = workspace;
  const float* lpb = lse + n;
  const float* lpl = lpb + n;
  const float* alpha = lpl + n;
  const float* beta = alpha + n;
  hipLaunchKernelGGL(rnnt_grad_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, logits, grad, labels, lse, lpb,
                     lpl, alpha, beta, tlens, ulens, gscale_dev, scale, T, U, V, blank, 0L, (unsigned short*)nullptr);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* ---- the same loss on lattice ROWS: the caller streams the joint logits through a buffer of `nrows` consecutive lattice
 * nodes at a time (node = (b*T + t)*U + u, rows node0 .. node0+nrows-1) instead of materialising [B,T,U,V];
 * `workspace` is the 5*B*T*U-float lattice workspace of eamd_rnnt_workspace.
 *   eamd_rnnt_node_stats : lse / log p(blank) / log p(label) of the rows into the workspace
 *   eamd_rnnt_alpha_beta : forward / backward variables and loss[b] from the workspace (all rows must be in)
 *   eamd_rnnt_node_grad  : d loss / d logits of the rows (fp32 and / or bf16 output; may alias the logits) */
int eamd_rnnt_node_stats(const float* logits_rows, const int32_t* labels, float* workspace, int64_t node0, int64_t nrows,
                         int B, int T, int U, int V, int blank, void* stream) {
  const long n = (long)B * T * U;
  if (!logits_rows || !labels || !workspace || node0 < 0 || nrows <= 0 || node0 + nrows > n || V <= 1 || blank < 0 ||
      blank >= V || nrows > 2147483647L)
    return EAMD_EINVAL;
  float* lse = workspace;
  launch_lse_gather(logits_rows, labels, lse, lse + n, lse + 2 * n, T, U, V, blank, (long)node0, (long)nrows, (hipStream_t)stream);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* eamd_rnnt_node_stats for logits that were never stored: part / zlab / zblank are what eamd_gemm's row-statistics epilogue
 * (epilogue 7, eamd_gemm_t.stats with col = the node's next label or -1, fix = blank) wrote for the nrows nodes
 * node0 .. node0 + nrows - 1; tiles_n = ceil(V / tile) of that launch */
int eamd_rnnt_node_stats_part(const float* part, const float* zlab, const float* zblank, float* workspace, int64_t node0,
                              int64_t nrows, int tiles_n, int B, int T, int U, void* stream) {
  if (!part || !zlab || !zblank || !workspace || B <= 0 || T <= 0 || U <= 0 || tiles_n <= 0 || node0 < 0 || nrows <= 0)
    return EAMD_EINVAL;
  const long n = (long)B * T * U;
  if (node0 + nrows > n) return EAMD_EINVAL;
  if (reinterpret_cast<uintptr_t>(part) & 7) return EAMD_EUNSUPPORTED;
  float* lse = workspace;
  hipLaunchKernelGGL(rnnt_stats_combine_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     zlab, zblank, lse, lse + n, lse + 2 * n, tiles_n, U, (long)node0, (long)nrows);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_rnnt_row_coef(const int32_t* labels, const int32_t* tlens, const int32_t* ulens, const float* workspace, float* rowc,
                       int32_t* col, int64_t node0, int64_t nrows, int B, int T, int U, void* stream) {
  if (!labels || !tlens || !ulens || !workspace || !rowc || !col || B <= 0 || T <= 0 || U <= 0 || node0 < 0 || nrows <= 0)
    return EAMD_EINVAL;
  const long n = (long)B * T * U;
  if (node0 + nrows > n) return EAMD_EINVAL;
  const float* lse = workspace;
  hipLaunchKernelGGL(rnnt_row_coef_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, lse,
                     lse + n, lse + 2 * n, lse + 3 * n, lse + 4 * n, tlens, ulens, rowc, col, T, U, (long)node0, (long)nrows);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_rnnt_alpha_beta(float* workspace, const int32_t* tlens, const int32_t* ulens, float* loss, int B, int T, int U,
                         void* stream) {
  if (!workspace || !tlens || !ulens || !loss || B <= 0 || T <= 0 || U <= 0) return EAMD_EINVAL;
  if ((size_t)2 * U * sizeof(float) > 64 * 1024) return EAMD_EUNSUPPORTED;
  const long n = (long)B * T * U;
  hipLaunchKernelGGL(rnnt_alpha_beta_kernel, dim3(2 * B), dim3(256), 2 * U * sizeof(float), (hipStream_t)stream,
                     workspace + n, workspace + 2 * n, tlens, ulens, workspace + 3 * n, workspace + 4 * n, loss, B, T, U);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_rnnt_node_grad(const float* logits_rows, float* grad_rows, void* grad_rows_bf16, const int32_t* labels,
                        const int32_t* tlens, const int32_t* ulens, const float* workspace, int64_t node0, int64_t nrows,
                        int B, int T, int U, int V, int blank, const float* gscale_dev, float scale, void* stream) {
  const long n = (long)B * T * U;
  if (!logits_rows || (!grad_rows && !grad_rows_bf16) || !labels || !tlens || !ulens || !workspace || node0 < 0 ||
      nrows <= 0 || node0 + nrows > n || V <= 1 || blank < 0 || blank >= V || nrows > 2147483647L)
    return EAMD_EINVAL;
  const float* lse = workspace;
  hipLaunchKernelGGL(rnnt_grad_kernel, dim3((unsigned)nrows), dim3(256), 0, (hipStream_t)stream, logits_rows, grad_rows, labels,
                     lse, lse + n, lse + 2 * n, lse + 3 * n, lse + 4 * n, tlens, ulens, gscale_dev, scale, T, U, V, blank,
                     (long)node0, (unsigned short*)grad_rows_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"
