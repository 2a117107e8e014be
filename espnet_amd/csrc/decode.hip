// Cached decoding of the Transformer decoder for a beam search, a few launches per layer instead of fourteen.
// reference: transformer/decoder.py:283-321 (forward_one_step), decoder_layer.py:81-134 (the cached step: only the newest position
// queries; the reference keeps every layer's OUTPUTS and re-projects keys / values of the whole prefix at every step).
//
// Here a layer keeps the KEYS and VALUES of the prefix instead (the same numbers: LayerNorm and the projections are row-wise),
// time-major in [Lcap, n, D] buffers that are only ever appended to, and a beam step's re-ordering of the hypotheses never moves
// them: slot_at[i][t] names the slot whose row at position t belongs to the history of the hypothesis now in slot i (a table of
// n x Lcap int32 per search, shared by all layers, re-ordered with the hypotheses).
//   eamd_linear_rows_ln_f32   y = alpha * act(LayerNorm(x) W^T + b) + R for M <= 16 rows: the pre-norm of a sub-block inside the
//                             product that follows it (norm1 + q/k/v, norm2 + q of the source attention, norm3 + w_1, after_norm
//                             + the output layer)
//   eamd_decode_self_attn     appends this step's k / v rows and attends the newest position over the prefix (one wave per
//                             (hypothesis, head), d_k = 64)
//   eamd_beam_slots           the slot table behind a beam step's selection
#include <stdlib.h>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

// one wave per output column (rowops.hip: linear_rows_f32_kernel), four columns per workgroup; K <= 1024: the M rows fit LDS whole.
// (tried: no staging - every wave reads its pieces of all rows to registers and repeats the LayerNorm there: 10.5 - 11.9 us
// against 7.3 - 7.8 for this form, the output layer 19.9 against 10.0; the repeated statistics cost more than the staging pass)
__global__ __launch_bounds__(256) void linear_rows_ln_f32_kernel(const float* __restrict__ x, const float* __restrict__ gam,
                                                                 const float* __restrict__ bet, float eps,
                                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                                 const float* __restrict__ R, float* __restrict__ y, int M, int N,
                                                                 int K, int act, float alpha, long ldx, long ldr, long ldy) {
  // the M rows are staged once per workgroup in LDS (all loads of the pass in flight together), wave w normalises rows w, w + 4, ...
  // in place (mean, then the centred sum of squares: layernorm_fwd's arithmetic), then every wave multiplies them with its column
  extern __shared__ __attribute__((aligned(16))) float xs_ln[];        // [M][K]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = min(blockIdx.x * 4 + w, N - 1);
  const bool store = blockIdx.x * 4 + w < N;
  const float* wr = W + (long)n * K;
  float4 w4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = lane * 4 + 256 * q;
    w4[q] = k < K ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int c4 = K >> 2;
  for (int idx = threadIdx.x; idx < M * c4; idx += 256) {
    const int m = idx / c4, c = idx - m * c4;
    *reinterpret_cast<float4*>(&xs_ln[m * K + c * 4]) = *reinterpret_cast<const float4*>(x + (long)m * ldx + c * 4);
  }
  __syncthreads();
  for (int m = w; m < M; m += 4) {
    float4 x4[4];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lane * 4 + 256 * q;
      x4[q] = k < K ? *reinterpret_cast<const float4*>(&xs_ln[m * K + k]) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (x4[q].x + x4[q].y) + (x4[q].z + x4[q].w);
    }
    const float mean = wave_sum(s) / K;
    float c = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lane * 4 + 256 * q;
      if (k < K) {
        const float a0 = x4[q].x - mean, a1 = x4[q].y - mean, a2 = x4[q].z - mean, a3 = x4[q].w - mean;
        c += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    }
    const float rstd = rsqrtf(wave_sum(c) / K + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lane * 4 + 256 * q;
      if (k < K) {
        const float4 g = *reinterpret_cast<const float4*>(gam + k), b = *reinterpret_cast<const float4*>(bet + k);
        *reinterpret_cast<float4*>(&xs_ln[m * K + k]) =
            make_float4((x4[q].x - mean) * rstd * g.x + b.x, (x4[q].y - mean) * rstd * g.y + b.y,
                        (x4[q].z - mean) * rstd * g.z + b.z, (x4[q].w - mean) * rstd * g.w + b.w);
      }
    }
  }
  __syncthreads();
  float mine = 0.f;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if (m < M) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane * 4 + 256 * q;
        if (k < K) {
          const float4 v = *reinterpret_cast<const float4*>(&xs_ln[m * K + k]);
          acc = fmaf(v.x, w4[q].x, fmaf(v.y, w4[q].y, fmaf(v.z, w4[q].z, fmaf(v.w, w4[q].w, acc))));
        }
      }
      const float r = wave_sum(acc);
      if (lane == m) mine = r;
    }
  }
  if (lane < M && store) {
    float v = mine + (bias ? bias[n] : 0.f);
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = eamd_swish(v);
    v *= alpha;
    if (R) v += R[(long)lane * ldr + n];
    y[(long)lane * ldy + n] = v;
  }
}

// The same for a few hundred rows (a batched beam search steps utterances x beam hypotheses: M = 320) with K <= 256: every WAVE owns
// a 16 x 16 output tile (rowops.hip: linear_mfma16_f32_kernel - operands straight from global memory into
// v_mfma_f32_16x16x4_f32); its 16 input rows are whole in the registers of the wave (lane = row x k-group of four), so their
// LayerNorm is two four-lane sums per row in front of the products - 19 LayerNorm launches per step go.
typedef float lnm_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void linear_mfma16_ln_f32_kernel(const float* __restrict__ x, const float* __restrict__ gam,
                                                                   const float* __restrict__ bet, float eps,
                                                                   const float* __restrict__ W, const float* __restrict__ bias,
                                                                   const float* __restrict__ R, float* __restrict__ y, int M, int N,
                                                                   int K, int act, float alpha, long ldx, long ldr, long ldy) {
  __shared__ __attribute__((aligned(16))) float gs[256], bs[256];        // gamma / beta once per workgroup (read from global inside
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;      // the product loop each pair waited a memory round trip)
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  const int m0 = blockIdx.y * 16;
  if (threadIdx.x * 4 < K) {
    *reinterpret_cast<float4*>(&gs[threadIdx.x * 4]) = *reinterpret_cast<const float4*>(gam + threadIdx.x * 4);
    *reinterpret_cast<float4*>(&bs[threadIdx.x * 4]) = *reinterpret_cast<const float4*>(bet + threadIdx.x * 4);
  }
  const float* xr = x + (long)min(m0 + fr, M - 1) * ldx + fq * 4;        // clamped rows / columns are never stored
  const float* wr = W + (long)min(n0 + fr, N - 1) * K + fq * 4;
  constexpr int U = 16;                                                   // K <= 256: 16-k chunks, all requested together
  float4 xa[U], wb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool in = 16 * u < K;                                           // (K % 16 == 0: checked on the host)
    xa[u] = in ? *reinterpret_cast<const float4*>(xr + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
    wb[u] = in ? *reinterpret_cast<const float4*>(wr + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) s += (xa[u].x + xa[u].y) + (xa[u].z + xa[u].w);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  const float mean = s / K;
  float c = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (16 * u < K) {
      const float a0 = xa[u].x - mean, a1 = xa[u].y - mean, a2 = xa[u].z - mean, a3 = xa[u].w - mean;
      c += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  c += __shfl_xor(c, 16, 64);
  c += __shfl_xor(c, 32, 64);
  const float rstd = rsqrtf(c / K + eps);
  __syncthreads();
  if (n0 >= N) return;
  lnm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (16 * u < K) {
      const float4 g = *reinterpret_cast<const float4*>(&gs[fq * 4 + 16 * u]), b = *reinterpret_cast<const float4*>(&bs[fq * 4 + 16 * u]);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((xa[u].x - mean) * rstd * g.x + b.x, wb[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((xa[u].y - mean) * rstd * g.y + b.y, wb[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((xa[u].z - mean) * rstd * g.z + b.z, wb[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32((xa[u].w - mean) * rstd * g.w + b.w, wb[u].w, acc, 0, 0, 0);
    }
  }
  const int n = n0 + fr;
  if (n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + fq * 4 + r;
      if (m < M) {
        float v = acc[r] + bv;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = eamd_swish(v);
        v *= alpha;
        if (R) v += R[(long)m * ldr + n];
        y[(long)m * ldy + n] = v;
      }
    }
  }
}

// one wave per (hypothesis slot, head); lane = channel of the head (d_k = 64).
//   qkv   [n, ldq]: q | k | v of the newest position (columns 0, D, 2 D)
//   Kc,Vc [Lcap, n, D]: rows [pos][slot] are written here, rows t < pos are read through slot_at[slot][t]
//   ctx   [n, D]
__global__ __launch_bounds__(64) void decode_self_attn_kernel(const float* __restrict__ qkv, long ldq, float* __restrict__ Kc,
                                                              float* __restrict__ Vc, const int* __restrict__ slot_at, int Lcap,
                                                              int pos, int n, int D, float* __restrict__ ctx, float scale,
                                                              const int* __restrict__ pos_dev) {
  extern __shared__ float sc[];                 // [pos + 1] scores, then probabilities; ints of the slots behind them
  if (pos_dev) pos = min(max(pos_dev[0] + pos, 0), Lcap - 1);      // the step index lives on the device (one graph for every step)
  int* sl = reinterpret_cast<int*>(sc + Lcap);
  const int h = blockIdx.x, row = blockIdx.y, lane = threadIdx.x;
  const float* qr = qkv + (long)row * ldq + h * 64;
  const float kq = qr[D + lane], vq = qr[2 * D + lane];
  Kc[((long)pos * n + row) * D + h * 64 + lane] = kq;
  Vc[((long)pos * n + row) * D + h * 64 + lane] = vq;
  float4 q4[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) q4[i] = *reinterpret_cast<const float4*>(qr + 4 * i);      // the whole query in every lane
  // scores of the prefix: lane = position inside a trip of 64 (each position's key is 256 contiguous bytes)
  float mx = -INFINITY;
  for (int t0 = 0; t0 < pos; t0 += 64) {
    const int t = t0 + lane;
    if (t < pos) {
      int s = slot_at[(long)row * Lcap + t];
      s = min(max(s, 0), n - 1);                                     // a slot outside the beam is never turned into an address
      const float4* kp = reinterpret_cast<const float4*>(Kc + ((long)t * n + s) * D + h * 64);
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 kv = kp[i];
        a = fmaf(q4[i].x, kv.x, fmaf(q4[i].y, kv.y, fmaf(q4[i].z, kv.z, fmaf(q4[i].w, kv.w, a))));
      }
      a *= scale;
      sc[t] = a;
      sl[t] = s;
      mx = fmaxf(mx, a);
    }
  }
  const float snew = wave_sum(qr[lane] * kq) * scale;                  // the newest position attends to itself from registers
  mx = fmaxf(wave_max(mx), snew);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float den = 0.f;
  for (int t0 = 0; t0 < pos; t0 += 64) {
    const int t = t0 + lane;
    if (t < pos) {
      const float e = __expf(sc[t] - mx);
      sc[t] = e;
      den += e;
    }
  }
  const float enew = __expf(snew - mx);
  den = wave_sum(den) + enew;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // context: lane = channel; four positions' value rows in flight
  float acc = enew * vq;
  int t = 0;
  for (; t + 4 <= pos; t += 4) {
    float p[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      p[u] = sc[t + u];
      v[u] = Vc[((long)(t + u) * n + sl[t + u]) * D + h * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fmaf(p[u], v[u], acc);
  }
  for (; t < pos; ++t) acc = fmaf(sc[t], Vc[((long)t * n + sl[t]) * D + h * 64 + lane], acc);
  ctx[(long)row * D + h * 64 + lane] = acc / den;
}

// slot_out[i][t] = slot_in[hyp[i]][t] for t < pos, slot_out[i][pos] = hyp[i]: the history of the hypothesis selected into slot i
__global__ __launch_bounds__(256) void beam_slots_kernel(const int* __restrict__ slot_in, int* __restrict__ slot_out,
                                                         const long long* __restrict__ hyp, int n, int Lcap, int pos,
                                                         const int* __restrict__ pos_dev) {
  if (pos_dev) pos = min(max(pos_dev[0] + pos, 0), Lcap - 1);
  const int i = blockIdx.x;
  long long p = hyp[i];
  p = p < 0 ? 0 : (p >= n ? n - 1 : p);
  for (int t = threadIdx.x; t <= pos && t < Lcap; t += blockDim.x)
    slot_out[(long)i * Lcap + t] = t < pos ? slot_in[p * Lcap + t] : (int)p;
}

}  // namespace

extern "C" {

int eamd_linear_rows_ln_f32(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias,
                            const float* R, float* y, int M, int N, int K, int act, float alpha, int64_t ldx, int64_t ldr,
                            int64_t ldy, void* stream) {
  if (!x || !gamma || !beta || !W || !y || M <= 0 || N <= 0 || K <= 0 || act < 0 || ldx < 0 || ldr < 0 || ldy < 0) return EAMD_EINVAL;
  if (ldx == 0) ldx = K;
  if (ldr == 0) ldr = N;
  if (ldy == 0) ldy = N;
  if (ldx < K || ldr < N || ldy < N) return EAMD_EINVAL;
  if (K > 1024 || K % 4 != 0 || ldx % 4 != 0 || act > 2) return EAMD_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta) & 15) return EAMD_EUNSUPPORTED;
  if (M > 16) {                                            // blocks of 16 rows on the matrix cores (the input rows of a wave are whole in its registers)
    if (M > 1024 || K > 256 || K % 16 != 0) return EAMD_EUNSUPPORTED;
    hipLaunchKernelGGL(linear_mfma16_ln_f32_kernel, dim3((N + 63) / 64, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                       eps, W, bias, R, y, M, N, K, act, alpha, (long)ldx, (long)ldr, (long)ldy);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(linear_rows_ln_f32_kernel, dim3((N + 3) / 4), dim3(256), (size_t)M * K * sizeof(float), (hipStream_t)stream, x,
                     gamma, beta, eps, W, bias, R,
                     y, M, N, K, act, alpha, (long)ldx, (long)ldr, (long)ldy);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_decode_self_attn(const float* qkv, int64_t ldq, float* kcache, float* vcache, const int32_t* slot_at, int Lcap, int pos,
                          int n, int H, int D, float* ctx, void* stream) {
  return eamd_decode_self_attn_dyn(qkv, ldq, kcache, vcache, slot_at, Lcap, pos, nullptr, n, H, D, ctx, stream);
}

int eamd_decode_self_attn_dyn(const float* qkv, int64_t ldq, float* kcache, float* vcache, const int32_t* slot_at, int Lcap, int pos,
                              const int32_t* pos_dev, int n, int H, int D, float* ctx, void* stream) {
  if (!qkv || !kcache || !vcache || !slot_at || !ctx || n <= 0 || H <= 0 || D <= 0 || Lcap <= 0 || (!pos_dev && pos < 0)) return EAMD_EINVAL;
  if ((!pos_dev && pos >= Lcap) || ldq < 3L * D) return EAMD_EINVAL;
  if (D != H * 64 || Lcap > 4096 || ldq % 4 != 0 || (((uintptr_t)qkv | (uintptr_t)kcache | (uintptr_t)vcache) & 15)) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(decode_self_attn_kernel, dim3(H, n), dim3(64), (size_t)Lcap * 8, (hipStream_t)stream, qkv, (long)ldq, kcache,
                     vcache, slot_at, Lcap, pos, n, D, ctx, 0.125f, pos_dev);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_beam_slots(const int32_t* slot_in, int32_t* slot_out, const int64_t* hyp, int n, int Lcap, int pos, void* stream) {
  return eamd_beam_slots_dyn(slot_in, slot_out, hyp, n, Lcap, pos, nullptr, stream);
}

int eamd_beam_slots_dyn(const int32_t* slot_in, int32_t* slot_out, const int64_t* hyp, int n, int Lcap, int pos, const int32_t* pos_dev,
                        void* stream) {
  if (!slot_in || !slot_out || !hyp || n <= 0 || Lcap <= 0 || (!pos_dev && (pos < 0 || pos >= Lcap))) return EAMD_EINVAL;
  hipLaunchKernelGGL(beam_slots_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, slot_in, slot_out, (const long long*)hyp, n, Lcap,
                     pos, pos_dev);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"

// ---- the selection of a beam step on the pre-beam candidates (reference: beam_search.py:296-334, :199-226) -------------------------
// BeamSearch drops every token outside the pre-beam (weighted_scores[:] = -inf; weighted_scores[ids] = tmp), so the `beam` best
// continuations of an utterance are among its beam x P candidates: one workgroup per utterance forms their scores in the
// reference's order of operations - (sum_k w_k logp_k)[token] + w_ctc (psi - s_prev), + the hypothesis's running score - and picks
// the best `beam` (value descending, ties by ascending slot * V + token: torch.topk's order on the flattened [beam, V] scores).
// Replaces, per step: a fill of [n, V] with -inf, a gather, a scatter, the running-score add and two top-k passes.
namespace {
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                           const float* __restrict__ l2, const float* __restrict__ l3, float w0,
                                                           float w1, float w2, float w3, float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    // (separately rounded products and sums, as the tensor expressions `weighted += w * logp` round them: the empty asm keeps
    // hipcc from fusing a * b + c)
    float4 a = reinterpret_cast<const float4*>(l0)[i];
    float4 v = make_float4(w0 * a.x, w0 * a.y, w0 * a.z, w0 * a.w);
#define EAMD_WS_ADD(L, W)                                                                                              \
    if (L) {                                                                                                           \
      a = reinterpret_cast<const float4*>(L)[i];                                                                       \
      float p0 = W * a.x, p1 = W * a.y, p2 = W * a.z, p3 = W * a.w;                                                     \
      asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));                                                      \
      v.x += p0; v.y += p1; v.z += p2; v.w += p3;                                                                      \
    }
    EAMD_WS_ADD(l1, w1)
    EAMD_WS_ADD(l2, w2)
    EAMD_WS_ADD(l3, w3)
#undef EAMD_WS_ADD
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

__device__ __forceinline__ bool sel_before(float va, long ia, float vb, long ib) { return va > vb || (va == vb && ia < ib); }

__global__ __launch_bounds__(256) void beam_select_kernel(const float* __restrict__ pre, const long long* __restrict__ ids,
                                                          const float* __restrict__ psi, const float* __restrict__ c_s,
                                                          const float* __restrict__ hyp, float w_ctc, int beam, int P, int V,
                                                          float* __restrict__ c_local, float* __restrict__ top_s,
                                                          long long* __restrict__ top_i) {
  __shared__ float sv[4];
  __shared__ long si[4];
  __shared__ float wv;
  __shared__ long wi;
  const int u = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int C = beam * P;
  constexpr int RMAX = 4;                         // beam * P <= 1024 candidates (host check)
  float val[RMAX];
  long key[RMAX];
#pragma unroll
  for (int q = 0; q < RMAX; ++q) {
    const int c = t + 256 * q;
    val[q] = -INFINITY;
    key[q] = 0x7ffffffffffffffeL;
    if (c < C) {
      const int slot = c / P, j = c - slot * P;
      const long h = (long)u * beam + slot;
      long long tok = ids[h * P + j];
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);                 // (a token outside the vocabulary never becomes an address)
      const float cl = __fsub_rn(psi[h * P + j], c_s[h]);
      c_local[h * P + j] = cl;
      float prod = w_ctc * cl;
      asm volatile("" : "+v"(prod));                    // keeps the product separately rounded, as the tensor expressions round
      float v = pre[h * V + tok] + prod;               // it (hipcc fuses a * b + c whatever the contraction pragma says)
      v = v + hyp[h];
      val[q] = (v != v) ? -INFINITY : v;
      key[q] = (long)slot * V + tok;
    }
  }
  float pv = INFINITY;
  long pi = -1;
  for (int r = 0; r < beam; ++r) {
    float bv = -INFINITY;
    long bi = 0x7fffffffffffffffL;
#pragma unroll
    for (int q = 0; q < RMAX; ++q)
      if (t + 256 * q < C && sel_before(pv, pi, val[q], key[q]) && sel_before(val[q], key[q], bv, bi)) { bv = val[q]; bi = key[q]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float ov = __shfl_xor(bv, m);
      const long oi = __shfl_xor(bi, m);
      if (sel_before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[w] = bv; si[w] = bi; }
    __syncthreads();
    if (t == 0) {
      float fv = sv[0];
      long fi = si[0];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (sel_before(sv[q], si[q], fv, fi)) { fv = sv[q]; fi = si[q]; }
      if (fi == 0x7fffffffffffffffL) { fv = -INFINITY; fi = 0; }     // fewer candidates than `beam`: a dead slot
      wv = fv; wi = fi;
      top_s[(long)u * beam + r] = fv;
      top_i[(long)u * beam + r] = fi;
    }
    __syncthreads();
    pv = wv; pi = wi;
    if (pi == 0 && pv == -INFINITY) pi = -1;       // (after a dead slot every further rank is dead too)
  }
}

// The selection above and the bookkeeping behind it (rowops.hip: beam_finish_kernel) in ONE launch per step, one workgroup per
// utterance: the `beam` winners stay in LDS; thread s then does slot s's scalars (which hypothesis / token, the scores carried
// along, the CTC prefix score of the new prefix = psi at the token's candidate position, the end tests, the next running score)
// and all threads copy the prefixes.  Same arithmetic as the two kernels; three launches (select, finish, the psi gather) and
// the int32 cast of the newest tokens become one.
struct BeamStepArgs {
  const float* logp[4];
  const float* pre; const long long* ids; const float* psi; const float* c_s; const float* hyp; const long long* maxlen;
  const float* sc_in; const long long* yseq_in;
  float* c_local; float* sc_out; long long* yseq_out; float* hyp_out; long long* hyp_i; long long* tok_i; int* tok32; float* cs_out;
  float* rec;
  float w_ctc;
  int n, beam, P, V, W, L, step, eos, ns, nf;
  // one graph for every step: the step index read from the device (L = step + 1), the log row written into a ring of `ring` rows
  // ([ring][n][3 + ns + W]: slot step % ring), step + 1 left in step_out for the next replay
  const int* step_dev; int* step_out; int ring;
  // the cached decoder's slot table behind the selection (beam_slots_kernel's work, for the one scorer that keeps such a table):
  // slot_out[s][t] = slot_in[hypothesis of s][t] for t < step, the hypothesis's slot itself at t = step
  const int* slot_in; int* slot_out; int Lcap;
};
__device__ __forceinline__ unsigned sel_bits(float v) {      // order-preserving bits (NaN was made -inf; -0 ranks as +0)
  if (v == 0.f) v = 0.f;
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(256) void beam_step_kernel(const BeamStepArgs a) {
  // (locals, not fields of the by-value argument struct: writing to `a` moved it to private memory - 7.5 -> 12.4 us)
  const int step_ = a.step_dev ? min(max(a.step_dev[0], 0), a.W - 2) : a.step;
  const int L_ = a.step_dev ? step_ + 1 : a.L;
  float* const rec_ = a.rec + ((a.step_dev && a.ring > 0) ? (long)(step_ % a.ring) * a.n * (3 + a.ns + a.W) : 0L);
  if (a.step_dev && a.step_out && blockIdx.x == 0 && threadIdx.x == 0) a.step_out[0] = step_ + 1;
  // the winners by counting: a candidate's rank is the number of candidates that come before it (value descending, then
  // slot * V + token ascending - one unsigned compare on (value bits, ~index)); ranks below `beam` are the selection
  __shared__ __attribute__((aligned(16))) unsigned long long ckey[1024];
  __shared__ float win_s[64];
  __shared__ long win_i[64];
  __shared__ long slot_h[64];
  __shared__ long slot_tok[64];
  const int u = blockIdx.x, t = threadIdx.x;
  const int beam = a.beam, P = a.P, V = a.V;
  const int C = beam * P;
  constexpr int RMAX = 4;                         // beam * P <= 1024 candidates (host check)
  float val[RMAX];
  unsigned long long key[RMAX];
#pragma unroll
  for (int q = 0; q < RMAX; ++q) {
    const int c = t + 256 * q;
    val[q] = -INFINITY;
    key[q] = 0;
    if (c < C) {
      const int slot = c / P, j = c - slot * P;
      const long h = (long)u * beam + slot;
      long long tok = a.ids[h * P + j];
      const bool none = tok < 0;                                       // no candidate in this column: ranks last, selects nothing
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
      const float cl = __fsub_rn(a.psi[h * P + j], a.c_s[h]);
      a.c_local[h * P + j] = cl;
      float prod = a.w_ctc * cl;
      asm volatile("" : "+v"(prod));
      float v = a.pre[h * V + tok] + prod;
      v = v + a.hyp[h];
      val[q] = (v != v || none) ? -INFINITY : v;
      // (its index lies beyond the utterance's beam x V continuations: distinct from every real key, a dead slot if it ever wins)
      key[q] = ((unsigned long long)sel_bits(val[q]) << 32) | (0xFFFFFFFFu - (unsigned)(none ? beam * V + c : slot * V + (int)tok));
      ckey[c] = key[q];
    }
  }
  if (t < beam) { win_s[t] = -INFINITY; win_i[t] = 0; }            // fewer candidates than `beam`: dead slots
  if (t == 0 && (C & 1)) ckey[C] = 0;                              // (the pair-wise walk below reads one past an odd count)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RMAX; ++q) {
    if (t + 256 * q < C) {
      int r = 0;
      for (int j = 0; j < C; j += 2) {
        const ulonglong2 o = *reinterpret_cast<const ulonglong2*>(&ckey[j]);
        r += (o.x > key[q]) + (o.y > key[q]);
      }
      if (r < beam) { win_s[r] = val[q]; win_i[r] = (long)(0xFFFFFFFFu - (unsigned)key[q]); }
    }
  }
  __syncthreads();
  // ---- the bookkeeping of this utterance's slots (c_local of the block above is visible: the barriers of the rounds) ----
  const int RW = 3 + a.ns + a.W;
  if (t < beam) {
    const long s = (long)u * beam + t;
    long ti = win_i[t];
    float ts = win_s[t];
    if (ti < 0 || ti >= (long)beam * V) { ti = 0; ts = -INFINITY; }
    const long h = (long)u * beam + ti / V;
    const long tok = ti % V;
    a.hyp_i[s] = h; a.tok_i[s] = tok; a.tok32[s] = (int)tok;
    slot_h[t] = h; slot_tok[t] = tok;
    float* rec = rec_ + s * RW;
    rec[0] = (float)step_; rec[1] = ts; rec[2] = (float)tok;
    for (int j = 0; j < a.nf; ++j) {
      const float v = a.sc_in[(long)j * a.n + h] + a.logp[j][h * V + tok];
      a.sc_out[(long)j * a.n + s] = v;
      rec[3 + j] = v;
    }
    long p = 0;                                            // position of the token among the hypothesis's candidates
    for (int q = 0; q < P; ++q)
      if (a.ids[h * P + q] == tok) { p = q; break; }
    if (a.ns > a.nf) {
      const float v = a.sc_in[(long)a.nf * a.n + h] + a.c_local[h * P + p];
      a.sc_out[(long)a.nf * a.n + s] = v;
      rec[3 + a.nf] = v;
    }
    a.cs_out[s] = a.psi[h * P + p];
    const bool finite = isfinite(ts);
    const bool at_cap = a.maxlen[u] <= step_ + 1;
    const bool done = finite && (tok == a.eos || at_cap);
    a.hyp_out[s] = (done || !finite) ? -INFINITY : ts;
  }
  __syncthreads();
  for (int idx = t; idx < beam * a.W; idx += 256) {
    const int sl = idx / a.W, wq = idx - sl * a.W;
    const long s = (long)u * beam + sl;
    const long long tk = wq == L_ ? slot_tok[sl] : a.yseq_in[slot_h[sl] * a.W + wq];
    a.yseq_out[s * a.W + wq] = tk;
    rec_[s * RW + 3 + a.ns + wq] = (float)tk;
  }
  if (a.slot_in) {
    const int pos = min(step_, a.Lcap - 1);
    for (int idx = t; idx < beam * (pos + 1); idx += 256) {
      const int sl = idx / (pos + 1), tt = idx - sl * (pos + 1);
      const long s = (long)u * beam + sl;
      const long h = min(max(slot_h[sl], 0L), (long)a.n - 1);
      a.slot_out[s * a.Lcap + tt] = tt < pos ? a.slot_in[h * a.Lcap + tt] : (int)h;
    }
  }
}
}  // namespace

extern "C" {

int eamd_beam_step(const float* pre, const int64_t* ids, const float* psi, const float* c_s, const float* hyp, float w_ctc, int nutt,
                   int beam, int P, int V, int W, int L, int step, int eos, const int64_t* maxlen, int ns, int nf, const float* sc_in,
                   const float* const* logps, const int64_t* yseq_in, float* c_local, float* sc_out, int64_t* yseq_out, float* hyp_out,
                   int64_t* hyp_i, int64_t* tok_i, int32_t* tok32, float* cs_out, float* rec, void* stream) {
  return eamd_beam_step_dyn(pre, ids, psi, c_s, hyp, w_ctc, nutt, beam, P, V, W, L, step, eos, maxlen, ns, nf, sc_in, logps, yseq_in, c_local,
                            sc_out, yseq_out, hyp_out, hyp_i, tok_i, tok32, cs_out, rec, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

int eamd_beam_step_dyn(const float* pre, const int64_t* ids, const float* psi, const float* c_s, const float* hyp, float w_ctc, int nutt,
                       int beam, int P, int V, int W, int L, int step, int eos, const int64_t* maxlen, int ns, int nf, const float* sc_in,
                       const float* const* logps, const int64_t* yseq_in, float* c_local, float* sc_out, int64_t* yseq_out, float* hyp_out,
                       int64_t* hyp_i, int64_t* tok_i, int32_t* tok32, float* cs_out, float* rec, const int32_t* step_dev, int32_t* step_out,
                       int ring, const int32_t* slot_in, int32_t* slot_out, int Lcap, void* stream) {
  if (!pre || !ids || !psi || !c_s || !hyp || !maxlen || !sc_in || !yseq_in || !c_local || !sc_out || !yseq_out || !hyp_out || !hyp_i ||
      !tok_i || !tok32 || !cs_out || !rec)
    return EAMD_EINVAL;
  if (step_dev) { L = 0; step = 0; }
  if (nutt <= 0 || beam <= 0 || P <= 0 || V <= 0 || W < 2 || L < 0 || L >= W || ns < 1 || nf < 0 || nf > 4 || ns != nf + 1 || ring < 0) return EAMD_EINVAL;
  if ((nf > 0 && !logps) || (long)nutt * beam > 0x7fffffffL) return EAMD_EINVAL;
  if ((long)beam * P > 1023 || beam > 64 || (long)beam * V + 1024 > 0x7fffffffL) return EAMD_EUNSUPPORTED;
  BeamStepArgs a;
  for (int j = 0; j < 4; ++j) a.logp[j] = j < nf ? logps[j] : nullptr;
  for (int j = 0; j < nf; ++j) if (!a.logp[j]) return EAMD_EINVAL;
  a.pre = pre; a.ids = (const long long*)ids; a.psi = psi; a.c_s = c_s; a.hyp = hyp; a.maxlen = (const long long*)maxlen; a.sc_in = sc_in;
  a.yseq_in = (const long long*)yseq_in; a.c_local = c_local; a.sc_out = sc_out; a.yseq_out = (long long*)yseq_out; a.hyp_out = hyp_out;
  a.hyp_i = (long long*)hyp_i; a.tok_i = (long long*)tok_i; a.tok32 = tok32; a.cs_out = cs_out; a.rec = rec; a.w_ctc = w_ctc;
  a.n = nutt * beam; a.beam = beam; a.P = P; a.V = V; a.W = W; a.L = L; a.step = step; a.eos = eos; a.ns = ns; a.nf = nf;
  a.step_dev = step_dev; a.step_out = step_out; a.ring = ring;
  if ((slot_in != nullptr) != (slot_out != nullptr) || (slot_in && Lcap <= 0)) return EAMD_EINVAL;
  a.slot_in = slot_in; a.slot_out = slot_out; a.Lcap = Lcap;
  hipLaunchKernelGGL(beam_step_kernel, dim3(nutt), dim3(256), 0, (hipStream_t)stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_weighted_sum(const float* const* logps, const float* weights, int nf, int64_t numel, float* out, void* stream) {
  if (!logps || !weights || !out || nf < 1 || nf > 4 || numel <= 0) return EAMD_EINVAL;
  for (int i = 0; i < nf; ++i)
    if (!logps[i] || ((uintptr_t)logps[i] & 15)) return logps[i] ? EAMD_EUNSUPPORTED : EAMD_EINVAL;
  if (numel % 4 != 0 || ((uintptr_t)out & 15)) return EAMD_EUNSUPPORTED;
  const long n4 = numel / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logps[0], nf > 1 ? logps[1] : nullptr,
                     nf > 2 ? logps[2] : nullptr, nf > 3 ? logps[3] : nullptr, weights[0], nf > 1 ? weights[1] : 0.f,
                     nf > 2 ? weights[2] : 0.f, nf > 3 ? weights[3] : 0.f, out, n4);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_beam_select(const float* pre, const int64_t* ids, const float* psi, const float* c_s, const float* hyp, float w_ctc,
                     int nutt, int beam, int P, int V, float* c_local, float* top_s, int64_t* top_i, void* stream) {
  if (!pre || !ids || !psi || !c_s || !hyp || !c_local || !top_s || !top_i || nutt <= 0 || beam <= 0 || P <= 0 || V <= 0)
    return EAMD_EINVAL;
  if ((long)beam * P > 1024) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(beam_select_kernel, dim3(nutt), dim3(256), 0, (hipStream_t)stream, pre, (const long long*)ids, psi, c_s, hyp,
                     w_ctc, beam, P, V, c_local, top_s, (long long*)top_i);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // extern "C"

// ---- source attention of a beam step: g hypotheses of an utterance over ITS memory (decoder_layer.py:109-121) -------------------
// One wave per (hypothesis, head), lane = key inside a trip of 64 for the scores (a key row of one head is 256 contiguous bytes),
// lane = channel for the context.  Keys / values are column blocks of the decoder stack's ONE projection of the memory
// (F_.SharedProj: row stride ldkv).  The training kernel (attn_f32_fwd_kernel: 64 queries x all keys per workgroup, four
// workgroups for one utterance) took 19 us per layer here; this one is bound by the latency of ~250 keys.
namespace {
__global__ __launch_bounds__(256) void decode_src_attn_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ Km,
                                                              const float* __restrict__ Vm, long ldkv, const unsigned char* __restrict__ mask,
                                                              int g, int T, int D, float* __restrict__ ctx, float scale) {
  // four waves per (hypothesis, head): thread = key for the scores (a trip covers 256 keys), wave w walks keys w, w + 4, ... with
  // eight value rows in flight for the context (one wave walking 249 keys four at a time took 25 us: 62 dependent round trips)
  extern __shared__ float sc[];                 // [T] scores, then probabilities; [4][64] partial contexts; 8 reduction slots
  float* part = sc + T;
  float* red = part + 256;
  const int h = blockIdx.x, row = blockIdx.y, t_ = threadIdx.x, lane = t_ & 63, w = t_ >> 6;
  const int u = row / g;                        // utterance of this hypothesis
  const float* qr = q + (long)row * ldq + h * 64;
  const float* kb = Km + (long)u * T * ldkv + h * 64;
  const float* vb = Vm + (long)u * T * ldkv + h * 64;
  const unsigned char* mk = mask ? mask + (long)u * T : nullptr;
  float4 q4[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) q4[i] = *reinterpret_cast<const float4*>(qr + 4 * i);
  float mx = -INFINITY;
  for (int t = t_; t < T; t += 256) {
    const float4* kp = reinterpret_cast<const float4*>(kb + (long)t * ldkv);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 kv = kp[i];
      a = fmaf(q4[i].x, kv.x, fmaf(q4[i].y, kv.y, fmaf(q4[i].z, kv.z, fmaf(q4[i].w, kv.w, a))));
    }
    a = (mk && !mk[t]) ? -INFINITY : a * scale;          // a masked frame: finfo.min -> softmax -> 0 (attention.py:80-88)
    sc[t] = a;
    mx = fmaxf(mx, a);
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float den = 0.f;
  for (int t = t_; t < T; t += 256) {
    const float e = (mx == -INFINITY) ? 0.f : __expf(sc[t] - mx);
    sc[t] = e;
    den += e;
  }
  den = wave_sum(den);
  if (lane == 0) red[4 + w] = den;
  __syncthreads();
  den = (red[4] + red[5]) + (red[6] + red[7]);
  float acc = 0.f;
  for (int t0 = w; t0 < T; t0 += 256) {          // the value rows of 64 keys per wave requested together (see the grouped kernel)
    float v[64];
    const int tw = __builtin_amdgcn_readfirstlane(t0);
#pragma unroll
    for (int k = 0; k < 64; ++k) v[k] = (vb + (long)min(tw + 4 * k, T - 1) * ldkv)[lane];        // (unconditional, clamped: see below)
#pragma unroll
    for (int k = 0; k < 64; ++k) acc = fmaf(tw + 4 * k < T ? sc[tw + 4 * k] : 0.f, v[k], acc);
  }
  part[w * 64 + lane] = acc;
  __syncthreads();
  if (w == 0) {
    const float s = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
    ctx[(long)row * D + h * 64 + lane] = den > 0.f ? s / den : 0.f;        // every frame masked: zeros (attention.py:84-88)
  }
}
}  // namespace

// Many utterances per search (32 x beam 10 hypotheses): one workgroup per (utterance, head) serves ALL g hypotheses of the utterance -
// a key row and a value row are read once and meet the g queries / probability rows from LDS (the kernel above re-reads the
// utterance's keys and values once per hypothesis: 431 -> 360 utt/s when it was tried at this size; the training kernel,
// 64 queries per workgroup, took 23.5 us per layer here).  Same score and context arithmetic; the softmax sums run per wave.
namespace {
constexpr int SRC_GQ = 16;                       // hypotheses per utterance served by one workgroup
// gridDim.z > 1: the keys of an utterance are split over gridDim.z workgroups (128 workgroups of ~250 keys each left the chip to
// latency: 23 us); a split leaves its row maxima, sums and UNNORMALISED partial contexts in `part` ([nutt][H][splits][g][66]:
// 64 channels, max, sum) and decode_src_attn_merge_kernel combines them in a fixed order.
__global__ __launch_bounds__(256) void decode_src_attn_group_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ Km,
                                                                    const float* __restrict__ Vm, long ldkv,
                                                                    const unsigned char* __restrict__ mask, int g, int Tall, int Tp, int D,
                                                                    float* __restrict__ ctx, float scale, float* __restrict__ pws) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* qs = gsm;                               // [SRC_GQ][64]
  float* sc = qs + SRC_GQ * 64;                  // [g][Tp] scores, then unnormalised probabilities
  float* part = sc + (long)SRC_GQ * Tp;          // [4][SRC_GQ][64]
  float* den = part + 4 * SRC_GQ * 64;           // [2][SRC_GQ]: sums, maxima
  const int h = blockIdx.x, u = blockIdx.y, t_ = threadIdx.x, lane = t_ & 63, w = t_ >> 6;
  const int ns = gridDim.z, sp = blockIdx.z;
  const int per = ((Tall + ns - 1) / ns + 3) & ~3;       // keys per split
  const int k0 = min(sp * per, Tall), T = min(per, Tall - k0);      // this workgroup's keys: k0 .. k0 + T - 1 (T may be 0)
  const float* kb = Km + ((long)u * Tall + k0) * ldkv + h * 64;
  const float* vb = Vm + ((long)u * Tall + k0) * ldkv + h * 64;
  const unsigned char* mk = mask ? mask + (long)u * Tall + k0 : nullptr;
  if (T <= 0) {                                          // a split past the last key (only with several splits): nothing to add
    for (int idx = t_; idx < g * 66; idx += 256) {
      const int r = idx / 66, d = idx - r * 66;
      pws[((((long)u * gridDim.x + h) * ns + sp) * g + r) * 66 + d] = d == 64 ? -INFINITY : 0.f;
    }
    return;
  }
  if (t_ < g * 16) {
    const int r = t_ >> 4, c = t_ & 15;
    *reinterpret_cast<float4*>(&qs[r * 64 + 4 * c]) = *reinterpret_cast<const float4*>(q + ((long)u * g + r) * ldq + h * 64 + 4 * c);
  }
  __syncthreads();
  for (int t = t_; t < T; t += 256) {
    const float4* kp = reinterpret_cast<const float4*>(kb + (long)t * ldkv);
    float4 kv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kv[i] = kp[i];
    const bool dead = mk && !mk[t];
    for (int r = 0; r < g; ++r) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 q4 = *reinterpret_cast<const float4*>(&qs[r * 64 + 4 * i]);
        a = fmaf(q4.x, kv[i].x, fmaf(q4.y, kv[i].y, fmaf(q4.z, kv[i].z, fmaf(q4.w, kv[i].w, a))));
      }
      sc[(long)r * Tp + t] = dead ? -INFINITY : a * scale;
    }
  }
  __syncthreads();
  for (int r = w; r < g; r += 4) {               // a wave per probability row
    float* sr = sc + (long)r * Tp;
    float mx = -INFINITY;
    for (int t = lane; t < T; t += 64) mx = fmaxf(mx, sr[t]);
    mx = wave_max(mx);
    float d = 0.f;
    for (int t = lane; t < T; t += 64) {
      const float e = mx == -INFINITY ? 0.f : __expf(sr[t] - mx);
      sr[t] = e;
      d += e;
    }
    d = wave_sum(d);
    if (lane == 0) { den[r] = d; den[SRC_GQ + r] = mx; }
    if (lane < Tp - T) sr[T + lane] = 0.f;               // (the 16-byte reads below run to the next multiple of 4)
  }
  __syncthreads();
  float acc[SRC_GQ];
#pragma unroll
  for (int r = 0; r < SRC_GQ; ++r) acc[r] = 0.f;
  // a wave takes a contiguous quarter of the keys, four at a time: four value rows (lane = channel) against the four
  // probabilities of every query in ONE 16-byte LDS read (probabilities read one by one - 640 LDS instructions per wave whatever
  // the number of keys - were what this kernel spent its time on: 23 us)
  {
    const int Q = ((T + 3) / 4 + 3) & ~3;                // keys per wave (a multiple of 4)
    const int kb0 = w * Q, ke0 = min(kb0 + Q, (T + 3) & ~3);
    for (int kk = kb0; kk < ke0; kk += 16) {             // sixteen value rows in flight per round trip
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = (vb + (long)min(kk + e, T - 1) * ldkv)[lane];       // (rows past T meet probability 0)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (kk + 4 * c < ke0) {
#pragma unroll
          for (int r = 0; r < SRC_GQ; ++r) {
            if (r < g) {
              const float4 p4 = *reinterpret_cast<const float4*>(&sc[(long)r * Tp + kk + 4 * c]);
              acc[r] = fmaf(p4.x, v[4 * c], fmaf(p4.y, v[4 * c + 1], fmaf(p4.z, v[4 * c + 2], fmaf(p4.w, v[4 * c + 3], acc[r]))));
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < SRC_GQ; ++r)
    if (r < g) part[(w * SRC_GQ + r) * 64 + lane] = acc[r];
  __syncthreads();
  for (int idx = t_; idx < g * 64; idx += 256) {
    const int r = idx >> 6, d = idx & 63;
    const float s = (part[(0 * SRC_GQ + r) * 64 + d] + part[(1 * SRC_GQ + r) * 64 + d]) +
                    (part[(2 * SRC_GQ + r) * 64 + d] + part[(3 * SRC_GQ + r) * 64 + d]);
    if (ns == 1) {
      ctx[((long)u * g + r) * D + h * 64 + d] = den[r] > 0.f ? s / den[r] : 0.f;      // every frame masked: zeros (attention.py:84-88)
    } else {
      float* pr = pws + ((((long)u * gridDim.x + h) * ns + sp) * g + r) * 66;
      pr[d] = s;
      if (d == 0) { pr[64] = den[SRC_GQ + r]; pr[65] = den[r]; }
    }
  }
}
// ctx[u, r, h] = sum_s e^(m_s - M) partial_s / sum_s e^(m_s - M) sum_s over the splits, in split order
__global__ __launch_bounds__(64) void decode_src_attn_merge_kernel(const float* __restrict__ part, int ns, int g, int H, int D,
                                                                   float* __restrict__ ctx) {
  const int h = blockIdx.x, r = blockIdx.y % g, u = blockIdx.y / g, d = threadIdx.x;
  const float* pr = part + ((((long)u * H + h) * ns) * g + r) * 66;
  float M = -INFINITY;
  for (int s = 0; s < ns; ++s) M = fmaxf(M, pr[(long)s * g * 66 + 64]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < ns; ++s) {
    const float* ps = pr + (long)s * g * 66;
    const float m = ps[64];
    const float wgt = (m == -INFINITY) ? 0.f : __expf(m - M);
    num = fmaf(wgt, ps[d], num);
    den = fmaf(wgt, ps[65], den);
  }
  ctx[((long)u * g + r) * D + h * 64 + d] = den > 0.f ? num / den : 0.f;
}
}  // namespace

// Up to 16 small device-to-device copies in ONE launch (the state a replayed beam step hands to the next replay of the same graph:
// prefixes, scores, slot table, CTC state - a memcpy node each would cost ~4 us apiece).  Sizes in bytes, multiples of 4.
namespace {
struct CopyJobs { const unsigned* src[16]; unsigned* dst[16]; unsigned words[16]; int n; };
__global__ __launch_bounds__(256) void copy_jobs_kernel(const CopyJobs j) {
  const int job = blockIdx.y;
  if (job >= j.n) return;
  const unsigned* s = j.src[job];
  unsigned* d = j.dst[job];
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < j.words[job]; i += gridDim.x * 256) d[i] = s[i];
}
}  // namespace

extern "C" int eamd_copy_jobs(const void* const* src, void* const* dst, const int64_t* nbytes, int njobs, void* stream) {
  if (!src || !dst || !nbytes || njobs <= 0 || njobs > 16) return EAMD_EINVAL;
  CopyJobs j;
  unsigned most = 0;
  for (int i = 0; i < 16; ++i) {
    j.src[i] = i < njobs ? (const unsigned*)src[i] : nullptr;
    j.dst[i] = i < njobs ? (unsigned*)dst[i] : nullptr;
    j.words[i] = 0;
    if (i < njobs) {
      if (!src[i] || !dst[i] || nbytes[i] < 0 || nbytes[i] % 4 != 0 || nbytes[i] > (1LL << 33)) return EAMD_EINVAL;
      if (((uintptr_t)src[i] | (uintptr_t)dst[i]) & 3) return EAMD_EUNSUPPORTED;
      j.words[i] = (unsigned)(nbytes[i] / 4);
      most = j.words[i] > most ? j.words[i] : most;
    }
  }
  j.n = njobs;
  const unsigned bx = most == 0 ? 1 : ((most + 255) / 256 < 64 ? (most + 255) / 256 : 64);
  hipLaunchKernelGGL(copy_jobs_kernel, dim3(bx, njobs), dim3(256), 0, (hipStream_t)stream, j);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int eamd_decode_src_attn_group(const float* q, int64_t ldq, const float* kmem, const float* vmem, int64_t ldkv,
                                          const uint8_t* mask, int nutt, int g, int T, int H, int D, float* ctx, void* stream) {
  if (!q || !kmem || !vmem || !ctx || nutt <= 0 || g <= 0 || T <= 0 || H <= 0 || D <= 0) return EAMD_EINVAL;
  if (ldq < D || ldkv < D) return EAMD_EINVAL;
  if (D != H * 64 || T > 1024 || g > SRC_GQ || ldq % 4 != 0 || ldkv % 4 != 0 || (((uintptr_t)q | (uintptr_t)kmem | (uintptr_t)vmem) & 15))
    return EAMD_EUNSUPPORTED;
  const int Tp = (T + 3) & ~3;
  const size_t lds = ((size_t)SRC_GQ * 64 + (size_t)SRC_GQ * Tp + 4 * SRC_GQ * 64 + 2 * SRC_GQ) * 4;
  hipLaunchKernelGGL(decode_src_attn_group_kernel, dim3(H, nutt, 1), dim3(256), lds, (hipStream_t)stream, q, (long)ldq, kmem, vmem, (long)ldkv,
                     mask, g, T, Tp, D, ctx, 0.125f, (float*)nullptr);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int64_t eamd_decode_src_attn_split_workspace(int nutt, int g, int H, int splits) {
  return (int64_t)nutt * H * splits * g * 66;            // floats
}

extern "C" int eamd_decode_src_attn_split(const float* q, int64_t ldq, const float* kmem, const float* vmem, int64_t ldkv,
                                          const uint8_t* mask, int nutt, int g, int T, int H, int D, int splits, float* ws, float* ctx,
                                          void* stream) {
  if (!q || !kmem || !vmem || !ctx || !ws || nutt <= 0 || g <= 0 || T <= 0 || H <= 0 || D <= 0 || splits < 2 || splits > 16) return EAMD_EINVAL;
  if (ldq < D || ldkv < D) return EAMD_EINVAL;
  if (D != H * 64 || T > 4096 || g > SRC_GQ || ldq % 4 != 0 || ldkv % 4 != 0 || (((uintptr_t)q | (uintptr_t)kmem | (uintptr_t)vmem) & 15))
    return EAMD_EUNSUPPORTED;
  const int per = ((T + splits - 1) / splits + 3) & ~3;
  const int Tp = per;
  const size_t lds = ((size_t)SRC_GQ * 64 + (size_t)SRC_GQ * Tp + 4 * SRC_GQ * 64 + 2 * SRC_GQ) * 4;
  if (lds > 64 * 1024) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(decode_src_attn_group_kernel, dim3(H, nutt, splits), dim3(256), lds, (hipStream_t)stream, q, (long)ldq, kmem, vmem,
                     (long)ldkv, mask, g, T, Tp, D, ctx, 0.125f, ws);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(decode_src_attn_merge_kernel, dim3(H, nutt * g), dim3(64), 0, (hipStream_t)stream, ws, splits, g, H, D, ctx);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int eamd_decode_src_attn(const float* q, int64_t ldq, const float* kmem, const float* vmem, int64_t ldkv,
                                    const uint8_t* mask, int nutt, int g, int T, int H, int D, float* ctx, void* stream) {
  if (!q || !kmem || !vmem || !ctx || nutt <= 0 || g <= 0 || T <= 0 || H <= 0 || D <= 0) return EAMD_EINVAL;
  if (ldq < D || ldkv < D) return EAMD_EINVAL;
  if (D != H * 64 || T > 8192 || ldq % 4 != 0 || ldkv % 4 != 0 || (((uintptr_t)q | (uintptr_t)kmem | (uintptr_t)vmem) & 15))
    return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL(decode_src_attn_kernel, dim3(H, nutt * g), dim3(256), (size_t)(T + 256 + 8) * 4, (hipStream_t)stream, q, (long)ldq, kmem, vmem,
                     (long)ldkv, mask, g, T, D, ctx, 0.125f);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
