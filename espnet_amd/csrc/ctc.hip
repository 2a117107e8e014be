// CTC loss forward + gradient for gfx950.
// Replaces warp-ctc (tools/installers/install_warp-ctc.sh; call sites ctc.py:40-43,62-63) and
// torch.nn.CTCLoss(reduction="sum")/B (ctc.py:37-39,53-61; espnet2/asr/ctc.py:32-52): takes raw
// activations, returns sum_b(-log p(y_b|x_b)) per utterance and d(loss)/d(activations).
//
// Four launches, no host synchronisation:
//   prep       : compact padded labels (ignore_id removed) -> extended label table
//   lse_gather : one 256-thread block per (t,b) row: log-sum-exp over V (the only pass that reads the
//                whole 159 MB logits), then gathers the <=2L+1 label log-probs into a compact lattice
//   alpha_beta : one block per (b, direction): wavefront scan over T in log space, states across
//                lanes, previous column exchanged through LDS (double buffered, 1 barrier per frame)
//   grad       : one block per (t,b): softmax row minus per-label occupancies accumulated in LDS
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

__device__ __forceinline__ float lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}
// the scan's critical path: hardware exp2 / log2 forms (v_exp_f32 / v_log_f32; arguments are <= 0 resp. in [1, 3]),
// the libm-accurate chains tripled the per-frame latency of the serial T-frame recursion
__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// one wave per utterance: 64 labels per step, kept ones compacted by ballot / popcount (a thread per utterance walked the
// Lmax labels through 100 dependent loads: 25 us at config 2)
__global__ __launch_bounds__(64) void ctc_prep_kernel(const long long* __restrict__ ys, int Lmax, int ignore_id,
                                                      int* __restrict__ lab, int* __restrict__ lablen, int B) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int n = 0;
  for (int i0 = 0; i0 < Lmax; i0 += 64) {
    const int i = i0 + lane;
    const long long y = i < Lmax ? ys[(long)b * Lmax + i] : (long long)ignore_id;
    const bool keep = i < Lmax && y != ignore_id;
    const unsigned long long m = __ballot(keep);
    if (keep) lab[(long)b * Lmax + n + __popcll(m & ((1ULL << lane) - 1ULL))] = (int)y;
    n += __popcll(m);
  }
  if (lane == 0) lablen[b] = n;
}

__global__ __launch_bounds__(256) void ctc_lse_gather_kernel(const float* __restrict__ x, long st, long sb,
                                                             const int* __restrict__ ilen,
                                                             const int* __restrict__ lab,
                                                             const int* __restrict__ lablen, int Lmax,
                                                             float* __restrict__ lse_out, float* __restrict__ lp,
                                                             int T, int V, int Smax, int blank) {
  __shared__ float red[16];
  const int t = blockIdx.x, b = blockIdx.y;
  if (t >= ilen[b]) return;
  const float* xr = x + t * st + b * sb;
  // one pass over the row: running maximum and rescaled sum per thread (16-byte loads where the row allows)
  float mx = -INFINITY, se = 0.f;
  if (((reinterpret_cast<uintptr_t>(xr) & 15) == 0) && (V % 4 == 0)) {
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    for (int q = threadIdx.x; q < V / 4; q += blockDim.x) {
      const float4 v = x4[q];
      const float m4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      if (m4 > mx) { se *= expf(mx - m4); mx = m4; }
      se += (expf(v.x - mx) + expf(v.y - mx)) + (expf(v.z - mx) + expf(v.w - mx));
    }
  } else {
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      const float xv = xr[v];
      if (xv > mx) { se *= expf(mx - xv); mx = xv; }
      se += expf(xv - mx);
    }
  }
  const float mt = mx;
  mx = block_max(mx, red);
  se = block_sum(mt == -INFINITY ? 0.f : se * expf(mt - mx), red);
  const float lse = mx + logf(se);
  if (threadIdx.x == 0) lse_out[(long)b * T + t] = lse;
  const int S = 2 * lablen[b] + 1;
  float* lpr = lp + ((long)b * T + t) * Smax;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    int l = (s & 1) ? lab[(long)b * Lmax + (s >> 1)] : blank;
    lpr[s] = xr[l] - lse;
  }
}

// blockIdx.y == 0: alpha (forward), == 1: beta (backward, includes the emission at t)
__global__ __launch_bounds__(1024) void ctc_alpha_beta_kernel(const float* __restrict__ lp,
                                                              const int* __restrict__ ilen,
                                                              const int* __restrict__ lab,
                                                              const int* __restrict__ lablen, int Lmax,
                                                              float* __restrict__ alpha, float* __restrict__ beta,
                                                              float* __restrict__ nll, int T, int Smax, int blank) {
  extern __shared__ float sh[];  // [2][Smax + 4]
  constexpr int PF = 8;
  const int b = blockIdx.x;
  const int dir = blockIdx.y;
  const int s = threadIdx.x;
  const int Tb = ilen[b];
  const int L = lablen[b];
  const int S = 2 * L + 1;
  const int W = Smax + 4;
  float* buf0 = sh;
  float* buf1 = sh + W;
  // pads: indices [0,1] and [S+2, S+3] hold -inf so s-1, s-2, s+1, s+2 never need bounds checks
  for (int i = threadIdx.x; i < 2 * W; i += blockDim.x) sh[i] = -INFINITY;
  __syncthreads();
  if (Tb <= 0) {
    if (dir == 0 && s == 0) nll[b] = (L == 0) ? 0.f : INFINITY;
    return;
  }
  const bool active = s < S;
  const int my = active ? ((s & 1) ? lab[(long)b * Lmax + (s >> 1)] : blank) : blank;
  bool skip = false;  // may take the s-2 (alpha) / s+2 (beta) transition
  if (active && (s & 1)) {
    if (dir == 0) skip = (s >= 2) && (lab[(long)b * Lmax + ((s - 2) >> 1)] != my);
    else          skip = (s + 2 < S) && (lab[(long)b * Lmax + ((s + 2) >> 1)] != my);
  }
  const float* lpb = lp + (long)b * T * Smax;
  float* out = (dir == 0 ? alpha : beta) + (long)b * T * Smax;

  if (dir == 0) {
    float cur = -INFINITY;
    if (active && s < 2) cur = lpb[s];
    if (active) { buf0[s + 2] = cur; out[s] = cur; }
    __syncthreads();
    float* prev = buf0; float* next = buf1;
    // emissions are fetched PF frames ahead (a frame's own compute + barrier is ~0.1 us, an L2 / HBM round trip
    // several times that: with one frame of look-ahead every step of the scan waited for its load)
    float er[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) er[i] = (active && 1 + i < Tb) ? lpb[(long)(1 + i) * Smax + s] : 0.f;
    for (int t0 = 1; t0 < Tb; t0 += PF) {
      float en[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) en[i] = (active && t0 + PF + i < Tb) ? lpb[(long)(t0 + PF + i) * Smax + s] : 0.f;
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int t = t0 + i;
        if (t < Tb) {                                 // block-uniform
          if (active) {
            float a = lse3(prev[s + 2], prev[s + 1], skip ? prev[s] : -INFINITY) + er[i];
            next[s + 2] = a;
            out[(long)t * Smax + s] = a;
          }
          __syncthreads();
          float* tmp = prev; prev = next; next = tmp;
        }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) er[i] = en[i];
    }
    if (s == 0) {
      float a1 = prev[S - 1 + 2];
      float a2 = S >= 2 ? prev[S - 2 + 2] : -INFINITY;
      nll[b] = -lse2(a1, a2);
    }
  } else {
    float cur = -INFINITY;
    if (active && s >= S - 2) cur = lpb[(long)(Tb - 1) * Smax + s];
    if (active) { buf0[s + 2] = cur; out[(long)(Tb - 1) * Smax + s] = cur; }
    __syncthreads();
    float* prev = buf0; float* next = buf1;
    float er[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) er[i] = (active && Tb - 2 - i >= 0) ? lpb[(long)(Tb - 2 - i) * Smax + s] : 0.f;
    for (int t0 = Tb - 2; t0 >= 0; t0 -= PF) {
      float en[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i) en[i] = (active && t0 - PF - i >= 0) ? lpb[(long)(t0 - PF - i) * Smax + s] : 0.f;
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int t = t0 - i;
        if (t >= 0) {                                 // block-uniform
          if (active) {
            float a = lse3(prev[s + 2], prev[s + 3], skip ? prev[s + 4] : -INFINITY) + er[i];
            next[s + 2] = a;
            out[(long)t * Smax + s] = a;
          }
          __syncthreads();
          float* tmp = prev; prev = next; next = tmp;
        }
      }
#pragma unroll
      for (int i = 0; i < PF; ++i) er[i] = en[i];
    }
  }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ x, long st, long sb,
                                                       const int* __restrict__ ilen, const int* __restrict__ lab,
                                                       const int* __restrict__ lablen, int Lmax,
                                                       const float* __restrict__ lse, const float* __restrict__ lp,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, const float* __restrict__ nll,
                                                       float* __restrict__ grad, long gst, long gsb, int T, int V,
                                                       int Smax, int blank, float scale) {
  extern __shared__ float occ[];  // [V]
  const int t = blockIdx.x, b = blockIdx.y;
  float* gr = grad + t * gst + b * gsb;
  if (t >= ilen[b]) {
    for (int v = threadIdx.x; v < V; v += blockDim.x) gr[v] = 0.f;
    return;
  }
  for (int v = threadIdx.x; v < V; v += blockDim.x) occ[v] = 0.f;
  __syncthreads();
  const int S = 2 * lablen[b] + 1;
  const long base = ((long)b * T + t) * Smax;
  const float nl = nll[b];
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    int l = (s & 1) ? lab[(long)b * Lmax + (s >> 1)] : blank;
    float g = expf(alpha[base + s] + beta[base + s] - lp[base + s] + nl);
    atomicAdd(&occ[l], g);
  }
  __syncthreads();
  const float* xr = x + t * st + b * sb;
  const float ls = lse[(long)b * T + t];
  for (int v = threadIdx.x; v < V; v += blockDim.x) gr[v] = (expf(xr[v] - ls) - occ[v]) * scale;
}

}  // namespace

extern "C" {

/* workspace bytes needed by eamd_ctc_loss */
int64_t eamd_ctc_workspace_bytes(int B, int T, int Lmax) {
  int64_t Smax = 2 * (int64_t)Lmax + 1;
  int64_t n = (int64_t)B * Lmax * 4 + (int64_t)B * 4      /* lab, lablen */
              + (int64_t)B * T * 4                          /* lse */
              + 3 * (int64_t)B * T * Smax * 4;              /* lp, alpha, beta */
  return n + 256;
}

/*
 * acts  : [T,B,V] or [B,T,V] fp32 via element strides (stride_t, stride_b); V contiguous
 * ys_pad: [B,Lmax] int64 padded with ignore_id; ilens: [B] int32 valid frames
 * nll   : [B] fp32 out (-log p per utterance, +inf if no valid alignment)
 * grad  : optional, same strides convention (gstride_t, gstride_b); = scale * d(sum_b nll_b)/d acts
 */
int eamd_ctc_loss(const float* acts, int64_t stride_t, int64_t stride_b, const int64_t* ys_pad,
                  const int32_t* ilens, float* nll, float* grad, int64_t gstride_t, int64_t gstride_b,
                  void* workspace, int B, int T, int V, int Lmax, int blank, int ignore_id, float grad_scale,
                  void* stream) {
  if (!acts || !ys_pad || !ilens || !nll || !workspace || B <= 0 || T <= 0 || V <= 0 || Lmax < 0) return EAMD_EINVAL;
  const int Smax = 2 * Lmax + 1;
  int threads = ((Smax + 63) / 64) * 64;
  if (threads > 1024) return EAMD_EUNSUPPORTED;
  if (grad && (size_t)V * 4 > 64 * 1024) return EAMD_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)workspace;
  int Lm = Lmax > 0 ? Lmax : 1;
  int* lab = (int*)w; w += (size_t)B * Lm * 4;
  int* lablen = (int*)w; w += (size_t)B * 4;
  w = (char*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
  float* lse = (float*)w; w += (size_t)B * T * 4;
  float* lp = (float*)w; w += (size_t)B * T * Smax * 4;
  float* alpha = (float*)w; w += (size_t)B * T * Smax * 4;
  float* beta = (float*)w;

  hipLaunchKernelGGL(ctc_prep_kernel, dim3(B), dim3(64), 0, s, (const long long*)ys_pad, Lmax, ignore_id, lab, lablen, B);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctc_lse_gather_kernel, dim3(T, B), dim3(256), 0, s, acts, (long)stride_t, (long)stride_b, ilens,
                     lab, lablen, Lm, lse, lp, T, V, Smax, blank);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B, 2), dim3(threads), 2 * (Smax + 4) * sizeof(float), s, lp, ilens,
                     lab, lablen, Lm, alpha, beta, nll, T, Smax, blank);
  EAMD_LAUNCH_CHECK();
  if (grad) {
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(256), (size_t)V * sizeof(float), s, acts, (long)stride_t,
                       (long)stride_b, ilens, lab, lablen, Lm, lse, lp, alpha, beta, nll, grad, (long)gstride_t,
                       (long)gstride_b, T, V, Smax, blank, grad_scale);
    EAMD_LAUNCH_CHECK();
  }
  return EAMD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// CTC prefix scoring for joint CTC/attention beam search (Watanabe et al. 2017, Algorithm 2).
// reference: espnet/nets/ctc_prefix_score.py:224-310 (CTCPrefixScore.__call__), :157-162 (the per-frame
// loop of CTCPrefixScoreTH).  One thread = one (hypothesis, candidate token): sequential scan over the
// T frames in log space; candidates of one hypothesis share the broadcast r_prev reads.
//   logp   [T, V]                  frame log-posteriors (CTC.log_softmax)
//   r_prev [nhyp, T, 2]            (r^n, r^b) of each hypothesis prefix
//   cand   [nhyp, ncand] int32     tokens to score;  last[nhyp], olen[nhyp] = last token / prefix length-1
//   psi    [nhyp, ncand]           log prefix probabilities;  r_new [nhyp, ncand, T, 2]
// ---------------------------------------------------------------------------------------------
namespace {
constexpr float kLogZero = -10000000000.0f;
// numpy.logaddexp for the prefix recursion: 249 frames x 4 of these in ONE wave's dependent chain - with the library expf / log1pf
// (~100 instructions each) a search step spent 345 us here.  Hardware exp2 / log2 (1 ulp each) instead; log1p(e) below 2^-12 by its
// series (1 + e would round the term away).  Absolute error of one call <= 1.2e-7 on values of magnitude 1 .. 100.
__device__ __forceinline__ float lae(float a, float b) {
  const float m = fmaxf(a, b);
  const float e = a == b ? 1.f : __expf(-fabsf(a - b));     // equal operands incl. (-inf, -inf): log 2 on top of m, not NaN
  const float l = e < 2.44140625e-4f ? e * (1.f - 0.5f * e) : __logf(1.f + e);
  return m + l;
}
__global__ void ctc_prefix_kernel(const float* __restrict__ logp, const float* __restrict__ r_prev,
                                  const int* __restrict__ cand, const int* __restrict__ last,
                                  const int* __restrict__ olen, float* __restrict__ psi, float* __restrict__ r_new,
                                  int T, int V, int ncand, int blank, int eos) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (j >= ncand) return;
  const int c = cand[(long)h * ncand + j];
  // a candidate outside the vocabulary is never dereferenced: its score is NaN (a selection treats NaN as -inf), its state untouched
  if (c < 0 || c >= V) { psi[(long)h * ncand + j] = __builtin_nanf(""); return; }
  const float* rp = r_prev + (long)h * T * 2;
  float* rn = r_new + ((long)h * ncand + j) * T * 2;
  const int ol = min(max(olen[h], 0), T);
  const bool same = ol > 0 && last[h] == c;
  const int start = max(ol, 1);
  // rows before start-1 are never read by later steps; keep them at log-zero like the reference's r
  for (int t = 0; t < start - 1; ++t) { rn[2 * t] = kLogZero; rn[2 * t + 1] = kLogZero; }
  float rn_n, rn_b;
  if (ol == 0) { rn_n = logp[c]; rn_b = kLogZero; }
  else { rn_n = kLogZero; rn_b = kLogZero; }
  rn[2 * (start - 1)] = rn_n; rn[2 * (start - 1) + 1] = rn_b;
  float lpsi = rn_n;
  // the recursion over t is serial, its operands are not: the posteriors and the previous prefix's rows of the next PF frames
  // are requested together (each logp row is its own cache line, 20 KB apart: fetched inside the chain every frame cost a
  // memory round trip - 344 us for 249 frames), phi is formed beside them, and only the log-add chain stays serial.
  // Same operations in the same order as the frame-by-frame loop: results bit for bit.
  // ... and a RING of NB such groups is kept in flight (the group NB - 1 ahead is requested before the current one is walked).
  // Measured at config 2 (249 frames, 10 hypotheses x 15 candidates): 152 us with one group in flight, 157 us with the ring and
  // 8-byte state stores - neither the round trips nor the stores bound this kernel: it is ONE wave per hypothesis walking a
  // dependent chain of ~60 VALU + 8 transcendental instructions per frame with nothing else on its SIMD (0.6 us per frame).
  constexpr int PF = 8, NB = 4;
  float xv[NB][PF], bv[NB][PF], pn[NB][PF], pb[NB][PF];
  auto request = [&](auto slot_c, int t0) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = min(max(t0 + q, 1), T - 1);          // (frames behind the end: the last one again; T = 1: frame 0, never walked)
      const int tp = max(t - 1, 0);
      xv[SL][q] = logp[(long)t * V + c];
      bv[SL][q] = logp[(long)t * V + blank];
      pn[SL][q] = rp[2 * tp];
      pb[SL][q] = rp[2 * tp + 1];
    }
  };
  auto walk = [&](auto slot_c, int t0) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = t0 + q;
      if (t < T) {
        const float phi = same ? pb[SL][q] : lae(pn[SL][q], pb[SL][q]);
        const float nn = lae(rn_n, phi) + xv[SL][q];
        const float nb = lae(rn_n, rn_b) + bv[SL][q];
        lpsi = lae(lpsi, phi + xv[SL][q]);
        rn_n = nn; rn_b = nb;
        *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(nn, nb);        // one 8-byte store per frame
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  request(S0{}, start);
  request(S1{}, start + PF);
  request(S2{}, start + 2 * PF);
  for (int t0 = start; t0 < T; t0 += NB * PF) {
    request(S3{}, t0 + 3 * PF);
    walk(S0{}, t0);
    request(S0{}, t0 + 4 * PF);
    walk(S1{}, t0 + PF);
    request(S1{}, t0 + 5 * PF);
    walk(S2{}, t0 + 2 * PF);
    request(S2{}, t0 + 6 * PF);
    walk(S3{}, t0 + 3 * PF);
  }
  if (c == eos) lpsi = lae(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);
  if (c == blank) lpsi = kLogZero;
  psi[(long)h * ncand + j] = lpsi;
}
// The same recursion for the hypotheses of SEVERAL utterances in one launch: hypothesis h belongs to utterance h / per_utt,
// whose posteriors are logp[u] ([Tmax, V], lens[u] valid frames); r_prev / r_new rows are padded to Tmax (rows from lens[u] on
// are never read).
__global__ void ctc_prefix_batch_kernel(const float* __restrict__ logp_all, const int* __restrict__ lens, int per_utt,
                                        const float* __restrict__ r_prev, const int* __restrict__ cand,
                                        const int* __restrict__ last, const int* __restrict__ olen, float* __restrict__ psi,
                                        float* __restrict__ r_new, int Tmax, int V, int ncand, int blank, int eos) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (j >= ncand) return;
  const int u = h / per_utt;
  const int T = min(max(lens[u], 1), Tmax);                 // a length outside 1 .. Tmax never becomes an address
  const float* logp = logp_all + (long)u * Tmax * V;
  const int c = cand[(long)h * ncand + j];
  if (c < 0 || c >= V) { psi[(long)h * ncand + j] = __builtin_nanf(""); return; }
  const float* rp = r_prev + (long)h * Tmax * 2;
  float* rn = r_new + ((long)h * ncand + j) * Tmax * 2;
  const int ol = min(max(olen[h], 0), T);
  const bool same = ol > 0 && last[h] == c;
  const int start = max(ol, 1);
  for (int t = 0; t < min(start - 1, T); ++t) { rn[2 * t] = kLogZero; rn[2 * t + 1] = kLogZero; }
  float rn_n, rn_b;
  if (ol == 0) { rn_n = logp[c]; rn_b = kLogZero; }
  else { rn_n = kLogZero; rn_b = kLogZero; }
  if (start - 1 < T) { rn[2 * (start - 1)] = rn_n; rn[2 * (start - 1) + 1] = rn_b; }
  float lpsi = rn_n;
  // the recursion over t is serial, its operands are not: the posteriors and the previous prefix's rows of the next PF frames
  // are requested together (each logp row is its own cache line, 20 KB apart: fetched inside the chain every frame cost a
  // memory round trip - 344 us for 249 frames), phi is formed beside them, and only the log-add chain stays serial.
  // Same operations in the same order as the frame-by-frame loop: results bit for bit.
  // ... and a RING of NB such groups is kept in flight (the group NB - 1 ahead is requested before the current one is walked).
  // Measured at config 2 (249 frames, 10 hypotheses x 15 candidates): 152 us with one group in flight, 157 us with the ring and
  // 8-byte state stores - neither the round trips nor the stores bound this kernel: it is ONE wave per hypothesis walking a
  // dependent chain of ~60 VALU + 8 transcendental instructions per frame with nothing else on its SIMD (0.6 us per frame).
  constexpr int PF = 8, NB = 4;
  float xv[NB][PF], bv[NB][PF], pn[NB][PF], pb[NB][PF];
  auto request = [&](auto slot_c, int t0) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = min(max(t0 + q, 1), T - 1);          // (frames behind the end: the last one again; T = 1: frame 0, never walked)
      const int tp = max(t - 1, 0);
      xv[SL][q] = logp[(long)t * V + c];
      bv[SL][q] = logp[(long)t * V + blank];
      pn[SL][q] = rp[2 * tp];
      pb[SL][q] = rp[2 * tp + 1];
    }
  };
  auto walk = [&](auto slot_c, int t0) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = t0 + q;
      if (t < T) {
        const float phi = same ? pb[SL][q] : lae(pn[SL][q], pb[SL][q]);
        const float nn = lae(rn_n, phi) + xv[SL][q];
        const float nb = lae(rn_n, rn_b) + bv[SL][q];
        lpsi = lae(lpsi, phi + xv[SL][q]);
        rn_n = nn; rn_b = nb;
        *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(nn, nb);        // one 8-byte store per frame
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  request(S0{}, start);
  request(S1{}, start + PF);
  request(S2{}, start + 2 * PF);
  for (int t0 = start; t0 < T; t0 += NB * PF) {
    request(S3{}, t0 + 3 * PF);
    walk(S0{}, t0);
    request(S0{}, t0 + 4 * PF);
    walk(S1{}, t0 + PF);
    request(S1{}, t0 + 5 * PF);
    walk(S2{}, t0 + 2 * PF);
    request(S2{}, t0 + 6 * PF);
    walk(S3{}, t0 + 3 * PF);
  }
  if (c == eos) lpsi = lae(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);
  if (c == blank) lpsi = kLogZero;
  psi[(long)h * ncand + j] = lpsi;
}
}  // namespace

// ---- the same scores without the serial chain on a beam step's critical path ------------------------------------------------
// log psi of a candidate is logsumexp over t of (phi(t-1) + x(t)) (ctc_prefix_score.py:290-296: log_psi never reads r[t]): a
// PARALLEL reduction over the frames.  Only the forward variables r^n / r^b of the NEXT step need the frame-by-frame recursion,
// and only for the `beam` continuations that survive the selection - so a step scores its candidates with ctc_prefix_psi_kernel
// (one wave per (hypothesis, candidate), lanes over frames) and the survivors' states are made by ctc_prefix_state_kernel at the
// START of the next step, on a second stream beside the decoder stack (160 us of serial recursion off the critical path).
namespace {
template <int NT>      // frames per lane: 64 NT >= Tmax
__global__ __launch_bounds__(64) void ctc_prefix_psi_kernel(const float* __restrict__ logp_all, const int* __restrict__ lens, int per_utt,
                                                            const float* __restrict__ r_prev, const int* __restrict__ cand,
                                                            const int* __restrict__ last, int ol, float* __restrict__ psi, int Tmax,
                                                            int V, int ncand, int blank, int eos, const int* __restrict__ ol_dev) {
  if (ol_dev) ol = ol_dev[0] + ol;                       // the prefix length from the device step index (+ host offset)
  const int j = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int u = h / per_utt;
  const int T = min(max(lens[u], 1), Tmax);
  const float* logp = logp_all + (long)u * Tmax * V;
  const int c = cand[(long)h * ncand + j];
  if (c < 0 || c >= V) { if (lane == 0) psi[(long)h * ncand + j] = __builtin_nanf(""); return; }
  const float* rp = r_prev + (long)h * Tmax * 2;
  ol = min(max(ol, 0), T);
  const bool same = ol > 0 && last[h] == c;
  const int start = max(ol, 1);
  // terms phi(t-1) + x(t), t = start .. T-1, and the initial r[start-1, 0]; two passes: maximum, then the sum of exponentials
  float term[NT];                                  // Tmax <= 64 NT frames per wave (host check)
  float mx = (ol == 0 && lane == 0) ? logp[c] : kLogZero;
  const float init = mx;
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int t = start + lane + 64 * q;
    term[q] = -INFINITY;
    if (t < T) {
      const float pn = rp[2 * (t - 1)], pb = rp[2 * (t - 1) + 1];
      const float m = fmaxf(pn, pb);
      const float phi = same ? pb : (m == -INFINITY ? -INFINITY : m + log1pf(expf(-fabsf(pn - pb))));
      term[q] = phi + logp[(long)t * V + c];
      mx = fmaxf(mx, term[q]);
    }
  }
  mx = wave_max(mx);
  float se = lane == 0 ? expf(init - mx) : 0.f;
#pragma unroll
  for (int q = 0; q < NT; ++q) se += (term[q] == -INFINITY) ? 0.f : expf(term[q] - mx);
  se = wave_sum(se);
  if (lane == 0) {
    float lpsi = mx == -INFINITY ? -INFINITY : mx + logf(se);
    if (c == eos) lpsi = lae(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);
    if (c == blank) lpsi = kLogZero;
    psi[(long)h * ncand + j] = lpsi;
  }
}

// forward variables of the surviving continuations: slot s continues the hypothesis of slot parent[s] with token tok[s]; the
// recursion of ctc_prefix_batch_kernel for that one (hypothesis, candidate) pair, written straight into r_out[s] ([n, Tmax, 2]).
// dead[s] != 0 (an ended or empty slot): the row is filled with log-zero.
__global__ __launch_bounds__(64) void ctc_prefix_state_kernel(const float* __restrict__ logp_all, const int* __restrict__ lens, int per_utt,
                                                              const float* __restrict__ r_prev, const long long* __restrict__ parent,
                                                              const long long* __restrict__ tok, const int* __restrict__ last, int ol,
                                                              const float* __restrict__ alive, float* __restrict__ r_out, int n, int Tmax,
                                                              int V, int blank, const int* __restrict__ ol_dev) {
  if (ol_dev) ol = ol_dev[0] + ol;
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n) return;
  const int u = s / per_utt;
  const int T = min(max(lens[u], 1), Tmax);
  const float* logp = logp_all + (long)u * Tmax * V;
  float* rn = r_out + (long)s * Tmax * 2;
  long long hh = parent[s];
  hh = hh < 0 ? 0 : (hh >= n ? n - 1 : hh);
  const int c = (int)tok[s];
  if (c < 0 || c >= V || !(alive[s] > -INFINITY)) {                      // nothing continues in this slot
    for (int t = 0; t < T; ++t) *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(kLogZero, kLogZero);
    return;
  }
  const float* rp = r_prev + hh * Tmax * 2;
  ol = min(max(ol, 0), T);
  const bool same = ol > 0 && last[hh] == c;
  const int start = max(ol, 1);
  for (int t = 0; t < min(start - 1, T); ++t) *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(kLogZero, kLogZero);
  float rn_n, rn_b;
  if (ol == 0) { rn_n = logp[c]; rn_b = kLogZero; }
  else { rn_n = kLogZero; rn_b = kLogZero; }
  if (start - 1 < T) *reinterpret_cast<float2*>(rn + 2 * (start - 1)) = make_float2(rn_n, rn_b);
  constexpr int PF = 8;
  for (int t0 = start; t0 < T; t0 += PF) {
    float xv[PF], bv[PF], phi[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = min(t0 + q, T - 1);
      xv[q] = logp[(long)t * V + c];
      bv[q] = logp[(long)t * V + blank];
      const float pn = rp[2 * (t - 1)], pb = rp[2 * (t - 1) + 1];
      phi[q] = same ? pb : lae(pn, pb);
    }
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int t = t0 + q;
      if (t < T) {
        const float nn = lae(rn_n, phi[q]) + xv[q];
        const float nb = lae(rn_n, rn_b) + bv[q];
        rn_n = nn; rn_b = nb;
        *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(nn, nb);
      }
    }
  }
}

// The same forward variables as a PARALLEL scan (Tmax <= 2048: 8 / 16 / 32 frames per lane): r^n does not read r^b -
//   r^n(t) = logaddexp(x(t) + r^n(t-1), x(t) + phi(t-1)),   then   r^b(t) = logaddexp(b(t) + r^b(t-1), b(t) + r^n(t-1))
// are two scalar recurrences s(t) = logaddexp(a(t) + s(t-1), c(t)); maps (a, c) compose as (a2 + a1, logaddexp(a2 + c1, c2)).
// One wave per slot, a lane owns Q consecutive frames: inclusive maps inside the lane, a six-step scan of the lanes' totals,
// then every frame applies its map to the state entering the lane - ~16 dependent logaddexp per recurrence instead of one per
// frame (130 us -> a few us at T = 249).  Same quantities; the order of the additions differs from the frame-by-frame recursion
// (both are within 1e-5 + 2e-6 |ref| of float64: test_ctc_prefix_score_vs_float64).
struct ScanMap { float a, c; };
__device__ __forceinline__ ScanMap scan_after(ScanMap later, ScanMap earlier) {     // later o earlier
  return ScanMap{later.a + earlier.a, lae(later.a + earlier.c, later.c)};
}
__device__ __forceinline__ ScanMap wave_scan_exclusive(ScanMap tot, int lane) {
  ScanMap inc = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    ScanMap o{__shfl_up(inc.a, d, 64), __shfl_up(inc.c, d, 64)};
    if (lane >= d) inc = scan_after(inc, o);
  }
  ScanMap ex{__shfl_up(inc.a, 1, 64), __shfl_up(inc.c, 1, 64)};
  if (lane == 0) ex = ScanMap{0.f, -INFINITY};
  return ex;
}
template <int Q>       // consecutive frames per lane: 64 Q >= Tmax
__global__ __launch_bounds__(64) void ctc_prefix_state_scan_kernel(const float* __restrict__ logp_all, const int* __restrict__ lens,
                                                                   int per_utt, const float* __restrict__ r_prev,
                                                                   const long long* __restrict__ parent, const long long* __restrict__ tok,
                                                                   const int* __restrict__ last, int ol, const float* __restrict__ alive,
                                                                   float* __restrict__ r_out, int n, int Tmax, int V, int blank,
                                                                   const int* __restrict__ ol_dev) {
  if (ol_dev) ol = ol_dev[0] + ol;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int u = s / per_utt;
  const int T = min(max(lens[u], 1), Tmax);
  const float* logp = logp_all + (long)u * Tmax * V;
  float* rn = r_out + (long)s * Tmax * 2;
  long long hh = parent[s];
  hh = hh < 0 ? 0 : (hh >= n ? n - 1 : hh);
  const int c = (int)tok[s];
  if (c < 0 || c >= V || !(alive[s] > -INFINITY)) {                      // nothing continues in this slot
    for (int t = lane; t < T; t += 64) *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(kLogZero, kLogZero);
    return;
  }
  const float* rp = r_prev + hh * Tmax * 2;
  ol = min(max(ol, 0), T);
  const bool same = ol > 0 && last[hh] == c;
  const int start = max(ol, 1);
  for (int t = lane; t < min(start - 1, T); t += 64) *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(kLogZero, kLogZero);
  const float n0 = ol == 0 ? logp[c] : kLogZero, b0 = kLogZero;
  if (lane == 0 && start - 1 < T) *reinterpret_cast<float2*>(rn + 2 * (start - 1)) = make_float2(n0, b0);
  float xv[Q], bv[Q], phi[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int t = start + Q * lane + q;
    xv[q] = 0.f; bv[q] = 0.f; phi[q] = -INFINITY;                        // past the last frame: the identity map
    if (t < T) {
      xv[q] = logp[(long)t * V + c];
      bv[q] = logp[(long)t * V + blank];
      const float2 p = *reinterpret_cast<const float2*>(rp + 2 * (t - 1));
      phi[q] = same ? p.y : lae(p.x, p.y);
    }
  }
  // r^n
  ScanMap mp[Q];
  mp[0] = ScanMap{xv[0], xv[0] + phi[0]};
#pragma unroll
  for (int q = 1; q < Q; ++q) mp[q] = scan_after(ScanMap{xv[q], xv[q] + phi[q]}, mp[q - 1]);
  ScanMap ex = wave_scan_exclusive(mp[Q - 1], lane);
  const float n_in = lae(ex.a + n0, ex.c);
  float nn[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) nn[q] = lae(mp[q].a + n_in, mp[q].c);
  // r^b
  mp[0] = ScanMap{bv[0], start + Q * lane < T ? bv[0] + n_in : -INFINITY};
#pragma unroll
  for (int q = 1; q < Q; ++q) mp[q] = scan_after(ScanMap{bv[q], start + Q * lane + q < T ? bv[q] + nn[q - 1] : -INFINITY}, mp[q - 1]);
  ex = wave_scan_exclusive(mp[Q - 1], lane);
  const float b_in = lae(ex.a + b0, ex.c);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int t = start + Q * lane + q;
    if (t < T) *reinterpret_cast<float2*>(rn + 2 * t) = make_float2(nn[q], lae(mp[q].a + b_in, mp[q].c));
  }
}
}  // namespace

extern "C" int eamd_ctc_prefix_psi(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev,
                                   const int32_t* cand, const int32_t* last, int olen, float* psi, int ncand, int Tmax, int V, int blank,
                                   int eos, void* stream) {
  return eamd_ctc_prefix_psi_dyn(logp, lens, nutt, per_utt, r_prev, cand, last, olen, nullptr, psi, ncand, Tmax, V, blank, eos, stream);
}

extern "C" int eamd_ctc_prefix_psi_dyn(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev,
                                       const int32_t* cand, const int32_t* last, int olen, const int32_t* olen_dev, float* psi, int ncand,
                                       int Tmax, int V, int blank, int eos, void* stream) {
  if (!logp || !lens || !r_prev || !cand || !last || !psi || nutt <= 0 || per_utt <= 0 || ncand <= 0 || Tmax <= 0 || V <= 0 ||
      (!olen_dev && olen < 0))
    return EAMD_EINVAL;
  const int* ol_dev = olen_dev;
  if (Tmax > 2048) return EAMD_EUNSUPPORTED;            // 8 / 16 / 32 frames per lane are held in registers
#define EAMD_PSI_(NT) hipLaunchKernelGGL(ctc_prefix_psi_kernel<NT>, dim3(ncand, nutt * per_utt), dim3(64), 0, (hipStream_t)stream, logp, lens, \
                                         per_utt, r_prev, cand, last, olen, psi, Tmax, V, ncand, blank, eos, ol_dev)
  if (Tmax <= 512) EAMD_PSI_(8);
  else if (Tmax <= 1024) EAMD_PSI_(16);
  else EAMD_PSI_(32);
#undef EAMD_PSI_
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int eamd_ctc_prefix_state(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev,
                                     const int64_t* parent, const int64_t* tok, const int32_t* last, int olen, const float* alive,
                                     float* r_out, int Tmax, int V, int blank, void* stream) {
  return eamd_ctc_prefix_state_dyn(logp, lens, nutt, per_utt, r_prev, parent, tok, last, olen, nullptr, alive, r_out, Tmax, V, blank, stream);
}

extern "C" int eamd_ctc_prefix_state_dyn(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev,
                                         const int64_t* parent, const int64_t* tok, const int32_t* last, int olen, const int32_t* olen_dev,
                                         const float* alive, float* r_out, int Tmax, int V, int blank, void* stream) {
  if (!logp || !lens || !r_prev || !parent || !tok || !last || !alive || !r_out || nutt <= 0 || per_utt <= 0 || Tmax <= 0 || V <= 0 ||
      (!olen_dev && olen < 0))
    return EAMD_EINVAL;
  const int* ol_dev = olen_dev;
  const int n = nutt * per_utt;
  static const int state_scan = getenv("EAMD_CTC_STATE_SCAN") ? atoi(getenv("EAMD_CTC_STATE_SCAN")) : 1;     // A/B knob: 0 = frame by frame
  if (state_scan && Tmax <= 2048) {
#define EAMD_SCAN_(Q) hipLaunchKernelGGL(ctc_prefix_state_scan_kernel<Q>, dim3(n), dim3(64), 0, (hipStream_t)stream, logp, lens, per_utt, r_prev, \
                                         (const long long*)parent, (const long long*)tok, last, olen, alive, r_out, n, Tmax, V, blank, ol_dev)
    if (Tmax <= 512) EAMD_SCAN_(8);
    else if (Tmax <= 1024) EAMD_SCAN_(16);
    else EAMD_SCAN_(32);
#undef EAMD_SCAN_
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
  hipLaunchKernelGGL(ctc_prefix_state_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, logp, lens, per_utt, r_prev,
                     (const long long*)parent, (const long long*)tok, last, olen, alive, r_out, n, Tmax, V, blank, ol_dev);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int eamd_ctc_prefix_score_batch(const float* logp, const int32_t* lens, int nutt, int per_utt, const float* r_prev,
                                           const int32_t* cand, const int32_t* last, const int32_t* olen, float* psi,
                                           float* r_new, int ncand, int Tmax, int V, int blank, int eos, void* stream) {
  if (!logp || !lens || !r_prev || !cand || !last || !olen || !psi || !r_new || nutt <= 0 || per_utt <= 0 || ncand <= 0 ||
      Tmax <= 0 || V <= 0)
    return EAMD_EINVAL;
  hipLaunchKernelGGL(ctc_prefix_batch_kernel, dim3((ncand + 63) / 64, nutt * per_utt), dim3(64), 0, (hipStream_t)stream, logp,
                     lens, per_utt, r_prev, cand, last, olen, psi, r_new, Tmax, V, ncand, blank, eos);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int eamd_ctc_prefix_score(const float* logp, const float* r_prev, const int32_t* cand, const int32_t* last,
                                     const int32_t* olen, float* psi, float* r_new, int nhyp, int ncand, int T, int V,
                                     int blank, int eos, void* stream) {
  if (!logp || !r_prev || !cand || !last || !olen || !psi || !r_new || nhyp <= 0 || ncand <= 0 || T <= 0 || V <= 0)
    return EAMD_EINVAL;
  hipLaunchKernelGGL(ctc_prefix_kernel, dim3((ncand + 63) / 64, nhyp), dim3(64), 0, (hipStream_t)stream, logp, r_prev,
                     cand, last, olen, psi, r_new, T, V, ncand, blank, eos);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
