// Fused position-wise feed-forward kernels for gfx950, bf16 operands (v_mfma_f32_16x16x32_bf16, fp32 accumulation): the
// bf16-operand mode's twin of ffn_f32.hip.
//
// reference: espnet/nets/pytorch_backend/transformer/positionwise_feed_forward.py:12-32
//     forward   out = R + alpha * drop_out( drop_in(act(x W1^T + b1)) W2^T + b2 )       (x, W, h, f bf16; R, out fp32)
//     backward  dz = alpha * (dy W2) (.) f,   dx = dz W1
//
// At the bf16 matrix rate the two products of 32 rows take 6.8 us per workgroup; what bounds the kernel is streaming the
// 2 MB of weights into every CU (L2 -> CU, ~14 us), the 65 MB of h / f written for backward, and the epilogue arithmetic.
// So, unlike the fp32 kernel (two wave roles, because there the matrix pipe is the bottleneck), all eight waves do the same
// work - an eighth of the first product, of its epilogue and of the second product per chunk of 128 hidden units:
//   * the 32 input rows are kept in LDS for the whole launch (A fragments read per k-step);
//   * weights go straight from global memory into MFMA B fragments (16 bytes along k per lane); BOTH
//     products' fragments of chunk c + 1 are requested at the top of chunk c (a whole chunk to arrive);
//   * wave w forms z[32 rows, 16 hidden units w*16..] (16 MFMAs), applies bias / activation / dropout to its 8 accumulator
//     values per lane and leaves h (and f) as bf16 in LDS; ONE barrier per chunk; then every thread stores one 16-byte piece
//     of h and of f to global memory (whole row pieces instead of 2-byte scatters) and wave w accumulates
//     out[32 rows, 32 columns w*32..] += h_chunk W2[:, chunk]^T (16 MFMAs, A fragments from LDS).
// Backward runs the SAME kernel on the packed images of the transposed weights (eamd_ffn_pack_bf16 makes all four images
// of a layer in one launch); its epilogue is dz = alpha * acc * f with f staged through LDS.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"
#include "ffn_ln.h"

namespace {

typedef unsigned short bf16_t;
constexpr int BBM = 32, BD = 256, BHC = 128, BNT = 512;
constexpr int HLD = BHC + 8;                   // bf16 elements per LDS row of a chunk image (272 bytes: 16-byte aligned rows)
constexpr int HSZ = BBM * HLD;                 // elements per chunk image
constexpr int YLD = BD + 4;                    // fp32 staging of the output rows
constexpr int XLDB = BD + 8;                   // bf16 elements per LDS row of the input image
constexpr size_t B_CHUNKS = (size_t)4 * HSZ * 2, B_XS = (size_t)BBM * XLDB * 2;
constexpr size_t B_SMEM = (B_CHUNKS + B_XS) > (size_t)BBM * YLD * 4 ? (B_CHUNKS + B_XS) : (size_t)BBM * YLD * 4;

__device__ __forceinline__ float bf2f_(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

template <bool BWD, int ACT>
__global__ __launch_bounds__(BNT, 2) void ffn_bf16_kernel(const eamd_ffn_t p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16_t* const hs = reinterpret_cast<bf16_t*>(smem_raw);          // [2][32][HLD]: h (forward) / dz (backward)
  bf16_t* const fs = hs + 2 * HSZ;                                  // [2][32][HLD]: f (forward: out; backward: in)
  bf16_t* const xs = fs + 2 * HSZ;                                  // [32][XLDB]: the input rows
  const int t = threadIdx.x;
  const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * BBM;
  const int F = p.F;
  const int nch = F / BHC;
  const bool full_rows = m0 + BBM <= p.M;
  const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* __restrict__ Wa = reinterpret_cast<const bf16_t*>(p.w1);      // [F][256]: rows = hidden units
  const bf16_t* __restrict__ Wb = reinterpret_cast<const bf16_t*>(p.w2);      // [256][F]: rows = outputs
  bf16_t* __restrict__ Hg = reinterpret_cast<bf16_t*>(p.h);
  bf16_t* __restrict__ Fg = reinterpret_cast<bf16_t*>(p.f);

  // ---- the 32 input rows: LDS image, A fragments (lane = row fr of row tile i, k-group fq: 8 consecutive k) read per k-step
  // (held in registers they cost 64 VGPRs - the room the one-chunk-ahead prefetch of BOTH weight sets needs) ----
  if (!BWD && p.ln_x) {        // LayerNorm in front: normalise the fp32 rows on their way into the bf16 LDS image (ffn_ln.h)
    ffn_ln_stage(p, m0, t, [&](int row, int col, float4 y) __attribute__((always_inline)) {
      uint2 hh;
      hh.x = eamd_f2bf(y.x) | ((unsigned)eamd_f2bf(y.y) << 16);
      hh.y = eamd_f2bf(y.z) | ((unsigned)eamd_f2bf(y.w) << 16);
      *reinterpret_cast<uint2*>(&xs[row * XLDB + col]) = hh;
    });
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = t + BNT * k, row = idx >> 5, c8 = idx & 31;          // 32 rows x 32 pieces of 8 elements
      *reinterpret_cast<uint4*>(&xs[row * XLDB + c8 * 8]) =
          *reinterpret_cast<const uint4*>(X + (long)min(m0 + row, p.M - 1) * BD + c8 * 8);
    }
  }
  // ---- weight fragments, from the PACKED images eamd_ffn_pack_bf16 makes (fragment order: one wave-instruction reads 1 KB of
  // consecutive bytes = 8 whole cache lines; fetched from the nn.Linear layout a fragment load touches 16 rows x 64 bytes -
  // half of 16 lines - and the kernel sat at the vector memory path's line rate: 72 us, no better than the two GEMMs) ----
  //   first product:  Pa[c][w][ks][lane] = 8 k-elements (ks*32 + fq*8 ..) of hidden unit c*128 + w*16 + fr
  //   second product: Pb[c][w][j][ks][lane] = 8 k-elements (hidden c*128 + ks*32 + fq*8 ..) of output w*32 + j*16 + fr
  const unsigned aoff = (unsigned)((w * 8 * 64 + lane) * 16);
  const unsigned boff = (unsigned)((w * 8 * 64 + lane) * 16);
  uint4 wu[2][8], wd[2][2][4];      // [register set][k-step] and [register set][column tile][k-step]
  auto load_up = [&](auto set_c, int c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const char* base = reinterpret_cast<const char*>(Wa) + (long)min(c, nch - 1) * (8 * 8 * 64 * 16);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wu[SET][ks] = *reinterpret_cast<const uint4*>(base + ks * 1024 + aoff);
  };
  auto load_down = [&](auto set_c, int c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const char* base = reinterpret_cast<const char*>(Wb) + (long)min(c, nch - 1) * (8 * 8 * 64 * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wd[SET][j][ks] = *reinterpret_cast<const uint4*>(base + (j * 4 + ks) * 1024 + boff);
  };
  // one 16-byte piece of a [32 x 128] bf16 chunk image per thread: row t / 16, columns (t % 16) * 8 ..
  const int pr = t >> 4, pc = (t & 15) * 8;
  const bool prow_ok = full_rows || m0 + pr < p.M;
  const long pgo = (long)min(m0 + pr, p.M - 1) * F + pc;

  f32x4 yacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned thr_in = eamd_drop_thr16(p.p_in);
  const float inv_in = p.p_in > 0.f ? eamd_drop_inv(thr_in) : 1.f;
  const unsigned seed_in = (!BWD && p.p_in > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_in) : 0u;

  float bias_next = (!BWD && p.b1) ? p.b1[w * 16 + fr] : 0.f;       // b1 of chunk 0; chunk c requests chunk c + 1's
  auto chunk = [&](auto set_c, int c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    bf16_t* hb = hs + (c & 1) * HSZ;
    bf16_t* fb = fs + (c & 1) * HSZ;
    // requests: the second product's weights of this chunk, the first product's of the next; backward: the factor piece of
    // the next chunk (global -> register now, -> LDS in front of this chunk's barrier)
    // (vmcnt retires in order: the small loads go FIRST, so that waiting for them does not wait for the weight requests
    // behind them - a bias load behind the weight loads cost a full memory round trip per chunk: 88 -> 73 us)
    uint4 fnext = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (BWD) fnext = *reinterpret_cast<const uint4*>(Fg + pgo + (long)min(c + 1, nch - 1) * BHC);
    const float bias = bias_next;
    if constexpr (!BWD) bias_next = p.b1 ? p.b1[min(c + 1, nch - 1) * BHC + w * 16 + fr] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    load_up(std::integral_constant<int, SET ^ 1>{}, c + 1);          // both weight sets of the NEXT chunk: a whole chunk to arrive
    load_down(std::integral_constant<int, SET ^ 1>{}, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- first product: z[32, 16] of this wave ----
    f32x4 z[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint4 xa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) xa[i] = *reinterpret_cast<const uint4*>(&xs[(i * 16 + fr) * XLDB + ks * 32 + fq * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        z[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xa[i]), __builtin_bit_cast(bf16x8, wu[SET][ks]),
                                                       z[i], 0, 0, 0);
    }
    // ---- epilogue on the accumulators: element (i, r) = row i*16 + fq*4 + r, hidden unit w*16 + fr of the chunk ----
    const int lc = w * 16 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = i * 16 + fq * 4 + r;
        float hv, fv = 0.f;
        if constexpr (!BWD) {
          eamd_act_dact(z[i][r] + bias, ACT, hv, fv);
          if (p.p_in > 0.f) {
            const unsigned gi = (unsigned)(m0 + lr) * (unsigned)F + (unsigned)(c * BHC + lc);
            const bool keep = eamd_drop_keep(seed_in, (unsigned long long)gi, thr_in);
            hv = keep ? hv * inv_in : 0.f;
            fv = keep ? fv * inv_in : 0.f;
          }
          fb[lr * HLD + lc] = eamd_f2bf(fv);
        } else {
          hv = (z[i][r] * bf2f_(fb[lr * HLD + lc])) * p.alpha;
        }
        hb[lr * HLD + lc] = eamd_f2bf(hv);
      }
    if constexpr (BWD) *reinterpret_cast<uint4*>(&fs[((c + 1) & 1) * HSZ + pr * HLD + pc]) = fnext;
    __syncthreads();
    // ---- copies for backward as whole 16-byte row pieces ----
    if (prow_ok) {
      if (Hg) *reinterpret_cast<uint4*>(Hg + pgo + (long)c * BHC) = *reinterpret_cast<const uint4*>(&hb[pr * HLD + pc]);
      if constexpr (!BWD) { if (Fg) *reinterpret_cast<uint4*>(Fg + pgo + (long)c * BHC) = *reinterpret_cast<const uint4*>(&fb[pr * HLD + pc]); }
    }
    // ---- second product: out[32, 32] of this wave += h_chunk W2[:, chunk]^T ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 ha[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) ha[i] = *reinterpret_cast<const uint4*>(&hb[(i * 16 + fr) * HLD + ks * 32 + fq * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          yacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ha[i]), __builtin_bit_cast(bf16x8, wd[SET][j][ks]),
                                                               yacc[i][j], 0, 0, 0);
    }
  };

  load_up(std::integral_constant<int, 0>{}, 0);
  load_down(std::integral_constant<int, 0>{}, 0);
  if constexpr (BWD) *reinterpret_cast<uint4*>(&fs[pr * HLD + pc]) = *reinterpret_cast<const uint4*>(Fg + pgo);      // the factor piece of chunk 0
  __syncthreads();                                  // the input image (and the factor piece) are in LDS
  for (int c = 0; c < nch; c += 2) {       // nch is even (F % 256 == 0)
    chunk(std::integral_constant<int, 0>{}, c);
    chunk(std::integral_constant<int, 1>{}, c + 1);
  }
  __syncthreads();                                  // every wave is done with the chunk images: the staging tile goes over them
  float* ys = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ys[(i * 16 + fq * 4 + r) * YLD + w * 32 + j * 16 + fr] = yacc[i][j][r];
  __syncthreads();
  const unsigned thr_out = eamd_drop_thr16(p.p_out);
  const float inv_out = eamd_drop_inv(thr_out);
  const unsigned seed_out = (!BWD && p.p_out > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_out) : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + BNT * i, lr = idx >> 6, c4 = idx & 63;
    const int row = m0 + lr;
    if (row >= p.M) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(&ys[lr * YLD + c4 * 4]);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const long gi = (long)row * BD + c4 * 4;
    if constexpr (!BWD) {
      if (p.b2) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.b2 + c4 * 4);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      if (p.p_out > 0.f) {
        bool keep[4];
        eamd_drop_keep4(seed_out, (unsigned long long)gi, thr_out, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * inv_out : 0.f;
      }
      float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.R) r4 = *reinterpret_cast<const float4*>(p.R + gi);
      v[0] = v[0] * p.alpha + r4.x; v[1] = v[1] * p.alpha + r4.y; v[2] = v[2] * p.alpha + r4.z; v[3] = v[3] * p.alpha + r4.w;
    }
    *reinterpret_cast<float4*>(p.out + gi) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// The four packed weight images of one FFN (see the kernel): one thread per 16-byte fragment piece.
//   which 0: forward first product  <- W1[n = c*128 + w*16 + fr][k = ks*32 + fq*8 + e]
//   which 1: forward second product <- W2[n = w*32 + j*16 + fr][k = c*128 + ks*32 + fq*8 + e]
//   which 2: backward first product  (dh = dy W2: n = hidden unit, k = output)  <- W2[k][n]
//   which 3: backward second product (dx = dz W1: n = input column, k = hidden unit) <- W1[k][n]
__global__ __launch_bounds__(256) void ffn_pack_bf16_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                            bf16_t* __restrict__ p0, bf16_t* __restrict__ p1,
                                                            bf16_t* __restrict__ p2, bf16_t* __restrict__ p3, int F) {
  const int which = blockIdx.y;
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;           // (c, w, slot 0..7, lane)
  const long npiece = (long)(F / BHC) * 8 * 8 * 64;
  if (piece >= npiece) return;
  const int lane = piece & 63, slot = (piece >> 6) & 7, w = (piece >> 9) & 7, c = (int)(piece >> 12);
  const int fr = lane & 15, fq = lane >> 4;
  unsigned short v[8];
  if (which == 0 || which == 2) {
    const int n = c * BHC + w * 16 + fr, k0 = slot * 32 + fq * 8;     // slot = ks
    if (which == 0) {
      const uint4 u = *reinterpret_cast<const uint4*>(w1 + (long)n * BD + k0);
      *reinterpret_cast<uint4*>(p0 + piece * 8) = u;
      return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w2[(long)(k0 + e) * F + n];
    uint4 u;
    u.x = v[0] | ((unsigned)v[1] << 16); u.y = v[2] | ((unsigned)v[3] << 16); u.z = v[4] | ((unsigned)v[5] << 16); u.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(p2 + piece * 8) = u;
  } else {
    const int j = slot >> 2, ks = slot & 3;
    const int n = w * 32 + j * 16 + fr, k0 = c * BHC + ks * 32 + fq * 8;
    if (which == 1) {
      const uint4 u = *reinterpret_cast<const uint4*>(w2 + (long)n * F + k0);
      *reinterpret_cast<uint4*>(p1 + piece * 8) = u;
      return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w1[(long)(k0 + e) * BD + n];
    uint4 u;
    u.x = v[0] | ((unsigned)v[1] << 16); u.y = v[2] | ((unsigned)v[3] << 16); u.z = v[4] | ((unsigned)v[5] << 16); u.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(p3 + piece * 8) = u;
  }
}

template <bool BWD, int ACT>
int launch_b(const eamd_ffn_t& p, hipStream_t stream) {
  const int nblk = (p.M + BBM - 1) / BBM;
  hipLaunchKernelGGL((ffn_bf16_kernel<BWD, ACT>), dim3(nblk), dim3(BNT), B_SMEM, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // namespace

bool eamd_ffn_bf16_ok(const eamd_ffn_t* p) {
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return p->D == BD && p->F % 256 == 0 && p->F >= 256 && (long)p->M * p->F < (1L << 31) && al(p->x) && al(p->w1) && al(p->w2) &&
         al(p->out) && (!p->R || al(p->R)) && (!p->b2 || al(p->b2)) && (!p->f || al(p->f)) && (!p->h || al(p->h));
}
int eamd_ffn_bf16_launch(const eamd_ffn_t* p, int bwd, void* stream) {
  if (bwd) return launch_b<true, EAMD_ACT_NONE>(*p, (hipStream_t)stream);
  return p->act == EAMD_ACT_SWISH ? launch_b<false, EAMD_ACT_SWISH>(*p, (hipStream_t)stream)
                                  : launch_b<false, EAMD_ACT_RELU>(*p, (hipStream_t)stream);
}

extern "C" int eamd_ffn_pack_bf16(const void* w1, const void* w2, void* fwd_first, void* fwd_second, void* bwd_first,
                                  void* bwd_second, int D, int F, void* stream) {
  if (!w1 || !w2 || !fwd_first || !fwd_second || !bwd_first || !bwd_second || F <= 0) return EAMD_EINVAL;
  if (D != BD || F % 256 != 0) return EAMD_EUNSUPPORTED;
  for (const void* q : {w1, w2, (const void*)fwd_first, (const void*)fwd_second, (const void*)bwd_first, (const void*)bwd_second})
    if (reinterpret_cast<uintptr_t>(q) & 15) return EAMD_EUNSUPPORTED;
  const long npiece = (long)(F / BHC) * 8 * 8 * 64;
  hipLaunchKernelGGL(ffn_pack_bf16_kernel, dim3((unsigned)((npiece + 255) / 256), 4), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w1, (const bf16_t*)w2, (bf16_t*)fwd_first, (bf16_t*)fwd_second, (bf16_t*)bwd_first,
                     (bf16_t*)bwd_second, F);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
