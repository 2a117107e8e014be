// Fused attention forward for bf16 activations, d_k = 64, up to 256 keys: scores (+ relative-position term with the
// legacy rel_shift), mask, softmax and the context product in ONE kernel; the fp32 scores never reach HBM.
// reference: transformer/attention.py:63-114 (forward_attention / MultiHeadedAttention.forward),
//            attention.py:141-206 (RelPositionMultiHeadedAttention: rel_shift, (ac + bd) / sqrt(d_k)).
//
// One workgroup = one (batch, head) pair x 64 queries, four waves; everything is computed in the transposed
// orientation S^T = K Q^T:
//   * the 16x16x32 MFMA operand layout of a k-contiguous matrix is 16 bytes of global memory per lane, so the K, q,
//     positional and (q+v) fragments are loaded straight into registers (no operand staging);
//   * SCORE phases split the KEYS over the waves (all 64 queries each: every K / position row is fetched by one wave
//     only) and leave their tiles in an fp32 LDS score matrix X[query][key], the legacy rel_shift applied while the bd
//     tiles are stored (see attn_fwd_kernel); for SOFTMAX and the products with P wave w owns queries 16 w .. 16 w + 15:
//     it reads its X rows back in the accumulator layout - four ADJACENT keys per lane = an 8-byte bf16 store into P,
//     and exactly the pairing the context product needs: with the (arbitrary) contraction order "keys 32 s + 4 q +
//     {0..3}, then 32 s + 16 + 4 q + {0..3}" the probabilities of tiles 2s, 2s+1 ARE the MFMA operand.  The V fragments
//     of the same order come from ds_read_b64_tr_b16 on a V panel staged over X once the scores are in registers;
//   * the context is accumulated as C^T = V^T P^T, which leaves four adjacent channels per lane (8-byte stores).
#include "gemm_bf16_common.h"

namespace {

constexpr int ATT_DK = 64;
constexpr int ATT_MAXK = 512;                 // keys per row: 8 / 16 / 32 accumulator tiles of 16 (32: one workgroup per CU)

struct AttnArgs {
  const bf16_t* qu; const bf16_t* qv; const bf16_t* k; const bf16_t* v; const bf16_t* pos;
  const unsigned char* mask;
  bf16_t* P; bf16_t* ctx;
  long ldq, ldqv, ldk, ldv, ldpos, ldc, ldp, mb, mi;
  int B, H, T1, T2, nqb;
  float scale;
  // attention dropout (attention.py:91): Pd = dropout(P) feeds the context; mask index = element index in P
  bf16_t* Pd; float drop_p; const unsigned long long* drop_step; unsigned long long drop_salt;
  const int* tshift;     // device scalar: length the legacy rel_shift works on (NULL: T2) - see eamd_attn_fwd
};

__device__ __forceinline__ float xor_max16_32(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor_sum16_32(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// NKT = key tiles of 16 the instantiation covers (T2 <= 16 NKT).  Every loop over tiles is fully unrolled and free of
// branches: rows past T2 / T1 are clamped re-reads whose scores are masked / never stored.
// Score phases (bd, ac): wave w takes a quarter of the KEY tiles and all 64 queries of the workgroup, so every K /
// position row is fetched by one wave only (round 1 had every wave fetch all keys for its own 16 queries, and computed
// the bd tiles of 32 queries to get 17 rows); the tiles S^T[key][query] go to an fp32 LDS score matrix X[query][key]
// (16-byte stores: an accumulator holds four adjacent keys of one query).  The legacy rel_shift is one-to-one -
// shifted[i][j] = bd[i][T-1-i+j] for j <= i, 0 for j = i + 1, bd[i+1][j-i-2] above - so the bd tiles are STORED at their
// shifted place and the ac tiles added by read-modify-write of 16-byte rows behind a barrier; bd row r0w + 64 (first query
// of the next workgroup) feeds query r0w + 63: 64-term dot products on the VALU, one position per thread.
template <int NKT>
__device__ __forceinline__ void score_tiles_bf16(const bf16_t* keys, long ldk, int T2, const bf16_t* qs, long ldq, int nq,
                                                 float* X, int XLD, int wave, int fr, int fq, bool acc) {
  constexpr int TPW = NKT / 4;
  uint4 qf[4][2];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const bf16_t* qr = qs + (long)min(qt * 16 + fr, nq - 1) * ldq + fq * 8;
    qf[qt][0] = *reinterpret_cast<const uint4*>(qr);
    qf[qt][1] = *reinterpret_cast<const uint4*>(qr + 32);
  }
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int kt = wave * TPW + u;
    const bf16_t* kr = keys + (long)min(kt * 16 + fr, T2 - 1) * ldk + fq * 8;
    const uint4 k0 = *reinterpret_cast<const uint4*>(kr), k1 = *reinterpret_cast<const uint4*>(kr + 32);
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
      float4* xp = reinterpret_cast<float4*>(&X[(qt * 16 + fr) * XLD + kt * 16 + 4 * fq]);
      if (acc) {                      // on top of the shifted bd values (one owner per element: plain read-modify-write)
        const float4 o = *xp;
        c = (f32x4){o.x, o.y, o.z, o.w};
      }
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k0), __builtin_bit_cast(bf16x8, qf[qt][0]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k1), __builtin_bit_cast(bf16x8, qf[qt][1]), c, 0, 0, 0);
      *xp = make_float4(c[0], c[1], c[2], c[3]);      // c[r]: key 16 kt + 4 fq + r, query 16 qt + fr
    }
  }
}

__device__ __forceinline__ float dot8_bf16(uint4 x, uint4 y) {
  const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    s += __uint_as_float(xs[i] << 16) * __uint_as_float(ys[i] << 16) +
         __uint_as_float(xs[i] & 0xffff0000u) * __uint_as_float(ys[i] & 0xffff0000u);
  return s;
}

template <bool REL, int NKT>
__global__ __launch_bounds__(256, NKT > 16 ? 1 : 2) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int XLD = NKT * 16 + 4;
  constexpr int TPW = NKT / 4;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // (batch, head) pairs are dealt to the XCDs in contiguous runs: the query blocks of one pair share an L2
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nqb) * 8 + xcd;            // z = h * B + b, as the score tensors are laid out
  const bool live = z < a.B * a.H;                 // whole workgroup; dead ones only keep the barriers company
  const int zz = live ? z : 0;
  const int h = zz / a.B, b = zz % a.B;
  const int T1 = a.T1, T2 = a.T2;
  const int r0w = min((jb % a.nqb) * 64, T1 - 1);  // first query of the workgroup
  const int nq = T1 - r0w;                         // its queries (up to 64)
  const int r0 = r0w + wave * 16;                  // first query this wave owns in the softmax / context phases
  const bool active = live && r0 < T1;
  const int qi = min(r0 + fr, T1 - 1);             // this lane's query (clamped lanes are never stored)

  float* X = reinterpret_cast<float*>(smem_raw);
  // Ts: the length the legacy rel_shift is taken over.  A batch padded beyond its own longest utterance (shape-bucketed
  // graphs) passes that utterance's length here: the shift then maps bd exactly as the reference's T' x T' matrix does;
  // keys from Ts on are masked by the caller's mask, and the score matrix is cleared first so that the rows / columns the
  // shift no longer reaches hold zeros instead of stale LDS.
  const int Ts = (REL && a.tshift) ? min(max(a.tshift[0], 1), T2) : T2;
  if (REL) {
    if (Ts < T2) {
      for (int e = t; e < 64 * XLD; e += 256) X[e] = 0.f;
      __syncthreads();
    }
    // ---- bd, stored at its rel-shifted place: bd[i][m] -> (i, m - (T-1-i)) if m >= T-1-i, else (i - 1, m + i + 1) ----
    uint4 qf[4][2];
    const bf16_t* qs = a.qv + ((long)b * T1 + r0w) * a.ldqv + h * ATT_DK;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const bf16_t* qr = qs + (long)min(qt * 16 + fr, nq - 1) * a.ldqv + fq * 8;
      qf[qt][0] = *reinterpret_cast<const uint4*>(qr);
      qf[qt][1] = *reinterpret_cast<const uint4*>(qr + 32);
    }
    if (t < 64 && r0w + t + 1 < T2) X[t * XLD + r0w + t + 1] = 0.f;      // (i, i + 1): the zero column of the padding
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int mt = wave * TPW + u;
      const bf16_t* pr = a.pos + (long)min(mt * 16 + fr, T2 - 1) * a.ldpos + h * ATT_DK + fq * 8;
      const uint4 p0 = *reinterpret_cast<const uint4*>(pr), p1 = *reinterpret_cast<const uint4*>(pr + 32);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p0), __builtin_bit_cast(bf16x8, qf[qt][0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, p1), __builtin_bit_cast(bf16x8, qf[qt][1]), c, 0, 0, 0);
        const int ql = qt * 16 + fr, i = r0w + ql;       // c[r] = bd[query i][m = 16 mt + 4 fq + r]
        const int lim = Ts - 1 - i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mt * 16 + 4 * fq + r;
          const int row = m >= lim ? ql : ql - 1, j = m >= lim ? m - lim : m + i + 1;
          if (m < Ts && i < Ts && ql < nq && row >= 0) X[row * XLD + j] = c[r];
        }
      }
    }
    const int i64 = r0w + 64;                        // first query of the next workgroup: its low positions feed query r0w + 63
    if (live && i64 < Ts) {
      const uint4* qr = reinterpret_cast<const uint4*>(a.qv + ((long)b * T1 + i64) * a.ldqv + h * ATT_DK);
      for (int m = t; m <= Ts - 2 - i64; m += 256) {
        const uint4* pr = reinterpret_cast<const uint4*>(a.pos + (long)m * a.ldpos + h * ATT_DK);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += dot8_bf16(qr[j], pr[j]);
        X[63 * XLD + m + i64 + 1] = s;
      }
    }
    __syncthreads();
  }
  // ---- ac: X[q][j] (+)= qu_q . k_j ----
  score_tiles_bf16<NKT>(a.k + (long)b * T2 * a.ldk + h * ATT_DK, a.ldk, T2, a.qu + ((long)b * T1 + r0w) * a.ldq + h * ATT_DK,
                        a.ldq, nq, X, XLD, wave, fr, fq, REL);
  // V panel [16 NKT keys][64 channels]: global loads now, LDS stores once the scores have left X
  uint4 vreg[NKT / 2];
#pragma unroll
  for (int q = 0; q < NKT / 2; ++q) {
    const int c = t + 256 * q, row = c >> 3, c16 = c & 7;
    vreg[q] = *reinterpret_cast<const uint4*>(a.v + ((long)b * T2 + min(row, T2 - 1)) * a.ldv + h * ATT_DK + c16 * 8);
    if (row >= T2) vreg[q] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();                                  // X is complete
  // ---- scale, rel-shift term, mask, softmax over the keys of query (r0 + fr) ----
  uint2 Pk[NKT];                                    // four bf16 probabilities per tile
  if (active) {
    f32x4 S[NKT];
    // mask bytes of this lane's keys: unconditional clamped loads (a load behind a lane-dependent guard is waited
    // for on the spot), all of them in flight together; keys past T2 are masked by index below
    unsigned mk[NKT];
    if (a.mask) {
      const unsigned char* mr = a.mask + (long)b * a.mb + (long)qi * a.mi;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int j0 = kt * 16 + fq * 4;
        const unsigned m0 = mr[min(j0, T2 - 1)], m1 = mr[min(j0 + 1, T2 - 1)], m2 = mr[min(j0 + 2, T2 - 1)],
                       m3 = mr[min(j0 + 3, T2 - 1)];
        mk[kt] = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) mk[kt] = 0x01010101u;
    }
    const float* xrow = &X[(wave * 16 + fr) * XLD + 4 * fq];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int j0 = kt * 16 + fq * 4;
      const float4 xv = *reinterpret_cast<const float4*>(xrow + kt * 16);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = xs[r] * a.scale;
        if (j0 + r >= T2 || ((mk[kt] >> (8 * r)) & 0xffu) == 0u) x = -INFINITY;
        S[kt][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = xor_max16_32(mx);
    const bool dead = mx == -INFINITY;              // every key masked: softmax(min, ...) = uniform, then masked_fill(0)
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = dead ? 0.f : __expf(S[kt][r] - mx);
        S[kt][r] = e;
        sum += e;
      }
    sum = xor_sum16_32(sum);
    const float inv = dead ? 0.f : 1.f / sum;
    const bool qok = r0 + fr < T1;
    const long pro = ((long)zz * T1 + qi) * a.ldp;
    bf16_t* prow = a.P + pro;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      Pk[kt].x = (unsigned)eamd_f2bf(S[kt][0] * inv) | ((unsigned)eamd_f2bf(S[kt][1] * inv) << 16);
      Pk[kt].y = (unsigned)eamd_f2bf(S[kt][2] * inv) | ((unsigned)eamd_f2bf(S[kt][3] * inv) << 16);
      const int j0 = kt * 16 + fq * 4;
      if (qok && j0 < a.ldp) *reinterpret_cast<uint2*>(prow + j0) = Pk[kt];     // pad columns receive zeros
    }
    if (a.drop_p > 0.f) {                            // wave-uniform: the dropped probabilities go on to the context product
      const unsigned seed = eamd_drop_seed(a.drop_step, a.drop_salt), thr = eamd_drop_thr16(a.drop_p);
      const float dinv = eamd_drop_inv(thr) * inv;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int j0 = kt * 16 + fq * 4;
        bool kp[4];
        eamd_drop_keep4(seed, (unsigned long long)(pro + min(j0, (int)a.ldp - 4)), thr, kp);
        Pk[kt].x = (unsigned)eamd_f2bf(kp[0] ? S[kt][0] * dinv : 0.f) | ((unsigned)eamd_f2bf(kp[1] ? S[kt][1] * dinv : 0.f) << 16);
        Pk[kt].y = (unsigned)eamd_f2bf(kp[2] ? S[kt][2] * dinv : 0.f) | ((unsigned)eamd_f2bf(kp[3] ? S[kt][3] * dinv : 0.f) << 16);
        if (qok && j0 < a.ldp) *reinterpret_cast<uint2*>(a.Pd + pro + j0) = Pk[kt];
      }
    }
  } else {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) Pk[kt] = make_uint2(0u, 0u);
  }
  __syncthreads();                                  // every wave has its scores in registers
  // ---- V panel -> LDS (k-strided image, rows past T2 are zero) ----
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem_raw);
#pragma unroll
  for (int q = 0; q < NKT / 2; ++q) {
    const int c = t + 256 * q, row = c >> 3, c16 = c & 7;
    *reinterpret_cast<uint4*>(&Vs[lds_chunk_off<true, 64>(row, c16)]) = vreg[q];
  }
  __syncthreads();
  if (!active) return;
  // ---- context^T = V^T P^T: rows = channels, columns = queries ----
  f32x4 C[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) C[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < NKT / 2; ++ks) {
    {
      const uint4 pw = make_uint4(Pk[2 * ks].x, Pk[2 * ks].y, Pk[2 * ks + 1].x, Pk[2 * ks + 1].y);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      const int rlo = ks * 32 + 4 * fq + (fr >> 2), rhi = rlo + 16;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int cc = dt * 16 + 4 * (fr & 3);
        const bf16_t* q0 = &Vs[lds_chunk_off<true, 64>(rlo, cc >> 3) + (cc & 7)];
        const bf16_t* q1 = &Vs[lds_chunk_off<true, 64>(rhi, cc >> 3) + (cc & 7)];
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q1);
        const bf16x8 vf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        C[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, C[dt], 0, 0, 0);
      }
    }
  }
  if (r0 + fr < T1) {
    bf16_t* crow = a.ctx + ((long)b * T1 + r0 + fr) * a.ldc + h * ATT_DK;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 o;
      o.x = (unsigned)eamd_f2bf(C[dt][0]) | ((unsigned)eamd_f2bf(C[dt][1]) << 16);
      o.y = (unsigned)eamd_f2bf(C[dt][2]) | ((unsigned)eamd_f2bf(C[dt][3]) << 16);
      *reinterpret_cast<uint2*>(crow + dt * 16 + fq * 4) = o;
    }
  }
}

// ---- backward, query side: dP = dctx V^T, dS = scale * P (dP - rowsum(P dP)), dq = dS K in one launch ----
// Same orientation and work splits as the forward: dP^T tiles by key quarter into the LDS matrix X, P re-read with
// the 8-byte accesses the forward stored it with, the row sum inside the wave, dS (bf16) stored for the key-side
// GEMMs (dK = dS^T q, and with relative positions dqv / dpos from the inverse rel_shift scatter dbd, written here
// element by element exactly as eamd_softmax_bwd does) and dq^T = K^T dS^T from the K panel in LDS - the context
// product of the forward with K in place of V.  reference: autograd of attention.py:63-114, :141-206.
struct AttnBwdArgs {
  const bf16_t* dctx; const bf16_t* k; const bf16_t* v; const bf16_t* P;
  bf16_t* dS; bf16_t* dbd; void* dq;
  long ldd, ldk, ldv, ldp, ldo;
  int B, H, T1, T2, nqb, dq_bf16;
  float scale;
  float drop_p; const unsigned long long* drop_step; unsigned long long drop_salt;   // the forward's attention dropout
  const int* tshift;
};

template <int NKT>
__global__ __launch_bounds__(256, NKT > 16 ? 1 : 2) void attn_bwd_q_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nqb) * 8 + xcd;
  const bool live = z < a.B * a.H;
  const int zz = live ? z : 0;
  const int h = zz / a.B, b = zz % a.B;
  constexpr int XLD = NKT * 16 + 4;
  const int T1 = a.T1, T2 = a.T2;
  const int Ts = (a.dbd && a.tshift) ? min(max(a.tshift[0], 1), T2) : T2;      // rel_shift length (see the forward kernel)
  const int r0w = min((jb % a.nqb) * 64, T1 - 1);
  const int nq = T1 - r0w;
  const int r0 = r0w + wave * 16;
  const bool active = live && r0 < T1;
  const int qi = min(r0 + fr, T1 - 1);
  const bool qok = r0 + fr < T1;

  // ---- dP: X[q][j] = dctx_q . v_j, the key tiles split over the waves (see attn_fwd_kernel) ----
  float* X = reinterpret_cast<float*>(smem_raw);
  score_tiles_bf16<NKT>(a.v + (long)b * T2 * a.ldv + h * ATT_DK, a.ldv, T2, a.dctx + ((long)b * T1 + r0w) * a.ldd + h * ATT_DK,
                        a.ldd, nq, X, XLD, wave, fr, fq, false);
  // this lane's probabilities: in flight across the barrier
  const bf16_t* prow = a.P + ((long)zz * T1 + qi) * a.ldp;
  uint2 Pr[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int j0 = kt * 16 + fq * 4;
    Pr[kt] = *reinterpret_cast<const uint2*>(prow + min(j0, (int)a.ldp - 4));
  }
  __syncthreads();
  uint2 Gk[NKT];                                    // four bf16 score gradients per tile
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) Gk[kt] = make_uint2(0u, 0u);
  if (active) {
    f32x4 S[NKT];
    const float* xrow = &X[(wave * 16 + fr) * XLD + 4 * fq];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const float4 xv = *reinterpret_cast<const float4*>(xrow + kt * 16);
      S[kt] = (f32x4){xv.x, xv.y, xv.z, xv.w};
    }
    // P is unpacked from its packed registers in both passes (a second fp32 copy costs 64 VGPRs and a wave per SIMD)
    auto unpack = [&](int kt, int r) __attribute__((always_inline)) -> float {
      const unsigned w = r < 2 ? Pr[kt].x : Pr[kt].y;
      const float f = __uint_as_float((r & 1) ? (w & 0xffff0000u) : (w << 16));
      return kt * 16 + fq * 4 < (int)a.ldp ? f : 0.f;   // pad columns of P hold zeros, columns past ldp do not exist
    };
    if (a.drop_p > 0.f) {                            // wave-uniform: gradient of the dropped probabilities -> gradient of P
      const unsigned seed = eamd_drop_seed(a.drop_step, a.drop_salt), thr = eamd_drop_thr16(a.drop_p);
      const float dinv = eamd_drop_inv(thr);
      const long pro = ((long)zz * T1 + qi) * a.ldp;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        bool kp[4];
        eamd_drop_keep4(seed, (unsigned long long)(pro + min(kt * 16 + fq * 4, (int)a.ldp - 4)), thr, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[kt][r] = kp[r] ? S[kt][r] * dinv : 0.f;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (kt * 16 + fq * 4 + r < T2) s += unpack(kt, r) * S[kt][r];
    s = xor_sum16_32(s);
    const long zo = (long)zz * T1 * a.ldp;
    bf16_t* srow = a.dS + zo + (long)qi * a.ldp;
    const int i = r0 + fr;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int j0 = kt * 16 + fq * 4;
      unsigned short g16[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + r;
        g16[r] = eamd_f2bf(j < T2 ? unpack(kt, r) * (S[kt][r] - s) * a.scale : 0.f);
        if (a.dbd && qok) {
          // inverse rel_shift for T1 == T2 = T (Ts when the shift is shorter than the padded matrix: everything outside the
          // Ts x Ts square is zero): padded index T + i T + j lands in row i (j <= i) or row i + 1
          if (j < Ts && i < Ts) {
            const int R = j <= i ? i : i + 1, c = j <= i ? Ts + j - i : j - i - 1;
            if (c != 0) a.dbd[zo + (long)R * a.ldp + (c - 1)] = g16[r];
          } else if (j < (int)a.ldp) {
            a.dbd[zo + (long)i * a.ldp + j] = 0;      // pad columns of this row
          }
        }
      }
      Gk[kt].x = (unsigned)g16[0] | ((unsigned)g16[1] << 16);
      Gk[kt].y = (unsigned)g16[2] | ((unsigned)g16[3] << 16);
      if (qok && j0 < (int)a.ldp) *reinterpret_cast<uint2*>(srow + j0) = Gk[kt];
    }
    if (a.dbd && r0 == 0)                            // the head of row 0 the scatter never reaches
      for (int f = 1 + lane; f < Ts; f += 64) a.dbd[zo + (f - 1)] = 0;
  }
  // K panel [16 NKT keys][64 channels]: loads in flight across the barrier, staged over X once every wave has its rows
  uint4 kreg[NKT / 2];
#pragma unroll
  for (int q = 0; q < NKT / 2; ++q) {
    const int c = t + 256 * q, row = c >> 3, c16 = c & 7;
    kreg[q] = *reinterpret_cast<const uint4*>(a.k + ((long)b * T2 + min(row, T2 - 1)) * a.ldk + h * ATT_DK + c16 * 8);
    if (row >= T2) kreg[q] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
#pragma unroll
  for (int q = 0; q < NKT / 2; ++q) {
    const int c = t + 256 * q, row = c >> 3, c16 = c & 7;
    *reinterpret_cast<uint4*>(&Ks[lds_chunk_off<true, 64>(row, c16)]) = kreg[q];
  }
  __syncthreads();
  if (!active) return;
  // ---- dq^T = K^T dS^T: rows = channels, columns = queries ----
  f32x4 C[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) C[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < NKT / 2; ++ks) {
    const uint4 gw = make_uint4(Gk[2 * ks].x, Gk[2 * ks].y, Gk[2 * ks + 1].x, Gk[2 * ks + 1].y);
    const bf16x8 gf = __builtin_bit_cast(bf16x8, gw);
    const int rlo = ks * 32 + 4 * fq + (fr >> 2), rhi = rlo + 16;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int cc = dt * 16 + 4 * (fr & 3);
      const bf16_t* q0 = &Ks[lds_chunk_off<true, 64>(rlo, cc >> 3) + (cc & 7)];
      const bf16_t* q1 = &Ks[lds_chunk_off<true, 64>(rhi, cc >> 3) + (cc & 7)];
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q0);
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q1);
      const bf16x8 kf = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      C[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, gf, C[dt], 0, 0, 0);
    }
  }
  if (qok) {
    const long ro = ((long)b * T1 + r0 + fr) * a.ldo + h * ATT_DK;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if (a.dq_bf16) {
        uint2 o;
        o.x = (unsigned)eamd_f2bf(C[dt][0]) | ((unsigned)eamd_f2bf(C[dt][1]) << 16);
        o.y = (unsigned)eamd_f2bf(C[dt][2]) | ((unsigned)eamd_f2bf(C[dt][3]) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.dq) + ro + dt * 16 + fq * 4) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.dq) + ro + dt * 16 + fq * 4) =
            make_float4(C[dt][0], C[dt][1], C[dt][2], C[dt][3]);
      }
    }
  }
}

// Key side of the backward in one launch (bf16 operands): dv = Pd^T dctx and dk = dS^T qu for one (batch, head) and a
// block of 64 keys.  Both operands of each product are read TRANSPOSED (the contraction runs over the queries, the
// leading index of both tiles), so each 64-query stage goes through LDS in the k-strided image and the fragments come
// out of ds_read_tr16_b64, exactly as the V panel does in the forward; the next stage's tiles are in registers under
// the MFMAs.  Wave w owns keys 16w .. 16w+15 and all 64 channels: out^T[channel][key], fp32 accumulation.
// reference: autograd of attention.py:63-114 (dv, dk of softmax(QK^T)V), :141-206 (with q + pos_bias_u for dk).
struct AttnKvArgs {
  const bf16_t* Pd; const bf16_t* dS; const bf16_t* dctx; const bf16_t* qu;
  bf16_t* dv; bf16_t* dk;
  long ldp, ldd, ldq, ldo;
  int B, H, T1, T2, nkb;
};

__device__ __forceinline__ void kv16_load(const bf16_t* src, long ld, int i0, int T1, int c0, int ncol, int t, uint4 (&r)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = t + 256 * q, row = c >> 3, c16 = c & 7;
    const bool ok = i0 + row < T1 && c0 + c16 * 8 < ncol;
    r[q] = *reinterpret_cast<const uint4*>(src + (long)min(i0 + row, T1 - 1) * ld + (ok ? c0 + c16 * 8 : 0));
    if (!ok) r[q] = make_uint4(0u, 0u, 0u, 0u);
  }
}
__device__ __forceinline__ void kv16_store(const uint4 (&r)[2], bf16_t* dst, int t) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = t + 256 * q, row = c >> 3, c16 = c & 7;
    *reinterpret_cast<uint4*>(&dst[lds_chunk_off<true, 64>(row, c16)]) = r[q];
  }
}
__device__ __forceinline__ bf16x8 kv16_frag(const bf16_t* tile, int ks, int col0, int fr, int fq) {
  const int rlo = ks * 32 + 4 * fq + (fr >> 2), rhi = rlo + 16, cc = col0 + 4 * (fr & 3);
  const bf16_t* q0 = &tile[lds_chunk_off<true, 64>(rlo, cc >> 3) + (cc & 7)];
  const bf16_t* q1 = &tile[lds_chunk_off<true, 64>(rhi, cc >> 3) + (cc & 7)];
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q1);
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// acc^T[channel tile ct][key tile of this wave] += X[query][channel]^T  W[query][key]
__device__ __forceinline__ void kv16_product(const bf16_t* Xc, const bf16_t* Wk, int wave, int fr, int fq, f32x4 (&acc)[4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8 wf = kv16_frag(Wk, ks, wave * 16, fr, fq);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kv16_frag(Xc, ks, ct * 16, fr, fq), wf, acc[ct], 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const AttnKvArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sW[64 * 64];     // [query][key] tile, k-strided image
  __shared__ __attribute__((aligned(16))) bf16_t sX[64 * 64];     // [query][channel] tile
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nkb) * 8 + xcd;            // z = h * B + b
  if (z >= a.B * a.H) return;                      // whole workgroup
  const int h = z / a.B, b = z % a.B;
  const int T1 = a.T1, T2 = a.T2;
  const int j0 = (jb % a.nkb) * 64;                // first key of the workgroup
  const long zo = (long)z * T1 * a.ldp;
  const bf16_t* W0 = a.Pd + zo;
  const bf16_t* W1 = a.dS + zo;
  const bf16_t* X0 = a.dctx + (long)b * T1 * a.ldd + h * ATT_DK;
  const bf16_t* X1 = a.qu + (long)b * T1 * a.ldq + h * ATT_DK;
  f32x4 acc[2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[p][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nstage = ((T1 + 63) / 64) * 2;
  uint4 rw[2], rx[2];
  kv16_load(W0, a.ldp, 0, T1, j0, (int)a.ldp, t, rw);
  kv16_load(X0, a.ldd, 0, T1, 0, 64, t, rx);
  for (int s = 0; s < nstage; ++s) {
    __syncthreads();                               // the previous stage's reads are done
    kv16_store(rw, sW, t);
    kv16_store(rx, sX, t);
    __syncthreads();
    if (s + 1 < nstage) {                          // next pair of tiles: in flight under the MFMAs
      const int i0 = (s + 1) / 2 * 64;
      if ((s + 1) & 1) { kv16_load(W1, a.ldp, i0, T1, j0, (int)a.ldp, t, rw); kv16_load(X1, a.ldq, i0, T1, 0, 64, t, rx); }
      else { kv16_load(W0, a.ldp, i0, T1, j0, (int)a.ldp, t, rw); kv16_load(X0, a.ldd, i0, T1, 0, 64, t, rx); }
    }
    if (s & 1) kv16_product(sX, sW, wave, fr, fq, acc[1]);
    else kv16_product(sX, sW, wave, fr, fq, acc[0]);
  }
  const int j = j0 + wave * 16 + fr;               // this lane's key; acc[.][ct][r] = channel 16 ct + 4 fq + r
  if (j < T2) {
    const long ro = ((long)b * T2 + j) * a.ldo + h * ATT_DK + 4 * fq;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      uint2 o;
      o.x = (unsigned)eamd_f2bf(acc[0][ct][0]) | ((unsigned)eamd_f2bf(acc[0][ct][1]) << 16);
      o.y = (unsigned)eamd_f2bf(acc[0][ct][2]) | ((unsigned)eamd_f2bf(acc[0][ct][3]) << 16);
      *reinterpret_cast<uint2*>(a.dv + ro + ct * 16) = o;
      o.x = (unsigned)eamd_f2bf(acc[1][ct][0]) | ((unsigned)eamd_f2bf(acc[1][ct][1]) << 16);
      o.y = (unsigned)eamd_f2bf(acc[1][ct][2]) | ((unsigned)eamd_f2bf(acc[1][ct][3]) << 16);
      *reinterpret_cast<uint2*>(a.dk + ro + ct * 16) = o;
    }
  }
}

template <int NKT>
int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t stream) {
  const int nz = (a.B * a.H + 7) / 8 * 8;
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<NKT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, NKT > 16 ? 144 * 1024 : 72 * 1024);
  if (attr_err != hipSuccess) return (int)attr_err;
  hipLaunchKernelGGL((attn_bwd_q_kernel<NKT>), dim3((unsigned)(a.nqb * nz)), dim3(256), (size_t)64 * (NKT * 16 + 4) * sizeof(float),
                     stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

template <bool REL, int NKT>
int launch_attn(const AttnArgs& a, size_t smem, hipStream_t stream) {
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<REL, NKT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, NKT > 16 ? 144 * 1024 : 72 * 1024);
  if (attr_err != hipSuccess) return (int)attr_err;
  const int nz = (a.B * a.H + 7) / 8 * 8;
  hipLaunchKernelGGL((attn_fwd_kernel<REL, NKT>), dim3((unsigned)(a.nqb * nz)), dim3(256), smem, stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int eamd_attn_long_fwd_bf16(const void* qu, int64_t ldq, const void* qv, int64_t ldqv, const void* k, int64_t ldk, const void* v,
                            int64_t ldv, const void* pos, int64_t ldpos, const unsigned char* mask, int64_t mb, int64_t mi,
                            void* P, int64_t ldp, void* ctx, int64_t ldc, int B, int H, int T1, int T2, float scale, void* Pd,
                            float drop_p, const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len, void* stream);   // attn_f32.hip
int eamd_attn_long_bwd_q_bf16(const void* dctx, int64_t ldd, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* P,
                              int64_t ldp, void* dS, void* dbd, void* dq, int64_t ldo, int dq_is_bf16, int B, int H, int T1, int T2,
                              float scale, float drop_p, const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len,
                              void* stream);

extern "C" int eamd_attn_fwd(const void* qu, int64_t ldq, const void* qv, int64_t ldqv, const void* k, int64_t ldk,
                             const void* v, int64_t ldv, const void* pos, int64_t ldpos, const unsigned char* mask,
                             int64_t mask_bstride, int64_t mask_qstride, void* P_bf16, int64_t ldp, void* ctx_bf16,
                             int64_t ldc, int B, int H, int T1, int T2, int dk, float scale, void* Pd_bf16, float drop_p,
                             const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len, void* stream) {
  if (!qu || !k || !v || !P_bf16 || !ctx_bf16 || B <= 0 || H <= 0 || T1 <= 0 || T2 <= 0) return EAMD_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (!Pd_bf16 || !drop_step || !al16(Pd_bf16)))) return EAMD_EINVAL;
  if ((pos == nullptr) != (qv == nullptr)) return EAMD_EINVAL;
  if (dk != ATT_DK || (pos && T1 != T2)) return EAMD_EUNSUPPORTED;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldc % 4 || ldp % 8 || ldp < T2 || (pos && (ldqv % 8 || ldpos % 8)))
    return EAMD_EUNSUPPORTED;
  if (!al16(qu) || !al16(k) || !al16(v) || !al16(P_bf16) || (reinterpret_cast<uintptr_t>(ctx_bf16) & 7) ||
      (pos && (!al16(qv) || !al16(pos))))
    return EAMD_EUNSUPPORTED;
  if ((long)B * H * ((T1 + 15) / 16) >= (1L << 28)) return EAMD_EUNSUPPORTED;
  if (T2 > ATT_MAXK)            // rows of 513 .. 2048 keys: the key-split long-row kernels (attn_f32.hip), operands widened on load
    return eamd_attn_long_fwd_bf16(qu, ldq, qv, ldqv, k, ldk, v, ldv, pos, ldpos, mask, mask_bstride, mask_qstride, P_bf16, ldp,
                                   ctx_bf16, ldc, B, H, T1, T2, scale, Pd_bf16, drop_p, drop_step, drop_salt, shift_len, stream);
  AttnArgs a;
  a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
  a.pos = (const bf16_t*)pos; a.mask = mask; a.P = (bf16_t*)P_bf16; a.ctx = (bf16_t*)ctx_bf16;
  a.ldq = ldq; a.ldqv = ldqv; a.ldk = ldk; a.ldv = ldv; a.ldpos = ldpos; a.ldc = ldc; a.ldp = ldp;
  a.mb = mask_bstride; a.mi = mask_qstride;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nqb = (T1 + 63) / 64; a.scale = scale;
  a.Pd = (bf16_t*)Pd_bf16; a.drop_p = drop_p; a.drop_step = (const unsigned long long*)drop_step; a.drop_salt = drop_salt;
  a.tshift = shift_len;
  const int nkt = T2 <= 128 ? 8 : T2 <= 256 ? 16 : 32;               // key tiles of 16 the instantiation covers
  const size_t smem = (size_t)64 * (nkt * 16 + 4) * sizeof(float);   // score matrix X (the V panel staged over it is smaller)
  hipStream_t s = (hipStream_t)stream;
  if (pos)
    return nkt == 8 ? launch_attn<true, 8>(a, smem, s) : nkt == 16 ? launch_attn<true, 16>(a, smem, s) : launch_attn<true, 32>(a, smem, s);
  return nkt == 8 ? launch_attn<false, 8>(a, smem, s) : nkt == 16 ? launch_attn<false, 16>(a, smem, s)
                                                                   : launch_attn<false, 32>(a, smem, s);
}

extern "C" int eamd_attn_bwd_q(const void* dctx, int64_t ldd, const void* k, int64_t ldk, const void* v, int64_t ldv,
                               const void* P_bf16, int64_t ldp, void* dS_bf16, void* dbd_bf16, void* dq, int64_t ldo,
                               int dq_is_bf16, int B, int H, int T1, int T2, int dk, float scale, float drop_p,
                               const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len, void* stream) {
  if (!dctx || !k || !v || !P_bf16 || !dS_bf16 || !dq || B <= 0 || H <= 0 || T1 <= 0 || T2 <= 0) return EAMD_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_step)) return EAMD_EINVAL;
  if (dk != ATT_DK || (dbd_bf16 && T1 != T2)) return EAMD_EUNSUPPORTED;
  if (ldd % 8 || ldk % 8 || ldv % 8 || ldp % 8 || ldp < T2 || ldo % 4) return EAMD_EUNSUPPORTED;
  if (!al16(dctx) || !al16(k) || !al16(v) || !al16(P_bf16) || !al16(dS_bf16) ||
      (reinterpret_cast<uintptr_t>(dq) & (dq_is_bf16 ? 7 : 15)))
    return EAMD_EUNSUPPORTED;
  if ((long)B * H * ((T1 + 15) / 16) >= (1L << 28)) return EAMD_EUNSUPPORTED;
  if (T2 > ATT_MAXK)
    return eamd_attn_long_bwd_q_bf16(dctx, ldd, k, ldk, v, ldv, P_bf16, ldp, dS_bf16, dbd_bf16, dq, ldo, dq_is_bf16, B, H, T1, T2,
                                     scale, drop_p, drop_step, drop_salt, shift_len, stream);
  AttnBwdArgs a;
  a.dctx = (const bf16_t*)dctx; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.P = (const bf16_t*)P_bf16;
  a.dS = (bf16_t*)dS_bf16; a.dbd = (bf16_t*)dbd_bf16; a.dq = dq;
  a.ldd = ldd; a.ldk = ldk; a.ldv = ldv; a.ldp = ldp; a.ldo = ldo;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nqb = (T1 + 63) / 64; a.dq_bf16 = dq_is_bf16; a.scale = scale;
  a.drop_p = drop_p; a.drop_step = (const unsigned long long*)drop_step; a.drop_salt = drop_salt;
  a.tshift = shift_len;
  return T2 <= 128 ? launch_attn_bwd<8>(a, (hipStream_t)stream) : T2 <= 256 ? launch_attn_bwd<16>(a, (hipStream_t)stream)
                                                                             : launch_attn_bwd<32>(a, (hipStream_t)stream);
}

extern "C" int eamd_attn_bwd_kv(const void* Pd_bf16, const void* dS_bf16, int64_t ldp, const void* dctx, int64_t ldd,
                                const void* qu, int64_t ldq, void* dv, void* dk_out, int64_t ldo, int B, int H, int T1,
                                int T2, int dk, void* stream) {
  if (!Pd_bf16 || !dS_bf16 || !dctx || !qu || !dv || !dk_out || B <= 0 || H <= 0 || T1 <= 0 || T2 <= 0) return EAMD_EINVAL;
  if (dk != ATT_DK) return EAMD_EUNSUPPORTED;
  if (ldp % 8 || ldp < T2 || ldd % 8 || ldq % 8 || ldo % 4) return EAMD_EUNSUPPORTED;
  if (!al16(Pd_bf16) || !al16(dS_bf16) || !al16(dctx) || !al16(qu) || (reinterpret_cast<uintptr_t>(dv) & 7) ||
      (reinterpret_cast<uintptr_t>(dk_out) & 7))
    return EAMD_EUNSUPPORTED;
  AttnKvArgs a;
  a.Pd = (const bf16_t*)Pd_bf16; a.dS = (const bf16_t*)dS_bf16; a.dctx = (const bf16_t*)dctx; a.qu = (const bf16_t*)qu;
  a.dv = (bf16_t*)dv; a.dk = (bf16_t*)dk_out;
  a.ldp = ldp; a.ldd = ldd; a.ldq = ldq; a.ldo = ldo;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nkb = (T2 + 63) / 64;
  if ((long)B * H * a.nkb >= (1L << 28)) return EAMD_EUNSUPPORTED;
  const int nz = (B * H + 7) / 8 * 8;
  hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3((unsigned)(a.nkb * nz)), dim3(256), 0, (hipStream_t)stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
