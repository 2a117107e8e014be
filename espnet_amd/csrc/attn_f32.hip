// Fused attention for fp32 activations (the reference's precision), d_k = 64, up to 256 keys, on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32): scores (+ relative-position term with the legacy rel_shift), mask, softmax and the
// context product in ONE kernel, and the query side of the backward in a second one; fp32 twins of attn_fused.hip.
// reference: transformer/attention.py:63-114 (forward_attention / MultiHeadedAttention.forward),
//            attention.py:141-206 (RelPositionMultiHeadedAttention: rel_shift, (ac + bd) / sqrt(d_k)).
//
// One workgroup = one (batch, head) pair x 64 queries, four waves.  Two work splits are used:
//   * SCORE phases (ac = q k^T, bd = (q+v) p^T; dP = dctx v^T in the backward) - wave w takes a quarter of the KEYS
//     (positions) and ALL 64 queries.  A 16x16x4 MFMA takes ONE float per lane per operand and the order of the
//     64-channel contraction is free, so lane (fr = lane & 15, fq = lane >> 4) takes the four 16-byte chunks
//     {4 j + fq} of row fr straight from global memory into MFMA operands: no operand staging.  Such row-strided
//     fragment loads are the slow part of the kernel (16 cache lines per instruction), which is why the split is by
//     keys: every K / position row is fetched by exactly one wave of the workgroup, the 64 query rows once per wave
//     (measured: 63 -> see DESIGN for the variant in which every wave fetched all keys for its own 16 queries).
//     The tiles S^T[key][query] land in an LDS score matrix X[query][key] (16-byte stores: an accumulator holds four
//     adjacent keys of one query);
//   * the legacy rel_shift is a flat re-indexing of bd padded with a zero column: shifted[i][j] = bd[i][T-1-i+j] for
//     j <= i, 0 for j = i+1, bd[i+1][j-i-2] above.  It is one-to-one, so the bd tiles are STORED into X at their
//     shifted place first (plain 4-byte LDS stores, one owner per element; as ds_add_f32 atomics on top of the ac
//     tiles the same scatter cost 29 us of a 76 us kernel) and the ac tiles are then added by read-modify-write of
//     16-byte rows behind a barrier.  bd row r0 + 64 (first query of the next workgroup) feeds query r0 + 63:
//     64-term dot products on the VALU, one position per thread;
//   * SOFTMAX and the products with P - wave w owns queries 16 w .. 16 w + 15 and all keys, so a softmax row never
//     leaves its wave: it reads its X rows back in the accumulator layout (lane = query fr, keys 16 t + 4 fq + {0..3}),
//     and each probability IS the operand of the context product C^T = V^T P^T for the contraction step
//     "keys 16 t + 4 fq + r, fq = 0..3".  V comes from a panel staged over X once the scores are in registers
//     ([key][68] floats: the four fq groups of a ds_read_b32 fall into disjoint banks; its global loads are issued
//     before the softmax).  C^T leaves four adjacent channels per lane (16-byte stores).
#include "common.h"

namespace {

constexpr int ATT_DK = 64;
constexpr int ATT_MAXK = 512;                 // keys per row: 8 / 16 key tiles of 16 at two workgroups per CU, 32 at one (LDS)
constexpr int PLD = 68;                       // row stride (floats) of the V / K panel in LDS

struct AttnF32Args {
  const float* qu; const float* qv; const float* k; const float* v; const float* pos;
  const unsigned char* mask;
  float* P; float* ctx;
  long ldq, ldqv, ldk, ldv, ldpos, ldc, ldp, mb, mi;
  int B, H, T1, T2, nqb;
  float scale;
  // attention dropout (attention.py:91): Pd = dropout(P) feeds the context; mask index = element index in P
  float* Pd; float drop_p; const unsigned long long* drop_step; unsigned long long drop_salt;
  const int* tshift;     // device scalar: length the legacy rel_shift works on (NULL: T2) - see eamd_attn_fwd
};

__device__ __forceinline__ float xmax16_32(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xsum16_32(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// this lane's share of a 64-channel row: chunks 4 j + fq
__device__ __forceinline__ void load_frag(const float* row, int fq, float4 (&f)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = *reinterpret_cast<const float4*>(row + 16 * j + 4 * fq);
}
// c += A B^T over the 64 channels, A rows / B rows held as load_frag leaves them
__device__ __forceinline__ f32x4 dot_tile(const float4 (&x)[4], const float4 (&y)[4], f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].x, y[j].x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].y, y[j].y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].z, y[j].z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j].w, y[j].w, c, 0, 0, 0);
  }
  return c;
}

// panel [16 NKT rows][64 channels] of one (batch, head): global -> registers (rows past `rows` zeroed) ...
template <int NKT>
__device__ __forceinline__ void panel_load(const float* src, long ld, int rows, int t, float4 (&r)[NKT]) {
#pragma unroll
  for (int q = 0; q < NKT; ++q) {
    const int c = t + 256 * q, row = c >> 4, c4 = c & 15;
    r[q] = *reinterpret_cast<const float4*>(src + (long)min(row, rows - 1) * ld + c4 * 4);
    if (row >= rows) r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// ... -> LDS
template <int NKT>
__device__ __forceinline__ void panel_store(const float4 (&r)[NKT], float* dst, int t) {
#pragma unroll
  for (int q = 0; q < NKT; ++q) {
    const int c = t + 256 * q, row = c >> 4, c4 = c & 15;
    *reinterpret_cast<float4*>(&dst[row * PLD + c4 * 4]) = r[q];
  }
}

// out^T[channel][query] += panel^T[channel][key] G^T[key][query] with G in the score-accumulator layout
template <int NKT>
__device__ __forceinline__ void panel_product(const float* panel, const f32x4 (&G)[NKT], int fr, int fq, f32x4 (&C)[4]) {
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row = &panel[(kt * 16 + 4 * fq + r) * PLD + fr];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) C[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[dt * 16], G[kt][r], C[dt], 0, 0, 0);
    }
}

// X[query][key] = (rows of `keys`) . (rows of `qs`)^T for this wave's quarter of the key tiles and all 64 queries.
// keys: first key row of the (batch, head), T2 rows; qs: first query row of the workgroup, rows clamped to nq - 1.
template <int NKT, bool ACC>
__device__ __forceinline__ void score_tiles(const float* keys, long ldk, int T2, const float* qs, long ldq, int nq,
                                            float* X, int XLD, int wave, int fr, int fq) {
  constexpr int TPW = NKT / 4;
  float4 qf[4][4];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) load_frag(qs + (long)min(qt * 16 + fr, nq - 1) * ldq, fq, qf[qt]);
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int kt = wave * TPW + u;
    float4 kf[4];
    load_frag(keys + (long)min(kt * 16 + fr, T2 - 1) * ldk, fq, kf);
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
      float4* xp = reinterpret_cast<float4*>(&X[(qt * 16 + fr) * XLD + kt * 16 + 4 * fq]);
      if (ACC) {                      // on top of what is there (one owner per element: plain read-modify-write)
        const float4 o = *xp;
        c = (f32x4){o.x, o.y, o.z, o.w};
      }
      c = dot_tile(kf, qf[qt], c);    // c[r]: key 16 kt + 4 fq + r, query 16 qt + fr
      *xp = make_float4(c[0], c[1], c[2], c[3]);
    }
  }
}

// NKT = key tiles of 16 the instantiation covers (T2 <= 16 NKT).  Loops over tiles are fully unrolled and free of
// branches: rows past T2 / T1 are clamped re-reads whose scores are masked / never stored.
template <bool REL, int NKT>
__global__ __launch_bounds__(256, NKT > 16 ? 1 : 2) void attn_f32_fwd_kernel(const AttnF32Args a) {
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
    // ---- bd, stored at its rel-shifted place: bd[i][m] -> (i, m - (T-1-i)) if m >= T-1-i, else (i - 1, m + i + 1).
    // The map is one-to-one; the only elements of a row it never reaches are (i, i + 1) - zeroed here - and, for the
    // last query of the workgroup, the part fed by bd row r0w + 64 (the dot products below) ----
    float4 qf[4][4];
    const float* qs = a.qv + ((long)b * T1 + r0w) * a.ldqv + h * ATT_DK;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) load_frag(qs + (long)min(qt * 16 + fr, nq - 1) * a.ldqv, fq, qf[qt]);
    if (t < 64 && r0w + t + 1 < T2) X[t * XLD + r0w + t + 1] = 0.f;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int mt = wave * TPW + u;
      float4 pf[4];
      load_frag(a.pos + (long)min(mt * 16 + fr, T2 - 1) * a.ldpos + h * ATT_DK, fq, pf);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const f32x4 c = dot_tile(pf, qf[qt], (f32x4){0.f, 0.f, 0.f, 0.f});   // c[r] = bd[query 16 qt + fr][m = 16 mt + 4 fq + r]
        const int ql = qt * 16 + fr, i = r0w + ql;
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
      const float4* qr = reinterpret_cast<const float4*>(a.qv + ((long)b * T1 + i64) * a.ldqv + h * ATT_DK);
      for (int m = t; m <= Ts - 2 - i64; m += 256) {
        const float4* pr = reinterpret_cast<const float4*>(a.pos + (long)m * a.ldpos + h * ATT_DK);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float4 x = qr[j], y = pr[j];
          s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        X[63 * XLD + m + i64 + 1] = s;
      }
    }
    __syncthreads();
  }
  // ---- ac: X[q][j] (+)= qu_q . k_j ----
  score_tiles<NKT, REL>(a.k + (long)b * T2 * a.ldk + h * ATT_DK, a.ldk, T2, a.qu + ((long)b * T1 + r0w) * a.ldq + h * ATT_DK,
                        a.ldq, nq, X, XLD, wave, fr, fq);
  // V panel: global loads now, LDS stores once the scores have left X
  float4 vreg[NKT];
  panel_load<NKT>(a.v + (long)b * T2 * a.ldv + h * ATT_DK, a.ldv, live ? T2 : 1, t, vreg);
  __syncthreads();                                  // X is complete
  // ---- scale, mask, softmax over the keys of query (r0 + fr) ----
  f32x4 S[NKT];
  if (active) {
    // mask bytes of this lane's keys: unconditional clamped loads, all in flight together; keys past T2 are masked
    // by index below
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
    mx = xmax16_32(mx);
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
    sum = xsum16_32(sum);
    const float inv = dead ? 0.f : 1.f / sum;
    const bool qok = r0 + fr < T1;
    const long pro = ((long)zz * T1 + qi) * a.ldp;
    float* prow = a.P + pro;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      S[kt] *= inv;
      const int j0 = kt * 16 + fq * 4;
      if (qok && j0 < a.ldp)                         // pad columns receive zeros
        *reinterpret_cast<float4*>(prow + j0) = make_float4(S[kt][0], S[kt][1], S[kt][2], S[kt][3]);
    }
    if (a.drop_p > 0.f) {                            // wave-uniform: the dropped probabilities go on to the context product
      const unsigned seed = eamd_drop_seed(a.drop_step, a.drop_salt), thr = eamd_drop_thr16(a.drop_p);
      const float dinv = eamd_drop_inv(thr);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const int j0 = kt * 16 + fq * 4;
        bool kp[4];
        eamd_drop_keep4(seed, (unsigned long long)(pro + min(j0, (int)a.ldp - 4)), thr, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[kt][r] = kp[r] ? S[kt][r] * dinv : 0.f;
        if (qok && j0 < a.ldp)
          *reinterpret_cast<float4*>(a.Pd + pro + j0) = make_float4(S[kt][0], S[kt][1], S[kt][2], S[kt][3]);
      }
    }
  } else {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) S[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();                                  // every wave has its scores in registers
  float* Vs = reinterpret_cast<float*>(smem_raw);
  panel_store<NKT>(vreg, Vs, t);
  __syncthreads();
  if (!active) return;
  // ---- context^T = V^T P^T: rows = channels, columns = queries ----
  f32x4 C[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) C[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  panel_product<NKT>(Vs, S, fr, fq, C);
  if (r0 + fr < T1) {
    float* crow = a.ctx + ((long)b * T1 + r0 + fr) * a.ldc + h * ATT_DK;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<float4*>(crow + dt * 16 + fq * 4) = make_float4(C[dt][0], C[dt][1], C[dt][2], C[dt][3]);
  }
}

// ---- backward, query side: dP = dctx V^T, dS = scale * P (dP - rowsum(P dP)), dq = dS K in one launch ----
// Same two work splits as the forward: dP^T tiles by key quarter into the LDS matrix X, then each wave takes its 16
// queries: P re-read with the 16-byte accesses the forward stored it with, the row sum inside the wave, dS stored for
// the key-side GEMMs (dK = dS^T q, and with relative positions dqv / dpos from the inverse rel_shift scatter dbd,
// written here element by element exactly as eamd_softmax_bwd does) and dq^T = K^T dS^T from a K panel staged over X -
// the context product of the forward with K in place of V.  reference: autograd of attention.py:63-114, :141-206.
struct AttnF32BwdArgs {
  const float* dctx; const float* k; const float* v; const float* P;
  float* dS; float* dbd; float* dq;
  long ldd, ldk, ldv, ldp, ldo;
  int B, H, T1, T2, nqb;
  float scale;
  float drop_p; const unsigned long long* drop_step; unsigned long long drop_salt;   // the forward's attention dropout
  const int* tshift;
};

template <int NKT>
__global__ __launch_bounds__(256, NKT > 16 ? 1 : 2) void attn_f32_bwd_q_kernel(const AttnF32BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int XLD = NKT * 16 + 4;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nqb) * 8 + xcd;
  const bool live = z < a.B * a.H;
  const int zz = live ? z : 0;
  const int h = zz / a.B, b = zz % a.B;
  const int T1 = a.T1, T2 = a.T2;
  const int Ts = (a.dbd && a.tshift) ? min(max(a.tshift[0], 1), T2) : T2;      // rel_shift length (see the forward kernel)
  const int r0w = min((jb % a.nqb) * 64, T1 - 1);
  const int nq = T1 - r0w;
  const int r0 = r0w + wave * 16;
  const bool active = live && r0 < T1;
  const int qi = min(r0 + fr, T1 - 1);
  const bool qok = r0 + fr < T1;

  float* X = reinterpret_cast<float*>(smem_raw);
  // ---- dP: X[q][j] = dctx_q . v_j ----
  score_tiles<NKT, false>(a.v + (long)b * T2 * a.ldv + h * ATT_DK, a.ldv, T2,
                          a.dctx + ((long)b * T1 + r0w) * a.ldd + h * ATT_DK, a.ldd, nq, X, XLD, wave, fr, fq);
  // this lane's probabilities: in flight across the barrier
  const float* prow = a.P + ((long)zz * T1 + qi) * a.ldp;
  float4 Pr[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int j0 = kt * 16 + fq * 4;
    Pr[kt] = *reinterpret_cast<const float4*>(prow + min(j0, (int)a.ldp - 4));
    if (j0 >= (int)a.ldp) Pr[kt] = make_float4(0.f, 0.f, 0.f, 0.f);   // columns past ldp do not exist
  }
  __syncthreads();
  f32x4 S[NKT];                                     // dP^T, then dS^T
  if (active) {
    const float* xrow = &X[(wave * 16 + fr) * XLD + 4 * fq];
    const bool drop = a.drop_p > 0.f;                // wave-uniform
    const unsigned seed = drop ? eamd_drop_seed(a.drop_step, a.drop_salt) : 0u, thr = eamd_drop_thr16(a.drop_p);
    const float dinv = eamd_drop_inv(thr);
    const long pro = ((long)zz * T1 + qi) * a.ldp;
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int j0 = kt * 16 + fq * 4;
      const float4 xv = *reinterpret_cast<const float4*>(xrow + kt * 16);
      S[kt] = (f32x4){xv.x, xv.y, xv.z, xv.w};
      if (drop) {                                    // gradient of the dropped probabilities -> gradient of P
        bool kp[4];
        eamd_drop_keep4(seed, (unsigned long long)(pro + min(j0, (int)a.ldp - 4)), thr, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[kt][r] = kp[r] ? S[kt][r] * dinv : 0.f;
      }
      const float pr[4] = {Pr[kt].x, Pr[kt].y, Pr[kt].z, Pr[kt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) if (j0 + r < T2) s += pr[r] * S[kt][r];
    }
    s = xsum16_32(s);
    const long zo = (long)zz * T1 * a.ldp;
    float* srow = a.dS + zo + (long)qi * a.ldp;
    const int i = r0 + fr;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int j0 = kt * 16 + fq * 4;
      const float pr[4] = {Pr[kt].x, Pr[kt].y, Pr[kt].z, Pr[kt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + r;
        const float g = j < T2 ? pr[r] * (S[kt][r] - s) * a.scale : 0.f;
        S[kt][r] = g;
        if (a.dbd && qok) {
          // inverse rel_shift for T1 == T2 = T (Ts when the shift is shorter than the padded matrix: everything outside the
          // Ts x Ts square is zero): padded index T + i T + j lands in row i (j <= i) or row i + 1
          if (j < Ts && i < Ts) {
            const int R = j <= i ? i : i + 1, c = j <= i ? Ts + j - i : j - i - 1;
            if (c != 0) a.dbd[zo + (long)R * a.ldp + (c - 1)] = g;
          } else if (j < (int)a.ldp) {
            a.dbd[zo + (long)i * a.ldp + j] = 0.f;    // pad columns of this row
          }
        }
      }
      if (qok && j0 < (int)a.ldp) *reinterpret_cast<float4*>(srow + j0) = make_float4(S[kt][0], S[kt][1], S[kt][2], S[kt][3]);
    }
    if (a.dbd && r0 == 0)                            // the head of row 0 the scatter never reaches
      for (int f = 1 + lane; f < Ts; f += 64) a.dbd[zo + (f - 1)] = 0.f;
  } else {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) S[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float4 kreg[NKT];                                 // K panel: loads in flight across the barrier (the probabilities are dead)
  panel_load<NKT>(a.k + (long)b * T2 * a.ldk + h * ATT_DK, a.ldk, live ? T2 : 1, t, kreg);
  __syncthreads();                                  // every wave has its dP rows in registers
  float* Ks = reinterpret_cast<float*>(smem_raw);
  panel_store<NKT>(kreg, Ks, t);
  __syncthreads();
  if (!active) return;
  // ---- dq^T = K^T dS^T: rows = channels, columns = queries ----
  f32x4 C[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) C[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  panel_product<NKT>(Ks, S, fr, fq, C);
  if (qok) {
    float* orow = a.dq + ((long)b * T1 + r0 + fr) * a.ldo + h * ATT_DK;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<float4*>(orow + dt * 16 + fq * 4) = make_float4(C[dt][0], C[dt][1], C[dt][2], C[dt][3]);
  }
}

// =====================================================================================================================
// Rows of more than 512 keys (up to ATT_LONG_MAXK = 4096).  The kernels above keep a query block's whole score row in registers
// (S[NKT]) and a V / K panel of all keys in LDS; that stops at 512 keys.  The long-row forms keep the structure - the
// score matrix X[query][key] of a query block lives in LDS for the whole kernel, so the legacy rel_shift stays the same
// one-to-one scatter of the bd tiles - but a workgroup takes 16 queries (X = 16 x (T2 + 4) floats: 131 KB at 2048 keys; 8 queries
// beyond that, up to 4096 keys: the 16 x 16 tiles then carry query LQ - 1 eight more times - same values stored to the same
// addresses by lanes of one instruction - which halves the useful MFMA work but keeps such rows off the GEMM + softmax path)
// and ALL FOUR phases are split over the keys: wave w owns key tiles w, w + 4, w + 8, ... in the score products, in the
// softmax (row maxima / sums meet through 128 floats of LDS) and in the product with V (K), whose 16-key tiles each wave
// stages in a private 4 KB LDS buffer; the four partial context (dq) tiles are summed through LDS at the end.  The
// softmax walks X three times (max, exp + sum, normalise) instead of holding the row in registers.
// Per workgroup K, V and the positions are read once (768 KB at 1024 keys) for 16 queries: four times the L2 traffic of
// the 64-query kernels per query, which is what the LDS budget allows.
constexpr int ATT_LONG_MAXK = 4096;                         // (LQ, the queries per workgroup, is a template parameter: 16, or 8 beyond 2048 keys)

// element access for the two storage types of the long-row kernels: float (fp32 mode) and bf16 bits (bf16-operand mode:
// same kernels, operands widened on load, fp32 MFMA; P / dS / dbd / ctx rounded to bf16 on store)
typedef unsigned short bfbits;
__device__ __forceinline__ float bf2f_(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bfbits* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf2f_(u.x & 0xffffu), bf2f_(u.x >> 16), bf2f_(u.y & 0xffffu), bf2f_(u.y >> 16));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bfbits* p, float4 v) {
  uint2 u;
  u.x = (unsigned)eamd_f2bf(v.x) | ((unsigned)eamd_f2bf(v.y) << 16);
  u.y = (unsigned)eamd_f2bf(v.z) | ((unsigned)eamd_f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bfbits* p, float v) { *p = eamd_f2bf(v); }
__device__ __forceinline__ float rnd(const float*, float v) { return v; }                      // value as the storage type keeps it
__device__ __forceinline__ float rnd(const bfbits*, float v) { return bf2f_(eamd_f2bf(v)); }
template <typename T>
__device__ __forceinline__ void load_frag_t(const T* row, int fq, float4 (&f)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = ld4(row + 16 * j + 4 * fq);
}

// stage one 16-key tile (rows key0 .. key0 + 15 of `src`, clamped to T2 - 1) into this wave's [16][PLD] LDS buffer
template <typename T>
__device__ __forceinline__ void tile_load(const T* src, long ld, int key0, int T2, int lane, float4 (&r)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = lane + 64 * q, row = c >> 4, c4 = c & 15;
    r[q] = ld4(src + (long)min(key0 + row, T2 - 1) * ld + c4 * 4);
    if (key0 + row >= T2) r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void tile_store(const float4 (&r)[4], float* dst, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = lane + 64 * q, row = c >> 4, c4 = c & 15;
    *reinterpret_cast<float4*>(&dst[row * PLD + c4 * 4]) = r[q];
  }
}
// C^T[channel][query] += tile^T[channel][key] G^T[key][query] for one 16-key tile, G in the accumulator layout
__device__ __forceinline__ void tile_product(const float* tile, const f32x4& G, int fr, int fq, f32x4 (&C)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float* row = &tile[(4 * fq + r) * PLD + fr];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) C[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[dt * 16], G[r], C[dt], 0, 0, 0);
  }
}
// sum of the NW waves' C^T tiles -> out[query][64 channels] (16-byte stores); `scr` = NW x [64][17] floats
template <int NW, typename T>
__device__ __forceinline__ void reduce_store_ct(const f32x4 (&C)[4], float* scr, int wave, int fr, int fq, int t, T* out,
                                                long ld, int nq) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(wave * 64 + dt * 16 + fq * 4 + r) * 17 + fr] = C[dt][r];
  __syncthreads();
  const int q = t >> 4, c4 = (t & 15) * 4;
  if (q < nq) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += scr[(w * 64 + c4 + e) * 17 + q];
      o[e] = v;
    }
    st4(out + (long)q * ld + c4, make_float4(o[0], o[1], o[2], o[3]));
  }
}

template <typename T, bool REL, int LQ, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_long_kernel(const AttnF32Args a, const int nkt) {
  constexpr int QT = LQ >= 16 ? LQ / 16 : 1;
  constexpr int NT = 64 * NW;                          // NW waves split the key tiles (4, or 8 where only one workgroup fits a CU)           // 16-query tiles of the workgroup: every key / value / position fragment a wave
  const T* const a_qu = reinterpret_cast<const T*>(a.qu); const T* const a_qv = reinterpret_cast<const T*>(a.qv);      // loads serves all of them
  const T* const a_k = reinterpret_cast<const T*>(a.k); const T* const a_v = reinterpret_cast<const T*>(a.v);
  const T* const a_pos = reinterpret_cast<const T*>(a.pos);
  T* const a_P = reinterpret_cast<T*>(a.P); T* const a_Pd = reinterpret_cast<T*>(a.Pd); T* const a_ctx = reinterpret_cast<T*>(a.ctx);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int XLD = nkt * 16 + 4;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nqb) * 8 + xcd;
  const bool live = z < a.B * a.H;
  const int zz = live ? z : 0;
  const int h = zz / a.B, b = zz % a.B;
  const int T1 = a.T1, T2 = a.T2;
  const int r0w = min((jb % a.nqb) * LQ, T1 - 1);
  const int nq = min(LQ, T1 - r0w);
  // per query tile: the row of X / red this lane works on (LQ = 8: lanes of the tile's upper half repeat query LQ - 1 - same values,
  // same addresses, one instruction), the query row it loads (clamped), the query whose mask / probability row it serves
  int rowx[QT], qrow[QT], qi[QT];
  bool qok[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    rowx[qt] = min(fr + 16 * qt, LQ - 1);
    qrow[qt] = r0w + min(fr + 16 * qt, nq - 1);
    qi[qt] = min(r0w + rowx[qt], T1 - 1);
    qok[qt] = live && fr + 16 * qt < nq;
  }
  float* X = reinterpret_cast<float*>(smem_raw);
  float* Tw = X + LQ * XLD + wave * 16 * PLD;           // this wave's tile buffer
  float* red = X + LQ * XLD + NW * 16 * PLD;            // [2][NW][LQ]
  const int Ts = (REL && a.tshift) ? min(max(a.tshift[0], 1), T2) : T2;
  if (REL) {
    if (Ts < T2) {
      for (int e = t; e < LQ * XLD; e += NT) X[e] = 0.f;
      __syncthreads();
    }
    float4 qf[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) load_frag_t(a_qv + ((long)b * T1 + qrow[qt]) * a.ldqv + h * ATT_DK, fq, qf[qt]);
    if (t < LQ && r0w + t + 1 < T2) X[t * XLD + r0w + t + 1] = 0.f;
    // (the next tile's fragment is requested before this tile's products: with one or two workgroups of 4 - 8 waves per CU nothing
    // else hides the round trip of these row-strided loads)
    float4 pf[4], pfn[4];
    if (wave < nkt) load_frag_t(a_pos + (long)min(wave * 16 + fr, T2 - 1) * a.ldpos + h * ATT_DK, fq, pf);
    for (int mt = wave; mt < nkt; mt += NW) {
      if (mt + NW < nkt) load_frag_t(a_pos + (long)min((mt + NW) * 16 + fr, T2 - 1) * a.ldpos + h * ATT_DK, fq, pfn);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4 c = dot_tile(pf, qf[qt], (f32x4){0.f, 0.f, 0.f, 0.f});     // c[r] = bd[query 16 qt + fr][m = 16 mt + 4 fq + r]
        const int lr = fr + 16 * qt;                     // the query inside the block
        const int i = r0w + lr, lim = Ts - 1 - i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mt * 16 + 4 * fq + r;
          const int row = m >= lim ? lr : lr - 1, j = m >= lim ? m - lim : m + i + 1;
          if (m < Ts && i < Ts && lr < nq && row >= 0) X[row * XLD + j] = c[r];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) pf[j] = pfn[j];
    }
    const int i16 = r0w + LQ;                         // first query of the next workgroup: its low positions feed query r0w + LQ - 1
    if (live && i16 < Ts) {
      const T* qr = a_qv + ((long)b * T1 + i16) * a.ldqv + h * ATT_DK;
      for (int m = t; m <= Ts - 2 - i16; m += NT) {
        const T* pr = a_pos + (long)m * a.ldpos + h * ATT_DK;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float4 x = ld4(qr + 4 * j), y = ld4(pr + 4 * j);
          s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        X[(LQ - 1) * XLD + m + i16 + 1] = s;
      }
    }
    __syncthreads();
  }
  // ---- ac on top, then scale + mask, in place; row maxima ----
  float mx[QT];
  {
    const unsigned char* mr[QT];
    float4 qf[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      mr[qt] = a.mask ? a.mask + (long)b * a.mb + (long)qi[qt] * a.mi : nullptr;
      mx[qt] = -INFINITY;
      load_frag_t(a_qu + ((long)b * T1 + qrow[qt]) * a.ldq + h * ATT_DK, fq, qf[qt]);
    }
    const T* keys = a_k + (long)b * T2 * a.ldk + h * ATT_DK;
    float4 kf[4], kfn[4];
    if (wave < nkt) load_frag_t(keys + (long)min(wave * 16 + fr, T2 - 1) * a.ldk, fq, kf);
    for (int kt = wave; kt < nkt; kt += NW) {
      if (kt + NW < nkt) load_frag_t(keys + (long)min((kt + NW) * 16 + fr, T2 - 1) * a.ldk, fq, kfn);
      const int j0 = kt * 16 + fq * 4;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float4* xp = reinterpret_cast<float4*>(&X[rowx[qt] * XLD + kt * 16 + 4 * fq]);
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (REL) { const float4 o = *xp; c = (f32x4){o.x, o.y, o.z, o.w}; }
        c = dot_tile(kf, qf[qt], c);                    // c[r]: key 16 kt + 4 fq + r, query 16 qt + fr
        unsigned mk = 0x01010101u;
        if (mr[qt]) mk = mr[qt][min(j0, T2 - 1)] | (mr[qt][min(j0 + 1, T2 - 1)] << 8) | (mr[qt][min(j0 + 2, T2 - 1)] << 16) |
                         (mr[qt][min(j0 + 3, T2 - 1)] << 24);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = c[r] * a.scale;
          if (j0 + r >= T2 || ((mk >> (8 * r)) & 0xffu) == 0u) x = -INFINITY;
          c[r] = x;
          mx[qt] = fmaxf(mx[qt], x);
        }
        *xp = make_float4(c[0], c[1], c[2], c[3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = kfn[j];
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mx[qt] = xmax16_32(mx[qt]);
    if (fq == 0) red[wave * LQ + rowx[qt]] = mx[qt];
  }
  __syncthreads();
  bool dead[QT];
  float sum[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int rx = rowx[qt];
    float m_ = red[rx];
#pragma unroll
    for (int w_ = 1; w_ < NW; ++w_) m_ = fmaxf(m_, red[w_ * LQ + rx]);
    mx[qt] = m_;
    dead[qt] = mx[qt] == -INFINITY;                    // every key masked: zeros (softmax of min, then masked_fill(0))
    sum[qt] = 0.f;
  }
  for (int kt = wave; kt < nkt; kt += NW) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float4* xp = reinterpret_cast<float4*>(&X[rowx[qt] * XLD + kt * 16 + 4 * fq]);
      float4 v = *xp;
      v.x = dead[qt] ? 0.f : __expf(v.x - mx[qt]); v.y = dead[qt] ? 0.f : __expf(v.y - mx[qt]);
      v.z = dead[qt] ? 0.f : __expf(v.z - mx[qt]); v.w = dead[qt] ? 0.f : __expf(v.w - mx[qt]);
      sum[qt] += (v.x + v.y) + (v.z + v.w);
      *xp = v;
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    sum[qt] = xsum16_32(sum[qt]);
    if (fq == 0) red[NW * LQ + wave * LQ + rowx[qt]] = sum[qt];
  }
  __syncthreads();
  float inv[QT];
  long pro[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int rx = rowx[qt];
    float sm = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < NW; w_ += 2) sm += red[(NW + w_) * LQ + rx] + red[(NW + w_ + 1) * LQ + rx];      // (NW = 4: the pairs' sum as before)
    inv[qt] = dead[qt] ? 0.f : 1.f / sm;
    pro[qt] = ((long)zz * T1 + qi[qt]) * a.ldp;
  }
  // ---- probabilities out, context^T = V^T Pd^T over this wave's key tiles ----
  const bool drop = a.drop_p > 0.f;
  const unsigned seed = drop ? eamd_drop_seed(a.drop_step, a.drop_salt) : 0u, thr = eamd_drop_thr16(a.drop_p);
  const float dinv = eamd_drop_inv(thr);
  const T* vs = a_v + (long)b * T2 * a.ldv + h * ATT_DK;
  f32x4 C[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) C[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 vreg[4];
  if (wave < nkt) tile_load(vs, a.ldv, wave * 16, live ? T2 : 1, lane, vreg);
  for (int kt = wave; kt < nkt; kt += NW) {
    tile_store(vreg, Tw, lane);
    if (kt + NW < nkt) tile_load(vs, a.ldv, (kt + NW) * 16, live ? T2 : 1, lane, vreg);
    const int j0 = kt * 16 + fq * 4;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float4 e = *reinterpret_cast<const float4*>(&X[rowx[qt] * XLD + kt * 16 + 4 * fq]);
      f32x4 Pv = (f32x4){e.x * inv[qt], e.y * inv[qt], e.z * inv[qt], e.w * inv[qt]};
      if (qok[qt] && j0 < a.ldp) st4(a_P + pro[qt] + j0, make_float4(Pv[0], Pv[1], Pv[2], Pv[3]));
      if (drop) {
        bool kp[4];
        eamd_drop_keep4(seed, (unsigned long long)(pro[qt] + min(j0, (int)a.ldp - 4)), thr, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) Pv[r] = kp[r] ? Pv[r] * dinv : 0.f;
        if (qok[qt] && j0 < a.ldp) st4(a_Pd + pro[qt] + j0, make_float4(Pv[0], Pv[1], Pv[2], Pv[3]));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) Pv[r] = rnd(a_P, Pv[r]);      // the context is built from the probabilities backward will read
      tile_product(Tw, Pv, fr, fq, C[qt]);
    }
  }
  __syncthreads();                                    // X is dead: its head becomes the reduction scratch
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (qt > 0) __syncthreads();
    reduce_store_ct<NW>(C[qt], X, wave, fr, fq, t, a_ctx + ((long)b * T1 + r0w + 16 * qt) * a.ldc + h * ATT_DK, a.ldc,
                    live ? max(0, min(16, nq - 16 * qt)) : 0);
  }
}

template <typename T, int LQ, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_q_long_kernel(const AttnF32BwdArgs a, const int nkt, const int dq_bf16) {
  constexpr int QT = LQ >= 16 ? LQ / 16 : 1;
  constexpr int NT = 64 * NW;                          // NW waves split the key tiles (4, or 8 where only one workgroup fits a CU)
  const T* const a_dctx = reinterpret_cast<const T*>(a.dctx); const T* const a_k = reinterpret_cast<const T*>(a.k);
  const T* const a_v = reinterpret_cast<const T*>(a.v); const T* const a_P = reinterpret_cast<const T*>(a.P);
  T* const a_dS = reinterpret_cast<T*>(a.dS); T* const a_dbd = reinterpret_cast<T*>(a.dbd);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int XLD = nkt * 16 + 4;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nqb) * 8 + xcd;
  const bool live = z < a.B * a.H;
  const int zz = live ? z : 0;
  const int h = zz / a.B, b = zz % a.B;
  const int T1 = a.T1, T2 = a.T2;
  const int Ts = (a.dbd && a.tshift) ? min(max(a.tshift[0], 1), T2) : T2;
  const int r0w = min((jb % a.nqb) * LQ, T1 - 1);
  const int nq = min(LQ, T1 - r0w);
  const long zo = (long)zz * T1 * a.ldp;
  int rowx[QT], qrow[QT];
  long pro[QT];
  bool qok[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    rowx[qt] = min(fr + 16 * qt, LQ - 1);
    qrow[qt] = r0w + min(fr + 16 * qt, nq - 1);
    pro[qt] = zo + (long)min(r0w + rowx[qt], T1 - 1) * a.ldp;
    qok[qt] = live && fr + 16 * qt < nq;
  }
  float* X = reinterpret_cast<float*>(smem_raw);
  float* Tw = X + LQ * XLD + wave * 16 * PLD;
  float* red = X + LQ * XLD + NW * 16 * PLD;
  const bool drop = a.drop_p > 0.f;
  const unsigned seed = drop ? eamd_drop_seed(a.drop_step, a.drop_salt) : 0u, thr = eamd_drop_thr16(a.drop_p);
  const float dinv = eamd_drop_inv(thr);
  // ---- dP tiles (gradient of the dropped probabilities -> of P), row sums of P dP ----
  float s[QT];
  {
    float4 df[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      s[qt] = 0.f;
      load_frag_t(a_dctx + ((long)b * T1 + qrow[qt]) * a.ldd + h * ATT_DK, fq, df[qt]);
    }
    const T* vs = a_v + (long)b * T2 * a.ldv + h * ATT_DK;
    float4 vf[4], vfn[4];
    if (wave < nkt) load_frag_t(vs + (long)min(wave * 16 + fr, T2 - 1) * a.ldv, fq, vf);
    for (int kt = wave; kt < nkt; kt += NW) {
      if (kt + NW < nkt) load_frag_t(vs + (long)min((kt + NW) * 16 + fr, T2 - 1) * a.ldv, fq, vfn);
      const int j0 = kt * 16 + fq * 4;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 c = dot_tile(vf, df[qt], (f32x4){0.f, 0.f, 0.f, 0.f});      // c[r]: key 16 kt + 4 fq + r, query 16 qt + fr
        float4 pr = ld4(a_P + pro[qt] + min(j0, (int)a.ldp - 4));
        if (j0 >= (int)a.ldp) pr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (drop) {
          bool kp[4];
          eamd_drop_keep4(seed, (unsigned long long)(pro[qt] + min(j0, (int)a.ldp - 4)), thr, kp);
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = kp[r] ? c[r] * dinv : 0.f;
        }
        const float p4[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) if (j0 + r < T2) s[qt] += p4[r] * c[r];
        *reinterpret_cast<float4*>(&X[rowx[qt] * XLD + kt * 16 + 4 * fq]) = make_float4(c[0], c[1], c[2], c[3]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) vf[j] = vfn[j];
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    s[qt] = xsum16_32(s[qt]);
    if (fq == 0) red[wave * LQ + rowx[qt]] = s[qt];
  }
  __syncthreads();
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int rx = rowx[qt];
    float s_ = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < NW; w_ += 2) s_ += red[w_ * LQ + rx] + red[(w_ + 1) * LQ + rx];
    s[qt] = s_;
  }
  // ---- dS (+ the inverse rel_shift scatter dbd), dq^T = K^T dS^T over this wave's key tiles ----
  const T* ks = a_k + (long)b * T2 * a.ldk + h * ATT_DK;
  f32x4 C[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) C[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 kreg[4];
  if (wave < nkt) tile_load(ks, a.ldk, wave * 16, live ? T2 : 1, lane, kreg);
  for (int kt = wave; kt < nkt; kt += NW) {
    tile_store(kreg, Tw, lane);
    if (kt + NW < nkt) tile_load(ks, a.ldk, (kt + NW) * 16, live ? T2 : 1, lane, kreg);
    const int j0 = kt * 16 + fq * 4;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int i = r0w + fr + 16 * qt;
      float4 pr = ld4(a_P + pro[qt] + min(j0, (int)a.ldp - 4));
      if (j0 >= (int)a.ldp) pr = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 dp = *reinterpret_cast<const float4*>(&X[rowx[qt] * XLD + kt * 16 + 4 * fq]);
      const float p4[4] = {pr.x, pr.y, pr.z, pr.w}, d4[4] = {dp.x, dp.y, dp.z, dp.w};
      f32x4 G;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + r;
        const float g = j < T2 ? p4[r] * (d4[r] - s[qt]) * a.scale : 0.f;
        G[r] = g;
        if (a.dbd && qok[qt]) {
          if (j < Ts && i < Ts) {
            const int R = j <= i ? i : i + 1, c = j <= i ? Ts + j - i : j - i - 1;
            if (c != 0) st1(a_dbd + zo + (long)R * a.ldp + (c - 1), g);
          } else if (j < (int)a.ldp) {
            st1(a_dbd + zo + (long)i * a.ldp + j, 0.f);
          }
        }
      }
      if (qok[qt] && j0 < (int)a.ldp) st4(a_dS + pro[qt] + j0, make_float4(G[0], G[1], G[2], G[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) G[r] = rnd(a_dS, G[r]);       // dq from the dS the key-side kernels will read
      tile_product(Tw, G, fr, fq, C[qt]);
    }
  }
  if (a.dbd && live && r0w == 0)                      // the head of row 0 the scatter never reaches
    for (int f = 1 + t; f < Ts; f += NT) st1(a_dbd + zo + (f - 1), 0.f);
  __syncthreads();
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    if (qt > 0) __syncthreads();
    const int nqt = live ? max(0, min(16, nq - 16 * qt)) : 0;
    const long orow = ((long)b * T1 + r0w + 16 * qt) * a.ldo + h * ATT_DK;
    if (dq_bf16)
      reduce_store_ct<NW>(C[qt], X, wave, fr, fq, t, reinterpret_cast<bfbits*>(a.dq) + orow, a.ldo, nqt);
    else
      reduce_store_ct<NW>(C[qt], X, wave, fr, fq, t, a.dq + orow, a.ldo, nqt);
  }
}

template <typename T, bool REL, int LQ, int NW>
int launch_fwd_long_q(AttnF32Args a, hipStream_t stream) {
  const int nkt = (a.T2 + 15) / 16;
  const size_t smem = ((size_t)LQ * (nkt * 16 + 4) + NW * 16 * PLD + 2 * NW * LQ) * sizeof(float);
  if (smem > 160 * 1024 || smem < (size_t)NW * 64 * 17 * sizeof(float)) return EAMD_EUNSUPPORTED;
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_long_kernel<T, REL, LQ, NW>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr_err != hipSuccess) return (int)attr_err;
  a.nqb = (a.T1 + LQ - 1) / LQ;
  const int nz = (a.B * a.H + 7) / 8 * 8;
  if ((long)a.nqb * nz >= (1L << 31)) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL((attn_fwd_long_kernel<T, REL, LQ, NW>), dim3((unsigned)(a.nqb * nz)), dim3(64 * NW), smem, stream, a, nkt);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
// queries and waves per workgroup.  These kernels are bound by latency, not by their L2 traffic: measured (tools/attn_long_lq_probe.py,
// B = 16, forward / query-side backward TFLOP/s) 16 queries x 4 waves give 45 - 49 / 32 - 39 while TWO workgroups share a CU (LDS <= 80 KB:
// up to ~976 keys) and 30 - 32 / 20 - 23 beyond that, where one workgroup = four waves per CU is left; 32 queries (two 16-query tiles per
// wave: every key / value / position fragment fetched serves both) give 41 - 46 / 26 - 30 wherever they fit (1100 keys) - worse than two
// resident workgroups, better than one.  Beyond 976 keys the one workgroup a CU holds therefore runs EIGHT waves (the key tiles
// split eight ways: 45 - 48 / 36 up to 1900 keys), six where the tile buffers of eight no longer fit (2048 keys); 8 queries beyond
// 2048 keys (to 4096: the MFMA tiles run half empty).
struct LongGeom { int lq, nw; };
inline LongGeom long_geom(int T1, int T2) {
  static const int force_lq = getenv("EAMD_ATTN_LONG_LQ") ? atoi(getenv("EAMD_ATTN_LONG_LQ")) : 0;      // A/B knobs
  static const int force_nw = getenv("EAMD_ATTN_LONG_NW") ? atoi(getenv("EAMD_ATTN_LONG_NW")) : 0;
  const int nkt = (T2 + 15) / 16;
  auto bytes = [&](int lq, int nw) { return ((size_t)lq * (nkt * 16 + 4) + nw * 16 * PLD + 2 * nw * lq) * sizeof(float); };
  LongGeom g{16, 4};
  if (bytes(16, 4) > 160 * 1024) g = LongGeom{8, 4};
  if (bytes(g.lq, 4) > 80 * 1024) g.nw = bytes(g.lq, 8) <= 160 * 1024 ? 8 : bytes(g.lq, 6) <= 160 * 1024 ? 6 : 4;   // one workgroup per CU: more waves
  if ((force_lq == 32 || force_lq == 16 || force_lq == 8) && bytes(force_lq, 4) <= 160 * 1024) g = LongGeom{force_lq, 4};
  if ((force_nw == 4 || force_nw == 6 || force_nw == 8) && g.lq != 32 && bytes(g.lq, force_nw) <= 160 * 1024) g.nw = force_nw;
  (void)T1;
  return g;
}
template <typename T, bool REL>
int launch_fwd_long(AttnF32Args a, hipStream_t stream) {
  const LongGeom g = long_geom(a.T1, a.T2);
  if (g.lq == 32) return launch_fwd_long_q<T, REL, 32, 4>(a, stream);
  if (g.lq == 16) return g.nw == 8 ? launch_fwd_long_q<T, REL, 16, 8>(a, stream) : g.nw == 6 ? launch_fwd_long_q<T, REL, 16, 6>(a, stream)
                                                                                               : launch_fwd_long_q<T, REL, 16, 4>(a, stream);
  return g.nw == 8 ? launch_fwd_long_q<T, REL, 8, 8>(a, stream) : g.nw == 6 ? launch_fwd_long_q<T, REL, 8, 6>(a, stream)
                                                                              : launch_fwd_long_q<T, REL, 8, 4>(a, stream);
}

template <typename T, int LQ, int NW>
int launch_bwd_long_q(AttnF32BwdArgs a, int dq_bf16, hipStream_t stream) {
  const int nkt = (a.T2 + 15) / 16;
  const size_t smem = ((size_t)LQ * (nkt * 16 + 4) + NW * 16 * PLD + 2 * NW * LQ) * sizeof(float);
  if (smem > 160 * 1024 || smem < (size_t)NW * 64 * 17 * sizeof(float)) return EAMD_EUNSUPPORTED;
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_long_kernel<T, LQ, NW>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr_err != hipSuccess) return (int)attr_err;
  a.nqb = (a.T1 + LQ - 1) / LQ;
  const int nz = (a.B * a.H + 7) / 8 * 8;
  if ((long)a.nqb * nz >= (1L << 31)) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL((attn_bwd_q_long_kernel<T, LQ, NW>), dim3((unsigned)(a.nqb * nz)), dim3(64 * NW), smem, stream, a, nkt, dq_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
template <typename T>
int launch_bwd_long(AttnF32BwdArgs a, int dq_bf16, hipStream_t stream) {
  const LongGeom g = long_geom(a.T1, a.T2);
  if (g.lq == 32) return launch_bwd_long_q<T, 32, 4>(a, dq_bf16, stream);
  // (six waves: measured SLOWER than four in this kernel at 2048 keys - 1058 against 760 us - while the forward gains, 639 against 820)
  if (g.lq == 16) return g.nw == 8 ? launch_bwd_long_q<T, 16, 8>(a, dq_bf16, stream) : launch_bwd_long_q<T, 16, 4>(a, dq_bf16, stream);
  return g.nw == 8 ? launch_bwd_long_q<T, 8, 8>(a, dq_bf16, stream) : launch_bwd_long_q<T, 8, 4>(a, dq_bf16, stream);
}

template <bool REL, int NKT>
int launch_fwd(const AttnF32Args& a, size_t smem, hipStream_t stream) {
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_fwd_kernel<REL, NKT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, NKT * 16 * PLD * 4);
  if (attr_err != hipSuccess) return (int)attr_err;
  const int nz = (a.B * a.H + 7) / 8 * 8;
  hipLaunchKernelGGL((attn_f32_fwd_kernel<REL, NKT>), dim3((unsigned)(a.nqb * nz)), dim3(256), smem, stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

template <int NKT>
int launch_bwd(const AttnF32BwdArgs& a, hipStream_t stream) {
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_bwd_q_kernel<NKT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, NKT * 16 * PLD * 4);
  if (attr_err != hipSuccess) return (int)attr_err;
  const int nz = (a.B * a.H + 7) / 8 * 8;
  hipLaunchKernelGGL((attn_f32_bwd_q_kernel<NKT>), dim3((unsigned)(a.nqb * nz)), dim3(256),
                     (size_t)NKT * 16 * PLD * sizeof(float), stream, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// bf16-operand mode, rows of 513 .. ATT_LONG_MAXK keys: called by eamd_attn_fwd / eamd_attn_bwd_q (attn_fused.hip), which
// have validated the operands; pointers are bf16 tensors, leading dimensions in elements
int eamd_attn_long_fwd_bf16(const void* qu, int64_t ldq, const void* qv, int64_t ldqv, const void* k, int64_t ldk, const void* v,
                            int64_t ldv, const void* pos, int64_t ldpos, const unsigned char* mask, int64_t mb, int64_t mi,
                            void* P, int64_t ldp, void* ctx, int64_t ldc, int B, int H, int T1, int T2, float scale, void* Pd,
                            float drop_p, const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len, void* stream) {
  if (T2 > ATT_LONG_MAXK || ldq % 4 || ldk % 4 || ldv % 4 || ldc % 4 || ldp % 4) return EAMD_EUNSUPPORTED;
  AttnF32Args a;
  a.qu = (const float*)qu; a.qv = (const float*)qv; a.k = (const float*)k; a.v = (const float*)v; a.pos = (const float*)pos;
  a.mask = mask; a.P = (float*)P; a.ctx = (float*)ctx;
  a.ldq = ldq; a.ldqv = ldqv; a.ldk = ldk; a.ldv = ldv; a.ldpos = ldpos; a.ldc = ldc; a.ldp = ldp; a.mb = mb; a.mi = mi;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nqb = 0; a.scale = scale;
  a.Pd = (float*)Pd; a.drop_p = drop_p; a.drop_step = (const unsigned long long*)drop_step; a.drop_salt = drop_salt;
  a.tshift = shift_len;
  return pos ? launch_fwd_long<bfbits, true>(a, (hipStream_t)stream) : launch_fwd_long<bfbits, false>(a, (hipStream_t)stream);
}

int eamd_attn_long_bwd_q_bf16(const void* dctx, int64_t ldd, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* P,
                              int64_t ldp, void* dS, void* dbd, void* dq, int64_t ldo, int dq_is_bf16, int B, int H, int T1, int T2,
                              float scale, float drop_p, const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len,
                              void* stream) {
  if (T2 > ATT_LONG_MAXK || ldd % 4 || ldk % 4 || ldv % 4 || ldp % 4 || ldo % 4) return EAMD_EUNSUPPORTED;
  AttnF32BwdArgs a;
  a.dctx = (const float*)dctx; a.k = (const float*)k; a.v = (const float*)v; a.P = (const float*)P;
  a.dS = (float*)dS; a.dbd = (float*)dbd; a.dq = (float*)dq;
  a.ldd = ldd; a.ldk = ldk; a.ldv = ldv; a.ldp = ldp; a.ldo = ldo;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nqb = 0; a.scale = scale;
  a.drop_p = drop_p; a.drop_step = (const unsigned long long*)drop_step; a.drop_salt = drop_salt;
  a.tshift = shift_len;
  return launch_bwd_long<bfbits>(a, dq_is_bf16, (hipStream_t)stream);
}

extern "C" int eamd_attn_fwd_f32(const float* qu, int64_t ldq, const float* qv, int64_t ldqv, const float* k, int64_t ldk,
                                 const float* v, int64_t ldv, const float* pos, int64_t ldpos, const unsigned char* mask,
                                 int64_t mask_bstride, int64_t mask_qstride, float* P, int64_t ldp, float* ctx,
                                 int64_t ldc, int B, int H, int T1, int T2, int dk, float scale, float* Pd, float drop_p,
                                 const uint64_t* drop_step, uint64_t drop_salt, const int32_t* shift_len, void* stream) {
  if (!qu || !k || !v || !P || !ctx || B <= 0 || H <= 0 || T1 <= 0 || T2 <= 0) return EAMD_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (!Pd || !drop_step || !al16(Pd)))) return EAMD_EINVAL;
  if ((pos == nullptr) != (qv == nullptr)) return EAMD_EINVAL;
  if (dk != ATT_DK || T2 > ATT_LONG_MAXK || (pos && T1 != T2)) return EAMD_EUNSUPPORTED;
  if (ldq % 4 || ldk % 4 || ldv % 4 || ldc % 4 || ldp % 4 || ldp < T2 || (pos && (ldqv % 4 || ldpos % 4)))
    return EAMD_EUNSUPPORTED;
  if (!al16(qu) || !al16(k) || !al16(v) || !al16(P) || !al16(ctx) || (pos && (!al16(qv) || !al16(pos))))
    return EAMD_EUNSUPPORTED;
  if ((long)B * H * ((T1 + 63) / 64) >= (1L << 28)) return EAMD_EUNSUPPORTED;
  AttnF32Args a;
  a.qu = qu; a.qv = qv; a.k = k; a.v = v; a.pos = pos; a.mask = mask; a.P = P; a.ctx = ctx;
  a.ldq = ldq; a.ldqv = ldqv; a.ldk = ldk; a.ldv = ldv; a.ldpos = ldpos; a.ldc = ldc; a.ldp = ldp;
  a.mb = mask_bstride; a.mi = mask_qstride;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nqb = (T1 + 63) / 64; a.scale = scale;
  a.Pd = Pd; a.drop_p = drop_p; a.drop_step = (const unsigned long long*)drop_step; a.drop_salt = drop_salt;
  a.tshift = shift_len;
  if (T2 > ATT_MAXK) return pos ? launch_fwd_long<float, true>(a, (hipStream_t)stream) : launch_fwd_long<float, false>(a, (hipStream_t)stream);
  const int nkt = T2 <= 128 ? 8 : T2 <= 256 ? 16 : 32;               // key tiles of 16 the instantiation covers
  const size_t smem = (size_t)nkt * 16 * PLD * sizeof(float);        // V panel (the score matrix X is smaller)
  hipStream_t s = (hipStream_t)stream;
  if (pos) return nkt == 8 ? launch_fwd<true, 8>(a, smem, s) : nkt == 16 ? launch_fwd<true, 16>(a, smem, s) : launch_fwd<true, 32>(a, smem, s);
  return nkt == 8 ? launch_fwd<false, 8>(a, smem, s) : nkt == 16 ? launch_fwd<false, 16>(a, smem, s) : launch_fwd<false, 32>(a, smem, s);
}

extern "C" int eamd_attn_bwd_q_f32(const float* dctx, int64_t ldd, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                   const float* P, int64_t ldp, float* dS, float* dbd, float* dq, int64_t ldo, int B, int H,
                                   int T1, int T2, int dk, float scale, float drop_p, const uint64_t* drop_step,
                                   uint64_t drop_salt, const int32_t* shift_len, void* stream) {
  if (!dctx || !k || !v || !P || !dS || !dq || B <= 0 || H <= 0 || T1 <= 0 || T2 <= 0) return EAMD_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_step)) return EAMD_EINVAL;
  if (dk != ATT_DK || T2 > ATT_LONG_MAXK || (dbd && T1 != T2)) return EAMD_EUNSUPPORTED;
  if (ldd % 4 || ldk % 4 || ldv % 4 || ldp % 4 || ldp < T2 || ldo % 4) return EAMD_EUNSUPPORTED;
  if (!al16(dctx) || !al16(k) || !al16(v) || !al16(P) || !al16(dS) || !al16(dq)) return EAMD_EUNSUPPORTED;
  if ((long)B * H * ((T1 + 63) / 64) >= (1L << 28)) return EAMD_EUNSUPPORTED;
  AttnF32BwdArgs a;
  a.dctx = dctx; a.k = k; a.v = v; a.P = P; a.dS = dS; a.dbd = dbd; a.dq = dq;
  a.ldd = ldd; a.ldk = ldk; a.ldv = ldv; a.ldp = ldp; a.ldo = ldo;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nqb = (T1 + 63) / 64; a.scale = scale;
  a.drop_p = drop_p; a.drop_step = (const unsigned long long*)drop_step; a.drop_salt = drop_salt;
  a.tshift = shift_len;
  if (T2 > ATT_MAXK) return launch_bwd_long<float>(a, 0, (hipStream_t)stream);
  return T2 <= 128 ? launch_bwd<8>(a, (hipStream_t)stream) : T2 <= 256 ? launch_bwd<16>(a, (hipStream_t)stream)
                                                                         : launch_bwd<32>(a, (hipStream_t)stream);
}

// ---- backward, key side: dV = Pd^T dctx, dK = dS^T qu and (relative positions) dpos += dbd^T qv in one launch ----
// One workgroup = one (batch, head) pair x 64 keys; wave w owns keys 16 w .. 16 w + 15 and all 64 channels, and the
// workgroup walks the queries in blocks of 64.  Both operands of a product come from row-major global tensors whose
// rows are the CONTRACTED index (queries): a [64 queries][64 keys] tile of Pd / dS / dbd and a [64 queries][64 channels]
// tile of dctx / qu / qv are staged in LDS with coalesced 16-byte loads and read back as MFMA operands along the
// query axis (ds_read_b32: lane (fr, fq) takes element [4 kk + fq][.. + fr]; row stride 80 floats keeps the four fq
// groups on disjoint banks).  The result is accumulated transposed, out^T[channel][key], so that a lane ends up with
// four adjacent channels of one key (16-byte stores).  The next pair of tiles is prefetched into registers while the
// current one feeds the matrix cores.  Replaces three batched GEMMs per layer (two without relative positions).
// reference: autograd of attention.py:63-114 (dv, dk), :141-206 (the positional term).
namespace {

constexpr int KLD = 80;

struct AttnF32KvArgs {
  const float* Pd; const float* dS; const float* dbd;
  const float* dctx; const float* qu; const float* qv;
  float* dv; float* dk; float* dpos;
  long ldp, ldd, ldq, ldqv, ldo, ldpos;
  int B, H, T1, T2, nkb;
};

// rows i0 .. i0 + 63 of a [T1][ld] matrix, 64 columns from column c0 on: global -> registers; rows past T1 and
// column chunks past `ncol` (the row length that exists) are zero
__device__ __forceinline__ void kv_tile_load(const float* src, long ld, int i0, int T1, int c0, int ncol, int t, float4 (&r)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = t + 256 * q, row = c >> 4, c4 = c & 15;
    const bool ok = i0 + row < T1 && c0 + c4 * 4 < ncol;
    r[q] = *reinterpret_cast<const float4*>(src + (long)min(i0 + row, T1 - 1) * ld + (ok ? c0 + c4 * 4 : 0));
    if (!ok) r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void kv_tile_store(const float4 (&r)[4], float* dst, int t) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = t + 256 * q, row = c >> 4, c4 = c & 15;
    *reinterpret_cast<float4*>(&dst[row * KLD + c4 * 4]) = r[q];
  }
}
// acc^T[channel tile ct][key tile of this wave] += X[query][channel]^T  W[query][key]
__device__ __forceinline__ void kv_product(const float* Xc, const float* Wk, int wave, int fr, int fq, f32x4 (&acc)[4]) {
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int i = 4 * kk + fq;
    const float b = Wk[i * KLD + wave * 16 + fr];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Xc[i * KLD + ct * 16 + fr], b, acc[ct], 0, 0, 0);
  }
}

template <bool REL>
__global__ __launch_bounds__(256, 2) void attn_f32_bwd_kv_kernel(const AttnF32KvArgs a) {
  __shared__ __attribute__((aligned(16))) float sW[64 * KLD];     // [query][key] tile
  __shared__ __attribute__((aligned(16))) float sX[64 * KLD];     // [query][channel] tile
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int z = (jb / a.nkb) * 8 + xcd;            // z = h * B + b
  if (z >= a.B * a.H) return;                      // whole workgroup
  const int h = z / a.B, b = z % a.B;
  const int T1 = a.T1, T2 = a.T2;
  const int j0 = (jb % a.nkb) * 64;                // first key of the workgroup
  const long zo = (long)z * T1 * a.ldp;
  const float* Wsrc[3] = {a.Pd + zo, a.dS + zo, REL ? a.dbd + zo : nullptr};
  const float* Xsrc[3] = {a.dctx + (long)b * T1 * a.ldd + h * ATT_DK, a.qu + (long)b * T1 * a.ldq + h * ATT_DK,
                          REL ? a.qv + (long)b * T1 * a.ldqv + h * ATT_DK : nullptr};
  const long Xld[3] = {a.ldd, a.ldq, a.ldqv};
  constexpr int NP = REL ? 3 : 2;
  f32x4 acc[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[p][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nstage = ((T1 + 63) / 64) * NP;
  float4 rw[4], rx[4];
  kv_tile_load(Wsrc[0], a.ldp, 0, T1, j0, (int)a.ldp, t, rw);
  kv_tile_load(Xsrc[0], Xld[0], 0, T1, 0, 64, t, rx);
  for (int s = 0; s < nstage; ++s) {
    const int p = s % NP;
    __syncthreads();                               // the previous stage's reads are done
    kv_tile_store(rw, sW, t);
    kv_tile_store(rx, sX, t);
    __syncthreads();
    if (s + 1 < nstage) {                          // next pair of tiles: in flight under the MFMAs
      const int pn = (s + 1) % NP, i0 = (s + 1) / NP * 64;
      if (pn == 0) { kv_tile_load(Wsrc[0], a.ldp, i0, T1, j0, (int)a.ldp, t, rw); kv_tile_load(Xsrc[0], Xld[0], i0, T1, 0, 64, t, rx); }
      else if (pn == 1) { kv_tile_load(Wsrc[1], a.ldp, i0, T1, j0, (int)a.ldp, t, rw); kv_tile_load(Xsrc[1], Xld[1], i0, T1, 0, 64, t, rx); }
      else if (REL) { kv_tile_load(Wsrc[2], a.ldp, i0, T1, j0, (int)a.ldp, t, rw); kv_tile_load(Xsrc[2], Xld[2], i0, T1, 0, 64, t, rx); }
    }
    if (p == 0) kv_product(sX, sW, wave, fr, fq, acc[0]);
    else if (p == 1) kv_product(sX, sW, wave, fr, fq, acc[1]);
    else if (REL) kv_product(sX, sW, wave, fr, fq, acc[NP - 1]);
  }
  const int j = j0 + wave * 16 + fr;               // this lane's key; acc[.][ct][r] = channel 16 ct + 4 fq + r
  if (j < T2) {
    float* vrow = a.dv + ((long)b * T2 + j) * a.ldo + h * ATT_DK + 4 * fq;
    float* krow = a.dk + ((long)b * T2 + j) * a.ldo + h * ATT_DK + 4 * fq;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      *reinterpret_cast<float4*>(vrow + ct * 16) = make_float4(acc[0][ct][0], acc[0][ct][1], acc[0][ct][2], acc[0][ct][3]);
      *reinterpret_cast<float4*>(krow + ct * 16) = make_float4(acc[1][ct][0], acc[1][ct][1], acc[1][ct][2], acc[1][ct][3]);
    }
    if (REL) {                                     // positions are shared by the batch: accumulate
      float* prow = a.dpos + (long)j * a.ldpos + h * ATT_DK + 4 * fq;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(prow + ct * 16 + r, acc[NP - 1][ct][r]);
    }
  }
}

}  // namespace

extern "C" int eamd_attn_bwd_kv_f32(const float* Pd, const float* dS, const float* dbd, int64_t ldp, const float* dctx,
                                    int64_t ldd, const float* qu, int64_t ldq, const float* qv, int64_t ldqv, float* dv,
                                    float* dk_out, int64_t ldo, float* dpos, int64_t ldpos, int B, int H, int T1, int T2, int dk,
                                    void* stream) {
  if (!Pd || !dS || !dctx || !qu || !dv || !dk_out || B <= 0 || H <= 0 || T1 <= 0 || T2 <= 0) return EAMD_EINVAL;
  if ((dbd == nullptr) != (qv == nullptr) || (dbd == nullptr) != (dpos == nullptr)) return EAMD_EINVAL;
  if (dk != ATT_DK || (dbd && T1 != T2)) return EAMD_EUNSUPPORTED;
  if (ldp % 4 || ldp < T2 || ldd % 4 || ldq % 4 || ldo % 4 || (dbd && (ldqv % 4 || ldpos % 4))) return EAMD_EUNSUPPORTED;
  if (!al16(Pd) || !al16(dS) || !al16(dctx) || !al16(qu) || !al16(dv) || !al16(dk_out) ||
      (dbd && (!al16(dbd) || !al16(qv) || !al16(dpos))))
    return EAMD_EUNSUPPORTED;
  AttnF32KvArgs a;
  a.Pd = Pd; a.dS = dS; a.dbd = dbd; a.dctx = dctx; a.qu = qu; a.qv = qv; a.dv = dv; a.dk = dk_out; a.dpos = dpos;
  a.ldp = ldp; a.ldd = ldd; a.ldq = ldq; a.ldqv = ldqv; a.ldo = ldo; a.ldpos = ldpos;
  a.B = B; a.H = H; a.T1 = T1; a.T2 = T2; a.nkb = (T2 + 63) / 64;
  if ((long)B * H * a.nkb >= (1L << 28)) return EAMD_EUNSUPPORTED;
  const int nz = (B * H + 7) / 8 * 8;
  hipStream_t s = (hipStream_t)stream;
  if (dbd) hipLaunchKernelGGL((attn_f32_bwd_kv_kernel<true>), dim3((unsigned)(a.nkb * nz)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_f32_bwd_kv_kernel<false>), dim3((unsigned)(a.nkb * nz)), dim3(256), 0, s, a);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
