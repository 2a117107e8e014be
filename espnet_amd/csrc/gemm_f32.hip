// fp32-operand MFMA GEMM for gfx950: the reference-precision path of eamd_gemm (in_dtype = 0, precision = 0).
//
// Arithmetic: v_mfma_f32_16x16x4_f32 - fp32 operands, fp32 products, fp32 accumulation (bitwise an fmaf chain):
// the reference's own number format.  Its matrix rate is 64 FLOP/clk/SIMD = 157 TFLOP/s for the chip, 1/16 of the
// bf16 rate, so unlike the bf16 kernels this one is bound by the matrix pipe, and the whole job of the code around
// the MFMAs is to never make them wait:
//   * operand tiles go global -> VGPR (16-byte chunks) -> LDS through a register ring; a tile's loads are issued
//     one or two MFMA phases (1024 - 4096 cycles each) before their first use and NOTHING touches the loaded
//     registers in between, so hipcc emits one counted s_waitcnt in front of the LDS stores instead of draining
//     vmcnt before the MFMAs (the generic kernel in gemm.hip guards every load with a branch and waits for each
//     on the spot: 30-50 TFLOP/s on every shape of the model);
//   * loads are branch-free: rows / column chunks past the M or N edge are clamped to the last valid one (they only
//     reach output rows / columns the epilogue never stores), only the ragged last K-tile is masked to zero;
//   * k-contiguous operands ([rows][K]) sit in LDS as [row][32 + 4] and are read as ds_read_b128 (4 k-steps per
//     read); k-strided operands ([K][cols]: dX = dY W, dW = dY^T X) sit as [k][cols + 4] and are read as
//     ds_read_b32, conflict-free for the 16x16x4 operand layout (lane = (row, k group)); no transposed copy of an
//     activation or weight is ever made;
//   * implicit-GEMM convolution (gathered A rows / gathered reduction rows), split-K with f32 atomics, fused
//     bias-gradient column sums, operand activations and the LDS-staged 16-byte epilogue (bias, ReLU / Swish and
//     their derivative masks, alpha, residual, beta, row maps) are the same features as the bf16 kernel's.
// 256 threads = 4 waves (2 x 2); block tile 128x128 (wave tile 64x64 = 16 accumulators) or 64x64; BK = 32.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"

#include "gemm_bf16_common.h"      // store_c_tile (result epilogue), gather helpers

namespace {

constexpr int FBK = 32;

template <int BM, int BN, bool TA, bool TB>
struct SmemF {
  // k-strided images [k][cols + pad], read with ds_read_b32 by lanes (col fr, k-group fq): the four fq groups are EQ k-rows
  // apart (EQ = 4 for the 64x64 tile, 2 for 128x128), and the two groups of a 32-lane half must land 16 banks apart:
  // EQ * (cols + pad) = 16 (mod 32) -> pad 4 at 64 columns, pad 8 at 128 (with pad 4 the 128-column image shifted by 8 banks:
  // SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE in the grouped weight-gradient launch)
  static constexpr int KPAD_A = BM >= 128 ? 8 : 4, KPAD_B = BN >= 128 ? 8 : 4;
  static constexpr int LDA = TA ? BM + KPAD_A : FBK + 4;     // floats per image row
  static constexpr int RA = TA ? FBK : BM;
  static constexpr int LDB = TB ? BN + KPAD_B : FBK + 4;
  static constexpr int RB = TB ? FBK : BN;
  static constexpr int OPS = 2 * (RA * LDA + RB * LDB);
  static constexpr int CT = BM * (BN + 4);
  float ab[OPS > CT ? OPS : CT];                         // operand buffers, re-used as the result staging tile
  int poff[8][FBK];
  __device__ float* a(int buf) { return ab + buf * RA * LDA; }
  __device__ float* b(int buf) { return ab + 2 * RA * LDA + buf * RB * LDB; }
};

__device__ __forceinline__ float4 act4(float4 v, int act) {
  if (act == EAMD_ACT_SWISH) return make_float4(eamd_swish(v.x), eamd_swish(v.y), eamd_swish(v.z), eamd_swish(v.w));
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ float4 mask4(float4 v, int nv) {
  return make_float4(nv > 0 ? v.x : 0.f, nv > 1 ? v.y : 0.f, nv > 2 ? v.z : 0.f, nv > 3 ? v.w : 0.f);
}

// One workgroup's share of a problem: `bid` of `nblk` workgroups (tile x split-K slice), batch index `zb`.
// gemm_f32_kernel runs it on a launch of its own; gemm_f32_group_kernel looks the problem up in a device table.
template <int BM, int BN, bool TA, bool TB, bool GAT, bool ACT, bool ROWEPI = false, bool NOPAD = false>
__device__ __forceinline__ void gemm_f32_body(const eamd_gemm_t& p, const int bid, const int nblk, const int zb) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 16, NTL = WN / 16;
  constexpr int NCA = BM / 32, NCB = BN / 32;           // 16-byte chunks per thread per tile
  constexpr int CPR_A = BM / 4, RPP_A = NT_ / CPR_A;    // k-strided image: chunks per k-row, k-rows per pass
  constexpr int CPR_B = BN / 4, RPP_B = NT_ / CPR_B;
  using S = SmemF<BM, BN, TA, TB>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  S& sm = *reinterpret_cast<S*>(smem_raw);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (see gemm_bf16.hip): every XCD gets one contiguous run of row-major tiles; with split-K
  // the split index is the fastest-varying part of the workgroup id
  const int tiles_n = (p.N + BN - 1) / BN;
  int tile_id, split;
  if (nblk < 0) {
    // grouped launch: -nblk workgroups (a multiple of 8, so bid & 7 is the XCD) serve tiles x splits in split-major
    // order, every XCD one contiguous run of it: an XCD works on one K-slice (or part of one) of neighbouring tiles,
    // so the operand panels they share are fetched from HBM once into that XCD's L2 (dealt round-robin, the same
    // launch fetched 3.8x its operand bytes)
    const int ntile = ((p.M + BM - 1) / BM) * tiles_n;
    const int per = (-nblk) >> 3;
    const int v = (bid & 7) * per + (bid >> 3);
    if (v >= ntile * p.splitk) return;       // padding workgroup
    split = v / ntile;
    tile_id = v - split * ntile;
  } else if (p.splitk > 1) {
    split = bid % p.splitk;
    tile_id = bid / p.splitk;
  } else {
    split = 0;
    const int ntile = nblk;
    const int id = bid, q = ntile >> 3, r = ntile & 7, xcd = id & 7, j = id >> 3;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m = tile_id / tiles_n, tile_n = tile_id % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int b1 = zb / p.batch2, b2 = zb % p.batch2;
  const float* __restrict__ A = p.A + b1 * p.sA1 + b2 * p.sA2;
  const float* __restrict__ B = p.B + b1 * p.sB1 + b2 * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;

  const int nkt_total = (p.K + FBK - 1) / FBK;
  const int per = (nkt_total + p.splitk - 1) / p.splitk;
  const int kt_begin = split * per;
  const int kt_end = min(nkt_total, kt_begin + per);
  const int nkt = kt_end - kt_begin;

  // ---- staging coordinates: LDS image (row, 16-byte chunk) of every staged chunk ----
  int a_r[NCA], a_c[NCA], b_r[NCB], b_c[NCB];
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    if constexpr (TA) { a_r[i] = t / CPR_A + RPP_A * i; a_c[i] = t % CPR_A; }
    else              { a_r[i] = t / 8 + 32 * i;        a_c[i] = t % 8; }
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    if constexpr (TB) { b_r[i] = t / CPR_B + RPP_B * i; b_c[i] = t % CPR_B; }
    else              { b_r[i] = t / 8 + 32 * i;        b_c[i] = t % 8; }
  }
  long a_off[NCA], b_off[NCB];
  if constexpr (!GAT) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      if constexpr (!TA) a_off[i] = (long)min(m0 + a_r[i], p.M - 1) * p.lda + a_c[i] * 4;
      else a_off[i] = (long)a_r[i] * p.lda + min(m0 + a_c[i] * 4, (p.M - 1) & ~3);
    }
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    if constexpr (!TB) b_off[i] = (long)min(n0 + b_r[i], p.N - 1) * p.ldb + b_c[i] * 4;
    else b_off[i] = (long)b_r[i] * p.ldb + min(n0 + b_c[i] * 4, (p.N - 1) & ~3);
  }
  RowStateB a_rs[NCA];
  // gathered A rows: everything the K loop needs of the descriptor is taken into scalar registers HERE.  A scalar load inside
  // the loop (p.gather.dh[tap] once per tile, and - when the descriptor sits behind a run-time index, as in the
  // eamd_gemm_multi kernel - every p.* the compiler may not speculate) is followed by s_waitcnt lgkmcnt(0), which also
  // drains the wave's LDS reads: the tap offsets travel as 4-bit fields of two 64-bit scalars (host: -8 <= dh, dw <= 7).
  unsigned long long dh_pack = 0ull, dw_pack = 0ull;
  int gC = 1, gHin = 0, gWin = 0;
  if constexpr (GAT) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      dh_pack |= (unsigned long long)((unsigned)(p.gather.dh[tp] + 8) & 15u) << (4 * tp);
      dw_pack |= (unsigned long long)((unsigned)(p.gather.dw[tp] + 8) & 15u) << (4 * tp);
    }
    gC = p.gather.C; gHin = p.gather.Hin; gWin = p.gather.Win;
  }
  const int pK = p.K;
  const long ldaL = p.lda, ldbL = p.ldb;
  auto tap_dh = [&](int tap) __attribute__((always_inline)) { return (int)((dh_pack >> (4 * tap)) & 15ull) - 8; };
  auto tap_dw = [&](int tap) __attribute__((always_inline)) { return (int)((dw_pack >> (4 * tap)) & 15ull) - 8; };
  auto goff = [&](const RowStateB& rs, int dh, int dw) __attribute__((always_inline)) -> long {   // gather_off_b on the local copies
    const int hh = rs.ih + dh, ww = rs.jw + dw;
    const bool ok = rs.ok && hh >= 0 && hh < gHin && ww >= 0 && ww < gWin;
    return ok ? ((long)(rs.base + hh * gWin + ww)) * gC : -1L;
  };
  // a convolution without padding (Conv2dSubsampling's second 3x3 / stride-2 layer: every tap of every output pixel lies
  // inside the source) needs no per-tile range check: offset = row base + a tap offset that is uniform over the workgroup
  constexpr bool g_nopad = NOPAD;         // checked on the host (gather_nopad): no tap of any output pixel leaves the source
  if constexpr (GAT && !TA) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) a_rs[i] = decompose_b(p.gather, m0 + a_r[i], p.M);
  }
  // transposed gather (weight gradient: the reduction runs over the output pixels): lanes 0 .. FBK-1 of the workgroup keep
  // the (b, i, j) of the next reduction row they publish and step it by FBK rows per K-tile instead of dividing again
  int pr_row = 0, pr_j = 0, pr_i = 0, pr_b = 0;
  if constexpr (GAT && TA) {
    if (t < FBK) {
      pr_row = kt_begin * FBK + t;
      pr_j = pr_row % p.gather.Wo;
      const int tt = pr_row / p.gather.Wo;
      pr_i = tt % p.gather.Ho;
      pr_b = tt / p.gather.Ho;
    }
  }
  // register ring: DEPTH-1 tiles of loads in flight across the MFMA phases.  A 128x128 phase is 4096 MFMA cycles
  // per wave (longer than an HBM round trip): one tile ahead is enough; 64x64 phases are 1024 cycles: two ahead.
  constexpr int DEPTH = BM >= 128 ? 2 : 3;
  constexpr int UNROLL = DEPTH % 2 ? 2 * DEPTH : DEPTH;
  float4 ra[DEPTH][NCA], rb[DEPTH][NCB];
  // operand-side dropout (ACT instantiations only): seeds once, one hash pair per staged 4-element chunk
  const unsigned a_dseed = (ACT && p.a_drop_p > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.a_drop_salt) : 0u;
  const unsigned b_dseed = (ACT && p.b_drop_p > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.b_drop_salt) : 0u;
  const bool do_colsum = TA && p.colsum != nullptr && !GAT && tile_n == 0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};

  const int gWo = GAT ? p.gather.Wo : 1, gHo = GAT ? p.gather.Ho : 1, gsh = GAT ? p.gather.sh : 0, gsw = GAT ? p.gather.sw : 0;
  const int ta_tap = (GAT && TA) ? m0 / gC : 0;                      // transposed gather: one tap per M-tile
  const int ta_dh = tap_dh(ta_tap), ta_dw = tap_dw(ta_tap);
  auto fill_poff = [&](int kt, int slot) __attribute__((always_inline)) {      // called for kt_begin, kt_begin + 1, ... in order
    if (t < FBK) {
      RowStateB s;
      s.ok = pr_row < pK;
      s.base = pr_b * gHin * gWin; s.ih = pr_i * gsh; s.jw = pr_j * gsw;
      sm.poff[slot][t] = (int)goff(s, ta_dh, ta_dw);
      pr_row += FBK; pr_j += FBK;
      while (pr_j >= gWo) { pr_j -= gWo; ++pr_i; }
      while (pr_i >= gHo) { pr_i -= gHo; ++pr_b; }
    }
  };

  // GUARD = the tile may be the ragged last one (k0 + FBK > K): out-of-range chunk starts are redirected to the
  // start of the row / to reduction row 0 and zeroed at store time
  auto load_tile = [&](auto set_c, auto guard_c, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    constexpr bool GUARD = decltype(guard_c)::value;
    const int k0 = kt * FBK;
    if constexpr (!TA) {
      if constexpr (!GAT) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const int kk = (GUARD && k0 + a_c[i] * 4 >= pK) ? -a_c[i] * 4 : k0;
          ra[SET][i] = *reinterpret_cast<const float4*>(A + a_off[i] + kk);
        }
      } else {
        const int tap = k0 / gC, c0 = k0 - tap * gC;
        const int dh = tap_dh(tap), dw = tap_dw(tap);                                                     // uniform
        const int tappix = dh * gWin + dw;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          long off;
          if (g_nopad) off = a_rs[i].ok ? (long)(a_rs[i].base + a_rs[i].ih * gWin + a_rs[i].jw + tappix) * gC : -1L;
          else off = goff(a_rs[i], dh, dw);
          ra[SET][i] = *reinterpret_cast<const float4*>(A + (off >= 0 ? off + c0 + a_c[i] * 4 : 0L));
        }
      }
    } else {
      if constexpr (!GAT) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const int kk = (GUARD && k0 + a_r[i] >= pK) ? -a_r[i] : k0;
          ra[SET][i] = *reinterpret_cast<const float4*>(A + a_off[i] + (long)kk * ldaL);
        }
      } else {
        const int c = (m0 % gC) + a_c[0] * 4;
        const int slot = kt % 8;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const int off = sm.poff[slot][a_r[i]];
          ra[SET][i] = *reinterpret_cast<const float4*>(A + (off >= 0 ? (long)off + c : 0L));
        }
      }
    }
    if constexpr (!TB) {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const int kk = (GUARD && k0 + b_c[i] * 4 >= pK) ? -b_c[i] * 4 : k0;
        rb[SET][i] = *reinterpret_cast<const float4*>(B + b_off[i] + kk);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const int kk = (GUARD && k0 + b_r[i] >= pK) ? -b_r[i] : k0;
        rb[SET][i] = *reinterpret_cast<const float4*>(B + b_off[i] + (long)kk * ldbL);
      }
    }
  };

  auto store_tile = [&](auto set_c, auto guard_c, int buf, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    constexpr bool GUARD = decltype(guard_c)::value;
    const int k0 = kt * FBK;
    if constexpr (GAT) {   // gathered rows are either fully valid or fully zero (padding taps / row tail)
      int sdh = 0, sdw = 0;
      if constexpr (!TA && !g_nopad) { const int tap = k0 / gC; sdh = tap_dh(tap); sdw = tap_dw(tap); }
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        bool ok;
        if constexpr (!TA) ok = g_nopad ? a_rs[i].ok != 0 : goff(a_rs[i], sdh, sdw) >= 0;
        else ok = sm.poff[kt % 8][a_r[i]] >= 0;
        ra[SET][i] = mask4(ra[SET][i], ok ? 4 : 0);
      }
    }
    if constexpr (GUARD) {
      if (k0 + FBK > pK) {      // ragged last K-tile: zero the out-of-range reduction elements
        if constexpr (!GAT) {
#pragma unroll
          for (int i = 0; i < NCA; ++i)
            ra[SET][i] = mask4(ra[SET][i], TA ? ((k0 + a_r[i]) < pK ? 4 : 0) : pK - (k0 + a_c[i] * 4));
        }
#pragma unroll
        for (int i = 0; i < NCB; ++i)
          rb[SET][i] = mask4(rb[SET][i], TB ? ((k0 + b_r[i]) < pK ? 4 : 0) : pK - (k0 + b_c[i] * 4));
      }
    }
    float* la = sm.a(buf);
    float* lb = sm.b(buf);
    if constexpr (ACT) {
      // activation / dropout of the staged operands, ONE chunk at a time (act -> mask -> bias-gradient sum -> LDS store,
      // then a scheduling fence): interleaving the hashes of all chunks cost 30+ VGPRs and spilled the 128x128 variant.
      // Mask index = element index in the operand's own [rows, ld] matrix = the chunk's load offset (chunks that were
      // clamped in only feed rows / columns that are never stored); operands are < 2^32 elements (checked on the host).
      const unsigned a_thr = eamd_drop_thr16(p.a_drop_p), b_thr = eamd_drop_thr16(p.b_drop_p);
      const float a_inv = eamd_drop_inv(a_thr), b_inv = eamd_drop_inv(b_thr);
      const unsigned a_k = TA ? (unsigned)k0 * (unsigned)p.lda : (unsigned)k0;
      const unsigned b_k = TB ? (unsigned)k0 * (unsigned)p.ldb : (unsigned)k0;
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        float4 v = ra[SET][i];
        if (p.a_act != EAMD_ACT_NONE) v = act4(v, p.a_act);
        if (p.a_drop_p > 0.f) {
          bool kp[4];
          eamd_drop_keep4(a_dseed, (unsigned long long)((unsigned)a_off[i] + a_k), a_thr, kp);
          v = make_float4(kp[0] ? v.x * a_inv : 0.f, kp[1] ? v.y * a_inv : 0.f, kp[2] ? v.z * a_inv : 0.f,
                          kp[3] ? v.w * a_inv : 0.f);
        }
        if (do_colsum) { cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w; }
        *reinterpret_cast<float4*>(&la[a_r[i] * S::LDA + a_c[i] * 4]) = v;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        float4 v = rb[SET][i];
        if (p.b_act != EAMD_ACT_NONE) v = act4(v, p.b_act);
        if (p.b_drop_p > 0.f) {
          bool kp[4];
          eamd_drop_keep4(b_dseed, (unsigned long long)((unsigned)b_off[i] + b_k), b_thr, kp);
          v = make_float4(kp[0] ? v.x * b_inv : 0.f, kp[1] ? v.y * b_inv : 0.f, kp[2] ? v.z * b_inv : 0.f,
                          kp[3] ? v.w * b_inv : 0.f);
        }
        *reinterpret_cast<float4*>(&lb[b_r[i] * S::LDB + b_c[i] * 4]) = v;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_colsum) {        // bias gradient of the transposed A operand
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          cs[0] += ra[SET][i].x; cs[1] += ra[SET][i].y; cs[2] += ra[SET][i].z; cs[3] += ra[SET][i].w;
        }
      }
#pragma unroll
      for (int i = 0; i < NCA; ++i) *reinterpret_cast<float4*>(&la[a_r[i] * S::LDA + a_c[i] * 4]) = ra[SET][i];
#pragma unroll
      for (int i = 0; i < NCB; ++i) *reinterpret_cast<float4*>(&lb[b_r[i] * S::LDB + b_c[i] * 4]) = rb[SET][i];
    }
  };

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr bool tgat = GAT && TA;
  if (nkt > 0) {
    if (tgat) {
      for (int q = 0; q < DEPTH && q < nkt; ++q) fill_poff(kt_begin + q, (kt_begin + q) % 8);
      __syncthreads();
    }
    load_tile(std::integral_constant<int, 0>{}, std::true_type{}, kt_begin);
    if constexpr (DEPTH == 3) {
      if (nkt > 1) load_tile(std::integral_constant<int, 1>{}, std::true_type{}, kt_begin + 1);
    }
    store_tile(std::integral_constant<int, 0>{}, std::true_type{}, 0, kt_begin);
    __syncthreads();
  }

  const int fr = lane & 15, fq = lane >> 4;
  // A K-tile is consumed in NS fragment sets of QD reduction steps each (64x64: 2 x 16 via ds_read_b128; 128x128:
  // 4 x 8 via ds_read_b64 - half the fragment registers, which is what keeps two 128x128 workgroups per CU).
  // MFMA e of set q contracts k = q*QD + EQ*fq + e.
  constexpr int QD = BM >= 128 ? 8 : 16;
  constexpr int NS = FBK / QD, EQ = QD / 4;
  auto read_frags = [&](int buf, int q, float (&af)[MT][EQ], float (&bfr)[NTL][EQ]) __attribute__((always_inline)) {
    const float* la = sm.a(buf);
    const float* lb = sm.b(buf);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (!TA) {
        const float* src = &la[(wm * WM + i * 16 + fr) * S::LDA + q * QD + fq * EQ];
        if constexpr (EQ == 4) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(src);
          af[i][0] = v.x; af[i][1] = v.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < EQ; ++e) af[i][e] = la[(q * QD + fq * EQ + e) * S::LDA + wm * WM + i * 16 + fr];
      }
    }
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      if constexpr (!TB) {
        const float* src = &lb[(wn * WN + j * 16 + fr) * S::LDB + q * QD + fq * EQ];
        if constexpr (EQ == 4) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          bfr[j][0] = v.x; bfr[j][1] = v.y; bfr[j][2] = v.z; bfr[j][3] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(src);
          bfr[j][0] = v.x; bfr[j][1] = v.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < EQ; ++e) bfr[j][e] = lb[(q * QD + fq * EQ + e) * S::LDB + wn * WN + j * 16 + fr];
      }
    }
  };
  auto mfma_set = [&](const float (&af)[MT][EQ], const float (&bfr)[NTL][EQ]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < EQ; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
  };
  // Software pipeline of one K-tile (fragment sets double-buffered in registers; set 0 of a tile is read at the END of
  // the previous phase, behind the barrier but in front of that phase's last MFMA group, so no MFMA waits on a cold
  // LDS read; the LDS stores of the next tile go in front of the second-to-last MFMA group - their buffer has had no
  // readers since the previous barrier - and drain under it):
  //     global loads (tile it+DEPTH-1) | read set 1 | MFMA set 0 | ... | read set NS-1 | LDS stores (tile it+1) |
  //     MFMA set NS-2 | barrier | read set 0 of tile it+1 | MFMA set NS-1
  // A 64x64 workgroup keeps a ring of 3 register tiles (tile it+1 landed a phase ago: its stores open the phase); a
  // 128x128 one a ring of 2 (loads issued at the top of the phase, stored 2048 MFMA cycles later).
  float fa[2][MT][EQ], fb[2][NTL][EQ];
  if (nkt > 0) read_frags(0, 0, fa[0], fb[0]);
  auto phase = [&](auto idx_c, auto guard_c, int it) __attribute__((always_inline)) {
    constexpr int IDX = decltype(idx_c)::value;
    constexpr int PAR = IDX % DEPTH;
    constexpr bool GUARD = decltype(guard_c)::value;
    using load_t = std::integral_constant<int, (PAR + DEPTH - 1) % DEPTH>;
    using other_t = std::integral_constant<int, (PAR + 1) % DEPTH>;
    constexpr int buf = IDX & 1;
    if constexpr (GUARD) {
      if (tgat && it + DEPTH < nkt) fill_poff(kt_begin + it + DEPTH, (kt_begin + it + DEPTH) % 8);
      if (it + DEPTH - 1 < nkt) load_tile(load_t{}, guard_c, kt_begin + it + DEPTH - 1);
    } else {
      if constexpr (tgat) fill_poff(kt_begin + it + DEPTH, (kt_begin + it + DEPTH) % 8);
      load_tile(load_t{}, guard_c, kt_begin + it + DEPTH - 1);
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (q + 1 < NS) {
        read_frags(buf, q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if constexpr (GUARD) {
          if (it + 1 < nkt) read_frags(buf ^ 1, 0, fa[0], fb[0]);
        } else {
          read_frags(buf ^ 1, 0, fa[0], fb[0]);
        }
        // MFMAs are pure register operations: instruction selection is free to hoist the last set above the barrier
        // (it did), which puts the barrier and the cold reads of the next tile back on the critical path.  Passing
        // the set's fragments through an empty volatile asm orders its MFMAs behind the barrier.
#pragma unroll
        for (int e = 0; e < EQ; ++e) {
#pragma unroll
          for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[q & 1][i][e]));
#pragma unroll
          for (int j = 0; j < NTL; ++j) asm volatile("" : "+v"(fb[q & 1][j][e]));
        }
      }
      // the machine scheduler otherwise moves the global loads / LDS traffic next to their consumers: pin the order
      __builtin_amdgcn_sched_barrier(0);
      if (q == NS - 2) {
        if constexpr (GUARD) {
          if (it + 1 < nkt) store_tile(other_t{}, guard_c, buf ^ 1, kt_begin + it + 1);
        } else {
          store_tile(other_t{}, guard_c, buf ^ 1, kt_begin + it + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      mfma_set(fa[q & 1], fb[q & 1]);
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  int it = 0;
  // unguarded groups: the last phase of a group loads tile it + UNROLL + DEPTH - 2, which must not be the (possibly
  // ragged) last tile
  for (; it + UNROLL + DEPTH - 1 < nkt; it += UNROLL) {
    phase(std::integral_constant<int, 0>{}, F_{}, it);
    phase(std::integral_constant<int, 1>{}, F_{}, it + 1);
    if constexpr (UNROLL == 6) {
      phase(std::integral_constant<int, 2>{}, F_{}, it + 2);
      phase(std::integral_constant<int, 3>{}, F_{}, it + 3);
      phase(std::integral_constant<int, 4>{}, F_{}, it + 4);
      phase(std::integral_constant<int, 5>{}, F_{}, it + 5);
    }
  }
  for (; it < nkt; it += UNROLL) {
    phase(std::integral_constant<int, 0>{}, T_{}, it);
    if (it + 1 < nkt) phase(std::integral_constant<int, 1>{}, T_{}, it + 1);
    if constexpr (UNROLL == 6) {
      if (it + 2 < nkt) phase(std::integral_constant<int, 2>{}, T_{}, it + 2);
      if (it + 3 < nkt) phase(std::integral_constant<int, 3>{}, T_{}, it + 3);
      if (it + 4 < nkt) phase(std::integral_constant<int, 4>{}, T_{}, it + 4);
      if (it + 5 < nkt) phase(std::integral_constant<int, 5>{}, T_{}, it + 5);
    }
  }

  __syncthreads();      // the last phase's trailing MFMAs ran behind its barrier: nothing reads the operand buffers now
  if (TA && p.colsum != nullptr && !GAT) {   // block-uniform
    // bias gradient: per-thread column sums -> one LDS row per k-row group of the staging layout -> one global
    // atomic per column per block (plain LDS stores: ds_add_f32 is slow on gfx950)
    float* csl = sm.ab;
    constexpr int NSLOT = NT_ / CPR_A;
    static_assert(NSLOT * BM <= S::OPS, "column-sum slots must fit in the operand buffers");
    __syncthreads();
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 4; ++j) csl[(t / CPR_A) * BM + a_c[0] * 4 + j] = cs[j];
    }
    __syncthreads();
    if (do_colsum) {
      for (int i = t; i < BM; i += NT_) {
        const int m = m0 + i;
        float v = 0.f;
#pragma unroll 8
        for (int sl = 0; sl < NSLOT; ++sl) v += csl[sl * BM + i];
        if (m < p.M) atomicAdd(p.colsum + (long)zb * p.M + m, v * p.alpha);
      }
    }
    __syncthreads();
  }

  // ---- epilogue ----
  if (p.splitk > 1) {
    const bool lead = split == 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * WM + i * 16 + fq * 4 + r;
        if (m >= p.M) continue;
        long prow = m;
        if (p.cmap.enabled) {
          const eamd_rowmap_t& c = p.cmap;
          int jj = m % c.Wo; int tt = m / c.Wo; int ii = tt % c.Ho; int bb = tt / c.Ho;
          prow = ((long)bb * c.Hc + ii * c.sh + c.oh) * c.Wc + jj * c.sw + c.ow;
        }
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          const int n = n0 + wn * WN + j * 16 + fr;
          if (n >= p.N) continue;
          float v = acc[i][j][r];
          if (lead && p.bias) v += p.bias[n];
          v *= p.alpha;
          if (lead && p.R) v += p.R[coff + prow * p.ldr + n];
          atomicAdd(p.C + coff + prow * p.ldc + n, v);
        }
      }
    }
    return;
  }
  store_c_tile<BM, BN, true, ROWEPI>(p, acc, sm.ab, m0, n0, coff);      // operand buffers are free now (last phase ended in a barrier)
}

template <int BM, int BN, bool TA, bool TB, bool GAT, bool ACT, bool ROWEPI = false, bool NOPAD = false>
__global__ __launch_bounds__(NT_, 2) void gemm_f32_kernel(const eamd_gemm_t p) {
  gemm_f32_body<BM, BN, TA, TB, GAT, ACT, ROWEPI, NOPAD>(p, blockIdx.x, gridDim.x, blockIdx.z);
}

// Grouped launch: workgroups first[i] .. first[i + 1] - 1 work on problem i of a device-resident descriptor table
// (independent weight-gradient GEMMs dW_i += dY_i^T X_i of one backward pass, each too small to fill the chip).
template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(NT_, 2) void gemm_f32_group_kernel(const eamd_gemm_t* __restrict__ tab,
                                                                const int* __restrict__ first, const int n) {
  int lo = 0, hi = n;                    // largest i with first[i] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const eamd_gemm_t p = tab[lo];         // wave-uniform: scalar loads
  gemm_f32_body<BM, BN, TA, TB, false, false>(p, (int)blockIdx.x - first[lo], -(first[lo + 1] - first[lo]), 0);
}

// Several independent implicit-convolution products in ONE launch (eamd_gemm_multi): the stride-parity classes of a strided
// convolution's input gradient are four x W^T products of 4, 2, 2 and 1 taps over the same rows; launched one after another
// each ends in a partly filled last tile round (2366 tiles = 4.6 rounds of 512) and a launch gap.  The problems follow each
// other inside one grid (groups of 8 workgroups, one per XCD, so that id & 7 stays the XCD inside a problem): config 2,
// 2003 -> 1890 us.  Dealing the groups to the problems in turn instead (EAMD_GEMM_MULTI_ORDER=0: tiles of all reduction
// lengths resident together, stores of one under the main loop of another) was much SLOWER, 3130 us: the four weight
// images (2.4 MB) and four result streams then compete for each XCD's 4 MB of L2.
struct MultiF {
  eamd_gemm_t p[EAMD_GEMM_MULTI_MAX];
  int nt[EAMD_GEMM_MULTI_MAX];
  int n, order;
};

template <int BM, int BN, bool TA, bool TB, bool GAT, bool NOPAD>
__global__ __launch_bounds__(NT_, 2) void gemm_f32_multi_kernel(const MultiF m) {
  const int g = (int)blockIdx.x >> 3;
  int cls, bid;
  if (m.order == 0) {
    cls = g % m.n;
    bid = (g / m.n) * 8 + ((int)blockIdx.x & 7);
  } else {
    const int per = (int)gridDim.x / (8 * m.n);
    cls = g / per;
    bid = (g - cls * per) * 8 + ((int)blockIdx.x & 7);
  }
  if (bid >= m.nt[cls]) return;            // this problem has fewer tiles than the largest one of the launch
  const eamd_gemm_t p = m.p[cls];          // wave-uniform copy: scalar loads up front, none inside the tile
  gemm_f32_body<BM, BN, TA, TB, GAT, false, false, NOPAD>(p, bid, m.nt[cls], 0);
}

template <int BM, int BN, bool TA, bool TB, bool GAT, bool ACT, bool NOPAD = false>
int launch_f2(const eamd_gemm_t& p, hipStream_t stream) {
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk, 1, p.batch1 * p.batch2);
  constexpr size_t smem = sizeof(SmemF<BM, BN, TA, TB>);
  if (p.epilogue >= 7) {      // row epilogues (eamd_gemm_t.stats): instantiated for plain x W^T products only
    if constexpr (!TA && !TB && !GAT && !ACT) {
      if (smem > 64 * 1024) {
        static const hipError_t attr_err_r = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, TA, TB, GAT, ACT, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (attr_err_r != hipSuccess) return (int)attr_err_r;
      }
      hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, GAT, ACT, true>), grid, dim3(NT_), smem, stream, p);
      EAMD_LAUNCH_CHECK();
      return EAMD_OK;
    } else {
      return EAMD_EUNSUPPORTED;
    }
  }
  if (smem > 64 * 1024) {
    static const hipError_t attr_err = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, TA, TB, GAT, ACT, false, NOPAD>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (attr_err != hipSuccess) return (int)attr_err;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, GAT, ACT, false, NOPAD>), grid, dim3(NT_), smem, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

// implicit-conv gather whose taps never leave the source (no padding): the kernel skips the per-tile range checks
inline bool gather_nopad(const eamd_gather_t& g) {
  for (int tp = 0; tp < g.ntap; ++tp)
    if (g.dh[tp] < 0 || (g.Ho - 1) * g.sh + g.dh[tp] >= g.Hin || g.dw[tp] < 0 || (g.Wo - 1) * g.sw + g.dw[tp] >= g.Win) return false;
  return true;
}

// the gathered kernels keep the tap offsets as 4-bit fields of two scalars
inline bool gather_taps_small(const eamd_gather_t& g) {
  for (int tp = 0; tp < g.ntap; ++tp)
    if (g.dh[tp] < -8 || g.dh[tp] > 7 || g.dw[tp] < -8 || g.dw[tp] > 7) return false;
  return true;
}

template <int BM, int BN, bool TA, bool TB, bool GAT>
int launch_f(const eamd_gemm_t& p, hipStream_t stream) {
  if constexpr (GAT && !TA) {
    if (gather_nopad(p.gather) && p.epilogue < 7) return launch_f2<BM, BN, TA, TB, GAT, false, true>(p, stream);
  }
  if constexpr (GAT) return launch_f2<BM, BN, TA, TB, GAT, false>(p, stream);
  else if (p.a_act != EAMD_ACT_NONE || p.b_act != EAMD_ACT_NONE || p.a_drop_p > 0.f || p.b_drop_p > 0.f)
    return launch_f2<BM, BN, TA, TB, GAT, true>(p, stream);
  return launch_f2<BM, BN, TA, TB, GAT, false>(p, stream);
}

template <int T>
int dispatch_layout_f(const eamd_gemm_t& p, hipStream_t s) {
  if (p.transA) return p.transB ? launch_f<T, T, true, true, false>(p, s) : launch_f<T, T, true, false, false>(p, s);
  return p.transB ? launch_f<T, T, false, true, false>(p, s) : launch_f<T, T, false, false, false>(p, s);
}

bool aligned16f(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// Called by eamd_gemm (gemm.hip) for in_dtype = 0, precision = 0 after argument validation.
// Returns EAMD_EUNSUPPORTED when the operands do not meet the branch-free staging conditions (16-byte aligned chunk
// starts that stay inside the operand); the caller then takes the generic kernel.
int eamd_gemm_f32_dispatch(const eamd_gemm_t& p, int tile, hipStream_t stream) {
  static const int on = [] { const char* e = getenv("EAMD_GEMM_F32_FAST"); return e ? atoi(e) : 1; }();
  if (!on) return EAMD_EUNSUPPORTED;
  if (p.Cb || (p.Hb && !p.h_dtype) || p.aux_dtype || (!p.C && p.epilogue != 7)) return EAMD_EUNSUPPORTED;
  if (p.Hb && p.drop_p <= 0.f) return EAMD_EINVAL;
  if (p.drop_p < 0.f || p.drop_p >= 1.f) return EAMD_EINVAL;
  if (p.a_drop_p < 0.f || p.a_drop_p >= 1.f || p.b_drop_p < 0.f || p.b_drop_p >= 1.f) return EAMD_EINVAL;
  if ((p.a_drop_p > 0.f || p.b_drop_p > 0.f) && (p.gather.enabled || p.batch1 * p.batch2 != 1 || !p.drop_step)) return EAMD_EINVAL;
  if (p.a_drop_p > 0.f && (int64_t)(p.transA ? p.K : p.M) * p.lda >= (1LL << 32)) return EAMD_EUNSUPPORTED;   // 32-bit mask index
  if (p.b_drop_p > 0.f && (int64_t)(p.transB ? p.K : p.N) * p.ldb >= (1LL << 32)) return EAMD_EUNSUPPORTED;
  // fused result dropout: mask index = row * N + col of a contiguous [M, N] result, as eamd_dropout draws it
  if (p.drop_p > 0.f && (p.cmap.enabled || p.batch1 * p.batch2 != 1 || p.ldc != p.N || p.splitk > 1)) return EAMD_EINVAL;
  const bool a_ok = aligned16f(p.A) && p.lda % 4 == 0 && p.sA1 % 4 == 0 && p.sA2 % 4 == 0 &&
                    p.lda >= (p.transA ? (p.M + 3) / 4 * 4 : (p.K + 3) / 4 * 4);
  const bool b_ok = aligned16f(p.B) && p.ldb % 4 == 0 && p.sB1 % 4 == 0 && p.sB2 % 4 == 0 &&
                    p.ldb >= (p.transB ? (p.N + 3) / 4 * 4 : (p.K + 3) / 4 * 4);
  if (p.gather.enabled) {
    const eamd_gather_t& g = p.gather;
    if (g.C % FBK != 0 || !p.transB || !aligned16f(p.A) || !b_ok) return EAMD_EUNSUPPORTED;
    if (p.a_act != EAMD_ACT_NONE || p.b_act != EAMD_ACT_NONE) return EAMD_EUNSUPPORTED;
    if (!gather_taps_small(g)) return EAMD_EUNSUPPORTED;       // tap offsets travel as 4-bit fields
    if (p.transA && g.C % tile != 0) return EAMD_EUNSUPPORTED;
    if (tile == 128)
      return p.transA ? launch_f<128, 128, true, true, true>(p, stream) : launch_f<128, 128, false, true, true>(p, stream);
    return p.transA ? launch_f<64, 64, true, true, true>(p, stream) : launch_f<64, 64, false, true, true>(p, stream);
  }
  if (!a_ok || !b_ok) return EAMD_EUNSUPPORTED;
  return tile == 128 ? dispatch_layout_f<128>(p, stream) : dispatch_layout_f<64>(p, stream);
}

// ---- several implicit-convolution products in one launch (eamd_gemm_multi, fp32 operands) ----
// ps[i] have passed eamd_gemm's argument validation; tiles[i] is the tile eamd_gemm would pick.  EAMD_EUNSUPPORTED: the caller
// launches them one by one.
template <bool NOPAD>
static int multi_launch_t(const MultiF& m, int grid, hipStream_t stream) {
  constexpr size_t smem = sizeof(SmemF<128, 128, false, true>);
  static const hipError_t attr_err = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&gemm_f32_multi_kernel<128, 128, false, true, true, NOPAD>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_err != hipSuccess) return (int)attr_err;
  hipLaunchKernelGGL((gemm_f32_multi_kernel<128, 128, false, true, true, NOPAD>), dim3((unsigned)grid), dim3(NT_), smem, stream, m);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_gemm_f32_multi(const eamd_gemm_t* ps, const int* tiles, int n, hipStream_t stream) {
  static const int on = [] { const char* e = getenv("EAMD_GEMM_MULTI"); return e ? atoi(e) : 1; }();
  if (!on || n < 2 || n > EAMD_GEMM_MULTI_MAX) return EAMD_EUNSUPPORTED;
  MultiF m;
  bool nopad = true;
  int maxnt = 0;
  for (int i = 0; i < n; ++i) {
    const eamd_gemm_t& p = ps[i];
    if (tiles[i] != 128 || p.in_dtype != 0 || p.precision != 0) return EAMD_EUNSUPPORTED;
    if (!p.gather.enabled || p.transA || !p.transB || p.splitk != 1 || p.batch1 * p.batch2 != 1) return EAMD_EUNSUPPORTED;
    if (!p.C || p.Cb || p.Hb || p.aux_dtype || p.epilogue > 5 || p.colsum) return EAMD_EUNSUPPORTED;
    if (p.a_act != EAMD_ACT_NONE || p.b_act != EAMD_ACT_NONE || p.drop_p != 0.f || p.a_drop_p != 0.f || p.b_drop_p != 0.f)
      return EAMD_EUNSUPPORTED;
    const bool b_ok = aligned16f(p.B) && p.ldb % 4 == 0 && p.ldb >= (p.N + 3) / 4 * 4;
    if (p.gather.C % FBK != 0 || !aligned16f(p.A) || !b_ok || !gather_taps_small(p.gather)) return EAMD_EUNSUPPORTED;
    const long nt = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (nt > (1L << 24)) return EAMD_EUNSUPPORTED;
    nopad = nopad && gather_nopad(p.gather);
    m.p[i] = p;
    m.nt[i] = (int)nt;
    maxnt = nt > maxnt ? (int)nt : maxnt;
  }
  for (int i = n; i < EAMD_GEMM_MULTI_MAX; ++i) { m.p[i] = ps[0]; m.nt[i] = 0; }
  m.n = n;
  static const int order = [] { const char* e = getenv("EAMD_GEMM_MULTI_ORDER"); return e ? atoi(e) : 1; }();
  m.order = order;
  const int grid = (maxnt + 7) / 8 * 8 * n;
  return nopad ? multi_launch_t<true>(m, grid, stream) : multi_launch_t<false>(m, grid, stream);
}

// ---- grouped weight-gradient launch (fp32 operands) ----
// plan: validates problem i for the grouped kernel (C += alpha A^T B with split-K atomics or beta = 1: transA, transB, no
// epilogue / bias / residual / gather / row map / dropout, batch 1, operands meeting the staging conditions above) and
// returns its workgroup count, or EAMD_EUNSUPPORTED.
int eamd_gemm_f32_group_count(const eamd_gemm_t& p) {
  if (p.in_dtype != 0 || p.precision != 0 || !p.transA || !p.transB || !p.C || p.Cb || p.Hb || p.aux || p.R || p.bias)
    return EAMD_EUNSUPPORTED;
  if (p.gather.enabled || p.cmap.enabled || p.epilogue || p.a_act || p.b_act || p.drop_p > 0.f || p.a_drop_p > 0.f ||
      p.b_drop_p > 0.f || p.batch1 * p.batch2 != 1 || p.splitk < 1)
    return EAMD_EUNSUPPORTED;
  if (p.splitk == 1 && p.beta != 1.f) return EAMD_EUNSUPPORTED;          // accumulate into the gradient buffer
  const bool a_ok = aligned16f(p.A) && p.lda % 4 == 0 && p.lda >= (p.M + 3) / 4 * 4;
  const bool b_ok = aligned16f(p.B) && p.ldb % 4 == 0 && p.ldb >= (p.N + 3) / 4 * 4;
  if (!a_ok || !b_ok) return EAMD_EUNSUPPORTED;
  const int T = p.tile == 128 ? 128 : 64;
  const long n = ((long)((p.M + T - 1) / T) * ((p.N + T - 1) / T) * p.splitk + 7) / 8 * 8;   // whole rounds of the 8 XCDs
  return n < (1L << 24) ? (int)n : EAMD_EUNSUPPORTED;
}

template <int T>
static int group_launch_t(const eamd_gemm_t* tab_dev, const int* first_dev, int n, int total, hipStream_t stream) {
  constexpr size_t smem = sizeof(SmemF<T, T, true, true>);
  if (smem > 64 * 1024) {
    static const hipError_t attr_err = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_f32_group_kernel<T, T, true, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (attr_err != hipSuccess) return (int)attr_err;
  }
  hipLaunchKernelGGL((gemm_f32_group_kernel<T, T, true, true>), dim3((unsigned)total), dim3(NT_), smem, stream, tab_dev,
                     first_dev, n);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_gemm_f32_group_launch(const eamd_gemm_t* tab_dev, const int* first_dev, int n, int total, int tile, hipStream_t stream) {
  return tile == 128 ? group_launch_t<128>(tab_dev, first_dev, n, total, stream)
                     : group_launch_t<64>(tab_dev, first_dev, n, total, stream);
}
