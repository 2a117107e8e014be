// bf16-operand MFMA GEMM for gfx950 (fast path of eamd_gemm: in_dtype = 1).
//
// Operands live in HBM as bf16 (activations written in bf16 by their producers, weights as a bf16
// shadow of the fp32 master copy); products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 and
// written as fp32 and/or bf16.  Compared with the fp32-operand kernel in gemm.hip this halves the
// operand bytes, removes the conversion pass and doubles the K depth per barrier (BK = 64).
//
// Staging is global -> VGPR (16-byte chunks) -> LDS with the next tile's loads in flight during the
// current tile's 32 MFMAs per wave (register double buffering, two LDS buffers, one barrier/tile).
//   * k-contiguous operand  ([rows][K], nn.Linear activations / weights): LDS image [row][64 + 8],
//     144-byte rows => conflict-free ds_read_b128 fragment reads.
//   * k-strided operand     ([K][cols], transposed use in dX = dY W and dW = dY^T X): LDS image
//     [k][cols + 16] (288 / 160-byte rows) read with ds_read_b64_tr_b16, the CDNA4 transposing LDS read,
//     so no transposed copy of any activation or weight is ever materialised in HBM.
// 256 threads = 4 waves (2x2), block tile 128x128 or 64x64, wave tile 64x64 / 32x32 of 16x16x32 MFMAs.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"

#include "gemm_bf16_common.h"

namespace {

// ACT: prologue activations (a_act / b_act) compiled in; the hot instantiations leave them out so the
// steady-state loop carries no transcendental code and no branches around it.
// One workgroup's share of a problem: `bid` of `nblk` workgroups (tile x split-K slice), batch index `zb`.
// gemm_bf16_kernel runs it on a launch of its own; gemm_bf16_group_kernel looks the problem up in a device table.
template <int BM, int BN, bool TA, bool TB, bool FAST, bool GAT, bool ACT, bool ROWEPI = false>
__device__ __forceinline__ void gemm_bf16_body(const eamd_gemm_t& p, const int bid, const int nblk, const int zb) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 16, NTL = WN / 16;
  constexpr int NCA = BM / 32, NCB = BN / 32;          // 16-byte chunks per thread per tile
  constexpr int CPR_A = BM / 8, RPP_A = NT_ / CPR_A;   // transposed image: chunks per k-row, k-rows per pass
  constexpr int CPR_B = BN / 8, RPP_B = NT_ / CPR_B;
  using S = SmemB<BM, BN, TA, TB>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  S& sm = *reinterpret_cast<S*>(smem_raw);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an
  // L2), so give every XCD one contiguous run of row-major (m-tile, n-tile) pairs: the N-tiles that
  // re-read one A row panel then hit the same L2 instead of pulling the panel through the fabric 8x.
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
    // split-K: the split index is the fastest-varying part of the workgroup id, so XCD i (ids = i mod 8)
    // owns K-slice i (mod splitk) of EVERY tile: each XCD streams its slice of A and B once through its
    // own L2 instead of all eight XCDs re-reading the whole reduction range
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
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(p.B) + b1 * p.sB1 + b2 * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;

  const int nkt_total = (p.K + BK - 1) / BK;
  const int per = (nkt_total + p.splitk - 1) / p.splitk;
  const int kt_begin = split * per;
  const int kt_end = min(nkt_total, kt_begin + per);
  const int nkt = kt_end - kt_begin;

  constexpr bool gat = GAT;
  const bool a_vec = (p.lda % 8 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool b_vec = (p.ldb % 8 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

  // ---- staging coordinates ----------------------------------------------------------------------
  int a_r[NCA], a_c[NCA], b_r[NCB], b_c[NCB];   // LDS image (row, chunk) of every staged chunk
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
  // FAST staging reads through clamped coordinates: rows / column chunks past the M or N edge re-read
  // the last valid one.  Whatever they hold only reaches output rows / columns that the epilogue never
  // stores, so full K-tiles need no masking at all; only the ragged last K-tile is masked (to zero).
  long a_off[NCA], b_off[NCB];
  if constexpr (FAST && !gat) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      if constexpr (!TA) a_off[i] = (long)min(m0 + a_r[i], p.M - 1) * p.lda + a_c[i] * 8;
      else a_off[i] = (long)a_r[i] * p.lda + min(m0 + a_c[i] * 8, (p.M - 1) & ~7);
    }
  }
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      if constexpr (!TB) b_off[i] = (long)min(n0 + b_r[i], p.N - 1) * p.ldb + b_c[i] * 8;
      else b_off[i] = (long)b_r[i] * p.ldb + min(n0 + b_c[i] * 8, (p.N - 1) & ~7);
    }
  }
  RowStateB a_rs[NCA];
  if constexpr (gat && !TA) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) a_rs[i] = decompose_b(p.gather, m0 + a_r[i], p.M);
  }
  GatherRegs gr = {0ull, 0ull, 1, 0, 0};          // tap offsets in scalar registers (gemm_bf16_common.h)
  if constexpr (gat) gr = gather_regs(p.gather);
  // register ring of DEPTH tile sets: DEPTH-1 tiles of loads stay in flight across the MFMA phases
  // (skinny N=256 GEMMs have ~2 workgroups per CU, so bytes in flight per workgroup hide HBM latency)
  constexpr int DEPTH = BM >= 128 ? 3 : 4;
  constexpr int UNROLL = DEPTH % 2 ? 2 * DEPTH : DEPTH;   // phases per steady-state iteration (set and LDS buffer both static)
  uint4 ra[DEPTH][NCA], rb[DEPTH][NCB];
  const bool do_colsum = TA && p.colsum != nullptr && !gat && tile_n == 0;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;

  auto fill_poff = [&](int kt, int slot) __attribute__((always_inline)) {
    if (t < BK) {
      const int tap = m0 / gr.C;
      RowStateB s = decompose_b(p.gather, kt * BK + t, p.K);
      sm.poff[slot][t] = (int)gather_off_r(gr, s, tap);
    }
  };

  // GUARD = the tile may be the ragged last one (k0 + BK > K): out-of-range chunk starts are redirected
  // to the start of the row / to row 0 and zeroed at store time.
  auto load_tile = [&](auto set_c, auto guard_c, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    constexpr bool GUARD = decltype(guard_c)::value;
    const int k0 = kt * BK;
    // ---- A ----
    if constexpr (!TA) {
      if constexpr (!gat) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const int m = m0 + a_r[i], k = k0 + a_c[i] * 8;
          if constexpr (FAST) {
            const int kk = (GUARD && k >= p.K) ? -a_c[i] * 8 : k0;
            ra[SET][i] = *reinterpret_cast<const uint4*>(A + a_off[i] + kk);
          } else {
            const int nv = m < p.M ? min(8, p.K - k) : 0;
            ra[SET][i] = load8(A, (long)m * p.lda + k, nv, a_vec);
          }
        }
      } else {
        const int tap = k0 / gr.C, c0 = k0 - tap * gr.C;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const long off = gather_off_r(gr, a_rs[i], tap);
          ra[SET][i] = load8_fast(A, off + c0 + a_c[i] * 8, off >= 0 ? 8 : 0);
        }
      }
    } else {
      if constexpr (!gat) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const int k = k0 + a_r[i], m = m0 + a_c[i] * 8;
          if constexpr (FAST) {
            const int kk = (GUARD && k >= p.K) ? -a_r[i] : k0;
            ra[SET][i] = *reinterpret_cast<const uint4*>(A + a_off[i] + (long)kk * p.lda);
          } else {
            const int nv = k < p.K ? min(8, p.M - m) : 0;
            ra[SET][i] = load8(A, (long)k * p.lda + m, nv, a_vec);
          }
        }
      } else {
        const int c = (m0 % p.gather.C) + a_c[0] * 8;
        const int slot = kt % 8;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const int off = sm.poff[slot][a_r[i]];
          ra[SET][i] = load8_fast(A, (long)off + c, off >= 0 ? 8 : 0);
        }
      }
    }
    // ---- B ----
    if constexpr (!TB) {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const int n = n0 + b_r[i], k = k0 + b_c[i] * 8;
        if constexpr (FAST) {
          const int kk = (GUARD && k >= p.K) ? -b_c[i] * 8 : k0;
          rb[SET][i] = *reinterpret_cast<const uint4*>(B + b_off[i] + kk);
        } else {
          const int nv = n < p.N ? min(8, p.K - k) : 0;
          rb[SET][i] = load8(B, (long)n * p.ldb + k, nv, b_vec);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const int k = k0 + b_r[i], n = n0 + b_c[i] * 8;
        if constexpr (FAST) {
          const int kk = (GUARD && k >= p.K) ? -b_r[i] : k0;
          rb[SET][i] = *reinterpret_cast<const uint4*>(B + b_off[i] + (long)kk * p.ldb);
        } else {
          const int nv = k < p.K ? min(8, p.N - n) : 0;
          rb[SET][i] = load8(B, (long)k * p.ldb + n, nv, b_vec);
        }
      }
    }
  };

  // validity (number of in-range elements) of staged chunk i of tile kt, recomputed at store time
  auto nv_a = [&](int i, int kt) __attribute__((always_inline)) -> int {
    const int k0 = kt * BK;
    if constexpr (gat) {
      if constexpr (!TA) return gather_off_r(gr, a_rs[i], k0 / gr.C) >= 0 ? 8 : 0;
      else return sm.poff[kt % 8][a_r[i]] >= 0 ? 8 : 0;
    } else if constexpr (!TA) {
      return min(8, p.K - (k0 + a_c[i] * 8));       // K edge only (M edge: clamped rows, never stored)
    } else {
      return (k0 + a_r[i]) < p.K ? 8 : 0;
    }
  };
  auto nv_b = [&](int i, int kt) __attribute__((always_inline)) -> int {
    const int k0 = kt * BK;
    if constexpr (!TB) return min(8, p.K - (k0 + b_c[i] * 8));
    else return (k0 + b_r[i]) < p.K ? 8 : 0;
  };

  auto store_tile = [&](auto set_c, auto guard_c, int buf, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    constexpr bool GUARD = decltype(guard_c)::value;
    if constexpr (gat) {   // gathered rows are either fully valid or fully zero (padding taps / row tail)
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        const bool ok = nv_a(i, kt) > 0;
        ra[SET][i] = make_uint4(ok ? ra[SET][i].x : 0u, ok ? ra[SET][i].y : 0u, ok ? ra[SET][i].z : 0u,
                                ok ? ra[SET][i].w : 0u);
      }
    }
    if constexpr (FAST && GUARD) {
      if (kt * BK + BK > p.K) {      // ragged last K-tile: zero the out-of-range reduction elements
        if constexpr (!gat) {
#pragma unroll
          for (int i = 0; i < NCA; ++i) ra[SET][i] = mask8(ra[SET][i], nv_a(i, kt));
        }
#pragma unroll
        for (int i = 0; i < NCB; ++i) rb[SET][i] = mask8(rb[SET][i], nv_b(i, kt));
      }
    }
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        cs[0] += bf2f(ra[SET][i].x & 0xffff); cs[1] += bf2f(ra[SET][i].x >> 16);
        cs[2] += bf2f(ra[SET][i].y & 0xffff); cs[3] += bf2f(ra[SET][i].y >> 16);
        cs[4] += bf2f(ra[SET][i].z & 0xffff); cs[5] += bf2f(ra[SET][i].z >> 16);
        cs[6] += bf2f(ra[SET][i].w & 0xffff); cs[7] += bf2f(ra[SET][i].w >> 16);
      }
    }
    if constexpr (ACT) {
      if (p.a_act == EAMD_ACT_SWISH) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) ra[SET][i] = swish8(ra[SET][i]);
      } else if (p.a_act == EAMD_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < NCA; ++i) ra[SET][i] = relu8(ra[SET][i]);
      }
      if (p.b_act == EAMD_ACT_SWISH) {
#pragma unroll
        for (int i = 0; i < NCB; ++i) rb[SET][i] = swish8(rb[SET][i]);
      } else if (p.b_act == EAMD_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < NCB; ++i) rb[SET][i] = relu8(rb[SET][i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NCA; ++i)
      *reinterpret_cast<uint4*>(&sm.a[buf][lds_chunk_off<TA, BM>(a_r[i], a_c[i])]) = ra[SET][i];
#pragma unroll
    for (int i = 0; i < NCB; ++i)
      *reinterpret_cast<uint4*>(&sm.b[buf][lds_chunk_off<TB, BN>(b_r[i], b_c[i])]) = rb[SET][i];
  };

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr bool tgat = gat && TA;
  if (nkt > 0) {
    if (tgat) {
      for (int q = 0; q < DEPTH && q < nkt; ++q) fill_poff(kt_begin + q, (kt_begin + q) % 8);
      __syncthreads();
    }
    load_tile(std::integral_constant<int, 0>{}, std::true_type{}, kt_begin);
    if (nkt > 1) load_tile(std::integral_constant<int, 1>{}, std::true_type{}, kt_begin + 1);
    if constexpr (DEPTH == 4) {
      if (nkt > 2) load_tile(std::integral_constant<int, 2>{}, std::true_type{}, kt_begin + 2);
    }
    store_tile(std::integral_constant<int, 0>{}, std::true_type{}, 0, kt_begin);
    __syncthreads();
  }

  const int fr = lane & 15, fq = lane >> 4;
  // one K-tile: issue the loads of tile it+DEPTH-1 into the register set freed one phase ago, run the
  // 32 (or 8) MFMAs of tile `it` from LDS, then move tile it+1 (loaded DEPTH-2 phases ago) into the other
  // LDS buffer: DEPTH-2 whole tiles of loads stay in flight behind a counted s_waitcnt.
  auto phase = [&](auto idx_c, auto guard_c, int it) __attribute__((always_inline)) {
    constexpr int IDX = decltype(idx_c)::value;                       // it % UNROLL
    constexpr int PAR = IDX % DEPTH;
    constexpr bool GUARD = decltype(guard_c)::value;
    using load_t = std::integral_constant<int, (PAR + DEPTH - 1) % DEPTH>;   // set freed one phase ago
    using other_t = std::integral_constant<int, (PAR + 1) % DEPTH>;          // tile it+1
    constexpr int buf = IDX & 1;
    if constexpr (GUARD) {
      if (tgat && it + DEPTH < nkt) fill_poff(kt_begin + it + DEPTH, (kt_begin + it + DEPTH) % 8);
      if (it + DEPTH - 1 < nkt) load_tile(load_t{}, guard_c, kt_begin + it + DEPTH - 1);
    } else {
      if constexpr (tgat) fill_poff(kt_begin + it + DEPTH, (kt_begin + it + DEPTH) % 8);
      load_tile(load_t{}, guard_c, kt_begin + it + DEPTH - 1);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MT], bfr[NTL];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (!TA) {
          af[i] = *reinterpret_cast<const bf16x8*>(&sm.a[buf][lds_chunk_off<false, BM>(wm * WM + i * 16 + fr, ks * 4 + fq)]);
        } else {
          const int rk = ks * 32 + 8 * fq + (fr >> 2), cc = wm * WM + i * 16 + 4 * (fr & 3);
          const bf16_t* q0 = &sm.a[buf][lds_chunk_off<true, BM>(rk, cc >> 3) + (cc & 7)];
          const bf16_t* q1 = &sm.a[buf][lds_chunk_off<true, BM>(rk + 4, cc >> 3) + (cc & 7)];
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q0);
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q1);
          af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        if constexpr (!TB) {
          bfr[j] = *reinterpret_cast<const bf16x8*>(&sm.b[buf][lds_chunk_off<false, BN>(wn * WN + j * 16 + fr, ks * 4 + fq)]);
        } else {
          const int rk = ks * 32 + 8 * fq + (fr >> 2), cc = wn * WN + j * 16 + 4 * (fr & 3);
          const bf16_t* q0 = &sm.b[buf][lds_chunk_off<true, BN>(rk, cc >> 3) + (cc & 7)];
          const bf16_t* q1 = &sm.b[buf][lds_chunk_off<true, BN>(rk + 4, cc >> 3) + (cc & 7)];
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q0);
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)q1);
          bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (GUARD) {
      if (it + 1 < nkt) store_tile(other_t{}, guard_c, buf ^ 1, kt_begin + it + 1);
    } else {
      store_tile(other_t{}, guard_c, buf ^ 1, kt_begin + it + 1);
    }
    __syncthreads();
  };
  // steady state: groups of DEPTH phases with NO conditionals, so the compiler sees one straight-line
  // body and can emit counted s_waitcnt vmcnt(N) that leave the newer tiles' loads in flight; the last
  // (up to 2*DEPTH-2) tiles run through the guarded form.
  using T_ = std::true_type;
  using F_ = std::false_type;
  int it = 0;
  // unguarded group: its last phase loads tile it + UNROLL + DEPTH - 2, which must not be the (possibly
  // ragged) last tile
  for (; it + UNROLL + DEPTH - 1 < nkt; it += UNROLL) {
    phase(std::integral_constant<int, 0>{}, F_{}, it);
    phase(std::integral_constant<int, 1>{}, F_{}, it + 1);
    phase(std::integral_constant<int, 2>{}, F_{}, it + 2);
    phase(std::integral_constant<int, 3>{}, F_{}, it + 3);
    if constexpr (UNROLL == 6) {
      phase(std::integral_constant<int, 4>{}, F_{}, it + 4);
      phase(std::integral_constant<int, 5>{}, F_{}, it + 5);
    }
  }
  for (; it < nkt; it += UNROLL) {
    phase(std::integral_constant<int, 0>{}, T_{}, it);
    if (it + 1 < nkt) phase(std::integral_constant<int, 1>{}, T_{}, it + 1);
    if (it + 2 < nkt) phase(std::integral_constant<int, 2>{}, T_{}, it + 2);
    if (it + 3 < nkt) phase(std::integral_constant<int, 3>{}, T_{}, it + 3);
    if constexpr (UNROLL == 6) {
      if (it + 4 < nkt) phase(std::integral_constant<int, 4>{}, T_{}, it + 4);
      if (it + 5 < nkt) phase(std::integral_constant<int, 5>{}, T_{}, it + 5);
    }
  }

  if (TA && p.colsum != nullptr && !gat) {   // wave-uniform: every thread of the block takes the same path
    // bias gradient: combine the per-thread column sums in LDS (operand buffers are free now), then
    // one global atomic per column per block
    // (one LDS row per k-row group of the staging layout, plain stores: ds_add_f32 is slow on gfx950)
    float* csl = reinterpret_cast<float*>(smem_raw);
    constexpr int NSLOT = NT_ / CPR_A;
    static_assert(sizeof(float) * NSLOT * BM <= sizeof(S), "column-sum slots must fit in the operand buffers");
    __syncthreads();
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) csl[(t / CPR_A) * BM + a_c[0] * 8 + j] = cs[j];
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

  // ---- epilogue ------------------------------------------------------------------------------
  const bool lead = split == 0;
  bf16_t* __restrict__ Cb = reinterpret_cast<bf16_t*>(p.Cb);
  const bf16_t* __restrict__ auxb = reinterpret_cast<const bf16_t*>(p.aux);
  if (p.splitk > 1) {
    // split-K partial sums: one f32 atomic per element straight from the accumulators (16 consecutive
    // columns = 64-byte segments per row; measured faster than staging them through LDS for 256-byte rows)
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

  static_assert(sizeof(float) * BM * (BN + 4) <= sizeof(S), "C tile must fit in the operand buffers");
  store_c_tile<BM, BN, true, ROWEPI>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, coff);   // operand buffers are free now
}

template <int BM, int BN, bool TA, bool TB, bool FAST, bool GAT, bool ACT, bool ROWEPI = false>
__global__ __launch_bounds__(NT_, 2) void gemm_bf16_kernel(const eamd_gemm_t p) {
  gemm_bf16_body<BM, BN, TA, TB, FAST, GAT, ACT, ROWEPI>(p, blockIdx.x, gridDim.x, blockIdx.z);
}

// Grouped launch: workgroups first[i] .. first[i + 1] - 1 work on problem i of a device-resident descriptor table
// (independent weight-gradient GEMMs dW_i += dY_i^T X_i of one backward pass, each too small to fill the chip).
template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(NT_) void gemm_bf16_group_kernel(const eamd_gemm_t* __restrict__ tab,
                                                              const int* __restrict__ first, const int n) {
  int lo = 0, hi = n;                    // largest i with first[i] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const eamd_gemm_t p = tab[lo];         // wave-uniform: scalar loads
  gemm_bf16_body<BM, BN, TA, TB, true, false, false>(p, (int)blockIdx.x - first[lo], -(first[lo + 1] - first[lo]), 0);
}

template <int BM, int BN, bool TA, bool TB, bool FAST, bool GAT, bool ACT>
int launch_b2(const eamd_gemm_t& p, hipStream_t stream) {
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk, 1, p.batch1 * p.batch2);
  size_t smem = sizeof(SmemB<BM, BN, TA, TB>);
  if (p.epilogue >= 7) {      // row epilogues (eamd_gemm_t.stats): instantiated for plain x W^T products only
    if constexpr (!TA && !TB && FAST && !GAT && !ACT) {
      if (smem > 64 * 1024) {
        static const hipError_t attr_err_r = hipFuncSetAttribute(
            reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, TA, TB, FAST, GAT, ACT, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemB<BM, BN, TA, TB>));
        if (attr_err_r != hipSuccess) return (int)attr_err_r;
      }
      hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, TA, TB, FAST, GAT, ACT, true>), grid, dim3(NT_), smem, stream, p);
      EAMD_LAUNCH_CHECK();
      return EAMD_OK;
    } else {
      return EAMD_EUNSUPPORTED;
    }
  }
  if (smem > 64 * 1024) {
    static const hipError_t attr_err = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, TA, TB, FAST, GAT, ACT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemB<BM, BN, TA, TB>));
    if (attr_err != hipSuccess) return (int)attr_err;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, TA, TB, FAST, GAT, ACT>), grid, dim3(NT_), smem, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

template <int BM, int BN, bool TA, bool TB, bool FAST, bool GAT>
int launch_b(const eamd_gemm_t& p, hipStream_t stream) {
  if constexpr (GAT) return launch_b2<BM, BN, TA, TB, FAST, GAT, false>(p, stream);
  else if (p.a_act != EAMD_ACT_NONE || p.b_act != EAMD_ACT_NONE) return launch_b2<BM, BN, TA, TB, FAST, GAT, true>(p, stream);
  return launch_b2<BM, BN, TA, TB, FAST, GAT, false>(p, stream);
}

template <int T, bool FAST>
int dispatch_layout(const eamd_gemm_t& p, hipStream_t s) {
  if (p.transA) return p.transB ? launch_b<T, T, true, true, FAST, false>(p, s) : launch_b<T, T, true, false, FAST, false>(p, s);
  return p.transB ? launch_b<T, T, false, true, FAST, false>(p, s) : launch_b<T, T, false, false, FAST, false>(p, s);
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

int eamd_gemm_bf16_persist(const eamd_gemm_t& p, int ntiles, hipStream_t stream);   // gemm_persist.hip
int eamd_gemm_bf16_direct(const eamd_gemm_t& p, hipStream_t stream);                // gemm_direct.hip

// called by eamd_gemm (gemm.hip) after argument validation when in_dtype == 1
int eamd_gemm_bf16_dispatch(const eamd_gemm_t& p, int tile, hipStream_t stream) {
  if (p.splitk > 1 && (!p.C || p.Cb || p.drop_p > 0.f)) return EAMD_EINVAL;
  if (p.drop_p < 0.f || p.drop_p >= 1.f || (p.Hb && p.drop_p <= 0.f)) return EAMD_EINVAL;
  if (p.drop_p > 0.f && (p.cmap.enabled || p.batch1 * p.batch2 != 1 || p.ldc != p.N)) return EAMD_EINVAL;   // mask index = row*N+col
  // branch-free staging needs 16-byte aligned chunk starts that stay inside the operand
  const bool a_ok = aligned16(p.A) && p.lda % 8 == 0 && p.sA1 % 8 == 0 && p.sA2 % 8 == 0 &&
                    p.lda >= (p.transA ? (p.M + 7) / 8 * 8 : (p.K + 7) / 8 * 8);
  const bool b_ok = aligned16(p.B) && p.ldb % 8 == 0 && p.sB1 % 8 == 0 && p.sB2 % 8 == 0 &&
                    p.ldb >= (p.transB ? (p.N + 7) / 8 * 8 : (p.K + 7) / 8 * 8);
  if (p.gather.enabled) {
    // implicit-conv operands: A gathered ([rows][(tap, C)] view of an NHWC tensor), B = tap-major weights [K][N]
    const eamd_gather_t& g = p.gather;
    if (g.C % BK != 0 || !p.transB || !aligned16(p.A) || !b_ok) return EAMD_EINVAL;
    if (p.a_act != EAMD_ACT_NONE || p.b_act != EAMD_ACT_NONE) return EAMD_EINVAL;
    if (!gather_taps_fit(g)) return EAMD_EUNSUPPORTED;          // tap offsets travel as 4-bit fields
    if (p.transA && g.C % tile != 0) return EAMD_EINVAL;
    if (tile == 128)
      return p.transA ? launch_b<128, 128, true, true, true, true>(p, stream)
                      : launch_b<128, 128, false, true, true, true>(p, stream);
    return p.transA ? launch_b<64, 64, true, true, true, true>(p, stream)
                    : launch_b<64, 64, false, true, true, true>(p, stream);
  }
  if (a_ok && b_ok && tile == 64 && !p.transA && !p.transB && (p.K == 64 || p.K == 128) && p.splitk == 1 && p.C && !p.Cb &&
      !p.Hb && !p.bias && !p.aux && !p.R && !p.colsum && p.epilogue == 0 && p.beta == 0.f && p.drop_p <= 0.f &&
      p.a_act == EAMD_ACT_NONE && p.b_act == EAMD_ACT_NONE && !p.cmap.enabled) {
    static const int direct_on = [] { const char* e = getenv("EAMD_GEMM_DIRECT"); return e ? atoi(e) : 1; }();
    if (direct_on) return eamd_gemm_bf16_direct(p, stream);      // attention score products: one or two K-tiles
  }
  if (a_ok && b_ok && tile == 64) {
    static const int persist_min = [] { const char* e = getenv("EAMD_GEMM_PERSIST_MIN"); return e ? atoi(e) : 512; }();
    const long ntiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    if (persist_min > 0 && ntiles >= persist_min && ntiles < (1L << 30) && p.K % 256 == 0 && p.splitk == 1 &&
        p.batch1 * p.batch2 == 1 && !p.colsum && p.a_act == EAMD_ACT_NONE && p.b_act == EAMD_ACT_NONE && !p.cmap.enabled &&
        !p.transB && !p.aux && p.epilogue < 7) {   // measured: the k-strided-B (NN) launches do not gain, aux epilogues want the prefetch
      return eamd_gemm_bf16_persist(p, (int)ntiles, stream);
    }
  }
  if (a_ok && b_ok) return tile == 128 ? dispatch_layout<128, true>(p, stream) : dispatch_layout<64, true>(p, stream);
  return tile == 128 ? dispatch_layout<128, false>(p, stream) : dispatch_layout<64, false>(p, stream);
}

// ---- grouped weight-gradient launch (bf16 operands): see gemm_f32.hip ----
int eamd_gemm_bf16_group_count(const eamd_gemm_t& p) {
  if (p.in_dtype != 1 || !p.transA || !p.transB || !p.C || p.Cb || p.Hb || p.aux || p.R || p.bias) return EAMD_EUNSUPPORTED;
  if (p.gather.enabled || p.cmap.enabled || p.epilogue || p.a_act || p.b_act || p.drop_p > 0.f || p.a_drop_p > 0.f ||
      p.b_drop_p > 0.f || p.batch1 * p.batch2 != 1 || p.splitk < 1)
    return EAMD_EUNSUPPORTED;
  if (p.splitk == 1 && p.beta != 1.f) return EAMD_EUNSUPPORTED;          // accumulate into the gradient buffer
  const bool a_ok = aligned16(p.A) && p.lda % 8 == 0 && p.lda >= (p.M + 7) / 8 * 8;
  const bool b_ok = aligned16(p.B) && p.ldb % 8 == 0 && p.ldb >= (p.N + 7) / 8 * 8;
  if (!a_ok || !b_ok) return EAMD_EUNSUPPORTED;
  const int T = p.tile == 128 ? 128 : 64;
  const long n = ((long)((p.M + T - 1) / T) * ((p.N + T - 1) / T) * p.splitk + 7) / 8 * 8;   // whole rounds of the 8 XCDs
  return n < (1L << 24) ? (int)n : EAMD_EUNSUPPORTED;
}

template <int T>
static int group_launch_t(const eamd_gemm_t* tab_dev, const int* first_dev, int n, int total, hipStream_t stream) {
  constexpr size_t smem = sizeof(SmemB<T, T, true, true>);
  if (smem > 64 * 1024) {
    static const hipError_t attr_err = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_bf16_group_kernel<T, T, true, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (attr_err != hipSuccess) return (int)attr_err;
  }
  hipLaunchKernelGGL((gemm_bf16_group_kernel<T, T, true, true>), dim3((unsigned)total), dim3(NT_), smem, stream, tab_dev,
                     first_dev, n);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_gemm_bf16_group_launch(const eamd_gemm_t* tab_dev, const int* first_dev, int n, int total, int tile, hipStream_t stream) {
  return tile == 128 ? group_launch_t<128>(tab_dev, first_dev, n, total, stream)
                     : group_launch_t<64>(tab_dev, first_dev, n, total, stream);
}
