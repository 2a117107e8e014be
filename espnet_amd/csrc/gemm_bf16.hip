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
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

constexpr int BK = 64;
constexpr int NT_ = 256;
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// LDS images are unpadded and XOR-swizzled at 16-byte chunk granularity so that both the staging
// stores (ds_write_b128) and the fragment reads are bank-conflict free:
//   k-contiguous image [row][64]: chunk' = chunk ^ ((row >> 1) & 7)          (ds_read_b128, 16-lane groups)
//   k-strided image   [k][W]    : 32-byte slot' = slot ^ f(k)                (ds_read_b64_tr_b16, 32-lane halves)
//       W = 128: f(k) = (k & 3) | ((k >> 3) & 1) << 2 ;  W = 64: f(k) = ((k >> 1) & 1) | ((k >> 3) & 1) << 1
template <bool T, int W>
__device__ __forceinline__ int lds_chunk_off(int r, int c16) {
  if constexpr (!T) {
    return r * 64 + ((c16 ^ ((r >> 1) & 7)) << 3);
  } else if constexpr (W == 128) {
    const int f = (r & 3) | (((r >> 3) & 1) << 2);
    return r * 128 + (((((c16 >> 1) ^ f) << 1) | (c16 & 1)) << 3);
  } else {
    const int f = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    return r * 64 + (((((c16 >> 1) ^ f) << 1) | (c16 & 1)) << 3);
  }
}

// same counter-based generator as dropout_kernel (elementwise.hip)
__device__ __forceinline__ unsigned drop_hash(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x;
}

template <int BM, int BN, bool TA, bool TB>
struct SmemB {
  static constexpr int LDA = TA ? BM : BK;
  static constexpr int RA = TA ? BK : BM;
  static constexpr int LDB = TB ? BN : BK;
  static constexpr int RB = TB ? BK : BN;
  bf16_t a[2][RA * LDA];
  bf16_t b[2][RB * LDB];
  float cpad[(BM * (BN + 4) * 4 > 2 * (RA * LDA + RB * LDB) * 2) ? (BM * (BN + 4) - (RA * LDA + RB * LDB)) : 1];
  int poff[8][BK];
};

struct RowStateB { int base, ih, jw, ok; };

__device__ __forceinline__ RowStateB decompose_b(const eamd_gather_t& g, int row, int nrows) {
  RowStateB s;
  s.ok = row < nrows;
  int r = s.ok ? row : 0;
  int j = r % g.Wo; int t = r / g.Wo; int i = t % g.Ho; int b = t / g.Ho;
  s.base = b * g.Hin * g.Win; s.ih = i * g.sh; s.jw = j * g.sw;
  return s;
}
__device__ __forceinline__ long gather_off_b(const eamd_gather_t& g, const RowStateB& s, int tap) {
  int hh = s.ih + g.dh[tap], ww = s.jw + g.dw[tap];
  bool ok = s.ok && hh >= 0 && hh < g.Hin && ww >= 0 && ww < g.Win;
  return ok ? ((long)(s.base + hh * g.Win + ww)) * g.C : -1L;
}

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// 8 consecutive bf16 starting at p[idx]; `nvalid` of them are in range (0..8); vec = 16-byte path usable
__device__ __forceinline__ uint4 load8(const bf16_t* __restrict__ p, long idx, int nvalid, bool vec) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (nvalid >= 8 && vec) {
    v = *reinterpret_cast<const uint4*>(p + idx);
  } else if (nvalid > 0) {
    bf16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = j < nvalid ? p[idx + j] : (bf16_t)0;
    v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
    v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
  }
  return v;
}

// Branch-free form used when every chunk start is 16-byte aligned and lies inside the tensor
// (host-checked: ld % 8 == 0, ld >= extent): invalid chunks read element 0 and are masked to zero,
// partially valid chunks are masked per element.  No control flow => the compiler keeps all of a
// tile's loads in flight behind one counted s_waitcnt instead of draining after each guarded load.
__device__ __forceinline__ unsigned mask2(int nvalid, int d) {
  const int r = nvalid - 2 * d;
  return r >= 2 ? 0xffffffffu : (r == 1 ? 0x0000ffffu : 0u);
}
// The load itself must not be followed by any use of its result (the masking happens when the
// chunk is moved to LDS, one or more MFMA phases later), otherwise hipcc waits for it on the spot.
__device__ __forceinline__ uint4 load8_fast(const bf16_t* __restrict__ p, long idx, int nvalid) {
  return *reinterpret_cast<const uint4*>(p + (nvalid > 0 ? idx : 0L));
}
__device__ __forceinline__ uint4 mask8(uint4 v, int nvalid) {
  return make_uint4(v.x & mask2(nvalid, 0), v.y & mask2(nvalid, 1), v.z & mask2(nvalid, 2), v.w & mask2(nvalid, 3));
}

// prologue activations on packed bf16 pairs; the activation kind is tested ONCE per chunk group
// (a per-element runtime switch costs hundreds of scalar branches per tile and fences the stores)
__device__ __forceinline__ unsigned swish2(unsigned w) {
  float lo = eamd_swish(__uint_as_float(w << 16));
  float hi = eamd_swish(__uint_as_float(w & 0xffff0000u));
  return (unsigned)eamd_f2bf(lo) | ((unsigned)eamd_f2bf(hi) << 16);
}
__device__ __forceinline__ unsigned relu2(unsigned w) {
  return (w & 0x00008000u ? 0u : (w & 0x0000ffffu)) | (w & 0x80000000u ? 0u : (w & 0xffff0000u));
}
__device__ __forceinline__ uint4 swish8(uint4 v) { return make_uint4(swish2(v.x), swish2(v.y), swish2(v.z), swish2(v.w)); }
__device__ __forceinline__ uint4 relu8(uint4 v) { return make_uint4(relu2(v.x), relu2(v.y), relu2(v.z), relu2(v.w)); }

// ACT: prologue activations (a_act / b_act) compiled in; the hot instantiations leave them out so the
// steady-state loop carries no transcendental code and no branches around it.
template <int BM, int BN, bool TA, bool TB, bool FAST, bool GAT, bool ACT>
__global__ __launch_bounds__(NT_) void gemm_bf16_kernel(const eamd_gemm_t p) {
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
  if (p.splitk > 1) {
    // split-K: the split index is the fastest-varying part of the workgroup id, so XCD i (ids = i mod 8)
    // owns K-slice i (mod splitk) of EVERY tile: each XCD streams its slice of A and B once through its
    // own L2 instead of all eight XCDs re-reading the whole reduction range
    split = blockIdx.x % p.splitk;
    tile_id = blockIdx.x / p.splitk;
  } else {
    split = 0;
    const int ntile = gridDim.x;
    const int id = blockIdx.x, q = ntile >> 3, r = ntile & 7, xcd = id & 7, j = id >> 3;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tile_m = tile_id / tiles_n, tile_n = tile_id % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int zb = blockIdx.z;
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
      const int tap = m0 / p.gather.C;
      RowStateB s = decompose_b(p.gather, kt * BK + t, p.K);
      sm.poff[slot][t] = (int)gather_off_b(p.gather, s, tap);
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
        const int tap = k0 / p.gather.C, c0 = k0 % p.gather.C;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          const long off = gather_off_b(p.gather, a_rs[i], tap);
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
      if constexpr (!TA) return gather_off_b(p.gather, a_rs[i], k0 / p.gather.C) >= 0 ? 8 : 0;
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
    float* csl = reinterpret_cast<float*>(smem_raw);
    if (do_colsum) {
      for (int i = t; i < BM; i += NT_) csl[i] = 0.f;
    }
    __syncthreads();
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&csl[a_c[0] * 8 + j], cs[j]);
    }
    __syncthreads();
    if (do_colsum) {
      for (int i = t; i < BM; i += NT_) {
        const int m = m0 + i;
        if (m < p.M) atomicAdd(p.colsum + (long)zb * p.M + m, csl[i] * p.alpha);
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

  // Full results go through LDS (the operand buffers are free after the last barrier) so that every
  // lane stores 16 contiguous bytes of one output row: 1 KiB per wave-instruction instead of four
  // 64-byte segments, and the residual / aux operands are read the same way.
  constexpr int LDC = BN + 4;
  static_assert(sizeof(float) * BM * LDC <= sizeof(S), "C tile must fit in the operand buffers");
  float* cl = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cl[(wm * WM + i * 16 + fq * 4 + r) * LDC + wn * WN + j * 16 + fr] = acc[i][j][r];
  __syncthreads();

  constexpr int V4_PER_ROW = BN / 4;
  constexpr int ROWS_PER_PASS = NT_ / V4_PER_ROW;
  const int c4 = t % V4_PER_ROW;
  const int n = n0 + c4 * 4;
  const bool cvec = (p.ldc % 4 == 0) && (coff % 4 == 0) &&
                    (!p.C || (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                    (!Cb || (reinterpret_cast<uintptr_t>(Cb) & 7) == 0) &&
                    (!p.Hb || (reinterpret_cast<uintptr_t>(p.Hb) & 7) == 0) &&
                    (!p.R || ((p.ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(p.R) & 15) == 0)) &&
                    (!p.aux || (p.ldaux % 4 == 0 && (reinterpret_cast<uintptr_t>(p.aux) & 15) == 0));
  bf16_t* __restrict__ Hb = reinterpret_cast<bf16_t*>(p.Hb);
  const unsigned long long drop_base =
      p.drop_p > 0.f ? (p.drop_step ? p.drop_step[0] : 0ULL) * 0x9E3779B97F4A7C15ULL + p.drop_salt * 0xD1B54A32D192ED03ULL
                     : 0ULL;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (n + e < p.N) bv[e] = p.bias[n + e];
  }
  for (int rr = t / V4_PER_ROW; rr < BM; rr += ROWS_PER_PASS) {
    const int m = m0 + rr;
    if (m >= p.M || n >= p.N) continue;
    long prow = m;
    if (p.cmap.enabled) {
      const eamd_rowmap_t& c = p.cmap;
      int jj = m % c.Wo; int tt = m / c.Wo; int ii = tt % c.Ho; int bb = tt / c.Ho;
      prow = ((long)bb * c.Hc + ii * c.sh + c.oh) * c.Wc + jj * c.sw + c.ow;
    }
    const float4 a4 = *reinterpret_cast<const float4*>(&cl[rr * LDC + c4 * 4]);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const bool full = cvec && (n + 3 < p.N);
    const long ci = coff + prow * p.ldc + n;
    float ax[4] = {0.f, 0.f, 0.f, 0.f}, rv[4] = {0.f, 0.f, 0.f, 0.f}, cold[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.epilogue >= 3) {
      const long ai = coff + prow * p.ldaux + n;
      if (full) {
        if (p.aux_dtype) {
          uint2 u = *reinterpret_cast<const uint2*>(auxb + ai);
          ax[0] = bf2f(u.x & 0xffff); ax[1] = bf2f(u.x >> 16); ax[2] = bf2f(u.y & 0xffff); ax[3] = bf2f(u.y >> 16);
        } else {
          float4 u = *reinterpret_cast<const float4*>(p.aux + ai);
          ax[0] = u.x; ax[1] = u.y; ax[2] = u.z; ax[3] = u.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < p.N) ax[e] = p.aux_dtype ? bf2f(auxb[ai + e]) : p.aux[ai + e];
      }
    }
    if (p.R) {
      const long ri = coff + prow * p.ldr + n;
      if (full) { float4 u = *reinterpret_cast<const float4*>(p.R + ri); rv[0] = u.x; rv[1] = u.y; rv[2] = u.z; rv[3] = u.w; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < p.N) rv[e] = p.R[ri + e];
      }
    }
    if (p.C && p.beta != 0.f) {
      if (full) { float4 u = *reinterpret_cast<const float4*>(p.C + ci); cold[0] = u.x; cold[1] = u.y; cold[2] = u.z; cold[3] = u.w; }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < p.N) cold[e] = p.C[ci + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bv[e];
    if (p.epilogue == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    } else if (p.epilogue == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = eamd_swish(v[e]);
    } else if (p.epilogue == 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ax[e] > 0.f ? v[e] : 0.f;
    } else if (p.epilogue == 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= eamd_dswish(ax[e]);
    }
    if (p.drop_p > 0.f) {
      // wave-uniform branch; mask index = element index of the contiguous [M, N] result
      const float inv = 1.f / (1.f - p.drop_p);
      const unsigned thr = (unsigned)fminf(p.drop_p * 4294967296.0f, 4294967040.0f);
      const unsigned long long base = drop_base + (unsigned long long)ci;
      if (Hb) {
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = eamd_act(v[e], p.h_act);
          h[e] = drop_hash(base + e) >= thr ? a * inv : 0.f;
        }
        if (full) {
          uint2 o;
          o.x = eamd_f2bf(h[0]) | ((unsigned)eamd_f2bf(h[1]) << 16);
          o.y = eamd_f2bf(h[2]) | ((unsigned)eamd_f2bf(h[3]) << 16);
          *reinterpret_cast<uint2*>(Hb + ci) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n + e < p.N) Hb[ci + e] = eamd_f2bf(h[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = drop_hash(base + e) >= thr ? v[e] * inv : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * p.alpha + rv[e] + p.beta * cold[e];
    if (full) {
      if (p.C) *reinterpret_cast<float4*>(p.C + ci) = make_float4(v[0], v[1], v[2], v[3]);
      if (Cb) {
        uint2 o;
        o.x = eamd_f2bf(v[0]) | ((unsigned)eamd_f2bf(v[1]) << 16);
        o.y = eamd_f2bf(v[2]) | ((unsigned)eamd_f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(Cb + ci) = o;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < p.N) {
          if (p.C) p.C[ci + e] = v[e];
          if (Cb) Cb[ci + e] = eamd_f2bf(v[e]);
        }
      }
    }
  }
}

template <int BM, int BN, bool TA, bool TB, bool FAST, bool GAT, bool ACT>
int launch_b2(const eamd_gemm_t& p, hipStream_t stream) {
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk, 1, p.batch1 * p.batch2);
  size_t smem = sizeof(SmemB<BM, BN, TA, TB>);
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
    if (p.transA && g.C % tile != 0) return EAMD_EINVAL;
    if (tile == 128)
      return p.transA ? launch_b<128, 128, true, true, true, true>(p, stream)
                      : launch_b<128, 128, false, true, true, true>(p, stream);
    return p.transA ? launch_b<64, 64, true, true, true, true>(p, stream)
                    : launch_b<64, 64, false, true, true, true>(p, stream);
  }
  if (a_ok && b_ok) return tile == 128 ? dispatch_layout<128, true>(p, stream) : dispatch_layout<64, true>(p, stream);
  return tile == 128 ? dispatch_layout<128, false>(p, stream) : dispatch_layout<64, false>(p, stream);
}
