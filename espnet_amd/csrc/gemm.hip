// MFMA GEMM family for gfx950 (MI355X).  See include/espnet_amd.h for the contract.
//
// One kernel template covers every dense contraction of the ASR hot path:
//   * nn.Linear forward (NT), input-gradient (NN) and weight-gradient (TN, split-K + f32 atomics);
//   * batched attention products QK^T / PV and their gradients (two-level strided batch);
//   * Conv2dSubsampling as implicit GEMM (gathered A rows, mapped C rows), no im2col buffer.
// Operands live in HBM as fp32 (reference dtype).  They are staged global -> VGPR -> LDS with the
// next tile's loads issued before the current tile's MFMAs (register double buffering + two LDS
// buffers, one barrier per K-tile).  precision=1 rounds operands to bf16 in the staging pass and
// uses v_mfma_f32_16x16x32_bf16; precision=0 keeps fp32 and uses v_mfma_f32_16x16x4_f32.
// 256 threads = 4 waves (64 lanes) in a 2x2 arrangement; block tile 128x128 or 64x64, BK = 32.
#include <stdlib.h>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

template <int BM, int BN, int PREC>
struct Smem {
  // bf16: rows of 32 + 8 pad halfwords (80 B, 16-B aligned); fp32: rows of 32 + 4 pad floats (144 B)
  static constexpr int LDK = PREC ? 40 : 36;
  using elem_t = typename std::conditional<PREC == 1, unsigned short, float>::type;
  elem_t a[2][BM][LDK];
  elem_t b[2][BN][LDK];
  int poff[3][BK];  // gathered-row offsets for the transposed gather loader
};

struct RowState {  // per-thread decomposition of a gathered logical row
  int base;        // b * Hin * Win
  int ih, jw;      // i*sh, j*sw
  int ok;          // row < nrows
};

__device__ __forceinline__ RowState decompose(const eamd_gather_t& g, int row, int nrows) {
  RowState s;
  s.ok = row < nrows;
  int r = s.ok ? row : 0;
  int j = r % g.Wo;
  int t = r / g.Wo;
  int i = t % g.Ho;
  int b = t / g.Ho;
  s.base = b * g.Hin * g.Win;
  s.ih = i * g.sh;
  s.jw = j * g.sw;
  return s;
}

// element offset (in floats, before adding the channel) of tap `tap` for a decomposed row, or -1
__device__ __forceinline__ long gather_off(const eamd_gather_t& g, const RowState& s, int tap) {
  int hh = s.ih + g.dh[tap];
  int ww = s.jw + g.dw[tap];
  bool ok = s.ok && hh >= 0 && hh < g.Hin && ww >= 0 && ww < g.Win;
  return ok ? ((long)(s.base + hh * g.Win + ww)) * g.C : -1L;
}

__device__ __forceinline__ float4 apply_act4(float4 v, int act) {
  if (act == EAMD_ACT_SWISH) {
    v.x = eamd_swish(v.x); v.y = eamd_swish(v.y); v.z = eamd_swish(v.z); v.w = eamd_swish(v.w);
  } else if (act == EAMD_ACT_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  return v;
}

// Loads the 4 k-consecutive elements (row, k..k+3) of a k-contiguous operand.
__device__ __forceinline__ float4 load_kcontig(const float* __restrict__ rowp, bool row_ok, int k, int K,
                                               bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row_ok) {
    if (vec_ok && k + 3 < K) {
      v = *reinterpret_cast<const float4*>(rowp + k);
    } else {
      if (k + 0 < K) v.x = rowp[k + 0];
      if (k + 1 < K) v.y = rowp[k + 1];
      if (k + 2 < K) v.z = rowp[k + 2];
      if (k + 3 < K) v.w = rowp[k + 3];
    }
  }
  return v;
}

// Loads (k..k+3, col) of an operand stored [K][cols] (col contiguous): 4 strided scalar loads.
__device__ __forceinline__ float4 load_kstrided(const float* __restrict__ p, long ld, bool col_ok, int k,
                                                int K) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) {
    const float* q = p + (long)k * ld;
    if (k + 3 < K) {
      v.x = q[0]; v.y = q[ld]; v.z = q[2 * ld]; v.w = q[3 * ld];
    } else {
      if (k + 0 < K) v.x = q[0];
      if (k + 1 < K) v.y = q[ld];
      if (k + 2 < K) v.z = q[2 * ld];
      if (k + 3 < K) v.w = q[3 * ld];
    }
  }
  return v;
}

template <int PREC, typename T>
__device__ __forceinline__ void lds_store4(T* dst, float4 v) {
  if constexpr (PREC == 1) {
    ushort4 h;
    h.x = eamd_f2bf(v.x); h.y = eamd_f2bf(v.y); h.z = eamd_f2bf(v.z); h.w = eamd_f2bf(v.w);
    *reinterpret_cast<ushort4*>(dst) = h;
  } else {
    *reinterpret_cast<float4*>(dst) = v;
  }
}

template <int BM, int BN, int PREC>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const eamd_gemm_t p) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int NIA = BM / 32, NIB = BN / 32;  // float4 groups per thread per tile
  using S = Smem<BM, BN, PREC>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  S& sm = *reinterpret_cast<S*>(smem_raw);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int zb = blockIdx.z / p.splitk, split = blockIdx.z % p.splitk;
  const int b1 = zb / p.batch2, b2 = zb % p.batch2;

  const float* __restrict__ A = p.A + b1 * p.sA1 + b2 * p.sA2;
  const float* __restrict__ B = p.B + b1 * p.sB1 + b2 * p.sB2;
  const long coff = b1 * p.sC1 + b2 * p.sC2;

  const int nkt_total = (p.K + BK - 1) / BK;
  const int per = (nkt_total + p.splitk - 1) / p.splitk;
  const int kt_begin = split * per;
  const int kt_end = min(nkt_total, kt_begin + per);
  const int nkt = kt_end - kt_begin;

  const bool gat = p.gather.enabled != 0;
  const bool a_vec = (p.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool b_vec = (p.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);

  // ---- per-thread staging coordinates -------------------------------------------------------
  // k-contiguous operand: row = t/8 + 32*i, kq = (t%8)*4
  // k-strided operand   : col = t % BM, kq = (t/BM + i*(256/BM))*4
  int a_row[NIA], a_kq[NIA];
  int b_row[NIB], b_kq[NIB];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    if (p.transA) { a_row[i] = t % BM; a_kq[i] = (t / BM + i * (NTHREADS / BM)) * 4; }
    else          { a_row[i] = t / 8 + 32 * i; a_kq[i] = (t % 8) * 4; }
  }
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    if (p.transB) { b_row[i] = t % BN; b_kq[i] = (t / BN + i * (NTHREADS / BN)) * 4; }
    else          { b_row[i] = t / 8 + 32 * i; b_kq[i] = (t % 8) * 4; }
  }
  RowState a_rs[NIA];
  if (gat && !p.transA) {
#pragma unroll
    for (int i = 0; i < NIA; ++i) a_rs[i] = decompose(p.gather, m0 + a_row[i], p.M);
  }

  float4 ra[NIA], rb[NIB];
  // bias gradient fused into the weight-gradient GEMM: column sums of the transposed A operand
  const bool do_colsum = p.colsum != nullptr && p.transA && !gat && blockIdx.y == 0;
  float cs_acc = 0.f;

  // transposed gather: the 32 reduction rows of a K-tile are decomposed once per block into LDS
  auto fill_poff = [&](int kt, int slot) {
    if (t < BK) {
      const int tap = m0 / p.gather.C;
      RowState s = decompose(p.gather, kt * BK + t, p.K);
      long off = gather_off(p.gather, s, tap);
      sm.poff[slot][t] = (int)off;  // source tensors on this path are < 2^31 elements
    }
  };

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    // ---- A ----
    if (!p.transA) {
      if (!gat) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          const int m = m0 + a_row[i];
          ra[i] = load_kcontig(A + (long)m * p.lda, m < p.M, k0 + a_kq[i], p.K, a_vec);
        }
      } else {
        const int tap = k0 / p.gather.C, c0 = k0 % p.gather.C;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          long off = gather_off(p.gather, a_rs[i], tap);
          ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (off >= 0) ra[i] = *reinterpret_cast<const float4*>(A + off + c0 + a_kq[i]);
        }
      }
    } else {
      if (!gat) {
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          const int m = m0 + a_row[i];
          ra[i] = load_kstrided(A + m, p.lda, m < p.M, k0 + a_kq[i], p.K);
          if (do_colsum) cs_acc += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
        }
      } else {
        const int c = (m0 % p.gather.C) + a_row[0];
        const int slot = kt % 3;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int off = sm.poff[slot][a_kq[i] + e];
            v[e] = off >= 0 ? A[(long)off + c] : 0.f;
          }
          ra[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    // ---- B ----
    if (!p.transB) {
#pragma unroll
      for (int i = 0; i < NIB; ++i) {
        const int n = n0 + b_row[i];
        rb[i] = load_kcontig(B + (long)n * p.ldb, n < p.N, k0 + b_kq[i], p.K, b_vec);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NIB; ++i) {
        const int n = n0 + b_row[i];
        rb[i] = load_kstrided(B + n, p.ldb, n < p.N, k0 + b_kq[i], p.K);
      }
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NIA; ++i)
      lds_store4<PREC>(&sm.a[buf][a_row[i]][a_kq[i]], apply_act4(ra[i], p.a_act));
#pragma unroll
    for (int i = 0; i < NIB; ++i)
      lds_store4<PREC>(&sm.b[buf][b_row[i]][b_kq[i]], apply_act4(rb[i], p.b_act));
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool tgat = gat && p.transA;
  if (nkt > 0) {
    if (tgat) {
      fill_poff(kt_begin, kt_begin % 3);
      if (nkt > 1) fill_poff(kt_begin + 1, (kt_begin + 1) % 3);
      __syncthreads();
    }
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
  }

  const int fr = lane & 15, fq = lane >> 4;
  for (int it = 0; it < nkt; ++it) {
    const int buf = it & 1;
    const bool more = it + 1 < nkt;
    if (more) {
      if (tgat && it + 2 < nkt) fill_poff(kt_begin + it + 2, (kt_begin + it + 2) % 3);
      load_tile(kt_begin + it + 1);
    }

    if constexpr (PREC == 1) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&sm.a[buf][wm * WM + i * 16 + fr][fq * 8]);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&sm.b[buf][wn * WN + j * 16 + fr][fq * 8]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f32x4 af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&sm.a[buf][wm * WM + i * 16 + fr][kk * 16 + fq * 4]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bfr[j] = *reinterpret_cast<const f32x4*>(&sm.b[buf][wn * WN + j * 16 + fr][kk * 16 + fq * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
      }
    }

    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (do_colsum) {
    const int m = m0 + a_row[0];
    if (m < p.M) atomicAdd(p.colsum + (long)zb * p.M + m, cs_acc * p.alpha);
  }

  // ---- epilogue ------------------------------------------------------------------------------
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
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WN + j * 16 + fr;
        if (n >= p.N) continue;
        float v = acc[i][j][r];
        if (p.splitk > 1) {
          if (lead && p.bias) v += p.bias[n];
          v *= p.alpha;
          if (lead && p.R) v += p.R[coff + prow * p.ldr + n];
          atomicAdd(p.C + coff + prow * p.ldc + n, v);
        } else {
          if (p.bias) v += p.bias[n];
          if (p.epilogue == 1) v = v > 0.f ? v : 0.f;
          else if (p.epilogue == 2) v = eamd_swish(v);
          else if (p.epilogue == 3) v = p.aux[coff + prow * p.ldaux + n] > 0.f ? v : 0.f;
          else if (p.epilogue == 4) v *= eamd_dswish(p.aux[coff + prow * p.ldaux + n]);
          else if (p.epilogue == 5) v *= p.aux[coff + prow * p.ldaux + n];
          v *= p.alpha;
          if (p.R) v += p.R[coff + prow * p.ldr + n];
          float* cp = p.C + coff + prow * p.ldc + n;
          if (p.beta != 0.f) v += p.beta * (*cp);
          *cp = v;
        }
      }
    }
  }
}

template <int BM, int BN, int PREC>
int launch(const eamd_gemm_t& p, hipStream_t stream) {
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.batch1 * p.batch2 * p.splitk);
  size_t smem = sizeof(Smem<BM, BN, PREC>);
  if (smem > 64 * 1024) {
    // opt in to > 64 KiB of dynamic LDS (idempotent, per kernel instantiation)
    static const hipError_t attr_err = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_kernel<BM, BN, PREC>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)sizeof(Smem<BM, BN, PREC>));
    if (attr_err != hipSuccess) return (int)attr_err;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, PREC>), grid, dim3(NTHREADS), smem, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // namespace

int eamd_gemm_bf16_dispatch(const eamd_gemm_t& p, int tile, hipStream_t stream);  // gemm_bf16.hip
int eamd_gemm_f32_dispatch(const eamd_gemm_t& p, int tile, hipStream_t stream);   // gemm_f32.hip

// argument validation and tile choice shared by eamd_gemm and eamd_gemm_multi (normalises p.splitk)
static int gemm_check(eamd_gemm_t& p, int& tile_out) {
  const bool stats = p.epilogue == 7;          // row statistics instead of a result (eamd_gemm_t.stats)
  const bool rowgrad = p.epilogue == 8;        // softmax-gradient rows from per-row coefficients
  if (!p.A || !p.B || (!p.C && !p.Cb && !stats)) return EAMD_EINVAL;
  if (stats && (!p.stats.part || !p.stats.zfix || (p.stats.col && !p.stats.zcol))) return EAMD_EINVAL;
  if (rowgrad && !p.stats.rowc) return EAMD_EINVAL;
  if ((stats || rowgrad) && (p.stats.fix < 0 || p.stats.fix >= p.N || (p.tile != 64 && p.tile != 128) || p.splitk > 1 ||
                             p.batch1 * p.batch2 != 1 || p.cmap.enabled || p.gather.enabled || p.Hb || p.drop_p != 0.f ||
                             p.R || p.colsum || p.beta != 0.f))
    return EAMD_EINVAL;
  if (p.in_dtype != 0 && p.in_dtype != 1) return EAMD_EINVAL;
  if (p.in_dtype == 1 && p.precision != 1) return EAMD_EINVAL;
  if (p.in_dtype == 0 && (p.Cb || p.aux_dtype || (!p.C && !stats))) return EAMD_EUNSUPPORTED;
  if (p.M <= 0 || p.N <= 0 || p.K < 0) return EAMD_EINVAL;
  if (p.batch1 <= 0 || p.batch2 <= 0) return EAMD_EINVAL;
  if (p.splitk < 1) p.splitk = 1;
  if (p.splitk > 1 && p.epilogue != 0) return EAMD_EINVAL;
  if (p.epilogue < 0 || p.epilogue > 8) return EAMD_EINVAL;
  if (p.epilogue >= 3 && p.epilogue <= 5 && !p.aux) return EAMD_EINVAL;
  if (p.epilogue == 6 && (!p.Hb || p.drop_p <= 0.f)) return EAMD_EINVAL;      // only with the dual-output dropout epilogue
  if (p.precision != 0 && p.precision != 1) return EAMD_EINVAL;
  if ((long)p.batch1 * p.batch2 * p.splitk > 65535) return EAMD_EUNSUPPORTED;

  int tile = p.tile;
  if (tile == 0) {
    long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch1 * p.batch2 * p.splitk;
    if (p.in_dtype == 1) {
      // measured on MI355X (tools/gemm_breakdown.py bf16, tools/gemm_kslope.py): the 64x64 tile (and its persistent
      // form) wins at K = 256; from K = 320 on the 128x128 tile, two workgroups per CU since its register cap, is ahead
      // once there are two full rounds of tiles - the implicit-conv GEMMs of the subsampling front end
      // (151392 x 256 x 2304: 461 -> 335 us, 160000 x 256 x 1024: 276 -> 220 us) and the transducer's logits products
      // (37774 x 5000 x 320: config 5 46.9 -> 44.4 ms)
      static const int kmin = [] { const char* e = getenv("EAMD_BF16_T128_KMIN"); return e ? atoi(e) : 320; }();
      static const int nmin = [] { const char* e = getenv("EAMD_BF16_T128_NMIN"); return e ? atoi(e) : 256; }();
      tile = (t128 >= 1024 && p.K >= kmin && p.M >= 2048 && p.N >= nmin) ? 128 : 64;
    } else {
      // measured on MI355X (tools/gemm_f32_probe.py): from ~1 tile per CU on the 128x128 tile wins (4x fewer LDS
      // stores and barriers per MFMA), below that the 64x64 tile's 4x as many workgroups do
      // (the generic kernel, which still serves fp32 operands with bf16 MFMA, keeps its old threshold)
      tile = (t128 >= (p.precision == 0 ? 240 : 512) && p.M >= 128 && p.N >= 128) ? 128 : 64;
      // ... except between one and one and a half tiles per CU: the second half-filled round of 128x128 tiles costs more than
      // the 64x64 tile's loop (QKV projection 7968 x 768 x 256 = 378 tiles: 46.5 us, 41.5 us with 64x64; 252 and 416 tiles
      // (7968 x 512, 3232 x 2048) are as good or better with 128x128)
      if (p.precision == 0 && tile == 128 && t128 > 256 && t128 < 400 && p.splitk == 1) tile = 64;
    }
  }
  if (tile != 64 && tile != 128) return EAMD_EINVAL;

  if (p.gather.enabled) {
    const eamd_gather_t& g = p.gather;
    if (g.ntap < 1 || g.ntap > 9 || g.C <= 0 || g.C % BK != 0) return EAMD_EINVAL;
    if (!p.transA) {
      if (p.K != g.ntap * g.C) return EAMD_EINVAL;       // columns are (tap, channel)
    } else {
      if (p.M != g.ntap * g.C) return EAMD_EINVAL;
      if (g.C % tile != 0) return EAMD_EINVAL;            // an M-tile must stay inside one tap
    }
  }
  if ((p.N + tile - 1) / tile > 65535) return EAMD_EUNSUPPORTED;
  // fused dropout: the bf16-operand kernels (incl. the dual bf16 output Hb) and, for a plain dropped result, the
  // pipelined fp32 kernel (same epilogue code)
  if (p.h_dtype != 0 && p.h_dtype != 1) return EAMD_EINVAL;
  if (p.in_dtype == 1 && p.h_dtype != 0) return EAMD_EUNSUPPORTED;            // bf16 operands: bf16 second output
  if (p.in_dtype != 1 && ((p.Hb && !(p.h_dtype == 1 && p.precision == 0)) || (p.drop_p != 0.f && p.precision != 0)))
    return EAMD_EUNSUPPORTED;
  if ((p.a_drop_p != 0.f || p.b_drop_p != 0.f) && (p.in_dtype != 0 || p.precision != 0)) return EAMD_EUNSUPPORTED;
  tile_out = tile;
  return EAMD_OK;
}

extern "C" int eamd_gemm(const eamd_gemm_t* pp, void* stream_) {
  if (!pp) return EAMD_EINVAL;
  eamd_gemm_t p = *pp;
  hipStream_t stream = (hipStream_t)stream_;
  int tile = 0;
  const int chk = gemm_check(p, tile);
  if (chk != EAMD_OK) return chk;
  const bool stats = p.epilogue == 7, rowgrad = p.epilogue == 8;
  if (p.in_dtype == 1) return eamd_gemm_bf16_dispatch(p, tile, stream);

  // fp32 operands asked to go through the bf16 matrix cores (precision 1, bf16 mode's leftovers: activations a producer
  // keeps in fp32): the generic kernel converts while staging behind guarded loads (30-65 TFLOP/s on skinny shapes);
  // the implicit-conv weight gradients among them (EAMD_P1_F32=0: off) go (gather + transA: tiny output, K in the millions,
  // bound by the operand stream) through the pipelined fp32 kernel instead - exact fp32 arithmetic, i.e. no less accurate
  // (config 4, bf16 mode: 92.5 -> 91.1 ms)
  static const int p1_f32 = [] { const char* e = getenv("EAMD_P1_F32"); return e ? atoi(e) : 1; }();
  if (p.precision == 1 && p1_f32 && p.gather.enabled && p.transA && !p.Hb && p.drop_p == 0.f && p.a_drop_p == 0.f &&
      p.b_drop_p == 0.f) {
    const int rc = eamd_gemm_f32_dispatch(p, tile, stream);
    if (rc != EAMD_EUNSUPPORTED) return rc;
  }
  if (p.precision == 0) {     // reference precision: the pipelined fp32-MFMA kernel wherever its staging conditions hold
    const int rc = eamd_gemm_f32_dispatch(p, tile, stream);
    if (rc != EAMD_EUNSUPPORTED) return rc;
    if (p.drop_p != 0.f || p.a_drop_p != 0.f || p.b_drop_p != 0.f || p.Hb) return EAMD_EUNSUPPORTED;   // generic kernel: no dropout
  }
  if (stats || rowgrad) return EAMD_EUNSUPPORTED;       // the generic kernel has neither row epilogue
  if (tile == 128) {
    return p.precision ? launch<128, 128, 1>(p, stream) : launch<128, 128, 0>(p, stream);
  }
  return p.precision ? launch<64, 64, 1>(p, stream) : launch<64, 64, 0>(p, stream);
}

int eamd_gemm_f32_multi(const eamd_gemm_t* ps, const int* tiles, int n, hipStream_t stream);   // gemm_f32.hip

extern "C" int eamd_gemm_multi(const eamd_gemm_t* descs, int n, void* stream) {
  if (!descs || n < 1) return EAMD_EINVAL;
  if (n >= 2 && n <= EAMD_GEMM_MULTI_MAX) {
    eamd_gemm_t ps[EAMD_GEMM_MULTI_MAX];
    int tiles[EAMD_GEMM_MULTI_MAX];
    bool f32 = true;
    for (int i = 0; i < n; ++i) {
      ps[i] = descs[i];
      const int chk = gemm_check(ps[i], tiles[i]);
      if (chk != EAMD_OK) return chk;              // nothing has been launched yet
      f32 = f32 && ps[i].in_dtype == 0 && ps[i].precision == 0;
    }
    if (f32) {
      const int rc = eamd_gemm_f32_multi(ps, tiles, n, (hipStream_t)stream);
      if (rc != EAMD_EUNSUPPORTED) return rc;
    }
  }
  for (int i = 0; i < n; ++i) {
    const int rc = eamd_gemm(&descs[i], stream);
    if (rc != EAMD_OK) return rc;
  }
  return EAMD_OK;
}

int eamd_gemm_f32_group_count(const eamd_gemm_t& p);                                            // gemm_f32.hip
int eamd_gemm_f32_group_launch(const eamd_gemm_t* tab, const int* first, int n, int total, int tile, hipStream_t s);
int eamd_gemm_bf16_group_count(const eamd_gemm_t& p);                                           // gemm_bf16.hip
int eamd_gemm_bf16_group_launch(const eamd_gemm_t* tab, const int* first, int n, int total, int tile, hipStream_t s);

extern "C" int eamd_gemm_group_plan(const eamd_gemm_t* descs, int n, int32_t* first) {
  if (!descs || !first || n <= 0) return EAMD_EINVAL;
  long total = 0;
  for (int i = 0; i < n; ++i) {
    const eamd_gemm_t& p = descs[i];
    if (!p.A || !p.B || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0) return EAMD_EINVAL;
    if (p.in_dtype != descs[0].in_dtype || p.precision != descs[0].precision) return EAMD_EINVAL;
    if ((p.tile == 128) != (descs[0].tile == 128)) return EAMD_EINVAL;          // one tile size per launch
    const int c = p.in_dtype == 1 ? eamd_gemm_bf16_group_count(p) : eamd_gemm_f32_group_count(p);
    if (c < 0) return c;
    first[i] = (int32_t)total;
    total += c;
    if (total >= (1L << 30)) return EAMD_EUNSUPPORTED;
  }
  first[n] = (int32_t)total;
  return (int)total;
}

extern "C" int eamd_gemm_group_launch(const eamd_gemm_t* descs_dev, const int32_t* first_dev, int n, int total, int in_dtype,
                                      int tile, void* stream) {
  if (!descs_dev || !first_dev || n <= 0 || total <= 0 || (tile != 64 && tile != 128)) return EAMD_EINVAL;
  return in_dtype == 1 ? eamd_gemm_bf16_group_launch(descs_dev, first_dev, n, total, tile, (hipStream_t)stream)
                       : eamd_gemm_f32_group_launch(descs_dev, first_dev, n, total, tile, (hipStream_t)stream);
}

extern "C" int eamd_abi_version(void) { return 1; }
