// Persistent short-K variant of the bf16-operand GEMM (see the comment above the kernel).
#include <stdlib.h>
#include "gemm_bf16_common.h"

namespace {

// ---- persistent variant -----------------------------------------------------------------------------
// For launches with thousands of 64x64 tiles and a SHORT reduction (K = 256: FFN up-projection and its input
// gradient, fused QKV, vocabulary projections) a tile lives ~5.7 us, almost all of it HBM->LDS latency and the
// epilogue, while its 4 K-tiles of MFMA take a fraction of a microsecond.  Here a workgroup walks a list of
// tiles and keeps ONE register ring running across tile boundaries: while the last K-tiles of tile i are
// multiplied and its result is staged and stored, the first three K-tiles of tile i+1 are already in flight,
// so the load latency hides behind the epilogue instead of adding to it.  The result staging area is separate
// from the operand buffers (the next tile's first K-tile is already in LDS when the epilogue runs).
// Preconditions (host-checked): bf16 FAST operands, no gather / prologue activation / split-K / column sums /
// batch, K % 256 == 0 (whole K-tiles, a multiple of the ring depth per tile).
template <bool TA, bool TB>
struct SmemP {
  bf16_t a[2][64 * 64];
  bf16_t b[2][64 * 64];
  float c[64 * 68];
};

template <bool TA, bool TB>
__global__ __launch_bounds__(NT_) void gemm_bf16_persist_kernel(const eamd_gemm_t p, int ntiles) {
  constexpr int BM = 64, BN = 64, WM = 32, WN = 32, MT = 2, NTL = 2, NCA = 2, NCB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SmemP<TA, TB>& sm = *reinterpret_cast<SmemP<TA, TB>*>(smem_raw);
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(p.B);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nkt = p.K / BK;

  // tile list of this workgroup: XCD x (workgroup ids = x mod 8) owns one contiguous run of tiles, so that the
  // N-tiles sharing an A row panel meet in one L2; the workgroups of the XCD stride through the run
  const int G = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int nbx = (G - xcd + 7) >> 3;                   // workgroups living on this XCD
  const int Q = (ntiles + 7) >> 3;
  const int tile_end = min(ntiles, (xcd + 1) * Q);
  int tid = xcd * Q + jb;
  if (tid >= tile_end) return;

  int a_r[NCA], a_c[NCA], b_r[NCB], b_c[NCB];
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    if constexpr (TA) { a_r[i] = t / 8 + 32 * i; a_c[i] = t % 8; }       // [k][64 cols]: 8 chunks per k-row
    else              { a_r[i] = t / 8 + 32 * i; a_c[i] = t % 8; }       // [row][64 k]
  }
#pragma unroll
  for (int i = 0; i < NCB; ++i) { b_r[i] = t / 8 + 32 * i; b_c[i] = t % 8; }

  struct Offs { long a0, a1, b0, b1; int m0, n0; };      // plain scalars: arrays passed by reference end up in scratch
  auto offsets = [&](int tile) __attribute__((always_inline)) -> Offs {
    Offs o;
    o.m0 = (tile / tiles_n) * BM; o.n0 = (tile % tiles_n) * BN;
    if constexpr (!TA) {
      o.a0 = (long)min(o.m0 + a_r[0], p.M - 1) * p.lda + a_c[0] * 8;
      o.a1 = (long)min(o.m0 + a_r[1], p.M - 1) * p.lda + a_c[1] * 8;
    } else {
      o.a0 = (long)a_r[0] * p.lda + min(o.m0 + a_c[0] * 8, (p.M - 1) & ~7);
      o.a1 = (long)a_r[1] * p.lda + min(o.m0 + a_c[1] * 8, (p.M - 1) & ~7);
    }
    if constexpr (!TB) {
      o.b0 = (long)min(o.n0 + b_r[0], p.N - 1) * p.ldb + b_c[0] * 8;
      o.b1 = (long)min(o.n0 + b_r[1], p.N - 1) * p.ldb + b_c[1] * 8;
    } else {
      o.b0 = (long)b_r[0] * p.ldb + min(o.n0 + b_c[0] * 8, (p.N - 1) & ~7);
      o.b1 = (long)b_r[1] * p.ldb + min(o.n0 + b_c[1] * 8, (p.N - 1) & ~7);
    }
    return o;
  };

  uint4 ra[4][NCA], rb[4][NCB];      // the ring: K-tile kt lives in set kt % 4
  // (zero-initialised: with an uninitialised set on any path hipcc keeps the whole ring in scratch memory)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ra[q][0] = make_uint4(0u, 0u, 0u, 0u); ra[q][1] = ra[q][0]; rb[q][0] = ra[q][0]; rb[q][1] = ra[q][0];
  }
  auto load = [&](auto set_c, const Offs o, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const long ka = TA ? (long)kt * BK * p.lda : (long)kt * BK;
    const long kb = TB ? (long)kt * BK * p.ldb : (long)kt * BK;
    ra[SET][0] = *reinterpret_cast<const uint4*>(A + o.a0 + ka);
    ra[SET][1] = *reinterpret_cast<const uint4*>(A + o.a1 + ka);
    rb[SET][0] = *reinterpret_cast<const uint4*>(B + o.b0 + kb);
    rb[SET][1] = *reinterpret_cast<const uint4*>(B + o.b1 + kb);
  };
  auto store = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    *reinterpret_cast<uint4*>(&sm.a[buf][lds_chunk_off<TA, BM>(a_r[0], a_c[0])]) = ra[SET][0];
    *reinterpret_cast<uint4*>(&sm.a[buf][lds_chunk_off<TA, BM>(a_r[1], a_c[1])]) = ra[SET][1];
    *reinterpret_cast<uint4*>(&sm.b[buf][lds_chunk_off<TB, BN>(b_r[0], b_c[0])]) = rb[SET][0];
    *reinterpret_cast<uint4*>(&sm.b[buf][lds_chunk_off<TB, BN>(b_r[1], b_c[1])]) = rb[SET][1];
  };

  f32x4 acc[MT][NTL];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto mfma = [&](int buf) __attribute__((always_inline)) {
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
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;

  Offs cur = offsets(tid), nxt = cur;
  // K-tile kt of any tile sits in register set kt % 4 and LDS buffer kt % 2 (nkt is a multiple of 4)
  load(S0{}, cur, 0); load(S1{}, cur, 1); load(S2{}, cur, 2);
  store(S0{}, 0);
  __syncthreads();
  zero_acc();
  while (true) {
    const int next = tid + nbx;
    const bool has_next = next < tile_end;
    for (int kg = 0; kg + 4 < nkt; kg += 4) {         // whole groups strictly inside the tile
      load(S3{}, cur, kg + 3); mfma(0); store(S1{}, 1); __syncthreads();
      load(S0{}, cur, kg + 4); mfma(1); store(S2{}, 0); __syncthreads();
      load(S1{}, cur, kg + 5); mfma(0); store(S3{}, 1); __syncthreads();
      load(S2{}, cur, kg + 6); mfma(1); store(S0{}, 0); __syncthreads();
    }
    // last group of the tile: its loads reach into the next tile of the list (the very last tile of the list
    // re-reads its own first K-tiles instead: harmless, and it keeps this block free of divergent paths)
    nxt = offsets(has_next ? next : tid);
    load(S3{}, cur, nkt - 1); mfma(0); store(S1{}, 1); __syncthreads();
    load(S0{}, nxt, 0);       mfma(1); store(S2{}, 0); __syncthreads();
    load(S1{}, nxt, 1);       mfma(0); store(S3{}, 1); __syncthreads();
    load(S2{}, nxt, 2);       mfma(1); store(S0{}, 0); __syncthreads();
    store_c_tile<BM, BN, false>(p, acc, sm.c, cur.m0, cur.n0, 0L);
    if (!has_next) break;
    zero_acc();
    tid = next;
    cur = nxt;
  }
}

template <bool TA, bool TB>
int launch_persist(const eamd_gemm_t& p, int ntiles, hipStream_t stream) {
  static const hipError_t attr_err = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&gemm_bf16_persist_kernel<TA, TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
      (int)sizeof(SmemP<TA, TB>));
  if (attr_err != hipSuccess) return (int)attr_err;
  // one workgroup per resident slot: CUs x occupancy (registers / 49 KB of LDS decide; queried once)
  static const int grid = [] {
    int dev = 0, cus = 256, occ = 2;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&gemm_bf16_persist_kernel<TA, TB>),
                                                     NT_, sizeof(SmemP<TA, TB>)) != hipSuccess || occ < 1)
      occ = 2;
    return cus * occ;
  }();
  hipLaunchKernelGGL((gemm_bf16_persist_kernel<TA, TB>), dim3(grid), dim3(NT_), sizeof(SmemP<TA, TB>), stream, p, ntiles);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // namespace

// called by eamd_gemm_bf16_dispatch (gemm_bf16.hip) when the launch qualifies
int eamd_gemm_bf16_persist(const eamd_gemm_t& p, int ntiles, hipStream_t stream) {
  if (p.transA) return p.transB ? launch_persist<true, true>(p, ntiles, stream) : launch_persist<true, false>(p, ntiles, stream);
  return p.transB ? launch_persist<false, true>(p, ntiles, stream) : launch_persist<false, false>(p, ntiles, stream);
}
