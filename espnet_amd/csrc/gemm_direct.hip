// Register-direct short-K variant of the bf16-operand GEMM (attention score products q k^T, (q+v) p^T, dO V^T).
#include "gemm_bf16_common.h"

namespace {

// C[z] = alpha * A[z] B[z]^T for k-contiguous A [M][K] and B [N][K] with K = 32 * KS <= 128 (d_k = 64 in every recipe).
// With one or two K-tiles there is nothing to pipeline: a 64x64 workgroup tile of the general kernel spends its
// life in global -> VGPR -> LDS -> barrier -> fragment reads for 8 MFMAs per wave.  The 16x16x32 MFMA operand
// layout of a k-contiguous matrix IS 16 bytes per lane of global memory (lane (r, q) holds row r, k = 8q .. 8q+7), so
// here every wave loads the fragments of its own 32x32 tile straight into registers, multiplies and stores from the
// accumulators: no LDS, no barrier, waves fully independent (<= 80 VGPRs: 6 waves per SIMD hide the one memory
// round trip).  The 2x2 waves of a workgroup cover a 64x64 tile so that the redundant fragment reads hit the CU's
// L1; the batches are dealt to the XCDs in contiguous runs so that a (batch, head) pair's panels live in ONE L2.
template <int KS>
__global__ __launch_bounds__(256) void gemm_bf16_direct_kernel(const eamd_gemm_t p, int tiles_m, int tiles_n, int nbatch) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  const int tpb = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int zb = (jb / tpb) * 8 + xcd;
  if (zb >= nbatch) return;
  const int tile = jb % tpb;
  const int m0 = (tile / tiles_n) * 64 + wm * 32, n0 = (tile % tiles_n) * 64 + wn * 32;
  if (m0 >= p.M || n0 >= p.N) return;          // wave-uniform; no barriers below
  const int b1 = zb / p.batch2, b2 = zb % p.batch2;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(p.B) + b1 * p.sB1 + b2 * p.sB2;
  float* __restrict__ Cp = p.C + b1 * p.sC1 + b2 * p.sC2;

  uint4 a[2][KS], b[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bf16_t* ar = A + (long)min(m0 + i * 16 + fr, p.M - 1) * p.lda + fq * 8;      // clamped rows are never stored
    const bf16_t* br = B + (long)min(n0 + i * 16 + fr, p.N - 1) * p.ldb + fq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a[i][ks] = *reinterpret_cast<const uint4*>(ar + ks * 32);
      b[i][ks] = *reinterpret_cast<const uint4*>(br + ks * 32);
    }
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][ks]),
                                                            __builtin_bit_cast(bf16x8, b[j][ks]), acc[i][j], 0, 0, 0);
  // accumulator (i, j)[r] = C[m0 + 16 i + 4 fq + r][n0 + 16 j + fr]: a store instruction covers 4 rows x 64 bytes
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + i * 16 + fq * 4 + r;
      if (row >= p.M) continue;
      float* crow = Cp + (long)row * p.ldc;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 16 + fr;
        if (col < p.N) crow[col] = acc[i][j][r] * p.alpha;
      }
    }
}

}  // namespace

// called by eamd_gemm_bf16_dispatch (gemm_bf16.hip) when the launch qualifies: bf16 k-contiguous operands with
// 16-byte aligned rows, K = 64 or 128, fp32 result only, plain epilogue (alpha), no split-K / gather / row map
int eamd_gemm_bf16_direct(const eamd_gemm_t& p, hipStream_t stream) {
  const int tiles_m = (p.M + 63) / 64, tiles_n = (p.N + 63) / 64, nbatch = p.batch1 * p.batch2;
  const long nblk = (long)tiles_m * tiles_n * ((nbatch + 7) / 8 * 8);
  if (nblk >= (1L << 31)) return EAMD_EUNSUPPORTED;
  if (p.K == 64)
    hipLaunchKernelGGL(gemm_bf16_direct_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, stream, p, tiles_m, tiles_n, nbatch);
  else if (p.K == 128)
    hipLaunchKernelGGL(gemm_bf16_direct_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, stream, p, tiles_m, tiles_n, nbatch);
  else
    return EAMD_EUNSUPPORTED;
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
