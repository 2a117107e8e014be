// Conformer convolution-module kernels on channels-last [B, T, C] activations (C contiguous, so a
// wave's 64 lanes read 256 contiguous bytes): depthwise Conv1d fwd/bwd, BatchNorm1d (training batch
// statistics over all B*T positions, padded frames included - reference behaviour) fused with the
// following activation, and the first Conv2dSubsampling convolution (C_in = 1) as a direct kernel.
// reference: conformer/convolution.py:13-79, transformer/subsampling.py:28-33.
#include <stdlib.h>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

constexpr int KMAX = 32;

// y[b,t,c] = bias[c] + sum_k w[c,k] * x[b, t + k - pad, c]         (flip=0, forward)
// y[b,t,c] =           sum_k w[c,k] * x[b, t - k + pad, c]         (flip=1, input gradient)
// One thread = one channel x TT consecutive frames: the K taps stay in registers and the input
// window (TT + K - 1 values) is read once, lanes along C (coalesced 256-B rows).
// The block's 256 x K taps are fetched with coalesced loads and handed out through LDS (a direct
// w[c*K + k] read is a 64-way scattered access per tap).
template <int TT>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     int B, int T, int C, int K, int pad, int flip) {
  __shared__ float wl[256 * KMAX];
  const int c0 = blockIdx.x * blockDim.x;
  const int c_raw = c0 + threadIdx.x;
  const int c = min(c_raw, C - 1);          // dead lanes shadow a valid channel (no divergent exit before the barrier)
  const int nchunk = (T + TT - 1) / TT;
  const int b = blockIdx.y / nchunk;
  const int t0 = (blockIdx.y % nchunk) * TT;
  // the whole input window first, branch-free (clamped rows, zeroed afterwards): TT + K - 1 independent loads in
  // flight together instead of one round trip per tap; window position j covers frame t0 - pad + j
  // the block's taps likewise: KMAX independent coalesced loads per thread (a runtime-bounded copy loop waits for
  // every load before issuing the next: K serialized round trips)
  const int nw = min(256, C - c0) * K;
  float tw[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; ++q) tw[q] = w[(long)c0 * K + min((int)threadIdx.x + q * 256, nw - 1)];
  const float* xb = x + (long)b * T * C + c;
  float xv[TT + KMAX - 1];
#pragma unroll
  for (int j = 0; j < TT + KMAX - 1; ++j) {
    const int ts = min(max(t0 - pad + j, 0), T - 1);
    xv[j] = xb[(long)ts * C];
  }
#pragma unroll
  for (int q = 0; q < KMAX; ++q)
    if ((int)threadIdx.x + q * 256 < nw) wl[threadIdx.x + q * 256] = tw[q];
  __syncthreads();
  if (c_raw >= C) return;
  float wr[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    int kk = flip ? K - 1 - k : k;          // flipped taps turn the gradient into the same correlation
    wr[k] = (k < K) ? wl[threadIdx.x * K + min(max(kk, 0), K - 1)] : 0.f;
  }
  const float bv = (bias && !flip) ? bias[c] : 0.f;
  float acc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) acc[i] = bv;
#pragma unroll
  for (int j = 0; j < TT + KMAX - 1; ++j) {
    const int ts = t0 - pad + j;
    const float v = (ts >= 0 && ts < T && j < TT + K - 1) ? xv[j] : 0.f;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const int k = j - i;                 // tap index feeding output t0 + i
      if (k >= 0 && k < KMAX) acc[i] += wr[k] * v;
    }
  }
  float* yb = y + (long)b * T * C + c;
#pragma unroll
  for (int i = 0; i < TT; ++i)
    if (t0 + i < T) yb[(long)(t0 + i) * C] = acc[i];
}

__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b);

// LDS-tiled form of the same correlation (the default): a block owns 64 channels x NWV*8 frames.  The NWV*8 + K - 1
// input rows are fetched ONCE per block (coalesced 256-byte rows, all loads of a thread in flight together) into LDS,
// as are the 64 x K taps; wave w then produces frames [8w, 8w+8) from LDS.  L2 -> CU traffic per output drops from
// (8 + K - 1) / 8 input rows + a 256 x K tap block per 8 frames to (NWV*8 + K - 1) / (NWV*8) rows + a 64 x K tap block
// per NWV*8 frames (K = 31, NWV = 8: 8.8x -> 1.6x the output bytes).
// MODE 1 (forward of the Conformer convolution module): the input is GLU(a) of the pointwise-conv output a [B, T, 2C]
// (value columns 0..C-1, gate columns C..2C-1), formed while the rows are loaded - the [B, T, C] GLU result is never
// written.  MODE 2 (its input gradient, flip = 1): the result dgl is turned into da = GLU'(a) . dgl in the store -
// da[.., c] = dgl * sigmoid(g), da[.., C + c] = dgl * v * sigmoid(g) (1 - sigmoid(g)) - fp32 or bf16 [B, T, 2C].
// reference: conformer/convolution.py:53-79 (glu after pointwise_conv1, depthwise_conv), torch.nn.functional.glu.
template <int NWV, int MODE>
__global__ __launch_bounds__(NWV * 64) void dwconv_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int B, int T, int C, int K, int pad, int flip,
                                                              const float* __restrict__ aglu, int y_bf16,
                                                              float* __restrict__ bn_part) {
  // bn_part (MODE 1): the block also leaves the BatchNorm partial statistics (count, mean, M2 per channel) of its
  // 64-frame x 64-channel output tile in slab blockIdx.y of the workspace eamd_bn_finalize merges - bn_partial's job
  // without re-reading the output
  constexpr int TB = NWV * 8;                       // frames per block
  constexpr int NR = (TB + KMAX - 1 + NWV - 1) / NWV;   // input rows per wave (upper bound)
  constexpr int NWL = (64 * KMAX + NWV * 64 - 1) / (NWV * 64);
  __shared__ float xs[(TB + KMAX - 1) * 64];
  __shared__ float wl[64 * KMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64;
  const int c = min(c0 + lane, C - 1);
  const int nblk = (T + TB - 1) / TB;
  const int b = blockIdx.y / nblk;
  const int tb0 = (blockIdx.y % nblk) * TB;
  const int rows = TB + K - 1;
  const int nw = min(64, C - c0) * K;
  const float* xb = x + (long)b * T * (MODE == 1 ? 2 * C : C) + c;
  float xr[NR], tw[NWL];
  if constexpr (MODE == 1) {
    float gr[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int ts = min(max(tb0 - pad + wave + i * NWV, 0), T - 1);
      xr[i] = xb[(long)ts * 2 * C];
      gr[i] = xb[(long)ts * 2 * C + C];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) xr[i] *= eamd_sigmoid(gr[i]);
  } else {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int ts = min(max(tb0 - pad + wave + i * NWV, 0), T - 1);
      xr[i] = xb[(long)ts * C];
    }
  }
#pragma unroll
  for (int q = 0; q < NWL; ++q) tw[q] = w[(long)c0 * K + min((int)threadIdx.x + q * NWV * 64, nw - 1)];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = wave + i * NWV, ts = tb0 - pad + r;
    if (r < rows) xs[r * 64 + lane] = (ts >= 0 && ts < T) ? xr[i] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NWL; ++q)
    if ((int)threadIdx.x + q * NWV * 64 < nw) wl[threadIdx.x + q * NWV * 64] = tw[q];
  __syncthreads();
  const int t0 = tb0 + wave * 8;
  if ((MODE != 1 || !bn_part) && (c0 + lane >= C || t0 >= T)) return;
  float wr[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int kk = flip ? K - 1 - k : k;
    wr[k] = (k < K) ? wl[lane * K + min(max(kk, 0), K - 1)] : 0.f;
  }
  const float bv = (bias && !flip) ? bias[c] : 0.f;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = bv;
#pragma unroll
  for (int j = 0; j < 8 + KMAX - 1; ++j) {
    const float v = (j < 8 + K - 1) ? xs[(wave * 8 + min(j, 8 + K - 2)) * 64 + lane] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = j - i;
      if (k >= 0 && k < KMAX) acc[i] += wr[k] * v;
    }
  }
  if constexpr (MODE == 2) {
    const float* ab = aglu + (long)b * T * 2 * C + c;
    float av[8], gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = (long)min(t0 + i, T - 1) * 2 * C;
      av[i] = ab[r]; gv[i] = ab[r + C];
    }
    unsigned short* y16 = reinterpret_cast<unsigned short*>(y) + (long)b * T * 2 * C + c;
    float* y32 = y + (long)b * T * 2 * C + c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (t0 + i >= T) continue;
      const float sg = eamd_sigmoid(gv[i]);
      const float g1 = acc[i] * sg, g2 = acc[i] * av[i] * sg * (1.f - sg);
      const long r = (long)(t0 + i) * 2 * C;
      if (y_bf16) { y16[r] = eamd_f2bf(g1); y16[r + C] = eamd_f2bf(g2); }
      else { y32[r] = g1; y32[r + C] = g2; }
    }
    return;
  }
  float* yb = y + (long)b * T * C + c;
  const bool clive = c0 + lane < C;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (clive && t0 + i < T) yb[(long)(t0 + i) * C] = acc[i];
  if constexpr (MODE == 1) {
    if (bn_part) {
      __shared__ float bsh[NWV][3][64];
      float n = 0.f, sum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) if (t0 + i < T) { n += 1.f; sum += acc[i]; }
      float mean = n > 0.f ? sum / n : 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) if (t0 + i < T) { const float d = acc[i] - mean; m2 += d * d; }
      bsh[wave][0][lane] = n; bsh[wave][1][lane] = mean; bsh[wave][2][lane] = m2;
      __syncthreads();
      if (wave == 0 && clive) {            // fixed merge order: bitwise reproducible
        for (int w2 = 1; w2 < NWV; ++w2) chan_merge(n, mean, m2, bsh[w2][0][lane], bsh[w2][1][lane], bsh[w2][2][lane]);
        float* pp = bn_part + (long)blockIdx.y * 3 * C + c0 + lane;
        pp[0] = n; pp[C] = mean; pp[2 * C] = m2;
      }
    }
  }
}

// dw[c,k] += sum_{b,t} dy[b,t,c] * x[b,t+k-pad,c];  db[c] += sum dy.
// Block = NWV waves on 64 channels, walking `tiles_per_block` tiles of NWV*8 frames: per tile the NWV*8 + K - 1 input
// rows and the NWV*8 gradient rows go through LDS once (coalesced, all of a thread's loads in flight together), wave
// w combines frames [8w, 8w+8) into K register accumulators with fully unrolled FMAs.  The waves' accumulators are
// summed through per-wave LDS slots and the block adds its [64, K] result to global memory with lanes along the
// contiguous (c, k) index (uncoalesced f32 atomics run ~17x slower on gfx950).
constexpr int WNWV = 8;
__global__ __launch_bounds__(WNWV * 64) void dwconv_bwd_w_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ x, float* __restrict__ dw,
                                                                 float* __restrict__ db, int B, int T, int C, int K,
                                                                 int pad, int tiles_per_block, int glu) {
  // glu: x is the pointwise-conv output a [B, T, 2C] and the convolution input is GLU(a), formed on load
  constexpr int NWV = WNWV, TB = NWV * 8;
  constexpr int NR = (TB + KMAX - 1 + NWV - 1) / NWV;
  // one buffer, two lives: the (x, dy) tile while accumulating, then one [64][K+1] slot per wave for the block sum
  // (LDS float atomics measured ~40 us here; plain stores + a strided read cost nothing)
  __shared__ float smem[NWV * 64 * (KMAX + 1)];
  static_assert((TB + KMAX - 1) * 64 + TB * 64 <= NWV * 64 * (KMAX + 1), "tile must fit in the reduction buffer");
  float* xs = smem;
  float* gs = smem + (TB + KMAX - 1) * 64;
  __shared__ float redb[NWV * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64;
  const bool live = c0 + lane < C;
  const int c = live ? c0 + lane : C - 1;    // dead lanes shadow a valid channel and contribute nothing
  const int nblk = (T + TB - 1) / TB;
  const long total = (long)B * nblk;
  const int rows = TB + K - 1;
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  float accb = 0.f;
  for (int r = 0; r < tiles_per_block; ++r) {
    const long q = (long)blockIdx.y * tiles_per_block + r;
    if (q >= total) break;
    const int b = q / nblk;
    const int tb0 = (int)(q % nblk) * TB;
    const int xld = glu ? 2 * C : C;
    const float* xb = x + (long)b * T * xld + c;
    const float* gb = dy + (long)b * T * C + c;
    float xr[NR], gr[8];
#pragma unroll
    for (int i = 0; i < NR; ++i) xr[i] = xb[(long)min(max(tb0 - pad + wave + i * NWV, 0), T - 1) * xld];
    if (glu) {                               // wave-uniform
      float gg[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) gg[i] = xb[(long)min(max(tb0 - pad + wave + i * NWV, 0), T - 1) * xld + C];
#pragma unroll
      for (int i = 0; i < NR; ++i) xr[i] *= eamd_sigmoid(gg[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gr[i] = gb[(long)min(tb0 + wave + i * NWV, T - 1) * C];
    __syncthreads();                       // previous tile fully consumed (also orders the zero-fill of red)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int rr = wave + i * NWV, ts = tb0 - pad + rr;
      if (rr < rows) xs[rr * 64 + lane] = (ts >= 0 && ts < T) ? xr[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = wave + i * NWV;
      gs[rr * 64 + lane] = (tb0 + rr < T) ? gr[i] : 0.f;
    }
    __syncthreads();
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[i] = gs[(wave * 8 + i) * 64 + lane]; accb += g[i]; }
#pragma unroll
    for (int j = 0; j < 8 + KMAX - 1; ++j) {
      const float xv = (j < 8 + K - 1) ? xs[(wave * 8 + min(j, 8 + K - 2)) * 64 + lane] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = j - i;
        if (k >= 0 && k < KMAX) acc[k] += g[i] * xv;
      }
    }
  }
  __syncthreads();
  float* slot = smem + wave * 64 * (KMAX + 1);
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) slot[lane * (KMAX + 1) + k] = live ? acc[k] : 0.f;
  redb[wave * 64 + lane] = live ? accb : 0.f;
  __syncthreads();
  const int nch = min(64, C - c0);
  for (int i = threadIdx.x; i < nch * K; i += blockDim.x) {
    const int cc = i / K, k = i % K;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NWV; ++wv) t += smem[wv * 64 * (KMAX + 1) + cc * (KMAX + 1) + k];
    atomicAdd(&dw[(long)c0 * K + i], t);
  }
  if (db && threadIdx.x < nch) {
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NWV; ++wv) t += redb[wv * 64 + threadIdx.x];
    atomicAdd(&db[c0 + threadIdx.x], t);
  }
}

// ---- BatchNorm1d over [M, C] -----------------------------------------------------------------
// Statistics are reduced in two launches sized for latency, not for a serial sweep: a slab is
// BN_R * (256 / (C/4)) rows; every thread pulls its BN_R rows of one float4 channel group into
// registers with all loads in flight, forms (n, mean, M2) there (exact two-pass, no re-read), and the
// row-subgroups of the block are merged with Chan's update through LDS.  The finalize block merges the
// slab partials the same way (fixed order => bitwise reproducible).
constexpr int BN_R = 4;
struct BnGeom { int cq, rs, slab_rows; };
__host__ __device__ inline BnGeom bn_geom(int C) {
  BnGeom g;
  g.cq = C / 4;                          // float4 channel groups (C % 4 == 0, C <= 1024)
  g.rs = 256 / g.cq;                     // row-subgroups per 256-thread block
  g.slab_rows = BN_R * g.rs;
  return g;
}
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  const float nt = n + nb;
  if (nt > 0.f) {
    const float d = mb - mean;
    const float f = nb / nt;
    mean += d * f;
    m2 += m2b + d * d * n * f;
  }
  n = nt;
}

// (T, bound): optional row bound - rows whose time index (row % T) is not below bound[0] take no part in the statistics (a
// batch padded beyond its own longest utterance by a shape-bucketed graph: conformer/convolution.py:56-79 never sees those frames)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                         long M, int C, int T, const int* __restrict__ bound) {
  __shared__ float4 sh_mean[256], sh_m2[256];
  __shared__ float sh_n[256];
  const BnGeom g = bn_geom(C);
  const int t = threadIdx.x;
  const int q = t % g.cq, sub = t / g.cq;
  const bool live = sub < g.rs;
  const long r0 = (long)blockIdx.x * g.slab_rows + sub;
  float4 v[BN_R];
  float n = 0.f;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int tb = bound ? bound[0] : T;
  unsigned okm = 0u;
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const long r = r0 + (long)i * g.rs;
    const bool ok = live && r < M && (!bound || (int)(r % T) < tb);
    okm |= ok ? (1u << i) : 0u;
    v[i] = ok ? reinterpret_cast<const float4*>(x + r * C)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    n += ok ? 1.f : 0.f;
    sum.x += v[i].x; sum.y += v[i].y; sum.z += v[i].z; sum.w += v[i].w;
  }
  const float inv = n > 0.f ? 1.f / n : 0.f;
  float4 mean = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
  float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const bool ok = (okm >> i) & 1u;
    if (ok) {
      float a = v[i].x - mean.x, b = v[i].y - mean.y, c = v[i].z - mean.z, d = v[i].w - mean.w;
      m2.x += a * a; m2.y += b * b; m2.z += c * c; m2.w += d * d;
    }
  }
  sh_n[t] = n; sh_mean[t] = mean; sh_m2[t] = m2;
  __syncthreads();
  if (sub == 0) {
    for (int s2 = 1; s2 < g.rs; ++s2) {
      const int o = s2 * g.cq + q;
      const float nb = sh_n[o];
      const float4 mb = sh_mean[o], qb = sh_m2[o];
      float n0 = n, n1 = n, n2 = n, n3 = n;
      chan_merge(n0, mean.x, m2.x, nb, mb.x, qb.x);
      chan_merge(n1, mean.y, m2.y, nb, mb.y, qb.y);
      chan_merge(n2, mean.z, m2.z, nb, mb.z, qb.z);
      chan_merge(n3, mean.w, m2.w, nb, mb.w, qb.w);
      n = n0;
    }
    float* p = part + (long)blockIdx.x * 3 * C;
    reinterpret_cast<float4*>(p)[q] = make_float4(n, n, n, n);
    reinterpret_cast<float4*>(p + C)[q] = mean;
    reinterpret_cast<float4*>(p + 2 * C)[q] = m2;
  }
}
// stage 2: one block of 1024 threads per 32 channels; thread (group gi of 32, channel c) merges slabs gi, gi+32, ...
// in order, then the 32 group results are merged in group order; writes mean, rstd, updates running stats.
// (One block for all channels walked nslab / 4 dependent Chan merges per thread: 10 us for 249 slabs.)
constexpr int BN_CPB = 32, BN_G = 1024 / BN_CPB;
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nslab, int C,
                                                           float eps, float momentum, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           long long* __restrict__ num_batches_tracked) {
  __shared__ float sh[3][1024];
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;   // batchnorm.py: += 1 per training forward
  const int t = threadIdx.x;
  const int cl = t % BN_CPB, gi = t / BN_CPB;
  const int c = blockIdx.x * BN_CPB + cl;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (c < C) {
    for (int s0 = gi; s0 < nslab; s0 += 8 * BN_G) {      // 8 slabs per trip, all 24 loads in flight together
      float pn[8], pm[8], pq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u * BN_G;
        const float* p = part + (long)(s < nslab ? s : s0) * 3 * C;
        pn[u] = s < nslab ? p[c] : 0.f; pm[u] = p[C + c]; pq[u] = s < nslab ? p[2 * C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) chan_merge(n, mean, m2, pn[u], pm[u], pq[u]);
    }
  }
  sh[0][t] = n; sh[1][t] = mean; sh[2][t] = m2;
  __syncthreads();
  if (gi == 0 && c < C) {
    for (int g2 = 1; g2 < BN_G; ++g2) {
      const int o = g2 * BN_CPB + cl;
      chan_merge(n, mean, m2, sh[0][o], sh[1][o], sh[2][o]);
    }
    const float var = m2 / n;
    mean_out[c] = mean;
    rstd_out[c] = rsqrtf(var + eps);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
    }
  }
}
// y = act((x - mean) * rstd * gamma + beta)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long M, int C, int act,
                                int bf16) {
  const long n = M * C;
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned short* y16 = reinterpret_cast<unsigned short*>(y);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c = i % C;
    float z = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    z = eamd_act(z, act);
    if (bf16) y16[i] = eamd_f2bf(z); else y[i] = z;
  }
}
// backward stage 1: dz = dy * act'(z); partial sums of dz and dz*xhat per (slab, channel); same slab
// geometry as the forward statistics
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ part,
                                                             long M, int C, int act, int T,
                                                             const int* __restrict__ bound) {
  __shared__ float4 sh1[256], sh2[256];
  const int tb = bound ? bound[0] : T;
  const BnGeom g = bn_geom(C);
  const int t = threadIdx.x;
  const int q = t % g.cq, sub = t / g.cq;
  const bool live = sub < g.rs;
  const long r0 = (long)blockIdx.x * g.slab_rows + sub;
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu, gm = mu, be = mu;
  if (live) {
    mu = reinterpret_cast<const float4*>(mean)[q]; rs = reinterpret_cast<const float4*>(rstd)[q];
    gm = reinterpret_cast<const float4*>(gamma)[q]; be = reinterpret_cast<const float4*>(beta)[q];
  }
  float4 xv[BN_R], dv[BN_R];
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const long r = r0 + (long)i * g.rs;
    const bool ok = live && r < M && (!bound || (int)(r % T) < tb);     // rows past the bound: dy = 0
    xv[i] = ok ? reinterpret_cast<const float4*>(x + r * C)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    dv[i] = ok ? reinterpret_cast<const float4*>(dy + r * C)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const float* xp = &xv[i].x; const float* dp = &dv[i].x;
    const float* mp = &mu.x; const float* rp = &rs.x; const float* gp = &gm.x; const float* bp = &be.x;
    float* a1 = &s1.x; float* a2 = &s2.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xp[e] - mp[e]) * rp[e];
      const float z = xh * gp[e] + bp[e];
      float d = dp[e];   // rows past M hold dy = 0 => contribute nothing
      if (act == EAMD_ACT_SWISH) d *= eamd_dswish(z);
      else if (act == EAMD_ACT_RELU) d = z > 0.f ? d : 0.f;
      else if (act != EAMD_ACT_NONE) d *= eamd_dact(z, act);
      a1[e] += d; a2[e] += d * xh;
    }
  }
  sh1[t] = s1; sh2[t] = s2;
  __syncthreads();
  if (sub == 0) {
    for (int k = 1; k < g.rs; ++k) {
      const float4 a = sh1[k * g.cq + q], b = sh2[k * g.cq + q];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    float* p = part + (long)blockIdx.x * 2 * C;
    reinterpret_cast<float4*>(p)[q] = s1;
    reinterpret_cast<float4*>(p + C)[q] = s2;
  }
}
// backward stage 2 (one block of 1024 threads per 32 channels): reduce partials in a fixed order -> sums[2][C];
// accumulate dgamma/dbeta
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nslab, int C,
                                                               float* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
  __shared__ float sh[2][1024];
  const int t = threadIdx.x;
  const int cl = t % BN_CPB, gi = t / BN_CPB;
  const int c = blockIdx.x * BN_CPB + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    for (int s0 = gi; s0 < nslab; s0 += 8 * BN_G) {
      float a1[8], a2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u * BN_G;
        const bool ok = s < nslab;
        const float* q = part + (long)(ok ? s : s0) * 2 * C;       // unconditional (clamped) loads, selected afterwards
        const float v1 = q[c], v2 = q[C + c];
        a1[u] = ok ? v1 : 0.f;
        a2[u] = ok ? v2 : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += a1[u]; s2 += a2[u]; }
    }
  }
  sh[0][t] = s1; sh[1][t] = s2;
  __syncthreads();
  if (gi == 0 && c < C) {
    for (int g2 = 1; g2 < BN_G; ++g2) { s1 += sh[0][g2 * BN_CPB + cl]; s2 += sh[1][g2 * BN_CPB + cl]; }
    sums[c] = s1; sums[C + c] = s2;
    dbeta[c] += s1; dgamma[c] += s2;
  }
}
// backward stage 3: dx = gamma*rstd*(dz - s1/M - xhat*s2/M)   (training-mode statistics)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ sums, float* __restrict__ dx, long M, int C, int act,
                                    int training, int T, const int* __restrict__ bound) {
  const long n = M * C;
  const long stride = (long)gridDim.x * blockDim.x;
  const int tb = bound ? bound[0] : T;
  const float invM = bound ? 1.f / ((float)(M / T) * (float)tb) : 1.f / (float)M;      // rows that took part in the statistics
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c = i % C;
    if (bound && (int)((i / C) % T) >= tb) { dx[i] = 0.f; continue; }
    float rs = rstd[c], g = gamma[c];
    float xh = (x[i] - mean[c]) * rs;
    float z = xh * g + beta[c];
    float d = dy[i];
    if (act == EAMD_ACT_SWISH) d *= eamd_dswish(z);
    else if (act == EAMD_ACT_RELU) d = z > 0.f ? d : 0.f;
    else if (act != EAMD_ACT_NONE) d *= eamd_dact(z, act);
    if (training) dx[i] = g * rs * (d - sums[c] * invM - xh * sums[C + c] * invM);
    else dx[i] = g * rs * d;
  }
}

// ---- first convolution of the front-ends: 1 -> C channels, 3x3, + ReLU, NHWC output ----------------
// x [B, T, F]; w [C, 1, 3, 3]; y [B, H, W, C], H = (T + 2 pad - 3)/st + 1, W = (F + 2 pad - 3)/st + 1
// (st 2 / pad 0: Conv2dSubsampling, subsampling.py:28-33; st 1 / pad 1: VGG2L, rnn/encoders.py:184).
// One workgroup per output row (b, hh): the three input rows it needs sit in LDS with a zeroed halo, a
// thread owns one output channel (its 9 taps in registers) and walks the W positions, so every store is
// a full 2*C / 4*C-byte row segment and the inner loop has neither index divisions nor bounds tests.
constexpr int C1_MAXF = 512;     // input feature dim + 2*pad must fit
// Eight positions per trip: their wave-uniform input window (ST*7+3 floats per kernel row) is read as float4
// broadcasts, 15 LDS instructions instead of 72 scalar ones (the scalar form was LDS-issue bound); a thread owns
// NC = 2 adjacent channels when the output is bf16 (one packed 4-byte store per position) or 1 channel otherwise.
template <int ST, int NC>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B,
                                                        int T, int F, int H, int W, int C, int bf16, int pad) {
  constexpr int XLD = C1_MAXF + 32;
  constexpr int NV = (ST * 7 + 3 + 3) / 4;
  __shared__ __attribute__((aligned(16))) float xs[3 * XLD];
  const int row = blockIdx.x;                 // b * H + hh
  const int b = row / H, hh = row % H;
  const int FP = F + 2 * pad;
  for (int i = threadIdx.x; i < 3 * XLD; i += blockDim.x) {
    const int kh = i / XLD, j = i % XLD, f = j - pad;
    const int tt = ST * hh - pad + kh;
    const bool ok = j < FP && tt >= 0 && tt < T && f >= 0 && f < F;
    const float v = x[((long)b * T + min(max(tt, 0), T - 1)) * F + min(max(f, 0), F - 1)];
    xs[i] = ok ? v : 0.f;
  }
  __syncthreads();
  unsigned short* y16 = reinterpret_cast<unsigned short*>(y);
  unsigned int* y32 = reinterpret_cast<unsigned int*>(y);
  for (int c = threadIdx.x * NC; c < C; c += blockDim.x * NC) {
    float wr[NC][9], bv[NC];
#pragma unroll
    for (int e = 0; e < NC; ++e) {
#pragma unroll
      for (int k = 0; k < 9; ++k) wr[e][k] = w[(long)(c + e) * 9 + k];
      bv[e] = bias[c + e];
    }
    const long o = (long)row * W * C + c;
    for (int w0 = 0; w0 < W; w0 += 8) {
      float xr[3][NV * 4];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(&xs[kh * XLD + ST * w0 + 4 * q]);
          xr[kh][4 * q] = v.x; xr[kh][4 * q + 1] = v.y; xr[kh][4 * q + 2] = v.z; xr[kh][4 * q + 3] = v.w;
        }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float acc[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) {
          acc[e] = bv[e];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[e] += wr[e][kh * 3 + kw] * xr[kh][ST * u + kw];
          acc[e] = acc[e] > 0.f ? acc[e] : 0.f;
        }
        if (w0 + u < W) {
          const long oi = o + (long)(w0 + u) * C;
          // the 320 / 640 MB result is written once and read by a later kernel: streaming (nt) stores keep it from
          // evicting the working set in L2 (fp32: 184 -> 140 us = 0.44 -> 0.58 of the HBM rate)
          if (NC == 2) __builtin_nontemporal_store((unsigned)eamd_f2bf(acc[0]) | ((unsigned)eamd_f2bf(acc[NC - 1]) << 16), &y32[oi >> 1]);
          else if (bf16) y16[oi] = eamd_f2bf(acc[0]);
          else __builtin_nontemporal_store(acc[0], &y[oi]);
        }
      }
    }
  }
}
// dW[c, kh, kw] += sum_pos dy[pos, c] * x[pos shifted]; db[c] += sum dy   (dy already ReLU-masked)
// grid (ceil(C/256), row groups): a workgroup walks `rows_per_block` output rows, staging the three input
// rows of each in LDS; thread = channel, 9 + 1 register accumulators.  Per output row the (up to 40) gradient values of
// the thread's channel are requested BEFORE the input rows are staged (one memory round trip per row, not one per
// eight positions), and the wave-uniform input window of eight positions is read as float4 broadcasts
// (ST*7+3 floats per kernel row: 15 LDS instructions per eight positions instead of 72 scalar ones - the scalar form
// was LDS-issue bound).
// NC = 2 (bf16 gradients): a thread owns two adjacent channels and reads them as one dword - sub-dword loads
// (global_load_ushort) run this kernel at 0.9 TB/s where the dword form reaches > 4 TB/s.
template <int ST, int NC>
__global__ __launch_bounds__(256) void conv1_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dw, float* __restrict__ db,
                                                          float* __restrict__ part, int B, int T, int F, int H, int W,
                                                          int C, int rows_per_block, int bf16, int pad) {
  constexpr int XLD = C1_MAXF + 32;          // row stride of the staged input (floats, multiple of 4)
  constexpr int NV = (ST * 7 + 3 + 3) / 4;   // float4 reads covering eight positions of one kernel row
  constexpr int NPOS = NC == 2 ? 24 : 40;    // gradient values requested ahead per output row
  __shared__ __attribute__((aligned(16))) float xs[3 * XLD];
  const int c_raw = (blockIdx.x * blockDim.x + threadIdx.x) * NC;
  const bool live = c_raw < C;
  const int c = live ? c_raw : C - NC;
  const int FP = F + 2 * pad;
  const long nrow = (long)B * H;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(nrow, r0 + (long)rows_per_block);
  const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(dy);
  const unsigned int* dy32 = reinterpret_cast<const unsigned int*>(dy);
  for (int i = threadIdx.x; i < 3 * XLD; i += blockDim.x) xs[i] = 0.f;   // the tail past FP stays zero (read, never NaN)
  float acc[NC][9], accb[NC];
#pragma unroll
  for (int e = 0; e < NC; ++e) {
    accb[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[e][k] = 0.f;
  }
  for (long row = r0; row < r1; ++row) {
    const int b = row / H, hh = row % H;
    const long o = row * W * C + c;
    for (int w00 = 0; w00 < W; w00 += NPOS) {
      float g[NPOS / 8][8][NC];
#pragma unroll
      for (int tq = 0; tq < NPOS / 8; ++tq)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ww = w00 + tq * 8 + u;
          const long oi = o + (long)min(ww, W - 1) * C;
          if (NC == 2) {
            const unsigned v = dy32[oi >> 1];
            g[tq][u][0] = __uint_as_float(v << 16); g[tq][u][NC - 1] = __uint_as_float(v & 0xffff0000u);
          } else {
            g[tq][u][0] = bf16 ? __uint_as_float(((unsigned)dy16[oi]) << 16) : dy[oi];
          }
        }
      if (w00 == 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * FP; i += blockDim.x) {
          const int kh = i / FP, f = i % FP - pad;
          const int tt = ST * hh - pad + kh;
          const bool ok = tt >= 0 && tt < T && f >= 0 && f < F;
          const float v = x[((long)b * T + min(max(tt, 0), T - 1)) * F + min(max(f, 0), F - 1)];
          xs[kh * XLD + i % FP] = ok ? v : 0.f;
        }
        __syncthreads();
      }
#pragma unroll
      for (int tq = 0; tq < NPOS / 8; ++tq) {
        const int w0 = w00 + tq * 8;
        if (w0 < W) {                          // block-uniform
          float xr[3][NV * 4];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int q = 0; q < NV; ++q) {
              const float4 v = *reinterpret_cast<const float4*>(&xs[kh * XLD + ST * w0 + 4 * q]);
              xr[kh][4 * q] = v.x; xr[kh][4 * q + 1] = v.y; xr[kh][4 * q + 2] = v.z; xr[kh][4 * q + 3] = v.w;
            }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int e = 0; e < NC; ++e) {
              const float gv = (w0 + u < W) ? g[tq][u][e] : 0.f;
              accb[e] += gv;
#pragma unroll
              for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc[e][kh * 3 + kw] += gv * xr[kh][ST * u + kw];
            }
          }
        }
      }
    }
  }
  // per-block partial sums [gridDim.y][10][C] (k-major: coalesced), reduced by conv1_bwd_w_reduce_kernel.  (A second
  // form that transposed the sums through LDS and added them to dw / db with atomics when the grid was small was
  // removed in round 2: it was the one code path implicated in a wrong conv.0.weight gradient seen once in round 1.)
  if (live) {
#pragma unroll
    for (int e = 0; e < NC; ++e) {
      float* pp = part + (long)blockIdx.y * 10 * C + c + e;
#pragma unroll
      for (int k = 0; k < 9; ++k) pp[(long)k * C] = acc[e][k];
      pp[(long)9 * C] = accb[e];
    }
  }
}
// part[nblk][10][C] -> dw[C][9] +=, db[C] +=.  grid (ceil(10*C/64), slices): 4 row-subgroups x 64 columns per block
__global__ __launch_bounds__(256) void conv1_bwd_w_reduce_kernel(const float* __restrict__ part, int nblk, int C,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;         // index into [10][C]
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nblk, r0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < 10 * C) {
    int r = r0 + sub;
    for (; r + 12 < r1; r += 16) {
      s0 += part[(long)r * 10 * C + col];        s1 += part[(long)(r + 4) * 10 * C + col];
      s2 += part[(long)(r + 8) * 10 * C + col];  s3 += part[(long)(r + 12) * 10 * C + col];
    }
    for (; r < r1; r += 4) s0 += part[(long)r * 10 * C + col];
  }
  red[sub][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0 && col < 10 * C) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int k = col / C, c = col % C;
    if (k < 9) atomicAdd(&dw[(long)c * 9 + k], v); else atomicAdd(&db[c], v);
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

// frames per thread of the depthwise convolution: fewer frames = more workgroups in flight (the kernel is
// latency-bound at B*T ~ 8k frames), more frames = less re-reading of the K-1 halo.  EAMD_DWCONV_TT overrides.
inline void dwconv_launch(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int K,
                          int flip, hipStream_t s) {
  static const int tt_env = [] { const char* e = getenv("EAMD_DWCONV_TT"); return e ? atoi(e) : 0; }();
  const int pad = (K - 1) / 2;
  if (tt_env == 0) {      // LDS-tiled default: 64 channels x 64 frames per block
    hipLaunchKernelGGL((dwconv_lds_kernel<8, 0>), dim3((C + 63) / 64, B * ((T + 63) / 64)), dim3(512), 0, s, x, w, bias, y, B, T, C,
                       K, pad, flip, (const float*)nullptr, 0, (float*)nullptr);
    return;
  }
  const int tt = tt_env;
  const int gx = (C + 255) / 256;
  if (tt == 4)
    hipLaunchKernelGGL(dwconv_kernel<4>, dim3(gx, B * ((T + 3) / 4)), dim3(256), 0, s, x, w, bias, y, B, T, C, K, pad, flip);
  else if (tt == 16)
    hipLaunchKernelGGL(dwconv_kernel<16>, dim3(gx, B * ((T + 15) / 16)), dim3(256), 0, s, x, w, bias, y, B, T, C, K, pad, flip);
  else
    hipLaunchKernelGGL(dwconv_kernel<8>, dim3(gx, B * ((T + 7) / 8)), dim3(256), 0, s, x, w, bias, y, B, T, C, K, pad, flip);
}

}  // namespace

extern "C" {

int eamd_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int C, int K,
                    void* stream) {
  if (!x || !w || !y || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  dwconv_launch(x, w, bias, y, B, T, C, K, 0, (hipStream_t)stream);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dwconv_bwd_x(const float* dy, const float* w, float* dx, int B, int T, int C, int K, void* stream) {
  if (!dy || !w || !dx || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  dwconv_launch(dy, w, (const float*)nullptr, dx, B, T, C, K, 1, (hipStream_t)stream);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dwconv_bwd_w(const float* dy, const float* x, float* dw, float* db, int B, int T, int C, int K,
                      void* stream) {
  if (!dy || !x || !dw || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  const long total = (long)B * ((T + WNWV * 8 - 1) / (WNWV * 8));      // tiles of 64 frames
  const int gx = (C + 63) / 64;
  // about two 8-wave blocks per CU; more tiles per block = fewer global atomics, fewer blocks in flight
  long gy = (512 + gx - 1) / gx;
  if (gy > total) gy = total;
  if (gy < 1) gy = 1;
  static const int tpb_env = [] { const char* e = getenv("EAMD_DWW_TPB"); return e ? atoi(e) : 0; }();
  const long tpb = tpb_env ? tpb_env : (total + gy - 1) / gy;
  gy = (total + tpb - 1) / tpb;
  hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3(gx, (unsigned)gy), dim3(WNWV * 64), 0, (hipStream_t)stream, dy, x, dw, db, B,
                     T, C, K, (K - 1) / 2, (int)tpb, 0);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* GLU-fused twins for the Conformer convolution module: `a` is the pointwise-conv output [B, T, 2C] (value | gate). */
int eamd_dwconv_glu_fwd(const float* a, const float* w, const float* bias, float* y, float* bn_part, int B, int T, int C, int K,
                        void* stream) {
  if (!a || !w || !y || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL((dwconv_lds_kernel<8, 1>), dim3((C + 63) / 64, B * ((T + 63) / 64)), dim3(512), 0, (hipStream_t)stream, a, w,
                     bias, y, B, T, C, K, (K - 1) / 2, 0, (const float*)nullptr, 0, bn_part);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dwconv_glu_bwd_x(const float* dy, const float* w, const float* a, void* da, int da_bf16, int B, int T, int C, int K,
                          void* stream) {
  if (!dy || !w || !a || !da || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  hipLaunchKernelGGL((dwconv_lds_kernel<8, 2>), dim3((C + 63) / 64, B * ((T + 63) / 64)), dim3(512), 0, (hipStream_t)stream, dy, w,
                     (const float*)nullptr, (float*)da, B, T, C, K, (K - 1) / 2, 1, a, da_bf16, (float*)nullptr);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_dwconv_glu_bwd_w(const float* dy, const float* a, float* dw, float* db, int B, int T, int C, int K, void* stream) {
  if (!dy || !a || !dw || B <= 0 || T <= 0 || C <= 0 || K <= 0 || (K & 1) == 0) return EAMD_EINVAL;
  if (K > KMAX) return EAMD_EUNSUPPORTED;
  const long total = (long)B * ((T + WNWV * 8 - 1) / (WNWV * 8));
  const int gx = (C + 63) / 64;
  long gy = (512 + gx - 1) / gx;
  if (gy > total) gy = total;
  if (gy < 1) gy = 1;
  const long tpb = (total + gy - 1) / gy;
  gy = (total + tpb - 1) / tpb;
  hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3(gx, (unsigned)gy), dim3(WNWV * 64), 0, (hipStream_t)stream, dy, a, dw, db, B,
                     T, C, K, (K - 1) / 2, (int)tpb, 1);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* workspace: 3*C*nslab floats where nslab = eamd_bn_nslab(M, C) */
static bool bn_shape_ok(int C) { return C >= 4 && C <= 1024 && C % 4 == 0; }
int eamd_bn_nslab(int64_t M, int C) {
  if (!bn_shape_ok(C) || M <= 0) return 0;
  const BnGeom g = bn_geom(C);
  return (int)((M + g.slab_rows - 1) / g.slab_rows);
}

static int bn_stats_impl(const float* x, float* workspace, float* mean, float* rstd, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float eps, float momentum, int T,
                         const int32_t* bound, void* stream) {
  if (!x || !workspace || !mean || !rstd || M <= 0 || C <= 0) return EAMD_EINVAL;
  if (bound && (T <= 0 || M % T != 0)) return EAMD_EINVAL;
  if (!bn_shape_ok(C)) return EAMD_EUNSUPPORTED;   // channel counts 4..1024, multiples of 4
  if ((uintptr_t)x & 15) return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nslab = eamd_bn_nslab(M, C);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(nslab), dim3(256), 0, s, x, workspace, (long)M, C, T, (const int*)bound);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BN_CPB - 1) / BN_CPB), dim3(1024), 0, s, workspace, nslab, C, eps, momentum, mean, rstd,
                     running_mean, running_var, (long long*)num_batches_tracked);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
int eamd_bn_stats(const float* x, float* workspace, float* mean, float* rstd, float* running_mean,
                  float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float eps, float momentum, void* stream) {
  return bn_stats_impl(x, workspace, mean, rstd, running_mean, running_var, num_batches_tracked, M, C, eps, momentum, 1, nullptr, stream);
}
int eamd_bn_stats_bounded(const float* x, float* workspace, float* mean, float* rstd, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, int64_t M, int C, float eps, float momentum, int T,
                          const int32_t* bound, void* stream) {
  if (!bound) return EAMD_EINVAL;
  return bn_stats_impl(x, workspace, mean, rstd, running_mean, running_var, num_batches_tracked, M, C, eps, momentum, T, bound, stream);
}

/* second stage of eamd_bn_stats alone, for partial statistics a producer kernel left behind (eamd_dwconv_glu_fwd's bn_part:
 * nslab = B * ceil(T / 64) slabs of [count | mean | M2] x C) */
int eamd_bn_finalize(const float* part, int nslab, float* mean, float* rstd, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, int C, float eps, float momentum, void* stream) {
  if (!part || !mean || !rstd || nslab <= 0 || C <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BN_CPB - 1) / BN_CPB), dim3(1024), 0, (hipStream_t)stream, part, nslab, C, eps,
                     momentum, mean, rstd, running_mean, running_var, (long long*)num_batches_tracked);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int eamd_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  void* y, int64_t M, int C, int act, int y_bf16, void* stream) {
  if (!x || !mean || !rstd || !gamma || !beta || !y || M <= 0 || C <= 0) return EAMD_EINVAL;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * C)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma,
                     beta, (float*)y, (long)M, C, act, y_bf16);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

/* workspace: (2*nslab + 2)*C floats */
static int bn_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, float* workspace, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                       int act, int training, int T, const int32_t* bound, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !beta || !workspace || !dx || !dgamma || !dbeta || M <= 0 || C <= 0)
    return EAMD_EINVAL;
  if (bound && (T <= 0 || M % T != 0)) return EAMD_EINVAL;
  if (!bn_shape_ok(C)) return EAMD_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)gamma | (uintptr_t)beta) & 15)
    return EAMD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nslab = eamd_bn_nslab(M, C);
  float* sums = workspace + (long)nslab * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nslab), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta,
                     workspace, (long)M, C, act, T, (const int*)bound);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BN_CPB - 1) / BN_CPB), dim3(1024), 0, s, workspace, nslab, C, sums, dgamma, dbeta);
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * C)), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta,
                     sums, dx, (long)M, C, act, training, T, (const int*)bound);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
int eamd_bn_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                const float* beta, float* workspace, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                int act, int training, void* stream) {
  return bn_bwd_impl(dy, x, mean, rstd, gamma, beta, workspace, dx, dgamma, dbeta, M, C, act, training, 1, nullptr, stream);
}
int eamd_bn_bwd_bounded(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, float* workspace, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                        int act, int training, int T, const int32_t* bound, void* stream) {
  if (!bound) return EAMD_EINVAL;
  return bn_bwd_impl(dy, x, mean, rstd, gamma, beta, workspace, dx, dgamma, dbeta, M, C, act, training, T, bound, stream);
}

static int conv_c1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int T, int F, int C,
                       int y_bf16, int st, int pad, void* stream) {
  if (!x || !w || !bias || !y || B <= 0 || T + 2 * pad < 3 || F + 2 * pad < 3 || C <= 0) return EAMD_EINVAL;
  if (F + 2 * pad > C1_MAXF) return EAMD_EUNSUPPORTED;
  int H = (T + 2 * pad - 3) / st + 1, W = (F + 2 * pad - 3) / st + 1;
  hipStream_t s = (hipStream_t)stream;
  const bool pair = y_bf16 && (C % 2 == 0) && (((uintptr_t)y & 3) == 0);
  const int nthr = pair ? ((C / 2 + 63) / 64) * 64 : 256;
#define EAMD_C1F(ST_, NC_) hipLaunchKernelGGL((conv1_fwd_kernel<ST_, NC_>), dim3(B * H), dim3(min(nthr, 256)), 0, s, x, w, \
                                              bias, (float*)y, B, T, F, H, W, C, y_bf16, pad)
  if (st == 2) { if (pair) EAMD_C1F(2, 2); else EAMD_C1F(2, 1); }
  else         { if (pair) EAMD_C1F(1, 2); else EAMD_C1F(1, 1); }
#undef EAMD_C1F
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
int eamd_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int T, int F, int C,
                   int y_bf16, void* stream) {
  return conv_c1_fwd(x, w, bias, y, B, T, F, C, y_bf16, 2, 0, stream);
}
int eamd_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int T, int F, int C,
                        int y_bf16, void* stream) {
  return conv_c1_fwd(x, w, bias, y, B, T, F, C, y_bf16, 1, 1, stream);
}

static void conv_c1_bwd_w_grid(int B, int H, int C, int* gx, int* gy, long* rpb) {
  long nrow = (long)B * H;
  *gx = (C + 255) / 256;
  static const int want_env = [] { const char* e = getenv("EAMD_C1W_BLOCKS"); return e ? atoi(e) : 2048; }();
  long want = want_env / *gx; if (want < 1) want = 1;
  *rpb = (nrow + want - 1) / want; if (*rpb < 1) *rpb = 1;
  *gy = (int)((nrow + *rpb - 1) / *rpb);
}
static int conv_c1_bwd_w(const void* dy, const float* x, float* dw, float* db, float* workspace, int B, int T, int F,
                         int C, int dy_bf16, int st, int pad, void* stream) {
  if (!dy || !x || !dw || !db || B <= 0 || T + 2 * pad < 3 || F + 2 * pad < 3 || C <= 0) return EAMD_EINVAL;
  if (F + 2 * pad > C1_MAXF) return EAMD_EUNSUPPORTED;
  int H = (T + 2 * pad - 3) / st + 1, W = (F + 2 * pad - 3) / st + 1;
  int gx, gy; long rpb;
  conv_c1_bwd_w_grid(B, H, C, &gx, &gy, &rpb);
  if (!workspace) return EAMD_EINVAL;
  float* part = workspace;
  hipStream_t s = (hipStream_t)stream;
  const bool pair = dy_bf16 && (C % 2 == 0) && (((uintptr_t)dy & 3) == 0);
  const int nthr = pair ? min(256, ((C / 2 + 63) / 64) * 64) : 256;
  if (pair) gx = (C / 2 + nthr - 1) / nthr;
#define EAMD_C1W(ST_, NC_) hipLaunchKernelGGL((conv1_bwd_w_kernel<ST_, NC_>), dim3(gx, gy), dim3(nthr), 0, s, \
                                              (const float*)dy, x, dw, db, part, B, T, F, H, W, C, (int)rpb, dy_bf16, pad)
  if (st == 2) { if (pair) EAMD_C1W(2, 2); else EAMD_C1W(2, 1); }
  else         { if (pair) EAMD_C1W(1, 2); else EAMD_C1W(1, 1); }
#undef EAMD_C1W
  EAMD_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv1_bwd_w_reduce_kernel, dim3((10 * C + 63) / 64, min(16, (gy + 31) / 32)), dim3(256), 0, s, part,
                     gy, C, dw, db);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
/* floats of scratch for eamd_conv1_bwd_w / eamd_conv3x3_c1_bwd_w (stride/pad as in the forward) */
static int64_t conv_c1_ws(int B, int T, int C, int st, int pad) {
  if (B <= 0 || T + 2 * pad < 3 || C <= 0) return 0;
  int H = (T + 2 * pad - 3) / st + 1;
  int gx, gy; long rpb;
  conv_c1_bwd_w_grid(B, H, C, &gx, &gy, &rpb);
  return (int64_t)gy * 10 * C;
}
int64_t eamd_conv1_bwd_w_workspace(int B, int T, int C) { return conv_c1_ws(B, T, C, 2, 0); }
int64_t eamd_conv3x3_c1_bwd_w_workspace(int B, int T, int C) { return conv_c1_ws(B, T, C, 1, 1); }
int eamd_conv1_bwd_w(const void* dy, const float* x, float* dw, float* db, float* workspace, int B, int T, int F,
                     int C, int dy_bf16, void* stream) {
  return conv_c1_bwd_w(dy, x, dw, db, workspace, B, T, F, C, dy_bf16, 2, 0, stream);
}
int eamd_conv3x3_c1_bwd_w(const void* dy, const float* x, float* dw, float* db, float* workspace, int B, int T, int F,
                          int C, int dy_bf16, void* stream) {
  return conv_c1_bwd_w(dy, x, dw, db, workspace, B, T, F, C, dy_bf16, 1, 1, stream);
}

}  // extern "C"
