// Fused position-wise feed-forward for gfx950, fp32 operands, SYMMETRIC form: all eight waves of a workgroup do the same
// work (an eighth of the first product, of its epilogue and of the second product per chunk of 256 hidden units).
//
// reference: espnet/nets/pytorch_backend/transformer/positionwise_feed_forward.py:12-32
//     forward   out = R + alpha * drop_out( drop_in(act(x W1^T + b1)) W2^T + b2 )
//     backward  dz = alpha * (dy W2) (.) f,   dx = dz W1
//
// Why a second form beside ffn_f32.hip's role-split kernel: there four waves form z and carry ALL of the epilogue arithmetic
// while the other four contract - the stamps showed the contracting waves waiting ~28 % of every chunk period at the barrier
// for the epilogue.  Here every wave owns 32 hidden units of a chunk (two 16-column MFMA tiles, columns interleaved so that a
// lane holds an adjacent PAIR: one dropout hash, 8-byte stores) and 32 output columns, so the epilogue is spread over all
// eight waves; h and f go to global memory straight from the accumulators (a lane's pair x 16 lanes = one full 128-byte line
// per row), only h passes through LDS - as the A operand of the second product.  Weights come from packed fragment-order
// images through a ring of four register sets, three 16-k groups ahead; the 32 input rows stay in LDS; ONE barrier per chunk
// (h chunks double-buffered).  Needs F % 256 == 0 (ffn_f32.hip's kernel takes the other multiples of 128).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"
#include "ffn_ln.h"

namespace {

constexpr int SBM = 32, SD = 256, SHC = 256, SNT = 512;
constexpr int SX_LD = SD + 8;              // 264: ds_read_b128 rows conflict-free (stride = 2 mod 16 chunks of 16 bytes)
constexpr int SH_LD = SHC + 8;
constexpr int SX_SZ = SBM * SX_LD, SH_SZ = SBM * SH_LD;
constexpr size_t S_SMEM = (size_t)(SX_SZ + 2 * SH_SZ) * sizeof(float);
constexpr int GROUP_BYTES = 2 * 64 * 16;   // one 16-k group of a wave: 2 column tiles x 64 lanes x 16 bytes

template <bool BWD, int ACT>
__global__ __launch_bounds__(SNT, 2) void ffn_f32_sym_kernel(const eamd_ffn_t p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const xs = sm;                    // [32][264]: input rows (forward: x, backward: dy); the output staging tile at the end
  float* const hs = xs + SX_SZ;            // [2][32][264]: h (forward) / dz (backward) chunk images
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * SBM;
  const int F = p.F, nch = F / SHC;
  const bool full_rows = m0 + SBM <= p.M;
  if (!BWD && p.ln_x) {        // LayerNorm in front: normalise the rows on their way into LDS (ffn_ln.h)
    ffn_ln_stage(p, m0, t, [&](int row, int col, float4 y) __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(&xs[row * SX_LD + col]) = (f32x4){y.x, y.y, y.z, y.w};
    });
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + SNT * i, row = idx >> 6, c4 = idx & 63;
      *reinterpret_cast<f32x4*>(&xs[row * SX_LD + c4 * 4]) =
          *reinterpret_cast<const f32x4*>(p.x + (long)min(m0 + row, p.M - 1) * SD + c4 * 4);
    }
  }
  // packed images (ffn_pack_f32_sym_kernel): image[c][w][g][j][lane] = four k-elements of column tile j for the 16-k group g
  const char* const Wa = reinterpret_cast<const char*>(p.w1) + (long)w * 16 * GROUP_BYTES + lane * 16;
  const char* const Wb = reinterpret_cast<const char*>(p.w2) + (long)w * 16 * GROUP_BYTES + lane * 16;
  constexpr long CHUNK_BYTES = 8L * 16 * GROUP_BYTES;
  f32x4 ring[4][2];
  // weight stream: step s of chunk c = group s of the first product (s < 16) or group s - 16 of the second
  auto load_w = [&](auto set_c, int c, int s) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    int cc = c + (s >> 5);
    const int ss = s & 31;
    cc = min(cc, nch - 1);
    const char* base = (ss < 16 ? Wa : Wb) + (long)cc * CHUNK_BYTES + (long)(ss & 15) * GROUP_BYTES;
    ring[SET][0] = *reinterpret_cast<const f32x4*>(base);
    ring[SET][1] = *reinterpret_cast<const f32x4*>(base + 1024);
  };
  f32x4 yacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned thr_in = eamd_drop_thr16(p.p_in);
  const float inv_in = p.p_in > 0.f ? eamd_drop_inv(thr_in) : 1.f;
  const unsigned seed_in = (!BWD && p.p_in > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_in) : 0u;
  const int lc0 = w * 32 + 2 * fr;                       // this lane's column pair inside a chunk
  // row offsets (elements) of this lane's accumulator rows in the [M, F] tensors h / f
  long grow[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) grow[i][r] = (long)min(m0 + i * 16 + fq * 4 + r, p.M - 1) * F + lc0;

  auto mfma16 = [&](auto set_c, const f32x4 (&a)[2], f32x4 (&acc)[2][2]) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], ring[SET][j][e], acc[i][j], 0, 0, 0);
  };

  load_w(std::integral_constant<int, 0>{}, 0, 0);
  load_w(std::integral_constant<int, 1>{}, 0, 1);
  load_w(std::integral_constant<int, 2>{}, 0, 2);
  __syncthreads();                                        // the input image is in LDS

  for (int c = 0; c < nch; ++c) {
    float* const hb = hs + (c & 1) * SH_SZ;
    // small loads first (vmcnt retires in order): bias pair / the factor values of this chunk
    float bpre[2] = {0.f, 0.f};
    float fpre[2][4][2];
    if constexpr (BWD) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 v = *reinterpret_cast<const float2*>(p.f + grow[i][r] + (long)c * SHC);
          fpre[i][r][0] = v.x; fpre[i][r][1] = v.y;
        }
    } else if (p.b1) {
      bpre[0] = p.b1[c * SHC + lc0]; bpre[1] = p.b1[c * SHC + lc0 + 1];
    }
    // ---- first product: z[32][this wave's 32 columns of the chunk] ----
    f32x4 zacc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if ((g & 3) == 0) load_w(std::integral_constant<int, 3>{}, c, g + 3);
      else if ((g & 3) == 1) load_w(std::integral_constant<int, 0>{}, c, g + 3);
      else if ((g & 3) == 2) load_w(std::integral_constant<int, 1>{}, c, g + 3);
      else load_w(std::integral_constant<int, 2>{}, c, g + 3);
      __builtin_amdgcn_sched_barrier(0);          // the requests stay here, three groups ahead of their use
      f32x4 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&xs[(i * 16 + fr) * SX_LD + g * 16 + fq * 4]);
      if ((g & 3) == 0) mfma16(std::integral_constant<int, 0>{}, a, zacc);
      else if ((g & 3) == 1) mfma16(std::integral_constant<int, 1>{}, a, zacc);
      else if ((g & 3) == 2) mfma16(std::integral_constant<int, 2>{}, a, zacc);
      else mfma16(std::integral_constant<int, 3>{}, a, zacc);
    }
    // ---- epilogue on the accumulators: element (i, j, r) = row i*16 + fq*4 + r, column lc0 + j of the chunk ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = i * 16 + fq * 4 + r;
        float hv[2], fv[2] = {0.f, 0.f};
        if constexpr (!BWD) {
#pragma unroll
          for (int j = 0; j < 2; ++j) eamd_act_dact(zacc[i][j][r] + bpre[j], ACT, hv[j], fv[j]);
          if (p.p_in > 0.f) {
            const unsigned gi = (unsigned)(m0 + lr) * (unsigned)F + (unsigned)(c * SHC + lc0);       // even
            const unsigned hsh = eamd_drop_pair(seed_in, (unsigned long long)(gi >> 1));
            const bool k0 = (hsh & 0xffffu) >= thr_in, k1 = (hsh >> 16) >= thr_in;
            hv[0] = k0 ? hv[0] * inv_in : 0.f; fv[0] = k0 ? fv[0] * inv_in : 0.f;
            hv[1] = k1 ? hv[1] * inv_in : 0.f; fv[1] = k1 ? fv[1] * inv_in : 0.f;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) hv[j] = (zacc[i][j][r] * fpre[i][r][j]) * p.alpha;
        }
        *reinterpret_cast<float2*>(&hb[lr * SH_LD + lc0]) = make_float2(hv[0], hv[1]);
        if (full_rows || m0 + lr < p.M) {
          if (p.h) *reinterpret_cast<float2*>(p.h + grow[i][r] + (long)c * SHC) = make_float2(hv[0], hv[1]);
          if constexpr (!BWD) {
            if (p.f) *reinterpret_cast<float2*>(p.f + grow[i][r] + (long)c * SHC) = make_float2(fv[0], fv[1]);
          }
        }
      }
    // the chunk image is complete: an LDS-only barrier (__syncthreads would also drain vmcnt - the h / f stores and the
    // weight requests three groups ahead - at every chunk)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- second product: out[32][this wave's 32 columns] += h_chunk W2[:, chunk]^T ----
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if ((g & 3) == 0) load_w(std::integral_constant<int, 3>{}, c, g + 19);
      else if ((g & 3) == 1) load_w(std::integral_constant<int, 0>{}, c, g + 19);
      else if ((g & 3) == 2) load_w(std::integral_constant<int, 1>{}, c, g + 19);
      else load_w(std::integral_constant<int, 2>{}, c, g + 19);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&hb[(i * 16 + fr) * SH_LD + g * 16 + fq * 4]);
      if ((g & 3) == 0) mfma16(std::integral_constant<int, 0>{}, a, yacc);
      else if ((g & 3) == 1) mfma16(std::integral_constant<int, 1>{}, a, yacc);
      else if ((g & 3) == 2) mfma16(std::integral_constant<int, 2>{}, a, yacc);
      else mfma16(std::integral_constant<int, 3>{}, a, yacc);
    }
  }
  __syncthreads();                                        // every wave is done with the images: the staging tile goes over xs
  float* const ys = xs;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ys[(i * 16 + fq * 4 + r) * SX_LD + w * 32 + j * 16 + fr] = yacc[i][j][r];
  __syncthreads();
  const unsigned thr_out = eamd_drop_thr16(p.p_out);
  const float inv_out = eamd_drop_inv(thr_out);
  const unsigned seed_out = (!BWD && p.p_out > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_out) : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + SNT * i, lr = idx >> 6, c4 = idx & 63;
    const int row = m0 + lr;
    if (row >= p.M) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(&ys[lr * SX_LD + c4 * 4]);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const long gi = (long)row * SD + c4 * 4;
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

// The four packed images of one FFN for the symmetric kernel: piece = ((((c*8 + w)*16 + g)*2 + j)*64 + lane), 16 bytes each.
//   which 0 (forward, first):   { W1[u][16g + 4fq + e] }   u = c*256 + w*32 + 2fr + j      (column tiles interleaved)
//   which 1 (forward, second):  { W2[n][c*256 + 16g + 4fq + e] }   n = w*32 + 16j + fr
//   which 2 (backward, first):  { W2[16g + 4fq + e][u] }   (dh = dy W2: contraction over the outputs)
//   which 3 (backward, second): { W1[c*256 + 16g + 4fq + e][n] }   (dx = dz W1: contraction over the hidden units)
__global__ __launch_bounds__(256) void ffn_pack_f32_sym_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                               float* __restrict__ p0, float* __restrict__ p1,
                                                               float* __restrict__ p2, float* __restrict__ p3, int F) {
  const int which = blockIdx.y;
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;
  const long npiece = (long)(F / SHC) * 8 * 16 * 2 * 64;
  if (piece >= npiece) return;
  const int lane = piece & 63, j = (piece >> 6) & 1, g = (piece >> 7) & 15, w = (piece >> 11) & 7, c = (int)(piece >> 14);
  const int fr = lane & 15, fq = lane >> 4;
  const int u = c * SHC + w * 32 + 2 * fr + j, n = w * 32 + 16 * j + fr;
  float4 o;
  if (which == 0) {
    o = *reinterpret_cast<const float4*>(w1 + (long)u * SD + 16 * g + 4 * fq);
    *reinterpret_cast<float4*>(p0 + piece * 4) = o;
  } else if (which == 1) {
    o = *reinterpret_cast<const float4*>(w2 + (long)n * F + c * SHC + 16 * g + 4 * fq);
    *reinterpret_cast<float4*>(p1 + piece * 4) = o;
  } else if (which == 2) {
    const float* q = w2 + (long)(16 * g + 4 * fq) * F + u;
    *reinterpret_cast<float4*>(p2 + piece * 4) = make_float4(q[0], q[F], q[2L * F], q[3L * F]);
  } else {
    const float* q = w1 + (long)(c * SHC + 16 * g + 4 * fq) * SD + n;
    *reinterpret_cast<float4*>(p3 + piece * 4) = make_float4(q[0], q[SD], q[2 * SD], q[3 * SD]);
  }
}

template <bool BWD, int ACT>
int launch_sym(const eamd_ffn_t& p, hipStream_t stream) {
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_f32_sym_kernel<BWD, ACT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)S_SMEM);
  if (attr_err != hipSuccess) return (int)attr_err;
  const int nblk = (p.M + SBM - 1) / SBM;
  hipLaunchKernelGGL((ffn_f32_sym_kernel<BWD, ACT>), dim3(nblk), dim3(SNT), S_SMEM, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // namespace

// which fp32 form a layer of hidden width F takes (the pack and the launch must agree): EAMD_FFN_F32_FORM = "role" (default) |
// "sym".  Measured at config 2 (tools/ffn_fused_probe.py fp32, forward with saved tensors / backward): role-split 145 / 146 us,
// symmetric 150 / 142-144 us - spreading the epilogue over all eight waves did NOT lift the matrix-pipe utilisation (both forms
// sit at ~74 %), so the role-split kernel stays the default and this one is kept as the measured alternative.
bool eamd_ffn_f32_sym(int F) {
  static const int form = [] {
    const char* e = getenv("EAMD_FFN_F32_FORM");
    return (e && e[0] == 's') ? 1 : 0;
  }();
  return form == 1 && F % SHC == 0;
}

int eamd_ffn_f32_sym_launch(const eamd_ffn_t* p, int bwd, void* stream) {
  if (bwd) return launch_sym<true, EAMD_ACT_NONE>(*p, (hipStream_t)stream);
  return p->act == EAMD_ACT_SWISH ? launch_sym<false, EAMD_ACT_SWISH>(*p, (hipStream_t)stream)
                                  : launch_sym<false, EAMD_ACT_RELU>(*p, (hipStream_t)stream);
}

int eamd_ffn_f32_sym_pack(const float* w1, const float* w2, float* p0, float* p1, float* p2, float* p3, int F, void* stream) {
  const long npiece = (long)(F / SHC) * 8 * 16 * 2 * 64;
  hipLaunchKernelGGL(ffn_pack_f32_sym_kernel, dim3((unsigned)((npiece + 255) / 256), 4), dim3(256), 0, (hipStream_t)stream, w1, w2,
                     p0, p1, p2, p3, F);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}
