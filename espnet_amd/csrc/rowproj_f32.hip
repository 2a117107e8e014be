// Row-block projections for gfx950, fp32 operands (v_mfma_f32_16x16x4_f32: exact fp32 products).
//
// The Conformer's K = 256 / N = 256 products around the attention and the convolution module
//     q, k, v = LN(x) W3^T + b3          (transformer/attention.py:40-61 behind conformer/encoder_layer.py:106-113)
//     out     = x + drop(ctx Wo^T + bo)  (attention.py:90-92, encoder_layer.py:126-129)
//     a       = LN(x) W1^T + b1          (conformer/convolution.py:63 pointwise_conv1 behind encoder_layer.py:132-135)
//     out     = x + drop(e W2^T + b2)    (convolution.py:76 pointwise_conv2, encoder_layer.py:136-138)
// and their input gradients dctx = dy Wo, dxn = dqkv W3, de = dy W2, dxn = da W1 (each dxn followed by the LayerNorm
// backward) are, as tile GEMMs, launches of 250 - 750 64x64 tiles with 8 K-tiles each: prologue, result store and tail of
// every tile are exposed (53 - 86 TFLOP/s at config 2), and every LayerNorm is one more pass over the rows.  They are local
// to a block of ROWS, like the feed-forward pair (ffn_f32.hip), so the same structure serves them: one workgroup takes 32 rows
// through the whole product,
//   * the rows are staged ONCE in LDS (optionally normalised on the way: LayerNorm in front, as ffn_ln.h),
//   * the weights come from a PACKED image in MFMA fragment order straight into operand registers (a wave-instruction
//     reads 1 KB of consecutive bytes; ring of four fragment sets, three K-steps ahead),
//   * wave w owns output columns 32 w .. 32 w + 31 of every 256-column chunk (wave tile 32 x 32, K-steps of 32),
//   * a chunk's results leave through a wave-private LDS tile as 16-byte row pieces with bias / dropout / residual applied -
//     or, for the input gradients that a LayerNorm backward follows (N = 256: whole rows in one workgroup), through the
//     LayerNorm backward itself (dx, the residual gradient added, the dropped copy the previous block's products read, and
//     the per-workgroup partial sums of d gamma / d beta for the batched second stage of rowops.hip).
// No barrier inside the K loop (every wave reads the shared rows, nobody writes them).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"
#include "ln_bwd_rows.h"

namespace {

constexpr int RBM = 32;          // rows per workgroup
constexpr int RNT = 512;         // 8 waves
constexpr int RCH = 256;         // output columns per chunk: 8 waves x 32
constexpr int ST_LD = 36;        // wave-private result tile [32][32 + 4]
constexpr int ST_SZ = RBM * ST_LD;
constexpr int TL_LD = 264;       // [32][256 + 8] whole-row tile of the LayerNorm-backward epilogue

__device__ __forceinline__ int xs_ld(int K) { return K + 8; }      // ds_read_b128 conflict-free (row stride = 2 mod 16 chunks)

// LayerNorm of the 32 input rows while they are staged (K = 256): thread t owns row t >> 4, columns 4 (l + 16 j) .. + 3
// (ffn_ln.h's arithmetic: mean, centred sum of squares of the register-resident values, rsqrt(var + eps)); the normalised rows
// also go to p.a (backward's weight gradient reads them), mean / rstd to ln_mean / ln_rstd.
__device__ __forceinline__ void stage_ln(const eamd_rowproj_t& p, const int m0, const int t, float* xs, const int LD) {
  constexpr int D = 256;
  const int row = t >> 4, l = t & 15;
  const bool live = m0 + row < p.M;
  const long gr = (long)min(m0 + row, p.M - 1);
  const float4* __restrict__ xr = reinterpret_cast<const float4*>(p.ln_x + gr * D);
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = xr[l + 16 * j]; s += v[j].x + v[j].y + v[j].z + v[j].w; }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
  const float mean = s / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += a * a + b * b + c * c + d * d;
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
  const float rstd = rsqrtf(q / D + p.ln_eps);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(p.ln_w);
  const float4* __restrict__ b4 = reinterpret_cast<const float4*>(p.ln_b);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 g = g4[l + 16 * j], b = b4[l + 16 * j];
    float4 o;
    o.x = (v[j].x - mean) * rstd * g.x + b.x;
    o.y = (v[j].y - mean) * rstd * g.y + b.y;
    o.z = (v[j].z - mean) * rstd * g.z + b.z;
    o.w = (v[j].w - mean) * rstd * g.w + b.w;
    const int col = (l + 16 * j) * 4;
    *reinterpret_cast<float4*>(&xs[row * LD + col]) = o;
    if (live) *reinterpret_cast<float4*>(const_cast<float*>(p.a) + gr * D + col) = o;
  }
  if (live && l == 0) { p.ln_mean[gr] = mean; p.ln_rstd[gr] = rstd; }
}

// Per-column affine + activation of the staged rows (K = 256): a' = act(a * scale[k] + shift[k]) - the BatchNorm apply + Swish
// in front of pointwise_conv2 (convolution.py:73-76).  a' also goes to p.a_out (the weight gradient's operand).
__device__ __forceinline__ void stage_affine(const eamd_rowproj_t& p, const int m0, const int t, float* xs, const int LD) {
  constexpr int D = 256;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + RNT * i, row = idx >> 6, c4 = idx & 63;
    const long gr = (long)min(m0 + row, p.M - 1);
    const float4 v = *reinterpret_cast<const float4*>(p.a + gr * p.lda + c4 * 4);
    const float4 sc = *reinterpret_cast<const float4*>(p.a_scale + c4 * 4), sh = *reinterpret_cast<const float4*>(p.a_shift + c4 * 4);
    float4 o;
    o.x = eamd_act(v.x * sc.x + sh.x, p.a_act); o.y = eamd_act(v.y * sc.y + sh.y, p.a_act);
    o.z = eamd_act(v.z * sc.z + sh.z, p.a_act); o.w = eamd_act(v.w * sc.w + sh.w, p.a_act);
    *reinterpret_cast<float4*>(&xs[row * LD + c4 * 4]) = o;
    if (p.a_out && m0 + row < p.M) *reinterpret_cast<float4*>(p.a_out + gr * D + c4 * 4) = o;
  }
}

template <bool LNB>
__global__ __launch_bounds__(RNT, 2) void rowproj_f32_kernel(const eamd_rowproj_t p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = p.K, N = p.N;
  const int LD = xs_ld(K);
  float* const xs = sm;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * RBM;
  const int S = K >> 5;                 // K-steps of 32 per chunk
  const int nchunk = N / RCH;
  const int nsteps = S * nchunk;
  // wave-private result tile (LNB: the whole-row tile lives over the input rows instead)
  float* const st = sm + RBM * LD + wave * ST_SZ;

  // ---- the 32 input rows -> LDS ----
  if (p.ln_x) {
    stage_ln(p, m0, t, xs, LD);
  } else if (p.a_scale) {
    stage_affine(p, m0, t, xs, LD);
  } else {
    const int c4n = K >> 2;             // 16-byte pieces per row
    for (int idx = t; idx < RBM * c4n; idx += RNT) {
      const int row = idx / c4n, c4 = idx - row * c4n;
      *reinterpret_cast<f32x4*>(&xs[row * LD + c4 * 4]) =
          *reinterpret_cast<const f32x4*>(p.a + (long)min(m0 + row, p.M - 1) * p.lda + c4 * 4);
    }
  }

  // ---- weights: packed image[step g][wave][v = q*2 + j][lane] (float4 over 4 k), ring of four sets ----
  const char* __restrict__ Wb = reinterpret_cast<const char*>(p.w) + (long)wave * 4096 + lane * 16;
  f32x4 bs[4][4];
  auto load_b = [&](auto set_c, int g) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const char* base = Wb + (long)min(g, nsteps - 1) * (8 * 4096);
#pragma unroll
    for (int v = 0; v < 4; ++v) bs[SET][v] = *reinterpret_cast<const f32x4*>(base + v * 1024);
  };
  float fA[2][2][4];
  auto read_a = [&](int kcol, auto half_c) __attribute__((always_inline)) {
    constexpr int hh = decltype(half_c)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(&xs[(i * 16 + fr) * LD + kcol + hh * 16 + fq * 4]);
      fA[hh][i][0] = v.x; fA[hh][i][1] = v.y; fA[hh][i][2] = v.z; fA[hh][i][3] = v.w;
    }
  };
  f32x4 acc[2][2];
  auto mfma_half = [&](auto set_c, auto half_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value, q = decltype(half_c)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fA[q][i][e], bs[SET][q * 2 + j][e], acc[i][j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // one K-step: g = global step (ring position), kcol / kcol_next = LDS column of this / the next step's rows
  auto step = [&](auto s_c, int g, int kcol, int kcol_next) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value;
    load_b(std::integral_constant<int, (s + 3) & 3>{}, g + 3);
    __builtin_amdgcn_sched_barrier(0);
    read_a(kcol, I1{});
    mfma_half(std::integral_constant<int, s & 3>{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_a(kcol_next, I0{});
    mfma_half(std::integral_constant<int, s & 3>{}, I1{});
  };

  load_b(std::integral_constant<int, 0>{}, 0);
  load_b(std::integral_constant<int, 1>{}, 1);
  load_b(std::integral_constant<int, 2>{}, 2);
  __syncthreads();                      // the staged rows are complete
  read_a(0, I0{});

  const unsigned thr_out = eamd_drop_thr16(p.p_out);
  const float inv_out = eamd_drop_inv(thr_out);
  const unsigned seed_out = p.p_out > 0.f ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_out) : 0u;

  int g = 0;
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kp = 0; kp < S; kp += 4) {               // S is a multiple of 8 (K % 256 == 0)
      const int k0 = kp * 32;
      step(std::integral_constant<int, 0>{}, g + 0, k0, k0 + 32);
      step(std::integral_constant<int, 1>{}, g + 1, k0 + 32, k0 + 64);
      step(std::integral_constant<int, 2>{}, g + 2, k0 + 64, k0 + 96);
      step(std::integral_constant<int, 3>{}, g + 3, k0 + 96, (kp + 4 < S) ? k0 + 128 : 0);
      g += 4;
    }
    if constexpr (!LNB) {
      // ---- chunk epilogue: accumulators -> the wave's LDS tile -> 16-byte row pieces with bias / dropout / alpha / residual ----
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float2*>(&st[(i * 16 + fq * 4 + r) * ST_LD + 2 * fr]) = make_float2(acc[i][0][r], acc[i][1][r]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int col0 = c * RCH + wave * 32 + (lane & 7) * 4;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + col0);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int lr = it * 8 + (lane >> 3);
        const int row = m0 + lr;
        const float4 a4 = *reinterpret_cast<const float4*>(&st[lr * ST_LD + (lane & 7) * 4]);
        if (row < p.M) {
          float v[4] = {a4.x + b4.x, a4.y + b4.y, a4.z + b4.z, a4.w + b4.w};
          const long gi = (long)row * N + col0;
          if (p.p_out > 0.f) {
            bool keep[4];
            eamd_drop_keep4(seed_out, (unsigned long long)gi, thr_out, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * inv_out : 0.f;
          }
          float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.R) r4 = *reinterpret_cast<const float4*>(p.R + (long)row * p.ldr + col0);
          *reinterpret_cast<float4*>(p.out + (long)row * p.ldo + col0) =
              make_float4(v[0] * p.alpha + r4.x, v[1] * p.alpha + r4.y, v[2] * p.alpha + r4.z, v[3] * p.alpha + r4.w);
        }
      }
      __builtin_amdgcn_wave_barrier();     // the tile is free again before the next chunk's accumulators land in it
    }
  }

  if constexpr (LNB) {
    // ---- LayerNorm backward over the finished rows (N = 256: one chunk).  reference: transformer/layer_norm.py:12-38 ----
    constexpr int D = 256;
    float* const tl = sm;                             // whole-row tile over the input rows (nobody reads those any more)
    float* const gs = sm + RBM * TL_LD;               // [32][256] d gamma contributions, then [32][256] d beta contributions
    float* const bsum = gs + RBM * D;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float2*>(&tl[(i * 16 + fq * 4 + r) * TL_LD + wave * 32 + 2 * fr]) = make_float2(acc[i][0][r], acc[i][1][r]);
    __syncthreads();
    const EamdLnbArgs la{p.lnb_x, p.lnb_gamma, p.lnb_mean, p.lnb_rstd, p.lnb_dres, p.lnb_ws, p.lnb_drop_out, p.lnb_drop_p,
                         (unsigned long long)p.lnb_drop_salt, p.drop_step, p.out, (long)p.ldo, p.M};
    eamd_ln_bwd_rows32<TL_LD>(la, tl, gs, bsum, m0, t, (int)blockIdx.x);
  }
}

// Packed image of one product: image[chunk c][step s][wave w][v = q*2 + j][lane] = float4 over e of
//     B[k = s*32 + q*16 + fq*4 + e][col = c*256 + w*32 + 2 fr + j],   B[k][col] = trans ? W[k * ldw + col] : W[col * ldw + k]
// (trans = 0: y = x W^T with W [N, K] as nn.Linear stores it; trans = 1: dx = dy W with W [K = rows, N = columns]).
struct PackJob { const float* w; float* img; int K, N, ldw, trans; long npiece; };
constexpr int PACK_JOBS = 48;
struct PackTable { PackJob j[PACK_JOBS]; };
__global__ __launch_bounds__(256) void rowproj_pack_kernel(const PackTable tab) {
  const PackJob jb = tab.j[blockIdx.y];
  for (long piece = (long)blockIdx.x * 256 + threadIdx.x; piece < jb.npiece; piece += (long)gridDim.x * 256) {
    const int lane = piece & 63, v = (piece >> 6) & 3, w = (piece >> 8) & 7;
    const long gs_ = piece >> 11;                     // c * S + s
    const int S = jb.K >> 5;
    const int c = (int)(gs_ / S), s = (int)(gs_ - (long)c * S);
    const int fr = lane & 15, fq = lane >> 4, q = v >> 1, j = v & 1;
    const int col = c * RCH + w * 32 + 2 * fr + j, k = s * 32 + q * 16 + fq * 4;
    float4 o;
    if (jb.trans) {
      const float* r0 = jb.w + (long)k * jb.ldw + col;
      o = make_float4(r0[0], r0[jb.ldw], r0[2L * jb.ldw], r0[3L * jb.ldw]);
    } else {
      o = *reinterpret_cast<const float4*>(jb.w + (long)col * jb.ldw + k);
    }
    *reinterpret_cast<float4*>(jb.img + piece * 4) = o;
  }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

size_t smem_bytes(int K, bool lnb) {
  size_t rows = (size_t)RBM * (K + 8);
  if (lnb) {
    const size_t need = (size_t)RBM * TL_LD + 2 * (size_t)RBM * 256;
    return (rows > need ? rows : need) * sizeof(float);
  }
  return (rows + 8 * (size_t)ST_SZ) * sizeof(float);
}

int check_rowproj(const eamd_rowproj_t* p) {
  if (!p || !p->a || !p->w || !p->out) return EAMD_EINVAL;
  if (p->M <= 0 || p->K <= 0 || p->N <= 0) return EAMD_EINVAL;
  if (p->K % 256 != 0 || p->N % RCH != 0 || p->K > 768) return EAMD_EUNSUPPORTED;
  if (p->lda < p->K || p->ldo < p->N || (p->R && p->ldr < p->N)) return EAMD_EINVAL;
  if (p->lda % 4 != 0 || p->ldo % 4 != 0 || (p->R && p->ldr % 4 != 0)) return EAMD_EUNSUPPORTED;
  if (!al16(p->a) || !al16(p->w) || !al16(p->out) || (p->R && !al16(p->R)) || (p->bias && !al16(p->bias))) return EAMD_EUNSUPPORTED;
  if (p->p_out < 0.f || p->p_out >= 1.f) return EAMD_EINVAL;
  if (p->p_out > 0.f && (!p->drop_step || p->ldo != p->N)) return EAMD_EINVAL;       // the mask is that of the contiguous [M, N] tensor
  if (p->ln_x) {
    if (p->K != 256 || p->a_scale || !p->ln_w || !p->ln_b || !p->ln_mean || !p->ln_rstd || p->lda != 256) return EAMD_EINVAL;
    if (!al16(p->ln_x) || !al16(p->ln_w) || !al16(p->ln_b)) return EAMD_EUNSUPPORTED;
  }
  if (p->a_scale) {
    if (p->K != 256 || !p->a_shift) return EAMD_EINVAL;
    if (!al16(p->a_scale) || !al16(p->a_shift) || (p->a_out && !al16(p->a_out))) return EAMD_EUNSUPPORTED;
  }
  if (p->lnb_x) {
    if (p->N != 256 || !p->lnb_gamma || !p->lnb_mean || !p->lnb_rstd || !p->lnb_ws || p->R || p->p_out > 0.f) return EAMD_EINVAL;
    if (p->lnb_drop_out && (!p->drop_step || p->lnb_drop_p < 0.f || p->lnb_drop_p >= 1.f)) return EAMD_EINVAL;
    if (!al16(p->lnb_x) || !al16(p->lnb_gamma) || (p->lnb_dres && !al16(p->lnb_dres)) || (p->lnb_drop_out && !al16(p->lnb_drop_out)))
      return EAMD_EUNSUPPORTED;
  }
  return EAMD_OK;
}

template <bool LNB>
int launch_rowproj(const eamd_rowproj_t& p, hipStream_t stream) {
  const size_t smem = smem_bytes(p.K, LNB);
  static int attr_done = 0;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowproj_f32_kernel<LNB>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes(768, LNB));
    if (e != hipSuccess) return (int)e;
    attr_done = 1;
  }
  hipLaunchKernelGGL((rowproj_f32_kernel<LNB>), dim3((p.M + RBM - 1) / RBM), dim3(RNT), smem, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

}  // namespace

extern "C" int64_t eamd_rowproj_lnb_workspace(int M) { return (int64_t)((M + RBM - 1) / RBM) * 2 * 256; }

extern "C" int eamd_rowproj(const eamd_rowproj_t* p, void* stream) {
  const int rc = check_rowproj(p);
  if (rc != EAMD_OK) return rc;
  return p->lnb_x ? launch_rowproj<true>(*p, (hipStream_t)stream) : launch_rowproj<false>(*p, (hipStream_t)stream);
}

extern "C" int eamd_rowproj_pack_f32(const eamd_rowproj_pack_t* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return EAMD_EINVAL;
  for (int i0 = 0; i0 < njobs; i0 += PACK_JOBS) {
    const int n = njobs - i0 < PACK_JOBS ? njobs - i0 : PACK_JOBS;
    PackTable tab;
    long most = 0;
    for (int i = 0; i < n; ++i) {
      const eamd_rowproj_pack_t& q = jobs[i0 + i];
      if (!q.w || !q.image || q.K <= 0 || q.N <= 0 || q.ldw <= 0) return EAMD_EINVAL;
      if (q.K % 256 != 0 || q.N % RCH != 0 || !al16(q.image)) return EAMD_EUNSUPPORTED;
      if (!q.trans && (!al16(q.w) || q.ldw % 4 != 0)) return EAMD_EUNSUPPORTED;
      if (q.ldw < (q.trans ? q.N : q.K)) return EAMD_EINVAL;
      tab.j[i] = PackJob{q.w, q.image, q.K, q.N, q.ldw, q.trans, (long)q.K * q.N / 4};
      if (tab.j[i].npiece > most) most = tab.j[i].npiece;
    }
    const unsigned bx = (unsigned)((most + 255) / 256 < 256 ? (most + 255) / 256 : 256);
    hipLaunchKernelGGL(rowproj_pack_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, tab);
    EAMD_LAUNCH_CHECK();
  }
  return EAMD_OK;
}
