// Fused position-wise feed-forward kernels for gfx950, fp32 operands (v_mfma_f32_16x16x4_f32: exact fp32 products).
//
// reference: espnet/nets/pytorch_backend/transformer/positionwise_feed_forward.py:12-32
//     forward   out = R + alpha * drop_out( drop_in(act(x W1^T + b1)) W2^T + b2 )
//     backward  dz = alpha * (dy W2) (.) f,   dx = dz W1          (f = mask / (1 - p) * act'(x W1^T + b1), kept by forward)
//
// Why one kernel: as two GEMMs the pair is a [M, 2048] x K = 256 product whose 65 MB result (x 2 outputs) is written
// and read back, and a [M, 256] x K = 2048 product that has only M / 64 x 4 = 500 tiles for 256 CUs; each launch pays
// its own prologue / epilogue / tail.  Both products are local to a block of ROWS, so one workgroup can take 32 rows
// through both: the hidden units never make a round trip for the second product, the skinny product disappears into a
// loop with no launch boundary, and M / 32 = 249 workgroups fill the chip for the whole launch.
//
// Structure (512 threads = 8 waves, one workgroup per CU, two waves per SIMD):
//   * waves 0-3 ("up") form z = x W1^T chunk by chunk of 128 hidden units (wave tile 32 x 32, K = 256 in 8 steps of 32),
//     apply bias / activation / dropout to the accumulators, leave h (and f) in global memory for backward and h in LDS;
//   * waves 4-7 ("down") accumulate out[32, 256] += h_chunk W2[:, chunk]^T one chunk behind (wave tile 32 x 64, 8 steps
//     of 16 hidden units): every SIMD holds one up and one down wave, so the up wave's epilogue arithmetic runs beside
//     the down wave's MFMAs, and both roles issue 32 MFMAs per step = the SIMD's matrix pipe never changes hands idle;
//   * all 8 waves stage the weight tiles of the next steps global -> VGPR -> LDS (clamped, branch-free 16-byte loads two
//     steps ahead of their use); the 32 input rows stay in LDS for the whole launch; one barrier per step.
// The backward kernel is the same skeleton with k-strided weight tiles (dh = dy W2 walks W2's rows, dx = dz W1 walks
// W1's rows; LDS images [k][cols + 4] read with ds_read_b32, conflict-free for k = 4 fq + e) and the factor f read at
// the accumulator positions.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"

namespace {

constexpr int FBM = 32;          // rows per workgroup
constexpr int FD = 256;          // model width (template constant of this kernel)
constexpr int FHC = 128;         // hidden units per chunk
constexpr int FNT = 512;
constexpr int XS_LD = 264;       // [32][256 + 8]: ds_read_b128 conflict-free (row stride = 2 mod 16 chunks)
constexpr int HS_LD = 136;       // [32][128 + 8]
constexpr int W1F_LD = 40;       // forward: [128 hidden][32 k + 8]
constexpr int W2F_LD = 24;       // forward: [256 outs][16 k + 8]
constexpr int W1B_LD = 132;      // backward: [32 k][128 hidden + 4]
constexpr int W2B_LD = 260;      // backward: [16 k][256 cols + 4]
constexpr int XS_SZ = FBM * XS_LD;                 // 8448 floats
constexpr int W1S_SZ = 128 * W1F_LD;               // 5120 (>= 32 * 132 = 4224)
constexpr int W2S_SZ = 256 * W2F_LD;               // 6144 (>= 16 * 260 = 4160)
constexpr int HS_SZ = FBM * HS_LD;                 // 4352
constexpr int FFN_SMEM_FLOATS = XS_SZ + 2 * W1S_SZ + 2 * W2S_SZ + 2 * HS_SZ;   // 39680 floats = 158720 B
static_assert(32 * W1B_LD <= W1S_SZ && 16 * W2B_LD <= W2S_SZ, "backward tile images fit the forward buffers");
static_assert(FFN_SMEM_FLOATS * 4 <= 160 * 1024, "LDS budget of one CU");

template <bool BWD, int ACT>
__global__ __launch_bounds__(FNT, 2) void ffn_f32_kernel(const eamd_ffn_t p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const xs = sm;
  float* const w1s = xs + XS_SZ;
  float* const w2s = w1s + 2 * W1S_SZ;
  float* const hs = w2s + 2 * W2S_SZ;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const bool up = wave < 4;            // wave-uniform role
  const int wq = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * FBM;
  const int F = p.F;
  const int nch = F / FHC;
  const float* __restrict__ X = p.x;
  // "first" weight = B operand of the up product, "second" = of the down product
  const float* __restrict__ Wa = BWD ? p.w2 : p.w1;
  const float* __restrict__ Wb = BWD ? p.w1 : p.w2;

  // ---- the 32 input rows (rows past M clamped: they only feed outputs that are never stored) ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + FNT * i, row = idx >> 6, c4 = idx & 63;
    *reinterpret_cast<f32x4*>(&xs[row * XS_LD + c4 * 4]) =
        *reinterpret_cast<const f32x4*>(X + (long)min(m0 + row, p.M - 1) * FD + c4 * 4);
  }

  // ---- weight-tile staging ----
  // up tile g = (chunk c = g / 8, step s = g % 8); down pair P = (chunk cd = P / 4, pair sp = P % 4) serves down steps 2P, 2P+1
  f32x4 r1[2][2], r2[2][4];      // native vectors: a float4 struct copied global -> array -> LDS becomes two memcpys that keep the array in scratch
  // per-thread byte offsets inside a tile (the tile's origin is wave-uniform: scalar base + 32-bit vector offset)
  const unsigned w1_toff = BWD ? (unsigned)(((t >> 5) * F + (t & 31) * 4) * 4) : (unsigned)(((t >> 3) * FD + (t & 7) * 4) * 4);
  const unsigned w2_toff = BWD ? (unsigned)(((t >> 6) * FD + (t & 63) * 4) * 4) : (unsigned)(((t >> 3) * F + (t & 7) * 4) * 4);
  auto load_w1 = [&](auto set_c, int g) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int c = min(g >> 3, nch - 1), s = g & 7;
    // forward: W1[c*128 + row][s*32 + kc*4], row = idx >> 3 (i adds 64 rows);  backward: W2[s*32 + row][c*128 + cc*4], row = idx >> 5 (i adds 16)
    const char* base = reinterpret_cast<const char*>(BWD ? Wa + (long)(s * 32) * F + c * FHC : Wa + (long)(c * FHC) * FD + s * 32);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* bi = base + (BWD ? (long)i * 16 * F * 4 : (long)i * 64 * FD * 4);
      r1[SET][i] = *reinterpret_cast<const f32x4*>(bi + w1_toff);
    }
  };
  auto store_w1 = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    float* dst = w1s + buf * W1S_SZ;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + FNT * i;
      if constexpr (!BWD) *reinterpret_cast<f32x4*>(&dst[(idx >> 3) * W1F_LD + (idx & 7) * 4]) = r1[SET][i];
      else *reinterpret_cast<f32x4*>(&dst[(idx >> 5) * W1B_LD + (idx & 31) * 4]) = r1[SET][i];
    }
  };
  auto load_w2 = [&](auto set_c, int P) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int Pc = min(P, 4 * nch - 1);
    const int cd = Pc >> 2, sp = Pc & 3;
    // forward: W2[row][cd*128 + sp*32 + kc*4], row = idx >> 3 (i adds 64 rows);  backward: W1[cd*128 + sp*32 + row][cc*4], row = idx >> 6 (i adds 8)
    const char* base = reinterpret_cast<const char*>(BWD ? Wb + (long)(cd * FHC + sp * 32) * FD : Wb + cd * FHC + sp * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* bi = base + (BWD ? (long)i * 8 * FD * 4 : (long)i * 64 * F * 4);
      r2[SET][i] = *reinterpret_cast<const f32x4*>(bi + w2_toff);
    }
  };
  // half = 0 / 1: the 16 reduction elements of the pair tile that the next down step contracts
  auto store_w2 = [&](auto set_c, int half, int buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    float* dst = w2s + buf * W2S_SZ;
    if constexpr (!BWD) {
      if (((t >> 2) & 1) == half) {      // this thread's chunks (kc = t & 7) belong to half kc >> 2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = t + FNT * i;
          *reinterpret_cast<f32x4*>(&dst[(idx >> 3) * W2F_LD + (idx & 3) * 4]) = r2[SET][i];
        }
      }
    } else {                             // rows 0-15 of the pair tile are chunks i = 0, 1; rows 16-31 chunks 2, 3
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = t + FNT * i;     // row within the half = idx >> 6 (0..15)
        const f32x4 v = half ? r2[SET][i + 2] : r2[SET][i];
        *reinterpret_cast<f32x4*>(&dst[(idx >> 6) * W2B_LD + (idx & 63) * 4]) = v;
      }
    }
  };

  f32x4 zacc[2][2], yacc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // dropout constants of the hidden units (forward)
  const unsigned thr_in = eamd_drop_thr16(p.p_in);
  const float inv_in = p.p_in > 0.f ? eamd_drop_inv(thr_in) : 1.f;
  const unsigned seed_in = (!BWD && p.p_in > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_in) : 0u;
  float fpre[2][2][4];     // backward: the factor f at this lane's accumulator positions (requested at the chunk's first step)

  // ---- one step ----
  auto up_mfma = [&](int s, int buf) __attribute__((always_inline)) {
    const float* wb = w1s + buf * W1S_SZ;
    float4 xa[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        xa[i][q] = *reinterpret_cast<const float4*>(&xs[(i * 16 + fr) * XS_LD + s * 32 + q * 16 + fq * 4]);
    if constexpr (!BWD) {
      float4 wv[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          wv[j][q] = *reinterpret_cast<const float4*>(&wb[(wq * 32 + j * 16 + fr) * W1F_LD + q * 16 + fq * 4]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              zacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][q][e], wv[j][q][e], zacc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float wv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[j][e] = wb[(q * 16 + fq * 4 + e) * W1B_LD + wq * 32 + j * 16 + fr];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              zacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][q][e], wv[j][e], zacc[i][j], 0, 0, 0);
      }
    }
  };
  auto down_mfma = [&](int s, int buf, int hb) __attribute__((always_inline)) {
    const float* wb = w2s + buf * W2S_SZ;
    const float* hp = hs + hb * HS_SZ;
    float4 ha[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ha[i] = *reinterpret_cast<const float4*>(&hp[(i * 16 + fr) * HS_LD + s * 16 + fq * 4]);
    if constexpr (!BWD) {
      float4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const float4*>(&wb[(wq * 64 + j * 16 + fr) * W2F_LD + fq * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            yacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[i][e], wv[j][e], yacc[i][j], 0, 0, 0);
    } else {
      float wv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) wv[j][e] = wb[(fq * 4 + e) * W2B_LD + wq * 64 + j * 16 + fr];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            yacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[i][e], wv[j][e], yacc[i][j], 0, 0, 0);
    }
  };
  // the up waves' chunk epilogue: accumulators -> hidden units (LDS for the down waves, global memory for backward).
  // Straight-line code: the activation is a template constant, the row guard is taken only by the ragged last workgroup.
  const bool full_rows = m0 + FBM <= p.M;
  const unsigned e_toff = (unsigned)(((fq * 4) * F + wq * 32 + fr) * 4);      // this lane's first accumulator element
  auto up_epilogue = [&](int c) __attribute__((always_inline)) {
    float* hp = hs + (c & 1) * HS_SZ;
    float hv[2][2][4], fv[2][2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lc = wq * 32 + j * 16 + fr;        // column inside the chunk
      float bj = 0.f;
      if constexpr (!BWD) bj = p.b1 ? p.b1[c * FHC + lc] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (!BWD) {
            float a, d;
            eamd_act_dact(zacc[i][j][r] + bj, ACT, a, d);
            hv[i][j][r] = a; fv[i][j][r] = d;
          } else {
            hv[i][j][r] = (zacc[i][j][r] * fpre[i][j][r]) * p.alpha;
          }
        }
      }
    }
    if constexpr (!BWD) {
      if (p.p_in > 0.f) {       // wave-uniform
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned gi = (unsigned)(m0 + i * 16 + fq * 4 + r) * (unsigned)F + (unsigned)(c * FHC + wq * 32 + j * 16 + fr);
              const bool keep = eamd_drop_keep(seed_in, (unsigned long long)gi, thr_in);
              hv[i][j][r] = keep ? hv[i][j][r] * inv_in : 0.f;
              fv[i][j][r] = keep ? fv[i][j][r] * inv_in : 0.f;
            }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[(i * 16 + fq * 4 + r) * HS_LD + wq * 32 + j * 16 + fr] = hv[i][j][r];
    // global copies for backward: element (i, j, r) sits (i*16 + r) rows and j*16 columns from the lane's first one
    auto put = [&](float* dstp, const float (&val)[2][2][4]) __attribute__((always_inline)) {
      char* base = reinterpret_cast<char*>(dstp + (long)m0 * F + c * FHC);
      if (full_rows) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<float*>(base + ((long)(i * 16 + r) * F + j * 16) * 4 + e_toff) = val[i][j][r];
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (m0 + i * 16 + fq * 4 + r < p.M)
                *reinterpret_cast<float*>(base + ((long)(i * 16 + r) * F + j * 16) * 4 + e_toff) = val[i][j][r];
      }
    };
    if (p.h) put(p.h, hv);
    if constexpr (!BWD) { if (p.f) put(p.f, fv); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto load_f = [&](int c) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(p.f + (long)m0 * F + c * FHC);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // rows past M: any valid address (their products only reach rows that are never stored)
          const long ro = full_rows ? (long)(i * 16 + r) * F : (long)(min(m0 + i * 16 + fq * 4 + r, p.M - 1) - m0 - fq * 4) * F;
          fpre[i][j][r] = *reinterpret_cast<const float*>(base + (ro + j * 16) * 4 + e_toff);
        }
  };

  // step g = 8 c + s.  UP: chunk c < nch is being formed; DOWN: chunk c - 1 is being contracted.
  auto step = [&](auto s_c, auto up_c, auto down_c, int c) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value;
    constexpr bool UP = decltype(up_c)::value, DOWN = decltype(down_c)::value;
    const int g = 8 * c + s;
    constexpr int buf = s & 1;
    // requests: up tile g + 2 (same register set as tile g, stored a step ago), down pair for steps g + 2, g + 3
    if constexpr (UP) load_w1(std::integral_constant<int, s & 1>{}, g + 2);
    if constexpr ((DOWN || s >= 6) && (s % 2 == 0)) load_w2(std::integral_constant<int, ((s + 2) / 2) & 1>{}, (g + 2 - 8) >> 1);
    if constexpr (BWD && UP && s == 0) { if (up) load_f(c); }
    __builtin_amdgcn_sched_barrier(0);
    if (up) {
      if constexpr (UP) up_mfma(s, buf);
    } else {
      if constexpr (DOWN) down_mfma(s, buf, (c - 1) & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the tiles of step g + 1 (their buffers were last read in step g - 1)
    if constexpr (UP) { if (g + 1 < 8 * nch) store_w1(std::integral_constant<int, (s + 1) & 1>{}, buf ^ 1); }
    if constexpr (DOWN || s == 7) {
      // down step gd = g + 1 - 8, pair gd / 2 (register set (gd / 2) & 1), half gd & 1
      constexpr int gdl = (s + 1) & 7;          // gd mod 8 (chunks are 8 steps: the pair parity repeats per chunk)
      if (g + 1 - 8 < 8 * nch) store_w2(std::integral_constant<int, (gdl / 2) & 1>{}, gdl & 1, buf ^ 1);
    }
    if constexpr (UP && s == 7) { if (up) up_epilogue(c); }
    __syncthreads();
  };
  auto chunk = [&](auto up_c, auto down_c, int c) __attribute__((always_inline)) {
    step(std::integral_constant<int, 0>{}, up_c, down_c, c);
    step(std::integral_constant<int, 1>{}, up_c, down_c, c);
    step(std::integral_constant<int, 2>{}, up_c, down_c, c);
    step(std::integral_constant<int, 3>{}, up_c, down_c, c);
    step(std::integral_constant<int, 4>{}, up_c, down_c, c);
    step(std::integral_constant<int, 5>{}, up_c, down_c, c);
    step(std::integral_constant<int, 6>{}, up_c, down_c, c);
    step(std::integral_constant<int, 7>{}, up_c, down_c, c);
  };

  // prologue: up tiles 0 (-> LDS) and 1 (in flight)
  load_w1(std::integral_constant<int, 0>{}, 0);
  load_w1(std::integral_constant<int, 1>{}, 1);
  store_w1(std::integral_constant<int, 0>{}, 0);
  __syncthreads();

  using T_ = std::true_type;
  using F_ = std::false_type;
  chunk(T_{}, F_{}, 0);
  for (int c = 1; c < nch; ++c) chunk(T_{}, T_{}, c);
  chunk(F_{}, T_{}, nch);

  // ---- output rows: accumulators of the down waves -> LDS (over the input rows) -> 16-byte row stores ----
  if (!up) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[(i * 16 + fq * 4 + r) * XS_LD + wq * 64 + j * 16 + fr] = yacc[i][j][r];
  }
  __syncthreads();
  const unsigned thr_out = eamd_drop_thr16(p.p_out);
  const float inv_out = eamd_drop_inv(thr_out);
  const unsigned seed_out = (!BWD && p.p_out > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_out) : 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + FNT * i, lr = idx >> 6, c4 = idx & 63;
    const int row = m0 + lr;
    if (row >= p.M) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(&xs[lr * XS_LD + c4 * 4]);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const long gi = (long)row * FD + c4 * 4;
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

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <bool BWD, int ACT>
int launch_ffn(const eamd_ffn_t& p, hipStream_t stream) {
  constexpr size_t smem = (size_t)FFN_SMEM_FLOATS * sizeof(float);
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_f32_kernel<BWD, ACT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_err != hipSuccess) return (int)attr_err;
  const int nblk = (p.M + FBM - 1) / FBM;
  hipLaunchKernelGGL((ffn_f32_kernel<BWD, ACT>), dim3(nblk), dim3(FNT), smem, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int check_ffn(const eamd_ffn_t* p, bool bwd) {
  if (!p || !p->x || !p->w1 || !p->w2 || !p->out) return EAMD_EINVAL;
  if (p->M <= 0 || p->D <= 0 || p->F <= 0) return EAMD_EINVAL;
  if (p->dtype != 0) return EAMD_EUNSUPPORTED;                       // fp32 operands (the bf16 twin: not built)
  if (p->D != FD || p->F % FHC != 0 || p->F < 2 * FHC) return EAMD_EUNSUPPORTED;
  if ((long)p->M * p->F >= (1L << 31)) return EAMD_EUNSUPPORTED;     // 32-bit dropout pair index space
  if (!al16(p->x) || !al16(p->w1) || !al16(p->w2) || !al16(p->out) || (p->R && !al16(p->R)) || (p->b2 && !al16(p->b2)))
    return EAMD_EUNSUPPORTED;
  if (bwd) {
    if (!p->f) return EAMD_EINVAL;
  } else {
    if (p->p_in < 0.f || p->p_in >= 1.f || p->p_out < 0.f || p->p_out >= 1.f) return EAMD_EINVAL;
    if ((p->p_in > 0.f || p->p_out > 0.f) && !p->drop_step) return EAMD_EINVAL;
    if (p->act != EAMD_ACT_RELU && p->act != EAMD_ACT_SWISH) return EAMD_EUNSUPPORTED;
  }
  return EAMD_OK;
}

}  // namespace

extern "C" int eamd_ffn_fwd(const eamd_ffn_t* p, void* stream) {
  const int rc = check_ffn(p, false);
  if (rc != EAMD_OK) return rc;
  return p->act == EAMD_ACT_SWISH ? launch_ffn<false, EAMD_ACT_SWISH>(*p, (hipStream_t)stream)
                                  : launch_ffn<false, EAMD_ACT_RELU>(*p, (hipStream_t)stream);
}

extern "C" int eamd_ffn_bwd(const eamd_ffn_t* p, void* stream) {
  const int rc = check_ffn(p, true);
  if (rc != EAMD_OK) return rc;
  return launch_ffn<true, EAMD_ACT_NONE>(*p, (hipStream_t)stream);
}
