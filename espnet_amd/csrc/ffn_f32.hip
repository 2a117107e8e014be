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

#ifdef FFN_STAMP
// diagnostic build only: s_memtime stamps of one up and one down wave of workgroup 100 during body 4
__device__ unsigned long long ffn_stamps[2][2][8][8];
#define STAMP(k)                                                                                         \
  do {                                                                                                   \
    if (blockIdx.x == 100 && c == 4 && (wave == 0 || wave == 4) && lane == 0)                            \
      ffn_stamps[BWD][ROLE][s][k] = __builtin_amdgcn_s_memtime();                                        \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

constexpr int FBM = 32;          // rows per workgroup
constexpr int FD = 256;          // model width (template constant of this kernel)
constexpr int FHC = 128;         // hidden units per chunk
constexpr int FNT = 512;
constexpr int FLAG_LAG = 12;      // steps the down waves run behind the up waves (8 = one chunk, + 2 epilogue steps, + 2)
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
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);     // scalar: role branches are s_cbranch, not exec masks
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
  f32x4 r1[2][2], r2[2][2];      // native vectors: a float4 struct copied global -> array -> LDS becomes two memcpys that keep the array in scratch
  // per-thread byte offsets inside a tile (the tile's origin is wave-uniform: scalar base + 32-bit vector offset)
  const unsigned w1_toff = BWD ? (unsigned)(((t >> 5) * F + (t & 31) * 4) * 4) : (unsigned)(((t >> 3) * FD + (t & 7) * 4) * 4);
  const unsigned w2_toff = BWD ? (unsigned)(((t >> 6) * FD + (t & 63) * 4) * 4) : (unsigned)(((t >> 2) * F + (t & 3) * 4) * 4);
  auto load_w1 = [&](auto set_c, int g) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int c = min(g >> 3, nch - 1), s = g & 7;
    // forward: W1[c*128 + row][s*32 + kc*4], row = idx >> 3 (i adds 64 rows);  backward: W2[s*32 + row][c*128 + cc*4], row = idx >> 5 (i adds 16)
    const char* base = reinterpret_cast<const char*>(BWD ? Wa + (long)(s * 32) * F + c * FHC : Wa + (long)(c * FHC) * FD + s * 32);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* bi = base + (BWD ? (long)i * 16 * F * 4 : (long)i * 64 * FD * 4);
      if (p.reserved & 1) r1[SET][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wa) + ((g & 7) * 2 + i) * 8192 + t * 16);   // timing experiment
      else r1[SET][i] = *reinterpret_cast<const f32x4*>(bi + w1_toff);
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
  // down tile gd = (chunk cd = gd / 8, step sd = gd % 8): the 16 hidden units that down step contracts
  auto load_w2 = [&](auto set_c, int gd) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int gc = min(max(gd, 0), 8 * nch - 1);
    const int cd = gc >> 3, sd = gc & 7;
    // forward: W2[row][cd*128 + sd*16 + kc*4], row = idx >> 2 (i adds 128 rows);  backward: W1[cd*128 + sd*16 + row][cc*4], row = idx >> 6 (i adds 8)
    const char* base = reinterpret_cast<const char*>(BWD ? Wb + (long)(cd * FHC + sd * 16) * FD : Wb + cd * FHC + sd * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* bi = base + (BWD ? (long)i * 8 * FD * 4 : (long)i * 128 * F * 4);
      if (p.reserved & 1) r2[SET][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Wb) + ((gc & 7) * 2 + i) * 8192 + t * 16);  // timing experiment
      else r2[SET][i] = *reinterpret_cast<const f32x4*>(bi + w2_toff);
    }
  };
  auto store_w2 = [&](auto set_c, int buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    float* dst = w2s + buf * W2S_SZ;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + FNT * i;
      if constexpr (!BWD) *reinterpret_cast<f32x4*>(&dst[(idx >> 2) * W2F_LD + (idx & 3) * 4]) = r2[SET][i];
      else *reinterpret_cast<f32x4*>(&dst[(idx >> 6) * W2B_LD + (idx & 63) * 4]) = r2[SET][i];
    }
  };

  f32x4 zacc[2][2], zold[2][2], yacc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; zold[i][j] = zacc[i][j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // dropout constants of the hidden units (forward)
  const unsigned thr_in = eamd_drop_thr16(p.p_in);
  const float inv_in = p.p_in > 0.f ? eamd_drop_inv(thr_in) : 1.f;
  const unsigned seed_in = (!BWD && p.p_in > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_in) : 0u;
  float bpre[2] = {0.f, 0.f};   // forward: b1 at this lane's two columns of the chunk in flight (requested mid-chunk: a load consumed
                                // at once would wait for every tile request in front of it in the vmcnt queue)
  float fpre[2][2][4];     // backward: the factor f at this lane's accumulator positions (requested mid-chunk)
  const bool full_rows = m0 + FBM <= p.M;
  const unsigned e_toff = (unsigned)(((fq * 4) * F + wq * 32 + fr) * 4);      // this lane's first accumulator element

  // ---- fragments: a step's 32 MFMAs per wave are two halves of 16; each half has its own fragment registers, so the
  // reads of one half fly under the MFMAs of the other and the first half of step g + 1 is read right behind the barrier
  // that ends step g, in front of the second half of step g.
  //   up wave:   half q = reduction elements q*16 .. q*16+15 of the step's 32 (A: 2 row tiles, B: 2 column tiles)
  //   down wave: half hh = output column tiles 2 hh, 2 hh + 1 (A: the step's 16 hidden units, read for both halves)
  float fA[2][2][4], fB[2][2][4];
  auto read_half = [&](auto role_c, auto s_c, auto half_c, int hbuf) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value, hh = decltype(half_c)::value;
    constexpr int buf = s & 1;
    if constexpr (decltype(role_c)::value == 0) {
      const float* wb = w1s + buf * W1S_SZ;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&xs[(i * 16 + fr) * XS_LD + s * 32 + hh * 16 + fq * 4]);
        fA[hh][i][0] = v.x; fA[hh][i][1] = v.y; fA[hh][i][2] = v.z; fA[hh][i][3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (!BWD) {
          const float4 v = *reinterpret_cast<const float4*>(&wb[(wq * 32 + j * 16 + fr) * W1F_LD + hh * 16 + fq * 4]);
          fB[hh][j][0] = v.x; fB[hh][j][1] = v.y; fB[hh][j][2] = v.z; fB[hh][j][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) fB[hh][j][e] = wb[(hh * 16 + fq * 4 + e) * W1B_LD + wq * 32 + j * 16 + fr];
        }
      }
    } else {
      constexpr int sd = (s + 8 - (FLAG_LAG & 7)) & 7;          // down step inside its chunk
      const float* wb = w2s + buf * W2S_SZ;
      const float* hp = hs + hbuf * HS_SZ;       // hbuf = parity of the chunk this down step contracts
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&hp[(i * 16 + fr) * HS_LD + sd * 16 + fq * 4]);
        fA[hh][i][0] = v.x; fA[hh][i][1] = v.y; fA[hh][i][2] = v.z; fA[hh][i][3] = v.w;
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if constexpr (!BWD) {
          const float4 v = *reinterpret_cast<const float4*>(&wb[(wq * 64 + (2 * hh + jj) * 16 + fr) * W2F_LD + fq * 4]);
          fB[hh][jj][0] = v.x; fB[hh][jj][1] = v.y; fB[hh][jj][2] = v.z; fB[hh][jj][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) fB[hh][jj][e] = wb[(fq * 4 + e) * W2B_LD + wq * 64 + (2 * hh + jj) * 16 + fr];
        }
      }
    }
  };
  auto mfma_half = [&](auto role_c, auto half_c, auto upon_c, auto downon_c) __attribute__((always_inline)) {
    constexpr int hh = decltype(half_c)::value;
    constexpr bool up_on = decltype(upon_c)::value, down_on = decltype(downon_c)::value;
    if constexpr (decltype(role_c)::value == 0) {
      if constexpr (up_on) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              zacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fA[hh][i][e], fB[hh][j][e], zacc[i][j], 0, 0, 0);
      }
    } else {
      if constexpr (down_on) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
              yacc[i][2 * hh + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fA[hh][i][e], fB[hh][jj][e], yacc[i][2 * hh + jj], 0, 0, 0);
      }
    }
  };

  // One quarter (row tile i, column tile j) of the up waves' chunk epilogue: 4 accumulator elements per lane -> hidden
  // units (LDS for the down waves, global memory for backward).  The four quarters of chunk c run in the four half steps
  // of steps 8 (c + 1) and 8 (c + 1) + 1, beside the MFMAs of the next chunk (zold = the finished accumulators).
  auto epi_quarter = [&](auto i_c, auto j_c, int c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value;
    float* hp = hs + (c & 1) * HS_SZ;
    const int lc = wq * 32 + j * 16 + fr;        // column inside the chunk
    float hv[4], fv[4];
    if constexpr (!BWD) {
      const float bj = bpre[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) eamd_act_dact(zold[i][j][r] + bj, ACT, hv[r], fv[r]);
      if (p.p_in > 0.f) {       // wave-uniform
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned gi = (unsigned)(m0 + i * 16 + fq * 4 + r) * (unsigned)F + (unsigned)(c * FHC + lc);
          const bool keep = eamd_drop_keep(seed_in, (unsigned long long)gi, thr_in);
          hv[r] = keep ? hv[r] * inv_in : 0.f;
          fv[r] = keep ? fv[r] * inv_in : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (zold[i][j][r] * fpre[i][j][r]) * p.alpha;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) hp[(i * 16 + fq * 4 + r) * HS_LD + lc] = hv[r];
    // global copies for backward: element r sits (i*16 + r) rows and j*16 columns from the lane's first one
    auto put = [&](float* dstp, const float (&val)[4]) __attribute__((always_inline)) {
      char* base = reinterpret_cast<char*>(dstp + (long)m0 * F + c * FHC);
      if (full_rows) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(base + ((long)(i * 16 + r) * F + j * 16) * 4 + e_toff) = val[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m0 + i * 16 + fq * 4 + r < p.M)
            *reinterpret_cast<float*>(base + ((long)(i * 16 + r) * F + j * 16) * 4 + e_toff) = val[r];
      }
    };
    if (p.h) put(p.h, hv);
    if constexpr (!BWD) { if (p.f) put(p.f, fv); }
  };
  auto load_f = [&](int c) __attribute__((always_inline)) {
    const char* base = reinterpret_cast<const char*>(p.f + (long)m0 * F + c * FHC);
    if (full_rows) {          // scalar base + this lane's 32-bit offset
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            fpre[i][j][r] = *reinterpret_cast<const float*>(base + ((long)(i * 16 + r) * F + j * 16) * 4 + e_toff);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // rows past M: any valid address (their products only reach rows that are never stored)
            const long ro = (long)(min(m0 + i * 16 + fq * 4 + r, p.M - 1) - m0 - fq * 4) * F;
            fpre[i][j][r] = *reinterpret_cast<const float*>(base + (ro + j * 16) * 4 + e_toff);
          }
    }
  };

  // Step g = 8 c + s of "body" c.  Up waves: chunk c is being formed (UP: c < nch); the epilogue of chunk c - 1 runs in
  // steps s = 0, 1 (DA).  Down waves run LAG = 12 steps behind: steps s >= 4 contract chunk c - 1 (DA: 1 <= c <= nch),
  // steps s < 4 chunk c - 2 (DB: 2 <= c <= nch + 1).  The flags are compile-time: a runtime condition around the MFMAs
  // makes the compiler copy the accumulators at every join (and wait for the matrix pipe to drain first).
  // ROLE 0 = up wave, 1 = down wave: each role runs its OWN copy of the step sequence (the role split sits outside the
  // loops - inside a step it makes the compiler merge the two arms and copy accumulators at every join); both copies
  // stage the weight tiles and meet at the same barriers.
  auto step = [&](auto role_c, auto s_c, auto up_c, auto da_c, auto db_c, int c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr int s = decltype(s_c)::value;
    constexpr bool UP = decltype(up_c)::value, DA = decltype(da_c)::value, DB = decltype(db_c)::value;
    constexpr bool DOWN = s >= FLAG_LAG - 8 ? DA : DB;              // this step's down product
    constexpr bool DOWN2 = (s + 2 < FLAG_LAG - 8) ? DB : DA;         // the down step two ahead (s = 6, 7: next body's DB = this DA)
    constexpr bool DOWN1 = (s + 1 < FLAG_LAG - 8) ? DB : DA;         // the down step one ahead
    const int g = 8 * c + s;
    constexpr int buf = s & 1;
    // requests: up tile g + 2 (same register set as tile g, stored a step ago); the down tile of step g + 2
    STAMP(0);
    if constexpr (UP) load_w1(std::integral_constant<int, s & 1>{}, g + 2);
    if constexpr (DOWN2) load_w2(std::integral_constant<int, s & 1>{}, g + 2 - FLAG_LAG);
    if constexpr (BWD && UP && s == 3 && ROLE == 0) load_f(c);
    if constexpr (!BWD && UP && s == 3 && ROLE == 0) {
      if (p.b1) { bpre[0] = p.b1[c * FHC + wq * 32 + fr]; bpre[1] = p.b1[c * FHC + wq * 32 + 16 + fr]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    // down step g contracts chunk c - 1 (s >= LAG - 8) or c - 2; the step after s = 7 is the next body's first
    read_half(role_c, s_c, std::integral_constant<int, 1>{}, (s >= FLAG_LAG - 8 ? c + 1 : c) & 1);
    mfma_half(role_c, std::integral_constant<int, 0>{}, std::integral_constant<bool, UP>{}, std::integral_constant<bool, DOWN>{});
    STAMP(1);
    if constexpr (ROLE == 0 && DA && s < 2) epi_quarter(std::integral_constant<int, 0>{}, std::integral_constant<int, s>{}, c - 1);
    STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    // the tiles of step g + 1 (their buffers were last read in step g - 1, or right behind the barrier that ended it)
    if constexpr (UP) { if (s < 7 || g + 1 < 8 * nch) store_w1(std::integral_constant<int, (s + 1) & 1>{}, buf ^ 1); }
    if constexpr (DOWN1) store_w2(std::integral_constant<int, (s + 1) & 1>{}, buf ^ 1);
    STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    // Stagger: the two waves of a SIMD (one up, one down) must not sit in their LDS-store / barrier window together, or the
    // matrix pipe idles through it.  The second half's fragments are in registers before the barrier, so the DOWN wave
    // issues its second half in FRONT of the barrier (under the up wave's stores and barrier wait) and the UP wave BEHIND
    // it (under the down wave's fragment reads and tile requests).
    if constexpr (ROLE == 1) {
      mfma_half(role_c, std::integral_constant<int, 1>{}, std::integral_constant<bool, UP>{}, std::integral_constant<bool, DOWN>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    read_half(role_c, std::integral_constant<int, (s + 1) & 7>{}, std::integral_constant<int, 0>{}, (s == 7 || s + 1 >= FLAG_LAG - 8 ? c + 1 : c) & 1);
    if constexpr (ROLE == 0) {
      // MFMAs are pure register operations: ordered behind the barrier (and the cold reads just issued) by passing their
      // fragments through an empty volatile asm
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(fA[1][a][e])); asm volatile("" : "+v"(fB[1][a][e])); }
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(role_c, std::integral_constant<int, 1>{}, std::integral_constant<bool, UP>{}, std::integral_constant<bool, DOWN>{});
    }
    STAMP(6);
    if constexpr (ROLE == 0 && DA && s < 2) epi_quarter(std::integral_constant<int, 1>{}, std::integral_constant<int, s>{}, c - 1);
    STAMP(7);
    if constexpr (ROLE == 0 && UP && s == 7) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { zold[i][j] = zacc[i][j]; zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
  };
  auto body = [&](auto role_c, auto up_c, auto da_c, auto db_c, int c) __attribute__((always_inline)) {
    step(role_c, std::integral_constant<int, 0>{}, up_c, da_c, db_c, c);
    step(role_c, std::integral_constant<int, 1>{}, up_c, da_c, db_c, c);
    step(role_c, std::integral_constant<int, 2>{}, up_c, da_c, db_c, c);
    step(role_c, std::integral_constant<int, 3>{}, up_c, da_c, db_c, c);
    if constexpr (decltype(up_c)::value || decltype(da_c)::value) {      // the last body ends with the down waves' step s = 3
      step(role_c, std::integral_constant<int, 4>{}, up_c, da_c, db_c, c);
      step(role_c, std::integral_constant<int, 5>{}, up_c, da_c, db_c, c);
      step(role_c, std::integral_constant<int, 6>{}, up_c, da_c, db_c, c);
      step(role_c, std::integral_constant<int, 7>{}, up_c, da_c, db_c, c);
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto program = [&](auto role_c) __attribute__((always_inline)) {
    read_half(role_c, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
    body(role_c, T_{}, F_{}, F_{}, 0);
    body(role_c, T_{}, T_{}, F_{}, 1);
    for (int c = 2; c < nch; ++c) body(role_c, T_{}, T_{}, T_{}, c);
    body(role_c, F_{}, T_{}, T_{}, nch);
    body(role_c, F_{}, F_{}, T_{}, nch + 1);
  };

  // prologue: up tiles 0 (-> LDS) and 1 (in flight)
  load_w1(std::integral_constant<int, 0>{}, 0);
  load_w1(std::integral_constant<int, 1>{}, 1);
  store_w1(std::integral_constant<int, 0>{}, 0);
  __syncthreads();
  if (up) {
    program(std::integral_constant<int, 0>{});
  } else {
    program(std::integral_constant<int, 1>{});
    // output rows: accumulators -> LDS, over the input rows (the up waves read those for the last time 12 steps ago)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[(i * 16 + fq * 4 + r) * XS_LD + wq * 64 + j * 16 + fr] = yacc[i][j][r];
  }
  __syncthreads();

  // ---- output rows: LDS -> 16-byte row stores (all 512 threads) ----
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

// ------------------------------------------------------------------------------------------------------------------
// Version 3: weights straight from global memory into MFMA operand registers.
// No weight element is shared between waves of a workgroup (an up wave owns 32 hidden columns of the chunk, a down wave
// 64 output columns), so staging the tiles through LDS only bought a layout change - and cost four ds_write_b128 per
// thread and step, the fragment reads, and ONE BARRIER PER STEP that made the two waves of every SIMD wait for each
// other 140 times per launch.  Here a lane fetches its own operand fragments (16 bytes along k for the k-contiguous
// forward weights; along n for the k-strided backward weights, the output columns of a tile being interleaved to
// match) three steps ahead into a ring of four register sets; LDS holds only the 32 input rows and the hidden-unit
// chunks, and the roles meet at ONE barrier per chunk (256 MFMAs per wave): the up waves' epilogue arithmetic no longer
// holds the down waves back.
//   period P (8 steps): up waves form chunk P and run the epilogue of chunk P - 1 in steps 0-3 (one quarter each);
//   down waves contract chunk P - 2; barrier at the end of every period.
constexpr int D_SMEM_FLOATS = XS_SZ + 4 * HS_SZ;      // input rows + 2 hidden-unit chunks + (forward) 2 factor chunks

template <bool BWD, int ACT>
__global__ __launch_bounds__(FNT, 2) void ffn_f32_direct_kernel(const eamd_ffn_t p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const xs = sm;
  float* const hs = xs + XS_SZ;
  float* const fs = hs + 2 * HS_SZ;     // forward: the factor chunks, staged like h for the down waves' 16-byte global stores
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool up = wave < 4;
  const int wq = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * FBM;
  const int F = p.F;
  const int nch = F / FHC;
  const int nsteps = 8 * nch;
  const bool full_rows = m0 + FBM <= p.M;
  using T_ = std::true_type;
  using F_ = std::false_type;

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + FNT * i, row = idx >> 6, c4 = idx & 63;
    *reinterpret_cast<f32x4*>(&xs[row * XS_LD + c4 * 4]) =
        *reinterpret_cast<const f32x4*>(p.x + (long)min(m0 + row, p.M - 1) * FD + c4 * 4);
  }
  __syncthreads();

  if (up) {
    // =============================== up waves ===============================
    const float* __restrict__ W = BWD ? p.w2 : p.w1;
    // column of accumulator tile j inside the chunk: forward j*16 + fr; backward (float2 fragments along n) 2*fr + j
    const int lc0 = BWD ? wq * 32 + 2 * fr : wq * 32 + fr;
    constexpr int LCJ = BWD ? 1 : 16;
    const unsigned boff = BWD ? (unsigned)(((fq * 4) * F + wq * 32 + 2 * fr) * 4) : (unsigned)(((wq * 32 + fr) * FD + fq * 4) * 4);
    f32x4 bs[4][4];            // forward: [set][q*2 + j] = 4 k-elements;  backward: [set][q*2 + e/2] = (e even: j0 j1, e odd: j0 j1)
    f32x4 zacc[2][2], zold[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) { zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; zold[i][j] = zacc[i][j]; }
    const unsigned thr_in = eamd_drop_thr16(p.p_in);
    const float inv_in = p.p_in > 0.f ? eamd_drop_inv(thr_in) : 1.f;
    const unsigned seed_in = (!BWD && p.p_in > 0.f) ? eamd_drop_seed((const unsigned long long*)p.drop_step, p.salt_in) : 0u;
    float bpre[2] = {0.f, 0.f};
    float fpre[2][2][4];
    const unsigned e_toff = (unsigned)(((fq * 4) * F + lc0) * 4);

    auto load_b = [&](auto set_c, int g) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      const int gc = min(g, nsteps - 1);
      const int c = gc >> 3, s = gc & 7;
      if constexpr (!BWD) {
        const char* base = reinterpret_cast<const char*>(W + (long)(c * FHC) * FD + s * 32);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bs[SET][q * 2 + j] = *reinterpret_cast<const f32x4*>(base + ((long)(j * 16) * FD + q * 16) * 4 + boff);
      } else {
        const char* base = reinterpret_cast<const char*>(W + (long)(s * 32) * F + c * FHC);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float2 v = *reinterpret_cast<const float2*>(base + ((long)(q * 16 + e) * F) * 4 + boff);
            bs[SET][q * 2 + (e >> 1)][(e & 1) * 2] = v.x;
            bs[SET][q * 2 + (e >> 1)][(e & 1) * 2 + 1] = v.y;
          }
      }
    };
    float fA[2][2][4];
    auto read_a = [&](auto s_c, auto half_c) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value, hh = decltype(half_c)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&xs[(i * 16 + fr) * XS_LD + s * 32 + hh * 16 + fq * 4]);
        fA[hh][i][0] = v.x; fA[hh][i][1] = v.y; fA[hh][i][2] = v.z; fA[hh][i][3] = v.w;
      }
    };
    auto mfma_half = [&](auto set_c, auto half_c) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value, q = decltype(half_c)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float b = BWD ? bs[SET][q * 2 + (e >> 1)][(e & 1) * 2 + j] : bs[SET][q * 2 + j][e];
            zacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fA[q][i][e], b, zacc[i][j], 0, 0, 0);
          }
    };
    // epilogue of rows i*16 + fq*4 + r (r = 0..3) x column tile j of chunk c
    auto epi_quarter = [&](auto i_c, auto j_c, int c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value;
      float* hp = hs + (c & 1) * HS_SZ;
      const int lc = lc0 + j * LCJ;
      float hv[4], fv[4];
      if constexpr (!BWD) {
        const float bj = bpre[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) eamd_act_dact(zold[i][j][r] + bj, ACT, hv[r], fv[r]);
        if (p.p_in > 0.f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned gi = (unsigned)(m0 + i * 16 + fq * 4 + r) * (unsigned)F + (unsigned)(c * FHC + lc);
            const bool keep = eamd_drop_keep(seed_in, (unsigned long long)gi, thr_in);
            hv[r] = keep ? hv[r] * inv_in : 0.f;
            fv[r] = keep ? fv[r] * inv_in : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (zold[i][j][r] * fpre[i][j][r]) * p.alpha;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) hp[(i * 16 + fq * 4 + r) * HS_LD + lc] = hv[r];
      // the copies for backward (h, f) leave from LDS: the down waves store them as whole 16-byte row pieces next period
      if constexpr (!BWD) {
        float* fp = fs + (c & 1) * HS_SZ;
#pragma unroll
        for (int r = 0; r < 4; ++r) fp[(i * 16 + fq * 4 + r) * HS_LD + lc] = fv[r];
      }
    };
    auto load_f = [&](int c) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(p.f + (long)m0 * F + c * FHC);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long ro = full_rows ? (long)(i * 16 + r) * F : (long)(min(m0 + i * 16 + fq * 4 + r, p.M - 1) - m0 - fq * 4) * F;
          const float2 v = *reinterpret_cast<const float2*>(base + ro * 4 + e_toff);       // columns 2 fr, 2 fr + 1 = tiles 0, 1
          fpre[i][0][r] = v.x; fpre[i][1][r] = v.y;
        }
    };
    auto step = [&](auto s_c, auto up_c, auto epi_c, int P) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      constexpr bool UP = decltype(up_c)::value, EPI = decltype(epi_c)::value;
      if constexpr (UP) {
        load_b(std::integral_constant<int, (s + 3) & 3>{}, 8 * P + s + 3);
        if constexpr (s == 4) {      // behind the last epilogue quarter of the previous chunk (step 3), which still reads them
          if constexpr (BWD) load_f(P);
          else if (p.b1) { bpre[0] = p.b1[P * FHC + lc0]; bpre[1] = p.b1[P * FHC + lc0 + 16]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        read_a(s_c, std::integral_constant<int, 1>{});
        mfma_half(std::integral_constant<int, s & 3>{}, std::integral_constant<int, 0>{});
      }
      if constexpr (EPI && s < 4) epi_quarter(std::integral_constant<int, s & 1>{}, std::integral_constant<int, (s / 2)>{}, P - 1);
      if constexpr (UP) {
        __builtin_amdgcn_sched_barrier(0);
        read_a(std::integral_constant<int, (s + 1) & 7>{}, std::integral_constant<int, 0>{});
        mfma_half(std::integral_constant<int, s & 3>{}, std::integral_constant<int, 1>{});
        if constexpr (s == 7) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { zold[i][j] = zacc[i][j]; zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
      }
      if constexpr (s == 7) __syncthreads();
    };
    auto period = [&](auto up_c, auto epi_c, int P) __attribute__((always_inline)) {
      step(std::integral_constant<int, 0>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 1>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 2>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 3>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 4>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 5>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 6>{}, up_c, epi_c, P);
      step(std::integral_constant<int, 7>{}, up_c, epi_c, P);
    };
    load_b(std::integral_constant<int, 0>{}, 0);
    load_b(std::integral_constant<int, 1>{}, 1);
    load_b(std::integral_constant<int, 2>{}, 2);
    read_a(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    period(T_{}, F_{}, 0);
    for (int P = 1; P < nch; ++P) period(T_{}, T_{}, P);
    period(F_{}, T_{}, nch);
    period(F_{}, F_{}, nch + 1);
  } else {
    // =============================== down waves ===============================
    const float* __restrict__ W = BWD ? p.w1 : p.w2;
    const unsigned boff = BWD ? (unsigned)(((fq * 4) * FD + wq * 64 + 4 * fr) * 4) : (unsigned)(((wq * 64 + fr) * F + fq * 4) * 4);
    f32x4 bs[4][4];            // forward: [set][j] = 4 k-elements of column tile j;  backward: [set][e] = column tiles 0..3 at k = fq*4 + e
    f32x4 yacc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_b = [&](auto set_c, int gd) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      const int gc = min(max(gd, 0), nsteps - 1);
      const int cd = gc >> 3, sd = gc & 7;
      if constexpr (!BWD) {
        const char* base = reinterpret_cast<const char*>(W + cd * FHC + sd * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[SET][j] = *reinterpret_cast<const f32x4*>(base + ((long)(j * 16) * F) * 4 + boff);
      } else {
        const char* base = reinterpret_cast<const char*>(W + (long)(cd * FHC + sd * 16) * FD);
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[SET][e] = *reinterpret_cast<const f32x4*>(base + ((long)e * FD) * 4 + boff);
      }
    };
    float fA[2][2][4];         // [step parity][row tile][4 k-elements]
    auto read_a = [&](auto par_c, int sd, int hbuf) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_c)::value;
      const float* hp = hs + hbuf * HS_SZ;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(&hp[(i * 16 + fr) * HS_LD + sd * 16 + fq * 4]);
        fA[PAR][i][0] = v.x; fA[PAR][i][1] = v.y; fA[PAR][i][2] = v.z; fA[PAR][i][3] = v.w;
      }
    };
    auto mfma_step = [&](auto set_c, auto par_c) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value, PAR = decltype(par_c)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float b = BWD ? bs[SET][e][j] : bs[SET][j][e];
            yacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fA[PAR][i][e], b, yacc[i][j], 0, 0, 0);
          }
    };
    // period P contracts chunk P - 2 (D: 2 <= P <= nch + 1); DN = the next period is a down period too (prefetch target)
    auto step = [&](auto s_c, auto d_c, auto dn_c, int P) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      constexpr bool D = decltype(d_c)::value, DN = decltype(dn_c)::value;
      const int gd = 8 * (P - 2) + s;
      if constexpr (s + 3 < 8 ? D : DN) load_b(std::integral_constant<int, (s + 3) & 3>{}, gd + 3);
      if constexpr (D && (s == 1 || s == 3)) {
        // global copies of chunk P - 2 for backward (forward: h at s = 1, f at s = 3; backward: dz at s = 1): 32 rows x 512 bytes
        // from the LDS image, four 16-byte pieces per lane
        float* dstp = (s == 1) ? p.h : (BWD ? nullptr : p.f);
        if (dstp) {
          const float* src = (s == 1 ? hs : fs) + (P & 1) * HS_SZ;
          const int dt = t - 256;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = dt + 256 * k, lr = idx >> 5, c4 = idx & 31;
            if (full_rows || m0 + lr < p.M)
              *reinterpret_cast<float4*>(dstp + (long)(m0 + lr) * F + (P - 2) * FHC + c4 * 4) =
                  *reinterpret_cast<const float4*>(&src[lr * HS_LD + c4 * 4]);
          }
        }
      }
      if constexpr (D) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s < 7) read_a(std::integral_constant<int, (s + 1) & 1>{}, s + 1, P & 1);
        mfma_step(std::integral_constant<int, s & 3>{}, std::integral_constant<int, s & 1>{});
      }
      if constexpr (s == 7) {
        __syncthreads();
        if constexpr (DN) read_a(std::integral_constant<int, 0>{}, 0, (P + 1) & 1);      // chunk P - 1 is complete behind this barrier
      }
    };
    auto period = [&](auto d_c, auto dn_c, int P) __attribute__((always_inline)) {
      step(std::integral_constant<int, 0>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 1>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 2>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 3>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 4>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 5>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 6>{}, d_c, dn_c, P);
      step(std::integral_constant<int, 7>{}, d_c, dn_c, P);
    };
    period(F_{}, F_{}, 0);
    period(F_{}, T_{}, 1);
    for (int P = 2; P <= nch; ++P) period(T_{}, T_{}, P);
    period(T_{}, F_{}, nch + 1);
    // output rows: accumulators -> LDS over the input rows (the up waves read those for the last time two periods ago)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (BWD) {
          *reinterpret_cast<float4*>(&xs[(i * 16 + fq * 4 + r) * XS_LD + wq * 64 + 4 * fr]) =
              make_float4(yacc[i][0][r], yacc[i][1][r], yacc[i][2][r], yacc[i][3][r]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) xs[(i * 16 + fq * 4 + r) * XS_LD + wq * 64 + j * 16 + fr] = yacc[i][j][r];
        }
      }
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
  if (!(p.reserved & 32)) {      // default: the register-direct version (bit 5 of `reserved` selects the LDS-staged one)
    constexpr size_t smem_d = (size_t)D_SMEM_FLOATS * sizeof(float);
    static const hipError_t attr_err_d = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_f32_direct_kernel<BWD, ACT>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_d);
    if (attr_err_d != hipSuccess) return (int)attr_err_d;
    hipLaunchKernelGGL((ffn_f32_direct_kernel<BWD, ACT>), dim3(nblk), dim3(FNT), smem_d, stream, p);
    EAMD_LAUNCH_CHECK();
    return EAMD_OK;
  }
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

#ifdef FFN_STAMP
extern "C" int eamd_ffn_debug_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ffn_stamps), sizeof(ffn_stamps));
}
#endif

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
