// Fused position-wise feed-forward kernels for gfx950, fp32 operands (v_mfma_f32_16x16x4_f32: exact fp32 products).
//
// reference: espnet/nets/pytorch_backend/transformer/positionwise_feed_forward.py:12-32
//     forward   out = R + alpha * drop_out( drop_in(act(x W1^T + b1)) W2^T + b2 )
//     backward  dz = alpha * (dy W2) (.) f,   dx = dz W1          (f = mask / (1 - p) * act'(x W1^T + b1), kept by forward)
//
// Why one kernel: as two GEMMs the pair is a [M, 2048] x K = 256 product whose 65 MB result (x 2 outputs) is written
// and read back, and a [M, 256] x K = 2048 product that has only M / 64 x 4 = 500 tiles for 256 CUs; each launch pays
// its own prologue / epilogue / tail.  Both products are local to a block of ROWS, so one workgroup takes 32 rows
// through both: the hidden units never make a round trip for the second product, the skinny product disappears into a
// loop with no launch boundary, and M / 32 = 249 workgroups fill the chip for the whole launch.
//
// Structure (512 threads = 8 waves, one workgroup per CU, two waves per SIMD - one of each role):
//   * waves 0-3 ("up") form z = x W1^T chunk by chunk of 128 hidden units (wave tile 32 x 32, K = 256 in 8 steps of 32),
//     apply bias / activation / dropout to the finished accumulators beside the next chunk's MFMAs, and leave h (and f)
//     in LDS;
//   * waves 4-7 ("down") accumulate out[32, 256] += h_chunk W2[:, chunk]^T two chunks behind (wave tile 32 x 64, 8 steps
//     of 16 hidden units) and write the LDS images of h and f to global memory as 16-byte row pieces for backward;
//   * weights go straight from global memory into MFMA operand registers (see the kernel's comment); LDS holds the 32
//     input rows and the hidden-unit chunks only; ONE barrier per chunk.
// History of the structure, measured on MI355X at config 2 (M = 7968, F = 2048; the GEMM pairs take 180 / 176 us):
//   weight tiles staged through LDS, one barrier per step, roles inside each step      208 / 171 us (forward / backward)
//   + role programs split at the top, fragments prefetched across the barrier          192 / 167 us
//   register-direct weights, one barrier per chunk                                      177 / 150 us
//   + h / f copies stored from LDS by the down waves                                    156 / 145 us
// What the s_memtime stamps of the diagnostic build (-DFFN_STAMP, tools/ffn_stamp.py) show is left: the up waves carry
// all of the epilogue arithmetic, so the down waves wait for them at the chunk barrier.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/espnet_amd.h"
#include "ffn_ln.h"
#include "ln_bwd_rows.h"

#ifdef FFN_STAMP
// diagnostic build only: s_memtime stamps of one up and one down wave of workgroup 100 during body 4
__device__ unsigned long long ffn_stamps[2][2][8][8];
#define STAMP3(role, P, s, k)                                                                            \
  do {                                                                                                   \
    if (blockIdx.x == 100 && (P) == 4 && (wave == 0 || wave == 4) && lane == 0)                          \
      ffn_stamps[BWD][role][s][k] = __builtin_amdgcn_s_memtime();                                        \
  } while (0)
#else
#define STAMP3(role, P, s, k) do { } while (0)
#endif

bool eamd_ffn_bf16_ok(const eamd_ffn_t* p);                      // ffn_bf16.hip
int eamd_ffn_bf16_launch(const eamd_ffn_t* p, int bwd, void* stream);
bool eamd_ffn_f32_sym(int F);                                    // ffn_f32_sym.hip: the symmetric fp32 form (F % 256 == 0)
int eamd_ffn_f32_sym_launch(const eamd_ffn_t* p, int bwd, void* stream);
int eamd_ffn_f32_sym_pack(const float* w1, const float* w2, float* p0, float* p1, float* p2, float* p3, int F, void* stream);

namespace {

constexpr int FBM = 32;          // rows per workgroup
constexpr int FD = 256;          // model width (template constant of this kernel)
constexpr int FHC = 128;         // hidden units per chunk
constexpr int FNT = 512;
constexpr int XS_LD = 264;       // [32][256 + 8]: ds_read_b128 conflict-free (row stride = 2 mod 16 chunks)
constexpr int HS_LD = 136;       // [32][128 + 8]
constexpr int XS_SZ = FBM * XS_LD;                 // 8448 floats
constexpr int HS_SZ = FBM * HS_LD;                 // 4352

// ------------------------------------------------------------------------------------------------------------------
// Weights straight from global memory into MFMA operand registers.
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
  // hidden split (eamd_ffn_t.hsplit = gridDim.y > 1, few rows): this workgroup takes hidden units hid0 .. hid0 + nch * 128 - 1 of
  // its 32 rows and ADDS its share of the second product to `out` (zeroed by the caller; slice 0 also brings bias / residual)
  const int nch = F / FHC / (int)gridDim.y;
  const int hid0 = (int)blockIdx.y * nch * FHC;
  const int nsteps = 8 * nch;
  const bool full_rows = m0 + FBM <= p.M;
  using T_ = std::true_type;
  using F_ = std::false_type;

  if (!BWD && p.ln_x) {        // LayerNorm in front: normalise the rows on their way into LDS (ffn_ln.h)
    ffn_ln_stage(p, m0, t, [&](int row, int col, float4 y) __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(&xs[row * XS_LD + col]) = (f32x4){y.x, y.y, y.z, y.w};
    });
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + FNT * i, row = idx >> 6, c4 = idx & 63;
      *reinterpret_cast<f32x4*>(&xs[row * XS_LD + c4 * 4]) =
          *reinterpret_cast<const f32x4*>(p.x + (long)min(m0 + row, p.M - 1) * FD + c4 * 4);
    }
  }
  __syncthreads();

  if (up) {
    // =============================== up waves ===============================
    const float* __restrict__ W = p.w1 + (long)hid0 * FD;      // packed image of the first product (forward: W1; backward: W2 read along its rows), chunk-major
    // column of accumulator tile j inside the chunk: forward j*16 + fr; backward (float2 fragments along n) 2*fr + j
    // both directions interleave the two column tiles (tile j holds columns 2 fr + j): a lane's (j = 0, 1) elements of one
    // row are neighbours - one dropout hash (an element PAIR), one 8-byte LDS store
    const int lc0 = wq * 32 + 2 * fr;
    constexpr int LCJ = 1;
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

    // operand fragments from the PACKED image of this product (eamd_ffn_pack_f32: fragment order, a wave-instruction reads
    // 1 KB of consecutive bytes; from the nn.Linear layout it touched 16 rows x 64 bytes): image[step g][wave][v][lane]
    auto load_b = [&](auto set_c, int g) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      const char* base = reinterpret_cast<const char*>(W) + ((long)min(g, nsteps - 1) * 4 + wq) * 4096 + lane * 16;
#pragma unroll
      for (int v = 0; v < 4; ++v) bs[SET][v] = *reinterpret_cast<const f32x4*>(base + v * 1024);
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
    // epilogue of rows i*16 + fq*4 + rr*2 + (0, 1) x both column tiles of chunk c: quarter (i, rr)
    auto epi_quarter = [&](auto i_c, auto rr_c, int c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value, rr = decltype(rr_c)::value;
      float* hp = hs + (c & 1) * HS_SZ;
      float* fp = fs + (c & 1) * HS_SZ;
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        constexpr int dummy = 0; (void)dummy;
        const int r = rr * 2 + r2;
        float hv[2], fv[2];
        if constexpr (!BWD) {
#pragma unroll
          for (int j = 0; j < 2; ++j) eamd_act_dact(zold[i][j][r] + bpre[j], ACT, hv[j], fv[j]);
          if (p.p_in > 0.f) {
            const unsigned gi = (unsigned)(m0 + i * 16 + fq * 4 + r) * (unsigned)F + (unsigned)(hid0 + c * FHC + lc0);      // even
            const unsigned hsh = eamd_drop_pair(seed_in, (unsigned long long)(gi >> 1));
            const bool k0 = (hsh & 0xffffu) >= thr_in, k1 = (hsh >> 16) >= thr_in;
            hv[0] = k0 ? hv[0] * inv_in : 0.f; fv[0] = k0 ? fv[0] * inv_in : 0.f;
            hv[1] = k1 ? hv[1] * inv_in : 0.f; fv[1] = k1 ? fv[1] * inv_in : 0.f;
          }
          *reinterpret_cast<float2*>(&fp[(i * 16 + fq * 4 + r) * HS_LD + lc0]) = make_float2(fv[0], fv[1]);
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) hv[j] = (zold[i][j][r] * fpre[i][j][r]) * p.alpha;
        }
        *reinterpret_cast<float2*>(&hp[(i * 16 + fq * 4 + r) * HS_LD + lc0]) = make_float2(hv[0], hv[1]);
      }
    };
    auto load_f = [&](int c) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(p.f + (long)m0 * F + hid0 + c * FHC);
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
      STAMP3(0, P, s, 0);
      if constexpr (UP) {
        load_b(std::integral_constant<int, (s + 3) & 3>{}, 8 * P + s + 3);
        if constexpr (s == 4) {      // behind the last epilogue quarter of the previous chunk (step 3), which still reads them
          if constexpr (BWD) load_f(P);
          else if (p.b1) { bpre[0] = p.b1[hid0 + P * FHC + lc0]; bpre[1] = p.b1[hid0 + P * FHC + lc0 + 1]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        read_a(s_c, std::integral_constant<int, 1>{});
        mfma_half(std::integral_constant<int, s & 3>{}, std::integral_constant<int, 0>{});
      }
      STAMP3(0, P, s, 1);
      if constexpr (EPI && s < 4) epi_quarter(std::integral_constant<int, s & 1>{}, std::integral_constant<int, (s / 2)>{}, P - 1);     // (row tile, row pair)
      STAMP3(0, P, s, 2);
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
      STAMP3(0, P, s, 3);
      if constexpr (s == 7) __syncthreads();
      STAMP3(0, P, s, 4);
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
    const float* __restrict__ W = p.w2 + (long)hid0 * FD;      // packed image of the second product, chunk-major
    f32x4 bs[4][4];            // forward: [set][j] = 4 k-elements of column tile j;  backward: [set][e] = column tiles 0..3 at k = fq*4 + e
    f32x4 yacc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_b = [&](auto set_c, int gd) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      const char* base = reinterpret_cast<const char*>(W) + ((long)min(max(gd, 0), nsteps - 1) * 4 + wq) * 4096 + lane * 16;
#pragma unroll
      for (int v = 0; v < 4; ++v) bs[SET][v] = *reinterpret_cast<const f32x4*>(base + v * 1024);
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
      STAMP3(1, P, s, 0);
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
              *reinterpret_cast<float4*>(dstp + (long)(m0 + lr) * F + hid0 + (P - 2) * FHC + c4 * 4) =
                  *reinterpret_cast<const float4*>(&src[lr * HS_LD + c4 * 4]);
          }
        }
      }
      if constexpr (D) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s < 7) read_a(std::integral_constant<int, (s + 1) & 1>{}, s + 1, P & 1);
        mfma_step(std::integral_constant<int, s & 3>{}, std::integral_constant<int, s & 1>{});
      }
      STAMP3(1, P, s, 3);
      if constexpr (s == 7) {
        __syncthreads();
        STAMP3(1, P, s, 4);
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
  if constexpr (BWD) {
    if (p.lnb_x) {       // LayerNorm backward over the finished dx rows (ln_bwd_rows.h); scratch = the hidden-unit chunk buffers
      const EamdLnbArgs la{p.lnb_x, p.lnb_gamma, p.lnb_mean, p.lnb_rstd, p.lnb_dres, p.lnb_ws, p.lnb_drop_out, p.lnb_drop_p,
                           (unsigned long long)p.lnb_drop_salt, p.drop_step, p.out, (long)FD, p.M};
      eamd_ln_bwd_rows32<XS_LD>(la, xs, hs, hs + FBM * FD, m0, t, (int)blockIdx.x);
      return;
    }
  }
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
    const bool first = blockIdx.y == 0;         // hidden split: bias and residual come with slice 0
    if constexpr (!BWD) {
      if (p.b2 && first) {
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
      if (p.R && first) r4 = *reinterpret_cast<const float4*>(p.R + gi);
      v[0] = v[0] * p.alpha + r4.x; v[1] = v[1] * p.alpha + r4.y; v[2] = v[2] * p.alpha + r4.z; v[3] = v[3] * p.alpha + r4.w;
    }
    if (gridDim.y > 1) {       // two addends onto zeros: the sum does not depend on their order
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(p.out + gi + e, v[e]);
    } else {
      *reinterpret_cast<float4*>(p.out + gi) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// The four packed weight images of one FFN, fragment order image[step g = chunk*8 + s][wave wq][v][lane] (float4 each):
//   which 0 forward first  : v = q*2 + j  <- W1[c*128 + wq*32 + 2 fr + j][s*32 + q*16 + fq*4 + (0..3)]
//   which 1 forward second : v = j        <- W2[wq*64 + j*16 + fr][c*128 + s*16 + fq*4 + (0..3)]
//   which 2 backward first : v = q*2 + e/2, component (e%2)*2 + j  <- W2[s*32 + q*16 + fq*4 + e][c*128 + wq*32 + 2 fr + j]
//   which 3 backward second: v = e        <- W1[c*128 + s*16 + fq*4 + e][wq*64 + 4 fr + (0..3)]
__global__ __launch_bounds__(256) void ffn_pack_f32_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                           float* __restrict__ p0, float* __restrict__ p1,
                                                           float* __restrict__ p2, float* __restrict__ p3, int F) {
  const int which = blockIdx.y;
  const long piece = (long)blockIdx.x * 256 + threadIdx.x;           // ((g*4 + wq)*4 + v)*64 + lane
  const long npiece = (long)(F / FHC) * 8 * 4 * 4 * 64;
  if (piece >= npiece) return;
  const int lane = piece & 63, v = (piece >> 6) & 3, wq = (piece >> 8) & 3;
  const int g = (int)(piece >> 10), c = g >> 3, s = g & 7;
  const int fr = lane & 15, fq = lane >> 4;
  float4 o;
  if (which == 0) {
    const int q = v >> 1, j = v & 1;
    o = *reinterpret_cast<const float4*>(w1 + (long)(c * FHC + wq * 32 + 2 * fr + j) * FD + s * 32 + q * 16 + fq * 4);
    *reinterpret_cast<float4*>(p0 + piece * 4) = o;
  } else if (which == 1) {
    o = *reinterpret_cast<const float4*>(w2 + (long)(wq * 64 + v * 16 + fr) * F + c * FHC + s * 16 + fq * 4);
    *reinterpret_cast<float4*>(p1 + piece * 4) = o;
  } else if (which == 2) {
    const int q = v >> 1, e0 = (v & 1) * 2;
    const float* r0 = w2 + (long)(s * 32 + q * 16 + fq * 4 + e0) * F + c * FHC + wq * 32 + 2 * fr;
    const float2 a = *reinterpret_cast<const float2*>(r0), b = *reinterpret_cast<const float2*>(r0 + F);
    *reinterpret_cast<float4*>(p2 + piece * 4) = make_float4(a.x, a.y, b.x, b.y);
  } else {
    o = *reinterpret_cast<const float4*>(w1 + (long)(c * FHC + s * 16 + fq * 4 + v) * FD + wq * 64 + 4 * fr);
    *reinterpret_cast<float4*>(p3 + piece * 4) = o;
  }
}

// the same packing for MANY layers in one launch (a table of jobs in the kernel arguments; blockIdx.z = job): a 12-layer macaron
// Conformer packs 24 feed-forward blocks per step - 24 launches of 5.7 us with their gaps, or one launch of all of them
constexpr int FFN_PACK_JOBS = 32;
struct FfnPackJob { const float* w1; const float* w2; float* p0; float* p1; float* p2; float* p3; int F; int pad; };
struct FfnPackTable { FfnPackJob j[FFN_PACK_JOBS]; };
__global__ __launch_bounds__(256) void ffn_pack_f32_multi_kernel(const FfnPackTable tab) {
  const FfnPackJob jb = tab.j[blockIdx.z];
  const int which = blockIdx.y, F = jb.F;
  const long npiece = (long)(F / FHC) * 8 * 4 * 4 * 64;
  for (long piece = (long)blockIdx.x * 256 + threadIdx.x; piece < npiece; piece += (long)gridDim.x * 256) {
    const int lane = piece & 63, v = (piece >> 6) & 3, wq = (piece >> 8) & 3;
    const int g = (int)(piece >> 10), c = g >> 3, s = g & 7;
    const int fr = lane & 15, fq = lane >> 4;
    if (which == 0) {
      const int q = v >> 1, j = v & 1;
      *reinterpret_cast<float4*>(jb.p0 + piece * 4) =
          *reinterpret_cast<const float4*>(jb.w1 + (long)(c * FHC + wq * 32 + 2 * fr + j) * FD + s * 32 + q * 16 + fq * 4);
    } else if (which == 1) {
      *reinterpret_cast<float4*>(jb.p1 + piece * 4) =
          *reinterpret_cast<const float4*>(jb.w2 + (long)(wq * 64 + v * 16 + fr) * F + c * FHC + s * 16 + fq * 4);
    } else if (which == 2) {
      const int q = v >> 1, e0 = (v & 1) * 2;
      const float* r0 = jb.w2 + (long)(s * 32 + q * 16 + fq * 4 + e0) * F + c * FHC + wq * 32 + 2 * fr;
      const float2 a = *reinterpret_cast<const float2*>(r0), b = *reinterpret_cast<const float2*>(r0 + F);
      *reinterpret_cast<float4*>(jb.p2 + piece * 4) = make_float4(a.x, a.y, b.x, b.y);
    } else {
      *reinterpret_cast<float4*>(jb.p3 + piece * 4) =
          *reinterpret_cast<const float4*>(jb.w1 + (long)(c * FHC + s * 16 + fq * 4 + v) * FD + wq * 64 + 4 * fr);
    }
  }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <bool BWD, int ACT>
int launch_ffn(const eamd_ffn_t& p, hipStream_t stream) {
  constexpr size_t smem = (size_t)D_SMEM_FLOATS * sizeof(float);
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_f32_direct_kernel<BWD, ACT>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (attr_err != hipSuccess) return (int)attr_err;
  const int nblk = (p.M + FBM - 1) / FBM;
  hipLaunchKernelGGL((ffn_f32_direct_kernel<BWD, ACT>), dim3(nblk, p.hsplit > 1 ? p.hsplit : 1), dim3(FNT), smem, stream, p);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

int check_ffn(const eamd_ffn_t* p, bool bwd) {
  if (!p || !p->x || !p->w1 || !p->w2 || !p->out) return EAMD_EINVAL;
  if (p->M <= 0 || p->D <= 0 || p->F <= 0) return EAMD_EINVAL;
  if (p->dtype != 0 && p->dtype != 1) return EAMD_EINVAL;
  if (p->ln_x) {      // LayerNorm in front (forward only): x is written, not read
    if (bwd || !p->ln_w || !p->ln_b || !p->ln_mean || !p->ln_rstd) return EAMD_EINVAL;
    if (!al16(p->ln_x) || !al16(p->ln_w) || !al16(p->ln_b)) return EAMD_EUNSUPPORTED;
  }
  if (p->hsplit < 0) return EAMD_EINVAL;
  if (p->dtype == 1) {                                               // bf16 operands: ffn_bf16.hip
    if (!eamd_ffn_bf16_ok(p) || p->hsplit > 1 || p->lnb_x) return EAMD_EUNSUPPORTED;
    if (bwd) return p->f ? EAMD_OK : EAMD_EINVAL;
    if (p->p_in < 0.f || p->p_in >= 1.f || p->p_out < 0.f || p->p_out >= 1.f) return EAMD_EINVAL;
    if ((p->p_in > 0.f || p->p_out > 0.f) && !p->drop_step) return EAMD_EINVAL;
    return (p->act == EAMD_ACT_RELU || p->act == EAMD_ACT_SWISH) ? EAMD_OK : EAMD_EUNSUPPORTED;
  }
  if (p->D != FD || p->F % FHC != 0 || p->F < 2 * FHC) return EAMD_EUNSUPPORTED;
  if (p->hsplit > 1 && (p->hsplit != 2 || (p->F / FHC) % 2 != 0 || p->F < 4 * FHC)) return EAMD_EUNSUPPORTED;
  if ((long)p->M * p->F >= (1L << 31)) return EAMD_EUNSUPPORTED;     // 32-bit dropout pair index space
  if (!al16(p->x) || !al16(p->w1) || !al16(p->w2) || !al16(p->out) || (p->R && !al16(p->R)) || (p->b2 && !al16(p->b2)))
    return EAMD_EUNSUPPORTED;
  if (p->lnb_x) {
    if (!bwd || p->hsplit > 1 || !p->lnb_gamma || !p->lnb_mean || !p->lnb_rstd || !p->lnb_ws) return EAMD_EINVAL;
    if (p->lnb_drop_out && (!p->drop_step || p->lnb_drop_p < 0.f || p->lnb_drop_p >= 1.f)) return EAMD_EINVAL;
    if (!al16(p->lnb_x) || !al16(p->lnb_gamma) || (p->lnb_dres && !al16(p->lnb_dres)) || (p->lnb_drop_out && !al16(p->lnb_drop_out)))
      return EAMD_EUNSUPPORTED;
  }
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

extern "C" int eamd_ffn_pack_f32(const float* w1, const float* w2, float* fwd_first, float* fwd_second, float* bwd_first,
                                 float* bwd_second, int D, int F, void* stream) {
  if (!w1 || !w2 || !fwd_first || !fwd_second || !bwd_first || !bwd_second || F <= 0) return EAMD_EINVAL;
  if (D != FD || F % FHC != 0 || F < 2 * FHC) return EAMD_EUNSUPPORTED;
  for (const void* q : {(const void*)w1, (const void*)w2, (const void*)fwd_first, (const void*)fwd_second, (const void*)bwd_first,
                        (const void*)bwd_second})
    if (!al16(q)) return EAMD_EUNSUPPORTED;
  if (eamd_ffn_f32_sym(F)) return eamd_ffn_f32_sym_pack(w1, w2, fwd_first, fwd_second, bwd_first, bwd_second, F, stream);
  const long npiece = (long)(F / FHC) * 8 * 4 * 4 * 64;
  hipLaunchKernelGGL(ffn_pack_f32_kernel, dim3((unsigned)((npiece + 255) / 256), 4), dim3(256), 0, (hipStream_t)stream, w1, w2,
                     fwd_first, fwd_second, bwd_first, bwd_second, F);
  EAMD_LAUNCH_CHECK();
  return EAMD_OK;
}

extern "C" int eamd_ffn_pack_f32_multi(const eamd_ffn_pack_t* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return EAMD_EINVAL;
  for (int i = 0; i < njobs; ++i) {
    const eamd_ffn_pack_t& q = jobs[i];
    if (!q.w1 || !q.w2 || !q.fwd_first || !q.fwd_second || !q.bwd_first || !q.bwd_second || q.F <= 0) return EAMD_EINVAL;
    if (q.D != FD || q.F % FHC != 0 || q.F < 2 * FHC || eamd_ffn_f32_sym(q.F)) return EAMD_EUNSUPPORTED;
    for (const void* a : {(const void*)q.w1, (const void*)q.w2, (const void*)q.fwd_first, (const void*)q.fwd_second,
                          (const void*)q.bwd_first, (const void*)q.bwd_second})
      if (!al16(a)) return EAMD_EUNSUPPORTED;
  }
  for (int i0 = 0; i0 < njobs; i0 += FFN_PACK_JOBS) {
    const int n = njobs - i0 < FFN_PACK_JOBS ? njobs - i0 : FFN_PACK_JOBS;
    FfnPackTable tab;
    int fmax = 0;
    for (int i = 0; i < n; ++i) {
      const eamd_ffn_pack_t& q = jobs[i0 + i];
      tab.j[i] = FfnPackJob{q.w1, q.w2, q.fwd_first, q.fwd_second, q.bwd_first, q.bwd_second, q.F, 0};
      fmax = q.F > fmax ? q.F : fmax;
    }
    const long npiece = (long)(fmax / FHC) * 8 * 4 * 4 * 64;
    const unsigned bx = (unsigned)((npiece + 255) / 256 < 128 ? (npiece + 255) / 256 : 128);
    hipLaunchKernelGGL(ffn_pack_f32_multi_kernel, dim3(bx, 4, n), dim3(256), 0, (hipStream_t)stream, tab);
    EAMD_LAUNCH_CHECK();
  }
  return EAMD_OK;
}

extern "C" int eamd_ffn_fwd(const eamd_ffn_t* p, void* stream) {
  const int rc = check_ffn(p, false);
  if (rc != EAMD_OK) return rc;
  if (p->dtype == 1) return eamd_ffn_bf16_launch(p, 0, stream);
  if (p->hsplit <= 1 && eamd_ffn_f32_sym(p->F)) return eamd_ffn_f32_sym_launch(p, 0, stream);
  return p->act == EAMD_ACT_SWISH ? launch_ffn<false, EAMD_ACT_SWISH>(*p, (hipStream_t)stream)
                                  : launch_ffn<false, EAMD_ACT_RELU>(*p, (hipStream_t)stream);
}

extern "C" int eamd_ffn_bwd(const eamd_ffn_t* p, void* stream) {
  const int rc = check_ffn(p, true);
  if (rc != EAMD_OK) return rc;
  if (p->dtype == 1) return eamd_ffn_bf16_launch(p, 1, stream);
  if (p->hsplit <= 1 && eamd_ffn_f32_sym(p->F) && !p->lnb_x) return eamd_ffn_f32_sym_launch(p, 1, stream);
  return launch_ffn<true, EAMD_ACT_NONE>(*p, (hipStream_t)stream);
}
