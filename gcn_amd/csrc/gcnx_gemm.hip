// Dense fp32 product  C = act(alpha * op(A) * op(B) + beta * C)  for gfx950.
//
// Two kernels behind one entry point, chosen by shape (they never serve the
// same shape):
//   * gemm_mfma_kernel   -- tall products with K and N multiples of 32 and at
//     least 64 wide (hidden >= 64): v_mfma_f32_32x32x2_f32, exact fp32
//     (a k-ordered fmaf chain), operands staged through LDS.
//   * gemm_tile_kernel   -- everything else (hidden < 64, the class dimension,
//     transposed operands): a 64x64x16 register-tiled VALU kernel.
// A product whose reduction dimension dwarfs its output (the weight gradient
// H^T * dP, reduced over all nodes) is split along K across workgroups; each
// split writes its tile to the workspace and gemm_reduce_kernel adds the tiles
// in split order, so the sum is reproducible.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gcnx_common.h"
#include "gcnx_philox.h"

namespace gcnx {

constexpr int TM = 64, TN = 64, TK = 16;

struct GemmArgs {
  int64_t M, N, K;
  const float *A;
  int64_t lda;
  const float *B;
  int64_t ldb;
  float *C;
  int64_t ldc;
  float alpha, beta;
  int relu;
  float *partial;  // [splits][M][N] when splits > 1
  int64_t k_per_split;
  int splits;
  const float *gate;   // nullable: out = gate > 0 ? out * gate_scale : 0
  int64_t ldgate;
  float gate_scale;
  // dropout applied to A while it is staged (keep >= 1: off); element (r, c) of
  // A as stored has Philox index adrop_offset + r * adrop_ld + c
  float adrop_keep;
  uint64_t adrop_seed;
  uint64_t adrop_offset;
  int64_t adrop_ld;
  // nullable: row i of A is row adrop_rows[i] of the matrix the mask is defined
  // on (A holds a gathered subset of its rows)
  const int32_t *adrop_rows;
  // nullable: the gate as one bit per element (bit n % 32 of word
  // gate_bits[m * ldbits + n / 32] set <=> keep)
  const uint32_t *gate_bits;
  int64_t ldbits;
  // dropout on the output behind the activation (keep >= 1: off); element (m, n)
  // has Philox index odrop_offset + m * odrop_ld + odrop_col0 + n
  float odrop_keep;
  uint64_t odrop_seed;
  uint64_t odrop_offset;
  int64_t odrop_ld;
  int64_t odrop_col0;
  int odrop_vec;   // the 16-byte epilogue of the MFMA kernel may be used
  // nullable, with the output dropout only: sign pattern of the stored C, bit
  // (n % 32) of osign[m * ld_osign + n / 32] set <=> stored element (m, n) > 0
  uint32_t *osign;
  int64_t ld_osign;
};

typedef float __attribute__((ext_vector_type(4))) f4v;

// four consecutive elements starting at stored position (r, c), c % 4 == 0
__device__ __forceinline__ f4v drop4(const GemmArgs &g, f4v v, int64_t r, int64_t c) {
  const uint64_t e0 = g.adrop_offset + (uint64_t)r * (uint64_t)g.adrop_ld + (uint64_t)c;
  const philox4 rnd = philox4x32_10(e0 >> 2, 0ull, g.adrop_seed);
  const float inv = 1.0f / g.adrop_keep;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = ((float)(rnd.v[i] >> 8) * (1.0f / 16777216.0f) < g.adrop_keep) ? v[i] * inv : 0.0f;
  return v;
}

__device__ __forceinline__ float gemm_epilogue(const GemmArgs &g, float acc, float *dst, int64_t gm,
                                               int64_t gn) {
  float out = g.alpha * acc;
  if (g.beta != 0.0f) out += g.beta * *dst;
  if (g.relu && out < 0.0f) out = 0.0f;   // a NaN stays a NaN
  if (g.gate) out = g.gate[gm * g.ldgate + gn] > 0.0f ? out * g.gate_scale : 0.0f;
  if (g.gate_bits) out = ((g.gate_bits[gm * g.ldbits + (gn >> 5)] >> (gn & 31)) & 1u) ? out * g.gate_scale : 0.0f;
  return out;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(kBlock) void gemm_tile_kernel(const GemmArgs g) {
  __shared__ float As[TK][TM + 4];
  __shared__ float Bs[TK][TN + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * TM, n0 = (int64_t)blockIdx.y * TN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

  for (int64_t kk = kbeg; kk < kend; kk += TK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * kBlock;
      int m, k;
      if (!TA) { m = e >> 4; k = e & 15; } else { k = e >> 6; m = e & 63; }
      const int64_t gm = m0 + m, gk = kk + k;
      float v = 0.0f;
      if (gm < g.M && gk < kend) v = TA ? g.A[gk * g.lda + gm] : g.A[gm * g.lda + gk];
      As[k][m] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * kBlock;
      int n, k;
      if (!TB) { k = e >> 6; n = e & 63; } else { n = e >> 4; k = e & 15; }
      const int64_t gn = n0 + n, gk = kk + k;
      float v = 0.0f;
      if (gn < g.N && gk < kend) v = TB ? g.B[gn * g.ldb + gk] : g.B[gk * g.ldb + gn];
      Bs[k][n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t gm = m0 + ty * 4 + i;
    if (gm >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gn = n0 + tx * 4 + j;
      if (gn >= g.N) continue;
      if (g.splits > 1) {
        g.partial[((int64_t)blockIdx.z * g.M + gm) * g.N + gn] = acc[i][j];
      } else {
        float *dst = g.C + gm * g.ldc + gn;
        *dst = gemm_epilogue(g, acc[i][j], dst, gm, gn);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void gemm_reduce_kernel(const GemmArgs g) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.M * g.N) return;
  float s = 0.0f;
  for (int z = 0; z < g.splits; ++z) s += g.partial[(int64_t)z * g.M * g.N + e];
  float *dst = g.C + (e / g.N) * g.ldc + (e % g.N);
  *dst = gemm_epilogue(g, s, dst, e / g.N, e % g.N);
}

// First stage of a long reduction: plane j of `stage` = the sum of planes
// [j * kReduceFan, (j + 1) * kReduceFan) of `partial`, added in plane order.
// With hundreds of splits and a small C the single-stage kernel above is a few
// hundred dependent loads on a handful of threads; two stages keep the order of
// the additions fixed (reproducible) and every load coalesced.
constexpr int kReduceFan = 16;
constexpr int kReduceStagedFrom = 32;   // split counts from here on take two stages
__global__ __launch_bounds__(kBlock) void gemm_reduce_stage_kernel(const float *__restrict__ partial,
                                                                    float *__restrict__ stage, int64_t mn,
                                                                    int splits, int planes_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= mn * planes_out) return;
  const int64_t e = t % mn;
  const int j = (int)(t / mn);
  const int z1 = (j + 1) * kReduceFan < splits ? (j + 1) * kReduceFan : splits;
  float s = 0.0f;
  for (int z = j * kReduceFan; z < z1; ++z) s += partial[(int64_t)z * mn + e];
  stage[t] = s;
}

// ---------------------------------------------------------------------------
// MFMA kernel:  C[M,N] = A[M,K] * B[K,N]  (no transposes), N % 32 == 0,
// K % 16 == 0.  Workgroup = 4 wavefronts, block tile 128 x BN (BN = 128);
// wavefront w owns rows [32w, 32w+32) of the block tile and all BN columns:
// BN/32 accumulator tiles of 32x32 (16 registers each).  Per K-step of 16 the
// block stages A[128x16] and B[16xBN] in LDS; one v_mfma_f32_32x32x2_f32 takes
// A[i=lane&31][k=lane>>5] and B[k=lane>>5][j=lane&31] as one float each.
// ---------------------------------------------------------------------------
typedef float __attribute__((ext_vector_type(16))) f16v;

constexpr int MB = 128;

// One 32 x 32 accumulator tile (MFMA C/D layout: col = lane & 31, row = (reg & 3)
// + 8 * (reg >> 2) + 4 * (lane >> 5)) -> wavefront-private LDS -> four rows x
// 16 bytes per lane -> alpha, relu, output dropout (one Philox block per 16
// bytes) -> global.
__device__ __forceinline__ void store_tile_relu_dropout(const GemmArgs &g, const f16v &acc, float *my,
                                                        int pitch, int lane, int64_t row0, int64_t col0) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int rgi = 0; rgi < 16; ++rgi) my[((rgi & 3) + 8 * (rgi >> 2) + 4 * lk) * pitch + li] = acc[rgi];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const float inv = 1.0f / g.odrop_keep;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = p * kWave + lane;
    const int r = idx >> 3, c4 = (idx & 7) * 4;
    const int64_t gm = row0 + r;
    const int64_t gn = col0 + c4;
    if (gm < g.M && gn < g.N) {
      f4v v = *reinterpret_cast<const f4v *>(my + r * pitch + c4) * g.alpha;
      const uint64_t e0 = g.odrop_offset + (uint64_t)gm * (uint64_t)g.odrop_ld + (uint64_t)(g.odrop_col0 + gn);
      const philox4 rnd = philox4x32_10(e0 >> 2, 0ull, g.odrop_seed);
      f4v o;
      o.x = ((float)(rnd.v[0] >> 8) * (1.0f / 16777216.0f) < g.odrop_keep) ? ((g.relu && v.x < 0.0f) ? 0.0f : v.x) * inv : 0.0f;
      o.y = ((float)(rnd.v[1] >> 8) * (1.0f / 16777216.0f) < g.odrop_keep) ? ((g.relu && v.y < 0.0f) ? 0.0f : v.y) * inv : 0.0f;
      o.z = ((float)(rnd.v[2] >> 8) * (1.0f / 16777216.0f) < g.odrop_keep) ? ((g.relu && v.z < 0.0f) ? 0.0f : v.z) * inv : 0.0f;
      o.w = ((float)(rnd.v[3] >> 8) * (1.0f / 16777216.0f) < g.odrop_keep) ? ((g.relu && v.w < 0.0f) ? 0.0f : v.w) * inv : 0.0f;
      *reinterpret_cast<f4v *>(g.C + gm * g.ldc + gn) = o;
      if (g.osign) {
        // the 8 lanes of a row of this 32-column tile hold one sign word between
        // them (rows are whole groups of 8 lanes, so the group is active together)
        int w = ((o.x > 0.0f ? 1 : 0) | (o.y > 0.0f ? 2 : 0) | (o.z > 0.0f ? 4 : 0) | (o.w > 0.0f ? 8 : 0))
                << (4 * (lane & 7));
        w |= __builtin_amdgcn_mov_dpp(w, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        w |= __builtin_amdgcn_mov_dpp(w, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        w |= __builtin_amdgcn_mov_dpp(w, 0x141, 0xF, 0xF, true);   // row_half_mirror: the other quad of the 8
        if ((lane & 7) == 0) g.osign[gm * g.ld_osign + (col0 >> 5)] = (uint32_t)w;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Epilogue shared by the tall MFMA kernels: accumulator tiles of one wavefront
// (TM_W x TN_W tiles of 32 x 32, tile (i, t) at rows (wm * TM_W + i) * 32, columns
// (wn * TN_W + t) * 32 of the BM x BN block tile) -> C.  `smem` is the block's
// staging memory, free once the K loop has ended (one 32 x 36-word tile per
// wavefront is used by the output-dropout path).
template <int TM_W, int TN_W, int BM, int BN>
__device__ __forceinline__ void store_acc_tiles(const GemmArgs &g, const f16v (&acc)[TM_W * TN_W], float *smem,
                                                int w, int wm, int wn, int lane, int64_t m0, int64_t n0) {
  constexpr int NT = TM_W * TN_W;
  constexpr int kTilePitch = 36;
  const int li = lane & 31, lk = lane >> 5;
  // C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
  if (g.odrop_vec) {
    // Output dropout: the tile is turned through LDS so that a lane holds four
    // consecutive columns = one Philox block, and stores 16 bytes.
    float *const my = smem + w * (32 * kTilePitch);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      store_tile_relu_dropout(g, acc[t], my, kTilePitch, lane, m0 + (wm * TM_W + t / TN_W) * 32,
                              n0 + (wn * TN_W + t % TN_W) * 32);
    return;
  }
  if (g.beta == 0.0f && !g.gate && !g.gate_bits && m0 + BM <= g.M && n0 + BN <= g.N) {
    // whole block tile in range, plain epilogue: wavefront-uniform base per (row,
    // tile) plus one per-lane offset, no per-element conditions.  relu is a
    // compare-and-select, so that a NaN accumulator stays NaN in either case (as it
    // does in the per-element epilogue below).
    const int c_off = 4 * lk * (int)g.ldc + li;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float *const base = g.C + (m0 + (wm * TM_W + t / TN_W) * 32) * g.ldc + n0 + (wn * TN_W + t % TN_W) * 32;
#pragma unroll
      for (int rgi = 0; rgi < 16; ++rgi) {
        const float v = g.alpha * acc[t][rgi];
        (base + ((rgi & 3) + 8 * (rgi >> 2)) * g.ldc)[c_off] = (g.relu && v < 0.0f) ? 0.0f : v;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int rgi = 0; rgi < 16; ++rgi) {
      const int row = (rgi & 3) + 8 * (rgi >> 2) + 4 * lk;
      const int64_t gm = m0 + (wm * TM_W + t / TN_W) * 32 + row;
      const int64_t gn = n0 + (wn * TN_W + t % TN_W) * 32 + li;
      if (gm >= g.M || gn >= g.N) continue;
      float *dst = g.C + gm * g.ldc + gn;
      *dst = gemm_epilogue(g, acc[t][rgi], dst, gm, gn);
    }
  }
}

// BN = 128 / 256: wavefront w owns rows [32w, 32w+32) and all BN columns (four
// or eight accumulator tiles); 256 is used when N allows it, so that A is
// streamed (and its fused dropout regenerated) once per 256 output columns.  BN = 32: the class-dimension / skinny case (N <= 32,
// columns beyond N read as zero, one tile per wavefront); such products are
// bound by streaming A (or writing C), the matrix pipe has room to spare.
// TB: B is stored [N, K] (the product with a transposed weight matrix); it is
// transposed while it is staged, so the inner loop is the same.
template <int BN, bool TB>
__global__ __launch_bounds__(kBlock, BN > 128 ? 2 : 3) void gemm_mfma_kernel(const GemmArgs g) {
  // K-step of 16 (32 with the 256-column tile measured 5 % slower: 254 registers
  // and 49 KiB of LDS per workgroup for half as many barriers)
  constexpr int KB = 16;
  constexpr int NA_V = (MB * KB) / (4 * kBlock);   // 16-byte A pieces per thread
  // A tile stored [m][k] with a 17-word pitch: lanes 0..31 of a fragment read
  // stride-17 words (conflict free), lanes 32..63 the next k.
  // wavefront grid: 4 x 1 (each 32 rows x BN columns), or 2 x 2 with the 256-column
  // tile (each 64 x 128: 6 instead of 9 fragment reads per 8 MFMAs)
  constexpr int WN = BN > 128 ? 2 : 1;
  constexpr int WM = kWavesPerBlock / WN;
  constexpr int TM_W = MB / (32 * WM);           // accumulator tiles per wavefront along M
  constexpr int TN_W = BN / (32 * WN);           // ... along N
  constexpr int NT = TM_W * TN_W;
  // after the K loop the same memory holds one 32 x 32 output tile per wavefront
  // (36-word pitch) for the 16-byte epilogue
  constexpr int kTilePitch = 36;
  constexpr int kStage = MB * (KB + 1) + KB * BN;
  constexpr int kSmem = kStage > kWavesPerBlock * 32 * kTilePitch ? kStage : kWavesPerBlock * 32 * kTilePitch;
  __shared__ __attribute__((aligned(16))) float smem[kSmem];
  float *const As = smem;
  float *const Bs = smem + MB * (KB + 1);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w / WN, wn = w % WN;
  const int64_t m0 = (int64_t)blockIdx.x * MB;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  f16v acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  const int li = lane & 31, lk = lane >> 5;
  // Software pipeline: the global loads of K-step i+1 are issued before the
  // MFMAs of K-step i and land in registers; they are written to LDS (A with
  // its dropout applied) after those MFMAs have been issued.
  constexpr int NB_S = (KB * BN) / kBlock;   // scalar B elements per thread (generic paths)
  constexpr int NB_V = NB_S / 4;             // 16-byte B pieces per thread (plain B, BN >= 128)
  f4v ra[NA_V], rb[NB_V > 0 ? NB_V : 1];
  float rbs[(TB || BN < 128) ? NB_S : 1];
  auto load_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < NA_V; ++i) {
      const int e = tid + i * kBlock;
      const int m = e / (KB / 4), q = e % (KB / 4);
      // rows past the end read the last row instead of branching: their results
      // are never stored
      const int64_t gm = m0 + m < g.M ? m0 + m : g.M - 1;
      ra[i] = *reinterpret_cast<const f4v *>(g.A + gm * g.lda + kk + q * 4);
    }
    if constexpr (!TB && BN >= 128) {
#pragma unroll
      for (int i = 0; i < NB_V; ++i) {
        const int e = tid + i * kBlock;
        const int k = e / (BN / 4), q = e % (BN / 4);
        rb[i] = *reinterpret_cast<const f4v *>(g.B + (kk + k) * g.ldb + n0 + q * 4);
      }
    } else if constexpr (!TB) {
#pragma unroll
      for (int i = 0; i < NB_S; ++i) {
        const int e = tid + i * kBlock;
        const int k = e / BN, n = e % BN;
        rbs[i] = (n0 + n < g.N) ? g.B[(kk + k) * g.ldb + n0 + n] : 0.0f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB_S; ++i) {
        const int e = tid + i * kBlock;
        const int n = e / KB, k = e % KB;
        rbs[i] = (n0 + n < g.N) ? g.B[(n0 + n) * g.ldb + kk + k] : 0.0f;
      }
    }
  };
  auto store_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < NA_V; ++i) {
      const int e = tid + i * kBlock;
      const int m = e / (KB / 4), q = e % (KB / 4);
      f4v v = ra[i];
      if (g.adrop_keep < 1.0f) {
        const int64_t gm = m0 + m < g.M ? m0 + m : g.M - 1;
        v = drop4(g, v, g.adrop_rows ? (int64_t)g.adrop_rows[gm] : gm, kk + q * 4);
      }
      float *d = As + m * (KB + 1) + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if constexpr (!TB && BN >= 128) {
#pragma unroll
      for (int i = 0; i < NB_V; ++i) {
        const int e = tid + i * kBlock;
        const int k = e / (BN / 4), q = e % (BN / 4);
        *reinterpret_cast<f4v *>(Bs + k * BN + q * 4) = rb[i];
      }
    } else if constexpr (!TB) {
#pragma unroll
      for (int i = 0; i < NB_S; ++i) {
        const int e = tid + i * kBlock;
        Bs[(e / BN) * BN + (e % BN)] = rbs[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB_S; ++i) {
        const int e = tid + i * kBlock;
        Bs[(e % KB) * BN + (e / KB)] = rbs[i];
      }
    }
  };
  load_tiles(0);
  for (int64_t kk = 0; kk < g.K; kk += KB) {
    store_tiles(kk);
    __syncthreads();
    if (kk + KB < g.K) load_tiles(kk + KB);
#pragma unroll
    for (int k2 = 0; k2 < KB; k2 += 2) {
      if constexpr (BN >= 128) {
        float af[TM_W], bf[TN_W];
#pragma unroll
        for (int i = 0; i < TM_W; ++i) af[i] = As[((wm * TM_W + i) * 32 + li) * (KB + 1) + k2 + lk];
#pragma unroll
        for (int t = 0; t < TN_W; ++t) bf[t] = Bs[(k2 + lk) * BN + (wn * TN_W + t) * 32 + li];
#pragma unroll
        for (int i = 0; i < TM_W; ++i)
#pragma unroll
          for (int t = 0; t < TN_W; ++t)
            acc[i * TN_W + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[t], acc[i * TN_W + t], 0, 0, 0);
      } else {
        const float a = As[(w * 32 + li) * (KB + 1) + k2 + lk];
        const float b = Bs[(k2 + lk) * BN + li];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  store_acc_tiles<TM_W, TN_W, MB, BN>(g, acc, smem, w, wm, wn, lane, m0, n0);
}

// ---------------------------------------------------------------------------
// The tall product on the bf16 matrix pipe at fp32 accuracy ("bf16 x 3").
//
// On gfx950 the f32-input MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD,
// 1/16 of the bf16 MFMA) and does not overlap with vector instructions, so an
// fp32 product on it is bound by the vector ALUs.  Here every fp32 operand x is
// split EXACTLY into three bf16 pieces, x = h + m + l (three times 8 significant
// bits = the 24 of an fp32: h = the top 16 bits of x, m = the top 16 bits of
// x - h, l = x - h - m, all subtractions exact), and a product a * b is formed
// from the six partial products of weight >= 2^-16,
//     ah*bh + am*bh + al*bh + ah*bm + am*bm + ah*bl,
// each one exact in fp32 (8 x 8 significant bits), accumulated in fp32 inside
// v_mfma_f32_32x32x16_bf16.  What is dropped (am*bl + al*bm + al*bl) is below
// 2^-23 of |a*b|, the size of the rounding of one fp32 product.  Six bf16 MFMAs of 32
// cycles replace eight f32 MFMAs of 64 cycles per 16 values of k, and they
// co-execute with the vector instructions (Philox, splitting, addressing).
//
// Block tile 128 x 256, K step 16 = one MFMA; the B tile [n][16 k] lies in LDS as
// bf16 rows of 32 bytes, three pieces, with an XOR swizzle of the two 16-byte
// halves of a row (16-byte fragment reads of 16 consecutive rows then fall on 16
// different bank groups).  B -- the weights -- is split once by split_b_kernel
// into the workspace, laid out so that the B tile of a K step is one contiguous
// block: Bsp[k / 16][piece][n][k % 16].  A is split by the wavefront that
// multiplies it, after its fused dropout (see gemm_split_nn_kernel).
// ---------------------------------------------------------------------------
typedef __bf16 __attribute__((ext_vector_type(8))) bf8v;
typedef int __attribute__((ext_vector_type(4))) i4v;
typedef unsigned __attribute__((ext_vector_type(2))) u2v;

// x = h + m + l exactly; each piece is a bf16 held in the upper half of a word
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(h);
  m = __float_as_uint(r) & 0xffff0000u;
  l = __float_as_uint(r - __uint_as_float(m));   // at most 8 significant bits: its lower half is zero
}
// two upper halves -> one word (lo in bits 0..15)
__device__ __forceinline__ unsigned pack_upper(unsigned lo, unsigned hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

__global__ __launch_bounds__(kBlock) void split_b_kernel(const float *__restrict__ B, int64_t ldb, int trans_b,
                                                         int64_t K, int64_t N, unsigned short *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * N) return;
  // thread e -> (k / 16, n, k % 16): writes are contiguous per piece
  const int kk = (int)(e & 15);
  const int64_t n = (e >> 4) % N, ks = (e >> 4) / N;
  const int64_t k = ks * 16 + kk;
  const float x = trans_b ? B[n * ldb + k] : B[k * ldb + n];
  unsigned h, m, l;
  split3(x, h, m, l);
  const int64_t base = (ks * 3 * N + n) * 16 + kk;
  out[base] = (unsigned short)(h >> 16);
  out[base + N * 16] = (unsigned short)(m >> 16);
  out[base + 2 * N * 16] = (unsigned short)(l >> 16);
}

// The K loop keeps the matrix pipe fed from one wavefront's own instruction
// stream: while the 48 MFMAs of K step s run, the same wavefront turns the raw
// fp32 rows of step s + 1 into bf16 fragments (dropout, splitting: vector
// instructions, which co-execute with bf16 MFMAs).  Wavefront w owns rows
// 32 w .. 32 w + 31 of the tile and all 256 columns, so the A fragments of a
// wavefront are nobody else's: they never pass through a shared tile.
//   * The raw rows of A travel by LDS-DMA (global_load_lds_dwordx4: no
//     registers) into a landing ring private to the wavefront (3 slots of 32
//     rows x 64 bytes) and are read back by the same wavefront in MFMA fragment
//     order: only the wavefront's own in-order load counter is involved.
//   * The weight tile of step s + 1 travels by LDS-DMA from the workspace straight
//     into the second of two B tiles, in its final layout; one barrier per step
//     publishes it.
// The DMA instructions are inline assembly, so the compiler inserts no waits of
// its own for them; the one wait per step is written out, its count derived
// from the issue order, which is the same in every step and wavefront:
//   [6 x B(s+1)] [2 x A(s+3)]
// so that "at most 2 outstanding" before the barrier of step s means B(s) and
// A(s+1) have landed and only A(s+2) may still be in flight.
constexpr int kNnBM = 128, kNnBN = 256, kNnKB = 16;
constexpr int kNnBPiece = kNnBN * 32;                           // bytes of one bf16 piece of the B tile
constexpr int kNnBTile = 3 * kNnBPiece;
constexpr int kNnRingSlots = 3, kNnRingSlot = 2048;             // 32 raw rows of one K step
constexpr int kNnLds = 2 * kNnBTile + 4 * kNnRingSlots * kNnRingSlot;   // 72 KiB: two workgroups per CU
// s_waitcnt immediates (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt 6:4
// and lgkmcnt 11:8 left at "no wait")
constexpr int kWaitVm0 = 0x0F70, kWaitVm2 = 0x0F72;

__device__ __forceinline__ unsigned lds_offset(const void *p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char *)p);
}
// 64 lanes x 16 bytes from (uniform base + per-lane byte offset) to LDS [lds_base, lds_base + 1 KiB).
// M0 (the LDS base of the DMA) is a reserved register: the compiler keeps no value in it across
// instructions and rewrites it before every use of its own, so the write here needs no clobber entry.
__device__ __forceinline__ void dma16(const void *base, unsigned lane_off, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base), "s"(lds_base)
               : "memory");   // M0 is a reserved register: the compiler sets it before each use of its own
}
__device__ __forceinline__ void block_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// 16-byte unit of (row r, half hf) inside a split tile of 32-byte rows: the XOR puts
// the 16 rows that a 16-lane group of ds_read_b128 touches on 16 different groups of four banks
__device__ __forceinline__ int split_unit(int r, int hf) { return 2 * r + (hf ^ ((r >> 3) & 1)); }
// 16-byte unit of (row r, floats 4 q ..) inside a ring slot of 32 raw rows, for the same reason
__device__ __forceinline__ int ring_unit(int r, int q) { return 4 * r + (q ^ ((r >> 2) & 3)); }

template <bool DROP>   // the Philox dropout of A is compiled in or out: the K loop stays one basic block
__global__ __launch_bounds__(kBlock, 2) void gemm_split_nn_kernel(const GemmArgs g,
                                                                  const unsigned short *__restrict__ Bsp) {
  constexpr int KB = kNnKB, kBPiece = kNnBPiece;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kNnLds];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * kNnBM;
  const int64_t n0 = (int64_t)blockIdx.y * kNnBN;
  const int nk = (int)(g.K / KB);
  f16v acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  // A: DMA instruction i of the wavefront fills units 64 i .. 64 i + 63 of a ring
  // slot; the lane fetches the piece (row, q) that belongs in its unit
  unsigned a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = w * 32 + i * 16 + (lane >> 2);
    const int q = (lane & 3) ^ ((lane >> 4) & 3);
    const int64_t rc = m0 + r < g.M ? r : g.M - 1 - m0;        // rows past the end: never stored
    a_off[i] = (unsigned)(rc * g.lda * 4 + q * 16);
  }
  // the fragment of this lane: row li of the wavefront's 32, floats 8 lk .. 8 lk + 7 of the step's 16
  const int64_t frow = m0 + w * 32 + li < g.M ? m0 + w * 32 + li : g.M - 1;
  const int64_t drow = DROP && g.adrop_rows ? (int64_t)g.adrop_rows[frow] : frow;
  // every load issued so far (the row id) completes here, where the compiler sees it
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  const float *const a_tile = g.A + m0 * g.lda;
  unsigned char *const ring = smem + 2 * kNnBTile + w * (kNnRingSlots * kNnRingSlot);
  const unsigned ring_lds = lds_offset(ring);
  // Steps past the end are issued all the same (clamped to the last step: L2
  // hits, landing in a slot nobody reads any more), so that every step issues
  // the same eight instructions and the loop body has no branch.
  auto issue_a = [&](int step, int slot) {
    const float *base = a_tile + (int64_t)(step < nk ? step : nk - 1) * KB;
    const unsigned dst = ring_lds + (unsigned)(slot * kNnRingSlot);
    dma16(base, a_off[0], dst);
    dma16(base, a_off[1], dst + 1024);
  };
  // weight tile: 24 DMA instructions of 64 units, six per wavefront; instruction
  // id = (piece, block of 32 columns); the lane picks the half-row that belongs in its unit
  const unsigned b_off = (unsigned)((lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
  const unsigned btile_lds = lds_offset(smem);
  const unsigned char *const b_src = reinterpret_cast<const unsigned char *>(Bsp) + n0 * 32;
  auto issue_b = [&](int step) {
    const unsigned dst = btile_lds + (unsigned)((step & 1) * kNnBTile);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int id = w * 6 + j, p = id >> 3, jj = id & 7;
      dma16(b_src + ((int64_t)step * 3 + p) * g.N * 32 + jj * 1024, b_off, dst + (unsigned)(p * kBPiece + jj * 1024));
    }
  };
  const unsigned char *const raw_lo = ring + ring_unit(li, 2 * lk) * 16;
  const unsigned char *const raw_hi = ring + ring_unit(li, 2 * lk + 1) * 16;
  // raw floats of one step -> dropout -> the three bf16 A fragments of this lane
  auto prep = [&](i4v (&af)[3], int step, int slot) {
    f4v v[2];
    v[0] = *reinterpret_cast<const f4v *>(raw_lo + slot * kNnRingSlot);
    v[1] = *reinterpret_cast<const f4v *>(raw_hi + slot * kNnRingSlot);
    const int64_t kk = (int64_t)step * KB + lk * 8;
    unsigned h[8], md[8], l[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (DROP) v[i] = drop4(g, v[i], drow, kk + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(v[i][j], h[i * 4 + j], md[i * 4 + j], l[i * 4 + j]);
    }
    af[0] = i4v{(int)pack_upper(h[0], h[1]), (int)pack_upper(h[2], h[3]), (int)pack_upper(h[4], h[5]),
                (int)pack_upper(h[6], h[7])};
    af[1] = i4v{(int)pack_upper(md[0], md[1]), (int)pack_upper(md[2], md[3]), (int)pack_upper(md[4], md[5]),
                (int)pack_upper(md[6], md[7])};
    af[2] = i4v{(int)pack_upper(l[0], l[1]), (int)pack_upper(l[2], l[3]), (int)pack_upper(l[4], l[5]),
                (int)pack_upper(l[6], l[7])};
  };

  // prologue: A(0), A(1), B(0), A(2); all but A(2) have landed at "2 outstanding"
  issue_a(0, 0);
  issue_a(1, 1);
  issue_b(0);
  issue_a(2, 2);
  __builtin_amdgcn_s_waitcnt(kWaitVm2);
  i4v af[3];
  prep(af, 0, 0);
  const int b_frag = split_unit(li, lk) * 16;
  int slot = 0;                           // ring slot of the raw rows of `step`
  auto one_step = [&](int step, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;    // another step follows
    // B(step) and A(step + 1) have landed once only A(step + 2) (2 instructions) may remain
    __builtin_amdgcn_s_waitcnt(kWaitVm2);
    block_barrier_lds();                  // B tile of this step visible; the other B tile is free
    const int next = slot + 1 == kNnRingSlots ? 0 : slot + 1;
    i4v afn[3];
    if (MORE) {
      issue_b(step + 1);
      issue_a(step + 3, slot);            // this slot's rows were consumed a step ago
      prep(afn, step + 1, next);
    }
    const int bt = b_frag + (step & 1) * kNnBTile;
#pragma unroll
    for (int pb = 0; pb < 3; ++pb)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        i4v bf[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          bf[tn] = *reinterpret_cast<const i4v *>(smem + bt + pb * kBPiece + (half * 4 + tn) * 32 * 32);
#pragma unroll
        for (int pa = 0; pa + pb < 3; ++pa)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn)
            acc[half * 4 + tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf8v, af[pa]), __builtin_bit_cast(bf8v, bf[tn]), acc[half * 4 + tn], 0, 0, 0);
      }
    if (MORE) {
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = afn[p];
      slot = next;
    }
  };
  for (int step = 0; step + 1 < nk; ++step) one_step(step, std::true_type{});
  one_step(nk - 1, std::false_type{});
  __builtin_amdgcn_s_waitcnt(kWaitVm0);   // the clamped DMAs of the last steps
  __syncthreads();            // the epilogue may reuse the staging memory
  store_acc_tiles<1, 8, kNnBM, kNnBN>(g, acc, reinterpret_cast<float *>(smem), w, w, 0, lane, m0, n0);
}

// ---------------------------------------------------------------------------
// MFMA kernel for the weight gradients  C[M,N] = A^T * B  with A stored [K, M],
// B stored [K, N] and K = the node dimension (huge), split along K over
// grid.z.  The A operand of v_mfma_f32_32x32x2_f32 wants A_op[i][k] = A[k][i]:
// for a fixed k the 32 values are contiguous in memory, so both operands are
// staged in LDS exactly as they lie in HBM ([k][m] and [k][n]) and fragment
// reads are conflict free.  Block tile 128 x BN: BN = 128 / 256 -> wavefronts
// 2 x 2, each 64 x 64 / 64 x 128 (four / eight accumulator tiles); BN = 32 -> wavefronts 4 x 1, each
// 32 x 32 (the class-dimension case, N <= 32, columns beyond N read as zero).
// Every split writes its tile to the workspace; gemm_reduce_kernel adds the
// splits in order.
// ---------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(kBlock, BN > 128 ? 2 : 3) void gemm_mfma_tn_kernel(const GemmArgs g) {
  constexpr int BM = 128, BK = 16;
  constexpr int WN = BN >= 128 ? 2 : 1;          // wavefronts along N
  constexpr int WM = 4 / WN;                     // wavefronts along M
  constexpr int TMW = BM / (32 * WM);            // accumulator tiles per wavefront along M
  constexpr int TNW = BN / (32 * WN);
  constexpr int NBV = BN >= 128 ? BN / 64 : 1;   // 16-byte B pieces per thread
  __shared__ float As[BK * BM];
  __shared__ float Bs[BK * BN];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w / WN, wn = w % WN;
  const int li = lane & 31, lk = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  f16v acc[TMW][TNW];
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  // software pipeline as in gemm_mfma_kernel: next K-step's tiles travel in
  // registers while this K-step's MFMAs run
  f4v ra[2], rb[NBV];
  float rbs[2];
  auto load_tiles = [&](int64_t kk) {
    // a whole K-step inside the split (all but possibly the last one): no
    // per-row conditions (wavefront-uniform test)
    const bool whole = kk + BK <= kend;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      if (whole) {
        ra[i] = *reinterpret_cast<const f4v *>(g.A + (kk + k) * g.lda + m0 + q * 4);
      } else {
        ra[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (kk + k < kend) ra[i] = *reinterpret_cast<const f4v *>(g.A + (kk + k) * g.lda + m0 + q * 4);
      }
      if constexpr (BN < 128)
        rbs[i] = (kk + k < kend && n0 + q < g.N) ? g.B[(kk + k) * g.ldb + n0 + q] : 0.0f;
    }
    if constexpr (BN >= 128) {
#pragma unroll
      for (int i = 0; i < NBV; ++i) {
        const int e = tid + i * kBlock;
        const int k = e / (BN / 4), q = e % (BN / 4);
        if (whole) {
          rb[i] = *reinterpret_cast<const f4v *>(g.B + (kk + k) * g.ldb + n0 + q * 4);
        } else {
          rb[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
          if (kk + k < kend) rb[i] = *reinterpret_cast<const f4v *>(g.B + (kk + k) * g.ldb + n0 + q * 4);
        }
      }
    }
  };
  auto store_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      f4v v = ra[i];
      if (g.adrop_keep < 1.0f) v = drop4(g, v, kk + k, m0 + q * 4);   // rows past kend hold zeros
      *reinterpret_cast<f4v *>(As + k * BM + q * 4) = v;
      if constexpr (BN < 128) Bs[k * BN + q] = rbs[i];
    }
    if constexpr (BN >= 128) {
#pragma unroll
      for (int i = 0; i < NBV; ++i) {
        const int e = tid + i * kBlock;
        const int k = e / (BN / 4), q = e % (BN / 4);
        *reinterpret_cast<f4v *>(Bs + k * BN + q * 4) = rb[i];
      }
    }
  };
  if (kbeg < kend) load_tiles(kbeg);
  for (int64_t kk = kbeg; kk < kend; kk += BK) {
    store_tiles(kk);
    __syncthreads();
    if (kk + BK < kend) load_tiles(kk + BK);
#pragma unroll
    for (int k2 = 0; k2 < BK; k2 += 2) {
      float af[TMW], bf[TNW];
#pragma unroll
      for (int a = 0; a < TMW; ++a) af[a] = As[(k2 + lk) * BM + (wm * TMW + a) * 32 + li];
#pragma unroll
      for (int b = 0; b < TNW; ++b) bf[b] = Bs[(k2 + lk) * BN + (wn * TNW + b) * 32 + li];
#pragma unroll
      for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b = 0; b < TNW; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < TNW; ++b)
#pragma unroll
      for (int rgi = 0; rgi < 16; ++rgi) {
        const int row = (rgi & 3) + 8 * (rgi >> 2) + 4 * lk;
        const int64_t gm = m0 + (wm * TMW + a) * 32 + row;
        const int64_t gn = n0 + (wn * TNW + b) * 32 + li;
        if (gm >= g.M || gn >= g.N) continue;
        if (g.splits > 1) {
          g.partial[((int64_t)blockIdx.z * g.M + gm) * g.N + gn] = acc[a][b][rgi];
        } else {
          float *dst = g.C + gm * g.ldc + gn;
          *dst = gemm_epilogue(g, acc[a][b][rgi], dst, gm, gn);
        }
      }
}

// ---------------------------------------------------------------------------
// The weight gradient  C[M,N] = A^T * B  (A stored [K, M], B stored [K, N], K =
// the node dimension, split over grid.z) on the bf16 matrix pipe with exactly
// split operands, as gemm_split_nn_kernel.  Both operands are node-major, i.e.
// k-major, while a lane of v_mfma_f32_32x32x16_bf16 wants 8 consecutive k of ONE
// column: the tiles are stored in LDS as they lie in memory ([k][m], [k][n], bf16)
// and read with ds_read_b64_tr_b16, which hands each lane of a 16-lane group one
// column of a 4 (k) x 16 block.  Row pitch = row bytes + 64, so the four k rows a
// 32-lane half reads at once fall on four different quarters of the 64 banks.
// Block tile 128 x 256, wavefronts 2 x 2 each 64 x 128, K step 16.
// ---------------------------------------------------------------------------
typedef short __attribute__((ext_vector_type(4))) s4v;
typedef short __attribute__((ext_vector_type(8))) s8v;

__device__ __forceinline__ bf8v tr_frag(const unsigned char *p, int row_pitch) {
  // rows k .. k+3 at p, rows k+4 .. k+7 four rows further down
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v *)(p));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v *)(p + 4 * row_pitch));
  const s8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf8v, v);
}

// ROWS: row k of A draws the dropout mask of row g.adrop_rows[k] (A holds a gathered
// subset of the rows the mask is defined on); the ids travel two K steps ahead of
// the draw that needs them, next to the tile loads, so that no wait for them falls
// between the loads and the MFMAs.
template <bool DROP, bool ROWS = false>   // the Philox dropout of A compiled in or out
__global__ __launch_bounds__(kBlock, 2) void gemm_split_tn_kernel(const GemmArgs g) {
  static_assert(DROP || !ROWS, "row ids only select dropout masks");
  constexpr int BM = 128, BN = 256, BK = 16;
  constexpr int PA = BM * 2 + 64, PB = BN * 2 + 64;           // row pitches in bytes
  constexpr int kAPiece = BK * PA, kBPiece = BK * PB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kAPiece + 3 * kBPiece];
  unsigned char *const As = smem;
  unsigned char *const Bs = smem + 3 * kAPiece;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lk = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  f16v acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  f4v ra[2], rb[4];
  auto load_tiles = [&](int64_t kk) {
    const bool whole = kk + BK <= kend;     // all but possibly the last K step of a split
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      ra[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      if (whole || kk + k < kend) ra[i] = *reinterpret_cast<const f4v *>(g.A + (kk + k) * g.lda + m0 + q * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 6, q = e & 63;
      rb[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      if (whole || kk + k < kend) rb[i] = *reinterpret_cast<const f4v *>(g.B + (kk + k) * g.ldb + n0 + q * 4);
    }
  };
  // The dropout mask of the A pieces this thread stages at step kk is a function
  // of indices only: it is drawn (Philox, ~150 vector instructions) in the shadow
  // of the previous step's MFMAs and kept as 8 bits.
  unsigned keep_bits = 0;
  int rid_cur[2] = {0, 0}, rid_nxt[2] = {0, 0};    // ROWS: mask rows of the next draw / of the one after
  auto load_rids = [&](int64_t kk, int (&dst)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t r = kk + ((tid + i * kBlock) >> 5);
      dst[i] = r < kend ? g.adrop_rows[r] : 0;
    }
  };
  auto draw_keep = [&](int64_t kk) {
    keep_bits = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      const uint64_t mask_row = ROWS ? (uint64_t)(int64_t)rid_cur[i] : (uint64_t)(kk + k);
      const uint64_t e0 = g.adrop_offset + mask_row * (uint64_t)g.adrop_ld + (uint64_t)(m0 + q * 4);
      const philox4 rnd = philox4x32_10(e0 >> 2, 0ull, g.adrop_seed);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        keep_bits |= ((float)(rnd.v[j] >> 8) * (1.0f / 16777216.0f) < g.adrop_keep ? 1u : 0u) << (4 * i + j);
    }
  };
  auto put3 = [&](unsigned char *d, int piece_bytes, f4v v) {
    unsigned h[4], md[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[j], h[j], md[j], l[j]);
    *reinterpret_cast<u2v *>(d) = u2v{pack_upper(h[0], h[1]), pack_upper(h[2], h[3])};
    *reinterpret_cast<u2v *>(d + piece_bytes) = u2v{pack_upper(md[0], md[1]), pack_upper(md[2], md[3])};
    *reinterpret_cast<u2v *>(d + 2 * piece_bytes) = u2v{pack_upper(l[0], l[1]), pack_upper(l[2], l[3])};
  };
  auto store_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      f4v v = ra[i];
      if (DROP) {
        // keep bits drawn while the previous step's MFMAs ran (draw_keep); rows past kend hold zeros
        const float inv = 1.0f / g.adrop_keep;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((keep_bits >> (4 * i + j)) & 1u) ? v[j] * inv : 0.0f;
      }
      put3(As + k * PA + q * 8, kAPiece, v);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 6, q = e & 63;
      put3(Bs + k * PB + q * 8, kBPiece, rb[i]);
    }
  };
  // transposed-read address of this lane inside a 32-column operand tile: lane
  // 4q + p of a 16-lane group supplies row q, columns 4p .. 4p+3 of its block;
  // groups 0, 1 take k 0..7 of columns 0..15 / 16..31, groups 2, 3 k 8..15
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int a_tr = ((grp >> 1) * 8 + gq) * PA + ((wm * 2) * 32 + (grp & 1) * 16 + gp * 4) * 2;
  const int b_tr = ((grp >> 1) * 8 + gq) * PB + ((wn * 4) * 32 + (grp & 1) * 16 + gp * 4) * 2;
  (void)li; (void)lk;
  if (kbeg < kend) load_tiles(kbeg);
  if (ROWS && kbeg < kend) {
    load_rids(kbeg, rid_cur);
    load_rids(kbeg + BK, rid_nxt);
  }
  if (DROP && kbeg < kend) draw_keep(kbeg);
  if (ROWS) { rid_cur[0] = rid_nxt[0]; rid_cur[1] = rid_nxt[1]; }
  for (int64_t kk = kbeg; kk < kend; kk += BK) {
    store_tiles(kk);
    __syncthreads();
    if (kk + BK < kend) load_tiles(kk + BK);
    if (ROWS) load_rids(kk + 2 * BK, rid_nxt);     // past the end: zeros, no load
    if (DROP) draw_keep(kk + BK);           // past the last step: drawn and not used (no branch in the MFMA block)
    if (ROWS) { rid_cur[0] = rid_nxt[0]; rid_cur[1] = rid_nxt[1]; }
    bf8v af[2][3];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[tm][p] = tr_frag(As + p * kAPiece + a_tr + tm * 64, PA);
#pragma unroll
    for (int pb = 0; pb < 3; ++pb) {
      bf8v bf[4];
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) bf[tn] = tr_frag(Bs + pb * kBPiece + b_tr + tn * 64, PB);
#pragma unroll
      for (int pa = 0; pa + pb < 3; ++pa)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn)
            acc[tm * 4 + tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][pa], bf[tn], acc[tm * 4 + tn], 0, 0, 0);
    }
    __syncthreads();
  }
  const int lk2 = lane >> 5, li2 = lane & 31;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int rgi = 0; rgi < 16; ++rgi) {
      const int row = (rgi & 3) + 8 * (rgi >> 2) + 4 * lk2;
      const int64_t gm = m0 + (wm * 2 + t / 4) * 32 + row;
      const int64_t gn = n0 + (wn * 4 + t % 4) * 32 + li2;
      if (gm >= g.M || gn >= g.N) continue;
      if (g.splits > 1) {
        g.partial[((int64_t)blockIdx.z * g.M + gm) * g.N + gn] = acc[t][rgi];
      } else {
        float *dst = g.C + gm * g.ldc + gn;
        *dst = gemm_epilogue(g, acc[t][rgi], dst, gm, gn);
      }
    }
}

// ---------------------------------------------------------------------------
// Streaming kernel for a tiny inner dimension (K <= 32) and a wide output:
// C[M,N] = A[M,K] * op(B), e.g. dP2 * W2^T with K = the class count.  The product
// is bound by writing C (and reading the gate): lane l keeps its 4 output
// columns' slice of op(B) (4 x K values) in registers, one wavefront produces one
// full row of a 256-column slab per step with 16-byte loads / stores (1 KiB
// coalesced per instruction), and the K values of the A row are wave-uniform
// (scalar loads).  Four rows are in flight per wavefront.
// ---------------------------------------------------------------------------
template <int KMAX, bool TB>
__global__ __launch_bounds__(kBlock) void gemm_small_k_kernel(const GemmArgs g) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t col0 = (int64_t)blockIdx.y * 256 + lane * 4;
  const bool active = col0 < g.N;   // N % 4 == 0
  float bw[KMAX][4];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = 0.0f;
      if (active && k < g.K) v = TB ? g.B[(col0 + j) * g.ldb + k] : g.B[(int64_t)k * g.ldb + col0 + j];
      bw[k][j] = v;
    }
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  constexpr int R = kWave / KMAX;          // rows whose A values fit one value per lane
  const int arow_l = lane / KMAX, ak_l = lane % KMAX;
  for (int64_t r0 = wave * R; r0 < g.M; r0 += nwaves * R) {
    // one coalesced load brings the K values of R rows, one per lane; they are
    // broadcast below with v_readlane (scalar operands of the FMAs)
    float aval = 0.0f;
    if (r0 + arow_l < g.M && ak_l < g.K) aval = g.A[(r0 + arow_l) * g.lda + ak_l];
    f4v acc[R], gt[R], old[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      acc[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      gt[i] = f4v{1.0f, 1.0f, 1.0f, 1.0f};
      old[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      if (r < g.M && active) {
        if (g.gate) gt[i] = *reinterpret_cast<const f4v *>(g.gate + r * g.ldgate + col0);
        if (g.beta != 0.0f) old[i] = *reinterpret_cast<const f4v *>(g.C + r * g.ldc + col0);
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(aval), i * KMAX + k));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a, bw[k][j], acc[i][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      if (r < g.M && active) {
        f4v out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float o = g.alpha * acc[i][j] + g.beta * old[i][j];
          if (g.relu && o < 0.0f) o = 0.0f;   // a NaN stays a NaN
          if (g.gate) o = gt[i][j] > 0.0f ? o * g.gate_scale : 0.0f;
          out[j] = o;
        }
        *reinterpret_cast<f4v *>(g.C + r * g.ldc + col0) = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The same product (K <= 16, wide output) on the matrix pipe: with the FMAs on
// the VALU the kernel above spends about as long computing as streaming; here
// the VALU only runs the epilogue.  One wavefront owns 32 rows and walks the
// output in 32 x 32 tiles (v_mfma_f32_32x32x2_f32, 8 steps cover K = 16).
// A is read as it lies in memory, one 16-byte load per 8 values of k:
// lane (i = lane & 31, h = lane >> 5) holds A[i][4h..4h+3] and A[i][8+4h..8+4h+3],
// so step s = 4*half + j multiplies k = 8*half + 4h + j; the op(B) fragments
// (kept in registers for the whole kernel) use the same k numbering.  The sum
// over k inside one accumulator is order-free, so this is the same fp32 result
// up to the order of the 16 terms.  A tile is stored with 128-byte row
// segments (lanes 0..31 one row, lanes 32..63 the row 4 below).
// ---------------------------------------------------------------------------
typedef float __attribute__((ext_vector_type(16))) f16v_;

// GATE: 0 none, 1 float gate, 2 bit gate (a template parameter so that the
// unrolled epilogue of a full tile carries no branches at all).
template <bool TB, int GATE>
__global__ __launch_bounds__(kBlock, 2) void gemm_small_k_mfma_kernel(const GemmArgs g) {
  constexpr int NT = 8;  // 32-column tiles of one 256-column slab
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n0 = (int64_t)blockIdx.y * 256;
  float bf[NT][8];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = 8 * (s >> 2) + 4 * lh + (s & 3);
      const int64_t n = n0 + 32 * t + li;
      float v = 0.0f;
      if (k < g.K && n < g.N) v = TB ? g.B[n * g.ldb + k] : g.B[(int64_t)k * g.ldb + n];
      bf[t][s] = v;
    }
  const bool vec_a = (g.K == 16) && (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t r0 = wave * 32; r0 < g.M; r0 += nwaves * 32) {
    float af[8];
    {
      const int64_t row = r0 + li;
      f4v a0 = f4v{0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
      if (row < g.M) {
        const float *ap = g.A + row * g.lda;
        if (vec_a) {
          a0 = *reinterpret_cast<const f4v *>(ap + 4 * lh);
          a1 = *reinterpret_cast<const f4v *>(ap + 8 + 4 * lh);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (4 * lh + j < g.K) a0[j] = ap[4 * lh + j];
            if (8 + 4 * lh + j < g.K) a1[j] = ap[8 + 4 * lh + j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { af[j] = a0[j]; af[4 + j] = a1[j]; }
    }
    // bit gate: lane i fetches the words of row r0 + i once (one word per tile);
    // a tile then takes its rows' words with v_readlane
    uint32_t wb[NT];
    if constexpr (GATE == 2) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        wb[t] = 0u;
        if (r0 + li < g.M && n0 + 32 * t < g.N) wb[t] = g.gate_bits[(r0 + li) * g.ldbits + ((n0 + 32 * t) >> 5)];
      }
    }
    // Addresses are a wavefront-uniform base (row and tile) plus one per-lane
    // 32-bit offset that is the same for every row and tile.
    const int c_off = 4 * lh * (int)g.ldc + li;
    const int g_off = 4 * lh * (int)g.ldgate + li;
    const bool rows_full = r0 + 32 <= g.M && g.beta == 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int64_t tc0 = n0 + 32 * t;
      if (tc0 >= g.N) break;
      f16v_ acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
      if (rows_full && tc0 + 32 <= g.N) {
        // full tile, beta == 0: no per-element conditions
        float gt[16];
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int ro = (rg & 3) + 8 * (rg >> 2);
          if constexpr (GATE == 1) gt[rg] = (g.gate + (r0 + ro) * g.ldgate + tc0)[g_off];
          if constexpr (GATE == 2) {
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wb[t], ro);
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)wb[t], ro + 4);
            gt[rg] = (((lh ? w1 : w0) >> li) & 1u) ? 1.0f : 0.0f;
          }
          if constexpr (GATE == 0) gt[rg] = 1.0f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[t][s], acc, 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
          const int ro = (rg & 3) + 8 * (rg >> 2);
          float o = g.alpha * acc[rg];
          if (g.relu && o < 0.0f) o = 0.0f;   // a NaN stays a NaN
          if constexpr (GATE != 0) o = gt[rg] > 0.0f ? o * g.gate_scale : 0.0f;
          __builtin_nontemporal_store(o, g.C + (r0 + ro) * g.ldc + tc0 + c_off);
        }
      } else {
        const int64_t col = tc0 + li;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[t][s], acc, 0, 0, 0);
        for (int rg = 0; rg < 16; ++rg) {
          const int64_t row = r0 + (rg & 3) + 8 * (rg >> 2) + 4 * lh;
          if (row < g.M && col < g.N) {
            float *dst = g.C + row * g.ldc + col;
            float o = g.alpha * acc[rg];
            if (g.beta != 0.0f) o += g.beta * *dst;
            if (g.relu && o < 0.0f) o = 0.0f;   // a NaN stays a NaN
            if constexpr (GATE == 1) o = g.gate[row * g.ldgate + col] > 0.0f ? o * g.gate_scale : 0.0f;
            if constexpr (GATE == 2)
              o = ((g.gate_bits[row * g.ldbits + (col >> 5)] >> (col & 31)) & 1u) ? o * g.gate_scale : 0.0f;
            *dst = o;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Skinny output, no transposes:  C[M, N<=16] = A[M,K] * B[K,N]  with K a
// multiple of 64 up to 256 (H' * W2 with N = the class count).  Bound by
// streaming A once, so A is read the way HBM likes it: one wavefront load
// instruction = one full row segment of up to 1 KiB, marked non-temporal
// (streaming reads measure 7.0 TB/s that way against 6.2 TB/s plain and
// 4.6 TB/s when an instruction touches 16 rows x 64 bytes).  v_mfma_f32_16x16x4_f32
// wants lane (i = lane & 15, q = lane >> 4) to hold A[row i][k], so the 16 rows
// of a pass go through a wavefront-private LDS tile (row pitch K + 4 floats):
// written as loaded, read back as 16-byte fragments A[i][16s + 4q .. +3]; MFMA
// (s, j) multiplies k = 16s + 4q + j and the B fragments (registers) use the
// same numbering.  No workgroup barrier: a wavefront only touches its own tile.
// The loads of the next two passes are in flight while this pass multiplies.
// ---------------------------------------------------------------------------
#ifndef GCNX_SKINNY_DIST
#define GCNX_SKINNY_DIST 2
#endif
template <int KS>   // K <= 16 * KS, KS = 4, 8, 16
__global__ __launch_bounds__(kBlock, 1) void gemm_skinny_n_kernel(const GemmArgs g) {
  constexpr int KMAX = 16 * KS;
  constexpr int PITCH = KMAX + 4;
  constexpr int LPR = KMAX / 4;          // lanes that cover one row with 16-byte loads
  constexpr int RPI = kWave / LPR;       // rows one load instruction covers (1, 2 or 4)
  constexpr int NLD = 16 / RPI;          // load instructions per 16-row pass
  __shared__ float tile[kWavesPerBlock][16 * PITCH];
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *const my = tile[wid];
  const int ks = (int)(g.K / 16);
  float bf[KS][4];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * s + 4 * lq + j;
      bf[s][j] = (s < ks && li < g.N) ? g.B[(int64_t)k * g.ldb + li] : 0.0f;
    }
  const int lrow = lane / LPR, lcol = (lane % LPR) * 4;
  const bool col_ok = lcol < g.K;
  const bool fast_store = g.N == 16 && g.beta == 0.0f && !g.gate && !g.gate_bits && (g.ldc % 4 == 0) &&
                          ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  // DIST passes of loads are kept in flight (one register set each)
  constexpr int DIST = GCNX_SKINNY_DIST;
  f4v ld[DIST][NLD];
  auto issue_loads = [&](int64_t r0, f4v (&dst)[NLD]) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int64_t row = r0 + u * RPI + lrow;
      dst[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      if (row < g.M && col_ok)
        dst[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(g.A + row * g.lda + lcol));
    }
  };
  auto pass = [&](int64_t r0, f4v (&cur)[NLD]) {
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      *reinterpret_cast<f4v *>(my + (u * RPI + lrow) * PITCH + lcol) = cur[u];
    const int64_t rn = r0 + DIST * nwaves * 16;
    if (rn < g.M) issue_loads(rn, cur);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // two accumulator chains so that consecutive MFMAs do not wait on each other
    f4v acc = f4v{0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc;
#pragma unroll
    for (int s = 0; s < KS; s += 2) {
      if (s < ks) {
        const f4v a = *reinterpret_cast<const f4v *>(my + li * PITCH + 16 * s + 4 * lq);
        const f4v a1 = *reinterpret_cast<const f4v *>(my + li * PITCH + 16 * (s + 1) + 4 * lq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bf[s][j], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bf[s + 1][j], acc1, 0, 0, 0);
        }
      }
    }
    acc += acc1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // D layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg
    if (fast_store) {
      // C rows are 64 bytes: turn the tile through LDS so that each lane stores
      // 16 contiguous bytes (a wavefront store covers whole 128-byte lines)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float o = g.alpha * acc[rg];
        if (g.relu && o < 0.0f) o = 0.0f;   // a NaN stays a NaN
        my[(4 * lq + rg) * 16 + li] = o;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int64_t gm = r0 + (lane >> 2);
      const f4v o = *reinterpret_cast<const f4v *>(my + lane * 4);
      if (gm < g.M) *reinterpret_cast<f4v *>(g.C + gm * g.ldc + (lane & 3) * 4) = o;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int64_t gm = r0 + 4 * lq + rg;
        if (gm < g.M && li < g.N) {
          float *dst = g.C + gm * g.ldc + li;
          *dst = gemm_epilogue(g, acc[rg], dst, gm, li);
        }
      }
    }
  };
  const int64_t step = nwaves * 16;
  int64_t r0 = wave * 16;
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (r0 + d * step < g.M) issue_loads(r0 + d * step, ld[d]);
  for (; r0 < g.M; r0 += DIST * step) {
#pragma unroll
    for (int d = 0; d < DIST; ++d)
      if (r0 + d * step < g.M) pass(r0 + d * step, ld[d]);
  }
}

// ---------------------------------------------------------------------------
// Skinny weight gradient:  C[M, N<=16] = A^T * B  with A stored [K, M], B
// stored [K, N], K = the node dimension, split along K over grid.z
// (H'^T * dP2).  One wavefront owns 64 columns of A: lane (i, q) loads
// A[k + q][m0 + 4i .. +3] (16 bytes; a wavefront reads 4 rows x 256 bytes) and
// B[k + q][i]; MFMA j of the group forms the 16 x 16 tile of output rows
// m0 + 4i + j.  Eight groups (32 rows of A and B) are in flight per pass.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gemm_skinny_tn_kernel(const GemmArgs g) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t m0 = ((int64_t)blockIdx.x * kWavesPerBlock + wid) * 64;
  if (m0 >= g.M) return;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  const bool m_ok = m0 + 4 * li < g.M;   // M % 4 == 0
  const bool n_ok = li < g.N;
  f4v acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
  for (int64_t kk = kbeg; kk < kend; kk += 4 * U) {
    f4v a[U];
    float b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = kk + 4 * u + lq;
      a[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      b[u] = 0.0f;
      if (k < kend) {
        if (m_ok) a[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(g.A + k * g.lda + m0 + 4 * li));
        if (n_ok) b[u] = g.B[k * g.ldb + li];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u], acc[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int64_t gm = m0 + 4 * (4 * lq + rg) + j;
      if (gm >= g.M || !n_ok) continue;
      if (g.splits > 1) {
        g.partial[((int64_t)blockIdx.z * g.M + gm) * g.N + li] = acc[j][rg];
      } else {
        float *dst = g.C + gm * g.ldc + li;
        *dst = gemm_epilogue(g, acc[j][rg], dst, gm, li);
      }
    }
}

static bool skinny_tn_shape(int64_t M, int64_t N, int64_t K) {
  return N >= 1 && N <= 16 && M >= 64 && M % 4 == 0 && K >= 4096;
}

static int skinny_tn_splits(int64_t M, int64_t K) {
  const int64_t blocks = ceil_div(M, 256);
  int64_t s = 1024 / blocks;
  const int64_t max_s = ceil_div(K, 512);
  if (s > max_s) s = max_s;
  if (s > 65535) s = 65535;
  if (s < 1) s = 1;
  return (int)s;
}

// Shapes served by gemm_mfma_tn_kernel (transposed A, plain B).
static bool tn_mfma_shape(int64_t M, int64_t N, int64_t K) {
  return M >= 64 && M % 128 == 0 && K >= 4096 && (N % 128 == 0 || N <= 32);
}

static int tn_mfma_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t bn = N % 256 == 0 ? 256 : N % 128 == 0 ? 128 : 32;
  const int64_t tiles = (M / 128) * ceil_div(N, bn);
  int64_t s = 2048 / tiles;
  const int64_t max_s = ceil_div(K, 1024);
  if (s > max_s) s = max_s;
  if (s > 65535) s = 65535;
  if (s < 1) s = 1;
  return (int)s;
}

static int choose_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ceil_div(M, TM) * ceil_div(N, TN);
  if (tiles >= 1024 || K <= 128) return 1;
  // one K step of the tile kernel is a full load -> LDS -> barrier round trip
  // (~2 us when nothing else hides it), so with few tiles even a K of a few
  // thousand is cut down to 64 per workgroup
  int64_t s = 2048 / tiles;
  const int64_t max_s = ceil_div(K, 64);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Workspace of a product cut into `splits` partial results: the partial planes
// and, for long reductions, the planes of the first reduction stage.
static size_t split_ws_bytes(int splits, int64_t M, int64_t N) {
  if (splits <= 1) return 0;
  const size_t planes = (size_t)splits + (splits >= kReduceStagedFrom ? (size_t)ceil_div(splits, kReduceFan) : 0);
  return planes * (size_t)M * (size_t)N * sizeof(float);
}

// Adds the partial planes in order and applies the epilogue (one or two launches).
static int launch_split_reduce(GemmArgs g, hipStream_t st) {
  const int64_t mn = g.M * g.N;
  if (g.splits >= kReduceStagedFrom) {
    const int planes = (int)ceil_div(g.splits, kReduceFan);
    float *stage = g.partial + (size_t)g.splits * (size_t)mn;
    hipLaunchKernelGGL(gemm_reduce_stage_kernel, dim3((unsigned)ceil_div(mn * planes, kBlock)), dim3(kBlock), 0, st,
                       g.partial, stage, mn, g.splits, planes);
    GCNX_CHECK_LAUNCH();
    g.partial = stage;
    g.splits = planes;
  }
  hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)ceil_div(mn, kBlock)), dim3(kBlock), 0, st, g);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

// Shapes served by gemm_split_nn_kernel (plain A; B plain or transposed).
static bool nn_split_shape(int64_t M, int64_t N, int64_t K) {
  return M >= 4 * 128 && K % 16 == 0 && K >= 64 && N % 256 == 0;
}
static size_t nn_split_ws_bytes(int64_t N, int64_t K) { return (size_t)K * (size_t)N * 3 * sizeof(unsigned short); }

}  // namespace gcnx

using namespace gcnx;

extern "C" size_t gcnx_gemm_workspace_bytes(int trans_a, int trans_b, int64_t M, int64_t N,
                                            int64_t K) {
  (void)trans_a; (void)trans_b;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // the largest split count any kernel that may serve this shape would use
  // (which one runs also depends on alignment and on the fused options)
  int s = choose_splits(M, N, K);
  if (trans_a && !trans_b && skinny_tn_shape(M, N, K)) s = std::max(s, skinny_tn_splits(M, K));
  if (trans_a && !trans_b && tn_mfma_shape(M, N, K)) s = std::max(s, tn_mfma_splits(M, N, K));
  size_t bytes = split_ws_bytes(s, M, N);
  if (!trans_a && nn_split_shape(M, N, K)) bytes = std::max(bytes, nn_split_ws_bytes(N, K));
  return bytes;
}

static int gemm_impl(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float *A,
                     int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, float alpha,
                     float beta, int fuse_relu, const float *gate, int64_t ldgate, float gate_scale,
                     const gcnx_dropout *a_drop, const int32_t *a_rows, const uint32_t *gate_bits,
                     int64_t ldbits, const gcnx_dropout *out_drop, void *ws, size_t ws_bytes, void *stream,
                     uint32_t *out_signs = nullptr, int64_t ld_out_signs = 0);

extern "C" int gcnx_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                             const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                             int64_t ldc, float alpha, float beta, int fuse_relu, void *ws,
                             size_t ws_bytes, void *stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, fuse_relu, nullptr,
                   0, 1.0f, nullptr, nullptr, nullptr, 0, nullptr, ws, ws_bytes, stream);
}

// Shape part of the MFMA dispatch rules (pointer alignment is checked at call time).
static bool nn_mfma_shape(int64_t M, int64_t N, int64_t K) {
  return M >= 4 * 128 && (K % 16 == 0) && (N >= 64 || K >= 64) && (N % 128 == 0 || N <= 32);
}

extern "C" int gcnx_gemm_dropout_a_supported(int trans_a, int trans_b, int64_t M, int64_t N,
                                             int64_t K) {
  if (trans_a) return (!trans_b && tn_mfma_shape(M, N, K) && M % 4 == 0) ? 1 : 0;
  return (nn_mfma_shape(M, N, K) && K % 4 == 0) ? 1 : 0;
}

extern "C" int gcnx_gemm_dropout_a_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                       const float *A, int64_t lda, const gcnx_dropout *a_dropout,
                                       const float *B, int64_t ldb, float *C, int64_t ldc,
                                       float alpha, float beta, void *ws, size_t ws_bytes,
                                       void *stream) {
  GCNX_REQUIRE(a_dropout, GCNX_ERR_INVALID_ARG, "gemm_dropout_a: null descriptor");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, 0, nullptr, 0, 1.0f,
                   a_dropout, nullptr, nullptr, 0, nullptr, ws, ws_bytes, stream);
}

extern "C" int gcnx_gemm_dropout_a_rows_f32(int trans_b, int64_t M, int64_t N, int64_t K,
                                            const float *A, int64_t lda,
                                            const gcnx_dropout *a_dropout, const int32_t *a_row_ids,
                                            const float *B, int64_t ldb, float *C, int64_t ldc,
                                            float alpha, float beta, void *ws, size_t ws_bytes,
                                            void *stream) {
  GCNX_REQUIRE(a_dropout && a_row_ids, GCNX_ERR_INVALID_ARG, "gemm_dropout_a_rows: null argument");
  return gemm_impl(0, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, 0, nullptr, 0, 1.0f,
                   a_dropout, a_row_ids, nullptr, 0, nullptr, ws, ws_bytes, stream);
}

extern "C" int gcnx_gemm_dropout_at_rows_supported(int64_t M, int64_t N, int64_t K) {
  return (tn_mfma_shape(M, N, K) && M % 4 == 0 && N % 256 == 0 && dense_arith_mode() == 0) ? 1 : 0;
}

extern "C" int gcnx_gemm_dropout_at_rows_f32(int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                                             const gcnx_dropout *a_dropout, const int32_t *a_row_ids,
                                             const float *B, int64_t ldb, float *C, int64_t ldc, float alpha,
                                             float beta, void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(a_dropout && a_row_ids, GCNX_ERR_INVALID_ARG, "gemm_dropout_at_rows: null argument");
  GCNX_REQUIRE(a_dropout->keep < 1.0f, GCNX_ERR_INVALID_ARG, "gemm_dropout_at_rows: keep must be below 1");
  GCNX_REQUIRE(gcnx_gemm_dropout_at_rows_supported(M, N, K), GCNX_ERR_UNSUPPORTED,
               "gemm_dropout_at_rows: shape or arithmetic not served (gcnx_gemm_dropout_at_rows_supported)");
  return gemm_impl(1, 0, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, 0, nullptr, 0, 1.0f, a_dropout,
                   a_row_ids, nullptr, 0, nullptr, ws, ws_bytes, stream);
}

extern "C" int gcnx_gemm_gated_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                   const float *A, int64_t lda, const float *B, int64_t ldb,
                                   float *C, int64_t ldc, float alpha, float beta,
                                   const float *gate, int64_t ldgate, float gate_scale, void *ws,
                                   size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(gate && ldgate >= N, GCNX_ERR_INVALID_ARG, "gemm_gated: bad gate");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, 0, gate, ldgate,
                   gate_scale, nullptr, nullptr, nullptr, 0, nullptr, ws, ws_bytes, stream);
}

extern "C" int gcnx_gemm_relu_dropout_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                          const float *A, int64_t lda, const float *B, int64_t ldb,
                                          float *C, int64_t ldc, float alpha, float beta, int fuse_relu,
                                          const gcnx_dropout *out_dropout, void *ws, size_t ws_bytes,
                                          void *stream) {
  GCNX_REQUIRE(out_dropout, GCNX_ERR_INVALID_ARG, "gemm_relu_dropout: null descriptor");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, fuse_relu, nullptr, 0,
                   1.0f, nullptr, nullptr, nullptr, 0, out_dropout, ws, ws_bytes, stream);
}

extern "C" int gcnx_gemm_relu_dropout_signs_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                                const float *A, int64_t lda, const float *B, int64_t ldb,
                                                float *C, int64_t ldc, float alpha, float beta, int fuse_relu,
                                                const gcnx_dropout *out_dropout, uint32_t *sign_bits,
                                                int64_t ld_words, void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(out_dropout && sign_bits, GCNX_ERR_INVALID_ARG, "gemm_relu_dropout_signs: null argument");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, fuse_relu, nullptr, 0,
                   1.0f, nullptr, nullptr, nullptr, 0, out_dropout, ws, ws_bytes, stream, sign_bits, ld_words);
}

extern "C" int gcnx_gemm_relu_dropout_supported(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K) {
  (void)trans_b;
  return (!trans_a && nn_mfma_shape(M, N, K) && N % 4 == 0) ? 1 : 0;
}

extern "C" int gcnx_gemm_bitgated_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                      const float *A, int64_t lda, const float *B, int64_t ldb,
                                      float *C, int64_t ldc, float alpha, float beta,
                                      const uint32_t *gate_bits, int64_t ld_words, float gate_scale,
                                      void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(gate_bits && ld_words * 32 >= N, GCNX_ERR_INVALID_ARG, "gemm_bitgated: bad gate");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, 0, nullptr, 0,
                   gate_scale, nullptr, nullptr, gate_bits, ld_words, nullptr, ws, ws_bytes, stream);
}

static int gemm_impl(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float *A,
                     int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, float alpha,
                     float beta, int fuse_relu, const float *gate, int64_t ldgate, float gate_scale,
                     const gcnx_dropout *a_drop, const int32_t *a_rows, const uint32_t *gate_bits,
                     int64_t ldbits, const gcnx_dropout *out_drop, void *ws, size_t ws_bytes, void *stream,
                     uint32_t *out_signs, int64_t ld_out_signs) {
  GCNX_REQUIRE(M >= 0 && N >= 0 && K >= 0, GCNX_ERR_INVALID_ARG, "gemm: negative size");
  if (M == 0 || N == 0) return GCNX_OK;
  GCNX_REQUIRE(C && ldc >= N, GCNX_ERR_INVALID_ARG, "gemm: bad C");
  GCNX_REQUIRE(K == 0 || (A && B), GCNX_ERR_INVALID_ARG, "gemm: null operand");
  GCNX_REQUIRE(K == 0 || (lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N)),
               GCNX_ERR_INVALID_ARG, "gemm: leading dimension too small");
  hipStream_t st = as_stream(stream);
  GemmArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta; g.relu = fuse_relu;
  g.partial = nullptr; g.splits = 1; g.k_per_split = K > 0 ? K : 1;
  g.gate = gate; g.ldgate = ldgate; g.gate_scale = gate_scale;
  g.adrop_keep = 1.0f; g.adrop_seed = 0; g.adrop_offset = 0; g.adrop_ld = 0;
  g.adrop_rows = nullptr; g.gate_bits = gate_bits; g.ldbits = ldbits;
  g.odrop_keep = 1.0f; g.odrop_seed = 0; g.odrop_offset = 0; g.odrop_ld = 0; g.odrop_col0 = 0; g.odrop_vec = 0;
  const bool want_odrop = out_drop && out_drop->keep < 1.0f;
  g.osign = nullptr; g.ld_osign = 0;
  if (out_signs) {
    GCNX_REQUIRE(want_odrop && N % 32 == 0 && ld_out_signs * 32 >= N, GCNX_ERR_UNSUPPORTED,
                 "gemm: sign words come with the fused output dropout (keep < 1), N %% 32 == 0, ld_words * 32 >= N");
    g.osign = out_signs; g.ld_osign = ld_out_signs;
  }
  if (want_odrop) {
    GCNX_REQUIRE(out_drop->keep > 0.0f && out_drop->ld >= N && out_drop->col0 >= 0, GCNX_ERR_INVALID_ARG,
                 "gemm: bad output dropout descriptor");
    g.odrop_keep = out_drop->keep; g.odrop_seed = out_drop->seed; g.odrop_offset = out_drop->offset;
    g.odrop_ld = out_drop->ld; g.odrop_col0 = out_drop->col0;
    g.odrop_vec = (beta == 0.0f && !gate && !gate_bits && N % 4 == 0 && ldc % 4 == 0 && aligned16(C) &&
                   out_drop->offset % 4 == 0 && out_drop->ld % 4 == 0 && out_drop->col0 % 4 == 0) ? 1 : 0;
  }
  // output dropout lives in the 16-byte epilogue of the tall MFMA kernel only
  GCNX_REQUIRE(!want_odrop || (g.odrop_vec && !trans_a && nn_mfma_shape(M, N, K) && (lda % 4 == 0) && aligned16(A) &&
                               (trans_b || N <= 32 || ((ldb % 4 == 0) && aligned16(B)))),
               GCNX_ERR_UNSUPPORTED, "gemm: output dropout is served for tall MFMA shapes with 16-byte aligned rows only");
  const bool want_adrop = a_drop && a_drop->keep < 1.0f;
  if (want_adrop) {
    GCNX_REQUIRE(a_drop->keep > 0.0f && a_drop->offset % 4 == 0 && a_drop->ld % 4 == 0 &&
                     a_drop->col0 == 0,
                 GCNX_ERR_INVALID_ARG, "gemm: A-dropout needs keep > 0, offset and ld multiples of 4, col0 = 0");
    g.adrop_keep = a_drop->keep; g.adrop_seed = a_drop->seed; g.adrop_offset = a_drop->offset;
    g.adrop_ld = a_drop->ld;
    // with a transposed A the row ids are served by the split-operand kernel only (checked at its dispatch)
    g.adrop_rows = a_rows;
  }

  if (K == 0) {
    // empty inner dimension: C = epilogue(0) (beta * C, gates, relu); the operands
    // may be null, so no product kernel is launched
    GCNX_REQUIRE(!want_odrop && !want_adrop, GCNX_ERR_UNSUPPORTED, "gemm: fused dropout with K = 0");
    GCNX_REQUIRE(M * N / kBlock < 0x7fffffffll, GCNX_ERR_UNSUPPORTED, "gemm: shape too large");
    g.splits = 0;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)ceil_div(M * N, kBlock)), dim3(kBlock), 0, st, g);
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }

  // tiny inner dimension, wide output: the streaming kernel
  const bool small_k = !trans_a && !want_adrop && K >= 1 && K <= 32 && N >= 64 && (N % 4 == 0) &&
                       M >= 1024 && (ldc % 4 == 0) && aligned16(C) &&
                       (!gate || ((ldgate % 4 == 0) && aligned16(gate)));
  if (small_k && !want_odrop && K <= 16 && N % 32 == 0) {
    int64_t blocks = ceil_div(ceil_div(M, 32), kWavesPerBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    const dim3 grid((unsigned)blocks, (unsigned)ceil_div(N, 256));
#define GCNX_SMALLK(TBV)                                                                                  \
  do {                                                                                                   \
    if (gate_bits) hipLaunchKernelGGL((gemm_small_k_mfma_kernel<TBV, 2>), grid, dim3(kBlock), 0, st, g);  \
    else if (gate) hipLaunchKernelGGL((gemm_small_k_mfma_kernel<TBV, 1>), grid, dim3(kBlock), 0, st, g);  \
    else hipLaunchKernelGGL((gemm_small_k_mfma_kernel<TBV, 0>), grid, dim3(kBlock), 0, st, g);            \
  } while (0)
    if (trans_b) GCNX_SMALLK(true);
    else GCNX_SMALLK(false);
#undef GCNX_SMALLK
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }
  if (small_k && !gate_bits && !want_odrop) {
    int64_t blocks = ceil_div(ceil_div(M, 2), kWavesPerBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    const dim3 grid((unsigned)blocks, (unsigned)ceil_div(N, 256));
    if (K <= 16) {
      if (trans_b) hipLaunchKernelGGL((gemm_small_k_kernel<16, true>), grid, dim3(kBlock), 0, st, g);
      else hipLaunchKernelGGL((gemm_small_k_kernel<16, false>), grid, dim3(kBlock), 0, st, g);
    } else {
      if (trans_b) hipLaunchKernelGGL((gemm_small_k_kernel<32, true>), grid, dim3(kBlock), 0, st, g);
      else hipLaunchKernelGGL((gemm_small_k_kernel<32, false>), grid, dim3(kBlock), 0, st, g);
    }
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }

  // skinny output (the class dimension): stream A once, no LDS
  const bool skinny_n = !trans_a && !trans_b && !want_adrop && !want_odrop && N <= 16 && K >= 64 && K <= 256 &&
                        K % 16 == 0 && M >= 1024 && (lda % 4 == 0) && aligned16(A);
  if (skinny_n) {
    int64_t blocks = ceil_div(ceil_div(M, 16), kWavesPerBlock);
    if (blocks > 256 * 8) blocks = 256 * 8;
    const dim3 grid((unsigned)blocks);
    if (K <= 64) hipLaunchKernelGGL((gemm_skinny_n_kernel<4>), grid, dim3(kBlock), 0, st, g);
    else if (K <= 128) hipLaunchKernelGGL((gemm_skinny_n_kernel<8>), grid, dim3(kBlock), 0, st, g);
    else hipLaunchKernelGGL((gemm_skinny_n_kernel<16>), grid, dim3(kBlock), 0, st, g);
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }
  const bool skinny_tn = trans_a && !trans_b && !want_adrop && skinny_tn_shape(M, N, K) &&
                         (lda % 4 == 0) && aligned16(A);
  if (skinny_tn) {
    const int sp = skinny_tn_splits(M, K);
    if (sp > 1) {
      const size_t need = split_ws_bytes(sp, M, N);
      GCNX_REQUIRE(ws && ws_bytes >= need, GCNX_ERR_WORKSPACE,
                   "gemm: workspace of %zu bytes required, got %zu", need, ws_bytes);
      g.partial = static_cast<float *>(ws);
      g.splits = sp;
      g.k_per_split = ceil_div(ceil_div(K, sp), 32) * 32;
    }
    hipLaunchKernelGGL(gemm_skinny_tn_kernel, dim3((unsigned)ceil_div(M, 256), 1u, (unsigned)g.splits),
                       dim3(kBlock), 0, st, g);
    GCNX_CHECK_LAUNCH();
    if (g.splits > 1) return launch_split_reduce(g, st);
    return GCNX_OK;
  }

  // MFMA for tall products once a weight dimension reaches 64 (hidden >= 64);
  // narrower layers stay on the VALU kernel.
  const bool mfma_ok = !trans_a && nn_mfma_shape(M, N, K) && (lda % 4 == 0) && aligned16(A) &&
                       (trans_b || N <= 32 || ((ldb % 4 == 0) && aligned16(B)));
  if (mfma_ok) {
    // row tiles ride on grid.x (2^31-1 limit); grid.y and .z stop at 65535
    const int64_t mt = ceil_div(M, MB);
    GCNX_REQUIRE(mt <= 0x7fffffffll && N / 128 <= 65535, GCNX_ERR_UNSUPPORTED, "gemm: shape too large");
    if (dense_arith_mode() == 0 && nn_split_shape(M, N, K) && K * N / kBlock < 0x7fffffffll && lda < (1 << 22)) {
      // exactly split operands on the bf16 matrix pipe; the weights are split first
      const size_t need = nn_split_ws_bytes(N, K);
      GCNX_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GCNX_ERR_WORKSPACE,
                   "gemm: workspace of %zu bytes (16-byte aligned) required, got %zu", need, ws_bytes);
      unsigned short *bsp = static_cast<unsigned short *>(ws);
      hipLaunchKernelGGL(split_b_kernel, dim3((unsigned)ceil_div(K * N, kBlock)), dim3(kBlock), 0, st, B, ldb, trans_b,
                         K, N, bsp);
      GCNX_CHECK_LAUNCH();
      if (g.adrop_keep < 1.0f)
        hipLaunchKernelGGL(gemm_split_nn_kernel<true>, dim3((unsigned)mt, (unsigned)(N / 256)), dim3(kBlock), 0, st, g, bsp);
      else
        hipLaunchKernelGGL(gemm_split_nn_kernel<false>, dim3((unsigned)mt, (unsigned)(N / 256)), dim3(kBlock), 0, st, g, bsp);
      GCNX_CHECK_LAUNCH();
      return GCNX_OK;
    }
    if (N % 256 == 0 && !trans_b) {
      // the whole 256-column slab per workgroup: A (and its dropout) is read once per slab
      hipLaunchKernelGGL((gemm_mfma_kernel<256, false>), dim3((unsigned)mt, (unsigned)(N / 256)), dim3(kBlock), 0, st, g);
      GCNX_CHECK_LAUNCH();
      return GCNX_OK;
    }
    const dim3 grid((unsigned)mt, (unsigned)(N % 128 == 0 ? N / 128 : 1));
    if (N % 128 == 0) {
      if (trans_b) hipLaunchKernelGGL((gemm_mfma_kernel<128, true>), grid, dim3(kBlock), 0, st, g);
      else hipLaunchKernelGGL((gemm_mfma_kernel<128, false>), grid, dim3(kBlock), 0, st, g);
    } else {
      if (trans_b) hipLaunchKernelGGL((gemm_mfma_kernel<32, true>), grid, dim3(kBlock), 0, st, g);
      else hipLaunchKernelGGL((gemm_mfma_kernel<32, false>), grid, dim3(kBlock), 0, st, g);
    }
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }

  const bool tn_ok = trans_a && !trans_b && tn_mfma_shape(M, N, K) && (lda % 4 == 0) &&
                     aligned16(A) && (N > 32 ? (ldb % 4 == 0 && aligned16(B)) : true);
  if (tn_ok) {
    const int sp = tn_mfma_splits(M, N, K);
    if (sp > 1) {
      const size_t need = split_ws_bytes(sp, M, N);
      GCNX_REQUIRE(ws && ws_bytes >= need, GCNX_ERR_WORKSPACE,
                   "gemm: workspace of %zu bytes required, got %zu", need, ws_bytes);
      g.partial = static_cast<float *>(ws);
      g.splits = sp;
      g.k_per_split = ceil_div(ceil_div(K, sp), 16) * 16;
    }
    GCNX_REQUIRE(!g.adrop_rows || (N % 256 == 0 && dense_arith_mode() == 0 && g.adrop_keep < 1.0f),
                 GCNX_ERR_UNSUPPORTED,
                 "gemm: row ids for the dropout of a transposed A need N %% 256 == 0 and the split-operand arithmetic");
    if (N % 256 == 0 && dense_arith_mode() == 0) {
      // exactly split operands on the bf16 matrix pipe
      if (g.adrop_keep < 1.0f && g.adrop_rows)
        hipLaunchKernelGGL((gemm_split_tn_kernel<true, true>), dim3((unsigned)(M / 128), (unsigned)(N / 256),
                           (unsigned)g.splits), dim3(kBlock), 0, st, g);
      else if (g.adrop_keep < 1.0f)
        hipLaunchKernelGGL(gemm_split_tn_kernel<true>, dim3((unsigned)(M / 128), (unsigned)(N / 256), (unsigned)g.splits),
                           dim3(kBlock), 0, st, g);
      else
        hipLaunchKernelGGL(gemm_split_tn_kernel<false>, dim3((unsigned)(M / 128), (unsigned)(N / 256), (unsigned)g.splits),
                           dim3(kBlock), 0, st, g);
    } else if (N % 256 == 0) {
      // 128 x 256 tiles: each workgroup reads its slice of A (and regenerates its
      // dropout) once per 256 output columns
      hipLaunchKernelGGL((gemm_mfma_tn_kernel<256>), dim3((unsigned)(M / 128), (unsigned)(N / 256), (unsigned)g.splits),
                         dim3(kBlock), 0, st, g);
    } else if (N % 128 == 0) {
      hipLaunchKernelGGL((gemm_mfma_tn_kernel<128>), dim3((unsigned)(M / 128), (unsigned)(N / 128), (unsigned)g.splits),
                         dim3(kBlock), 0, st, g);
    } else {
      hipLaunchKernelGGL((gemm_mfma_tn_kernel<32>), dim3((unsigned)(M / 128), 1u, (unsigned)g.splits),
                         dim3(kBlock), 0, st, g);
    }
    GCNX_CHECK_LAUNCH();
    if (g.splits > 1) return launch_split_reduce(g, st);
    return GCNX_OK;
  }

  GCNX_REQUIRE(!want_adrop, GCNX_ERR_UNSUPPORTED,
               "gemm: dropout on A is only fused into the MFMA kernels (tall A, a weight dimension >= 64)");
  const int splits = choose_splits(M, N, K);
  if (splits > 1) {
    const size_t need = split_ws_bytes(splits, M, N);
    GCNX_REQUIRE(ws && ws_bytes >= need, GCNX_ERR_WORKSPACE,
                 "gemm: workspace of %zu bytes required, got %zu", need, ws_bytes);
    g.partial = static_cast<float *>(ws);
    g.splits = splits;
    g.k_per_split = ceil_div(ceil_div(K, splits), TK) * TK;
  }
  GCNX_REQUIRE(ceil_div(M, TM) <= 0x7fffffffll && ceil_div(N, TN) <= 65535, GCNX_ERR_UNSUPPORTED,
               "gemm: shape too large");
  const dim3 grid((unsigned)ceil_div(M, TM), (unsigned)ceil_div(N, TN), (unsigned)g.splits);
  if (!trans_a && !trans_b) hipLaunchKernelGGL((gemm_tile_kernel<false, false>), grid, dim3(kBlock), 0, st, g);
  else if (!trans_a && trans_b) hipLaunchKernelGGL((gemm_tile_kernel<false, true>), grid, dim3(kBlock), 0, st, g);
  else if (trans_a && !trans_b) hipLaunchKernelGGL((gemm_tile_kernel<true, false>), grid, dim3(kBlock), 0, st, g);
  else hipLaunchKernelGGL((gemm_tile_kernel<true, true>), grid, dim3(kBlock), 0, st, g);
  GCNX_CHECK_LAUNCH();
  if (g.splits > 1) return launch_split_reduce(g, st);
  return GCNX_OK;
}
