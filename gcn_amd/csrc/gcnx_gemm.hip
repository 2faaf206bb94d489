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
  if (g.relu) out = fmaxf(out, 0.0f);
  if (g.gate) out = g.gate[gm * g.ldgate + gn] > 0.0f ? out * g.gate_scale : 0.0f;
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

// ---------------------------------------------------------------------------
// MFMA kernel:  C[M,N] = A[M,K] * B[K,N]  (no transposes), N % 32 == 0,
// K % 16 == 0.  Workgroup = 4 wavefronts, block tile 128 x BN (BN = 128);
// wavefront w owns rows [32w, 32w+32) of the block tile and all BN columns:
// BN/32 accumulator tiles of 32x32 (16 registers each).  Per K-step of 16 the
// block stages A[128x16] and B[16xBN] in LDS; one v_mfma_f32_32x32x2_f32 takes
// A[i=lane&31][k=lane>>5] and B[k=lane>>5][j=lane&31] as one float each.
// ---------------------------------------------------------------------------
typedef float __attribute__((ext_vector_type(16))) f16v;

constexpr int MB = 128, KB = 16;

// BN = 128: wavefront w owns rows [32w, 32w+32) and all 128 columns (four
// accumulator tiles).  BN = 32: the class-dimension / skinny case (N <= 32,
// columns beyond N read as zero, one tile per wavefront); such products are
// bound by streaming A (or writing C), the matrix pipe has room to spare.
// TB: B is stored [N, K] (the product with a transposed weight matrix); it is
// transposed while it is staged, so the inner loop is the same.
template <int BN, bool TB>
__global__ __launch_bounds__(kBlock, 3) void gemm_mfma_kernel(const GemmArgs g) {
  // A tile stored [m][k] with a 17-word pitch: lanes 0..31 of a fragment read
  // stride-17 words (conflict free), lanes 32..63 the next k.
  constexpr int NT = BN / 32;
  __shared__ float As[MB * (KB + 1)];
  __shared__ float Bs[KB * BN];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
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
  f4v ra[2], rb[2];
  float rbs[NB_S];
  auto load_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int m = e >> 2, q = e & 3;
      const int64_t gm = m0 + m;
      ra[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      if (gm < g.M) ra[i] = *reinterpret_cast<const f4v *>(g.A + gm * g.lda + kk + q * 4);
    }
    if constexpr (!TB && BN == 128) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + i * kBlock;
        const int k = e >> 5, q = e & 31;
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
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int m = e >> 2, q = e & 3;
      f4v v = ra[i];
      if (g.adrop_keep < 1.0f && m0 + m < g.M) v = drop4(g, v, m0 + m, kk + q * 4);
      float *d = As + m * (KB + 1) + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if constexpr (!TB && BN == 128) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + i * kBlock;
        const int k = e >> 5, q = e & 31;
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
      if constexpr (BN == 128) {
        const float a = As[(w * 32 + li) * (KB + 1) + k2 + lk];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float b = Bs[(k2 + lk) * BN + t * 32 + li];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      } else {
        const float a = As[(w * 32 + li) * (KB + 1) + k2 + lk];
        const float b = Bs[(k2 + lk) * BN + li];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int rgi = 0; rgi < 16; ++rgi) {
      const int row = (rgi & 3) + 8 * (rgi >> 2) + 4 * lk;
      const int64_t gm = m0 + w * 32 + row;
      const int64_t gn = n0 + t * 32 + li;
      if (gm >= g.M || gn >= g.N) continue;
      float *dst = g.C + gm * g.ldc + gn;
      *dst = gemm_epilogue(g, acc[t][rgi], dst, gm, gn);
    }
  }
}

// ---------------------------------------------------------------------------
// MFMA kernel for the weight gradients  C[M,N] = A^T * B  with A stored [K, M],
// B stored [K, N] and K = the node dimension (huge), split along K over
// grid.z.  The A operand of v_mfma_f32_32x32x2_f32 wants A_op[i][k] = A[k][i]:
// for a fixed k the 32 values are contiguous in memory, so both operands are
// staged in LDS exactly as they lie in HBM ([k][m] and [k][n]) and fragment
// reads are conflict free.  Block tile 128 x BN: BN = 128 -> wavefronts 2 x 2,
// each 64 x 64 (four accumulator tiles); BN = 32 -> wavefronts 4 x 1, each
// 32 x 32 (the class-dimension case, N <= 32, columns beyond N read as zero).
// Every split writes its tile to the workspace; gemm_reduce_kernel adds the
// splits in order.
// ---------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(kBlock, 3) void gemm_mfma_tn_kernel(const GemmArgs g) {
  constexpr int BM = 128, BK = 16;
  constexpr int WN = BN == 128 ? 2 : 1;          // wavefronts along N
  constexpr int WM = 4 / WN;                     // wavefronts along M
  constexpr int TMW = BM / (32 * WM);            // accumulator tiles per wavefront along M
  constexpr int TNW = BN / (32 * WN);
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
  f4v ra[2], rb[2];
  float rbs[2];
  auto load_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      ra[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
      if (kk + k < kend) ra[i] = *reinterpret_cast<const f4v *>(g.A + (kk + k) * g.lda + m0 + q * 4);
      if constexpr (BN == 128) {
        rb[i] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (kk + k < kend) rb[i] = *reinterpret_cast<const f4v *>(g.B + (kk + k) * g.ldb + n0 + q * 4);
      } else {
        rbs[i] = (kk + k < kend && n0 + q < g.N) ? g.B[(kk + k) * g.ldb + n0 + q] : 0.0f;
      }
    }
  };
  auto store_tiles = [&](int64_t kk) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * kBlock;
      const int k = e >> 5, q = e & 31;
      f4v v = ra[i];
      if (g.adrop_keep < 1.0f && kk + k < kend) v = drop4(g, v, kk + k, m0 + q * 4);
      *reinterpret_cast<f4v *>(As + k * BM + q * 4) = v;
      if constexpr (BN == 128) {
        *reinterpret_cast<f4v *>(Bs + k * BN + q * 4) = rb[i];
      } else {
        Bs[k * BN + q] = rbs[i];
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
          if (g.relu) o = fmaxf(o, 0.0f);
          if (g.gate) o = gt[i][j] > 0.0f ? o * g.gate_scale : 0.0f;
          out[j] = o;
        }
        *reinterpret_cast<f4v *>(g.C + r * g.ldc + col0) = out;
      }
    }
  }
}

// Shapes served by gemm_mfma_tn_kernel (transposed A, plain B).
static bool tn_mfma_shape(int64_t M, int64_t N, int64_t K) {
  return M >= 64 && M % 128 == 0 && K >= 4096 && (N % 128 == 0 || N <= 32);
}

static int tn_mfma_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t bn = N % 128 == 0 ? 128 : 32;
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

}  // namespace gcnx

using namespace gcnx;

extern "C" size_t gcnx_gemm_workspace_bytes(int trans_a, int trans_b, int64_t M, int64_t N,
                                            int64_t K) {
  (void)trans_a; (void)trans_b;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s = (trans_a && !trans_b && tn_mfma_shape(M, N, K)) ? tn_mfma_splits(M, N, K)
                                                                : choose_splits(M, N, K);
  return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}

static int gemm_impl(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float *A,
                     int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, float alpha,
                     float beta, int fuse_relu, const float *gate, int64_t ldgate, float gate_scale,
                     const gcnx_dropout *a_drop, void *ws, size_t ws_bytes, void *stream);

extern "C" int gcnx_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                             const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                             int64_t ldc, float alpha, float beta, int fuse_relu, void *ws,
                             size_t ws_bytes, void *stream) {
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, fuse_relu, nullptr,
                   0, 1.0f, nullptr, ws, ws_bytes, stream);
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
                   a_dropout, ws, ws_bytes, stream);
}

extern "C" int gcnx_gemm_gated_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                   const float *A, int64_t lda, const float *B, int64_t ldb,
                                   float *C, int64_t ldc, float alpha, float beta,
                                   const float *gate, int64_t ldgate, float gate_scale, void *ws,
                                   size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(gate && ldgate >= N, GCNX_ERR_INVALID_ARG, "gemm_gated: bad gate");
  return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, 0, gate, ldgate,
                   gate_scale, nullptr, ws, ws_bytes, stream);
}

static int gemm_impl(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float *A,
                     int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, float alpha,
                     float beta, int fuse_relu, const float *gate, int64_t ldgate, float gate_scale,
                     const gcnx_dropout *a_drop, void *ws, size_t ws_bytes, void *stream) {
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
  const bool want_adrop = a_drop && a_drop->keep < 1.0f;
  if (want_adrop) {
    GCNX_REQUIRE(a_drop->keep > 0.0f && a_drop->offset % 4 == 0 && a_drop->ld % 4 == 0 &&
                     a_drop->col0 == 0,
                 GCNX_ERR_INVALID_ARG, "gemm: A-dropout needs keep > 0, offset and ld multiples of 4, col0 = 0");
    g.adrop_keep = a_drop->keep; g.adrop_seed = a_drop->seed; g.adrop_offset = a_drop->offset;
    g.adrop_ld = a_drop->ld;
  }

  // tiny inner dimension, wide output: the streaming kernel
  const bool small_k = !trans_a && !want_adrop && K >= 1 && K <= 32 && N >= 64 && (N % 4 == 0) &&
                       M >= 1024 && (ldc % 4 == 0) && aligned16(C) &&
                       (!gate || ((ldgate % 4 == 0) && aligned16(gate)));
  if (small_k) {
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

  // MFMA for tall products once a weight dimension reaches 64 (hidden >= 64);
  // narrower layers stay on the VALU kernel.
  const bool mfma_ok = !trans_a && nn_mfma_shape(M, N, K) && (lda % 4 == 0) && aligned16(A) &&
                       (trans_b || N <= 32 || ((ldb % 4 == 0) && aligned16(B)));
  if (mfma_ok) {
    // row tiles ride on grid.x (2^31-1 limit); grid.y and .z stop at 65535
    const int64_t mt = ceil_div(M, MB);
    GCNX_REQUIRE(mt <= 0x7fffffffll && N / 128 <= 65535, GCNX_ERR_UNSUPPORTED, "gemm: shape too large");
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
      const size_t need = (size_t)sp * (size_t)M * (size_t)N * sizeof(float);
      GCNX_REQUIRE(ws && ws_bytes >= need, GCNX_ERR_WORKSPACE,
                   "gemm: workspace of %zu bytes required, got %zu", need, ws_bytes);
      g.partial = static_cast<float *>(ws);
      g.splits = sp;
      g.k_per_split = ceil_div(ceil_div(K, sp), 16) * 16;
    }
    if (N % 128 == 0) {
      hipLaunchKernelGGL((gemm_mfma_tn_kernel<128>), dim3((unsigned)(M / 128), (unsigned)(N / 128), (unsigned)g.splits),
                         dim3(kBlock), 0, st, g);
    } else {
      hipLaunchKernelGGL((gemm_mfma_tn_kernel<32>), dim3((unsigned)(M / 128), 1u, (unsigned)g.splits),
                         dim3(kBlock), 0, st, g);
    }
    GCNX_CHECK_LAUNCH();
    if (g.splits > 1) {
      hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)ceil_div(M * N, kBlock)), dim3(kBlock), 0, st, g);
      GCNX_CHECK_LAUNCH();
    }
    return GCNX_OK;
  }

  GCNX_REQUIRE(!want_adrop, GCNX_ERR_UNSUPPORTED,
               "gemm: dropout on A is only fused into the MFMA kernels (tall A, a weight dimension >= 64)");
  const int splits = choose_splits(M, N, K);
  if (splits > 1) {
    const size_t need = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
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
  if (g.splits > 1) {
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)ceil_div(M * N, kBlock)), dim3(kBlock), 0, st, g);
    GCNX_CHECK_LAUNCH();
  }
  return GCNX_OK;
}
