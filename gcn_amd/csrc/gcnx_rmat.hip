// Synthetic RMAT edge generator (benchmark inputs).  Every edge is a pure
// function of (seed, edge index): Philox4x32-10 supplies one 32-bit uniform per
// recursion level, so any device can generate any slice of the edge list and a
// graph is identical for every GPU count.  Vertex ids are scrambled by a seeded
// bijection of [0, 2^scale) so that hub rows are spread over the row range
// instead of clustering at id 0.
#include "gcnx_common.h"
#include "gcnx_philox.h"

namespace gcnx {

struct Scramble {
  uint64_t m0, a0, m1, a1;
  int s0, s1;
  uint64_t mask;
};

__host__ __device__ inline uint64_t splitmix64(uint64_t &x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// multiply-by-odd, add, xor-shift-right: each step is a bijection mod 2^scale
__host__ __device__ inline uint64_t scramble_id(uint64_t v, const Scramble &s) {
  v = (v * s.m0 + s.a0) & s.mask;
  v ^= v >> s.s0;
  v = (v * s.m1 + s.a1) & s.mask;
  v ^= v >> s.s1;
  return v;
}

__global__ __launch_bounds__(kBlock) void rmat_edges_kernel(
    int scale, int64_t e_begin, int64_t e_end, uint32_t ta, uint32_t tab, uint32_t tabc,
    uint64_t seed, Scramble sc, int64_t row_begin, int64_t row_end, uint64_t *__restrict__ keys,
    int64_t capacity, unsigned long long *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // every lane of a wavefront runs the same number of iterations (ballots below)
  const int64_t n = e_end - e_begin;
  const int64_t iters = (n + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t e = e_begin + it * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool w0 = false, w1 = false;
    uint64_t u = 0, v = 0;
    if (e < e_end) {
      for (int lvl = 0; lvl < scale; lvl += 4) {
        const philox4 r = philox4x32_10((uint64_t)e, (uint64_t)(lvl >> 2), seed);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (lvl + j < scale) {
            const uint32_t x = r.v[j];
            const int q = x < ta ? 0 : (x < tab ? 1 : (x < tabc ? 2 : 3));
            u = (u << 1) | (uint64_t)(q >> 1);
            v = (v << 1) | (uint64_t)(q & 1);
          }
        }
      }
      u = scramble_id(u, sc);
      v = scramble_id(v, sc);
      if (u != v) {
        w0 = (int64_t)u >= row_begin && (int64_t)u < row_end;
        w1 = (int64_t)v >= row_begin && (int64_t)v < row_end;
      }
    }
    const unsigned long long b0 = __ballot(w0), b1 = __ballot(w1);
    const int c0 = __popcll(b0), c1 = __popcll(b1);
    unsigned long long base = 0;
    if (lane == 0 && (c0 + c1) > 0) base = atomicAdd(counter, (unsigned long long)(c0 + c1));
    base = __shfl(base, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (w0) {
      const int64_t p = (int64_t)base + __popcll(b0 & below);
      if (p < capacity) keys[p] = (u << 32) | v;
    }
    if (w1) {
      const int64_t p = (int64_t)base + c0 + __popcll(b1 & below);
      if (p < capacity) keys[p] = (v << 32) | u;
    }
  }
}

static uint32_t prob_threshold(double p) {
  if (p <= 0.0) return 0u;
  if (p >= 1.0) return 0xffffffffu;
  return (uint32_t)(p * 4294967296.0);
}

}  // namespace gcnx

using namespace gcnx;

extern "C" size_t gcnx_rmat_workspace_bytes(void) { return 256; }

extern "C" int gcnx_rmat_edges(int32_t scale, int64_t edge_begin, int64_t edge_end, double a,
                               double b, double c, uint64_t seed, int64_t row_begin,
                               int64_t row_end, uint64_t *keys_out, int64_t capacity,
                               int64_t *h_count, void *ws, void *stream) {
  GCNX_REQUIRE(scale >= 1 && scale <= 31, GCNX_ERR_INVALID_ARG, "rmat: scale must be in [1, 31]");
  GCNX_REQUIRE(edge_begin >= 0 && edge_end >= edge_begin && row_begin >= 0 && row_end >= row_begin,
               GCNX_ERR_INVALID_ARG, "rmat: bad range");
  GCNX_REQUIRE(a >= 0 && b >= 0 && c >= 0 && a + b + c <= 1.0, GCNX_ERR_INVALID_ARG,
               "rmat: bad quadrant probabilities");
  GCNX_REQUIRE(h_count && ws && (capacity == 0 || keys_out), GCNX_ERR_INVALID_ARG,
               "rmat: null argument");
  hipStream_t st = as_stream(stream);
  Scramble sc;
  uint64_t x = seed ^ 0x6a09e667f3bcc908ull;
  sc.mask = (1ull << scale) - 1ull;
  sc.m0 = splitmix64(x) | 1ull;
  sc.a0 = splitmix64(x);
  sc.m1 = splitmix64(x) | 1ull;
  sc.a1 = splitmix64(x);
  sc.s0 = scale / 2 + 1;
  sc.s1 = scale / 3 + 1;
  unsigned long long *counter = static_cast<unsigned long long *>(ws);
  GCNX_CHECK_HIP(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
  const int64_t n = edge_end - edge_begin;
  if (n > 0) {
    int64_t blocks = ceil_div(n, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(rmat_edges_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, (int)scale,
                       edge_begin, edge_end, prob_threshold(a), prob_threshold(a + b),
                       prob_threshold(a + b + c), seed, sc, row_begin, row_end, keys_out, capacity,
                       counter);
    GCNX_CHECK_LAUNCH();
  }
  unsigned long long h = 0;
  GCNX_CHECK_HIP(hipMemcpyAsync(&h, counter, sizeof(h), hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  *h_count = (int64_t)h;
  GCNX_REQUIRE((int64_t)h <= capacity, GCNX_ERR_WORKSPACE,
               "rmat: %lld keys produced but capacity is %lld", (long long)h, (long long)capacity);
  return GCNX_OK;
}
