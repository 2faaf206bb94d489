// Calibration probe for the memory-side counters on narrow gathers (gfx950).
// A table far larger than L2 + Infinity Cache is read ONCE, row by row, in a
// random order (a bijection of the row ids), with the access shape of the narrow
// SpMM kernel: one wavefront load instruction covers 64 / LW rows, LW lanes x 16
// bytes each.  The bytes that must cross the fabric are then known exactly
// (rows x row bytes), so FETCH_SIZE / TCC_EA0_RDREQ readings of this program say
// how many bytes the L2 requests per gathered row of 64, 128 or 256 bytes.
//
//   hipcc -O3 --offload-arch=gfx950 gather_probe.hip -o gather_probe
//   ./gather_probe <LW: 4|8|16> [log2 rows = 25]
//   rocprofv3 --pmc FETCH_SIZE -- ./gather_probe 4
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float __attribute__((ext_vector_type(4))) f4v;

// odd multiplier + xor-shift rounds: a bijection on [0, 2^bits)
__device__ __forceinline__ unsigned scramble(unsigned i, int bits) {
  const unsigned mask = bits == 32 ? 0xffffffffu : ((1u << bits) - 1u);
  unsigned x = i;
  x = (x * 0x9E3779B1u) & mask;
  x ^= x >> (bits / 2);
  x = (x * 0x85EBCA6Bu) & mask;
  x ^= x >> (bits / 2 + 1);
  x = (x * 0xC2B2AE35u) & mask;
  return x;
}

// The same gather through buffer loads with an explicit cache policy (AUX: bit 0 =
// sc0, bit 1 = nt, bit 4 = sc1), to see whether any policy makes the L2 ask the
// fabric for less than a whole 128-byte line per 64-byte row.
typedef int __attribute__((ext_vector_type(4))) i4v;
template <int LW, int U, int AUX>
__global__ __launch_bounds__(256) void gather_policy_kernel(const f4v *__restrict__ table, int bits, float *out) {
  constexpr int S = 64 / LW;
  const int lane = threadIdx.x & 63;
  const int slot = lane / LW, part = lane % LW;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t n_rows = (size_t)1 << bits;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<f4v *>(table), 0, (int)0xffffffffu, 0x00020000);
  f4v acc = {0, 0, 0, 0};
  for (size_t base = wave * (size_t)(S * U); base < n_rows; base += nwaves * (size_t)(S * U)) {
    i4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned r = scramble((unsigned)(base + (size_t)u * S + slot), bits);
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((r * LW + part) * 16u), 0, AUX);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += __builtin_bit_cast(f4v, v[u]);
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}

template <int LW, int U>
__global__ __launch_bounds__(256) void gather_kernel(const f4v *__restrict__ table, int bits, float *out) {
  constexpr int S = 64 / LW;            // rows per load instruction
  const int lane = threadIdx.x & 63;
  const int slot = lane / LW, part = lane % LW;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const size_t n_rows = (size_t)1 << bits;
  f4v acc = {0, 0, 0, 0};
  for (size_t base = wave * (size_t)(S * U); base < n_rows; base += nwaves * (size_t)(S * U)) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned r = scramble((unsigned)(base + (size_t)u * S + slot), bits);
      v[u] = table[(size_t)r * LW + part];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}

template <int AUX>
static void run_policy(const f4v *table, int bits, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t bytes = ((size_t)1 << bits) * 4 * 16;
  for (int it = 0; it < 3; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((gather_policy_kernel<4, 4, AUX>), dim3(256 * 32), dim3(256), 0, 0, table, bits, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("policy aux=%d rows=2^%d row=64 B: %.3f ms = %.0f GB/s of row bytes, %.1f G rows/s\n", AUX, bits, ms,
                bytes / (ms * 1e-3) / 1e9, ((size_t)1 << bits) / (ms * 1e-3) / 1e9);
  }
}

template <int LW>
static void run(const f4v *table, int bits, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t bytes = ((size_t)1 << bits) * LW * 16;
  for (int it = 0; it < 3; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((gather_kernel<LW, 4>), dim3(256 * 32), dim3(256), 0, 0, table, bits, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("LW=%d rows=2^%d row=%d B table=%.2f GB: %.3f ms = %.0f GB/s of row bytes, %.1f G rows/s\n", LW, bits,
                LW * 16, bytes / 1e9, ms, bytes / (ms * 1e-3) / 1e9, ((size_t)1 << bits) / (ms * 1e-3) / 1e9);
  }
}

int main(int argc, char **argv) {
  const int lw = argc > 1 ? std::atoi(argv[1]) : 4;
  const int bits = argc > 2 ? std::atoi(argv[2]) : 25;
  const int aux = argc > 3 ? std::atoi(argv[3]) : -1;      // >= 0: 64-byte rows through buffer loads with this policy
  const int uncached = argc > 4 ? std::atoi(argv[4]) : 0;  // 1: table in uncached device memory
  if ((lw != 4 && lw != 8 && lw != 16) || bits < 10 || bits > 27 || (aux >= 0 && (lw != 4 || bits > 25))) {
    std::fprintf(stderr, "usage: gather_probe <4|8|16> [log2 rows, 10..27]\n");
    return 2;
  }
  const size_t bytes = ((size_t)1 << bits) * lw * 16;
  f4v *table = nullptr;
  float *out = nullptr;
  if ((uncached ? hipExtMallocWithFlags((void **)&table, bytes, hipDeviceMallocUncached) : hipMalloc(&table, bytes)) !=
          hipSuccess ||
      hipMalloc(&out, 4) != hipSuccess)
    return 2;
  (void)hipMemset(table, 0, bytes);
  (void)hipDeviceSynchronize();
  if (aux == 0) run_policy<0>(table, bits, out);
  else if (aux == 1) run_policy<1>(table, bits, out);
  else if (aux == 2) run_policy<2>(table, bits, out);
  else if (aux == 16) run_policy<16>(table, bits, out);
  else if (aux == 17) run_policy<17>(table, bits, out);
  else if (aux == 18) run_policy<18>(table, bits, out);
  else if (aux == 19) run_policy<19>(table, bits, out);
  else if (aux >= 0) return 2;
  else if (lw == 4) run<4>(table, bits, out);
  else if (lw == 8) run<8>(table, bits, out);
  else run<16>(table, bits, out);
  (void)hipDeviceSynchronize();
  (void)hipFree(table);
  (void)hipFree(out);
  return 0;
}
