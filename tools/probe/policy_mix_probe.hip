// Which load policy keeps a hot set resident in the L2 while a warm set streams by?
// Gathers of 128-byte rows (8 lanes x 16 bytes, 8 rows per wavefront instruction)
// from two tables: a HOT one that fits an XCD's L2 (default 2 MiB) and a WARM one
// that fits the Infinity Cache but not the L2 (default 128 MiB); a third of the
// load instructions go to the hot table, the rest to the warm one, rows uniformly
// random.  The hot loads are plain; the warm loads carry the cache policy under
// test (buffer-load AUX bits: 1 = sc0, 2 = nt, 16 = sc1, and their sums).  If a
// policy makes the warm rows bypass the L2 but still be served on-die, the hot
// rows keep hitting (TCC_HIT up, TCC_EA0_RDREQ down) and the time falls.
//
//   hipcc -O3 --offload-arch=gfx950 policy_mix_probe.hip -o policy_mix_probe
//   ./policy_mix_probe <aux> [hot KiB = 2048] [warm MiB = 128] [million instructions = 64]
//   rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -- ./policy_mix_probe 16
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float __attribute__((ext_vector_type(4))) f4v;
typedef int __attribute__((ext_vector_type(4))) i4v;

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int AUX, int U>
__global__ __launch_bounds__(256) void mix_kernel(const f4v *hot, unsigned hot_rows, const f4v *warm,
                                                  unsigned warm_rows, size_t n_instr, float *out) {
  const int lane = threadIdx.x & 63;
  const int slot = lane >> 3, part = lane & 7;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4v *>(hot), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4v *>(warm), 0, (int)0xffffffffu, 0x00020000);
  f4v acc = {0, 0, 0, 0};
  for (size_t i = wave * U; i < n_instr; i += nwaves * U) {
    i4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned h = mix((unsigned)(i + u) * 2654435761u + 12345u);
      const unsigned r = mix(h + (unsigned)slot * 0x9E3779B9u);
      if (h % 3u == 0u) {      // wavefront-uniform choice: one load instruction, one policy
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rh, (int)(((r % hot_rows) * 8u + part) * 16u), 0, 0);
      } else {
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(((r % warm_rows) * 8u + part) * 16u), 0, AUX);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += __builtin_bit_cast(f4v, v[u]);
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}

template <int AUX>
static void run(const f4v *hot, unsigned hot_rows, const f4v *warm, unsigned warm_rows, size_t n_instr, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<AUX, 4>), dim3(256 * 8), dim3(256), 0, 0, hot, hot_rows, warm, warm_rows, n_instr, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("warm policy aux=%d: %.3f ms, %.1f G rows/s, %.0f GB/s of row bytes\n", AUX, ms,
                n_instr * 8 / (ms * 1e-3) / 1e9, n_instr * 1024.0 / (ms * 1e-3) / 1e9);
  }
}

int main(int argc, char **argv) {
  const int aux = argc > 1 ? std::atoi(argv[1]) : 0;
  const size_t hot_kib = argc > 2 ? std::atoi(argv[2]) : 2048;
  const size_t warm_mib = argc > 3 ? std::atoi(argv[3]) : 128;
  const size_t n_instr = (size_t)(argc > 4 ? std::atoi(argv[4]) : 64) * 1000000;
  const size_t hot_bytes = hot_kib << 10, warm_bytes = warm_mib << 20;
  f4v *hot = nullptr, *warm = nullptr;
  float *out = nullptr;
  if (hipMalloc(&hot, hot_bytes) != hipSuccess || hipMalloc(&warm, warm_bytes) != hipSuccess ||
      hipMalloc(&out, 4) != hipSuccess)
    return 2;
  (void)hipMemset(hot, 0, hot_bytes);
  (void)hipMemset(warm, 0, warm_bytes);
  (void)hipDeviceSynchronize();
  const unsigned hr = (unsigned)(hot_bytes / 128), wr = (unsigned)(warm_bytes / 128);
  std::printf("hot %zu KiB (%u rows), warm %zu MiB (%u rows), %zu M load instructions of 8 rows, 1 in 3 hot\n", hot_kib, hr,
              warm_mib, wr, n_instr / 1000000);
  switch (aux) {
    case 0: run<0>(hot, hr, warm, wr, n_instr, out); break;
    case 1: run<1>(hot, hr, warm, wr, n_instr, out); break;
    case 2: run<2>(hot, hr, warm, wr, n_instr, out); break;
    case 3: run<3>(hot, hr, warm, wr, n_instr, out); break;
    case 16: run<16>(hot, hr, warm, wr, n_instr, out); break;
    case 17: run<17>(hot, hr, warm, wr, n_instr, out); break;
    case 18: run<18>(hot, hr, warm, wr, n_instr, out); break;
    case 19: run<19>(hot, hr, warm, wr, n_instr, out); break;
    default: return 2;
  }
  (void)hipDeviceSynchronize();
  return 0;
}
