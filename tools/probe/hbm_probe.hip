// Streaming-read / copy bandwidth probe for gfx950: which load shape reaches
// the highest HBM rate.  Build: hipcc -O3 --offload-arch=gfx950 hbm_probe.hip -o hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float __attribute__((ext_vector_type(4))) f4v;

// Each wavefront reads U consecutive 1 KiB lines per pass (U loads in flight per lane).
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const f4v *__restrict__ x, size_t n_vec, float *out) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  f4v acc = {0, 0, 0, 0};
  for (size_t base = wave * (size_t)(64 * U); base < n_vec; base += nwaves * (size_t)(64 * U)) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 64 + lane;
      if (NT) v[u] = i < n_vec ? __builtin_nontemporal_load(x + i) : f4v{0, 0, 0, 0};
      else v[u] = i < n_vec ? x[i] : f4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const f4v *__restrict__ x, f4v *__restrict__ y, size_t n_vec) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (size_t base = wave * (size_t)(64 * U); base < n_vec; base += nwaves * (size_t)(64 * U)) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 64 + lane;
      v[u] = i < n_vec ? x[i] : f4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 64 + lane;
      if (i < n_vec) { if (NT) __builtin_nontemporal_store(v[u], y + i); else y[i] = v[u]; }
    }
  }
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < 5; ++i) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main() {
  const size_t bytes = (size_t)16 << 30;
  const size_t n_vec = bytes / 16;
  f4v *x, *y; float *out;
  if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4);
  hipMemset(x, 0, bytes); hipMemset(y, 0, bytes);
  // occupancy sensitivity: dynamic LDS limits resident workgroups per CU
  for (int lds : {0, 33000, 66000}) {
    const int g = 2048;
#define RL(U) { float ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<U, true>), dim3(g), dim3(256), lds, 0, x, n_vec, out); }); \
    printf("read  U=%2d nt=1 grid=%6d lds=%6d %7.2f ms  %6.0f GB/s\n", U, g, lds, ms, bytes / ms / 1e6); fflush(stdout); }
    RL(4) RL(8) RL(16)
  }
  const int grids[] = {2048};
  for (int g : grids) {
#define RD(U, NT) { float ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(g), dim3(256), 0, 0, x, n_vec, out); }); \
    printf("read  U=%2d nt=%d grid=%6d  %7.2f ms  %6.0f GB/s\n", U, (int)NT, g, ms, bytes / ms / 1e6); fflush(stdout); }
    RD(4, false) RD(8, false) RD(16, false) RD(8, true)
#define CP(U, NT) { float ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(g), dim3(256), 0, 0, x, y, n_vec); }); \
    printf("copy  U=%2d nt=%d grid=%6d  %7.2f ms  %6.0f GB/s\n", U, (int)NT, g, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); }
    CP(4, false) CP(8, false) CP(8, true)
  }
  return 0;
}
