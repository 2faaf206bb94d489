// Write bandwidth of a [rows, 256] fp32 matrix for different shapes of one
// wavefront store instruction (which rows / how many contiguous bytes per row).
// Build: hipcc -O3 --offload-arch=gfx950 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float __attribute__((ext_vector_type(4))) f4v;

// MODE 0: dword stores, lanes 0..31 one row (128 B), lanes 32..63 the row 4 below (MFMA C layout)
// MODE 1: 16-byte stores, 8 lanes per row (128 B), 8 rows per instruction
// MODE 2: 16-byte stores, 16 lanes per row (256 B), 4 rows per instruction
// MODE 3: 16-byte stores, 64 lanes per row (1 KiB), 1 row per instruction
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float *__restrict__ c, long rows) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r0 = wave * 32; r0 < rows; r0 += nwaves * 32) {
    if (MODE == 0) {
      for (int t = 0; t < 8; ++t)
        for (int rg = 0; rg < 16; ++rg) {
          const long row = r0 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
          c[row * 256 + 32 * t + (lane & 31)] = (float)rg;
        }
    } else if (MODE == 1) {
      for (int t = 0; t < 8; ++t)
        for (int p = 0; p < 4; ++p) {
          const long row = r0 + p * 8 + (lane >> 3);
          *reinterpret_cast<f4v *>(c + row * 256 + 32 * t + (lane & 7) * 4) = f4v{1, 2, 3, (float)p};
        }
    } else if (MODE == 2) {
      for (int t = 0; t < 4; ++t)
        for (int p = 0; p < 8; ++p) {
          const long row = r0 + p * 4 + (lane >> 4);
          *reinterpret_cast<f4v *>(c + row * 256 + 64 * t + (lane & 15) * 4) = f4v{1, 2, 3, (float)p};
        }
    } else {
      for (int p = 0; p < 32; ++p)
        *reinterpret_cast<f4v *>(c + (r0 + p) * 256 + lane * 4) = f4v{1, 2, 3, (float)p};
    }
  }
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < 5; ++i) {
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main() {
  const long rows = 1L << 24;
  const size_t bytes = (size_t)rows * 256 * 4;
  float *c;
  if (hipMalloc(&c, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const int g = 2048;
#define RUN(M, what) { float ms = time_ms([&] { hipLaunchKernelGGL((store_kernel<M>), dim3(g), dim3(256), 0, 0, c, rows); }); \
  printf("%-58s %7.2f ms %6.0f GB/s\n", what, ms, bytes / ms / 1e6); fflush(stdout); }
  RUN(0, "dword stores, 2 rows x 128 B per instruction")
  RUN(1, "16-byte stores, 8 rows x 128 B per instruction")
  RUN(2, "16-byte stores, 4 rows x 256 B per instruction")
  RUN(3, "16-byte stores, 1 row x 1 KiB per instruction")
  return 0;
}
