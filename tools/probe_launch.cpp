// Launch-ramp probe (scratch tool, not part of the product): how long after the first workgroup's entry does the last
// workgroup of a 256-workgroup grid enter, as a function of dynamic LDS size, block size and register footprint?
//   hipcc --offload-arch=gfx950 -O3 -o tools/dbg/probe_launch tools/probe_launch.cpp && tools/dbg/probe_launch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int NREG>
__global__ void probe(unsigned long long* out, int spin_ticks, float* sink) {
  extern __shared__ char smem[];
  // force the kernarg fetch in front of the first stamp (like a real kernel: nothing starts before its arguments)
  asm volatile("" ::"s"(out), "s"(spin_ticks), "s"(sink));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = r[i] * 1.0001f + 0.5f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += r[i];
  if (s == 123.456f) sink[0] = s + smem[threadIdx.x];
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = t0;
    out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ void thrash(const uint4* __restrict__ src, size_t n, uint4* sink) {
  uint4 a = make_uint4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
  }
  if (a.x == 0x12345678u && a.y == 77u) sink[0] = a;
}
static uint4* g_big = nullptr;
static bool g_thrash = false;

template <int NREG>
void run(int grid, int block, int lds, const char* tag) {
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * grid);
  hipMalloc(&sink, 64);
  auto k = probe<NREG>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<double> last, med;
  std::vector<unsigned long long> h(2 * grid);
  for (int it = 0; it < 30; ++it) {
    if (g_thrash) {  // stream 1 GiB through the caches / TLBs first, like the kernels around a decode-attention launch
      hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, 0, g_big, (size_t)(1u << 30) / 16, (uint4*)sink);
      hipDeviceSynchronize();
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d, 600, sink);  // spin 6 us: nobody exits before all started
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> e(grid);
    for (int i = 0; i < grid; ++i) e[i] = h[2 * i];
    std::sort(e.begin(), e.end());
    if (it >= 5) {
      last.push_back((e[grid - 1] - e[0]) / 100.0);
      med.push_back((e[grid / 2] - e[0]) / 100.0);
    }
  }
  std::sort(last.begin(), last.end());
  std::sort(med.begin(), med.end());
  printf("%-28s grid %4d block %4d lds %6d regs~%3d : entry skew median-wg %.2f us, last-wg %.2f us (median of %zu)\n", tag,
         grid, block, lds, NREG, med[med.size() / 2], last[last.size() / 2], last.size());
  hipFree(d);
  hipFree(sink);
}

int main(int argc, char** argv) {
  g_thrash = argc > 1;
  if (g_thrash) {
    hipMalloc(&g_big, (size_t)1 << 30);
    hipMemset(g_big, 1, (size_t)1 << 30);
    printf("with a 1 GiB streaming read in front of every launch:\n");
  }
  run<4>(256, 256, 0, "small");
  run<4>(256, 256, 65536, "lds 64K");
  run<4>(256, 256, 131072, "lds 128K");
  run<4>(256, 256, 160 * 1024, "lds 160K");
  run<128>(256, 256, 0, "regs 128");
  run<128>(256, 256, 131072, "regs 128 + lds 128K");
  run<4>(256, 512, 131072, "block 512 + lds 128K");
  run<4>(256, 1024, 131072, "block 1024 + lds 128K");
  run<4>(512, 256, 65536, "grid 512 + lds 64K");
  run<4>(1024, 256, 32768, "grid 1024 + lds 32K");
  run<4>(1024, 256, 0, "grid 1024");
  run<4>(2048, 64, 0, "grid 2048 x 64");
  return 0;
}
