// Does an out-of-range lane of buffer_load_dwordx4 ... lds write zeros to LDS, or leave the bytes alone?
// hipcc --offload-arch=gfx950 -O2 tools/dbg/probe_dma.cpp -o tools/dbg/probe_dma && tools/dbg/probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const char* p, unsigned* o, unsigned nrec) {
  extern __shared__ char smem[];
  unsigned* w = (unsigned*)smem;
  for (int i = threadIdx.x; i < 512; i += 64) w[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned long long a = (unsigned long long)p;
  u32x4 rs = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, nrec, 0x00020000u};
  unsigned voff = threadIdx.x * 16;
  asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "i"(0) : "memory", "scc");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) o[i] = w[i];
}
int main() {
  char* d; unsigned* o;
  hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 512u);  // lanes 0..31 in range, 32..63 out of range
  std::vector<unsigned> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  printf("lane 0: %08x %08x  lane 31: %08x  lane 32: %08x %08x  lane 63: %08x  word 256 (untouched): %08x\n", r[0], r[1], r[31 * 4], r[32 * 4],
         r[32 * 4 + 1], r[63 * 4 + 3], r[256]);
  return 0;
}
