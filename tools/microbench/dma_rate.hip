// Per-CU global -> LDS staging rate on MI355X: LDS-DMA (global_load_lds_dwordx4) vs register staging
// (global_load_dwordx4 + ds_write_b128), by source residency (L2-resident table shared by all workgroups / a stream
// far larger than the caches), waves per workgroup and pieces in flight.  One workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip && ./dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: LDS-DMA, 1: registers + ds_write.  PPS pieces (1 KB each) per wave per step, DEPTH steps in flight.
template <int MODE, int PPS, int DEPTH>
__global__ void stream_kernel(const char* __restrict__ src, size_t span_bytes, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t step_bytes = (size_t)nw * PPS * 1024;
  // every workgroup walks the span from its own start (L2-resident span: all share it; large span: disjoint parts)
  size_t pos = ((size_t)blockIdx.x * 7919u * step_bytes) % span_bytes;
  char* my = smem + (size_t)wave * PPS * 1024 * (DEPTH + 1);
  u32x4 regs[PPS];
  unsigned acc = 0;
  for (int s = 0; s < steps + DEPTH; ++s) {
    if (s < steps) {
      const char* g = src + pos + (size_t)wave * PPS * 1024 + lane * 16;
      char* l = my + (s % (DEPTH + 1)) * PPS * 1024;
#pragma unroll
      for (int i = 0; i < PPS; ++i) {
        if (MODE == 0) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 1024),
                                           (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
        } else {
          regs[i] = *reinterpret_cast<const u32x4*>(g + i * 1024);
        }
      }
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < PPS; ++i) *reinterpret_cast<u32x4*>(l + i * 1024 + lane * 16) = regs[i];
      }
      pos += step_bytes;
      if (pos + step_bytes > span_bytes) pos = 0;
    }
    if (MODE == 0) {
      if (s >= DEPTH) {
        if (s < steps) wait_vm<DEPTH * PPS>(); else wait_vm<0>();
        acc += *reinterpret_cast<unsigned*>(my + ((s - DEPTH) % (DEPTH + 1)) * PPS * 1024 + lane * 4);
      }
    } else {
      acc += regs[0].x;
    }
  }
  if (acc == 0x12345678u) sink[0] = 1.f;
}

template <int MODE, int PPS, int DEPTH>
void run(const char* name, const char* src, size_t span, int nwaves, int steps, float* sink) {
  const size_t lds = (size_t)nwaves * PPS * 1024 * (DEPTH + 1);
  hipFuncSetAttribute((const void*)stream_kernel<MODE, PPS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, PPS, DEPTH>), dim3(256), dim3(nwaves * 64), lds, 0, src, span, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = 256.0 * steps * nwaves * PPS * 1024;
  printf("%-34s waves %d pieces/step %d depth %d : %7.1f GB/s per CU  (%5.2f TB/s chip, %.0f us)\n", name, nwaves, PPS, DEPTH,
         bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12, best * 1e3);
}

int main() {
  const size_t big = (size_t)2 << 30;
  char* buf;
  float* sink;
  hipMalloc(&buf, big);
  hipMalloc(&sink, 4);
  hipMemset(buf, 1, big);
  const size_t l2span = 1 << 20;        // 1 MB: resident in every XCD's L2
  const size_t mall = (size_t)96 << 20; // 96 MB: Infinity Cache
  const int steps = 400;
#define BOTH(P, D, W)                                                           \
  run<0, P, D>("LDS-DMA    L2-resident", buf, l2span, W, steps, sink);           \
  run<1, P, D>("reg+dswrite L2-resident", buf, l2span, W, steps, sink);          \
  run<0, P, D>("LDS-DMA    96 MB (MALL)", buf, mall, W, steps, sink);            \
  run<1, P, D>("reg+dswrite 96 MB (MALL)", buf, mall, W, steps, sink);           \
  run<0, P, D>("LDS-DMA    2 GB (HBM)", buf, big, W, steps, sink);               \
  run<1, P, D>("reg+dswrite 2 GB (HBM)", buf, big, W, steps, sink);
  BOTH(2, 2, 8)
  BOTH(4, 2, 8)
  BOTH(4, 3, 4)
  BOTH(8, 2, 4)
  BOTH(4, 2, 16)
  return 0;
}
