// Cost of the end-of-workgroup flush of a reduction on MI355X, by accumulator type.  Every workgroup of a launch ends
// with ONE wave instruction of C atomics on the SAME C addresses (per-channel statistics) or on its own addresses
// (weight-gradient tiles): fp32 atomic add (order-dependent rounding) against 64-bit integer atomic adds on
// fixed-point accumulators (order-independent: the deterministic form) -- one word, or two words {lo, hi} stored
// interleaved (one 16-byte slot) or as two planes.  Also the same inside a workgroup on LDS.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// MODE 0: float add; 1: one int64 add; 2: two int64 adds, interleaved {lo,hi}; 3: two int64 adds, two planes
template <int MODE>
__global__ void same_addr_kernel(float* f, long long* q, int C, int reps) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const float v = 1.0f + (float)((blockIdx.x * 131 + c) & 1023) * 0x1p-20f;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) atomicAdd(f + c, v);
    if (MODE == 1) atomicAdd((unsigned long long*)q + c, (unsigned long long)(long long)(v * 0x1p20f));
    if (MODE == 2) {
      atomicAdd((unsigned long long*)q + 2 * c, (unsigned long long)(long long)(v * 0x1p20f));
      atomicAdd((unsigned long long*)q + 2 * c + 1, (unsigned long long)(long long)(v * 0x1p10f));
    }
    if (MODE == 3) {
      atomicAdd((unsigned long long*)q + c, (unsigned long long)(long long)(v * 0x1p20f));
      atomicAdd((unsigned long long*)q + 4096 + c, (unsigned long long)(long long)(v * 0x1p10f));
    }
  }
}

// every workgroup flushes a TILE of `tile` values to addresses shared by `share` workgroups (weight-gradient splits)
template <int MODE>
__global__ void tile_flush_kernel(float* f, long long* q, int tile, int ntiles) {
  const int t = blockIdx.x % ntiles;
  for (int i = threadIdx.x; i < tile; i += blockDim.x) {
    const size_t a = (size_t)t * tile + i;
    const float v = 1.0f + (float)((blockIdx.x * 131 + i) & 1023) * 0x1p-20f;
    if (MODE == 0) atomicAdd(f + a, v);
    if (MODE == 1) atomicAdd((unsigned long long*)q + a, (unsigned long long)(long long)(v * 0x1p20f));
    if (MODE == 2) {
      atomicAdd((unsigned long long*)q + 2 * a, (unsigned long long)(long long)(v * 0x1p20f));
      atomicAdd((unsigned long long*)q + 2 * a + 1, (unsigned long long)(long long)(v * 0x1p10f));
    }
  }
}

// LDS: 256 threads add into C slots `reps` times, result to global so nothing is optimised away
template <int MODE>
__global__ void lds_kernel(float* out, int C, int reps) {
  __shared__ float sf[256];
  __shared__ unsigned long long sq[512];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) sf[i] = 0.f;
  for (int i = threadIdx.x; i < 512; i += blockDim.x) sq[i] = 0;
  __syncthreads();
  const int c = threadIdx.x % C;
  const float v = 1.0f + (float)(threadIdx.x & 63) * 0x1p-20f;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) atomicAdd(&sf[c], v);
    if (MODE == 1) atomicAdd(&sq[c], (unsigned long long)(long long)(v * 0x1p20f));
    if (MODE == 2) {
      atomicAdd(&sq[2 * c], (unsigned long long)(long long)(v * 0x1p20f));
      atomicAdd(&sq[2 * c + 1], (unsigned long long)(long long)(v * 0x1p10f));
    }
  }
  __syncthreads();
  if (threadIdx.x < C) out[blockIdx.x * C + threadIdx.x] = sf[threadIdx.x] + (float)sq[threadIdx.x] + (float)sq[2 * threadIdx.x + 1];
}

template <typename F>
float time_us(F launch, int iters = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}

int main() {
  float* f; long long* q; float* out;
  hipMalloc(&f, 64 << 20); hipMalloc(&q, 256 << 20); hipMalloc(&out, 16 << 20);
  hipMemset(f, 0, 64 << 20); hipMemset(q, 0, 256 << 20);
  const char* names[4] = {"f32", "i64", "i64x2 interleaved", "i64x2 planes"};
  printf("## same C addresses, one wave instruction per workgroup (us per launch)\n");
  printf("| C | workgroups | f32 | i64 | i64x2 interleaved | i64x2 planes |\n|---|---|---|---|---|---|\n");
  for (int C : {8, 64, 256}) {
    for (int nwg : {128, 512, 2048, 8192}) {
      float t[4];
      t[0] = time_us([&] { same_addr_kernel<0><<<nwg, 256>>>(f, q, C, 1); });
      t[1] = time_us([&] { same_addr_kernel<1><<<nwg, 256>>>(f, q, C, 1); });
      t[2] = time_us([&] { same_addr_kernel<2><<<nwg, 256>>>(f, q, C, 1); });
      t[3] = time_us([&] { same_addr_kernel<3><<<nwg, 256>>>(f, q, C, 1); });
      printf("| %d | %d | %.1f | %.1f | %.1f | %.1f |\n", C, nwg, t[0], t[1], t[2], t[3]);
    }
  }
  printf("\n## tile flush: `splits` workgroups add a tile of `tile` values each onto the same tile (us per launch)\n");
  printf("| tile values | tiles | splits | f32 | i64 | i64x2 interleaved |\n|---|---|---|---|---|---|\n");
  for (int tile : {16384, 73728}) {
    for (int ntiles : {1, 4}) {
      for (int splits : {16, 64, 128}) {
        float t[3];
        const int nwg = ntiles * splits;
        t[0] = time_us([&] { tile_flush_kernel<0><<<nwg, 256>>>(f, q, tile, ntiles); });
        t[1] = time_us([&] { tile_flush_kernel<1><<<nwg, 256>>>(f, q, tile, ntiles); });
        t[2] = time_us([&] { tile_flush_kernel<2><<<nwg, 256>>>(f, q, tile, ntiles); });
        printf("| %d | %d | %d | %.1f | %.1f | %.1f |\n", tile, ntiles, splits, t[0], t[1], t[2]);
      }
    }
  }
  printf("\n## LDS: 256 threads x 64 adds into C slots, 1024 workgroups (us per launch)\n| C | f32 | i64 | i64x2 |\n|---|---|---|---|\n");
  for (int C : {8, 32, 128, 256}) {
    float t[3];
    t[0] = time_us([&] { lds_kernel<0><<<1024, 256>>>(out, C, 64); });
    t[1] = time_us([&] { lds_kernel<1><<<1024, 256>>>(out, C, 64); });
    t[2] = time_us([&] { lds_kernel<2><<<1024, 256>>>(out, C, 64); });
    printf("| %d | %.1f | %.1f | %.1f |\n", C, t[0], t[1], t[2]);
  }
  (void)names;
  return 0;
}
