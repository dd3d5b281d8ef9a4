// micro-benchmark: dependent load->store chains through (a) ds_ (address_space 3), (b) flat pointer into LDS,
// (c) global pointer (private 64 KB region per wave), with and without interleaved global stores; 1 lane active.
// Decides how the engine addresses LDS-resident candidate pools.  Build: hipcc --offload-arch=gfx950 -O3 lds_flat.hip -o lds_flat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define AS3 __attribute__((address_space(3)))
__global__ void __launch_bounds__(64) k_chain(int mode, int iters, int *gbuf, long long *out, int *sink)
{
  __shared__ int lds[4096];
  int *g = gbuf + (size_t)blockIdx.x * 16384;
  for (int i = threadIdx.x; i < 4096; i += 64) { lds[i] = (i * 7 + 1) & 4095; }
  for (int i = threadIdx.x; i < 16384; i += 64) g[i] = (i * 7 + 1) & 16383;
  __syncthreads();
  long long t0 = clock64();
  int idx = 0;
  if (threadIdx.x == 0) {
    if (mode == 0) { AS3 int *p = (AS3 int *)lds; for (int k = 0; k < iters; k++) { idx = p[idx]; p[(idx + 2048) & 4095] = idx; } }
    else if (mode == 1) { int *p = (int *)lds; asm volatile("" : "+v"(p)); for (int k = 0; k < iters; k++) { idx = p[idx]; p[(idx + 2048) & 4095] = idx; } }
    else if (mode == 2) { int *p = g; for (int k = 0; k < iters; k++) { idx = p[idx]; } }                          // global loads only
    else if (mode == 3) { int *p = g; for (int k = 0; k < iters; k++) { idx = p[idx]; p[(idx + 8192) & 16383] = idx; } }   // + a store per step
    else if (mode == 4) { AS3 int *p = (AS3 int *)lds; int *q = g; for (int k = 0; k < iters; k++) { idx = p[idx]; q[(idx + 8192) & 16383] = idx; } }  // LDS load chain + global store per step
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = idx; }
}
int main()
{
  const int blocks[3] = { 1, 1024, 4096 }, iters = 2000;
  int *gbuf; long long *out; int *sink;
  hipMalloc(&gbuf, sizeof(int) * 16384 * 4096); hipMalloc(&out, sizeof(long long) * 4096); hipMalloc(&sink, sizeof(int) * 4096);
  const char *names[5] = { "ds_ (AS3) load+store", "flat->LDS load+store", "global load chain", "global load+store chain", "ds_ load + global store" };
  for (int b = 0; b < 3; b++)
    for (int mode = 0; mode < 5; mode++) {
      for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_chain, dim3(blocks[b]), dim3(64), 0, 0, mode, iters, gbuf, out, sink); hipDeviceSynchronize(); }
      long long *h = (long long *)malloc(sizeof(long long) * blocks[b]);
      hipMemcpy(h, out, sizeof(long long) * blocks[b], hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < blocks[b]; i++) s += (double)h[i];
      printf("blocks %4d  %-26s %8.1f cycles/step\n", blocks[b], names[mode], s / blocks[b] / iters);
      free(h);
    }
  return 0;
}
