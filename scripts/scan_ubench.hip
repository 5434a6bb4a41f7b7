// scripts/scan_ubench.hip -- standalone microbenchmark of the streaming scan kernel (scan.hip).
// Build (one binary per ablation level):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMX_SCAN_ABLATE=N -I memex_amd/csrc \
//         scripts/scan_ubench.hip memex_amd/csrc/scan.hip -o /tmp/scan_ub_N
// Not product code: prints GB/s of the main-stage launch on random data with theta = +inf.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "index_kernels.h"
using namespace mx;
#ifndef MX_SCAN_ABLATE
#define MX_SCAN_ABLATE 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(float* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = ((int)(h & 0xffff) - 32768) / 32768.0f; }
}
int main(int argc, char** argv) {
  size_t rows = argc > 1 ? atoll(argv[1]) : 10000000; int ds = argc > 2 ? atoi(argv[2]) : 384; int reps = argc > 3 ? atoi(argv[3]) : 10; int nwg = argc > 4 ? atoi(argv[4]) : 256;
  int kc = ds / 128; rows = rows / 32 * 32;
  float *x, *scale, *theta; void* qf; float* lb; uint32_t *lc, *ovf;
  CK(hipMalloc(&x, rows * ds * 4)); CK(hipMalloc(&scale, rows * 4)); CK(hipMalloc(&theta, 1024)); CK(hipMalloc(&qf, 256 * ds * 2));
  CK(hipMalloc(&lb, (size_t)256 * 512 * kRecCap * 64)); uint32_t* lt; CK(hipMalloc(&lt, (size_t)256 * 512 * kRecCap * 4)); CK(hipMalloc(&lc, 256 * 512 * 4)); CK(hipMalloc(&ovf, 1024));
  fill<<<4096, 256>>>(x, rows * ds, 1); fill<<<1024, 256>>>(scale, rows, 2); fill<<<64, 256>>>((float*)qf, 256 * ds / 2, 3);
  std::vector<float> th(256, INFINITY); CK(hipMemcpy(theta, th.data(), 1024, hipMemcpyHostToDevice));
  CK(scan_setup());
  ScanParams p; p.x = x; p.scale = scale; p.qfrag = qf; p.theta = theta; p.n_rows = rows; p.tile_begin = 0; p.tile_end = rows / 32; p.tile_stride = 1; p.ds = ds; p.lane_max = (float*)lc; p.lane_rec = lb; p.lane_tile = lt; p.lane_cnt = lc; p.overflow = ovf;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) CK(launch_scan(0, kc, true, nwg, p));
  CK(hipDeviceSynchronize());
  float best = 1e9, tot = 0;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); CK(launch_scan(0, kc, true, nwg, p)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
  double gb = (double)rows * ds * 4 / 1e9;
  printf("ablate=%d nwg=%d rows=%zu ds=%d: avg %.3f ms (%.0f GB/s)  best %.3f ms (%.0f GB/s)\n", MX_SCAN_ABLATE, nwg, rows, ds, tot / reps, gb / (tot / reps) * 1e3, best, gb / best * 1e3);
  return 0;
}
