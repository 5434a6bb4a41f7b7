// scripts/scan8_ubench.hip -- standalone microbenchmark of the int8 filter-copy scan (scan8.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMX_SCAN8_ABLATE=N] -I memex_amd/csrc scripts/scan8_ubench.hip memex_amd/csrc/scan8.hip -o build_ub/scan8_ub_N
// Not product code: prints GB/s of the collect launch on random bytes with theta = +inf (no records) or a theta that
// passes some of the lanes (argv[5]); argv[6] = 0: small non-negative bytes instead of signed Gaussian-like ones (package power
// depends on how many bits toggle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "index_kernels.h"
using namespace mx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill8(unsigned* p, size_t n, unsigned seed, int signed_fill) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; unsigned o = 0;
    for (int b = 0; b < 4; ++b) {  // signed, roughly Gaussian with sigma 32 (what a quantised unit vector looks like): sum of 4 uniforms
      unsigned g = h * 2654435761u + b * 40503u; g ^= g >> 16; g *= 2246822519u; g ^= g >> 13;
      int v = (int)((g & 63) + ((g >> 6) & 63) + ((g >> 12) & 63) + ((g >> 18) & 63)) - 126; v = v * (signed_fill ? 1 : 0) + (signed_fill ? 0 : (int)(g & 63));
      o |= ((unsigned)v & 0xffu) << (8 * b);
    }
    p[i] = o; }
}
__global__ void fillf(float* p, size_t n, float v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main(int argc, char** argv) {
  size_t rows = argc > 1 ? atoll(argv[1]) : 10000000; int ds = argc > 2 ? atoi(argv[2]) : 384; int reps = argc > 3 ? atoi(argv[3]) : 10; int nwg = argc > 4 ? atoi(argv[4]) : 256;
  float thr = argc > 5 ? atof(argv[5]) : INFINITY;
  int kc = ds / 128; rows = rows / 64 * 64;
  float *tsc, *qsc, *theta; void *qf, *xh; float* lb; uint32_t *lc, *ovf;
  CK(hipMalloc(&xh, rows * ds)); CK(hipMalloc(&tsc, rows / 64 * kTscaleFloats * 4)); float* zq; CK(hipMalloc(&zq, 1024)); CK(hipMemset(zq, 0, 1024)); CK(hipMalloc(&qsc, 1024)); CK(hipMalloc(&theta, 1024)); CK(hipMalloc(&qf, 256 * ds));
  CK(hipMalloc(&lb, (size_t)256 * 512 * kRecCap * 64)); uint32_t* lt; CK(hipMalloc(&lt, (size_t)256 * 512 * kRecCap * 4)); CK(hipMalloc(&lc, 256 * 512 * 4)); CK(hipMalloc(&ovf, 1024));
  int sf = argc > 6 ? atoi(argv[6]) : 1; fill8<<<4096, 256>>>((unsigned*)xh, rows * ds / 4, 1, sf); fillf<<<1024, 256>>>(tsc, rows / 64 * kTscaleFloats, 1.0f); fillf<<<1, 256>>>(qsc, 256, 1.0f); fill8<<<64, 256>>>((unsigned*)qf, 256 * ds / 4, 3, sf);
  std::vector<float> th(256, thr); CK(hipMemcpy(theta, th.data(), 1024, hipMemcpyHostToDevice));
  CK(scan8_setup());
  ScanParams p; p.x = nullptr; p.xh = xh; p.scale = nullptr; p.qfrag = qf; p.theta = theta; p.n_rows = rows; p.tile_begin = 0; p.tile_end = rows / 64; p.tile_stride = 1; p.ds = ds; p.lane_max = (float*)lc; p.lane_rec = lb; p.lane_tile = lt; p.lane_cnt = lc; p.overflow = ovf; p.tscale = tsc; p.qscale = qsc; p.qa = zq; p.qb = zq;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) CK(launch_scan8(0, kc, true, nwg, p));
  CK(hipDeviceSynchronize());
  float best = 1e9, tot = 0;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); CK(launch_scan8(0, kc, true, nwg, p)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
  std::vector<uint32_t> cnt((size_t)nwg * 512); CK(hipMemcpy(cnt.data(), lc, cnt.size() * 4, hipMemcpyDeviceToHost)); double recs = 0; for (auto c : cnt) recs += c;
  double gb = (double)rows * ds / 1e9;
  printf("scan8 ablate=%d nwg=%d rows=%zu ds=%d theta=%g: avg %.3f ms (%.0f GB/s)  best %.3f ms (%.0f GB/s)  records/launch %.0f\n",
#ifdef MX_SCAN8_ABLATE
         MX_SCAN8_ABLATE,
#else
         0,
#endif
         nwg, rows, ds, thr, tot / reps, gb / (tot / reps) * 1e3, best, gb / best * 1e3, recs);
  return 0;
}
