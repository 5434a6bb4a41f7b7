// scripts/mlp_ubench.hip -- standalone microbenchmark of the fused MLP kernel (encoder_mlp.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMX_MLP_ABLATE=N -I memex_amd/csrc \
//        scripts/mlp_ubench.hip memex_amd/csrc/encoder_mlp.hip -o build_ub/mlp_ub_N
// Not product code: times the kernel on random bf16 data (MiniLM shape: hidden 384, ffn 1536).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "encoder_kernels.h"
using namespace mx;
#ifndef MX_MLP_ABLATE
#define MX_MLP_ABLATE 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill16(unsigned short* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (unsigned short)(0x3a00u + (h & 0x3ff) + ((h >> 16) & 0x8000u)); }
}
__global__ void fillf(float* p, size_t n, float v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
#ifdef MX_MLP_TRACE
namespace mx { extern __device__ unsigned long long g_mlp_trace[32]; }
#endif
int main(int argc, char** argv) {
  int m = argc > 1 ? atoi(argv[1]) : 131072; int f = argc > 2 ? atoi(argv[2]) : 1536; int reps = argc > 3 ? atoi(argv[3]) : 20; int zero = argc > 4 ? atoi(argv[4]) : 0;
  bf16_t *x, *w1, *w2, *out; float *b1, *b2, *g, *b;
  CK(hipMalloc(&x, (size_t)m * 384 * 2)); CK(hipMalloc(&out, (size_t)m * 384 * 2)); CK(hipMalloc(&w1, (size_t)f * 384 * 2)); CK(hipMalloc(&w2, (size_t)f * 384 * 2));
  CK(hipMalloc(&b1, f * 4)); CK(hipMalloc(&b2, 384 * 4)); CK(hipMalloc(&g, 384 * 4)); CK(hipMalloc(&b, 384 * 4));
  fill16<<<4096, 256>>>((unsigned short*)x, (size_t)m * 384, 1); fill16<<<256, 256>>>((unsigned short*)w1, (size_t)f * 384, 2); fill16<<<256, 256>>>((unsigned short*)w2, (size_t)f * 384, 3);
  fillf<<<8, 256>>>(b1, f, 0.01f); fillf<<<2, 256>>>(b2, 384, 0.01f); fillf<<<2, 256>>>(g, 384, 1.0f); fillf<<<2, 256>>>(b, 384, 0.0f);
  if (zero) { CK(hipMemset(x, 0, (size_t)m * 384 * 2)); CK(hipMemset(w1, 0, (size_t)f * 384 * 2)); CK(hipMemset(w2, 0, (size_t)f * 384 * 2)); }
  CK(mlp_setup());
  MlpParams p{}; p.x = x; p.ldx = 384; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.f = f; p.m = m; p.out = out; p.ldo = 384; p.gamma = g; p.beta = b; p.eps = 1e-12f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(launch_mlp(0, p));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK(launch_mlp(0, p)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
#ifdef MX_MLP_TRACE
  { unsigned long long tr[32]; CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(mx::g_mlp_trace), sizeof(tr)));
    double tot = 0; for (int t = 1; t < 10; ++t) tot += (double)tr[t];
    printf("trace (workgroup 0, wave 0; last launch): %.0f ticks in the main loop\n", tot);
    const char* nm[16] = {"", "MFMA block before a publish", "vmcnt wait", "barrier", "DMA issue", "G1 stage MFMAs+reads", "E1", "drain publish + exposed reads", "G2 k-step 0", "G2 k-step 1 (after its publish)"};
    for (int t = 1; t < 10; ++t) if (tr[16 + t]) printf("  tag %d %-34s n=%5llu  %9.0f ticks  %5.1f%%  avg %.0f\n", t, nm[t], tr[16 + t], (double)tr[t], 100.0 * tr[t] / tot, (double)tr[t] / tr[16 + t]); }
#endif
  double fl = 4.0 * (double)m * 384 * f;
  printf("mlp ablate=%d zero=%d m=%d f=%d: %.1f us  %.0f TFLOP/s (%.1f%% of 2.5 PF)\n", MX_MLP_ABLATE, zero, m, f, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0);
  return 0;
}
