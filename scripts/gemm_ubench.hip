// scripts/gemm_ubench.hip -- standalone microbenchmark of the encoder's GEMM kernel (encoder_kernels.hip) in
// several tile configurations on the shapes of one layer (argv: tokens hidden ffn reps).  Every configuration
// sums k in the same order, so the output checksums of two configurations of one GEMM must be equal: a new
// configuration is checked against the one the GPU tests cover.  Random bf16 data (never zero-filled).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I memex_amd/csrc scripts/gemm_ubench.hip -o build_ub/gemm_ub
#include "encoder_kernels.hip"

#include <cstdio>
#include <cstring>
#include <vector>
using namespace mx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill16(unsigned short* p, size_t n, unsigned seed, unsigned expo) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (unsigned short)(expo + (h & 0x7f) + ((h >> 16) & 0x8000u)); }
}
__global__ void fillf(float* p, size_t n, float v, float step) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + step * (float)(i % 37); }

static unsigned long long checksum(const void* d, size_t bytes) {
  std::vector<unsigned short> c(bytes / 2); if (hipMemcpy(c.data(), d, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  unsigned long long h = 1469598103934665603ull; for (unsigned short v : c) { h ^= v; h *= 1099511628211ull; } return h;
}

template <int EPI, int WM, int WN, int MI, int S>
static int run(const char* name, GemmParams p, int reps, size_t out_bytes, const void* out) {
  using G = GemmGeom<WM, WN, MI, 32, S>;
  if (p.m % G::BM || p.n % G::BN) { printf("%-8s <%d,%d,%d,S%d> tile %dx%d: shape not divisible\n", name, WM, WN, MI, S, G::BM, G::BN); return 0; }
  CK((gemm_attr<EPI, WM, WN, MI, 32, S>()));
  CK(hipMemset((void*)out, 0xff, out_bytes));
  CK((gemm_go<EPI, WM, WN, MI, 32, S>(0, p))); CK(hipDeviceSynchronize());
  const unsigned long long cs = checksum(out, out_bytes);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK((gemm_go<EPI, WM, WN, MI, 32, S>(0, p)));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK((gemm_go<EPI, WM, WN, MI, 32, S>(0, p))); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double fl = 2.0 * p.m * (double)p.n * p.k;
  printf("%-8s <%d,%d,%d,S%d> tile %dx%d lds %3d KiB: %8.1f us %6.0f TFLOP/s (%.1f%%)  checksum %016llx\n", name, WM, WN, MI, S, G::BM, G::BN, G::LDS / 1024, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0, cs);
  return 0;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 131072, H = argc > 2 ? atoi(argv[2]) : 768, F = argc > 3 ? atoi(argv[3]) : 3072, reps = argc > 4 ? atoi(argv[4]) : 200;
  bf16_t *x, *hbuf, *wqkv, *w1, *w2, *q, *k, *out; float *bias, *g, *b;
  CK(hipMalloc(&x, (size_t)m * H * 2)); CK(hipMalloc(&hbuf, (size_t)m * F * 2)); CK(hipMalloc(&q, (size_t)m * H * 2)); CK(hipMalloc(&k, (size_t)m * H * 2)); CK(hipMalloc(&out, (size_t)m * F * 2));
  CK(hipMalloc(&wqkv, (size_t)3 * H * H * 2)); CK(hipMalloc(&w1, (size_t)F * H * 2)); CK(hipMalloc(&w2, (size_t)F * H * 2));
  CK(hipMalloc(&bias, (size_t)F * 4)); CK(hipMalloc(&g, H * 4)); CK(hipMalloc(&b, H * 4));
  fill16<<<4096, 256>>>((unsigned short*)x, (size_t)m * H, 1, 0x3f00); fill16<<<4096, 256>>>((unsigned short*)hbuf, (size_t)m * F, 7, 0x3e00);
  fill16<<<256, 256>>>((unsigned short*)wqkv, (size_t)3 * H * H, 5, 0x3d00); fill16<<<256, 256>>>((unsigned short*)w1, (size_t)F * H, 2, 0x3d00); fill16<<<256, 256>>>((unsigned short*)w2, (size_t)F * H, 3, 0x3c80);
  fillf<<<8, 256>>>(bias, F, 0.01f, 0.003f); fillf<<<2, 256>>>(g, H, 1.0f, 0.01f); fillf<<<2, 256>>>(b, H, 0.0f, 0.005f);
  CK(hipDeviceSynchronize());
  { GemmParams p{}; p.a = x; p.lda = H; p.w = wqkv; p.w_rows = 3 * H; p.w_row0 = 0; p.bias = bias; p.m = m; p.n = 2 * H; p.k = H; p.out = q; p.out_k = k; p.ldo = H; p.hidden = H; p.qscale = 0.18f;
    run<EPI_QKV, 2, 2, 2, 4>("qk", p, reps, (size_t)m * H * 2, q); run<EPI_QKV, 2, 4, 2, 4>("qk", p, reps, (size_t)m * H * 2, q);
    run<EPI_QKV, 4, 2, 2, 4>("qk", p, reps, (size_t)m * H * 2, q); run<EPI_QKV, 2, 4, 4, 3>("qk", p, reps, (size_t)m * H * 2, q); run<EPI_QKV, 2, 2, 4, 2>("qk", p, reps, (size_t)m * H * 2, q); }
  { GemmParams p{}; p.a = x; p.lda = H; p.w = w1; p.w_rows = F; p.w_row0 = 0; p.bias = bias; p.m = m; p.n = F; p.k = H; p.out = out; p.ldo = F;
    run<EPI_BIAS_GELU, 2, 2, 2, 4>("ffn1", p, reps, (size_t)m * F * 2, out); run<EPI_BIAS_GELU, 2, 4, 2, 4>("ffn1", p, reps, (size_t)m * F * 2, out);
    run<EPI_BIAS_GELU, 4, 2, 2, 4>("ffn1", p, reps, (size_t)m * F * 2, out); run<EPI_BIAS_GELU, 2, 4, 4, 3>("ffn1", p, reps, (size_t)m * F * 2, out); run<EPI_BIAS_GELU, 2, 2, 4, 2>("ffn1", p, reps, (size_t)m * F * 2, out); }
  { GemmParams p{}; p.a = x; p.lda = H; p.w = wqkv; p.w_rows = 3 * H; p.w_row0 = 2 * H; p.bias = bias; p.m = m; p.n = H; p.k = H; p.out_vt = q; p.ldvt = m; p.hidden = H;
    run<EPI_VT, 2, 2, 2, 4>("vt", p, reps, (size_t)m * H * 2, q); run<EPI_VT, 2, 4, 4, 3>("vt", p, reps, (size_t)m * H * 2, q); run<EPI_VT, 2, 4, 2, 4>("vt", p, reps, (size_t)m * H * 2, q); }
  if (H == 768) { GemmParams p{}; p.a = hbuf; p.lda = F; p.w = w2; p.w_rows = H; p.w_row0 = 0; p.bias = bias; p.m = m; p.n = H; p.k = F; p.out = out; p.ldo = H; p.res = x; p.ldres = H; p.gamma = g; p.beta = b; p.eps = 1e-12f;
    run<EPI_BIAS_RES_LN, 1, 8, 2, 3>("ffn2+ln", p, reps, (size_t)m * H * 2, out); run<EPI_BIAS_RES_LN, 1, 8, 4, 2>("ffn2+ln", p, reps, (size_t)m * H * 2, out);
    p.a = x; p.lda = H; p.w = wqkv; p.k = H;
    run<EPI_BIAS_RES_LN, 1, 8, 2, 3>("oproj+ln", p, reps, (size_t)m * H * 2, out); run<EPI_BIAS_RES_LN, 1, 8, 4, 2>("oproj+ln", p, reps, (size_t)m * H * 2, out); }
  return 0;
}
