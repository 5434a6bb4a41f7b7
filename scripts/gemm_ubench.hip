// scripts/gemm_ubench.hip -- standalone microbenchmark of the encoder's GEMM kernels on the shapes of one layer
// (argv: tokens hidden ffn reps): gemm_kernel (encoder_kernels.hip) in several tile configurations and pgemm_kernel
// (encoder_pgemm.hip).  Every configuration sums k in the same order and shares the epilogue arithmetic, so outputs
// must be bit-identical: each run is compared element by element with the first configuration of its GEMM (the one the
// GPU tests cover).  Random bf16 data (never zero-filled).  -DMX_GEMM_ABLATE=<bits> / -DMX_PGEMM_ABLATE=<bits> build
// the ablation variants (no DMA in the loop / no epilogue / ...): their outputs are not compared.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I memex_amd/csrc scripts/gemm_ubench.hip -o build_ub/gemm_ub
#include "encoder_kernels.hip"
#include "encoder_pgemm.hip"
#include "pgemm4.hip"

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
// res[0] = number of differing elements, res[1] = first differing index
__global__ void cmp16(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* res) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) if (a[i] != b[i]) { atomicAdd(&res[0], 1ull); atomicMin(&res[1], (unsigned long long)i); }
}

static const bool kAblated = (MX_GEMM_ABLATE != 0) || (MX_PGEMM_ABLATE != 0);
static unsigned short* g_ref = nullptr;   // reference output of the current GEMM (first configuration)
static unsigned long long* g_res = nullptr;

// verdict of `out` against the reference of the current GEMM (first = this run IS the reference)
static int verdict_of(const void* out, size_t out_bytes, bool first, char* verdict, size_t vn) {
  verdict[0] = 0;
  if (kAblated) return 0;
  if (first) { CK(hipMemcpy(g_ref, out, out_bytes, hipMemcpyDeviceToDevice)); snprintf(verdict, vn, "reference"); return 0; }
  unsigned long long h[2] = {0ull, ~0ull};
  CK(hipMemcpy(g_res, h, sizeof h, hipMemcpyHostToDevice));
  cmp16<<<2048, 256>>>((const unsigned short*)out, g_ref, out_bytes / 2, g_res);
  CK(hipMemcpy(h, g_res, sizeof h, hipMemcpyDeviceToHost));
  if (h[0] == 0) snprintf(verdict, vn, "bit-identical");
  else snprintf(verdict, vn, "MISMATCH %llu of %zu, first at %llu", h[0], out_bytes / 2, h[1]);
  return 0;
}

template <int EPI, int WM, int WN, int MI, int S>
static int run(const char* name, GemmParams p, int reps, size_t out_bytes, const void* out, bool first = false) {
  using G = GemmGeom<WM, WN, MI, 32, S>;
  char cfg[64]; snprintf(cfg, sizeof cfg, "gemm<%d,%d,%d,S%d> %dx%d", WM, WN, MI, S, G::BM, G::BN);
  if (p.m % G::BM || p.n % G::BN) { printf("%-11s %-26s shape not divisible\n", name, cfg); return 0; }
  CK((gemm_attr<EPI, WM, WN, MI, 32, S>()));
  CK(hipMemset((void*)out, 0xff, out_bytes));
  CK((gemm_go<EPI, WM, WN, MI, 32, S>(0, p))); CK(hipDeviceSynchronize());
  char verdict[128];
  if (verdict_of(out, out_bytes, first, verdict, sizeof verdict)) return 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK((gemm_go<EPI, WM, WN, MI, 32, S>(0, p)));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK((gemm_go<EPI, WM, WN, MI, 32, S>(0, p))); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double fl = 2.0 * p.m * (double)p.n * p.k;
  printf("%-11s %-26s %8.1f us %6.0f TFLOP/s (%.1f%%)  %s\n", name, cfg, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0, verdict);
  fflush(stdout);
  return 0;
}

static int run_p(const char* name, int epi, GemmParams p, int reps, size_t out_bytes, const void* out) {
  if (!pgemm_supported(epi, p)) { printf("%-11s %-26s shape not supported\n", name, "pgemm 256x256"); return 0; }
  CK(hipMemset((void*)out, 0xff, out_bytes));
  CK(launch_pgemm(0, epi, p)); CK(hipDeviceSynchronize());
  char verdict[128], verdict2[128];
  if (verdict_of(out, out_bytes, false, verdict, sizeof verdict)) return 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK(launch_pgemm(0, epi, p));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK(launch_pgemm(0, epi, p)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  // a second comparison after the timed launches: a race that only shows under back-to-back launches
  if (verdict_of(out, out_bytes, false, verdict2, sizeof verdict2)) return 1;
  const double fl = 2.0 * p.m * (double)p.n * p.k;
  printf("%-11s %-26s %8.1f us %6.0f TFLOP/s (%.1f%%)  %s%s%s\n", name, "pgemm 256x256", ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0, verdict,
         strcmp(verdict, verdict2) ? " | after timing: " : "", strcmp(verdict, verdict2) ? verdict2 : "");
  fflush(stdout);
  return 0;
}

// pgemm4_kernel (scripts/pgemm4.hip): the 1-wave-per-SIMD / 512-register form of pgemm_kernel
static int run_p4(const char* name, int epi, GemmParams p, int reps, size_t out_bytes, const void* out) {
  const bool kAbl4 = MX_PGEMM4_ABLATE != 0;
  if (!pgemm4_supported(epi, p)) { printf("%-11s %-26s shape not supported\n", name, "pgemm4 256x256 1w/SIMD"); return 0; }
  CK(hipMemset((void*)out, 0xff, out_bytes));
  CK(launch_pgemm4(0, epi, p, 256)); CK(hipDeviceSynchronize());
  char verdict[128] = "", verdict2[128] = "";
  if (!kAbl4 && verdict_of(out, out_bytes, false, verdict, sizeof verdict)) return 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) CK(launch_pgemm4(0, epi, p, 256));
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK(launch_pgemm4(0, epi, p, 256)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  if (!kAbl4 && verdict_of(out, out_bytes, false, verdict2, sizeof verdict2)) return 1;
  const double fl = 2.0 * p.m * (double)p.n * p.k;
  printf("%-11s %-26s %8.1f us %6.0f TFLOP/s (%.1f%%)  %s%s%s\n", name, "pgemm4 256x256 1w/SIMD", ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0, verdict,
         strcmp(verdict, verdict2) ? " | after timing: " : "", strcmp(verdict, verdict2) ? verdict2 : "");
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 131072, H = argc > 2 ? atoi(argv[2]) : 768, F = argc > 3 ? atoi(argv[3]) : 3072, reps = argc > 4 ? atoi(argv[4]) : 100;
  printf("gemm_ubench: tokens %d hidden %d ffn %d reps %d  MX_GEMM_ABLATE=%d MX_PGEMM_ABLATE=%d MX_PGEMM4_ABLATE=%d\n", m, H, F, reps, MX_GEMM_ABLATE, MX_PGEMM_ABLATE, MX_PGEMM4_ABLATE);
  bf16_t *x, *hbuf, *wqkv, *w1, *w2, *q, *k, *out; float *bias, *g, *b;
  CK(hipMalloc(&x, (size_t)m * H * 2)); CK(hipMalloc(&hbuf, (size_t)m * F * 2)); CK(hipMalloc(&q, (size_t)m * H * 2)); CK(hipMalloc(&k, (size_t)m * H * 2)); CK(hipMalloc(&out, (size_t)m * F * 2));
  CK(hipMalloc(&g_ref, (size_t)m * F * 2)); CK(hipMalloc(&g_res, 16));
  CK(hipMalloc(&wqkv, (size_t)3 * H * H * 2)); CK(hipMalloc(&w1, (size_t)F * H * 2)); CK(hipMalloc(&w2, (size_t)F * H * 2));
  CK(hipMalloc(&bias, (size_t)F * 4)); CK(hipMalloc(&g, H * 4)); CK(hipMalloc(&b, H * 4));
  fill16<<<4096, 256>>>((unsigned short*)x, (size_t)m * H, 1, 0x3f00); fill16<<<4096, 256>>>((unsigned short*)hbuf, (size_t)m * F, 7, 0x3e00);
  fill16<<<256, 256>>>((unsigned short*)wqkv, (size_t)3 * H * H, 5, 0x3d00); fill16<<<256, 256>>>((unsigned short*)w1, (size_t)F * H, 2, 0x3d00); fill16<<<256, 256>>>((unsigned short*)w2, (size_t)F * H, 3, 0x3c80);
  fillf<<<8, 256>>>(bias, F, 0.01f, 0.003f); fillf<<<2, 256>>>(g, H, 1.0f, 0.01f); fillf<<<2, 256>>>(b, H, 0.0f, 0.005f);
  CK(hipDeviceSynchronize());
  CK(pgemm_setup());
  { GemmParams p{}; p.a = x; p.lda = H; p.w = wqkv; p.w_rows = 3 * H; p.w_row0 = 0; p.bias = bias; p.m = m; p.n = 2 * H; p.k = H; p.out = q; p.out_k = k; p.ldo = H; p.hidden = H; p.qscale = 0.18f;
    // q and k are two buffers: compare q (the scaled half) and k separately through two passes
    run<EPI_QKV, 2, 2, 2, 4>("qk(q)", p, reps, (size_t)m * H * 2, q, true); run<EPI_QKV, 2, 4, 4, 3>("qk(q)", p, reps, (size_t)m * H * 2, q);
    run_p("qk(q)", EPI_QKV, p, reps, (size_t)m * H * 2, q); run_p4("qk(q)", EPI_QKV, p, reps, (size_t)m * H * 2, q);
    run<EPI_QKV, 2, 2, 2, 4>("qk(k)", p, 1, (size_t)m * H * 2, k, true); run_p("qk(k)", EPI_QKV, p, 1, (size_t)m * H * 2, k); run_p4("qk(k)", EPI_QKV, p, 1, (size_t)m * H * 2, k); }
  { GemmParams p{}; p.a = x; p.lda = H; p.w = w1; p.w_rows = F; p.w_row0 = 0; p.bias = bias; p.m = m; p.n = F; p.k = H; p.out = out; p.ldo = F;
    run<EPI_BIAS_GELU, 2, 2, 2, 4>("ffn1", p, reps, (size_t)m * F * 2, out, true); run<EPI_BIAS_GELU, 2, 4, 4, 3>("ffn1", p, reps, (size_t)m * F * 2, out);
    run_p("ffn1", EPI_BIAS_GELU, p, reps, (size_t)m * F * 2, out); run_p4("ffn1", EPI_BIAS_GELU, p, reps, (size_t)m * F * 2, out);
    run<EPI_BIAS, 2, 2, 2, 4>("ffn1-nogelu", p, reps, (size_t)m * F * 2, out, true); run_p("ffn1-nogelu", EPI_BIAS, p, reps, (size_t)m * F * 2, out); run_p4("ffn1-nogelu", EPI_BIAS, p, reps, (size_t)m * F * 2, out); }
  { GemmParams p{}; p.a = x; p.lda = H; p.w = wqkv; p.w_rows = 3 * H; p.w_row0 = 2 * H; p.bias = bias; p.m = m; p.n = H; p.k = H; p.out_vt = q; p.ldvt = m; p.hidden = H;
    run<EPI_VT, 2, 2, 2, 4>("vt", p, reps, (size_t)m * H * 2, q, true); run<EPI_VT, 2, 4, 4, 3>("vt", p, reps, (size_t)m * H * 2, q);
    run_p("vt", EPI_VT, p, reps, (size_t)m * H * 2, q); run_p4("vt", EPI_VT, p, reps, (size_t)m * H * 2, q); }
  if (H == 768) { GemmParams p{}; p.a = hbuf; p.lda = F; p.w = w2; p.w_rows = H; p.w_row0 = 0; p.bias = bias; p.m = m; p.n = H; p.k = F; p.out = out; p.ldo = H; p.res = x; p.ldres = H; p.gamma = g; p.beta = b; p.eps = 1e-12f;
    run<EPI_BIAS_RES_LN, 1, 8, 2, 3>("ffn2+ln", p, reps, (size_t)m * H * 2, out, true);
    // the same product without the LayerNorm epilogue (bias only): what a 256 x 256 schedule does on K = 3072
    run<EPI_BIAS, 2, 2, 2, 4>("ffn2", p, reps, (size_t)m * H * 2, out, true); run_p("ffn2", EPI_BIAS, p, reps, (size_t)m * H * 2, out); run_p4("ffn2", EPI_BIAS, p, reps, (size_t)m * H * 2, out);
    p.a = x; p.lda = H; p.w = wqkv; p.w_rows = 3 * H; p.k = H;
    run<EPI_BIAS_RES_LN, 1, 8, 2, 3>("oproj+ln", p, reps, (size_t)m * H * 2, out, true);
    run<EPI_BIAS, 2, 2, 2, 4>("oproj", p, reps, (size_t)m * H * 2, out, true); run_p("oproj", EPI_BIAS, p, reps, (size_t)m * H * 2, out); run_p4("oproj", EPI_BIAS, p, reps, (size_t)m * H * 2, out); }
  return 0;
}
