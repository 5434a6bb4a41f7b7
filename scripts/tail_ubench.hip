// scripts/tail_ubench.hip -- standalone microbenchmark of the fused layer-tail kernel (encoder_tail.hip;
// argv[4] = 0: the MLP block alone) on random
// bf16 data (MiniLM shape: hidden 384, ffn 1536), with a checksum of the output so that two builds can be
// compared (bit-equality with the two-GEMM path is tests/test_encoder_gpu.py's job).  Use >= 1000 reps:
// the first milliseconds run at ramp-up clocks.  -DMX_TAIL_ABLATE=N selects the ablations of the kernel.
// argv: rows ffn reps po skew_iters skew_shift skew_hi.  Then the same for tail2_kernel (encoder_tail2.hip) with the
// largest difference between the two outputs (different rounding points: a few bf16 ulps).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I memex_amd/csrc -I scripts scripts/tail_ubench.hip \
//        memex_amd/csrc/encoder_tail.hip scripts/encoder_tail2.hip -o build_ub/tail_ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <utility>
#include "encoder_kernels.h"
#include "encoder_tail2.h"
using namespace mx;
#ifndef MX_TAIL_ABLATE
#define MX_TAIL_ABLATE 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill16(unsigned short* p, size_t n, unsigned seed, unsigned expo) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (unsigned short)(expo + (h & 0x7f) + ((h >> 16) & 0x8000u)); }
}
__global__ void fillf(float* p, size_t n, float v, float step) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + step * (float)(i % 37); }
// K-blocked bf16 [k/32][rows][32] -> logical f32 [rows][k]
static void to_logical(const std::vector<unsigned short>& kb, size_t rows, size_t k, std::vector<float>& w) {
  w.resize(rows * k);
  for (size_t n = 0; n < rows; ++n) for (size_t c = 0; c < k; ++c) {
    const unsigned int u = (unsigned int)kb[((c >> 5) * rows + n) * 32 + (c & 31)] << 16; float f; memcpy(&f, &u, 4); w[n * k + c] = f;
  }
}
static uint16_t bf16_exact(float f) { unsigned int u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
int main(int argc, char** argv) {
  int m = argc > 1 ? atoi(argv[1]) : 131072; int f = argc > 2 ? atoi(argv[2]) : 1536; int reps = argc > 3 ? atoi(argv[3]) : 2000; int po = argc > 4 ? atoi(argv[4]) : 1;
  const int skew_iters = argc > 5 ? atoi(argv[5]) : 0, skew_shift = argc > 6 ? atoi(argv[6]) : 8, skew_hi = argc > 7 ? atoi(argv[7]) : 512;
  const int wmask = argc > 8 ? atoi(argv[8]) : 3;  // debugging: bit 0 = Wo non-zero, bit 1 = W1 / W2 non-zero
  bf16_t *x, *ctx, *w1, *w2, *wo, *wf, *out2; float *b1, *b2, *g, *b;
  CK(hipMalloc(&x, (size_t)m * 384 * 2)); CK(hipMalloc(&out2, (size_t)m * 384 * 2));
  CK(hipMalloc(&w1, (size_t)f * 384 * 2)); CK(hipMalloc(&w2, (size_t)f * 384 * 2)); CK(hipMalloc(&wf, tail_stream_elems(f) * 2)); CK(hipMalloc(&wo, 384 * 384 * 2)); CK(hipMalloc(&ctx, (size_t)m * 384 * 2));
  CK(hipMalloc(&b1, f * 4)); CK(hipMalloc(&b2, 384 * 4)); CK(hipMalloc(&g, 384 * 4)); CK(hipMalloc(&b, 384 * 4));
  // activations of magnitude 0.5 .. 1, weights 0.03 .. 0.06 (0.016 .. 0.03 for W2): every term of the tail matters
  fill16<<<4096, 256>>>((unsigned short*)x, (size_t)m * 384, 1, 0x3f00); fill16<<<4096, 256>>>((unsigned short*)ctx, (size_t)m * 384, 7, 0x3f00); fill16<<<64, 256>>>((unsigned short*)wo, (size_t)384 * 384, 5, 0x3d00); fill16<<<256, 256>>>((unsigned short*)w1, (size_t)f * 384, 2, 0x3d00); fill16<<<256, 256>>>((unsigned short*)w2, (size_t)f * 384, 3, 0x3c80);
  if (!(wmask & 1)) CK(hipMemset(wo, 0, 384 * 384 * 2)); if (!(wmask & 2)) { CK(hipMemset(w1, 0, (size_t)f * 384 * 2)); CK(hipMemset(w2, 0, (size_t)f * 384 * 2)); }
  fillf<<<8, 256>>>(b1, f, 0.01f, 0.003f); fillf<<<2, 256>>>(b2, 384, 0.01f, -0.002f); fillf<<<2, 256>>>(g, 384, 1.0f, 0.01f); fillf<<<2, 256>>>(b, 384, 0.0f, 0.005f);
  CK(hipDeviceSynchronize());
  { std::vector<unsigned short> kb((size_t)f * 384), kbo((size_t)384 * 384); std::vector<float> l1, l2, lo; std::vector<uint16_t> st(tail_stream_elems(f));
    CK(hipMemcpy(kb.data(), w1, kb.size() * 2, hipMemcpyDeviceToHost)); to_logical(kb, f, 384, l1);
    CK(hipMemcpy(kb.data(), w2, kb.size() * 2, hipMemcpyDeviceToHost)); to_logical(kb, 384, f, l2);
    CK(hipMemcpy(kbo.data(), wo, kbo.size() * 2, hipMemcpyDeviceToHost)); to_logical(kbo, 384, 384, lo);
    tail_stream_layout(lo.data(), l1.data(), l2.data(), f, st.data(), bf16_exact); CK(hipMemcpy(wf, st.data(), st.size() * 2, hipMemcpyHostToDevice)); }
  CK(tail_setup());
  TailParams p2{}; p2.ctx = po ? ctx : nullptr; p2.ldc = 384; p2.bo = b2; p2.ln1g = g; p2.ln1b = b; p2.x = x; p2.ldx = 384; p2.b1 = b1; p2.b2 = b2; p2.f = f; p2.m = m; p2.out = out2; p2.ldo = 384; p2.gamma = g; p2.beta = b; p2.eps = 1e-12f; p2.wf = wf;
  p2.skew_iters = skew_iters; p2.skew_shift = skew_shift; p2.skew_hi = skew_hi; p2.trace = nullptr;
  CK(hipMemset(out2, 0xff, (size_t)m * 384 * 2));
  CK(launch_tail(0, p2)); CK(hipDeviceSynchronize());
  { std::vector<unsigned short> c((size_t)m * 384);
    CK(hipMemcpy(c.data(), out2, c.size() * 2, hipMemcpyDeviceToHost));
    unsigned long long h = 1469598103934665603ull; for (unsigned short v : c) { h ^= v; h *= 1099511628211ull; }
    printf("output checksum %016llx\n", h); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double fl = 4.0 * (double)m * 384 * f + (po ? 2.0 * (double)m * 384 * 384 : 0.0);
  for (int i = 0; i < 3; ++i) CK(launch_tail(0, p2));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK(launch_tail(0, p2)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("tail po=%d ablate=%d skew=%d(bit %d, < %d) m=%d f=%d: %.1f us  %.0f TFLOP/s (%.1f%% of 2.5 PF)\n", po, MX_TAIL_ABLATE, skew_iters, skew_shift, skew_hi, m, f, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0);
  if (argc > 9 && atoi(argv[9]) > 0) {
    // argv[9] = MB: the same launches with COLD inputs -- between two launches a kernel streams that many MB through the
    // caches (in the encoder the tail's x was written three kernels and ~400 MB of q / k / v / ctx traffic earlier); each
    // launch timed by its own event pair.  argv[10] = 1: the flush touches x and ctx themselves last (warm inputs, same gaps)
    const size_t fb = (size_t)atoi(argv[9]) << 20; const int rewarm = argc > 10 ? atoi(argv[10]) : 0; unsigned short* junk; CK(hipMalloc(&junk, fb)); CK(hipMemset(junk, 0, fb));
    const int n2 = reps < 200 ? reps : 200; double tot = 0.0;
    for (int i = 0; i < n2 + 3; ++i) {
      fill16<<<4096, 256>>>(junk, fb / 2, 11 + i, 0x3f00);
      if (rewarm) { fill16<<<4096, 256>>>((unsigned short*)x, (size_t)m * 384, 1, 0x3f00); fill16<<<4096, 256>>>((unsigned short*)ctx, (size_t)m * 384, 7, 0x3f00); }
      CK(hipEventRecord(e0)); CK(launch_tail(0, p2)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t1; CK(hipEventElapsedTime(&t1, e0, e1)); if (i >= 3) tot += t1;
    }
    printf("tail with %d MB streamed between launches (rewarm %d): %.1f us per launch\n", atoi(argv[9]), rewarm, tot / n2 * 1e3);
    CK(hipFree(junk));
  }
#if MX_TAIL_TRACE
  { // one traced launch in steady state: phase durations per dispatch round, and who shares a CU
    const int nb = m / 64; unsigned long long* tr; CK(hipMalloc(&tr, (size_t)nb * 64)); CK(hipMemset(tr, 0, (size_t)nb * 64));
    p2.trace = tr; for (int i = 0; i < 20; ++i) CK(launch_tail(0, p2)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> t((size_t)nb * 8); CK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull; for (int b = 0; b < nb; ++b) if (t[b * 8] < t0) t0 = t[b * 8];
    const char* names[6] = {"skew", "prologue(ctx DMA)", "out-proj", "LN1", "chunk loop", "LN2+store"};
    for (int r0 = 0; r0 < nb; r0 += 512) {
      double st = 0, en = 0, d[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
      for (int b = r0; b < r0 + 512 && b < nb; ++b, ++n) { st += (double)(t[b * 8] - t0); en += (double)(t[b * 8 + 6] - t0); for (int i = 0; i < 6; ++i) d[i] += (double)(t[b * 8 + i + 1] - t[b * 8 + i]); }
      printf("blocks %4d..%4d: start %.1f us end %.1f us |", r0, r0 + n - 1, st / n / 100, en / n / 100);
      for (int i = 0; i < 6; ++i) printf(" %s %.1f", names[i], d[i] / n / 100); printf(" us\n");
    }
    // start times of the first 520 blocks, and whether block b and b + 256 run on the same CU
    int same = 0, cnt = 0; for (int b = 0; b + 256 < nb && b < 256; ++b, ++cnt) { const unsigned long long ka = t[b * 8 + 7], kb = t[(b + 256) * 8 + 7]; same += ((ka >> 32) == (kb >> 32)) && ((ka & 0xff00) == (kb & 0xff00)); }
    printf("block b and b+256 on the same CU: %d of %d\n", same, cnt);
    { // who shares a CU among the first 512 blocks: histogram of (second block - first block)
      std::vector<std::pair<unsigned long long, int>> ks; for (int b = 0; b < 512 && b < nb; ++b) ks.push_back({((t[b * 8 + 7] >> 32) << 16) | (t[b * 8 + 7] & 0xff00), b});
      std::sort(ks.begin(), ks.end()); int ncu = 0, pairs = 0; std::vector<int> deltas;
      for (size_t i = 0; i < ks.size();) { size_t j = i; while (j < ks.size() && ks[j].first == ks[i].first) ++j; ++ncu; if (j - i == 2) { ++pairs; deltas.push_back(ks[i + 1].second - ks[i].second); } i = j; }
      std::sort(deltas.begin(), deltas.end()); printf("first 512 blocks sit on %d distinct CUs, %d of them hold exactly two; delta between the two block ids: min %d median %d max %d\n", ncu, pairs, deltas.empty() ? -1 : deltas.front(), deltas.empty() ? -1 : deltas[deltas.size() / 2], deltas.empty() ? -1 : deltas.back()); }
    for (int b : {0, 1, 8, 255, 256, 257, 264, 511, 512, 513, 767, 768, 1023, 1024}) if (b < nb) printf("  block %4d: xcc %llu hw_id %08llx  start %.1f us end %.1f us\n", b, t[b * 8 + 7] >> 32, t[b * 8 + 7] & 0xffffffffull, (double)(t[b * 8] - t0) / 100, (double)(t[b * 8 + 6] - t0) / 100);
    unsigned long long tend = 0; for (int b = 0; b < nb; ++b) if (t[b * 8 + 6] > tend) tend = t[b * 8 + 6];
    printf("traced launch: %.1f us first start -> last end\n", (double)(tend - t0) / 100);
  }
#endif
  if (po) { // ---- tail2_kernel on the same inputs
    bf16_t *wf2, *out3; float *pf;
    CK(hipMalloc(&wf2, tail2_stream_elems(f) * 2)); CK(hipMalloc(&pf, tail2_param_floats() * 4)); CK(hipMalloc(&out3, (size_t)m * 384 * 2));
    { std::vector<unsigned short> kb((size_t)f * 384), kbo((size_t)384 * 384); std::vector<float> l1, l2, lo; std::vector<uint16_t> st(tail2_stream_elems(f));
      CK(hipMemcpy(kb.data(), w1, kb.size() * 2, hipMemcpyDeviceToHost)); to_logical(kb, f, 384, l1);
      CK(hipMemcpy(kb.data(), w2, kb.size() * 2, hipMemcpyDeviceToHost)); to_logical(kb, 384, f, l2);
      CK(hipMemcpy(kbo.data(), wo, kbo.size() * 2, hipMemcpyDeviceToHost)); to_logical(kbo, 384, 384, lo);
      tail2_stream_layout(lo.data(), l1.data(), l2.data(), f, st.data(), bf16_exact); CK(hipMemcpy(wf2, st.data(), st.size() * 2, hipMemcpyHostToDevice));
      std::vector<float> hb1(f), hb2(384), hg(384), hb(384), pp(tail2_param_floats());
      CK(hipMemcpy(hb1.data(), b1, f * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb2.data(), b2, 384 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hg.data(), g, 384 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, 384 * 4, hipMemcpyDeviceToHost));
      tail2_param_layout(hb2.data(), hg.data(), hb.data(), hb1.data(), hb2.data(), hg.data(), hb.data(), f, pp.data()); CK(hipMemcpy(pf, pp.data(), pp.size() * 4, hipMemcpyHostToDevice)); }
    // f64 reference of a few rows (exact erf GELU, no intermediate rounding): which kernel is off, and where
    std::vector<float> rl1, rl2, rlo; { std::vector<unsigned short> kb((size_t)f * 384), kbo((size_t)384 * 384);
      CK(hipMemcpy(kb.data(), w1, kb.size() * 2, hipMemcpyDeviceToHost)); to_logical(kb, f, 384, rl1);
      CK(hipMemcpy(kb.data(), w2, kb.size() * 2, hipMemcpyDeviceToHost)); to_logical(kb, 384, f, rl2);
      CK(hipMemcpy(kbo.data(), wo, kbo.size() * 2, hipMemcpyDeviceToHost)); to_logical(kbo, 384, 384, rlo); }
    auto bf = [](unsigned short u) { unsigned v = (unsigned)u << 16; float x_; memcpy(&x_, &v, 4); return (double)x_; };
    auto ref_row = [&](int r, std::vector<double>& outv) {
      std::vector<unsigned short> xrw(384), crw(384); std::vector<float> hb1(f), hb2(384), hg(384), hb(384);
      hipMemcpy(xrw.data(), (unsigned short*)x + (size_t)r * 384, 768, hipMemcpyDeviceToHost); hipMemcpy(crw.data(), (unsigned short*)ctx + (size_t)r * 384, 768, hipMemcpyDeviceToHost);
      hipMemcpy(hb1.data(), b1, f * 4, hipMemcpyDeviceToHost); hipMemcpy(hb2.data(), b2, 384 * 4, hipMemcpyDeviceToHost); hipMemcpy(hg.data(), g, 384 * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, 384 * 4, hipMemcpyDeviceToHost);
      std::vector<double> v(384), x1(384), hh(f), y(384);
      auto ln = [&](std::vector<double>& a, std::vector<double>& o) { double m_ = 0, q = 0; for (double t : a) m_ += t; m_ /= 384; for (double t : a) q += (t - m_) * (t - m_); q /= 384; for (int i = 0; i < 384; ++i) o[i] = (a[i] - m_) / sqrt(q + 1e-12) * hg[i] + hb[i]; };
      for (int n = 0; n < 384; ++n) { double a = hb2[n] + bf(xrw[n]); for (int k = 0; k < 384; ++k) a += bf(crw[k]) * rlo[(size_t)n * 384 + k]; v[n] = a; }
      ln(v, x1);
      for (int j = 0; j < f; ++j) { double a = hb1[j]; for (int k = 0; k < 384; ++k) a += x1[k] * rl1[(size_t)j * 384 + k]; hh[j] = 0.5 * a * (1.0 + erf(a / sqrt(2.0))); }
      for (int n = 0; n < 384; ++n) { double a = hb2[n] + x1[n]; for (int j = 0; j < f; ++j) a += hh[j] * rl2[(size_t)n * f + j]; y[n] = a; }
      outv.resize(384); ln(y, outv); };
    CK(tail2_setup());
    Tail2Params p3; static_cast<TailParams &>(p3) = p2; p3.wf2 = wf2; p3.pf = pf; p3.out = out3; p3.trace = nullptr;
    CK(hipMemset(out3, 0xff, (size_t)m * 384 * 2));
    CK(launch_tail2(0, p3)); CK(hipDeviceSynchronize());
    { std::vector<unsigned short> a((size_t)m * 384), c((size_t)m * 384);
      CK(hipMemcpy(a.data(), out2, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), out3, c.size() * 2, hipMemcpyDeviceToHost));
      double mx = 0, sum = 0, ref = 0; size_t bad = 0, worst = 0;
      for (size_t i = 0; i < a.size(); ++i) { unsigned ua = (unsigned)a[i] << 16, uc = (unsigned)c[i] << 16; float fa, fc; memcpy(&fa, &ua, 4); memcpy(&fc, &uc, 4);
        const double d = fabs((double)fa - fc); if (!(d <= 1e30)) { ++bad; continue; } if (d > mx) { mx = d; worst = i; } sum += d; ref += fabs(fa); }
      printf("tail2 vs tail: max |diff| %.5f (row %zu col %zu), mean |diff| %.6f, mean |value| %.4f, non-finite %zu\n", mx, worst / 384, worst % 384, sum / a.size(), ref / a.size(), bad);
      for (int r : {0, 37, 64, 127, (int)(worst / 384)}) { if (r >= m) continue; std::vector<double> rv; ref_row(r, rv); double e1 = 0, e2 = 0; int w1_ = 0, w2_ = 0;
        for (int i = 0; i < 384; ++i) { const double d1 = fabs(bf(a[(size_t)r * 384 + i]) - rv[i]), d2 = fabs(bf(c[(size_t)r * 384 + i]) - rv[i]); if (d1 > e1) { e1 = d1; w1_ = i; } if (d2 > e2) { e2 = d2; w2_ = i; } }
        printf("  row %6d vs f64 reference: tail max err %.5f (col %d)   tail2 max err %.5f (col %d)\n", r, e1, w1_, e2, w2_);
        if (r == 0) { printf("  row 0 tail2 - ref, features 0..63:"); for (int i = 0; i < 64; ++i) printf("%s%+.3f", i % 16 == 0 ? "\n    " : " ", bf(c[i]) - rv[i]); printf("\n"); } } }
    for (int i = 0; i < 3; ++i) CK(launch_tail2(0, p3));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) CK(launch_tail2(0, p3)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("tail2 m=%d f=%d: %.1f us  %.0f TFLOP/s (%.1f%% of 2.5 PF)\n", m, f, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 25.0);
#if MX_TAIL2_TRACE
    { const int nb = m / 128; unsigned long long* tr; CK(hipMalloc(&tr, (size_t)nb * 64)); CK(hipMemset(tr, 0, (size_t)nb * 64));
      p3.trace = tr; for (int i = 0; i < 20; ++i) CK(launch_tail2(0, p3)); CK(hipDeviceSynchronize());
      std::vector<unsigned long long> t((size_t)nb * 8); CK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, tend = 0; for (int b2_ = 0; b2_ < nb; ++b2_) { if (t[b2_ * 8] < t0) t0 = t[b2_ * 8]; if (t[b2_ * 8 + 5] > tend) tend = t[b2_ * 8 + 5]; }
      const char* names[5] = {"prologue", "out-proj", "LN1", "MLP loop", "LN2+store"};
      for (int r0 = 0; r0 < nb; r0 += 256) { double st = 0, en = 0, d[5] = {0, 0, 0, 0, 0}; int n = 0;
        for (int b2_ = r0; b2_ < r0 + 256 && b2_ < nb; ++b2_, ++n) { st += (double)(t[b2_ * 8] - t0); en += (double)(t[b2_ * 8 + 5] - t0); for (int i = 0; i < 5; ++i) d[i] += (double)(t[b2_ * 8 + i + 1] - t[b2_ * 8 + i]); }
        printf("tail2 blocks %4d..%4d: start %.1f us end %.1f us |", r0, r0 + n - 1, st / n / 100, en / n / 100); for (int i = 0; i < 5; ++i) printf(" %s %.1f", names[i], d[i] / n / 100); printf(" us\n"); }
      printf("tail2 traced launch: %.1f us first start -> last end\n", (double)(tend - t0) / 100); }
#endif
  }
  return 0;
}
