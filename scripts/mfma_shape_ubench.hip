// mfma_shape_ubench.hip -- sustained rate of bare MFMA streams at the package power cap, by instruction shape and type:
// does v_mfma_f32_16x16x32_bf16 deliver more flops per joule than the 32x32x16 form every encoder kernel is built on?
// (no memory traffic: operands are random register contents, 4 independent accumulator chains per wave, 8 waves per CU)
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape_ubench.hip -o build_ub/mfma_shape_ub && build_ub/mfma_shape_ub
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

template <int MODE>
__global__ __launch_bounds__(512) void mfma_loop(const uint4 *__restrict__ seed, float *__restrict__ sink, int iters) {
    // eight distinct A and B fragments per lane, rotated: the multiplier inputs toggle between instructions as in a real GEMM
    uint4 av[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        av[k] = seed[(k * 1024 + threadIdx.x) & 8191];
        bv[k] = seed[(k * 1024 + 512 + threadIdx.x) & 8191];
    }
#define a0 av[(u + c) & 7]
#define b0 bv[(u * 3 + c) & 7]
    if (MODE == 0 || MODE == 2) {  // 32x32x16 bf16 / f16
        f32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int i = 0; i < iters; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[c], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (MODE == 1 || MODE == 3) {  // 16x16x32 bf16 / f16
        f32x4 acc[8];
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
        for (int i = 0; i < iters; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE == 1) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[c], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) s += acc[c][r];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (MODE == 5) {  // int8 16x16x64
        typedef __attribute__((ext_vector_type(4))) int i32x4v;
        i32x4v acc[8];
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) acc[c][r] = 0;
        for (int i = 0; i < iters; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c)
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), acc[c], 0, 0, 0);
        }
        int s = 0;
        for (int c = 0; c < 8; ++c)
            for (int r = 0; r < 4; ++r) s += acc[c][r];
        sink[blockIdx.x * 512 + threadIdx.x] = (float)s;
    } else {  // int8 32x32x32
        i32x16 acc[4];
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = 0;
        for (int i = 0; i < iters; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, b0), acc[c], 0, 0, 0);
        }
        int s = 0;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
        sink[blockIdx.x * 512 + threadIdx.x] = (float)s;
    }
}

template <int MODE>
static void run(const char *name, double flops_per_iter_per_wave, const uint4 *seed, float *sink, int cus, double seconds) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(cus), dim3(512), 0, 0, seed, sink, iters);
    hipDeviceSynchronize();
    // run for `seconds` so that the clock settles at the cap, time the last launches
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(cus), dim3(512), 0, 0, seed, sink, iters);
        hipDeviceSynchronize();
        n += 8;
    }
    hipEventRecord(e0);
    const int timed = 16;
    for (int k = 0; k < timed; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(cus), dim3(512), 0, 0, seed, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = flops_per_iter_per_wave * iters * 8.0 * cus * timed;
    printf("%-28s %8.1f T(FL)OP/s sustained (%.3f ms per launch, %d settle launches)\n", name, ops / (ms * 1e-3) / 1e12, ms / timed, n);
}

int main(int argc, char **argv) {
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;  // 0: all-zero operands (what the data-independent part costs)
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<uint4> h(8192);
    srand(1);
    for (auto &v : h) {
        // random finite bf16 / f16 / int8 patterns: exponents kept in a narrow band so that nothing overflows
        auto r16 = [&]() -> uint32_t { return zeros ? 0u : (uint32_t)((rand() & 0x807f) | 0x3c00 | (rand() & 0x0380)); };
        v.x = r16() | (r16() << 16); v.y = r16() | (r16() << 16); v.z = r16() | (r16() << 16); v.w = r16() | (r16() << 16);
    }
    uint4 *seed;
    float *sink;
    hipMalloc(&seed, h.size() * sizeof(uint4));
    hipMalloc(&sink, (size_t)cus * 512 * sizeof(float));
    hipMemcpy(seed, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    printf("%d CUs, operands %s\n", cus, zeros ? "all zero" : "random");
    const double secs = 1.5;
    run<0>("bf16 32x32x16 (4 chains)", 4.0 * 2 * 32 * 32 * 16, seed, sink, cus, secs);
    run<1>("bf16 16x16x32 (8 chains)", 8.0 * 2 * 16 * 16 * 32, seed, sink, cus, secs);
    run<2>("f16  32x32x16 (4 chains)", 4.0 * 2 * 32 * 32 * 16, seed, sink, cus, secs);
    run<3>("f16  16x16x32 (8 chains)", 8.0 * 2 * 16 * 16 * 32, seed, sink, cus, secs);
    run<4>("i8   32x32x32 (4 chains)", 4.0 * 2 * 32 * 32 * 32, seed, sink, cus, secs);
    run<5>("i8   16x16x64 (8 chains)", 8.0 * 2 * 16 * 16 * 64, seed, sink, cus, secs);
    return 0;
}
