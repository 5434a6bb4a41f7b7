// encoder_tail2.hip -- an EXPERIMENTAL second form of the layer tail (MEMEX_HIP_TAIL=2|3; default: encoder_tail.hip), hidden = 384:
//     x1  = LayerNorm1(x + Wo ctx + bo);   out = LayerNorm2(x1 + W2 gelu(W1 x1 + b1) + b2)
// Same role as encoder_tail.hip (reference: the BERT self-output / intermediate / output blocks behind
// `model.encode(&segments)`, lib/libmemex/src/llm/embedding.rs:109; restated in oracle/bert_oracle.py), built
// the other way round.  What bounded tail_kernel (round 2: 0.315 of the MFMA peak) is the CU's vector-memory
// return path: every wave pulled its own copy of the weights through L1 at 1 KiB per 2 MFMAs, i.e. 64 B/clk
// per CU at full MFMA rate -- exactly the L1 peak (profiles/r3_tail_trace.txt: 60 us of a 97 us workgroup
// in the chunk loop at 61 % MFMA issue; LayerNorm phases 27 us behind five barriers).  Here:
//   * ACTIVATION-STATIONARY: a wave owns 32 tokens from the attention output to the layer output.  A lane
//     owns one token (MFMA column), so the 384 features of a token sit in the lane pair (l, l+32): both
//     LayerNorms are register arithmetic plus one half-wave exchange -- no LDS staging tile, no barrier --
//     and h = gelu(W1 x1 + b1) never leaves the registers: the accumulator of G1 becomes the B operand of
//     G2 through v_cvt_pk_bf16_f32 + two v_permlane32_swap per 16 features.
//   * ONE 512-register wave per SIMD (4 waves = 128 tokens per workgroup, one workgroup per CU): y (32 x 384
//     f32 = 192 registers), the G1 accumulator of a 64-feature ffn chunk (32) and two h fragment sets (2 x 16)
//     stay resident.
//   * THE WEIGHTS ARE THE STREAM: Wo, W1, W2 stored once more as ONE sequence of 1-KiB MFMA A-fragments in
//     consumption order (tail2_stream_layout), pulled by LDS-DMA into a 48-KiB ring (6 slots x 8
//     fragments) that all four waves read: 2.65 MB of weights cross L2 -> CU once per 128 tokens (round 2:
//     once per 64 tokens and per wave pair), the L1 return path carries nothing in the loop, every fragment
//     read is a conflict-free lane-linear ds_read_b128, one s_barrier per 8 MFMAs publishes a slot.
//   * the GELU of chunk c runs under the MFMAs of G2(c-1):
//     a single wave issues ~5 instructions per 32-cycle MFMA, so the epilogue is cut to fit:
//     gelu(v) = v / (1 + exp(-(a v + b v^3 + c v^5))), |error| <= 2.6e-5 against the erf form (tests), 9
//     single-issue VALU operations per value.
// Arithmetic differences to the GEMM-by-GEMM path (all inside the 1e-3 cosine bar, tests/test_encoder_gpu.py):
// the residuals enter in f32 instead of through a bf16 staging tile, x1 feeds LayerNorm2 unrounded, the GELU
// form above.  tail_kernel stays the path for small passes (a query: more, smaller workgroups) and for
// MEMEX_HIP_TAIL=1.
#include <type_traits>

#include "encoder_kernels.h"
#include "encoder_tail2.h"

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#ifndef MX_TAIL2_TRACE
#define MX_TAIL2_TRACE 0
#endif
// Ablation switch for scripts/tail_ubench.hip only (0 = production kernel; results are wrong with any bit set):
//   1 = no slot wait / barrier, 2 = no DMA issue in the loops, 4 = no GELU sub-steps under the MFMAs,
//   8 = no A-fragment reads in the loops (the registers keep the prologue's fragments)
#ifndef MX_TAIL2_ABLATE
#define MX_TAIL2_ABLATE 0
#endif
#if MX_TAIL2_TRACE
#define MX_TRACE2(i)                                                                                        \
    do {                                                                                                    \
        if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#else
#define MX_TRACE2(i) do { } while (0)
#endif

namespace {

constexpr int kHid = 384;
constexpr int kTok = 128;                               // token rows per workgroup
constexpr int kWTok = 32;                               // ... per wave (one MFMA column block)
constexpr int kFC = 64;                                 // ffn features per chunk (2 MFMA blocks: y 192 + h 32 accumulator registers leave the allocator 32 of the 256 AGPRs)
constexpr int kFrag = 1024;                             // bytes of one MFMA fragment (64 lanes x 16 B)
constexpr int kSlotFrags = 8;
constexpr int kRingSlots = 6;
constexpr int kRingFrags = kSlotFrags * kRingSlots;     // 48
constexpr int kActBytes = kWTok * kHid * 2;             // 24576: one wave's activation fragments (ctx, then x1)
constexpr int kRingOff = 4 * kActBytes;                 // 98304
constexpr int kParOff = kRingOff + kRingFrags * kFrag;  // 147456
constexpr int kMaxF = 1536;
constexpr int kParFloats = 6 * kHid + kMaxF;            // bo g1 be1 b2 g2 be2 | b1
constexpr int kLds2 = kParOff + kParFloats * 4;         // 162816 of 163840
constexpr int kPoFrags = 12 * 24;                       // Wo: 24 k-steps x 12 feature blocks
constexpr int kG1Frags = 2 * 24;                        // W1 chunk: 24 k-steps x 2 feature blocks
constexpr int kG2Frags = 12 * 4;                        // W2 chunk: 4 k-steps x 12 feature blocks
constexpr int kAhead = 8;                               // A fragments read ahead into registers
static_assert(kPoFrags % kRingFrags == 0 && kG1Frags % kRingFrags == 0 && kG2Frags % kRingFrags == 0,
              "every segment starts at ring position 0: all LDS offsets are compile-time constants");
static_assert(kAhead <= kSlotFrags, "the read-ahead may reach into the next slot only");

// parameter block offsets (floats)
constexpr int kPBo = 0, kPG1 = kHid, kPBe1 = 2 * kHid, kPB2 = 3 * kHid, kPG2 = 4 * kHid, kPBe2 = 5 * kHid, kPB1 = 6 * kHid;

// exp(-(a v + b v^3 + c v^5)) as exp2(v (C1 + C3 t + C5 t^2)), t = min(v^2, 100)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kGC1 = -1.59501577f * kLog2e, kGC3 = -7.40112855e-2f * kLog2e, kGC5 = 7.03032486e-4f * kLog2e;

template <int N>
using ic = std::integral_constant<int, N>;

// compile-time loop: every index inside the body is a constant expression (register arrays must never be
// indexed by anything the compiler could mistake for a run-time value: that sends them to scratch)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

}  // namespace

__device__ __forceinline__ float gelu_sig5(float v) {
    const float t = fminf(v * v, 100.0f);
    float pl = __builtin_fmaf(t, kGC5, kGC3);
    pl = __builtin_fmaf(pl, t, kGC1);
    const float e = __builtin_amdgcn_exp2f(pl * v);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// accumulator layout (lane = token, 8 consecutive registers = features {0-3, 8-11} + 4 (lane >> 5) of a
// 16-feature group)  <->  MFMA B-fragment layout (lane holds features 8 (lane >> 5) .. +7 as 4 packed pairs):
// the same two half-wave swaps in both directions
// v_permlane32_swap a, b: lanes 32-63 of a <-> lanes 0-31 of b.  Written as inline assembly: hipcc (ROCm 7.2)
// miscompiles code that combines the two results of __builtin_amdgcn_permlane32_swap (seen: r[0] + r[1] emitted as
// r[0] + r[0]).  The instruction needs two wait states after a VALU write of either operand; the s_nop covers it.
__device__ __forceinline__ void lane32_swap(uint32_t &a, uint32_t &b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void frag_swap(uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t &r3) {
    lane32_swap(r0, r2);
    lane32_swap(r1, r3);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const bf16x2 pk = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, pk);
}
// 8 accumulator values of one 16-feature group -> this lane's B fragment of that group
__device__ __forceinline__ bf16x8 to_frag(const float (&v)[8]) {
    uint32_t r0 = pack2(v[0], v[1]), r1 = pack2(v[2], v[3]), r2 = pack2(v[4], v[5]), r3 = pack2(v[6], v[7]);
    frag_swap(r0, r1, r2, r3);
    const u32x4 o = {r0, r1, r2, r3};
    return __builtin_bit_cast(bf16x8, o);
}
__device__ __forceinline__ void from_frag(bf16x8 f, float (&v)[8]) {
    const u32x4 u = __builtin_bit_cast(u32x4, f);
    uint32_t r0 = u[0], r1 = u[1], r2 = u[2], r3 = u[3];
    frag_swap(r0, r1, r2, r3);
    const uint32_t r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
        v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
    }
}

// ---- the accumulators live in AGPRs that the inline assembly below owns (hipcc cannot keep 256 accumulator
// registers of a 512-register wave in place: left to the allocator the kernel spills ~700 registers and every
// reload drains the DMA ring).  a[16 fb .. +15], fb < 12: y block fb;  a[192 + 32 P + 16 fb .. +15]: G1
// accumulator block fb < 2 of chunk parity P.  The compiler only ever sees VGPRs (fragments, GELU temporaries).
// Audit after every edit (scripts/check_tail2_isa.sh): .vgpr_spill_count 0, no scratch, and no v_accvgpr_* /
// v_mfma outside ;;#ASMSTART .. ;;#ASMEND.
#define MX_A16(n) "a" #n
#define MX_CLOB_1(b) "a" #b
#define MX_STR(x) #x
#define MX_MFMA(BASE, A, B)                                                                                  \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(A), "v"(B), "i"(BASE), "i"((BASE) + 15))
#define MX_MFMA_VB(BASE, A, B) /* B was just written by VALU code: two wait states inside the statement */   \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(A), "v"(B), "i"(BASE), "i"((BASE) + 15))
#define MX_MFMA_Z(BASE, A, B) /* C = 0 */                                                                    \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(A), "v"(B), "i"(BASE), "i"((BASE) + 15))
#define MX_ACC_RD(DST, IDX) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(DST) : "i"(IDX))
#define MX_ACC_WR(IDX, SRC) asm volatile("v_accvgpr_write_b32 a%c0, %1" ::"i"(IDX), "v"(SRC))
#define MX_ACC_RD4(D, IDX) /* D[0..3] = a[IDX .. IDX+3]: one statement, one boundary pad */                              \
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7" \
                 : "=v"((D)[0]), "=v"((D)[1]), "=v"((D)[2]), "=v"((D)[3])                                                        \
                 : "i"(IDX), "i"((IDX) + 1), "i"((IDX) + 2), "i"((IDX) + 3))
#define MX_ACC_WR4(IDX, S)                                                                                                   \
    asm volatile("v_accvgpr_write_b32 a%c0, %4\n\tv_accvgpr_write_b32 a%c1, %5\n\tv_accvgpr_write_b32 a%c2, %6\n\tv_accvgpr_write_b32 a%c3, %7" \
                 ::"i"(IDX), "i"((IDX) + 1), "i"((IDX) + 2), "i"((IDX) + 3), "v"((S)[0]), "v"((S)[1]), "v"((S)[2]), "v"((S)[3]))
#define MX_MFMA_W(BASE, A, B, ANEXT) /* also waits for the NEXT A fragment: one s_waitcnt per two MFMAs */             \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(A), "v"(B), "i"(BASE), "i"((BASE) + 15), "v"(ANEXT))
#define MX_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")  /* the last MFMAs' results are readable */

#define MX_Z1(n) "v_accvgpr_write_b32 a" #n ", 0\n\t"
#define MX_Z8(a, b, c, d, e, f, g, h) MX_Z1(a) MX_Z1(b) MX_Z1(c) MX_Z1(d) MX_Z1(e) MX_Z1(f) MX_Z1(g) MX_Z1(h)
#define MX_C8(a, b, c, d, e, f, g, h) "a" #a, "a" #b, "a" #c, "a" #d, "a" #e, "a" #f, "a" #g, "a" #h

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tail2_kernel(const Tail2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const size_t row = (size_t)blockIdx.x * kTok + w * kWTok + l31;  // this lane's token
    const int nch = p.f / kFC;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t act = (uint32_t)w * kActBytes + lane16;   // + s * 1024: this lane's 16 bytes of activation fragment s
    const uint32_t ringl = (uint32_t)kRingOff + lane16;      // + pos * 1024: ... of the fragment at ring position pos
    const float *par = reinterpret_cast<const float *>(smem + kParOff);
    MX_TRACE2(0);

    // y = 0, and the declaration that a0 .. a255 belong to this kernel (the clobber list is what makes the
    // kernel descriptor allocate them)
    asm volatile(
        MX_Z8(0, 1, 2, 3, 4, 5, 6, 7) MX_Z8(8, 9, 10, 11, 12, 13, 14, 15) MX_Z8(16, 17, 18, 19, 20, 21, 22, 23) MX_Z8(24, 25, 26, 27, 28, 29, 30, 31)
        MX_Z8(32, 33, 34, 35, 36, 37, 38, 39) MX_Z8(40, 41, 42, 43, 44, 45, 46, 47) MX_Z8(48, 49, 50, 51, 52, 53, 54, 55) MX_Z8(56, 57, 58, 59, 60, 61, 62, 63)
        MX_Z8(64, 65, 66, 67, 68, 69, 70, 71) MX_Z8(72, 73, 74, 75, 76, 77, 78, 79) MX_Z8(80, 81, 82, 83, 84, 85, 86, 87) MX_Z8(88, 89, 90, 91, 92, 93, 94, 95)
        MX_Z8(96, 97, 98, 99, 100, 101, 102, 103) MX_Z8(104, 105, 106, 107, 108, 109, 110, 111) MX_Z8(112, 113, 114, 115, 116, 117, 118, 119) MX_Z8(120, 121, 122, 123, 124, 125, 126, 127)
        MX_Z8(128, 129, 130, 131, 132, 133, 134, 135) MX_Z8(136, 137, 138, 139, 140, 141, 142, 143) MX_Z8(144, 145, 146, 147, 148, 149, 150, 151) MX_Z8(152, 153, 154, 155, 156, 157, 158, 159)
        MX_Z8(160, 161, 162, 163, 164, 165, 166, 167) MX_Z8(168, 169, 170, 171, 172, 173, 174, 175) MX_Z8(176, 177, 178, 179, 180, 181, 182, 183) MX_Z8(184, 185, 186, 187, 188, 189, 190, 191)
        "s_nop 3"
        :
        :
        : "memory", MX_C8(0, 1, 2, 3, 4, 5, 6, 7), MX_C8(8, 9, 10, 11, 12, 13, 14, 15), MX_C8(16, 17, 18, 19, 20, 21, 22, 23), MX_C8(24, 25, 26, 27, 28, 29, 30, 31),
          MX_C8(32, 33, 34, 35, 36, 37, 38, 39), MX_C8(40, 41, 42, 43, 44, 45, 46, 47), MX_C8(48, 49, 50, 51, 52, 53, 54, 55), MX_C8(56, 57, 58, 59, 60, 61, 62, 63),
          MX_C8(64, 65, 66, 67, 68, 69, 70, 71), MX_C8(72, 73, 74, 75, 76, 77, 78, 79), MX_C8(80, 81, 82, 83, 84, 85, 86, 87), MX_C8(88, 89, 90, 91, 92, 93, 94, 95),
          MX_C8(96, 97, 98, 99, 100, 101, 102, 103), MX_C8(104, 105, 106, 107, 108, 109, 110, 111), MX_C8(112, 113, 114, 115, 116, 117, 118, 119), MX_C8(120, 121, 122, 123, 124, 125, 126, 127),
          MX_C8(128, 129, 130, 131, 132, 133, 134, 135), MX_C8(136, 137, 138, 139, 140, 141, 142, 143), MX_C8(144, 145, 146, 147, 148, 149, 150, 151), MX_C8(152, 153, 154, 155, 156, 157, 158, 159),
          MX_C8(160, 161, 162, 163, 164, 165, 166, 167), MX_C8(168, 169, 170, 171, 172, 173, 174, 175), MX_C8(176, 177, 178, 179, 180, 181, 182, 183), MX_C8(184, 185, 186, 187, 188, 189, 190, 191),
          MX_C8(192, 193, 194, 195, 196, 197, 198, 199), MX_C8(200, 201, 202, 203, 204, 205, 206, 207), MX_C8(208, 209, 210, 211, 212, 213, 214, 215), MX_C8(216, 217, 218, 219, 220, 221, 222, 223),
          MX_C8(224, 225, 226, 227, 228, 229, 230, 231), MX_C8(232, 233, 234, 235, 236, 237, 238, 239), MX_C8(240, 241, 242, 243, 244, 245, 246, 247), MX_C8(248, 249, 250, 251, 252, 253, 254, 255));

    // ---- the weight stream: fragment g at byte g * 1024 of p.wf2; a slot = 8 fragments, wave w moves two of
    // them (LDS-DMA, 1 KiB per instruction).  The stream position sits in the VGPR offset so that the buffer
    // range check kills the loads past the end of the stream (they touch no memory): the loop issues
    // unconditionally and the vmcnt arithmetic stays uniform.
    const uint32_t nfrag = (uint32_t)(kPoFrags + nch * (kG1Frags + kG2Frags));
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)p.wf2, 0, nfrag * (uint32_t)kFrag, 0x00020000);
    uint32_t dma_voff = lane16 + (uint32_t)(2 * w) * kFrag;  // this lane's bytes of the wave's first fragment of the next slot
    const uint32_t dma_dst = (uint32_t)kRingOff + (uint32_t)(2 * w) * kFrag;
    auto dma_slot = [&](auto postag) __attribute__((always_inline)) {  // ring slot 0..5
        constexpr int pos = decltype(postag)::value;
        char *dst = smem + (dma_dst + (uint32_t)pos * (kSlotFrags * kFrag));  // wave-uniform (w went through readfirstlane)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t *)dst, 16, dma_voff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t *)(dst + kFrag), 16, dma_voff + kFrag, 0, 0, 0);
        dma_voff += kSlotFrags * kFrag;
    };

    // ---- prologue: residual rows (registers, fragment layout), parameter block and ctx fragments (LDS-DMA),
    // then the first five slots of the stream
    bf16x8 xr[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) xr[s] = *reinterpret_cast<const bf16x8 *>(p.x + row * p.ldx + 16 * s + 8 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = w + 4 * i;  // 15 pieces of 1 KiB
        if (piece * 256 < kParFloats)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(p.pf + piece * 256 + lane * 4),
                                             (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane(kParOff + piece * kFrag)), 16, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 24; ++s)
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(p.ctx + row * p.ldc + 16 * s + 8 * h),
                                         (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane((uint32_t)w * kActBytes + s * kFrag)), 16, 0, 0);
    {   // touch the residual rows here: the compiler's wait for them (it drains everything issued so far) then
        // falls before the ring starts, not into the out-projection
        float sink = 0.0f;
#pragma unroll
        for (int s = 0; s < 24; ++s) sink += (float)xr[s][0];
        asm volatile("" ::"v"(sink));
    }
    static_for<0, kRingSlots - 1>([&](auto j) __attribute__((always_inline)) { dma_slot(j); });
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // slots 0 and 1 landed (three newer slots in flight)
    __builtin_amdgcn_s_barrier();
    MX_TRACE2(1);

    // ---- fragment pipeline.  abuf[f % 8] holds A fragment f of the current segment; after the MFMA that
    // consumed it the register is refilled with fragment f + 8 (ring position (f + 8) % 48).
    bf16x8 abuf[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) abuf[i] = *reinterpret_cast<const bf16x8 *>(smem + ringl + i * kFrag);
    auto ring_read = [&](auto ftag) __attribute__((always_inline)) -> bf16x8 {  // fragment f of the segment (f may run into the next segment)
        return *reinterpret_cast<const bf16x8 *>(smem + ringl + (decltype(ftag)::value % kRingFrags) * kFrag);
    };
    // start of a slot (f % 8 == 0): my pieces of the NEXT slot have landed, everyone's are published by the
    // barrier, and everyone is done with the previous slot, whose ring position the DMA below refills
    auto slot_open = [&]() __attribute__((always_inline)) {
#if !(MX_TAIL2_ABLATE & 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
    };
    // the bookkeeping behind MFMA f of a segment: refill the A register, issue the DMA of the slot five ahead
    auto after_mfma = [&](auto ft) __attribute__((always_inline)) {
        constexpr int f = decltype(ft)::value;
#if !(MX_TAIL2_ABLATE & 8)
        abuf[f % kAhead] = ring_read(ic<f + kAhead>{});
#endif
        if constexpr (f % kSlotFrags == 1 && !(MX_TAIL2_ABLATE & 2)) dma_slot(ic<((f % kRingFrags) / kSlotFrags + kRingSlots - 1) % kRingSlots>{});
    };

    // =============== out-projection: y = ctx Wo^T.  6 ring revolutions of 4 k-steps x 12 blocks ===============
    {
        uint32_t actp = act;
        bf16x8 bb[2];
        bb[0] = *reinterpret_cast<const bf16x8 *>(smem + actp);
#pragma unroll 1
        for (int rev = 0; rev < kPoFrags / kRingFrags; ++rev) {
            static_for<0, kRingFrags>([&](auto ft) __attribute__((always_inline)) {
                constexpr int f = decltype(ft)::value, ks = f / 12, fb = f % 12;
                // the next k-step's ctx fragment (the read past the last k-step stays inside LDS: harmless)
                if constexpr (fb == 0) bb[(ks + 1) & 1] = *reinterpret_cast<const bf16x8 *>(smem + actp + (ks + 1) * kFrag);
                if constexpr (f % kSlotFrags == 0) {
                    if (f != 0 || rev != 0) slot_open();
                }
                if constexpr (f % 2 == 0) MX_MFMA_W(16 * fb, abuf[f % kAhead], bb[ks & 1], abuf[(f + 1) % kAhead]);
                else MX_MFMA(16 * fb, abuf[f % kAhead], bb[ks & 1]);
                after_mfma(ft);
                __builtin_amdgcn_sched_barrier(0);
            });
            actp += 4 * kFrag;
        }
    }
    MX_TRACE2(2);

    // =============== Add & LayerNorm in registers ===============
    // v = y (+ residual fragments + bias); mean / variance over the token's 384 features = this lane's 192
    // values + the partner lane's (one half-wave exchange; single pass, f32: sum and sum of squares);
    // (v - mean) rstd gamma + beta comes out 8 values (one B fragment) at a time
    auto par4 = [&](int off, int fb, int rg) __attribute__((always_inline)) -> f32x4 {  // 4 parameters of features fb*32 + 8 rg + 4 h ..
        return *reinterpret_cast<const f32x4 *>(par + off + fb * 32 + rg * 8 + h * 4);
    };
    auto half_sum = [&](float s) __attribute__((always_inline)) -> float {  // s(lane) + s(lane ^ 32) in every lane
        uint32_t a = __builtin_bit_cast(uint32_t, s), b = a;
        lane32_swap(a, b);  // lower lanes: a = own, b = partner's; upper lanes: a = partner's, b = own
        return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    };
    float ln_mean = 0.0f, ln_rstd = 0.0f;
    auto ln_stats = [&](auto first) __attribute__((always_inline)) {  // first: LayerNorm1 (y += residual rows + bo)
        constexpr bool kFirst = decltype(first)::value != 0;
        float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        MX_MFMA_DRAIN();
        static_for<0, 24>([&](auto gt) __attribute__((always_inline)) {
            constexpr int g = decltype(gt)::value, fb = g >> 1, hf = g & 1;
            float v[8];  // 8 registers per statement: the eight chains behind it are independent, the compiler interleaves them
            MX_ACC_RD4(v, 16 * fb + 8 * hf);
            MX_ACC_RD4(v + 4, 16 * fb + 8 * hf + 4);
            if constexpr (kFirst) {
                float rv[8];
                from_frag(xr[g], rv);
                const f32x4 b0 = par4(kPBo, fb, 2 * hf), b1v = par4(kPBo, fb, 2 * hf + 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] + rv[i] + (i < 4 ? b0[i & 3] : b1v[i & 3]);
                MX_ACC_WR4(16 * fb + 8 * hf, v);
                MX_ACC_WR4(16 * fb + 8 * hf + 4, v + 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s4[i & 3] += v[i];
                q4[i & 3] = __builtin_fmaf(v[i], v[i], q4[i & 3]);
            }
        });
        ln_mean = half_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / kHid);
        const float ex2 = half_sum((q4[0] + q4[1]) + (q4[2] + q4[3])) * (1.0f / kHid);
        ln_rstd = 1.0f / sqrtf(fmaxf(ex2 - ln_mean * ln_mean, 0.0f) + p.eps);
    };
    // normalised values of fragment g (block g >> 1, half g & 1) -> v[8]
    auto ln_values = [&](auto gt, int pg, int pbe, float (&v)[8]) __attribute__((always_inline)) {
        constexpr int g = decltype(gt)::value, fb = g >> 1, hf = g & 1;
        const f32x4 g0 = par4(pg, fb, 2 * hf), g1 = par4(pg, fb, 2 * hf + 1), e0 = par4(pbe, fb, 2 * hf), e1 = par4(pbe, fb, 2 * hf + 1);
        float a[8];
        MX_ACC_RD4(a, 16 * fb + 8 * hf);
        MX_ACC_RD4(a + 4, 16 * fb + 8 * hf + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf((a[i] - ln_mean) * ln_rstd, i < 4 ? g0[i & 3] : g1[i & 3], i < 4 ? e0[i & 3] : e1[i & 3]);
    };

    // x1 = LN1(x + ctx Wo^T + bo) -> this wave's activation fragments (over ctx: only this wave reads them, and
    // its reads are done); y = x1 + b2: the residual and the bias of the MLP output, in f32
    ln_stats(ic<1>{});
    static_for<0, 24>([&](auto gt) __attribute__((always_inline)) {
        constexpr int g = decltype(gt)::value, fb = g >> 1, hf = g & 1;
        float v[8];
        ln_values(gt, kPG1, kPBe1, v);
        *reinterpret_cast<bf16x8 *>(smem + act + g * kFrag) = to_frag(v);
        const f32x4 c0 = par4(kPB2, fb, 2 * hf), c1 = par4(kPB2, fb, 2 * hf + 1);
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = v[i] + (i < 4 ? c0[i & 3] : c1[i & 3]);
        MX_ACC_WR4(16 * fb + 8 * hf, y);
        MX_ACC_WR4(16 * fb + 8 * hf + 4, y + 4);
    });
    asm volatile("s_nop 3" ::: "memory");
    MX_TRACE2(3);

    // =============== MLP.  stream: G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(nch-1) G2(nch-2) | G2(nch-1) ===============
    bf16x8 hfr[2][4];   // gelu(h) as B fragments of G2, by chunk parity: 4 k-steps of 16 ffn features

    // The epilogue E1(c), one PIECE = one G2 k-step of 16 ffn features: hfr[P][k] = B fragment of
    // gelu(acc1[P] + b1) for features 16 k .. 16 k + 15 of chunk c (P = c & 1).  A single wave hides ~5
    // instructions behind an MFMA, so a piece is cut into kE1Steps sub-steps that the segments below spread
    // over 24 MFMAs.
    // Sub-steps of a piece: 24, one behind each MFMA it is spread over.  A step applies ONE operation to FOUR
    // values (a single wave issues in order: a dependent chain per value would pay the full VALU / transcendental
    // latency at every instruction; four independent chains issue back to back):
    //   steps 0 .. 10: values 0-3 (read | + bias | v^2 | clamp | fma | fma | * v | exp2 | + 1 | rcp | * v), 11 .. 21: values
    //   4-7, 22: pack to bf16, 23: half-wave swaps -> hfr
    constexpr int kE1Steps = 24;
    float ev[8], ea[4], et[4];
    f32x4 eba = {0, 0, 0, 0}, ebc = {0, 0, 0, 0};
    uint32_t ep[4] = {0, 0, 0, 0};
    auto e1_step = [&](auto ptag, auto ktag, auto sttag, int chunk) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value, k = decltype(ktag)::value, step = decltype(sttag)::value;
        constexpr int fb = k >> 1, half = k & 1;
        if constexpr (step < 22) {
            constexpr int grp = step / 11, op = step % 11;  // values 4 grp .. 4 grp + 3
            if constexpr (op == 0) {
                if constexpr (grp == 0) {
                    eba = *reinterpret_cast<const f32x4 *>(par + kPB1 + chunk * kFC + fb * 32 + half * 16 + h * 4);
                    ebc = *reinterpret_cast<const f32x4 *>(par + kPB1 + chunk * kFC + fb * 32 + half * 16 + 8 + h * 4);
                }
                MX_ACC_RD4(ea, 192 + 32 * P + 16 * fb + 8 * half + 4 * grp);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (op == 1) ea[j] += grp == 0 ? eba[j] : ebc[j];
                    else if constexpr (op == 2) et[j] = ea[j] * ea[j];
                    else if constexpr (op == 3) et[j] = fminf(et[j], 100.0f);
                    else if constexpr (op == 4) ev[4 * grp + j] = __builtin_fmaf(et[j], kGC5, kGC3);
                    else if constexpr (op == 5) et[j] = __builtin_fmaf(ev[4 * grp + j], et[j], kGC1);
                    else if constexpr (op == 6) et[j] = et[j] * ea[j];
                    else if constexpr (op == 7) et[j] = __builtin_amdgcn_exp2f(et[j]);
                    else if constexpr (op == 8) et[j] = 1.0f + et[j];
                    else if constexpr (op == 9) et[j] = __builtin_amdgcn_rcpf(et[j]);
                    else ev[4 * grp + j] = ea[j] * et[j];
                }
            }
        } else if constexpr (step == 22) {
            ep[0] = pack2(ev[0], ev[1]); ep[1] = pack2(ev[2], ev[3]); ep[2] = pack2(ev[4], ev[5]); ep[3] = pack2(ev[6], ev[7]);
        } else {
            frag_swap(ep[0], ep[1], ep[2], ep[3]);
            const u32x4 o = {ep[0], ep[1], ep[2], ep[3]};
            hfr[P][k] = __builtin_bit_cast(bf16x8, o);
        }
    };
    // the sub-step of piece k that goes behind MFMA g (0..23) of the 24 it is spread over
    auto e1_under = [&](auto ptag, auto ktag, auto gtag, int chunk) __attribute__((always_inline)) {
        if constexpr (!(MX_TAIL2_ABLATE & 4)) e1_step(ptag, ktag, gtag, chunk);
    };
    auto e1_alone = [&](auto ptag, auto k0tag, int chunk) __attribute__((always_inline)) {  // two pieces with nothing to hide under
        MX_MFMA_DRAIN();
        static_for<0, 2>([&](auto kt) __attribute__((always_inline)) {
            static_for<0, kE1Steps>([&](auto st) __attribute__((always_inline)) { e1_step(ptag, ic<decltype(k0tag)::value + decltype(kt)::value>{}, st, chunk); });
        });
    };

    // G1(c): acc1[P] = x1 W1[chunk]^T, 24 k-steps x 2 blocks; with E1, pieces 2 and 3 of E1(c-1) (parity 1-P) run under it
    auto g1_segment = [&](auto ptag, auto e1tag, int chunk) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        constexpr bool E1 = decltype(e1tag)::value != 0;
        bf16x8 bb[3];
        static_for<0, 3>([&](auto st) __attribute__((always_inline)) {
            bb[decltype(st)::value] = *reinterpret_cast<const bf16x8 *>(smem + act + decltype(st)::value * kFrag);
        });
        static_for<0, kG1Frags>([&](auto ft) __attribute__((always_inline)) {
            constexpr int f = decltype(ft)::value, s = f / 2, fb = f % 2;
            if constexpr (f % kSlotFrags == 0) slot_open();
            if constexpr (s == 0) MX_MFMA_Z(192 + 32 * P + 16 * fb, abuf[f % kAhead], bb[s % 3]);
            else if constexpr (f % 2 == 0) MX_MFMA_W(192 + 32 * P + 16 * fb, abuf[f % kAhead], bb[s % 3], abuf[(f + 1) % kAhead]);
            else MX_MFMA(192 + 32 * P + 16 * fb, abuf[f % kAhead], bb[s % 3]);
            after_mfma(ft);
            if constexpr (fb == 1 && s + 3 < 24) bb[s % 3] = *reinterpret_cast<const bf16x8 *>(smem + act + (s + 3) * kFrag);
            if constexpr (E1) e1_under(ic<1 - P>{}, ic<2 + f / 24>{}, ic<f % 24>{}, chunk - 1);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // G2(c): y += gelu(h)(c) W2[:, chunk]^T, 4 k-steps x 12 blocks; with E1, pieces 0 and 1 of E1(c+1) (parity 1-P) run under it
    auto g2_segment = [&](auto ptag, auto e1tag, int chunk) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        constexpr bool E1 = decltype(e1tag)::value != 0;
        static_for<0, kG2Frags>([&](auto ft) __attribute__((always_inline)) {
            constexpr int f = decltype(ft)::value, s2 = f / 12, fb = f % 12;
            if constexpr (f % kSlotFrags == 0) slot_open();
            if constexpr (fb == 0) MX_MFMA_VB(16 * fb, abuf[f % kAhead], hfr[P][s2]);
            else if constexpr (f % 2 == 0) MX_MFMA_W(16 * fb, abuf[f % kAhead], hfr[P][s2], abuf[(f + 1) % kAhead]);
            else MX_MFMA(16 * fb, abuf[f % kAhead], hfr[P][s2]);
            after_mfma(ft);
            if constexpr (E1) e1_under(ic<1 - P>{}, ic<f / 24>{}, ic<f % 24>{}, chunk + 1);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    g1_segment(ic<0>{}, ic<0>{}, 0);   // G1(0)
    e1_alone(ic<0>{}, ic<0>{}, 0);     // E1(0), pieces 0 and 1: nothing to hide under
#pragma unroll 1
    for (int c = 1; c < nch; ++c) {    // G1(c) with the rest of E1(c-1) under it, then G2(c-1) with the start of E1(c) under it
        if (c & 1) {
            g1_segment(ic<1>{}, ic<1>{}, c);
            g2_segment(ic<0>{}, ic<1>{}, c - 1);
        } else {
            g1_segment(ic<0>{}, ic<1>{}, c);
            g2_segment(ic<1>{}, ic<1>{}, c - 1);
        }
    }
    if ((nch - 1) & 1) {
        e1_alone(ic<1>{}, ic<2>{}, nch - 1);
        g2_segment(ic<1>{}, ic<0>{}, nch - 1);
    } else {
        e1_alone(ic<0>{}, ic<2>{}, nch - 1);
        g2_segment(ic<0>{}, ic<0>{}, nch - 1);
    }
    MX_TRACE2(4);

    // =============== out = LayerNorm2(x1 + b2 + gelu(h) W2^T) -> global, 16 bytes per lane and fragment ===============
    ln_stats(ic<0>{});
    static_for<0, 24>([&](auto gt) __attribute__((always_inline)) {
        constexpr int g = decltype(gt)::value;
        float v[8];
        ln_values(gt, kPG2, kPBe2, v);
        *reinterpret_cast<bf16x8 *>(p.out + row * p.ldo + 16 * g + 8 * h) = to_frag(v);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead DMA ops must not outlive the workgroup's LDS
    MX_TRACE2(5);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The stream: fragment (rows r0 .. r0+31 of a weight matrix, k0 .. k0+15) = 64 lanes x 8 bf16, lane (h, r) holding
// row r0 + r, k0 + 8 h .. +7 -- the A operand of v_mfma_f32_32x32x16_bf16 -- in the order
//   Wo:    for s < 24: for fb < 12: (rows 32 fb,            k 16 s)
//   G1(c): for s < 24: for fb < 2:  (W1 rows 64 c + 32 fb,  k 16 s)
//   G2(c): for s < 4:  for fb < 12: (W2 rows 32 fb,         k 64 c + 16 s)
//   Wo | G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(nch-1) G2(nch-2) | G2(nch-1)
size_t tail2_stream_elems(int F) { return (size_t)(kPoFrags + (F / kFC) * (kG1Frags + kG2Frags)) * (kFrag / 2); }

void tail2_stream_layout(const float *wo, const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float)) {
    const int nch = F / kFC;
    size_t o = 0;
    auto frag = [&](const float *wm, size_t ld, int row0, int k0) {
        for (int ln = 0; ln < 64; ++ln)
            for (int e = 0; e < 8; ++e) out[o++] = to_bf16(wm[(size_t)(row0 + (ln & 31)) * ld + k0 + 8 * (ln >> 5) + e]);
    };
    auto g1 = [&](int c) {
        for (int s = 0; s < 24; ++s)
            for (int fb = 0; fb < kFC / 32; ++fb) frag(w1, kHid, c * kFC + fb * 32, 16 * s);
    };
    auto g2 = [&](int c) {
        for (int s = 0; s < kFC / 16; ++s)
            for (int fb = 0; fb < 12; ++fb) frag(w2, (size_t)F, fb * 32, c * kFC + 16 * s);
    };
    for (int s = 0; s < 24; ++s)
        for (int fb = 0; fb < 12; ++fb) frag(wo, kHid, fb * 32, 16 * s);
    g1(0);
    for (int c = 1; c < nch; ++c) {
        g1(c);
        g2(c - 1);
    }
    g2(nch - 1);
}

// the parameter block the kernel copies into LDS: bo g1 be1 b2 g2 be2 (384 each) | b1 (F, zero-padded to 1536)
size_t tail2_param_floats() { return kParFloats; }
void tail2_param_layout(const float *bo, const float *g1, const float *be1, const float *b1, const float *b2, const float *g2,
                        const float *be2, int F, float *out) {
    for (int i = 0; i < kParFloats; ++i) out[i] = 0.0f;
    for (int i = 0; i < kHid; ++i) {
        out[kPBo + i] = bo[i]; out[kPG1 + i] = g1[i]; out[kPBe1 + i] = be1[i];
        out[kPB2 + i] = b2[i]; out[kPG2 + i] = g2[i]; out[kPBe2 + i] = be2[i];
    }
    for (int i = 0; i < F; ++i) out[kPB1 + i] = b1[i];
}

hipError_t tail2_setup() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&tail2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds2);
}

bool tail2_supported(int hidden, int ffn) { return hidden == kHid && ffn >= 128 && ffn % 128 == 0 && ffn <= kMaxF; }

hipError_t launch_tail2(hipStream_t s, const Tail2Params &p) {
    if (p.m % kTok || p.f % 128 || p.f < 128 || p.f > kMaxF || !p.wf2 || !p.pf || !p.ctx) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tail2_kernel, dim3(p.m / kTok), dim3(256), kLds2, s, p);
    return hipGetLastError();
}

}  // namespace mx
