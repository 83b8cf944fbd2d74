// fp32 contractions on the 16-bit matrix cores from split operands ("split formats").
//
// The contractions of the path are compute-bound on the fp32 matrix rate (64 FLOP/clk/SIMD), 1/16 of the bf16 / fp16 one.
// An fp32 operand is written as a short sum of 16-bit terms whose pairwise products are exact in the fp32 accumulator:
//
//   terms = 3  (bf16 x 3)   x = hi + mid + lo, 8 + 8 + 8 significant bits, fp32's exponent range.  Six MFMAs per block:
//                           a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1   (dropped products <= 2^-24 |a b|)
//   terms = 2  (fp16 x 2)   x = hi + lo, 11 + 11 significant bits.  THREE MFMAs per block:
//                           a1 b1 + a1 b2 + a2 b1                           (dropped product  <= 2^-22 |a b|)
//              fp16 keeps 5 exponent bits: (1) a finite |x| >= 65520 becomes Inf; (2) the low term of |x| < 2^-3 is an fp16
//              subnormal (the matrix cores keep subnormal operands: tools/ubench/mfma_f16_probe.hip), so x is carried to
//              max(2^-24 |x|, 2^-25) -- absolute below 2^-3, relative above.  Weights are therefore packed times a power of two
//              that puts the layer's largest magnitude into [2^14, 2^15) (every weight down to 2^-17 of the largest keeps 22
//              bits; the accumulator is multiplied by the inverse, exact), activations go in as they are.
//   Measured on the chip against float64 (K = 576, post-ReLU operands, same probe): fp32 MFMA 4.4e-7 of max|ref|, three bf16
//   terms 6.6e-7, two fp16 terms 4.2e-7 -- the error of all three is the fp32 accumulation, not the operands.
//
// Layouts (terms = NT_):  activations [N][ceil(C/16)][H][W][terms][16 channels], 32 * terms bytes per pixel and chunk ("SB16");
// in LDS at a pitch of 32 * terms + 16 bytes (an odd number of 16-byte slots: sixteen lanes' 16-byte reads fall into sixteen
// different bank groups).  Weights in A-fragment order [..][term][64 lanes][8].
#pragma once
#include <hip/hip_runtime.h>

namespace bde {

typedef __bf16 sb_b8 __attribute__((ext_vector_type(8)));
typedef _Float16 sb_h8 __attribute__((ext_vector_type(8)));
typedef int sb8 __attribute__((ext_vector_type(4)));          // one MFMA operand fragment of either format: 8 x 16 bit
typedef float sb_f4 __attribute__((ext_vector_type(4)));
typedef float sb_f16v __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int sb_pix_bytes(int terms) { return 32 * terms; }
__host__ __device__ constexpr int sb_lds_slots(int terms) { return 2 * terms + 1; }      // 16-byte slots per pixel in LDS
__host__ __device__ constexpr int sb_lds_pitch(int terms) { return 16 * (2 * terms + 1); }

// ---- bf16 terms ------------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned short sb_bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);     // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ __forceinline__ float sb_bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// x = hi + mid + lo (exact for every finite fp32 x whose low term does not underflow; Inf / NaN ride in the leading term alone:
// Inf - Inf would put a NaN into the second term of an infinite value)
__host__ __device__ __forceinline__ void sb_split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = sb_bf16_rne(x);
    if ((__builtin_bit_cast(unsigned, x) & 0x7f800000u) == 0x7f800000u) { mid = lo = 0; return; }
    const float r1 = x - sb_bf16_to_f32(hi);
    mid = sb_bf16_rne(r1);
    const float r2 = r1 - sb_bf16_to_f32(mid);
    lo = sb_bf16_rne(r2);
}

// ---- fp16 terms ------------------------------------------------------------------------------------------------------------------
// round to nearest even, subnormals kept, |x| >= 65520 -> Inf (bit-identical to v_cvt_f16_f32 in the default kernel mode)
__host__ __device__ __forceinline__ unsigned short sb_f16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    const unsigned short s = (unsigned short)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (unsigned short)(s | 0x7e00u);
    if (u >= 0x47800000u) return (unsigned short)(s | 0x7c00u);                          // >= 2^16
    if (u < 0x38800000u) {                                                              // < 2^-14: multiples of 2^-24
        const float f = __builtin_bit_cast(float, u) + 0.5f;                            // ulp of [0.5, 1) is 2^-24: the add rounds
        return (unsigned short)(s | (__builtin_bit_cast(unsigned, f) - 0x3f000000u));
    }
    u += 0xfffu + ((u >> 13) & 1u);
    return (unsigned short)(s | ((u - 0x38000000u) >> 13));                             // (a carry into exponent 31 is Inf)
}
__host__ __device__ __forceinline__ float sb_f16_to_f32(unsigned short h) {
    const unsigned s = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 31u) return __builtin_bit_cast(float, s | 0x7f800000u | (m << 13));
    if (e == 0u) {
        const float f = (float)m * 5.9604644775390625e-08f;                             // m * 2^-24
        return s ? -f : f;
    }
    return __builtin_bit_cast(float, s | ((e + 112u) << 23) | (m << 13));
}
__host__ __device__ __forceinline__ void sb_split2(float x, unsigned short& hi, unsigned short& lo) {
    hi = sb_f16_rne(x);
    if ((hi & 0x7c00u) == 0x7c00u) { lo = 0; return; }                                  // Inf / NaN / out of range: leading term alone
    lo = sb_f16_rne(x - sb_f16_to_f32(hi));
}
// the power of two that puts max|w| into [2^14, 2^15) (1 for an all-zero or non-finite layer)
static inline float sb_weight_scale(const float* w, long n) {
    float mx = 0.f;
    for (long i = 0; i < n; ++i) {
        const float v = w[i] < 0.f ? -w[i] : w[i];
        if (v == v && v < 3.0e38f && v > mx) mx = v;
    }
    if (!(mx > 0.f)) return 1.f;
    int e = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127;               // mx in [2^e, 2^(e+1))
    int sh = 14 - e;
    if (sh > 100) sh = 100;
    if (sh < -100) sh = -100;
    return __builtin_bit_cast(float, (unsigned)(127 + sh) << 23);
}

// Largest finite fp32 magnitude the two-term format carries: anything at or above it rounds to the fp16 infinity.
constexpr float SB_F16_LIMIT = 65520.f;

#if defined(__HIPCC__)
// ---- range guard of the two-term format -------------------------------------------------------------------------------------------
// Every kernel that turns fp32 activations into two fp16 terms keeps the running maximum of |x| over the values it splits (one
// v_max_f32 / half a v_max3_f32 per value: the |.| is an operand modifier) and, should that maximum reach SB_F16_LIMIT, ORs bit 0
// into the forward's overflow word.  The host reads the word when the forward's frames are handed over (bde_api.hip,
// resolve_overflow): the frames of a forward that set it are recomputed in the three-term bf16 format, which has fp32's
// exponent range ("sb_auto", default), or the call fails with BDE_ERR_RANGE.  A NaN does not raise the flag (max drops it) and
// travels through either format as a NaN; an infinite input does (three bf16 terms carry it exactly as fp32 does).
__device__ __forceinline__ float sb_guard_max(float gm, float x) { return fmaxf(gm, fabsf(x)); }
__device__ __forceinline__ float sb_guard_max2(float gm, float x0, float x1) { return fmaxf(fmaxf(gm, fabsf(x0)), fabsf(x1)); }
__device__ __forceinline__ void sb_guard_flush(float gm, unsigned* flag) {
    if (flag != nullptr && gm >= SB_F16_LIMIT) atomicOr(flag, 1u);
}

// ---- LDS-DMA the waitcnt pass does not see ------------------------------------------------------------------------------------------
// 16 bytes per lane, global -> LDS, block of 1 KiB at the WAVE-UNIFORM LDS address `lds_addr` (lane l lands at lds_addr + 16 l).  As the
// builtin (__builtin_amdgcn_global_load_lds) the instruction makes SIInsertWaitcnts give up counting: the first wait on ANY load
// after it becomes s_waitcnt vmcnt(0) lgkmcnt(0), which drains a register ring of weight fragments that was meant to stay in
// flight (ISA of lstm_sb_step_kernel, round 4: one full L2 round trip per stage in the middle of the tap loop).  As inline asm the
// compiler does not know the load exists: its own counted waits stay counted (and, loads returning in order, only ever wait for
// MORE than they need when such a DMA is older than the load they want); whoever reads the tile waits for it by hand
// (s_waitcnt vmcnt(n) + s_barrier as asm with a memory clobber).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void sb_lds_dma16(const void* src, unsigned lds_addr) {
    // (readfirstlane: an "s" operand the compiler cannot prove uniform would be handed over in a VGPR)
    const unsigned m = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}
// LDS address of byte 0 of a kernel's `extern __shared__` array (it follows the kernel's static LDS)
__device__ __forceinline__ unsigned sb_dyn_lds_base() { return __builtin_amdgcn_groupstaticsize(); }
#pragma clang diagnostic pop

// device-side splits on the conversion instructions
__device__ __forceinline__ void split2_dev(float x, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)x;
    hi = __builtin_bit_cast(unsigned short, h);
    const _Float16 l = (_Float16)(x - (float)h);
    lo = (hi & 0x7c00u) == 0x7c00u ? (unsigned short)0 : __builtin_bit_cast(unsigned short, l);
}
// four consecutive channels -> 8 bytes of each term
__device__ __forceinline__ void split2_quad(const float (&v)[4], uint2& hi, uint2& lo) {
    unsigned short h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2_dev(v[i], h[i], l[i]);
    hi = uint2{h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16)};
    lo = uint2{l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16)};
}
__device__ __forceinline__ void split3_quad(const float (&v)[4], uint2& hi, uint2& mid, uint2& lo) {
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sb_split3(v[i], h[i], m[i], l[i]);
    hi = uint2{h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16)};
    mid = uint2{m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16)};
    lo = uint2{l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16)};
}
// x -> its terms as 16-bit patterns, t[0] the leading one
template <int TERMS>
__device__ __forceinline__ void sb_split_dev(float x, unsigned short (&t)[TERMS]) {
    if constexpr (TERMS == 2) split2_dev(x, t[0], t[1]);
    else sb_split3(x, t[0], t[1], t[2]);
}

// acc += A B for one block of split operands, small products first, the leading one last
template <int TERMS>
__device__ __forceinline__ sb_f16v sb_mma32(const sb8 (&af)[TERMS], const sb8 (&bf)[TERMS], sb_f16v acc) {
    if constexpr (TERMS == 2) {
#define BDE_H(x) __builtin_bit_cast(sb_h8, x)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(BDE_H(af[0]), BDE_H(bf[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(BDE_H(af[1]), BDE_H(bf[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(BDE_H(af[0]), BDE_H(bf[0]), acc, 0, 0, 0);
    } else {
#define BDE_B(x) __builtin_bit_cast(sb_b8, x)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BDE_B(af[0]), BDE_B(bf[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BDE_B(af[2]), BDE_B(bf[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BDE_B(af[1]), BDE_B(bf[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BDE_B(af[0]), BDE_B(bf[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BDE_B(af[1]), BDE_B(bf[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BDE_B(af[0]), BDE_B(bf[0]), acc, 0, 0, 0);
    }
    return acc;
}
template <int TERMS>
__device__ __forceinline__ sb_f4 sb_mma16(const sb8 (&af)[TERMS], const sb8 (&bf)[TERMS], sb_f4 acc) {
    if constexpr (TERMS == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(BDE_H(af[0]), BDE_H(bf[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(BDE_H(af[1]), BDE_H(bf[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(BDE_H(af[0]), BDE_H(bf[0]), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BDE_B(af[0]), BDE_B(bf[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BDE_B(af[2]), BDE_B(bf[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BDE_B(af[1]), BDE_B(bf[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BDE_B(af[0]), BDE_B(bf[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BDE_B(af[1]), BDE_B(bf[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BDE_B(af[0]), BDE_B(bf[0]), acc, 0, 0, 0);
    }
    return acc;
}
#endif

}  // namespace bde
