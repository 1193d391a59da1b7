// Split-bf16 ("X6") operand helpers shared by the forward / dgrad GEMM (gemm.hip) and the weight-gradient GEMM
// (gemm_tn.hip): an f32 value is the exact sum of three bf16 (truncation splits), the six largest of the nine
// cross products go through v_mfma_f32_32x32x16_bf16 with f32 accumulation -- the accuracy of an f32 fma chain at
// several times the f32 matrix-core rate (gemm.hip's header has the error budget).
#pragma once
#include "common.hpp"

namespace gs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// a = hi + mid + lo exactly, each with 8 significand bits (the upper half of an f32 word): truncation splits.
// Non-finite operands: a = +-inf gives hi = +-inf and mid = NaN (inf - inf), so every output the element contributes
// to is NaN where an f32 fma chain would keep +-inf (or produce NaN itself against a zero weight); NaN stays NaN.  Both
// say "not a number came in" -- a select per element to keep the infinity would cost a tenth of the split's VALU
// work on every finite element (tests/test_gpu_folded.py pins the behaviour).  Subnormal lo parts keep their value:
// the split is by subtraction, bf16 has the f32 exponent range.
__device__ __forceinline__ void gs_split3(float a, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
  hi = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(hi);          // exact: the low 16 significand bits
  mid = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mid);        // exact: at most 8 significand bits are left
  lo = __float_as_uint(r2);                          // (its low half is zero)
}
// two bf16 (upper halves of x0, x1) in one dword, x0 in the low half (k order = memory order)
__device__ __forceinline__ uint32_t gs_pack_hi16(uint32_t x0, uint32_t x1) { return (x0 >> 16) | (x1 & 0xffff0000u); }

constexpr int kX6RowBytes = 48;   // 16 bf16 + 16 B of padding: 12 dwords = 4 x odd -> conflict-free b128 fragment reads

bool gemm_x6_enabled();   // gemm.hip: GNNSAFT_GEMM_X6 = 0 selects the f32 matrix-core kernels everywhere

}  // namespace gs
