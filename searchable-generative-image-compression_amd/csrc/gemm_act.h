// Epilogue activations shared by the GEMM kernels (gemm.hip: exact-fp32 MFMA; gemm_split.hip: bf16x3 split).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_TANH = 3, ACT_LRELU = 4 };

// Exact (erf) GELU for the fc1 epilogues: gelu(x) = x * Phi(x) = 0.5 x (1 + erf(x / sqrt 2)), nn.GELU's default.
// The epilogue of a 128x128 tile evaluates it 64 times per lane while the matrix pipe waits, so its VALU cost is
// GEMM time (measured with tools/micro/gemm_stamps.hip: a (6,4) rational erf + IEEE division, ~35 VALU operations per
// element, made the epilogue 15 % of an fc1 tile).  This form is Abramowitz & Stegun 7.1.26 for erfc,
//     erfc(a) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-a^2),  t = 1 / (1 + p a),  a >= 0,  |error| <= 1.5e-7,
// with the hardware reciprocal and exp2 (1 ulp each): ~14 VALU operations, branch-free.  1 + erf(z) is taken as
// 2 - erfc(|z|) for z >= 0 and as erfc(|z|) itself for z < 0, so the negative tail has no cancellation at all
// (max |gelu error| 4e-7 over [-12, 12], the level of the fp32 rounding of the result).
// contract(off): the value must not depend on which epilogue instantiation the compiler inlined it into (left to itself it
// contracts the plain multiplies / adds below differently per call site: 1-ulp differences between kernel variants).
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
  const float z = x * 0.70710678118654752440f;
  const float a = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(a * a * -1.44269504088896340736f);   // exp(-a^2)
  const float erfc_a = p * t * e;
  const float one_plus_erf = z >= 0.f ? 2.0f - erfc_a : erfc_a;
  return 0.5f * x * one_plus_erf;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_GELU: return gelu_erf(v);
    case ACT_SILU: return v / (1.0f + expf(-v));
    case ACT_TANH: return tanhf(v);
    case ACT_LRELU: return v >= 0.f ? v : 0.01f * v;
    default: return v;
  }
}

template <int ACT>
__device__ __forceinline__ float apply_act_c(float v) {
  if constexpr (ACT == ACT_GELU) return gelu_erf(v);
  else if constexpr (ACT == ACT_SILU) return v / (1.0f + expf(-v));
  else if constexpr (ACT == ACT_TANH) return tanhf(v);
  else if constexpr (ACT == ACT_LRELU) return v >= 0.f ? v : 0.01f * v;
  else return v;
}

template <int V>
struct IntC {
  static constexpr int value = V;
};

