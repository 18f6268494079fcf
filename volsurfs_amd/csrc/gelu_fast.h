// Exact-GELU building blocks in ~13 vector instructions per element (erff + expf of the device library: ~45).
//
// torch.nn.GELU() (models/mlp.py:34, models/nerfhash.py:56): GELU(z) = z Phi(z), GELU'(z) = Phi(z) + z phi(z).
// Phi(-u) = 0.5 erfc(u / sqrt 2) = 2^Q(u) on u in [0, 8] with ONE degree-9 polynomial Q (tools/fit_gelu_cdf.py:
// Lawson-reweighted least squares on the ABSOLUTE error of Phi; in fp32 Horner arithmetic max |Phi error| =
// 6.6e-8 = half an ulp of 1, max |GELU error| = 1.6e-7 over [-12, 12]); Phi(z) = 1 - Phi(-z) for z > 0.
// Beyond |z| = 8 Phi(-u) < 7e-16: the argument is clamped.
#pragma once

__device__ __forceinline__ float gelu_phi_neg(float u) {        // Phi(-u), u >= 0
  u = fminf(u, 8.0f);
  float q = 5.068341908e-07f;
  q = __builtin_fmaf(q, u, -9.472626408e-06f);
  q = __builtin_fmaf(q, u, 7.445209598e-05f);
  q = __builtin_fmaf(q, u, -2.826963563e-04f);
  q = __builtin_fmaf(q, u, 1.197748286e-05f);
  q = __builtin_fmaf(q, u, 6.934337472e-03f);
  q = __builtin_fmaf(q, u, -5.243671404e-02f);
  q = __builtin_fmaf(q, u, -4.592208355e-01f);
  q = __builtin_fmaf(q, u, -1.151104352e+00f);
  q = __builtin_fmaf(q, u, -9.999999969e-01f);
  return __builtin_amdgcn_exp2f(q);
}

__device__ __forceinline__ float gelu_cdf(float z) {
  const float e = gelu_phi_neg(__builtin_fabsf(z));
  return z < 0.0f ? e : 1.0f - e;
}

__device__ __forceinline__ float gelu_fast(float z) { return z * gelu_cdf(z); }

// Phi(z) and the standard normal density phi(z) = exp(-z^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ void gelu_cdf_pdf(float z, float& cdf, float& pdf) {
  cdf = gelu_cdf(z);
  pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);
}
