#pragma once
// PCG32 (O'Neill 2014; the reference carries Wenzel Jakob's pcg32.h) — the
// published generator restated: LCG state, XSH-RR output, log-time skip-ahead.
struct Pcg32 {
  unsigned long long state, inc;
  __device__ unsigned int next_uint() {
    const unsigned long long old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    const unsigned int xs = (unsigned int)(((old >> 18u) ^ old) >> 27u);
    const unsigned int rot = (unsigned int)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
  }
  __device__ float next_float() { return __uint_as_float((next_uint() >> 9) | 0x3f800000u) - 1.0f; }
  __device__ void advance(unsigned long long delta) {
    unsigned long long cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
      if (delta & 1) {
        acc_mult *= cur_mult;
        acc_plus = acc_plus * cur_mult + cur_plus;
      }
      cur_plus = (cur_mult + 1) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1;
    }
    state = acc_mult * state + acc_plus;
  }
};
