// Host build of csrc/kd6d_det.h's pure functions for tests/test_det_accumulator.py (g++, no GPU).
#include "../kd-6d-pose-adlp_amd/csrc/kd6d_det.h"

extern "C" void det_split_act(float v, long long* out) {
  const kd6d_detail::det_words w = kd6d_detail::det_split<KD6D_DET_ACT>(v);
  out[0] = w.lo; out[1] = w.hi;
}
extern "C" void det_split_grad(float v, long long* out) {
  const kd6d_detail::det_words w = kd6d_detail::det_split<KD6D_DET_GRAD>(v);
  out[0] = w.lo; out[1] = w.hi;
}
extern "C" float det_value_act(long long lo, long long hi) { return kd6d_detail::det_value<KD6D_DET_ACT>(lo, hi); }
extern "C" float det_value_grad(long long lo, long long hi) { return kd6d_detail::det_value<KD6D_DET_GRAD>(lo, hi); }
// sum n floats through the accumulator (what a reduction of n workgroup partials does), any order
extern "C" float det_sum_act(const float* v, long long n) {
  long long lo = 0, hi = 0;
  for (long long i = 0; i < n; ++i) { const auto w = kd6d_detail::det_split<KD6D_DET_ACT>(v[i]); lo += w.lo; hi += w.hi; }
  return kd6d_detail::det_value<KD6D_DET_ACT>(lo, hi);
}
extern "C" float det_sum_grad(const float* v, long long n) {
  long long lo = 0, hi = 0;
  for (long long i = 0; i < n; ++i) { const auto w = kd6d_detail::det_split<KD6D_DET_GRAD>(v[i]); lo += w.lo; hi += w.hi; }
  return kd6d_detail::det_value<KD6D_DET_GRAD>(lo, hi);
}
