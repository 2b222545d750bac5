// Host build of rap_amd/csrc/kabsch.h (the exact code the device solve kernel runs) for CPU unit tests.
#include "../../rap_amd/csrc/kabsch.h"
extern "C" void kabsch_from_points(const float* src, const float* tgt, int n, float* R, float* t) {
  double m[15];
  for (int k = 0; k < 15; ++k) m[k] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double s[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
    const double q[3] = {tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2]};
    for (int a = 0; a < 3; ++a) { m[a] += s[a]; m[3 + a] += q[a]; for (int b = 0; b < 3; ++b) m[6 + 3 * a + b] += s[a] * q[b]; }
  }
  rap_kabsch_from_moments(m, n, R, t);
}
