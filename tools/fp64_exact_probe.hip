// Probe: are double sqrt / divide / reciprocal on gfx950 bit-identical to the host's IEEE results
// when compiled with -ffp-contract=off?  (Design gate for bit-exact GPU<->oracle parity.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const double* a, const double* b, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = a[i], y = b[i];
  o[4 * i + 0] = sqrt(x);
  o[4 * i + 1] = x / y;
  o[4 * i + 2] = 1.0 / y;
  o[4 * i + 3] = x * y + x;   // must NOT contract into an fma
}
static uint64_t s = 88172645463325252ull;
static uint64_t nx() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
  const int n = 1 << 22;
  std::vector<double> a(n), b(n), o(4 * (size_t)n);
  for (int i = 0; i < n; i++) {
    double u = (nx() >> 11) * (1.0 / 9007199254740992.0), v = (nx() >> 11) * (1.0 / 9007199254740992.0);
    int ea = (int)(nx() % 80) - 40, eb = (int)(nx() % 80) - 40;
    a[i] = ldexp(u + 0.5, (i & 1) ? ea : 0); b[i] = ldexp(v + 0.5, (i & 2) ? eb : 0);
  }
  double *da, *db, *dout;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, 4 * (size_t)n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(da, db, dout, n);
  hipMemcpy(o.data(), dout, 4 * (size_t)n * 8, hipMemcpyDeviceToHost);
  long bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    volatile double x = a[i], y = b[i];
    double r[4]; r[0] = sqrt(x); r[1] = x / y; r[2] = 1.0 / y; volatile double m = x * y; r[3] = m + x;
    for (int j = 0; j < 4; j++) if (memcmp(&r[j], &o[4 * (size_t)i + j], 8)) bad[j]++;
  }
  printf("mismatches of %d: sqrt=%ld div=%ld rcp=%ld mul_add=%ld\n", n, bad[0], bad[1], bad[2], bad[3]);
  return (bad[0] | bad[1] | bad[2] | bad[3]) ? 1 : 0;
}
