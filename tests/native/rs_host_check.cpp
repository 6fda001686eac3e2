// Host check of the Reeds-Shepp product core (rpp_rs.h) against the reference's known-answer vectors, read from a
// flat binary dump written by tests/test_core_host.py (no numpy on this side).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rpp_rs.h"
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  long bad = 0, cases = 0;
  for (;;) {
    double inp[8];
    int32_t hdr[2];   // n, n_len
    if (fread(inp, 8, 8, f) != 8) break;
    if (fread(hdr, 4, 2, f) != 2) break;
    char mode[8] = {0};
    if (fread(mode, 1, 8, f) != 8) break;
    double len[5];
    if (fread(len, 8, 5, f) != 5) break;
    const int n = hdr[0];
    std::vector<double> ex(n > 0 ? n : 0), ey(n > 0 ? n : 0), eyaw(n > 0 ? n : 0);
    if (n > 0) {
      if (fread(ex.data(), 8, n, f) != (size_t)n || fread(ey.data(), 8, n, f) != (size_t)n || fread(eyaw.data(), 8, n, f) != (size_t)n) break;
    }
    static double px[8192], py[8192], pyaw[8192];
    rpp::RsResult R;
    rpp::rs_plan(inp[0], inp[1], inp[2], inp[3], inp[4], inp[5], inp[6], inp[7], px, py, pyaw, 8192, &R);
    cases++;
    bool ok = true;
    if (n < 0) {
      ok = R.err == ((strcmp(mode, "ZeroDivisionError") == 0 || strncmp(mode, "ZeroDiv", 7) == 0) ? -3 : -4);
    } else if (n == 0) {
      ok = R.err == 0 && R.n == 0;
    } else {
      ok = R.err == 0 && R.n == n && strcmp(R.ct, mode) == 0 && R.nl == hdr[1] && !memcmp(R.len, len, 8 * hdr[1]) &&
           !memcmp(px, ex.data(), 8 * n) && !memcmp(py, ey.data(), 8 * n) && !memcmp(pyaw, eyaw.data(), 8 * n);
    }
    if (!ok) {
      if (bad++ < 5) printf("case %ld: n %d vs %d, mode %s vs %s, err %d\n", cases - 1, R.n, n, R.ct, mode, R.err);
    }
  }
  printf("rs cases %ld mismatches %ld\n", cases, bad);
  return bad || cases == 0;
}
