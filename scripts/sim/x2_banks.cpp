// Development aid: LDS bank-conflict cost of the last-layout side of X2 (ds_read/write_b32, banks = dword address mod 32
// per 32-lane half) for the generated slot table, under candidate XOR swizzles of the lane part of the address.
// Build: /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++20 -o /tmp/x2_banks scripts/sim/x2_banks.cpp (from the repo root: cd scripts/sim first)
#include <cstdio>
#include <vector>
#include <set>
#include <functional>
#include "../../regularizepsf_amd/csrc/rpsf_core.hpp"
using namespace rpsf;
template <class C>
double cost(const std::vector<uint16_t>& tab, std::function<int(int)> s) {
  // last-side accesses: per wave, per GI (slot-major, member minor), per EE: lanes read lds[EE*STRIDE + phys(gid)]
  long cyc = 0, ops = 0;
  for (int w = 0; w < C::T / 64; ++w)
    for (int gi = 0; gi < C::P; ++gi)
      for (int half = 0; half < 2; ++half) {
        int cnt[32] = {0};
        std::set<int> seen;
        for (int l = 0; l < 32; ++l) {
          int t = w * 64 + half * 32 + l;
          int gid = tab[(size_t)t * C::P + gi];
          int lane = gid & 63, j = gid >> 6;
          int phys = (j << 6) + (lane ^ s(j));
          if (seen.insert(phys).second) cnt[phys & 31]++;
        }
        int mx = 0; for (int b = 0; b < 32; ++b) mx = cnt[b] > mx ? cnt[b] : mx;
        cyc += mx; ops += 1;
      }
  return (double)cyc / ops;
}
template <class C> void run(const char* name) {
  std::vector<uint16_t> tab((size_t)C::T * C::NSLOT * 2);
  build_slot_table<C>(tab.data());
  printf("%s: T=%d P=%d NSLOT=%d E=%d\n", name, C::T, C::P, C::NSLOT, C::E);
  printf("  none        %.3f\n", cost<C>(tab, [](int){return 0;}));
  printf("  j&31        %.3f\n", cost<C>(tab, [](int j){return j & 31;}));
  for (int sh = 0; sh < 6; ++sh) for (int mul : {1,2,4,8,16}) for (int nb : {1,2,3}) {
    int mask = (1<<nb)-1;
    double c = cost<C>(tab, [=](int j){return (((j >> sh) & mask) * mul) & 31;});
    printf("  ((j>>%d)&%d)*%d  %.3f\n", sh, mask, mul, c);
  }
}
int main() { run<Cfg256>("Cfg256"); run<Cfg128>("Cfg128"); }
