// Development aid: LDS bank-conflict cost of the last-layout side of X2 for slot tables generated in different walk
// orders and with different row strides of the group grid ('tried and measured').
// Build from scripts/sim: /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++20 -o /tmp/x2_orders x2_orders.cpp
#include <cstdio>
#include <vector>
#include <set>
#include <functional>
#include "../../regularizepsf_amd/csrc/rpsf_core.hpp"
using namespace rpsf;
// variant slot table: pass 1 walks g in a custom order
template <class C>
void build_tab(uint16_t* tab, std::function<int(int)> order) {
  const int G = C::G;
  std::vector<char> seen(G, 0); std::vector<int> slots; int selfs[4], nself = 0;
  for (int g = 0; g < G; ++g) if (partner_gid<C>(g) == g) selfs[nself++] = g;
  for (int i = 0; i + 1 < nself; i += 2) { slots.push_back(selfs[i]); slots.push_back(selfs[i+1]); seen[selfs[i]] = seen[selfs[i+1]] = 1; }
  for (int pass = 0; pass < 2; ++pass)
    for (int n = 0; n < G; ++n) {
      int g = pass == 0 ? n : order(n);
      if (seen[g]) continue;
      int q, m; gid_to_qm<C>(g, q, m);
      if (pass == 0 && q != 0 && m != 0) continue;
      int p = partner_gid<C>(g);
      slots.push_back(g); slots.push_back(p); seen[g] = seen[p] = 1;
    }
  int ns = slots.size() / 2;
  for (int sigma = 0; sigma < ns; ++sigma) { int s = sigma / C::T, t = sigma % C::T; tab[(t*C::NSLOT+s)*2] = slots[2*sigma]; tab[(t*C::NSLOT+s)*2+1] = slots[2*sigma+1]; }
}
template <class C>
double cost(const std::vector<uint16_t>& tab, int RS, bool verbose=false) {
  long cyc = 0, ops = 0;
  for (int gi = 0; gi < C::P; ++gi) { if (verbose) printf("gi%d:", gi);
    for (int w = 0; w < C::T / 64; ++w)
      for (int half = 0; half < 2; ++half) {
        int cnt[32] = {0};
        for (int l = 0; l < 32; ++l) {
          int gid = tab[(size_t)(w*64+half*32+l) * C::P + gi];
          int phys = (gid >> 6) * RS + (gid & 63);
          cnt[phys & 31]++;
        }
        int mx = 0; for (int b = 0; b < 32; ++b) mx = cnt[b] > mx ? cnt[b] : mx;
        cyc += mx; ops += 1; if (verbose) printf(" %d", mx);
      }
    if (verbose) printf("\n"); }
  return (double)cyc / ops;
}
template <class C> void run(const char* name) {
  std::vector<uint16_t> tab((size_t)C::T * C::NSLOT * 2);
  printf("%s\n", name);
  for (int RS : {64, 66, 68, 70, 72})
    for (int rd = 0; rd < 2; ++rd) for (int ld = 0; ld < 2; ++ld) for (int blk : {64, 8, 4, 2, 1}) {
      auto ord = [=](int n){ int j = n >> 6, lane = n & 63; if (rd) j = 63 - j; int hi = lane / blk, lo = lane % blk; if (ld) hi = 64/blk - 1 - hi; return (j << 6) + hi*blk + lo; };
      build_tab<C>(tab.data(), ord); printf("  RS=%d rows %s, %d-lane blocks %s: %.3f\n", RS, rd?"desc":"asc", blk, ld?"desc":"asc", cost<C>(tab, RS));
    }
}
int main() { run<Cfg256>("Cfg256"); run<Cfg128>("Cfg128"); }
