// rpsf_core2.hpp - second-generation per-thread phases of the fused patch kernel for the three-stage plans
// (N = 128 and N = 256).  Same algorithm as rpsf_core.hpp (packed real 2-D DFT, Hermitian-folded K, overlap-add;
// regularizepsf/transform.py:151-169), re-laid out around the two widths the memory system likes:
//   * a thread's unit of global traffic is 16 bytes = four consecutive pixels of one row = the packed columns 2c', 2c'+1.
//     The low packed-column bit c3 ("half" h) therefore lives in a register digit that is never exchanged: a thread
//     owns 32 units = 2 halves x 32 complex values, and every stage works on one half (32 values, 5 index bits)
//     at a time, so the LDS traffic of one half overlaps the butterflies of the other;
//   * the unit of LDS traffic is one complex value (8 bytes): ds_write_b64 / ds_read_b64 / ds_read_b128, half the
//     instructions of the re/im passes of the first generation.
// Digits (most significant first): rows (A1 | A2 | AL), packed columns (B1 | B2 | 1); A1 + B1 = 5, A2 + B2 = 5.
//   stage 1 owns (r1, c1), stage 2 (r2, c2), the last stage (r3, c3): E = 2^(AL+1) bins per group.
//   r = r1 2^(A2+AL) + r2 2^AL + r3  ->  kr = k1 + k2 2^A1 + k3 2^(A1+A2) = q + Q k3
//   c = c1 2^(B2+1) + 2 c2 + c3      ->  kc = l1 + l2 2^B1 + l3 2^(B1+B2) = m + M l3
// Thread t = 64 wave + 32 hb + l5:  r3 = 2 wave + hb in the stage-1 / stage-2 layouts; l5 = (r2, c2) in the
// stage-1 layout and (k1, l1) in the stage-2 layout (X1 swaps the five exchanged register bits with l5 inside each
// half-wave).  Register v[2 j + h] in stages 1-2, v[group E + 2 r3 + c3] in the last stage, so the exchange of half h
// replaces exactly the registers of parity h.
// Groups whose partner rule differs (q = 0 or m = 0) are brought to the general rule "bin e <-> bin E-1-e of the
// partner group" by modulating the partner group before / after its last-stage DFT (a shift by one bin in k3, or a
// swap in l3); only the four self-paired groups need their own path (two threads park them in LDS and wave 0 walks
// their bin pairs, one pair per lane).
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "rpsf_core.hpp"

namespace rpsf {

struct alignas(16) cf2 {
  cf a, b;
};

template <int LOGN_, int A1_, int A2_, int AL_, int B1_, int B2_>
struct Cfg2 {
  static constexpr int LOGN = LOGN_, N = 1 << LOGN_, NC = N / 2, LOGR = LOGN_, ROWS = N;
  static constexpr int A1 = A1_, A2 = A2_, AL = AL_, B1 = B1_, B2 = B2_, BL = 1;
  static_assert(A1_ + B1_ == 5 && A2_ + B2_ == 5, "32 values per thread, half and stage");
  static_assert(A1_ + A2_ + AL_ == LOGN_ && B1_ + B2_ + 1 == LOGN_ - 1, "digits must cover the index");
  static_assert(A1_ >= 1 && B1_ >= 1, "the top row / column bits must be stage-1 register digits (quadrants)");
  static constexpr bool S3 = true;
  static constexpr int EA = 1 << AL_, EB = 2, E = EA * EB, P = 64 / E, NSLOT = P / 2;
  static constexpr int T = ROWS * NC / 64, WAVES = T / 64;
  static_assert(WAVES * 2 == EA, "r3 = 2 wave + hb");
  static constexpr int LQ = A1_ + A2_, Q = 1 << LQ, M = 1 << (B1_ + B2_), G = Q * M;
  static_assert(G == 1024, "32 x 32 groups");
  static constexpr bool SPLIT_ROWS = NSLOT == 1;  // see stage3_rows
  static constexpr int KCH = 8, NWORDS = NSLOT * E, NCHUNK = NWORDS / KCH;
  static_assert(NWORDS == 32, "32 pair words per thread");
  static constexpr int G_PER_PATCH = NWORDS * T * 2;  // complex values
  // self-paired groups (0,0), (Q/2,0), (0,M/2), (Q/2,M/2): slots 0 of threads 0 and 1; their bin pairs
  static constexpr int NORBIT = 2 * E + 2;            // E/2 + 2 pairs in group (0,0) (four fixed points), E/2 in the others
  static constexpr int ORBIT_ROUNDS = (NORBIT + 63) / 64;
  static constexpr int GS_PER_PATCH = ORBIT_ROUNDS * 64 * 2;  // complex values: one pair word per (round, lane)
  // LDS, in 8-byte units
  static constexpr int X1_ROWU = 34;  // 32 + 2: rows stay 16-byte aligned and a 16-lane group of b128 reads covers all banks
  // X2 image: EA planes of 32 rows (register digit j'') x 32 units (lane digit l5'), rows X2_ROWU units apart.  With 33 (34 for the 128-pixel plan)
  // instead of 32 the bank of a group's unit depends on both digits, which is what lets build_slot_table2 deal the slots so that the
  // gid-indexed side of the exchange (x2_last_read2 / x2_last_write2) is free of bank conflicts; the lane-indexed side touches runs of
  // consecutive units either way.  Both fit under the X1 regions' size.
  static constexpr int X2_ROWU = LOGN_ == 8 ? 33 : 34;
  // A plane is X2_G = 32 x 34 units long whatever the row step, so that two planes are exactly one wave's X1 region:
  // X1 region of wave w (64 rows of X1_ROWU units) = the wave's OWN two planes of the X2 image, r3 = 2w and 2w + 1 - the planes only its own
  // x2_mid_write2 / x2_mid_read2 touch - so that a wave goes from X1 to X2 and back without a workgroup barrier (two of the ten barriers of a
  // pass until round 4).
  static constexpr int X2_G = 32 * X1_ROWU;
  static_assert(X2_G >= 32 * X2_ROWU && X2_G % 32 == 0, "rows of a plane; a unit's bank must not depend on the plane");
  static constexpr int X2_UNITS = EA * X2_G;
  static constexpr int BUF_UNITS = X2_UNITS;
  static constexpr int PARK_UNITS = 2 * 2 * E;  // two threads x two groups
  static constexpr int SYNC_UNITS = 4;  // eight words behind the park area: arrival counters of the split barriers (rpsf_kernels2.hpp, RPSF_SPLIT_BARRIERS)
  static constexpr int LDS_UNITS = BUF_UNITS + PARK_UNITS + SYNC_UNITS;
  static constexpr float SCALE = 1.0f / (2.0f * (float)N * (float)N);  // 1/4 (pair algebra) * 1/(N*N/2) (inverse DFT)
};

// ------------------------------------------------------------------------------------------
// group ids: gid = (j'' << 5) + l5',  j'' = (k2, l2) register digit of the stage-2 layout, l5' = (k1, l1)
// ------------------------------------------------------------------------------------------
template <class C>
RPSF_HD void gid_to_qm2(int gid, int& q, int& m) {
  const int l5 = gid & 31, j = gid >> 5;
  const int k1 = l5 >> C::B1, l1 = l5 & ((1 << C::B1) - 1);
  const int k2 = j >> C::B2, l2 = j & ((1 << C::B2) - 1);
  q = k1 + (k2 << C::A1);
  m = l1 + (l2 << C::B1);
}
template <class C>
RPSF_HD int qm_to_gid2(int q, int m) {
  const int k1 = q & ((1 << C::A1) - 1), k2 = q >> C::A1;
  const int l1 = m & ((1 << C::B1) - 1), l2 = m >> C::B1;
  return (((k2 << C::B2) + l2) << 5) + (k1 << C::B1) + l1;
}
template <class C>
RPSF_HD int partner_gid2(int gid) {
  int q, m;
  gid_to_qm2<C>(gid, q, m);
  return qm_to_gid2<C>((C::Q - q) & (C::Q - 1), (C::M - m) & (C::M - 1));
}
// unit of group gid = (j'' << 5) + l5' inside a plane of the X2 image
template <class C>
RPSF_HD int x2_unit(int gid) { return gid + (gid >> 5) * (C::X2_ROWU - 32); }
enum SlotKind : int { SLOT_GENERAL = 0, SLOT_Q0 = 1, SLOT_M0 = 2, SLOT_SELF = 3 };
template <class C>
RPSF_HD int slot_kind2(int gid_a) {
  int q, m;
  gid_to_qm2<C>(gid_a, q, m);
  if (partner_gid2<C>(gid_a) == gid_a) return SLOT_SELF;
  if (q == 0) return SLOT_Q0;
  if (m == 0) return SLOT_M0;
  return SLOT_GENERAL;
}

// Slot table: tab[(t*NSLOT + s)*2 + member] = gid.  Slot sigma = s*T + t.  The two self-paired slots are threads 0 and 1 of slot 0
// ((0,0)+(Q/2,0) and (0,M/2)+(Q/2,M/2)), the q = 0 / m = 0 pairs sit in slot 0 of wave 0 (only that wave runs their code).  Host only.
//
// Which thread takes which pair, and which group of a pair is member A, is free otherwise - K is folded by the same table (pack_kernel2) -
// and decides the LDS bank conflicts of the gid-indexed side of the X2 exchange: a ds_read_b64 serves 32 lanes at once from 64 banks, a
// ds_write_b64 16 lanes from 32, one LDS-array cycle per distinct address on a bank, so the 32 lanes of a half wave want member-A units that
// differ modulo 32, member-B units likewise, and each run of 16 lanes units that differ modulo 16.  In ascending order of gid (the table until
// round 4) the reads took 2.8 x and the writes 1.95 x the conflict-free cycles at both sizes - the q = 0 pairs, 31 slots of wave 0, sat on two
// bank pairs.  deal_slots2 deals the pairs so that every half wave is conflict-free (tests/test_emulator.py counts the cycles):
//  1. the special pairs go to the two half waves of wave 0 / slot 0, half and orientation chosen depth-first, a branch left as soon as the bank
//     values still missing in a half cannot be supplied by general pairs (a bipartite matching of bank values);
//  2. the two halves are completed by such a matching;
//  3. the remaining pairs form a regular multigraph on the 32 bank values; an Euler orientation makes every value a member-A bank as often as a
//     member-B bank, and the regular bipartite multigraph (A bank -> B bank) falls apart into perfect matchings - one half wave each;
//  4. inside a half the 32 slots split into two runs of 16 whose units differ modulo 16 (two-colouring of the union of two perfect matchings).
namespace slots2 {
// perfect matching, left vertex i -> adj[i] = (right vertex, tag); result[i] = (right, tag), empty if there is none (Kuhn's augmenting paths)
inline std::vector<std::pair<int, int>> match(const std::vector<std::vector<std::pair<int, int>>>& adj, int n_right) {
  const int nl = (int)adj.size();
  std::vector<int> owner(n_right, -1), tag(n_right, -1);
  std::vector<char> seen;
  auto grow = [&](auto&& self, int a) -> bool {
    for (const auto& [b, e] : adj[a]) {
      if (seen[b]) continue;
      seen[b] = 1;
      if (owner[b] < 0 || self(self, owner[b])) {
        owner[b] = a, tag[b] = e;
        return true;
      }
    }
    return false;
  };
  for (int a = 0; a < nl; ++a) {
    seen.assign(n_right, 0);
    if (!grow(grow, a)) return {};
  }
  std::vector<std::pair<int, int>> out(nl);
  for (int b = 0; b < n_right; ++b)
    if (owner[b] >= 0) out[owner[b]] = {b, tag[b]};
  return out;
}
}  // namespace slots2

template <class C>
inline bool deal_slots2(std::vector<std::pair<int, int>>& slots) {
  using Pair = std::pair<int, int>;
  constexpr int NB = 32;
  const int T = C::T, NS = C::NSLOT, nhalves = T * NS / 32;
  auto bank = [](int g) { return x2_unit<C>(g) & (NB - 1); };
  std::vector<char> seen(C::G, 0);
  const Pair selfs[2] = {{qm_to_gid2<C>(0, 0), qm_to_gid2<C>(C::Q / 2, 0)}, {qm_to_gid2<C>(0, C::M / 2), qm_to_gid2<C>(C::Q / 2, C::M / 2)}};
  for (const Pair& s : selfs) seen[s.first] = seen[s.second] = 1;
  std::vector<Pair> special, general;
  for (int g = 0; g < C::G; ++g) {
    if (seen[g]) continue;
    const int p = partner_gid2<C>(g);
    if (p == g) return false;
    seen[g] = seen[p] = 1;
    int q, m;
    gid_to_qm2<C>(g, q, m);
    (q == 0 || m == 0 ? special : general).push_back({g, p});
  }
  if ((int)special.size() + 2 > 64) return false;
  // general pairs by bank type, either way round: (pair index, swapped)
  std::vector<Pair> pool[NB][NB];
  for (int e = 0; e < (int)general.size(); ++e) {
    const int x = bank(general[e].first), y = bank(general[e].second);
    pool[x][y].push_back({e, 0});
    if (x != y) pool[y][x].push_back({e, 1});
  }
  std::vector<std::vector<Pair>> halves(2);
  bool used_a[2][NB] = {}, used_b[2][NB] = {};
  for (const Pair& s : selfs) {
    if (used_a[0][bank(s.first)] || used_b[0][bank(s.second)]) return false;
    halves[0].push_back(s), used_a[0][bank(s.first)] = used_b[0][bank(s.second)] = true;
  }
  // (tag of a matched edge: missing A bank * 32 + missing B bank)
  auto completion = [&](int h) {
    std::vector<int> mb, ib(NB, -1);
    for (int y = 0; y < NB; ++y)
      if (!used_b[h][y]) ib[y] = (int)mb.size(), mb.push_back(y);
    std::vector<std::vector<Pair>> adj;
    for (int x = 0; x < NB; ++x) {
      if (used_a[h][x]) continue;
      adj.emplace_back();
      for (int y : mb)
        if (!pool[x][y].empty()) adj.back().push_back({ib[y], x * NB + y});
    }
    return slots2::match(adj, (int)mb.size());
  };
  long budget = 200000;  // (both plans finish in a few dozen steps; a bound, so that a configuration without a solution falls back quickly)
  auto place = [&](auto&& self, int i) -> bool {
    if (i == (int)special.size()) return true;
    if (--budget < 0) return false;
    for (int h = 0; h < 2; ++h) {
      if (halves[h].size() >= 32) continue;
      for (int o = 0; o < 2; ++o) {
        const Pair pr = o ? Pair{special[i].second, special[i].first} : special[i];
        const int x = bank(pr.first), y = bank(pr.second);
        if (used_a[h][x] || used_b[h][y]) continue;
        halves[h].push_back(pr), used_a[h][x] = used_b[h][y] = true;
        if ((halves[h].size() == 32 || !completion(h).empty()) && self(self, i + 1)) return true;
        halves[h].pop_back(), used_a[h][x] = used_b[h][y] = false;
      }
    }
    return false;
  };
  if (!place(place, 0)) return false;
  std::vector<char> taken(general.size(), 0);
  for (int h = 0; h < 2; ++h) {
    if (halves[h].size() == 32) continue;
    const auto ml = completion(h);
    if (ml.empty()) return false;
    for (const auto& [b, tag] : ml) {
      const int x = tag / NB, y = tag % NB;
      bool found = false;
      for (const auto& [e, sw] : pool[x][y]) {
        if (taken[e]) continue;
        taken[e] = 1, found = true;
        halves[h].push_back(sw ? Pair{general[e].second, general[e].first} : general[e]);
        break;
      }
      if (!found) return false;
    }
  }
  // the rest: Euler orientation on the bank values ...
  std::vector<Pair> rest;
  for (int e = 0; e < (int)general.size(); ++e)
    if (!taken[e]) rest.push_back(general[e]);
  std::vector<std::vector<int>> inc(NB);
  for (int e = 0; e < (int)rest.size(); ++e) {
    inc[bank(rest[e].first)].push_back(e);
    if (bank(rest[e].second) != bank(rest[e].first)) inc[bank(rest[e].second)].push_back(e);
  }
  std::vector<char> done(rest.size(), 0);
  std::vector<size_t> ptr(NB, 0);
  for (int start = 0; start < NB; ++start)
    for (;;) {
      int v = start;
      bool moved = false;
      for (;;) {
        while (ptr[v] < inc[v].size() && done[inc[v][ptr[v]]]) ++ptr[v];
        if (ptr[v] == inc[v].size()) break;
        const int e = inc[v][ptr[v]];
        if (bank(rest[e].first) != v) std::swap(rest[e].first, rest[e].second);
        done[e] = 1, v = bank(rest[e].second), moved = true;
      }
      if (!moved) break;
    }
  int deg_a[NB] = {}, deg_b[NB] = {};
  for (const Pair& pr : rest) ++deg_a[bank(pr.first)], ++deg_b[bank(pr.second)];
  for (int x = 0; x < NB; ++x)
    if (deg_a[x] != nhalves - 2 || deg_b[x] != nhalves - 2) return false;
  // ... and one perfect matching per remaining half wave
  std::vector<char> gone(rest.size(), 0);
  for (int k = 2; k < nhalves; ++k) {
    std::vector<std::vector<Pair>> adj(NB);
    for (int e = 0; e < (int)rest.size(); ++e)
      if (!gone[e]) adj[bank(rest[e].first)].push_back({bank(rest[e].second), e});
    const auto ml = slots2::match(adj, NB);
    if (ml.empty()) return false;
    halves.emplace_back();
    for (const auto& [b, e] : ml) halves.back().push_back(rest[e]), gone[e] = 1;
  }
  // runs of 16 lanes
  slots.assign((size_t)T * NS, Pair{-1, -1});
  for (int hi = 0; hi < nhalves; ++hi) {
    const std::vector<Pair>& hv = halves[hi];
    if (hv.size() != 32) return false;
    std::vector<int> nb[32];
    for (int member = 0; member < 2; ++member) {
      int first[16];
      for (int& f : first) f = -1;
      for (int i = 0; i < 32; ++i) {
        const int r = bank(member ? hv[i].second : hv[i].first) & 15;
        if (first[r] < 0) first[r] = i;
        else nb[first[r]].push_back(i), nb[i].push_back(first[r]);
      }
    }
    int col[32];
    for (int& c : col) c = -1;
    for (int s0 = 0; s0 < 32; ++s0) {
      if (col[s0] >= 0) continue;
      col[s0] = 0;
      std::vector<int> st{s0};
      while (!st.empty()) {
        const int x = st.back();
        st.pop_back();
        for (int y : nb[x])
          if (col[y] < 0) col[y] = 1 - col[x], st.push_back(y);
      }
    }
    std::vector<int> run[2];
    for (int i = 0; i < 32; ++i) run[col[i]].push_back(i);
    if (hi == 0) {  // lanes 0 and 1 are the self-paired slots (entries 0 and 1), whatever that costs the first run
      if (col[0] == 1) std::swap(run[0], run[1]);
      auto it = std::find(run[1].begin(), run[1].end(), 1);
      if (it != run[1].end()) {
        run[1].erase(it);
        int sw = -1;
        for (int i : run[0])
          if (i > 1) sw = i;
        run[0].erase(std::find(run[0].begin(), run[0].end(), sw));
        run[0].push_back(1), run[1].push_back(sw);
      }
      std::vector<int> r0{0, 1};
      for (int i : run[0])
        if (i > 1) r0.push_back(i);
      run[0] = r0;
    }
    if (run[0].size() != 16 || run[1].size() != 16) return false;
    const int s = hi / (2 * (T / 64)), w = hi % (2 * (T / 64)) / 2, hb = hi & 1;
    for (int l = 0; l < 32; ++l) slots[(size_t)s * T + w * 64 + hb * 32 + l] = hv[l < 16 ? run[0][l] : run[1][l - 16]];
  }
  return true;
}

template <class C>
inline void build_slot_table2(uint16_t* tab) {
  std::vector<std::pair<int, int>> slots;
  if (!deal_slots2<C>(slots)) {  // (not reached by the plans in use: any order with the special slots in wave 0 is correct, only slower)
    const int G = C::G;
    std::vector<char> seen(G, 0);
    slots.clear();
    auto push = [&](int a, int b) {
      slots.push_back({a, b});
      seen[a] = seen[b] = 1;
    };
    push(qm_to_gid2<C>(0, 0), qm_to_gid2<C>(C::Q / 2, 0));
    push(qm_to_gid2<C>(0, C::M / 2), qm_to_gid2<C>(C::Q / 2, C::M / 2));
    for (int pass = 0; pass < 2; ++pass)
      for (int g = 0; g < G; ++g) {
        if (seen[g]) continue;
        int q, m;
        gid_to_qm2<C>(g, q, m);
        if (pass == 0 && q != 0 && m != 0) continue;
        push(g, partner_gid2<C>(g));
      }
  }
  for (int sigma = 0; sigma < (int)slots.size(); ++sigma) {
    const int s = sigma / C::T, t = sigma % C::T;
    tab[(t * C::NSLOT + s) * 2 + 0] = (uint16_t)slots[sigma].first;
    tab[(t * C::NSLOT + s) * 2 + 1] = (uint16_t)slots[sigma].second;
  }
}
template <class C>
inline int special_slots2() { return 2 + (C::Q / 2 - 1) + (C::M / 2 - 1); }  // all must sit in slot 0 of wave 0

// Bin pairs of the four self-paired groups.  Entry: bits 0-7 x1, 8-15 x2 (positions among the 4E parked values:
// thread*2E + member*E + e), 16-23 twiddle index kc of the first bin, bit 31 valid.  Host only; NORBIT entries used.
template <class C>
inline int build_orbit_table2(const uint16_t* tab, uint32_t* ot) {
  int n = 0;
  for (int i = 0; i < C::ORBIT_ROUNDS * 64; ++i) ot[i] = 0;
  for (int t = 0; t < 2; ++t)
    for (int member = 0; member < 2; ++member) {
      int q, m;
      gid_to_qm2<C>(tab[(t * C::NSLOT + 0) * 2 + member], q, m);
      for (int e = 0; e < C::E; ++e) {
        const int k3 = e >> 1, l3 = e & 1;
        const int pk = q == 0 ? (C::EA - k3) % C::EA : C::EA - 1 - k3;
        const int pl = m == 0 ? l3 : 1 - l3;
        const int pe = pk * 2 + pl;
        if (e > pe) continue;
        if (n < C::ORBIT_ROUNDS * 64)
          ot[n] = pair_entry(t * 2 * C::E + member * C::E + e, t * 2 * C::E + member * C::E + pe, m + C::M * l3);
        ++n;
      }
    }
  return n;
}

// ------------------------------------------------------------------------------------------
// Thread coordinates
// ------------------------------------------------------------------------------------------
template <class C>
struct ThreadPos2 {
  int wave, hb, l5, r3;
  int r_low, c2;  // stage-1 layout: r = (r1 << (A2+AL)) + r_low; unit column c' = (c1 << B2) + c2 (pixels 4c' .. 4c'+3)
  RPSF_HD explicit ThreadPos2(int t) {
    wave = t >> 6, hb = (t >> 5) & 1, l5 = t & 31;
    r3 = 2 * wave + hb;
    const int r2 = l5 >> C::B2;
    c2 = l5 & ((1 << C::B2) - 1);
    r_low = (r2 << C::AL) + r3;
  }
};

// ------------------------------------------------------------------------------------------
// Stages.  v[2 j + H] in stages 1 and 2; tw[k] = exp(-2 pi i k / N)
// ------------------------------------------------------------------------------------------
template <class C, int H, bool INV>
RPSF_HD void stage1h(int t, cf* v, const cf* __restrict__ tw) {
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  ThreadPos2<C> tp(t);
#if defined(__HIP_DEVICE_COMPILE__)
  // The inverse stage recomputes its twiddle addresses (two integer instructions each): left to itself the compiler keeps the forward stage's fifteen
  // alive across the whole pass and spills three of them - scratch reloads in the last stage of the chain, behind the in-order vector-memory pipe.
  // (256-pixel plan: 3 spilled VGPRs / 16 B of scratch -> none, timing unchanged, profiles/r04x; the 128-pixel plan never spilled them)
  if constexpr (INV && C::SPLIT_ROWS) asm volatile("" : "+v"(tp.r_low), "+v"(tp.c2));
#endif
  const int c_low = 2 * tp.c2 + H;
  auto row_tw = [&]() RPSF_AI {
    StaticFor<1, NR>::run([&]<int K1>() RPSF_AI {
      const cf w = tw[(K1 * tp.r_low) & (C::N - 1)];
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        cf& x = v[2 * (K1 * NCOL + C1) + H];
        x = INV ? cmulc(x, w) : cmul(x, w);
      });
    });
  };
  auto col_tw = [&]() RPSF_AI {
    StaticFor<1, NCOL>::run([&]<int L1>() RPSF_AI {
      const cf w = tw[(2 * L1 * c_low) & (C::N - 1)];
      StaticFor<0, NR>::run([&]<int K1>() RPSF_AI {
        cf& x = v[2 * (K1 * NCOL + L1) + H];
        x = INV ? cmulc(x, w) : cmul(x, w);
      });
    });
  };
  if constexpr (!INV) {
    fft_axis<C::A1, 2 * NCOL, NCOL, 2, false, H>(v);
    row_tw();
    fft_axis<C::B1, 2, NR, 2 * NCOL, false, H>(v);
    col_tw();
  } else {
    col_tw();
    fft_axis<C::B1, 2, NR, 2 * NCOL, true, H>(v);
    row_tw();
    fft_axis<C::A1, 2 * NCOL, NCOL, 2, true, H>(v);
  }
}

template <class C, int H, bool INV>
RPSF_HD void stage2h(int t, cf* v, const cf* __restrict__ tw) {
  constexpr int NR = 1 << C::A2, NCOL = 1 << C::B2;
  ThreadPos2<C> tp(t);
  auto row_tw = [&]() RPSF_AI {  // W_{2^(A2+AL)}^(k2 r3)
    StaticFor<1, NR>::run([&]<int K2>() RPSF_AI {
      const cf w = tw[((K2 * tp.r3) << C::A1) & (C::N - 1)];
      StaticFor<0, NCOL>::run([&]<int C2>() RPSF_AI {
        cf& x = v[2 * (K2 * NCOL + C2) + H];
        x = INV ? cmulc(x, w) : cmul(x, w);
      });
    });
  };
  auto col_tw = [&]() RPSF_AI {  // W_{2^(B2+1)}^(l2 c3): compile-time, only for c3 = 1
    if constexpr (H == 1) {
      StaticFor<1, NCOL>::run([&]<int L2>() RPSF_AI {
        constexpr int k64 = L2 * (64 >> (C::B2 + 1));  // on the 64-point circle
        constexpr float c = cos64(k64), s = -sin64(k64);
        StaticFor<0, NR>::run([&]<int K2>() RPSF_AI {
          cf& x = v[2 * (K2 * NCOL + L2) + H];
          x = INV ? cmulc(x, cf{c, s}) : cmul(x, cf{c, s});
        });
      });
    }
  };
  if constexpr (!INV) {
    fft_axis<C::A2, 2 * NCOL, NCOL, 2, false, H>(v);
    row_tw();
    fft_axis<C::B2, 2, NR, 2 * NCOL, false, H>(v);
    col_tw();
  } else {
    col_tw();
    fft_axis<C::B2, 2, NR, 2 * NCOL, true, H>(v);
    row_tw();
    fft_axis<C::A2, 2 * NCOL, NCOL, 2, true, H>(v);
  }
}

// ------------------------------------------------------------------------------------------
// LDS exchanges of half H (lds in 8-byte units)
// ------------------------------------------------------------------------------------------
template <class C, int H>
RPSF_HD void x1_write2(int t, const cf* v, cf* lds) {
  ThreadPos2<C> tp(t);
  cf* base = lds + tp.wave * (2 * C::X2_G) + tp.hb * (32 * C::X1_ROWU) + tp.l5;  // (the wave's own two planes of the X2 image, see Cfg2::X2_G)
  StaticFor<0, 32>::run([&]<int J>() RPSF_AI { base[J * C::X1_ROWU] = v[2 * J + H]; });
}
template <class C, int H>
RPSF_HD void x1_read2(int t, cf* v, const cf* lds) {
  ThreadPos2<C> tp(t);
  const cf2* row = reinterpret_cast<const cf2*>(lds + tp.wave * (2 * C::X2_G) + tp.hb * (32 * C::X1_ROWU) + tp.l5 * C::X1_ROWU);
  StaticFor<0, 16>::run([&]<int K>() RPSF_AI {
    const cf2 u = row[K];
    v[2 * (2 * K) + H] = u.a;
    v[2 * (2 * K + 1) + H] = u.b;
  });
}
template <class C, int H>
RPSF_HD void x2_mid_write2(int t, const cf* v, cf* lds) {
  ThreadPos2<C> tp(t);
  cf* base = lds + tp.r3 * C::X2_G + tp.l5;
  StaticFor<0, 32>::run([&]<int J>() RPSF_AI { base[J * C::X2_ROWU] = v[2 * J + H]; });
}
template <class C, int H>
RPSF_HD void x2_mid_read2(int t, cf* v, const cf* lds) {
  ThreadPos2<C> tp(t);
  const cf* base = lds + tp.r3 * C::X2_G + tp.l5;
  StaticFor<0, 32>::run([&]<int J>() RPSF_AI { v[2 * J + H] = base[J * C::X2_ROWU]; });
}
template <class C, int H>
RPSF_HD void x2_last_read2(const GroupIds<C>& gids, cf* v, const cf* lds) {
  StaticFor<0, C::P>::run([&]<int GI>() RPSF_AI {
    const cf* base = lds + x2_unit<C>(gids[GI]);
    StaticFor<0, C::EA>::run([&]<int R3>() RPSF_AI { v[GI * C::E + 2 * R3 + H] = base[R3 * C::X2_G]; });
  });
}
template <class C, int H>
RPSF_HD void x2_last_write2(const GroupIds<C>& gids, const cf* v, cf* lds) {
  StaticFor<0, C::P>::run([&]<int GI>() RPSF_AI {
    cf* base = lds + x2_unit<C>(gids[GI]);
    StaticFor<0, C::EA>::run([&]<int R3>() RPSF_AI { base[R3 * C::X2_G] = v[GI * C::E + 2 * R3 + H]; });
  });
}

// ------------------------------------------------------------------------------------------
// Frequency step
// ------------------------------------------------------------------------------------------
// Modulation of the partner group (member B) of a q = 0 / m = 0 slot, in the (r3, c3) domain: b[r3] *= W_EA^(+-r3)
// shifts its k3 bins by one, negating the c3 = 1 values swaps its l3 bins.  kind is per lane; the select is on values.
// Done per column parity C3, next to the row DFTs of that parity.
template <class C, int S, bool POST, int C3>
RPSF_HD void modulate_partner(int kind, cf* v) {
  cf* zb = v + (2 * S + 1) * C::E;
  const bool q0 = kind == SLOT_Q0, m0 = kind == SLOT_M0;
  StaticFor<0, C::EA>::run([&]<int R3>() RPSF_AI {
    cf& x = zb[2 * R3 + C3];
    if constexpr (R3 > 0) {
      constexpr int k64 = R3 * (64 / C::EA);
      constexpr float c = cos64(k64), s = POST ? sin64(k64) : -sin64(k64);  // W_EA^r3 before, its conjugate after
      x = sel(q0, cmul(x, cf{c, s}), x);
    }
    if constexpr (C3 == 1) x = sel(m0, -x, x);
  });
}

// Last stage in two steps: the DFTs along r3 of the values of one column parity of a slot, and the 2-point DFT along
// c3.  With one slot per thread (N = 256) the row DFTs of a parity need only that half of the exchange, so they run
// while the other half is still moving through LDS (Cfg2::SPLIT_ROWS); with several slots they stay next to the slot's
// pair words.
template <class C, bool INV, int C3, int S>
RPSF_HD void stage3_rows(int t, const GroupIds<C>& gids, cf* v) {
  if constexpr (!INV && S == 0) {
    if (t < 64) modulate_partner<C, 0, false, C3>(slot_kind2<C>(gids[0]), v);  // wave-uniform branch
  }
  fft_axis<C::AL, 2, 1, 1, INV, (2 * S) * C::E + C3>(v);
  fft_axis<C::AL, 2, 1, 1, INV, (2 * S + 1) * C::E + C3>(v);
  if constexpr (INV && S == 0) {
    if (t < 64) modulate_partner<C, 0, true, C3>(slot_kind2<C>(gids[0]), v);
  }
}
template <class C, bool INV, int S>
RPSF_HD void stage3_cols(cf* v) {
  fft_axis<1, 1, C::EA, 2, INV, (2 * S) * C::E>(v);
  fft_axis<1, 1, C::EA, 2, INV, (2 * S + 1) * C::E>(v);
}

// K words: word w of thread t at cf index (w*T + t)*2 - (K'_h(p), K'_h(p + (0,N/2))) for p = bin e of member A of slot
// w / E, e = w % E (every slot, the modulated ones included).
template <class C, int CI, bool NT = true>
RPSF_HD void load_k_chunk2(int t, cf* k, const cf* __restrict__ g) {
  StaticFor<0, C::KCH>::run([&]<int I>() RPSF_AI {
#if defined(RPSF2_ABL_NOK)
    k[2 * I] = cf{1.0f + (float)I, 0.5f};
    k[2 * I + 1] = cf{0.25f, (float)t};
    return;
#endif
    load_k16<NT>(g + ((size_t)(CI * C::KCH + I) * C::T + t) * 2, k[2 * I], k[2 * I + 1]);
  });
}

// Self-paired groups: threads 0 and 1 park their slot 0 (after its forward DFT) ...
template <class C>
RPSF_HD void self_park(int t, const cf* v, cf* park) {
  if (t < 2) StaticFor<0, 2 * C::E>::run([&]<int I>() RPSF_AI { park[t * 2 * C::E + I] = v[I]; });
}
// ... lane `lane` of wave 0 handles pair `round*64 + lane` ...
template <class C>
RPSF_HD void self_orbit(int lane, int round, const uint32_t* __restrict__ ot, cf ka, cf kb, const cf* __restrict__ tw, cf* park) {
  const uint32_t ent = ot[round * 64 + lane];
  const int x1 = ent & 0xff, x2 = (ent >> 8) & 0xff, kc = (ent >> 16) & 0xff;
  const cf z1 = park[x1], z2 = park[x2];
  const PairOut o = pair_op(z1, z2, ka, kb, tw[kc]);
  if (ent >> 31) {
    park[x1] = o.a;
    if (x2 != x1) park[x2] = o.b;
  }
}
// ... and the two threads take the results back before the inverse DFT.
template <class C>
RPSF_HD void self_unpark(int t, cf* v, const cf* park) {
  if (t < 2) StaticFor<0, 2 * C::E>::run([&]<int I>() RPSF_AI { v[I] = park[t * 2 * C::E + I]; });
}

// pair words of slot S held in k[0 .. 2*COUNT): bins E0 .. E0+COUNT-1
template <class C, int S, int E0, int COUNT>
RPSF_HD void pair_words(const GroupIds<C>& gids, cf* v, const cf* k, const cf* __restrict__ tw) {
  cf* za = v + (2 * S) * C::E;
  cf* zb = za + C::E;
  int qa, ma;
  gid_to_qm2<C>(gids[2 * S], qa, ma);
  const cf w0 = tw[ma], w1 = tw[ma + C::M];
  StaticFor<0, COUNT>::run([&]<int I>() RPSF_AI {
    constexpr int EE = E0 + I;
    const PairOut o = pair_op(za[EE], zb[C::E - 1 - EE], k[2 * I], k[2 * I + 1], (EE & 1) ? w1 : w0);
    za[EE] = o.a;
    zb[C::E - 1 - EE] = o.b;
  });
}

// The frequency step of one thread in three calls (the kernel puts wave-level LDS ordering between them; with
// SPLIT_ROWS it also runs the row DFTs of slot 0 itself, around the exchanges):
//   freq_a: (row DFTs and) column DFT of slot 0, self-paired slots parked;
//   self_orbit (wave 0 only, one call per round);
//   freq_b: pair words chunk by chunk (k holds chunk 0 on entry; each later chunk is requested as soon as the
//           buffer is free), the other slots' DFTs around their words, results of the self-paired slots taken back.
template <class C>
RPSF_HD void freq_a(int t, const GroupIds<C>& gids, cf* v, cf* park) {
  if constexpr (!C::SPLIT_ROWS) {
    stage3_rows<C, false, 0, 0>(t, gids, v);
    stage3_rows<C, false, 1, 0>(t, gids, v);
  }
  stage3_cols<C, false, 0>(v);
  if (t < 64) self_park<C>(t, v, park);
}
template <class C, bool NT = true>
RPSF_HD void freq_b(int t, const GroupIds<C>& gids, cf* v, cf* k, const cf* __restrict__ g, const cf* __restrict__ tw,
                    const cf* park) {
  StaticFor<0, C::NCHUNK>::run([&]<int CI>() RPSF_AI {
    constexpr int S = CI * C::KCH / C::E, E0 = CI * C::KCH % C::E;
    if constexpr (E0 == 0 && S > 0) {
      stage3_rows<C, false, 0, S>(t, gids, v);
      stage3_rows<C, false, 1, S>(t, gids, v);
      stage3_cols<C, false, S>(v);
    }
    pair_words<C, S, E0, C::KCH>(gids, v, k, tw);
    if constexpr (CI + 1 < C::NCHUNK) load_k_chunk2<C, CI + 1, NT>(t, k, g);
    if constexpr (E0 + C::KCH == C::E) {
      if constexpr (S == 0) {
        if (t < 64) self_unpark<C>(t, v, park);
      }
      stage3_cols<C, true, S>(v);
      if constexpr (!(C::SPLIT_ROWS && S == 0)) {
        stage3_rows<C, true, 0, S>(t, gids, v);
        stage3_rows<C, true, 1, S>(t, gids, v);
      }
    }
  });
}

// Value of the packed K stream at (thread t, word w, side b)
template <class C, class KF>
RPSF_HD cf pack_value2(const KF& kfull, const uint16_t* __restrict__ tab, int t, int w, int b) {
  const int s = w / C::E, e = w % C::E;
  int q, m;
  gid_to_qm2<C>(tab[(t * C::NSLOT + s) * 2], q, m);
  const int kr = q + C::Q * (e >> 1), kc = m + C::M * (e & 1);
  return kh_at<C>(kfull, kr, b ? kc + C::NC : kc);
}
// ... and of the side array of the self-paired bin pairs (entry i of the orbit table)
template <class C, class KF>
RPSF_HD cf pack_orbit2(const KF& kfull, const uint16_t* __restrict__ tab, const uint32_t* __restrict__ ot, int i, int b) {
  const uint32_t ent = ot[i];
  if (!(ent >> 31)) return cf{0.f, 0.f};
  const int x1 = ent & 0xff, t = x1 / (2 * C::E), member = (x1 / C::E) & 1, e = x1 % C::E;
  int q, m;
  gid_to_qm2<C>(tab[(t * C::NSLOT + 0) * 2 + member], q, m);
  const int kr = q + C::Q * (e >> 1), kc = m + C::M * (e & 1);
  return kh_at<C>(kfull, kr, b ? kc + C::NC : kc);
}

// ------------------------------------------------------------------------------------------
// Image side.  Unit (R1, C1) of a thread: row r = (R1 << (A2+AL)) + r_low, pixels 4c' .. 4c'+3, c' = (C1 << B2) + c2.
// ------------------------------------------------------------------------------------------
template <class C>
RPSF_HD bool patch_inside2(int pr, int pc, int H, int W, int row0, int rows) {
  return pr >= 0 && pc >= 0 && pr + C::ROWS <= H && pc + C::N <= W && pr >= row0 && pr + C::ROWS <= row0 + rows;
}
RPSF_HD bool quads_aligned(const void* base, int ld, int pc) { return ((ld | pc) & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0; }

// Gather in two steps so that a persistent workgroup can request the next patch's pixels while the stores of the
// current one drain: load_raw2 issues the loads (np.pad index maps from LDS for patches that hang over the edge),
// window_patch2 applies the sine window (transform.py:151-155,163) once the values are needed.
// HOT (the specialised persistent kernels, rpsf_kernels2.hpp): the launcher has checked that every unit of a rim patch maps to four
// consecutive image columns or to the fill (hot_geometry in rpsf.hip), so the pixel-by-pixel path is not compiled in.
template <class C, bool HOT = false>
RPSF_HD void load_raw2(int t, cf* v, const ImageView& im, int pr, int pc, bool fast, const int* maps) {
  ThreadPos2<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  if (__builtin_expect(fast, 1)) {
    // (one row multiplication per thread: the rows of a thread's units are a fixed, workgroup-uniform step apart - v_mul_lo_u32 is a quarter-rate
    // instruction, and the compiler otherwise spends one per unit row)
    const float* base = im.img + (size_t)(pr - im.row0) * im.ld + pc + (size_t)tp.r_low * im.ld;
    const size_t rstep = (size_t)im.ld << (C::A2 + C::AL);
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        const int cp = (C1 << C::B2) + tp.c2;
        const f32x4 q = *reinterpret_cast<const f32x4*>(base + R1 * rstep + 4 * cp);
        v[2 * (R1 * NCOL + C1)] = cf{q.x, q.y};
        v[2 * (R1 * NCOL + C1) + 1] = cf{q.z, q.w};
      });
    });
  } else {
    // Rim patches.  A thread's two column units are the same for all its rows: where the np.pad index map sends a unit to
    // four consecutive image columns - ascending (inside the image, 'wrap') or descending ('symmetric' mirrors whole units
    // when the image edge is unit-aligned) - or to the constant fill, the unit is still one 16-byte load (reversed in
    // registers); only units the map tears apart ('reflect', 'edge', widths that are no multiple of 4) go pixel by pixel.
    int xq[NCOL];
    bool rev[NCOL], cst[NCOL];
    bool quad = HOT || quads_aligned(im.img, im.ld, 0);
    StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
      const int* m = maps + C::N + 4 * ((C1 << C::B2) + tp.c2);
      const int m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3];
      const bool up = m0 >= 0 && m1 == m0 + 1 && m2 == m0 + 2 && m3 == m0 + 3 && (m0 & 3) == 0;
      const bool dn = m3 >= 0 && m2 == m3 + 1 && m1 == m3 + 2 && m0 == m3 + 3 && (m3 & 3) == 0;
      cst[C1] = (m0 & m1 & m2 & m3) < 0;  // all four are "constant value"
      rev[C1] = dn;
      xq[C1] = up ? m0 : dn ? m3 : 0;
      quad = HOT || (quad && (up || dn || cst[C1]));
    });
    if (HOT || quad) {
      StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
        const int r = (R1 << (C::A2 + C::AL)) + tp.r_low;
        const int yl = maps[r];
        const float* row = im.img + (size_t)(yl < 0 ? 0 : yl) * im.ld;
        StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
          const f32x4 q = *reinterpret_cast<const f32x4*>(row + xq[C1]);  // always in bounds; select afterwards
          const bool fill = yl < 0 || cst[C1];
          const float pv = im.pad_value;
          v[2 * (R1 * NCOL + C1)] = cf{fill ? pv : rev[C1] ? q.w : q.x, fill ? pv : rev[C1] ? q.z : q.y};
          v[2 * (R1 * NCOL + C1) + 1] = cf{fill ? pv : rev[C1] ? q.y : q.z, fill ? pv : rev[C1] ? q.x : q.w};
        });
      });
      return;
    }
    if constexpr (!HOT)
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      const int r = (R1 << (C::A2 + C::AL)) + tp.r_low;
      const int yl = maps[r];
      const float* row = im.img + (size_t)(yl < 0 ? 0 : yl) * im.ld;
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        const int cp = (C1 << C::B2) + tp.c2;
        float px[4];
        StaticFor<0, 4>::run([&]<int I>() RPSF_AI {
          const int x = maps[C::N + 4 * cp + I];
          const float raw = row[x < 0 ? 0 : x];  // always in bounds; select afterwards
          px[I] = (yl < 0 || x < 0) ? im.pad_value : raw;
        });
        v[2 * (R1 * NCOL + C1)] = cf{px[0], px[1]};
        v[2 * (R1 * NCOL + C1) + 1] = cf{px[2], px[3]};
      });
    });
  }
}
template <class C>
RPSF_HD void window_patch2(int t, cf* v, const float* __restrict__ win) {
  ThreadPos2<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
    const float wr = win[(R1 << (C::A2 + C::AL)) + tp.r_low];
    StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
      const int cp = (C1 << C::B2) + tp.c2;
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(win + 4 * cp);
      cf& a = v[2 * (R1 * NCOL + C1)];
      cf& b = v[2 * (R1 * NCOL + C1) + 1];
      a = cf{a.x * (w4.x * wr), a.y * (w4.y * wr)};
      b = cf{b.x * (w4.z * wr), b.y * (w4.w * wr)};
    });
  });
}
template <class C>
RPSF_HD void load_patch2(int t, cf* v, const ImageView& im, int pr, int pc, const float* __restrict__ win, bool fast, const int* maps) {
  load_raw2<C>(t, v, im, pr, pc, fast, maps);
  window_patch2<C>(t, v, win);
}

// Overlap-add of the finished patch.  qw: the four quadrant words of rpsf_core.hpp (store_patch_direct) or nullptr:
//   nullptr + pv.plane_stride != 0: every pixel into colour plane `plane` (streaming stores);
//   nullptr + pv.plane_stride == 0: float atomics into pv.out (ADD);
//   else: QUAD_DIRECT quadrants into dv.out (accumulating onto what is there when QUAD_ACC is set, read with LOAD4
//   - called as load4<R1, C1>(address) for unit (R1, C1), so that a caller may have prefetched the values - /
//   LOAD1 = L1-bypassing loads), QUAD_SIDE quadrants into the colour plane.
// PSTORE4 / PSTORE1: 16- and 4-byte stores into the colour plane (streaming, or write-through when the plane sum runs
// in the same launch).
// HOT: colour planes whose geometry the launcher has checked (whole units inside or outside the image): interior patches and the
// 16-byte rim path only.
template <class C, bool HOT = false, class ADD, class LOAD4, class LOAD1, class PSTORE4, class PSTORE1>
RPSF_HD void store_patch2(int t, const cf* v, const OutView& pv, const OutView& dv, int plane, int pr, int pc,
                          const float* __restrict__ win, const uint32_t* qw, ADD&& add, LOAD4&& load4, LOAD1&& load1,
                          PSTORE4&& pstore4, PSTORE1&& pstore1) {
  ThreadPos2<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  const bool planes = HOT || pv.plane_stride != 0;
  float* pbase = pv.out + (size_t)plane * pv.plane_stride;
  const bool fast = patch_inside2<C>(pr, pc, pv.H, pv.W, pv.row0, pv.rows) && (HOT || (quads_aligned(pbase, pv.ld, pc) &&
                    (planes || qw) && (!qw || quads_aligned(dv.out, dv.ld, pc))));
  if (__builtin_expect(fast, 1)) {
    float* prow0 = pbase + (size_t)(pr - pv.row0) * pv.ld + pc;
    float* drow0 = qw ? dv.out + (size_t)(pr - dv.row0) * dv.ld + pc : nullptr;
    float* const prow_t = prow0 + (size_t)tp.r_low * pv.ld;             // (one row multiplication per thread, as in load_raw2)
    const size_t pstep = (size_t)pv.ld << (C::A2 + C::AL);
    constexpr int BATCH = 4;  // rows of units whose running sums are in flight together
    StaticFor<0, NR / BATCH>::run([&]<int RB>() RPSF_AI {
      f32x4 old[BATCH * NCOL];
      StaticFor<0, BATCH * NCOL>::run([&]<int U>() RPSF_AI {
        constexpr int R1 = RB * BATCH + U / NCOL, C1 = U % NCOL, QD = 2 * (R1 >= NR / 2) + (C1 >= NCOL / 2);
        old[U] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qw && quad_mode(qw[QD]) == QUAD_DIRECT && (qw[QD] & QUAD_ACC)) {
          const int r = (R1 << (C::A2 + C::AL)) + tp.r_low, cp = (C1 << C::B2) + tp.c2;
          old[U] = load4.template operator()<R1, C1>(drow0 + (size_t)r * dv.ld + 4 * cp);
        }
      });
      StaticFor<0, BATCH * NCOL>::run([&]<int U>() RPSF_AI {
        constexpr int R1 = RB * BATCH + U / NCOL, C1 = U % NCOL, QD = 2 * (R1 >= NR / 2) + (C1 >= NCOL / 2);
        const int r = (R1 << (C::A2 + C::AL)) + tp.r_low, cp = (C1 << C::B2) + tp.c2;
        const float wr = win[r];
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(win + 4 * cp);
        const cf a = v[2 * (R1 * NCOL + C1)], b = v[2 * (R1 * NCOL + C1) + 1];
        f32x4 val = {a.x * (w4.x * wr), a.y * (w4.y * wr), b.x * (w4.z * wr), b.y * (w4.w * wr)};
        if (!qw || quad_mode(qw[QD]) == QUAD_SIDE) {
#if defined(RPSF2_SKEL_PRESUM)
          if constexpr (C1 >= NCOL / 2) (void)val;  // (not stored - and not kept alive: the register file of these kernels has no room for it)
          else
#endif
          pstore4(prow_t + R1 * pstep + 4 * cp, val);
        } else if (quad_mode(qw[QD]) == QUAD_DIRECT) {
          val += old[U];
          *reinterpret_cast<f32x4*>(drow0 + (size_t)r * dv.ld + 4 * cp) = val;
        }
      });
    });
    return;
  }
  if (HOT || (planes && !qw && quads_aligned(pbase, pv.ld, pc) && (pv.W & 3) == 0)) {
    // Rim patches in plane mode: with the corner column and the image width multiples of 4 a unit lies wholly inside or
    // wholly outside the image, so the part of the patch that is inside still goes out in 16-byte stores (and a 128-byte
    // line is still written whole by one store instruction whenever the image edges are line-aligned, which is what
    // the fused plane sum asks of its producers).
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      const int r = (R1 << (C::A2 + C::AL)) + tp.r_low;
      const float wr = win[r];
      const int y = pr + r, yl = y - pv.row0;
      const bool row_ok = y >= 0 && y < pv.H && yl >= 0 && yl < pv.rows;
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        const int cp = (C1 << C::B2) + tp.c2, x = pc + 4 * cp;
#if defined(RPSF2_SKEL_PRESUM)
        if (C1 < NCOL / 2)
#endif
        if (row_ok && x >= 0 && x + 4 <= pv.W) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(win + 4 * cp);
          const cf a = v[2 * (R1 * NCOL + C1)], b = v[2 * (R1 * NCOL + C1) + 1];
          pstore4(pbase + (size_t)yl * pv.ld + x, f32x4{a.x * (w4.x * wr), a.y * (w4.y * wr), b.x * (w4.z * wr), b.y * (w4.w * wr)});
        }
      });
    });
    return;
  }
  if constexpr (!HOT)
  StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
    const int r = (R1 << (C::A2 + C::AL)) + tp.r_low;
    const float wr = win[r];
    const int y = pr + r, yl = y - pv.row0;
    if (y >= 0 && y < pv.H && yl >= 0 && yl < pv.rows) {
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        constexpr int QD = 2 * (R1 >= NR / 2) + (C1 >= NCOL / 2);
        const int cp = (C1 << C::B2) + tp.c2;
        const cf a = v[2 * (R1 * NCOL + C1)], b = v[2 * (R1 * NCOL + C1) + 1];
        const float px[4] = {a.x, a.y, b.x, b.y};
        StaticFor<0, 4>::run([&]<int I>() RPSF_AI {
          const int x = pc + 4 * cp + I;
          if (x < 0 || x >= pv.W) return;
          const float val = px[I] * (wr * win[4 * cp + I]);
          if (!qw) {
            float* dst = pbase + (size_t)yl * pv.ld + x;
            if (planes) pstore1(dst, val); else add(dst, val);
          } else if (quad_mode(qw[QD]) == QUAD_DIRECT) {
            float* dst = dv.out + (size_t)(y - dv.row0) * dv.ld + x;
            *dst = ((qw[QD] & QUAD_ACC) ? load1(dst) : 0.f) + val;
          } else if (quad_mode(qw[QD]) == QUAD_SIDE) {
            pstore1(pbase + (size_t)yl * pv.ld + x, val);
          }
        });
      });
    }
  });
}

// Plans compiled into the library
using Cfg256v2 = Cfg2<8, 4, 0, 4, 1, 5>;
using Cfg128v2 = Cfg2<7, 4, 1, 2, 1, 4>;

}  // namespace rpsf
