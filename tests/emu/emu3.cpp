// CPU lane emulator for rpsf_core3.hpp + rpsf_plan3.hpp (test infrastructure, never shipped in the product path).
// Runs the per-lane phases of the third-generation (sweep) kernel lane by lane, with a phase boundary wherever the wave
// exchanges data through LDS, over the very job lists the library builds - so the index algebra (transposes, packed-K
// format, ring addressing, store / add / flush rules, dependency lists) is checked against the oracle without a GPU.
// order_seed != 0: the jobs of a region run in a random order that respects their dependency lists, as the waves of a
// workgroup may run them; the result must not change by a bit.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../regularizepsf_amd/csrc/rpsf_core3.hpp"
#include "../../regularizepsf_amd/csrc/rpsf_plan3.hpp"

using namespace rpsf;

template <class C>
static int emu3_apply_t(int n_patches, const int32_t* coords, int Himg, int Wimg, int pad_mode, float pad_value, const float* img,
                        const float* kfull, float* out, int target_regions, unsigned order_seed, int aligned, int64_t* stats) {
  constexpr int N = C::N, H = C::H;
  // ---- lattice (as setup_lattice in rpsf.hip) ----
  int r0 = coords[0], c0 = coords[1], r1 = r0, c1 = c0;
  for (int i = 0; i < n_patches; ++i) {
    r0 = std::min(r0, coords[2 * i]), r1 = std::max(r1, coords[2 * i]);
    c0 = std::min(c0, coords[2 * i + 1]), c1 = std::max(c1, coords[2 * i + 1]);
  }
  const int nli = (r1 - r0) / H + 1, nlj = (c1 - c0) / H + 1;
  if ((long)nli * nlj != n_patches) return -2;
  std::vector<int32_t> cell((size_t)nli * nlj, -1);
  for (int i = 0; i < n_patches; ++i) {
    if ((coords[2 * i] - r0) % H || (coords[2 * i + 1] - c0) % H) return -2;
    cell[(size_t)((coords[2 * i] - r0) / H) * nlj + (coords[2 * i + 1] - c0) / H] = i;
  }
  for (int32_t x : cell)
    if (x < 0) return -2;
  Plan3 plan;
  if (!plan3_build(N, C::KSMAX, C::WAVES, nli, nlj, cell.data(), 0, target_regions, plan)) return -3;
  if (stats) stats[0] = (int64_t)plan.regions.size(), stats[1] = (int64_t)plan.jobs.size(), stats[2] = plan.patch_slots, stats[3] = plan.ks;
  // ---- packed K ----
  std::vector<float> k3((size_t)n_patches * C::K_FLOATS);
  for (int p = 0; p < n_patches; ++p) {
    const cf* kf = reinterpret_cast<const cf*>(kfull) + (size_t)p * N * N;
    for (int i = 0; i < C::K_FLOATS / 2; ++i) {
      const cf x = pack_value3<C>(kf, i);
      k3[(size_t)p * C::K_FLOATS + 2 * i] = x.x, k3[(size_t)p * C::K_FLOATS + 2 * i + 1] = x.y;
    }
  }
  std::vector<float> win(N);
  for (int k = 0; k < N; ++k) win[k] = (float)std::sin((k + 0.5) * (M_PI / N));
  alignas(16) static const float zeros[4] = {0, 0, 0, 0};
  ImageView im{img, Himg, Wimg, Wimg, pad_mode, pad_value, 0, Himg};
  Flush3 fl{out, Wimg, 0, Himg, Himg, Wimg, aligned};
  // every pixel must be written exactly once: count the stores
  std::vector<uint8_t> written((size_t)Himg * Wimg, 0);
  int double_writes = 0;
  auto st1 = [&](float* dst, float x) {
    if (written[dst - out]++) ++double_writes;
    *dst = x;
  };
  auto st4 = [&](float* dst, f32x4 x) {
    st1(dst, x.x), st1(dst + 1, x.y), st1(dst + 2, x.z), st1(dst + 3, x.w);
  };
  std::vector<float> ring(C::RINGF), xb(C::XF);
  std::vector<f32x4> g((size_t)64 * H);
  std::vector<cf> v((size_t)64 * N);
  std::mt19937 rng(order_seed);
  for (const Region3& reg : plan.regions) {
    // garbage in the ring: nothing may depend on what a previous region left there
    for (float& x : ring) x = std::nanf("");
    // ---- order ----
    std::vector<int> order;
    if (order_seed == 0) {
      for (int j = 0; j < reg.njobs; ++j) order.push_back(j);
    } else {
      std::vector<char> done(reg.njobs, 0);
      std::vector<int> ready;
      while ((int)order.size() < reg.njobs) {
        ready.clear();
        // like the workgroup: jobs are DRAWN in list order (a window of `WAVES` jobs in flight), and finish in any order the flags allow
        int inflight = 0;
        for (int j = 0; j < reg.njobs && inflight < C::WAVES; ++j) {
          if (done[j]) continue;
          ++inflight;
          const Job3& d = plan.jobs[reg.job0 + j];
          if ((d.dep0 < 0 || done[d.dep0]) && (d.dep1 < 0 || done[d.dep1])) ready.push_back(j);
        }
        if (ready.empty()) return -4;  // deadlock: a job in flight waits for one that has not been drawn
        const int j = ready[rng() % ready.size()];
        done[j] = 1, order.push_back(j);
      }
    }
    for (int j : order) {
      const Job3& d = plan.jobs[reg.job0 + j];
      if (d.dep0 >= j || d.dep1 >= j) return -5;
      const int row0 = r0 + d.row, col0 = c0 + d.col;
      const bool fast = aligned && row0 >= 0 && row0 + N <= Himg && col0 >= 0 && col0 + C::SLABW <= Wimg;
      auto lanes = [&](auto&& f) {
        for (int l = 0; l < 64; ++l) f(l, &g[(size_t)l * H], &v[(size_t)l * N]);
      };
      if constexpr (C::DIRECT_GATHER) {  // the lanes read their own rows (no first transpose)
        lanes([&](int l, f32x4*, cf* vl) {
          if (fast) r3_load_fast<C, 0>(l, vl, img + (size_t)row0 * Wimg + col0, Wimg), r3_load_fast<C, 1>(l, vl, img + (size_t)row0 * Wimg + col0, Wimg);
          else r3_load_generic<C, 0>(l, vl, im, row0, col0), r3_load_generic<C, 1>(l, vl, im, row0, col0);
        });
      } else {
        for (int l = 0; l < 64; ++l) {
          if (fast) g3_load_fast<C>(l, &g[(size_t)l * H], img + (size_t)row0 * Wimg + col0, Wimg);
          else g3_load_generic<C>(l, &g[(size_t)l * H], im, row0, col0);
        }
        StaticFor<0, C::NSUB>::run([&]<int S>() {
          lanes([&](int l, f32x4* gl, cf*) { t0_write<C, 0, S>(l, gl, xb.data()); });
          lanes([&](int l, f32x4*, cf* vl) { t0_read<C, 0, S>(l, vl, xb.data()); });
          lanes([&](int l, f32x4* gl, cf*) { t0_write<C, 1, S>(l, gl, xb.data()); });
          lanes([&](int l, f32x4*, cf* vl) { t0_read<C, 1, S>(l, vl, xb.data()); });
        });
      }
      lanes([&](int l, f32x4*, cf* vl) {
        window_in_fft_rows<C>(vl, win[l % H], win[l % H + H]);
        unpack_rows<C>(vl);
      });
      StaticFor<0, C::NSUB>::run([&]<int S>() {
        lanes([&](int l, f32x4*, cf* vl) { t1_write<C, 0, S>(l, vl, xb.data()); });
        lanes([&](int l, f32x4*, cf* vl) { t1_read<C, 0, S>(l, vl, xb.data()); });
        lanes([&](int l, f32x4*, cf* vl) { t1_write<C, 1, S>(l, vl, xb.data()); });
        lanes([&](int l, f32x4*, cf* vl) { t1_read<C, 1, S>(l, vl, xb.data()); });
      });
      lanes([&](int l, f32x4*, cf* vl) {
        const int q = l / H, p = l % H;
        FftSmall<C::LOGN, false>::run(vl);
        const float* kp = k3.data() + (size_t)d.kslot[q] * C::K_FLOATS;
        f32x4 pre[C::KPRE > 0 ? C::KPRE : 1];
        for (int jw = 0; jw < C::KPRE; ++jw) pre[jw] = *reinterpret_cast<const f32x4*>(kp + (size_t)(jw * H + p) * 4);
        kmul3<C, false>(vl, pre, kp + p * 4, p == 0 ? kp + C::KA_FLOATS : zeros, p == 0 ? 4 : 0);
        FftSmall<C::LOGN, true>::run(vl);
      });
      StaticFor<0, C::NSUB>::run([&]<int S>() {
        lanes([&](int l, f32x4*, cf* vl) { t2_write<C, 0, S>(l, vl, xb.data()); });
        lanes([&](int l, f32x4*, cf* vl) { t2_read<C, 0, S>(l, vl, xb.data()); });
        lanes([&](int l, f32x4*, cf* vl) { t2_write<C, 1, S>(l, vl, xb.data()); });
        lanes([&](int l, f32x4*, cf* vl) { t2_read<C, 1, S>(l, vl, xb.data()); });
      });
      const int hs = (d.flags & J3_RING_HALF) ? 1 : 0;
      lanes([&](int l, f32x4*, cf* vl) {
        const int q = l / H, p = l % H;
        repack_rows<C>(vl);
        FftSmall<C::LOGN, true>::run(vl);
        const bool valid = ((d.flags >> (J3_VALID_SHIFT + q)) & 1u) != 0;
        if (d.ring_col < 0 || d.ring_col + C::SLABW > C::RINGW) std::abort();
        float* ru = ring.data() + (hs * H + p) * C::RP + d.ring_col + q * N;
        float* rl = ring.data() + ((hs ^ 1) * H + p) * C::RP + d.ring_col + q * N;
        window_out<C>(vl, win[p], win[p + H], valid);
        accumulate3<C>(vl, ru, rl, (int)((d.flags >> J3_UPPER_SHIFT) & 3u), (int)((d.flags >> J3_LOWER_SHIFT) & 3u));
      });
      const int oc0 = c0 + d.own_c0, oc1 = c0 + d.own_c1;
      if (d.flags & J3_FLUSH_UPPER)
        for (int l = 0; l < 64; ++l) flush3<C>(l, ring.data() + (hs * H) * C::RP + d.ring_col, fl, row0, col0, oc0, oc1, st4, st1);
      if (d.flags & J3_FLUSH_LOWER)
        for (int l = 0; l < 64; ++l) flush3<C>(l, ring.data() + ((hs ^ 1) * H) * C::RP + d.ring_col, fl, row0 + H, col0, oc0, oc1, st4, st1);
    }
  }
  if (double_writes) return -6;
  // pixels the lattice covers must all have been written
  int missing = 0;
  for (int y = 0; y < Himg; ++y)
    for (int x = 0; x < Wimg; ++x) {
      const bool covered = y >= r0 && y < r0 + (nli + 1) * H && x >= c0 && x < c0 + (nlj + 1) * H;
      if (covered && !written[(size_t)y * Wimg + x]) ++missing;
      if (!covered && !written[(size_t)y * Wimg + x]) out[(size_t)y * Wimg + x] = 0.0f;  // (the launcher clears these)
    }
  return missing ? -7 : 0;
}

// Independent check of the job lists: any two jobs of a region that touch the same ring words inside the owned columns, one of them writing,
// must be ordered by the (transitive) dependency lists - whatever the waves' timing.  Returns the number of unordered conflicting pairs.
extern "C" long emu3_check_plan(int N, int ksmax, int waves, int nli, int nlj, int target_regions, int64_t* stats) {
  const int H = N / 2;
  std::vector<int32_t> cell((size_t)nli * nlj);
  for (size_t i = 0; i < cell.size(); ++i) cell[i] = (int32_t)i;
  Plan3 plan;
  if (!plan3_build(N, ksmax, waves, nli, nlj, cell.data(), 0, target_regions, plan)) return -1;
  if (stats) stats[0] = (int64_t)plan.regions.size(), stats[1] = (int64_t)plan.jobs.size(), stats[2] = plan.patch_slots, stats[3] = plan.ks;
  long bad = 0;
  // coverage: every band cell (row band, column band) of the lattice is flushed by exactly one job
  std::vector<int> flushed((size_t)(nli + 1) * (nlj + 1), 0);
  for (const Region3& reg : plan.regions) {
    const int n = reg.njobs;
    std::vector<std::vector<uint64_t>> anc(n, std::vector<uint64_t>((n + 63) / 64, 0));
    for (int j = 0; j < n; ++j) {
      const Job3& d = plan.jobs[reg.job0 + j];
      for (int dep : {d.dep0, d.dep1}) {
        if (dep < 0) continue;
        if (dep >= j) return -2;
        anc[j][dep / 64] |= 1ull << (dep % 64);
        for (size_t w = 0; w < anc[j].size(); ++w) anc[j][w] |= anc[dep][w];
      }
    }
    struct Rect {
      int half, c0, c1;
      bool write;
    };
    auto rects = [&](const Job3& d, std::vector<Rect>& out) {
      out.clear();
      const int org = d.col - d.ring_col;  // lattice column of ring column 0
      const int lo = std::max(d.ring_col, d.own_c0 - org), hi = std::min(d.ring_col + 128, d.own_c1 - org);
      if (lo >= hi) return;
      const int hs = (d.flags & J3_RING_HALF) ? 1 : 0;
      if ((d.flags >> J3_UPPER_SHIFT) & 3u) out.push_back({hs, lo, hi, true});
      if ((d.flags >> J3_LOWER_SHIFT) & 3u) out.push_back({hs ^ 1, lo, hi, true});
      if (d.flags & J3_FLUSH_UPPER) out.push_back({hs, lo, hi, false});
      if (d.flags & J3_FLUSH_LOWER) out.push_back({hs ^ 1, lo, hi, false});
    };
    std::vector<std::vector<Rect>> all(n);
    for (int j = 0; j < n; ++j) rects(plan.jobs[reg.job0 + j], all[j]);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < j; ++i) {
        if ((anc[j][i / 64] >> (i % 64)) & 1) continue;
        for (const Rect& a : all[i])
          for (const Rect& b : all[j])
            if (a.half == b.half && a.c0 < b.c1 && b.c0 < a.c1 && (a.write || b.write)) ++bad;
      }
    for (int j = 0; j < n; ++j) {
      const Job3& d = plan.jobs[reg.job0 + j];
      for (int which = 0; which < 2; ++which) {
        if (!(d.flags & (which ? J3_FLUSH_LOWER : J3_FLUSH_UPPER))) continue;
        const int band_r = d.row / H + which;
        const int lo = std::max(d.col, d.own_c0), hi = std::min(d.col + 128, d.own_c1);
        for (int c = lo; c < hi; c += H) {
          if (c / H < 0 || c / H > nlj) return -3;
          ++flushed[(size_t)band_r * (nlj + 1) + c / H];
        }
      }
    }
  }
  for (int x : flushed)
    if (x != 1) ++bad;
  return bad;
}

extern "C" int emu3_apply(int N, int n_patches, const int32_t* coords, int H, int W, int pad_mode, float pad_value, const float* img,
                          const float* kfull, float* out, int target_regions, unsigned order_seed, int aligned, int64_t* stats) {
  switch (N) {
    case 64: return emu3_apply_t<Cfg3_64>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out, target_regions, order_seed, aligned, stats);
    case 32: return emu3_apply_t<Cfg3_32>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out, target_regions, order_seed, aligned, stats);
    case 16: return emu3_apply_t<Cfg3_16>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out, target_regions, order_seed, aligned, stats);
    default: return -1;
  }
}
