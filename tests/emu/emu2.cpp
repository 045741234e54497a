// CPU thread emulator for rpsf_core2.hpp (test infrastructure, never shipped in the product path): the second-generation
// three-stage plans driven thread by thread, with explicit phase boundaries where the kernel has barriers or
// wave-level LDS ordering.  Checks the index algebra (digits, LDS addresses, slot table, modulated and self-paired
// slots, packed-K format, quadrant stores) against the oracle without a GPU.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../regularizepsf_amd/csrc/rpsf_core2.hpp"

using namespace rpsf;

// direct: 0 = atomics (plain adds here), 1 = direct with the quadrant words below, 2 = colour planes + sum
template <class C>
static int emu2_apply_t(int n_patches, const int32_t* coords, int H, int W, int pad_mode, float pad_value,
                        const float* img, const float* kfull, float* out, int direct) {
  constexpr int T = C::T, N = C::N;
  std::vector<cf> tw(N);
  std::vector<float> win(N);
  for (int k = 0; k < N; ++k) {
    double a = -2.0 * M_PI * k / N;
    tw[k] = cf{(float)std::cos(a), (float)std::sin(a)};
    win[k] = (float)std::sin((k + 0.5) * (M_PI / N));
  }
  std::vector<uint16_t> tab((size_t)T * C::NSLOT * 2);
  build_slot_table2<C>(tab.data());
  if (special_slots2<C>() > 64) return -4;
  std::vector<uint32_t> ot(C::ORBIT_ROUNDS * 64);
  if (build_orbit_table2<C>(tab.data(), ot.data()) != C::NORBIT) return -3;
  std::vector<cf> regs((size_t)T * 64);
  std::vector<cf> lds(C::LDS_UNITS);
  std::vector<cf> g((size_t)C::G_PER_PATCH), gs((size_t)C::GS_PER_PATCH);
  std::vector<GroupIds<C>> gids(T);
  for (int t = 0; t < T; ++t) gids[t].load(tab.data(), t);
  ImageView im{img, H, W, W, pad_mode, pad_value, 0, H};
  std::vector<float> sink(128);
  OutView ov{out, H, W, W, 0, H, 0, sink.data()};
  memset(out, 0, sizeof(float) * (size_t)H * W);
  // colour planes: 16-byte aligned like the device buffer (the 16-byte rim stores need it)
  std::vector<float> plane_mem(direct == 2 ? (size_t)4 * H * W + 4 : 0, 0.f);
  float* planes = plane_mem.data();
  while (reinterpret_cast<uintptr_t>(planes) & 15) ++planes;
  OutView pv{planes, H, W, W, 0, H, (size_t)H * W, sink.data()};
  auto add = [](float* p, float v) { *p += v; };
  auto load4 = []<int R1, int C1>(const float* p) { return *reinterpret_cast<const f32x4*>(p); };
  auto load1 = [](const float* p) { return *p; };
  auto pstore4 = [](float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; };
  auto pstore1 = [](float* p, float v) { *p = v; };
  std::vector<int> maps(2 * N);
  cf* park = lds.data() + C::BUF_UNITS;
  std::vector<uint8_t> touched;  // direct mode: which quadrant tiles of the output have been initialised (by N/2 tiles from the first corner)
  for (int p = 0; p < n_patches; ++p) {
    const cf* kf = reinterpret_cast<const cf*>(kfull) + (size_t)p * N * N;
    for (int t = 0; t < T; ++t)
      for (int w = 0; w < C::NWORDS; ++w)
        for (int b = 0; b < 2; ++b) g[((size_t)w * T + t) * 2 + b] = pack_value2<C>(kf, tab.data(), t, w, b);
    for (int i = 0; i < C::ORBIT_ROUNDS * 64; ++i)
      for (int b = 0; b < 2; ++b) gs[(size_t)i * 2 + b] = pack_orbit2<C>(kf, tab.data(), ot.data(), i, b);
    int pr = coords[2 * p], pc = coords[2 * p + 1];
    const bool fast = patch_inside<C>(pr, pc, H, W, 0, H) && quads_aligned(img, W, pc);
    if (!fast)
      for (int t = 0; t < T; ++t) build_pad_maps<C>(t, maps.data(), im, pr, pc);
    auto R = [&](int t) { return &regs[(size_t)t * 64]; };
    for (int t = 0; t < T; ++t) load_patch2<C>(t, R(t), im, pr, pc, win.data(), fast, maps.data());
    for (int t = 0; t < T; ++t) stage1h<C, 0, false>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x1_write2<C, 0>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) stage1h<C, 1, false>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x1_read2<C, 0>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) x1_write2<C, 1>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) stage2h<C, 0, false>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x1_read2<C, 1>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) x2_mid_write2<C, 0>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) stage2h<C, 1, false>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x2_last_read2<C, 0>(gids[t], R(t), lds.data());
    for (int t = 0; t < T; ++t) x2_mid_write2<C, 1>(t, R(t), lds.data());
    if constexpr (C::SPLIT_ROWS)
      for (int t = 0; t < T; ++t) stage3_rows<C, false, 0, 0>(t, gids[t], R(t));
    for (int t = 0; t < T; ++t) x2_last_read2<C, 1>(gids[t], R(t), lds.data());
    if constexpr (C::SPLIT_ROWS)
      for (int t = 0; t < T; ++t) stage3_rows<C, false, 1, 0>(t, gids[t], R(t));
    for (int t = 0; t < T; ++t) freq_a<C>(t, gids[t], R(t), park);
    for (int round = 0; round < C::ORBIT_ROUNDS; ++round)
      for (int lane = 0; lane < 64; ++lane) {
        const cf* kw = gs.data() + (size_t)(round * 64 + lane) * 2;
        self_orbit<C>(lane, round, ot.data(), kw[0], kw[1], tw.data(), park);
      }
    for (int t = 0; t < T; ++t) {
      cf k[2 * C::KCH];
      load_k_chunk2<C, 0>(t, k, g.data());
      freq_b<C>(t, gids[t], R(t), k, g.data(), tw.data(), park);
      if constexpr (C::SPLIT_ROWS) stage3_rows<C, true, 0, 0>(t, gids[t], R(t));
    }
    for (int t = 0; t < T; ++t) x2_last_write2<C, 0>(gids[t], R(t), lds.data());
    if constexpr (C::SPLIT_ROWS)
      for (int t = 0; t < T; ++t) stage3_rows<C, true, 1, 0>(t, gids[t], R(t));
    for (int t = 0; t < T; ++t) x2_mid_read2<C, 0>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) x2_last_write2<C, 1>(gids[t], R(t), lds.data());
    for (int t = 0; t < T; ++t) stage2h<C, 0, true>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x2_mid_read2<C, 1>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) x1_write2<C, 0>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) stage2h<C, 1, true>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x1_read2<C, 0>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) x1_write2<C, 1>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) stage1h<C, 0, true>(t, R(t), tw.data());
    for (int t = 0; t < T; ++t) x1_read2<C, 1>(t, R(t), lds.data());
    for (int t = 0; t < T; ++t) stage1h<C, 1, true>(t, R(t), tw.data());
    if (!direct) {
      for (int t = 0; t < T; ++t) store_patch2<C>(t, R(t), ov, ov, 0, pr, pc, win.data(), nullptr, add, load4, load1, pstore4, pstore1);
    } else if (direct == 2) {  // the plane of the patch's lattice parity (the caller passes a lattice with corners at multiples of N/2)
      const int half = N / 2, plane = 2 * (((pr + 8 * N) / half) & 1) + (((pc + 8 * N) / half) & 1);
      for (int t = 0; t < T; ++t) store_patch2<C>(t, R(t), pv, pv, plane, pr, pc, win.data(), nullptr, add, load4, load1, pstore4, pstore1);
    } else {
      // direct stores, sequential: the first patch over a tile stores, later ones accumulate (the flags' job on the GPU);
      // tiles are indexed from the first patch corner (the caller passes a lattice)
      const int half = N / 2, r0 = coords[0] - 4 * N, c0 = coords[1] - 4 * N, ntj = (W + 8 * N) / half + 2;
      if (touched.empty()) touched.assign((size_t)((H + 8 * N) / half + 2) * ntj, 0);
      uint32_t qw[4];
      for (int q = 0; q < 4; ++q) {
        const int ti = (pr - r0) / half + (q >> 1), tj = (pc - c0) / half + (q & 1);
        uint8_t& seen = touched[(size_t)ti * ntj + tj];
        qw[q] = quad_word(QUAD_DIRECT, 0, 0) | (seen ? QUAD_ACC : 0u);
        seen = 1;
      }
      for (int t = 0; t < T; ++t) store_patch2<C>(t, R(t), ov, ov, 0, pr, pc, win.data(), qw, add, load4, load1, pstore4, pstore1);
    }
  }
  if (direct == 2)
    for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = ((planes[i] + planes[i + pv.plane_stride]) + planes[i + 2 * pv.plane_stride]) + planes[i + 3 * pv.plane_stride];
  return 0;
}

extern "C" int emu2_apply(int N, int n_patches, const int32_t* coords, int H, int W, int pad_mode, float pad_value,
                          const float* img, const float* kfull, float* out, int direct) {
  switch (N) {
    case 256: return emu2_apply_t<Cfg256v2>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out, direct);
    case 128: return emu2_apply_t<Cfg128v2>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out, direct);
    default: return -1;
  }
}

// The slot table the plan uploads and the X2 image's geometry, for the bank-conflict count of tests/test_emulator.py:
// tab[(t*nslot + s)*2 + member] = gid (2 * threads * nslot entries), units[gid] = 8-byte unit of the group inside a plane of the image.
template <class C>
static int emu2_slot_table_t(uint16_t* tab, int32_t* units, int* threads, int* nslot, int* dealt) {
  build_slot_table2<C>(tab);
  for (int g = 0; g < C::G; ++g) units[g] = x2_unit<C>(g);
  std::vector<std::pair<int, int>> slots;
  *threads = C::T, *nslot = C::NSLOT, *dealt = deal_slots2<C>(slots) ? 1 : 0;
  return 0;
}
extern "C" int emu2_slot_table(int N, uint16_t* tab, int32_t* units, int* threads, int* nslot, int* dealt) {
  switch (N) {
    case 256: return emu2_slot_table_t<Cfg256v2>(tab, units, threads, nslot, dealt);
    case 128: return emu2_slot_table_t<Cfg128v2>(tab, units, threads, nslot, dealt);
    default: return -1;
  }
}
