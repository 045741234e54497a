// CPU thread emulator for rpsf_core.hpp (test infrastructure, never shipped in the product path).
// Runs the exact per-thread phases the HIP kernel runs, one "thread" at a time with explicit
// phase boundaries where the kernel has barriers, so the index algebra (digit layouts, LDS
// addressing, slot table, packed-K format) can be checked against the oracle without a GPU.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../regularizepsf_amd/csrc/rpsf_core.hpp"

using namespace rpsf;

template <class C>
static int emu_apply_t(int n_patches, const int32_t* coords, int H, int W, int pad_mode, float pad_value,
                       const float* img, const float* kfull, float* out) {
  constexpr int T = C::T, N = C::N;
  std::vector<cf> tw(N);
  std::vector<float> win(N);
  for (int k = 0; k < N; ++k) {
    double a = -2.0 * M_PI * k / N;
    tw[k] = cf{(float)std::cos(a), (float)std::sin(a)};
    win[k] = (float)std::sin((k + 0.5) * (M_PI / N));
  }
  std::vector<uint16_t> tab((size_t)T * C::NSLOT * 2);
  build_slot_table<C>(tab.data());
  std::vector<uint32_t> pt((size_t)C::PT_WORDS + 1);
  if (build_pair_table<C>(tab.data(), pt.data()) > C::NP) return -3;
  std::vector<cf> regs((size_t)T * 64);
  std::vector<float> lds(C::LDS_FLOATS);
  std::vector<cf> g((size_t)C::G_PER_PATCH), gs((size_t)C::GS_PER_PATCH + 1);
  std::vector<GroupIds<C>> gids(T);
  for (int t = 0; t < T; ++t) gids[t].load(tab.data(), t);
  ImageView im{img, H, W, W, pad_mode, pad_value, 0, H};
  std::vector<float> sink(128);
  OutView ov{out, H, W, W, 0, H, 0, sink.data()};
  memset(out, 0, sizeof(float) * (size_t)H * W);
  auto add = [](float* p, float v) { *p += v; };
  for (int p = 0; p < n_patches; ++p) {
    const cf* kf = reinterpret_cast<const cf*>(kfull) + (size_t)p * N * N;
    // pack K for this patch (what the pack kernel does)
    for (int t = 0; t < T; ++t)
      for (int rho = 0; rho < 2 * C::NWORDS; ++rho) {
        g[((size_t)(rho / 2) * T + t) * 2 + (rho & 1)] = pack_value<C>(kf, tab.data(), pt.data(), t, rho, 0);
        if constexpr (!C::INLINE_GS) {
          const int w = rho >> 1, b = rho & 1, s = w / C::E, e = w % C::E;
          if (slot_is_special<C>(s, t))
            gs[(size_t)C::spec_prefix(s) * 2 * C::E + ((size_t)e * C::spec_t(s) + t) * 2 + b] =
                pack_value<C>(kf, tab.data(), pt.data(), t, rho, 1);
        }
      }
    int pr = coords[2 * p], pc = coords[2 * p + 1];
    const bool fast = patch_inside<C>(pr, pc, H, W, 0, H) && pairs_aligned(img, W, pc);
    int* maps = reinterpret_cast<int*>(lds.data());
    if (!fast)
      for (int t = 0; t < T; ++t) build_pad_maps<C>(t, maps, im, pr, pc);
    for (int t = 0; t < T; ++t) load_patch<C>(t, &regs[(size_t)t * 64], im, pr, pc, win.data(), fast, maps);
    for (int t = 0; t < T; ++t) stage1<C, false>(t, &regs[(size_t)t * 64], tw.data());
    if constexpr (C::S3) {
      for (int t = 0; t < T; ++t) x1_write<C, 0>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) x1_read<C, 0>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) x1_write<C, 1>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) x1_read<C, 1>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) stage2<C, false>(t, &regs[(size_t)t * 64], tw.data());
    }
    for (int t = 0; t < T; ++t) x2_mid_write<C, 0>(t, &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) x2_last_read<C, 0>(gids[t], &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) x2_mid_write<C, 1>(t, &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) x2_last_read<C, 1>(gids[t], &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) {
      cf* v = &regs[(size_t)t * 64];
      KRing<C> kring;
      kring_fill<C>(t, kring, g.data());
      freq_step<C>(t, gids[t], v, kring, g.data(), gs.data(), tw.data(), reinterpret_cast<cf*>(lds.data() + C::PARK_OFFSET), pt.data());
    }
    for (int t = 0; t < T; ++t) x2_last_write<C, 0>(gids[t], &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) x2_mid_read<C, 0>(t, &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) x2_last_write<C, 1>(gids[t], &regs[(size_t)t * 64], lds.data());
    for (int t = 0; t < T; ++t) x2_mid_read<C, 1>(t, &regs[(size_t)t * 64], lds.data());
    if constexpr (C::S3) {
      for (int t = 0; t < T; ++t) stage2<C, true>(t, &regs[(size_t)t * 64], tw.data());
      for (int t = 0; t < T; ++t) x1_write<C, 0>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) x1_read<C, 0>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) x1_write<C, 1>(t, &regs[(size_t)t * 64], lds.data());
      for (int t = 0; t < T; ++t) x1_read<C, 1>(t, &regs[(size_t)t * 64], lds.data());
    }
    for (int t = 0; t < T; ++t) {
      cf* v = &regs[(size_t)t * 64];
      stage1<C, true>(t, v, tw.data());
      store_patch<C>(t, v, ov, 0, pr, pc, win.data(), add);
    }
  }
  return 0;
}

extern "C" int emu_apply(int N, int n_patches, const int32_t* coords, int H, int W, int pad_mode, float pad_value,
                         const float* img, const float* kfull, float* out) {
  switch (N) {
    case 256: return emu_apply_t<Cfg256>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out);
    case 128: return emu_apply_t<Cfg128>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out);
    case 64: return emu_apply_t<Cfg64>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out);
    case 32: return emu_apply_t<Cfg32>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out);
    case 16: return emu_apply_t<Cfg16>(n_patches, coords, H, W, pad_mode, pad_value, img, kfull, out);
    default: return -1;
  }
}
