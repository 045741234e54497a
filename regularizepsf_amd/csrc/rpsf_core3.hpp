// rpsf_core3.hpp - per-lane building blocks of the third kernel generation (N = 16, 32, 64): the "sweep" kernel of
// rpsf_kernels3.hpp, in which a workgroup owns a region of OUTPUT pixels, walks every patch that touches it and meets the four
// contributions to a pixel on chip (an N-row ring of the region in LDS) - no colour planes, no plane sum, every output pixel
// written once.  Like rpsf_core.hpp these functions compile for the GPU and, as plain C++, for the CPU emulator (tests/emu/emu3.cpp).
//
// Stands in for regularizepsf/transform.py:151-169 (window -> fft2 -> x K -> ifft2 -> real -> window -> overlap-add).
//
// Work unit: a SLAB = N rows x 128 columns of the (padded) image = PPW = 128 / N patches of one lattice row and one column
// parity, side by side, handled by ONE wave with no workgroup barrier.  A lane is one 1-D transform:
//   R-layout  lane (q, p), q = lane / H patch of the slab, p = lane % H:  v[c] = x[p][c] + i x[p + H][c]   (H = N / 2, rows p and p + H
//             of the patch packed into one complex row; c = column)
//   C-layout  lane (q, k): v[r] = column k of the half spectrum, r = row / row frequency; column 0 carries the two real columns
//             k = 0 and k = N/2 packed as one complex column
// with three transposes through a wave-private LDS buffer (real parts, then imaginary parts: half the buffer), all by 16-byte
// stores and 4-byte strided loads whose offsets are compile-time constants.
//
// Algebra (w = window, K_h = Hermitian fold of the caller's K, s = 1 / (2 N^2)):
//   u_p[c] = (w x)[p][c] + i (w x)[p+H][c];   U_p = DFT_c(u_p)
//   A_p[k] = U_p[k] + conj U_p[N-k] = 2 F_p[k],  B_p[k] = -i (U_p[k] - conj U_p[N-k]) = 2 F_{p+H}[k]      (F_r = spectrum of row r)
//   column k (0 < k < H): C_k[r] = 2 F_r[k];  column 0: D[r] = 2 F_r[0] + i 2 F_r[H]  (both real)
//   G_k = DFT_r(C_k) = 2 X[.][k];  G_k' = G_k * s K_h[.][k];   G_0'[q] = a[q] G_0[q] + b[q] conj G_0[-q],
//   a = s (K_h[q][0] + K_h[q][H]) / 2,  b = s (K_h[q][0] - K_h[q][H]) / 2
//   inverse column DFT, back to rows, U'[k] = V_p[k] + i V_{p+H}[k], U'[N-k] = conj V_p[k] + i conj V_{p+H}[k], inverse row DFT:
//   u'[c] = y[p][c] + i y[p+H][c], the patch's contribution before the second window.
#pragma once
#include <type_traits>

#include "rpsf_core.hpp"

namespace rpsf {

constexpr int pad_mod32(int x, int want) {  // smallest y >= x with y % 32 == want % 32
  int y = x;
  while ((y & 31) != (want & 31)) ++y;
  return y;
}

constexpr double sin_poly3(double x) {  // |x| <= pi / 2
  double x2 = x * x, term = x, sum = x;
  for (int n = 1; n <= 12; ++n) {
    term *= -x2 / ((2.0 * n) * (2.0 * n + 1.0));
    sum += term;
  }
  return sum;
}
template <int N>
constexpr float win3(int i) {  // sin((i + 1/2) pi / N), transform.py:151-155
  constexpr double pi = 3.14159265358979323846;
  double x = (i + 0.5) * pi / N;
  if (x > pi / 2) x = pi - x;
  return (float)sin_poly3(x);
}

template <int LOGN_, int KSMAX_ = 2, int WAVES_ = 8>
struct Cfg3 {
  static constexpr int LOGN = LOGN_, N = 1 << LOGN_, H = N / 2;
  static constexpr int SLABW = 128, PPW = SLABW / N;  // patches per slab
#if defined(RPSF3_PPP32)
  static constexpr int PPP = N == 32 ? RPSF3_PPP32 : N <= 32 ? PPW : 1;  // (development sweeps)
#else
  static constexpr int PPP = N <= 32 ? PPW : 1;       // patches per exchange pass (N = 64: one, the buffer holds half a slab)
#endif
  static constexpr int NSUB = PPW / PPP;
  static constexpr int WAVES = WAVES_, WG = 64 * WAVES_;
  // exchange sub-layouts (floats): X0[q][row < H][c < N] (gather -> rows, and columns -> rows on the way back as X2[q][k][r]),
  // X1[q][r < N][j < H] (rows -> columns).  Row pitches = 4 mod 32 (16-byte stores of 8 consecutive lanes hit 32 different banks),
  // patch strides = H mod 32 (the 4-byte loads of the 32 lanes of a half wave - two or four patches - hit 32 different banks).
  static constexpr int P0 = N + 4, Q0 = pad_mod32(H * P0, H);
  static constexpr int P1 = H + 4, Q1 = pad_mod32(N * P1, H);
  static constexpr int XF = PPP * (Q0 > Q1 ? Q0 : Q1);  // floats per wave
  // ring of the region's output rows in LDS: N rows x RP floats; RP / 4 odd (the 16-byte accesses of 16 consecutive rows - one lane each -
  // hit 64 different banks; the flush reads whole 16-byte units)
  static constexpr int KSMAX = KSMAX_;                 // slabs per parity and lattice row of a region
  static constexpr int RINGW = SLABW * KSMAX + H;
  static constexpr int RP = RINGW + 4;
  static_assert(RP % 4 == 0 && (RP / 4) % 2 == 1, "ring pitch");
  static constexpr int RINGF = N * RP;
  static constexpr int NFLAGS = 256;
  // The lanes reading their own rows straight into the R-layout (r3_load_*: no first transpose, a quarter less LDS traffic) LOSES: 64 different runs per load
  // instruction through the texture path cost more than the transpose saves - 4096^2: +13 % at N = 32, +25 % at N = 64, 0 at N = 16, 512^2 frames +20 %
  // (profiles/r06zy2_sweep_direct_gather.log).  Kept behind the switch.
#if defined(RPSF3_DIRECT_GATHER)
  static constexpr bool DIRECT_GATHER = true;
#else
  static constexpr bool DIRECT_GATHER = false;
#endif
  static constexpr bool DESC_PREFETCH = N >= 32;  // (rpsf_kernels3.hpp: the job descriptor WAVES jobs ahead is touched early)
  // N = 64: the slab is requested half by half (128 registers of pixels beside 128 of the transform do not fit a lane)
  static constexpr bool SPLIT_GATHER = N == 64;
  // a words (rpsf_kernels3.hpp: the transfer kernel of the lane's column) that a lane requests at the start of its job, half a job before it uses
  // them; the words behind them roll through the same registers (kmul3: a used word's register requests the word KPRE further on)
#if defined(RPSF3_KPRE)
  static constexpr int KPRE = RPSF3_KPRE < H ? RPSF3_KPRE : H;
#else
  static constexpr int KPRE = 8;  // (N = 64: 16 rolling words spill 17 registers, 0.198 -> 0.214 ms; 8 do not)
#endif
  // N = 64: requested only once the pixels' registers are free (behind the first transposes), N <= 32: at the start of the job
#if defined(RPSF3_KPRE_EARLY)
  static constexpr bool KPRE_LATE = false;  // (development sweeps)
#else
  static constexpr bool KPRE_LATE = N == 64;
#endif
  // per wave: the b words of the slab's column-0 lanes (one 16-byte word per lane, requested by all 64 lanes in one coalesced load) + 16 bytes of zeros
  static constexpr int SIDE_ZERO = 256, SIDEF = 260;
  static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)RINGF + (size_t)WAVES * (XF + SIDEF)) + sizeof(uint32_t) * (NFLAGS + 4);
  // packed K, per patch (floats): A[j < H][k < H][4] = (K'[r][k], K'[r'][k]) for the row pair (r, r') = (0, H) if j = 0, else (j, N - j); K' = s K_h, column 0 holding a; then B[j < H][4] = (b[r], b[r'])
  static constexpr int KA_FLOATS = H * H * 4, KB_FLOATS = H * 4, K_FLOATS = KA_FLOATS + KB_FLOATS;
  static constexpr float SCALE = 1.0f / (2.0f * (float)N * (float)N);
};

// ---- LDS accessors (plain memory on the host) -------------------------------------------------------------------------------
RPSF_HD f32x4 lds_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
RPSF_HD void lds_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
RPSF_HD f32x2 lds_ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
RPSF_HD void lds_add1(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // ds_add_f32, no return: the jobs of a region are ordered by their flags (rpsf_kernels3.hpp), so nobody else touches these words meanwhile;
  // the atomic form is used for its one-instruction read-modify-write
  __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
#else
  *p += v;
#endif
}

// ---- gather (transform.py:117-123,141-149,157-162) ----------------------------------------------------------------------------
// G-layout: load i < H covers slab rows 2i and 2i + 1; lane = (row parity, 16-byte unit u = lane & 31 of the 128 columns)
// PARTS = 2: both halves of the slab (rows [0, H) -> g[0 .. H/2), rows [H, N) -> g[H/2 .. H)); PARTS = 1: half PART only
template <class C, int PART = 0, int PARTS = 2>
RPSF_HD void g3_load_fast(int lane, f32x4* g, const float* slab, int ld) {
  const float* base = slab + (size_t)((lane >> 5) + PART * C::H) * ld + 4 * (lane & 31);
  StaticFor<0, PARTS * C::H / 2>::run([&]<int I>() RPSF_AI { g[PART * (C::H / 2) + I] = *reinterpret_cast<const f32x4*>(base + (size_t)(2 * I) * ld); });
}
// any slab: np.pad's index maps, pixel by pixel (slabs on the rim of the image, unaligned geometry)
template <class C, int PART = 0, int PARTS = 2>
RPSF_HD void g3_load_generic(int lane, f32x4* g, const ImageView& im, int row0, int col0) {
  int cx[4];
  for (int d = 0; d < 4; ++d) cx[d] = pad_index(col0 + 4 * (lane & 31) + d, im.W, im.pad_mode);
  StaticFor<0, PARTS * C::H / 2>::run([&]<int I>() RPSF_AI {
    int y = pad_index(row0 + PART * C::H + 2 * I + (lane >> 5), im.H, im.pad_mode);
    if (y >= 0) {
      y -= im.row0;
      if (y < 0 || y >= im.rows) y = -1;  // not resident: treated as fill (the launcher keeps every row a band needs resident)
    }
    const float* row = im.img + (size_t)(y < 0 ? 0 : y) * im.ld;
    float px[4];
    for (int d = 0; d < 4; ++d) {
      const float t = row[cx[d] < 0 ? 0 : cx[d]];  // always in bounds; select afterwards
      px[d] = (y < 0 || cx[d] < 0) ? im.pad_value : t;
    }
    g[PART * (C::H / 2) + I] = f32x4{px[0], px[1], px[2], px[3]};
  });
}

// ---- direct gather: the lane's own two rows, straight into the R-layout (no first transpose) ---------------------------------------------------
// (-DRPSF3_DIRECT_GATHER: measured, slower - see Cfg3::DIRECT_GATHER.)  Lane (q, p) reads row p (PART 0 -> real parts) or row p + H (PART 1 -> imaginary parts)
// of patch q: N / 4 sixteen-byte loads of one 4 N-byte run.  The 64 lanes of an instruction touch 64 different runs - the texture path takes them line by line.
template <class C, int PART>
RPSF_HD void r3_load_fast(int lane, cf* v, const float* slab, int ld) {
  const int q = lane / C::H, p = lane % C::H;
  const float* base = slab + (size_t)(p + PART * C::H) * ld + q * C::N;
  f32x4 t[C::N / 4];
  StaticFor<0, C::N / 4>::run([&]<int J>() RPSF_AI { t[J] = *reinterpret_cast<const f32x4*>(base + 4 * J); });
  StaticFor<0, C::N / 4>::run([&]<int J>() RPSF_AI {
    if constexpr (PART == 0) v[4 * J].x = t[J].x, v[4 * J + 1].x = t[J].y, v[4 * J + 2].x = t[J].z, v[4 * J + 3].x = t[J].w;
    else v[4 * J].y = t[J].x, v[4 * J + 1].y = t[J].y, v[4 * J + 2].y = t[J].z, v[4 * J + 3].y = t[J].w;
  });
}
// any slab: np.pad's index maps, pixel by pixel
template <class C, int PART>
RPSF_HD void r3_load_generic(int lane, cf* v, const ImageView& im, int row0, int col0) {
  const int q = lane / C::H, p = lane % C::H;
  int y = pad_index(row0 + p + PART * C::H, im.H, im.pad_mode);
  if (y >= 0) {
    y -= im.row0;
    if (y < 0 || y >= im.rows) y = -1;  // not resident: treated as fill (the launcher keeps every row a band needs resident)
  }
  const float* row = im.img + (size_t)(y < 0 ? 0 : y) * im.ld;
  StaticFor<0, C::N>::run([&]<int K>() RPSF_AI {
    const int cx = pad_index(col0 + q * C::N + K, im.W, im.pad_mode);
    const float t = row[cx < 0 ? 0 : cx];  // always in bounds; select afterwards
    const float px = (y < 0 || cx < 0) ? im.pad_value : t;
    if constexpr (PART == 0) v[K].x = px;
    else v[K].y = px;
  });
}

// ---- T0: G-layout -> R-layout.  PART 0: slab rows [0, H) -> real parts, PART 1: rows [H, N) -> imaginary parts ------------------------
template <class C, int PART, int SUB>
RPSF_HD void t0_write(int lane, const f32x4* g, float* xb) {
  const int u = lane & 31, hf = lane >> 5;
  const int q = (4 * u) / C::N, c = (4 * u) % C::N;
  if (q / C::PPP != SUB) return;
  float* base = xb + (q % C::PPP) * C::Q0 + hf * C::P0 + c;
  StaticFor<0, C::H / 2>::run([&]<int I>() RPSF_AI { lds_st4(base + (2 * I) * C::P0, g[PART * (C::H / 2) + I]); });
}
template <class C, int PART, int SUB>
RPSF_HD void t0_read(int lane, cf* v, const float* xb) {
  const int q = lane / C::H, p = lane % C::H;
  if (q / C::PPP != SUB) return;
  const float* base = xb + (q % C::PPP) * C::Q0 + p * C::P0;
  StaticFor<0, C::N / 4>::run([&]<int J>() RPSF_AI {
    const f32x4 t = lds_ld4(base + 4 * J);
    if constexpr (PART == 0) v[4 * J].x = t.x, v[4 * J + 1].x = t.y, v[4 * J + 2].x = t.z, v[4 * J + 3].x = t.w;
    else v[4 * J].y = t.x, v[4 * J + 1].y = t.y, v[4 * J + 2].y = t.z, v[4 * J + 3].y = t.w;
  });
}

// first window (transform.py:151-155,163) and the forward row transform.  The rows' weights (the lane's two) are plain multiplications; the columns'
// - compile-time constants - ride on the first butterflies of the decimation-in-time recursion, which pair element i with i + N/2 and have no twiddle:
// (wa e + wb o, wa e - wb o) is one multiplication and two fused multiply-adds per component where "weigh, then add and subtract" takes four operations.
template <int N, int LOG, int BASE, int STRIDE>
struct FftRowsW {  // forward, natural order in and out, elements x[i] stand for original positions BASE + i STRIDE
  static RPSF_HD void run(cf* x) {
    constexpr int h = 1 << (LOG - 1);
    if constexpr (LOG == 1) {
      constexpr float wa = win3<N>(BASE), wb = win3<N>(BASE + STRIDE);
      const cf e = x[0], o = x[1];
      const float tx = wa * e.x, ty = wa * e.y;
      x[0] = cf{__builtin_fmaf(wb, o.x, tx), __builtin_fmaf(wb, o.y, ty)};
      x[1] = cf{__builtin_fmaf(-wb, o.x, tx), __builtin_fmaf(-wb, o.y, ty)};
    } else {
      cf ev[h], od[h];
      StaticFor<0, h>::run([&]<int I>() RPSF_AI {
        ev[I] = x[2 * I];
        od[I] = x[2 * I + 1];
      });
      FftRowsW<N, LOG - 1, BASE, 2 * STRIDE>::run(ev);
      FftRowsW<N, LOG - 1, BASE + STRIDE, 2 * STRIDE>::run(od);
      StaticFor<0, h>::run([&]<int I>() RPSF_AI { butterfly_dit<I, LOG, false>(ev[I], od[I], x[I], x[I + h]); });
    }
  }
};
template <class C>
RPSF_HD void window_in_fft_rows(cf* v, float w_re, float w_im) {
  StaticFor<0, C::N>::run([&]<int I>() RPSF_AI {
    v[I].x *= w_re;
    v[I].y *= w_im;
  });
  FftRowsW<C::N, C::LOGN, 0, 1>::run(v);
}
// row spectra of the two packed rows, in place: v[k] (k < H) = column k of row p (k = 0: D[p]); v[H + j] = column (j ? H - j : 0) of row p + H
template <class C>
RPSF_HD void unpack_rows(cf* v) {
  constexpr int N = C::N, H = C::H;
  const cf u0 = v[0], uh = v[H];
  v[0] = cf{2.0f * u0.x, 2.0f * uh.x};
  v[H] = cf{2.0f * u0.y, 2.0f * uh.y};
  StaticFor<1, H>::run([&]<int K>() RPSF_AI {
    const cf a = v[K], b = v[N - K];                 // U[k] = a, U[N-k] = b
    v[K] = cf{a.x + b.x, a.y - b.y};                 // A[k] = U[k] + conj U[N-k]
    v[N - K] = cf{a.y + b.y, b.x - a.x};             // B[k] = -i (U[k] - conj U[N-k])
  });
}
// and back: from V_p[k] (v[k]) and V_{p+H}[k] (v[N-k]) to the spectrum of u' = y_p + i y_{p+H}
template <class C>
RPSF_HD void repack_rows(cf* v) {
  constexpr int N = C::N, H = C::H;
  const cf d0 = v[0], d1 = v[H];  // D'[p], D'[p+H]
  v[0] = cf{d0.x, d1.x};
  v[H] = cf{d0.y, d1.y};
  StaticFor<1, H>::run([&]<int K>() RPSF_AI {
    const cf a = v[K], b = v[N - K];                 // Va, Vb
    v[K] = cf{a.x - b.y, a.y + b.x};                 // Va + i Vb
    v[N - K] = cf{a.x + b.y, b.x - a.y};             // conj Va + i conj Vb
  });
}

// ---- T1: R-layout -> C-layout through X1[q][r][j] (row p: j = k; row p + H: j = 0 for k = 0, else H - k: the registers v[H + j] in order) ----
template <class C, int PART, int SUB>
RPSF_HD void t1_write(int lane, const cf* v, float* xb) {
  const int q = lane / C::H, p = lane % C::H;
  if (q / C::PPP != SUB) return;
  float* base = xb + (q % C::PPP) * C::Q1 + p * C::P1;
  StaticFor<0, C::N / 4>::run([&]<int J>() RPSF_AI {
    constexpr int R = (4 * J) / C::H, J0 = (4 * J) % C::H;  // R = 0: row p, R = 1: row p + H
    f32x4 t;
    if constexpr (PART == 0) t = f32x4{v[4 * J].x, v[4 * J + 1].x, v[4 * J + 2].x, v[4 * J + 3].x};
    else t = f32x4{v[4 * J].y, v[4 * J + 1].y, v[4 * J + 2].y, v[4 * J + 3].y};
    lds_st4(base + R * C::H * C::P1 + J0, t);
  });
}
template <class C, int PART, int SUB>
RPSF_HD void t1_read(int lane, cf* v, const float* xb) {
  const int q = lane / C::H, k = lane % C::H;
  if (q / C::PPP != SUB) return;
  const float* lo = xb + (q % C::PPP) * C::Q1 + k;
  const float* hi = xb + (q % C::PPP) * C::Q1 + C::H * C::P1 + (k == 0 ? 0 : C::H - k);
  StaticFor<0, C::H>::run([&]<int R>() RPSF_AI {
    const float a = lo[R * C::P1], b = hi[R * C::P1];
    if constexpr (PART == 0) v[R].x = a, v[C::H + R].x = b;
    else v[R].y = a, v[C::H + R].y = b;
  });
}

// ---- frequency step (transform.py:164): x s K_h, lane = column k ----------------------------------------------------------------
// Rows pair as (r, N - r): word 0 of a lane's stream holds the two self-paired rows (0, H), word j >= 1 the rows (j, N - j).
// ka: the lane's first 16-byte word of A (H words apart from one j to the next); kb: the lane's B words (column 0, kb_stride = 4 floats) or 16
// bytes of zeros (every other column, kb_stride = 0: b = 0, so that the rule of column 0 costs no branch).
template <int N, int H>
constexpr int k3_row(int j, int m) { return j == 0 ? (m ? H : 0) : (m ? N - j : j); }
template <bool NT>
RPSF_HD f32x4 load_k3(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#endif
  return *reinterpret_cast<const f32x4*>(p);
}
// ka_pre: the first KPRE words, requested at the start of the job.  The words behind them ROLL through the same registers: as soon as word J has been
// used its register requests word J + KPRE, so KPRE loads stay in flight and the lane waits for memory once, not once per word (a frame of a few
// hundred patches has no other wave to hide a load's latency behind).  4096^2 / 32: -2.5 %, N = 64: -1 %, 512^2 frames: 0 (profiles/r06zl_sweep_rolling_k.log).
template <class C, bool NT>
RPSF_HD void kmul3(cf* v, f32x4* ka_pre, const float* ka, const float* kb, int kb_stride) {
  constexpr int N = C::N, H = C::H, KP = C::KPRE;
  // the b words come out of LDS eight at a time (N <= 32; N = 64 has no registers to spare: two at a time, as the compiler pairs them anyway) - read one by
  // one where they are used they cost an LDS round trip each, sixteen per job at N = 32
  constexpr int BB = H <= 16 ? (H < 8 ? H : 8) : 2;
  f32x4 kbv[BB];
  StaticFor<0, H>::run([&]<int J>() RPSF_AI {
    if constexpr (J % BB == 0) StaticFor<0, BB>::run([&]<int I>() RPSF_AI { kbv[I] = lds_ld4(kb + (size_t)(J + I) * kb_stride); });
    cf ae, ao, be, bo;
    if constexpr (KP > 0) {
      const f32x4 w = ka_pre[J % KP];
      ae = cf{w.x, w.y}, ao = cf{w.z, w.w};
      if constexpr (J + KP < H) ka_pre[J % KP] = load_k3<NT>(ka + (size_t)(J + KP) * (H * 4));
    } else {
      load_k16<NT>(ka + (size_t)J * (H * 4), ae, ao);
    }
    be = cf{kbv[J % BB].x, kbv[J % BB].y}, bo = cf{kbv[J % BB].z, kbv[J % BB].w};
    constexpr int RE = k3_row<N, H>(J, 0), RO = k3_row<N, H>(J, 1);
    const cf x = v[RE], y = v[RO];
    if constexpr (J == 0) {  // v'[r] = a[r] v[r] + b[r] conj v[-r], -r = r for both rows
      v[RE] = cmul(ae, x) + cmul(be, cconj(x));
      v[RO] = cmul(ao, y) + cmul(bo, cconj(y));
    } else {
      v[RE] = cmul(ae, x) + cmul(be, cconj(y));
      v[RO] = cmul(ao, y) + cmul(bo, cconj(x));
    }
  });
}

// ---- T2: C-layout -> R-layout through X2[q][k][r] (same pitches as X0) ---------------------------------------------------------------
template <class C, int PART, int SUB>
RPSF_HD void t2_write(int lane, const cf* v, float* xb) {
  const int q = lane / C::H, k = lane % C::H;
  if (q / C::PPP != SUB) return;
  float* base = xb + (q % C::PPP) * C::Q0 + k * C::P0;
  StaticFor<0, C::N / 4>::run([&]<int J>() RPSF_AI {
    f32x4 t;
    if constexpr (PART == 0) t = f32x4{v[4 * J].x, v[4 * J + 1].x, v[4 * J + 2].x, v[4 * J + 3].x};
    else t = f32x4{v[4 * J].y, v[4 * J + 1].y, v[4 * J + 2].y, v[4 * J + 3].y};
    lds_st4(base + 4 * J, t);
  });
}
template <class C, int PART, int SUB>
RPSF_HD void t2_read(int lane, cf* v, const float* xb) {
  const int q = lane / C::H, p = lane % C::H;
  if (q / C::PPP != SUB) return;
  const float* base = xb + (q % C::PPP) * C::Q0 + p;
  StaticFor<0, C::H>::run([&]<int J>() RPSF_AI {
    constexpr int KHI = J == 0 ? 0 : C::H - J;  // the column that register v[H + J] holds
    const float a = base[J * C::P0], b = base[KHI * C::P0 + C::H];
    if constexpr (PART == 0) v[J].x = a, v[C::H + J].x = b;
    else v[J].y = a, v[C::H + J].y = b;
  });
}

// ---- second window and overlap-add into the ring (transform.py:165-169) ---------------------------------------------------------
enum Acc3 : int { ACC_SKIP = 0, ACC_STORE = 1, ACC_ADD = 2 };
// second window, in place (transform.py:165).  An invalid patch (a virtual column beside the lattice, an empty slot of a slab) gets row weights of zero: its
// pixels are image pixels that a real patch of the same rows covers too, so where they are finite it contributes exact zeros and where they are not
// the real patch has spread the NaN over the same pixels already (one select per job instead of one per value: 2 N fewer instructions).
template <class C>
RPSF_HD void window_out(cf* v, float w_re, float w_im, bool valid) {
  const float wr = valid ? w_re : 0.0f, wi = valid ? w_im : 0.0f;
  StaticFor<0, C::N>::run([&]<int I>() RPSF_AI {
    constexpr float wc = win3<C::N>(I);
    v[I].x = (v[I].x * wr) * wc;
    v[I].y = (v[I].y * wi) * wc;
  });
}
// upper: the lane's row p of the slab (real parts), lower: row p + H (imaginary parts); ru / rl: their ring rows at the patch's first column.
// The jobs of a region are ordered by their flags (rpsf_kernels3.hpp), so nobody else touches these words meanwhile: plain 16-byte
// read - add - write.  (LDS float atomics - ds_add_f32 - take about two clocks per LANE on gfx950: 64 of them per job kept the LDS of the
// CU busy for 9,750 of a job's 12,600 clocks, profiles/r06c.)
template <class C>
RPSF_HD void accumulate3(const cf* v, float* ru, float* rl, int mode_u, int mode_l) {
  // The mode is decided ONCE per band, outside the loops: inside them it was a branch per 16-byte unit, which kept the reads from being issued back to back -
  // every unit paid an LDS round trip of its own (1.9 us per job at N = 32, inside the section the jobs behind this one wait for; profiles/r06zw).  BATCH units
  // are read together, added and written back (N = 64 has no registers for more than eight at a time).
  constexpr int UNITS = C::N / 4, BATCH = UNITS < 8 ? UNITS : 8;
  auto unit = [&]<bool RE, int J>() RPSF_AI {
    const float x0 = RE ? v[4 * J].x : v[4 * J].y, x1 = RE ? v[4 * J + 1].x : v[4 * J + 1].y;
    const float x2 = RE ? v[4 * J + 2].x : v[4 * J + 2].y, x3 = RE ? v[4 * J + 3].x : v[4 * J + 3].y;
    return f32x4{x0, x1, x2, x3};
  };
  if constexpr (UNITS <= 8) {
    // N <= 32: both bands' reads in flight together - one round trip for the whole job
    f32x4 ou[UNITS], ol[UNITS];
    if (mode_u == ACC_ADD) StaticFor<0, UNITS>::run([&]<int J>() RPSF_AI { ou[J] = lds_ld4(ru + 4 * J); });
    if (mode_l == ACC_ADD) StaticFor<0, UNITS>::run([&]<int J>() RPSF_AI { ol[J] = lds_ld4(rl + 4 * J); });
    auto finish = [&](auto re_part, float* r, int mode, const f32x4* o) RPSF_AI {
      constexpr bool RE = decltype(re_part)::value;
      if (mode == ACC_ADD) {
        StaticFor<0, UNITS>::run([&]<int J>() RPSF_AI {
          const f32x4 t = unit.template operator()<RE, J>();
          lds_st4(r + 4 * J, f32x4{o[J].x + t.x, o[J].y + t.y, o[J].z + t.z, o[J].w + t.w});
        });
      } else if (mode == ACC_STORE) {
        StaticFor<0, UNITS>::run([&]<int J>() RPSF_AI { lds_st4(r + 4 * J, unit.template operator()<RE, J>()); });
      }
    };
    finish(std::true_type(), ru, mode_u, ou);
    finish(std::false_type(), rl, mode_l, ol);
  } else {
    auto half = [&](auto re_part, float* r, int mode) RPSF_AI {
      constexpr bool RE = decltype(re_part)::value;
      if (mode == ACC_ADD) {
        StaticFor<0, UNITS / BATCH>::run([&]<int B>() RPSF_AI {
          f32x4 o[BATCH];
          StaticFor<0, BATCH>::run([&]<int I>() RPSF_AI { o[I] = lds_ld4(r + 4 * (B * BATCH + I)); });
          StaticFor<0, BATCH>::run([&]<int I>() RPSF_AI {
            const f32x4 t = unit.template operator()<RE, B * BATCH + I>();
            lds_st4(r + 4 * (B * BATCH + I), f32x4{o[I].x + t.x, o[I].y + t.y, o[I].z + t.z, o[I].w + t.w});
          });
        });
      } else if (mode == ACC_STORE) {
        StaticFor<0, UNITS>::run([&]<int J>() RPSF_AI { lds_st4(r + 4 * J, unit.template operator()<RE, J>()); });
      }
    };
    half(std::true_type(), ru, mode_u);
    half(std::false_type(), rl, mode_l);
  }
}

// ---- flush: H finished ring rows x the slab's 128 columns -> the output image (transform.py:174-177, float32 here) ------------------
// band_row: image row of the band's first row; col0: image column of the slab's first column; [oc0, oc1): image columns this region owns
struct Flush3 {
  float* out;  // rows [row0, row0 + rows) of the output, row stride ld
  int ld, row0, rows, Himg, Wimg;
  int aligned;  // 16-byte stores allowed (ld, column origin and pointer multiples of four floats)
};
// G-layout: unit i < H / 2 = rows 2i and 2i + 1 of the band, lane = (row parity, 16-byte unit of the 128 columns)
template <class C, class Store4, class Store1>
RPSF_HD void flush3(int lane, const float* ring_band, const Flush3& f, int band_row, int col0, int oc0, int oc1, Store4&& st4, Store1&& st1) {
  const int u = lane & 31, hf = lane >> 5;
  const int c = col0 + 4 * u;
  const int lo = oc0 > 0 ? oc0 : 0, hi = oc1 < f.Wimg ? oc1 : f.Wimg;
  // (the band's units are read first, all of them, and stored afterwards: with the read inside the loop of conditional stores every unit paid an LDS round
  // trip of its own - 0.8 us per flushing job at N = 32, inside the section the jobs behind it wait for)
  constexpr int UNITS = C::H / 2, BATCH = UNITS < 8 ? UNITS : 8;
  StaticFor<0, UNITS / BATCH>::run([&]<int B>() RPSF_AI {
  f32x4 band[BATCH];
  StaticFor<0, BATCH>::run([&]<int K>() RPSF_AI { band[K] = lds_ld4(ring_band + (2 * (B * BATCH + K) + hf) * C::RP + 4 * u); });
  StaticFor<0, BATCH>::run([&]<int K>() RPSF_AI {
    constexpr int I = B * BATCH + K;
    const int r = band_row + 2 * I + hf;
    const f32x4 ab = band[K];
    if (r >= f.row0 && r < f.row0 + f.rows && r >= 0 && r < f.Himg) {
      float* dst = f.out + (size_t)(r - f.row0) * f.ld + c;
      if (f.aligned && c >= lo && c + 4 <= hi) {
        st4(dst, ab);
      } else {
        if (c >= lo && c < hi) st1(dst, ab.x);
        if (c + 1 >= lo && c + 1 < hi) st1(dst + 1, ab.y);
        if (c + 2 >= lo && c + 2 < hi) st1(dst + 2, ab.z);
        if (c + 3 >= lo && c + 3 < hi) st1(dst + 3, ab.w);
      }
    }
  });
  });
}

// ---- K pack (one-time, at set_transfer): value `idx` of a patch's K_FLOATS / 2 complex values ---------------------------------------------
template <class C, class KF>
RPSF_HD cf kh3_at(const KF& kfull, int kr, int kc) {  // Hermitian fold (legal for any K: only Re(ifft2) is kept, transform.py:164), not scaled
  const cf a = k_at(kfull, kr * C::N + kc);
  const cf b = k_at(kfull, ((C::N - kr) & (C::N - 1)) * C::N + ((C::N - kc) & (C::N - 1)));
  return cf{(a.x + b.x) * 0.5f, (a.y - b.y) * 0.5f};
}
template <class C, class KF>
RPSF_HD cf pack_value3(const KF& kfull, int idx) {
  constexpr int H = C::H, N = C::N;
  const bool side = idx >= H * H * 2;  // B[j][2] behind A[j][k][2]
  const int i2 = side ? idx - H * H * 2 : idx;
  const int m = i2 & 1, k = side ? 0 : (i2 >> 1) % H, j = side ? i2 >> 1 : (i2 >> 1) / H;
  const int r = j == 0 ? (m ? H : 0) : (m ? N - j : j);
  if (k != 0) {
    const cf x = kh3_at<C>(kfull, r, k);
    return cf{x.x * C::SCALE, x.y * C::SCALE};
  }
  const cf x0 = kh3_at<C>(kfull, r, 0), xh = kh3_at<C>(kfull, r, H);
  if (!side) return cf{(x0.x + xh.x) * (0.5f * C::SCALE), (x0.y + xh.y) * (0.5f * C::SCALE)};
  return cf{(x0.x - xh.x) * (0.5f * C::SCALE), (x0.y - xh.y) * (0.5f * C::SCALE)};
}

// (development sweeps: -DRPSF3_W16=.. etc. select other wave counts)
#if !defined(RPSF3_W16)
#define RPSF3_W16 16
#endif
#if !defined(RPSF3_W32)
#define RPSF3_W32 8
#endif
#if !defined(RPSF3_W64)
#define RPSF3_W64 8
#endif
#if !defined(RPSF3_KS16)
#define RPSF3_KS16 4
#endif
#if !defined(RPSF3_KS32)
#define RPSF3_KS32 2
#endif
using Cfg3_16 = Cfg3<4, RPSF3_KS16, RPSF3_W16>;
using Cfg3_32 = Cfg3<5, RPSF3_KS32, RPSF3_W32>;
using Cfg3_64 = Cfg3<6, 2, RPSF3_W64>;

}  // namespace rpsf
