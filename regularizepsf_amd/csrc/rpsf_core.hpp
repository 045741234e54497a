// rpsf_core.hpp - per-thread building blocks of the fused patch kernel (K1), written once and
// used twice: by the HIP kernels in rpsf.hip and, compiled as plain C++, by the CPU thread
// emulator in tests/emu/ (which drives these phases thread by thread to check the index algebra
// without a GPU).  Nothing here is a port of reference code: the reference path
// (regularizepsf/transform.py:151-169) is pad -> gather -> window -> fft2 -> *K -> ifft2 -> real ->
// window -> overlap-add on NumPy arrays; this file restructures it for one workgroup per patch with
// the whole patch resident in the register file.
//
// Algorithm (N x N real patch x, N = 2^LOGN):
//   * pack column pairs:  z[r][c] = x[r][2c] + i x[r][2c+1],  r < N, c < N/2          (N*N/2 complex)
//   * Z = 2-D complex DFT of z, done as 2 or 3 "stages"; every thread owns 64 complex values in
//     registers and a stage is a small in-register DFT along the row digit and the column digit it
//     owns, followed by twiddles; between stages the block transposes through LDS (two passes, real
//     parts then imaginary parts, because a 256x256 patch is 256 KiB and LDS is 160 KiB).
//   * in the last layout every thread holds, for each of its "slots", the two groups of bins
//     {(q,m)} and {(-q,-m)} so that the bins p and -p needed to unpack the real-input spectrum
//     X[kr][kc] = E + W^kc O,  X[kr][kc+N/2] = E - W^kc O   (E,O from Z[p], conj Z[-p])
//     sit in the same thread.  There X is multiplied by the Hermitian-folded transfer kernel
//     K_h(k) = (K(k) + conj K(-k))/2  (legal for any K because only Re(ifft2) is kept,
//     transform.py:164) and re-packed; the inverse DFT then retraces the stages backwards.
//   * the result is windowed again and overlap-added into the output image (transform.py:165-169).
//
// Index algebra.  Row index r has LR = LOGN bits split in digits (A1 | A2 | AL), most significant
// first; packed-column index c has LC = LOGN-1 bits split in (B1 | B2 | BL).  Decimation in frequency:
//   r = r1*2^(A2+AL) + r2*2^AL + r3        ->  kr = k1 + k2*2^A1 + k3*2^(A1+A2) = q + Q*k3
//   c = c1*2^(B2+BL) + c2*2^BL + c3        ->  kc = l1 + l2*2^B1 + l3*2^(B1+B2) = m + M*l3
// stage 1 owns (r1,c1) in registers, stage 2 (r2,c2), the last stage (r3,c3): element e = k3*2^BL + l3 of
// group (q,m).  -p maps element e of a group to element E-1-e of the partner group (-q,-m) whenever
// q != 0 and m != 0 (both digits are complemented).  A1+B1 = 6 and A2+B2 = 6 (three-stage plans,
// N >= 128, BL = 0) or 0 (two-stage plans, N <= 64, which put column bits in the last stage so that
// consecutive lanes own consecutive columns).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RPSF_HD __host__ __device__ __forceinline__
#else
#define RPSF_HD inline
#endif
// lambdas passed to StaticFor must be inlined, or the register tile they touch is forced to memory
#define RPSF_AI __attribute__((always_inline))

namespace rpsf {

// Scalar complex.  Measured on MI355X: v_pk_{add,mul,fma}_f32 issue at half the rate of the scalar
// forms (no throughput gain) and need their operands in aligned register pairs, which cost the
// packed build ~1800 v_mov and ~800 s_nop per thread; scalar code lets re/im be renamed freely.
// Build with -fno-slp-vectorize so the compiler does not re-pack.
struct alignas(8) cf {
  float x, y;
};
RPSF_HD cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
RPSF_HD cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
RPSF_HD cf operator*(cf a, cf b) { return cf{a.x * b.x, a.y * b.y}; }
RPSF_HD cf operator*(cf a, float b) { return cf{a.x * b, a.y * b}; }
RPSF_HD cf operator-(cf a) { return cf{-a.x, -a.y}; }

RPSF_HD cf cmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
RPSF_HD cf cmulc(cf a, cf b) { return cf{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }  // a * conj(b)
RPSF_HD cf cconj(cf a) { return cf{a.x, -a.y}; }
RPSF_HD cf mul_pi(cf a) { return cf{-a.y, a.x}; }   // a * (+i)
RPSF_HD cf mul_mi(cf a) { return cf{a.y, -a.x}; }   // a * (-i)
// select on VALUES: `c ? arr[i] : arr[j]` would become a load through a selected pointer and
// push the whole register tile into scratch
RPSF_HD cf sel(bool c, cf a, cf b) { return cf{c ? a.x : b.x, c ? a.y : b.y}; }

// Streaming (non-temporal) accesses for data that is touched exactly once per apply - the packed K
// stream and the colour-plane stores - so they do not evict the image tiles that four patches share in L2.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
RPSF_HD void load_stream16(const void* p, cf& a, cf& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  f32x4 q = *reinterpret_cast<const f32x4*>(p);
#endif
  a = cf{q.x, q.y};
  b = cf{q.z, q.w};
}
// the pair words of K: streamed (nontemporal) where a launch's K would push everything else out of the caches, plain where it fits beside the rest
// and the next apply - or the next frame of a batch - finds it there (rpsf.hip, k_cached)
template <bool NT>
RPSF_HD void load_k16(const void* p, cf& a, cf& b) {
  if constexpr (NT) load_stream16(p, a, b);
  else {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p);
    a = cf{q.x, q.y};
    b = cf{q.z, q.w};
  }
}
RPSF_HD void store_stream8(void* p, cf v) {
  f32x2 q = {v.x, v.y};
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_nontemporal_store(q, reinterpret_cast<f32x2*>(p));
#else
  *reinterpret_cast<f32x2*>(p) = q;
#endif
}

// cos(2 pi k / 64) for the first quadrant; everything else by symmetry so that 0 and +-1 are exact.
constexpr float kCosQ[17] = {1.0f,          0.99518472f, 0.98078525f, 0.95694035f, 0.92387950f, 0.88192129f,
                             0.83146960f,   0.77301043f, 0.70710677f, 0.63439327f, 0.55557024f, 0.47139674f,
                             0.38268343f,   0.29028466f, 0.19509032f, 0.09801714f, 0.0f};
constexpr float cos64(int k) {
  k &= 63;
  return k <= 16 ? kCosQ[k] : k <= 32 ? -kCosQ[32 - k] : k <= 48 ? -kCosQ[k - 32] : kCosQ[64 - k];
}
constexpr float sin64(int k) { return cos64(k - 16); }

template <int I, int CNT>
struct StaticFor {
  template <class F>
  static RPSF_HD void run(F&& f) {
    f.template operator()<I>();
    StaticFor<I + 1, CNT>::run(f);
  }
};
template <int CNT>
struct StaticFor<CNT, CNT> {
  template <class F>
  static RPSF_HD void run(F&&) {}
};

// In-register DFT of 2^LOG points, natural order in and out, radix-2 decimation in time with compile-time
// twiddles.  All indices are compile-time, so x[] lives in registers.  A butterfly with a non-trivial twiddle
// W = c + i s is six fused multiply-adds (X = E + W O accumulated straight onto E, Y = 2E - X) instead of
// the eight operations of "multiply, then add and subtract".
template <int I, int LOG, bool INV>
RPSF_HD void butterfly_dit(cf e, cf o, cf& x, cf& y) {
  constexpr int n = 1 << LOG;
  constexpr int k = (I & (n - 1)) * (64 / n);  // index on the 64-point circle
  if constexpr (k == 0) {
    x = e + o, y = e - o;
  } else if constexpr (k == 16) {
    cf t = INV ? mul_pi(o) : mul_mi(o);
    x = e + t, y = e - t;
  } else {
    constexpr float c = cos64(k);
    constexpr float s = INV ? sin64(k) : -sin64(k);  // W = c + i s
    x.x = __builtin_fmaf(-s, o.y, __builtin_fmaf(c, o.x, e.x));
    x.y = __builtin_fmaf(c, o.y, __builtin_fmaf(s, o.x, e.y));
    y.x = __builtin_fmaf(2.0f, e.x, -x.x);
    y.y = __builtin_fmaf(2.0f, e.y, -x.y);
  }
}
template <int LOG, bool INV>
struct FftSmall {
  static RPSF_HD void run(cf* x) {
    constexpr int h = 1 << (LOG - 1);
    cf ev[h], od[h];
    StaticFor<0, h>::run([&]<int I>() RPSF_AI {
      ev[I] = x[2 * I];
      od[I] = x[2 * I + 1];
    });
    FftSmall<LOG - 1, INV>::run(ev);
    FftSmall<LOG - 1, INV>::run(od);
    StaticFor<0, h>::run([&]<int I>() RPSF_AI { butterfly_dit<I, LOG, INV>(ev[I], od[I], x[I], x[I + h]); });
  }
};
template <bool INV>
struct FftSmall<0, INV> {
  static RPSF_HD void run(cf*) {}
};

// DFT along one axis of the thread's register tile: for every "other" index o < OTHER,
// elements v[BASE + o*OSTRIDE + i*STRIDE], i < 2^LOG.
template <int LOG, int STRIDE, int OTHER, int OSTRIDE, bool INV, int BASE = 0>
RPSF_HD void fft_axis(cf* v) {
  if constexpr (LOG > 0) {
    constexpr int n = 1 << LOG;
    StaticFor<0, OTHER>::run([&]<int O>() RPSF_AI {
      cf x[n];
      StaticFor<0, n>::run([&]<int I>() RPSF_AI { x[I] = v[BASE + O * OSTRIDE + I * STRIDE]; });
      FftSmall<LOG, INV>::run(x);
      StaticFor<0, n>::run([&]<int I>() RPSF_AI { v[BASE + O * OSTRIDE + I * STRIDE] = x[I]; });
    });
  }
}

// ------------------------------------------------------------------------------------------
// Plan geometry
// ------------------------------------------------------------------------------------------
template <int LOGN_, int A1_, int A2_, int AL_, int B1_, int B2_, int BL_ = 0>
struct Cfg {
  static constexpr int LOGN = LOGN_, N = 1 << LOGN_, NC = N / 2, LR = LOGN_, LC = LOGN_ - 1;
  static constexpr int A1 = A1_, A2 = A2_, AL = AL_, B1 = B1_, B2 = B2_, BL = BL_;
  static constexpr bool S3 = (A2_ + B2_) != 0;  // three stages
  static_assert(A1_ + A2_ + AL_ == LOGN_ && B1_ + B2_ + BL_ == LOGN_ - 1, "digits must cover the index");
  static_assert(A1_ + B1_ == 6 && (A2_ + B2_ == 6 || A2_ + B2_ == 0) && AL_ + BL_ >= 1 && B1_ + B2_ >= 1,
                "64 values per thread per stage");
  static constexpr int EA = 1 << AL_, EB = 1 << BL_;
  static constexpr int E = EA * EB;           // bins per group (last-stage 2-D DFT size), e = k3*EB + l3
  static constexpr int P = 64 / E;            // groups per thread in the last layout
  static constexpr int NSLOT = P / 2;         // pair slots per thread
  static constexpr int KCH = E >= 8 ? 8 : 4;   // pair words of K per streaming chunk (2 cf = 4 registers each)
  // K chunks in flight.  One everywhere a wave has 256 registers (a second chunk spills, and spill reloads queue
  // behind the K stream); two in the N = 64 plan, which is compiled for one wave per SIMD (see Launch in
  // rpsf.hip) and has the registers: 2048^2 / N=64 81 -> 71.5 us, a depth of four gives nothing more.
  static constexpr int KDEPTH = (LOGN_ == 6 && A2_ + B2_ == 0) ? 2 : 1;
  static constexpr bool FUSE_LAST = true;  // the last stage runs slot by slot around the multiplication
  static constexpr int T = N * NC / 64;       // threads per patch
  static constexpr int LQ = A1_ + A2_, Q = 1 << LQ, M = 1 << (B1_ + B2_), G = Q * M;  // kr = q + Q*k3, kc = m + M*l3
  static constexpr int NSPEC = (Q + M) / 2;   // slots whose groups have q == 0, m == 0 or are self-paired
  static constexpr int WAVE = T < 64 ? T : 64;  // threads that share the special/general decision
  // threads t' < spec_t(s) of slot s use the "special" K format (natural K_h plus the Nyquist-side array)
  static constexpr int spec_t(int s) {
    int left = NSPEC - s * T;
    if (left <= 0) return 0;
    int r = (left + WAVE - 1) / WAVE * WAVE;
    return r > T ? T : r;
  }
  static constexpr int spec_prefix(int s) {  // threads in special format in slots < s
    int acc = 0;
    for (int i = 0; i < s; ++i) acc += spec_t(i);
    return acc;
  }
  static constexpr int NSPEC_THREADSLOTS = spec_prefix(NSLOT);
  // Two-stage plans: a slot is special-format for the whole team or not at all.  Such a slot is worked through as
  // a list of bin pairs ("orbits" of k -> -k inside the slot's two groups) read from a small table: both bins of
  // a pair are fetched from the LDS parking area by computed address, one pair_op serves both, both results go
  // back to the parking area, and the stream carries one ordinary K word per pair - (K_h(p), K_h(p + (0,N/2)))
  // for the first bin p of the pair - instead of separate words for the two members.  (Before: two pair_ops
  // per bin pair and twice the K bytes, because the partner of a bin was a run-time-indexed register.)
  // Three-stage plans keep the per-bin variant with the side array gs: only their leading waves are special.
  static constexpr bool ORBIT = !S3;
  static constexpr bool INLINE_GS = ORBIT;                  // no side array
  static constexpr int NP = E + 2;                          // pair words of a special slot: E orbits, two more in the slot that holds group (0,0) and its four fixed points
  static constexpr int slot_words(int s) { return (ORBIT && spec_t(s) > 0) ? NP : E; }
  static constexpr int special_index(int s) {               // special slots before s
    int acc = 0;
    for (int i = 0; i < s; ++i) acc += spec_t(i) > 0 ? 1 : 0;
    return acc;
  }
  static constexpr int NSS = special_index(NSLOT);          // special slots per thread
  static constexpr int PT_WORDS = ORBIT ? T * NSS * NP : 0; // pair table, uint32 per (thread, special slot, pair)
  static constexpr int word_base(int s) {
    int acc = 0;
    for (int i = 0; i < s; ++i) acc += slot_words(i);
    return acc;
  }
  static constexpr int word_slot(int w) {  // slot that word w belongs to (NSLOT for padding words)
    int s = 0;
    while (s < NSLOT && w >= word_base(s + 1)) ++s;
    return s;
  }
  static constexpr int NWORDS_USED = word_base(NSLOT);
  static constexpr int NWORDS = (NWORDS_USED + KCH - 1) / KCH * KCH;  // whole chunks
  static constexpr int GS_PER_PATCH = INLINE_GS ? 0 : NSPEC_THREADSLOTS * 2 * E;  // complex values
  static constexpr int G_PER_PATCH = NWORDS * T * 2;                              // complex values
  // LDS floats for one exchange pass
  static constexpr int X1_ROW = 68;  // floats per X1 row: 16-byte aligned rows (wide reads), 4*lane + c mod 64 covers every bank once
  static constexpr int X1_FLOATS = S3 ? (T / 64) * 64 * X1_ROW : 0;
  static constexpr int X2_STRIDE = S3 ? G : G + 1;
  static constexpr bool TWO_PASS_SPECIAL = S3 && KCH == E;  // see special_pass1 (N = 256; at N = 128 a chunk holds four slots)
  static constexpr int X2_FLOATS = E * X2_STRIDE;
  static constexpr int LDS_FLOATS0 = X1_FLOATS > X2_FLOATS ? X1_FLOATS : X2_FLOATS;
  // Parked special slot: thread-private LDS columns scratch[(h*E + e)*PARK_STRIDE + t].  Two-stage plans are
  // single-wave workgroups and park inside the (idle) exchange buffer; three-stage plans park behind it,
  // because faster waves are already rewriting the exchange buffer while a special wave still reads its columns.
  static constexpr int PARK_STRIDE = S3 ? spec_t(0) : T;
  static constexpr int PARK_FLOATS = 2 * E * PARK_STRIDE * 2;  // cf = 2 floats
  static constexpr int PARK_OFFSET = S3 ? LDS_FLOATS0 : 0;
  static constexpr int LDS_FLOATS = S3 ? LDS_FLOATS0 + PARK_FLOATS : (LDS_FLOATS0 > PARK_FLOATS ? LDS_FLOATS0 : PARK_FLOATS);
  static constexpr float SCALE = 1.0f / (2.0f * (float)N * (float)N);  // 1/4 (pair algebra) * 1/(N*N/2) (inverse DFT)
};

// gid' (the LDS-friendly group numbering) <-> (q, m)
template <class C>
RPSF_HD void gid_to_qm(int gid, int& q, int& m) {
  if constexpr (C::S3) {
    int lane = gid & 63, j = gid >> 6;
    int k1 = lane >> C::B1, l1 = lane & ((1 << C::B1) - 1);
    int k2 = j >> C::B2, l2 = j & ((1 << C::B2) - 1);
    q = k1 + (k2 << C::A1);
    m = l1 + (l2 << C::B1);
  } else {
    q = gid >> C::B1;
    m = gid & ((1 << C::B1) - 1);
  }
}
template <class C>
RPSF_HD int qm_to_gid(int q, int m) {
  if constexpr (C::S3) {
    int k1 = q & ((1 << C::A1) - 1), k2 = q >> C::A1;
    int l1 = m & ((1 << C::B1) - 1), l2 = m >> C::B1;
    return (((k2 << C::B2) + l2) << 6) + (k1 << C::B1) + l1;
  } else {
    return (q << C::B1) + m;
  }
}
template <class C>
RPSF_HD int partner_gid(int gid) {
  int q, m;
  gid_to_qm<C>(gid, q, m);
  return qm_to_gid<C>((C::Q - q) & (C::Q - 1), (C::M - m) & (C::M - 1));
}

// Slot table: tab[(t*NSLOT + s)*2 + member] = gid' of the group.  Special slots (self-paired groups,
// q == 0 or m == 0) come first so they land in the leading threads of slot 0.  Host only.
template <class C>
inline void build_slot_table(uint16_t* tab) {
  static_assert(C::G <= 65536, "gid must fit uint16");
  const int G = C::G;
  bool* seen = new bool[G]();
  int* slots = new int[G];  // pairs, flattened
  int ns = 0;
  int selfs[4], nself = 0;
  for (int g = 0; g < G; ++g)
    if (partner_gid<C>(g) == g) selfs[nself++] = g;
  for (int i = 0; i + 1 < nself; i += 2) {
    slots[2 * ns] = selfs[i], slots[2 * ns + 1] = selfs[i + 1], ++ns;
    seen[selfs[i]] = seen[selfs[i + 1]] = true;
  }
  for (int pass = 0; pass < 2; ++pass)  // pass 0: special pairs, pass 1: the rest
    for (int g = 0; g < G; ++g) {
      if (seen[g]) continue;
      int q, m;
      gid_to_qm<C>(g, q, m);
      if (pass == 0 && q != 0 && m != 0) continue;
      int p = partner_gid<C>(g);
      slots[2 * ns] = g, slots[2 * ns + 1] = p, ++ns;
      seen[g] = seen[p] = true;
    }
  for (int sigma = 0; sigma < ns; ++sigma) {
    int s = sigma / C::T, t = sigma % C::T;
    tab[(t * C::NSLOT + s) * 2 + 0] = (uint16_t)slots[2 * sigma];
    tab[(t * C::NSLOT + s) * 2 + 1] = (uint16_t)slots[2 * sigma + 1];
  }
  delete[] seen;
  delete[] slots;
}

// Pair table of the special slots of two-stage plans (Cfg::ORBIT).  Entry (t, special slot i, j):
//   bits 0-7 x1, 8-15 x2: positions of the pair's two bins among the slot's 2E parked values (member*E + e),
//   bits 16-23 the twiddle index kc of the first bin (W_N^kc), bit 31 valid.  A fixed point has x1 == x2.
RPSF_HD uint32_t pair_entry(int x1, int x2, int kc) { return (uint32_t)x1 | ((uint32_t)x2 << 8) | ((uint32_t)kc << 16) | 0x80000000u; }
template <class C>
RPSF_HD int partner_element(int q, int m, int e) {  // element index of bin -p inside the partner group
  const int k3 = e / C::EB, l3 = e % C::EB;
  const int pk = q == 0 ? (C::EA - k3) % C::EA : C::EA - 1 - k3;
  const int pl = m == 0 ? (C::EB - l3) % C::EB : C::EB - 1 - l3;
  return pk * C::EB + pl;
}
// Host only.  Returns the largest number of pairs any slot needs (must be <= C::NP).
template <class C>
inline int build_pair_table(const uint16_t* tab, uint32_t* pt) {
  int worst = 0;
  if constexpr (C::ORBIT) {
    for (int i = 0; i < C::PT_WORDS; ++i) pt[i] = 0;
    for (int t = 0; t < C::T; ++t)
      for (int s = 0; s < C::NSLOT; ++s) {
        if (C::spec_t(s) == 0) continue;
        uint32_t* out = pt + ((size_t)t * C::NSS + C::special_index(s)) * C::NP;
        const int ga = tab[(t * C::NSLOT + s) * 2], gb = tab[(t * C::NSLOT + s) * 2 + 1];
        int qa, ma, qb, mb, n = 0;
        gid_to_qm<C>(ga, qa, ma);
        gid_to_qm<C>(gb, qb, mb);
        if (partner_gid<C>(ga) != ga) {  // two groups that are each other's partners: bin e of A with its mirror in B
          for (int e = 0; e < C::E; ++e, ++n)
            if (n < C::NP) out[n] = pair_entry(e, C::E + partner_element<C>(qa, ma, e), ma + C::M * (e % C::EB));
        } else {  // two self-paired groups: the orbits inside A, then inside B
          for (int member = 0; member < 2; ++member) {
            const int q = member ? qb : qa, m = member ? mb : ma;
            for (int e = 0; e < C::E; ++e) {
              const int pe = partner_element<C>(q, m, e);
              if (e > pe) continue;
              if (n < C::NP) out[n] = pair_entry(member * C::E + e, member * C::E + pe, m + C::M * (e % C::EB));
              ++n;
            }
          }
        }
        worst = n > worst ? n : worst;
      }
  }
  return worst;
}

// ------------------------------------------------------------------------------------------
// Padding index maps (np.pad modes the kernel evaluates itself; transform.py:119-123)
// ------------------------------------------------------------------------------------------
enum PadMode : int { PAD_CONSTANT = 0, PAD_SYMMETRIC = 1, PAD_REFLECT = 2, PAD_EDGE = 3, PAD_WRAP = 4 };

RPSF_HD int pad_index(int i, int n, int mode) {  // returns -1 for "constant value"
  if (i >= 0 && i < n) return i;
  switch (mode) {
    case PAD_SYMMETRIC: {
      int p = 2 * n, k = i % p;
      if (k < 0) k += p;
      return k < n ? k : p - 1 - k;
    }
    case PAD_REFLECT: {
      if (n == 1) return 0;
      int p = 2 * n - 2, k = i % p;
      if (k < 0) k += p;
      return k < n ? k : p - k;
    }
    case PAD_EDGE: return i < 0 ? 0 : n - 1;
    case PAD_WRAP: {
      int k = i % n;
      return k < 0 ? k + n : k;
    }
    default: return -1;
  }
}

// ------------------------------------------------------------------------------------------
// Thread coordinates in the stage-1 / stage-2 layouts
// ------------------------------------------------------------------------------------------
template <class C>
struct ThreadPos {
  int r_rest, c_rest;  // stage-1 layout: r = r1*2^(A2+AL) + r_rest, c = c1*2^(B2+BL) + c_rest
  int r3, c3;          // last digits (same thread bits in the stage-1 and stage-2 layouts)
  RPSF_HD explicit ThreadPos(int t) {
    // thread index = e * (T / E) + (stage digit bits), e = r3 * 2^BL + c3
    const int e = C::S3 ? (t >> 6) : t;
    r3 = e >> C::BL;
    c3 = e & ((1 << C::BL) - 1);
    if constexpr (C::S3) {
      int r2 = (t >> C::B2) & ((1 << C::A2) - 1);
      int c2 = t & ((1 << C::B2) - 1);
      r_rest = (r2 << C::AL) + r3;
      c_rest = (c2 << C::BL) + c3;
    } else {
      r_rest = r3;
      c_rest = c3;
    }
  }
};

// ------------------------------------------------------------------------------------------
// Stage 1 (digits r1, c1) and stage 2 (digits r2, c2); tw[k] = exp(-2 pi i k / N), k < N
// ------------------------------------------------------------------------------------------
template <class C, bool INV>
RPSF_HD void stage1(int t, cf* v, const cf* __restrict__ tw) {
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  ThreadPos<C> tp(t);
  if constexpr (!INV) {
    fft_axis<C::A1, NCOL, NCOL, 1, false>(v);
    StaticFor<1, NR>::run([&]<int K1>() RPSF_AI {
      cf w = tw[(K1 * tp.r_rest) & (C::N - 1)];
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI { v[K1 * NCOL + C1] = cmul(v[K1 * NCOL + C1], w); });
    });
    fft_axis<C::B1, 1, NR, NCOL, false>(v);
    if constexpr (C::B2 + C::BL > 0) {
      StaticFor<1, NCOL>::run([&]<int L1>() RPSF_AI {
        cf w = tw[(2 * L1 * tp.c_rest) & (C::N - 1)];
        StaticFor<0, NR>::run([&]<int K1>() RPSF_AI { v[K1 * NCOL + L1] = cmul(v[K1 * NCOL + L1], w); });
      });
    }
  } else {
    if constexpr (C::B2 + C::BL > 0) {
      StaticFor<1, NCOL>::run([&]<int L1>() RPSF_AI {
        cf w = tw[(2 * L1 * tp.c_rest) & (C::N - 1)];
        StaticFor<0, NR>::run([&]<int K1>() RPSF_AI { v[K1 * NCOL + L1] = cmulc(v[K1 * NCOL + L1], w); });
      });
    }
    fft_axis<C::B1, 1, NR, NCOL, true>(v);
    StaticFor<1, NR>::run([&]<int K1>() RPSF_AI {
      cf w = tw[(K1 * tp.r_rest) & (C::N - 1)];
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI { v[K1 * NCOL + C1] = cmulc(v[K1 * NCOL + C1], w); });
    });
    fft_axis<C::A1, NCOL, NCOL, 1, true>(v);
  }
}

template <class C, bool INV>
RPSF_HD void stage2(int t, cf* v, const cf* __restrict__ tw) {
  if constexpr (C::S3) {
    constexpr int NR = 1 << C::A2, NCOL = 1 << C::B2;
    ThreadPos<C> tp(t);
    auto col_twiddle = [&]() RPSF_AI {  // W_{2^(B2+BL)}^(l2 c3), only when the last stage owns column bits
      if constexpr (C::BL > 0) {
        StaticFor<1, NCOL>::run([&]<int L2>() RPSF_AI {
          cf w = tw[((2 * L2 * tp.c3) << C::B1) & (C::N - 1)];
          StaticFor<0, NR>::run([&]<int K2>() RPSF_AI {
            v[K2 * NCOL + L2] = INV ? cmulc(v[K2 * NCOL + L2], w) : cmul(v[K2 * NCOL + L2], w);
          });
        });
      }
    };
    if constexpr (!INV) {
      fft_axis<C::A2, NCOL, NCOL, 1, false>(v);
      StaticFor<1, NR>::run([&]<int K2>() RPSF_AI {
        cf w = tw[((K2 * tp.r3) << C::A1) & (C::N - 1)];  // W_{2^(A2+AL)}^(k2 r3)
        StaticFor<0, NCOL>::run([&]<int C2>() RPSF_AI { v[K2 * NCOL + C2] = cmul(v[K2 * NCOL + C2], w); });
      });
      fft_axis<C::B2, 1, NR, NCOL, false>(v);
      col_twiddle();
    } else {
      col_twiddle();
      fft_axis<C::B2, 1, NR, NCOL, true>(v);
      StaticFor<1, NR>::run([&]<int K2>() RPSF_AI {
        cf w = tw[((K2 * tp.r3) << C::A1) & (C::N - 1)];
        StaticFor<0, NCOL>::run([&]<int C2>() RPSF_AI { v[K2 * NCOL + C2] = cmulc(v[K2 * NCOL + C2], w); });
      });
      fft_axis<C::A2, NCOL, NCOL, 1, true>(v);
    }
  }
}

// Last stage: 2-D DFT over (r3, c3) inside every group (register index rho = group*E + k3*EB + l3).
template <class C, bool INV, int GI>
RPSF_HD void stage_last_group(cf* v) {
  fft_axis<C::AL, C::EB, C::EB, 1, INV, GI * C::E>(v);   // along r3 for every l3
  fft_axis<C::BL, 1, C::EA, C::EB, INV, GI * C::E>(v);   // along c3 for every k3
}
template <class C, bool INV>
RPSF_HD void stage_last(cf* v) {
  StaticFor<0, C::P>::run([&]<int GI>() RPSF_AI { stage_last_group<C, INV, GI>(v); });
}

// ------------------------------------------------------------------------------------------
// LDS exchanges.  PART 0 moves real parts, PART 1 imaginary parts.
// X1: 64x64 transpose between register index and lane inside each 64-thread team (stage 1 <-> stage 2).
// X2: stage-2 layout (or stage-1 layout for two-stage plans) <-> last layout, element-major:
//     address = e * X2_STRIDE + gid'.
// ------------------------------------------------------------------------------------------
struct alignas(16) quad { float a, b, c, d; };
template <class C, int PART>
RPSF_HD void x1_write(int t, const cf* v, float* lds) {
  int base = (t >> 6) * (64 * C::X1_ROW) + (t & 63);
  StaticFor<0, 64>::run([&]<int J>() RPSF_AI { lds[base + J * C::X1_ROW] = PART ? v[J].y : v[J].x; });
}
// a thread reads its whole row: sixteen 16-byte reads (measured 2.7x the rate of 64 dword reads)
template <class C, int PART>
RPSF_HD void x1_read(int t, cf* v, const float* lds) {
  const quad* row = reinterpret_cast<const quad*>(lds + (t >> 6) * (64 * C::X1_ROW) + (t & 63) * C::X1_ROW);
  StaticFor<0, 16>::run([&]<int J>() RPSF_AI {
    const quad f = row[J];
    if (PART) v[4 * J].y = f.a, v[4 * J + 1].y = f.b, v[4 * J + 2].y = f.c, v[4 * J + 3].y = f.d;
    else v[4 * J].x = f.a, v[4 * J + 1].x = f.b, v[4 * J + 2].x = f.c, v[4 * J + 3].x = f.d;
  });
}

template <class C>
RPSF_HD int x2_mid_base(int t) {  // address of register 0 of a stage-2-layout (or stage-1-layout) thread
  if constexpr (C::S3) return (t >> 6) * C::X2_STRIDE + (t & 63);
  else return t * C::X2_STRIDE;
}
template <class C, int PART>
RPSF_HD void x2_mid_write(int t, const cf* v, float* lds) {
  int base = x2_mid_base<C>(t);
  constexpr int RS = C::S3 ? 64 : 1;
  StaticFor<0, 64>::run([&]<int J>() RPSF_AI { lds[base + J * RS] = PART ? v[J].y : v[J].x; });
}
template <class C, int PART>
RPSF_HD void x2_mid_read(int t, cf* v, const float* lds) {
  int base = x2_mid_base<C>(t);
  constexpr int RS = C::S3 ? 64 : 1;
  StaticFor<0, 64>::run([&]<int J>() RPSF_AI {
    float f = lds[base + J * RS];
    if (PART) v[J].y = f; else v[J].x = f;
  });
}
// The thread's P group ids in the last layout (slot-major, member minor), two 16-bit ids per register:
// at N = 128 a thread owns 32 groups and unpacked ids alone would take 32 of the 256 registers.
template <class C>
struct GroupIds {
  uint32_t w[C::P / 2];
  RPSF_HD int operator[](int i) const { return (int)((w[i >> 1] >> ((i & 1) * 16)) & 0xffffu); }
  RPSF_HD void load(const uint16_t* __restrict__ tab, int t) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tab + (size_t)t * C::P);
    StaticFor<0, C::P / 2>::run([&]<int I>() RPSF_AI { w[I] = src[I]; });
  }
};

// last layout: gids[g] for the thread's P groups (slot-major, member minor)
template <class C, int PART>
RPSF_HD void x2_last_read(const GroupIds<C>& gids, cf* v, const float* lds) {
  StaticFor<0, C::P>::run([&]<int GI>() RPSF_AI {
    StaticFor<0, C::E>::run([&]<int EE>() RPSF_AI {
      float f = lds[EE * C::X2_STRIDE + gids[GI]];
      if (PART) v[GI * C::E + EE].y = f; else v[GI * C::E + EE].x = f;
    });
  });
}
template <class C, int PART>
RPSF_HD void x2_last_write(const GroupIds<C>& gids, const cf* v, float* lds) {
  StaticFor<0, C::P>::run([&]<int GI>() RPSF_AI {
    StaticFor<0, C::E>::run([&]<int EE>() RPSF_AI {
      lds[EE * C::X2_STRIDE + gids[GI]] = PART ? v[GI * C::E + EE].y : v[GI * C::E + EE].x;
    });
  });
}

// ------------------------------------------------------------------------------------------
// Frequency-domain step: unpack real spectrum, multiply by folded K, re-pack.
// ------------------------------------------------------------------------------------------
struct PairOut { cf a, b; };
// two-sided: bins p (value za) and -p (value zb); ka = K'_h(p), kb = K'_h(p + (0,N/2)); w = W_N^kc(p)
RPSF_HD PairOut pair_op(cf za, cf zb, cf ka, cf kb, cf w) {
  cf zbc = cconj(zb);
  cf e2 = za + zbc;
  cf o2 = mul_mi(za - zbc);
  cf wo = cmul(w, o2);
  cf y1 = cmul(e2 + wo, ka);
  cf y2 = cmul(e2 - wo, kb);
  cf ep = y1 + y2;
  cf op = cmulc(y1 - y2, w);
  PairOut r;
  r.a = ep + mul_pi(op);
  r.b = cconj(ep) + mul_pi(cconj(op));
  return r;
}

// ------------------------------------------------------------------------------------------
// Packed K layout (device memory, per patch) and its streaming through registers
//   g : 32 "pair words" of 16 bytes per thread, word w of thread t at cf index (w*T + t)*2.
//       w = S*E + e addresses bins e of slot S.  General slot: (K'_h(p), K'_h(p + (0,N/2))) for p = bin e of
//       member A - exactly the two factors of pair_op(A_e, B_{E-1-e}).  Special-format slot:
//       (K'_h(A_e), K'_h(B_e)), and the Nyquist-side factors (K'_h(A_e + (0,N/2)), K'_h(B_e + (0,N/2))) sit in
//   gs: cf index prefix(S)*2E + (e*spec_t(S) + t)*2, for t < spec_t(S)  (three-stage plans).
//       Two-stage plans have no gs: a special slot takes NP ordinary words, one per bin pair of its pair table
//       (Cfg::ORBIT, build_pair_table), and the stream is padded to whole chunks.
// K is consumed in chunks of C::KCH pair words (8, or 4 for the plans with tiny groups whose other
// temporaries are larger): 16-byte streaming loads, 1 KiB per wave instruction.
// ------------------------------------------------------------------------------------------
template <class C, int CI>
RPSF_HD void load_k_chunk(int t, cf* k, const cf* __restrict__ g) {
  StaticFor<0, C::KCH>::run([&]<int I>() RPSF_AI {
    load_stream16(g + ((size_t)(CI * C::KCH + I) * C::T + t) * 2, k[2 * I], k[2 * I + 1]);
  });
}
#define RPSF_KRING 1
template <class C>
struct KRing {
  static constexpr int DEPTH = C::KDEPTH > RPSF_KRING ? C::KDEPTH : RPSF_KRING;  // chunks in flight
  cf k[DEPTH][2 * C::KCH];
};
template <class C>
RPSF_HD void kring_fill(int t, KRing<C>& r, const cf* __restrict__ g) {
  StaticFor<0, KRing<C>::DEPTH>::run([&]<int D>() RPSF_AI { load_k_chunk<C, D>(t, r.k[D], g); });
}

// Special slot: parked in thread-private LDS columns (scratch[(h*E + e)*PARK_STRIDE + t]); every bin fetches
// its partner by computed address, so no register temporaries and no 4-way selects are needed.  (The special
// waves are the ones every barrier waits for: a register-select version of this path cost 8 % of the N=256
// kernel.)
template <class C, int S>
RPSF_HD void special_slot_park(int t, const cf* v, cf* scratch) {
  constexpr int E = C::E;
  StaticFor<0, 2 * E>::run([&]<int I>() RPSF_AI { scratch[(size_t)I * C::PARK_STRIDE + t] = v[(2 * S) * E + I]; });
}
template <class C, int S, int EE>
RPSF_HD void special_pair_parked(int t, const GroupIds<C>& gids, cf* v, cf ka_a, cf ka_b, cf ks_a, cf ks_b,
                                 const cf* __restrict__ tw, const cf* scratch) {
  constexpr int E = C::E, EA = C::EA, EB = C::EB, K3 = EE / EB, L3 = EE % EB;
  cf* za = v + (2 * S) * E;
  cf* zb = za + E;
  const int ga = gids[2 * S], gb = gids[2 * S + 1];
  const bool self = partner_gid<C>(ga) == ga;
  int qa, ma, qb, mb;
  gid_to_qm<C>(ga, qa, ma);
  gid_to_qm<C>(gb, qb, mb);
  auto partner = [&](int q, int m, int own_member) RPSF_AI {  // bin of -p: digit negated if the low part is zero, else reversed
    const int k3 = q == 0 ? (EA - K3) % EA : EA - 1 - K3;
    const int l3 = m == 0 ? (EB - L3) % EB : EB - 1 - L3;
    const int member = self ? own_member : 1 - own_member;
    return scratch[(size_t)(member * E + k3 * EB + l3) * C::PARK_STRIDE + t];
  };
  const cf pa = partner(qa, ma, 0), pb = partner(qb, mb, 1);
  za[EE] = pair_op(za[EE], pa, ka_a, ks_a, tw[ma + C::M * L3]).a;
  zb[EE] = pair_op(zb[EE], pb, ka_b, ks_b, tw[mb + C::M * L3]).a;
}

// The same step in two passes, for plans whose K chunk is exactly one slot (E == KCH): r.a of pair_op is linear in
// (ka, ks), so pass 1 applies the main-stream factor, each word's registers are refilled with its side factors as
// soon as it is consumed, and pass 2 adds the side term from the parked originals.  One-pass code reads the side
// factors word by word inside the loop - eight dependent memory round trips in the waves every barrier waits for.
template <class C, int S, int EE, int MEMBER>
RPSF_HD void special_terms(int t, const GroupIds<C>& gids, const cf* __restrict__ tw, const cf* scratch, cf z, cf& e2,
                           cf& wo, cf& w) {
  constexpr int E = C::E, EA = C::EA, EB = C::EB, K3 = EE / EB, L3 = EE % EB;
  const int ga = gids[2 * S], gm = gids[2 * S + MEMBER];
  const bool self = partner_gid<C>(ga) == ga;
  int q, m;
  gid_to_qm<C>(gm, q, m);
  const int k3 = q == 0 ? (EA - K3) % EA : EA - 1 - K3;  // bin of -p: digit negated if the low part is zero, else reversed
  const int l3 = m == 0 ? (EB - L3) % EB : EB - 1 - L3;
  const int member = self ? MEMBER : 1 - MEMBER;
  const cf pc = cconj(scratch[(size_t)(member * E + k3 * EB + l3) * C::PARK_STRIDE + t]);
  w = tw[m + C::M * L3];
  e2 = z + pc;
  wo = cmul(w, mul_mi(z - pc));
}
template <class C, int S, int EE>
RPSF_HD void special_pass1(int t, const GroupIds<C>& gids, cf* v, cf ka_a, cf ka_b, const cf* __restrict__ tw,
                           const cf* scratch) {
  cf* za = v + (2 * S) * C::E;
  cf* zb = za + C::E;
  cf e2, wo, w;
  special_terms<C, S, EE, 0>(t, gids, tw, scratch, za[EE], e2, wo, w);
  za[EE] = cmul(e2 + wo, ka_a);
  special_terms<C, S, EE, 1>(t, gids, tw, scratch, zb[EE], e2, wo, w);
  zb[EE] = cmul(e2 + wo, ka_b);
}
template <class C, int S, int EE>
RPSF_HD void special_pass2(int t, const GroupIds<C>& gids, cf* v, cf ks_a, cf ks_b, const cf* __restrict__ tw,
                           const cf* scratch) {
  constexpr int E = C::E;
  cf* za = v + (2 * S) * E;
  cf* zb = za + E;
  cf e2, wo, w;
  special_terms<C, S, EE, 0>(t, gids, tw, scratch, scratch[(size_t)EE * C::PARK_STRIDE + t], e2, wo, w);
  cf y2 = cmul(e2 - wo, ks_a);
  za[EE] = (za[EE] + y2) + mul_pi(cmulc(za[EE] - y2, w));
  special_terms<C, S, EE, 1>(t, gids, tw, scratch, scratch[(size_t)(E + EE) * C::PARK_STRIDE + t], e2, wo, w);
  y2 = cmul(e2 - wo, ks_b);
  zb[EE] = (zb[EE] + y2) + mul_pi(cmulc(zb[EE] - y2, w));
}

// Two-stage plans: pair J of special slot S, both bins through the parking area (see Cfg::ORBIT).
template <class C, int S, int J>
RPSF_HD void orbit_pair(int t, const uint32_t* pt, cf ka, cf kb, const cf* __restrict__ tw, cf* scratch) {
  const uint32_t ent = pt[((size_t)t * C::NSS + C::special_index(S)) * C::NP + J];
  const int x1 = ent & 0xff, x2 = (ent >> 8) & 0xff, kc = (ent >> 16) & 0xff;
  const cf z1 = scratch[(size_t)x1 * C::PARK_STRIDE + t], z2 = scratch[(size_t)x2 * C::PARK_STRIDE + t];
  const PairOut o = pair_op(z1, z2, ka, kb, tw[kc]);
  if (ent >> 31) {
    scratch[(size_t)x1 * C::PARK_STRIDE + t] = o.a;
    if (x2 != x1) scratch[(size_t)x2 * C::PARK_STRIDE + t] = o.b;
  }
}
template <class C, int S>
RPSF_HD void special_slot_unpark(int t, cf* v, const cf* scratch) {
  constexpr int E = C::E;
  StaticFor<0, 2 * E>::run([&]<int I>() RPSF_AI { v[(2 * S) * E + I] = scratch[(size_t)I * C::PARK_STRIDE + t]; });
}

// The whole frequency step of one thread: 32 pair words in 4 chunks.  r holds chunk 0 (loaded by the caller
// before the exchange into the last layout); each later chunk is requested as soon as its buffer is free.
// scratch: thread-private LDS columns for the parked special path.
// FUSE: the last stage is a DFT inside each group, so when chunks hold whole slots it runs slot by slot
// around the multiplication (forward DFT of the chunk's groups, multiply, request the next chunk, inverse
// DFT of the same groups): the K round trip of chunk i+1 hides behind the butterflies of chunks i and i+1.
template <class C, bool FUSE>
RPSF_HD void pointwise(int t, const GroupIds<C>& gids, cf* v, KRing<C>& r, const cf* __restrict__ g,
                       const cf* __restrict__ gs, const cf* __restrict__ tw, cf* scratch, const uint32_t* pt) {
  constexpr int E = C::E, EB = C::EB;
  StaticFor<0, C::NWORDS / C::KCH>::run([&]<int CI>() RPSF_AI {
    constexpr int DEPTH = KRing<C>::DEPTH;
    cf* rk = r.k[CI % DEPTH];
    StaticFor<0, C::KCH>::run([&]<int I>() RPSF_AI {
      constexpr int W = CI * C::KCH + I;
      if constexpr (W < C::NWORDS_USED) {
        constexpr int S = C::word_slot(W), R = W - C::word_base(S), ST = C::spec_t(S);
        constexpr bool WIDE = C::ORBIT && ST > 0;            // a special slot of a two-stage plan: word R is pair R of its table
        constexpr int EE = WIDE ? 0 : R;
        if constexpr (FUSE && R == 0) {  // first word of a slot: forward DFT of its two groups
          stage_last_group<C, false, 2 * S>(v);
          stage_last_group<C, false, 2 * S + 1>(v);
        }
        cf* za = v + (2 * S) * E;
        cf* zb = za + E;
        bool special = false;
        if constexpr (ST > 0) special = (t & ~(C::WAVE - 1)) < ST;  // uniform over a 64-thread team
        if (!special) {
          int qa, ma;
          gid_to_qm<C>(gids[2 * S], qa, ma);
          PairOut o = pair_op(za[EE], zb[E - 1 - EE], rk[2 * I], rk[2 * I + 1], tw[ma + C::M * (EE % EB)]);
          za[EE] = o.a;
          zb[E - 1 - EE] = o.b;
        } else if constexpr (ST > 0) {
          static_assert(!C::S3 || C::spec_t(S) <= C::PARK_STRIDE, "parking area sized by slot 0");
          if constexpr (R == 0) special_slot_park<C, S>(t, v, scratch);
          if constexpr (WIDE) {
            orbit_pair<C, S, R>(t, pt, rk[2 * I], rk[2 * I + 1], tw, scratch);
            if constexpr (R == C::NP - 1) special_slot_unpark<C, S>(t, v, scratch);
          } else if constexpr (C::TWO_PASS_SPECIAL) {
            static_assert(C::KCH == E && I == EE, "the chunk in the ring is this slot");
            const cf* gsp = gs + (size_t)C::spec_prefix(S) * 2 * E + ((size_t)EE * ST + t) * 2;
            special_pass1<C, S, EE>(t, gids, v, rk[2 * I], rk[2 * I + 1], tw, scratch);
            load_stream16(gsp, rk[2 * I], rk[2 * I + 1]);  // the word's registers now wait for its side factors
            if constexpr (EE == E - 1)
              StaticFor<0, E>::run([&]<int E2>() RPSF_AI {
                special_pass2<C, S, E2>(t, gids, v, rk[2 * E2], rk[2 * E2 + 1], tw, scratch);
                if constexpr (CI + DEPTH < C::NWORDS / C::KCH)  // ... and then for the next chunk's word (instead of the refill below)
                  load_stream16(g + ((size_t)((CI + DEPTH) * C::KCH + E2) * C::T + t) * 2, rk[2 * E2], rk[2 * E2 + 1]);
              });
          } else {
            const cf* gsp = gs + (size_t)C::spec_prefix(S) * 2 * E + ((size_t)EE * ST + t) * 2;
            special_pair_parked<C, S, EE>(t, gids, v, rk[2 * I], rk[2 * I + 1], gsp[0], gsp[1], tw, scratch);
          }
        }
      }
    });
    if constexpr (CI + DEPTH < C::NWORDS / C::KCH) {
      constexpr int SC = C::word_slot(CI * C::KCH), STC = C::spec_t(SC);
      bool refilled = false;  // the two-pass special path has already asked for the next chunk, word by word
      if constexpr (C::TWO_PASS_SPECIAL && STC > 0) refilled = (t & ~(C::WAVE - 1)) < STC;
      if (!refilled) load_k_chunk<C, CI + DEPTH>(t, rk, g);
    }
    StaticFor<0, C::KCH>::run([&]<int I>() RPSF_AI {  // slots that ended in this chunk: inverse DFT of their groups
      constexpr int W = CI * C::KCH + I;
      if constexpr (W < C::NWORDS_USED) {
        constexpr int S = C::word_slot(W), R = W - C::word_base(S);
        if constexpr (FUSE && R == C::slot_words(S) - 1) {
          stage_last_group<C, true, 2 * S>(v);
          stage_last_group<C, true, 2 * S + 1>(v);
        }
      }
    });
  });
}
// Last stage forward, multiplication by K, last stage inverse.
template <class C>
RPSF_HD void freq_step(int t, const GroupIds<C>& gids, cf* v, KRing<C>& r, const cf* __restrict__ g,
                       const cf* __restrict__ gs, const cf* __restrict__ tw, cf* scratch, const uint32_t* pt) {
  if constexpr (C::FUSE_LAST) {
    pointwise<C, true>(t, gids, v, r, g, gs, tw, scratch, pt);
  } else {
    stage_last<C, false>(v);
    pointwise<C, false>(t, gids, v, r, g, gs, tw, scratch, pt);
    stage_last<C, true>(v);
  }
}

// Value of the packed K arrays at (thread t, register rho).  kfull = one patch of the caller's
// transfer kernel, N x N complex64 (IndexedCube values, transform.py:164).  which = 0: g, 1: gs.
// (KF: where K comes from - the caller's full array, or a functor that evaluates transform.py:78-82 from the two PSF spectra on the fly, so that
// construct -> pack never writes or reads the full K: rpsf_kernels.hpp, KFromSpectra)
RPSF_HD cf k_at(const cf* k, int i) { return k[i]; }
template <class F>
RPSF_HD cf k_at(const F& f, int i) { return f(i); }
template <class C, class KF>
RPSF_HD cf kh_at(const KF& kfull, int kr, int kc) {
  cf a = k_at(kfull, kr * C::N + kc);
  cf b = k_at(kfull, ((C::N - kr) & (C::N - 1)) * C::N + ((C::N - kc) & (C::N - 1)));
  return cf{(a.x + b.x) * (0.5f * C::SCALE), (a.y - b.y) * (0.5f * C::SCALE)};
}
template <class C>
RPSF_HD bool slot_is_special(int s, int t) {
  return (t & ~(C::WAVE - 1)) < C::spec_t(s);
}
template <class C, class KF>
RPSF_HD cf pack_value(const KF& kfull, const uint16_t* __restrict__ tab, const uint32_t* __restrict__ pt, int t,
                      int rho, int which) {
  // rho = 2*w + b, w = word index in the thread's stream;  which = 0: g, 1: gs (special-format slots of
  // three-stage plans only).  See the layout comment above load_k_chunk.
  const int w = rho >> 1, b = rho & 1;
  if (w >= C::NWORDS_USED) return cf{0.f, 0.f};  // padding to whole chunks
  const int s = C::word_slot(w), r = w - C::word_base(s);
  const bool special = slot_is_special<C>(s, t);
  int member = special ? b : 0, e = r;  // general words describe bin e of member A only
  bool nyquist_side = special ? which == 1 : b == 1;
  if (C::ORBIT && special) {  // word r = pair r of the slot's table: (K_h(p), K_h(p + (0,N/2))) of its first bin p
    const uint32_t ent = pt[((size_t)t * C::NSS + C::special_index(s)) * C::NP + r];
    if (!(ent >> 31)) return cf{0.f, 0.f};
    const int x1 = ent & 0xff;
    member = x1 / C::E, e = x1 % C::E, nyquist_side = b == 1;
  }
  int q, m;
  gid_to_qm<C>(tab[(t * C::NSLOT + s) * 2 + member], q, m);
  const int kr = q + C::Q * (e / C::EB), kc = m + C::M * (e % C::EB);
  return kh_at<C>(kfull, kr, nyquist_side ? kc + C::NC : kc);
}

// ------------------------------------------------------------------------------------------
// Image side: gather + window (transform.py:151-163) and window + overlap-add (transform.py:165-169)
// ------------------------------------------------------------------------------------------
struct ImageView {
  const float* img;  // rows [row0, row0 + rows) of the H x W image, row stride ld
  int H, W, ld;
  int pad_mode;
  float pad_value;
  int row0, rows;
};
// Overlap-add target.  plane_stride == 0: one H x W image accumulated with float atomics (any
// corner list).  plane_stride != 0: four "colour planes"; a patch stores (plain, coalesced) into
// the plane of its lattice parity class, inside which patches never overlap, and a small kernel
// sums the planes afterwards (regular half-overlap lattices only).
struct OutView {
  float* out;  // rows [row0, row0 + rows) of the H x W output (plane 0 in plane mode), row stride ld
  int H, W, ld;
  int row0, rows;
  size_t plane_stride;  // floats
  float* sink;          // >= 128 floats of write-only scratch: where the pixels a rim patch hangs over the edge go
};

// A patch may use 8-byte vector accesses when it lies inside the resident window and pixel pairs are aligned.
template <class C>
RPSF_HD bool patch_inside(int pr, int pc, int H, int W, int row0, int rows) {
  return pr >= 0 && pc >= 0 && pr + C::N <= H && pc + C::N <= W && pr >= row0 && pr + C::N <= row0 + rows;
}
RPSF_HD bool pairs_aligned(const void* base, int ld, int pc) {
  return ((ld | pc) & 1) == 0 && (reinterpret_cast<uintptr_t>(base) & 7) == 0;
}

// Boundary patches: the np.pad index maps of the patch's N rows and N columns, built once per patch
// in LDS (maps[0..N) = resident row index or -1, maps[N..2N) = column or -1).
template <class C>
RPSF_HD void build_pad_maps(int t, int* maps, const ImageView& im, int pr, int pc) {
  for (int i = t; i < 2 * C::N; i += C::T) {
    int v;
    if (i < C::N) {
      v = pad_index(pr + i, im.H, im.pad_mode);
      if (v >= 0) {
        v -= im.row0;
        if (v < 0 || v >= im.rows) v = -1;  // not resident: treated as constant fill
      }
    } else {
      v = pad_index(pc + (i - C::N), im.W, im.pad_mode);
    }
    maps[i] = v;
  }
}

// Interior patch, step 1: issue the 64 eight-byte loads (no table needed yet, so this can start before
// the twiddle/window tables have been staged in LDS).
template <class C>
RPSF_HD void load_patch_raw(int t, cf* v, const ImageView& im, int pr, int pc) {
  ThreadPos<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  const float* base = im.img + (size_t)(pr - im.row0) * im.ld + pc;
  StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
    int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
    const float* row = base + (size_t)r * im.ld;
    StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
      int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
      v[R1 * NCOL + C1] = *reinterpret_cast<const cf*>(row + 2 * c);
    });
  });
}
// Interior patch, step 2: sine window (transform.py:151-155,163).
template <class C>
RPSF_HD void window_patch(int t, cf* v, const float* __restrict__ win) {
  ThreadPos<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
    float wr = win[(R1 << (C::A2 + C::AL)) + tp.r_rest];
    StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
      int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
      cf w2 = *reinterpret_cast<const cf*>(win + 2 * c);
      v[R1 * NCOL + C1] = v[R1 * NCOL + C1] * (w2 * wr);
    });
  });
}

template <class C>
RPSF_HD void load_patch(int t, cf* v, const ImageView& im, int pr, int pc, const float* __restrict__ win,
                        bool fast, const int* maps) {
  ThreadPos<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  if (fast) {
    load_patch_raw<C>(t, v, im, pr, pc);
    window_patch<C>(t, v, win);
  } else {
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
      float wr = win[r];
      int yl = maps[r];
      const float* row = im.img + (size_t)(yl < 0 ? 0 : yl) * im.ld;
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
        int xa = maps[C::N + 2 * c], xb = maps[C::N + 2 * c + 1];
        float p0 = row[xa < 0 ? 0 : xa], p1 = row[xb < 0 ? 0 : xb];  // always in bounds; select afterwards
        p0 = (yl < 0 || xa < 0) ? im.pad_value : p0;
        p1 = (yl < 0 || xb < 0) ? im.pad_value : p1;
        v[R1 * NCOL + C1] = cf{p0 * (wr * win[2 * c]), p1 * (wr * win[2 * c + 1])};
      });
    });
  }
}

// ADD(ptr, value): accumulate one pixel (float atomic on the device, plain add in the emulator).
template <class C, class ADD>
RPSF_HD void store_patch(int t, const cf* v, const OutView& ov, int plane, int pr, int pc,
                         const float* __restrict__ win, ADD&& add) {
  ThreadPos<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  const bool planes = ov.plane_stride != 0;
  float* dst = ov.out + (size_t)plane * ov.plane_stride;
  if (planes && patch_inside<C>(pr, pc, ov.H, ov.W, ov.row0, ov.rows) && pairs_aligned(dst, ov.ld, pc)) {
    float* base = dst + (size_t)(pr - ov.row0) * ov.ld + pc;
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
      float wr = win[r];
      float* row = base + (size_t)r * ov.ld;
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
        cf w2 = *reinterpret_cast<const cf*>(win + 2 * c);
        store_stream8(row + 2 * c, v[R1 * NCOL + C1] * (w2 * wr));
      });
    });
    return;
  }
  if constexpr (C::S3) {  // large patches: one predicated store per pixel (half a 256-px rim patch is outside:
                          // sending that to one sink line serialises in L2 - measured 217 -> 252 us)
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
      float wr = win[r];
      int y = pr + r, yl = y - ov.row0;
      if (y >= 0 && y < ov.H && yl >= 0 && yl < ov.rows) {
        float* row = dst + (size_t)yl * ov.ld;
        StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
          int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
          int x0 = pc + 2 * c;
          cf val = v[R1 * NCOL + C1];
          float a0 = val.x * (wr * win[2 * c]), a1 = val.y * (wr * win[2 * c + 1]);
          if (planes) {
            if (x0 >= 0 && x0 < ov.W) row[x0] = a0;
            if (x0 + 1 >= 0 && x0 + 1 < ov.W) row[x0 + 1] = a1;
          } else {
            if (x0 >= 0 && x0 < ov.W) add(row + x0, a0);
            if (x0 + 1 >= 0 && x0 + 1 < ov.W) add(row + x0 + 1, a1);
          }
        });
      }
    });
    return;
  }
  // Small patches (several per wave, so the lanes of a wave disagree about every condition): no branch per
  // pixel - a pixel outside the image or outside the resident rows is written to a sink instead
  // (1024^2 / N=32: 40.5 -> 35 us; 512^2 / N=64: 43 -> 36 us).
  float* sink = ov.sink + 2 * (t & 63);
  StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
    int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
    float wr = win[r];
    int y = pr + r, yl = y - ov.row0;
    const bool row_ok = y >= 0 && y < ov.H && yl >= 0 && yl < ov.rows;
    float* row = dst + (size_t)(row_ok ? yl : 0) * ov.ld;
    StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
      int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
      int x0 = pc + 2 * c;
      cf val = v[R1 * NCOL + C1];
      float a0 = val.x * (wr * win[2 * c]), a1 = val.y * (wr * win[2 * c + 1]);
      float* p0 = (row_ok && x0 >= 0 && x0 < ov.W) ? row + x0 : sink;
      float* p1 = (row_ok && x0 + 1 >= 0 && x0 + 1 < ov.W) ? row + x0 + 1 : sink + 1;
      if (planes) {
        *p0 = a0;
        *p1 = a1;
      } else {
        add(p0, a0);
        add(p1, a1);
      }
    });
  });
}

// ------------------------------------------------------------------------------------------
// Direct overlap-add (three-stage plans on a regular lattice).  Every lattice tile (N/2 x N/2 pixels) has up to
// four contributing patches, one per colour class.  The plan cuts the processing order into 8 chunks, one per
// XCD, and gives every tile to the chunk that holds most of its contributors: those accumulate straight into
// the output image, in colour order, through that XCD's L2 - the first one stores, the later ones wait for
// their predecessor's flag, read the running sum (L1-bypassing loads), add and store.  Contributors from other
// chunks ("side") store into their colour plane as before and a small fix-up kernel adds them afterwards.
// Quadrant word (one per patch quadrant, plan constant): bits 0-1 mode, 2-4 rank among the tile's direct
// contributors, 8-31 tile index; bits 5 and 7 are set at run time (accumulate onto what the output holds / sent
// to the plane although the table said direct).
// Tile flag word: (epoch << 8) | (output tile initialised << 3) | direct contributors done.
// ------------------------------------------------------------------------------------------
enum QuadMode : uint32_t { QUAD_NONE = 0, QUAD_SIDE = 1, QUAD_DIRECT = 2 };
RPSF_HD uint32_t quad_word(uint32_t mode, uint32_t rank, uint32_t tile) { return mode | (rank << 2) | (tile << 8); }
RPSF_HD uint32_t quad_mode(uint32_t w) { return w & 3u; }
RPSF_HD uint32_t quad_rank(uint32_t w) { return (w >> 2) & 7u; }
RPSF_HD uint32_t quad_tile(uint32_t w) { return w >> 8; }
constexpr uint32_t QUAD_ACC = 0x20u, QUAD_DEMOTED = 0x80u;

// qw[q], q = 2*(bottom half) + (right half).  LOAD2 / LOAD1: coherent (L1-bypassing) 8- and 4-byte loads of the
// running sum.  A quadrant whose word says QUAD_NONE lies outside the image and is skipped.
template <class C, class LOAD2, class LOAD1>
RPSF_HD void store_patch_direct(int t, const cf* v, const OutView& pv, const OutView& dv, int plane, int pr, int pc,
                                const float* __restrict__ win, const uint32_t* qw, LOAD2&& load2, LOAD1&& load1) {
  static_assert(C::S3 && C::A1 >= 1 && C::B1 >= 1, "the top row / column bits must be register digits");
  ThreadPos<C> tp(t);
  constexpr int NR = 1 << C::A1, NCOL = 1 << C::B1;
  float* pbase = pv.out + (size_t)plane * pv.plane_stride;
  const bool fast = patch_inside<C>(pr, pc, dv.H, dv.W, dv.row0, dv.rows) && pairs_aligned(pbase, pv.ld, pc) &&
                    pairs_aligned(dv.out, dv.ld, pc);
  if (fast) {
    float* prow0 = pbase + (size_t)(pr - pv.row0) * pv.ld + pc;
    float* drow0 = dv.out + (size_t)(pr - dv.row0) * dv.ld + pc;
    StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
      const int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
      const float wr = win[r];
      cf old[NCOL];
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        constexpr int Q = 2 * (R1 >= NR / 2) + (C1 >= NCOL / 2);
        const int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
        old[C1] = cf{0.f, 0.f};
        if (quad_mode(qw[Q]) == QUAD_DIRECT && (qw[Q] & QUAD_ACC)) old[C1] = load2(drow0 + (size_t)r * dv.ld + 2 * c);
      });
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        constexpr int Q = 2 * (R1 >= NR / 2) + (C1 >= NCOL / 2);
        const int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
        const cf w2 = *reinterpret_cast<const cf*>(win + 2 * c);
        const cf val = v[R1 * NCOL + C1] * (w2 * wr);
        if (quad_mode(qw[Q]) == QUAD_DIRECT) *reinterpret_cast<cf*>(drow0 + (size_t)r * dv.ld + 2 * c) = old[C1] + val;
        else if (quad_mode(qw[Q]) == QUAD_SIDE) store_stream8(prow0 + (size_t)r * pv.ld + 2 * c, val);
      });
    });
    return;
  }
  StaticFor<0, NR>::run([&]<int R1>() RPSF_AI {
    const int r = (R1 << (C::A2 + C::AL)) + tp.r_rest;
    const float wr = win[r];
    const int y = pr + r, yl = y - dv.row0;
    if (y >= 0 && y < dv.H && yl >= 0 && yl < dv.rows) {
      StaticFor<0, NCOL>::run([&]<int C1>() RPSF_AI {
        constexpr int Q = 2 * (R1 >= NR / 2) + (C1 >= NCOL / 2);
        const uint32_t mode = quad_mode(qw[Q]);
        const bool rmw = (qw[Q] & QUAD_ACC) != 0;
        const int c = (C1 << (C::B2 + C::BL)) + tp.c_rest;
        const int x0 = pc + 2 * c;
        const cf val = v[R1 * NCOL + C1];
        const float a0 = val.x * (wr * win[2 * c]), a1 = val.y * (wr * win[2 * c + 1]);
        const bool in0 = x0 >= 0 && x0 < dv.W, in1 = x0 + 1 >= 0 && x0 + 1 < dv.W;
        if (mode == QUAD_DIRECT) {
          float* row = dv.out + (size_t)yl * dv.ld;
          if (in0) row[x0] = (rmw ? load1(row + x0) : 0.f) + a0;
          if (in1) row[x0 + 1] = (rmw ? load1(row + x0 + 1) : 0.f) + a1;
        } else if (mode == QUAD_SIDE) {
          float* row = pbase + (size_t)(y - pv.row0) * pv.ld;
          if (in0) row[x0] = a0;
          if (in1) row[x0 + 1] = a1;
        }
      });
    }
  });
}

// Sum of the colour planes at one pixel.  cover = 4-bit mask of the classes that have a patch over the
// pixel's lattice tile (planes are never cleared, so classes without a patch must not be read).
RPSF_HD float sum_planes_at(const float* planes, size_t plane_stride, size_t offset, int cover) {
  float acc = 0.0f;
  if (cover & 1) acc += planes[offset];
  if (cover & 2) acc += planes[plane_stride + offset];
  if (cover & 4) acc += planes[2 * plane_stride + offset];
  if (cover & 8) acc += planes[3 * plane_stride + offset];
  return acc;
}

// Plans compiled into the library
// digits chosen so that the 64 lanes of a wave cover 2 rows x 32 packed columns (2 x 256 B contiguous)
using Cfg256 = Cfg<8, 4, 1, 3, 2, 5>;
using Cfg128 = Cfg<7, 5, 1, 1, 1, 5>;
// two-stage plans: all row bits but one (or none) in stage 1, column bits in the last stage, so that the
// threads of a patch own consecutive packed columns (the earlier row-only last stage made a wave touch 32-64 rows)
using Cfg64 = Cfg<6, 5, 0, 1, 1, 0, 4>;  // 32 threads: 2 rows x 16 packed columns (128 B)
using Cfg32 = Cfg<5, 5, 0, 0, 1, 0, 3>;  // 8 threads: 8 packed columns (64 B) of one row
using Cfg16 = Cfg<4, 4, 0, 0, 2, 0, 1>;  // 2 threads

}  // namespace rpsf
