// rpsf_plan3.hpp - host side of the third kernel generation (rpsf_kernels3.hpp): how a regular half-overlap lattice of patches is cut
// into REGIONS of output pixels (one workgroup each) and every region into an ordered list of JOBS (one slab = PPW patches of one lattice
// row and one column parity, one wave each).  Plain C++: rpsf.hip builds the lists at plan creation, tests/emu/emu3.cpp replays them.
//
// Geometry (h = N / 2; everything below in pixels relative to the lattice origin = the corner of patch (0, 0)):
//   patch (a, b) has its corner at (a h, b h) and covers the row bands a, a + 1 and the column bands b, b + 1 (band i = [i h, (i + 1) h)).
//   A region takes the lattice rows [a0, a1] and the patch columns [b0, b1] and OWNS the bands that all their contributors lie in:
//   columns b0 + 1 .. b1, rows a0 + 1 .. a1 (plus row band a0 if a0 = 0 and a1 + 1 if a1 is the last lattice row).  Neighbouring regions
//   share one lattice row / patch column, which both compute (no hand-off between workgroups).  The patch columns are extended by one
//   virtual column on either side (-1 and nlj: patches that contribute zeros), so that the column rule has no special cases.
// Order inside a region = the order in which the contributions to a pixel are added (fixed, so results are bit-reproducible and do not
// depend on how the lattice was cut): lattice rows top-down; inside a row first the patches of even global column (phase A), then
// the odd ones (phase B).  A phase-A job STOREs the row's lower band (its first touch), everything else is added; a phase-B job
// flushes the finished upper band of its columns to the output image.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace rpsf {

struct Job3 {  // 64 bytes, read through the scalar cache
  int32_t row, col;        // slab corner
  int32_t ring_col;        // its first column in the region's ring
  uint32_t flags;          // J3_* below
  int32_t dep0, dep1;      // jobs of the same region (index in the region) whose adds - and flush - must be over first, -1: none
  int32_t own_c0, own_c1;  // columns the region owns
  int32_t kslot[8];        // transfer-kernel slot of each patch of the slab (0 for an invalid one)
};
static_assert(sizeof(Job3) == 64, "Job3 layout");
enum : uint32_t {
  J3_UPPER_SHIFT = 0,  // 2 bits: Acc3 of the slab's upper H rows
  J3_LOWER_SHIFT = 2,  // 2 bits: Acc3 of the lower H rows
  J3_FLUSH_UPPER = 1u << 4,
  J3_FLUSH_LOWER = 1u << 5,
  J3_RING_HALF = 1u << 6,  // ring half (H rows) of the slab's upper rows; the lower rows go to the other one
  J3_VALID_SHIFT = 8,      // 8 bits: which patches of the slab exist
};
struct Region3 {
  int32_t job0, njobs;
};

struct Plan3 {
  std::vector<Job3> jobs;
  std::vector<Region3> regions;
  int strips = 0, segments = 0, ks = 0;
  long slabs = 0, patch_slots = 0;  // bookkeeping: patch_slots / (nli * nlj) = the recompute factor
};

// cell: nli x nlj transfer-kernel slots (all >= 0); par_j: parity of the view's first patch column in its parent's lattice (0 for a plan of its own)
inline bool plan3_build(int N, int ksmax, int waves, int nli, int nlj, const int32_t* cell, int par_j, int target_regions, Plan3& out) {
  const int h = N / 2, ppw = 128 / N;
  if (nli < 2 || nlj < 2 || ppw < 1 || ppw > 8) return false;
  const int ncols = nlj + 2;  // extended patch columns -1 .. nlj
  // ---- choose slabs per phase and strip (ks) and the number of row segments: least estimated time, then least work ----
  int best_ks = 1, best_seg = 1;
  double best_t = 1e300, best_w = 1e300;
  for (int ks = 1; ks <= ksmax; ++ks) {
    const int adv = 2 * ppw * ks - 1;
    const int nstrips = (ncols - 1 + adv - 1) / adv;
    for (int nseg = 1; nseg <= nli - 1; ++nseg) {
      const int advr = (nli - 1 + nseg - 1) / nseg;
      if ((nli - 1 + advr - 1) / advr != nseg) continue;  // the same cut under another name
      const int rows = advr + 1;
      const double jobs = 2.0 * ks * rows;
      const double rounds = (double)std::max(1L, ((long)nstrips * nseg + target_regions - 1) / target_regions);
      // a region runs its jobs `waves` at a time; the last lattice row of a region keeps only 2 ks waves busy; the ordered adds of its
      // 2 x rows layers come one after the other (about a seventh of a job each: what decides between cuts of a small frame)
      const double t = rounds * (std::max(jobs / waves, 1.0) + 1.0 + 0.15 * 2.0 * rows);
      const double w = (double)nstrips * nseg * jobs;
      if (t < best_t - 1e-9 || (t < best_t + 1e-9 && w < best_w)) best_t = t, best_w = w, best_ks = ks, best_seg = nseg;
    }
  }
  const int ks = best_ks, adv = 2 * ppw * ks - 1;
  const int advr = (nli - 1 + best_seg - 1) / best_seg;
  out.jobs.clear(), out.regions.clear();
  out.ks = ks, out.slabs = 0, out.patch_slots = 0;
  int nstrips = 0, nseg = 0;
  for (int e0 = 0; e0 < ncols - 1; e0 += adv, ++nstrips) {
    const int b0 = e0 - 1, b1 = std::min(b0 + adv, nlj);  // patch columns of the strip (extended numbering - 1)
    // the two phases' patches, each in runs of ppw
    std::vector<int> cols[2];
    for (int b = b0; b <= b1; ++b) cols[(b + par_j) & 1].push_back(b);
    struct Slab {
      int b_first;
      uint32_t valid;
    };
    std::vector<Slab> slabs[2];
    for (int ph = 0; ph < 2; ++ph)
      for (size_t i = 0; i < cols[ph].size(); i += ppw) {
        Slab s{cols[ph][i], 0};
        for (int m = 0; m < ppw; ++m) {
          const int b = s.b_first + 2 * m;
          if (i + m < cols[ph].size() && b >= 0 && b < nlj) s.valid |= 1u << m;
        }
        slabs[ph].push_back(s);
      }
    if (slabs[0].empty() || slabs[1].empty()) return false;
    const int ring_b0 = std::min(slabs[0][0].b_first, slabs[1][0].b_first);
    nseg = 0;
    for (int a0 = 0; a0 < nli - 1; a0 += advr, ++nseg) {
      const int a1 = std::min(a0 + advr, nli - 1);
      Region3 reg{(int32_t)out.jobs.size(), 0};
      const int own_r0 = a0 + (a0 > 0 ? 1 : 0), own_r1 = a1 + (a1 == nli - 1 ? 1 : 0);  // owned row bands, inclusive
      const int per_row = (int)(slabs[0].size() + slabs[1].size());
      for (int a = a0; a <= a1; ++a) {
        const bool up = a >= own_r0 && a <= own_r1, low = a + 1 >= own_r0 && a + 1 <= own_r1;
        for (int ph = 0; ph < 2; ++ph)
          for (size_t x = 0; x < slabs[ph].size(); ++x) {
            Job3 j{};
            j.row = a * h, j.col = slabs[ph][x].b_first * h;
            j.ring_col = (slabs[ph][x].b_first - ring_b0) * h;
            const uint32_t mu = !up ? 0u : ph == 0 ? (a == a0 ? 1u : 2u) : 2u;
            const uint32_t ml = !low ? 0u : ph == 0 ? 1u : 2u;
            j.flags = (mu << J3_UPPER_SHIFT) | (ml << J3_LOWER_SHIFT) | (slabs[ph][x].valid << J3_VALID_SHIFT);
            if (((a - a0) & 1) != 0) j.flags |= J3_RING_HALF;
            if (ph == 1 && up) j.flags |= J3_FLUSH_UPPER;
            if (ph == 1 && low && a == a1) j.flags |= J3_FLUSH_LOWER;
            j.own_c0 = (b0 + 1) * h, j.own_c1 = (b1 + 1) * h;
            for (int m = 0; m < 8; ++m) {
              const int b = slabs[ph][x].b_first + 2 * m;
              j.kslot[m] = (m < ppw && ((slabs[ph][x].valid >> m) & 1)) ? cell[(size_t)a * nlj + b] : 0;
              if (m < ppw) ++out.patch_slots;
            }
            // dependencies: the jobs of the other phase - this row's for phase B, the previous row's for phase A - whose columns overlap
            j.dep0 = j.dep1 = -1;
            const int other = ph ^ 1;
            const int base = (a - a0 - (ph == 0 ? 1 : 0)) * per_row + (other == 1 ? (int)slabs[0].size() : 0);
            if (ph == 1 || a > a0) {
              int nd = 0;
              for (size_t y = 0; y < slabs[other].size(); ++y) {
                const int d = (slabs[other][y].b_first - slabs[ph][x].b_first) * h;
                if (d > -128 && d < 128) {
                  if (nd == 0) j.dep0 = base + (int)y;
                  else if (nd == 1) j.dep1 = base + (int)y;
                  else return false;
                  ++nd;
                }
              }
            }
            out.jobs.push_back(j);
            ++out.slabs;
          }
      }
      reg.njobs = (int32_t)out.jobs.size() - reg.job0;
      out.regions.push_back(reg);
    }
  }
  out.strips = nstrips, out.segments = nseg;
  return true;
}

}  // namespace rpsf
