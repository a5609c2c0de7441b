// Two-phase product epilogues, shared by the single-workgroup MFMA kernels (svgp_fit.hip) and the cluster kernel.
#pragma once
#include <type_traits>

#include "common.h"

// ---- two-phase epilogues (round 4) --------------------------------------------------------------------------------
// An epilogue that READS memory (G_A: A, m, g_mu, g_v; G_LS: L_S and its Adam moments) used to be called block by block
// behind the k-loop: loads of block n + 1 were issued behind the stores of block n (the compiler cannot prove that GA
// and A do not alias) and vmcnt counts loads and stores in issue order, so every 16 x 16 block of a tile paid one full
// memory round trip -- the G_A and G_LS phases ran at 19 .. 42 % matrix-pipe busy where the plain-store G_KX^T product
// runs at 41 .. 72 % (tools/phase_table.py).  A two-phase epilogue splits into load(i0, j0) -> Pre (loads only) and
// store(i0, j0, block, pre): the product engines call load() for a group of up to four blocks back to back, then
// store() for the group -- one round trip per group.  Same arithmetic per element: bit-identical.
// GROUP = blocks whose loads are issued together: 4 with the whole register file (256 VGPRs: M = 320 -4.6 %, 384
// -3.2 %, 448 -1.9 % in time with the register look-ahead Cholesky), 1 in the 128-VGPR two-per-CU build, where
// the ~100 registers of four blocks' operands spill and the second workgroup hides the round trips anyway (group 4
// there: M = 144 .. 200 +9 %, 256 +5 % in time).
template <int GROUP, typename LoadF, typename StoreF>
struct TwoPhaseEpi {
  LoadF load;
  StoreF store;
  static constexpr bool two_phase = true;
  static constexpr int group = GROUP;
};
template <int GROUP = 4, typename LoadF, typename StoreF>
__device__ inline TwoPhaseEpi<GROUP, LoadF, StoreF> two_phase_epi(LoadF l, StoreF st) {
  return TwoPhaseEpi<GROUP, LoadF, StoreF>{l, st};
}
template <typename T, typename = void>
struct is_two_phase : std::false_type {};
template <typename T>
struct is_two_phase<T, std::void_t<decltype(T::two_phase)>> : std::true_type {};
// an epilogue shifted by (r0, c0): what product<> hands to the per-wave strips of a workgroup-tiled product
template <typename Epi>
__device__ inline auto shifted_epi(Epi epi, int r0, int c0) {
  if constexpr (is_two_phase<Epi>::value) {
    return two_phase_epi<Epi::group>([=](int i, int j) { return epi.load(r0 + i, c0 + j); },
                         [=](int i, int j, const gapro_mfma::d4& v, const auto& pre) { epi.store(r0 + i, c0 + j, v, pre); });
  } else {
    return [=](int i, int j, const gapro_mfma::d4& v) { epi(r0 + i, c0 + j, v); };
  }
}
// run the epilogue over NB blocks; blk(b, &i, &j) gives block b's position (or i < 0: not part of the output),
// acc(b) its accumulator
template <int NB, typename Epi, typename BlkPos, typename AccOf>
__device__ inline void run_epilogue(Epi& epi, BlkPos blk, AccOf acc) {
  if constexpr (is_two_phase<Epi>::value) {
    constexpr int G = NB < Epi::group ? NB : Epi::group;
#pragma unroll
    for (int g0 = 0; g0 < NB; g0 += G) {
      decltype(epi.load(0, 0)) pv[G];
#pragma unroll
      for (int b = 0; b < G; ++b) {
        int i, j;
        blk(g0 + b, &i, &j);
        if (i >= 0) pv[b] = epi.load(i, j);
      }
#pragma unroll
      for (int b = 0; b < G; ++b) {
        int i, j;
        blk(g0 + b, &i, &j);
        if (i >= 0) epi.store(i, j, acc(g0 + b), pv[b]);
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      int i, j;
      blk(b, &i, &j);
      if (i >= 0) epi(i, j, acc(b));
    }
  }
}

