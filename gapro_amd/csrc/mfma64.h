// FP64 matrix-core step of every fit kernel: D(16 x 16) += A(16 x 4) B(4 x 16), one double per lane and operand.
//
// gfx950 has two FP64 MFMA shapes, and they do not run at the same rate (measured on MI355X with dependent-free
// accumulator chains, all CUs, tools/mfma_peak.py; round 3):
//     v_mfma_f64_16x16x4_f64      48.3 TFLOP/s   (one wave per SIMD: 34)      -- what rounds 1 and 2 were built on
//     v_mfma_f64_4x4x4_4b_f64     75.2 TFLOP/s   (one wave per SIMD: 73.5)    -- 96 % of the datasheet's 78.6
// The products of the staged and cluster kernels already ran at 84 .. 96 % of the FIRST figure: the instruction was
// the bound.  The 4x4x4_4b form computes four independent 4 x 4 x 4 blocks, D_b = A_b B_b (+ C_b), b = 0 .. 3; its
// lane maps (tools/probes/mfma4x4_probe.hip, run on the hardware):
//     A: lane 16 k + 4 b + i      B: lane 16 k + 4 b + j      C / D: lane 16 i + 4 b + j
// i.e. with row = 4 b + i and column = 4 b + j the A and B registers of the 16x16x4 form (A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15]) ARE A and B registers of the four-block form, block b holding rows / columns
// 4 b .. 4 b + 3.  (CBSZ / ABID, which could broadcast one A block to all four, are ignored by the FP64 forms: probed.)
// So one 16 x 16 x 4 step is four block instructions, A's blocks rotated by r = 0 .. 3 blocks (a rotation by 4 r lanes
// inside every row of 16 lanes: DPP row_ror, two 32-bit moves per rotation):
//     acc[r] at lane l  +=  sum_k A[4 ((b + r) & 3) + i][k] B[k][4 b + j]       with b = (l >> 2) & 3, i = l >> 4, j = l & 3
//                        =  C[row = (l >> 4) + 4 ((b + r) & 3)][column = l & 15]
// -- the 16x16x4 form's accumulator layout (register r' <-> row (l >> 4) + 4 r', column l & 15) with the four registers
// of a lane rotated by the lane's b.  mma16() accumulates in that rotated layout, unrotate() turns a finished block
// into the standard layout once per block (16 selects), so every epilogue written for the 16x16x4 form is unchanged.
// Per element the contraction still runs over ascending k in groups of four.
#pragma once
#include <hip/hip_runtime.h>

namespace gapro_mfma {

typedef double d4 __attribute__((ext_vector_type(4)));

// value of `v` in the lane whose position inside its row of 16 lanes is (own position + 4 R) mod 16
template <int R>
__device__ inline double rot_blocks(double v) {
  if (R == 0) return v;
  constexpr int ctrl = 0x120 + (16 - 4 * R);  // DPP row_ror:n -- lane p receives lane (p - n) mod 16
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

struct AFrag {  // an A fragment and its three block rotations
  double r[4];
};
__device__ inline AFrag make_afrag(double a) {
  AFrag f;
  f.r[0] = a;
  f.r[1] = rot_blocks<1>(a);
  f.r[2] = rot_blocks<2>(a);
  f.r[3] = rot_blocks<3>(a);
  return f;
}

// acc (rotated layout) += A B for one 16 x 16 x 4 step
__device__ inline void mma16(const AFrag& a, double b, d4& acc) {
#ifdef GAPRO_MFMA_16X16  // A/B experiments: the 16x16x4 form (standard layout; unrotate() is then the identity)
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.r[0], b, acc, 0, 0, 0);
#else
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[0], b, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[1], b, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[2], b, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[3], b, acc[3], 0, 0, 0);
#endif
}
__device__ inline void mma16(double a, double b, d4& acc) { mma16(make_afrag(a), b, acc); }

// rotated layout -> the 16x16x4 form's layout: register r' of a lane = rotated register (r' - b) & 3, b = (lane >> 2) & 3
__device__ inline d4 unrotate(const d4& acc) {
#ifdef GAPRO_MFMA_16X16
  return acc;
#else
  const int b = (threadIdx.x >> 2) & 3;
  d4 t, o;
  // barrel shifter: by one register if b & 1, by two if b & 2
  t[0] = (b & 1) ? acc[3] : acc[0];
  t[1] = (b & 1) ? acc[0] : acc[1];
  t[2] = (b & 1) ? acc[1] : acc[2];
  t[3] = (b & 1) ? acc[2] : acc[3];
  o[0] = (b & 2) ? t[2] : t[0];
  o[1] = (b & 2) ? t[3] : t[1];
  o[2] = (b & 2) ? t[0] : t[2];
  o[3] = (b & 2) ? t[1] : t[3];
  return o;
#endif
}
// the inverse: a block in the standard layout (an accumulator to continue from) -> rotated layout
__device__ inline d4 rotate_in(const d4& c) {
#ifdef GAPRO_MFMA_16X16
  return c;
#else
  const int b = (threadIdx.x >> 2) & 3;
  d4 t, o;  // rotated register r = standard register (r + b) & 3
  t[0] = (b & 1) ? c[1] : c[0];
  t[1] = (b & 1) ? c[2] : c[1];
  t[2] = (b & 1) ? c[3] : c[2];
  t[3] = (b & 1) ? c[0] : c[3];
  o[0] = (b & 2) ? t[2] : t[0];
  o[1] = (b & 2) ? t[3] : t[1];
  o[2] = (b & 2) ? t[0] : t[2];
  o[3] = (b & 2) ? t[1] : t[3];
  return o;
#endif
}

}  // namespace gapro_mfma
