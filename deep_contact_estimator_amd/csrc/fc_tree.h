// fc_tree.h -- the ONE summation tree every fp32 Linear kernel of libdce.so evaluates (fc.0 and fc.3 of the reference's
// src/contact_cnn.py:48-54; fc.6 has its own, fc6_chain.h), so that a window's activations are the same bits whichever
// kernel -- GEMV, MFMA chain, tile GEMM, phased GEMM -- a batch size selects:
//
//     K is cut into FC_RANGES = 4 ranges at multiples of 3 x 128 floats:  range r = [ cut(r) 128, cut(r+1) 128 ),  U = K / 128,
//         cut(0) = 0,  cut(4) = U,  cut(r) = 3 ((U r / 4 + 1) / 3)  (integer divisions: the quarter points rounded to a multiple
//         of 3 -- fc.0: 0, 9, 18, 27, 37;  fc.3: 0, 3, 9, 12, 16 -- so that a cut falls on a whole round of the phased GEMMs'
//         K loop, which is unrolled by its three LDS buffers, whether its K-tiles are 32 or 64 floats)
//     p_r = the fmaf chain over range r, started at 0, every 8 consecutive k walked as 0,4,1,5,2,6,3,7
//           (an fp32 MFMA is an ordered fmaf chain; that is the order the 32x32x2 / 16x16x4 kernels feed the pipe)
//     y   = act( ((((0 + p_0) + p_1) + p_2) + p_3) + bias )
//
// Round 2 walked K as one chain of 4736 links: every kernel that is bound by the LENGTH of that chain (GEMV at batch
// size 1: 9.4 cycles per link = 18.5 us for fc.0) paid for the property.  Four ranges cut the chain to 1280 links where
// the ranges can run side by side (the GEMV's four waves), and cost the throughput kernels one accumulator-sized add
// at three K-tiles out of 148.  (The tests hold a bit-level CPU model of this tree, built on fmaf.)
#pragma once

namespace dce {

constexpr int FC_RANGES = 4;

// first 128-float unit of range r (r = 0 .. 4; cut(4) = U)
__host__ __device__ constexpr int fc_tree_unit(int U, int r) { return r <= 0 ? 0 : r >= FC_RANGES ? U : 3 * ((U * r / 4 + 1) / 3); }

// first K-tile (tiles of `unit` floats, unit | 128) of ranges 1, 2, 3; K % 128 != 0 (not a shape of this model): no cut
struct FcTree { int b1, b2, b3; };
__host__ __device__ inline FcTree fc_tree(int K, int unit)
{
    if (K % 128) return FcTree{-1, -1, -1};
#ifdef FC_TREE_OFF           // timing probe only (WRONG results: one chain in the MFMA GEMMs, four ranges elsewhere)
    return FcTree{-1, -1, -1};
#endif
    const int U = K / 128, m = 128 / unit;
    return FcTree{fc_tree_unit(U, 1) * m, fc_tree_unit(U, 2) * m, fc_tree_unit(U, 3) * m};
}
__host__ __device__ inline bool fc_tree_cut(const FcTree& t, int tile) { return tile == t.b1 || tile == t.b2 || tile == t.b3; }

}  // namespace dce
