// Device helpers shared by the two attention kernels (attn.hip, bert_attn.hip): DPP row reductions and the
// stateless dropout hash.
#pragma once
#include "common.h"

namespace hopmi {

// Reductions over the 16 lanes of a DPP row (= the 16 key columns j a lane quad-group holds), on the VALU
// with DPP modifiers instead of ds_bpermute round trips: xor 1, xor 2 (quad_perm), then row_half_mirror
// (lane i <-> 7 - i of each half: the two quads of a half), then row_mirror (i <-> 15 - i: the two halves).
// Every lane of the row ends up with the row's result.
template <int CTRL>
__device__ __forceinline__ float dpp_(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_<0x141>(v));     // row_half_mirror
  v = fmaxf(v, dpp_<0x140>(v));     // row_mirror
  return v;
}
// Maximum over the whole wave of a NON-NEGATIVE value, as a wave-uniform number: the 16-lane DPP rows first, then row_bcast15 /
// row_bcast31 (gfx9 DPP: lane 15 of every row into the next row, lane 31 into the upper half) leave the total in lane 63, read back
// with v_readlane -- 6 VALU operations + one readlane instead of six ds_bpermute round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep_(float v) {      // lanes outside ROW_MASK keep v
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = row16_max(v);
  v = fmaxf(v, dpp_keep_<0x142, 0xA>(v));      // row_bcast15 -> rows 1, 3
  v = fmaxf(v, dpp_keep_<0x143, 0xC>(v));      // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_<0xB1>(v);
  v += dpp_<0x4E>(v);
  v += dpp_<0x141>(v);
  v += dpp_<0x140>(v);
  return v;
}

__device__ __forceinline__ unsigned attn_hash(unsigned seed, unsigned row, unsigned head, unsigned key) {
  unsigned x = seed ^ (row * 0x9E3779B1u) ^ (key * 0x85EBCA77u) ^ (head * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;      // murmur3 fmix32
  return x;
}

// One hash per pair of adjacent keys (reprogramming attention): keep bits of keys 2 p and 2 p + 1 are the low and the high
// 16 bits of the hash compared with p_drop * 2^16.  `rowhead` = attn_rowhead(seed, row, head) is hoisted per row.
__device__ __forceinline__ unsigned attn_rowhead(unsigned seed, unsigned row, unsigned head) {
  return seed ^ (row * 0x9E3779B1u) ^ (head * 0xC2B2AE3Du);
}
__device__ __forceinline__ unsigned attn_hash_pair(unsigned rowhead, unsigned pair) {
  unsigned x = rowhead ^ (pair * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;      // murmur3 fmix32
  return x;
}

}  // namespace hopmi
