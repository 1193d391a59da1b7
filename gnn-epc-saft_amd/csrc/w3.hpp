// "W3" weight images: the B operand of the split-bf16 GEMM (gemm_w3.hip), split ONCE per forward instead of once per
// workgroup and k step.  An f32 weight matrix W[n][k] (reference: the nn.Linear weights behind PyG PNAConv's pre_nns /
// post_nns / lin, /root/reference/gnnepcsaft/train/models.py:69-80,128) is stored as the three bf16 planes of its
// exact split (x6.hpp: w = hi + mid + lo) in exactly the byte order the GEMM's LDS stage wants, so a workgroup copies
// its B tile global -> LDS with direct-to-LDS loads (no VGPRs, no VALU, 1 KiB contiguous per wave instruction):
//
//   image[kt][plane][n][64 B]      kt = k / 32 (one LDS stage), plane = hi | mid | lo, n = output row
//
// A 64-byte row holds the stage's 32 k as four 16-byte chunks of 8 bf16; chunk c = 2 s + h feeds lane half h of the
// MFMA k16 step s and holds k = 4c .. 4c+3, 16+4c .. 16+4c+3 (relative to the stage) -- the eight values a staging
// lane of the A operand owns after its two float4 loads of one cache line, so A needs no cross-lane exchange to form
// 16-byte LDS writes; the order of k inside a dot product is free as long as A and B agree.  Chunk c sits at 16-byte
// position c ^ ((n >> 2) & 3): with that swizzle the ds_read_b128 fragment reads (lane = row, 64-byte row pitch) hit 64
// distinct banks in each of the instruction's four 16-lane groups (MI355X_MICROARCH.md, LDS table) without padding.
#pragma once
#include "common.hpp"
#include "x6.hpp"

namespace gs {

constexpr int kW3Kt = 32;         // k per stage
constexpr int kW3RowBytes = 64;   // bytes per row, plane and stage

// bytes of the image of an [n_pad, k] matrix (k a multiple of 32)
static inline size_t w3_bytes(int64_t n_pad, int64_t k) { return (size_t)(k / kW3Kt) * 3 * (size_t)n_pad * kW3RowBytes; }

// 16-byte position of chunk c in row n
__device__ __forceinline__ int w3_chunk_pos(int c, int n) { return c ^ ((n >> 2) & 3); }

// stores W[n][k .. k+3] (k a multiple of 4) into the image: 8 bytes per plane
__device__ __forceinline__ void w3_store4(char *__restrict__ img, int64_t n_pad, int64_t n, int k, f32x4 v) {
  uint32_t h[4], md[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) gs_split3(v[e], h[e], md[e], l[e]);
  const int kt = k >> 5, kk = k & 31;
  const int c = (kk & 15) >> 2, half = kk >> 4;
  char *p = img + ((int64_t)kt * 3 * n_pad + n) * kW3RowBytes + w3_chunk_pos(c, (int)(n & 31)) * 16 + half * 8;
  const int64_t plane = n_pad * kW3RowBytes;
  *reinterpret_cast<uint2 *>(p) = uint2{gs_pack_hi16(h[0], h[1]), gs_pack_hi16(h[2], h[3])};
  *reinterpret_cast<uint2 *>(p + plane) = uint2{gs_pack_hi16(md[0], md[1]), gs_pack_hi16(md[2], md[3])};
  *reinterpret_cast<uint2 *>(p + 2 * plane) = uint2{gs_pack_hi16(l[0], l[1]), gs_pack_hi16(l[2], l[3])};
}

// One weight matrix to convert (launch_w3_pack): rows [0, rows) of `src` (row pitch ld floats, k columns) become rows
// [row0, row0 + rows) of the image `dst` with n_pad rows per plane.
struct W3PackItem {
  const float *src;
  char *dst;
  int64_t ld;
  int32_t rows, k, n_pad, row0;
};
constexpr int kMaxW3PackBatch = 48;
int launch_w3_pack(int count, const W3PackItem *items, hipStream_t st);

// tile configurations of k_gemm_w3 (gemm_w3.hip)
enum W3Cfg { kW3_128x128 = 0, kW3_128x256 = 1, kW3_64x128 = 2, kW3_64x64 = 3, kW3_128x64 = 4, kW3_64x256 = 5,
             kW3_128x128d = 6, kW3_128x256d = 7 /* double-buffered, one workgroup per CU */, kNumW3Cfg = 8 };
int w3_cfg_bm(int cfg);
// measured choice for a plain GEMM of m rows (-1: the shape is outside the W3 kernels' envelope)
int w3_pick_cfg(int64_t m, int n_out, int k, bool stats);
bool gemm_w3_enabled();   // GNNSAFT_GEMM_W3 = 0 keeps the in-kernel weight split (k_gemm_f32<X6>) everywhere

// tile configurations of k_gemm_ar (gemm_ar.hip: A operand in registers, a wave owns 32 rows x all BN columns)
enum ArCfg { kAr_128x128 = 0, kAr_128x64 = 1, kAr_64x128 = 2, kAr_64x64 = 3, kAr_256x128 = 4, kNumArCfg = 5 };

}  // namespace gs
