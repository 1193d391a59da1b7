// Split-bf16 GEMM with PRE-SPLIT weights ("W3" images, w3.hpp):  out = epilogue( A_virtual x W^T + b ).
//
// Same arithmetic as k_gemm_f32<X6> (gemm.hip): every f32 operand is the exact sum of three bf16 numbers, six
// v_mfma_f32_32x32x16_bf16 per k16 step with f32 accumulation -- the accuracy of an f32 fma chain (the 1e-5 parity bar
// of PNAPCSAFT.forward, /root/reference/gnnepcsaft/train/models.py:105-135, sees no difference) at 6/16 of the f32
// matrix cores' time.  What changes is the operand path, which bounded the X6 kernel (profiles/r03_gemm_mfma_pmc_*:
// 6-17 VALU instructions per MFMA, VALU issue time ADDING to the matrix cores' time):
//   * the weights are split ONCE per forward into bf16 planes laid out as the LDS stage wants them (w3.hpp); a
//     workgroup copies its B tile with direct-to-LDS loads (global_load_lds_dwordx4: 1 KiB contiguous per wave
//     instruction, no VGPRs, no VALU) instead of splitting the same weight tile in every one of ~1300 row tiles;
//   * a stage is 32 k = one 128-byte line per A row: a staging lane loads the two float4 at k = 4q and 16 + 4q of its
//     row (the two halves of ONE cache line, back to back) and owns a whole 16-byte chunk of each bf16 plane, so
//     the plane writes are ds_write_b128 (no 8-byte writes, no bank conflicts) and the split is 5.5 VALU per element
//     (and / sub / and / sub + v_perm_b32 packing two bf16 at a time);
//   * LDS rows are unpadded 64 B per plane with an XOR swizzle (conflict-free ds_read_b128 fragments), 192 B per
//     row and stage: 128 x 256 and 128 x 128 tiles double-buffered in 144 / 96 KB.
// Per stage and 128 x 128 tile on 8 waves that is ~45 VALU per wave against 24 MFMAs (was ~100 per 12).
//
// Pipeline per stage t: the loads of stage t + 2 (A rows, B pieces) are issued, the MFMAs of stage t run out of LDS
// buffer t & 1, then stage t + 1 (loaded one stage earlier) is split / copied into buffer (t + 1) & 1; one barrier.
#include <cstdlib>

#include "common.hpp"
#include "gemm_epi.hpp"
#include "gemm_prov.hpp"
#include "w3.hpp"

namespace gs {

// two bf16 (the upper halves of x0, x1) in one dword, x0 low: v_perm_b32
__device__ __forceinline__ uint32_t w3_pack2(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// NBUF: LDS stages.  2 = double buffer, one barrier per stage, B through registers, every load two stages ahead.
//       1 = ONE LDS stage and two barriers per stage, B by direct-to-LDS loads: a third of the LDS, so that 2-3 workgroups
//       share a CU and fill each other's load / split / barrier phases with their MFMAs (and one's epilogue stores run
//       under the others' main loops).  Measured (tools/w3_variants.py): with one big workgroup per
//       CU the two waves of a SIMD run in lockstep between barriers -- fragment reads, staging and MFMAs take turns
//       instead of overlapping (the kernel took the SUM of its parts: 214 us of MFMA + fragment reads and 216 us of
//       staging gave 332 us on the C3 update shape).
// MINWG: waves per SIMD the register allocation must allow (hipcc reads the second launch bound as waves per execution
//        unit): workgroups per CU x waves per workgroup / 4.
// VAR: timing experiments only (tools/w3_variants.py; results are garbage): bit 0 no A loads, 1 no A split / LDS
// writes, 2 no B copies, 3 no MFMAs, 4 no fragment reads, 5 no barriers, 6 loads NOT pinned at the top of the stage
template <int BM, int BN, int WAVES_M, int WAVES_N, int NBUF, int MINWG, class AProv, bool STATS, bool AFFINE, bool RESID,
          int VAR = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINWG) void k_gemm_w3(AProv ap, GemmBatch batch, int n_pad,
                                                                            int64_t ldo, int n_out, int k, EpiArgs epi) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M;
  constexpr int WTN = BN / WAVES_N;
  static_assert(!STATS || WTM == kBnRowsPerGroup, "BatchNorm partials assume 64 rows per wave");
  constexpr int TM = WTM / 32;
  constexpr int TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA");
  constexpr int RPP = NT / 4;                    // A rows covered by one pass of the workgroup (4 lanes per row)
  constexpr int A_R = BM / RPP;                  // rows per thread
  static_assert(A_R >= 1 && BM % RPP == 0 && RPP % 16 == 0, "the A tile is a whole number of passes");
  constexpr int PLANE = (BM + BN) * kW3RowBytes; // bytes per bf16 plane and stage: A rows, then B rows
  constexpr int STAGE = 3 * PLANE;
  constexpr int B_PIECES = 3 * BN / 16;          // 1-KiB pieces per stage (16 rows of one plane each)
  static_assert(B_PIECES % NW == 0, "every wave copies the same number of pieces");
  constexpr int B_PW = B_PIECES / NW;
  constexpr bool kDma = NBUF == 1;               // B by direct-to-LDS loads (no registers) / through registers
  static_assert(NBUF == 1 || NBUF == 2, "one or two LDS stages");
  extern __shared__ __attribute__((aligned(256))) char lds[];   // NBUF * STAGE

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const TileInfo ti = ap.tile(blockIdx.x, BM);
  if (ti.count <= 0) return;  // block-uniform, before any barrier
  const int n0 = blockIdx.y * BN;
  const GemmBatchEntry ent = batch.e[blockIdx.z];
  const int nk = k / kW3Kt;

  // ---- A staging map: lane q = tid & 3 of row r0 + RPP j owns chunk q of the row's 32 k
  const int q = tid & 3;
  const int r0 = tid >> 2;
  typename AProv::Row arow[A_R];
#pragma unroll
  for (int j = 0; j < A_R; ++j) {
    const int lr = r0 + RPP * j;
    arow[j] = ap.row(ti.row0 + (lr < ti.count ? lr : ti.count - 1), ent.a_off);   // clamped rows are never stored
  }
  const int a_lds = r0 * kW3RowBytes + w3_chunk_pos(q, r0) * 16;   // (RPP is a multiple of 16: same swizzle for every j)

  // ---- B pieces of this wave: piece i = (plane, block of 16 rows) = 1 KiB contiguous in the image AND in the LDS stage;
  //      lane -> row (lane >> 2), 16 bytes (lane & 3)
  const char *bsrc[B_PW];
  int bdst[B_PW];
#pragma unroll
  for (int jj = 0; jj < B_PW; ++jj) {
    const int i = wave + NW * jj;
    const int p = i / (BN / 16), rb = i % (BN / 16);
    int n = n0 + rb * 16 + (lane >> 2);
    n = n < n_pad ? n : n_pad - 1;                                   // clamped columns are never stored
    bsrc[jj] = ent.w3 + ti.w_off * 6 + ((int64_t)p * n_pad + n) * kW3RowBytes + (lane & 3) * 16;
    bdst[jj] = p * PLANE + (BM + rb * 16) * kW3RowBytes + (kDma ? 0 : lane * 16);   // (the DMA adds lane * 16 itself)
  }
  const int64_t bstep = (int64_t)3 * n_pad * kW3RowBytes;             // bytes between two stages of the image

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typename AProv::Raw ra[2][A_R][2];
  f32x4 rbw[kDma ? 1 : 2][kDma ? 1 : B_PW];
  // fetch / stash / copy take ANY stage index: past the last stage they repeat the last one into a buffer nobody reads
  // (no branch inside the k loop: see gemm.hip on what a conditional load does to hipcc's s_waitcnt placement)
  auto fetch_a = [&](int kt, typename AProv::Raw(&r)[A_R][2]) {
    if constexpr ((VAR & 1) != 0) return;
    const int kc = kt < nk ? kt : nk - 1;
#pragma unroll
    for (int j = 0; j < A_R; ++j) {
      r[j][0] = ap.load(arow[j], kc * kW3Kt, 4 * q);
      r[j][1] = ap.load(arow[j], kc * kW3Kt, 16 + 4 * q);
    }
  };
  auto fetch_b = [&](int kt, f32x4(&rb)[kDma ? 1 : B_PW]) {   // registers (NBUF == 2)
    if constexpr ((VAR & 4) != 0 || kDma) return;
    const int kc = kt < nk ? kt : nk - 1;
#pragma unroll
    for (int jj = 0; jj < (kDma ? 1 : B_PW); ++jj) rb[jj] = *reinterpret_cast<const f32x4 *>(bsrc[jj] + kc * bstep);
  };
  auto dma_b = [&](int kt, int buf) {                          // direct to LDS (NBUF == 1)
    if constexpr ((VAR & 4) != 0 || !kDma) return;
    const int64_t off = (int64_t)(kt < nk ? kt : nk - 1) * bstep;
#pragma unroll
    for (int jj = 0; jj < B_PW; ++jj)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc[jj] + off),
                                       (__attribute__((address_space(3))) void *)(lds + buf * STAGE + bdst[jj]), 16, 0, 0);
  };
  auto pin = [&]() {
    if constexpr ((VAR & 64) == 0) __builtin_amdgcn_sched_barrier(0);   // the loads stay at the top of the stage
  };
  auto stash_b = [&](int buf, const f32x4(&rb)[kDma ? 1 : B_PW]) {
    if constexpr ((VAR & 4) != 0 || kDma) return;
#pragma unroll
    for (int jj = 0; jj < (kDma ? 1 : B_PW); ++jj) *reinterpret_cast<f32x4 *>(lds + buf * STAGE + bdst[jj]) = rb[jj];
  };
  auto stash_a = [&](int kt, int buf, const typename AProv::Raw(&r)[A_R][2]) {
    if constexpr ((VAR & 2) != 0) return;
    const int k0 = (kt < nk ? kt : nk - 1) * kW3Kt;
#pragma unroll
    for (int j = 0; j < A_R; ++j) {
      const f32x4 v0 = ap.finish_full(r[j][0], arow[j], k0, 4 * q);
      const f32x4 v1 = ap.finish_full(r[j][1], arow[j], k0, 16 + 4 * q);
      uint32_t e[8], m1[8], m2[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        e[t] = __float_as_uint(v0[t]);
        e[4 + t] = __float_as_uint(v1[t]);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {   // a = hi + mid + lo exactly (x6.hpp); the planes keep the upper 16 bits of each
        const float a = __uint_as_float(e[t]);
        const float r1 = a - __uint_as_float(e[t] & 0xffff0000u);
        m1[t] = __float_as_uint(r1);
        m2[t] = __float_as_uint(r1 - __uint_as_float(m1[t] & 0xffff0000u));
      }
      char *p = lds + buf * STAGE + a_lds + j * (RPP * kW3RowBytes);
      *reinterpret_cast<uint4 *>(p) = uint4{w3_pack2(e[0], e[1]), w3_pack2(e[2], e[3]), w3_pack2(e[4], e[5]), w3_pack2(e[6], e[7])};
      *reinterpret_cast<uint4 *>(p + PLANE) =
          uint4{w3_pack2(m1[0], m1[1]), w3_pack2(m1[2], m1[3]), w3_pack2(m1[4], m1[5]), w3_pack2(m1[6], m1[7])};
      *reinterpret_cast<uint4 *>(p + 2 * PLANE) =
          uint4{w3_pack2(m2[0], m2[1]), w3_pack2(m2[2], m2[3]), w3_pack2(m2[4], m2[5]), w3_pack2(m2[6], m2[7])};
    }
  };

  // fragment addresses: lane (row fr, half hh) reads chunk 2 s + hh of its row at step s
  const int fr = lane & 31, hh = lane >> 5;
  const int f0 = fr * kW3RowBytes + w3_chunk_pos(hh, fr) * 16;   // step 0; step 1 = f0 ^ 32
  const int fa = (wm * WTM) * kW3RowBytes, fb = (BM + wn * WTN) * kW3RowBytes;
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char *base = lds + buf * STAGE + (s == 0 ? f0 : (f0 ^ 32));
      bf16x8 af[TM][3], bf[TN][3];
      if constexpr ((VAR & 16) != 0) {   // operands from registers (values irrelevant)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(bf16x8, f32x4{acc[i][0][p], 1.f, 2.f, (float)s});
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[j][p] = __builtin_bit_cast(bf16x8, f32x4{acc[0][j][p + 3], 3.f, 1.f, (float)s});
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            af[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + fa + p * PLANE + i * 32 * kW3RowBytes));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            bf[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + fb + p * PLANE + j * 32 * kW3RowBytes));
      }
      if constexpr ((VAR & 8) != 0) {   // no MFMAs: keep the fragments alive
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) acc[i][0][p] += __builtin_bit_cast(f32x4, af[i][p])[s];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) acc[0][j][p + 3] += __builtin_bit_cast(f32x4, bf[j][p])[s];
        continue;
      }
      // six of the nine cross products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
      constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa[t]], bf[j][pb[t]], acc[i][j], 0, 0, 0);
    }
  };

  auto sync = [&]() {
    if constexpr ((VAR & 32) == 0) __syncthreads();
  };
  if constexpr ((VAR & 1) != 0) {   // (experiment: the A registers hold something)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < A_R; ++j) ra[u][j][0] = ra[u][j][1] = ap.load(arow[j], 0, 4 * q);
  }
  if constexpr ((VAR & 4) != 0 && !kDma) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int jj = 0; jj < B_PW; ++jj) rbw[u][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (NBUF == 2) {
    // loads run TWO stages ahead of the MFMAs (issued at the top of stage t for stage t + 2, split into LDS at the end
    // of stage t + 1): two register sets, exact vmcnt counts
    fetch_b(0, rbw[0]);
    fetch_a(0, ra[0]);
    fetch_b(1, rbw[kDma ? 0 : 1]);
    fetch_a(1, ra[1]);
    stash_b(0, rbw[0]);
    stash_a(0, 0, ra[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      fetch_b(kt + 2, rbw[0]);
      fetch_a(kt + 2, ra[0]);
      pin();
      compute(0);
      stash_b(1, rbw[kDma ? 0 : 1]);
      stash_a(kt + 1, 1, ra[1]);
      sync();
      fetch_b(kt + 3, rbw[kDma ? 0 : 1]);
      fetch_a(kt + 3, ra[1]);
      pin();
      if (kt + 1 < nk) compute(1);
      stash_b(0, rbw[0]);
      stash_a(kt + 2, 0, ra[0]);
      sync();
    }
  } else {
    // one LDS stage: MFMAs of stage t | barrier | B(t+1) by DMA + split of A(t+1) into the same buffer | barrier.  The
    // A registers of stage t + 2 are requested at the top of stage t (two sets); the DMA's L2 round trip and both
    // barriers are covered by the MFMAs of the OTHER workgroups on the CU.
    dma_b(0, 0);
    fetch_a(0, ra[0]);
    fetch_a(1, ra[1]);
    stash_a(0, 0, ra[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      fetch_a(kt + 2, ra[0]);
      pin();
      compute(0);
      sync();
      dma_b(kt + 1, 0);
      stash_a(kt + 1, 0, ra[1]);
      sync();
      fetch_a(kt + 3, ra[1]);
      pin();
      if (kt + 1 < nk) compute(0);
      sync();
      dma_b(kt + 2, 0);
      stash_a(kt + 2, 0, ra[0]);
      sync();
    }
  }

  gemm_epilogue<TM, TN, WTM, WTN, BM, BN, WAVES_M, STATS, AFFINE, RESID>(acc, ap, ti, ent, epi, n0, n_out, ldo, wm, wn, lane);
}

// --------------------------------------------------------------------------
// weight images
// --------------------------------------------------------------------------
struct W3PackArgs {
  W3PackItem item[kMaxW3PackBatch];
};

// one thread per float4 of a source matrix; blockIdx.y = item
__global__ __launch_bounds__(256) void k_w3_pack(W3PackArgs args) {
  const W3PackItem it = args.item[blockIdx.y];
  const int per_row = it.k >> 2;
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= (int64_t)it.rows * per_row) return;
  const int row = (int)(slot / per_row), c4 = (int)(slot - (int64_t)row * per_row);
  w3_store4(it.dst, it.n_pad, it.row0 + row, 4 * c4, gs_ld4(it.src + (int64_t)row * it.ld + 4 * c4));
}

int launch_w3_pack(int count, const W3PackItem *items, hipStream_t st) {
  GS_REQUIRE(count >= 0 && items != nullptr, GNNSAFT_ERR_SHAPE);
  for (int i0 = 0; i0 < count; i0 += kMaxW3PackBatch) {
    const int nb = count - i0 < kMaxW3PackBatch ? count - i0 : kMaxW3PackBatch;
    W3PackArgs a;
    int64_t most = 0;
    for (int i = 0; i < kMaxW3PackBatch; ++i) {
      a.item[i] = items[i0 + (i < nb ? i : 0)];
      const W3PackItem &it = a.item[i];
      GS_REQUIRE(it.src != nullptr && it.dst != nullptr, GNNSAFT_ERR_NULL);
      GS_REQUIRE(it.rows >= 1 && it.k >= kW3Kt && (it.k % kW3Kt) == 0 && (it.ld % 4) == 0 && it.row0 >= 0 &&
                     it.row0 + it.rows <= it.n_pad && (reinterpret_cast<uintptr_t>(it.src) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(it.dst) & 15) == 0,
                 GNNSAFT_ERR_SHAPE);
      const int64_t slots = (int64_t)it.rows * (it.k / 4);
      most = slots > most ? slots : most;
    }
    hipLaunchKernelGGL(k_w3_pack, dim3((unsigned)gs_ceil_div(most, 256), (unsigned)nb), dim3(256), 0, st, a);
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

// --------------------------------------------------------------------------
// host-side dispatch
// --------------------------------------------------------------------------
static const int kW3BM[kNumW3Cfg] = {128, 128, 64, 64, 128, 64, 128, 128};
int w3_cfg_bm(int cfg) { return cfg >= 0 && cfg < kNumW3Cfg ? kW3BM[cfg] : 0; }

bool gemm_w3_enabled() {
  static const bool on = [] {
    const char *e = getenv("GNNSAFT_GEMM_W3");   // 0: the in-kernel weight split everywhere
    return (e == nullptr || e[0] != '0') && gemm_x6_enabled();
  }();
  return on;
}

int w3_pick_cfg(int64_t m, int n_out, int k, bool stats) {
  if (!gemm_w3_enabled() || k < kW3Kt || (k % kW3Kt) != 0 || n_out < 64 || (n_out % 16) != 0 || m < 1) return -1;
  static const int forced = [] {
    const char *e = getenv("GNNSAFT_W3_CFG");
    return e != nullptr ? atoi(e) : -1;
  }();
  if (forced >= 0 && forced < kNumW3Cfg) return forced;
  (void)stats;   // every configuration has 64-row wave tiles
  // measured (tools/w3_tune.py, profiles/r04_w3_tile_sweep.txt): from 64 k rows up the images win 3-5 % on every shape
  // of the forward; below, the in-kernel split with its smaller tiles is as fast or faster (latency-bound launches)
  if (m >= 65536) return n_out >= 256 ? kW3_128x256 : (n_out >= 128 ? kW3_128x128 : kW3_128x64);
  // (A/B in the replayed C2 step, round 4: the images for the source terms below 64 k rows -- 128 x 64 tiles, 14.6 us
  //  against 17.7 stand-alone -- moved the step by 0.3748 -> 0.3722 ms, inside the run-to-run spread: not taken)
  return -1;
}

template <int BM, int BN, int WM, int WN, int NBUF, int MINWG, class AProv, bool STATS, bool AFFINE, bool RESID, int VAR = 0>
static int launch_w3_one(const AProv &ap, int nbatch, const GemmBatch &b, int n_pad, int64_t ldo, int64_t m, int n_out,
                         int k, const EpiArgs &ea, int64_t grid_x, hipStream_t stream) {
  constexpr size_t kLds = (size_t)NBUF * 3 * (BM + BN) * kW3RowBytes;
  static_assert(kLds * (MINWG * 4 / (WM * WN)) <= 160 * 1024, "the workgroups MINWG stands for share the CU's LDS");
  auto kern = k_gemm_w3<BM, BN, WM, WN, NBUF, MINWG, AProv, STATS, AFFINE, RESID, VAR>;
  static std::atomic<unsigned long long> raised{0ull};
  if (kLds > 64 * 1024) GS_HIP(gs_raise_dynamic_lds(reinterpret_cast<const void *>(kern), kLds, raised));
  const dim3 grid((unsigned)(grid_x > 0 ? grid_x : gs_ceil_div(m, BM)), (unsigned)gs_ceil_div(n_out, BN), (unsigned)nbatch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), kLds, stream, ap, b, n_pad, ldo, n_out, k, ea);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

template <class AProv, bool STATS, bool AFFINE, bool RESID>
static int launch_w3_cfg(int cfg, const AProv &ap, int nbatch, const GemmBatch &b, int n_pad, int64_t ldo, int64_t m,
                         int n_out, int k, const EpiArgs &ea, int64_t grid_x, hipStream_t stream) {
  switch (cfg) {
#define GS_W3(BM, BN, WM, WN, NBUF, MINWG) \
  return launch_w3_one<BM, BN, WM, WN, NBUF, MINWG, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream)
    case kW3_128x128: GS_W3(128, 128, 2, 2, 1, 3);    // 4 waves of 64 x 64, 48 KB: three workgroups per CU
    case kW3_128x256: GS_W3(128, 256, 2, 2, 1, 2);    // 4 waves of 64 x 128, 72 KB: two per CU
    case kW3_64x128: GS_W3(64, 128, 1, 4, 1, 4);      // 4 waves of 64 x 32, 36 KB: four per CU
    case kW3_64x64: GS_W3(64, 64, 1, 2, 1, 3);        // 2 waves of 64 x 32, 24 KB: six per CU
    case kW3_128x64: GS_W3(128, 64, 2, 2, 1, 4);      // 4 waves of 64 x 32, 36 KB: four per CU
    case kW3_64x256: GS_W3(64, 256, 1, 4, 1, 2);      // 4 waves of 64 x 64, 60 KB: two per CU
    case kW3_128x128d: GS_W3(128, 128, 2, 4, 2, 1);   // 8 waves of 64 x 32, double-buffered 96 KB: one per CU
    case kW3_128x256d: GS_W3(128, 256, 2, 4, 2, 1);   // 8 waves of 64 x 64, double-buffered 144 KB: one per CU
#undef GS_W3
    default:
      return GNNSAFT_ERR_UNSUPPORTED;
  }
}

// out = epilogue(a W^T + b) with every entry's weights given as a W3 image (entries[i].w3; n_pad rows per plane)
int launch_linear_w3(const float *a, int64_t lda, int nbatch, const GemmBatchEntry *entries, int n_pad, int64_t ldo,
                     int64_t m, int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg) {
  GS_REQUIRE(a != nullptr && entries != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(nbatch >= 1 && nbatch <= kMaxGemmBatch && cfg >= 0 && cfg < kNumW3Cfg, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(m >= 0 && n_out >= 1 && k >= kW3Kt && (k % kW3Kt) == 0 && (lda % 4) == 0 && n_pad >= n_out &&
                 (reinterpret_cast<uintptr_t>(a) & 15) == 0,
             GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(ldo >= 0 && epi.ldr >= 0 && (m + 1) * (ldo > epi.ldr ? ldo : epi.ldr) < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);
  if (m == 0) return GNNSAFT_OK;
  GemmBatch b;
  for (int i = 0; i < kMaxGemmBatch; ++i) b.e[i] = entries[i < nbatch ? i : 0];
  for (int i = 0; i < nbatch; ++i) {
    GS_REQUIRE(entries[i].w3 != nullptr && entries[i].out != nullptr, GNNSAFT_ERR_NULL);
    GS_REQUIRE((reinterpret_cast<uintptr_t>(entries[i].w3) & 15) == 0, GNNSAFT_ERR_SHAPE);
  }
  EpiArgs ea{epi.scale, epi.shift, epi.relu_out, epi.residual, epi.ldr, epi.stats, epi.residual_is_mask,
             epi.bn_mean, epi.bn_var, epi.bn_eps};
  GS_REQUIRE((epi.bn_var == nullptr) == (epi.bn_mean == nullptr) && (epi.bn_var == nullptr || epi.scale != nullptr),
             GNNSAFT_ERR_NULL);
  GS_REQUIRE((epi.scale == nullptr) == (epi.shift == nullptr), GNNSAFT_ERR_NULL);
  GS_REQUIRE(epi.stats == nullptr || nbatch == 1, GNNSAFT_ERR_SHAPE);
  PlainA ap{a, lda, m, k};
  const bool st = epi.stats != nullptr, af = epi.scale != nullptr, rs = epi.residual != nullptr;
  if (st) {
    GS_REQUIRE(!af && !rs, GNNSAFT_ERR_UNSUPPORTED);
    return launch_w3_cfg<PlainA, true, false, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  }
  if (af && rs) return launch_w3_cfg<PlainA, false, true, true>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  if (af) return launch_w3_cfg<PlainA, false, true, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  if (rs) return launch_w3_cfg<PlainA, false, false, true>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  return launch_w3_cfg<PlainA, false, false, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
}

// the W3 configuration whose BM equals the degree tile table's rows (pna_fold_tile_rows), or -1
int w3_cfg_for_update(int hidden, int64_t n) {
  if (!gemm_w3_enabled() || (hidden % 64) != 0 || hidden < 128 || n < 65536) return -1;
  const int rows = pna_fold_tile_rows(hidden), n_out = hidden / 2;
  if (rows == 128) return n_out >= 128 ? kW3_128x128 : kW3_128x64;
  if (rows == 64) return n_out >= 128 ? kW3_64x128 : kW3_64x64;
  return -1;
}

// degree-folded update (launch_pna_update_folded) on W3 images of the folded weights:
// w_eff3 = images of [D][2][F/2, 5F], one image of (F/2) * 5F * 6 bytes per (degree, tower)
int launch_pna_update_folded_w3(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const char *w_eff3,
                                const float *b_post0, const float *b_post1, float *u, hipStream_t stream) {
  GS_REQUIRE(x && agg && perm && tiles && num_tiles && w_eff3 && u, GNNSAFT_ERR_NULL);
  const int cfg = w3_cfg_for_update(hidden, (int64_t)1 << 20);   // (the caller decided; the size rule is the forward's)
  GS_REQUIRE(cfg >= 0, GNNSAFT_ERR_UNSUPPORTED);
  if (n == 0) return GNNSAFT_OK;
  GS_REQUIRE((n + 1) * (int64_t)hidden < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);   // 32-bit epilogue offsets
  const int64_t per_tower = (int64_t)(hidden / 2) * 5 * hidden;
  PostFoldA ap{x, agg, perm, tiles, num_tiles, 2 * per_tower, hidden};
  GemmBatch b;
  b.e[0] = GemmBatchEntry{nullptr, b_post0, u, 0, w_eff3};
  b.e[1] = GemmBatchEntry{nullptr, b_post1, u + hidden / 2, 4 * (int64_t)hidden, w_eff3 + per_tower * 6};
  for (int i = 2; i < kMaxGemmBatch; ++i) b.e[i] = b.e[0];
  EpiArgs ea{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
  return launch_w3_cfg<PostFoldA, false, false, false>(cfg, ap, 2, b, hidden / 2, hidden, n, hidden / 2, 5 * hidden, ea,
                                                       max_tiles, stream);
}

}  // namespace gs

// ---- C ABI: stage tests and tuning of the W3 kernels (include/gnnsaft.h)
extern "C" size_t gnnsaft_w3_image_bytes(int64_t rows, int64_t k) {
  if (rows < 1 || k < gs::kW3Kt || (k % gs::kW3Kt) != 0) return 0;
  return gs::w3_bytes(rows, k);
}

extern "C" int gnnsaft_w3_pack(const float *w, int64_t ldw, int32_t rows, int32_t k, void *image, gnnsaft_stream_t stream) {
  gs::W3PackItem it{w, static_cast<char *>(image), ldw, rows, k, rows, 0};
  return gs::launch_w3_pack(1, &it, static_cast<hipStream_t>(stream));
}

#ifdef GS_W3_VARIANTS
// timing experiments (tools/w3_variants.py): tile_config = cfg + 64 * variant, cfg 0 / 1 only, plain epilogue
template <int VAR>
static int gs_w3_variant(int cfg, const gs::PlainA &ap, const gs::GemmBatch &b, int n_pad, int64_t ldo, int64_t m, int n_out,
                         int k, const gs::EpiArgs &ea, hipStream_t st) {
  if (cfg == 0) return gs::launch_w3_one<128, 128, 2, 2, 1, 3, gs::PlainA, false, false, false, VAR>(ap, 1, b, n_pad, ldo, m, n_out, k, ea, 0, st);
  return gs::launch_w3_one<128, 256, 2, 2, 1, 2, gs::PlainA, false, false, false, VAR>(ap, 1, b, n_pad, ldo, m, n_out, k, ea, 0, st);
}
#endif

extern "C" int gnnsaft_debug_linear_w3(const float *a, int64_t lda, const void *w_image, const float *bias, float *out,
                                       int64_t ldo, int64_t m, int32_t n_out, int32_t k, float *stats,
                                       int32_t tile_config, gnnsaft_stream_t stream) {
  gs::GemmBatchEntry ent{nullptr, bias, out, 0, static_cast<const char *>(w_image)};
#ifdef GS_W3_VARIANTS
  if (tile_config >= 64) {
    const int var = tile_config / 64, cfg = tile_config % 64;
    GS_REQUIRE(cfg <= 1 && stats == nullptr, GNNSAFT_ERR_SHAPE);
    gs::GemmBatch b;
    for (int i = 0; i < gs::kMaxGemmBatch; ++i) b.e[i] = ent;
    gs::EpiArgs ea{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
    gs::PlainA ap{a, lda, m, k};
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (var) {
#define GS_V(v) case v: return gs_w3_variant<v>(cfg, ap, b, n_out, ldo, m, n_out, k, ea, st);
      GS_V(1) GS_V(2) GS_V(3) GS_V(4) GS_V(7) GS_V(8) GS_V(15) GS_V(16) GS_V(24) GS_V(31) GS_V(32) GS_V(39) GS_V(55) GS_V(64) GS_V(23)
#undef GS_V
      default: return GNNSAFT_ERR_UNSUPPORTED;
    }
  }
#endif
  gs::LinearEpilogue epi;
  epi.stats = stats;
  GS_REQUIRE(stats == nullptr || ldo == n_out, GNNSAFT_ERR_SHAPE);
  return gs::launch_linear_w3(a, lda, 1, &ent, n_out, ldo, m, n_out, k, epi, static_cast<hipStream_t>(stream), tile_config);
}
