// Wave-specialised split-bf16 GEMM on pre-split weight images (w3.hpp):  out = epilogue( A_virtual x W^T + b ).
//
// The arithmetic, the LDS stage image and the weight images are those of k_gemm_w3 (gemm_w3.hip): every f32 operand is
// the exact sum of three bf16 numbers, six v_mfma_f32_32x32x16_bf16 per k16 step, f32 accumulation -- the GEMMs behind
// PyG PNAConv's pre_nns / post_nns / lin (/root/reference/gnnepcsaft/train/models.py:69-80,128) at f32 accuracy.
// What differs is WHO does what.  Measured on the C3 update shape (tools/w3_variants.py): the
// matrix cores plus their fragment reads need 202-219 us, the operand staging (loads, split, LDS writes) 210-216 us,
// and a kernel whose every wave does both takes ~316 us whatever the tile, the buffering or the prefetch depth:
// inside one wave the two are a single in-order instruction stream, and waves that all run the same program meet at
// the same barriers with the same needs.  Here the workgroup's waves have ROLES:
//   * consumer waves (the first CW_M x CW_N): nothing but fragment reads and MFMAs -- the fragments of the next k16
//     step (across the stage boundary too) are requested before the MFMAs of the current one, so the matrix pipe's
//     instruction stream never waits for the LDS; then the epilogue;
//   * producer waves (the last four; one per SIMD, beside one consumer wave each): nothing but the operand path --
//     A rows and B pieces requested three stages ahead (register ring), A split into its bf16 planes, 16-byte LDS
//     writes -- under the consumers' MFMAs, on the SIMD's otherwise idle vector / memory issue slots.
// Three LDS stages: consumers read stage t (and pre-read the head of stage t + 1) while producers fill stage t + 2;
// ONE barrier per stage, on which nobody's next instruction depends.
// The A operand comes from a provider (gemm_prov.hpp) -- or from AggStager below: the PNA aggregation itself
// (aggregate.hip: k_pna_aggregate<kFusedQ>) done by the producers while they stage, so that the aggregates
// [N, 2, 4F] never exist in HBM (the no-tape forward's fused aggregate + update).
#include <cstdlib>

#include "common.hpp"
#include "gemm_epi.hpp"
#include "gemm_prov.hpp"
#include "w3.hpp"

namespace gs {

__device__ __forceinline__ uint32_t w3s_pack2(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// a lane's eight f32 of one 16-byte chunk (k = 4q .. 4q+3, 16+4q .. 16+4q+3) -> the chunk of each bf16 plane, written
// at `p` (plane pitch `plane` bytes).  a = hi + mid + lo exactly (x6.hpp), the planes keep the upper 16 bits of each.
__device__ __forceinline__ void w3s_split_store(char *p, int plane, const f32x4 v0, const f32x4 v1) {
  uint32_t e[8], m1[8], m2[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    e[t] = __float_as_uint(v0[t]);
    e[4 + t] = __float_as_uint(v1[t]);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float a = __uint_as_float(e[t]);
    const float r1 = a - __uint_as_float(e[t] & 0xffff0000u);
    m1[t] = __float_as_uint(r1);
    m2[t] = __float_as_uint(r1 - __uint_as_float(m1[t] & 0xffff0000u));
  }
  *reinterpret_cast<uint4 *>(p) = uint4{w3s_pack2(e[0], e[1]), w3s_pack2(e[2], e[3]), w3s_pack2(e[4], e[5]), w3s_pack2(e[6], e[7])};
  *reinterpret_cast<uint4 *>(p + plane) =
      uint4{w3s_pack2(m1[0], m1[1]), w3s_pack2(m1[2], m1[3]), w3s_pack2(m1[4], m1[5]), w3s_pack2(m1[6], m1[7])};
  *reinterpret_cast<uint4 *>(p + 2 * plane) =
      uint4{w3s_pack2(m2[0], m2[1]), w3s_pack2(m2[2], m2[3]), w3s_pack2(m2[4], m2[5]), w3s_pack2(m2[6], m2[7])};
}

constexpr int kW3sProducerWaves = 4;
constexpr int kW3sBufs = 3;

// VAR (timing experiments, tools/w3_variants.py; garbage results): bit 0 producers idle (no loads, no stash), bit 1
// consumers idle (no fragment reads, no MFMAs)
template <int BM, int BN, int CW_M, int CW_N, class AProv, bool STATS, bool AFFINE, bool RESID, int VAR = 0>
__global__ __launch_bounds__(64 * (CW_M * CW_N + kW3sProducerWaves)) void k_gemm_w3s(AProv ap, GemmBatch batch, int n_pad,
                                                                                       int64_t ldo, int n_out, int k,
                                                                                       EpiArgs epi) {
  constexpr int CW = CW_M * CW_N;                // consumer waves
  constexpr int PW = kW3sProducerWaves;
  constexpr int WTM = BM / CW_M;
  constexpr int WTN = BN / CW_N;
  static_assert(!STATS || WTM == kBnRowsPerGroup, "BatchNorm partials assume 64 rows per wave");
  constexpr int TM = WTM / 32;
  constexpr int TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA");
  constexpr int RPP = PW * 64 / 4;               // A rows covered by one pass of the producers (4 lanes per row)
  constexpr int A_R = BM / RPP;                  // rows per producer thread
  static_assert(A_R >= 1 && BM % RPP == 0, "the A tile is a whole number of producer passes");
  constexpr int PLANE = (BM + BN) * kW3RowBytes; // bytes per bf16 plane and stage: A rows, then B rows
  constexpr int STAGE = 3 * PLANE;
  constexpr int B_PIECES = 3 * BN / 16;          // 1-KiB pieces per stage (16 rows of one plane each)
  static_assert(B_PIECES % PW == 0, "every producer wave copies the same number of pieces");
  constexpr int B_PW = B_PIECES / PW;
  extern __shared__ __attribute__((aligned(256))) char lds[];   // kW3sBufs * STAGE

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileInfo ti = ap.tile(blockIdx.x, BM);
  if (ti.count <= 0) return;  // block-uniform, before any barrier
  const int n0 = blockIdx.y * BN;
  const GemmBatchEntry ent = batch.e[blockIdx.z];
  const int nk = k / kW3Kt;

  if (wave >= CW) {
    // ================================================================ producers
    const int ptid = tid - CW * 64;
    const int pw = wave - CW;
    const int q = ptid & 3;
    const int r0 = ptid >> 2;
    typename AProv::Row arow[A_R];
#pragma unroll
    for (int j = 0; j < A_R; ++j) {
      const int lr = r0 + RPP * j;
      arow[j] = ap.row(ti.row0 + (lr < ti.count ? lr : ti.count - 1), ent.a_off);   // clamped rows are never stored
    }
    const int a_lds = r0 * kW3RowBytes + w3_chunk_pos(q, r0) * 16;   // (RPP = 64: same swizzle for every j)
    const char *bsrc[B_PW];
    int bdst[B_PW];
#pragma unroll
    for (int jj = 0; jj < B_PW; ++jj) {
      const int i = pw + PW * jj;
      const int p = i / (BN / 16), rb = i % (BN / 16);
      int n = n0 + rb * 16 + (lane >> 2);
      n = n < n_pad ? n : n_pad - 1;                                   // clamped columns are never stored
      bsrc[jj] = ent.w3 + ti.w_off * 6 + ((int64_t)p * n_pad + n) * kW3RowBytes + (lane & 3) * 16;
      bdst[jj] = p * PLANE + (BM + rb * 16) * kW3RowBytes + lane * 16;
    }
    const int64_t bstep = (int64_t)3 * n_pad * kW3RowBytes;             // bytes between two stages of the image

    typename AProv::Raw ra[3][A_R][2];
    f32x4 rb[3][B_PW];
    // (any stage index: past the last stage the last one is repeated into a buffer nobody reads -- no branches)
    auto fetch = [&](int kt, typename AProv::Raw(&r)[A_R][2], f32x4(&b)[B_PW]) {
      if constexpr ((VAR & 1) != 0) return;
      const int kc = kt < nk ? kt : nk - 1;
#pragma unroll
      for (int jj = 0; jj < B_PW; ++jj) b[jj] = *reinterpret_cast<const f32x4 *>(bsrc[jj] + kc * bstep);
#pragma unroll
      for (int j = 0; j < A_R; ++j) {
        r[j][0] = ap.load(arow[j], kc * kW3Kt, 4 * q);
        r[j][1] = ap.load(arow[j], kc * kW3Kt, 16 + 4 * q);
      }
    };
    auto stash = [&](int kt, int buf, const typename AProv::Raw(&r)[A_R][2], const f32x4(&b)[B_PW]) {
      if constexpr ((VAR & 1) != 0) return;
      const int k0 = (kt < nk ? kt : nk - 1) * kW3Kt;
      char *st = lds + buf * STAGE;
#pragma unroll
      for (int jj = 0; jj < B_PW; ++jj) *reinterpret_cast<f32x4 *>(st + bdst[jj]) = b[jj];
#pragma unroll
      for (int j = 0; j < A_R; ++j)
        w3s_split_store(st + a_lds + j * (RPP * kW3RowBytes), PLANE, ap.finish_full(r[j][0], arow[j], k0, 4 * q),
                        ap.finish_full(r[j][1], arow[j], k0, 16 + 4 * q));
    };
    // prologue: stages 0 and 1 in LDS, stages 2, 3, 4 in flight
    fetch(0, ra[0], rb[0]);
    fetch(1, ra[1], rb[1]);
    fetch(2, ra[2], rb[2]);
    stash(0, 0, ra[0], rb[0]);
    fetch(3, ra[0], rb[0]);
    stash(1, 1, ra[1], rb[1]);
    fetch(4, ra[1], rb[1]);
    __syncthreads();
    // interval t: stage t + 2 into buffer (t + 2) % 3 (read last in interval t - 1), then the loads of stage t + 5
    for (int t = 0; t < nk; t += 3) {
      stash(t + 2, 2, ra[2], rb[2]);
      fetch(t + 5, ra[2], rb[2]);
      __syncthreads();
      if (t + 1 < nk) {
        stash(t + 3, 0, ra[0], rb[0]);
        fetch(t + 6, ra[0], rb[0]);
        __syncthreads();
      }
      if (t + 2 < nk) {
        stash(t + 4, 1, ra[1], rb[1]);
        fetch(t + 7, ra[1], rb[1]);
        __syncthreads();
      }
    }
    return;
  }

  // ================================================================== consumers
  const int wm = wave / CW_N;
  const int wn = wave % CW_N;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: lane (row fr, half hh) reads chunk 2 s + hh of its row at step s
  const int fr = lane & 31, hh = lane >> 5;
  const int f0 = fr * kW3RowBytes + w3_chunk_pos(hh, fr) * 16;   // step 0; step 1 = f0 ^ 32
  const int fa = (wm * WTM) * kW3RowBytes, fb = (BM + wn * WTN) * kW3RowBytes;
  bf16x8 af[2][TM][3], bf[2][TN][3];
  auto read = [&](int buf, int s, bf16x8(&a)[TM][3], bf16x8(&b)[TN][3]) {
    if constexpr ((VAR & 2) != 0) return;
    const char *base = lds + buf * STAGE + (s == 0 ? f0 : (f0 ^ 32));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + fa + p * PLANE + i * 32 * kW3RowBytes));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        b[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + fb + p * PLANE + j * 32 * kW3RowBytes));
  };
  auto mfma = [&](const bf16x8(&a)[TM][3], const bf16x8(&b)[TN][3]) {
    if constexpr ((VAR & 2) != 0) return;
    // six of the nine cross products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa[t]], b[j][pb[t]], acc[i][j], 0, 0, 0);
  };
  // interval t: MFMAs of stage t out of buffer t % 3; the fragments of its second step, then of the FIRST step of
  // stage t + 1 (filled during interval t - 1, behind the barrier that closed it) are requested ahead of the MFMAs
  // that hide them.  Past the last stage the pre-read takes a buffer of finite garbage nobody multiplies.
  // (sched_group_barrier: hipcc otherwise sinks each fragment read to just in front of its first use and waits for it
  // on the spot -- four exposed LDS round trips per stage)
  auto interval = [&](int buf, int next) {
    read(buf, 1, af[1], bf[1]);
    mfma(af[0], bf[0]);
    if constexpr ((VAR & 2) == 0) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);   // the step's fragment reads, all of them ...
      __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);     // ... then the MFMAs of the step before
    }
    read(next, 0, af[0], bf[0]);
    mfma(af[1], bf[1]);
    if constexpr ((VAR & 2) == 0) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);
    }
  };
  __syncthreads();
  read(0, 0, af[0], bf[0]);
  for (int t = 0; t < nk; t += 3) {
    interval(0, 1);
    __syncthreads();
    if (t + 1 < nk) {
      interval(1, 2);
      __syncthreads();
    }
    if (t + 2 < nk) {
      interval(2, 0);
      __syncthreads();
    }
  }
  gemm_epilogue<TM, TN, WTM, WTN, BM, BN, CW_M, STATS, AFFINE, RESID>(acc, ap, ti, ent, epi, n0, n_out, ldo, wm, wn, lane);
}

// --------------------------------------------------------------------------
// host-side dispatch
// --------------------------------------------------------------------------
template <int BM, int BN, int CWM, int CWN, class AProv, bool STATS, bool AFFINE, bool RESID, int VAR = 0>
static int launch_w3s_one(const AProv &ap, int nbatch, const GemmBatch &b, int n_pad, int64_t ldo, int64_t m, int n_out,
                          int k, const EpiArgs &ea, int64_t grid_x, hipStream_t stream) {
  constexpr size_t kLds = (size_t)kW3sBufs * 3 * (BM + BN) * kW3RowBytes;
  static_assert(kLds <= 160 * 1024, "three stages fit the CU's LDS");
  auto kern = k_gemm_w3s<BM, BN, CWM, CWN, AProv, STATS, AFFINE, RESID, VAR>;
  static std::atomic<unsigned long long> raised{0ull};
  if (kLds > 64 * 1024) GS_HIP(gs_raise_dynamic_lds(reinterpret_cast<const void *>(kern), kLds, raised));
  const dim3 grid((unsigned)(grid_x > 0 ? grid_x : gs_ceil_div(m, BM)), (unsigned)gs_ceil_div(n_out, BN), (unsigned)nbatch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * (CWM * CWN + kW3sProducerWaves)), kLds, stream, ap, b, n_pad, ldo, n_out, k, ea);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

// tile configurations: 0 = 128 x 128 (4 consumer waves of 64 x 64), 1 = 64 x 128 (4 consumer waves of 64 x 32)
template <class AProv, bool STATS, bool AFFINE, bool RESID>
static int launch_w3s_cfg(int cfg, const AProv &ap, int nbatch, const GemmBatch &b, int n_pad, int64_t ldo, int64_t m,
                          int n_out, int k, const EpiArgs &ea, int64_t grid_x, hipStream_t stream) {
  switch (cfg) {
    case 0: return launch_w3s_one<128, 128, 2, 2, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    case 1: return launch_w3s_one<64, 128, 1, 4, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    default: return GNNSAFT_ERR_UNSUPPORTED;
  }
}

int launch_linear_w3s(const float *a, int64_t lda, int nbatch, const GemmBatchEntry *entries, int n_pad, int64_t ldo,
                      int64_t m, int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg) {
  GS_REQUIRE(a != nullptr && entries != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(nbatch >= 1 && nbatch <= kMaxGemmBatch, GNNSAFT_ERR_SHAPE);
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
    return launch_w3s_cfg<PlainA, true, false, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  }
  if (af && rs) return launch_w3s_cfg<PlainA, false, true, true>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  if (af) return launch_w3s_cfg<PlainA, false, true, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  if (rs) return launch_w3s_cfg<PlainA, false, false, true>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
  return launch_w3s_cfg<PlainA, false, false, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
}

}  // namespace gs

#ifdef GS_W3_VARIANTS
template <int VAR>
static int gs_w3s_variant(const gs::PlainA &ap, const gs::GemmBatch &b, int n_pad, int64_t ldo, int64_t m, int n_out, int k,
                          const gs::EpiArgs &ea, hipStream_t st) {
  return gs::launch_w3s_one<128, 128, 2, 2, gs::PlainA, false, false, false, VAR>(ap, 1, b, n_pad, ldo, m, n_out, k, ea, 0, st);
}
#endif

// test / tuning hook (include/gnnsaft.h): tile_config 0 / 1 as launch_w3s_cfg; + 64 * variant in the ablation build
extern "C" int gnnsaft_debug_linear_w3s(const float *a, int64_t lda, const void *w_image, const float *bias, float *out,
                                        int64_t ldo, int64_t m, int32_t n_out, int32_t k, float *stats,
                                        int32_t tile_config, gnnsaft_stream_t stream) {
  gs::GemmBatchEntry ent{nullptr, bias, out, 0, static_cast<const char *>(w_image)};
#ifdef GS_W3_VARIANTS
  if (tile_config >= 64) {
    GS_REQUIRE(stats == nullptr && (tile_config % 64) == 0, GNNSAFT_ERR_SHAPE);
    gs::GemmBatch b;
    for (int i = 0; i < gs::kMaxGemmBatch; ++i) b.e[i] = ent;
    gs::EpiArgs ea{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
    gs::PlainA ap{a, lda, m, k};
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (tile_config / 64) {
      case 1: return gs_w3s_variant<1>(ap, b, n_out, ldo, m, n_out, k, ea, st);
      case 2: return gs_w3s_variant<2>(ap, b, n_out, ldo, m, n_out, k, ea, st);
      default: return GNNSAFT_ERR_UNSUPPORTED;
    }
  }
#endif
  gs::LinearEpilogue epi;
  epi.stats = stats;
  GS_REQUIRE(stats == nullptr || ldo == n_out, GNNSAFT_ERR_SHAPE);
  return gs::launch_linear_w3s(a, lda, 1, &ent, n_out, ldo, m, n_out, k, epi, static_cast<hipStream_t>(stream), tile_config);
}
