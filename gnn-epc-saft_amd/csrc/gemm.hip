// f32 matrix-core GEMM for gfx950:  out = epilogue( A_virtual  x  W^T + b ).
//
// One kernel template serves every dense op of the PNAPCSAFT forward
// (/root/reference/gnnepcsaft/train/models.py:84-103,128 and PyG PNAConv's
// pre_nns / post_nns / edge_encoder / lin).  The A operand is produced by a
// "provider" while it is staged global -> registers -> LDS, so the tensors
// PyG materialises ([E',T,3F] message input, [N,T,13F] update input) never
// exist in HBM:
//   PlainA : a row-major matrix, optional ReLU on load (extra pre/post layers)
//   PostA  : cat[x_i, A, A*amp_i, A*att_i]   (PNAConv update, scalers on load)
//   EdgeA  : relu(pq[dst_r] + pq[src_r] + rtab[combo_r])  (2nd pre-layer input)
//
// Arithmetic: __builtin_amdgcn_mfma_f32_32x32x2f32 (v_mfma_f32_32x32x2_f32),
// exact f32 fma chains -- bf16 MFMA is not admissible under the 1e-5 parity
// bar.  256 threads = 4 waves, every wave owns a 64 x (32|64) accumulator
// tile; BK = 32; LDS rows are padded to 36 floats so that the ds_read_b128
// fragment reads and the ds_write_b128 staging writes are conflict-free.
// K order inside a BK tile is permuted identically for A and B (lane half h
// reads k = 8g+4h .. 8g+4h+3), which only changes the (unspecified) summation
// order of the dot product.
//
// X6 mode (template flag): the same kernel on the bf16 matrix cores at f32 accuracy.  Every f32 operand value is
// split EXACTLY into three bf16 numbers, a = hi + mid + lo (8 significand bits each, by truncation: 3 x 8 = 24), while
// it is staged into LDS; the product a b is then the sum of nine bf16 x bf16 products, each exact in f32, of which the
// six largest are issued -- (hi,hi) (hi,mid) (mid,hi) (hi,lo) (lo,hi) (mid,mid) -- as v_mfma_f32_32x32x16_bf16 with
// f32 accumulation.  The three dropped terms are below 2^-24 |a b|: measured against f64 the result is as close as
// the f32 fma chain's (tests/test_gpu_stages.py), i.e. the 1e-5 parity bar sees no difference -- but the bf16 pipe
// runs 16x the f32 MFMA rate on gfx950 (2.5 PF vs 157 TF dense), so six of its instructions cost 6/16 of the f32
// form.  LDS holds three bf16 planes per operand (6 B per element, rows of 16 k padded to 48 B: conflict-free
// ds_read_b128 fragments), one k16 step per stage.
//
// Pipeline per BK tile: issue the raw global loads of tile t+1 (no consumer
// before the MFMAs, so no s_waitcnt in front of them), run the 16 k-steps of
// tile t out of LDS, then `finish` (scale / add / ReLU) the raw registers and
// write them to the other LDS buffer; one barrier per tile.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "gemm_epi.hpp"
#include "gemm_prov.hpp"
#include "x6.hpp"

namespace gs {

constexpr int LDS_LD = BK + 4;

// --------------------------------------------------------------------------
// kernel
// --------------------------------------------------------------------------
template <class T, class = void>
struct provider_needs_affine : std::false_type {};
template <class T>
struct provider_needs_affine<T, std::void_t<decltype(T::kNeedsAffine)>> : std::bool_constant<T::kNeedsAffine> {};

template <int BM, int BN, int WAVES_M, int WAVES_N, class AProv, bool STATS, bool AFFINE, bool RESID, bool X6 = false,
          int NT = 256>
__global__ __launch_bounds__(NT) void k_gemm_f32(AProv ap, GemmBatch batch, int64_t ldw, int64_t ldo, int64_t m,
                                                   int n_out, int k, EpiArgs epi) {
  static_assert(WAVES_M * WAVES_N * 64 == NT, "one wave per 64 threads");
  constexpr int WTM = BM / WAVES_M;
  constexpr int WTN = BN / WAVES_N;
  static_assert(!STATS || WTM == kBnRowsPerGroup, "BatchNorm partials assume 64 rows per wave");
  constexpr int TM = WTM / 32;
  constexpr int TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile is a multiple of the 32x32 MFMA");
  constexpr int KT = X6 ? 16 : BK;          // k per LDS stage
  constexpr int QPR = KT / 4;                // float4 per staged row
  constexpr int RSTEP = NT / QPR;            // rows covered by one pass of the NT threads
  constexpr int A_LD4 = BM * QPR / NT;
  constexpr int B_LD4 = (BN * QPR + NT - 1) / NT;
  static_assert(A_LD4 >= 1 && (BM * QPR) % NT == 0, "the A tile is a whole number of passes");
  // f32: rows of 36 floats; X6: three planes of 48-B rows -- 144 B per row and stage either way
  constexpr int STAGE = (BM + BN) * LDS_LD;
  static_assert(3 * kX6RowBytes == LDS_LD * 4, "both LDS formats spend 144 B per row and stage");
  constexpr int PLANE = (BM + BN) * kX6RowBytes;   // bytes per bf16 plane (X6)
  // (Tried and dropped in round 3: a wave-specialised variant -- 4 MFMA-only waves + 4 staging-only waves per
  // workgroup, 4 LDS buffers, one barrier per stage.  On [163840,256] x [256,512] it ran 349 us against 302 us for
  // this form: with the MFMAs removed its staging waves alone took 270 us, with the staging removed its MFMA waves
  // 192 us (53 % of the matrix-core rate) -- both sides are bound by their own load -> use latency chains, which
  // two mixed waves per SIMD hide better than one wave of each kind.)
  constexpr int kStages = 2;
  __shared__ __attribute__((aligned(16))) float lds[kStages * STAGE];
  constexpr bool kAffineA = provider_needs_affine<AProv>::value;
  __shared__ __attribute__((aligned(16))) float s_aff[kAffineA ? 2 * kMaxAffineK : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const TileInfo ti = ap.tile(blockIdx.x, BM);
  if (ti.count <= 0) return;  // block-uniform, before any barrier
  const int n0 = blockIdx.y * BN;
  const GemmBatchEntry ent = batch.e[blockIdx.z];

  // staging map: QPR float4 per staged row; thread -> (row r0 + RSTEP j, float4 column c4)
  const int c4 = tid & (QPR - 1);
  const int r0 = tid / QPR;

  typename AProv::Row arow[A_LD4];
#pragma unroll
  for (int j = 0; j < A_LD4; ++j) {
    const int lr = r0 + RSTEP * j;
    arow[j] = ap.row(ti.row0 + (lr < ti.count ? lr : ti.count - 1), ent.a_off);  // clamped rows are never stored
  }
  const float *wrow[B_LD4];
#pragma unroll
  for (int j = 0; j < B_LD4; ++j) {
    const int n = n0 + r0 + RSTEP * j;
    wrow[j] = ent.w + ti.w_off + (int64_t)(n < n_out ? n : n_out - 1) * ldw;  // clamped columns never stored
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // two register sets: the global loads run TWO k-tiles ahead of the MFMAs (one tile of MFMA work,
  // ~0.4 us for a 32x32 wave tile, is shorter than an L2/HBM round trip)
  typename AProv::Raw ra0[A_LD4], ra1[A_LD4];
  f32x4 rb0[B_LD4], rb1[B_LD4];
  const int nk = (k + KT - 1) / KT;

  // fetch / stash take ANY tile index: past the last tile the addresses are those of the last tile and the staged
  // data is zero.  That keeps the k-loop free of branches -- a conditional fetch makes hipcc lose track of the
  // outstanding loads and put an s_waitcnt vmcnt(0) at the loop head, which serialises prefetch and MFMAs.
  auto fetch = [&](int kt, typename AProv::Raw(&ra)[A_LD4], f32x4(&rb)[B_LD4]) {
    const int k0 = (kt < nk ? kt : nk - 1) * KT;
    const int kk = k0 + c4 * 4;
#pragma unroll
    for (int j = 0; j < A_LD4; ++j) ra[j] = ap.load(arow[j], k0, c4 * 4);
#pragma unroll
    for (int j = 0; j < B_LD4; ++j) rb[j] = gs_ld4(wrow[j] + (kk < k ? kk : 0));
  };
  // X6: split a staged float4 into its three bf16 planes (8 B each) at (row, c4)
  auto put3 = [&](char *plane0, int row, f32x4 v) {
    uint32_t h[4], md[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) gs_split3(v[e], h[e], md[e], l[e]);
    char *p = plane0 + row * kX6RowBytes + c4 * 8;
    *reinterpret_cast<uint2 *>(p) = uint2{gs_pack_hi16(h[0], h[1]), gs_pack_hi16(h[2], h[3])};
    *reinterpret_cast<uint2 *>(p + PLANE) = uint2{gs_pack_hi16(md[0], md[1]), gs_pack_hi16(md[2], md[3])};
    *reinterpret_cast<uint2 *>(p + 2 * PLANE) = uint2{gs_pack_hi16(l[0], l[1]), gs_pack_hi16(l[2], l[3])};
  };
  // `fast` (a std::bool_constant): the caller guarantees a live tile and K % KT == 0 -- no zero-fill selects
  auto stash_impl = [&](auto fast, int kt, const typename AProv::Raw(&ra)[A_LD4], const f32x4(&rb)[B_LD4]) {
    constexpr bool kFast = decltype(fast)::value;
    float *as = lds + (kt % kStages) * STAGE;
    float *bs = as + BM * LDS_LD;
    char *xs = reinterpret_cast<char *>(lds + (kt % kStages) * STAGE);   // X6: plane 0, A rows then B rows
    // (a provider with a side output -- BnResA writes x_{l+1} -- must not run on the tile one past the end)
    const bool live = (kFast && !kAffineA) || kt < nk;
    const int k0 = (live ? kt : nk - 1) * KT;
    const bool kok = kFast || (live && k0 + c4 * 4 < k);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < A_LD4; ++j) {
      f32x4 v;
      if constexpr (kAffineA) {
        v = live ? ap.finish(ra[j], arow[j], k0, c4 * 4, s_aff) : zero;   // (side output only for real tiles)
      } else if constexpr (kFast) {
        v = ap.finish_full(ra[j], arow[j], k0, c4 * 4);                     // no K tail to zero
      } else {
        v = ap.finish(ra[j], arow[j], k0, c4 * 4);
      }
      if constexpr (X6) {
        put3(xs, r0 + RSTEP * j, live ? v : zero);
      } else {
        gs_st4(as + (r0 + RSTEP * j) * LDS_LD + c4 * 4, live ? v : zero);
      }
    }
#pragma unroll
    for (int j = 0; j < B_LD4; ++j)
      if (BN >= RSTEP * (j + 1) || r0 + RSTEP * j < BN) {
        if constexpr (X6) {
          put3(xs, BM + r0 + RSTEP * j, kok ? rb[j] : zero);
        } else {
          gs_st4(bs + (r0 + RSTEP * j) * LDS_LD + c4 * 4, kok ? rb[j] : zero);
        }
      }
  };
  auto stash = [&](int kt, const typename AProv::Raw(&ra)[A_LD4], const f32x4(&rb)[B_LD4]) {
    stash_impl(std::false_type{}, kt, ra, rb);
  };

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;
  auto compute = [&](int kt) {
    if constexpr (X6) {
      // one k16 step per stage: lane (row frag_row, half h) reads k = 8h .. 8h+7 of its row from each plane
      const char *xa = reinterpret_cast<const char *>(lds + (kt % kStages) * STAGE) +
                       (wm * WTM + frag_row) * kX6RowBytes + (lane >> 5) * 16;
      const char *xb = reinterpret_cast<const char *>(lds + (kt % kStages) * STAGE) +
                       (BM + wn * WTN + frag_row) * kX6RowBytes + (lane >> 5) * 16;
      bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          af[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xa + p * PLANE + i * 32 * kX6RowBytes));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xb + p * PLANE + j * 32 * kX6RowBytes));
      // six of the nine cross products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
      constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa[t]], bf[j][pb[t]], acc[i][j], 0, 0, 0);
      return;
    }
    const float *as = lds + (kt & 1) * STAGE + (wm * WTM + frag_row) * LDS_LD + frag_k;
    const float *bs = lds + (kt & 1) * STAGE + BM * LDS_LD + (wn * WTN + frag_row) * LDS_LD + frag_k;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = gs_ld4(as + i * 32 * LDS_LD + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = gs_ld4(bs + j * 32 * LDS_LD + g * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (kAffineA) {   // [scale | shift] of the BatchNorm the provider applies: resident for the whole kernel
    for (int i = tid; i < 2 * ap.k; i += NT) s_aff[i] = i < ap.k ? ap.scale[i] : ap.shift[i - ap.k];
    __syncthreads();
  }
  if constexpr (X6) {
    // A k16 stage is 24 bf16 MFMAs per wave (~0.3 us): the global loads must run several stages ahead of the matrix
    // cores to cover an L2 / HBM round trip -- a ring of kRing = 4 register sets, set = stage mod 4.  A stage reads
    // 64 B of every row, HALF a cache line: the two stages of a line are fetched by back-to-back loads (the second
    // merges with the first's miss), at every other step, instead of one stage per step -- otherwise the second half
    // comes back to an L1 that has turned over and every line crosses the L2 interface twice (measured: the kernel
    // ran at L2 bandwidth).  At odd step t: stages t + 3 and t + 4 into the sets of stages t - 1 and t (both already
    // split into LDS); stage t + 1 is split into LDS after the step's MFMAs.  Unrolled: set indices are compile-time,
    // the waitcnt of every stash counts exactly the younger fetches.  Tiles past the end: clamped addresses, zero
    // data, no MFMAs.
    constexpr int kRing = 4;
    constexpr int kValuPerMfma = (32 * (A_LD4 + B_LD4) + 6 * TM * TN - 1) / (6 * TM * TN);   // ~32 VALU per staged float4
    typename AProv::Raw rra[kRing][A_LD4];
    f32x4 rrb[kRing][B_LD4];
#pragma unroll
    for (int u = 0; u < kRing; ++u) fetch(u, rra[u], rrb[u]);
    stash(0, rra[0], rrb[0]);
    __syncthreads();
    // Inside a step the wave's own MFMAs (24 per k16 stage at a 64 x 64 wave tile, 32 cycles each, only 8 of which
    // hold the vector issue port) and its staging work for the next stage (~100 VALU, the LDS writes) are independent:
    // asked to (sched_group_barrier), the scheduler interleaves them -- one MFMA, a handful of VALU -- instead of 24
    // MFMAs with an idle VALU followed by the split with an idle matrix core.
    // full ring trips: every tile live, no branch inside the step; with K % 16 == 0 (every op of the network) also
    // no zero-fill selects in the staging path -- the split's VALU work is what paces this kernel (measured: 225 VALU
    // per wave and stage against 24 MFMAs; 2 waves per SIMD issue-bound at 46 % matrix-core occupancy)
    const int nk_main = (k % KT) == 0 ? nk - nk % kRing : 0;
    for (int kt = 0; kt < nk_main; kt += kRing) {
#pragma unroll
      for (int u = 0; u < kRing; ++u) {
        if (u & 1) {
          fetch(kt + u + 3, rra[(u + 3) % kRing], rrb[(u + 3) % kRing]);
          fetch(kt + u + 4, rra[u], rrb[u]);
        }
        compute(kt + u);
        // (the tile one past the end is "staged" too: finite garbage in an LDS buffer nobody reads)
        stash_impl(std::true_type{}, kt + u + 1, rra[(u + 1) % kRing], rrb[(u + 1) % kRing]);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);        // fragment reads first
#pragma unroll
        for (int q = 0; q < 6 * TM * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA ...
          __builtin_amdgcn_sched_group_barrier(0x002, kValuPerMfma, 0);       // ... then a share of the split
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 3 * (A_LD4 + B_LD4), 0);  // LDS writes of the next stage
        __syncthreads();
      }
    }
    for (int kt = nk_main; kt < nk; kt += kRing) {   // (K not a multiple of 64: at most one trip, guarded)
#pragma unroll
      for (int u = 0; u < kRing; ++u) {
        if (u & 1) {
          fetch(kt + u + 3, rra[(u + 3) % kRing], rrb[(u + 3) % kRing]);
          fetch(kt + u + 4, rra[u], rrb[u]);
        }
        if (kt + u < nk) compute(kt + u);
        stash(kt + u + 1, rra[(u + 1) % kRing], rrb[(u + 1) % kRing]);
        __syncthreads();
      }
    }
  } else {
  // the big tile cannot afford the second register set (it would halve the waves per SIMD)
  constexpr bool kDeepPrefetch = BM * BN <= 128 * 64;
  fetch(0, ra0, rb0);
  stash(0, ra0, rb0);
  if (kDeepPrefetch) fetch(1, ra0, rb0);
  __syncthreads();

  if (!kDeepPrefetch) {
    // single register set (128x128): here the guarded form measured FASTER than both branch-free variants
    // (unconditional fetch, or last tile peeled): update at C3 0.98 ms vs 1.07 ms
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) fetch(kt + 1, ra0, rb0);
      compute(kt);
      if (more) stash(kt + 1, ra0, rb0);
      __syncthreads();
    }
  }
  // two tiles per trip; an odd tile count runs one extra all-zero tile (K of every op of the path is a multiple
  // of 2 BK except at H = 32)
  for (int kt = 0; kDeepPrefetch && kt < nk; kt += 2) {
    // LDS[0] = tile kt, set 0 = tile kt+1 in flight; start tile kt+2 into set 1
    fetch(kt + 2, ra1, rb1);
    compute(kt);
    stash(kt + 1, ra0, rb0);
    __syncthreads();
    // LDS[1] = tile kt+1, set 1 = tile kt+2 in flight; start tile kt+3 into set 0
    fetch(kt + 3, ra0, rb0);
    compute(kt + 1);
    stash(kt + 2, ra1, rb1);
    __syncthreads();
  }
  }

  gemm_epilogue<TM, TN, WTM, WTN, BM, BN, WAVES_M, STATS, AFFINE, RESID>(acc, ap, ti, ent, epi, n0, n_out, ldo, wm, wn, lane);
}

// --------------------------------------------------------------------------
// host-side dispatch
// --------------------------------------------------------------------------
// Tile configurations (block BM x BN, 4 waves).  Skinny f32 GEMMs (M large, N and K small) are
// latency- and quantisation-bound rather than operand-bandwidth-bound (f32 MFMA is 16x slower
// than bf16 per FLOP), so small tiles with more resident waves win; see DESIGN.md.
enum GemmCfg { kCfg256x32 = 0, kCfg128x64 = 1, kCfg128x128 = 2, kCfg64x64 = 3, kCfg64x128 = 4, kCfg128x32 = 5,
               kCfg128x128w8 = 6 /* 8 waves of 64 x 32: split-bf16 mode only */, kNumCfg = 7 };
static const int kCfgBM[kNumCfg] = {256, 128, 128, 64, 64, 128, 128};
static const bool kCfgStatsOk[kNumCfg] = {true, true, true, false, true, false, true};


static int pick_cfg(int64_t m, int n_out, int k, bool stats) {
  // measured on MI355X (tools/gemm_tune.py, tools/gemm_one.py; profiles/r01_gemm_tile_sweep.txt, r03_gemm_*)
  if (n_out <= 32) return stats ? kCfg256x32 : kCfg128x32;
  // enough rows that the big tile's lower operand traffic wins (C3: 164 k rows); in split-bf16 mode with 8 waves of
  // 64 x 32 (four waves per SIMD hide more of the staging latency than two: 3-9 % on the C3 shapes)
  const int big = gemm_x6_enabled() ? kCfg128x128w8 : kCfg128x128;
  // stats tiles have 64-row waves; up to ~40k rows the 64-row workgroups of 64x128 spread over more CUs than 128x128
  if (stats) return n_out <= 64 ? kCfg128x64 : (m < 40000 ? kCfg64x128 : big);
  // long-K update at H = 256, or many rows
  if (n_out >= 128 && (k >= 1024 || m >= 65536)) return big;
  return kCfg64x64;
}

// GNNSAFT_GEMM_X6 = 0 / 1 (default 1): the f32 matrix-core path, or the split-bf16 path at the same accuracy
bool gemm_x6_enabled() {
  static const bool on = [] {
    const char *e = getenv("GNNSAFT_GEMM_X6");
    return e == nullptr || e[0] != '0';
  }();
  return on;
}

template <int BM, int BN, int WM, int WN, class AProv, bool STATS, bool AFFINE, bool RESID, int NT = 256>
static void launch_one(const AProv &ap, int nbatch, const GemmBatch &b, int64_t ldw, int64_t ldo, int64_t m,
                       int n_out, int k, const EpiArgs &ea, int64_t grid_x, hipStream_t stream, int x6 = -1) {
  if constexpr (STATS && BM / WM != kBnRowsPerGroup) {
    return;  // not instantiated: BatchNorm partials need 64-row wave tiles
  } else {
    const dim3 grid((unsigned)(grid_x > 0 ? grid_x : gs_ceil_div(m, BM)), (unsigned)gs_ceil_div(n_out, BN),
                    (unsigned)nbatch);
    const bool use_x6 = x6 < 0 ? gemm_x6_enabled() : x6 != 0;
    if constexpr (NT != 256) {
      hipLaunchKernelGGL((k_gemm_f32<BM, BN, WM, WN, AProv, STATS, AFFINE, RESID, true, NT>), grid, dim3(NT), 0, stream,
                         ap, b, ldw, ldo, m, n_out, k, ea);
    } else if (use_x6)
      hipLaunchKernelGGL((k_gemm_f32<BM, BN, WM, WN, AProv, STATS, AFFINE, RESID, true>), grid, dim3(256), 0, stream,
                         ap, b, ldw, ldo, m, n_out, k, ea);
    else
      hipLaunchKernelGGL((k_gemm_f32<BM, BN, WM, WN, AProv, STATS, AFFINE, RESID, false>), grid, dim3(256), 0, stream,
                         ap, b, ldw, ldo, m, n_out, k, ea);
  }
}

template <class AProv, bool STATS, bool AFFINE, bool RESID>
static int launch_cfg(const AProv &ap, int nbatch, const GemmBatch &b, int64_t ldw, int64_t ldo, int64_t m, int n_out,
                      int k, const EpiArgs &ea, hipStream_t stream, int cfg = -1, int64_t grid_x = 0) {
  // explicit configurations (tuning / tests): cfg + 16 forces the split-bf16 kernels, cfg + 32 the f32 ones
  int x6 = -1;
  if (cfg >= 32) {
    x6 = 0;
    cfg -= 32;
  } else if (cfg >= 16) {
    x6 = 1;
    cfg -= 16;
  }
  if (cfg < 0 || cfg >= kNumCfg) cfg = pick_cfg(m, n_out, k, STATS);
  GS_REQUIRE(cfg >= 0 && cfg < kNumCfg && (!STATS || kCfgStatsOk[cfg]), GNNSAFT_ERR_UNSUPPORTED);
  switch (cfg) {
    case kCfg256x32:
      launch_one<256, 32, 4, 1, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x, stream, x6);
      break;
    case kCfg128x64:
      launch_one<128, 64, 2, 2, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x, stream, x6);
      break;
    case kCfg128x128:
      launch_one<128, 128, 2, 2, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x, stream, x6);
      break;
    case kCfg64x64:
      launch_one<64, 64, 2, 2, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x, stream, x6);
      break;
    case kCfg64x128:
      launch_one<64, 128, 1, 4, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x, stream, x6);
      break;
    case kCfg128x128w8:
      launch_one<128, 128, 2, 4, AProv, STATS, AFFINE, RESID, 512>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x,
                                                                   stream, x6);
      break;
    default:
      launch_one<128, 32, 4, 1, AProv, STATS, AFFINE, RESID>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, grid_x, stream, x6);
      break;
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

template <class AProv, bool FULL_EPILOGUES>
static int dispatch(const AProv &ap, int nbatch, const GemmBatchEntry *entries, int64_t ldw, int64_t ldo, int64_t m,
                    int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg = -1) {
  GS_REQUIRE(nbatch >= 1 && nbatch <= kMaxGemmBatch, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(m >= 0 && n_out >= 1 && k >= 4 && (k % 4) == 0 && (ldw % 4) == 0, GNNSAFT_ERR_SHAPE);
  // the epilogue indexes `out` / `residual` with 32-bit element offsets (row * ld + column)
  GS_REQUIRE(ldo >= 0 && epi.ldr >= 0 && (m + 1) * (ldo > epi.ldr ? ldo : epi.ldr) < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);
  if (m == 0) return GNNSAFT_OK;
  GemmBatch b;
  for (int i = 0; i < kMaxGemmBatch; ++i) b.e[i] = entries[i < nbatch ? i : 0];
  for (int i = 0; i < nbatch; ++i) {
    GS_REQUIRE(entries[i].w != nullptr && entries[i].out != nullptr, GNNSAFT_ERR_NULL);
    GS_REQUIRE((reinterpret_cast<uintptr_t>(entries[i].w) & 15) == 0, GNNSAFT_ERR_SHAPE);
  }
  EpiArgs ea{epi.scale, epi.shift, epi.relu_out, epi.residual, epi.ldr, epi.stats, epi.residual_is_mask,
             epi.bn_mean, epi.bn_var, epi.bn_eps};
  GS_REQUIRE((epi.bn_var == nullptr) == (epi.bn_mean == nullptr) && (epi.bn_var == nullptr || epi.scale != nullptr),
             GNNSAFT_ERR_NULL);
  GS_REQUIRE((epi.scale == nullptr) == (epi.shift == nullptr), GNNSAFT_ERR_NULL);
  GS_REQUIRE(epi.stats == nullptr || nbatch == 1, GNNSAFT_ERR_SHAPE);
  const bool st = epi.stats != nullptr, af = epi.scale != nullptr, rs = epi.residual != nullptr;
  if constexpr (!FULL_EPILOGUES) {
    GS_REQUIRE(!st && !af && !rs, GNNSAFT_ERR_UNSUPPORTED);
    return launch_cfg<AProv, false, false, false>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, stream, cfg);
  } else {
    if (st) {
      GS_REQUIRE(!af && !rs, GNNSAFT_ERR_UNSUPPORTED);  // train mode writes the pre-BN tensor
      return launch_cfg<AProv, true, false, false>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, stream, cfg);
    }
    if (af && rs) return launch_cfg<AProv, false, true, true>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, stream, cfg);
    if (af) return launch_cfg<AProv, false, true, false>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, stream, cfg);
    if (rs) return launch_cfg<AProv, false, false, true>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, stream, cfg);
    return launch_cfg<AProv, false, false, false>(ap, nbatch, b, ldw, ldo, m, n_out, k, ea, stream, cfg);
  }
}

int launch_linear(const float *a, int64_t lda, int relu_in, int nbatch, const GemmBatchEntry *entries, int64_t ldw,
                  int64_t ldo, int64_t m, int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg) {
  GS_REQUIRE(a != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE((lda % 4) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0, GNNSAFT_ERR_SHAPE);
  if (relu_in) {
    PlainReluA ap{a, lda, m, k};
    GS_REQUIRE(epi.stats == nullptr && epi.scale == nullptr && epi.residual == nullptr, GNNSAFT_ERR_UNSUPPORTED);
    return dispatch<PlainReluA, false>(ap, nbatch, entries, ldw, ldo, m, n_out, k, epi, stream, cfg);
  }
  PlainA ap{a, lda, m, k};
  return dispatch<PlainA, true>(ap, nbatch, entries, ldw, ldo, m, n_out, k, epi, stream, cfg);
}

int launch_linear_bnres(const float *y, const float *xprev, const float *scale, const float *shift, float *xout,
                        int nbatch, const GemmBatchEntry *entries, int64_t ldw, int64_t ldo, int64_t m, int n_out, int k,
                        hipStream_t stream) {
  GS_REQUIRE(y && scale && shift, GNNSAFT_ERR_NULL);
  GS_REQUIRE(k >= BK && (k % BK) == 0 && k <= kMaxAffineK && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(xprev) & 15) == 0 && (reinterpret_cast<uintptr_t>(xout) & 15) == 0,
             GNNSAFT_ERR_SHAPE);
  BnResA ap{y, xprev, scale, shift, xout, m, k};
  LinearEpilogue epi;
  return dispatch<BnResA, false>(ap, nbatch, entries, ldw, ldo, m, n_out, k, epi, stream);
}

int launch_pna_update(const float *x, const float *agg, const float *log_amp, const float *log_att,
                      const float *avg_deg_log, int64_t n, int hidden, const GemmBatchEntry *entries, int64_t ldo,
                      hipStream_t stream) {
  GS_REQUIRE(x && agg && log_amp && log_att && avg_deg_log, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  PostA ap{x, agg, log_amp, log_att, avg_deg_log, n, hidden};
  GemmBatchEntry e[2] = {entries[0], entries[1]};
  e[0].a_off = 0;
  e[1].a_off = 4 * (int64_t)hidden;
  LinearEpilogue epi;
  return dispatch<PostA, false>(ap, 2, e, 13 * (int64_t)hidden, ldo, n, hidden / 2, 13 * hidden, epi, stream);
}

int launch_pna_edge_mlp(const int32_t *src, const int32_t *dst, const int32_t *combo, int64_t rows, int hidden,
                        const float *pq, const float *rtab, const GemmBatchEntry *entries, int64_t ldo,
                        hipStream_t stream) {
  GS_REQUIRE(src && dst && combo && pq && rtab, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  EdgeA ap{src, dst, combo, pq, rtab, rows, hidden};
  GemmBatchEntry e[2] = {entries[0], entries[1]};
  e[0].a_off = 0;
  e[1].a_off = hidden;
  LinearEpilogue epi;
  return dispatch<EdgeA, false>(ap, 2, e, hidden, ldo, rows, hidden, hidden, epi, stream);
}

static int tiled_cfg_for(int hidden) { return pick_cfg(1 << 20, hidden / 2, 5 * hidden, false); }

int pna_fold_tile_rows(int hidden) {
  // rows per degree tile = BM of the configuration the folded update runs with (n_out = F/2 per tower);
  // every launch that walks the tile table must use a configuration with the same BM
  return kCfgBM[tiled_cfg_for(hidden)];
}

int launch_pna_update_folded(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                             const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden,
                             const float *w_eff /* [D,2,F/2,5F] */, const float *b_post0, const float *b_post1,
                             float *u, hipStream_t stream) {
  GS_REQUIRE(x && agg && perm && tiles && num_tiles && w_eff && u, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  if (n == 0) return GNNSAFT_OK;
  GS_REQUIRE((n + 1) * (int64_t)hidden < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);   // 32-bit epilogue offsets
  const int64_t per_tower = (int64_t)(hidden / 2) * 5 * hidden;
  PostFoldA ap{x, agg, perm, tiles, num_tiles, 2 * per_tower, hidden};
  GemmBatch b;
  b.e[0] = GemmBatchEntry{w_eff, b_post0, u, 0};
  b.e[1] = GemmBatchEntry{w_eff + per_tower, b_post1, u + hidden / 2, 4 * (int64_t)hidden};
  for (int i = 2; i < kMaxGemmBatch; ++i) b.e[i] = b.e[0];
  EpiArgs ea{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
  return launch_cfg<PostFoldA, false, false, false>(ap, 2, b, 5 * (int64_t)hidden, hidden, n, hidden / 2, 5 * hidden,
                                                    ea, stream, tiled_cfg_for(hidden), max_tiles);
}

int launch_linear_degree_tiled(const float *a, int64_t lda, const int32_t *perm, const int32_t *tiles,
                               const int32_t *num_tiles, int64_t max_tiles, int64_t w_stride, int nbatch,
                               const GemmBatchEntry *entries, int64_t ldw, int64_t ldo, int64_t n, int n_out, int k,
                               int hidden, hipStream_t stream) {
  GS_REQUIRE(a && perm && tiles && num_tiles && entries, GNNSAFT_ERR_NULL);
  GS_REQUIRE(nbatch >= 1 && nbatch <= kMaxGemmBatch && (k % 4) == 0 && (ldw % 4) == 0 && (lda % 4) == 0,
             GNNSAFT_ERR_SHAPE);
  if (n == 0) return GNNSAFT_OK;
  GS_REQUIRE((n + 1) * ldo < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);   // 32-bit epilogue offsets
  PermPlainA ap{a, lda, perm, tiles, num_tiles, w_stride, k};
  GemmBatch b;
  for (int i = 0; i < kMaxGemmBatch; ++i) b.e[i] = entries[i < nbatch ? i : 0];
  EpiArgs ea{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
  return launch_cfg<PermPlainA, false, false, false>(ap, nbatch, b, ldw, ldo, n, n_out, k, ea, stream,
                                                     tiled_cfg_for(hidden), max_tiles);
}

int launch_linear_concat2(const float *a0, int64_t lda0, int k0, const float *a1, int64_t lda1, int k1,
                          const GemmBatchEntry &entry, int64_t ldw, int64_t ldo, int64_t m, int n_out,
                          const LinearEpilogue &epi, hipStream_t stream) {
  GS_REQUIRE(a0 != nullptr && a1 != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE((lda0 % 4) == 0 && (lda1 % 4) == 0 && k0 >= BK && (k0 % BK) == 0 && k1 >= 4 && (k1 % 4) == 0 &&
                 (reinterpret_cast<uintptr_t>(a0) & 15) == 0 && (reinterpret_cast<uintptr_t>(a1) & 15) == 0,
             GNNSAFT_ERR_SHAPE);
  Concat2A ap{a0, a1, lda0, lda1, m, k0, k0 + k1};
  GS_REQUIRE(epi.stats == nullptr && epi.scale == nullptr, GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(entry.w != nullptr && entry.out != nullptr && (ldw % 4) == 0 && m >= 0 && n_out >= 1, GNNSAFT_ERR_SHAPE);
  if (m == 0) return GNNSAFT_OK;
  GS_REQUIRE((m + 1) * (ldo > epi.ldr ? ldo : epi.ldr) < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);   // 32-bit epilogue offsets
  GemmBatch b;
  for (int i = 0; i < kMaxGemmBatch; ++i) b.e[i] = entry;
  EpiArgs ea{nullptr, nullptr, epi.relu_out, epi.residual, epi.ldr, nullptr, epi.residual_is_mask, nullptr, nullptr,
             0.f};
  if (epi.residual != nullptr)
    return launch_cfg<Concat2A, false, false, true>(ap, 1, b, ldw, ldo, m, n_out, k0 + k1, ea, stream);
  return launch_cfg<Concat2A, false, false, false>(ap, 1, b, ldw, ldo, m, n_out, k0 + k1, ea, stream);
}

}  // namespace gs
