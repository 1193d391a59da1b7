// Fused PNA aggregation + degree-folded update for the no-tape forward:  u = post_nns[t][0]( cat[x, A_t * scalers] )
// with A_t = [mean | min | max | std] of the in-edge messages m~ = q[src] + rtab[class] formed INSIDE the GEMM's
// operand path -- the aggregates [N, 2, 4F] (1.34 GB per layer at BASELINE config 3, written by k_pna_aggregate and
// read straight back by the update GEMM) never exist in HBM.  Replaces, for that path, PyG PNAConv's
// aggregate + post_nns (/root/reference/gnnepcsaft/train/models.py:69-80,128).
//
// Structure: the wave-specialised split-bf16 GEMM of gemm_w3s.hip (three LDS stages of 32 k, one barrier per stage).
//   * producer waves (the last four) own the A operand.  A producer lane owns 8 of a row's 32 k per stage (w3.hpp).
//     K is walked in super-steps of four stages: first the x columns (F / 128 super-steps of plain loads), then one
//     super-step per 32-column slab c of the messages: the lane gathers its 8 columns of q[src_e] + rtab[class_e] for
//     every in-edge of its node, reduces them exactly as k_pna_aggregate<kFusedQ> does (sums of m - m_first, min, max;
//     std with PyG's clamp and mask) and holds mean | min | max | std of the slab -- the A values of the four stages
//     k = F (1 + a) + 32 c, a = 0..3 -- in registers; each stage's values are split into bf16 planes and written to
//     LDS when its buffer comes free.  The gathers of slab c + 1 are in flight while slab c is being written.  The
//     degree scalers never touch the activations: the rows of a tile share one in-degree and W_eff(d) (fold.hpp).
//   * consumer waves: fragment reads + MFMAs as in k_gemm_w3s, and the B operand: direct-to-LDS copies of the weight
//     image stage that matches the producers' k order (no registers, no VALU), requested two stages ahead.
// Same arithmetic as k_pna_aggregate<kFusedQ> + k_gemm_w3 on its output; the f32 accumulation runs over the k stages
// in another order (slab-major instead of aggregator-major), so results agree to rounding, not bit for bit.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "gemm_epi.hpp"
#include "gemm_prov.hpp"
#include "w3.hpp"

namespace gs {

__device__ __forceinline__ uint32_t ua_pack2(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// a lane's eight f32 of one 16-byte chunk -> the chunk of each bf16 plane (a = hi + mid + lo exactly, x6.hpp)
__device__ __forceinline__ void ua_split_store(char *p, int plane, const f32x4 v0, const f32x4 v1) {
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
  *reinterpret_cast<uint4 *>(p) = uint4{ua_pack2(e[0], e[1]), ua_pack2(e[2], e[3]), ua_pack2(e[4], e[5]), ua_pack2(e[6], e[7])};
  *reinterpret_cast<uint4 *>(p + plane) =
      uint4{ua_pack2(m1[0], m1[1]), ua_pack2(m1[2], m1[3]), ua_pack2(m1[4], m1[5]), ua_pack2(m1[6], m1[7])};
  *reinterpret_cast<uint4 *>(p + 2 * plane) =
      uint4{ua_pack2(m2[0], m2[1]), ua_pack2(m2[2], m2[3]), ua_pack2(m2[4], m2[5]), ua_pack2(m2[6], m2[7])};
}

// the same for ONE float4 slice: 8 bytes of each plane (PW = 8: two lanes share a chunk)
__device__ __forceinline__ void ua_split_store_half(char *p, int plane, const f32x4 v0) {
  uint32_t e[4], m1[4], m2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    e[t] = __float_as_uint(v0[t]);
    const float r1 = v0[t] - __uint_as_float(e[t] & 0xffff0000u);
    m1[t] = __float_as_uint(r1);
    m2[t] = __float_as_uint(r1 - __uint_as_float(m1[t] & 0xffff0000u));
  }
  *reinterpret_cast<uint2 *>(p) = uint2{ua_pack2(e[0], e[1]), ua_pack2(e[2], e[3])};
  *reinterpret_cast<uint2 *>(p + plane) = uint2{ua_pack2(m1[0], m1[1]), ua_pack2(m1[2], m1[3])};
  *reinterpret_cast<uint2 *>(p + 2 * plane) = uint2{ua_pack2(m2[0], m2[1]), ua_pack2(m2[2], m2[3])};
}

struct UpdateAggArgs {
  const float *x;        // [N, F] node state
  const float *q;        // [N, 2F] source terms of both towers
  const float *rtab;     // [classes, 2F] edge-class terms
  int classes;           // <= kUaMaxClasses
  int producer_prio;     // 1: s_setprio 1 for the producer waves
  int debug_var;         // development probe (GNNSAFT_UA_VAR; garbage results): 1 consumers idle, 2 no reductions, 4 no gathers, 8 no stash
  unsigned long long *stamps;   // development probe (tools/update_agg_stamps.py) or null: s_memtime stamps of tile 5
  const int32_t *rowptr, *src, *combo;          // destination-sorted CSR (self-loop rows included)
  const int32_t *perm, *tiles, *num_tiles;      // degree tiles (degree.hip)
  int64_t w_stride;      // f32 elements between two degrees' weight blocks (both towers)
  int f;                 // F = hidden
  const char *w3[2];     // W3 images of W_eff(0, t): [F/2, 5F]
  const float *bias[2];
  float *out[2];         // u + t F/2, row pitch F
};

// epilogue row map (gemm_epi.hpp): output row of tile row lr
struct UaRows {
  const int32_t *perm;
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return perm[t.row0 + lr]; }
};

constexpr int kUaEdgeBatch = 4;   // gathers in flight per lane and row
constexpr int kUaMaxClasses = 64; // edge classes whose slab of the class-term table fits the LDS left over (the reference: 60)

// PW producer waves: 4 -- a lane owns a whole 16-byte chunk (two float4 column slices) of its rows -- or 8 -- two lanes
// share a chunk, one float4 slice each (half the serial work per wave, two producer waves per SIMD: a single wave's
// dependent VALU / LDS stream issues at 5-7 cycles per instruction, measured with the consumers idle)
template <int BM, int BN, int CW_M, int CW_N, int PW>
__global__ __launch_bounds__(64 * (CW_M * CW_N + PW)) void k_update_agg_w3s(UpdateAggArgs a) {
  constexpr int CW = CW_M * CW_N;
  static_assert(PW == 4 || PW == 8, "four or eight producer waves");
  constexpr int HH = 8 / PW;                     // float4 column slices per lane and (row, stage): 2 or 1
  constexpr int WTM = BM / CW_M, WTN = BN / CW_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA");
  constexpr int RPP = 64;                        // rows per producer pass: PW lanes per row
  constexpr int R = BM / RPP;                    // rows per producer thread
  static_assert(R >= 1 && BM % RPP == 0, "the A tile is a whole number of producer passes");
  constexpr int PLANE = (BM + BN) * kW3RowBytes;
  constexpr int STAGE = 3 * PLANE;
  constexpr int B_PIECES = 3 * BN / 16;
  static_assert(B_PIECES % CW == 0, "every consumer wave copies the same number of pieces");
  constexpr int B_PW = B_PIECES / CW;
  constexpr int kRtabLds = 3 * STAGE;            // behind the stages: the class terms of ONE 32-column slab, [class][32] f32
  extern __shared__ __attribute__((aligned(256))) char lds[];   // 3 * STAGE + kUaMaxClasses * 128

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= a.num_tiles[0]) return;
  const int32_t *tt = a.tiles + 4 * (int64_t)blockIdx.x;
  const TileInfo ti{tt[1], tt[2], (int64_t)tt[0] * a.w_stride};
  const bool getenv_prio = a.producer_prio != 0;
  // stamps[role][stage][k]: role 0 = first producer wave, 1 = first consumer wave; lane 0 of tile 5, tower 0
  unsigned long long *stamp = (a.stamps != nullptr && blockIdx.x == 5 && blockIdx.z == 0 && lane == 0) ? a.stamps : nullptr;
  auto mark = [&](int role, int stage, int k) {
    if (stamp != nullptr && stage < 64) stamp[(role * 64 + stage) * 4 + k] = __builtin_readcyclecounter();
  };
  if (ti.count <= 0) return;   // block-uniform, before any barrier
  const int f = a.f;
  const int tower = blockIdx.z;
  const int n0 = blockIdx.y * BN;
  const int n_pad = f / 2;
  const int nx = f / 32;               // x stages
  const int nk = 5 * nx;               // all stages
  const int ns = nk / 4;               // super-steps of four stages

  if (wave >= CW) {
    // ================================================================ producers: the A operand
    // (younger than the consumer wave on their SIMD, they would lose every vector-issue arbitration against its MFMAs;
    // an MFMA needs the port 8 cycles in 32 and its pipe keeps running meanwhile -- MI355X_MICROARCH.md, two waves per SIMD)
    if (getenv_prio) __builtin_amdgcn_s_setprio(1);
    const int ptid = tid - CW * 64;
    const int q = ptid & 3;                       // chunk of the row's 32 k
    const int h0 = HH == 1 ? (ptid / 4) & 1 : 0;  // PW = 8: which float4 slice of the chunk this lane owns
    const int r0 = ptid / PW;
    const int a_lds = r0 * kW3RowBytes + w3_chunk_pos(q, r0) * 16 + h0 * 8;
    const float *xrow[R];
    int beg[R], cnt[R], sidx[R][kUaEdgeBatch], cidx[R][kUaEdgeBatch];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int lr = r0 + RPP * j;
      const int node = a.perm[ti.row0 + (lr < ti.count ? lr : ti.count - 1)];   // clamped rows are never stored
      xrow[j] = a.x + (int64_t)node * f;
      beg[j] = a.rowptr[node];
      cnt[j] = a.rowptr[node + 1] - beg[j];
#pragma unroll
      for (int e = 0; e < kUaEdgeBatch; ++e) {
        const int r = cnt[j] > 0 ? beg[j] + (e < cnt[j] ? e : cnt[j] - 1) : 0;   // (no edges: any valid row, unused)
        sidx[j][e] = a.src[r];
        cidx[j][e] = a.combo[r];
      }
    }
    const float *qt = a.q + tower * f;   // this tower's columns of the 2F-wide rows

    // values of the four stages of the CURRENT super-step: [stage][row][half] -- one 16-byte chunk of 8 k per
    // (stage, row); the loads of the NEXT super-step in flight: [row][edge | stage][half]; row 0 of the next super-step
    // reduced one stage early (its stage slots in `v` are still waiting for their buffers)
    f32x4 v[4][R][HH], g[R][kUaEdgeBatch][HH], t_std[HH];
    static_assert(kUaEdgeBatch == 4, "the in-flight registers double as the four x stages of a super-step");

    auto issue_x = [&](int S) __attribute__((always_inline)) {            // x columns 128 S .. 128 S + 127: plain loads, all rows
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int hh = 0; hh < HH; ++hh)
            g[j][s4][hh] = gs_ld4(xrow[j] + 128 * S + 32 * s4 + 4 * q + 16 * (HH == 2 ? hh : h0));
    };
    // (32-bit byte offsets from a uniform base instead of the 64-bit address per gather: measured SLOWER, 778 vs 733 us
    // at C3 -- the saddr form's scheduling, not the address arithmetic, is what the gathers wait for)
    auto issue_gather = [&](int c) __attribute__((always_inline)) {       // slab c: the first kUaEdgeBatch in-edges of every row
      const int cb = 32 * c + 4 * q;
#pragma unroll
      for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < kUaEdgeBatch; ++e)
#pragma unroll
          for (int hh = 0; hh < HH; ++hh)
            g[j][e][hh] = gs_ld4(qt + (int64_t)sidx[j][e] * (2 * f) + cb + 16 * (HH == 2 ? hh : h0));
    };
    const char *rl = lds + kRtabLds + (4 * q) * 4;    // this lane's columns of the slab's class rows
    // the reduction of k_pna_aggregate<kFusedQ> (aggregate.hip) on the lane's two float4 column slices of row j:
    // out[a][h] = mean | min | max | std
    // (the in-degree of every row of a degree tile is the tile's degree: a block-uniform count keeps the edge loop
    // free of per-lane masks.  Degrees the table clamps -- >= kDegreeBuckets, flagged GNNSAFT_FLAG_BAD_DEGREE -- are
    // garbage in, garbage out on either path.)
    const int deg = tt[0];
    // Straight-line reduction for the in-degrees a lane's in-flight gathers cover (1 .. kUaEdgeBatch: every heavy atom
    // with up to three bonds and its self-loop), the degree a compile-time constant: no branch per edge (each branch
    // cost a block of register moves and made hipcc canonicalise every operand of min / max again), the first edge
    // folded by hand (d = m - m_first is +0, its sums stay 0, min = max = m_first: what the loop of k_pna_aggregate
    // computes, bit for bit).  Same operations in the same order as aggregate.hip otherwise.
    // DEGC: the in-degree as a compile-time constant (0: run-time `deg`).  A quotient by 1, 2 or 4 is an exact
    // multiplication; by 3 (and by any run-time count) the correctly rounded quotient of ua_div1.
    auto finalize = [&](auto degc, const float (&v0)[4], const float (&s)[4], const float (&s2)[4], const float (&mn)[4],
                        const float (&mx)[4], int h, f32x4(&o_mean)[HH], f32x4(&o_min)[HH], f32x4(&o_max)[HH],
                        f32x4(&o_std)[HH]) __attribute__((always_inline)) {
      constexpr int DEGC = decltype(degc)::value;
      constexpr bool kPow2 = DEGC == 1 || DEGC == 2 || DEGC == 4;
      const float fc = DEGC > 0 ? (float)DEGC : (float)deg, inv = 1.f / fc;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float dmean = kPow2 ? s[t] * inv : gs_div_count(s[t], fc, inv);
        o_mean[h][t] = v0[t] + dmean;
        const float var = (kPow2 ? s2[t] * inv : gs_div_count(s2[t], fc, inv)) - dmean * dmean;
        o_std[h][t] = var;   // (the root is taken once behind the switch over the in-degree: std_of_var)
        o_min[h][t] = mn[t];
        o_max[h][t] = mx[t];
      }
    };
    auto reduce_fixed = [&](auto degc, int j, f32x4(&o_mean)[HH], f32x4(&o_min)[HH], f32x4(&o_max)[HH], f32x4(&o_std)[HH]) __attribute__((always_inline)) {
      constexpr int DEG = decltype(degc)::value;
      f32x4 tr[DEG][HH];
#pragma unroll
      for (int e = 0; e < DEG; ++e)
#pragma unroll
        for (int h = 0; h < HH; ++h)
          tr[e][h] = *reinterpret_cast<const f32x4 *>(rl + cidx[j][e] * 128 + 64 * (HH == 2 ? h : h0));
#pragma unroll
      for (int h = 0; h < HH; ++h) {
        float v0[4], s[4], s2[4], mn[4], mx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v0[t] = g[j][0][h][t] + tr[0][h][t];
          s[t] = 0.f;
          s2[t] = 0.f;
          mn[t] = v0[t];
          mx[t] = v0[t];
        }
#pragma unroll
        for (int e = 1; e < DEG; ++e)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float m = g[j][e][h][t] + tr[e][h][t];
            const float d = m - v0[t];
            s[t] = e == 1 ? d : s[t] + d;              // (0 + d = d)
            s2[t] = e == 1 ? d * d : s2[t] + d * d;    // (0 + d d = d d)
            mn[t] = fminf(mn[t], m);
            mx[t] = fmaxf(mx[t], m);
          }
        finalize(std::integral_constant<int, DEG>{}, v0, s, s2, mn, mx, h, o_mean, o_min, o_max, o_std);
      }
    };
    // any in-degree: zero (isolated node without self-loops) and more than kUaEdgeBatch in-edges (the further ones
    // gathered on the spot)
    auto reduce_any = [&](int c, int j, f32x4(&o_mean)[HH], f32x4(&o_min)[HH], f32x4(&o_max)[HH], f32x4(&o_std)[HH]) __attribute__((always_inline)) {
      const float inf = __builtin_huge_valf();
#pragma unroll
      for (int h = 0; h < HH; ++h) {
        const int hsel = HH == 2 ? h : h0;
        float s[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        float mn[4] = {inf, inf, inf, inf}, mx[4] = {-inf, -inf, -inf, -inf}, v0[4];
        const f32x4 t0v = *reinterpret_cast<const f32x4 *>(rl + cidx[j][0] * 128 + 64 * hsel);
#pragma unroll
        for (int t = 0; t < 4; ++t) v0[t] = g[j][0][h][t] + t0v[t];
        for (int e = 0; e < deg; ++e) {
          f32x4 mq, mt;
          if (e < kUaEdgeBatch) {   // (block-uniform)
            mq = e == 0 ? g[j][0][h] : (e == 1 ? g[j][1][h] : (e == 2 ? g[j][2][h] : g[j][3][h]));
            const int cl = e == 0 ? cidx[j][0] : (e == 1 ? cidx[j][1] : (e == 2 ? cidx[j][2] : cidx[j][3]));
            mt = *reinterpret_cast<const f32x4 *>(rl + cl * 128 + 64 * hsel);
          } else {
            const int r = beg[j] + e;
            mq = gs_ld4(qt + (int64_t)a.src[r] * (2 * f) + 32 * c + 4 * q + 16 * hsel);
            mt = *reinterpret_cast<const f32x4 *>(rl + a.combo[r] * 128 + 64 * hsel);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float m = mq[t] + mt[t];
            const float d = m - v0[t];
            s[t] += d;
            s2[t] += d * d;
            mn[t] = fminf(mn[t], m);
            mx[t] = fmaxf(mx[t], m);
          }
        }
        if (deg > 0) {
          finalize(std::integral_constant<int, 0>{}, v0, s, s2, mn, mx, h, o_mean, o_min, o_max, o_std);
        } else {
          o_mean[h] = o_min[h] = o_max[h] = o_std[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    auto reduce = [&](int c, int j, f32x4(&o_mean)[HH], f32x4(&o_min)[HH], f32x4(&o_max)[HH], f32x4(&o_std)[HH]) __attribute__((always_inline)) {
      static_assert(kUaEdgeBatch == 4, "one straight-line form per in-degree the in-flight gathers cover");
      switch (deg) {   // (block-uniform)
        case 1: reduce_fixed(std::integral_constant<int, 1>{}, j, o_mean, o_min, o_max, o_std); break;
        case 2: reduce_fixed(std::integral_constant<int, 2>{}, j, o_mean, o_min, o_max, o_std); break;
        case 3: reduce_fixed(std::integral_constant<int, 3>{}, j, o_mean, o_min, o_max, o_std); break;
        case 4: reduce_fixed(std::integral_constant<int, 4>{}, j, o_mean, o_min, o_max, o_std); break;
        default: reduce_any(c, j, o_mean, o_min, o_max, o_std); break;
      }
      // variance -> PyG's std, once for every in-degree.  std = var.clamp(min=1e-5).sqrt(), zeroed where std <=
      // sqrt(1e-5): with a correctly rounded root that mask is exactly `var <= 1e-5f` (sqrt(1e-5f) rounds to the
      // threshold, the next float above it does not), so it is taken on the variance; the clamp then has nothing left
      // to do (what it would clamp is masked, a NaN variance stays NaN as in torch).  The root is gs_sqrt_rn
      // (common.hpp: v_sqrt_f32 + the one-ulp residual test, 9 instructions for the ~13 of hipcc's sqrtf(), same bits
      // as k_pna_aggregate).  A bare 1-ulp v_sqrt_f32 was measured first (745 -> 707 us per launch at C3) and dropped: an
      // ulp of std is visible through train-mode BatchNorm (the population bar of tests/helpers.py).
#pragma unroll
      for (int h = 0; h < HH; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float var = o_std[h][t];
          o_std[h][t] = var <= 1e-5f ? 0.f : gs_sqrt_rn(var);
        }
    };
    // the piece of building super-step S that rides in stage slot s4 of the super-step before it (called AFTER the
    // slot's own stage left `v`): loads requested in slot 0; row 0 reduced in slot 2; in slot 3, with all of `v` written
    // out, row 1 (and the x values of an x super-step).  The slab's class terms were copied to LDS by the
    // consumers in the interval that runs beside slot 0.
    auto build = [&](int S, int s4) __attribute__((always_inline)) {
      if (S >= ns) return;                 // (block-uniform)
      const bool is_x = S < nx / 4;
      const int c = S - nx / 4;
      if (s4 == 0) {
        if (is_x)
          issue_x(S);
        else if ((a.debug_var & 4) == 0)
          issue_gather(c);
      }
      if ((a.debug_var & 2) != 0 && !is_x) return;
      // row 0 in slot 2: the stage slots 0..2 of `v` have been written out by then, only its std waits one more slot
      if (s4 == 2 && !is_x) reduce(c, 0, v[0][0], v[1][0], v[2][0], t_std);
      if (s4 == 3) {
        if (is_x) {
#pragma unroll
          for (int j = 0; j < R; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int hh = 0; hh < HH; ++hh) v[u][j][hh] = g[j][u][hh];
        } else {
#pragma unroll
          for (int hh = 0; hh < HH; ++hh) v[3][0][hh] = t_std[hh];
          if constexpr (R == 2) reduce(c, 1, v[0][1], v[1][1], v[2][1], v[3][1]);
          static_assert(R <= 2, "at most two rows per producer lane");
        }
      }
    };
    auto stash = [&](int s, const f32x4(&vs)[R][HH]) __attribute__((always_inline)) {   // stage s -> buffer s % 3
      char *st = lds + (s % 3) * STAGE + a_lds;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if constexpr (HH == 2)
          ua_split_store(st + j * (RPP * kW3RowBytes), PLANE, vs[j][0], vs[j][1]);
        else
          ua_split_store_half(st + j * (RPP * kW3RowBytes), PLANE, vs[j][0]);
      }
    };
    // stage s is written into its buffer during interval s - 2, i.e. after the barrier that closes interval s - 3 (the
    // buffer's last reader): one barrier behind every stage but the first, two more at the end -- 1 + nk in all, as
    // the consumers
    issue_x(0);
    build(0, 3);     // (waits for the loads: super-step 0 starts the tile)
    const bool pm = wave == CW;
    for (int S = 0; S < ns; ++S) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int s = 4 * S + s4;
        if (pm) mark(0, s, 0);
        if ((a.debug_var & 8) == 0) stash(s, v[s4]);
        if (pm) mark(0, s, 1);
        build(S + 1, s4);
        if (pm) mark(0, s, 2);
        if (s >= 1) __syncthreads();
        if (pm) mark(0, s, 3);
      }
    }
    __syncthreads();
    __syncthreads();
    return;
  }

  // ================================================================== consumers: B copies, fragment reads, MFMAs
  const int wm = wave / CW_N;
  const int wn = wave % CW_N;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // B pieces of this wave: piece i = (plane, block of 16 rows) = 1 KiB contiguous in the image and in the LDS stage
  const char *bsrc[B_PW];
  int bdst[B_PW];
#pragma unroll
  for (int jj = 0; jj < B_PW; ++jj) {
    const int i = wave + CW * jj;
    const int p = i / (BN / 16), rb = i % (BN / 16);
    int n = n0 + rb * 16 + (lane >> 2);
    n = n < n_pad ? n : n_pad - 1;                                   // clamped columns are never stored
    bsrc[jj] = a.w3[tower] + ti.w_off * 6 + ((int64_t)p * n_pad + n) * kW3RowBytes + (lane & 3) * 16;
    bdst[jj] = p * PLANE + (BM + rb * 16) * kW3RowBytes;              // wave-uniform; the hardware adds lane * 16
  }
  const int64_t bstep = (int64_t)3 * n_pad * kW3RowBytes;
  // the image stage behind the producers' stage s: x stages in place, then slab-major (mean, min, max, std of slab c)
  auto dma_b = [&](int s, int buf) {
    const int sc = s < nk ? s : nk - 1;
    const int r = sc - nx;
    const int kt = sc < nx ? sc : nx * (1 + (r & 3)) + (r >> 2);
#pragma unroll
    for (int jj = 0; jj < B_PW; ++jj)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc[jj] + kt * bstep),
                                       (__attribute__((address_space(3))) void *)(lds + buf * STAGE + bdst[jj]), 16, 0, 0);
  };

  // the class terms of slab c (this tower's columns 32 c .. 32 c + 31 of every class row: 128 bytes each) into LDS for
  // the producers' reductions: direct-to-LDS copies of 8 rows per wave instruction, shared out among the consumer waves
  const float *rt = a.rtab + tower * f;
  auto dma_rtab = [&](int c) {
    for (int piece = wave; piece * 8 < a.classes; piece += CW) {
      int cls = piece * 8 + (lane >> 3);
      cls = cls < a.classes ? cls : a.classes - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(rt + (int64_t)cls * (2 * f) + 32 * c + (lane & 7) * 4),
          (__attribute__((address_space(3))) void *)(lds + kRtabLds + piece * 1024), 16, 0, 0);
    }
  };
  const int fr = lane & 31, hh = lane >> 5;
  const int f0 = fr * kW3RowBytes + w3_chunk_pos(hh, fr) * 16;   // step 0; step 1 = f0 ^ 32
  const int fa = (wm * WTM) * kW3RowBytes, fb = (BM + wn * WTN) * kW3RowBytes;
  bf16x8 af[2][TM][3], bf[2][TN][3];
  auto read = [&](int buf, int s, bf16x8(&af_)[TM][3], bf16x8(&bf_)[TN][3]) {
    const char *base = lds + buf * STAGE + (s == 0 ? f0 : (f0 ^ 32));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        af_[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + fa + p * PLANE + i * 32 * kW3RowBytes));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bf_[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + fb + p * PLANE + j * 32 * kW3RowBytes));
  };
  auto mfma = [&](const bf16x8(&af_)[TM][3], const bf16x8(&bf_)[TN][3]) {
    // six of the nine cross products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_[i][pa[t]], bf_[j][pb[t]], acc[i][j], 0, 0, 0);
  };
  // interval t: B of stage t + 2 requested into buffer (t + 2) % 3 (read last in interval t - 1); MFMAs of stage t;
  // the fragments of the next k16 step (across the stage boundary too) requested ahead of the MFMAs that hide them
  auto interval = [&](int t, int buf, int next, int fill) {
    if (wave == 0) mark(1, t, 0);
    if ((a.debug_var & 1) != 0) return;
    dma_b(t + 2, fill);
    // beside the producers' slot 0 of super-step S = (t + 2) / 4 (they write stage t + 2 now): the class terms of the
    // slab super-step S + 1 is built from; its previous contents were last read two stages ago, behind a barrier
    if (((t + 2) & 3) == 0) {
      const int c = (t + 2) / 4 + 1 - nx / 4;
      if (c >= 0 && c < nx) dma_rtab(c);
    }
    read(buf, 1, af[1], bf[1]);
    mfma(af[0], bf[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);
    read(next, 0, af[0], bf[0]);
    mfma(af[1], bf[1]);
    __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);
    if (wave == 0) mark(1, t, 1);
  };
  dma_b(0, 0);
  dma_b(1, 1);
  if (nx / 4 == 1) dma_rtab(0);   // (F = 128: super-step 1, built during super-step 0, is already a message slab)
  __syncthreads();
  read(0, 0, af[0], bf[0]);
  for (int t = 0; t < nk; t += 3) {
    interval(t, 0, 1, 2);
    __syncthreads();
    if (t + 1 < nk) {
      interval(t + 1, 1, 2, 0);
      __syncthreads();
    }
    if (t + 2 < nk) {
      interval(t + 2, 2, 0, 1);
      __syncthreads();
    }
  }
  const UaRows rows{a.perm};
  const GemmBatchEntry ent{nullptr, a.bias[tower], a.out[tower], 0, nullptr};
  const EpiArgs epi{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
  gemm_epilogue<TM, TN, WTM, WTN, BM, BN, CW_M, false, false, false>(acc, rows, ti, ent, epi, n0, n_pad, (int64_t)f, wm, wn, lane);
}

template <int BM, int BN, int CWM, int CWN, int PW>
static int launch_update_agg_one(const UpdateAggArgs &a, int64_t max_tiles, hipStream_t stream) {
  constexpr size_t kLds = (size_t)3 * 3 * (BM + BN) * kW3RowBytes + kUaMaxClasses * 128;
  static_assert(kLds <= 160 * 1024, "three stages and a slab of class terms fit the CU's LDS");
  auto kern = k_update_agg_w3s<BM, BN, CWM, CWN, PW>;
  static std::atomic<unsigned long long> raised{0ull};
  if (kLds > 64 * 1024) GS_HIP(gs_raise_dynamic_lds(reinterpret_cast<const void *>(kern), kLds, raised));
  const dim3 grid((unsigned)max_tiles, (unsigned)gs_ceil_div(a.f / 2, BN), 2u);
  hipLaunchKernelGGL(kern, grid, dim3(64 * (CWM * CWN + PW)), kLds, stream, a);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

static unsigned long long *g_ua_stamps = nullptr;   // development probe: see gnnsaft_debug_update_agg_stamps

// does the fused form cover this hidden size?  (F a multiple of 128: super-steps of four 32-k stages; the tile rows of
// the degree table must be a tile height built here)
bool update_agg_supported(int hidden, int classes) {
  if (!gemm_x6_enabled() || hidden < 128 || (hidden % 128) != 0 || classes < 1 || classes > kUaMaxClasses) return false;
  const int rows = pna_fold_tile_rows(hidden);
  return rows == 128 || rows == 64;
}

int launch_pna_update_agg(const float *x, const float *q, const float *rtab, int classes, const int32_t *rowptr,
                          const int32_t *src, const int32_t *combo, const int32_t *perm, const int32_t *tiles,
                          const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const char *w_eff3,
                          const float *b_post0, const float *b_post1, float *u, hipStream_t stream) {
  GS_REQUIRE(x && q && rtab && rowptr && src && combo && perm && tiles && num_tiles && w_eff3 && u, GNNSAFT_ERR_NULL);
  GS_REQUIRE(update_agg_supported(hidden, classes), GNNSAFT_ERR_UNSUPPORTED);
  if (n == 0) return GNNSAFT_OK;
  GS_REQUIRE((n + 1) * (int64_t)(2 * hidden) < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);   // 32-bit epilogue offsets
  const int64_t per_tower = (int64_t)(hidden / 2) * 5 * hidden;
  UpdateAggArgs a;
  a.x = x;
  a.q = q;
  a.rtab = rtab;
  a.classes = classes;
  static const int prio = [] {
    const char *e = getenv("GNNSAFT_UA_PRIO");
    return e != nullptr && e[0] == '1' ? 1 : 0;   // (measured: no gain, 757 vs 733 us at C3)
  }();
  a.producer_prio = prio;
  a.stamps = g_ua_stamps;
  {
    const char *e = getenv("GNNSAFT_UA_VAR");
    a.debug_var = e != nullptr ? atoi(e) : 0;
  }
  a.rowptr = rowptr;
  a.src = src;
  a.combo = combo;
  a.perm = perm;
  a.tiles = tiles;
  a.num_tiles = num_tiles;
  a.w_stride = 2 * per_tower;
  a.f = hidden;
  a.w3[0] = w_eff3;
  a.w3[1] = w_eff3 + per_tower * 6;
  a.bias[0] = b_post0;
  a.bias[1] = b_post1;
  a.out[0] = u;
  a.out[1] = u + hidden / 2;
  const int rows = pna_fold_tile_rows(hidden);
  // GNNSAFT_UA_WAVES=16: 8 consumer (64 x 32) + 8 producer waves (two lanes per chunk) on the 128-row tile instead of
  // 4 + 4.  Measured at C3: 775 vs 786 us per launch, the producers alone 455 vs 518 us -- the kernel's time stays near
  // the SUM of what its two roles need alone, so halving each producer wave's serial work buys
  // 1.5 %; the 4 + 4 form (no register spills, 512 threads) stays the default.
  static const int wide = [] {
    const char *e = getenv("GNNSAFT_UA_WAVES");
    return e != nullptr && atoi(e) == 16 ? 1 : 0;
  }();
  if (rows == 128 && wide) return launch_update_agg_one<128, 128, 2, 4, 8>(a, max_tiles, stream);   // 8 + 8 waves
  if (rows == 128) return launch_update_agg_one<128, 128, 2, 2, 4>(a, max_tiles, stream);
  return launch_update_agg_one<64, 64, 1, 2, 4>(a, max_tiles, stream);
}

}  // namespace gs

// development probe (tools/update_agg_stamps.py; not part of the product path): a device buffer of 2 * 64 * 4 uint64 that
// the next launches fill with s_memtime stamps of one tile's first producer / consumer wave; NULL switches it off
extern "C" int gnnsaft_debug_update_agg_stamps(void *device_buffer) {
  gs::g_ua_stamps = static_cast<unsigned long long *>(device_buffer);
  return GNNSAFT_OK;
}

// stage-test entry point (include/gnnsaft.h): aggregation + folded update in one launch, operands as the two-launch
// pair gnnsaft_pna_aggregate_src + gnnsaft_pna_update_folded takes them, the folded weights as W3 images
extern "C" int gnnsaft_pna_update_agg(const float *x, const float *q, const float *rtab, int32_t num_classes,
                                      const int32_t *rowptr, const int32_t *src, const int32_t *combo,
                                      const int32_t *perm, const int32_t *tiles, const int32_t *num_tiles,
                                      int64_t num_nodes, int32_t hidden, const void *w_eff_images, const float *b_post0,
                                      const float *b_post1, float *u, gnnsaft_stream_t stream) {
  return gs::launch_pna_update_agg(x, q, rtab, num_classes, rowptr, src, combo, perm, tiles, num_tiles,
                                   gnnsaft_degree_tiles_capacity(num_nodes, hidden), num_nodes, hidden,
                                   static_cast<const char *>(w_eff_images), b_post0, b_post1, u,
                                   static_cast<hipStream_t>(stream));
}
