// Split-bf16 GEMM on W3 weight images with the A OPERAND IN REGISTERS:  out = epilogue( A_virtual x W^T + b ).
//
// Same arithmetic as k_gemm_w3 / k_gemm_f32<X6> (x6.hpp: every f32 operand is the exact sum of three bf16 numbers, six
// v_mfma_f32_32x32x16_bf16 per k16 step, f32 accumulation) for the GEMMs behind PyG PNAConv's pre_nns / post_nns / lin
// (/root/reference/gnnepcsaft/train/models.py:69-80,128).  What changes is, again, the operand path.  In k_gemm_w3 an A
// element travels global -> registers (staging lane) -> split -> LDS -> registers (fragment lane) and every stage ends in
// a barrier that both operands wait for; the measured result was a kernel that costs the SUM of
// its MFMA time and its operand-path time in every arrangement of waves.  Here:
//   * a wave owns 32 rows and ALL BN columns of the workgroup's tile, so nobody else needs its A rows: the lane that
//     feeds row r, k-half h of the MFMA loads exactly those f32 from global memory (the W3 image fixes the order of k
//     inside a 32-k stage: chunk c = 2 s + h holds k = 4c..4c+3, 16+4c..16+4c+3 -- two float4 of one cache line per
//     k16 step s), splits them in registers and hands the three bf16x8 to the matrix core.  No LDS writes, no A
//     fragment reads, no barrier between an A load and its use; the split (5.5 VALU per element, each element split
//     ONCE per BN output columns) is 1.8 VALU instructions per MFMA at BN = 128 and rides in the MFMAs' issue gaps
//     (MI355X_MICROARCH.md: <= 5 single-issue instructions hide per v_mfma_f32_32x32x16_bf16);
//   * A loads run three stages ahead in a ring of raw registers;
//   * B (the weight image) is copied global -> LDS by direct-to-LDS loads into a ring of three stages, two stages
//     ahead; the wave that issued a piece waits for it with a COUNTED s_waitcnt vmcnt (its own younger A loads stay
//     in flight) in front of a bare s_barrier -- __syncthreads() would drain vmcnt(0), i.e. every prefetch, once per
//     stage.  One barrier per 32-k stage; LDS traffic is B fragments only.
// Two workgroups of four waves per CU (two waves per SIMD, <= 256 registers each, 72 KB of LDS each at BN = 128).
#include <cstdlib>

#include "common.hpp"
#include "gemm_epi.hpp"
#include "gemm_prov.hpp"
#include "w3.hpp"

namespace gs {

__device__ __forceinline__ uint32_t ar_pack2(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// eight f32 (the two float4 of one k16 step) -> the three bf16x8 fragments hi | mid | lo  (a = hi + mid + lo, x6.hpp)
__device__ __forceinline__ void ar_split8(const f32x4 v0, const f32x4 v1, bf16x8 (&o)[3]) {
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
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  o[0] = __builtin_bit_cast(bf16x8, u32x4{ar_pack2(e[0], e[1]), ar_pack2(e[2], e[3]), ar_pack2(e[4], e[5]), ar_pack2(e[6], e[7])});
  o[1] = __builtin_bit_cast(bf16x8, u32x4{ar_pack2(m1[0], m1[1]), ar_pack2(m1[2], m1[3]), ar_pack2(m1[4], m1[5]), ar_pack2(m1[6], m1[7])});
  o[2] = __builtin_bit_cast(bf16x8, u32x4{ar_pack2(m2[0], m2[1]), ar_pack2(m2[2], m2[3]), ar_pack2(m2[4], m2[5]), ar_pack2(m2[6], m2[7])});
}

// The A loads are issued by hand (inline asm), not by the compiler: with direct-to-LDS copies and register loads in
// flight together hipcc's wait insertion gives up counting and drains vmcnt(0) once per loop trip (seen in the ISA), i.e.
// it would wait for loads issued a few instructions earlier.  The kernel waits itself: one counted s_waitcnt vmcnt in
// front of the stage's barrier.  The registers a wait makes valid pass THROUGH the wait statement ("+v"), so no use
// of them can be scheduled above it; the memory clobber keeps LDS / global accesses on their side of the barrier.
__device__ __forceinline__ void ar_load16(f32x4 &dst, const float *base, int byte_off_sel) {
  // (immediate offsets: the four pieces of a stage sit at +0, +64, +32, +96 bytes of one address register pair)
  switch (byte_off_sel) {
    case 0: asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(base)); break;
    case 1: asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(dst) : "v"(base)); break;
    case 2: asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(dst) : "v"(base)); break;
    default: asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(dst) : "v"(base)); break;
  }
}
template <int N>
__device__ __forceinline__ void ar_wait_vm_barrier(f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3) {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void ar_wait_vm(f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(N) : "memory");
}

// WAVES x 32 rows, TN x 32 columns per workgroup; MINW: waves per SIMD the register allocation must allow.
// TOWERS = 2: the workgroup's waves split into two groups, one per batch entry (the two towers of the degree-folded PNA
// update: same rows, each tower its own A columns, weight image, bias and output block); the workgroup then owns
// 32 WAVES / 2 rows x (2 x 32 TN) columns and one launch covers both towers with twice the waves per row tile.
template <int WAVES, int TN, int MINW, class AProv, bool AFFINE, bool RESID, int TOWERS = 1>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_gemm_ar(AProv ap, GemmBatch batch, int n_pad, int64_t ldo, int n_out,
                                                              int k, EpiArgs epi, unsigned long long *stamps) {
  static_assert(TOWERS == 1 || TOWERS == 2, "one or two batch entries per workgroup");
  constexpr int WR = WAVES / TOWERS;             // waves (= 32-row tiles) per tower
  constexpr int BM = 32 * WR, BNT = 32 * TN;     // rows per workgroup, columns per tower
  constexpr int BN = BNT * TOWERS;               // weight rows per LDS stage
  constexpr int PLANE = BN * kW3RowBytes;        // bytes per bf16 plane and stage (B only)
  constexpr int STAGE = 3 * PLANE;
  constexpr int B_PIECES = 3 * BN / 16;          // 1-KiB pieces per stage (16 rows of one plane each)
  static_assert(B_PIECES % WAVES == 0, "every wave copies the same number of pieces");
  constexpr int B_PW = B_PIECES / WAVES;
  static_assert(sizeof(typename AProv::Raw) == sizeof(f32x4), "one 16-byte load per provider call (the counted waits)");
  extern __shared__ __attribute__((aligned(256))) char lds[];   // 3 * STAGE

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tw = TOWERS == 2 ? wave / WR : 0;    // this wave's tower
  const int wr = TOWERS == 2 ? wave % WR : wave; // ... and row tile
  const TileInfo ti = ap.tile(blockIdx.x, BM);
  if (ti.count <= 0) return;   // block-uniform, before any barrier
  const int n0 = blockIdx.y * BNT;
  const GemmBatchEntry ent = batch.e[TOWERS == 2 ? tw : (int)blockIdx.z];
  const int nk = k / kW3Kt;

  // development probe (make stamps: -DGS_AR_STAMPS; gnnsaft_debug_ar_stamps, tools/ar_stamps.py): s_memtime of wave 0 of
  // one mid-grid tile, two per stage -- in front of the stage's wait + barrier and behind it.  NOT in the product
  // build: the extra branches change hipcc's register assignment around the hand-issued loads (tools/check_ar_isa.py).
#ifdef GS_AR_STAMPS
  unsigned long long *stamp =
      (stamps != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0 && lane == 0) ? stamps : nullptr;
  auto mark = [&](int i) {
    if (stamp != nullptr && i < 256) stamp[i] = __builtin_readcyclecounter();
  };
#else
  (void)stamps;
  auto mark = [](int) {};
#endif

  const int fr = lane & 31, hh = lane >> 5;
  const int lr = wr * 32 + fr;
  const typename AProv::Row arow = ap.row(ti.row0 + (lr < ti.count ? lr : ti.count - 1), ent.a_off);   // clamped rows are never stored

  // ---- B pieces of this wave: piece i = (plane, block of 16 rows) = 1 KiB contiguous in the image AND in the LDS stage
  const char *bsrc[B_PW];
  int bdst[B_PW];
#pragma unroll
  for (int jj = 0; jj < B_PW; ++jj) {
    const int i = wave + WAVES * jj;
    const int p = i / (BN / 16), rr = i % (BN / 16);                 // plane, 16-row block of the stage
    const int pt = rr / (BNT / 16), rb = rr % (BNT / 16);            // ... = tower, block inside the tower's columns
    int n = n0 + rb * 16 + (lane >> 2);
    n = n < n_pad ? n : n_pad - 1;                                   // clamped columns are never stored
    bsrc[jj] = batch.e[TOWERS == 2 ? pt : (int)blockIdx.z].w3 + ti.w_off * 6 + ((int64_t)p * n_pad + n) * kW3RowBytes + (lane & 3) * 16;
    bdst[jj] = p * PLANE + rr * 16 * kW3RowBytes;                    // wave-uniform; the hardware adds lane * 16
  }
  const int64_t bstep = (int64_t)3 * n_pad * kW3RowBytes;             // bytes between two stages of the image
  auto dma_b = [&](int kt, int buf) {
    const int64_t off = (int64_t)(kt < nk ? kt : nk - 1) * bstep;    // past the end: the last stage again, into a buffer nobody reads
#pragma unroll
    for (int jj = 0; jj < B_PW; ++jj)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc[jj] + off),
                                       (__attribute__((address_space(3))) void *)(lds + buf * STAGE + bdst[jj]), 16, 0, 0);
  };
  typename AProv::Raw raw[3][4];
  auto fetch_a = [&](int kt, typename AProv::Raw(&r)[4]) {
    const int k0 = (kt < nk ? kt : nk - 1) * kW3Kt;   // past the end: the last stage again (never multiplied)
    const float *p = ap.ptr(arow, k0, 4 * hh);
    ar_load16(r[0].v, p, 0);      // step 0: chunk hh      = k0 + 4 hh .. +3, k0 + 16 + 4 hh .. +3
    ar_load16(r[1].v, p, 1);
    ar_load16(r[2].v, p, 2);      // step 1: chunk 2 + hh  = k0 + 8 + 4 hh .. +3, k0 + 24 + 4 hh .. +3
    ar_load16(r[3].v, p, 3);
  };
  bf16x8 af[2][3];   // [k16 step][plane] of the stage being multiplied
  auto split = [&](int kt, int s, const typename AProv::Raw(&r)[4]) {   // half s of stage kt -> af[s]
    const int k0 = (kt < nk ? kt : nk - 1) * kW3Kt;
    ar_split8(ap.finish_full(r[2 * s], arow, k0, 8 * s + 4 * hh), ap.finish_full(r[2 * s + 1], arow, k0, 16 + 8 * s + 4 * hh), af[s]);
  };

  f32x16 acc[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  // B fragment of lane (column fr of a 32-column tile, half hh): chunk 2 s + hh of its row at step s
  const int f0 = (tw * BNT + fr) * kW3RowBytes + w3_chunk_pos(hh, fr) * 16;   // step 0; step 1 = f0 ^ 32
  bf16x8 bf[2][TN][3];
  auto read_b = [&](int buf, int s) {
    const char *base = lds + buf * STAGE + (s == 0 ? f0 : (f0 ^ 32));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        bf[s][j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(base + p * PLANE + j * 32 * kW3RowBytes));
  };
  auto mfma = [&](int s) {
    // six of the nine cross products, smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][pa[t]], bf[s][j][pb[t]], acc[0][j], 0, 0, 0);
  };
  // the scheduler's order for one k16 step: the 3 TN fragment reads of the NEXT step first, then the step's 6 TN MFMAs
  // with the split of the next A half (44 VALU) spread over their issue gaps
  auto order = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 3 * TN, 0);
#pragma unroll
    for (int i = 0; i < 6 * TN; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, (44 + 6 * TN - 1) / (6 * TN), 0);
    }
  };

  // Stage s (ring buffer s % 3; bf[0] / af[0] hold its step 0 on entry):
  //   B(s + 2) requested | fragments of step 1 read | MFMAs of step 0, beside them the split of A(s)'s second half |
  //   A(s + 3) requested into the raw slot that just came free | this wave's pieces of B(s + 1) waited for (all but the
  //   B_PW + 4 operations of this stage) and the workgroup meets: B(s + 1) is complete, every read of buffer (s - 1) % 3
  //   is done | fragments of step 0 of stage s + 1 read | MFMAs of step 1, beside them the split of A(s + 1)'s first half.
  //   Counted wait: vmcnt retires in issue order, and B(s + 1) was issued at the top of stage s - 1, in front of that
  //   stage's A request -- so the youngest B_PW + 8 operations (A(s + 2); B(s + 2) and A(s + 3) of this stage) may stay
  //   in flight and the wait makes exactly A(s + 1) and B(s + 1) valid: two stages of A per lane outstanding at every
  //   barrier (98 KB per CU with two workgroups; MI355X_MICROARCH.md: ~72 KB in flight per CU sustain the HBM rate).
  auto stage = [&](int s, int buf, int next, int fill, typename AProv::Raw(&r_cur)[4], typename AProv::Raw(&r_next)[4]) {
    dma_b(s + 2, fill);
    read_b(buf, 1);
    mfma(0);
    split(s, 1, r_cur);
    order();
    fetch_a(s + 3, r_cur);
    mark(2 * s);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ar_wait_vm_barrier<B_PW + 8>(r_next[0].v, r_next[1].v, r_next[2].v, r_next[3].v);
    mark(2 * s + 1);
    read_b(next, 0);
    mfma(1);
    split(s + 1, 0, r_next);
    order();
  };
  dma_b(0, 0);
  dma_b(1, 1);
  fetch_a(0, raw[0]);
  fetch_a(1, raw[1]);
  fetch_a(2, raw[2]);
  ar_wait_vm_barrier<8>(raw[0][0].v, raw[0][1].v, raw[0][2].v, raw[0][3].v);     // all but A(1), A(2): both B stages, A(0)
  split(0, 0, raw[0]);
  read_b(0, 0);
  const int ntrip = nk / 3;
  for (int it = 0; it < ntrip; ++it) {
    const int t = 3 * it;
    stage(t, 0, 1, 2, raw[0], raw[1]);
    stage(t + 1, 1, 2, 0, raw[1], raw[2]);
    stage(t + 2, 2, 0, 1, raw[2], raw[0]);
  }
  {
    const int t = 3 * ntrip;   // nk % 3 stages are left (block-uniform)
    if (t < nk) stage(t, 0, 1, 2, raw[0], raw[1]);
    if (t + 1 < nk) stage(t + 1, 1, 2, 0, raw[1], raw[2]);
  }
  // the requests past the last stage land in registers the epilogue is about to reuse: drain them first
  ar_wait_vm<0>(raw[0][0].v, raw[0][1].v, raw[0][2].v, raw[0][3].v);
  ar_wait_vm<0>(raw[1][0].v, raw[1][1].v, raw[1][2].v, raw[1][3].v);
  ar_wait_vm<0>(raw[2][0].v, raw[2][1].v, raw[2][2].v, raw[2][3].v);
  gemm_epilogue<1, TN, 32, BNT, BM, BNT, WR, false, AFFINE, RESID>(acc, ap, ti, ent, epi, n0, n_out, ldo, wr, 0, lane);
}

static unsigned long long *g_ar_stamps = nullptr;   // development probe: see gnnsaft_debug_ar_stamps

template <int WAVES, int TN, int MINW, class AProv, bool AFFINE, bool RESID, int TOWERS = 1>
static int launch_ar_one(const AProv &ap, int nbatch, const GemmBatch &b, int n_pad, int64_t ldo, int64_t m, int n_out, int k,
                         const EpiArgs &ea, int64_t grid_x, hipStream_t stream) {
  constexpr int BM = 32 * WAVES / TOWERS, BN = 32 * TN;
  constexpr size_t kLds = (size_t)3 * 3 * BN * TOWERS * kW3RowBytes;
  static_assert(kLds * (WAVES <= 4 ? 2 : 1) <= 160 * 1024, "two four-wave workgroups share the CU's LDS");
  auto kern = k_gemm_ar<WAVES, TN, MINW, AProv, AFFINE, RESID, TOWERS>;
  static std::atomic<unsigned long long> raised{0ull};
  if (kLds > 64 * 1024) GS_HIP(gs_raise_dynamic_lds(reinterpret_cast<const void *>(kern), kLds, raised));
  if (TOWERS == 2 && nbatch != 2) return GNNSAFT_ERR_SHAPE;
  const dim3 grid((unsigned)(grid_x > 0 ? grid_x : gs_ceil_div(m, BM)), (unsigned)gs_ceil_div(n_out, BN),
                  (unsigned)(TOWERS == 2 ? 1 : nbatch));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES), kLds, stream, ap, b, n_pad, ldo, n_out, k, ea, g_ar_stamps);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

// cfg: kAr_128x128 (4 waves, two workgroups per CU)
template <class AProv, bool AFFINE, bool RESID>
static int launch_ar_cfg(int cfg, const AProv &ap, int nbatch, const GemmBatch &b, int n_pad, int64_t ldo, int64_t m, int n_out,
                         int k, const EpiArgs &ea, int64_t grid_x, hipStream_t stream) {
  switch (cfg) {
    case kAr_128x128: return launch_ar_one<4, 4, 2, AProv, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    case kAr_128x64: return launch_ar_one<4, 2, 2, AProv, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    case kAr_64x128: return launch_ar_one<2, 4, 2, AProv, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    case kAr_64x64: return launch_ar_one<2, 2, 2, AProv, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    case kAr_256x128: return launch_ar_one<8, 4, 2, AProv, AFFINE, RESID>(ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, grid_x, stream);
    default: return GNNSAFT_ERR_UNSUPPORTED;
  }
}

// out = epilogue(a W^T + b), every entry's weights given as a W3 image (entries[i].w3; n_pad rows per plane)
int launch_linear_ar(const float *a, int64_t lda, int nbatch, const GemmBatchEntry *entries, int n_pad, int64_t ldo, int64_t m,
                     int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg) {
  GS_REQUIRE(a != nullptr && entries != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(nbatch >= 1 && nbatch <= kMaxGemmBatch && cfg >= 0 && cfg < kNumArCfg, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(m >= 0 && n_out >= 1 && k >= kW3Kt && (k % kW3Kt) == 0 && (lda % 4) == 0 && n_pad >= n_out &&
                 (reinterpret_cast<uintptr_t>(a) & 15) == 0,
             GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(ldo >= 0 && epi.ldr >= 0 && (m + 1) * (ldo > epi.ldr ? ldo : epi.ldr) < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(epi.stats == nullptr, GNNSAFT_ERR_UNSUPPORTED);
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
  // (plain epilogue only: the kernel is an opt-in experiment behind gnnsaft_debug_linear_ar, nothing in the forward
  //  launches it -- the affine / residual epilogues of gemm_epi.hpp would instantiate, but are not built untested)
  GS_REQUIRE(epi.scale == nullptr && epi.residual == nullptr && epi.relu_out == 0, GNNSAFT_ERR_UNSUPPORTED);
  PlainA ap{a, lda, m, k};
  return launch_ar_cfg<PlainA, false, false>(cfg, ap, nbatch, b, n_pad, ldo, m, n_out, k, ea, 0, stream);
}

// ---- the degree-folded PNA update (launch_pna_update_folded) on k_gemm_ar, both towers in ONE workgroup per degree
// tile: half the workgroup's waves per tower, every wave 32 rows x F/2 columns with its own A stream [x | A_t] in
// registers, the two towers' weight-image stages side by side in the LDS ring.  For the small batches the fused
// aggregation + update does not pay for (below 64 k nodes): hidden 128 (64-row tiles, 4 waves, two workgroups per CU)
// and hidden 256 (128-row tiles, 8 waves).
bool ar_update_supported(int hidden) {
  if (!gemm_w3_enabled()) return false;
  const int rows = pna_fold_tile_rows(hidden);
  return (hidden == 128 && rows == 64) || (hidden == 256 && rows == 128);
}

int launch_pna_update_folded_ar(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const char *w_eff3,
                                const float *b_post0, const float *b_post1, float *u, hipStream_t stream) {
  GS_REQUIRE(x && agg && perm && tiles && num_tiles && w_eff3 && u, GNNSAFT_ERR_NULL);
  GS_REQUIRE(ar_update_supported(hidden), GNNSAFT_ERR_UNSUPPORTED);
  if (n == 0) return GNNSAFT_OK;
  GS_REQUIRE((n + 1) * (int64_t)hidden < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);   // 32-bit epilogue offsets
  const int64_t per_tower = (int64_t)(hidden / 2) * 5 * hidden;
  PostFoldA ap{x, agg, perm, tiles, num_tiles, 2 * per_tower, hidden};
  GemmBatch b;
  b.e[0] = GemmBatchEntry{nullptr, b_post0, u, 0, w_eff3};
  b.e[1] = GemmBatchEntry{nullptr, b_post1, u + hidden / 2, 4 * (int64_t)hidden, w_eff3 + per_tower * 6};
  for (int i = 2; i < kMaxGemmBatch; ++i) b.e[i] = b.e[0];
  EpiArgs ea{nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, 0.f};
  if (hidden == 128)
    return launch_ar_one<4, 2, 2, PostFoldA, false, false, 2>(ap, 2, b, hidden / 2, hidden, n, hidden / 2, 5 * hidden, ea,
                                                              max_tiles, stream);
  return launch_ar_one<8, 4, 2, PostFoldA, false, false, 2>(ap, 2, b, hidden / 2, hidden, n, hidden / 2, 5 * hidden, ea,
                                                            max_tiles, stream);
}

}  // namespace gs

// development probe (tools/ar_stamps.py; not part of the product path): a device buffer of 256 uint64 that the next
// launches of k_gemm_ar fill with s_memtime stamps of one tile's first wave; NULL switches it off
extern "C" int gnnsaft_debug_ar_stamps(void *device_buffer) {
  gs::g_ar_stamps = static_cast<unsigned long long *>(device_buffer);
  return GNNSAFT_OK;
}

// stage-test entry point (include/gnnsaft.h): the degree-folded update through k_gemm_ar, both towers per workgroup;
// operands as gnnsaft_pna_update_folded takes them, the folded weights as W3 images (as gnnsaft_pna_update_agg)
extern "C" int gnnsaft_pna_update_folded_ar(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                            const int32_t *num_tiles, int64_t num_nodes, int32_t hidden,
                                            const void *w_eff_images, const float *b_post0, const float *b_post1, float *u,
                                            gnnsaft_stream_t stream) {
  return gs::launch_pna_update_folded_ar(x, agg, perm, tiles, num_tiles, gnnsaft_degree_tiles_capacity(num_nodes, hidden),
                                         num_nodes, hidden, static_cast<const char *>(w_eff_images), b_post0, b_post1, u,
                                         static_cast<hipStream_t>(stream));
}

// ---- C ABI: stage tests and tuning (include/gnnsaft.h)
extern "C" int gnnsaft_debug_linear_ar(const float *a, int64_t lda, const void *w_image, const float *bias, float *out,
                                       int64_t ldo, int64_t m, int32_t n_out, int32_t k, int32_t tile_config,
                                       gnnsaft_stream_t stream) {
  gs::GemmBatchEntry ent{nullptr, bias, out, 0, static_cast<const char *>(w_image)};
  gs::LinearEpilogue epi;
  return gs::launch_linear_ar(a, lda, 1, &ent, n_out, ldo, m, n_out, k, epi, static_cast<hipStream_t>(stream), tile_config);
}
