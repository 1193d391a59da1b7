// Readout MLP of the PNAPCSAFT forward in ONE launch (after k_add_pool): [Linear -> BatchNorm1d -> ReLU] x (m + 2) ->
// Linear(H/4, P) -> MAPE  (/root/reference/gnnepcsaft/train/models.py:84-103,133-134,191-194).
//
// The per-op form (4 GEMM launches, 3 BatchNorm launches, k_mape after k_add_pool) is 8 dependent launches of ~5 us each
// for < 0.1 % of the step's FLOPs: 58 us of a 467 us step at BASELINE config 2.  Here a workgroup owns 64 graphs
// (rows): it loads their pooled rows, keeps the activations of all layers in LDS, runs the four small GEMMs on the matrix cores
// (v_mfma_f32_32x32x2_f32, weights straight from L2 -- they total < 100 KB -- in the k order of gemm.hip), and writes
// the outputs.  Train-mode BatchNorm needs column statistics over ALL graphs: every workgroup publishes the (mean, M2)
// of its 64 rows, the grid meets at a device-scope barrier (release / acquire atomics on a counter zeroed by the
// forward's prologue kernel), and every workgroup folds the <= 256 partials itself (plain f64 sums S1, S2 of
// bn_fold.hpp -- no pivot: see its header for the error budget).  The barrier is only safe while all workgroups are co-resident: the launcher uses this kernel for up
// to as many workgroups as the device's occupancy calculator says are co-resident (readout_resident_workgroups:
// hipOccupancyMaxActiveBlocksPerMultiprocessor x CU count, per device) and the per-op path beyond.  What the calculator
// cannot see -- another process on the GPU, a CU mask -- ends in the bounded spin: the kernel then raises
// GNNSAFT_FLAG_BARRIER_TIMEOUT AND writes NaN to its outputs (forward: out / loss3; backward: the pooled-row
// gradient, which poisons every gradient below it), so that a caller who never reads the flag word cannot train on
// incomplete BatchNorm statistics.  The MAPE sum is closed by the last workgroup to arrive (an
// atomic ticket), which adds the per-workgroup partials in workgroup order: deterministic, no float atomics.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "plan.hpp"
#include "readout.hpp"

namespace gs {

#ifdef GS_GF_TIMING   // development probe (make timing): wall-clock (100 MHz) stamps of workgroup 0
__device__ long long g_rd_stamp[64];
#define RD_STAMP(i)                                                                        \
  do {                                                                                     \
    __syncthreads();                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0 && (i) < 64) g_rd_stamp[(i)] = wall_clock64(); \
  } while (0)
#else
#define RD_STAMP(i) do {} while (0)
#endif

constexpr int kRdRows = 64;      // graphs per workgroup
constexpr int kRdThreads = 512;  // 8 waves: one 32 x 32 output tile each at n_out = 128
constexpr int kRdPad = 4;    // floats of row padding in the LDS tiles

struct ReadoutArgs {
  const float *x;            // [N, H] node state after the last layer
  const int32_t *graph_ptr;  // [G + 1]
  int64_t g, n;
  int h, num_para, nblocks;  // nblocks BatchNorm blocks (m + 2), then the final Linear
  int training;
  float momentum, eps;
  const float *w[kRdMaxBlocks + 1], *b[kRdMaxBlocks + 1];
  const float *gamma[kRdMaxBlocks], *beta[kRdMaxBlocks];
  float *rmean[kRdMaxBlocks], *rvar[kRdMaxBlocks];
  int64_t *nbt[kRdMaxBlocks];
  int n_in[kRdMaxBlocks + 1], n_out[kRdMaxBlocks + 1];
  const float *target;       // [G, P] or null
  float *out;                // [G, P]
  float *loss3;              // [3] or null
  float *pooled, *ry, *ro, *rstat;   // tape: [G,H], [nb][G,H], [nb][G,H], [nb][2][H]   (ry / ro row stride H)
  float *part;               // [nb][W][2][H] per-workgroup (mean, M2)
  float *mape_part;          // [W]
  int32_t *sync;             // [nb + 1] counters, zero at launch; sync[kRdSyncInts - 1]: "this call lost a barrier"
  int32_t *err;
  const int32_t *k0_lost;    // or null: the structure chain of this call lost a barrier (forward.hip)
  int barrier_extra;         // 0; > 0 (test hook): the barriers expect that many arrivals more than there are workgroups
  float dropout_p;           // 0: no dropout (or eval mode)
  uint64_t dropout_seed;
  const uint64_t *dropout_step;   // or null: device word mixed into the key when the kernel runs (hipGraph replays)
};

// all workgroups of the grid meet here; returns true after every one of them has arrived, false when the spin bound
// was hit (the flag word is raised; the caller poisons its outputs)
__device__ __forceinline__ bool grid_barrier(int32_t *counter, int expected, int32_t *err, int32_t *lost_word) {
  __shared__ int s_barrier_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    int ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expected) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1L << 21)) {  // ~0.5 s: a workgroup never became resident; give up loudly instead of hanging
        if (err) atomicOr(err, GNNSAFT_FLAG_BARRIER_TIMEOUT);
        __hip_atomic_store(lost_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    s_barrier_ok = ok;
  }
  __syncthreads();
  return s_barrier_ok != 0;
}

// has THIS call lost a barrier anywhere?  `lost_word`: the last of the kernel's sync words (zero at launch), set by
// every workgroup that gave up -- one that arrived late and found the counters complete still learns that the others
// computed with incomplete statistics; `k0_lost` (forward, or null): the structure chain in front of this kernel.
// Per call: the sticky flag word is for the host, a later call is not poisoned by it.
constexpr int kRdLostWord = kRdSyncInts - 1;
__device__ __forceinline__ bool barrier_lost(bool all_ok, const int32_t *lost_word, const int32_t *k0_lost = nullptr) {
  return !all_ok || __hip_atomic_load(lost_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
         (k0_lost != nullptr && k0_lost[0] != 0);
}

// y[64][n_out] = a[64][n_in] (LDS) x W^T + bias : 32 x 32 output tiles round-robin over the 8 waves, k in the order
// of gemm.hip (8 k per step, lane half h takes k0 + 4h .. k0 + 4h + 3).  The weight fragments come straight from L2;
// a wave issues the loads of a whole 128-wide K chunk (16 float4 per lane) before its first MFMA -- one exposed
// round trip per chunk instead of one per step (measured: 84 us -> for the whole readout with a 1-deep prefetch).
__device__ __forceinline__ void rd_gemm(const float *a, int lda, const float *__restrict__ w, const float *__restrict__ bias,
                                        int n_in, int n_out, float *y, int ldy) {
  constexpr int kChunk = 128, kSteps = kChunk / 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int col_tiles = (n_out + 31) >> 5;
  for (int t = wave; t < 2 * col_tiles; t += kRdThreads / 64) {
    const int rt = t & 1, ct = t >> 1;
    const int col = ct * 32 + l31;
    const int colc = col < n_out ? col : n_out - 1;
    const float *ap = a + (rt * 32 + l31) * lda + 4 * half;
    const float *wp = w + (int64_t)colc * n_in + 4 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < n_in; kc += kChunk) {
      f32x4 bf[kSteps];
#pragma unroll
      for (int i = 0; i < kSteps; ++i) {
        const int k0 = kc + 8 * i;
        bf[i] = gs_ld4(wp + (k0 < n_in ? k0 : 0));   // past the end: any valid address, the step is skipped below
      }
#pragma unroll
      for (int i = 0; i < kSteps; ++i) {
        const int k0 = kc + 8 * i;
        if (k0 < n_in) {                              // wave-uniform
          const f32x4 af = gs_ld4(ap + k0);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[i][s], acc, 0, 0, 0);
        }
      }
    }
    const float bv = bias != nullptr ? bias[colc] : 0.f;
    if (col < n_out) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        y[row * ldy + col] = acc[r] + bv;
      }
    }
  }
}

__global__ __launch_bounds__(kRdThreads) void k_readout_fused(ReadoutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rd_lds[];
  const int tid = threadIdx.x;
  const int h = a.h, ld = h + kRdPad;
  float *at = rd_lds;                 // [64][ld] activations (input of the current GEMM)
  float *yt = at + kRdRows * ld;      // [64][ld] pre-BatchNorm output
  float *s_scale = yt + kRdRows * ld; // [H]
  float *s_shift = s_scale + h;       // [H]
  float *s_red = s_shift + h;         // [8]
  double *s_d = reinterpret_cast<double *>(s_red + 8);   // [2][kRdThreads] doubles | [2][kRdThreads] floats + [H] pivots
  float *s_f = reinterpret_cast<float *>(s_d);
  const int64_t row0 = (int64_t)blockIdx.x * kRdRows;
  const int rows = a.g - row0 < kRdRows ? (int)(a.g - row0) : kRdRows;   // >= 1
  const int nwg = gridDim.x;
  bool barriers_ok = true;

  // ---- the pooled rows of this workgroup's graphs (k_add_pool ran as its own launch: one thread per (graph, float4)
  //      keeps ~1e5 loads in flight over the whole chip; pooling here, 64 graphs per workgroup, was measured at 20 us
  //      of pure load latency)
  for (int idx = tid; idx < kRdRows * (h / 4); idx += kRdThreads) {
    const int r = idx / (h / 4), c = (idx - r * (h / 4)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) v = gs_ld4(a.pooled + (row0 + r) * h + c);
    gs_st4(at + r * ld + c, v);
  }
  __syncthreads();

  const int64_t rs = a.g * (int64_t)h;  // floats per tape block buffer
  RD_STAMP(0);
  for (int b = 0; b < a.nblocks; ++b) {
    const int n_in = a.n_in[b], n_out = a.n_out[b];
    rd_gemm(at, ld, a.w[b], a.b[b], n_in, n_out, yt, ld);
    __syncthreads();
    RD_STAMP(1 + 6 * b);
    const int q = n_out >> 2;           // float4 per row (n_out is a multiple of 4: H, H/2, H/4 with H % 32 == 0)
    if (a.ry != nullptr)
      for (int idx = tid; idx < rows * q; idx += kRdThreads) {
        const int r = idx / q, c = (idx - r * q) * 4;
        gs_st4(a.ry + b * rs + (row0 + r) * n_out + c, gs_ld4(yt + r * ld + c));
      }
    // columns x row-slices: thread (c, part) of np2 x tpc
    int np2 = 8;
    while (np2 < n_out) np2 <<= 1;
    const int tpc = kRdThreads / np2;
    const int c = tid & (np2 - 1), part = tid / np2;
    const bool col_ok = c < n_out;
    if (a.training) {
      // (mean, M2) of this workgroup's rows per column: one pass over d = y - y[row 0] (no cancellation), the
      // row slices meet in LDS
      {
        const float p0 = col_ok ? yt[c] : 0.f;
        float sd = 0.f, sd2 = 0.f;
        if (col_ok)
          for (int r = part; r < rows; r += tpc) {
            const float d = yt[r * ld + c] - p0;
            sd += d;
            sd2 += d * d;
          }
        s_f[tid] = sd;
        s_f[kRdThreads + tid] = sd2;
        __syncthreads();
        if (part == 0 && col_ok) {
          for (int o = 1; o < tpc; ++o) {
            sd += s_f[o * np2 + c];
            sd2 += s_f[kRdThreads + o * np2 + c];
          }
          const float n_w = (float)rows;
          float *p = a.part + ((int64_t)b * nwg + blockIdx.x) * 2 * h;
          const float mean_w = p0 + sd / n_w;
          p[c] = mean_w;
          p[h + c] = fmaxf(sd2 - sd * sd / n_w, 0.f);
          s_f[2 * kRdThreads + c] = mean_w;   // pivot of this workgroup's fold (behind the two reduction arrays)
        }
      }
      RD_STAMP(2 + 6 * b);
      barriers_ok &= grid_barrier(a.sync + b, nwg + a.barrier_extra, a.err, a.sync + kRdLostWord);
      RD_STAMP(3 + 6 * b);
      {
        // every workgroup folds all partials: S1 = sum n_w (mean_w - K), S2 = sum (M2_w + n_w (mean_w - K)^2) in f64
        // around the pivot K = mean of workgroup 0 (bn_train.hip's scheme); partial slices per thread, LDS combine
        // The pivot is this workgroup's OWN column mean (already here: no extra cross-XCD round trip, ~2 us each);
        // any value near the mean serves, the f64 sums differ between workgroups by ~1e-16 relative.
        const float *p = a.part + (int64_t)b * nwg * 2 * h;
        const int cc = col_ok ? c : 0;
        const double pivot = (double)s_f[2 * kRdThreads + cc];
        double s1 = 0.0, s2 = 0.0;
        for (int w0 = part; w0 < nwg; w0 += 8 * tpc) {
          float gm[8], g2[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int w = w0 + u * tpc < nwg ? w0 + u * tpc : nwg - 1;
            gm[u] = __builtin_nontemporal_load(p + (int64_t)w * 2 * h + cc);
            g2[u] = __builtin_nontemporal_load(p + (int64_t)w * 2 * h + h + cc);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int w = w0 + u * tpc;
            if (w < nwg) {
              const int64_t left = a.g - (int64_t)w * kRdRows;
              const double gn = (double)(left < kRdRows ? left : kRdRows);
              const double m = (double)gm[u] - pivot;
              s1 += gn * m;
              s2 += (double)g2[u] + gn * m * m;
            }
          }
        }
        __syncthreads();   // s_f (aliases s_d) was read above
        s_d[tid] = s1;
        s_d[kRdThreads + tid] = s2;
        __syncthreads();
        if (part == 0 && col_ok) {
          for (int o = 1; o < tpc; ++o) {
            s1 += s_d[o * np2 + c];
            s2 += s_d[kRdThreads + o * np2 + c];
          }
          const double nn = (double)a.g;
          const double dmean = s1 / nn;
          const double mean = pivot + dmean;
          double m2 = s2 - nn * dmean * dmean;
          m2 = m2 > 0.0 ? m2 : 0.0;
          const float mean_f = (float)mean;
          const float var_f = (float)(m2 / nn);
          const float rstd = 1.f / sqrtf(var_f + a.eps);
          const float sc = rstd * a.gamma[b][c];
          s_scale[c] = sc;
          s_shift[c] = a.beta[b][c] - mean_f * sc;
          if (blockIdx.x == 0) {
            if (a.rstat != nullptr) {
              a.rstat[(int64_t)b * 2 * h + c] = mean_f;           // layout of k_bn_train_apply: [mean | rstd] of n_out
              a.rstat[(int64_t)b * 2 * h + n_out + c] = rstd;
            }
            const float unbiased = (float)(nn > 1.0 ? m2 / (nn - 1.0) : m2);
            a.rmean[b][c] = (1.f - a.momentum) * a.rmean[b][c] + a.momentum * mean_f;
            a.rvar[b][c] = (1.f - a.momentum) * a.rvar[b][c] + a.momentum * unbiased;
            if (c == 0 && a.nbt[b] != nullptr) a.nbt[b][0] += 1;
          }
        }
      }
    } else if (part == 0 && col_ok) {
      const float rstd = 1.f / sqrtf(a.rvar[b][c] + a.eps);
      const float sc = rstd * a.gamma[b][c];
      s_scale[c] = sc;
      s_shift[c] = a.beta[b][c] - a.rmean[b][c] * sc;
    }
    __syncthreads();
    RD_STAMP(4 + 6 * b);
    for (int idx = tid; idx < kRdRows * q; idx += kRdThreads) {
      const int r = idx / q, c4 = (idx - r * q) * 4;
      const f32x4 sc = gs_ld4(s_scale + c4), sh = gs_ld4(s_shift + c4);
      f32x4 v = gs_ld4(yt + r * ld + c4) * sc + sh;
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
      if (a.dropout_p > 0.f) v = v * dropout_scale4(rd_key(a.dropout_seed, a.dropout_step), row0 + r, b, c4, a.dropout_p);   // block-uniform
      gs_st4(at + r * ld + c4, v);
      if (a.ro != nullptr && r < rows) gs_st4(a.ro + b * rs + (row0 + r) * n_out + c4, v);
    }
    __syncthreads();
    RD_STAMP(5 + 6 * b);
  }
  // ---- final Linear(H/4, P)
  {
    const int b = a.nblocks;
    rd_gemm(at, ld, a.w[b], a.b[b], a.n_in[b], a.num_para, yt, ld);
    __syncthreads();
    float ape = 0.f;
    const bool lost = barrier_lost(barriers_ok, a.sync + kRdLostWord, a.k0_lost);   // block-uniform; every mode
    for (int idx = tid; idx < rows * a.num_para; idx += kRdThreads) {
      const int r = idx / a.num_para, c = idx - r * a.num_para;
      const float v = lost ? __builtin_nanf("") : yt[r * ld + c];
      a.out[(row0 + r) * a.num_para + c] = v;
      if (a.target != nullptr) {
        const float t = a.target[(row0 + r) * a.num_para + c];
        ape += fabsf(v - t) / fmaxf(fabsf(t), 1.17e-06f);
      }
    }
    if (a.target != nullptr && a.loss3 != nullptr) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ape += __shfl_xor(ape, o);
      if ((tid & 63) == 0) s_red[tid >> 6] = ape;
      __syncthreads();
      if (tid == 0) {
        float wsum = 0.f;
        for (int w = 0; w < kRdThreads / 64; ++w) wsum += s_red[w];
        a.mape_part[blockIdx.x] = wsum;
        const int ticket = __hip_atomic_fetch_add(a.sync + a.nblocks, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == nwg - 1) {  // last workgroup: add the partials in workgroup order
          float tot = 0.f;
          for (int w = 0; w < nwg; ++w) tot += __builtin_nontemporal_load(a.mape_part + w);
          const float cnt = (float)(a.g * a.num_para);
          if (barrier_lost(barriers_ok, a.sync + kRdLostWord, a.k0_lost)) tot = __builtin_nanf("");
          a.loss3[0] = tot / cnt;
          a.loss3[1] = tot;
          a.loss3[2] = cnt;
        }
      }
    }
  }
  RD_STAMP(60);
}

// --------------------------------------------------------------------------------------------------------------
// Backward of the readout in ONE launch (+ the batched slab reduction of backward.hip): what 23 launches did over
// [G, H] matrices -- per block BatchNorm+ReLU backward (2), weight-gradient TN GEMM + slab sum (2), input-gradient
// GEMM (1), plus the final Linear's -- for < 1 % of the backward's FLOPs and ~10 % of its time at BASELINE config 2.
// Same decomposition as the forward kernel: a workgroup owns 64 graphs, keeps the gradient tile in LDS through all
// blocks, meets the grid once per BatchNorm block (the column sums of dz and dz*yhat run over ALL graphs) and leaves
// per-workgroup partial weight gradients [W][n_out][n_in] for the reduction launch (fixed order: reproducible).
struct ReadoutBwdArgs {
  int64_t g;
  int h, num_para, nblocks;
  const float *grad_out;                       // [G, P]
  const float *w_final;                        // [P, H/4]
  const float *wt[kRdMaxBlocks];               // transposed block weights [n_in][n_out]
  const float *gamma[kRdMaxBlocks], *beta[kRdMaxBlocks];
  int n_in[kRdMaxBlocks + 1], n_out[kRdMaxBlocks + 1];
  const float *pooled, *ry, *ro, *rstat;       // tape of the forward
  float *dgamma[kRdMaxBlocks], *dbeta[kRdMaxBlocks];
  float *slab_w[kRdMaxBlocks + 1];             // [W][n_out][n_in] per layer (final Linear last)
  float *bias_part;                            // [W][8] column sums of dOut per workgroup
  float *dbias_final;                          // [P]
  float *part;                                 // [nb][W][2][H] per-workgroup (sum dz, sum dz yhat)
  float *dpooled;                              // [G, H]
  int32_t *sync;                               // [nb + 1] counters, zero at launch
  int32_t *err;
  int barrier_extra;
  float dropout_p;
  uint64_t dropout_seed;
  const uint64_t *dropout_step;
};

// slab[n][k] = sum over this workgroup's 64 rows of dy[r][n] * in[r][k]: 32 x 32 output tiles round-robin over the
// waves, contraction in row pairs (lane half = row parity) exactly as k_gemm_tn; dy from the LDS tile, `in` rows
// straight from L2 (the forward's tape).
__device__ __forceinline__ void rd_wgrad(const float *dy, int ld, const float *__restrict__ in, int64_t row0, int rows,
                                         int n_out, int n_in, float *__restrict__ slab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nt_n = (n_out + 31) >> 5, nt_k = (n_in + 31) >> 5;
  for (int t = wave; t < nt_n * nt_k; t += kRdThreads / 64) {
    const int nt = t / nt_k, kt = t - nt * nt_k;
    const int ncol = nt * 32 + l31, kcol = kt * 32 + l31;
    const bool n_ok = ncol < n_out, k_ok = kcol < n_in;
    const float *bp = in + (k_ok ? kcol : 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int q0 = 0; q0 < kRdRows / 2; q0 += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = 2 * (q0 + u) + half;
        const int rc = row < rows ? row : rows - 1;       // rows past the end: dy is zero there, any valid address
        av[u] = dy[row * ld + (n_ok ? ncol : 0)];
        bv[u] = bp[(row0 + rc) * n_in];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(n_ok ? av[u] : 0.f, k_ok ? bv[u] : 0.f, acc, 0, 0, 0);
    }
    if (k_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nr = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (nr < n_out) slab[(int64_t)nr * n_in + kcol] = acc[r];
      }
    }
  }
}

__global__ __launch_bounds__(kRdThreads) void k_readout_bwd_fused(ReadoutBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rd_lds[];
  const int tid = threadIdx.x;
  const int h = a.h, ld = h + kRdPad;
  float *dt = rd_lds;                  // [64][ld] gradient w.r.t. the current block's output, then its input
  float *yt = dt + kRdRows * ld;       // [64][ld] yhat, then dy
  float *s_a = yt + kRdRows * ld;      // [H] gamma rstd
  float *s_b = s_a + h;                // [H] gamma rstd mean(dz)
  float *s_c = s_b + h;                // [H] gamma rstd mean(dz yhat)
  float *s_do = s_c + h;               // [64][8] dOut rows
  double *s_d = reinterpret_cast<double *>(s_do + kRdRows * 8);   // [2][kRdThreads]
  const int64_t row0 = (int64_t)blockIdx.x * kRdRows;
  const int rows = a.g - row0 < kRdRows ? (int)(a.g - row0) : kRdRows;   // >= 1
  const int nwg = gridDim.x, nb = a.nblocks, P = a.num_para;
  const int64_t rs = a.g * (int64_t)h;
  bool barriers_ok = true;

  // ---- final Linear(H/4 -> P): dW_f = dOut^T x, db_f = column sums of dOut, dx = dOut W_f
  {
    const int n_in = a.n_in[nb];
    for (int idx = tid; idx < kRdRows * 8; idx += kRdThreads) {
      const int r = idx >> 3, c = idx & 7;
      s_do[idx] = (r < rows && c < P) ? a.grad_out[(row0 + r) * P + c] : 0.f;
    }
    __syncthreads();
    const float *in = a.ro + (int64_t)(nb - 1) * rs;     // output of the last BatchNorm block, row stride n_in
    // staged through LDS: 64 dependent-address row loads per thread straight from L2 were most of this phase
    for (int idx = tid; idx < kRdRows * (n_in / 4); idx += kRdThreads) {
      const int r = idx / (n_in / 4), c4 = (idx - r * (n_in / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows) v = gs_ld4(in + (row0 + r) * n_in + c4);
      gs_st4(yt + r * ld + c4, v);
    }
    __syncthreads();
    float *slab = a.slab_w[nb] + (int64_t)blockIdx.x * P * n_in;
    for (int idx = tid; idx < P * n_in; idx += kRdThreads) {
      const int pp = idx / n_in, j = idx - pp * n_in;
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s += s_do[r * 8 + pp] * yt[r * ld + j];
      slab[idx] = s;
    }
    if (tid < 8) {
      float s = 0.f;
      for (int r = 0; r < rows; ++r) s += s_do[r * 8 + tid];
      s_a[tid] = s;     // published by thread 0 below (one writer in front of the ticket)
    }
    for (int idx = tid; idx < kRdRows * n_in; idx += kRdThreads) {
      const int r = idx / n_in, j = idx - r * n_in;
      float s = 0.f;
      for (int pp = 0; pp < P; ++pp) s += s_do[r * 8 + pp] * a.w_final[pp * n_in + j];
      dt[r * ld + j] = s;                                  // rows past the end: dOut is zero there
    }
    __syncthreads();
    if (tid == 0) {   // the last workgroup to arrive adds the bias partials in workgroup order
      for (int pp = 0; pp < 8; ++pp) a.bias_part[(int64_t)blockIdx.x * 8 + pp] = s_a[pp];
      const int ticket = __hip_atomic_fetch_add(a.sync + nb, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == nwg - 1)
        for (int pp = 0; pp < P; ++pp) {
          float tot = 0.f;
          for (int w = 0; w < nwg; ++w) tot += __builtin_nontemporal_load(a.bias_part + (int64_t)w * 8 + pp);
          a.dbias_final[pp] = tot;
        }
    }
  }

  for (int b = nb - 1; b >= 0; --b) {
    const int n_in = a.n_in[b], n_out = a.n_out[b];
    const int q = n_out >> 2;
    const float *stat = a.rstat + (int64_t)b * 2 * h;
    // ---- yhat and dz = dOut masked by the ReLU (z = yhat gamma + beta > 0), in place
    for (int idx = tid; idx < kRdRows * q; idx += kRdThreads) {
      const int r = idx / q, c4 = (idx - r * q) * 4;
      f32x4 yh = {0.f, 0.f, 0.f, 0.f}, dz = {0.f, 0.f, 0.f, 0.f};
      if (r < rows) {
        const f32x4 y = gs_ld4(a.ry + b * rs + (row0 + r) * n_out + c4);
        const f32x4 mean = gs_ld4(stat + c4), rstd = gs_ld4(stat + n_out + c4);
        const f32x4 gm = gs_ld4(a.gamma[b] + c4), bt = gs_ld4(a.beta[b] + c4);
        f32x4 dv = gs_ld4(dt + r * ld + c4);
        if (a.dropout_p > 0.f) dv = dv * dropout_scale4(rd_key(a.dropout_seed, a.dropout_step), row0 + r, b, c4, a.dropout_p);   // the forward's mask
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          yh[j] = (y[j] - mean[j]) * rstd[j];
          const float sc = rstd[j] * gm[j], sh = bt[j] - mean[j] * sc;   // the gate as the forward took it
          dz[j] = (y[j] * sc + sh) > 0.f ? dv[j] : 0.f;
        }
      }
      gs_st4(yt + r * ld + c4, yh);
      gs_st4(dt + r * ld + c4, dz);
    }
    __syncthreads();
    // ---- per-workgroup column sums, published for the grid
    int np2 = 8;
    while (np2 < n_out) np2 <<= 1;
    const int tpc = kRdThreads / np2;
    const int c = tid & (np2 - 1), part = tid / np2;
    const bool col_ok = c < n_out;
    {
      double s1 = 0.0, s2 = 0.0;
      if (col_ok)
        for (int r = part; r < rows; r += tpc) {
          const float dz = dt[r * ld + c];
          s1 += (double)dz;
          s2 += (double)dz * (double)yt[r * ld + c];
        }
      s_d[tid] = s1;
      s_d[kRdThreads + tid] = s2;
      __syncthreads();
      if (part == 0 && col_ok) {
        for (int o = 1; o < tpc; ++o) {
          s1 += s_d[o * np2 + c];
          s2 += s_d[kRdThreads + o * np2 + c];
        }
        float *pp = a.part + ((int64_t)b * nwg + blockIdx.x) * 2 * h;
        pp[c] = (float)s1;
        pp[h + c] = (float)s2;
      }
    }
    barriers_ok &= grid_barrier(a.sync + b, nwg + a.barrier_extra, a.err, a.sync + kRdLostWord);
    {
      const float *pp = a.part + (int64_t)b * nwg * 2 * h;
      const int cc = col_ok ? c : 0;
      double s1 = 0.0, s2 = 0.0;
      for (int w0 = part; w0 < nwg; w0 += 8 * tpc) {
        float g1[8], g2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int w = w0 + u * tpc < nwg ? w0 + u * tpc : nwg - 1;
          g1[u] = __builtin_nontemporal_load(pp + (int64_t)w * 2 * h + cc);
          g2[u] = __builtin_nontemporal_load(pp + (int64_t)w * 2 * h + h + cc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (w0 + u * tpc < nwg) {
            s1 += (double)g1[u];
            s2 += (double)g2[u];
          }
      }
      __syncthreads();
      s_d[tid] = s1;
      s_d[kRdThreads + tid] = s2;
      __syncthreads();
      if (part == 0 && col_ok) {
        for (int o = 1; o < tpc; ++o) {
          s1 += s_d[o * np2 + c];
          s2 += s_d[kRdThreads + o * np2 + c];
        }
        // dy = gamma rstd (dz - s1/G - yhat s2/G) = a dz - b - c yhat
        const float gr = a.gamma[b][c] * stat[n_out + c];
        s_a[c] = gr;
        s_b[c] = gr * (float)(s1 / (double)a.g);
        s_c[c] = gr * (float)(s2 / (double)a.g);
        if (blockIdx.x == 0) {
          if (a.dgamma[b] != nullptr) a.dgamma[b][c] = (float)s2;
          if (a.dbeta[b] != nullptr) a.dbeta[b][c] = (float)s1;
        }
      }
    }
    __syncthreads();
    // ---- dy (rows past the end stay zero: they must not reach the weight gradient)
    for (int idx = tid; idx < kRdRows * q; idx += kRdThreads) {
      const int r = idx / q, c4 = (idx - r * q) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows) {
        const f32x4 dz = gs_ld4(dt + r * ld + c4), yh = gs_ld4(yt + r * ld + c4);
        v = gs_ld4(s_a + c4) * dz - gs_ld4(s_b + c4) - gs_ld4(s_c + c4) * yh;
      }
      gs_st4(yt + r * ld + c4, v);
    }
    __syncthreads();
    // ---- dW_b partial = dy^T in_b ; dIn = dy W_b
    const float *in = b == 0 ? a.pooled : a.ro + (int64_t)(b - 1) * rs;
    rd_wgrad(yt, ld, in, row0, rows, n_out, n_in, a.slab_w[b] + (int64_t)blockIdx.x * n_out * n_in);
    rd_gemm(yt, ld, a.wt[b], nullptr, n_out, n_in, dt, ld);
    __syncthreads();
  }
  // ---- gradient of the pooled rows (NaN after a lost barrier: it reaches every gradient below the readout, and
  //      through the flat gradient buffer the optimizer's parameters -- wrong statistics cannot train silently)
  const bool lost = barrier_lost(barriers_ok, a.sync + kRdLostWord);
  const float nanv = __builtin_nanf("");
  for (int idx = tid; idx < rows * (h / 4); idx += kRdThreads) {
    const int r = idx / (h / 4), c4 = (idx - r * (h / 4)) * 4;
    gs_st4(a.dpooled + (row0 + r) * h + c4, lost ? f32x4{nanv, nanv, nanv, nanv} : gs_ld4(dt + r * ld + c4));
  }
  if (lost && blockIdx.x == 0 && tid < P) a.dbias_final[tid] = nanv;
}

#ifdef GS_GF_TIMING
extern "C" GNNSAFT_API int gnnsaft_debug_readout_stamps(long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rd_stamp), sizeof(long long) * 64);
}
#endif

size_t readout_fused_scratch_bytes(int64_t g, int h, int nblocks) {
  const int64_t wgs = gs_ceil_div(g > 0 ? g : 1, (int64_t)kRdRows);
  return gs_align_up((size_t)nblocks * wgs * 2 * h * 4, 256) + gs_align_up((size_t)wgs * 4, 256);
}

bool readout_fused_supported(int64_t g, int h, int num_para, int nblocks) {
  return g >= 1 && gs_ceil_div(g, (int64_t)kRdRows) <= kRdMaxWorkgroups && h >= 32 && h <= 256 && (h % 32) == 0 &&
         num_para >= 1 && num_para <= 32 && nblocks >= 2 && nblocks <= kRdMaxBlocks;
}

static size_t rd_fwd_lds(int h) {
  return ((size_t)2 * kRdRows * (h + kRdPad) + 2 * (size_t)h + 8) * sizeof(float) + 2 * (size_t)kRdThreads * sizeof(double);
}
static size_t rd_bwd_lds(int h) {
  return ((size_t)2 * kRdRows * (h + kRdPad) + 3 * (size_t)h + kRdRows * 8) * sizeof(float) +
         2 * (size_t)kRdThreads * sizeof(double);
}

// Workgroups of the fused readout kernels the CURRENT device keeps resident at once (0: unknown / query failed).
// A property of (device, kernel, LDS size): cached per device and hidden size; the first query of a device also
// raises that device's dynamic-LDS limit for the kernel (the attribute is per device, not per process).
static int readout_resident_workgroups(int h, bool backward) {
  constexpr int kMaxDev = 64;
  static std::atomic<int> cache[kMaxDev][2][9];   // zero-initialised: not yet queried
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  const int hi = h / 32;
  if (hi < 1 || hi > 8) return 0;
  const bool cached = dev >= 0 && dev < kMaxDev;
  if (cached) {
    const int v = cache[dev][backward ? 1 : 0][hi].load(std::memory_order_acquire);
    if (v != 0) return v > 0 ? v : 0;
  }
  const void *fn = backward ? reinterpret_cast<const void *>(&k_readout_bwd_fused)
                            : reinterpret_cast<const void *>(&k_readout_fused);
  const size_t lds = backward ? rd_bwd_lds(h) : rd_fwd_lds(h);
  int result = -1, cus = 0, per_cu = 0;
  // (the kernels also hold a few bytes of static LDS: ask for what the launch needs, not for the device's 160 KB)
  const hipError_t e1 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const hipError_t e2 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const hipError_t e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kRdThreads, lds);
  if (getenv("GNNSAFT_DEBUG"))
    fprintf(stderr, "readout residency h=%d bwd=%d: setattr %d, cus %d (%d), occupancy %d (%d), lds %zu\n", h,
            (int)backward, (int)e1, cus, (int)e2, per_cu, (int)e3, lds);
  if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && per_cu >= 1 && cus >= 1) result = per_cu * cus;
  (void)hipGetLastError();   // a failed query must not surface as the next launch's error
  if (cached) cache[dev][backward ? 1 : 0][hi].store(result, std::memory_order_release);
  return result > 0 ? result : 0;
}

}  // namespace gs

extern "C" int32_t gnnsaft_readout_resident_workgroups(int32_t hidden, int32_t backward) {
  return gs::readout_resident_workgroups(hidden, backward != 0);
}

namespace gs {

bool readout_fused_launchable(int64_t g, int h, int num_para, int nblocks, bool backward) {
  const bool shape = backward ? readout_bwd_fused_supported(g, h, num_para, nblocks)
                              : readout_fused_supported(g, h, num_para, nblocks);
  return shape && gs_ceil_div(g, (int64_t)kRdRows) <= readout_resident_workgroups(h, backward);
}

int launch_readout_fused(const ReadoutFusedParams &p, hipStream_t st) {
  GS_REQUIRE(readout_fused_launchable(p.g, p.h, p.num_para, p.nblocks, false), GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(p.pooled && p.out && p.scratch && p.sync, GNNSAFT_ERR_NULL);
  GS_REQUIRE(!p.training || p.g >= 2, GNNSAFT_ERR_SHAPE);   // torch: "Expected more than 1 value per channel"
  ReadoutArgs a;
  a.x = p.x;
  a.graph_ptr = p.graph_ptr;
  a.g = p.g;
  a.n = p.n;
  a.h = p.h;
  a.num_para = p.num_para;
  a.nblocks = p.nblocks;
  a.training = p.training;
  a.momentum = p.momentum;
  a.eps = p.eps;
  int width = p.h;
  for (int i = 0; i <= kRdMaxBlocks; ++i) {
    const bool live = i <= p.nblocks;
    a.w[i] = live ? p.w[i] : nullptr;
    a.b[i] = live ? p.b[i] : nullptr;
    a.n_in[i] = width;
    int n_out = p.h;
    if (i == p.nblocks - 2) n_out = p.h / 2;
    if (i == p.nblocks - 1) n_out = p.h / 4;
    if (i == p.nblocks) n_out = p.num_para;
    a.n_out[i] = n_out;
    width = n_out;
    GS_REQUIRE(!live || (a.w[i] != nullptr && a.b[i] != nullptr), GNNSAFT_ERR_NULL);
    if (i < kRdMaxBlocks) {
      const bool bl = i < p.nblocks;
      a.gamma[i] = bl ? p.bn[i].gamma : nullptr;
      a.beta[i] = bl ? p.bn[i].beta : nullptr;
      a.rmean[i] = bl ? p.bn[i].rmean : nullptr;
      a.rvar[i] = bl ? p.bn[i].rvar : nullptr;
      a.nbt[i] = bl ? p.bn[i].nbt : nullptr;
      GS_REQUIRE(!bl || (a.gamma[i] && a.beta[i] && a.rmean[i] && a.rvar[i]), GNNSAFT_ERR_NULL);
    }
  }
  a.target = p.target;
  a.out = p.out;
  a.loss3 = p.loss3;
  a.pooled = p.pooled;
  a.ry = p.ry;
  a.ro = p.ro;
  a.rstat = p.rstat;
  const int64_t wgs = gs_ceil_div(p.g, (int64_t)kRdRows);
  a.part = static_cast<float *>(p.scratch);
  a.mape_part = reinterpret_cast<float *>(static_cast<char *>(p.scratch) +
                                          gs_align_up((size_t)p.nblocks * wgs * 2 * p.h * 4, 256));
  a.sync = p.sync;
  a.err = p.err;
  a.k0_lost = p.k0_lost;
  a.barrier_extra = p.barrier_extra > 0 ? p.barrier_extra : 0;
  GS_REQUIRE(p.dropout_p >= 0.f && p.dropout_p < 1.f, GNNSAFT_ERR_SHAPE);
  a.dropout_p = p.training ? p.dropout_p : 0.f;
  a.dropout_seed = p.dropout_seed;
  a.dropout_step = p.dropout_step;
  const size_t lds = rd_fwd_lds(p.h);   // (the dynamic-LDS limit of this device was raised by the residency query)
  hipLaunchKernelGGL(k_readout_fused, dim3((unsigned)wgs), dim3(kRdThreads), lds, st, a);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

size_t readout_bwd_scratch_floats(int64_t g, int h, int nblocks) {
  const int64_t wgs = gs_ceil_div(g > 0 ? g : 1, (int64_t)kRdRows);
  return (size_t)nblocks * wgs * 2 * h + (size_t)wgs * 8 + 64;
}

size_t readout_bwd_slab_floats(int64_t g, int h, int num_para, int nblocks) {
  const int64_t wgs = gs_ceil_div(g > 0 ? g : 1, (int64_t)kRdRows);
  size_t per_wg = (size_t)num_para * (h / 4) + 64;
  int width = h;
  for (int i = 0; i < nblocks; ++i) {
    int n_out = h;
    if (i == nblocks - 2) n_out = h / 2;
    if (i == nblocks - 1) n_out = h / 4;
    per_wg += (size_t)n_out * width + 64;
    width = n_out;
  }
  return per_wg * (size_t)wgs;
}

bool readout_bwd_fused_supported(int64_t g, int h, int num_para, int nblocks) {
  return readout_fused_supported(g, h, num_para, nblocks) && num_para <= 8 && (h % 64) == 0 && g >= 2 &&
         nblocks + 1 <= kMaxSlabJobs;
}

int launch_readout_bwd_fused(const ReadoutBwdParams &p, SlabQueue &q, hipStream_t st) {
  GS_REQUIRE(readout_fused_launchable(p.g, p.h, p.num_para, p.nblocks, true), GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(p.grad_out && p.pooled && p.ry && p.ro && p.rstat && p.scratch && p.sync && p.dpooled, GNNSAFT_ERR_NULL);
  GS_REQUIRE(q.count + p.nblocks + 1 <= kMaxSlabJobs, GNNSAFT_ERR_WORKSPACE);
  const int64_t wgs = gs_ceil_div(p.g, (int64_t)kRdRows);
  ReadoutBwdArgs a;
  a.g = p.g;
  a.h = p.h;
  a.num_para = p.num_para;
  a.nblocks = p.nblocks;
  a.grad_out = p.grad_out;
  a.w_final = p.w[p.nblocks];
  a.pooled = p.pooled;
  a.ry = p.ry;
  a.ro = p.ro;
  a.rstat = p.rstat;
  a.dpooled = p.dpooled;
  a.dbias_final = p.db_final;
  a.sync = p.sync;
  a.err = p.err;
  a.barrier_extra = p.barrier_extra > 0 ? p.barrier_extra : 0;
  a.dropout_p = p.dropout_p;
  a.dropout_seed = p.dropout_seed;
  a.dropout_step = p.dropout_step;
  a.part = p.scratch;
  a.bias_part = p.scratch + (size_t)p.nblocks * wgs * 2 * p.h;
  GS_REQUIRE(a.w_final != nullptr && a.dbias_final != nullptr, GNNSAFT_ERR_NULL);
  int width = p.h;
  for (int i = 0; i <= kRdMaxBlocks; ++i) {
    const bool live = i <= p.nblocks;
    int n_out = p.h;
    if (i == p.nblocks - 2) n_out = p.h / 2;
    if (i == p.nblocks - 1) n_out = p.h / 4;
    if (i == p.nblocks) n_out = p.num_para;
    a.n_in[i] = width;
    a.n_out[i] = n_out;
    a.slab_w[i] = nullptr;
    if (live) {
      GS_REQUIRE(p.dw[i] != nullptr, GNNSAFT_ERR_NULL);
      a.slab_w[i] = q.take((size_t)wgs * n_out * width);
      GS_REQUIRE(a.slab_w[i] != nullptr, GNNSAFT_ERR_WORKSPACE);
      SlabOut so{{p.dw[i], p.dw[i], p.dw[i], p.dw[i]}, (int64_t)1 << 40};
      q.jobs[q.count++] = SlabJob{a.slab_w[i], (int64_t)n_out * width, wgs, width, so, width, 0};
    }
    width = n_out;
    if (i < kRdMaxBlocks) {
      const bool bl = i < p.nblocks;
      a.wt[i] = bl ? p.wt[i] : nullptr;
      a.gamma[i] = bl ? p.gamma[i] : nullptr;
      a.beta[i] = bl ? p.beta[i] : nullptr;
      a.dgamma[i] = bl ? p.dgamma[i] : nullptr;
      a.dbeta[i] = bl ? p.dbeta[i] : nullptr;
      GS_REQUIRE(!bl || (a.wt[i] && a.gamma[i] && a.beta[i]), GNNSAFT_ERR_NULL);
    }
  }
  const size_t lds = rd_bwd_lds(p.h);
  hipLaunchKernelGGL(k_readout_bwd_fused, dim3((unsigned)wgs), dim3(kRdThreads), lds, st, a);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

}  // namespace gs
