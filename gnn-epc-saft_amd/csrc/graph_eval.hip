// Per-graph fused EVAL-mode forward: PNAPCSAFT.forward
// (/root/reference/gnnepcsaft/train/models.py:105-135) for one molecular graph per workgroup, the whole network
// in ONE launch, templated on the arithmetic type (float | double).
//
// Why it exists (SURVEY.md section 8(b) "Module behaviours callers rely on", 8(f) rank 4):
//   * the reference's inference callers run the model in float64, eval mode, one un-batched `Data` at a time
//     (evaluations/evaluate_ensemble.py:67-77,145,185; demo/utils.py:23-27,141-152; validation_step
//     models.py:204-211).  Eval mode has no BatchNorm barrier between layers, so a graph never has to leave its CU;
//   * the batched pipeline of forward.hip is ~50 dependent launches: 216 us for one ethanol molecule.
//
// Layout of the work.  `gnnsaft_eval_pack` turns the module's parameters into a PACK once per weight version:
// every Linear transposed to k-major (consecutive lanes read consecutive output columns), eval-mode BatchNorm folded
// into the preceding Linear, the edge branch collapsed to a per-layer table over the <= 60 bond-attribute classes
// (rtab[c] = W_pre[:, 2F:3F] (W_e emb_c + b_e) + b_pre).  `k_graph_forward` then walks the layers for its graph:
//   P|Q = x W_pq        (destination / source halves of pre_nns[t][0], per node)
//   m_e = P[dst] + Q[src] + rtab[class]   (+ extra pre layers on edge rows), aggregated mean|min|max|std per node
//   u   = post_nns[t](cat[x, A, A amp, A att])   (three degree scalers applied per F-block of K, never materialised)
//   x'  = relu(lin'(u)) (+ x)           (BatchNorm folded into lin')
// then add-pool and the readout MLP.  Node tiles of R rows live in LDS; the node state x / P|Q lives in LDS when the
// graph is small enough and in a global scratch otherwise (flat pointers: one code path).
//
// GEMM inside the workgroup (1024 threads = 16 waves): a wave owns 16 output columns x 4 interleaved k-lanes
// (64-B coalesced weight segments, conflict-free LDS reads of the A rows), partial sums meet through two DPP shuffles,
// waves split K further when there are fewer than 256 columns; every thread keeps R row accumulators.  f32 / f64 FMA
// on the vector ALU: with <= 64 rows per graph the matrix cores would idle on fragment padding, and the whole thing
// is latency-bound on the weight stream from L2 (one CU), which is why 8 independent loads are kept in flight.
#include <atomic>

#include "plan.hpp"

namespace gs {

struct EvalShape {
  int h, L, pre, post, mlp, P, skip, loops, combos;
  int n_atom_cols, n_bond_cols, atom_rows;
  int atom_dims[GNNSAFT_MAX_TABLES], atom_row0[GNNSAFT_MAX_TABLES], bond_dims[GNNSAFT_MAX_TABLES];
};

// element offsets inside the pack (all multiples of 4)
struct EvalLayout {
  int64_t atoms, layer0, layer_stride;
  int64_t wpq, rtab, prex, prex_stride, wpost, bpost, postx, postx_stride, wlin, slin, blin, avg;   // inside a layer
  int64_t readout0, tmp_cemb, tmp_cenc, total;
};

__host__ __device__ inline int64_t up4(int64_t v) { return (v + 3) & ~(int64_t)3; }

__host__ __device__ inline EvalLayout eval_layout(const EvalShape &s) {
  EvalLayout l;
  const int64_t h = s.h;
  int64_t o = 0;
  l.atoms = o;
  o += up4((int64_t)s.atom_rows * h);
  l.layer0 = o;
  int64_t p = 0;
  l.wpq = p;
  p += 4 * h * h;
  l.rtab = p;
  p += up4((int64_t)s.combos * 2 * h);
  l.prex = p;
  l.prex_stride = 2 * h * h + 2 * h;   // w [F][2F] (own tower's F inputs per column) + b [2F]
  p += (int64_t)(s.pre - 1) * l.prex_stride;
  l.wpost = p;
  p += 13 * h * h;
  l.bpost = p;
  p += h;
  l.postx = p;
  l.postx_stride = (h / 2) * h + h;    // w [F/2][F] + b [F]
  p += (int64_t)(s.post - 1) * l.postx_stride;
  l.wlin = p;
  p += h * h;
  l.slin = p;    // eval-mode BatchNorm: y = (x W^T) * slin + blin,  slin = gamma rstd, blin = (b - mean) slin + beta
  p += h;
  l.blin = p;
  p += h;
  l.avg = p;
  p += 4;
  l.layer_stride = up4(p);
  o += (int64_t)s.L * l.layer_stride;
  l.readout0 = o;
  int width = s.h;
  for (int i = 0; i < s.mlp + 3; ++i) {
    const int n_out = i < s.mlp ? s.h : (i == s.mlp ? s.h / 2 : (i == s.mlp + 1 ? s.h / 4 : s.P));
    o += up4((int64_t)width * n_out) + 2 * up4(n_out);
    width = n_out;
  }
  l.tmp_cemb = o;
  o += up4((int64_t)s.combos * h);
  l.tmp_cenc = o;
  o += up4((int64_t)s.combos * h);
  l.total = o;
  return l;
}

// readout block i: offsets of its k-major weight [n_in][n_out], scale [n_out] (at b_off - up4(n_out)) and shift
__host__ __device__ inline void readout_block(const EvalShape &s, const EvalLayout &l, int i, int64_t &w_off,
                                              int64_t &b_off, int &n_in, int &n_out) {
  int64_t o = l.readout0;
  int width = s.h;
  for (int j = 0;; ++j) {
    const int no = j < s.mlp ? s.h : (j == s.mlp ? s.h / 2 : (j == s.mlp + 1 ? s.h / 4 : s.P));
    if (j == i) {
      w_off = o;
      b_off = o + up4((int64_t)width * no) + up4(no);
      n_in = width;
      n_out = no;
      return;
    }
    o += up4((int64_t)width * no) + 2 * up4(no);
    width = no;
  }
}

static int make_shape(const gnnsaft_model_desc *d, EvalShape &s) {
  GS_REQUIRE(d != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(d->hidden >= 32 && (d->hidden % 32) == 0 && d->hidden <= 256, GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(d->num_layers >= 0 && d->pre_layers >= 1 && d->pre_layers <= 8 && d->post_layers >= 1 &&
                 d->post_layers <= 8 && d->num_mlp_layers >= 0 && d->num_mlp_layers <= 8 && d->num_para >= 1 &&
                 d->num_para <= 64,
             GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(d->num_atom_cols >= 1 && d->num_atom_cols <= GNNSAFT_MAX_TABLES && d->num_bond_cols >= 1 &&
                 d->num_bond_cols <= GNNSAFT_MAX_TABLES,
             GNNSAFT_ERR_UNSUPPORTED);
  s.h = d->hidden;
  s.L = d->num_layers;
  s.pre = d->pre_layers;
  s.post = d->post_layers;
  s.mlp = d->num_mlp_layers;
  s.P = d->num_para;
  s.skip = d->skip_connections;
  s.loops = d->self_loops;
  s.n_atom_cols = d->num_atom_cols;
  s.n_bond_cols = d->num_bond_cols;
  int rows = 0;
  int64_t combos = 1;
  for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) {
    s.atom_dims[k] = k < d->num_atom_cols ? d->atom_dims[k] : 1;
    s.atom_row0[k] = rows;
    if (k < d->num_atom_cols) rows += d->atom_dims[k];
    s.bond_dims[k] = k < d->num_bond_cols ? d->bond_dims[k] : 1;
    if (k < d->num_bond_cols) {
      GS_REQUIRE(d->bond_dims[k] >= 1, GNNSAFT_ERR_SHAPE);
      combos *= d->bond_dims[k];
      GS_REQUIRE(combos <= 4096, GNNSAFT_ERR_UNSUPPORTED);
    }
  }
  s.atom_rows = rows;
  s.combos = (int)combos;
  return GNNSAFT_OK;
}

// ------------------------------------------------------------------------------------------ pack kernels
template <typename T>
__device__ __forceinline__ T t_sqrt(T v);
template <>
__device__ __forceinline__ float t_sqrt<float>(float v) { return sqrtf(v); }
template <>
__device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }
template <typename T>
__device__ __forceinline__ T t_log(T v);
template <>
__device__ __forceinline__ float t_log<float>(float v) { return (float)log((double)v); }   // correctly rounded, as common.hpp's degree scalers
template <>
__device__ __forceinline__ double t_log<double>(double v) { return log(v); }
template <typename T>
__device__ __forceinline__ T t_max(T a, T b) { return a > b ? a : b; }
template <typename T>
__device__ __forceinline__ T t_min(T a, T b) { return a < b ? a : b; }
// fused multiply-add of the dot products.  The library is built with -ffp-contract=off (the batched pipeline keeps
// torch's rounding points), which would make every multiply-add of this VALU-bound kernel two instructions; a GEMM's
// summation order is free anyway (the oracle's BLAS uses FMA too), so the dots contract explicitly.
template <typename T>
__device__ __forceinline__ T t_fma(T a, T b, T c);
template <>
__device__ __forceinline__ float t_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <>
__device__ __forceinline__ double t_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// dst[k * ldd + c0 + j] = src[j * lds + k] * s_j,  s_j = gamma_j / sqrt(var_j + eps) (eval-mode BatchNorm) or 1
template <typename T>
__global__ void k_pack_transpose(const T *__restrict__ src, int64_t lds, int rows_j, int cols_k, T *__restrict__ dst,
                                 int64_t ldd, int c0, const T *__restrict__ gamma, const T *__restrict__ var, T eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows_j * cols_k) return;
  const int k = (int)(i / rows_j), j = (int)(i - (int64_t)k * rows_j);
  T s = (T)1;
  if (gamma != nullptr) s = gamma[j] / t_sqrt<T>(var[j] + eps);
  dst[(int64_t)k * ldd + c0 + j] = src[(int64_t)j * lds + k] * s;
}

// dst[c0 + j] = (b_j - mean_j) s_j + beta_j   (BatchNorm folded)   or  b_j
template <typename T>
__global__ void k_pack_bias(const T *__restrict__ b, int n, T *__restrict__ dst, int c0, const T *__restrict__ gamma,
                            const T *__restrict__ beta, const T *__restrict__ mean, const T *__restrict__ var, T eps) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  T v = b != nullptr ? b[j] : (T)0;
  if (gamma != nullptr) v = (v - mean[j]) * (gamma[j] / t_sqrt<T>(var[j] + eps)) + beta[j];
  dst[c0 + j] = v;
}

// dst[j] = gamma_j / sqrt(var_j + eps)  (or 1)
template <typename T>
__global__ void k_pack_scale(int n, T *__restrict__ dst, const T *__restrict__ gamma, const T *__restrict__ var, T eps) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  dst[j] = gamma != nullptr ? gamma[j] / t_sqrt<T>(var[j] + eps) : (T)1;
}

template <typename T>
struct PackTables {
  int n;
  int dims[GNNSAFT_MAX_TABLES];
  const T *tab[GNNSAFT_MAX_TABLES];
};

// atom tables concatenated; cemb[c,:] = sum_k bond_tab_k[digit_k(c), :]
template <typename T>
__global__ void k_pack_tables(PackTables<T> atoms, PackTables<T> bonds, int h, int combos, T *__restrict__ atoms_out,
                              T *__restrict__ cemb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int rows = 0;
  for (int k = 0; k < atoms.n; ++k) rows += atoms.dims[k];
  if (i < (int64_t)rows * h) {
    int v = (int)(i / h);
    const int c = (int)(i - (int64_t)v * h);
    int t = 0;
    while (t + 1 < atoms.n && v >= atoms.dims[t]) v -= atoms.dims[t++];
    atoms_out[i] = atoms.tab[t][(int64_t)v * h + c];
  }
  if (i < (int64_t)combos * h) {
    const int cid = (int)(i / h), c = (int)(i - (int64_t)cid * h);
    int digit[GNNSAFT_MAX_TABLES];
    int rem = cid;
    for (int k = bonds.n - 1; k >= 0; --k) {
      digit[k] = rem % bonds.dims[k];
      rem /= bonds.dims[k];
    }
    T acc = (T)0;
    for (int k = 0; k < bonds.n; ++k) acc += bonds.tab[k][(int64_t)digit[k] * h + c];
    cemb[i] = acc;
  }
}

// out[r, c0 + j] = sum_k a[r, k] w[j, k0 + k] + b[j]   (tiny: rows = bond classes)
template <typename T>
__global__ void k_pack_linear(const T *__restrict__ a, int64_t lda, int rows, int k, const T *__restrict__ w,
                              int64_t ldw, int k0, const T *__restrict__ b, int n_out, T *__restrict__ out,
                              int64_t ldo, int c0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * n_out) return;
  const int r = (int)(i / n_out), j = (int)(i - (int64_t)r * n_out);
  T acc = (T)0;
  for (int kk = 0; kk < k; ++kk) acc += a[(int64_t)r * lda + kk] * w[(int64_t)j * ldw + k0 + kk];
  out[(int64_t)r * ldo + c0 + j] = acc + (b != nullptr ? b[j] : (T)0);
}

template <typename T>
static int pack_impl(const gnnsaft_model_desc *d, const EvalShape &s, const EvalLayout &lay, const ParsedWeights &pw,
                     T *pack, hipStream_t st) {
  const int h = s.h;
  const T eps = d->bn_eps_f64 > 0.0 ? (T)d->bn_eps_f64 : (T)d->bn_eps;
  auto C = [](const float *p) { return reinterpret_cast<const T *>(p); };
  auto transpose = [&](const T *src, int64_t lds, int rows_j, int cols_k, T *dst, int64_t ldd, int c0, const T *gamma,
                       const T *var) {
    const int64_t tot = (int64_t)rows_j * cols_k;
    hipLaunchKernelGGL(k_pack_transpose<T>, dim3((unsigned)gs_ceil_div(tot, 256)), dim3(256), 0, st, src, lds, rows_j,
                       cols_k, dst, ldd, c0, gamma, var, eps);
  };
  auto bias = [&](const T *b, int n, T *dst, int c0, const BnPtrs *bn) {
    hipLaunchKernelGGL(k_pack_bias<T>, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, st, b, n, dst, c0,
                       bn ? C(bn->gamma) : nullptr, bn ? C(bn->beta) : nullptr,
                       bn ? reinterpret_cast<const T *>(bn->rmean) : nullptr,
                       bn ? reinterpret_cast<const T *>(bn->rvar) : nullptr, eps);
  };
  auto scale = [&](int n, T *dst, const BnPtrs *bn) {
    hipLaunchKernelGGL(k_pack_scale<T>, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, st, n, dst,
                       bn ? C(bn->gamma) : nullptr, bn ? reinterpret_cast<const T *>(bn->rvar) : nullptr, eps);
  };
  {
    PackTables<T> at, bt;
    at.n = s.n_atom_cols;
    bt.n = s.n_bond_cols;
    for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) {
      at.dims[k] = s.atom_dims[k];
      bt.dims[k] = s.bond_dims[k];
      at.tab[k] = k < s.n_atom_cols ? C(pw.atom_tab[k]) : nullptr;
      bt.tab[k] = k < s.n_bond_cols ? C(pw.bond_tab[k]) : nullptr;
    }
    const int64_t tot = (int64_t)(s.atom_rows > s.combos ? s.atom_rows : s.combos) * h;
    hipLaunchKernelGGL(k_pack_tables<T>, dim3((unsigned)gs_ceil_div(tot, 256)), dim3(256), 0, st, at, bt, h, s.combos,
                       pack + lay.atoms, pack + lay.tmp_cemb);
  }
  for (int l = 0; l < s.L; ++l) {
    const LayerW &w = pw.layers[l];
    T *base = pack + lay.layer0 + (int64_t)l * lay.layer_stride;
    // wpq [H][4F]: P_t0 | P_t1 | Q_t0 | Q_t1
    for (int t = 0; t < 2; ++t) {
      transpose(C(w.wpre[t][0]), 3 * (int64_t)h, h, h, base + lay.wpq, 4 * (int64_t)h, t * h, nullptr, nullptr);
      transpose(C(w.wpre[t][0]) + h, 3 * (int64_t)h, h, h, base + lay.wpq, 4 * (int64_t)h, 2 * h + t * h, nullptr,
                nullptr);
    }
    // rtab[c, tF + f] = W_pre,t[f, 2F:3F] (W_e cemb_c + b_e) + b_pre,t[f]
    {
      const int64_t tot = (int64_t)s.combos * h;
      hipLaunchKernelGGL(k_pack_linear<T>, dim3((unsigned)gs_ceil_div(tot, 256)), dim3(256), 0, st,
                         pack + lay.tmp_cemb, (int64_t)h, s.combos, h, C(w.we), (int64_t)h, 0, C(w.be), h,
                         pack + lay.tmp_cenc, (int64_t)h, 0);
      for (int t = 0; t < 2; ++t)
        hipLaunchKernelGGL(k_pack_linear<T>, dim3((unsigned)gs_ceil_div(tot, 256)), dim3(256), 0, st,
                           pack + lay.tmp_cenc, (int64_t)h, s.combos, h, C(w.wpre[t][0]), 3 * (int64_t)h, 2 * h,
                           C(w.bpre[t][0]), h, base + lay.rtab, 2 * (int64_t)h, t * h);
    }
    for (int j = 1; j < s.pre; ++j) {
      T *px = base + lay.prex + (int64_t)(j - 1) * lay.prex_stride;
      for (int t = 0; t < 2; ++t) {
        transpose(C(w.wpre[t][j]), h, h, h, px, 2 * (int64_t)h, t * h, nullptr, nullptr);
        bias(C(w.bpre[t][j]), h, px + 2 * (int64_t)h * h, t * h, nullptr);
      }
    }
    for (int t = 0; t < 2; ++t) {
      transpose(C(w.wpost[t][0]), 13 * (int64_t)h, h / 2, 13 * h, base + lay.wpost, h, t * (h / 2), nullptr, nullptr);
      bias(C(w.bpost[t][0]), h / 2, base + lay.bpost, t * (h / 2), nullptr);
    }
    for (int j = 1; j < s.post; ++j) {
      T *px = base + lay.postx + (int64_t)(j - 1) * lay.postx_stride;
      for (int t = 0; t < 2; ++t) {
        transpose(C(w.wpost[t][j]), h / 2, h / 2, h / 2, px, h, t * (h / 2), nullptr, nullptr);
        bias(C(w.bpost[t][j]), h / 2, px + (int64_t)(h / 2) * h, t * (h / 2), nullptr);
      }
    }
    transpose(C(w.wlin), h, h, h, base + lay.wlin, h, 0, nullptr, nullptr);
    scale(h, base + lay.slin, &w.bn);
    bias(C(w.blin), h, base + lay.blin, 0, &w.bn);
    bias(C(w.avg), 1, base + lay.avg, 0, nullptr);
  }
  for (int i = 0; i < s.mlp + 3; ++i) {
    int64_t wo, bo;
    int n_in, n_out;
    readout_block(s, lay, i, wo, bo, n_in, n_out);
    const ReadoutW &r = pw.readout[i];
    const bool bn = i < s.mlp + 2;
    transpose(C(r.w), n_in, n_out, n_in, pack + wo, n_out, 0, nullptr, nullptr);
    scale(n_out, pack + bo - up4(n_out), bn ? &r.bn : nullptr);
    bias(C(r.b), n_out, pack + bo, 0, bn ? &r.bn : nullptr);
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

// ------------------------------------------------------------------------------------------ the graph kernel
constexpr int kGfThreads = 1024;       // 16 waves: R <= 8 (float32) / 4 (float64) accumulator rows per thread
constexpr int kGfThreadsWide = 512;    // 8 waves, twice the registers per thread: 16 / 8 rows -- a 9-16 atom molecule
                                       // is ONE node tile (every tile re-streams the layer's weights)
constexpr int kGfLdsNodes = 64;    // in-kernel CSR (single small graph): at most this many nodes ...
constexpr int kGfLdsEdges = 256;   // ... and directed edges

template <typename T>
struct GraphArgs {
  EvalShape s;
  EvalLayout lay;
  const T *pack;
  const int64_t *x_idx;
  // CSR built by csr.hip (batch / large graph) -- or, when rowptr == nullptr, the raw edge list of ONE small graph
  const int32_t *rowptr, *src, *combo, *graph_ptr;
  const int64_t *edge_index, *edge_attr;
  int64_t n, e;
  T *out;
  T *gx0, *gx1, *gpq, *gmsg;   // global scratch: node state ping-pong [N,H], P|Q [N,4F], messages [E',2F] (pre > 1)
  int lds_state_elems;          // T elements of LDS left for the node state
  int32_t *err;
};

#define GS_LDS(T) __attribute__((address_space(3))) T

#ifdef GS_GF_TIMING   // development probe: wall-clock (100 MHz) stamps of workgroup 0 at phase boundaries
__device__ long long g_gf_stamp[256];
#define GF_STAMP(i)                                                      \
  do {                                                                   \
    __syncthreads();                                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0 && (i) < 256) g_gf_stamp[(i)] = wall_clock64(); \
  } while (0)
#else
#define GF_STAMP(i) do {} while (0)
#endif

template <typename T, int R, int NT, class Accum, class Epi>
__device__ __forceinline__ void wg_gemm(int n_out, int K, GS_LDS(T) *red, Accum accum, Epi epi) {
  constexpr int kWaves = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane >> 4, cl = lane & 15;
  int np2 = 16;
  while (np2 < n_out && np2 < 16 * kWaves) np2 <<= 1;  // columns per pass
  const int groups = np2 >> 4;                 // waves side by side over the columns
  const int ksw = kWaves / groups;             // waves stacked over K
  const int cg = wave % groups, ks = wave / groups;
  constexpr int KSTEP = 4 * 16 / (int)sizeof(T);   // k covered by one wave step (4 k-lanes x 16 B)
  const int kper = (((K + ksw - 1) / ksw) + KSTEP - 1) / KSTEP * KSTEP;
  const int kb = ks * kper;
  const int ke = K < kb + kper ? K : kb + kper;
  for (int c0 = 0; c0 < n_out; c0 += np2) {
    const int col = c0 + cg * 16 + cl;
    const bool col_ok = col < n_out;
    T acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = (T)0;
    accum(acc, col_ok ? col : n_out - 1, kb, ke, kl);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r] += __shfl_xor(acc[r], 16);
      acc[r] += __shfl_xor(acc[r], 32);
    }
    if (ksw == 1) {
      if (kl == 0 && col_ok) {
#pragma unroll
        for (int r = 0; r < R; ++r) epi(r, col, acc[r]);
      }
    } else {
      if (kl == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[(ks * R + r) * np2 + cg * 16 + cl] = acc[r];
      }
      __syncthreads();
      for (int idx = tid; idx < R * np2; idx += NT) {
        const int r = idx / np2, cc = idx - r * np2;
        if (c0 + cc < n_out) {
          T v = (T)0;
          for (int s2 = 0; s2 < ksw; ++s2) v += red[(s2 * R + r) * np2 + cc];
          epi(r, c0 + cc, v);
        }
      }
      __syncthreads();
    }
  }
}

// acc[r] += sum over this lane's k of f(A[r][k]) * W[k][col] for the wave's K range [kb, ke).  The four k-lanes of a
// wave take V = 16 B / sizeof(T) CONSECUTIVE k each per step (k = s + V kl + j): one ds_read_b128 per A row and step
// (rows past the tile's fill hold garbage that is never stored), V weight loads from L2 per step, 64 B per lane in
// flight while the previous batch is consumed.
template <typename T, int R, bool RELU>
__device__ __forceinline__ void dot_range(T (&acc)[R], const GS_LDS(T) *a, int lda, const T *__restrict__ w, int64_t ldw,
                                          int col, int kb, int ke, int kl) {
  constexpr int V = 16 / (int)sizeof(T);   // k per lane and step
  // steps per batch: 64 B of weights per lane in flight (32 B with 8 accumulator rows: the register budget)
  constexpr int S = (R >= 8 ? 8 : 16) / V / ((int)sizeof(T) / 4);
  typedef T vecT __attribute__((ext_vector_type(V)));
  if (kb >= ke) return;
  // 32-bit unsigned element offsets from the (wave-uniform) base: `global_load v, v_off, s[base]` -- one address
  // register per load in flight instead of two
  const unsigned ld = (unsigned)ldw;
  auto fetch = [&](int s0, T(&wv)[S * V]) {
#pragma unroll
    for (int st = 0; st < S; ++st) {
      const int k4 = s0 + st * 4 * V + V * kl;
      const unsigned o = (unsigned)(k4 < ke ? k4 : ke - V) * ld + (unsigned)col;
#pragma unroll
      for (int j2 = 0; j2 < V; ++j2) wv[st * V + j2] = w[o + (unsigned)j2 * ld];
    }
  };
  T wv[S * V];
  fetch(kb, wv);
  for (int s0 = kb; s0 < ke; s0 += S * 4 * V) {
    T wn[S * V];
    fetch(s0 + S * 4 * V < ke ? s0 + S * 4 * V : s0, wn);   // next batch (re-reads this one at the end: harmless)
#pragma unroll
    for (int st = 0; st < S; ++st) {
      const int k4 = s0 + st * 4 * V + V * kl;
      const bool live = k4 < ke;
      const int kc = live ? k4 : ke - V;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const vecT av = *reinterpret_cast<const GS_LDS(vecT) *>(a + r * lda + kc);
#pragma unroll
        for (int j2 = 0; j2 < V; ++j2) {
          T x = av[j2];
          if (RELU) x = t_max<T>(x, (T)0);
          acc[r] = t_fma<T>(x, live ? wv[st * V + j2] : (T)0, acc[r]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < S * V; ++q) wv[q] = wn[q];
  }
}

template <typename T, int R, int NT>
__global__ __launch_bounds__(NT) void k_graph_forward(GraphArgs<T> a) {
  constexpr int kRed = 16 * (NT / 64) * R;   // wg_gemm's cross-wave reduction buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const EvalShape &s = a.s;
  const EvalLayout &lay = a.lay;
  const int h = s.h, f = s.h, tid = threadIdx.x;
  // ---- LDS carve-up
  GS_LDS(T) *red = (GS_LDS(T) *)smem;          // [kRed]
  GS_LDS(T) *xt = red + kRed;                  // [R][F]   x tile (update) / scratch
  GS_LDS(T) *zt = xt + R * f;                  // [R][2][13F] update operand cat[x, A, A amp, A att] per tower
                                               //             (also: edge-row tiles of the extra pre layers)
  GS_LDS(T) *ut = zt + R * 26 * f;             // [R][H]   update output
  GS_LDS(T) *ut2 = ut + R * h;                 // [R][H]   ping-pong for extra post layers
  T *state = (T *)(ut2 + R * h + 2 * R);                 // optional: x0 [n][H], x1 [n][H], pq [n][4F] (flat: LDS or global)
  int32_t *l_int = reinterpret_cast<int32_t *>(state + a.lds_state_elems);  // in-kernel CSR (single small graph)

  GF_STAMP(250);
  const int g = blockIdx.x;
  int64_t node0, n_g;
  if (a.graph_ptr != nullptr) {
    node0 = a.graph_ptr[g];
    n_g = a.graph_ptr[g + 1] - node0;
  } else {
    node0 = 0;
    n_g = a.n;
  }
  const int n = (int)n_g;

  // ---- graph structure: CSR rows of node i (local) = [rp[i], rp[i+1]) ; src ids GLOBAL node ids ; combo class ids
  const int32_t *rp, *srcs, *combos;
  int64_t row_base = 0;  // rp values are offsets relative to...: global CSR -> absolute rows
  if (a.rowptr != nullptr) {
    rp = a.rowptr + node0;
    srcs = a.src;
    combos = a.combo;
  } else {
    // one small graph: build the destination-sorted CSR here (edge order kept, self-loop last)
    int32_t *l_rp = l_int, *l_cnt = l_int + (kGfLdsNodes + 1), *l_src = l_cnt + kGfLdsNodes,
            *l_combo = l_src + (kGfLdsEdges + kGfLdsNodes);
    const int e = (int)a.e;
    for (int i = tid; i < n; i += NT) l_cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < e; i += NT) {
      const int64_t sv = a.edge_index[i], dv = a.edge_index[e + i];
      if (sv < 0 || sv >= n || dv < 0 || dv >= n) {
        if (a.err) atomicOr(a.err, GNNSAFT_FLAG_BAD_EDGE);
      } else {
        atomicAdd(&l_cnt[dv], 1);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int i = 0; i < n; ++i) {
        l_rp[i] = run;
        run += l_cnt[i] + (s.loops ? 1 : 0);
      }
      l_rp[n] = run;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {  // thread per destination: edges in edge_index order
      int pos = l_rp[i];
      for (int ed = 0; ed < e; ++ed) {
        const int64_t sv = a.edge_index[ed], dv = a.edge_index[e + ed];
        if (dv != i || sv < 0 || sv >= n) continue;
        int cid = 0;
        for (int k = 0; k < s.n_bond_cols; ++k) {
          int64_t v = a.edge_attr[(int64_t)ed * s.n_bond_cols + k];
          if (v < 0 || v >= s.bond_dims[k]) {
            if (a.err) atomicOr(a.err, GNNSAFT_FLAG_BAD_ATTR);
            v = 0;
          }
          cid = cid * s.bond_dims[k] + (int)v;
        }
        l_src[pos] = (int)sv;
        l_combo[pos] = cid;
        ++pos;
      }
      if (s.loops) {
        l_src[pos] = i;
        l_combo[pos] = 0;
      }
    }
    __syncthreads();
    rp = l_rp;
    srcs = l_src;
    combos = l_combo;
  }
  (void)row_base;

  // ---- node state: LDS if it fits, else the global scratch (flat pointers either way)
  T *x_cur, *x_nxt, *pq;
  if ((int64_t)n * 6 * h <= a.lds_state_elems) {
    x_cur = state;
    x_nxt = state + (int64_t)n * h;
    pq = state + (int64_t)n * 2 * h;
  } else {
    x_cur = a.gx0 + node0 * h;
    x_nxt = a.gx1 + node0 * h;
    pq = a.gpq + node0 * 4 * h;
  }
  // source ids are global node ids with the csr.hip structure, graph-local with the in-kernel one
  const int64_t src_shift = a.rowptr != nullptr ? node0 : 0;

  // ---- AtomEncoder: x[i, c] = sum_k tab_k[idx[i, k], c]  (left to right)
  {
    const T *atoms = a.pack + lay.atoms;
    for (int idx = tid; idx < n * h; idx += NT) {
      const int i = idx / h, c = idx - i * h;
      T acc = (T)0;
      for (int k = 0; k < s.n_atom_cols; ++k) {
        int64_t v = a.x_idx[(node0 + i) * s.n_atom_cols + k];
        if (v < 0 || v >= s.atom_dims[k]) {
          if (a.err) atomicOr(a.err, GNNSAFT_FLAG_BAD_ATTR);
          v = 0;
        }
        acc += atoms[(int64_t)(s.atom_row0[k] + (int)v) * h + c];
      }
      x_cur[(int64_t)i * h + c] = acc;
    }
  }
  __syncthreads();
  GF_STAMP(0);

  for (int l = 0; l < s.L; ++l) {
    const T *lw = a.pack + lay.layer0 + (int64_t)l * lay.layer_stride;
    const T *rtab = lw + lay.rtab;
    const T avg = lw[lay.avg];
    // ---- P | Q for every node of the graph
    for (int t0 = 0; t0 < n; t0 += R) {
      const int rows = n - t0 < R ? n - t0 : R;
      for (int idx = tid; idx < rows * f; idx += NT) {
        const int r = idx / f, c = idx - r * f;
        xt[r * f + c] = x_cur[(int64_t)(t0 + r) * h + c];
      }
      __syncthreads();
      const T *w = lw + lay.wpq;
      wg_gemm<T, R, NT>(
          4 * f, h, red,
          [&](T(&acc)[R], int col, int kb, int ke, int kl) { dot_range<T, R, false>(acc, xt, f, w, 4 * (int64_t)h, col, kb, ke, kl); },
          [&](int r, int col, T v) {
            if (r < rows) pq[(int64_t)(t0 + r) * 4 * h + col] = v;
          });
      __syncthreads();
    }
    __syncthreads();
    GF_STAMP(1 + 5 * l);
    // ---- extra pre layers: messages of every CSR row of the graph -> gmsg[row, 2F]
    const int row_lo = rp[0], row_hi = rp[n];
    if (s.pre > 1) {
      GS_LDS(T) *ea = zt, *eb = zt + R * 2 * f;  // two [R][2F] edge-row tiles
      for (int e0 = row_lo; e0 < row_hi; e0 += R) {
        const int rows = row_hi - e0 < R ? row_hi - e0 : R;
        // h1 = P[dst] + Q[src] + rtab[class]   (pre-activation of the first pre layer)
        for (int idx = tid; idx < rows * 2 * f; idx += NT) {
          const int r = idx / (2 * f), c = idx - r * 2 * f;
          const int row = e0 + r;
          // destination of a CSR row: binary search in rp
          int lo = 0, hi = n - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rp[mid] <= row) lo = mid; else hi = mid - 1;
          }
          const int64_t sj = (int64_t)srcs[row] - src_shift;
          ea[r * 2 * f + c] = (pq[(int64_t)lo * 4 * h + c] + pq[sj * 4 * h + 2 * f + c]) + rtab[(int64_t)combos[row] * 2 * f + c];
        }
        __syncthreads();
        GS_LDS(T) *cur = ea, *nxt = eb;
        for (int j = 1; j < s.pre; ++j) {
          const T *px = lw + lay.prex + (int64_t)(j - 1) * lay.prex_stride;
          const T *bx = px + 2 * (int64_t)h * h;
          const bool last = j == s.pre - 1;
          wg_gemm<T, R, NT>(
              2 * f, f, red,
              [&](T(&acc)[R], int col, int kb, int ke, int kl) {
                dot_range<T, R, true>(acc, cur + (col / f) * f, 2 * f, px, 2 * (int64_t)h, col, kb, ke, kl);
              },
              [&](int r, int col, T v) {
                if (r >= rows) return;
                v += bx[col];
                if (last) a.gmsg[(int64_t)(e0 + r) * 2 * f + col] = v;
                else nxt[r * 2 * f + col] = v;
              });
          __syncthreads();
          GS_LDS(T) *tswap = cur;
          cur = nxt;
          nxt = tswap;
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    // ---- node tiles: aggregate -> update -> lin (+BN) -> ReLU -> (+x)
    for (int t0 = 0; t0 < n; t0 += R) {
      const int rows = n - t0 < R ? n - t0 : R;
      for (int idx = tid; idx < rows * f; idx += NT) {
        const int r = idx / f, c = idx - r * f;
        const T xv = x_cur[(int64_t)(t0 + r) * h + c];
        xt[r * f + c] = xv;
        zt[(r * 2 + 0) * 13 * f + c] = xv;
        zt[(r * 2 + 1) * 13 * f + c] = xv;
      }
      // mean | min | max | std over the in-edges (sums of m - m_first: no cancellation in the variance)
      for (int idx = tid; idx < rows * 2 * f; idx += NT) {
        const int r = idx / (2 * f), c = idx - r * 2 * f;
        const int i = t0 + r;
        const int beg = rp[i], end = rp[i + 1];
        const int tw = c >= f ? 1 : 0, col = c - tw * f;
        T mean = (T)0, mn = (T)0, mx = (T)0, sd = (T)0;
        if (end > beg) {
          const T p = s.pre > 1 ? (T)0 : pq[(int64_t)i * 4 * h + c];
          T v0 = (T)0, sm = (T)0, s2 = (T)0;
          for (int row = beg; row < end; ++row) {
            T v;
            if (s.pre > 1) {
              v = a.gmsg[(int64_t)row * 2 * f + c];
            } else {
              const int64_t sj = (int64_t)srcs[row] - src_shift;
              v = (p + pq[sj * 4 * h + 2 * f + c]) + rtab[(int64_t)combos[row] * 2 * f + c];
            }
            if (row == beg) {
              v0 = v;
              mn = v;
              mx = v;
            }
            const T dlt = v - v0;
            sm += dlt;
            s2 += dlt * dlt;
            mn = t_min<T>(mn, v);
            mx = t_max<T>(mx, v);
          }
          const T cnt = (T)(end - beg);
          const T dm = sm / cnt;
          mean = v0 + dm;
          const T var = s2 / cnt - dm * dm;
          const T o = t_sqrt<T>(t_max<T>(var, (T)1e-5));
          sd = o <= (T)0.0031622776601683794 ? (T)0 : o;   // torch compares with the scalar rounded to T
        }
        // cat[A, A amp, A att] with A = [mean | min | max | std]: the three degree scalers applied while writing the tile
        const int deg = end - beg;
        const T amp = t_log<T>((T)deg + (T)1) / avg, att = avg / t_log<T>(t_max<T>((T)deg, (T)1) + (T)1);
        GS_LDS(T) *o4 = zt + (r * 2 + tw) * 13 * f + f + col;
        const T vals[4] = {mean, mn, mx, sd};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o4[q * f] = vals[q];
          o4[4 * f + q * f] = vals[q] * amp;
          o4[8 * f + q * f] = vals[q] * att;
        }
      }
      __syncthreads();
      if (t0 == 0) GF_STAMP(2 + 5 * l);
      // update: u[r, t F/2 + o] = W_t[o, :] . cat[x, A_t, amp A_t, att A_t] + b   (K = 13 F, one weight stream)
      {
        const T *w = lw + lay.wpost, *b = lw + lay.bpost;
        wg_gemm<T, R, NT>(
            f, 13 * f, red,
            [&](T(&acc)[R], int col, int kb, int ke, int kl) {
              dot_range<T, R, false>(acc, zt + (col / (f / 2)) * 13 * f, 26 * f, w, (int64_t)h, col, kb, ke, kl);
            },
            [&](int r, int col, T v) {
              if (r < rows) ut[r * h + col] = v + b[col];
            });
      }
      __syncthreads();
      if (t0 == 0) GF_STAMP(3 + 5 * l);
      GS_LDS(T) *ucur = ut, *unxt = ut2;
      for (int j = 1; j < s.post; ++j) {
        const T *px = lw + lay.postx + (int64_t)(j - 1) * lay.postx_stride;
        const T *bx = px + (int64_t)(h / 2) * h;
        wg_gemm<T, R, NT>(
            h, h / 2, red,
            [&](T(&acc)[R], int col, int kb, int ke, int kl) {
              dot_range<T, R, true>(acc, ucur + (col / (h / 2)) * (h / 2), h, px, (int64_t)h, col, kb, ke, kl);
            },
            [&](int r, int col, T v) {
              if (r < rows) unxt[r * h + col] = v + bx[col];
            });
        __syncthreads();
        GS_LDS(T) *tswap = ucur;
        ucur = unxt;
        unxt = tswap;
      }
      // lin with the eval-mode BatchNorm folded in, ReLU, residual
      {
        const T *w = lw + lay.wlin, *sc = lw + lay.slin, *b = lw + lay.blin;
        wg_gemm<T, R, NT>(
            h, h, red,
            [&](T(&acc)[R], int col, int kb, int ke, int kl) { dot_range<T, R, false>(acc, ucur, h, w, (int64_t)h, col, kb, ke, kl); },
            [&](int r, int col, T v) {
              if (r >= rows) return;
              v = t_max<T>(v * sc[col] + b[col], (T)0);
              if (s.skip) v += xt[r * f + col];
              x_nxt[(int64_t)(t0 + r) * h + col] = v;
            });
      }
      __syncthreads();
      if (t0 == 0) GF_STAMP(4 + 5 * l);
    }
    T *tswap = x_cur;
    x_cur = x_nxt;
    x_nxt = tswap;
    __threadfence_block();
    __syncthreads();
    GF_STAMP(5 + 5 * l);
  }

  // ---- global_add_pool + readout MLP (BatchNorm folded), one row
  for (int c = tid; c < h; c += NT) {
    T acc = (T)0;
    for (int i = 0; i < n; ++i) acc += x_cur[(int64_t)i * h + c];
    ut[c] = acc;
  }
  __syncthreads();
  GS_LDS(T) *cur = ut, *nxt = ut2;
  for (int i = 0; i < s.mlp + 3; ++i) {
    int64_t wo, bo;
    int n_in, n_out;
    readout_block(s, lay, i, wo, bo, n_in, n_out);
    const T *w = a.pack + wo, *b = a.pack + bo, *sc = b - up4(n_out);
    const bool last = i == s.mlp + 2;
    wg_gemm<T, R, NT>(
        n_out, n_in, red,
        [&](T(&acc)[R], int col, int kb, int ke, int kl) {   // lda = 0: every accumulator row sees the one pooled row
          dot_range<T, R, false>(acc, cur, 0, w, (int64_t)n_out, col, kb, ke, kl);
        },
        [&](int r, int col, T v) {
          if (r != 0) return;
          v = v * sc[col] + b[col];
          if (last) a.out[(int64_t)g * s.P + col] = v;
          else nxt[col] = t_max<T>(v, (T)0);
        });
    __syncthreads();
    GS_LDS(T) *tswap = cur;
    cur = nxt;
    nxt = tswap;
  }
  GF_STAMP(251);
}

static size_t gf_fixed_lds_elems(int h, int r, int threads = kGfThreads) {
  return (size_t)16 * (threads / 64) * r + (size_t)r * h + (size_t)r * 26 * h + 2 * (size_t)r * h + 2 * (size_t)r;
}
static size_t gf_csr_lds_bytes() { return (size_t)(kGfLdsNodes + 1 + kGfLdsNodes + 2 * (kGfLdsEdges + kGfLdsNodes)) * 4 + 16; }
constexpr size_t kGfLdsBudget = 150 * 1024;  // of the 160 KB of a gfx950 CU

// rows per node tile = accumulators per thread: at most 8 in float32 / 4 in float64 (128 VGPRs at 1024 threads), and
// what the [R][2][13F] operand tile leaves of the LDS; halved for a single molecule that would leave the tile half empty
static int gf_tile_rows(int h, size_t elem, int64_t n, int64_t g, int &threads) {
  threads = kGfThreads;
  int r = elem == 4 ? 8 : 4;
  while (r > 2 && gf_fixed_lds_elems(h, r) * elem + gf_csr_lds_bytes() > kGfLdsBudget - 16 * 1024) r /= 2;
  const int floor_r = elem == 4 ? 4 : 2;
  if (g == 1 && n <= r / 2 && r / 2 >= floor_r) r /= 2;
  // a single molecule with more atoms than one tile holds: 512 threads with twice the accumulator rows, when the
  // wider operand tile still fits (hidden_dim 64, the reference's default model)
  const int wide_r = elem == 4 ? 16 : 8;
  if (g == 1 && n > r && r == wide_r / 2 &&
      gf_fixed_lds_elems(h, wide_r, kGfThreadsWide) * elem + gf_csr_lds_bytes() <= kGfLdsBudget - 16 * 1024) {
    r = wide_r;
    threads = kGfThreadsWide;
  }
  return r;
}

struct GfPlan {
  size_t csr_ws, rowptr, src, dst, combo, log_amp, log_att, graph_ptr, gx0, gx1, gpq, gmsg, total;
  bool lds_csr;
};

static int gf_plan(const gnnsaft_model_desc *d, size_t elem, int64_t n, int64_t e, int64_t g, GfPlan &p) {
  GS_REQUIRE(n >= 1 && e >= 0 && g >= 1, GNNSAFT_ERR_SHAPE);
  const int64_t ep = e + (d->self_loops ? n : 0);
  GS_REQUIRE(ep + n < ((int64_t)1 << 31) - 1, GNNSAFT_ERR_SHAPE);
  p.lds_csr = g == 1 && n <= kGfLdsNodes && e <= kGfLdsEdges;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += gs_align_up(bytes, 256);
    return o;
  };
  const size_t nn = (size_t)n, ee = (size_t)(ep > 0 ? ep : 1), h = (size_t)d->hidden;
  p.csr_ws = take(p.lds_csr ? 0 : gnnsaft_csr_workspace_bytes(n, e));
  p.rowptr = take((nn + 1) * 4);
  p.src = take(ee * 4);
  p.dst = take(ee * 4);
  p.combo = take(ee * 4);
  p.log_amp = take(nn * 4);
  p.log_att = take(nn * 4);
  p.graph_ptr = take((size_t)(g + 1) * 4);
  p.gx0 = take(nn * h * elem);
  p.gx1 = take(nn * h * elem);
  p.gpq = take(nn * 4 * h * elem);
  p.gmsg = take(d->pre_layers > 1 ? ee * 2 * h * elem : 0);
  p.total = off;
  return GNNSAFT_OK;
}

template <typename T, int R, int NT = kGfThreads>
static int gf_launch(const GraphArgs<T> &a, int64_t g, size_t lds_bytes, hipStream_t st) {
  // the dynamic-LDS limit is a property of (kernel, DEVICE): once per device this process drives, and safe from
  // several host threads (the loader's prefetch thread makes HIP calls beside the main thread)
  static std::atomic<unsigned long long> attr_devices{0};   // bit d: raised on device d (devices >= 64: every launch)
  int dev = 0;
  GS_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if ((attr_devices.load(std::memory_order_acquire) & bit) == 0ull) {
    GS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_graph_forward<T, R, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGfLdsBudget));
    attr_devices.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_graph_forward<T, R, NT>), dim3((unsigned)g), dim3(NT), lds_bytes, st, a);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

template <typename T>
static int graph_forward_impl(const gnnsaft_model_desc *d, const void *pack, const int64_t *x_idx,
                              const int64_t *edge_index, const int64_t *edge_attr, const int64_t *batch, int64_t n,
                              int64_t e, int64_t g, void *out, int32_t *err_flag, void *workspace,
                              size_t workspace_bytes, hipStream_t st) {
  EvalShape s;
  GS_TRY(make_shape(d, s));
  GfPlan p;
  GS_TRY(gf_plan(d, sizeof(T), n, e, g, p));
  GS_REQUIRE(workspace_bytes >= p.total && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, GNNSAFT_ERR_WORKSPACE);
  GS_REQUIRE(batch != nullptr || g == 1, GNNSAFT_ERR_SHAPE);
  char *ws = static_cast<char *>(workspace);
  GraphArgs<T> a;
  a.s = s;
  a.lay = eval_layout(s);
  a.pack = static_cast<const T *>(pack);
  a.x_idx = x_idx;
  a.edge_index = edge_index;
  a.edge_attr = edge_attr;
  a.n = n;
  a.e = e;
  a.out = static_cast<T *>(out);
  a.gx0 = reinterpret_cast<T *>(ws + p.gx0);
  a.gx1 = reinterpret_cast<T *>(ws + p.gx1);
  a.gpq = reinterpret_cast<T *>(ws + p.gpq);
  a.gmsg = reinterpret_cast<T *>(ws + p.gmsg);
  a.err = err_flag;
  if (p.lds_csr) {
    a.rowptr = a.src = a.combo = a.graph_ptr = nullptr;
  } else {
    auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    GS_TRY(launch_csr_build(edge_index, edge_attr, n, e, d->num_bond_cols, d->bond_dims, d->self_loops, I(p.rowptr),
                            I(p.src), I(p.dst), I(p.combo), F(p.log_amp), F(p.log_att), err_flag, ws + p.csr_ws,
                            gnnsaft_csr_workspace_bytes(n, e), batch, g, I(p.graph_ptr), nullptr, false, st));
    a.rowptr = I(p.rowptr);
    a.src = I(p.src);
    a.combo = I(p.combo);
    a.graph_ptr = I(p.graph_ptr);
  }
  int threads = kGfThreads;
  const int r = gf_tile_rows(s.h, sizeof(T), n, g, threads);
  const size_t fixed = gf_fixed_lds_elems(s.h, r, threads) * sizeof(T);
  const size_t csr_b = gf_csr_lds_bytes();
  GS_REQUIRE(fixed + csr_b <= kGfLdsBudget, GNNSAFT_ERR_UNSUPPORTED);
  size_t state_elems = (kGfLdsBudget - fixed - csr_b) / sizeof(T);
  state_elems &= ~(size_t)3;
  // LDS is allocated per workgroup: do not reserve more state than the graph can use (batches: room for 28-atom
  // molecules, which keeps two workgroups per CU at H = 64; larger graphs keep their state in the global scratch)
  const size_t want = (size_t)(g == 1 ? n : 28) * 6 * (size_t)s.h;
  if (state_elems > want) state_elems = (want + 3) & ~(size_t)3;
  a.lds_state_elems = (int)state_elems;
  const size_t lds_bytes = fixed + state_elems * sizeof(T) + csr_b;
  if constexpr (sizeof(T) == 4) {
    if (threads == kGfThreadsWide) return gf_launch<T, 16, kGfThreadsWide>(a, g, lds_bytes, st);
    return r == 8 ? gf_launch<T, 8>(a, g, lds_bytes, st) : gf_launch<T, 4>(a, g, lds_bytes, st);
  } else {
    if (threads == kGfThreadsWide) return gf_launch<T, 8, kGfThreadsWide>(a, g, lds_bytes, st);
    return r == 4 ? gf_launch<T, 4>(a, g, lds_bytes, st) : gf_launch<T, 2>(a, g, lds_bytes, st);
  }
}

}  // namespace gs

using namespace gs;

#ifdef GS_GF_TIMING
extern "C" GNNSAFT_API int gnnsaft_debug_graph_stamps(long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gf_stamp), sizeof(long long) * 256);
}
#endif

extern "C" size_t gnnsaft_eval_pack_bytes(const gnnsaft_model_desc *desc, int32_t dtype) {
  EvalShape s;
  if (make_shape(desc, s) != GNNSAFT_OK || (dtype != GNNSAFT_DTYPE_F32 && dtype != GNNSAFT_DTYPE_F64)) return 0;
  return (size_t)eval_layout(s).total * (dtype == GNNSAFT_DTYPE_F64 ? 8 : 4);
}

extern "C" int gnnsaft_eval_pack(const gnnsaft_model_desc *desc, const void *const *weights_host, int32_t num_weights,
                                 int32_t dtype, void *pack, size_t pack_bytes, gnnsaft_stream_t stream) {
  GS_REQUIRE(desc && weights_host && pack, GNNSAFT_ERR_NULL);
  GS_REQUIRE(dtype == GNNSAFT_DTYPE_F32 || dtype == GNNSAFT_DTYPE_F64, GNNSAFT_ERR_UNSUPPORTED);
  EvalShape s;
  GS_TRY(make_shape(desc, s));
  GS_REQUIRE(num_weights == gnnsaft_num_weights(desc), GNNSAFT_ERR_SHAPE);
  const EvalLayout lay = eval_layout(s);
  GS_REQUIRE(pack_bytes >= gnnsaft_eval_pack_bytes(desc, dtype) && (reinterpret_cast<uintptr_t>(pack) & 15) == 0,
             GNNSAFT_ERR_WORKSPACE);
  ParsedWeights pw;
  GS_TRY(parse_weights(desc, weights_host, num_weights, pw));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == GNNSAFT_DTYPE_F64) return pack_impl<double>(desc, s, lay, pw, static_cast<double *>(pack), st);
  return pack_impl<float>(desc, s, lay, pw, static_cast<float *>(pack), st);
}

extern "C" size_t gnnsaft_graph_forward_workspace_bytes(const gnnsaft_model_desc *desc, int32_t dtype,
                                                        int64_t num_nodes, int64_t num_edges, int64_t num_graphs) {
  EvalShape s;
  GfPlan p;
  if (make_shape(desc, s) != GNNSAFT_OK) return 0;
  if (gf_plan(desc, dtype == GNNSAFT_DTYPE_F64 ? 8 : 4, num_nodes, num_edges, num_graphs, p) != GNNSAFT_OK) return 0;
  return p.total;
}

extern "C" int gnnsaft_graph_forward(const gnnsaft_model_desc *desc, int32_t dtype, const void *pack,
                                     const int64_t *x, const int64_t *edge_index, const int64_t *edge_attr,
                                     const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                                     void *out, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                                     gnnsaft_stream_t stream) {
  GS_REQUIRE(desc && pack && x && out && workspace, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_edges == 0 || (edge_index && edge_attr), GNNSAFT_ERR_NULL);
  GS_REQUIRE(!desc->training, GNNSAFT_ERR_UNSUPPORTED);  // eval-mode BatchNorm only: no barrier between layers
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == GNNSAFT_DTYPE_F64)
    return graph_forward_impl<double>(desc, pack, x, edge_index, edge_attr, batch, num_nodes, num_edges, num_graphs, out,
                                      err_flag, workspace, workspace_bytes, st);
  GS_REQUIRE(dtype == GNNSAFT_DTYPE_F32, GNNSAFT_ERR_UNSUPPORTED);
  return graph_forward_impl<float>(desc, pack, x, edge_index, edge_attr, batch, num_nodes, num_edges, num_graphs, out,
                                   err_flag, workspace, workspace_bytes, st);
}
