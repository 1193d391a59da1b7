// Fused readout (readout.hip): parameters of the one-launch add-pool -> MLP -> MAPE kernel.
#pragma once
#include "plan.hpp"

namespace gs {

constexpr int kRdMaxBlocks = 10;        // BatchNorm blocks of the readout: num_mlp_layers (<= 8) + 2
constexpr int kRdMaxWorkgroups = 1024;  // static upper bound (sizing); the launch-time bound is what the DEVICE can
                                        // keep co-resident for the grid barrier: readout_resident_workgroups()
constexpr int kRdSyncInts = 16;         // barrier / ticket counters, zeroed by the forward's prologue kernel

struct ReadoutFusedParams {
  const float *x;
  const int32_t *graph_ptr;
  int64_t g, n;
  int h, num_para, nblocks, training;
  float momentum, eps;
  const float *w[kRdMaxBlocks + 1], *b[kRdMaxBlocks + 1];   // nblocks BatchNorm blocks, then the final Linear
  BnPtrs bn[kRdMaxBlocks];
  const float *target;
  float *out, *loss3;
  float *pooled, *ry, *ro, *rstat;       // kept for the backward (may be null)
  void *scratch;                          // readout_fused_scratch_bytes
  int32_t *sync;                          // kRdSyncInts ints, zero at launch (the last one: "this call lost a barrier")
  int32_t *err;
  const int32_t *k0_lost;                 // or null: 1 = the structure chain of this call lost a barrier (NaN outputs)
  int barrier_extra;                      // test hook (desc->debug_barrier_extra): arrivals the barriers wait for in vain
  float dropout_p;                        // Dropout behind every ReLU of the readout (models.py:88,95,99); training only
  uint64_t dropout_seed;                  // Philox key of this call's masks
  const uint64_t *dropout_step = nullptr; // device word mixed into the key at run time (gnnsaft_model_desc), or null
};

// Backward of the readout in one launch (k_readout_bwd_fused): per-workgroup partial weight gradients go to `q`'s
// arena and are reduced by its flush (launch_slab_queue_flush); BatchNorm and final-bias gradients are written directly.
struct ReadoutBwdParams {
  int64_t g;
  int h, num_para, nblocks;
  const float *grad_out;                     // [G, P]
  const float *w[kRdMaxBlocks + 1];          // only the final Linear's [P, H/4] is read
  const float *wt[kRdMaxBlocks];             // transposed block weights [n_in][n_out]
  const float *gamma[kRdMaxBlocks], *beta[kRdMaxBlocks];
  float *dw[kRdMaxBlocks + 1];               // weight gradients (final Linear last)
  float *dgamma[kRdMaxBlocks], *dbeta[kRdMaxBlocks];
  float *db_final;                           // [P]
  const float *pooled, *ry, *ro, *rstat;     // the forward's tape
  float *dpooled;                            // [G, H] out
  float *scratch;                            // readout_bwd_scratch_floats
  int32_t *sync;                             // kRdSyncInts ints, zero at launch
  int32_t *err;                              // or null
  int barrier_extra;                         // test hook, as in the forward
  float dropout_p;                           // the forward's Dropout: the masks are regenerated from the same key
  uint64_t dropout_seed;
  const uint64_t *dropout_step = nullptr;    // as in the forward
};

// Dropout masks of the readout: Philox4x32-10 keyed by the call's seed, counter = (graph row, block, column / 4):
// four keep decisions for four consecutive columns of one row of one block -- a pure function of (seed, position), so
// forward and backward agree without storing a mask, whatever workgroup evaluates it.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}
// keep-scale factors (0 or 1 / (1 - p)) of columns c4 .. c4 + 3 of (row, block)
// the key of THIS run of the kernel: the call's seed, plus -- under hipGraph replay, where the seed is a frozen kernel
// argument -- a device word the caller bumps in front of every replay (times an odd constant: distinct keys)
__device__ __forceinline__ uint64_t rd_key(uint64_t seed, const uint64_t *step) {
  return step != nullptr ? seed + step[0] * 0x9E3779B97F4A7C15ull : seed;
}

__device__ __forceinline__ f32x4 dropout_scale4(uint64_t seed, int64_t row, int block, int c4, float p) {
  uint32_t r[4];
  philox4x32_10((uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)block, (uint32_t)(c4 >> 2), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  const float keep = 1.f / (1.f - p);
  f32x4 m;
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = (float)(r[j] >> 8) * (1.f / 16777216.f) >= p ? keep : 0.f;   // uniform [0,1) >= p
  return m;
}
size_t readout_bwd_scratch_floats(int64_t g, int h, int nblocks);
size_t readout_bwd_slab_floats(int64_t g, int h, int num_para, int nblocks);
bool readout_bwd_fused_supported(int64_t g, int h, int num_para, int nblocks);
int launch_readout_bwd_fused(const ReadoutBwdParams &p, SlabQueue &q, hipStream_t st);

size_t readout_fused_scratch_bytes(int64_t g, int h, int nblocks);
// shape envelope of the fused kernels (host-only: sizing functions use it) ...
bool readout_fused_supported(int64_t g, int h, int num_para, int nblocks);
// ... and the launch-time decision: the envelope AND every workgroup co-resident on the CURRENT device
// (hipOccupancyMaxActiveBlocksPerMultiprocessor x multiProcessorCount, cached per device and hidden size)
bool readout_fused_launchable(int64_t g, int h, int num_para, int nblocks, bool backward);
int launch_readout_fused(const ReadoutFusedParams &p, hipStream_t st);

}  // namespace gs
