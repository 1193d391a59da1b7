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
  int32_t *sync;                          // kRdSyncInts ints, zero at launch
  int32_t *err;
  int barrier_extra;                      // test hook (desc->debug_barrier_extra): arrivals the barriers wait for in vain
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
};
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
