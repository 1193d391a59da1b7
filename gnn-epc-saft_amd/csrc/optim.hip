// Fused optimizer steps over ONE flat f32 parameter buffer -- the training-step host loop next to
// the hot path (the reference's choices are
// /root/reference/gnnepcsaft/train/models.py:162-178: torch.optim.AdamW(amsgrad=True, eps=1e-5) or
// torch.optim.SGD(nesterov=True), stepped once per batch).  torch's foreach implementations issue
// ~10 launches over ~50 small tensors; here a step is one HBM-bound pass: AdamW reads p, g, m, v,
// vmax and writes p, m, v, vmax (36 B / parameter), SGD reads p, g, buf and writes p, buf (20 B).
// Arithmetic follows torch's single-tensor reference update op for op (same f32 rounding points),
// -ffp-contract=off keeps hipcc from fusing what torch leaves unfused.
#include <cmath>
#include <cstring>

#include "common.hpp"

namespace gs {

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, grad_scale;
  float step_size;            // lr / (1 - beta1^t)
  float bias2_sqrt;           // sqrt(1 - beta2^t)
  float decay, omb1, omb2;    // 1 - lr * wd, 1 - beta1, 1 - beta2: Python-float (double) expressions in torch
};

__device__ __forceinline__ float adamw_one(float &p, float g, float &m, float &v, float &vmax, const AdamArgs &a,
                                           bool amsgrad) {
  g *= a.grad_scale;
  p = p * a.decay;                                  // param.mul_(1 - lr * wd)
  m = m + a.omb1 * (g - m);                         // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + (a.omb2 * g) * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  float denom_src = v;
  if (amsgrad) {
    vmax = fmaxf(vmax, v);
    denom_src = vmax;
  }
  const float denom = sqrtf(denom_src) / a.bias2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);                // param.addcdiv_(exp_avg, denom, value=-step_size)
  return p;
}

__device__ __forceinline__ void adamw_span(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                           float *__restrict__ v, float *__restrict__ vmax, int64_t count,
                                           const AdamArgs &a) {
  const bool amsgrad = vmax != nullptr;
  const int64_t n4 = count >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pp = gs_ld4(p + 4 * i), mm = gs_ld4(m + 4 * i), vv = gs_ld4(v + 4 * i);
    const f32x4 gg = gs_ld4(g + 4 * i);
    f32x4 xx = amsgrad ? gs_ld4(vmax + 4 * i) : vv;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float pc = pp[c], mc = mm[c], vc = vv[c], xc = xx[c];
      adamw_one(pc, gg[c], mc, vc, xc, a, amsgrad);
      pp[c] = pc;
      mm[c] = mc;
      vv[c] = vc;
      xx[c] = xc;
    }
    gs_st4(p + 4 * i, pp);
    gs_st4(m + 4 * i, mm);
    gs_st4(v + 4 * i, vv);
    if (amsgrad) gs_st4(vmax + 4 * i, xx);
  }
  // scalar tail
  const int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && t < count) {
    float x = amsgrad ? vmax[t] : 0.f;
    adamw_one(p[t], g[t], m[t], v[t], x, a, amsgrad);
    if (amsgrad) vmax[t] = x;
  }
}

__global__ __launch_bounds__(256) void k_adamw(float *__restrict__ p, const float *__restrict__ g,
                                               float *__restrict__ m, float *__restrict__ v,
                                               float *__restrict__ vmax, int64_t count, AdamArgs a) {
  adamw_span(p, g, m, v, vmax, count, a);
}

// the step's scalars (learning rate, bias corrections: they change every step) read from DEVICE memory, so that the
// launch can sit in a captured hipGraph and be replayed with new values.  gnnsaft_adamw_args publishes them with a
// one-thread launch whose values travel as KERNEL ARGUMENTS (copied at enqueue time): stream-ordered in front of the
// replay that reads them, and no host buffer a queued replay could see in a later step's state.
__global__ void k_adamw_publish(AdamArgs a, AdamArgs *__restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = a;
}

__global__ __launch_bounds__(256) void k_adamw_dev(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   float *__restrict__ vmax, int64_t count,
                                                   const AdamArgs *__restrict__ ap) {
  const AdamArgs a = *ap;
  adamw_span(p, g, m, v, vmax, count, a);
}

// torch.optim.SGD(momentum, nesterov=True, dampening=0): g += wd p; buf = first ? g : mu buf + g; g += mu buf; p -= lr g
__global__ __launch_bounds__(256) void k_sgd_nesterov(float *__restrict__ p, const float *__restrict__ g,
                                                      float *__restrict__ buf, int64_t count, float lr, float mu,
                                                      float wd, int first, float grad_scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    if (mu != 0.f) {
      const float b = first ? gi : buf[i] * mu + gi;
      buf[i] = b;
      gi = gi + mu * b;
    }
    p[i] = pi - lr * gi;
  }
}

static unsigned optim_grid(int64_t items) {
  int64_t b = gs_ceil_div(items, 256);
  if (b > 256 * 8) b = 256 * 8;  // grid-stride beyond 8 workgroups per CU
  return (unsigned)(b > 0 ? b : 1);
}

}  // namespace gs

static gs::AdamArgs adam_args(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                              float grad_scale) {
  // bias corrections in double on the host, as torch computes them from the Python-float step count
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  return gs::AdamArgs{lr, beta1, beta2, eps, weight_decay, grad_scale, (float)((double)lr / bc1), (float)std::sqrt(bc2),
                      (float)(1.0 - (double)lr * (double)weight_decay), (float)(1.0 - (double)beta1),
                      (float)(1.0 - (double)beta2)};
}

static int adam_check(const float *param, const float *grad, const float *exp_avg, const float *exp_avg_sq,
                      const float *max_exp_avg_sq, int64_t count) {
  GS_REQUIRE(param && grad && exp_avg && exp_avg_sq, GNNSAFT_ERR_NULL);
  GS_REQUIRE(count >= 0, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
               reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq) |
               reinterpret_cast<uintptr_t>(max_exp_avg_sq)) & 15) == 0,
             GNNSAFT_ERR_SHAPE);
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                  float *max_exp_avg_sq, int64_t count, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int64_t step, float grad_scale, gnnsaft_stream_t stream) {
  if (const int rc = adam_check(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, count); rc != GNNSAFT_OK) return rc;
  GS_REQUIRE(step >= 1, GNNSAFT_ERR_SHAPE);
  if (count == 0) return GNNSAFT_OK;
  const gs::AdamArgs a = adam_args(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  hipLaunchKernelGGL(gs::k_adamw, dim3(gs::optim_grid(count / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, count, a);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int32_t gnnsaft_adamw_args_floats(void) { return (int32_t)(sizeof(gs::AdamArgs) / sizeof(float)); }

extern "C" int gnnsaft_adamw_args(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                  float grad_scale, float *args_dev, gnnsaft_stream_t stream) {
  GS_REQUIRE(args_dev != nullptr && (reinterpret_cast<uintptr_t>(args_dev) & 3) == 0, GNNSAFT_ERR_NULL);
  GS_REQUIRE(step >= 1, GNNSAFT_ERR_SHAPE);
  const gs::AdamArgs a = adam_args(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  hipLaunchKernelGGL(gs::k_adamw_publish, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a,
                     reinterpret_cast<gs::AdamArgs *>(args_dev));
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_adamw_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                      float *max_exp_avg_sq, int64_t count, const float *args_dev,
                                      gnnsaft_stream_t stream) {
  if (const int rc = adam_check(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, count); rc != GNNSAFT_OK) return rc;
  GS_REQUIRE(args_dev != nullptr && (reinterpret_cast<uintptr_t>(args_dev) & 3) == 0, GNNSAFT_ERR_NULL);
  if (count == 0) return GNNSAFT_OK;
  hipLaunchKernelGGL(gs::k_adamw_dev, dim3(gs::optim_grid(count / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, count,
                     reinterpret_cast<const gs::AdamArgs *>(args_dev));
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_sgd_step(float *param, const float *grad, float *momentum_buf, int64_t count, float lr,
                                float momentum, float weight_decay, int32_t first_step, float grad_scale,
                                gnnsaft_stream_t stream) {
  GS_REQUIRE(param && grad, GNNSAFT_ERR_NULL);
  GS_REQUIRE(momentum == 0.f || momentum_buf != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(count >= 0, GNNSAFT_ERR_SHAPE);
  if (count == 0) return GNNSAFT_OK;
  hipLaunchKernelGGL(gs::k_sgd_nesterov, dim3(gs::optim_grid(count)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     param, grad, momentum_buf, count, lr, momentum, weight_decay, first_step, grad_scale);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}
