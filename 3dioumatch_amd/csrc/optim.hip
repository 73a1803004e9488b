// 3dioumatch_amd/csrc/optim.hip -- the optimizer step of the train scripts on one flat buffer.
//
// What it replaces: optimizer.step() of torch.optim.Adam(net.parameters(), lr, weight_decay)
// (pretrain.py:186 / :289, train.py:201 / :339) and, in the semi-supervised stage, the EMA update
// of the teacher that follows it (train.py:232-236 update_ema_variables).  The detector's
// parameters live in one flat buffer (votenet/step.py), so the whole update is one pass:
// read p, g, m, v (and the teacher), write p, m, v (and the teacher) -- 28 bytes per parameter
// instead of the ~15 element-wise kernels of the for-each implementation.
//
// The arithmetic is that of torch's Adam (no amsgrad, not maximize):
//     g'  = g * grad_scale + weight_decay * p          (g * grad_scale is also written back)
//     m   = m + (g' - m) * (1 - beta1);   v = beta2 * v + (1 - beta2) * g' * g'
//     p   = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with t = step + 1; the step counter and the learning rate are device scalars, so a captured
// HIP graph of this launch replays with the current values.
#include "common.h"
#include <math.h>

namespace {

// one lane: t = ++step, and the two scalars every parameter needs
__global__ void adam_prepare_kernel(float *__restrict__ step, const float *__restrict__ lr,
                                    double beta1, double beta2, float *__restrict__ scratch) {
  const float t = step[0] + 1.f;
  step[0] = t;
  const double bc1 = 1.0 - pow(beta1, (double)t);
  const double bc2 = 1.0 - pow(beta2, (double)t);
  scratch[0] = (float)((double)lr[0] / bc1);  // step size
  scratch[1] = (float)sqrt(bc2);
}

__global__ void __launch_bounds__(256)
adam_update_kernel(long long n, float *__restrict__ p, float *__restrict__ g,
                   float *__restrict__ m, float *__restrict__ v,
                   const float *__restrict__ scratch, float beta1, float beta2, float one_m_beta1,
                   float one_m_beta2, float eps, float weight_decay, float grad_scale,
                   float *__restrict__ ema,
                   const float *__restrict__ ema_weight) {
  const float step_size = scratch[0], bc2_sqrt = scratch[1];
  const float w = ema ? ema_weight[0] : 0.f;
  auto one = [&](float &pp, float gg, float &mm, float &vv) {
    gg = gg * grad_scale;
    if (weight_decay != 0.f) gg = gg + weight_decay * pp;
    mm = mm + (gg - mm) * one_m_beta1;
    vv = beta2 * vv + (one_m_beta2 * gg) * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp = pp - step_size * (mm / denom);
  };
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    float4 pp = *reinterpret_cast<float4 *>(p + i4);
    const float4 gg = *reinterpret_cast<const float4 *>(g + i4);
    if (grad_scale != 1.f)  // leave the mean behind where the sum was (p.grad of the modules)
      *reinterpret_cast<float4 *>(g + i4) = make_float4(gg.x * grad_scale, gg.y * grad_scale,
                                                        gg.z * grad_scale, gg.w * grad_scale);
    float4 mm = *reinterpret_cast<float4 *>(m + i4), vv = *reinterpret_cast<float4 *>(v + i4);
    one(pp.x, gg.x, mm.x, vv.x); one(pp.y, gg.y, mm.y, vv.y);
    one(pp.z, gg.z, mm.z, vv.z); one(pp.w, gg.w, mm.w, vv.w);
    *reinterpret_cast<float4 *>(p + i4) = pp;
    *reinterpret_cast<float4 *>(m + i4) = mm;
    *reinterpret_cast<float4 *>(v + i4) = vv;
    if (ema) {  // teacher <- teacher + (student - teacher) * (1 - alpha)
      float4 tt = *reinterpret_cast<float4 *>(ema + i4);
      tt.x = tt.x + w * (pp.x - tt.x); tt.y = tt.y + w * (pp.y - tt.y);
      tt.z = tt.z + w * (pp.z - tt.z); tt.w = tt.w + w * (pp.w - tt.w);
      *reinterpret_cast<float4 *>(ema + i4) = tt;
    }
  } else {
    for (long long i = i4; i < n; ++i) {
      float pp = p[i], mm = m[i], vv = v[i];
      one(pp, g[i], mm, vv);
      if (grad_scale != 1.f) g[i] = g[i] * grad_scale;
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (ema) ema[i] = ema[i] + w * (pp - ema[i]);
    }
  }
}

}  // namespace

// One Adam step on flat buffers of n floats (16-byte aligned); step (1 float, the count of steps
// taken so far; incremented), lr (1 float) and ema_weight (1 float, = 1 - alpha) live on the
// device; scratch: 2 floats.  ema == NULL: no teacher update.  The hyper-parameters arrive as
// doubles (Python floats): 1 - beta is rounded to fp32 once, as torch does.
PN2_API int votenet_adam_step(long long n, float *p, float *g, float *m, float *v, float *step,
                              const float *lr, double beta1, double beta2, double eps,
                              double weight_decay, double grad_scale, float *ema,
                              const float *ema_weight, float *scratch, void *stream_) {
  if (n <= 0) return 0;
  if (!p || !g || !m || !v || !step || !lr || !scratch || (ema && !ema_weight))
    return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, stream, step, lr, beta1, beta2, scratch);
  hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)pn2_ceil_div(n, 1024LL)), dim3(256), 0, stream, n,
                     p, g, m, v, scratch, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                     (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)grad_scale, ema,
                     ema_weight);
  return pn2_launch_status();
}
