// SPDX-License-Identifier: MIT
// Fused host step: what the reference does around every raytrace with ~60 small torch kernels -
//   gradient import   renderer/gaussian_raytracer.py:50-58  (pc._x.grad.add_(raytracer dL_dx), 8 tensors)
//   scale decay       train.py:224-226                      (_scaling = log(exp(_scaling) * decay))
//   Adam              scene/gaussian_model.py:333-338       (8 groups, eps 1e-15, betas from the config, lr per group)
//   clamps            train.py:251-254                      (diffuse >= 0, roughness / f0 in [0,1])
//   zero_grad x2      train.py:248-249
//   parameter export  renderer/gaussian_raytracer.py:36-48  (raytracer tensors = model parameters, next iteration)
// - as ONE launch: blockIdx.y = parameter group, one thread per element (fully coalesced streams, ~0.6 KB per gaussian).
// The arithmetic follows torch.optim.Adam's default (non-capturable, non-amsgrad, no weight decay) single-tensor path step
// for step: lerp, mul+addcmul, sqrt / bias_correction2_sqrt + eps, addcdiv with step_size = lr / bias_correction1 (the two
// bias corrections are evaluated on the host in double like torch does).
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/egr_raytracer.h"

namespace {

struct StepArgs {
    egr_param_group g[EGR_MAX_PARAM_GROUPS];
    float step_size[EGR_MAX_PARAM_GROUPS]; // lr / (1 - beta1^t)
    float bc2_sqrt[EGR_MAX_PARAM_GROUPS];  // sqrt(1 - beta2^t); t is per group (torch keeps one step count per parameter tensor)
    float w1, beta2, w2, eps;              // w1 = 1 - beta1, w2 = 1 - beta2
    uint32_t n;
};

__global__ void __launch_bounds__(256) k_fused_step(StepArgs a) {
    const egr_param_group &G = a.g[blockIdx.y];
    const size_t total = (size_t)a.n * G.width;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float p = G.param[i];
        float g = G.grad ? G.grad[i] : 0.0f;
        if (G.rt_grad) g += G.rt_grad[i]; // import
        if (G.log_decay != 1.0f) p = logf(expf(p) * G.log_decay);
        if (G.exp_avg) { // Adam
            float m = G.exp_avg[i], v = G.exp_avg_sq[i];
            m = m + a.w1 * (g - m);         // lerp_(grad, 1 - beta1)
            v = v * a.beta2 + a.w2 * g * g; // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
            const float denom = sqrtf(v) / a.bc2_sqrt[blockIdx.y] + a.eps;
            p = p - a.step_size[blockIdx.y] * (m / denom); // addcdiv_(exp_avg, denom, value = -step_size)
            G.exp_avg[i] = m, G.exp_avg_sq[i] = v;
        }
        p = fminf(fmaxf(p, G.clamp_min), G.clamp_max);
        G.param[i] = p;
        if (G.rt_param) G.rt_param[i] = p; // export
        if (G.grad) G.grad[i] = 0.0f;
        if (G.rt_grad) G.rt_grad[i] = 0.0f;
    }
}

thread_local std::string g_step_error;

} // namespace

extern "C" const char *egr_fused_step_last_error(void) { return g_step_error.c_str(); }

extern "C" int egr_fused_adam_step(int device, const egr_param_group *groups, int num_groups, uint32_t n, uint32_t step, double beta1, double beta2,
                                   double eps, void *hip_stream) {
    if (!groups || num_groups < 1 || num_groups > EGR_MAX_PARAM_GROUPS || step < 1) {
        g_step_error = "libegr_hip: egr_fused_adam_step: 1..EGR_MAX_PARAM_GROUPS groups and a 1-based step are required";
        return 1;
    }
    if (n == 0) return 0;
    StepArgs a{};
    uint32_t wmax = 1;
    for (int k = 0; k < num_groups; k++) {
        a.g[k] = groups[k];
        if (!groups[k].param || groups[k].width == 0 || (groups[k].exp_avg == nullptr) != (groups[k].exp_avg_sq == nullptr)) {
            g_step_error = "libegr_hip: egr_fused_adam_step: group without parameter / width, or with only one Adam moment";
            return 1;
        }
        const double t = (double)(groups[k].step ? groups[k].step : step); // a group whose optimizer state was re-created counts from its own 1
        a.step_size[k] = (float)((double)groups[k].lr / (1.0 - std::pow(beta1, t)));
        a.bc2_sqrt[k] = (float)std::sqrt(1.0 - std::pow(beta2, t));
        wmax = std::max(wmax, groups[k].width);
    }
    a.w1 = (float)(1.0 - beta1), a.beta2 = (float)beta2, a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.n = n;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        const size_t total = (size_t)n * wmax;
        const uint32_t bx = (uint32_t)std::min<size_t>((total + 255) / 256, 65535u * 16u);
        hipLaunchKernelGGL(k_fused_step, dim3(bx, (uint32_t)num_groups), dim3(256), 0, (hipStream_t)hip_stream, a);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        g_step_error = std::string("libegr_hip: egr_fused_adam_step: ") + hipGetErrorString(e);
        return 1;
    }
    return 0;
}
