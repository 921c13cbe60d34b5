// k_step_epilogue: the per-ray step epilogue (egr_epilogue.hpp) as a kernel of its own, one thread per ray, once per bounce
// step. This translation unit is compiled with -ffp-contract=off (build.py); the header carries the same setting as a pragma.
#include <algorithm>

#include "egr_epilogue.hpp"

namespace {

__global__ void __launch_bounds__(EGR_WAVE) k_step_epilogue(DeviceView v, int step, int grads) {
    const int lane = threadIdx.x;
    const int num_bounces = min(*v.cfg.num_bounces, EGR_MAX_BOUNCES);
    if (step > num_bounces) return;
    for (uint32_t task = v.task_begin + blockIdx.x; task < v.task_begin + v.task_count; task += gridDim.x) {
        const TaskGeom tg = task_geom(v, task, lane);
        if (!tg.inside) continue;
        const StateRef S{v.state, v.state_stride, task * EGR_WAVE + (uint32_t)lane};
        step_epilogue_lane(v, step, grads != 0, num_bounces, tg, S);
    }
}

} // namespace

void egr_launch_step_epilogue(const DeviceView &v, int step, bool grads, hipStream_t s) {
    hipLaunchKernelGGL(k_step_epilogue, dim3(std::max(1u, std::min(v.task_count, 65535u))), dim3(EGR_WAVE), 0, s, v, step, grads ? 1 : 0);
}
