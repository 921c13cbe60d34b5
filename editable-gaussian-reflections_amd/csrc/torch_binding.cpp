// libraytracer.so -- TORCH_LIBRARY(raytracer, m) shim over the C ABI of libegr_hip.so.
//
// Registers the same eight TorchScript custom classes, with the same names, constructor signature, methods and
// read-only tensor attributes (names, shapes, dtypes, defaults) as the reference's
// editable_gauss_refl/cuda/csrc/raytracer.cpp:122-218 and core/*.h holders, so
//     torch.classes.load_library(".../libraytracer.so"); torch.classes.raytracer.Raytracer(W, H, N, fwd, bwd)
// works unchanged (editable_gauss_refl/__init__.py:15-27). This file only owns tensors and forwards raw device
// pointers; all arithmetic is in the HIP library. There is no CPU fallback: construction fails without a GPU.
//
// "torch::kCUDA" below is PyTorch's device name for HIP devices on ROCm builds, not a compatibility layer.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/custom_class.h>
#include <torch/library.h>
#include <torch/torch.h>

#include <string>
#include <tuple>
#include <vector>

#include "../../include/egr_raytracer.h"

using at::Tensor;

namespace {
inline at::TensorOptions F32() { return torch::dtype(torch::kFloat32).device(torch::kCUDA); }
inline at::TensorOptions I32() { return torch::dtype(torch::kInt32).device(torch::kCUDA); }
inline at::TensorOptions B8() { return torch::dtype(torch::kBool).device(torch::kCUDA); }
inline void *current_stream() { return (void *)c10::hip::getCurrentHIPStream().stream(); }
template <class T> T *ptr(const Tensor &t) { return reinterpret_cast<T *>(t.data_ptr()); }
} // namespace

// core/camera.h:43-77
struct CameraDataHolder : torch::CustomClassHolder {
    Tensor origin = torch::zeros({3}, F32());
    Tensor vertical_fov_radians = torch::zeros({1}, F32());
    Tensor rotation_c2w = torch::zeros({3, 3}, F32());
    Tensor rotation_w2c = torch::zeros({3, 3}, F32());
    Tensor znear = torch::zeros({1}, F32());
    Tensor zfar = torch::zeros({1}, F32());
    egr_camera reify() {
        return egr_camera{ptr<float>(origin), ptr<float>(vertical_fov_radians), ptr<float>(rotation_c2w), ptr<float>(rotation_w2c),
                          ptr<float>(znear), ptr<float>(zfar)};
    }
    void set_pose(const Tensor &c2w_origin, const Tensor &c2w_rotation) { // camera.h:62-68
        TORCH_CHECK(c2w_rotation.sizes() == torch::IntArrayRef({3, 3}), "c2w_rotation must be 3x3");
        TORCH_CHECK(c2w_origin.sizes() == torch::IntArrayRef({3}), "c2w_origin must be 3");
        rotation_c2w.copy_(c2w_rotation);
        origin.copy_(c2w_origin);
        rotation_w2c.copy_(c2w_rotation.transpose(0, 1));
    }
    static void bind(torch::Library &m) {
        m.class_<CameraDataHolder>("CameraDataHolder")
            .def("set_pose", &CameraDataHolder::set_pose)
            .def_readonly("vertical_fov_radians", &CameraDataHolder::vertical_fov_radians)
            .def_readonly("znear", &CameraDataHolder::znear)
            .def_readonly("zfar", &CameraDataHolder::zfar);
    }
};

// core/config.h:31-101 (defaults :32-51)
struct ConfigDataHolder : torch::CustomClassHolder {
    Tensor exp_power = torch::tensor({3.0f}, F32());
    Tensor alpha_threshold = torch::tensor({0.005f}, F32());
    Tensor transmittance_threshold = torch::tensor({0.01f}, F32());
    Tensor accumulate_samples = torch::zeros({1}, B8());
    Tensor jitter_primary_rays = torch::ones({1}, B8());
    Tensor num_bounces = torch::full({1}, 2, I32());
    Tensor global_scale_factor = torch::ones({1}, F32());
    Tensor loss_weight_diffuse = torch::ones({1}, F32());
    Tensor loss_weight_specular = torch::ones({1}, F32());
    Tensor loss_weight_depth = torch::ones({1}, F32());
    Tensor loss_weight_normal = torch::ones({1}, F32());
    Tensor loss_weight_f0 = torch::ones({1}, F32());
    Tensor loss_weight_roughness = torch::ones({1}, F32());
    Tensor eps_forward_normalization = torch::tensor({1e-12f}, F32());
    Tensor eps_scale_grad = torch::tensor({1e-12f}, F32());
    Tensor eps_ray_surface_offset = torch::tensor({0.01f}, F32());
    Tensor eps_min_roughness = torch::tensor({0.01f}, F32());
    Tensor reflection_invalid_normal_threshold = torch::tensor({0.7f}, F32());
    Tensor backfacing_invalid_normal_threshold = torch::tensor({0.9f}, F32());
    Tensor backfacing_max_dist = torch::tensor({0.1f}, F32());
    egr_config reify() {
        egr_config c;
        c.exp_power = ptr<float>(exp_power), c.alpha_threshold = ptr<float>(alpha_threshold);
        c.transmittance_threshold = ptr<float>(transmittance_threshold);
        c.accumulate_samples = ptr<uint8_t>(accumulate_samples), c.jitter_primary_rays = ptr<uint8_t>(jitter_primary_rays);
        c.num_bounces = ptr<int32_t>(num_bounces), c.global_scale_factor = ptr<float>(global_scale_factor);
        c.loss_weight_diffuse = ptr<float>(loss_weight_diffuse), c.loss_weight_specular = ptr<float>(loss_weight_specular);
        c.loss_weight_depth = ptr<float>(loss_weight_depth), c.loss_weight_normal = ptr<float>(loss_weight_normal);
        c.loss_weight_f0 = ptr<float>(loss_weight_f0), c.loss_weight_roughness = ptr<float>(loss_weight_roughness);
        c.eps_forward_normalization = ptr<float>(eps_forward_normalization), c.eps_scale_grad = ptr<float>(eps_scale_grad);
        c.eps_ray_surface_offset = ptr<float>(eps_ray_surface_offset), c.eps_min_roughness = ptr<float>(eps_min_roughness);
        c.reflection_invalid_normal_threshold = ptr<float>(reflection_invalid_normal_threshold);
        c.backfacing_invalid_normal_threshold = ptr<float>(backfacing_invalid_normal_threshold);
        c.backfacing_max_dist = ptr<float>(backfacing_max_dist);
        return c;
    }
    static void bind(torch::Library &m) {
        m.class_<ConfigDataHolder>("ConfigDataHolder")
            .def_readonly("exp_power", &ConfigDataHolder::exp_power)
            .def_readonly("alpha_threshold", &ConfigDataHolder::alpha_threshold)
            .def_readonly("transmittance_threshold", &ConfigDataHolder::transmittance_threshold)
            .def_readonly("accumulate_samples", &ConfigDataHolder::accumulate_samples)
            .def_readonly("jitter_primary_rays", &ConfigDataHolder::jitter_primary_rays)
            .def_readonly("num_bounces", &ConfigDataHolder::num_bounces)
            .def_readonly("global_scale_factor", &ConfigDataHolder::global_scale_factor)
            .def_readonly("loss_weight_diffuse", &ConfigDataHolder::loss_weight_diffuse)
            .def_readonly("loss_weight_specular", &ConfigDataHolder::loss_weight_specular)
            .def_readonly("loss_weight_depth", &ConfigDataHolder::loss_weight_depth)
            .def_readonly("loss_weight_normal", &ConfigDataHolder::loss_weight_normal)
            .def_readonly("loss_weight_f0", &ConfigDataHolder::loss_weight_f0)
            .def_readonly("loss_weight_roughness", &ConfigDataHolder::loss_weight_roughness)
            .def_readonly("eps_forward_normalization", &ConfigDataHolder::eps_forward_normalization)
            .def_readonly("eps_scale_grad", &ConfigDataHolder::eps_scale_grad)
            .def_readonly("eps_ray_surface_offset", &ConfigDataHolder::eps_ray_surface_offset)
            .def_readonly("eps_min_roughness", &ConfigDataHolder::eps_min_roughness)
            .def_readonly("reflection_invalid_normal_threshold", &ConfigDataHolder::reflection_invalid_normal_threshold)
            .def_readonly("backfacing_invalid_normal_threshold", &ConfigDataHolder::backfacing_invalid_normal_threshold)
            .def_readonly("backfacing_max_dist", &ConfigDataHolder::backfacing_max_dist);
    }
};

// core/framebuffer.h:159-288
struct FramebufferDataHolder : torch::CustomClassHolder {
    Tensor output_rgb, output_depth, output_normal, output_f0, output_roughness, output_transmittance, output_total_transmittance,
        output_ray_origin, output_ray_direction, output_final, output_denoised;
    Tensor accumulated_rgb, accumulated_transmittance, accumulated_total_transmittance, accumulated_depth, accumulated_normal,
        accumulated_f0, accumulated_roughness, accumulated_sample_count;
    Tensor target_diffuse, target_specular, target_depth, target_normal, target_f0, target_roughness;
    FramebufferDataHolder(int64_t w, int64_t h) {
        const int64_t S = EGR_NUM_STEPS;
        auto z = [&](int64_t lead, int64_t c) { return torch::zeros({lead, h, w, c}, F32()); };
        output_rgb = z(S, 3), output_depth = z(S, 1), output_normal = z(S, 3), output_f0 = z(S, 3), output_roughness = z(S, 1);
        output_transmittance = z(S, 1), output_total_transmittance = z(S, 1), output_ray_origin = z(S, 3), output_ray_direction = z(S, 3);
        output_final = z(1, 3), output_denoised = z(1, 3);
        accumulated_rgb = z(S, 3), accumulated_transmittance = z(S, 1), accumulated_total_transmittance = z(S, 1);
        accumulated_depth = z(S, 1), accumulated_normal = z(S, 3), accumulated_f0 = z(S, 3), accumulated_roughness = z(S, 1);
        accumulated_sample_count = torch::zeros({1}, I32());
        auto t = [&](int64_t c) { return torch::zeros({h, w, c}, F32()); };
        target_diffuse = t(3), target_specular = t(3), target_depth = t(1), target_normal = t(3), target_f0 = t(3), target_roughness = t(1);
    }
    egr_framebuffer reify() {
        egr_framebuffer f;
        f.output_rgb = ptr<float>(output_rgb), f.output_depth = ptr<float>(output_depth), f.output_normal = ptr<float>(output_normal);
        f.output_f0 = ptr<float>(output_f0), f.output_roughness = ptr<float>(output_roughness);
        f.output_transmittance = ptr<float>(output_transmittance), f.output_total_transmittance = ptr<float>(output_total_transmittance);
        f.output_ray_origin = ptr<float>(output_ray_origin), f.output_ray_direction = ptr<float>(output_ray_direction);
        f.output_final = ptr<float>(output_final), f.output_denoised = ptr<float>(output_denoised);
        f.accumulated_rgb = ptr<float>(accumulated_rgb), f.accumulated_transmittance = ptr<float>(accumulated_transmittance);
        f.accumulated_total_transmittance = ptr<float>(accumulated_total_transmittance), f.accumulated_depth = ptr<float>(accumulated_depth);
        f.accumulated_normal = ptr<float>(accumulated_normal), f.accumulated_f0 = ptr<float>(accumulated_f0);
        f.accumulated_roughness = ptr<float>(accumulated_roughness), f.accumulated_sample_count = ptr<int32_t>(accumulated_sample_count);
        f.target_diffuse = ptr<float>(target_diffuse), f.target_specular = ptr<float>(target_specular), f.target_depth = ptr<float>(target_depth);
        f.target_normal = ptr<float>(target_normal), f.target_f0 = ptr<float>(target_f0), f.target_roughness = ptr<float>(target_roughness);
        return f;
    }
    void reset_accumulators() { // framebuffer.h:249-258
        accumulated_rgb.zero_(), accumulated_transmittance.zero_(), accumulated_total_transmittance.zero_(), accumulated_depth.zero_();
        accumulated_normal.zero_(), accumulated_f0.zero_(), accumulated_roughness.zero_(), accumulated_sample_count.zero_();
    }
    static void bind(torch::Library &m) {
        m.class_<FramebufferDataHolder>("Framebuffer")
            .def_readonly("output_rgb", &FramebufferDataHolder::output_rgb)
            .def_readonly("output_depth", &FramebufferDataHolder::output_depth)
            .def_readonly("output_normal", &FramebufferDataHolder::output_normal)
            .def_readonly("output_f0", &FramebufferDataHolder::output_f0)
            .def_readonly("output_roughness", &FramebufferDataHolder::output_roughness)
            .def_readonly("output_transmittance", &FramebufferDataHolder::output_transmittance)
            .def_readonly("output_total_transmittance", &FramebufferDataHolder::output_total_transmittance)
            .def_readonly("output_ray_origin", &FramebufferDataHolder::output_ray_origin)
            .def_readonly("output_ray_direction", &FramebufferDataHolder::output_ray_direction)
            .def_readonly("output_final", &FramebufferDataHolder::output_final)
            .def_readonly("output_denoised", &FramebufferDataHolder::output_denoised)
            .def_readonly("accumulated_rgb", &FramebufferDataHolder::accumulated_rgb)
            .def_readonly("accumulated_transmittance", &FramebufferDataHolder::accumulated_transmittance)
            .def_readonly("accumulated_total_transmittance", &FramebufferDataHolder::accumulated_total_transmittance)
            .def_readonly("accumulated_depth", &FramebufferDataHolder::accumulated_depth)
            .def_readonly("accumulated_normal", &FramebufferDataHolder::accumulated_normal)
            .def_readonly("accumulated_f0", &FramebufferDataHolder::accumulated_f0)
            .def_readonly("accumulated_roughness", &FramebufferDataHolder::accumulated_roughness)
            .def_readonly("accumulated_sample_count", &FramebufferDataHolder::accumulated_sample_count)
            .def_readonly("target_diffuse", &FramebufferDataHolder::target_diffuse)
            .def_readonly("target_specular", &FramebufferDataHolder::target_specular)
            .def_readonly("target_depth", &FramebufferDataHolder::target_depth)
            .def_readonly("target_normal", &FramebufferDataHolder::target_normal)
            .def_readonly("target_f0", &FramebufferDataHolder::target_f0)
            .def_readonly("target_roughness", &FramebufferDataHolder::target_roughness);
    }
};

// core/gaussians.h:30-135. The nine gradient tensors are windows into ONE contiguous buffer (grad_flat, 22 floats
// per Gaussian, tensor-major) so that multi-GPU training needs a single all-reduce; each keeps the reference's
// shape and is installed as .grad of its parameter (gaussians.h:54-61).
struct GaussianDataHolder : torch::CustomClassHolder {
    int64_t count = 1;
    Tensor rgb = torch::zeros({1, 3}, F32()), normal = torch::zeros({1, 3}, F32()), f0 = torch::zeros({1, 3}, F32());
    Tensor roughness = torch::zeros({1, 1}, F32()), opacity = torch::zeros({1, 1}, F32()), scale = torch::zeros({1, 3}, F32());
    Tensor mean = torch::zeros({1, 3}, F32()), rotation = torch::zeros({1, 4}, F32());
    Tensor grad_flat = torch::zeros({22}, F32());
    Tensor dL_drgb = torch::empty({0}, F32()), dL_dnormal = torch::empty({0}, F32()), dL_df0 = torch::empty({0}, F32());
    Tensor dL_droughness = torch::empty({0}, F32()), dL_dopacity = torch::empty({0}, F32()), dL_dscale = torch::empty({0}, F32());
    Tensor dL_dmean = torch::empty({0}, F32()), dL_drotation = torch::empty({0}, F32()), total_weight = torch::empty({0}, F32());

    void point_grads() {
        int64_t off = 0;
        auto win = [&](Tensor &t, int64_t c) {
            t.set_(grad_flat.storage(), off, {count, c}, {c, 1}); // same TensorImpl, new window
            off += count * c;
        };
        win(dL_drgb, 3), win(dL_dnormal, 3), win(dL_df0, 3), win(dL_droughness, 1), win(dL_dopacity, 1);
        win(dL_dscale, 3), win(dL_dmean, 3), win(dL_drotation, 4), win(total_weight, 1);
    }
    GaussianDataHolder() {
        torch::NoGradGuard no_grad;
        point_grads();
        rgb.mutable_grad() = dL_drgb, normal.mutable_grad() = dL_dnormal, f0.mutable_grad() = dL_df0;
        roughness.mutable_grad() = dL_droughness, opacity.mutable_grad() = dL_dopacity, scale.mutable_grad() = dL_dscale;
        mean.mutable_grad() = dL_dmean, rotation.mutable_grad() = dL_drotation;
    }
    // Multi-GPU (not in the reference): with `use_delta` a grad launch STORES its gradients and weights in a [22N] buffer of its own
    // (grad_delta; egr_set_grad_overwrite: rows no ray touched read 0, nobody has to clear it) instead of adding them to the persistent
    // one, so the caller can all-reduce exactly one launch's contribution and then fold it in (renderer.py: all_reduce_grads) - summing
    // the persistent buffer would multiply whatever it already holds (total_weight across a pruning interval, accumulated gradients)
    // by the world size on every iteration.
    Tensor grad_delta = torch::empty({0}, F32());
    bool use_delta = false;
    void set_use_delta(bool on) {
        use_delta = on;
        grad_delta = on ? torch::zeros({22 * count}, F32()) : torch::empty({0}, F32());
    }
    void resize(int64_t n) { // gaussians.h:64-86: resize_ keeps the first min(old, n) rows of every tensor; grown gradient memory is zeroed here
        torch::NoGradGuard no_grad;
        const int64_t keep = std::min(count, n);
        Tensor old_rows[9] = {dL_drgb.narrow(0, 0, keep).clone(), dL_dnormal.narrow(0, 0, keep).clone(), dL_df0.narrow(0, 0, keep).clone(),
                              dL_droughness.narrow(0, 0, keep).clone(), dL_dopacity.narrow(0, 0, keep).clone(), dL_dscale.narrow(0, 0, keep).clone(),
                              dL_dmean.narrow(0, 0, keep).clone(), dL_drotation.narrow(0, 0, keep).clone(), total_weight.narrow(0, 0, keep).clone()};
        count = n;
        rgb.resize_({n, 3}), normal.resize_({n, 3}), f0.resize_({n, 3}), roughness.resize_({n, 1}), opacity.resize_({n, 1});
        scale.resize_({n, 3}), mean.resize_({n, 3}), rotation.resize_({n, 4});
        grad_flat = torch::zeros({22 * n}, F32());
        point_grads();
        Tensor *win[9] = {&dL_drgb, &dL_dnormal, &dL_df0, &dL_droughness, &dL_dopacity, &dL_dscale, &dL_dmean, &dL_drotation, &total_weight};
        for (int k = 0; k < 9 && keep > 0; k++) win[k]->narrow(0, 0, keep).copy_(old_rows[k]);
        if (use_delta) grad_delta = torch::zeros({22 * n}, F32());
    }
    egr_gaussians reify() {
        egr_gaussians g;
        g.count = (uint32_t)count;
        g.rgb = ptr<float>(rgb), g.normal = ptr<float>(normal), g.f0 = ptr<float>(f0), g.roughness = ptr<float>(roughness);
        g.opacity = ptr<float>(opacity), g.scale = ptr<float>(scale), g.mean = ptr<float>(mean), g.rotation = ptr<float>(rotation);
        float *base = use_delta ? ptr<float>(grad_delta) : ptr<float>(grad_flat); // same tensor-major layout as point_grads()
        const size_t n = (size_t)count;
        g.dL_drgb = base, g.dL_dnormal = base + 3 * n, g.dL_df0 = base + 6 * n, g.dL_droughness = base + 9 * n, g.dL_dopacity = base + 10 * n;
        g.dL_dscale = base + 11 * n, g.dL_dmean = base + 14 * n, g.dL_drotation = base + 17 * n, g.total_weight = base + 21 * n;
        return g;
    }
    static void bind(torch::Library &m) {
        m.class_<GaussianDataHolder>("GaussianDataHolder")
            .def_readonly("rgb", &GaussianDataHolder::rgb)
            .def_readonly("normal", &GaussianDataHolder::normal)
            .def_readonly("f0", &GaussianDataHolder::f0)
            .def_readonly("roughness", &GaussianDataHolder::roughness)
            .def_readonly("opacity", &GaussianDataHolder::opacity)
            .def_readonly("scale", &GaussianDataHolder::scale)
            .def_readonly("mean", &GaussianDataHolder::mean)
            .def_readonly("rotation", &GaussianDataHolder::rotation)
            .def_readonly("dL_drgb", &GaussianDataHolder::dL_drgb)
            .def_readonly("dL_dnormal", &GaussianDataHolder::dL_dnormal)
            .def_readonly("dL_df0", &GaussianDataHolder::dL_df0)
            .def_readonly("dL_droughness", &GaussianDataHolder::dL_droughness)
            .def_readonly("dL_dopacity", &GaussianDataHolder::dL_dopacity)
            .def_readonly("dL_dscale", &GaussianDataHolder::dL_dscale)
            .def_readonly("dL_dmean", &GaussianDataHolder::dL_dmean)
            .def_readonly("dL_drotation", &GaussianDataHolder::dL_drotation)
            .def_readonly("total_weight", &GaussianDataHolder::total_weight)
            .def_readonly("grad_flat", &GaussianDataHolder::grad_flat)    // addition: [22N] view of all of the above
            .def_readonly("grad_delta", &GaussianDataHolder::grad_delta); // addition: per-launch gradient buffer (empty unless use_grad_delta): stored by the first grad launch after grad_delta_consumed(), added to by further ones
    }
};

// core/metadata.h:12-39
struct MetaDataHolder : torch::CustomClassHolder {
    Tensor grads_enabled = torch::ones({1}, B8());
    Tensor total_num_calls = torch::zeros({1}, I32());
    Tensor random_seeds;
    MetaDataHolder(int64_t w, int64_t h) { random_seeds = torch::randint(0, 1000000000, {h, w, 1}, I32()); }
    egr_metadata reify() { return egr_metadata{ptr<uint8_t>(grads_enabled), ptr<int32_t>(total_num_calls), ptr<int32_t>(random_seeds)}; }
    static void bind(torch::Library &m) {
        m.class_<MetaDataHolder>("MetaDataHolder")
            .def_readonly("grads_enabled", &MetaDataHolder::grads_enabled)
            .def_readonly("total_num_calls", &MetaDataHolder::total_num_calls)
            .def_readonly("random_seeds", &MetaDataHolder::random_seeds);
    }
};

// core/stats.h:11-37
struct StatsDataHolder : torch::CustomClassHolder {
    Tensor num_accumulated_per_pixel, num_traversed_per_pixel;
    StatsDataHolder(int64_t w, int64_t h) {
        num_accumulated_per_pixel = torch::zeros({h, w}, I32());
        num_traversed_per_pixel = torch::zeros({h, w}, I32());
    }
    egr_stats reify() { return egr_stats{ptr<int32_t>(num_accumulated_per_pixel), ptr<int32_t>(num_traversed_per_pixel)}; }
    static void bind(torch::Library &m) {
        m.class_<StatsDataHolder>("StatsDataHolder")
            .def_readonly("num_accumulated_per_pixel", &StatsDataHolder::num_accumulated_per_pixel)
            .def_readonly("num_traversed_per_pixel", &StatsDataHolder::num_traversed_per_pixel);
    }
};

// core/per_pixel_linked_list.h:80-136. The HIP path has no global linked list; the class is kept (no Python
// caller reads its entries) with one-entry stubs so attribute access and NULL_PTR() keep working. `size` is
// remembered because it sizes the candidate scratch / hit arena instead.
struct PPLLDataHolder : torch::CustomClassHolder {
    int64_t requested_size;
    Tensor head_per_pixel, total_hits = torch::zeros({1}, I32()) - 1, gaussian_ids, distances, gaussvals, alphas, local_hits, transmittances,
                           previous_entries;
    PPLLDataHolder(int64_t w, int64_t h, int64_t size) : requested_size(size) {
        head_per_pixel = torch::full({h, w}, (int64_t)(int32_t)EGR_PPLL_NULL_PTR, I32());
        gaussian_ids = torch::zeros({1}, I32()), distances = torch::zeros({1}, F32()), gaussvals = torch::zeros({1}, F32());
        alphas = torch::zeros({1}, F32()), local_hits = torch::zeros({1, 3}, F32()), transmittances = torch::zeros({1}, F32());
        previous_entries = torch::zeros({1}, I32());
    }
    static void bind(torch::Library &m) {
        m.class_<PPLLDataHolder>("PPLLDataHolder")
            .def_readonly("head_per_pixel", &PPLLDataHolder::head_per_pixel)
            .def_readonly("total_hits", &PPLLDataHolder::total_hits)
            .def_readonly("gaussian_ids", &PPLLDataHolder::gaussian_ids)
            .def_readonly("distances", &PPLLDataHolder::distances)
            .def_readonly("alphas", &PPLLDataHolder::alphas)
            .def_readonly("transmittances", &PPLLDataHolder::transmittances)
            .def_readonly("local_hits", &PPLLDataHolder::local_hits)
            .def_readonly("gaussvals", &PPLLDataHolder::gaussvals)
            .def_readonly("previous_entries", &PPLLDataHolder::previous_entries)
            .def_static("NULL_PTR", []() { return (int64_t)EGR_PPLL_NULL_PTR; });
    }
};

// raytracer.cpp:24-205
struct Raytracer : torch::CustomClassHolder {
    int64_t width, height;
    c10::intrusive_ptr<CameraDataHolder> camera_data;
    c10::intrusive_ptr<ConfigDataHolder> config_data;
    c10::intrusive_ptr<FramebufferDataHolder> framebuffer_data;
    c10::intrusive_ptr<GaussianDataHolder> gaussian_data;
    c10::intrusive_ptr<MetaDataHolder> meta_data;
    c10::intrusive_ptr<StatsDataHolder> stats_data;
    c10::intrusive_ptr<PPLLDataHolder> ppll_forward_data, ppll_backward_data;
    egr_context *ctx = nullptr;
    Tensor pixel_mask; // debug_set_pixel_mask: the mask the context points at

    void check(int rc, const char *what) {
        if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (ctx ? egr_last_error(ctx) : "no context"));
    }

    Raytracer(int64_t width_, int64_t height_, int64_t num_gaussians, int64_t forward_ppl_size, int64_t backward_ppl_size)
        : width(width_), height(height_), camera_data(c10::make_intrusive<CameraDataHolder>()),
          config_data(c10::make_intrusive<ConfigDataHolder>()),
          framebuffer_data(c10::make_intrusive<FramebufferDataHolder>(width_, height_)),
          gaussian_data(c10::make_intrusive<GaussianDataHolder>()), meta_data(c10::make_intrusive<MetaDataHolder>(width_, height_)),
          stats_data(c10::make_intrusive<StatsDataHolder>(width_, height_)),
          ppll_forward_data(c10::make_intrusive<PPLLDataHolder>(width_, height_, forward_ppl_size)),
          ppll_backward_data(c10::make_intrusive<PPLLDataHolder>(width_, height_, backward_ppl_size)) {
        if (num_gaussians > 0) gaussian_data->resize(num_gaussians);
        int device = (int)camera_data->origin.get_device();
        int rc = egr_create(&ctx, device, (int)width, (int)height, forward_ppl_size, backward_ppl_size);
        if (rc != 0) throw std::runtime_error("raytracer: egr_create failed (no usable HIP device? there is no CPU fallback)");
        egr_camera cam = camera_data->reify();
        egr_config cfg = config_data->reify();
        egr_framebuffer fb = framebuffer_data->reify();
        egr_metadata md = meta_data->reify();
        egr_stats st = stats_data->reify();
        check(egr_bind(ctx, &cam, &cfg, &fb, &md, &st), "egr_bind");
        egr_gaussians g = gaussian_data->reify();
        check(egr_set_gaussians(ctx, &g), "egr_set_gaussians");
        // upstream builds the TLAS over `count` (zero-initialised) instances in the constructor (bvh_wrapper.h:17-22)
        check(egr_rebuild_bvh(ctx, current_stream()), "egr_rebuild_bvh");
    }
    ~Raytracer() override {
        if (ctx) egr_destroy(ctx);
    }

    void raytrace() { // raytracer.cpp:81-94
        check(egr_raytrace(ctx, torch::autograd::GradMode::is_enabled() ? 1 : 0, current_stream()), "raytrace");
    }
    void denoise() { check(egr_denoise(ctx, current_stream()), "denoise"); }
    void reset_accumulators() { framebuffer_data->reset_accumulators(); }
    // fuse_live (addition; default = the reference's update_bvh()): the pass also writes the live per-gaussian records and the NEXT raytrace()
    // skips its own pass over the cloud - for callers that run update_bvh() and raytrace() back to back (egr_update_bvh_ex)
    void update_bvh(bool fuse_live) { check(egr_update_bvh_ex(ctx, fuse_live ? EGR_UPDATE_FUSE_LIVE : 0u, current_stream()), "update_bvh"); }
    void rebuild_bvh() { check(egr_rebuild_bvh(ctx, current_stream()), "rebuild_bvh"); }
    void resize(int64_t n) { // raytracer.cpp:112-120
        gaussian_data->resize(n);
        egr_gaussians g = gaussian_data->reify();
        check(egr_set_gaussians(ctx, &g), "resize");
    }
    // ---- additions (not in the reference) ----
    void set_partition(int64_t rank, int64_t world) { check(egr_set_partition(ctx, (int)rank, (int)world), "set_partition"); }
    void use_grad_delta(bool on) { // see GaussianDataHolder::grad_delta
        gaussian_data->set_use_delta(on);
        egr_gaussians g = gaussian_data->reify();
        check(egr_set_gaussians(ctx, &g), "use_grad_delta");
        check(egr_set_grad_overwrite(ctx, on ? 1 : 0), "use_grad_delta"); // the first grad launch after grad_delta_consumed() STORES its sums in grad_delta (nobody clears it), further ones add
    }
    void grad_delta_consumed() { check(egr_grad_delta_consumed(ctx), "grad_delta_consumed"); } // the caller folded grad_delta into grad_flat: the next grad launch stores again
    // exact statistics: num_traversed_per_pixel / the candidate counters become the reference's intersection-program invocation
    // count (cube boxes, slower). Takes effect with the next update_bvh() / rebuild_bvh(); raytrace() refuses to run in between.
    void set_exact_stats(bool on) { check(egr_set_exact_stats(ctx, on ? 1 : 0), "set_exact_stats"); }
    // debug: trace only the pixels with mask != 0 ([H*W] or [H, W] uint8 CUDA tensor; an empty tensor clears the mask). The tensor is kept alive here.
    void debug_set_pixel_mask(Tensor mask) {
        if (mask.numel() == 0) {
            pixel_mask = Tensor();
            check(egr_debug_set_pixel_mask(ctx, nullptr), "debug_set_pixel_mask");
            return;
        }
        TORCH_CHECK(mask.is_cuda() && mask.scalar_type() == torch::kUInt8 && mask.numel() == width * height, "debug_set_pixel_mask: uint8 CUDA tensor with H*W elements expected");
        pixel_mask = mask.contiguous();
        check(egr_debug_set_pixel_mask(ctx, pixel_mask.data_ptr<uint8_t>()), "debug_set_pixel_mask");
    }
    // pose + scalars of a view in one launch (egr_set_camera_from_dataset): R = the dataset's c2w rotation [3,3], centre = camera_center [3], fp32 CUDA tensors
    void set_camera(Tensor R, Tensor centre, double fov, double znear, double zfar) {
        // (the reference's copy_ calls - gaussian_raytracer.py:98-100 - take tensors of any device: a CPU tensor is moved to this tracer's device first)
        TORCH_CHECK(R.dim() == 2 && R.size(0) == 3 && R.size(1) == 3 && centre.numel() == 3, "set_camera: tensors [3,3] and [3] expected");
        const auto dev = framebuffer_data->output_rgb.device();
        Tensor r = R.to(dev, torch::kFloat32).contiguous(), cc = centre.to(dev, torch::kFloat32).contiguous();
        check(egr_set_camera_from_dataset(ctx, r.data_ptr<float>(), cc.data_ptr<float>(), (float)fov, (float)znear, (float)zfar, current_stream()), "set_camera");
    }
    // the six target images of a training view, channel-major ([C,H,W] contiguous fp32 CUDA tensors; an undefined / empty tensor = absent = zeros), written into
    // the framebuffer's pixel-major target buffers for this context's own tiles in one launch (egr_set_targets_chw)
    void set_targets_chw(c10::optional<Tensor> diffuse, c10::optional<Tensor> specular, c10::optional<Tensor> depth, c10::optional<Tensor> normal, c10::optional<Tensor> roughness,
                         c10::optional<Tensor> f0) {
        const c10::optional<Tensor> *in[6] = {&diffuse, &specular, &depth, &normal, &roughness, &f0};
        const int64_t ch[6] = {3, 3, 1, 3, 1, 3};
        Tensor keep[6];
        const float *p[6];
        for (int b = 0; b < 6; b++) {
            p[b] = nullptr;
            if (!in[b]->has_value() || (*in[b])->numel() == 0) continue;
            // (the reference's `buf.copy_(val.moveaxis(0, -1))` raises on any other shape and accepts any device: same here - a [H,W,C] tensor must not be read as [C,H,W])
            const Tensor &t = **in[b];
            TORCH_CHECK(t.dim() == 3 && t.size(0) == ch[b] && t.size(1) == height && t.size(2) == width, "set_targets_chw: target ", b, " must be a [", ch[b], ",", height, ",", width, "] tensor (channel-major), got ", t.sizes());
            keep[b] = t.to(framebuffer_data->output_rgb.device(), torch::kFloat32).contiguous();
            p[b] = keep[b].data_ptr<float>();
        }
        check(egr_set_targets_chw(ctx, p[0], p[1], p[2], p[3], p[4], p[5], current_stream()), "set_targets_chw");
    }
    void set_rays_per_task(int64_t n) { TORCH_CHECK(egr_set_rays_per_task(ctx, (int)n) == 0, "set_rays_per_task: 0 (automatic), 16, 32 or 64 expected"); }
    void set_team_help(bool on) { TORCH_CHECK(egr_set_team_help(ctx, on ? 1 : 0) == 0, "set_team_help failed"); }
    void set_team_help_auto() { TORCH_CHECK(egr_set_team_help(ctx, -1) == 0, "set_team_help_auto failed"); } // on for under-filled ranks of a partition only (egr_set_team_help(-1))
    void set_strands(int64_t n) { TORCH_CHECK(egr_set_strands(ctx, (int)n) == 0, "set_strands: 1..EGR_STRANDS (value at creation) expected"); }
    std::vector<int64_t> get_counters() { // synchronises
        egr_counters c{};
        check(egr_get_counters(ctx, &c, current_stream()), "get_counters");
        // [rays0..2, candidates0..2, composited0..2, lifetime_rays, lifetime_launches, status, bvh_depth, bucket_records, device_bytes, arena_blocks_used, arena_blocks_cap,
        //  ext_blocks_used, ext_blocks_cap, accepted0..2]
        return {(int64_t)c.rays[0], (int64_t)c.rays[1], (int64_t)c.rays[2], (int64_t)c.candidates[0], (int64_t)c.candidates[1],
                (int64_t)c.candidates[2], (int64_t)c.composited[0], (int64_t)c.composited[1], (int64_t)c.composited[2],
                (int64_t)c.lifetime_rays, (int64_t)c.lifetime_launches, (int64_t)c.status, (int64_t)c.bvh_depth, (int64_t)c.bucket_records,
                (int64_t)c.device_bytes, (int64_t)c.arena_blocks_used, (int64_t)c.arena_blocks_cap, (int64_t)c.ext_blocks_used, (int64_t)c.ext_blocks_cap,
                (int64_t)c.accepted[0], (int64_t)c.accepted[1], (int64_t)c.accepted[2]};
    }
    void reset_lifetime_counters() { check(egr_reset_lifetime_counters(ctx, current_stream()), "reset_lifetime_counters"); }
    void enable_timing(bool on) { egr_enable_timing(ctx, on ? 1 : 0); }
    double last_raytrace_ms() { return egr_last_raytrace_ms(ctx); }
    double last_update_bvh_ms() { return egr_last_update_bvh_ms(ctx); }
    std::vector<std::tuple<std::string, double>> last_kernel_ms() {
        float ms[32];
        const char *names[32];
        int n = egr_last_kernel_ms(ctx, ms, names, 32);
        std::vector<std::tuple<std::string, double>> out;
        for (int i = 0; i < n; i++) out.emplace_back(std::string(names[i]), (double)ms[i]);
        return out;
    }
    Tensor debug_step_hits() { // [3,H,W] int32 on the host: composited hits per bounce step of the last grad launch
        Tensor t = torch::zeros({EGR_NUM_STEPS, height, width}, torch::kInt32);
        check(egr_debug_get_step_hits(ctx, t.data_ptr<int32_t>(), current_stream()), "debug_step_hits");
        return t;
    }
    Tensor debug_hit_sequence_hash() { // [3,H,W] int64 (the bits of the uint64 hashes) on the host: ordered composited gaussian ids per pixel and step of the last grad launch
        Tensor t = torch::zeros({EGR_NUM_STEPS, height, width}, torch::kInt64);
        check(egr_debug_get_hit_sequence_hash(ctx, reinterpret_cast<uint64_t *>(t.data_ptr<int64_t>()), current_stream()), "debug_hit_sequence_hash");
        return t;
    }
    int64_t check_bvh() { return egr_debug_check_bvh(ctx, current_stream()); }
    std::string last_error() { return egr_last_error(ctx); }
    std::vector<Tensor> debug_instances() {
        int64_t n = gaussian_data->count;
        Tensor M = torch::zeros({n, 3, 4}, torch::kFloat32), W = torch::zeros({n, 3, 4}, torch::kFloat32), A = torch::zeros({n, 6}, torch::kFloat32);
        check(egr_debug_get_instances(ctx, M.data_ptr<float>(), W.data_ptr<float>(), A.data_ptr<float>(), current_stream()), "debug_instances");
        return {M, W, A};
    }

    static void bind(torch::Library &m) {
        using Self = c10::intrusive_ptr<Raytracer>;
        m.class_<Raytracer>("Raytracer")
            .def(torch::init<int64_t, int64_t, int64_t, int64_t, int64_t>())
            .def("raytrace", &Raytracer::raytrace)
            .def("denoise", &Raytracer::denoise)
            .def("reset_accumulators", &Raytracer::reset_accumulators)
            .def("update_bvh", &Raytracer::update_bvh, "", {torch::arg("fuse_live") = false})
            .def("rebuild_bvh", &Raytracer::rebuild_bvh)
            .def("resize", &Raytracer::resize)
            .def("get_camera", [](const Self &self) { return self->camera_data; })
            .def("get_config", [](const Self &self) { return self->config_data; })
            .def("get_framebuffer", [](const Self &self) { return self->framebuffer_data; })
            .def("get_gaussians", [](const Self &self) { return self->gaussian_data; })
            .def("get_metadata", [](const Self &self) { return self->meta_data; })
            .def("get_stats", [](const Self &self) { return self->stats_data; })
            .def("get_ppll_forward_data", [](const Self &self) { return self->ppll_forward_data; })
            .def("get_ppll_backward_data", [](const Self &self) { return self->ppll_backward_data; })
            .def_static("MAX_BOUNCES", []() { return (int64_t)EGR_MAX_BOUNCES; })
            .def_static("MAX_ALPHA", []() { return (double)EGR_MAX_ALPHA; })
            .def_static("ROUGHNESS_DOWNWEIGHT_GRAD", []() { return (bool)EGR_ROUGHNESS_DOWNWEIGHT_GRAD; })
            .def_static("ROUGHNESS_DOWNWEIGHT_GRAD_POWER", []() { return (double)EGR_ROUGHNESS_DOWNWEIGHT_GRAD_POWER; })
            .def("describe_output_buffers", // raytracer.cpp:151-167
                 [](const Self &) {
                     std::vector<std::tuple<std::string, int64_t>> t = {
                         {"output_rgb", 3}, {"output_depth", 1}, {"output_normal", 3}, {"output_f0", 3}, {"output_roughness", 1},
                         {"output_transmittance", 1}, {"output_total_transmittance", 1}, {"output_brdf", 3}, {"output_ray_origin", 3},
                         {"output_ray_direction", 3}, {"output_final", 3}};
                     return t;
                 })
            .def("describe_accumulation_buffers", // :168-180
                 [](const Self &) {
                     std::vector<std::tuple<std::string, int64_t>> t = {
                         {"accumulated_rgb", 3}, {"accumulated_transmittance", 1}, {"accumulated_total_transmittance", 1},
                         {"accumulated_depth", 1}, {"accumulated_normal", 3}, {"accumulated_f0", 3}, {"accumulated_roughness", 1}};
                     return t;
                 })
            .def("describe_target_buffers", // :181-190
                 [](const Self &) {
                     std::vector<std::tuple<std::string, int64_t>> t = {
                         {"target_depth", 1}, {"target_normal", 3}, {"target_f0", 3}, {"target_roughness", 1}};
                     return t;
                 })
            .def("describe_gaussian_attributes", // :193-204
                 [](const Self &) {
                     std::vector<std::tuple<std::string, int64_t>> t = {
                         {"gaussian_rgb", 3}, {"gaussian_normal", 3}, {"gaussian_f0", 3}, {"gaussian_roughness", 1},
                         {"gaussian_opacity", 1}, {"gaussian_scale", 3}, {"gaussian_mean", 3}, {"gaussian_rotation", 4}};
                     return t;
                 })
            // additions for multi-GPU tile partitioning, measurement and tests
            .def("set_partition", &Raytracer::set_partition)
            .def("use_grad_delta", &Raytracer::use_grad_delta)
            .def("grad_delta_consumed", &Raytracer::grad_delta_consumed)
            .def("debug_set_pixel_mask", &Raytracer::debug_set_pixel_mask)
            .def("set_targets_chw", &Raytracer::set_targets_chw)
            .def("set_camera", &Raytracer::set_camera)
            .def("set_exact_stats", &Raytracer::set_exact_stats)
            .def("set_strands", &Raytracer::set_strands)
            .def("set_team_help", &Raytracer::set_team_help)
            .def("set_team_help_auto", &Raytracer::set_team_help_auto)
            .def("set_rays_per_task", &Raytracer::set_rays_per_task)
            .def("get_counters", &Raytracer::get_counters)
            .def("reset_lifetime_counters", &Raytracer::reset_lifetime_counters)
            .def("enable_timing", &Raytracer::enable_timing)
            .def("last_raytrace_ms", &Raytracer::last_raytrace_ms)
            .def("last_update_bvh_ms", &Raytracer::last_update_bvh_ms)
            .def("last_kernel_ms", &Raytracer::last_kernel_ms)
            .def("check_bvh", &Raytracer::check_bvh)
            .def("last_error", &Raytracer::last_error)
            .def("debug_instances", &Raytracer::debug_instances)
            .def("debug_step_hits", &Raytracer::debug_step_hits)
            .def("debug_hit_sequence_hash", &Raytracer::debug_hit_sequence_hash);
    }
};

// simple_knn._C.distCUDA2 (gaussian_model.py:17): float CUDA(HIP) tensor [N,3] -> [N] mean squared distance to the 3 nearest
// neighbours. Registered as torch.ops.simple_knn.distCUDA2; the package's simple_knn.py re-exports it under the reference's name.
static torch::Tensor dist_hip2(const torch::Tensor &points) {
    TORCH_CHECK(points.is_cuda(), "distCUDA2: points must live on the GPU (there is no CPU path)");
    TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "distCUDA2: expected an [N,3] tensor");
    auto p = points.to(torch::kFloat32).contiguous();
    auto out = torch::zeros({p.size(0)}, p.options());
    const int rc = egr_knn_mean_dist2(p.get_device(), p.data_ptr<float>(), (uint32_t)p.size(0), out.data_ptr<float>(), current_stream());
    TORCH_CHECK(rc == 0, egr_knn_last_error());
    return out;
}
// One launch for import + scale decay + Adam + clamps + zero_grad + export (csrc/step.hip). Tensor lists are parallel, one
// entry per parameter group; an undefined / empty tensor means "absent" for grads, rt_params, rt_grads and the Adam moments.
static void fused_adam_step(std::vector<torch::Tensor> params, std::vector<torch::Tensor> grads, std::vector<torch::Tensor> rt_params,
                            std::vector<torch::Tensor> rt_grads, std::vector<torch::Tensor> exp_avg, std::vector<torch::Tensor> exp_avg_sq,
                            std::vector<double> lrs, std::vector<double> clamp_min, std::vector<double> clamp_max, std::vector<double> log_decay,
                            int64_t step, double beta1, double beta2, double eps, std::vector<int64_t> group_steps) {
    const size_t G = params.size();
    TORCH_CHECK(G >= 1 && G <= EGR_MAX_PARAM_GROUPS, "fused_adam_step: 1..8 parameter groups");
    TORCH_CHECK(grads.size() == G && rt_params.size() == G && rt_grads.size() == G && exp_avg.size() == G && exp_avg_sq.size() == G && lrs.size() == G &&
                    clamp_min.size() == G && clamp_max.size() == G && log_decay.size() == G,
                "fused_adam_step: all lists need one entry per group");
    TORCH_CHECK(group_steps.empty() || group_steps.size() == G, "fused_adam_step: group_steps must be empty (every group uses `step`) or hold one count per group");
    egr_param_group g[EGR_MAX_PARAM_GROUPS];
    const int64_t n = params[0].size(0);
    auto ptr = [&](const torch::Tensor &t, const torch::Tensor &like, const char *what) -> float * {
        if (!t.defined() || t.numel() == 0) return nullptr;
        TORCH_CHECK(t.is_cuda() && t.scalar_type() == torch::kFloat32 && t.is_contiguous() && t.numel() == like.numel(), "fused_adam_step: ", what,
                    " must be a contiguous float GPU tensor of the parameter's size");
        return t.data_ptr<float>();
    };
    for (size_t k = 0; k < G; k++) {
        TORCH_CHECK(params[k].is_cuda() && params[k].scalar_type() == torch::kFloat32 && params[k].is_contiguous() && params[k].size(0) == n,
                    "fused_adam_step: parameters must be contiguous float GPU tensors with the same leading size");
        g[k].param = params[k].data_ptr<float>();
        g[k].grad = ptr(grads[k], params[k], "grad"), g[k].rt_param = ptr(rt_params[k], params[k], "rt_param");
        g[k].rt_grad = ptr(rt_grads[k], params[k], "rt_grad");
        g[k].exp_avg = ptr(exp_avg[k], params[k], "exp_avg"), g[k].exp_avg_sq = ptr(exp_avg_sq[k], params[k], "exp_avg_sq");
        g[k].width = n ? (uint32_t)(params[k].numel() / n) : 1u;
        g[k].lr = (float)lrs[k], g[k].clamp_min = (float)clamp_min[k], g[k].clamp_max = (float)clamp_max[k], g[k].log_decay = (float)log_decay[k];
        g[k].step = group_steps.size() == G ? (uint32_t)group_steps[k] : 0u;
    }
    const int rc = egr_fused_adam_step(params[0].get_device(), g, (int)G, (uint32_t)n, (uint32_t)step, beta1, beta2, eps, current_stream());
    TORCH_CHECK(rc == 0, egr_fused_step_last_error());
}

// unit-test hook (egr_debug_lean_arith): (a / b, sqrt(a)) as the hot kernels' division and square root compute them
static std::tuple<torch::Tensor, torch::Tensor> debug_lean_arith(const torch::Tensor &a, const torch::Tensor &b) {
    TORCH_CHECK(a.is_cuda() && b.is_cuda() && a.numel() == b.numel(), "debug_lean_arith: two GPU tensors of one size expected");
    auto x = a.to(torch::kFloat32).contiguous(), y = b.to(torch::kFloat32).contiguous();
    auto q = torch::zeros_like(x), r = torch::zeros_like(x);
    TORCH_CHECK(egr_debug_lean_arith(x.get_device(), x.data_ptr<float>(), y.data_ptr<float>(), q.data_ptr<float>(), r.data_ptr<float>(), (uint32_t)x.numel(), current_stream()) == 0, "debug_lean_arith failed");
    return {q, r};
}

TORCH_LIBRARY(simple_knn, m) { m.def("distCUDA2(Tensor points) -> Tensor", &dist_hip2); }
TORCH_LIBRARY(egr, m) {
    m.def("fused_adam_step(Tensor[] params, Tensor[] grads, Tensor[] rt_params, Tensor[] rt_grads, Tensor[] exp_avg, Tensor[] exp_avg_sq, float[] lrs, "
          "float[] clamp_min, float[] clamp_max, float[] log_decay, int step, float beta1, float beta2, float eps, int[] group_steps=[]) -> ()",
          &fused_adam_step);
    m.def("debug_lean_arith(Tensor a, Tensor b) -> (Tensor, Tensor)", &debug_lean_arith);
}

TORCH_LIBRARY(raytracer, m) {
    CameraDataHolder::bind(m);
    ConfigDataHolder::bind(m);
    FramebufferDataHolder::bind(m);
    GaussianDataHolder::bind(m);
    MetaDataHolder::bind(m);
    PPLLDataHolder::bind(m);
    StatsDataHolder::bind(m);
    Raytracer::bind(m);
}
