// The reference's compiled `_C` module, rebuilt over the C ABI of include/fr_rasterizer.h.
//
// Exports exactly what submodules/diff-gaussian-rasterization/ext.cpp:15-19 exports, with the signatures of
// rasterize_points.h:18-67:
//     rasterize_gaussians            RasterizeGaussiansCUDA            rasterize_points.cu:35-115
//     rasterize_gaussians_backward   RasterizeGaussiansBackwardCUDA    rasterize_points.cu:117-196
//     mark_visible                   markVisible                       rasterize_points.cu:198-217
// plus distCUDA2 (simple-knn ext.cpp / spatial.cu:14-25).  The bodies are torch glue only — allocate the outputs and
// the three opaque byte buffers with the caching allocator, pass data pointers and the current HIP stream to
// libfr_hip.so.  Where the reference calls CudaRasterizer::Rasterizer::{forward,backward,markVisible} this calls
// fr_forward / fr_backward / fr_mark_visible.  Built by __graft_entry__.build() (torch.utils.cpp_extension, host
// compiler only: there is no device code in this file).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>   // PyTorch-ROCm presents its HIP devices as device type "cuda"
#include <c10/core/DeviceGuard.h>
#include <algorithm>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include "fr_rasterizer.h"

namespace {

struct DeviceState {
    fr_handle* handle = nullptr;
    uint64_t capacity = 0;  // high-water mark of the binning capacity (instances)
};

std::mutex g_mutex;
std::unordered_map<int, DeviceState> g_state;

DeviceState& state_of(int device)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState& s = g_state[device];
    if (!s.handle) TORCH_CHECK(fr_create(&s.handle) == FR_OK, "fr_create failed: ", fr_last_error());
    return s;
}

// empty tensor -> null pointer, like data_ptr() of the reference's empty placeholder tensors (kernels branch on it)
const float* fptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

torch::Tensor f32c(const torch::Tensor& t)
{
    if (t.numel() == 0) return t;
    TORCH_CHECK(t.is_cuda(), "fateavatar_amd rasterizer: tensors must be on a HIP device (torch device 'cuda'); there is no CPU path");
    return t.to(torch::kFloat32).contiguous();
}

}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                       const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                       const torch::Tensor& campos, const bool prefiltered, const bool debug)
{
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    TORCH_CHECK(means3D.is_cuda(), "fateavatar_amd rasterizer: tensors must be on a HIP device (torch device 'cuda'); there is no CPU path");
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    c10::DeviceGuard guard(means3D.device());
    auto f32 = means3D.options().dtype(torch::kFloat32);
    auto u8 = means3D.options().dtype(torch::kByte);
    torch::Tensor out_color = torch::empty({3, H, W}, f32);
    torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
    if (P == 0) {  // rasterize_points.cu:81 skips the rasterizer
        out_color.zero_();
        return std::make_tuple(0, out_color, radii, torch::empty({0}, u8), torch::empty({0}, u8), torch::empty({0}, u8));
    }
    const torch::Tensor bg = f32c(background), m3 = f32c(means3D), col = f32c(colors), op = f32c(opacity), sc = f32c(scales),
                        rot = f32c(rotations), cov = f32c(cov3D_precomp), view = f32c(viewmatrix), proj = f32c(projmatrix),
                        shs = f32c(sh), cam = f32c(campos);
    const int M = shs.numel() ? (int)shs.size(1) : 0;
    DeviceState& st = state_of(means3D.get_device());
    torch::Tensor geom = torch::empty({(int64_t)fr_geometry_bytes(P)}, u8);
    torch::Tensor img = torch::empty({(int64_t)fr_image_bytes(W, H)}, u8);
    fr_params prm{P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, prefiltered ? 1 : 0, debug ? 1 : 0, /*flags*/ 0, /*aux*/ nullptr};
    fr_inputs in{fptr(bg), fptr(m3), fptr(shs), fptr(col), fptr(op), fptr(sc), fptr(rot), fptr(cov), fptr(view), fptr(proj), fptr(cam)};
    uint64_t cap = std::max<uint64_t>(st.capacity, 4ull * (uint64_t)P + 65536ull);
    fr_counts counts{};
    torch::Tensor binning;
    void* stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(means3D.get_device()).stream();
    for (;;) {
        binning = torch::empty({(int64_t)fr_binning_bytes(cap, W, H)}, u8);
        const int rc = fr_forward(st.handle, &prm, &in, out_color.data_ptr<float>(), radii.data_ptr<int>(), geom.data_ptr(),
                                  img.data_ptr(), binning.data_ptr(), cap, &counts, stream);
        if (rc == FR_ERR_BINNING_CAPACITY) {
            cap = (uint64_t)counts.num_instances * 5 / 4 + 1024;
            continue;
        }
        TORCH_CHECK(rc == FR_OK, "fr_forward failed (code ", rc, "): ", fr_last_error());
        break;
    }
    st.capacity = std::max<uint64_t>(st.capacity, (uint64_t)counts.num_instances * 5 / 4 + 1024);
    return std::make_tuple((int)counts.num_rendered, out_color, radii, geom, binning, img);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                               const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const bool debug)
{
    (void)R;  // the binning layout is re-derived from the counts the forward left in the image buffer
    TORCH_CHECK(means3D.is_cuda(), "fateavatar_amd rasterizer: tensors must be on a HIP device (torch device 'cuda'); there is no CPU path");
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    c10::DeviceGuard guard(means3D.device());
    const torch::Tensor shs = f32c(sh);
    const int M = shs.numel() ? (int)shs.size(1) : 0;
    auto f32 = means3D.options().dtype(torch::kFloat32);
    // the kernels write every row of every array: no zero-fill (the reference needs torch::zeros here, :151-159)
    auto mk = [&](std::vector<int64_t> shape) { return P ? torch::empty(shape, f32) : torch::zeros(shape, f32); };
    torch::Tensor dL_dmeans3D = mk({P, 3}), dL_dmeans2D = mk({P, 3}), dL_dcolors = mk({P, 3}), dL_dconic = torch::Tensor(),
                  dL_dopacity = mk({P, 1}), dL_dcov3D = mk({P, 6}), dL_dsh = mk({P, M, 3}), dL_dscales = mk({P, 3}),
                  dL_drotations = mk({P, 4});
    if (P != 0) {
        const torch::Tensor bg = f32c(background), m3 = f32c(means3D), col = f32c(colors), sc = f32c(scales), rot = f32c(rotations),
                            cov = f32c(cov3D_precomp), view = f32c(viewmatrix), proj = f32c(projmatrix), cam = f32c(campos),
                            dpix = f32c(dL_dout_color);
        const torch::Tensor rad = radii.contiguous();
        DeviceState& st = state_of(means3D.get_device());
        fr_params prm{P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, 0, debug ? 1 : 0, 0, nullptr};
        fr_inputs in{fptr(bg), fptr(m3), fptr(shs), fptr(col), nullptr, fptr(sc), fptr(rot), fptr(cov), fptr(view), fptr(proj), fptr(cam)};
        fr_grads g{dL_dmeans2D.data_ptr<float>(), dL_dcolors.data_ptr<float>(), dL_dopacity.data_ptr<float>(),
                   dL_dmeans3D.data_ptr<float>(), dL_dcov3D.data_ptr<float>(), M ? dL_dsh.data_ptr<float>() : nullptr,
                   dL_dscales.data_ptr<float>(), dL_drotations.data_ptr<float>()};
        const int rc = fr_backward(st.handle, &prm, &in, rad.data_ptr<int>(), geomBuffer.data_ptr(), imageBuffer.data_ptr(),
                                   binningBuffer.data_ptr(), dpix.data_ptr<float>(), &g,
                                   c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(means3D.get_device()).stream());
        TORCH_CHECK(rc == FR_OK, "fr_backward failed (code ", rc, "): ", fr_last_error());
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix)
{
    TORCH_CHECK(means3D.is_cuda(), "fateavatar_amd rasterizer: tensors must be on a HIP device (torch device 'cuda'); there is no CPU path");
    const int P = (int)means3D.size(0);
    c10::DeviceGuard guard(means3D.device());
    torch::Tensor present = torch::full({P}, false, means3D.options().dtype(at::kBool));
    if (P != 0) {
        const torch::Tensor m3 = f32c(means3D), view = f32c(viewmatrix), proj = f32c(projmatrix);
        const int rc = fr_mark_visible(P, m3.data_ptr<float>(), view.data_ptr<float>(), proj.data_ptr<float>(),
                                       reinterpret_cast<uint8_t*>(present.data_ptr<bool>()),
                                       c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(means3D.get_device()).stream());
        TORCH_CHECK(rc == FR_OK, "fr_mark_visible failed: ", fr_last_error());
    }
    return present;
}

// simple-knn: distCUDA2 (spatial.cu:14-25)
torch::Tensor distCUDA2(const torch::Tensor& points)
{
    TORCH_CHECK(points.is_cuda(), "simple_knn: points must be on a HIP device");
    const int P = (int)points.size(0);
    c10::DeviceGuard guard(points.device());
    const torch::Tensor pts = points.to(torch::kFloat32).contiguous();
    torch::Tensor means = torch::full({P}, 0.0, pts.options());
    if (P != 0) {
        torch::Tensor ws = torch::empty({(int64_t)fr_knn_workspace_bytes(P)}, pts.options().dtype(torch::kByte));
        const int rc = fr_knn_mean_dist2(P, pts.data_ptr<float>(), means.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(),
                                         c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(points.get_device()).stream());
        TORCH_CHECK(rc == FR_OK, "fr_knn_mean_dist2 failed: ", fr_last_error());
    }
    return means;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("rasterize_gaussians", &RasterizeGaussiansCUDA);
    m.def("rasterize_gaussians_backward", &RasterizeGaussiansBackwardCUDA);
    m.def("mark_visible", &markVisible);
    m.def("distCUDA2", &distCUDA2);
}
