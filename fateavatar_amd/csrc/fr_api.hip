// extern "C" entry points declared in include/fr_rasterizer.h.  Argument checking, handle
// lifetime, error strings; the work is in fr_preprocess.hip / fr_blend.hip /
// fr_preprocess_bwd.hip / fr_knn.hip.
#include "fr_common.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace fr {

thread_local char g_err[512] = "";

int fail_hip(hipError_t e, const char* what, const char* file, int line)
{
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
    return FR_ERR_HIP;
}
int fail_msg(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

static int check_frame(const fr_params* prm, const fr_inputs* in, bool forward)
{
    if (!prm || !in) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null params/inputs");
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0) return fail_msg(FR_ERR_INVALID_ARGUMENT, "bad P/W/H");
    if (prm->W > 8 * 65535 || prm->H > 8 * 65535) return fail_msg(FR_ERR_UNSUPPORTED, "image larger than 65535 tiles per axis");
    if (prm->D < 0 || prm->D > 3) return fail_msg(FR_ERR_INVALID_ARGUMENT, "SH degree must be 0..3");
    if (prm->P == 0) return FR_OK;
    // opacities are only read by the forward (the backward takes them from the geometry state)
    if (!in->means3D || (forward && !in->opacities) || !in->viewmatrix || !in->projmatrix || !in->campos || !in->background)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "missing required input pointer");
    // exactly one of shs / colors_precomp, exactly one of (scales+rotations) / cov3D_precomp
    // (the reference enforces this in Python: diff_gaussian_rasterization/__init__.py:191-195)
    if ((in->shs == nullptr) == (in->colors_precomp == nullptr))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "provide exactly one of shs / colors_precomp");
    const bool sr = in->scales && in->rotations;
    if ((in->scales == nullptr) != (in->rotations == nullptr) || (sr == (in->cov3D_precomp != nullptr)))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "provide exactly one of scales+rotations / cov3D_precomp");
    if ((prm->flags & FR_FLAG_RAW_ACTIVATIONS) && !sr)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "FR_FLAG_RAW_ACTIVATIONS needs scales + rotations");
    if (in->shs && prm->M < (prm->D + 1) * (prm->D + 1))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "M smaller than (D+1)^2");
    if (prm->aux && prm->aux->binding) {
        const fr_binding& b = *prm->aux->binding;
        if (!(prm->flags & FR_FLAG_RAW_ACTIVATIONS) || !sr)
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_aux::binding needs FR_FLAG_RAW_ACTIVATIONS and scales + rotations");
        if (b.N != prm->P || !b.verts || !b.faces || !b.face_index || !b.bary || !b.offset || !b.rotation || !b.scaling ||
            (b.resize_scale && !b.face_scale_canonical))
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_aux::binding: N must equal P and every array must be given");
    }
    return FR_OK;
}

__global__ void __launch_bounds__(256) k_zero16(uint4* p, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

int launch_zero(void* ptr, size_t bytes, hipStream_t s)
{
    const size_t n16 = (bytes + 15) / 16;
    if (n16 == 0) return FR_OK;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_zero16, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint4*>(ptr), n16);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

bool note_capture(fr_handle_impl* h, hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &st);
    if (st != hipStreamCaptureStatusNone) h->captured = true;
    return st != hipStreamCaptureStatusNone;
}

int release_buffer(fr_handle_impl* h, void* p, hipStream_t s)
{
    if (!p) return FR_OK;
    if (h->captured) {   // a captured graph may replay on it: kept until fr_destroy
        h->retired.push_back(p);
        return FR_OK;
    }
    // (frames in flight still hold the old pointer in kernels already enqueued on this stream: free behind them)
    FR_HIP(hipStreamSynchronize(s));
    FR_HIP(hipFree(p));
    return FR_OK;
}

int ensure_accum(fr_handle_impl* h, size_t P, hipStream_t s)
{
    if (P <= h->accum_rows) return FR_OK;
    {
        int rc = release_buffer(h, h->accum, s);
        if (rc) return rc;
    }
    h->accum = nullptr, h->accum_rows = 0;
    const size_t rows = P + P / 4 + 1024;
    FR_HIP(hipMalloc(reinterpret_cast<void**>(&h->accum), rows * kAccumStride * sizeof(float)));
    h->accum_rows = rows;
    return launch_zero(h->accum, rows * kAccumStride * sizeof(float), s);
}

}  // namespace fr

using namespace fr;

extern "C" {

const char* fr_last_error(void) { return g_err; }
const char* fr_version(void) { return "fateavatar_amd rasterizer 0.1 (gfx950, wave64, 8x8 tiles)"; }

int fr_create(fr_handle** out)
{
    if (!out) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null out");
    fr_handle_impl* h = new (std::nothrow) fr_handle_impl();
    if (!h) return fail_msg(FR_ERR_HIP, "out of host memory");
    FR_HIP(hipGetDevice(&h->device));
    FR_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->host_counts), 64, hipHostMallocMapped));
    memset(h->host_counts, 0, 64);
    FR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->host_counts_dev), h->host_counts, 0));
    FR_HIP(hipEventCreateWithFlags(&h->counts_ready, hipEventDisableTiming));
    FR_HIP(hipEventCreateWithFlags(&h->frame_done, hipEventDisableTiming));
    FR_HIP(hipEventCreateWithFlags(&h->bwd_done, hipEventDisableTiming));
    FR_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    FR_HIP(hipEventCreateWithFlags(&h->side_fork, hipEventDisableTiming));
    FR_HIP(hipEventCreateWithFlags(&h->side_join, hipEventDisableTiming));
    const char* bf = getenv("FR_BLEND_FWD");
    h->gather_in_chain = !(bf && strcmp(bf, "gather") == 0);
    const char* pf = getenv("FR_DENSE_PAIRS_FWD");
    const char* pb = getenv("FR_DENSE_PAIRS_BWD");
    h->dense_pairs_fwd = pf ? (uint32_t)strtoul(pf, nullptr, 10) : kDensePairsFwd;
    h->dense_pairs_bwd = pb ? (uint32_t)strtoul(pb, nullptr, 10) : kDensePairsBwd;
    if (const char* hp = getenv("FR_HEAVY_PAIRS")) h->heavy_pairs = (uint32_t)strtoul(hp, nullptr, 10);
    if (const char* cs = getenv("FR_CHAIN_SPINS")) h->chain_spins = (uint32_t)strtoul(cs, nullptr, 10);
    const char* ph = getenv("FR_DEBUG_PAIR_HIST");
    h->debug_pair_hist = ph && ph[0] == '1';
    *out = reinterpret_cast<fr_handle*>(h);
    return FR_OK;
}

int fr_destroy(fr_handle* hh)
{
    fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(hh);
    if (!h) return FR_OK;
    (void)hipEventDestroy(h->counts_ready);
    if (h->frame_done) (void)hipEventDestroy(h->frame_done);
    if (h->bwd_done) (void)hipEventDestroy(h->bwd_done);
    if (h->side_fork) (void)hipEventDestroy(h->side_fork);
    if (h->side_join) (void)hipEventDestroy(h->side_join);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    for (int st = 0; st < ST_COUNT; st++)
        for (size_t i = 0; i < h->ev[st].start.size(); i++) {
            (void)hipEventDestroy(h->ev[st].start[i]);
            (void)hipEventDestroy(h->ev[st].stop[i]);
        }
    (void)hipHostFree(h->host_counts);
    if (h->tile_counters) (void)hipFree(h->tile_counters);
    if (h->accum) (void)hipFree(h->accum);
    if (h->key_buckets) (void)hipFree(h->key_buckets);
    for (void* p : h->retired) (void)hipFree(p);
    delete h;
    return FR_OK;
}

int fr_profile_enable(fr_handle* hh, int32_t on)
{
    fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(hh);
    if (!h) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null handle");
    h->profiling = on != 0;
    for (int st = 0; st < ST_COUNT; st++) h->ev[st].used = 0;
    return FR_OK;
}

int fr_profile_read(fr_handle* hh, int32_t stage, double* total_ms, uint32_t* launches)
{
    fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(hh);
    if (!h || stage < 0 || stage >= ST_COUNT || !total_ms || !launches) return fail_msg(FR_ERR_INVALID_ARGUMENT, "bad argument");
    StageEvents& e = h->ev[stage];
    double tot = 0;
    for (size_t i = 0; i < e.used; i++) {
        float ms = 0;
        FR_HIP(hipEventElapsedTime(&ms, e.start[i], e.stop[i]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (uint32_t)e.used;
    return FR_OK;
}

size_t fr_geometry_bytes(int32_t P) { return GeomView::bytes((size_t)(P > 0 ? P : 0)); }
size_t fr_image_bytes(int32_t W, int32_t H) { return ImageView::bytes(W, H); }
size_t fr_binning_bytes(uint64_t capacity, int32_t W, int32_t H)
{
    ImageView v = ImageView::make(nullptr, W, H);
    return BinningView::bytes((size_t)capacity, (size_t)v.tiles_x * v.tiles_y);
}

int fr_forward(fr_handle* hh, const fr_params* prm, const fr_inputs* in, float* out_color, int32_t* radii,
               void* geometry, void* image, void* binning, uint64_t binning_capacity, fr_counts* counts, void* stream)
{
    fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(hh);
    if (!h) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null handle");
    int rc = check_frame(prm, in, true);
    if (rc) return rc;
    if (!out_color || !image || (prm->P > 0 && (!radii || !geometry)) || (binning_capacity > 0 && !binning))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "null output / scratch pointer");
    if (binning_capacity >= (1ull << 32)) return fail_msg(FR_ERR_UNSUPPORTED, "binning capacity must be < 2^32 instances");
    const ForwardCall c = {h, prm, in, out_color, radii, geometry, image, binning, binning_capacity, counts};
    return launch_forward(1, &c, static_cast<hipStream_t>(stream));
}

int fr_forward_batch(int32_t n_views, fr_handle* const* handles, const fr_params* const* prm, const fr_inputs* const* in,
                     float* const* out_color, int32_t* const* radii, void* const* geometry, void* const* image,
                     void* const* binning, const uint64_t* binning_capacity, fr_counts* counts, void* stream)
{
    if (n_views < 1 || n_views > kMaxBatch) return fail_msg(FR_ERR_INVALID_ARGUMENT, "n_views must be 1 .. FR_MAX_BATCH");
    if (!handles || !prm || !in || !out_color || !radii || !geometry || !image || !binning || !binning_capacity)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "null argument array");
    ForwardCall c[kMaxBatch];
    for (int k = 0; k < n_views; k++) {
        fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(handles[k]);
        if (!h) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null handle");
        for (int j = 0; j < k; j++)
            if (handles[j] == handles[k]) return fail_msg(FR_ERR_INVALID_ARGUMENT, "the views of a batch need a handle each");
        int rc = check_frame(prm[k], in[k], true);
        if (rc) return rc;
        if (n_views > 1 && prm[k]->P <= 0) return fail_msg(FR_ERR_INVALID_ARGUMENT, "batched views need P > 0");
        if (!out_color[k] || !image[k] || (prm[k]->P > 0 && (!radii[k] || !geometry[k])) || (binning_capacity[k] > 0 && !binning[k]))
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "null output / scratch pointer");
        if (binning_capacity[k] >= (1ull << 32)) return fail_msg(FR_ERR_UNSUPPORTED, "binning capacity must be < 2^32 instances");
        c[k] = ForwardCall{h, prm[k], in[k], out_color[k], radii[k], geometry[k], image[k], binning[k], binning_capacity[k],
                           counts ? counts + k : nullptr};
    }
    return launch_forward(n_views, c, static_cast<hipStream_t>(stream));
}

int fr_read_counts(fr_handle* hh, fr_counts* counts)
{
    fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(hh);
    if (!h || !counts) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null argument");
    *counts = *h->host_counts;
    // (valid once the frame's stream has been synchronised: the caller's responsibility.)  A handle driven only with
    // FR_FLAG_NO_WAIT learns here that the pinned slot holds a completed frame's counts: the next eager frame sizes the
    // key buckets from them (word 4) and decides about the big-list sorter.
    h->counts_seen = true;
    return FR_OK;
}

int fr_backward(fr_handle* hh, const fr_params* prm, const fr_inputs* in, const int32_t* radii, void* geometry,
                const void* image, const void* binning, const float* dL_dpix, const fr_grads* grads, void* stream)
{
    fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(hh);
    if (!h) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null handle");
    int rc = check_frame(prm, in, false);
    if (rc) return rc;
    if (prm->P == 0) return FR_OK;
    if (!radii || !geometry || !image || !binning || !dL_dpix || !grads)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "null pointer");
    const BackwardCall c = {h, prm, in, radii, geometry, image, binning, dL_dpix, grads};
    return launch_backward(1, &c, static_cast<hipStream_t>(stream));
}

int fr_backward_batch(int32_t n_views, fr_handle* const* handles, const fr_params* const* prm, const fr_inputs* const* in,
                      const int32_t* const* radii, void* const* geometry, const void* const* image, const void* const* binning,
                      const float* const* dL_dpix, const fr_grads* const* grads, void* stream)
{
    if (n_views < 1 || n_views > kMaxBatch) return fail_msg(FR_ERR_INVALID_ARGUMENT, "n_views must be 1 .. FR_MAX_BATCH");
    if (!handles || !prm || !in || !radii || !geometry || !image || !binning || !dL_dpix || !grads)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "null argument array");
    BackwardCall c[kMaxBatch];
    for (int k = 0; k < n_views; k++) {
        fr_handle_impl* h = reinterpret_cast<fr_handle_impl*>(handles[k]);
        if (!h) return fail_msg(FR_ERR_INVALID_ARGUMENT, "null handle");
        for (int j = 0; j < k; j++)
            if (handles[j] == handles[k]) return fail_msg(FR_ERR_INVALID_ARGUMENT, "the views of a batch need a handle each");
        int rc = check_frame(prm[k], in[k], false);
        if (rc) return rc;
        if (prm[k]->P <= 0) return fail_msg(FR_ERR_INVALID_ARGUMENT, "batched views need P > 0");
        if (!radii[k] || !geometry[k] || !image[k] || !binning[k] || !dL_dpix[k] || !grads[k])
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "null pointer");
        c[k] = BackwardCall{h, prm[k], in[k], radii[k], geometry[k], image[k], binning[k], dL_dpix[k], grads[k]};
    }
    return launch_backward(n_views, c, static_cast<hipStream_t>(stream));
}

int fr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                    void* stream)
{
    (void)projmatrix;  // the reference computes but does not use the projected point (auxiliary.h:149-154)
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail_msg(FR_ERR_INVALID_ARGUMENT, "bad argument");
    return launch_mark_visible(P, means3D, viewmatrix, present, static_cast<hipStream_t>(stream));
}

const float* fr_image_final_T(const void* image, int32_t W, int32_t H)
{
    return ImageView::make(const_cast<void*>(image), W, H).final_T;
}
const uint32_t* fr_image_n_contrib(const void* image, int32_t W, int32_t H)
{
    return ImageView::make(const_cast<void*>(image), W, H).n_contrib;
}

const void* fr_debug_geometry_field(const void* geometry, int32_t P, int32_t field)
{
    GeomView g = GeomView::make(const_cast<void*>(geometry), (size_t)(P > 0 ? P : 0));
    switch (field) {
        case 0: return nullptr;   // (pixel-space centres: columns 0-1 of field 8)
        case 1: return nullptr;   // (view-space depth: column 10 of field 8)
        case 2: return nullptr;   // (conic + opacity: columns 2-5 of field 8 hold (-0.5 a, -b, -0.5 c, opacity))
        case 3: return nullptr;   // (colours: columns 6-8 of field 8)
        case 4: return nullptr;   // (Sigma3D is not stored any more: both per-Gaussian kernels compute it)
        case 5: return nullptr;   // (the 8x8-tile rectangle is no longer stored)
        case 6: return g.clamped;
        case 7: return nullptr;   // (the gradient accumulators moved into the handle)
        case 8: return g.rec_tmpl;   // [P][12]: x, y, a', b', c', opacity, r, g, b, id bits, depth, 0
        default: return nullptr;
    }
}

int fr_debug_selftest_reduce(const float* in, float* out, void* stream)
{
    return launch_selftest_reduce(in, out, static_cast<hipStream_t>(stream));
}

size_t fr_knn_workspace_bytes(int32_t P) { return knn_workspace_bytes(P); }

int fr_knn_mean_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (P < 0 || (P > 0 && (!points || !out || !workspace))) return fail_msg(FR_ERR_INVALID_ARGUMENT, "bad argument");
    return launch_knn(P, points, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream), 0);
}

int fr_knn_nearest_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (P < 0 || (P > 0 && (!points || !out || !workspace))) return fail_msg(FR_ERR_INVALID_ARGUMENT, "bad argument");
    return launch_knn(P, points, out, workspace, workspace_bytes, static_cast<hipStream_t>(stream), 1);
}

static int check_binding(const fr_binding* b)
{
    if (!b || b->N < 0 || b->V < 0 || b->F < 0) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_binding: null or negative sizes");
    if (b->N > 0 && (!b->verts || !b->faces || !b->face_index || !b->bary || !b->offset || !b->rotation || !b->scaling ||
                     (b->resize_scale && !b->face_scale_canonical)))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_binding: missing array");
    return FR_OK;
}

int fr_face_scale(int32_t V, int32_t F, const float* verts, const int32_t* faces, float* out_scale, void* stream)
{
    if (V < 0 || F < 0 || (F > 0 && (!verts || !faces || !out_scale))) return fail_msg(FR_ERR_INVALID_ARGUMENT, "bad argument");
    return launch_face_scale(F, verts, faces, out_scale, static_cast<hipStream_t>(stream));
}

int fr_bind_forward(const fr_binding* b, float* xyz, float* rotation_out, float* scaling_out, void* stream)
{
    int rc = check_binding(b);
    if (rc) return rc;
    if (b->N > 0 && (!xyz || !rotation_out || !scaling_out)) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_bind_forward: null output");
    return launch_bind_forward(*b, xyz, rotation_out, scaling_out, static_cast<hipStream_t>(stream));
}

int fr_bind_backward(const fr_binding* b, const float* g_xyz, const float* g_rotation, const float* g_scaling,
                     float* d_verts, float* d_offset, float* d_rotation, float* d_scaling, void* stream)
{
    int rc = check_binding(b);
    if (rc) return rc;
    return launch_bind_backward(*b, g_xyz, g_rotation, g_scaling, d_verts, d_offset, d_rotation, d_scaling,
                                static_cast<hipStream_t>(stream));
}

int fr_adam_step(const fr_adam_config* cfg, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                 uint64_t n, float* state, void* stream)
{
    if (!cfg || cfg->n_segments < 1 || cfg->n_segments > FR_ADAM_MAX_SEGMENTS)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step: 1..FR_ADAM_MAX_SEGMENTS segments");
    if (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !state))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step: null array");
    uint64_t prev = 0;
    for (int i = 0; i < cfg->n_segments; i++) {
        if (cfg->segment_end[i] < prev) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step: segment ends must ascend");
        prev = cfg->segment_end[i];
    }
    if (prev != n) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step: the last segment must end at n");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step: arrays must be 16-byte aligned");
    const float* one[1] = {grad};
    return launch_adam(*cfg, param, one, 1, exp_avg, exp_avg_sq, n, state, static_cast<hipStream_t>(stream));
}

int fr_adam_step_multi(const fr_adam_config* cfg, float* param, const float* const* grads, int32_t n_grads, float* exp_avg,
                       float* exp_avg_sq, uint64_t n, float* state, void* stream)
{
    if (!cfg || cfg->n_segments < 1 || cfg->n_segments > FR_ADAM_MAX_SEGMENTS)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: 1..FR_ADAM_MAX_SEGMENTS segments");
    if (!grads || n_grads < 1 || n_grads > FR_ADAM_MAX_GRADS)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: 1..FR_ADAM_MAX_GRADS gradient buffers");
    for (int k = 0; k < n_grads; k++)
        if (n > 0 && (!grads[k] || (reinterpret_cast<uintptr_t>(grads[k]) & 15)))
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: null or misaligned gradient buffer");
    if (n > 0 && (!param || !exp_avg || !exp_avg_sq || !state)) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: null array");
    uint64_t prev = 0;
    for (int i = 0; i < cfg->n_segments; i++) {
        if (cfg->segment_end[i] < prev) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: segment ends must ascend");
        prev = cfg->segment_end[i];
    }
    if (prev != n) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: the last segment must end at n");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_adam_step_multi: arrays must be 16-byte aligned");
    return launch_adam(*cfg, param, grads, n_grads, exp_avg, exp_avg_sq, n, state, static_cast<hipStream_t>(stream));
}

size_t fr_l1_workspace_bytes(void) { return 17 * 128 + 1024 * sizeof(float); }

int fr_l1_loss_grad(uint64_t n, const float* img, const float* gt, float* grad, float* loss, void* workspace, void* stream)
{
    if (n > 0 && (!img || !gt || !loss || !workspace)) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad: null array");
    if ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(grad)) & 15)
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad: arrays must be 16-byte aligned");
    return launch_l1_loss_grad(n, img, gt, grad, loss, workspace, static_cast<hipStream_t>(stream));
}

int fr_l1_loss_grad_batch(int32_t n_images, uint64_t n, const float* const* img, const float* const* gt, float* const* grad,
                          float* const* loss, void* const* workspace, void* stream)
{
    if (n_images < 1 || n_images > kMaxBatch) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad_batch: 1 .. FR_MAX_BATCH images");
    if (!img || !gt || !loss || !workspace) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad_batch: null argument array");
    for (int k = 0; k < n_images; k++) {
        if (n > 0 && (!img[k] || !gt[k] || !loss[k] || !workspace[k]))
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad_batch: null array");
        if ((reinterpret_cast<uintptr_t>(img[k]) | reinterpret_cast<uintptr_t>(gt[k]) | (grad ? reinterpret_cast<uintptr_t>(grad[k]) : 0)) & 15)
            return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad_batch: arrays must be 16-byte aligned");
        for (int j = 0; j < k; j++)
            if (workspace[j] == workspace[k]) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_l1_loss_grad_batch: one workspace per image");
    }
    return launch_l1_loss_grad_batch(n_images, n, img, gt, grad, loss, workspace, static_cast<hipStream_t>(stream));
}

int fr_multi_copy(int32_t n_segments, float* const* dst, const float* const* src, const uint64_t* count, void* stream)
{
    if (n_segments < 0 || n_segments > FR_COPY_MAX_SEGMENTS || (n_segments > 0 && (!dst || !src || !count)))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_multi_copy: 0..FR_COPY_MAX_SEGMENTS segments");
    unsigned long long cnt[FR_COPY_MAX_SEGMENTS];
    for (int i = 0; i < n_segments; i++) {
        if (count[i] > 0 && (!dst[i] || !src[i])) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_multi_copy: null segment");
        cnt[i] = count[i];
    }
    return launch_multi_copy(n_segments, dst, src, cnt, static_cast<hipStream_t>(stream));
}

int fr_scaled_sum(int32_t n_src, const float* const* src, float* dst, uint64_t count, float scale, void* stream)
{
    if (n_src < 1 || n_src > FR_ADAM_MAX_GRADS || !src || (count > 0 && !dst))
        return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_scaled_sum: 1..FR_ADAM_MAX_GRADS sources");
    uintptr_t bits = reinterpret_cast<uintptr_t>(dst);
    for (int k = 0; k < n_src; k++) {
        if (count > 0 && !src[k]) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_scaled_sum: null source");
        bits |= reinterpret_cast<uintptr_t>(src[k]);
    }
    if (bits & 15) return fail_msg(FR_ERR_INVALID_ARGUMENT, "fr_scaled_sum: arrays must be 16-byte aligned");
    return launch_scaled_sum(n_src, src, dst, count, scale, static_cast<hipStream_t>(stream));
}

}  // extern "C"
