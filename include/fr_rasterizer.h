/*
 * fr_rasterizer.h — C ABI of the MI355X (gfx950) Gaussian-splat rasterizer.
 *
 * Drop-in boundary for FateAvatar's render path.  Every entry point takes plain
 * DEVICE pointers, sizes and a HIP stream; no torch types, no exceptions across
 * the ABI, integer return codes.  The entry points are what the reference's own
 * torch glue binds (reference paths are relative to
 * submodules/diff-gaussian-rasterization/ and submodules/simple-knn/):
 *
 *   fr_forward        replaces CudaRasterizer::Rasterizer::forward
 *                     (cuda_rasterizer/rasterizer.h:36-60, rasterizer_impl.cu:198-336),
 *                     called from RasterizeGaussiansCUDA (rasterize_points.cu:35-115)
 *   fr_backward       replaces CudaRasterizer::Rasterizer::backward
 *                     (cuda_rasterizer/rasterizer.h:62-85, rasterizer_impl.cu:340-434),
 *                     called from RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196)
 *   fr_mark_visible   replaces CudaRasterizer::Rasterizer::markVisible
 *                     (cuda_rasterizer/rasterizer.h:24-29, rasterizer_impl.cu:141-153),
 *                     called from markVisible (rasterize_points.cu:198-217)
 *   fr_bind_forward / fr_bind_backward / fr_face_scale replace the ~40 PyTorch kernels of the mesh
 *                     binding (model/fateavatar.py:225-258, volume_rendering/mesh_compute.py:27-59)
 *   fr_adam_step      replaces torch.optim.Adam.step() over the Gaussian groups (train/optim.py:11-37)
 *   fr_knn_mean_dist2 replaces SimpleKNN::knn (simple_knn.h, simple_knn.cu:186-222),
 *                     called from distCUDA2 (spatial.cu:14-25)
 *   fr_knn_nearest_dist2 replaces pytorch3d knn_points(p, p, K=6).dists[..., 1] in get_init_scale_by_knn
 *                     (model/fateavatar.py:597-608)
 *
 * Differences from the reference interface, all deliberate:
 *   - Scratch is three caller-owned byte buffers sized by fr_*_bytes() instead of
 *     std::function<char*(size_t)> resize callbacks: the caller's allocator stays in
 *     charge (torch caching allocator in the Python host).  The only device memory the
 *     library owns lives in the fr_handle: per-tile counters, the key buckets the counting
 *     pass writes into (tiles x 8 x capacity keys; 34 MB at 512 x 512) and the gradient
 *     accumulators of the blend backward (64 B per Gaussian), all allocated when a size is
 *     first seen and grown on demand.
 *   - The binning buffer has a CAPACITY.  fr_forward never blocks the GPU on the
 *     instance count (the reference does a blocking cudaMemcpy, rasterizer_impl.cu:281):
 *     it enqueues the whole frame, then reads the counts the sort kernel wrote to pinned
 *     host memory.  If the capacity was too small it returns FR_ERR_BINNING_CAPACITY and
 *     the required size; the caller regrows and calls again.
 *   - Work is enqueued on the stream passed in, not on the legacy default stream.
 *   - Tiles are 8x8 pixels = one 64-lane wavefront (the reference uses 16x16 = 256
 *     threads) and per-tile lists only hold Gaussians whose alpha >= 1/255 footprint
 *     touches the tile, clipped to the reference's own 16x16 tile rectangle, so every
 *     pixel blends exactly the sequence the reference blends.  `num_rendered` is still
 *     reported in reference semantics (sum of 16x16 tiles touched).
 */
#ifndef FR_RASTERIZER_H_INCLUDED
#define FR_RASTERIZER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_OK 0
#define FR_ERR_INVALID_ARGUMENT 1
#define FR_ERR_BINNING_CAPACITY 2 /* binning buffer too small; *instances_needed tells how big */
#define FR_ERR_HIP 3              /* a HIP runtime call failed; see fr_last_error() */
#define FR_ERR_UNSUPPORTED 4

/* Per-device context: pinned count slot, event, and the per-tile binning counters (device memory, kept zero
 * between frames so that a frame needs no zeroing launch), the key buckets and the gradient accumulators.  Calls on
 * one handle must not overlap on the host (one thread at a time); on the device, frames enqueued on one stream are
 * ordered by it and a frame on another stream first waits for the previous frame's event (forward passes wait for the
 * previous forward, backward passes for the previous backward: they share the counters resp. the accumulators).  Frames
 * enqueued while a stream is being CAPTURED are ordered by the capture; replays of the graph by whoever launches them —
 * give frames that are to run concurrently their own handles.  The handle's device memory
 * grows with the tile grid / Gaussian count / longest per-XCD tile sub-list (hipMalloc — not while the stream is being
 * captured into a hipGraph: run one eager frame of that size first; fr_forward on a capturing stream that would have
 * to grow something returns FR_ERR_UNSUPPORTED before enqueuing anything, which leaves the capture valid).  A captured
 * graph holds the handle's buffer pointers: once a frame of the handle has been captured, buffers that a later eager
 * frame outgrows are retired (kept until fr_destroy) instead of freed, so replays stay valid. */
typedef struct fr_handle fr_handle;

/* (defined with fr_bind_forward below) */
typedef struct fr_binding fr_binding;

/* Optional fused side inputs / outputs: per-Gaussian work the caller's step otherwise does
 * with extra kernels around the rasterizer.  Device pointers unless stated otherwise; any member may be NULL. */
typedef struct fr_aux {
    uint8_t* visible;   /* fr_forward  out    [P]: radii > 0, i.e. render()'s visibility_filter (render_3dgs.py:80) */
    float* grad_accum;  /* fr_backward in/out [P]: += ||dL_dmeans2D[i,:2]|| where radii > 0 — xyz_gradient_accum of
                           _add_densification_stats (model/fateavatar.py:734-737) */
    float* denom;       /* fr_backward in/out [P]: += 1 where radii > 0 (same function) */
    /* The frame rendered straight from its MESH BINDING (model/fateavatar.py:225-258; HOST pointer to the descriptor
     * fr_bind_forward takes, binding->N == P).  Needs FR_FLAG_RAW_ACTIVATIONS and scales + rotations.
     * fr_forward: the per-Gaussian kernel evaluates fr_bind_forward's expressions in front of its own work — same bits —
     * and inputs.means3D / rotations / scales are then OUTPUTS of the call: every row is written with the bound values
     * (what the reference assigns to gaussian._xyz / _rotation / _scaling before render()); they are not read.
     * fr_backward (same descriptor, the arrays fr_forward wrote handed back in inputs): the per-Gaussian backward
     * continues through the binding — fr_bind_backward's expressions on the gradients it would have written to
     * dL_dmeans3D / dL_drotations / dL_dscales (which may then be NULL) — and writes d_offset [P], d_rotation [P,4],
     * d_scaling [P,3] (every row; zeros for culled Gaussians) and ADDS dL/dverts into d_verts [V,3] (float atomics; the
     * caller zeroes it).  FR_FLAG_ACCUMULATE does not apply to these four. */
    const fr_binding* binding;
    float* d_verts;
    float* d_offset;
    float* d_rotation;
    float* d_scaling;
    /* fr_backward out [1] (device float, or NULL): 1.0f if the frame this backward belongs to overflowed its binning capacity
     * — nothing was blended, every gradient of the call is zero —, else 0.0f.  A step replayed from a hipGraph cannot react on
     * the host before its optimizer kernel runs; fr_adam_config::skip points the update at these words instead (a step
     * whose gradient is partly zeros for that reason is skipped, moments and step count included).  The word is OVERWRITTEN by
     * every backward call: frames that feed one optimizer step need ONE WORD EACH (fr_adam_config::skip takes up to
     * FR_ADAM_MAX_GRADS of them); the views of one fr_backward_batch call must not share a word. */
    float* overflow_out;
} fr_aux;

/* Frame parameters: the scalar arguments of Rasterizer::forward/backward. */
typedef struct fr_params {
    int32_t P;            /* number of Gaussians */
    int32_t D;            /* active SH degree (0..3) */
    int32_t M;            /* SH coefficients stored per Gaussian (shs is [P,M,3]); 0 if no shs */
    int32_t W, H;         /* image size in pixels */
    float tan_fovx, tan_fovy;
    float scale_modifier;
    int32_t prefiltered;  /* accepted for interface parity; the near-plane cull is always applied */
    int32_t debug;        /* !=0: synchronise and check after every stage (auxiliary.h:166-173) */
    int32_t flags;        /* FR_FLAG_* */
    const fr_aux* aux;    /* optional fused side outputs (host pointer to a struct of device pointers), or NULL */
} fr_params;

/* fr_forward does not wait for the frame counts: nothing in the call blocks or touches an event, so it can
 * be captured into a hipGraph.  `counts` is not filled; after the stream has been synchronised read them with
 * fr_read_counts.  If counts.overflow is set the frame's outputs are invalid: rerun with a larger capacity. */
#define FR_FLAG_NO_WAIT 1

/* Fused activations (reference volume_rendering/gaussian_model.py:39-50,105-128):
 * inputs.opacities / scales / rotations hold the RAW parameters and the kernels apply sigmoid / exp /
 * normalize (x / max(|x|, 1e-12)) themselves; fr_backward then returns dL/d(raw) in dL_dopacity, dL_dscales,
 * dL_drotations.  Requires scales + rotations (not cov3D_precomp).  Pass the same flag to fr_backward. */
#define FR_FLAG_RAW_ACTIVATIONS 2

/* fr_backward only: FR_FLAG_ACCUMULATE(k) makes the k-th array of fr_grads (k = position of the pointer in the
 * struct: 0 = dL_dmeans2D ... 7 = dL_drotations) ACCUMULATE: the frame's gradient is added to what the array holds
 * instead of overwriting it (culled Gaussians then touch nothing).  A batch of frames rendered from the same
 * parameters (the reference's batch loop, model/fateavatar.py:251-276, whose gradients autograd sums with one add
 * kernel and one temporary per frame and parameter) accumulates in one buffer: first frame without the flag, the
 * others with it.  FR_FLAG_ACCUMULATE_ALL = every array. */
#define FR_FLAG_ACCUMULATE_SHIFT 8
#define FR_FLAG_ACCUMULATE(k) (1 << (FR_FLAG_ACCUMULATE_SHIFT + (k)))
#define FR_FLAG_ACCUMULATE_ALL (0xFF << FR_FLAG_ACCUMULATE_SHIFT)

/* Device pointers.  NULL = the "empty tensor" of the reference glue
 * (rasterize_points.cu:94-103 passes data_ptr() of empty tensors; kernels branch on nullptr). */
typedef struct fr_inputs {
    const float* background;     /* [3] */
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL */
    const float* colors_precomp; /* [P,3]   or NULL (exactly one of shs / colors_precomp) */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3]   or NULL */
    const float* rotations;      /* [P,4]   or NULL (r,x,y,z; not normalised in-kernel) */
    const float* cov3D_precomp;  /* [P,6]   or NULL (exactly one of scales+rotations / cov3D_precomp) */
    const float* viewmatrix;     /* [16] row-major "transposed" world->view (camera_3dgs.py:53) */
    const float* projmatrix;     /* [16] full projection, same convention (camera_3dgs.py:71) */
    const float* campos;         /* [3] */
} fr_inputs;

/* Gradient outputs of fr_backward (device pointers).  Every non-NULL array is fully written
 * (rows of culled Gaussians are zero), so the caller may pass uninitialised memory; the
 * reference instead requires nine zero-filled tensors (rasterize_points.cu:151-159).
 * NULL = not wanted. */
typedef struct fr_grads {
    float* dL_dmeans2D;   /* [P,3] screen-space mean gradient, NDC-scaled, z = 0 (backward.cu:545-546) */
    float* dL_dcolors;    /* [P,3] */
    float* dL_dopacity;   /* [P]   */
    float* dL_dmeans3D;   /* [P,3] */
    float* dL_dcov3D;     /* [P,6] */
    float* dL_dsh;        /* [P,M,3] */
    float* dL_dscales;    /* [P,3] */
    float* dL_drotations; /* [P,4] */
} fr_grads;

/* What the binning stages report for a frame. */
typedef struct fr_counts {
    uint32_t num_rendered;   /* reference semantics: sum over Gaussians of 16x16 tiles touched */
    uint32_t num_instances;  /* (8x8 tile, Gaussian) instances this implementation bins and sorts */
    uint32_t max_tile_list;  /* longest per-tile list */
    uint32_t overflow;       /* 1 if num_instances exceeded the binning capacity */
} fr_counts;

int fr_create(fr_handle** out);
int fr_destroy(fr_handle* h);
const char* fr_last_error(void);
const char* fr_version(void);

/* Stage timing.  While enabled, every kernel launch of fr_forward / fr_backward is bracketed by HIP
 * events on the stream it is launched on; fr_profile_read sums the elapsed time of one stage over all
 * launches since fr_profile_enable(h, 1) (the stream must have been synchronised by the caller).
 * stage: 0 preprocess_fwd (+ key binning), 1 scan (per-tile totals and range allocation), 2 emit (no launch any more:
 * the preprocess kernel writes the keys), 3 tile_sort, 4 blend_fwd, 5 blend_bwd, 6 preprocess_bwd. */
int fr_profile_enable(fr_handle* h, int32_t on);
int fr_profile_read(fr_handle* h, int32_t stage, double* total_ms, uint32_t* launches);

/* Scratch sizes in bytes.  geometry: per-Gaussian state + gradient accumulators; image: per-pixel
 * final transmittance / contributor count and per-tile ranges; binning: `capacity` instances. */
size_t fr_geometry_bytes(int32_t P);
size_t fr_image_bytes(int32_t W, int32_t H);
size_t fr_binning_bytes(uint64_t capacity, int32_t W, int32_t H);

/* out_color [3,H,W], radii [P] (reference semantics: ceil(3*sigma_max), 0 if culled).
 * Returns FR_OK, or FR_ERR_BINNING_CAPACITY with counts->num_instances = capacity required
 * (outputs are then undefined and the call must be repeated with a larger binning buffer). */
int fr_forward(fr_handle* h, const fr_params* prm, const fr_inputs* in, float* out_color, int32_t* radii,
               void* geometry, void* image, void* binning, uint64_t binning_capacity, fr_counts* counts,
               void* hip_stream);

/* Counts of the most recent frame enqueued through this handle (valid once its stream has been synchronised). */
int fr_read_counts(fr_handle* h, fr_counts* counts);

/* geometry/image/binning: the buffers a successful fr_forward of the same frame filled.
 * dL_dpix [3,H,W]. */
int fr_backward(fr_handle* h, const fr_params* prm, const fr_inputs* in, const int32_t* radii, void* geometry,
                const void* image, const void* binning, const float* dL_dpix, const fr_grads* grads,
                void* hip_stream);

/* ---- several views through ONE launch chain (SURVEY.md §8e / reference model/fateavatar.py:251-276: the frames of a
 * batch are rendered one after the other with shared Gaussians; one frame's kernels leave most of an MI355X idle).
 * fr_forward_batch / fr_backward_batch do exactly what n_views calls of fr_forward / fr_backward would — every view has
 * its own handle, parameter block, inputs (which may or may not share pointers), outputs and scratch buffers, and the
 * results are the same bit for bit (the forward) resp. to atomic-summation order (the backward) — but every kernel of
 * the frame is launched once with a (grid, n_views) grid, so that the views fill the chip together without any stream
 * or hardware-queue arrangement on the caller's side.  1 <= n_views <= FR_MAX_BATCH; batched views need P > 0, distinct
 * handles and the default forward path (FR_BLEND_FWD unset).  `counts` (may be NULL): n_views entries.  Returns
 * FR_ERR_BINNING_CAPACITY if ANY view overflowed its binning capacity (counts[k].overflow says which). */
#define FR_MAX_BATCH 4
int fr_forward_batch(int32_t n_views, fr_handle* const* handles, const fr_params* const* prm, const fr_inputs* const* in,
                     float* const* out_color, int32_t* const* radii, void* const* geometry, void* const* image,
                     void* const* binning, const uint64_t* binning_capacity, fr_counts* counts, void* hip_stream);
int fr_backward_batch(int32_t n_views, fr_handle* const* handles, const fr_params* const* prm, const fr_inputs* const* in,
                      const int32_t* const* radii, void* const* geometry, const void* const* image,
                      const void* const* binning, const float* const* dL_dpix, const fr_grads* const* grads,
                      void* hip_stream);

/* ---- fused Adam over a flat parameter buffer (replaces torch.optim.Adam.step() over the
 * Gaussian parameter groups of train/optim.py:11-37: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
 * The buffer is cut into up to FR_ADAM_MAX_SEGMENTS consecutive segments, each with its own learning rate (the
 * reference's param groups); param / grad / exp_avg / exp_avg_sq are device arrays of n floats with the same
 * layout.  `state` is a device array of FR_ADAM_STATE_FLOATS floats owned by the caller, zero-initialised once: {step,
 * 1 - beta1^step, 1 - beta2^step, unused, ..., kernel bookkeeping from word 32 on that is zero between calls}; every call
 * advances it on the device (so the call is hipGraph-capturable: nothing step-dependent is a kernel argument) and applies
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * with g = grad_scale * grad (grad_scale: e.g. 1/world_size after a SUM all-reduce). */
#define FR_ADAM_MAX_SEGMENTS 16
#define FR_ADAM_STATE_FLOATS 576
typedef struct fr_adam_config {
    int32_t n_segments;
    uint64_t segment_end[FR_ADAM_MAX_SEGMENTS]; /* exclusive end offset (in floats) of each segment, ascending; last == n */
    float segment_lr[FR_ADAM_MAX_SEGMENTS];
    /* optional two-rate pattern inside a segment (SH coefficients stored [P,M,3] with the DC term at lr and the
     * rest at lr/20, train/optim.py:49-50): element e (relative to the segment start) uses segment_lr if
     * e % segment_period < segment_split, else segment_lr2.  segment_period == 0: segment_lr everywhere. */
    uint32_t segment_period[FR_ADAM_MAX_SEGMENTS];
    uint32_t segment_split[FR_ADAM_MAX_SEGMENTS];
    float segment_lr2[FR_ADAM_MAX_SEGMENTS];
    double beta1, beta2, eps; /* doubles, like torch's hyper-parameters: 1 - beta is rounded to float from here */
    float grad_scale;
    /* optional: n_skip (0 .. FR_ADAM_MAX_GRADS) device floats; if ANY of them is non-zero when the kernel runs the step does
     * nothing — no parameter, no moment, no step count changes.  Point them at the fr_aux::overflow_out words of the frames
     * whose gradients feed this step; in a data-parallel step keep those words behind the gradients in the exchanged buffer,
     * so that the all-reduce sums them and every rank skips the same steps. */
    const float* skip[4 /* FR_ADAM_MAX_GRADS */];
    int32_t n_skip;
} fr_adam_config;
int fr_adam_step(const fr_adam_config* cfg, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                 uint64_t n, float* state, void* hip_stream);
/* The same step with the gradient given as the SUM of n_grads (1 .. FR_ADAM_MAX_GRADS) buffers of n floats: the views of
 * a batch (the reference's `for bs_ in range(bs)` loop, model/fateavatar.py:251-276) each back-propagate into their own
 * buffer, in flight together; grad_scale = 1 / (views x ranks) makes it the batch mean of train/loss.py:92-105. */
#define FR_ADAM_MAX_GRADS 4
int fr_adam_step_multi(const fr_adam_config* cfg, float* param, const float* const* grads, int32_t n_grads,
                       float* exp_avg, float* exp_avg_sq, uint64_t n, float* state, void* hip_stream);

/* ---- the image loss of the optimisation step (reference nn.L1Loss(reduction='mean') on the rendered
 * image, model/loss.py:92, followed by loss.backward()): loss = mean |img - gt| and grad = sign(img - gt) / n (what
 * autograd hands to the rasterizer's backward for a unit upstream gradient) in ONE launch.  `grad` may be NULL.
 * `workspace`: fr_l1_workspace_bytes() bytes of device memory, zeroed ONCE by the caller (the kernel leaves it zeroed).
 * A workspace must NOT be shared by launches that can overlap on the device (other streams, other graphs): the kernel
 * elects its last workgroup and sums per-workgroup partials through it;
 * `loss`: one device float.  All arrays on the device, n floats each. */
size_t fr_l1_workspace_bytes(void);
int fr_l1_loss_grad(uint64_t n, const float* img, const float* gt, float* grad, float* loss, void* workspace,
                    void* hip_stream);
/* The same for the n_images (1 .. FR_MAX_BATCH) images of the frames of a batch (n floats each; the reference's batch loop,
 * model/fateavatar.py:251-276 with train/loss.py:92-105) in ONE launch: image k has its own gt / grad / loss / workspace
 * (distinct workspaces).  `grad` may be NULL (no gradients), or hold NULL entries. */
int fr_l1_loss_grad_batch(int32_t n_images, uint64_t n, const float* const* img, const float* const* gt, float* const* grad,
                          float* const* loss, void* const* workspace, void* hip_stream);

/* ---- dst = scale * (src[0] + ... + src[n_src - 1]), n_src in 1 .. FR_ADAM_MAX_GRADS arrays of `count` floats, 16-byte
 * aligned: the mean of the gradient buffers of the views a rank rendered in flight together, written into the exchange
 * buffer of the data-parallel all-reduce in one pass.  dst may be one of the sources. */
int fr_scaled_sum(int32_t n_src, const float* const* src, float* dst, uint64_t count, float scale, void* hip_stream);

/* ---- up to FR_COPY_MAX_SEGMENTS device-to-device copies of float arrays in one launch (the per-frame inputs of a
 * captured step: camera block, posed vertices, target image).  Segments must not overlap each other. */
#define FR_COPY_MAX_SEGMENTS 12
int fr_multi_copy(int32_t n_segments, float* const* dst, const float* const* src, const uint64_t* count, void* hip_stream);

/* ---- FateAvatar's mesh binding (reference model/fateavatar.py:225-258 with
 * volume_rendering/mesh_compute.py:27-59 and pytorch3d's matrix_to_quaternion / quaternion_multiply): from the posed
 * mesh and each Gaussian's binding (face index, barycentrics) and raw parameters to what the reference assigns to
 * gaussian._xyz / _rotation / _scaling before render():
 *   xyz = sum_k bary_k v_k + (e1 x e2) * shell_len * tanh(offset);  rotation = standardize(q_face (x) rotation);
 *   scaling = scaling + log(face_scale / face_scale_canonical)   (resize_scale != 0; unchanged otherwise).
 * All pointers are device pointers; one frame per call. */
struct fr_binding {
    int32_t N, V, F;
    const float* verts;                 /* [V,3] posed vertices */
    const int32_t* faces;               /* [F,3] */
    const int32_t* face_index;          /* [N]   face every Gaussian is bound to */
    const float* bary;                  /* [N,3] barycentric coordinates */
    const float* face_scale_canonical;  /* [F]   fr_face_scale of the canonical mesh */
    float shell_len;                    /* cfg_model.normal_offset */
    int32_t resize_scale;
    const float* offset;                /* [N]   raw: tanh is applied here */
    const float* rotation;              /* [N,4] raw quaternion (r,x,y,z) */
    const float* scaling;               /* [N,3] raw log-scale */
};
int fr_face_scale(int32_t V, int32_t F, const float* verts, const int32_t* faces, float* out_scale, void* hip_stream);
int fr_bind_forward(const fr_binding* b, float* xyz, float* rotation_out, float* scaling_out, void* hip_stream);
/* Gradients of the three outputs in, gradients of offset / rotation / scaling out (fully written), and dL/dverts
 * ADDED into d_verts [V,3] with float atomics (the caller zeroes it).  Any of the seven arrays may be NULL. */
int fr_bind_backward(const fr_binding* b, const float* g_xyz, const float* g_rotation, const float* g_scaling,
                     float* d_verts, float* d_offset, float* d_rotation, float* d_scaling, void* hip_stream);

/* present[i] = view-space z of means3D[i] > 0.2 (auxiliary.h:154). */
int fr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    uint8_t* present, void* hip_stream);

/* Per-pixel auxiliaries of the last forward held in `image` (device pointers into it). */
const float* fr_image_final_T(const void* image, int32_t W, int32_t H);
const uint32_t* fr_image_n_contrib(const void* image, int32_t W, int32_t H);

/* Test/diagnostic accessor: device pointer of one per-Gaussian array inside a geometry buffer filled by
 * fr_forward.  field: 6 clamped (uint8 bitmask), 8 the blend-record
 * template (12 floats: x, y, a', b', c', opacity, r, g, b, id bits, depth, 0 — the pixel-space centre, the view-space
 * depth and the colour that fields 0, 1 and 3 used to hold are its columns 0-1, 10 and 6-8; 2 was conic_opacity (the
 * conic as (-0.5 a, -b, -0.5 c) and the opacity are columns 2-5; the backward inverts its own 2D covariance), 4 was cov3D, which is no
 * longer stored (both per-Gaussian kernels compute it); 5 was the tile rectangle, 7 the gradient accumulators: they
 * live in the handle now).  NULL for any other field. */
const void* fr_debug_geometry_field(const void* geometry, int32_t P, int32_t field);

/* Test hook for the wave reduce-scatter used by the blend backward: in[64*36] (lane-major), out[64]:
 * out[l] = sum over lanes of in[lane*36 + bitrev6(l)] for bitrev6(l) < 36. */
int fr_debug_selftest_reduce(const float* in, float* out, void* hip_stream);

/* simple-knn: out[i] = mean of the 3 smallest squared distances from points[i] to the other points. */
size_t fr_knn_workspace_bytes(int32_t P);
int fr_knn_mean_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes,
                      void* hip_stream);
/* out[i] = squared distance from points[i] to the nearest OTHER point (FLT_MAX if P == 1): the quantity
 * FateAvatar.get_init_scale_by_knn takes from pytorch3d's knn_points(p, p, K=6).dists[..., 1]
 * (model/fateavatar.py:597-608).  Same workspace as fr_knn_mean_dist2. */
int fr_knn_nearest_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes,
                         void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
