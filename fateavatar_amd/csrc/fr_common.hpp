// Shared declarations for the gfx950 rasterizer kernels: scratch-buffer layouts, error
// plumbing, wave-level helpers.  CDNA4 only: wavefront = 64 lanes, tile = 8x8 pixels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/fr_rasterizer.h"

namespace fr {

constexpr int kWave = 64;
constexpr int kTile = 8;    // 8x8 pixels = one wavefront
constexpr int kRefTile = 16;  // the reference's tile edge (config.h:16-17): radii / num_rendered semantics
constexpr int kAccumStride = 16;  // floats per Gaussian in the gradient accumulator (one 64-B line)
constexpr int kXcds = 8;          // binning counters are privatised per XCD (indexed by HW_REG_XCC_ID & 7)
constexpr int kSubWords = 8;      // per-tile sub-list table: start (within the tile's list) of each XCD's sub-list
// Units in which many of the 64 x 64 (pixel, record) pairs pass are cheaper in the all-pairs form (wave-uniform record
// reads, independent alpha evaluations in flight, no divergent walk).
constexpr uint32_t kDensePairsFwd = 1000;  // local forward blend: all-pairs loop above this many mask bits per unit
constexpr uint32_t kDensePairsBwd = 1000;  // backward: the same choice inside k_unit_blend_bwd_sparse
#ifndef FR_PRE_WG
#define FR_PRE_WG 128
#endif
constexpr int kPreWG = FR_PRE_WG;   // threads per workgroup of k_preprocess_fwd
constexpr uint32_t kBucketCapInit = 64;  // initial capacity of a (tile, XCD) key bucket; grows (power of two) on overflow
constexpr int kSortWaveMax = 256;    // longest tile list one wave sorts alone in registers (4 keys per lane)
constexpr int kSortGroupMax = 2048;  // longest list k_tile_sort handles (4 waves x 8 keys per lane).  (1 024 until round 6: the reference's own
                                     // initialisation at 100 k Gaussians has lists of 1 241 entries, and the big-list sorter on its side stream cost
                                     // that step 20 us a frame)
constexpr int kSortRegMax = 4096;    // longest list k_tile_sort_big sorts in registers (4 waves x 16 keys per lane);
                                     // longer ones take its global-memory fallback

// accumulator slots (blend backward -> preprocess backward).  With q = dL/dG * G of a (pixel, Gaussian) pair
// and d = splat centre - pixel, the slots hold the sums over pixels of:
//   MX: q*gx   MY: q*gy   CA: q*dx*dx   CB: q*dx*dy   CC: q*dy*dy   OP: q   R,G,B: alpha*T*dL/dpixel
// with (gx, gy) = -(a dx + b dy, b dx + c dy) = (dG/d delx, dG/d dely) / G: the pair's offset combined with the conic PER PIXEL,
// as the reference does (backward.cu:540-546), and with the reference's conic itself (the record stores it scaled by powers
// of two only) — summing q*dx and q*dy and combining afterwards lets the two products of an elongated splat cancel only after
// N pixels' worth of rounding, and a conic whose entries were rounded independently puts its rounding into that cancellation.
// k_preprocess_bwd turns them into the reference's dL_dmean2D / dL_dconic / dL_dopacity / dL_dcolor.
enum { ACC_MX = 0, ACC_MY = 1, ACC_CA = 2, ACC_CB = 3, ACC_CC = 4, ACC_OP = 5, ACC_R = 6, ACC_G = 7, ACC_B = 8 };

extern thread_local char g_err[512];
int fail_hip(hipError_t e, const char* what, const char* file, int line);
int fail_msg(int code, const char* msg);

#define FR_HIP(expr)                                                          \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return ::fr::fail_hip(_e, #expr, __FILE__, __LINE__); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
__host__ __device__ static inline T* carve(char*& p, size_t count)
{
    // (pointer arithmetic, not an integer round trip: inside a kernel the compiler then still knows that the result points
    // to GLOBAL memory and emits global_load / global_store — a pointer rebuilt from an integer is generic: flat_* accesses,
    // which also count against lgkmcnt and so tie every LDS wait to the outstanding memory loads)
    T* r = reinterpret_cast<T*>(p + ((uintptr_t(0) - reinterpret_cast<uintptr_t>(p)) & uintptr_t(255)));
    p = reinterpret_cast<char*>(r + count);
    return r;
}

// Per-Gaussian state written by the forward preprocess and read by every later stage.
struct GeomView {
    float4* rec_tmpl;       // [P*3] the 48-byte splat RECORD the blend kernels gather for every (tile, Gaussian) instance:
                            // (x, y, a', b'), (c', opacity, r, g), (b, id, depth, 0) with the pixel-space centre, the
                            // conic as (a', b', c') = (-0.5 a, -b, -0.5 c) (fr_blend.hip: pair_power) and the colour fed to the blend (SH result or
                            // colors_precomp) — ONE 48-byte gather per instance, not three
    float* opacity_act;     // [P] the activated opacity (the backward's API does not take the opacities; the conic is NOT stored:
                            // k_preprocess_bwd computes the 2D covariance again anyway and inverts it with the forward's expressions)
    uint8_t* clamped;       // [P] bit c set if SH colour channel c was clamped at 0
    float* dcolor_ddir;     // [P*9] d(SH colour)/d(view direction): (dR,dG,dB)/dx, /dy, /dz — computed by the forward, which has
                            // the 48 coefficients in LDS anyway, so that the backward reads 36 bytes per Gaussian instead
                            // of its 192-byte SH row (the largest array of the frame) a second time
    float* accum;           // [P*kAccumStride] gradient accumulators of the blend backward.  NOT part of the geometry
                            // buffer: they belong to the handle (fr_handle_impl::accum), are all zero between
                            // backward passes (k_preprocess_bwd zeroes each row after reading it) and so cost the
                            // forward no zeroing writes; launch_forward / launch_backward point this member at them
    uint32_t* block_ref_tiles;  // [ceil(P/kPreWG)] per-workgroup sums of the reference-semantics tiles_touched
    static GeomView make(void* buf, size_t P)
    {
        char* p = static_cast<char*>(buf);
        GeomView g;
        g.rec_tmpl = carve<float4>(p, P * 3);
        g.opacity_act = carve<float>(p, P);
        g.clamped = carve<uint8_t>(p, P);
        g.dcolor_ddir = carve<float>(p, P * 9);
        g.accum = nullptr;
        g.block_ref_tiles = carve<uint32_t>(p, (P + kPreWG - 1) / kPreWG + 1);
        return g;
    }
    static size_t bytes(size_t P)
    {
        char* p = nullptr;
        GeomView g = make(p, P);
        return reinterpret_cast<size_t>(g.block_ref_tiles + (P + kPreWG - 1) / kPreWG + 1) + 256;
    }
};

struct DeviceCounts {  // lives at the head of the image buffer
    uint32_t num_rendered;   // reference semantics
    uint32_t num_instances;
    uint32_t max_tile_list;
    uint32_t overflow;
    uint32_t large_tiles;    // number of tiles whose list exceeds kSortRegMax
    uint32_t max_bucket;     // largest (tile, XCD) list: above ImageView::bucket_cap the key buckets overflowed
    uint32_t medium_tiles;   // number of tiles sorted cooperatively by a 4-wave workgroup
    uint32_t big_tiles;      // number of tiles with kSortGroupMax < entries <= kSortRegMax (4 waves x 16 keys per lane)
    uint32_t pad2;
    uint32_t pair_hist[5];   // FR_DEBUG_PAIR_HIST=1 only: units by pairs named (<= 500, <= 1000, <= 1500, <= 2500, more)
    uint32_t num_units;      // total number of blend units (64-record segments of tile lists)
    uint32_t capacity;       // binning capacity of this frame (the backward re-derives the binning layout from it)
};

struct ImageView {
    DeviceCounts* counts;
    // The per-tile counters do NOT live in the image buffer: they belong to the handle (fr_handle_impl::tile_counters),
    // are zero between frames (k_tile_totals re-zeroes tile_count after reading it) and so need no zeroing launch per
    // frame.  launch_forward points tile_count / buckets at them.
    // They are PRIVATE PER XCD: a counting atomic from XCD x goes to copy x, so a counter's cache line stays in one
    // XCD's L2 instead of bouncing between the eight (device-scope atomics from several XCDs on one line serialise
    // at the fabric); a tile's segment is the concatenation of its eight per-XCD sub-segments.
    uint32_t* tile_count;    // [kXcds][tpad] instances counted by XCD x (block-major, counter_index)
    uint64_t* buckets;       // [T][kXcds][bucket_cap] the keys of the instances, written by the counting pass itself:
                             // slot s of (tile, XCD x) is the s-th instance XCD x counted into the tile (handle-owned)
    uint32_t bucket_cap;
    uint32_t* tile_sub;      // [T][kSubWords] start, within the tile's list, of each XCD's sub-list
    uint32_t tpad;           // row pitch of the two counter arrays
    uint32_t* tile_total;    // [T rounded up to 16] instances per tile (sum over the XCD copies)
    uint32_t* tile_offset;   // [T+1] first record of the tile's list (allocated by k_tile_totals, in no particular tile order)
    uint32_t* large_list;    // [T]   ids of tiles with more than kSortRegMax entries
    uint32_t* medium_list;   // [T]   ids of tiles with kSortWaveMax < entries <= kSortGroupMax (sorted by 4 waves)
    uint32_t* big_list;      // [T]   ids of tiles with kSortGroupMax < entries <= kSortRegMax
    uint32_t* unit_offset;   // [T+1] first blend unit of each tile (ceil(tile_total / 64) consecutive units, allocated likewise)
    float* final_T;          // [W*H]
    uint32_t* n_contrib;     // [W*H] index+1 (in the 8x8 tile list) of the last blended entry
    int tiles_x, tiles_y;
    // position of tile (tx, ty)'s counter inside one XCD copy: block-major, 4x4 tiles per 64-byte line
    __host__ __device__ uint32_t counter_index(uint32_t tx, uint32_t ty) const
    {
        const uint32_t bx = (uint32_t)(tiles_x + 3) / 4;
        return ((ty >> 2) * bx + (tx >> 2)) * 16u + ((ty & 3u) << 2 | (tx & 3u));
    }
    static ImageView make(void* buf, int W, int H)
    {
        char* p = static_cast<char*>(buf);
        ImageView v;
        v.tiles_x = (W + kTile - 1) / kTile;
        v.tiles_y = (H + kTile - 1) / kTile;
        size_t T = (size_t)v.tiles_x * v.tiles_y;
        v.counts = carve<DeviceCounts>(p, 1);
        v.tile_count = nullptr, v.buckets = nullptr, v.bucket_cap = 0;
        // counters are stored in 4x4-tile blocks (one 64-byte line per block): lanes of one atomic instruction that
        // fall into the same line are merged into ONE request (27 requests/ns vs 250 lane-atomics/ns,
        // tools/micro_atomics.hip), and the tiles of one Gaussian's rectangle are 2-D neighbours
        v.tpad = (uint32_t)(((size_t)((v.tiles_x + 3) / 4) * ((v.tiles_y + 3) / 4)) * 16);
        v.tile_sub = carve<uint32_t>(p, T * kSubWords);
        v.tile_total = carve<uint32_t>(p, (T + 15) & ~(size_t)15);
        v.tile_offset = carve<uint32_t>(p, T + 1);
        v.large_list = carve<uint32_t>(p, T);
        v.medium_list = carve<uint32_t>(p, T);
        v.big_list = carve<uint32_t>(p, T);
        v.unit_offset = carve<uint32_t>(p, T + 1);
        v.final_T = carve<float>(p, (size_t)W * H);
        v.n_contrib = carve<uint32_t>(p, (size_t)W * H);
        return v;
    }
    static size_t bytes(int W, int H)
    {
        ImageView v = make(nullptr, W, H);
        return reinterpret_cast<size_t>(v.n_contrib + (size_t)W * H) + 256;
    }
};

constexpr int kUnit = 64;  // records per blend unit (= one LDS batch of one wavefront)

// Batched frames (fr_forward_batch / fr_backward_batch): up to kMaxBatch views share every launch of the frame; a
// launch's grid is (largest per-view grid, views) and a workgroup takes the argument block of view blockIdx.y.
constexpr int kMaxBatch = 4;
template <typename A>
struct BatchOf {
    A v[kMaxBatch];
};

// A blend UNIT is one 64-record segment of one tile's sorted list: the independent work item of the
// blend kernels.  Per unit and pixel (lane) the forward leaves what the other passes need.
// The blend BACKWARD's work list: slot w = the unit wave w takes.  The forward blend fills it, heavy units (long walks)
// from the front and light ones from the back: the waves of a dispatch start over several microseconds, in slot order,
// and the kernel ends with its slowest unit — the long ones go first.
struct BwdUnit {
    uint4 d;       // the unit's descriptor (BinningView::unit_tile)
    uint32_t u;    // the unit
    uint32_t pad[3];
};
// The list is filled without a global cursor (four thousand returning atomics on one address take 100 us): unit u
// belongs to stripe u % kStripes, which owns the slots j, j + kStripes, j + 2 kStripes ... and has a heavy cursor (from
// its front) and a light cursor (from its back), each on a cache line of its own.
constexpr uint32_t kStripes = 64;
constexpr uint32_t kStripeWords = 16;   // words between two cursors

struct BinningView {
    size_t cap, unit_cap;
    BwdUnit* bwd_units;   // [unit_cap]
    uint32_t* stripe_cursor;   // [kStripes * 2 * kStripeWords] (zeroed by k_tile_sort)
    uint64_t* keys;       // [cap] (depth_bits << 32 | gaussian id), grouped per tile, unsorted
    uint32_t* ids;        // [cap]   the Gaussian id of every (tile, Gaussian) instance, per tile in blend order (depth, then
                          //         id): the blend kernels gather the 48-byte record from GeomView::rec_tmpl through it
    uint2* masks;         // [cap]   footprint mask of every record (k_unit_blend_chained writes it, the backward reads it)
    uint2* walks;         // [unit_cap*64] the same bits pixel-major: bit j of word u*64 + p = record j of unit u names pixel p
                          //         (the forward has the 64 x 64 transpose in registers; the backward walks by pixel)
    uint4* unit_tile;     // [unit_cap] descriptor of each unit: (tile y << 16 | tile x, segment, list start, list length)
    uint32_t* unit_done;  // [unit_cap] k_unit_blend_chained: the unit's final contribution is in memory (zeroed by k_tile_sort)
    float* unit_tseg;     // [unit_cap*64]   product of (1-alpha) over the unit's blendable records, per pixel
    float* unit_out;      // [unit_cap*5*64] forward partials per pixel: Cr, Cg, Cb, T_out, (last | done<<31)
    float4* unit_state;   // [unit_cap*64]   backward entry state per pixel: colour behind the unit / T_out, T_out
    __host__ __device__ static size_t units_for(size_t cap, size_t T) { return cap / kUnit + T + 1; }
    __host__ __device__ static BinningView make(void* buf, size_t cap, size_t T)
    {
        char* p = static_cast<char*>(buf);
        BinningView b;
        b.cap = cap;
        b.unit_cap = units_for(cap, T);
        // (the backward's work list comes FIRST: its address does not depend on the capacity, so the blend backward, which
        // learns the capacity from the device counts, can request a unit's descriptor together with the counts)
        b.bwd_units = carve<BwdUnit>(p, b.unit_cap);
        b.unit_tile = carve<uint4>(p, b.unit_cap);
        b.stripe_cursor = carve<uint32_t>(p, kStripes * 2 * kStripeWords);
        b.keys = carve<uint64_t>(p, cap);
        b.masks = carve<uint2>(p, cap);
        b.walks = carve<uint2>(p, b.unit_cap * kUnit);
        b.ids = carve<uint32_t>(p, cap);
        b.unit_done = carve<uint32_t>(p, b.unit_cap);
        b.unit_tseg = carve<float>(p, b.unit_cap * kUnit);
        b.unit_out = carve<float>(p, b.unit_cap * 5 * kUnit);
        b.unit_state = carve<float4>(p, b.unit_cap * kUnit);
        return b;
    }
    static size_t bytes(size_t cap, size_t T)
    {
        BinningView b = make(nullptr, cap, T);
        return reinterpret_cast<size_t>(b.unit_state + b.unit_cap * kUnit) + 256;
    }
};

// Where the blend kernels read the record at position `pos` of the sorted lists: the Gaussian's template record
// (GeomView::rec_tmpl, 4.8 MB at 100 k Gaussians: L2 / MALL resident), found through the sorted id.  (Round 2 had
// k_tile_sort gather the template and write a 48-byte copy per instance: 3.4 us more in the sort, 1.5 us less in each
// blend kernel, but 10 MB more written and read per frame — with frames in flight together the copies lost by 3-5 %.)
struct RecSrc {
    const uint32_t* ids;
    const float4* tmpl;
    __device__ __forceinline__ const float4* at(size_t pos) const { return tmpl + (size_t)ids[pos] * 3; }
};

enum Stage { ST_PREPROCESS_FWD = 0, ST_SCAN, ST_EMIT, ST_SORT, ST_BLEND_FWD, ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_COUNT };

struct StageEvents {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
};

struct fr_handle_impl {
    int device;
    fr_counts* host_counts;      // pinned, mapped
    fr_counts* host_counts_dev;  // device view of the same memory
    hipEvent_t counts_ready;
    bool counts_seen = false;    // host_counts holds the counts of a completed frame of this handle
    // per-tile instance counters (see ImageView): device memory owned by the handle, all zero between frames
    uint32_t* tile_counters = nullptr;
    size_t tile_counter_tiles = 0;
    bool counters_clean = false;
    // key buckets (ImageView::buckets): [tiles][8 XCDs][bucket_cap] u64, handle-owned; the capacity doubles when a
    // frame reports a (tile, XCD) count above it (that frame is invalid and the caller repeats it)
    uint64_t* key_buckets = nullptr;
    size_t bucket_tiles = 0;
    uint32_t bucket_cap = 0;
    // gradient accumulators of the blend backward (GeomView::accum): device memory owned by the handle, all zero
    // between backward passes
    float* accum = nullptr;
    size_t accum_rows = 0;
    // Frames of one handle share those counters, so they must not overlap on the device.  Frames enqueued on ONE
    // stream are ordered by it; when the stream changes, the new frame first waits for `frame_done`, recorded
    // behind the last kernel that touches the counters of the previous frame (not while a stream is being captured:
    // a capture is ordered by its own stream, and replays of the graph are ordered by whoever launches them).
    hipEvent_t frame_done = nullptr;
    // the big-list sorter (lists > 1024, only launched for frames that have had one) runs NEXT TO k_tile_sort on this
    // stream: fork after the totals kernel, join before the blend (captured as a parallel branch in a HIP graph)
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    // ... and their backward passes share the gradient accumulators: a backward enqueued on a different stream than the
    // handle's previous backward first waits for `bwd_done`, recorded behind that one's last kernel
    hipEvent_t bwd_done = nullptr;
    hipStream_t last_bwd_stream = nullptr;
    bool have_last_bwd = false;
    // A captured graph bakes the handle's buffer pointers (and the bucket capacity) into its kernel arguments.  Once a
    // frame of this handle has been captured, outgrown buffers are therefore RETIRED, not freed: replays keep working on
    // the buffers they were captured with (every invariant — counters and accumulators zero between frames — holds per
    // buffer), eager frames use the new ones, and fr_destroy frees them all.  Capacities grow geometrically, so the
    // retired memory stays below the live memory.
    bool captured = false;
    std::vector<void*> retired;
    uint32_t dense_pairs_fwd = 0, dense_pairs_bwd = 0;  // per-unit pair counts above which the all-pairs loops take a unit (FR_DENSE_PAIRS_FWD / _BWD)
    bool debug_pair_hist = false;
    uint32_t heavy_pairs = 641;  // a unit that names at least this many pairs goes to the FRONT of the backward work list (FR_HEAVY_PAIRS; default = more than the pair slots of k_unit_blend_bwd_sparse hold at once, FR_PAIR_CAP + 1)
    uint32_t chain_spins = 1u << 16;   // polls before a blend unit stops waiting for another one and computes its product / row itself (FR_CHAIN_SPINS)
    bool gather_in_chain = true;    // FR_BLEND_FWD=gather: a separate k_tile_gather launch instead of the tile's last unit gathering
    bool profiling = false;      // fr_profile_enable: bracket every stage launch with HIP events
    StageEvents ev[ST_COUNT];
};

// Brackets one kernel launch with HIP events on its stream when profiling is on.
struct StageScope {
    fr_handle_impl* h;
    int st;
    hipStream_t s;
    StageScope(fr_handle_impl* h_, int st_, hipStream_t s_) : h(h_), st(st_), s(s_)
    {
        if (!h || !h->profiling) { h = nullptr; return; }
        StageEvents& e = h->ev[st];
        if (e.used == e.start.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { h = nullptr; return; }
            e.start.push_back(a), e.stop.push_back(b);
        }
        (void)hipEventRecord(e.start[e.used], s);
    }
    ~StageScope()
    {
        if (!h) return;
        StageEvents& e = h->ev[st];
        (void)hipEventRecord(e.stop[e.used], s);
        e.used++;
    }
};

// The next (start, stop) event pair of a stage, for launches that record their own events (hipExtLaunchKernelGGL takes
// them from the dispatch packet: the kernel's execution alone, as a profiler sees it, without the launch path).
static inline bool next_stage_events(fr_handle_impl* h, int st, hipEvent_t* a, hipEvent_t* b)
{
    if (!h || !h->profiling) return false;
    StageEvents& e = h->ev[st];
    if (e.used == e.start.size()) {
        hipEvent_t x, y;
        if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return false;
        e.start.push_back(x), e.stop.push_back(y);
    }
    *a = e.start[e.used], *b = e.stop[e.used];
    e.used++;
    return true;
}

// ---- stage launchers (defined in the .hip files) ----
// One view of a (possibly batched) frame as the launchers see it: its handle, the caller's arguments and buffers.
struct ForwardCall {
    fr_handle_impl* h;
    const fr_params* prm;
    const fr_inputs* in;
    float* out_color;
    int32_t* radii;
    void* geometry;
    void* image;
    void* binning;
    uint64_t cap;
    fr_counts* counts;   // may be null
};
struct BackwardCall {
    fr_handle_impl* h;
    const fr_params* prm;
    const fr_inputs* in;
    const int32_t* radii;
    void* geometry;
    const void* image;
    const void* binning;
    const float* dL_dpix;
    const fr_grads* grads;
};
// n views (1 .. kMaxBatch) through the SAME launches: n == 1 uses the plain kernels, n > 1 their *_batch twins with a
// (grid, n) launch.  Every view has its own handle and buffers; the views' frames are independent of each other.
int launch_forward(int n, const ForwardCall* calls, hipStream_t s);
int launch_backward(int n, const BackwardCall* calls, hipStream_t s);
// what launch_forward has prepared per view when it hands over to the sort + blend launches (fr_blend.hip)
struct FrameView {
    fr_handle_impl* h;
    const fr_params* prm;
    const fr_inputs* in;
    GeomView g;
    ImageView v;
    BinningView b;
    float* out_color;
};
// kernel for one view, its *_batch twin for several: the twin takes BatchOf<A> and a (gx, n) grid
template <typename A, typename KS, typename KB>
static inline void launch_views(KS single, KB batch, int n, const A* args, uint32_t gx, uint32_t threads, size_t lds, hipStream_t s)
{
    if (n == 1) {
        hipLaunchKernelGGL(single, dim3(gx), dim3(threads), lds, s, args[0]);
    } else {
        BatchOf<A> b;
        for (int k = 0; k < kMaxBatch; k++) b.v[k] = args[k < n ? k : 0];
        hipLaunchKernelGGL(batch, dim3(gx, (uint32_t)n), dim3(threads), lds, s, b);
    }
}
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
int launch_knn(int P, const float* points, float* out, void* ws, size_t ws_bytes, hipStream_t s, int mode);
size_t knn_workspace_bytes(int P);
// zero `bytes` (multiple of 16, 16-byte aligned) with a plain kernel.  hipMemsetAsync is avoided on purpose: as a
// memset NODE of a captured graph it stopped taking effect once an eager kernel had been launched between two
// replays (ROCm 7.0 runtime shipped with PyTorch 2.10; reproduced with tools/dbg_graph.py).
int launch_zero(void* ptr, size_t bytes, hipStream_t s);
// make h->accum hold at least P zeroed rows (hipMalloc when it grows: not while the stream is being captured)
int ensure_accum(fr_handle_impl* h, size_t P, hipStream_t s);
// remember that frames of this handle are being captured into a graph (see fr_handle_impl::captured); true while capturing
bool note_capture(fr_handle_impl* h, hipStream_t s);
// give up a handle-owned device buffer: freed behind the stream's work, or retired if a captured graph may still name it
int release_buffer(fr_handle_impl* h, void* p, hipStream_t s);
struct BindArgs;
BindArgs bind_args(const fr_binding& b);   // (fr_bind_math.hpp / fr_binding.hip)
int launch_bind_forward(const fr_binding& b, float* xyz, float* rot, float* scale, hipStream_t s);
int launch_bind_backward(const fr_binding& b, const float* g_xyz, const float* g_rot, const float* g_scale, float* d_verts,
                         float* d_offset, float* d_rotation, float* d_scaling, hipStream_t s);
int launch_face_scale(int F, const float* verts, const int* faces, float* out, hipStream_t s);
int launch_adam(const fr_adam_config& cfg, float* param, const float* const* grad_bufs, int n_grads, float* exp_avg,
                float* exp_avg_sq, unsigned long long n, float* state, hipStream_t s);
int launch_l1_loss_grad_batch(int n_images, unsigned long long n, const float* const* img, const float* const* gt, float* const* grad,
                              float* const* loss, void* const* workspace, hipStream_t s);
int launch_l1_loss_grad(unsigned long long n, const float* img, const float* gt, float* grad, float* loss, void* workspace,
                        hipStream_t s);
int launch_scaled_sum(int n_src, const float* const* src, float* dst, unsigned long long count, float scale, hipStream_t s);
int launch_multi_copy(int n_seg, float* const* dst, const float* const* src, const unsigned long long* count, hipStream_t s);
int launch_selftest_reduce(const float* in, float* out, hipStream_t s);

#if defined(__HIPCC__)
// ---------------------------------------------------------------- device helpers
// the reference GaussianModel's activations (gaussian_model.py:39-50), applied in-kernel when
// FR_FLAG_RAW_ACTIVATIONS is set
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float act_exp(float x) { return expf(x); }
__device__ __forceinline__ float act_rot_inv_norm(float r, float x, float y, float z)
{
    return 1.0f / fmaxf(sqrtf(r * r + x * x + y * y + z * z), 1e-12f);  // torch.nn.functional.normalize, eps 1e-12
}

// Sigma3D = R S^2 R^T as its 6 upper-triangular floats (forward.cu:118-152) from the ACTIVATED scale and unit quaternion.
// Both per-Gaussian kernels call this: the forward does not store the matrix (24 bytes per Gaussian written and read
// back), the backward computes the same bits again from the scales and rotations it reads anyway.
__device__ __forceinline__ void cov3d_from_scale_rot(float sc0, float sc1, float sc2, float r, float x, float y, float z,
                                                     float scale_modifier, float (&c3)[6])
{
    const float s0 = scale_modifier * sc0, s1 = scale_modifier * sc1, s2 = scale_modifier * sc2;
    // rows of the rotation matrix, each entry scaled by the scale of its COLUMN index
    const float m00 = s0 * (1.f - 2.f * (y * y + z * z)), m01 = s1 * (2.f * (x * y - r * z)), m02 = s2 * (2.f * (x * z + r * y));
    const float m10 = s0 * (2.f * (x * y + r * z)), m11 = s1 * (1.f - 2.f * (x * x + z * z)), m12 = s2 * (2.f * (y * z - r * x));
    const float m20 = s0 * (2.f * (x * z - r * y)), m21 = s1 * (2.f * (y * z + r * x)), m22 = s2 * (1.f - 2.f * (x * x + y * y));
    c3[0] = m00 * m00 + m01 * m01 + m02 * m02;
    c3[1] = m10 * m00 + m11 * m01 + m12 * m02;
    c3[2] = m20 * m00 + m21 * m01 + m22 * m02;
    c3[3] = m10 * m10 + m11 * m11 + m12 * m12;
    c3[4] = m20 * m10 + m21 * m11 + m22 * m12;
    c3[5] = m20 * m20 + m21 * m21 + m22 * m22;
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// HW fp32 atomic add (global_atomic_add_f32), no return value needed.
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// The frame's camera (view 4x4, projection 4x4, position) as wave-uniform values: every lane fetches ONE of the 35
// floats (one memory round trip, issued together with the kernel's other input loads) and v_readlane broadcasts
// them to scalar registers.  Reading the matrices through their pointers instead cost the per-Gaussian kernels a
// dependent global load (= a serialised round trip) at every use.
struct CameraRegs {
    float view[16], proj[16], campos[3];
};
__device__ __forceinline__ CameraRegs load_camera(const float* view, const float* proj, const float* campos, int lane)
{
    float x = 0.f;
    if (lane < 16) x = view[lane];
    else if (lane < 32) x = proj[lane - 16];
    else if (lane < 35) x = campos[lane - 32];
    const int xi = __float_as_int(x);
    CameraRegs c;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        c.view[k] = __int_as_float(__builtin_amdgcn_readlane(xi, k));
        c.proj[k] = __int_as_float(__builtin_amdgcn_readlane(xi, 16 + k));
    }
#pragma unroll
    for (int k = 0; k < 3; k++) c.campos[k] = __int_as_float(__builtin_amdgcn_readlane(xi, 32 + k));
    return c;
}

// reference auxiliary.h:58-77 — matrices are indexed column-major
__device__ __forceinline__ float3 xform4x3(const float3 p, const float* m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const float* m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
#endif

}  // namespace fr
