// simple-knn replacement for gfx950: out[i] = mean of the 3 smallest squared distances from
// points[i] to the other points (reference: submodules/simple-knn/simple_knn.cu:148-222).
//
// The reference orders points along a Morton curve and prunes 1024-point boxes; the result is
// the exact 3-NN, so any exact search gives the same numbers.  Here: bucket the points into a
// dense uniform grid (atomic histogram -> scan -> scatter, the same pattern as the rasterizer's
// tile binning), then every point scans grid shells of growing Chebyshev radius until the shell
// distance bound exceeds its third-best distance.  Distances use the reference's expression
// (dx*dx + dy*dy + dz*dz, no FMA contraction: this file is built with -ffp-contract=off) and the
// final (b0 + b1 + b2) / 3, so results are bit-identical to the CPU oracle.
#include "fr_common.hpp"
#include <cfloat>

namespace fr {

struct KnnHeader {
    uint32_t bmin[3], bmax[3];  // order-preserving uint encodings of the bounding box
    uint32_t pad[2];
};

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void __launch_bounds__(256) k_knn_bbox(int P, const float* pts, KnnHeader* h)
{
    uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 3; k++) {
            const uint32_t o = f2ord(pts[3 * (size_t)i + k]);
            mn[k] = min(mn[k], o);
            mx[k] = max(mx[k], o);
        }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = min(mn[k], (uint32_t)__shfl_xor(mn[k], off));
            mx[k] = max(mx[k], (uint32_t)__shfl_xor(mx[k], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&h->bmin[k], mn[k]);
            atomicMax(&h->bmax[k], mx[k]);
        }
    }
}

struct KnnGrid {
    int G;               // cells per axis
    float ox, oy, oz;    // origin
    float cs, inv_cs;    // cell edge
};

__device__ __forceinline__ KnnGrid make_grid(const KnnHeader* h, int G)
{
    KnnGrid g;
    g.G = G;
    g.ox = ord2f(h->bmin[0]), g.oy = ord2f(h->bmin[1]), g.oz = ord2f(h->bmin[2]);
    const float ex = ord2f(h->bmax[0]) - g.ox, ey = ord2f(h->bmax[1]) - g.oy, ez = ord2f(h->bmax[2]) - g.oz;
    float e = fmaxf(ex, fmaxf(ey, ez));
    e = fmaxf(e, 1e-30f);
    g.cs = e / (float)G * 1.0001f;
    g.inv_cs = 1.0f / g.cs;
    return g;
}
__device__ __forceinline__ int cell_coord(float v, float o, const KnnGrid& g)
{
    int c = (int)((v - o) * g.inv_cs);
    return c < 0 ? 0 : (c >= g.G ? g.G - 1 : c);
}

__global__ void __launch_bounds__(256) k_knn_count(int P, const float* pts, const KnnHeader* h, int G, uint32_t* cell_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const KnnGrid g = make_grid(h, G);
    const int cx = cell_coord(pts[3 * (size_t)i], g.ox, g), cy = cell_coord(pts[3 * (size_t)i + 1], g.oy, g),
              cz = cell_coord(pts[3 * (size_t)i + 2], g.oz, g);
    atomicAdd(&cell_count[((size_t)cz * G + cy) * G + cx], 1u);
}

// exclusive scan of n counts by one workgroup; writes offsets[n] = total and cursor = offsets
__global__ void __launch_bounds__(1024) k_knn_scan(uint32_t n, const uint32_t* count, uint32_t* offset, uint32_t* cursor)
{
    __shared__ uint32_t s_sum[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = min(n, tid * per), e = min(n, b + per);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += count[i];
    s_sum[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t t = (tid >= off) ? s_sum[tid - off] : 0u;
        __syncthreads();
        s_sum[tid] += t;
        __syncthreads();
    }
    uint32_t run = s_sum[tid] - sum;
    for (uint32_t i = b; i < e; i++) {
        offset[i] = run;
        cursor[i] = run;
        run += count[i];
    }
    if (tid == 1023) offset[n] = s_sum[1023];
}

__global__ void __launch_bounds__(256) k_knn_scatter(int P, const float* pts, const KnnHeader* h, int G, uint32_t* cursor,
                                                     float4* sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const KnnGrid g = make_grid(h, G);
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    const int cx = cell_coord(x, g.ox, g), cy = cell_coord(y, g.oy, g), cz = cell_coord(z, g.oz, g);
    const uint32_t slot = atomicAdd(&cursor[((size_t)cz * G + cy) * G + cx], 1u);
    sorted[slot] = make_float4(x, y, z, __uint_as_float((uint32_t)i));
}

__device__ __forceinline__ void update3(float dist, float (&best)[3])
{
    // reference updateKBest<3>, simple_knn.cu:131-146
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
}

// mode 0: mean of the three smallest squared distances (distCUDA2); mode 1: the smallest one (the nearest other
// point: what pytorch3d's knn_points(p, p, K=6).dists[..., 1] is used for in model/fateavatar.py:597-608)
__global__ void __launch_bounds__(256) k_knn_search(int P, const KnnHeader* h, int G, const uint32_t* __restrict__ offset,
                                                    const float4* __restrict__ sorted, float* out, int mode)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= P) return;
    const KnnGrid g = make_grid(h, G);
    const float4 me = sorted[s];
    const int cx = cell_coord(me.x, g.ox, g), cy = cell_coord(me.y, g.oy, g), cz = cell_coord(me.z, g.oz, g);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int r = 0; r < G; r++) {
        // every point outside the cube of Chebyshev radius r-1 around (cx,cy,cz) is farther than (r-1)*cs
        if (r >= 2) {
            const float bound = (float)(r - 1) * g.cs * 0.999f;
            if (best[2] <= bound * bound) break;
        }
        const int z0 = max(0, cz - r), z1 = min(G - 1, cz + r);
        const int y0 = max(0, cy - r), y1 = min(G - 1, cy + r);
        const int x0 = max(0, cx - r), x1 = min(G - 1, cx + r);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const bool shell_row = (z == cz - r) || (z == cz + r) || (y == cy - r) || (y == cy + r);
                const size_t rowbase = ((size_t)z * G + y) * G;
                if (shell_row) {
                    // whole x-run of this row belongs to the shell: contiguous cells -> one point range
                    const uint32_t b = offset[rowbase + x0], e = offset[rowbase + x1 + 1];
                    for (uint32_t q = b; q < e; q++) {
                        if ((int)q == s) continue;
                        const float4 o = sorted[q];
                        const float dx = o.x - me.x, dy = o.y - me.y, dz = o.z - me.z;
                        update3(dx * dx + dy * dy + dz * dz, best);
                    }
                } else {
                    // only the two end cells x = cx-r and x = cx+r
                    for (int side = 0; side < 2; side++) {
                        const int x = side ? cx + r : cx - r;
                        if (x < 0 || x >= G || (side && r == 0)) continue;
                        const uint32_t b = offset[rowbase + x], e = offset[rowbase + x + 1];
                        for (uint32_t q = b; q < e; q++) {
                            if ((int)q == s) continue;
                            const float4 o = sorted[q];
                            const float dx = o.x - me.x, dy = o.y - me.y, dz = o.z - me.z;
                            update3(dx * dx + dy * dy + dz * dz, best);
                        }
                    }
                }
            }
        if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == G - 1 && y1 == G - 1 && z1 == G - 1) break;  // whole grid seen
    }
    out[__float_as_uint(me.w)] = mode == 1 ? best[0] : (best[0] + best[1] + best[2]) / 3.0f;
}

static int knn_grid_size(int P)
{
    int G = 8;
    while (G < 256 && (long long)G * G * G < (long long)P) G <<= 1;
    return G;
}

size_t knn_workspace_bytes(int P)
{
    const size_t G = (size_t)knn_grid_size(P), cells = G * G * G;
    return 256 + align_up(sizeof(KnnHeader), 256) + 3 * align_up((cells + 1) * 4, 256) +
           align_up((size_t)(P > 0 ? P : 1) * 16, 256);
}

int launch_knn(int P, const float* points, float* out, void* ws, size_t ws_bytes, hipStream_t s, int mode)
{
    if (P <= 0) return FR_OK;
    if (ws_bytes < knn_workspace_bytes(P)) return fail_msg(FR_ERR_INVALID_ARGUMENT, "knn workspace too small");
    const int G = knn_grid_size(P);
    const size_t cells = (size_t)G * G * G;
    char* p = static_cast<char*>(ws);
    KnnHeader* h = carve<KnnHeader>(p, 1);
    uint32_t* cell_count = carve<uint32_t>(p, cells + 1);
    uint32_t* cell_offset = carve<uint32_t>(p, cells + 1);
    uint32_t* cursor = carve<uint32_t>(p, cells + 1);
    float4* sorted = carve<float4>(p, (size_t)P);

    FR_HIP(hipMemsetAsync(h->bmin, 0xff, sizeof(h->bmin), s));
    FR_HIP(hipMemsetAsync(h->bmax, 0x00, sizeof(h->bmax), s));
    FR_HIP(hipMemsetAsync(cell_count, 0, (cells + 1) * 4, s));
    const int nb = (P + 255) / 256;
    hipLaunchKernelGGL(k_knn_bbox, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, P, points, h);
    hipLaunchKernelGGL(k_knn_count, dim3(nb), dim3(256), 0, s, P, points, h, G, cell_count);
    hipLaunchKernelGGL(k_knn_scan, dim3(1), dim3(1024), 0, s, (uint32_t)cells, cell_count, cell_offset, cursor);
    hipLaunchKernelGGL(k_knn_scatter, dim3(nb), dim3(256), 0, s, P, points, h, G, cursor, sorted);
    hipLaunchKernelGGL(k_knn_search, dim3(nb), dim3(256), 0, s, P, h, G, cell_offset, sorted, out, mode);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

}  // namespace fr
