// Iso-surface extraction on the device: replaces the reference's
//   to_binary -> .dist file -> os.system("./isosurface/computeMarchingCubes f.dist f.obj -i iso")
// round trip (test/create_sdf.py:292-323) for the grid that is already resident in HBM.
//
// Indexed marching cubes: one vertex per cut grid edge (shared by the up to four cells around
// it), one index triple per triangle; case table derived in tools/gen_mc_tables.py (face-
// consistent, crack-free).  Order and arithmetic are those of oracle/mc_oracle.py, so the
// result is compared bit for bit (THIS FILE IS COMPILED WITH -ffp-contract=off):
//   vertex id  = rank of the cut edge in the flat order 3*p + axis, p = (iz*n + iy)*n + ix
//   position   = c0 + t*(c1 - c0) on the cut axis, t = (iso - v0)/(v1 - v0), float32
//   faces      = cells in flat order (iz,iy,ix) over R^3, triangles in table order
// All kernels are HBM-bound streaming passes over the (R+1)^3 grid; the scans are three-pass
// (block scan, scan of block sums, add).
#include "kernels.hpp"

#define DISN_MC_QUAL __constant__ const
#include "mc_tables.h"

namespace disn {

// ---------------------------------------------------------------------------
// exclusive scan of uint32, 4096 items per block
// ---------------------------------------------------------------------------
constexpr int kScanItems = 16;
constexpr int kScanBlock = 256 * kScanItems;

__global__ __launch_bounds__(256) void scan_block_kernel(const unsigned* __restrict__ in,
                                                         unsigned* __restrict__ out, size_t n,
                                                         unsigned* __restrict__ bsum) {
  __shared__ unsigned wsum[4];
  const size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * kScanItems;
  unsigned v[kScanItems];
  unsigned tsum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    tsum += v[i];
  }
  // inclusive scan of the 64 thread sums of a wave, then of the 4 wave sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = tsum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned u = __shfl_up(inc, off);
    if (lane >= off) inc += u;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  unsigned run = woff + inc - tsum;  // exclusive prefix of this thread
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 255) bsum[blockIdx.x] = woff + inc;
}

// one block: exclusive scan of the block sums in place, grand total -> *total
__global__ __launch_bounds__(1024) void scan_sums_kernel(unsigned* __restrict__ bsum, int nb,
                                                         unsigned long long* __restrict__ total) {
  __shared__ unsigned long long carry;
  __shared__ unsigned wsum[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned v = i < nb ? bsum[i] : 0u;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned u = __shfl_up(inc, off);
      if (lane >= off) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const unsigned long long c = carry;
    if (i < nb) bsum[i] = (unsigned)(c + woff + inc - v);
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void scan_add_kernel(unsigned* __restrict__ out, size_t n,
                                                       const unsigned* __restrict__ bsum) {
  const size_t base = (size_t)blockIdx.x * kScanBlock + (size_t)threadIdx.x * kScanItems;
  const unsigned add = bsum[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) out[base + i] += add;
}

static hipError_t exclusive_scan(const unsigned* in, unsigned* out, size_t n, unsigned* bsum,
                                 unsigned long long* total, hipStream_t st) {
  const int nb = (int)((n + kScanBlock - 1) / kScanBlock);
  hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(256), 0, st, in, out, n, bsum);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, bsum, nb, total);
  hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, st, out, n, bsum);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// pass 1: cut-edge flags (3 per grid point) and triangle counts (1 per cell)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mc_flags_kernel(const float* __restrict__ vol, int n,
                                                       float iso, unsigned* __restrict__ eflag,
                                                       unsigned* __restrict__ ccount) {
  const size_t total = (size_t)n * n * n;
  const int R = n - 1;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(p % n), iy = (int)((p / n) % n), iz = (int)(p / ((size_t)n * n));
    const bool in0 = vol[p] < iso;
    const bool hx = ix < R, hy = iy < R, hz = iz < R;
    const bool inx = hx ? vol[p + 1] < iso : in0;
    const bool iny = hy ? vol[p + n] < iso : in0;
    const bool inz = hz ? vol[p + (size_t)n * n] < iso : in0;
    eflag[3 * p + 0] = (hx && inx != in0) ? 1u : 0u;
    eflag[3 * p + 1] = (hy && iny != in0) ? 1u : 0u;
    eflag[3 * p + 2] = (hz && inz != in0) ? 1u : 0u;
    if (hx && hy && hz) {
      const size_t nn = (size_t)n * n;
      unsigned mask = in0 ? 1u : 0u;
      mask |= inx ? 2u : 0u;
      mask |= (vol[p + 1 + n] < iso) ? 4u : 0u;
      mask |= iny ? 8u : 0u;
      mask |= inz ? 16u : 0u;
      mask |= (vol[p + nn + 1] < iso) ? 32u : 0u;
      mask |= (vol[p + nn + 1 + n] < iso) ? 64u : 0u;
      mask |= (vol[p + nn + n] < iso) ? 128u : 0u;
      ccount[((size_t)iz * R + iy) * R + ix] = kMcNtri[mask];
    }
  }
}

__device__ __forceinline__ float grid_coord(const GridSpec& g, int a, int i) {
  double v = (double)i * g.step[a];
  v = v + g.start[a];
  if (i == g.res - 1 && g.res > 1) v = g.stop[a];
  return (float)v;
}

// pass 2: one vertex per cut edge
__global__ __launch_bounds__(256) void mc_verts_kernel(const float* __restrict__ vol, GridSpec g,
                                                       float iso, const unsigned* __restrict__ eflag,
                                                       const unsigned* __restrict__ eidx,
                                                       float* __restrict__ verts) {
  const int n = g.res;
  const size_t total = (size_t)3 * n * n * n;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    if (!eflag[e]) continue;
    const size_t p = e / 3;
    const int a = (int)(e - 3 * p);
    const int idx[3] = {(int)(p % n), (int)((p / n) % n), (int)(p / ((size_t)n * n))};
    const size_t step = a == 0 ? 1 : (a == 1 ? (size_t)n : (size_t)n * n);
    const float v0 = vol[p], v1 = vol[p + step];
    const float t = (iso - v0) / (v1 - v0);
    float pos[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[k] = grid_coord(g, k, idx[k]);
    const float c0 = pos[a], c1 = grid_coord(g, a, idx[a] + 1);
    const float d = t * (c1 - c0);
    pos[a] = c0 + d;
    float* o = verts + (size_t)eidx[e] * 3;
    o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
  }
}

// pass 3: index triples
__global__ __launch_bounds__(256) void mc_faces_kernel(const float* __restrict__ vol, int n,
                                                       float iso, const unsigned* __restrict__ coff,
                                                       const unsigned* __restrict__ eidx,
                                                       int* __restrict__ faces) {
  const int R = n - 1;
  const size_t cells = (size_t)R * R * R, nn = (size_t)n * n;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells;
       c += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(c % R), iy = (int)((c / R) % R), iz = (int)(c / ((size_t)R * R));
    const size_t p = ((size_t)iz * n + iy) * n + ix;
    unsigned mask = (vol[p] < iso) ? 1u : 0u;
    mask |= (vol[p + 1] < iso) ? 2u : 0u;
    mask |= (vol[p + 1 + n] < iso) ? 4u : 0u;
    mask |= (vol[p + n] < iso) ? 8u : 0u;
    mask |= (vol[p + nn] < iso) ? 16u : 0u;
    mask |= (vol[p + nn + 1] < iso) ? 32u : 0u;
    mask |= (vol[p + nn + 1 + n] < iso) ? 64u : 0u;
    mask |= (vol[p + nn + n] < iso) ? 128u : 0u;
    const int nt = kMcNtri[mask];
    if (!nt) continue;
    int* o = faces + (size_t)coff[c] * 3;
    for (int k = 0; k < 3 * nt; ++k) {
      const int e = kMcTri[mask][k];
      const size_t gp = ((size_t)(iz + kMcEdge[e][2]) * n + (iy + kMcEdge[e][1])) * n + (ix + kMcEdge[e][0]);
      o[k] = (int)eidx[3 * gp + kMcEdge[e][3]];
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct McWs {
  unsigned *eflag, *eidx, *ccount, *coff, *bsum;
  size_t total;
};

static McWs mc_layout(void* ws, int R) {
  const size_t n = (size_t)R + 1, ne = 3 * n * n * n, nc = (size_t)R * R * R;
  char* base = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = (off + 255) & ~size_t(255);
    unsigned* p = base ? reinterpret_cast<unsigned*>(base + off) : nullptr;
    off += bytes;
    return p;
  };
  McWs w;
  w.eflag = take(ne * 4);
  w.eidx = take(ne * 4);
  w.ccount = take(nc * 4);
  w.coff = take(nc * 4);
  w.bsum = take(((ne + kScanBlock - 1) / kScanBlock + 1) * 4);
  w.total = (off + 255) & ~size_t(255);
  return w;
}

size_t mc_ws_bytes(int R) { return mc_layout(nullptr, R).total; }

static inline int blocks_for(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  return (int)(b < 1 ? 1 : b);
}

hipError_t mc_count_launch(const float* vol, int R, float iso, unsigned long long* counts, void* ws,
                           hipStream_t st) {
  const McWs w = mc_layout(ws, R);
  const int n = R + 1;
  const size_t np = (size_t)n * n * n, nc = (size_t)R * R * R;
  hipLaunchKernelGGL(mc_flags_kernel, dim3(blocks_for(np)), dim3(256), 0, st, vol, n, iso, w.eflag,
                     w.ccount);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = exclusive_scan(w.eflag, w.eidx, 3 * np, w.bsum, counts, st)) != hipSuccess) return e;
  return exclusive_scan(w.ccount, w.coff, nc, w.bsum, counts + 1, st);
}

hipError_t mc_emit_launch(const float* vol, const GridSpec& g, float iso, float* verts, int* faces,
                          void* ws, hipStream_t st) {
  const int R = g.res - 1;
  const McWs w = mc_layout(ws, R);
  const size_t np = (size_t)g.res * g.res * g.res, nc = (size_t)R * R * R;
  hipLaunchKernelGGL(mc_verts_kernel, dim3(blocks_for(3 * np)), dim3(256), 0, st, vol, g, iso,
                     w.eflag, w.eidx, verts);
  hipLaunchKernelGGL(mc_faces_kernel, dim3(blocks_for(nc)), dim3(256), 0, st, vol, g.res, iso, w.coff,
                     w.eidx, faces);
  return hipGetLastError();
}

}  // namespace disn
