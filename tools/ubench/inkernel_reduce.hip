// Mechanism microbenchmark (not part of the product; queued experiment): can the
// split-K reduce of the small conv layers happen INSIDE the GEMM kernel without the agent-scope
// release / acquire fences that cost +65 % when tried (an L2 write-back per workgroup on the 8-XCD part)?
//
// Model of one layer: T tiles of 64x64 fp32, S producer workgroups per tile; each producer "computes"
// its partial (deterministic small integers, so every sum is exact in fp32 and any stale read shows up
// as a mismatch), spins `work` iterations, and delivers it.  Variants:
//   0  baseline      partial -> ws with plain float4 stores; a second kernel sums the S slabs (+bias)
//   1  agent atomics partial -> ws with relaxed agent-scope atomic stores (write-through, nothing dirty
//                    in L2, so no write-back is needed), s_waitcnt vmcnt(0), relaxed agent-scope
//                    arrival counter; the last arriver sums with relaxed agent-scope atomic loads
//   2  same XCD      the S producers of a tile get hardware workgroup ids that are equal mod 8 (= one XCD,
//                    checked against HW_REG_XCC_ID); plain stores (the XCD's L2 is their coherence
//                    point), s_waitcnt vmcnt(0), agent-scope counter; the last arriver reads with
//                    workgroup-scope (sc0) atomic loads so that its own L1 cannot serve a line of the
//                    previous launch
//   3  as 2, with the last arriver reading four slabs per iteration into independent registers (the
//      r01p measurement showed variant 2's slab-after-slab loop costing ~1.5 us per split)
// Every launch uses new data (seed), the output is compared on the host after every R launches, and the
// reported time is per launch (HIP events, back-to-back launches on one stream).
// Build: hipcc --offload-arch=gfx950 -O3 -o inkernel_reduce inkernel_reduce.hip ; run: ./inkernel_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

constexpr int TILE = 64 * 64;  // floats per tile; 256 threads x 16 floats

__device__ __forceinline__ float part_value(int tile, int s, int idx, int seed) {
  return (float)(((tile * 131 + s * 17 + idx * 7 + seed * 29) & 63) - 31);  // |v| < 32: sums exact
}

__device__ __forceinline__ unsigned xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
  return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
}

__device__ __forceinline__ void spin(int work, float& sink) {
  for (int i = 0; i < work; ++i) sink = sink * 1.0000001f + 1e-9f;
}

// ---- variant 0 --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void produce_plain(float* __restrict__ ws, int T, int S, int seed, int work) {
  const int tile = blockIdx.x, s = blockIdx.y;
  float sink = 1.f;
  spin(work, sink);
  float4* dst = reinterpret_cast<float4*>(ws + ((size_t)s * T + tile) * TILE) + threadIdx.x * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = (threadIdx.x * 4 + j) * 4;
    dst[j] = make_float4(part_value(tile, s, idx, seed), part_value(tile, s, idx + 1, seed),
                         part_value(tile, s, idx + 2, seed), part_value(tile, s, idx + 3, seed) + (sink > 1e30f));
  }
}
__global__ __launch_bounds__(256) void reduce_plain(const float* __restrict__ ws, float* __restrict__ out, int T, int S) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index over T*TILE/4
  if (i >= (size_t)T * TILE / 4) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const float4 u = reinterpret_cast<const float4*>(ws + (size_t)s * T * TILE)[i];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  reinterpret_cast<float4*>(out)[i] = v;
}

// ---- variants 1 and 2 ---------------------------------------------------------------------------------
// ids: a 1-D grid of 8 * ceil(T/8) * S workgroups; hardware id h -> xcd slot c = h & 7, j = h >> 3,
// tile = (j / S) * 8 + c, split = j % S: all splits of a tile have ids that are equal mod 8.
template <int VARIANT>
__global__ __launch_bounds__(256) void produce_reduce(float* ws, float* __restrict__ out, unsigned* ctr,
                                                      unsigned* xcc_seen, unsigned* xcc_mismatch, int T, int S,
                                                      int seed, int work) {
  __shared__ unsigned last_flag;
  const int h = blockIdx.x, c = h & 7, j = h >> 3;
  const int tile = (j / S) * 8 + c, s = j % S;
  if (tile >= T) return;
  float sink = 1.f;
  spin(work, sink);
  float* mine = ws + ((size_t)s * T + tile) * TILE;
  if (VARIANT == 1) {
    // 8-byte relaxed agent-scope atomic stores: write-through, never dirty in this XCD's L2
    unsigned long long* d = reinterpret_cast<unsigned long long*>(mine) + threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = (threadIdx.x * 8 + k) * 2;
      const float a = part_value(tile, s, idx, seed), b = part_value(tile, s, idx + 1, seed) + (sink > 1e30f);
      const unsigned long long bits = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
      __hip_atomic_store(d + k, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    float4* d = reinterpret_cast<float4*>(mine) + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = (threadIdx.x * 4 + k) * 4;
      d[k] = make_float4(part_value(tile, s, idx, seed), part_value(tile, s, idx + 1, seed),
                         part_value(tile, s, idx + 2, seed), part_value(tile, s, idx + 3, seed) + (sink > 1e30f));
    }
    if (threadIdx.x == 0) xcc_seen[(size_t)tile * S + s] = xcc_id();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores are acknowledged
  __syncthreads();                                   // ... and everyone else's of this workgroup
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (old == (unsigned)S - 1);
    if (last_flag) __hip_atomic_store(ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning
  }
  __syncthreads();
  if (!last_flag) return;
  if (VARIANT >= 2 && threadIdx.x == 0) {
    const unsigned me = xcc_id();
    for (int t = 0; t < S; ++t)
      if (__hip_atomic_load(xcc_seen + (size_t)tile * S + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != me)
        atomicAdd(xcc_mismatch, 1u);
  }
  // the last arriver: fixed summation order s = 0 .. S-1
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  int t0 = 0;
  if (VARIANT == 3) {
    // four slabs in flight: 32 independent 8-byte loads per thread, then added in slab order
    for (; t0 + 4 <= S; t0 += 4) {
      unsigned long long b[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long* q =
            reinterpret_cast<const unsigned long long*>(ws + ((size_t)(t0 + u) * T + tile) * TILE) + threadIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) b[u][k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc[2 * k] += __uint_as_float((unsigned)b[u][k]);
          acc[2 * k + 1] += __uint_as_float((unsigned)(b[u][k] >> 32));
        }
    }
  }
  for (int t = t0; t < S; ++t) {
    const float* src = ws + ((size_t)t * T + tile) * TILE;
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(src) + threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long bits =
          VARIANT == 1 ? __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // variants 2, 3
      acc[2 * k] += __uint_as_float((unsigned)bits);
      acc[2 * k + 1] += __uint_as_float((unsigned)(bits >> 32));
    }
  }
  float4* o = reinterpret_cast<float4*>(out + (size_t)tile * TILE) + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
}

static long check(const std::vector<float>& got, int T, int S, int seed) {
  long bad = 0;
  for (int tile = 0; tile < T; ++tile)
    for (int idx = 0; idx < TILE; ++idx) {
      float want = 0.f;
      for (int s = 0; s < S; ++s) want += (float)(((tile * 131 + s * 17 + idx * 7 + seed * 29) & 63) - 31);
      bad += got[(size_t)tile * TILE + idx] != want;
    }
  return bad;
}

int main() {
  const int shapes[][2] = {{196, 6}, {104, 11}, {32, 16}, {392, 3}};  // (tiles, splits) of conv3 / conv4 / conv5 / conv2_2
  for (auto& sh : shapes) {
    const int T = sh[0], S = sh[1];
    float *ws, *out;
    unsigned *ctr, *seen, *mism;
    CHECK(hipMalloc(&ws, (size_t)S * T * TILE * 4));
    CHECK(hipMalloc(&out, (size_t)T * TILE * 4));
    CHECK(hipMalloc(&ctr, T * 4));
    CHECK(hipMalloc(&seen, (size_t)T * S * 4));
    CHECK(hipMalloc(&mism, 4));
    CHECK(hipMemset(ctr, 0, T * 4));
    CHECK(hipMemset(mism, 0, 4));
    std::vector<float> host((size_t)T * TILE);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int work : {0, 1500}) {  // 1500 spin iterations ~ a 20-30 us producer
      for (int variant = 0; variant < 4; ++variant) {
        long bad = 0;
        int seed = 1;
        auto launch = [&](int sd) {
          if (variant == 0) {
            hipLaunchKernelGGL(produce_plain, dim3(T, S), dim3(256), 0, 0, ws, T, S, sd, work);
            hipLaunchKernelGGL(reduce_plain, dim3((T * TILE / 4 + 255) / 256), dim3(256), 0, 0, ws, out, T, S);
          } else {
            const int G = 8 * ((T + 7) / 8) * S;
            if (variant == 1)
              hipLaunchKernelGGL((produce_reduce<1>), dim3(G), dim3(256), 0, 0, ws, out, ctr, seen, mism, T, S, sd, work);
            else if (variant == 2)
              hipLaunchKernelGGL((produce_reduce<2>), dim3(G), dim3(256), 0, 0, ws, out, ctr, seen, mism, T, S, sd, work);
            else
              hipLaunchKernelGGL((produce_reduce<3>), dim3(G), dim3(256), 0, 0, ws, out, ctr, seen, mism, T, S, sd, work);
          }
        };
        // correctness: 50 launches with fresh data, each checked
        for (int r = 0; r < 50; ++r, ++seed) {
          launch(seed);
          CHECK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
          bad += check(host, T, S, seed);
        }
        // time: 200 back-to-back launches
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 200; ++r) launch(seed + r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned mm = 0;
        CHECK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
        printf("T %3d S %2d work %5d  variant %d : %7.2f us per launch, %ld wrong values in 50 checked launches%s\n", T, S,
               work, variant, ms * 1e3 / 200, bad, variant >= 2 ? (mm ? "  [XCC_ID mismatch seen!]" : "  [same XCC_ID per tile]") : "");
      }
    }
    CHECK(hipFree(ws)); CHECK(hipFree(out)); CHECK(hipFree(ctr)); CHECK(hipFree(seen)); CHECK(hipFree(mism));
  }
  return 0;
}
