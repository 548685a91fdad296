// Micro-benchmark (gfx950): what a GRID-WIDE barrier costs inside one persistent kernel, against what the boundary
// between two kernels of a replayed hipGraph costs — the number that decides whether a persistent decoder-layer kernel
// (eight dependent phases per layer, 261 launches per decode step today) can pay.
//
//   (a) `barrier_kernel`: G workgroups (one or two per CU, launched cooperatively so that all are resident), each
//       phase = every workgroup writes one line of a buffer, crosses the barrier, reads the line of a workgroup on
//       ANOTHER XCD and checks it (so the barrier is timed with the release/acquire traffic a real phase change needs,
//       and a missing fence shows up as a wrong value, not as a good number).  Barrier = one device-scope atomic
//       counter, sense reversal by generation; a spin cap turns a would-be hang into an error exit.
//   (b) `phase_kernel` x N captured into a hipGraph and replayed: the same write / read-other-XCD work with a kernel
//       boundary in the place of the barrier.
//
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar {
  unsigned count;
  unsigned gen;
  unsigned error;
  unsigned pad;
};

// all threads of the workgroup call it; returns false when the spin cap was hit (the caller leaves the kernel)
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned G, unsigned& my_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                                 // release this workgroup's writes (device scope)
    const unsigned next = my_gen + 1;
    const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == G) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != next) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000u || __hip_atomic_load(&b->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&b->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
      }
    }
    __threadfence();                                                 // acquire the others' writes
  }
  my_gen += 1;
  ok = __syncthreads_and(ok ? 1 : 0) != 0;
  return ok;
}

// (c) the XCD-HIERARCHICAL form the guide prices at 4.1 / 5.9 / 9.7 us (MI355X_MICROARCH.md, row "barrier-xcd"; round 6,
//     second session — the verdicts of rounds 4 and 5 asked for this number on this code base): one counter per XCD, the
//     LAST arriver of an XCD is its leader: release fence (ONE L2 write-back per XCD: the XCD's workgroups share that L2 and
//     have drained their stores before arriving) -> top counter over the 8 leaders -> acquire fence -> the XCD's generation
//     word; every other workgroup polls ITS XCD's generation with relaxed L1-bypassing loads (served by the local L2) and
//     then takes its own acquire fence (its CU's L1).
struct XBar {
  unsigned count, gen, pad[30];       // one 128-byte line per XCD
};
struct HBar {
  XBar x[8];
  unsigned top_count, top_gen, error, pad;
};
__device__ __forceinline__ bool grid_barrier_xcd(HBar* b, unsigned G, unsigned& my_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned xcd = blockIdx.x & 7u;                            // workgroups go round-robin over the XCDs
    const unsigned per = G >> 3;                                     // (G is a multiple of 8)
    const unsigned next = my_gen + 1;
    XBar* xb = &b->x[xcd];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // this workgroup's stores have left for the L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned arrived = __hip_atomic_fetch_add(&xb->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    unsigned spins = 0;
    if (arrived == per) {                                            // the XCD's leader
      __hip_atomic_store(&xb->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // the XCD's dirty lines -> memory
      const unsigned a2 = __hip_atomic_fetch_add(&b->top_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (a2 == 8u) {
        __hip_atomic_store(&b->top_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&b->top_gen, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&b->top_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != next) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 20000000u || __hip_atomic_load(&b->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(&xb->gen, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&xb->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != next) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000u || __hip_atomic_load(&b->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (!ok) __hip_atomic_store(&b->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  my_gen += 1;
  ok = __syncthreads_and(ok ? 1 : 0) != 0;
  return ok;
}

// one phase of "work": a 256-byte line per workgroup, tagged with the phase number
__device__ __forceinline__ void phase_write(unsigned* buf, unsigned G, unsigned phase) {
  if (threadIdx.x < 64) buf[(size_t)blockIdx.x * 64 + threadIdx.x] = phase * 1000003u + blockIdx.x * 64 + threadIdx.x;
}
__device__ __forceinline__ unsigned phase_check(const unsigned* buf, unsigned G, unsigned phase) {
  // consecutive workgroup ids go round-robin over the 8 XCDs: +G/2+1 lands on another XCD for every G used here
  const unsigned other = (blockIdx.x + G / 2 + 1) % G;
  unsigned bad = 0;
  if (threadIdx.x < 64) {
    const unsigned v = __builtin_nontemporal_load(&buf[(size_t)other * 64 + threadIdx.x]);
    bad = v != phase * 1000003u + other * 64 + threadIdx.x;
  }
  return bad;
}

__global__ __launch_bounds__(256) void barrier_kernel(Bar* bar, unsigned* buf, unsigned* mism, long long* cyc, unsigned G,
                                                      int phases) {
  unsigned my_gen = __hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned bad = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int p = 1; p <= phases; ++p) {
    phase_write(buf, G, (unsigned)p);
    if (!grid_barrier(bar, G, my_gen)) return;
    bad += phase_check(buf, G, (unsigned)p);
    if (!grid_barrier(bar, G, my_gen)) return;     // (the line is rewritten next phase: readers first)
  }
  const long long t1 = __builtin_readcyclecounter();
  if (bad) atomicAdd(mism, bad);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void barrier_xcd_kernel(HBar* bar, unsigned* buf, unsigned* mism, long long* cyc, unsigned G,
                                                          int phases) {
  unsigned my_gen = __hip_atomic_load(&bar->top_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned bad = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int p = 1; p <= phases; ++p) {
    phase_write(buf, G, (unsigned)p);
    if (!grid_barrier_xcd(bar, G, my_gen)) return;
    bad += phase_check(buf, G, (unsigned)p);
    if (!grid_barrier_xcd(bar, G, my_gen)) return;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (bad) atomicAdd(mism, bad);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void phase_kernel(unsigned* buf, unsigned* mism, unsigned G, unsigned phase, int check) {
  if (check) {
    const unsigned bad = phase_check(buf, G, phase);
    if (bad) atomicAdd(mism, bad);
  } else {
    phase_write(buf, G, phase);
  }
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  printf("%s: %d CUs, cooperative launch %d\n", prop.name, cus, prop.cooperativeLaunch);
  Bar* bar;
  unsigned *buf, *mism;
  long long* cyc;
  const int GMAX = 1024;
  CHECK(hipMalloc(&bar, sizeof(Bar)));
  CHECK(hipMalloc(&buf, (size_t)GMAX * 64 * 4));
  CHECK(hipMalloc(&mism, 4));
  CHECK(hipMalloc(&cyc, GMAX * 8));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, barrier_kernel, 256, 0));
  printf("barrier_kernel: %d workgroups of 256 threads fit a CU\n", per_cu);
  const int phases = 2000;
  for (int G : {cus / 2, cus, 2 * cus}) {
    if (G > GMAX || G > per_cu * cus) continue;
    CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
    CHECK(hipMemsetAsync(mism, 0, 4, st));
    unsigned Gu = (unsigned)G;
    int ph = 20;
    void* args[] = {&bar, &buf, &mism, &cyc, &Gu, &ph};
    CHECK(hipLaunchCooperativeKernel((const void*)barrier_kernel, dim3(G), dim3(256), args, 0, st));   // warm
    ph = phases;
    CHECK(hipEventRecord(e0, st));
    CHECK(hipLaunchCooperativeKernel((const void*)barrier_kernel, dim3(G), dim3(256), args, 0, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb;
    unsigned hm = 0;
    CHECK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("persistent kernel, %4d workgroups: %.2f us per barrier (write + barrier + read another XCD's line + barrier = "
           "%.2f us per phase), stale reads %u, spin-cap exits %u\n",
           G, 1e3 * ms / (2.0 * phases), 1e3 * ms / phases, hm, hb.error);
  }

  // (c) the XCD-hierarchical barrier, same work, same check
  {
    HBar* hbar;
    CHECK(hipMalloc(&hbar, sizeof(HBar)));
    for (int G : {cus / 2, cus, 2 * cus, 4 * cus}) {
      if (G > GMAX || G > per_cu * cus || (G & 7)) continue;
      CHECK(hipMemsetAsync(hbar, 0, sizeof(HBar), st));
      CHECK(hipMemsetAsync(mism, 0, 4, st));
      unsigned Gu = (unsigned)G;
      int ph = 20;
      void* args[] = {&hbar, &buf, &mism, &cyc, &Gu, &ph};
      CHECK(hipLaunchCooperativeKernel((const void*)barrier_xcd_kernel, dim3(G), dim3(256), args, 0, st));   // warm
      CHECK(hipStreamSynchronize(st));
      ph = phases;
      CHECK(hipEventRecord(e0, st));
      CHECK(hipLaunchCooperativeKernel((const void*)barrier_xcd_kernel, dim3(G), dim3(256), args, 0, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      HBar hb;
      unsigned hm = 0;
      CHECK(hipMemcpy(&hb, hbar, sizeof(HBar), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
      printf("persistent kernel, %4d workgroups, XCD-hierarchical barrier: %.2f us per barrier (%.2f us per write + read "
             "phase), stale reads %u, spin-cap exits %u\n",
             G, 1e3 * ms / (2.0 * phases), 1e3 * ms / phases, hm, hb.error);
    }
    CHECK(hipFree(hbar));
  }

  // (b) the same work with kernel boundaries, replayed from a graph
  for (int G : {cus, 2 * cus}) {
    const int N = 400;   // phases per graph
    CHECK(hipMemsetAsync(mism, 0, 4, st));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 1; p <= N; ++p) {
      phase_kernel<<<G, 256, 0, st>>>(buf, mism, (unsigned)G, (unsigned)p, 0);
      phase_kernel<<<G, 256, 0, st>>>(buf, mism, (unsigned)G, (unsigned)p, 1);
    }
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, st));
    CHECK(hipStreamSynchronize(st));
    const int reps = 5;
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(exec, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned hm = 0;
    CHECK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("graph replay,      %4d workgroups: %.2f us per kernel boundary (%.2f us per write + read phase), stale reads %u\n",
           G, 1e3 * ms / (2.0 * N * reps), 1e3 * ms / (N * reps), hm);
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
  }
  return 0;
}
