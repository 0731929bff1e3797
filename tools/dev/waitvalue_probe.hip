// Development probe (GPU): do HIP's stream memory operations order a side stream after a value a KERNEL stores?
//   side stream:   hipStreamWaitValue32(sig >= 1) -> consumer kernel (stamps s_memrealtime, copies a payload) -> hipStreamWriteValue32(done = 7)
//   launch stream: producer kernel: spins ~300 us, writes the payload, then stores sig = 1 (system-scope release)
// Prints whether the consumer saw the payload, the store -> consumer-start latency, and the wall time of an un-satisfied wait.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/waitvalue_probe.hip -o /tmp/waitvalue_probe && /tmp/waitvalue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(uint32_t* sig, float* payload, unsigned long long* stamp, unsigned spin_ticks, uint32_t value) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  payload[0] = 42.0f + (float)value;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  stamp[0] = __builtin_amdgcn_s_memrealtime();
  __hip_atomic_store(sig, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void consumer(const float* payload, float* seen, unsigned long long* stamp) {
  stamp[1] = __builtin_amdgcn_s_memrealtime();
  seen[0] = payload[0];
}

int main() {
  int can = -1;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t launch, side;
  CK(hipStreamCreateWithFlags(&launch, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  uint32_t* sig = nullptr;
  CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
  *(volatile uint64_t*)sig = 0;
  float *payload, *seen; unsigned long long* stamp; uint32_t* done;
  CK(hipMalloc(&payload, 4)); CK(hipMalloc(&seen, 4)); CK(hipMalloc(&stamp, 16)); CK(hipMalloc(&done, 4));
  CK(hipMemset(payload, 0, 4)); CK(hipMemset(seen, 0, 4)); CK(hipMemset(done, 0, 4));
  CK(hipDeviceSynchronize());
  for (uint32_t round = 1; round <= 5; ++round) {
    CK(hipStreamWaitValue32(side, sig, round, hipStreamWaitValueGte, 0xFFFFFFFFu));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, side, payload, seen, stamp);
    CK(hipStreamWriteValue32(side, done, 100 + round, 0));
    hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, launch, sig, payload, stamp, 30000u, round);   // 300 us at 100 MHz
    CK(hipStreamSynchronize(side));
    float s = 0; unsigned long long st[2]; uint32_t d = 0;
    CK(hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(st, stamp, 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&d, done, 4, hipMemcpyDeviceToHost));
    printf("round %u: consumer saw %.1f (want %.1f), started %.2f us after the producer's store, done word %u\n", round, s,
           42.0f + round, (double)((long long)(st[1] - st[0])) / 100.0, d);
  }
  // a wait that is already satisfied, enqueue cost on the host
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, side));
  for (int i = 0; i < 100; ++i) { CK(hipStreamWaitValue32(side, sig, 1, hipStreamWaitValueGte, 0xFFFFFFFFu)); CK(hipStreamWriteValue32(side, done, i, 0)); }
  CK(hipEventRecord(b, side)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  printf("100 satisfied wait + write pairs on the stream: %.1f us each\n", ms * 10.0f);
  return 0;
}
