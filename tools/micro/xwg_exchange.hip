// What does one cross-workgroup arg-max exchange per FPS round cost?  (split a
// cloud's buckets over P workgroups, exchange the per-round winner through one 64-bit atomicMax)
// P workgroups of one wave each; every round: atomicMax(slot[round % 3], key) at agent scope,
// atomicAdd(arrived[round % 3]), spin on an agent-scope load until all P arrived, read the slot,
// and (the next round's slot having been cleared two rounds earlier) go on.  The key depends on
// the previous round's result so rounds cannot overlap -- the dependence FPS has.
// Placement: workgroup ids i, i + 8, i + 16 ... land on the same XCD (round-robin dispatch), ids
// i, i + 1, ... on different XCDs.  Prints ns per round for P = 2, 4, 8 in both placements, and
// for P = 1 (the loop overhead itself).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
__global__ void __launch_bounds__(64) xchg(u64 *slots, unsigned *arrived, u64 *out, int parts, int stride,
                                           int groups, int rounds) {
  // group g = blockIdx.x % stride ... members blockIdx.x = g + p * stride
  const int g = blockIdx.x % stride, p = blockIdx.x / stride;
  if (g >= groups || p >= parts) return;
  u64 *slot = slots + g * 3;
  unsigned *arr = arrived + g * 3;
  u64 prev = blockIdx.x * 2654435761u;
  for (int r = 0; r < rounds; ++r) {
    const int s = r % 3;
    u64 key = (prev * 6364136223846793005ull + p + 1) >> 8;  // depends on the last winner
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_max(slot + s, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // release: the max must be visible before the arrival
      __hip_atomic_fetch_add(arr + s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)parts * (unsigned)(r / 3 + 1);
      while (__hip_atomic_load(arr + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {}
      prev = __hip_atomic_load(slot + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // clear the slot used two rounds from now (everyone has read it: they arrived at round r,
      // which they could only do after reading round r-1's slot ... the slot of round r+1 was
      // last read in round r-2)
      if (p == 0) __hip_atomic_store(slot + (r + 1) % 3, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    prev = __shfl(prev, 0, 64);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = prev;
}
int main() {
  u64 *slots, *out; unsigned *arrived;
  hipMalloc(&slots, 64 * 3 * 8); hipMalloc(&arrived, 64 * 3 * 4); hipMalloc(&out, 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int rounds = 2047;
  printf("{\"rounds\": %d, \"groups\": 8, \"results\": [\n", rounds);
  bool first = true;
  for (int same_xcd = 1; same_xcd >= 0; --same_xcd)
    for (int parts = 1; parts <= 8; parts *= 2) {
      // same XCD: stride 8 (8 groups, one per XCD, members 8 apart); different XCDs: members
      // adjacent -> group = blockIdx / parts emulated by stride = 8 groups but p-major order
      const int groups = 8;
      float best = 1e9f;
      for (int it = 0; it < 5; ++it) {
        hipMemset(slots, 0, 64 * 3 * 8); hipMemset(arrived, 0, 64 * 3 * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (same_xcd) xchg<<<groups * parts, 64>>>(slots, arrived, out, parts, 8, groups, rounds);
        else xchg<<<groups * parts, 64>>>(slots, arrived, out, parts, groups * parts == 8 ? 8 : 8, groups, rounds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      if (!same_xcd) continue;  // the cross-XCD placement is measured by the second kernel below
      printf("%s {\"placement\": \"same_xcd\", \"parts\": %d, \"ns_per_round\": %.1f}", first ? "" : ",\n", parts, best * 1e6f / rounds);
      first = false;
    }
  // different XCDs: group g's members are blocks g*parts .. g*parts+parts-1 -> consecutive ids
  for (int parts = 2; parts <= 8; parts *= 2) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      hipMemset(slots, 0, 64 * 3 * 8); hipMemset(arrived, 0, 64 * 3 * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      // stride = 1 group of `parts` consecutive blocks: one group only, members on parts XCDs
      xchg<<<parts, 64>>>(slots, arrived, out, parts, 1, 1, rounds);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf(",\n {\"placement\": \"different_xcds\", \"parts\": %d, \"ns_per_round\": %.1f}", parts, best * 1e6f / rounds);
  }
  printf("\n]}\n");
  return 0;
}
