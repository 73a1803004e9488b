// Flag-in-data variant of xwg_exchange.hip: no read-modify-write atomics, no counters.  Part p of
// a group publishes two 16-byte records {value, index, tag, 0} and {x, y, z, tag} into ITS slot of
// the round's parity, then one load instruction over 2 * P lanes polls every part's two records
// until all tags equal the round.  A part can run at most one round ahead of the slowest, so two
// parities are enough and nothing is ever cleared.  SCOPE picks the cache-policy bits of the
// 16-byte loads / stores: 0 = sc0 (coherent through the XCD's L2 only -- valid only when the P
// workgroups share an XCD), 1 = sc1 (agent), 2 = sc0 sc1 (system).
// Prints ns per round for 8 groups (one per XCD, members 8 workgroup ids apart) and for one
// group whose members sit on different XCDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
template <int SCOPE> __device__ __forceinline__ void st16(u32x4 *p, u32x4 v) {
  if (SCOPE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (SCOPE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (SCOPE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int SCOPE> __device__ __forceinline__ u32x4 ld16(const u32x4 *p) {
  u32x4 v;
  if (SCOPE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (SCOPE == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (SCOPE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// slots: [group][parity][record 0/1][part] of 16 bytes
template <int SCOPE>
__global__ void __launch_bounds__(64) xchg(u32x4 *slots, u32 *out, int *failed, int parts, int stride, int groups, int rounds) {
  const int g = blockIdx.x % stride, p = blockIdx.x / stride, lane = threadIdx.x;
  if (g >= groups || p >= parts) return;
  u32x4 *base = slots + (size_t)g * 2 * 2 * 64;
  u32 prev = blockIdx.x * 2654435761u;
  for (int r = 1; r <= rounds; ++r) {
    u32x4 *par = base + (r & 1) * 2 * 64;
    const u32 val = (prev * 1664525u + 1013904223u + p) >> 1;  // depends on the last winner
    if (lane < 2) {
      u32x4 rec;
      rec.x = lane == 0 ? val : 1u; rec.y = lane == 0 ? (u32)p : 2u; rec.z = lane == 0 ? (u32)r : 3u; rec.w = lane == 0 ? 0u : (u32)r;
      st16<SCOPE>(par + lane * 64 + p, rec);
    }
    u32x4 got = {0, 0, 0, 0};
    const bool poller = lane < 2 * parts;
    const u32x4 *src = par + (lane >= parts ? 64 + lane - parts : lane);
    int spins = 0;
    for (;;) {
      bool ok = true;
      if (poller) { got = ld16<SCOPE>(src); ok = (lane < parts ? got.z : got.w) == (u32)r; }
      if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
      if (++spins > (1 << 20)) { if (lane == 0) *failed = 1; return; }
    }
    // winner = max value over the first `parts` lanes
    u32 best = lane < parts ? got.x : 0u;
    for (int o = 32; o >= 1; o >>= 1) { const u32 t = __shfl_xor(best, o, 64); best = t > best ? t : best; }
    prev = best;
  }
  if (lane == 0) out[blockIdx.x] = prev;
}
template <int SCOPE> static void run(const char *name, u32x4 *slots, u32 *out, int *failed, bool &first) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int rounds = 2047;
  for (int same = 1; same >= 0; --same)
    for (int parts = same ? 1 : 2; parts <= (same ? 32 : 8); parts *= 2) {
      if (SCOPE == 0 && !same) continue;
      const int groups = same ? 8 : 1, stride = same ? 8 : 1;
      float best = 1e9f; int bad = 0;
      for (int it = 0; it < 4; ++it) {
        (void)hipMemset(slots, 0, 8 * 2 * 2 * 64 * 16); (void)hipMemset(failed, 0, 4);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        xchg<SCOPE><<<groups * parts, 64>>>(slots, out, failed, parts, stride, groups, rounds);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int f; (void)hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost); bad |= f;
        if (ms < best) best = ms;
      }
      printf("%s {\"scope\": \"%s\", \"placement\": \"%s\", \"groups\": %d, \"parts\": %d, \"ns_per_round\": %.1f, \"gave_up\": %d}",
             first ? "" : ",\n", name, same ? "one_xcd_per_group" : "one_group_across_xcds", groups, parts, best * 1e6f / rounds, bad);
      first = false;
    }
}
int main() {
  u32x4 *slots; u32 *out; int *failed;
  (void)hipMalloc(&slots, 8 * 2 * 2 * 64 * 16); (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&failed, 4);
  printf("{\"rounds\": 2047, \"results\": [\n");
  bool first = true;
  run<1>("sc1", slots, out, failed, first);
  run<2>("sc0 sc1", slots, out, failed, first);
  run<0>("sc0", slots, out, failed, first);
  printf("\n]}\n");
  return 0;
}
