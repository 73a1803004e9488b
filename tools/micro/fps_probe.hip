// tools/micro/fps_probe.hip -- the bucketed FPS kernel built with its round probes
// (see fps_probe.py); includes the product source so the probed code is the shipped code.
#include "pn2_fps_bucket.hip"

extern "C" __attribute__((visibility("default"))) size_t fps_probe_scratch(int b, int n) {
  return pn2_fps_bucket_scratch_bytes(b, n);
}
extern "C" __attribute__((visibility("default"))) int fps_probe_run(int b, int n, int m, int log2bs, const void *xyz,
                                                                      void *scratch, void *idx, size_t bytes,
                                                                      void *stream) {
  int handled = 0;
  const int rc = pn2_fps_bucket_try(b, n, m, log2bs, (const float *)xyz, scratch, bytes, (int *)idx,
                                    (hipStream_t)stream, &handled, 0.f, nullptr, nullptr, nullptr);
  return rc ? rc : (handled ? 0 : -1);
}
extern "C" __attribute__((visibility("default"))) int fps_probe_read(void *t, void *v) {
  int rc = (int)hipMemcpyFromSymbol(t, HIP_SYMBOL(fps_probe_t), sizeof(unsigned long long) * 2048 * 16 * 16);
  if (rc) return rc;
  return (int)hipMemcpyFromSymbol(v, HIP_SYMBOL(fps_probe_v), 2048 * 16);
}
