// tools/micro/grid_probe.hip -- the cell-list query kernel built with its per-centroid probe
// (see grid_probe.py); includes the product source so the probed code is the shipped code.
#include "pn2_ball_grid.hip"

extern "C" __attribute__((visibility("default"))) int grid_probe_run(
    int b, int n, int m, int c, float radius, int nsample, const void *new_xyz, const void *xyz,
    const void *feat, void *idx, void *out, void *ws, size_t ws_bytes, void *stream, int prebuilt) {
  int handled = 0;
  const int rc = pn2_query_group_grid_try(b, n, m, c, 3 + c, radius, nsample, 1, (const float *)new_xyz,
                                          (const float *)xyz, (const float *)feat, (int *)idx, (float *)out,
                                          ws, ws_bytes, (hipStream_t)stream, prebuilt, &handled);
  return rc ? rc : (handled ? 0 : -1);
}
extern "C" __attribute__((visibility("default"))) size_t grid_probe_ws(int b, int n, int m, int ns) {
  return pn2_ball_query_grid_workspace(b, n, m, ns);
}
extern "C" __attribute__((visibility("default"))) int grid_probe_read(void *t) {
  return (int)hipMemcpyFromSymbol(t, HIP_SYMBOL(grid_probe_t), sizeof(unsigned long long) * 16384 * 8);
}
