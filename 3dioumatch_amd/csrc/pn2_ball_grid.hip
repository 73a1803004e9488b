// 3dioumatch_amd/csrc/pn2_ball_grid.hip -- cell-list tier of ball_query (large clouds).
// Placeholder: reports "not handled" so pn2_ball_query uses the brute-force tier.
#include "common.h"

int pn2_ball_query_grid_try(int, int, int, float, int, const float *, const float *, int *, void *,
                            size_t, hipStream_t, int *handled) {
  *handled = 0;
  return 0;
}

size_t pn2_ball_query_grid_workspace(int, int, int, int) { return 0; }
