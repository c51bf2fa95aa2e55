// Internal interface of the fused tile-local kernels (fused_tile.hip <-> adapter.hip).
#pragma once

#include "common.h"

namespace gconv {

struct TileFwdArgs {
  const gconv_tile_t* tiles; const int32_t* lcol; const int32_t* halo;   // gconv_tiling_t of S
  const int32_t* rowptr; const float* dis;
  const float* x; float* out; float* saved_h; float* saved_h1; float* saved_p;
  const float* w_down; const float* b_down; const float* w_up; const float* b_up;
  const float* scalar;
  int d, r, fp, act, skip, cap, n_tiles, n;   // n = nodes of the graph (rows of x)
  int stagger;   // in: the caller's walk bit (0 / 1); the launcher turns it into the kernels' mode (0, 1, 2)
  int store_nt;  // output rows with non-temporal stores (set by the launcher: tile_store_nt)
  int npass;     // ceil(n_tiles / grid) (launcher)
  long long* stamps;   // diagnostic build only (-DGCONV_TILE_STAMPS, tools/tilestamps.py)
};

struct TileBwdArgs {
  const gconv_tile_t* tiles; const int32_t* lcol; const int32_t* halo;   // gconv_tiling_t of S^T
  const int32_t* rowptr; const float* dis;
  const float* x; const float* g; const float* saved_h; const float* saved_h1;
  const float* saved_p;
  const float* w_down; const float* w_up; const float* b_up; const float* scalar;
  float* dx;        // nullable
  // one partial row per (persistent) workgroup: [grid, Fp, d] x2, [grid, d], [grid, Fp], [grid]
  float* part_wd; float* part_wu; float* part_cs; float* part_bd;
  float* part_ds;   // complete scalar-gradient partial: (G Wu).P + [skip] G.X + b_up.colsum(G)
  int d, r, fp, act, skip, cap, want_ds, n_tiles, n;
  int stagger, store_nt, npass;   // as in TileFwdArgs
  long long* stamps;   // diagnostic build only
};

// true if the graph is tiled and one tile's rank-width state fits the LDS budget
bool tile_path_ok(const gconv_graph_t* g, int d, int fp);
size_t tile_lds_bytes(int d, int fp, int tile_nodes_max, int tile_edges_max);

// workgroups the tile kernels are launched with (= partial rows of the backward)
int tile_grid(int n_tiles);

// done_event: nullable hipEvent_t recorded by the launch's own completion (gconv_arm_completion_event)
gconv_status_t launch_tile_fwd(TileFwdArgs a, int n_tiles, int tile_nodes_max, int tile_edges_max,
                               hipStream_t stream, void* done_event = nullptr);
gconv_status_t launch_tile_bwd(TileBwdArgs a, int n_tiles, int tile_nodes_max, int tile_edges_max,
                               hipStream_t stream);

}  // namespace gconv
