// Tile geometry shared by the fused WaveNet-layer forward (wavenet.hip) and backward (wavenet_bwd.hip).
#pragma once
#include "gcn_dev.h"

namespace hopmi {

struct LayerGeom {
  GcnGeom g;       // V, S = output slabs per tile, mtiles, rows_lds, mix-matrix geometry, ntiles
  int B, T_in, T_out, d;
  int n_slabs;     // B * T_out
  float invV, invT;  // 1/V, 1/T_out for the (row + 0.5) * inv index splits (exact for the ranges validated)
};

constexpr int WN_MAX_MT = 5;                       // <= 80 rows per tile

static inline LayerGeom make_layer_geom(int B, int T_in, int V, int d, int grid_target, int max_mt = WN_MAX_MT) {
  LayerGeom L;
  L.B = B; L.T_in = T_in; L.d = d; L.T_out = T_in - d;
  L.n_slabs = B * L.T_out;
  int S = (L.n_slabs + grid_target - 1) / grid_target;            // one tile per workgroup when it fits ...
  const int smax = (16 * max_mt) / V > 0 ? (16 * max_mt) / V : 1;
  if (S > smax) S = smax;                                         // ... else walk several
  if (S < 1) S = 1;
  L.g = make_geom(L.n_slabs, V, S);
  L.invV = 1.0f / V;
  L.invT = 1.0f / L.T_out;
  return L;
}

static inline int wn_env_int(const char* name, int dflt) { return env_int(name, dflt); }

static inline int wn_validate(int B, int T_in, int V, int d) {
  if (B <= 0 || V < 1 || V > HOPMI_MAX_NODES || d < 1 || T_in - d < 4) {
    set_error("hopmi_wn_layer: bad geometry B=%d T_in=%d V=%d dilation=%d (need T_in - dilation >= 4, V in [1,%d])", B, T_in,
              V, d, HOPMI_MAX_NODES);
    return HOPMI_EINVAL;
  }
  if ((long long)B * T_in * V >= (1LL << 20) * 16) {
    set_error("hopmi_wn_layer: B*T*V too large for the 32-bit row index math");
    return HOPMI_EINVAL;
  }
  if ((long long)B * (T_in - d) >= (1 << 20)) {
    set_error("hopmi_wn_layer: B*T_out >= 2^20 not supported (float index split)");
    return HOPMI_EINVAL;
  }
  return HOPMI_OK;
}

}  // namespace hopmi
