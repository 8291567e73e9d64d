// Internal declarations shared by the translation units of libupamd.so.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/upamd.h"

namespace upamd {

int fail(int code, const char *fmt, ...);   // records the thread-local error message, returns code

#define UPAMD_HIP(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::upamd::fail(UPAMD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                            \
    } while (0)

// ---- flat parameter layout -------------------------------------------------------------
struct ParamTensor {
    std::string name;
    int64_t offset;   // floats
    int32_t rows, cols;   // bias: rows = len, cols = 1
    int32_t group;
};

struct ParamLayout {
    std::vector<ParamTensor> tensors;
    int64_t n_floats = 0;
    int64_t group_begin[3] = {0, 0, 0}, group_end[3] = {0, 0, 0};
    // indices into `tensors`
    std::vector<int> num_w, num_b;            // numerical encoder
    int node_w = -1, node_b = -1;
    std::vector<int> edge_w, edge_b;          // per GCN layer: first sub-layer (linear_0, [D, 2D])
    std::vector<std::vector<int>> edge_wk, edge_bk;   // [layer][k - 1]: sub-layers k = 1 .. K-1 (linear_k, [D, D]); empty when K = 1
    int inproj_w = -1, inproj_b = -1, outproj_w = -1, outproj_b = -1;
    int q_w = -1, q_b = -1, k_w = -1, k_b = -1, v_w = -1, v_b = -1;
    std::vector<int> value_w, value_b;
    std::vector<int> land_w; int land_b0 = -1;
    std::vector<int> road_w; int road_b0 = -1;
    int64_t off(int idx) const { return tensors[idx].offset; }
};

int validate_desc(const upamd_model_desc *d);
inline int edge_fc_layers(const upamd_model_desc &d) { return d.edge_fc_layers <= 0 ? 1 : d.edge_fc_layers; }
int build_param_layout(const upamd_model_desc *d, ParamLayout *out);

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

}  // namespace upamd
