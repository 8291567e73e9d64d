// Error reporting + model description / flat parameter layout (host only).
#include "upamd_internal.h"

#include <cstdarg>
#include <cstring>

namespace upamd {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int validate_desc(const upamd_model_desc *d) {
    if (!d) return fail(UPAMD_E_INVALID, "model desc is null");
    if (d->node_dim <= 0 || d->node_dim > UPAMD_NODE_PAD) return fail(UPAMD_E_INVALID, "node_dim must be in [1,24]");
    if (d->numerical_dim <= 0) return fail(UPAMD_E_INVALID, "numerical_dim must be > 0");
    if (d->D <= 0 || d->D % 16 != 0) return fail(UPAMD_E_INVALID, "gcn_node_dim must be a positive multiple of 16 (got %d)", d->D);
    if (d->encoder != UPAMD_ENCODER_SGNN && d->encoder != UPAMD_ENCODER_MLP) return fail(UPAMD_E_INVALID, "unknown encoder kind %d", d->encoder);
    const bool mlp = d->encoder == UPAMD_ENCODER_MLP;
    if (!mlp && (d->L <= 0 || d->L > 16)) return fail(UPAMD_E_INVALID, "num_gcn_layers must be in [1,16]");
    if (mlp && d->L != 0) return fail(UPAMD_E_INVALID, "the rl-mlp encoder has no GCN layers (L must be 0)");
    if (d->edge_fc_layers < 0 || d->edge_fc_layers > UPAMD_MAX_EDGE_FC)
        return fail(UPAMD_E_INVALID, "num_edge_fc_layers must be in [1,%d] (got %d)", UPAMD_MAX_EDGE_FC, d->edge_fc_layers);
    if (d->heads <= 0 || d->D % d->heads != 0) return fail(UPAMD_E_INVALID, "gcn_node_dim must be divisible by num_attention_heads");
    auto chk = [&](int n, const int32_t *h, const char *what, bool last_one, bool mult16) -> int {
        if (n <= 0 || n > UPAMD_MAX_MLP) return fail(UPAMD_E_INVALID, "%s: between 1 and %d layers supported", what, UPAMD_MAX_MLP);
        for (int i = 0; i < n; ++i) {
            if (h[i] <= 0) return fail(UPAMD_E_INVALID, "%s: sizes must be positive", what);
            if (mult16 && i < n - 1 && h[i] % 16 != 0)
                return fail(UPAMD_E_INVALID, "%s: hidden sizes must be multiples of 16 (got %d)", what, h[i]);
        }
        if (last_one && h[n - 1] != 1) return fail(UPAMD_E_INVALID, "%s: last size must be 1", what);
        if (last_one && n < 2) return fail(UPAMD_E_INVALID, "%s: need at least one hidden layer before the size-1 output", what);
        return 0;
    };
    int rc;
    if ((rc = chk(d->n_num, d->num_hidden, "state_encoder_hidden_size", false, false))) return rc;
    if ((rc = chk(d->n_land, d->land_hidden, "policy_land_use_head_hidden_size", true, true))) return rc;
    if ((rc = chk(d->n_road, d->road_hidden, "policy_road_head_hidden_size", true, true))) return rc;
    if ((rc = chk(d->n_value, d->value_hidden, "value_head_hidden_size", true, false))) return rc;
    return 0;
}

int build_param_layout(const upamd_model_desc *d, ParamLayout *out) {
    int rc = validate_desc(d);
    if (rc) return rc;
    ParamLayout P;
    int64_t off = 0;
    auto add = [&](const std::string &name, int rows, int cols, int group) -> int {
        ParamTensor t;
        t.name = name; t.offset = off; t.rows = rows; t.cols = cols; t.group = group;
        P.tensors.push_back(t);
        off = align_up(off + (int64_t)rows * cols, 4);
        return (int)P.tensors.size() - 1;
    };
    const int D = d->D;
    const std::string e = "shared_net.";
    // group 0: shared encoder (state_encoder.py:13-33) + value head (value.py:15-34)
    P.group_begin[0] = off;
    int prev = d->numerical_dim;
    for (int i = 0; i < d->n_num; ++i) {
        P.num_w.push_back(add(e + "numerical_feature_encoder.linear_" + std::to_string(i) + ".weight", d->num_hidden[i], prev, 0));
        P.num_b.push_back(add(e + "numerical_feature_encoder.linear_" + std::to_string(i) + ".bias", d->num_hidden[i], 1, 0));
        prev = d->num_hidden[i];
    }
    P.node_w = add(e + "node_encoder.weight", D, d->node_dim, 0);
    P.node_b = add(e + "node_encoder.bias", D, 1, 0);
    const bool mlp = d->encoder == UPAMD_ENCODER_MLP;     // MLPStateEncoder (state_encoder.py:217-236): numerical + node encoder only
    for (int l = 0; l < d->L; ++l) {
        P.edge_w.push_back(add(e + "edge_fc_layers." + std::to_string(l) + ".linear_0.weight", D, 2 * D, 0));
        P.edge_b.push_back(add(e + "edge_fc_layers." + std::to_string(l) + ".linear_0.bias", D, 1, 0));
        // sub-layers behind the first one (state_encoder.py:59-82), only when num_edge_fc_layers > 1
        P.edge_wk.emplace_back();
        P.edge_bk.emplace_back();
        for (int k = 1; k < edge_fc_layers(*d); ++k) {
            const std::string s = e + "edge_fc_layers." + std::to_string(l) + ".linear_" + std::to_string(k);
            P.edge_wk.back().push_back(add(s + ".weight", D, D, 0));
            P.edge_bk.back().push_back(add(s + ".bias", D, 1, 0));
        }
    }
    if (!mlp) {
    P.inproj_w = add(e + "attention_layer.in_proj_weight", 3 * D, D, 0);
    P.inproj_b = add(e + "attention_layer.in_proj_bias", 3 * D, 1, 0);
    P.outproj_w = add(e + "attention_layer.out_proj.weight", D, D, 0);
    P.outproj_b = add(e + "attention_layer.out_proj.bias", D, 1, 0);
    P.q_w = add(e + "attention_query_layer.weight", D, D, 0);
    P.q_b = add(e + "attention_query_layer.bias", D, 1, 0);
    P.k_w = add(e + "attention_key_layer.weight", D, D, 0);
    P.k_b = add(e + "attention_key_layer.bias", D, 1, 0);
    P.v_w = add(e + "attention_value_layer.weight", D, D, 0);
    P.v_b = add(e + "attention_value_layer.bias", D, 1, 0);
    }
    // output_value_size: state_encoder.py:33 (sgnn: + the attended current node) / :236 (mlp)
    prev = (mlp ? 2 : 3) * D + d->num_hidden[d->n_num - 1] + 3;
    for (int i = 0; i < d->n_value; ++i) {
        P.value_w.push_back(add("value_head.linear_" + std::to_string(i) + ".weight", d->value_hidden[i], prev, 0));
        P.value_b.push_back(add("value_head.linear_" + std::to_string(i) + ".bias", d->value_hidden[i], 1, 0));
        prev = d->value_hidden[i];
    }
    P.group_end[0] = off;
    // group 1: land-use head (policy.py:19-43; first Linear has a bias, the rest do not)
    P.group_begin[1] = off;
    prev = 4 * D;
    for (int i = 0; i < d->n_land; ++i) {
        P.land_w.push_back(add("policy_land_use_head.land_use_linear_" + std::to_string(i) + ".weight", d->land_hidden[i], prev, 1));
        if (i == 0) P.land_b0 = add("policy_land_use_head.land_use_linear_0.bias", d->land_hidden[0], 1, 1);
        prev = d->land_hidden[i];
    }
    P.group_end[1] = off;
    // group 2: road head
    P.group_begin[2] = off;
    prev = D;
    for (int i = 0; i < d->n_road; ++i) {
        P.road_w.push_back(add("policy_road_head.road_linear_" + std::to_string(i) + ".weight", d->road_hidden[i], prev, 2));
        if (i == 0) P.road_b0 = add("policy_road_head.road_linear_0.bias", d->road_hidden[0], 1, 2);
        prev = d->road_hidden[i];
    }
    P.group_end[2] = off;
    P.n_floats = off;
    *out = P;
    return 0;
}

}  // namespace upamd

extern "C" int upamd_abi_version(void) { return UPAMD_ABI_VERSION; }
extern "C" const char *upamd_last_error(void) { return upamd::g_err; }

extern "C" int upamd_param_count(const upamd_model_desc *desc, int64_t *n_floats, int32_t *n_tensors) {
    upamd::ParamLayout P;
    int rc = upamd::build_param_layout(desc, &P);
    if (rc) return rc;
    if (n_floats) *n_floats = P.n_floats;
    if (n_tensors) *n_tensors = (int32_t)P.tensors.size();
    return UPAMD_OK;
}

extern "C" int upamd_param_info(const upamd_model_desc *desc, int32_t index, char *name_out, int32_t name_cap,
                                int64_t *offset, int32_t *rows, int32_t *cols, int32_t *group) {
    upamd::ParamLayout P;
    int rc = upamd::build_param_layout(desc, &P);
    if (rc) return rc;
    if (index < 0 || index >= (int32_t)P.tensors.size()) return upamd::fail(UPAMD_E_INVALID, "upamd_param_info: index out of range");
    const upamd::ParamTensor &t = P.tensors[index];
    if (name_out && name_cap > 0) {
        std::strncpy(name_out, t.name.c_str(), name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    if (offset) *offset = t.offset;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    if (group) *group = t.group;
    return UPAMD_OK;
}

extern "C" int upamd_param_groups(const upamd_model_desc *desc, int64_t begin_out[3], int64_t end_out[3]) {
    upamd::ParamLayout P;
    int rc = upamd::build_param_layout(desc, &P);
    if (rc) return rc;
    for (int g = 0; g < 3; ++g) {
        begin_out[g] = P.group_begin[g];
        end_out[g] = P.group_end[g];
    }
    return UPAMD_OK;
}
