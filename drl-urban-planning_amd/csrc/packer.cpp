// Host replay packer: padded 9-field states -> ragged/CSR batch (see include/upamd.h).
//
// Replaces the reference's per-minibatch tensorfy/batch_data
// (urban_planning/agents/urban_planning_agent.py:16-20, urban_planning/models/state_encoder.py:163-177).
// Graph semantics follow the reference exactly:
//   * live edges  = slots with edge_mask; each live edge (i,j) contributes its message to node i
//     AND node j, so it appears once in each endpoint's incidence list (a self-loop twice in one
//     list: state_encoder.py:145-147 counts both roles);
//   * nodes [0,n) are kept un-compacted, n = 1 + max(last node_mask slot, any live endpoint, last
//     road candidate) so padded-slot indices stay valid node ids; node_mask is carried per node
//     (mean + attention use it, state_encoder.py:155-159,179-182,200);
//   * pointer-head candidates = land_use_mask slots (stage 0) / road_mask slots (stage 1) in
//     padded-slot order (urban_planning/models/policy.py:48-61); everything else has probability
//     exactly 0 in the reference's masked softmax and is never materialised.
#include "upamd_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <vector>

namespace {

struct StateView {
    const float *numerical;
    const float *feat;
    const int64_t *edge_index;
    const float *cur;
    const uint8_t *node_mask, *edge_mask, *land_mask, *road_mask;
    const float *stage;
};

inline StateView view(const uint64_t *ptrs, int64_t T, int64_t t) {
    StateView s;
    s.numerical = reinterpret_cast<const float *>(ptrs[0 * T + t]);
    s.feat = reinterpret_cast<const float *>(ptrs[1 * T + t]);
    s.edge_index = reinterpret_cast<const int64_t *>(ptrs[2 * T + t]);
    s.cur = reinterpret_cast<const float *>(ptrs[3 * T + t]);
    s.node_mask = reinterpret_cast<const uint8_t *>(ptrs[4 * T + t]);
    s.edge_mask = reinterpret_cast<const uint8_t *>(ptrs[5 * T + t]);
    s.land_mask = reinterpret_cast<const uint8_t *>(ptrs[6 * T + t]);
    s.road_mask = reinterpret_cast<const uint8_t *>(ptrs[7 * T + t]);
    s.stage = reinterpret_cast<const float *>(ptrs[8 * T + t]);
    return s;
}

inline int argmax3(const float *s) {
    int best = 0;
    if (s[1] > s[best]) best = 1;
    if (s[2] > s[best]) best = 2;
    return best;
}

// Default packer thread count: this process's share of the host -- the CPUs it may run on (affinity mask), divided by
// the ranks torchrun started on this node (LOCAL_WORLD_SIZE, else WORLD_SIZE), capped at 64 (the fill is memory-bound:
// more threads than that only contend).  Eight ranks on a 256-CPU host get 32 threads each instead of 8 x 256.
// CPUs the container may actually burn: the cgroup's CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  A box that
// shows 128 CPUs but grants 16 cores' worth of time per period throttles a process that runs 64 busy threads: the whole process is
// frozen for the rest of the period once the quota is spent (profiles/archive/r04_lab_host_stalls.log saw it with OpenMP; the packer's own
// threads did the same to the packing phase).
int cgroup_cpu_quota() {
    auto read2 = [](const char *path, long long *a, long long *b) -> int {
        FILE *f = fopen(path, "r");
        if (!f) return 0;
        char w1[64] = "", w2[64] = "";
        const int n = fscanf(f, "%63s %63s", w1, w2);
        fclose(f);
        if (n >= 1) *a = strcmp(w1, "max") == 0 ? -1 : atoll(w1);
        if (n >= 2 && b) *b = atoll(w2);
        return n;
    };
    long long quota = -1, period = 100000;
    if (read2("/sys/fs/cgroup/cpu.max", &quota, &period) >= 1) {
        if (quota > 0 && period > 0) return (int)std::max<long long>(1, (quota + period - 1) / period);
        return 0;
    }
    long long q1 = -1, p1 = 100000;
    if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q1, nullptr) >= 1 && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p1, nullptr) >= 1 &&
        q1 > 0 && p1 > 0)
        return (int)std::max<long long>(1, (q1 + p1 - 1) / p1);
    return 0;       // no quota
}

int default_pack_threads() {
    const char *ev = getenv("UPAMD_PACK_THREADS");
    if (ev && atoi(ev) > 0) return std::min(256, atoi(ev));
    int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) cpus = std::min(cpus, (int)CPU_COUNT(&set));
    // A CPU quota limits CPU-TIME per period, not threads: the packer runs in bursts of 10-40 ms, and measured on a 256-CPU box
    // with a 16-core quota (profiles/r05_lab_prepare.md) the fill of 8192 HLG states takes 73 / 39 / 21 / 13 ms on 8 / 16 / 32 / 64
    // threads -- the burst is over before the period's budget is spent.  Only a burst far beyond the budget gets throttled:
    // cap at four times the quota.
    const int quota = cgroup_cpu_quota();
    if (quota > 0) cpus = std::min(cpus, 4 * quota);
    int ranks = 1;
    const char *lw = getenv("LOCAL_WORLD_SIZE");
    if (!lw || !*lw) lw = getenv("WORLD_SIZE");
    if (lw && atoi(lw) > 1) ranks = atoi(lw);
    return std::max(1, std::min(64, cpus / ranks));
}

// Persistent worker pool: the streamed pack calls the fill once per chunk of the replay, and spawning 16-64 threads per call cost
// more than packing a small chunk.  Workers sleep on a condition variable between jobs.  fork(): the child owns none of the
// threads -- the atfork handler drops the pool object there (leaked on purpose: its threads do not exist in the child) and the
// next call builds a new one.
class Pool {
public:
    explicit Pool(int n) : stop_(false), gen_(0), pending_(0) {
        for (int w = 0; w < n; ++w) threads_.emplace_back([this]() { loop(); });
    }
    int size() const { return (int)threads_.size(); }
    // runs job() on `use` workers (<= size()) and returns when all of them are done
    void run(int use, const std::function<void()> &job) {
        std::unique_lock<std::mutex> lk(mu_);
        job_ = &job;
        want_ = use;
        pending_ = use;
        ++gen_;
        cv_.notify_all();
        done_.wait(lk, [this]() { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void()> *job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&]() { return stop_ || (gen_ != seen && want_ > 0); });
                if (stop_) return;
                seen = gen_;
                --want_;
                job = job_;
            }
            (*job)();
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    bool stop_;
    uint64_t gen_;
    int want_ = 0, pending_;
    const std::function<void()> *job_ = nullptr;
};

std::mutex g_pool_mu;             // one parallel_for at a time per process (the packer is called from one host thread)
Pool *g_pool = nullptr;
bool g_atfork = false;
void pool_after_fork_child() {
    g_pool = nullptr;             // (its threads do not exist here)
    new (&g_pool_mu) std::mutex();
}

constexpr int64_t GRAIN = 16;     // states a worker takes at a time

template <typename F>
int parallel_for(int64_t T, int n_threads, F &&fn) {
    if (n_threads <= 0) n_threads = default_pack_threads();
    // small batches (action serving packs 8-64 states per round, ~70 us each): a finer grain so that up to 16 workers share them
    // -- with the 16-state grain a 64-state round ran on 4 threads, 1.1 ms of a 1.7 ms serving round
    int64_t grain = GRAIN;
    if (T < GRAIN * n_threads) {
        const int64_t w = std::min<int64_t>(n_threads, 16);
        grain = std::min<int64_t>(GRAIN, std::max<int64_t>(1, (T + w - 1) / w));
    }
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, T / grain));
    std::atomic<int> status{0};
    if (n_threads <= 1) {
        for (int64_t t = 0; t < T; ++t) {
            int rc = fn(t);
            if (rc != 0) return rc;
        }
        return 0;
    }
    std::atomic<int64_t> next{0};
    const std::function<void()> job = [&]() {
        for (;;) {
            int64_t t0 = next.fetch_add(grain);
            if (t0 >= T || status.load() != 0) return;
            int64_t t1 = std::min(T, t0 + grain);
            for (int64_t t = t0; t < t1; ++t) {
                int rc = fn(t);
                if (rc != 0) {
                    status.store(rc);
                    return;
                }
            }
        }
    };
    std::lock_guard<std::mutex> guard(g_pool_mu);
    if (!g_atfork) {
        pthread_atfork(nullptr, nullptr, pool_after_fork_child);
        g_atfork = true;
    }
    if (!g_pool || g_pool->size() < n_threads) g_pool = new Pool(std::max(n_threads, g_pool ? g_pool->size() : 0));      // (an outgrown pool is leaked: rare, small)
    g_pool->run(n_threads, job);
    return status.load();
}

inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

}  // namespace

// Compact wire records (drl-urban-planning_amd/packer.py: _REC_HEADER / _record_sections) -> the packer's address table, without
// a Python object per field: header = 10 little-endian 32-bit words (magic 'UPS1', node_dim, numerical_len, cur_len, stage_len,
// pad_n, pad_e, n_rows, e_rows, edge_fill), then the nine arrays, each 8-byte aligned, trimmed to n_rows / e_rows.
extern "C" int upamd_record_table(int64_t T, const uint64_t *rec_ptrs, const int64_t *rec_sizes, int32_t node_dim,
                                  int32_t numerical_dim, uint64_t *ptrs, int32_t *pad_n, int32_t *pad_e) {
    if (T <= 0 || !rec_ptrs || !rec_sizes || !ptrs || !pad_n || !pad_e) return upamd::fail(UPAMD_E_INVALID, "upamd_record_table: bad argument");
    auto a8 = [](int64_t x) { return (x + 7) & ~int64_t(7); };
    for (int64_t t = 0; t < T; ++t) {
        const uint32_t *h = reinterpret_cast<const uint32_t *>(rec_ptrs[t]);
        if (!h || rec_sizes[t] < 40 || h[0] != 0x31535055u) return upamd::fail(UPAMD_E_INVALID, "upamd_record_table: state %lld is not a compact record", (long long)t);
        const int64_t F = h[1], Fn = h[2], cur = h[3], st = h[4], nr = h[7], er = h[8];
        if (F != node_dim || Fn != numerical_dim || cur != node_dim || st != 3)
            return upamd::fail(UPAMD_E_INVALID, "upamd_record_table: state %lld has node_dim %lld / numerical %lld / current-node %lld / stage %lld entries, "
                                                "expected %d / %d / %d / 3", (long long)t, (long long)F, (long long)Fn, (long long)cur, (long long)st, node_dim, numerical_dim, node_dim);
        const int64_t bytes[9] = {Fn * 4, nr * F * 4, er * 16, cur * 4, nr, er, er, nr, st * 4};
        int64_t off = a8(40);
        for (int f = 0; f < 9; ++f) {
            ptrs[f * T + t] = rec_ptrs[t] + (uint64_t)off;
            off = a8(off + bytes[f]);
        }
        if (off > rec_sizes[t]) return upamd::fail(UPAMD_E_INVALID, "upamd_record_table: state %lld: truncated record (%lld < %lld bytes)", (long long)t, (long long)rec_sizes[t], (long long)off);
        if (nr > INT32_MAX || er > INT32_MAX) return upamd::fail(UPAMD_E_LIMIT, "upamd_record_table: record too large");
        pad_n[t] = (int32_t)nr;
        pad_e[t] = (int32_t)er;
    }
    return UPAMD_OK;
}

extern "C" int upamd_pack_plan(int64_t T, const uint64_t *ptrs, const int32_t *pad_n, const int32_t *pad_e,
                               const float *actions, int32_t node_dim, int32_t numerical_dim, int32_t n_threads,
                               int32_t *meta, upamd_pack_layout *layout) {
    return upamd_pack_plan_ex(T, ptrs, pad_n, pad_e, actions, node_dim, numerical_dim, n_threads, 1, meta, layout);
}

extern "C" int upamd_pack_plan_ex(int64_t T, const uint64_t *ptrs, const int32_t *pad_n, const int32_t *pad_e,
                                  const float *actions, int32_t node_dim, int32_t numerical_dim, int32_t n_threads,
                                  int32_t exact, int32_t *meta, upamd_pack_layout *layout) {
    if (T <= 0 || !ptrs || !pad_n || !pad_e || !actions || !meta || !layout)
        return upamd::fail(UPAMD_E_INVALID, "upamd_pack_plan: null argument or T <= 0");
    if (node_dim <= 0 || node_dim > UPAMD_NODE_PAD)
        return upamd::fail(UPAMD_E_INVALID, "upamd_pack_plan: node_dim must be in [1, 24]");
    const bool no_mlp_fields = (exact & 2) != 0;      // flag bit 1: leave out what only the rl-mlp encoder reads (he_sel, xbar)
    exact &= 1;
    int rc = parallel_for(T, n_threads, [&](int64_t t) -> int {
        StateView s = view(ptrs, T, t);
        const int N = pad_n[t], E = pad_e[t];
        int32_t *m = meta + t * UPAMD_META_STRIDE;
        std::memset(m, 0, sizeof(int32_t) * UPAMD_META_STRIDE);
        const int stage = argmax3(s.stage);
        int n = 1, n_mask = 0;
        for (int v = 0; v < N; ++v)
            if (s.node_mask[v]) {
                n = std::max(n, v + 1);
                ++n_mask;
            }
        int e = 0, nh = 0, nr = 0;
        if (exact) {
            for (int k = 0; k < E; ++k) {
                if (s.edge_mask[k]) {
                    const int64_t i = s.edge_index[2 * k], j = s.edge_index[2 * k + 1];
                    if (i < 0 || j < 0 || i >= N || j >= N) return -1000;
                    n = std::max(n, (int)std::max(i, j) + 1);
                    ++e;
                }
                if (stage == 0 && s.land_mask[k]) ++nh;
            }
        } else {
            // masks only (a 16th of the bytes: the int64 edge list is not touched): the extent of the graph is taken from the node /
            // road masks, which is what it is for every state the extractor emits (an edge's endpoints are live nodes,
            // observation_extractor.py:84-132); the fill checks every live endpoint against it and asks for the exact pass if not
            for (int k = 0; k < E; ++k) {
                e += s.edge_mask[k] ? 1 : 0;
                if (stage == 0 && s.land_mask[k]) ++nh;
            }
        }
        if (stage == 1)
            for (int v = 0; v < N; ++v)
                if (s.road_mask[v]) {
                    n = std::max(n, v + 1);
                    ++nr;
                }
        if (n > 65535 || nh >= 65535) return -1001;
        int act = -1;
        if (stage == 0) {
            const int a = (int)actions[2 * t];
            if (a >= 0 && a < E && s.land_mask[a]) {
                act = 0;
                for (int k = 0; k < a; ++k) act += s.land_mask[k] ? 1 : 0;
            }
        } else if (stage == 1) {
            const int a = (int)actions[2 * t + 1];
            if (a >= 0 && a < N && s.road_mask[a]) {
                act = 0;
                for (int v = 0; v < a; ++v) act += s.road_mask[v] ? 1 : 0;
            }
        }
        m[0] = n; m[1] = e; m[2] = nh; m[3] = nr; m[4] = stage; m[5] = act; m[6] = n_mask; m[7] = N; m[8] = E;
        return 0;
    });
    if (rc == -1000) return upamd::fail(UPAMD_E_INVALID, "upamd_pack_plan: live edge endpoint outside [0, pad_n)");
    if (rc == -1001) return upamd::fail(UPAMD_E_LIMIT, "upamd_pack_plan: a graph exceeds the 16-bit local index range");
    if (rc != 0) return upamd::fail(UPAMD_E_INVALID, "upamd_pack_plan: failed");
    int64_t nodes = 0, edges = 0, he = 0, rn = 0;
    for (int64_t t = 0; t < T; ++t) {
        int32_t *m = meta + t * UPAMD_META_STRIDE;
        if (nodes + t > INT32_MAX - 70000 || 2 * edges > INT32_MAX - 200000 || he > INT32_MAX - 70000)
            return upamd::fail(UPAMD_E_LIMIT, "upamd_pack_plan: replay too large for 32-bit offsets");
        m[9] = (int32_t)nodes; m[10] = (int32_t)edges; m[11] = (int32_t)he; m[12] = (int32_t)rn;
        m[13] = (int32_t)(nodes + t);
        nodes += m[0]; edges += m[1]; he += m[2]; rn += m[3];
    }
    upamd_pack_layout L;
    std::memset(&L, 0, sizeof(L));
    L.T = T; L.total_nodes = nodes; L.total_edges = edges; L.total_he = he; L.total_rn = rn;
    L.node_dim = node_dim; L.numerical_dim = numerical_dim;
    int64_t off = 0;
    auto place = [&](int64_t bytes) { int64_t o = off; off = align256(off + std::max<int64_t>(bytes, 4)); return o; };
    L.off_meta = place(T * UPAMD_META_STRIDE * 4);
    L.off_x = place(nodes * UPAMD_NODE_PAD * 4);
    L.off_nmask = place(nodes);
    L.off_rowptr = place((nodes + T) * 4);
    L.off_inc_nbr = place(2 * edges * 2);
    L.off_he_src = place(he * 2);
    L.off_he_dst = place(he * 2);
    L.off_he_live = place(he);
    L.off_he_slot = place(he * 4);
    L.off_rn_node = place(rn * 2);
    L.off_numerical = place(T * (int64_t)numerical_dim * 4);
    L.off_cur = place(T * UPAMD_NODE_PAD * 4);
    L.off_order = place(nodes * 2);
    L.off_hinc_ptr = place((nodes + T) * 4);
    L.off_hinc_nbr = place(2 * he * 2);
    L.off_hinc_he = place(2 * he * 2);
    if (no_mlp_fields) {
        L.off_he_sel = L.off_xbar = -1;                // not present: upamd_pack_fill* skips them, an rl-mlp engine refuses the replay
    } else {
        L.off_he_sel = place(he * 2);
        L.off_xbar = place(T * UPAMD_NODE_PAD * 4);
    }
    L.total_bytes = off;
    *layout = L;
    return UPAMD_OK;
}

extern "C" int upamd_pack_fill(int64_t T, const uint64_t *ptrs, const int32_t *meta, const upamd_pack_layout *layout,
                               int32_t n_threads, void *out) {
    return upamd_pack_fill_range(T, ptrs, meta, layout, 0, T, n_threads, out);
}

extern "C" int upamd_pack_fill_range(int64_t T, const uint64_t *ptrs, const int32_t *meta, const upamd_pack_layout *layout,
                                     int64_t t_begin, int64_t t_end, int32_t n_threads, void *out) {
    if (T <= 0 || !ptrs || !meta || !layout || !out || layout->T != T || t_begin < 0 || t_end > T || t_begin > t_end)
        return upamd::fail(UPAMD_E_INVALID, "upamd_pack_fill: bad argument");
    const upamd_pack_layout &L = *layout;
    char *base = static_cast<char *>(out);
    std::memcpy(base + L.off_meta + sizeof(int32_t) * UPAMD_META_STRIDE * t_begin, meta + UPAMD_META_STRIDE * t_begin,
                sizeof(int32_t) * UPAMD_META_STRIDE * (t_end - t_begin));
    float *X = reinterpret_cast<float *>(base + L.off_x);
    uint8_t *nmask = reinterpret_cast<uint8_t *>(base + L.off_nmask);
    int32_t *rowptr = reinterpret_cast<int32_t *>(base + L.off_rowptr);
    uint16_t *inc_nbr = reinterpret_cast<uint16_t *>(base + L.off_inc_nbr);
    uint16_t *he_src = reinterpret_cast<uint16_t *>(base + L.off_he_src);
    uint16_t *he_dst = reinterpret_cast<uint16_t *>(base + L.off_he_dst);
    uint8_t *he_live = reinterpret_cast<uint8_t *>(base + L.off_he_live);
    int32_t *he_slot = reinterpret_cast<int32_t *>(base + L.off_he_slot);
    uint16_t *rn_node = reinterpret_cast<uint16_t *>(base + L.off_rn_node);
    float *numerical = reinterpret_cast<float *>(base + L.off_numerical);
    float *cur = reinterpret_cast<float *>(base + L.off_cur);
    uint16_t *order = reinterpret_cast<uint16_t *>(base + L.off_order);
    int32_t *hinc_ptr = reinterpret_cast<int32_t *>(base + L.off_hinc_ptr);
    uint16_t *hinc_nbr = reinterpret_cast<uint16_t *>(base + L.off_hinc_nbr);
    uint16_t *hinc_he = reinterpret_cast<uint16_t *>(base + L.off_hinc_he);
    // rl-mlp fields (he_sel, xbar): the mean of the selected endpoints' features walks every live edge with a 14-way arg-max and
    // 23 double adds -- two thirds of a state's fill time (95 of 140 us on the build container).  An SGNN replay is planned without
    // them (upamd_pack_plan_ex flag 2) and skips all of it.
    const bool mlp_fields = L.off_xbar >= 0 && L.off_he_sel >= 0;
    uint16_t *he_sel = mlp_fields ? reinterpret_cast<uint16_t *>(base + L.off_he_sel) : nullptr;
    float *xbar = mlp_fields ? reinterpret_cast<float *>(base + L.off_xbar) : nullptr;
    const int F = L.node_dim, Fn = L.numerical_dim;

    int rc = parallel_for(t_end - t_begin, n_threads, [&](int64_t tt) -> int {
        const int64_t t = t_begin + tt;
        StateView s = view(ptrs, T, t);
        const int32_t *m = meta + t * UPAMD_META_STRIDE;
        const int n = m[0], e = m[1], nh = m[2], nr = m[3], stage = m[4], N = m[7], E = m[8];
        const int64_t o_node = m[9], o_edge = m[10], o_he = m[11], o_rn = m[12], o_rp = m[13];
        // every live endpoint inside the planned extent?  (always true behind the exact plan; the masks-only plan relies on it --
        // checked before anything indexes with an endpoint)
        for (int k = 0; k < E; ++k)
            if (s.edge_mask[k]) {
                const int64_t i = s.edge_index[2 * k], j = s.edge_index[2 * k + 1];
                if (i < 0 || j < 0 || i >= n || j >= n) return -5;
            }
        // node features + mask
        for (int v = 0; v < n; ++v) {
            float *dst = X + (o_node + v) * UPAMD_NODE_PAD;
            std::memcpy(dst, s.feat + (int64_t)v * F, sizeof(float) * F);
            for (int c = F; c < UPAMD_NODE_PAD; ++c) dst[c] = 0.f;
            nmask[o_node + v] = s.node_mask[v] ? 1 : 0;
        }
        // rl-mlp encoder: the endpoint that represents an edge (state_encoder.py:263-282) and the mean of those
        // endpoints' raw features over the live edges (the encoder is linear, so mean(W x + b) = W mean(x) + b)
        auto selected = [&](int k) -> int {
            const int i = (int)s.edge_index[2 * k], j = (int)s.edge_index[2 * k + 1];
            const float *xj = s.feat + (int64_t)j * F;
            int best = 0;
            for (int c = 1; c < std::min(F, UPAMD_MLP_TYPE_COLS); ++c)
                if (xj[c] > xj[best]) best = c;
            return best == UPAMD_MLP_FEASIBLE ? j : i;
        };
        if (mlp_fields) {
            double acc[UPAMD_NODE_PAD] = {0};
            for (int k = 0; k < E; ++k)
                if (s.edge_mask[k]) {
                    const float *xs = s.feat + (int64_t)selected(k) * F;
                    for (int c = 0; c < F; ++c) acc[c] += xs[c];
                }
            float *xb = xbar + t * UPAMD_NODE_PAD;
            for (int c = 0; c < UPAMD_NODE_PAD; ++c) xb[c] = (c < F && e > 0) ? (float)(acc[c] / e) : 0.f;
        }
        // candidates
        std::vector<int32_t> he_of_slot;
        if (stage == 0) {
            he_of_slot.assign(E, -1);
            int q = 0;
            for (int k = 0; k < E; ++k)
                if (s.land_mask[k]) {
                    const bool live = s.edge_mask[k] != 0;
                    he_of_slot[k] = q;
                    he_src[o_he + q] = live ? (uint16_t)s.edge_index[2 * k] : 0;
                    he_dst[o_he + q] = live ? (uint16_t)s.edge_index[2 * k + 1] : 0;
                    he_live[o_he + q] = live ? 1 : 0;
                    if (mlp_fields) he_sel[o_he + q] = live ? (uint16_t)selected(k) : 0;
                    he_slot[o_he + q] = k;
                    ++q;
                }
            if (q != nh) return -1;
        }
        if (stage == 1) {
            int q = 0;
            for (int v = 0; v < N; ++v)
                if (s.road_mask[v]) rn_node[o_rn + q++] = (uint16_t)v;
            if (q != nr) return -1;
        }
        // incidence CSR (counting sort by endpoint, slot order preserved)
        int32_t *rp = rowptr + o_rp;
        for (int v = 0; v <= n; ++v) rp[v] = 0;
        for (int k = 0; k < E; ++k)
            if (s.edge_mask[k]) {
                rp[s.edge_index[2 * k] + 1]++;
                rp[s.edge_index[2 * k + 1] + 1]++;
            }
        for (int v = 0; v < n; ++v) rp[v + 1] += rp[v];
        if (rp[n] != 2 * e) return -1;
        std::vector<int32_t> fill(rp, rp + n);
        uint16_t *nb = inc_nbr + 2 * o_edge;
        for (int k = 0; k < E; ++k)
            if (s.edge_mask[k]) {
                const int i = (int)s.edge_index[2 * k], j = (int)s.edge_index[2 * k + 1];
                nb[fill[i]++] = (uint16_t)j;
                nb[fill[j]++] = (uint16_t)i;
            }
        // processing order of the edge kernels: nodes by degree, descending (stable), so that the four
        // nodes a wave handles together have similar neighbour counts
        {
            std::vector<uint16_t> ord(n);
            for (int v = 0; v < n; ++v) ord[v] = (uint16_t)v;
            std::stable_sort(ord.begin(), ord.end(), [&](uint16_t a, uint16_t b) {
                return (rp[a + 1] - rp[a]) > (rp[b + 1] - rp[b]);
            });
            std::memcpy(order + o_node, ord.data(), sizeof(uint16_t) * n);
        }
        // candidate-incidence lists: for every LIVE candidate edge h = (i, j): i <- (j, h), j <- (i, h)
        {
            int32_t *hp = hinc_ptr + o_rp;
            for (int v = 0; v <= n; ++v) hp[v] = 0;
            for (int q = 0; q < nh; ++q)
                if (he_live[o_he + q]) {
                    hp[he_src[o_he + q] + 1]++;
                    hp[he_dst[o_he + q] + 1]++;
                }
            for (int v = 0; v < n; ++v) hp[v + 1] += hp[v];
            std::vector<int32_t> hf(hp, hp + n);
            uint16_t *hn = hinc_nbr + 2 * o_he, *hh = hinc_he + 2 * o_he;
            for (int q = 0; q < nh; ++q)
                if (he_live[o_he + q]) {
                    const int i = he_src[o_he + q], j = he_dst[o_he + q];
                    hn[hf[i]] = (uint16_t)j; hh[hf[i]] = (uint16_t)q; hf[i]++;
                    hn[hf[j]] = (uint16_t)i; hh[hf[j]] = (uint16_t)q; hf[j]++;
                }
        }
        // per-state dense fields
        std::memcpy(numerical + t * Fn, s.numerical, sizeof(float) * Fn);
        float *cdst = cur + t * UPAMD_NODE_PAD;
        std::memcpy(cdst, s.cur, sizeof(float) * F);
        for (int c = F; c < UPAMD_NODE_PAD; ++c) cdst[c] = 0.f;
        return 0;
    });
    if (rc == -5)
        return upamd::fail(UPAMD_E_REPLAN, "upamd_pack_fill: a live edge touches a node beyond the extent the node / road masks give "
                                           "(masks-only plan): plan again with exact = 1");
    if (rc != 0) return upamd::fail(UPAMD_E_INVALID, "upamd_pack_fill: meta table inconsistent with the states");
    return UPAMD_OK;
}
