// ctg_pathfind.cpp -- host-side tree tools behind the C ABI (no GPU involved):
// a greedy pairwise path finder and a greedy slicer.
//
// This is the slot the reference fills with its optional Rust accelerator
// (`cotengrust`: cotengra/pathfinders/path_basic.py:1351-1383 picks it up for
// `optimize_greedy`) and, for slicing, with the pure-Python `SliceFinder`
// (cotengra/slicer.py:204-430 over the incremental cost model of
// `ContractionCosts`, slicer.py:17-201).  The hyper-optimizers that drive
// these building blocks stay in the reference, on the host, unchanged; what is
// native here is the inner loop they call thousands of times -- and what the
// stand-alone front ends (`cotengra_amd.einsum`, `ContractionTree.slice_`) use
// when no tree is supplied.
//
// Semantics restated from the reference (not its code):
//  * greedy: every pair of tensors sharing an index is a candidate with
//      score = size(ab) / costmod - (size(a) + size(b)) * costmod
//    (path_basic.py:624-639); with a temperature the score becomes
//    sign(s) log|s| - T * gumbel() (Boltzmann sampling); the best candidate is
//    contracted, candidates of the new tensor with its neighbours are added;
//    indices shared by more than `max_neighbors` tensors (batch-like) do not
//    generate candidates; what is left disconnected is combined smallest first
//    (path_basic.py:1098-1106).  An index survives on an intermediate while it
//    still appears elsewhere (inputs + output), which is what makes hyper
//    indices work (core.py:246-258).
//  * slicing: removing an index of extent d divides the flops of every
//    contraction it is involved in and the size of every intermediate it is a
//    leg of by d, and multiplies the number of slices by d (slicer.py:60-70,
//    136-192).  Greedy rule: an index is worth the bits it takes off the
//    intermediates that are still too large; among the most useful indices
//    the one that leaves the smallest total flops is removed; repeat until
//    every intermediate fits the target.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <queue>
#include <random>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ctg_hip.h"

namespace {

struct Network {
    int64_t n_inputs = 0, n_inds = 0;
    std::vector<std::map<int64_t, int>> legs;  // per tensor: index -> appearances absorbed
    std::vector<int> appearances;              // per index: total over inputs + output
    std::vector<double> log2size;              // per index
    std::vector<double> size;                  // per index
};

bool build_network(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                   const int64_t* out_inds, int64_t n_inds, const double* sizes, Network& net) {
    if (n_inputs < 1 || n_inds < 0 || !offsets || (!inds && offsets[n_inputs] > 0) || !sizes) return false;
    net.n_inputs = n_inputs;
    net.n_inds = n_inds;
    net.legs.assign(n_inputs, {});
    net.appearances.assign(n_inds, 0);
    net.log2size.resize(n_inds);
    for (int64_t i = 0; i < n_inds; ++i) {
        if (!(sizes[i] >= 1.0)) return false;
        net.log2size[i] = std::log2(sizes[i]);
    }
    net.size.assign(sizes, sizes + n_inds);
    for (int64_t t = 0; t < n_inputs; ++t) {
        if (offsets[t + 1] < offsets[t]) return false;
        for (int64_t q = offsets[t]; q < offsets[t + 1]; ++q) {
            const int64_t ix = inds[q];
            if (ix < 0 || ix >= n_inds) return false;
            net.legs[t][ix] += 1;
            net.appearances[ix] += 1;
        }
    }
    for (int64_t q = 0; q < n_out; ++q) {
        if (out_inds[q] < 0 || out_inds[q] >= n_inds) return false;
        net.appearances[out_inds[q]] += 1;
    }
    return true;
}

// legs of the contraction of a and b: an index stays while it appears elsewhere
std::map<int64_t, int> contract_legs(const std::map<int64_t, int>& a, const std::map<int64_t, int>& b,
                                     const std::vector<int>& appearances) {
    std::map<int64_t, int> out = a;
    for (const auto& kv : b) out[kv.first] += kv.second;
    for (auto it = out.begin(); it != out.end();) {
        if (it->second >= appearances[it->first]) it = out.erase(it);
        else ++it;
    }
    return out;
}

double legs_log2size(const std::map<int64_t, int>& legs, const std::vector<double>& l2) {
    double s = 0;
    for (const auto& kv : legs) s += l2[kv.first];
    return s;
}

// element count as a product (exact below 2^53, like the reference's integers)
double legs_size(const std::map<int64_t, int>& legs, const std::vector<double>& sz) {
    double s = 1;
    for (const auto& kv : legs) s *= sz[kv.first];
    return s;
}

int fail(const char* msg);

}  // namespace

// error reporting shares the runtime's thread-local message (ctg_runtime.hip)
extern "C" __attribute__((visibility("hidden"))) void ctg_set_error_(const char* msg);
namespace {
int fail(const char* msg) {
    ctg_set_error_(msg);
    return CTG_E_INVALID;
}
}  // namespace

// cost of one contraction in the reconfiguration: `flops + write_factor * size`
// (rates == nullptr), or modelled seconds (see ctg_subtree_reconfigure_timed)
struct CostModel {
    double write_factor = 0;
    const double* rates = nullptr;   // complex MACs / s by floor(log2 K)
    int64_t n_rates = 0;
    double elem_rate = 1;            // elements / s through memory
    double narrow_factor = 0.8;      // MAC rate of steps whose narrower kept side has 16..63 columns

    // time model: the step runs at the matrix-core rate its contracted extent K
    // and its narrower kept side N allow, or at the memory rate
    double seconds(double macs, double elems, double kk, double nn) const {
        int lk = 0;
        while (lk + 1 < n_rates && std::ldexp(1.0, lk + 1) <= kk) ++lk;
        double rate = rates[lk];
        if (nn < 16) rate *= std::max(nn, 1.0) / 16.0;   // a matrix-core tile has 16 complex columns
        else if (nn < 64 && kk >= 64) rate *= narrow_factor;   // 128x32 block tiles (the K < 64 rates
                                                                // were measured on such steps already)
        return std::max(macs / rate, elems / elem_rate);
    }
};

extern "C" {

int ctg_path_greedy(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                    const int64_t* out_inds, int64_t n_inds, const double* sizes, double costmod,
                    double temperature, int64_t max_neighbors, uint64_t seed, int64_t* ssa_path) {
    Network net;
    if (!ssa_path && n_inputs > 1) return fail("ctg_path_greedy: null output");
    if (!build_network(n_inputs, offsets, inds, n_out, out_inds, n_inds, sizes, net))
        return fail("ctg_path_greedy: malformed network");
    if (!(costmod > 0)) return fail("ctg_path_greedy: costmod must be positive");
    if (n_inputs == 1) return CTG_OK;

    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> uni(1e-300, 1.0);
    auto gumbel = [&]() { return -std::log(-std::log(uni(rng))); };
    // sizes as doubles: 2^1000 is representable, far above any real tensor
    auto score_of = [&](double sa, double sb, double sab) {
        const double s = sab / costmod - (sa + sb) * costmod;
        if (temperature == 0.0) return s;
        if (s > 0) return std::log(s) - temperature * gumbel();
        if (s < 0) return -std::log(-s) - temperature * gumbel();
        return -temperature * gumbel();
    };

    std::unordered_map<int64_t, std::map<int64_t, int>> nodes;  // alive tensors by ssa id
    std::unordered_map<int64_t, double> node_l2;   // element count of every alive tensor
    std::vector<std::set<int64_t>> edges(n_inds);                // index -> alive tensors
    for (int64_t t = 0; t < n_inputs; ++t) {
        nodes[t] = net.legs[t];
        node_l2[t] = legs_size(net.legs[t], net.size);
        for (const auto& kv : net.legs[t]) edges[kv.first].insert(t);
    }

    struct Cand {
        double score;
        int64_t c, i, j;
        bool operator<(const Cand& o) const {  // min-heap on (score, c)
            return score != o.score ? score > o.score : c > o.c;
        }
    };
    std::priority_queue<Cand> queue;
    int64_t counter = 0;
    auto push = [&](int64_t i, int64_t j) {
        const auto k = contract_legs(nodes[i], nodes[j], net.appearances);
        queue.push(Cand{score_of(node_l2[i], node_l2[j], legs_size(k, net.size)), counter++, i, j});
    };
    for (int64_t ix = 0; ix < n_inds; ++ix) {
        const auto& ts = edges[ix];
        if (max_neighbors > 0 && (int64_t)ts.size() > max_neighbors) continue;  // batch-like index
        for (auto a = ts.begin(); a != ts.end(); ++a)
            for (auto b = std::next(a); b != ts.end(); ++b) push(*a, *b);
    }

    int64_t next_ssa = n_inputs, n_steps = 0;
    auto contract = [&](int64_t i, int64_t j) {
        auto k = contract_legs(nodes[i], nodes[j], net.appearances);
        for (const auto& kv : nodes[i]) edges[kv.first].erase(i);
        for (const auto& kv : nodes[j]) edges[kv.first].erase(j);
        nodes.erase(i);
        nodes.erase(j);
        const int64_t id = next_ssa++;
        for (const auto& kv : k) edges[kv.first].insert(id);
        node_l2[id] = legs_size(k, net.size);
        nodes[id] = std::move(k);
        ssa_path[2 * n_steps] = i;
        ssa_path[2 * n_steps + 1] = j;
        ++n_steps;
        return id;
    };

    while (!queue.empty()) {
        const Cand c = queue.top();
        queue.pop();
        if (!nodes.count(c.i) || !nodes.count(c.j)) continue;  // stale
        const int64_t k = contract(c.i, c.j);
        // neighbours in the order the reference meets them (by leg, then by id),
        // each once: the push order breaks score ties
        std::vector<int64_t> nbrs;
        std::set<int64_t> seen;
        for (const auto& kv : nodes[k]) {
            const auto& ts = edges[kv.first];
            if (max_neighbors > 0 && (int64_t)ts.size() > max_neighbors) continue;
            for (int64_t t : ts)
                if (t != k && seen.insert(t).second) nbrs.push_back(t);
        }
        for (int64_t l : nbrs) push(k, l);
    }

    // disconnected remainder: smallest two first
    while (nodes.size() > 1) {
        std::vector<std::pair<double, int64_t>> by_size;
        for (const auto& kv : nodes) by_size.push_back({node_l2[kv.first], kv.first});
        std::sort(by_size.begin(), by_size.end());
        contract(by_size[0].second, by_size[1].second);
    }
    return n_steps == n_inputs - 1 ? CTG_OK : fail("ctg_path_greedy: internal error");
}

int ctg_slice_greedy(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                     const int64_t* out_inds, int64_t n_inds, const double* sizes,
                     const int64_t* ssa_path, double target_log2_size, int allow_outer,
                     int64_t max_sliced, int64_t* sliced, int64_t* n_sliced) {
    Network net;
    if (!sliced || !n_sliced || (!ssa_path && n_inputs > 1)) return fail("ctg_slice_greedy: null argument");
    if (!build_network(n_inputs, offsets, inds, n_out, out_inds, n_inds, sizes, net))
        return fail("ctg_slice_greedy: malformed network");
    *n_sliced = 0;
    if (n_inputs == 1) return CTG_OK;

    // contractions of the tree: involved indices, surviving legs, log2 flops, log2 size
    struct Con {
        std::vector<int64_t> involved, legs;
        double l2flops, l2size;
    };
    std::vector<Con> cons;
    {
        std::unordered_map<int64_t, std::map<int64_t, int>> nodes;
        for (int64_t t = 0; t < n_inputs; ++t) nodes[t] = net.legs[t];
        for (int64_t s = 0; s < n_inputs - 1; ++s) {
            const int64_t i = ssa_path[2 * s], j = ssa_path[2 * s + 1];
            if (!nodes.count(i) || !nodes.count(j) || i == j) return fail("ctg_slice_greedy: bad ssa path");
            Con c;
            std::map<int64_t, int> inv = nodes[i];
            for (const auto& kv : nodes[j]) inv[kv.first] += kv.second;
            auto k = contract_legs(nodes[i], nodes[j], net.appearances);
            c.l2flops = 0;
            for (const auto& kv : inv) {
                c.involved.push_back(kv.first);
                c.l2flops += net.log2size[kv.first];
            }
            c.l2size = 0;
            for (const auto& kv : k) {
                c.legs.push_back(kv.first);
                c.l2size += net.log2size[kv.first];
            }
            cons.push_back(std::move(c));
            nodes.erase(i);
            nodes.erase(j);
            nodes[n_inputs + s] = std::move(k);
        }
    }
    std::vector<char> is_out(n_inds, 0), gone(n_inds, 0);
    for (int64_t q = 0; q < n_out; ++q) is_out[out_inds[q]] = 1;

    auto total_flops_without = [&](int64_t ix) {  // relative units: sum of 2^l2flops, times d
        double f = 0;
        const double d = net.log2size[ix];
        for (const Con& c : cons) {
            const bool has = std::find(c.involved.begin(), c.involved.end(), ix) != c.involved.end();
            f += std::exp2(c.l2flops - (has ? d : 0.0));
        }
        return f * std::exp2(d);
    };

    for (;;) {
        double mx = -1;
        for (const Con& c : cons) mx = std::max(mx, c.l2size);
        if (mx <= target_log2_size + 1e-9) break;
        if (*n_sliced >= max_sliced) return fail("ctg_slice_greedy: more sliced indices than the caller allows");
        // candidates: legs of the intermediates that are still too large.  An index
        // is worth the excess (in bits, capped by its own extent) it takes off each
        // of them; among the most useful ones the cheapest in total flops wins.
        std::map<int64_t, double> gain;
        for (const Con& c : cons) {
            const double excess = c.l2size - target_log2_size;
            if (excess <= 1e-9) continue;
            for (int64_t ix : c.legs)
                if (!gone[ix] && (allow_outer || !is_out[ix]) && net.log2size[ix] > 0)
                    gain[ix] += std::min(excess, net.log2size[ix]);
        }
        if (gain.empty()) return fail("ctg_slice_greedy: target size unreachable (only output indices left)");
        double best_gain = 0;
        for (const auto& kv : gain) best_gain = std::max(best_gain, kv.second);
        int64_t best = -1;
        double best_flops = 0;
        for (const auto& kv : gain) {
            if (kv.second < 0.9 * best_gain) continue;
            const double f = total_flops_without(kv.first);
            if (best < 0 || f < best_flops) {
                best = kv.first;
                best_flops = f;
            }
        }
        const double d = net.log2size[best];
        for (Con& c : cons) {
            auto it = std::find(c.involved.begin(), c.involved.end(), best);
            if (it != c.involved.end()) {
                c.involved.erase(it);
                c.l2flops -= d;
            }
            auto jt = std::find(c.legs.begin(), c.legs.end(), best);
            if (jt != c.legs.end()) {
                c.legs.erase(jt);
                c.l2size -= d;
            }
        }
        gone[best] = 1;
        sliced[(*n_sliced)++] = best;
    }
    return CTG_OK;
}


// Subtree reconfiguration (reference `ContractionTree.subtree_reconfigure`,
// core.py:2316-2449): repeatedly take an internal node, grow a subtree below it
// breadth-first until it has `subtree_size` leaves (themselves arbitrary
// subtrees, treated as fixed tensors), find the *optimal* contraction order of
// those leaves by dynamic programming over subsets, and splice it in if it is
// cheaper.  Cost of one contraction = flops + write_factor * size of its result
// (the reference's `combo-<f>` objective, scoring.py; write_factor = 0 is plain
// flops).  Nodes are visited by decreasing cost (`select="max"`); a node whose
// subtree came out unchanged is not revisited until something below it changes.
static int subtree_reconfigure_impl(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                                   const int64_t* out_inds, int64_t n_inds, const double* sizes,
                                   const int64_t* ssa_path_in, int64_t subtree_size, int64_t maxiter,
                                   const CostModel& cm, int64_t* ssa_path_out) {
    Network net;
    if ((!ssa_path_in || !ssa_path_out) && n_inputs > 1) return fail("ctg_subtree_reconfigure: null argument");
    if (!build_network(n_inputs, offsets, inds, n_out, out_inds, n_inds, sizes, net))
        return fail("ctg_subtree_reconfigure: malformed network");
    if (subtree_size < 2 || subtree_size > 16) return fail("ctg_subtree_reconfigure: subtree_size must be 2..16");
    const int64_t n = n_inputs;
    if (n < 2) return CTG_OK;
    if (n == 2) {
        ssa_path_out[0] = ssa_path_in[0];
        ssa_path_out[1] = ssa_path_in[1];
        return CTG_OK;
    }

    struct Node {
        int64_t l = -1, r = -1, parent = -1;
        std::map<int64_t, int> legs;
        double cost = 0;   // flops + write_factor * size of this contraction (0 for leaves)
        bool alive = true, settled = false;
    };
    std::vector<Node> nodes(n);
    for (int64_t t = 0; t < n; ++t) nodes[t].legs = net.legs[t];
    auto pair_cost = [&](const std::map<int64_t, int>& a, const std::map<int64_t, int>& b,
                         const std::map<int64_t, int>& k) {
        double flops = 1;
        for (const auto& kv : a) flops *= net.size[kv.first];
        for (const auto& kv : b)
            if (!a.count(kv.first)) flops *= net.size[kv.first];
        if (cm.rates == nullptr) return flops + cm.write_factor * legs_size(k, net.size);
        const double sa = legs_size(a, net.size), sb = legs_size(b, net.size), sc = legs_size(k, net.size);
        double kk = 1, keep_a = 1, keep_b = 1;
        for (const auto& kv : a) {
            if (k.count(kv.first)) {
                if (!b.count(kv.first)) keep_a *= net.size[kv.first];
            } else if (b.count(kv.first)) {
                kk *= net.size[kv.first];
            }
        }
        for (const auto& kv : b)
            if (k.count(kv.first) && !a.count(kv.first)) keep_b *= net.size[kv.first];
        return cm.seconds(flops, sa + sb + sc, kk, std::min(keep_a, keep_b));
    };
    int64_t root = -1;
    {
        std::vector<char> used(2 * n, 0);
        for (int64_t s = 0; s < n - 1; ++s) {
            const int64_t i = ssa_path_in[2 * s], j = ssa_path_in[2 * s + 1];
            if (i < 0 || j < 0 || i >= n + s || j >= n + s || i == j || used[i] || used[j])
                return fail("ctg_subtree_reconfigure: bad ssa path");
            used[i] = used[j] = 1;
            Node p;
            p.l = i;
            p.r = j;
            p.legs = contract_legs(nodes[i].legs, nodes[j].legs, net.appearances);
            p.cost = pair_cost(nodes[i].legs, nodes[j].legs, p.legs);
            nodes.push_back(std::move(p));
            nodes[i].parent = nodes[j].parent = n + s;
        }
        root = 2 * n - 2;
    }

    const int64_t S = subtree_size;
    std::vector<double> best, msize;
    std::vector<int32_t> split;
    std::vector<uint64_t> bits;
    if (maxiter <= 0) maxiter = std::min<int64_t>(n, 1024);
    for (int64_t iter = 0; iter < maxiter; ++iter) {
        // the most expensive node not known to be locally optimal
        int64_t pick = -1;
        for (int64_t v = n; v < (int64_t)nodes.size(); ++v)
            if (nodes[v].alive && !nodes[v].settled && (pick < 0 || nodes[v].cost > nodes[pick].cost)) pick = v;
        if (pick < 0) break;
        // frontier: breadth-first expansion of internal nodes until S leaves
        std::vector<int64_t> frontier, inner, fifo{pick};
        int64_t n_leaves = 1;
        for (size_t q = 0; q < fifo.size(); ++q) {
            const int64_t v = fifo[q];
            if (nodes[v].l < 0 || n_leaves >= S) {
                frontier.push_back(v);   // stays a (fixed) leaf of the subtree
                continue;
            }
            inner.push_back(v);
            ++n_leaves;
            fifo.push_back(nodes[v].l);
            fifo.push_back(nodes[v].r);
        }
        const int m = (int)frontier.size();
        if (m < 3) {
            nodes[pick].settled = true;
            continue;
        }
        double cur = 0;
        for (int64_t v : inner) cur += nodes[v].cost;
        // dynamic programming over subsets of the frontier.  Indices are numbered
        // locally and every subset's legs are a bitset: an index whose appearances
        // all lie inside the frontier ("internal") survives in a subset until the
        // subset holds every leaf that carries it; any other index survives always.
        const int full = (1 << m) - 1;
        std::vector<int64_t> loc_ix;
        std::vector<uint32_t> holders;     // which frontier leaves carry the index
        std::vector<int> inside;           // appearances inside the frontier
        {
            std::unordered_map<int64_t, int> loc;
            for (int t = 0; t < m; ++t)
                for (const auto& kv : nodes[frontier[t]].legs) {
                    auto it = loc.find(kv.first);
                    if (it == loc.end()) {
                        it = loc.emplace(kv.first, (int)loc_ix.size()).first;
                        loc_ix.push_back(kv.first);
                        holders.push_back(0);
                        inside.push_back(0);
                    }
                    holders[it->second] |= 1u << t;
                    inside[it->second] += kv.second;
                }
        }
        const int L = (int)loc_ix.size(), W = (L + 63) / 64;
        std::vector<double> lsz(L);
        for (int i = 0; i < L; ++i) lsz[i] = net.size[loc_ix[i]];
        bits.assign((size_t)(full + 1) * W, 0);
        msize.assign(full + 1, 1.0);
        for (int i = 0; i < L; ++i) {
            const bool internal = inside[i] >= net.appearances[loc_ix[i]];
            const uint32_t h = holders[i];
            for (int mask = 1; mask <= full; ++mask) {
                if (!(mask & h)) continue;
                if (internal && !(h & ~(uint32_t)mask)) continue;
                bits[(size_t)mask * W + (i >> 6)] |= 1ull << (i & 63);
                msize[mask] *= lsz[i];
            }
        }
        // product of the extents of the indices in (x & y & ~z) / (x & y & z)
        auto prod_bits = [&](const uint64_t* x, const uint64_t* y, const uint64_t* z, bool with_z) {
            double p = 1;
            for (int w = 0; w < W; ++w) {
                uint64_t v = x[w] & y[w] & (with_z ? z[w] : ~z[w]);
                while (v) {
                    p *= lsz[(w << 6) + __builtin_ctzll(v)];
                    v &= v - 1;
                }
            }
            return p;
        };
        auto split_cost = [&](int left, int right, int mask) {
            const uint64_t* bl = &bits[(size_t)left * W];
            const uint64_t* br = &bits[(size_t)right * W];
            const uint64_t* bo = &bits[(size_t)mask * W];
            // every index of either side is involved once: shared ones divide out
            const double contracted = prod_bits(bl, br, bo, false), batch = prod_bits(bl, br, bo, true);
            const double flops = msize[left] * msize[right] / (contracted * batch);
            if (cm.rates == nullptr) return flops + cm.write_factor * msize[mask];
            const double keep_a = msize[left] / (contracted * batch), keep_b = msize[right] / (contracted * batch);
            return cm.seconds(flops, msize[left] + msize[right] + msize[mask], contracted, std::min(keep_a, keep_b));
        };
        best.assign(full + 1, 0.0);
        split.assign(full + 1, 0);
        for (int mask = 1; mask <= full; ++mask) {
            const int low = mask & -mask;
            if (mask == low) continue;
            double b = -1;
            int bs = 0;
            // splits with the lowest member on the left side (each unordered split once)
            const int rest = mask ^ low;
            for (int sub = rest;; sub = (sub - 1) & rest) {
                const int left = low | (rest ^ sub), right = sub;   // right may not be empty
                if (right != 0) {
                    const double c = best[left] + best[right] + split_cost(left, right, mask);
                    if (b < 0 || c < b) {
                        b = c;
                        bs = left;
                    }
                }
                if (sub == 0) break;
            }
            best[mask] = b;
            split[mask] = bs;
        }
        if (!(best[full] < cur * (1.0 - 1e-12))) {
            nodes[pick].settled = true;
            continue;
        }
        // splice the optimal order in: the internal nodes are recycled for it
        size_t reuse = 0;
        std::vector<int64_t> freed = inner;   // pick is inner[0]: it stays the subtree root
        std::function<int64_t(int, int64_t)> build = [&](int mask, int64_t as) -> int64_t {
            if ((mask & (mask - 1)) == 0) return frontier[__builtin_ctz(mask)];
            const int64_t id = as >= 0 ? as : freed[++reuse];
            const int left = split[mask], right = mask ^ left;
            const int64_t a = build(left, -1), b2 = build(right, -1);
            Node& nd = nodes[id];
            nd.l = a;
            nd.r = b2;
            nd.legs = contract_legs(nodes[a].legs, nodes[b2].legs, net.appearances);
            nd.cost = pair_cost(nodes[a].legs, nodes[b2].legs, nd.legs);
            nd.settled = false;
            nodes[a].parent = nodes[b2].parent = id;
            return id;
        };
        build(full, pick);
        // everything above may now be improvable again
        for (int64_t v = nodes[pick].parent; v >= 0; v = nodes[v].parent) nodes[v].settled = false;
    }

    // emit the SSA path (post-order)
    std::vector<int64_t> ssa_of(nodes.size(), -1);
    for (int64_t t = 0; t < n; ++t) ssa_of[t] = t;
    int64_t next_ssa = n, step = 0;
    std::vector<std::pair<int64_t, int>> stack{{root, 0}};
    while (!stack.empty()) {
        auto [v, st] = stack.back();
        stack.pop_back();
        if (nodes[v].l < 0) continue;
        if (st == 0) {
            stack.push_back({v, 1});
            stack.push_back({nodes[v].r, 0});
            stack.push_back({nodes[v].l, 0});
        } else {
            ssa_path_out[2 * step] = ssa_of[nodes[v].l];
            ssa_path_out[2 * step + 1] = ssa_of[nodes[v].r];
            ssa_of[v] = next_ssa++;
            ++step;
        }
    }
    return step == n - 1 ? CTG_OK : fail("ctg_subtree_reconfigure: internal error");
}

int ctg_subtree_reconfigure(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                            const int64_t* out_inds, int64_t n_inds, const double* sizes,
                            const int64_t* ssa_path_in, int64_t subtree_size, int64_t maxiter,
                            double write_factor, int64_t* ssa_path_out) {
    CostModel cm;
    cm.write_factor = write_factor;
    return subtree_reconfigure_impl(n_inputs, offsets, inds, n_out, out_inds, n_inds, sizes, ssa_path_in,
                                    subtree_size, maxiter, cm, ssa_path_out);
}

// The same search with a machine model as the objective: a contraction costs
// max(MACs / mac_rate[floor(log2 K)], (size_a + size_b + size_out) / elem_rate)
// seconds, where K is its contracted extent (the last table entry serves every
// larger K) and the MAC rate is scaled by N/16 when the narrower kept side N
// has fewer than 16 columns, by 0.8 when it has 16..63 and K >= 64 (128x32 tiles).  The tables are the caller's measurements of the
// executor's kernels (cotengra_amd.pathfind.MI355X_C64).
int ctg_subtree_reconfigure_timed(int64_t n_inputs, const int64_t* offsets, const int64_t* inds,
                                  int64_t n_out, const int64_t* out_inds, int64_t n_inds,
                                  const double* sizes, const int64_t* ssa_path_in, int64_t subtree_size,
                                  int64_t maxiter, const double* mac_rate_by_log2k, int64_t n_rates,
                                  double elem_rate, int64_t* ssa_path_out) {
    if (!mac_rate_by_log2k || n_rates < 1 || !(elem_rate > 0))
        return fail("ctg_subtree_reconfigure_timed: bad machine model");
    for (int64_t i = 0; i < n_rates; ++i)
        if (!(mac_rate_by_log2k[i] > 0)) return fail("ctg_subtree_reconfigure_timed: bad machine model");
    CostModel cm;
    cm.rates = mac_rate_by_log2k;
    cm.n_rates = n_rates;
    cm.elem_rate = elem_rate;
    return subtree_reconfigure_impl(n_inputs, offsets, inds, n_out, out_inds, n_inds, sizes, ssa_path_in,
                                    subtree_size, maxiter, cm, ssa_path_out);
}

}  // extern "C"
