// ctg_lds_host.hip -- host side of the LDS-resident subtrees (kernel: ctg_lds_run.hip, plan:
// cotengra_amd/ldsrun.py): validation of the component descriptors a plan carries (every table inside
// the blob, every LDS / global address inside its buffer -- the descriptor comes through the C ABI and is
// not trusted) and the packing of a component into the blob its workgroup copies into LDS.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "ctg_exec_state.h"

using namespace ctg;

namespace {

int lfail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctg_set_error_(buf);
    return code;
}

bool in_blob(const ctg_plan* p, int64_t off, int64_t len) {
    return off >= 0 && len >= 1 && off + len <= (int64_t)p->tables.size();
}

void tab_range(const ctg_plan* p, int64_t off, int64_t len, int64_t* mn, int64_t* mx) {
    *mn = INT64_MAX;
    *mx = INT64_MIN;
    for (int64_t i = 0; i < len; ++i) {
        *mn = std::min(*mn, p->tables[off + i]);
        *mx = std::max(*mx, p->tables[off + i]);
    }
}

int64_t space_cap(const ctg_plan* p, int64_t space) {
    switch (space) {
        case SPACE_INPUTS: return p->inputs_elems;
        case SPACE_ARENA: return p->arena_elems;
        case SPACE_RESULT: return p->result_elems;
    }
    return -1;
}

int log2_of(int64_t v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1ll << s) < v) ++s;
    return s;
}

}  // namespace

// Every component descriptor the step records point at: structure, table ranges, that every LDS address
// stays inside the component's data area and every global address inside its space.
int ctg_lds_validate(const ctg_plan* p) {
    const int64_t isz = ctg_item_size(p->dtype);
    std::map<int64_t, int64_t> comp_desc;   // component id -> descriptor offset
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        const int64_t c = r[W_LDS_COMP];
        if (c == 0) continue;
        if (c < 0 || c > p->n_steps) return lfail(CTG_E_INVALID, "step %lld: bad LDS component", (long long)s);
        if (r[W_KIND] != KIND_PAIR && r[W_KIND] != KIND_SINGLE)
            return lfail(CTG_E_INVALID, "step %lld: only pair and single steps are members of an LDS component", (long long)s);
        if (r[W_INVARIANT] == 1) return lfail(CTG_E_INVALID, "step %lld: a slice-invariant step cannot be a member", (long long)s);
        const int64_t d = r[W_LDS_DESC];
        if (!in_blob(p, d, LR_HEAD_WORDS)) return lfail(CTG_E_BOUNDS, "step %lld: LDS descriptor outside the blob", (long long)s);
        auto it = comp_desc.find(c - 1);
        if (it == comp_desc.end()) comp_desc[c - 1] = d;
        else if (it->second != d) return lfail(CTG_E_INVALID, "step %lld: two descriptors for one LDS component", (long long)s);
    }
    for (const auto& cd : comp_desc) {
        const int64_t cid = cd.first, d = cd.second;
        const long long cl = (long long)cid;
        const int64_t* h = &p->tables[d];
        if (h[LH_MAGIC] != LR_MAGIC || h[LH_ID] != cid) return lfail(CTG_E_INVALID, "LDS component %lld: bad descriptor", cl);
        const int64_t n_rec = h[LH_NREC], elems = h[LH_ELEMS];
        if (n_rec < 1 || n_rec > 4096 || elems < 0 || elems * isz > LDS_RUN_MAX_BYTES)
            return lfail(CTG_E_INVALID, "LDS component %lld: bad size", cl);
        if (!in_blob(p, d, LR_HEAD_WORDS + n_rec * LR_WORDS))
            return lfail(CTG_E_BOUNDS, "LDS component %lld: records outside the blob", cl);
        int64_t phase = 0, sharing = -1;
        std::vector<int64_t> seen;
        for (int64_t i = 0; i < n_rec; ++i) {
            const int64_t* q = h + LR_HEAD_WORDS + i * LR_WORDS;
            const int64_t kind = q[LR_KIND], main = q[LR_MAIN], gop = q[LR_GOP];
            if (kind != 0 && kind != 1) return lfail(CTG_E_INVALID, "LDS component %lld: bad record kind", cl);
            if (q[LR_PHASE] < phase || (kind == 0) != (q[LR_PHASE] == 0))
                return lfail(CTG_E_INVALID, "LDS component %lld: bad phase order", cl);
            phase = q[LR_PHASE];
            if (main < 0 || main >= p->n_steps || p->steps[main * STEP_WORDS + W_LDS_COMP] != cid + 1)
                return lfail(CTG_E_INVALID, "LDS component %lld: record of a step that is not a member", cl);
            const int64_t* m = &p->steps[main * STEP_WORDS];
            if (sharing < 0) sharing = m[W_INVARIANT];
            if (m[W_INVARIANT] != sharing) return lfail(CTG_E_INVALID, "LDS component %lld: members of two sharing classes", cl);
            const int64_t R = q[LR_R], K = q[LR_K], N = q[LR_N], row_lo = q[LR_ROW_LO], row_hi = q[LR_ROW_HI_LEN],
                          k_lo = q[LR_K_LO], k_hi = q[LR_K_HI_LEN];
            if (R < 1 || K < 1 || N < 1 || row_lo < 1 || row_hi < 1 || k_lo < 1 || k_hi < 1 || R > (1 << 24) ||
                K > (1 << 16) || N > (1 << 16) || row_lo * row_hi != R || k_lo * k_hi != K)
                return lfail(CTG_E_INVALID, "LDS component %lld: bad extents", cl);
            if (kind == 0) {
                if (N != 1 || q[LR_A_LDS] != 0 || q[LR_C_LDS] != 1 || (gop != 0 && gop != 1) ||
                    (gop == 1 && m[W_KIND] != KIND_PAIR))
                    return lfail(CTG_E_INVALID, "LDS component %lld: bad load record", cl);
                if (m[W_KIND] == KIND_SINGLE) seen.push_back(main);
            } else {
                if (q[LR_A_LDS] != 1 || q[LR_B_LDS] != 1 || m[W_KIND] != KIND_PAIR ||
                    (q[LR_C_LDS] == 1 ? gop != -1 : gop != 2))
                    return lfail(CTG_E_INVALID, "LDS component %lld: bad pair record", cl);
                seen.push_back(main);
            }
            // operands: tables inside the blob, addresses inside the data area / the global space
            struct Op { int lds_w, off_w; int64_t rhi, rlo, khi, klo, n; bool used; int main_space, main_off, main_leaf; };
            const Op ops[3] = {
                {LR_A_LDS, LR_A_OFF, q[LR_ROWA_HI], q[LR_ROWA_LO], q[LR_KA_HI], q[LR_KA_LO], -1, true,
                 gop == 1 ? W_B_SPACE : W_A_SPACE, gop == 1 ? W_B_OFF : W_A_OFF, gop == 1 ? W_B_LEAF : W_A_LEAF},
                {LR_B_LDS, LR_B_OFF, q[LR_ROWB_HI], q[LR_ROWB_LO], q[LR_KB_HI], q[LR_KB_LO], q[LR_NB], kind == 1, 0, 0, 0},
                {LR_C_LDS, LR_C_OFF, q[LR_ROWC_HI], q[LR_ROWC_LO], -1, -1, kind == 1 ? q[LR_NC] : -1, true,
                 W_C_SPACE, W_C_OFF, W_C_LEAF},
            };
            for (int o = 0; o < 3; ++o) {
                const Op& op = ops[o];
                if (!op.used) continue;
                int64_t lo_addr = 0, hi_addr = 0;
                const struct { int64_t t, len; } tabs[5] = {{op.rhi, row_hi}, {op.rlo, row_lo}, {op.khi, k_hi}, {op.klo, k_lo}, {op.n, N}};
                for (const auto& t : tabs) {
                    if (t.t < 0) continue;
                    if (!in_blob(p, t.t, t.len)) return lfail(CTG_E_BOUNDS, "LDS component %lld: table outside the blob", cl);
                    int64_t mn, mx;
                    tab_range(p, t.t, t.len, &mn, &mx);
                    if (mn < 0) return lfail(CTG_E_BOUNDS, "LDS component %lld: negative offset", cl);
                    lo_addr += mn;
                    hi_addr += mx;
                }
                if (o != 1 && (op.rhi < 0 || op.rlo < 0)) return lfail(CTG_E_INVALID, "LDS component %lld: missing row table", cl);
                if (q[op.lds_w] == 1) {
                    const int64_t off = q[op.off_w];
                    if (off < 0 || off + hi_addr >= elems || hi_addr >= 65536)
                        return lfail(CTG_E_BOUNDS, "LDS component %lld: record %lld operand %c reaches LDS element %lld of %lld",
                                     cl, (long long)i, "ABC"[o], (long long)(off + hi_addr), (long long)elems);
                } else {
                    const int64_t space = m[op.main_space], off = m[op.main_off], leaf = m[op.main_leaf];
                    const int64_t cap = space_cap(p, space);
                    if (cap < 0 || leaf < -1 || leaf > p->n_inputs) return lfail(CTG_E_INVALID, "LDS component %lld: bad global operand", cl);
                    if (o == 2 && space == SPACE_INPUTS) return lfail(CTG_E_INVALID, "LDS component %lld: writes into the inputs space", cl);
                    const int64_t top = off + (leaf >= 0 ? p->max_soff[leaf] : 0) + hi_addr;
                    if (off < 0 || top >= cap || hi_addr >= (1ll << 31))
                        return lfail(CTG_E_BOUNDS, "LDS component %lld: record %lld operand %c reaches element %lld of a space of %lld",
                                     cl, (long long)i, "ABC"[o], (long long)top, (long long)cap);
                }
            }
        }
        // every member is lowered exactly once
        std::sort(seen.begin(), seen.end());
        for (size_t i = 1; i < seen.size(); ++i)
            if (seen[i] == seen[i - 1]) return lfail(CTG_E_INVALID, "LDS component %lld: a member lowered twice", cl);
        for (int64_t s = 0; s < p->n_steps; ++s)
            if (p->steps[s * STEP_WORDS + W_LDS_COMP] == cid + 1 && !std::binary_search(seen.begin(), seen.end(), s))
                return lfail(CTG_E_INVALID, "LDS component %lld: member step %lld has no record", cl, (long long)s);
    }
    return CTG_OK;
}

// Pack the components the executor will run LDS-resident: per component one blob (LdsStepDev records, then
// tables with 16-bit LDS offsets) in device memory.  A component whose blob + data do not fit the LDS, or
// whose offsets do not fit their fields, stays off: its members launch one by one as ordinary steps.
// Fills e->lds_comp_of / lds_first / lds_count / lds_bytes, e->d_lds_comps, e->d_lds_blob.
int ctg_lds_build(ctg_exec* e) {
    const ctg_plan* p = e->plan;
    const int64_t isz = ctg_item_size(p->dtype);
    e->lds_comp_of.assign(p->n_steps, -1);
    for (int c = 0; c < 2; ++c) e->lds_first[c] = e->lds_count[c] = e->lds_bytes[c] = 0;
    if (e->d_lds_comps) (void)hipFree(e->d_lds_comps);
    if (e->d_lds_blob) (void)hipFree(e->d_lds_blob);
    e->d_lds_comps = nullptr;
    e->d_lds_blob = nullptr;
    // (strip_exponent keeps a scale per step: the ordinary steps run)
    if (e->strip || env_on("CTG_NO_LDS_RUNS")) return CTG_OK;
    std::map<int64_t, int64_t> comp_desc;
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        if (r[W_LDS_COMP] > 0) comp_desc[r[W_LDS_COMP] - 1] = r[W_LDS_DESC];
    }
    if (comp_desc.empty()) return CTG_OK;
    struct Packed { int64_t cid; std::vector<char> blob; uint32_t n_steps, data_off; int lds_bytes; bool shared; };
    const bool no_mfma = env_on("CTG_LDS_NO_MFMA");
    std::vector<Packed> packed;
    for (const auto& cd : comp_desc) {
        const int64_t* h = &p->tables[cd.second];
        const int64_t n_rec = h[LH_NREC], elems = h[LH_ELEMS];
        Packed pk;
        pk.cid = cd.first;
        pk.n_steps = (uint32_t)n_rec;
        std::vector<LdsStepDev> recs((size_t)n_rec);
        std::vector<char> tabs;
        bool ok = true;
        auto put = [&](const void* src, size_t bytes) -> uint32_t {
            const size_t at = (tabs.size() + 15) / 16 * 16;
            tabs.resize(at + bytes);
            memcpy(tabs.data() + at, src, bytes);
            return (uint32_t)at;
        };
        const size_t rec_bytes = (size_t)n_rec * sizeof(LdsStepDev);
        // per phase: small steps are dealt to single waves (least loaded first), large ones shared by all
        std::map<int64_t, std::vector<int64_t>> wave_load;
        std::map<int64_t, int64_t> mfma_rot;
        pk.shared = false;
        for (int64_t i = 0; i < n_rec && ok; ++i) {
            const int64_t* q = h + LR_HEAD_WORDS + i * LR_WORDS;
            const int64_t main = q[LR_MAIN], gop = q[LR_GOP], kind = q[LR_KIND];
            const int64_t* m = &p->steps[main * STEP_WORDS];
            pk.shared = m[W_INVARIANT] == 2;
            if (e->invariant[main]) ok = false;
            LdsStepDev& st = recs[(size_t)i];
            memset(&st, 0, sizeof(st));
            st.kind = (int32_t)kind;
            st.phase = (int32_t)q[LR_PHASE];
            st.R = (int32_t)q[LR_R];
            st.K = (int32_t)q[LR_K];
            st.N = (int32_t)q[LR_N];
            st.row_lo = (int32_t)q[LR_ROW_LO];
            st.row_shift = log2_of(q[LR_ROW_LO]);
            st.row_magic = 0;
            if (st.row_shift < 0) {
                if (q[LR_R] >= 65536) ok = false;
                st.row_magic = (uint32_t)((1ull << 32) / (uint64_t)q[LR_ROW_LO]) + 1u;
            }
            st.a_off = q[LR_A_LDS] ? (int32_t)q[LR_A_OFF] : -1;
            st.b_off = (kind == 1 && q[LR_B_LDS]) ? (int32_t)q[LR_B_OFF] : -1;
            st.c_off = q[LR_C_LDS] ? (int32_t)q[LR_C_OFF] : -1;
            // rows
            const int64_t row_hi = q[LR_ROW_HI_LEN], row_lo = q[LR_ROW_LO];
            auto rows = [&](int64_t wa, int64_t wb, int64_t wc, int64_t len) -> uint32_t {
                std::vector<uint32_t> ent((size_t)len * 2);
                for (int64_t j = 0; j < len; ++j) {
                    const int64_t a = p->tables[wa + j], b = wb >= 0 ? p->tables[wb + j] : 0, c = p->tables[wc + j];
                    if (kind == 1) {
                        if (a >= 65536 || b >= 65536 || c >= (1ll << 31)) ok = false;
                        ent[2 * j] = (uint32_t)a | ((uint32_t)b << 16);
                    } else {
                        if (a >= (1ll << 31) || c >= 65536) ok = false;
                        ent[2 * j] = (uint32_t)a;
                    }
                    ent[2 * j + 1] = (uint32_t)c;
                }
                return put(ent.data(), ent.size() * 4);
            };
            st.t_row_hi = rows(q[LR_ROWA_HI], kind == 1 ? q[LR_ROWB_HI] : -1, q[LR_ROWC_HI], row_hi);
            st.t_row_lo = rows(q[LR_ROWA_LO], kind == 1 ? q[LR_ROWB_LO] : -1, q[LR_ROWC_LO], row_lo);
            {
                const int64_t k_hi = q[LR_K_HI_LEN], k_lo = q[LR_K_LO];
                std::vector<uint32_t> ent((size_t)(k_hi * k_lo));
                for (int64_t kh = 0; kh < k_hi; ++kh)
                    for (int64_t kl = 0; kl < k_lo; ++kl) {
                        const int64_t ka = p->tables[q[LR_KA_HI] + kh] + p->tables[q[LR_KA_LO] + kl];
                        if (kind == 1) {
                            const int64_t kb = p->tables[q[LR_KB_HI] + kh] + p->tables[q[LR_KB_LO] + kl];
                            if (ka >= 65536 || kb >= 65536) ok = false;
                            ent[(size_t)(kh * k_lo + kl)] = (uint32_t)ka | ((uint32_t)kb << 16);
                        } else {
                            if (ka >= (1ll << 31)) ok = false;
                            ent[(size_t)(kh * k_lo + kl)] = (uint32_t)ka;
                        }
                    }
                st.t_k = put(ent.data(), ent.size() * 4);
            }
            if (kind == 1) {
                std::vector<uint32_t> ent((size_t)q[LR_N]);
                for (int64_t j = 0; j < q[LR_N]; ++j) {
                    const int64_t nb = p->tables[q[LR_NB] + j], nc = p->tables[q[LR_NC] + j];
                    if (nb >= 65536 || nc >= 65536) ok = false;
                    ent[(size_t)j] = (uint32_t)nb | ((uint32_t)nc << 16);
                }
                st.t_n = put(ent.data(), ent.size() * 4);
            }
            // the global side, exactly as the ordinary step addresses it
            if (gop >= 0) {
                const StepArgs& a = e->args[main];
                if (gop == 0) { st.gptr = (const char*)a.A; st.gsoff = a.soffA; st.gz = a.zA; st.gzs = (int32_t)a.zsA; st.gzq = a.zqA; }
                else if (gop == 1) { st.gptr = (const char*)a.B; st.gsoff = a.soffB; st.gz = a.zB; st.gzs = (int32_t)a.zsB; st.gzq = a.zqB; }
                else { st.gptr = (const char*)a.C; st.gsoff = a.soffC; st.gz = a.zC; st.gzs = (int32_t)a.zsC; st.gzq = 0; }
            }
            // matrix cores: complex64 GEMM-like pair steps with enough rows and columns to fill a good part of a
            // 32 x 16 tile (B must not depend on the row: no batch index)
            st.mfma = 0;
            if (kind == 1 && p->dtype == CTG_C64 && q[LR_N] >= 8 && q[LR_R] >= 16 && !no_mfma) {
                bool gemm = true;
                for (int64_t j = 0; j < row_hi && gemm; ++j) gemm = p->tables[q[LR_ROWB_HI] + j] == 0;
                for (int64_t j = 0; j < row_lo && gemm; ++j) gemm = p->tables[q[LR_ROWB_LO] + j] == 0;
                st.mfma = gemm ? 1 : 0;
                if (gemm) {
                    // which way the result is stored: the lanes of a store run along the index group that is
                    // denser in C (1: columns on the lanes, 2: rows on the lanes)
                    auto stride = [&](int64_t w, int64_t len) -> int64_t {
                        return len > 1 ? std::llabs(p->tables[w + 1] - p->tables[w]) : INT64_MAX;
                    };
                    const int64_t srow = stride(q[LR_ROWC_LO], row_lo), scol = stride(q[LR_NC], q[LR_N]);
                    if (srow <= scol) st.mfma = 2;
                }
            }
            // wave assignment
            const int tn = q[LR_N] >= 3 ? 4 : (int)q[LR_N];
            int64_t items = kind == 1 ? q[LR_R] * ((q[LR_N] + tn - 1) / tn) : q[LR_R];
            const int64_t tasks = ((q[LR_R] + 31) / 32) * ((q[LR_N] + 15) / 16);
            if (st.mfma) items = tasks <= 1 ? 1 : 1 << 20;
            // (loads: one wave each, so that the gathers of a component's leaves are all in flight together)
            if (kind == 0 && items <= 4096) items = std::min<int64_t>(items, 128);
            st.wave = -1;
            st.rot = 0;
            if (st.mfma && tasks > 1) {
                int64_t& rot = mfma_rot[q[LR_PHASE]];
                st.rot = (int32_t)(rot % (LDS_RUN_THREADS / 64));
                rot += tasks;
            }
            if (items <= 128) {
                std::vector<int64_t>& load = wave_load[q[LR_PHASE]];
                if (load.empty()) load.assign(LDS_RUN_THREADS / 64, 0);
                const int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                load[(size_t)w] += 64 + items * q[LR_K] * tn;
                st.wave = w;
            }
        }
        if (!ok) continue;
        // table offsets are relative to the blob: records first
        const size_t tab_at = (rec_bytes + 15) / 16 * 16;
        for (LdsStepDev& st : recs) {
            st.t_row_hi += (uint32_t)tab_at;
            st.t_row_lo += (uint32_t)tab_at;
            st.t_k += (uint32_t)tab_at;
            if (st.kind == 1) st.t_n += (uint32_t)tab_at;
        }
        const size_t blob_bytes = (tab_at + tabs.size() + 15) / 16 * 16;
        pk.blob.assign(blob_bytes, 0);
        memcpy(pk.blob.data(), recs.data(), rec_bytes);
        memcpy(pk.blob.data() + tab_at, tabs.data(), tabs.size());
        pk.data_off = (uint32_t)blob_bytes;
        const int64_t total = (int64_t)blob_bytes + elems * isz;
        if (total > LDS_RUN_MAX_BYTES) continue;
        pk.lds_bytes = (int)((total + 15) / 16 * 16);
        packed.push_back(std::move(pk));
    }
    if (packed.empty()) return CTG_OK;
    // group-shared components first, per-slice ones after
    std::stable_sort(packed.begin(), packed.end(), [](const Packed& a, const Packed& b) { return a.shared > b.shared; });
    size_t total = 0;
    for (const Packed& pk : packed) total += pk.blob.size();
    if (hipMalloc((void**)&e->d_lds_blob, total) != hipSuccess) {
        (void)hipGetLastError();
        e->d_lds_blob = nullptr;
        return lfail(CTG_E_NOMEM, "out of device memory for the LDS component blobs");
    }
    if (hipMalloc((void**)&e->d_lds_comps, packed.size() * sizeof(LdsCompDev)) != hipSuccess) {
        (void)hipGetLastError();
        e->d_lds_comps = nullptr;
        return lfail(CTG_E_NOMEM, "out of device memory for the LDS component table");
    }
    std::vector<char> all(total);
    std::vector<LdsCompDev> comps(packed.size());
    size_t at = 0;
    for (size_t i = 0; i < packed.size(); ++i) {
        const Packed& pk = packed[i];
        memcpy(all.data() + at, pk.blob.data(), pk.blob.size());
        comps[i] = LdsCompDev{e->d_lds_blob + at, (uint32_t)pk.blob.size(), pk.n_steps, pk.data_off, 0};
        at += pk.blob.size();
        const int cls = pk.shared ? 0 : 1;
        if (e->lds_count[cls] == 0) e->lds_first[cls] = (int)i;
        e->lds_count[cls] += 1;
        e->lds_bytes[cls] = std::max(e->lds_bytes[cls], pk.lds_bytes);
        for (int64_t s = 0; s < p->n_steps; ++s)
            if (p->steps[s * STEP_WORDS + W_LDS_COMP] == pk.cid + 1) e->lds_comp_of[s] = (int32_t)i;
    }
    if (hipMemcpy(e->d_lds_blob, all.data(), total, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_lds_comps, comps.data(), comps.size() * sizeof(LdsCompDev), hipMemcpyHostToDevice) != hipSuccess)
        return lfail(CTG_E_HIP, "copy of the LDS component blobs failed: %s", hipGetErrorString(hipGetLastError()));
    return CTG_OK;
}
