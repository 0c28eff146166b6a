// ctg_runtime.hip -- host side of libctg_hip.so: plan validation, device
// residency, the per-slice launch sequence and the C ABI of include/ctg_hip.h.
//
// One ctg_exec per GPU.  A slice is a fixed sequence of kernel launches
// (prologue + one kernel per plan step); the slice id lives on the device and
// is advanced by the prologue kernel, so the loop over slices needs no
// host<->device traffic and no host synchronisation.
#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ctg_exec_state.h"

using namespace ctg;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return fail(_e == hipErrorOutOfMemory ? CTG_E_NOMEM : CTG_E_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(_e));                                   \
    } while (0)

const int64_t kItemSize[4] = {4, 8, 8, 16};
const int64_t kScratchBytes = 64ll << 20;

int log2_exact(int64_t v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1ll << s) < v) ++s;
    return s;
}

}  // namespace

namespace {

// max value of a table range [off, off+len)
int64_t tab_max(const ctg_plan* p, int64_t off, int64_t len, int64_t* mn) {
    int64_t mx = INT64_MIN;
    *mn = INT64_MAX;
    for (int64_t i = 0; i < len; ++i) {
        mx = std::max(mx, p->tables[off + i]);
        *mn = std::min(*mn, p->tables[off + i]);
    }
    return mx;
}

bool tab_ok(const ctg_plan* p, int64_t off, int64_t len) {
    return off >= 0 && len >= 1 && off + len <= (int64_t)p->tables.size();
}

int64_t space_elems(const ctg_plan* p, int64_t space) {
    switch (space) {
        case SPACE_INPUTS: return p->inputs_elems;
        case SPACE_ARENA: return p->arena_elems;
        case SPACE_RESULT: return p->result_elems;
    }
    return -1;
}

// A fused stem pair: every table inside the blob, every address inside its buffer, the
// tile geometry one the kernel takes.
int validate_stem(const ctg_plan* p, int64_t s) {
    const int64_t* r = &p->steps[s * STEP_WORDS];
    const long long sl = (long long)s;
    if (p->dtype != CTG_C64) return fail(CTG_E_INVALID, "step %lld: fused stem pairs are complex64 only", sl);
    const int64_t d = r[W_STEM];
    if (!tab_ok(p, d, STEM_WORDS)) return fail(CTG_E_BOUNDS, "step %lld: stem descriptor outside the blob", sl);
    const int64_t* h = &p->tables[d];
    if (h[SW_MAGIC] != STEM_MAGIC) return fail(CTG_E_INVALID, "step %lld: bad stem descriptor", sl);
    const int64_t K1 = h[SW_K1], N1 = h[SW_N1], K2 = h[SW_K2], N2 = h[SW_N2], nr1 = h[SW_NR1],
                  rows2 = h[SW_ROWS2], n_tiles = h[SW_NTILES], g_lo = h[SW_GLO];
    StemArgs a{};
    a.K1 = (int)K1; a.N1 = (int)N1; a.K2 = (int)K2; a.N2 = (int)N2; a.nr1 = (int)nr1;
    a.rows2 = (int)rows2; a.ng2 = (int)h[SW_NG2]; a.ld2 = (int)h[SW_LD2];
    a.n_tiles = n_tiles;
    if (h[SW_ONE] != 0 && h[SW_ONE] != 1) return fail(CTG_E_INVALID, "step %lld: bad stem step count", sl);
    const bool one = h[SW_ONE] == 1;
    a.one = one ? 1 : 0;
    if (one) {
        // the first half alone: no second operand, no intermediate tile
        if (K1 < 1 || K1 > 128 || N1 < 1 || N1 > 128 || K2 != 0 || N2 != 0 || rows2 != 0 || h[SW_NG2] != 0 ||
            h[SW_LD2] != 0 || nr1 < 5 || nr1 > 9 || !stem2_supported(a))
            return fail(CTG_E_INVALID, "step %lld: single stem step shape the kernel does not take", sl);
    } else if (K1 < 1 || K1 > 128 || N1 < 1 || N1 > 128 || K2 < 1 || K2 > 128 || N2 < 1 || N2 > 128 || nr1 < 5 ||
               nr1 > 9 || rows2 < 1 || rows2 > (1 << 16) || (h[SW_TRI] != 1 && !stem2_supported(a)))
        return fail(CTG_E_INVALID, "step %lld: stem pair shape the kernel does not take", sl);
    if (n_tiles < 1 || g_lo < 1 || log2_exact(g_lo) < 0 || n_tiles % g_lo != 0 || log2_exact(n_tiles) < 0)
        return fail(CTG_E_INVALID, "step %lld: bad stem grid", sl);
    const int64_t rows1 = 1ll << nr1;
    const int64_t len[ST_COUNT] = {n_tiles / g_lo, g_lo, n_tiles / g_lo, g_lo, 8, 64, rows1 / 32, K1 / 16,
                                   K1 * N1, one ? 1 : K2 * N2, one ? 1 : rows1, one ? 1 : N1,
                                   one ? rows1 : rows2, one ? N1 : N2};
    int64_t mx[ST_COUNT], mn[ST_COUNT];
    for (int t = 0; t < ST_COUNT; ++t) {
        if (!tab_ok(p, h[SW_TABS + t], len[t]))
            return fail(CTG_E_BOUNDS, "step %lld: stem table outside the blob", sl);
        mx[t] = tab_max(p, h[SW_TABS + t], len[t], &mn[t]);
        if (mn[t] < 0) return fail(CTG_E_BOUNDS, "step %lld: negative stem offset", sl);
    }
    // the lane part of a gather address is a 32-bit byte offset in the kernel
    if (mx[ST_LANE_A] >= (1ll << 29))
        return fail(CTG_E_INVALID, "step %lld: stem lane offsets beyond 32 bits", sl);
    if (h[SW_VEC] != 0 && h[SW_VEC] != 1) return fail(CTG_E_INVALID, "step %lld: bad stem gather mode", sl);
    if (h[SW_VEC]) {
        // 16-byte gathers: slot pairs adjacent, every address even
        const int64_t* kj = &p->tables[h[SW_TABS + ST_KJ_A]];
        for (int q = 0; q < 4; ++q)
            if (kj[2 * q + 1] != kj[2 * q] + 1 || (kj[2 * q] & 1))
                return fail(CTG_E_INVALID, "step %lld: stem slots not paired for 16-byte gathers", sl);
        const int odd_tabs[5] = {ST_GA_HI, ST_GA_LO, ST_LANE_A, ST_RT_A, ST_CHUNK_A};
        for (int t : odd_tabs)
            for (int64_t i = 0; i < len[t]; ++i)
                if (p->tables[h[SW_TABS + t] + i] & 1)
                    return fail(CTG_E_INVALID, "step %lld: odd stem offset under 16-byte gathers", sl);
        if ((r[W_A_OFF] & 1) || r[W_A_LEAF] >= 0)
            return fail(CTG_E_INVALID, "step %lld: stem operand not aligned for 16-byte gathers", sl);
    }
    if (h[SW_TRI] != 0 && h[SW_TRI] != 1) return fail(CTG_E_INVALID, "step %lld: bad stem stage count", sl);
    const bool tri = h[SW_TRI] == 1;
    int64_t bm_top = 0;
    if (tri) {
        // a middle stage: its shape one the kernel is instantiated for, its tables inside the blob,
        // both intermediates inside the LDS region they share
        const int64_t KM = h[SW_KM], NM = h[SW_NM], rowsM = h[SW_ROWSM];
        if (one || KM < 16 || KM > 128 || NM < 16 || NM > 128 || rowsM < 32 || rowsM > (1 << 16) ||
            h[SW_LDM] != KM + 4 || rows1 * N1 != rowsM * KM || rowsM * NM != rows2 * K2)
            return fail(CTG_E_INVALID, "step %lld: three-step tile shape the kernel does not take", sl);
        a.tri = 1; a.KM = (int)KM; a.NM = (int)NM; a.rowsM = (int)rowsM; a.ngM = (int)h[SW_NGM]; a.ldM = (int)h[SW_LDM];
        a.vec = (int)h[SW_VEC];
        if (!stem3_supported(a))
            return fail(CTG_E_INVALID, "step %lld: three-step tile shape the kernel does not take", sl);
        const int64_t lenm[3] = {KM * NM, rowsM, NM};
        int64_t mxm[3], mnm[3];
        for (int t = 0; t < 3; ++t) {
            if (!tab_ok(p, h[SW_TABS_M + t], lenm[t]))
                return fail(CTG_E_BOUNDS, "step %lld: stem table outside the blob", sl);
            mxm[t] = tab_max(p, h[SW_TABS_M + t], lenm[t], &mnm[t]);
            if (mnm[t] < 0) return fail(CTG_E_BOUNDS, "step %lld: negative stem offset", sl);
        }
        bm_top = mxm[0];
        if (mx[ST_MID_ROW] + mx[ST_MID_COL] >= rowsM * (KM + 4) || mxm[1] + mxm[2] >= rows2 * (K2 + 4))
            return fail(CTG_E_BOUNDS, "step %lld: intermediate tile overflows its LDS", sl);
        const int64_t cap = space_elems(p, h[SW_BM_SPACE]);
        if (cap < 0) return fail(CTG_E_INVALID, "step %lld: bad space", sl);
        if (h[SW_BM_LEAF] < -1 || h[SW_BM_LEAF] > p->n_inputs) return fail(CTG_E_INVALID, "step %lld: bad leaf", sl);
        const int64_t hi = h[SW_BM_OFF] + (h[SW_BM_LEAF] >= 0 ? p->max_soff[h[SW_BM_LEAF]] : 0) + bm_top;
        if (h[SW_BM_OFF] < 0 || hi >= cap)
            return fail(CTG_E_BOUNDS, "step %lld: stem operand m reaches element %lld of a space of %lld", sl,
                        (long long)hi, (long long)cap);
    }
    if (!one && !tri && mx[ST_MID_ROW] + mx[ST_MID_COL] >= rows2 * (K2 + 4))
        return fail(CTG_E_BOUNDS, "step %lld: intermediate tile overflows its LDS", sl);
    struct Op { int64_t space, off, leaf, size, top; char name; };
    const Op ops[4] = {
        {r[W_A_SPACE], r[W_A_OFF], r[W_A_LEAF], r[W_A_SIZE],
         mx[ST_GA_HI] + mx[ST_GA_LO] + mx[ST_RT_A] + mx[ST_CHUNK_A] + mx[ST_KJ_A] + mx[ST_LANE_A], 'A'},
        {r[W_B_SPACE], r[W_B_OFF], r[W_B_LEAF], r[W_B_SIZE], mx[ST_B1_OFF], 'B'},
        {h[SW_B2_SPACE], h[SW_B2_OFF], h[SW_B2_LEAF], h[SW_B2_SIZE], mx[ST_B2_OFF], 'b'},
        {r[W_C_SPACE], r[W_C_OFF], r[W_C_LEAF], r[W_C_SIZE],
         mx[ST_GC_HI] + mx[ST_GC_LO] + mx[ST_OUT_ROW] + mx[ST_OUT_COL], 'C'},
    };
    for (const Op& op : ops) {
        if (one && op.name == 'b') continue;
        const int64_t cap = space_elems(p, op.space);
        if (cap < 0) return fail(CTG_E_INVALID, "step %lld: bad space", sl);
        if (op.leaf < -1 || op.leaf > p->n_inputs) return fail(CTG_E_INVALID, "step %lld: bad leaf", sl);
        if (op.name == 'C' && op.space == SPACE_INPUTS)
            return fail(CTG_E_INVALID, "step %lld: writes into the inputs space", sl);
        const int64_t hi = op.off + (op.leaf >= 0 ? p->max_soff[op.leaf] : 0) + op.top;
        if (op.off < 0 || hi >= cap)
            return fail(CTG_E_BOUNDS, "step %lld: stem operand %c reaches element %lld of a space of %lld",
                        sl, op.name, (long long)hi, (long long)cap);
    }
    return CTG_OK;
}

int validate_plan(ctg_plan* p) {
    if (p->dtype < 0 || p->dtype > 3) return fail(CTG_E_INVALID, "bad dtype %d", p->dtype);
    if (p->n_inputs < 1) return fail(CTG_E_INVALID, "plan needs at least one input");
    for (int64_t i = 0; i < p->n_inputs; ++i) {
        if (p->input_sizes[i] < 1 || p->input_offsets[i] < 0 ||
            p->input_offsets[i] + p->input_sizes[i] > p->inputs_elems)
            return fail(CTG_E_BOUNDS, "input %lld does not fit the inputs space", (long long)i);
    }
    // slices
    p->nslices = 1;
    p->max_soff.assign(p->n_inputs + 1, 0);
    for (int64_t j = 0; j < p->n_sliced; ++j) {
        const int64_t d = p->slice_sizes[j], f = p->slice_fixed[j];
        if (d < 1) return fail(CTG_E_INVALID, "bad slice size");
        if (f >= 0 && d != 1) return fail(CTG_E_INVALID, "projected index must have size 1");
        // saturate: trees narrowed for tests can have more than 2^63 slices
        p->nslices = (p->nslices > INT64_MAX / d) ? INT64_MAX : p->nslices * d;
        for (int64_t l = 0; l <= p->n_inputs; ++l) {
            const int64_t st = p->slice_strides[l * p->n_sliced + j];
            if (st < 0) return fail(CTG_E_INVALID, "negative slice stride");
            p->max_soff[l] += st * (f >= 0 ? f : d - 1);
        }
    }
    // slice groups: flags 0 / 1 on sliced indices that are not projected
    bool any_group = false;
    for (int64_t j = 0; j < p->n_sliced; ++j) {
        const int64_t g = p->slice_group[j];
        if (g != 0 && g != 1) return fail(CTG_E_INVALID, "bad slice group flag");
        if (g == 1 && p->slice_fixed[j] >= 0) return fail(CTG_E_INVALID, "a projected index cannot be a group index");
        any_group = any_group || g == 1;
    }
    p->has_groups = false;
    // steps
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        const int64_t kind = r[W_KIND];
        if (kind < 0 || kind > 3) return fail(CTG_E_INVALID, "step %lld: bad kind", (long long)s);
        if (r[W_INVARIANT] < 0 || r[W_INVARIANT] > 2 || (r[W_INVARIANT] == 2 && !any_group))
            return fail(CTG_E_INVALID, "step %lld: bad sharing class", (long long)s);
        p->has_groups = p->has_groups || (r[W_INVARIANT] == 2 && kind != KIND_ACCUM);
        if (kind == KIND_STEM2) {
            const int rc = validate_stem(p, s);
            if (rc != CTG_OK) return rc;
            continue;
        }
        if (r[W_KERNEL] < 0 || r[W_KERNEL] > 1)
            return fail(CTG_E_INVALID, "step %lld: bad kernel", (long long)s);
        if (r[W_KERNEL] == KERNEL_MFMA && kind != KIND_PAIR)
            return fail(CTG_E_INVALID, "step %lld: the matrix-core kernels execute pair steps only",
                        (long long)s);
        const int64_t R = r[W_R], Bt = r[W_BT], K = r[W_K], N = r[W_N];
        const int64_t row_lo = r[W_ROW_LO], row_hi = r[W_ROW_HI_LEN];
        const int64_t k_lo = r[W_K_LO], k_hi = r[W_K_HI_LEN];
        if (R < 1 || Bt < 1 || K < 1 || N < 1 || row_lo < 1 || row_hi < 1 || k_lo < 1 || k_hi < 1)
            return fail(CTG_E_INVALID, "step %lld: non-positive extent", (long long)s);
        if (row_lo * row_hi != R) return fail(CTG_E_INVALID, "step %lld: row split != R", (long long)s);
        if (k_lo * k_hi != K) return fail(CTG_E_INVALID, "step %lld: k split != K", (long long)s);
        if (r[W_KERNEL] == KERNEL_MFMA && Bt > 65535)
            return fail(CTG_E_INVALID, "step %lld: MFMA batch too large", (long long)s);

        struct Op { int space_w, off_w, leaf_w, size_w; int64_t rhi, rlo, khi, klo, n, b; bool used; };
        const bool pair = kind == KIND_PAIR, mfma = r[W_KERNEL] == KERNEL_MFMA;
        Op ops[3] = {
            {W_A_SPACE, W_A_OFF, W_A_LEAF, W_A_SIZE, r[W_ROWA_HI], r[W_ROWA_LO],
             kind != KIND_ACCUM ? r[W_KA_HI] : -1, kind != KIND_ACCUM ? r[W_KA] : -1, -1,
             mfma ? r[W_BA] : -1, true},
            {W_B_SPACE, W_B_OFF, W_B_LEAF, W_B_SIZE, mfma ? -1 : r[W_ROWB_HI], mfma ? -1 : r[W_ROWB_LO],
             r[W_KB_HI], r[W_KB], r[W_NB], mfma ? r[W_BB] : -1, pair},
            {W_C_SPACE, W_C_OFF, W_C_LEAF, W_C_SIZE, r[W_ROWC_HI], r[W_ROWC_LO], -1, -1,
             pair ? r[W_NC] : -1, mfma ? r[W_BC] : -1, true},
        };
        for (int o = 0; o < 3; ++o) {
            const Op& op = ops[o];
            if (!op.used) continue;
            const int64_t space = r[op.space_w], off = r[op.off_w], leaf = r[op.leaf_w];
            const int64_t cap = space_elems(p, space);
            if (cap < 0) return fail(CTG_E_INVALID, "step %lld: bad space", (long long)s);
            if (leaf < -1 || leaf > p->n_inputs)
                return fail(CTG_E_INVALID, "step %lld: bad leaf", (long long)s);
            if (o == 2 && space == SPACE_INPUTS)
                return fail(CTG_E_INVALID, "step %lld: writes into the inputs space", (long long)s);
            int64_t lo_addr = off, hi_addr = off + (leaf >= 0 ? p->max_soff[leaf] : 0);
            struct { int64_t t, len; } tabs[6] = {
                {op.rhi, row_hi}, {op.rlo, row_lo}, {op.khi, k_hi}, {op.klo, k_lo},
                {op.n, N}, {op.b, Bt}};
            for (auto& t : tabs) {
                if (t.t < 0) continue;
                if (!tab_ok(p, t.t, t.len))
                    return fail(CTG_E_BOUNDS, "step %lld: table outside the blob", (long long)s);
                int64_t mn;
                const int64_t mx = tab_max(p, t.t, t.len, &mn);
                hi_addr += mx;
                lo_addr += mn;
            }
            if (lo_addr < 0 || hi_addr >= cap)
                return fail(CTG_E_BOUNDS,
                            "step %lld: operand %c addresses [%lld, %lld] outside its space of %lld elements",
                            (long long)s, "ABC"[o], (long long)lo_addr, (long long)hi_addr, (long long)cap);
        }
    }
    if (p->has_groups) {
        // (group indices of extent 1 only: every "group" is one slice -- the plan shares nothing, and
        // ctg_exec_run_slices / ctg_exec_run_share must not hand the work to each other for ever)
        int64_t gsize = 1;
        for (int64_t j = 0; j < p->n_sliced; ++j)
            if (p->slice_group[j] == 1 && p->slice_fixed[j] < 0) gsize *= p->slice_sizes[j];
        if (gsize <= 1) p->has_groups = false;
    }
    if (p->has_groups) {
        // What a per-slice step reads of a step that its group shares must survive the group: no
        // per-slice step may write into that range of the arena (element ranges of the records; the
        // planner keeps such results out of the recycled part, cotengra_amd/plan.py: kept_for_group).
        struct Range { int64_t lo, hi; };
        auto out_of = [&](int64_t s) {
            const int64_t* r = &p->steps[s * STEP_WORDS];
            return Range{r[W_C_OFF], r[W_C_OFF] + r[W_C_SIZE]};
        };
        auto reads = [&](int64_t s, const Range& x) {
            const int64_t* r = &p->steps[s * STEP_WORDS];
            auto hit = [&](int64_t space, int64_t off, int64_t size) {
                return space == SPACE_ARENA && off < x.hi && x.lo < off + size;
            };
            if (hit(r[W_A_SPACE], r[W_A_OFF], r[W_A_SIZE]) || hit(r[W_B_SPACE], r[W_B_OFF], r[W_B_SIZE])) return true;
            if (r[W_KIND] == KIND_STEM2) {
                const int64_t* h = &p->tables[r[W_STEM]];
                if (h[SW_ONE] != 1 && hit(h[SW_B2_SPACE], h[SW_B2_OFF], h[SW_B2_SIZE])) return true;
                if (h[SW_TRI] == 1 && hit(h[SW_BM_SPACE], h[SW_BM_OFF], h[SW_BM_SIZE])) return true;
            }
            return false;
        };
        auto shared = [&](int64_t s) { return p->steps[s * STEP_WORDS + W_INVARIANT] == 2; };
        for (int64_t g = 0; g < p->n_steps; ++g) {
            const int64_t* rg = &p->steps[g * STEP_WORDS];
            if (!shared(g) || rg[W_KIND] == KIND_ACCUM || rg[W_C_SPACE] != SPACE_ARENA) continue;
            const Range x = out_of(g);
            // (the value lives until the next step that writes into its range: ranges are recycled)
            int64_t until = p->n_steps;
            for (int64_t s = g + 1; s < p->n_steps && until == p->n_steps; ++s) {
                const int64_t* r = &p->steps[s * STEP_WORDS];
                if (r[W_KIND] == KIND_ACCUM || r[W_C_SPACE] != SPACE_ARENA) continue;
                const Range y = out_of(s);
                if (y.lo < x.hi && x.lo < y.hi) until = s;
            }
            bool kept = false;
            for (int64_t s = g + 1; s < until && !kept; ++s)
                kept = p->steps[s * STEP_WORDS + W_INVARIANT] == 0 && reads(s, x);
            if (!kept) continue;
            for (int64_t s = 0; s < p->n_steps; ++s) {
                const int64_t* r = &p->steps[s * STEP_WORDS];
                if (s == g || r[W_INVARIANT] == 1 || r[W_KIND] == KIND_ACCUM || r[W_C_SPACE] != SPACE_ARENA) continue;
                const Range y = out_of(s);
                if (y.lo < x.hi && x.lo < y.hi)
                    return fail(CTG_E_INVALID, "step %lld writes into what step %lld keeps for its slice group",
                                (long long)s, (long long)g);
            }
        }
    }
    return CTG_OK;
}

void* space_ptr(const ctg_exec* e, int64_t space) {
    switch (space) {
        case SPACE_INPUTS: return e->d_inputs;
        case SPACE_ARENA: return e->d_arena;
        case SPACE_RESULT: return e->d_result;
    }
    return nullptr;
}

void resolve_args(ctg_exec* e) {
    const ctg_plan* p = e->plan;
    const int64_t isz = kItemSize[p->dtype];
    e->args.resize(p->n_steps);
    e->stem_args.resize(p->n_steps);
    // (hints depend on the plan only: kept when the arguments are re-resolved,
    // e.g. by ctg_exec_set_strip_exponent on a live executor)
    if ((int64_t)e->hints.size() != p->n_steps)
        e->hints.assign(p->n_steps, MfmaHints{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0});
    const int64_t* T = e->d_tables;
    // arena ranges written by slice-invariant steps: never recycled, the same in every slice
    std::vector<std::pair<int64_t, int64_t>> persistent;
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        if (r[W_INVARIANT] == 1 && r[W_KIND] != KIND_ACCUM && r[W_C_SPACE] == SPACE_ARENA)
            persistent.emplace_back(r[W_C_OFF], r[W_C_OFF] + r[W_C_SIZE]);
    }
    auto per_slice = [&](int64_t space, int64_t off) -> int64_t {
        if (space != SPACE_ARENA) return 0;
        for (const auto& iv : persistent)
            if (off >= iv.first && off < iv.second) return 0;
        return p->arena_elems;
    };
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        StepArgs& a = e->args[s];
        memset(&a, 0, sizeof(a));
        auto ptr = [&](int sw, int ow) -> char* {
            if (r[sw] < 0) return nullptr;
            return (char*)space_ptr(e, r[sw]) + r[ow] * isz;
        };
        auto soff = [&](int lw) -> const int64_t* {
            return r[lw] >= 0 ? e->d_soff + r[lw] : e->d_zero;
        };
        auto tab = [&](int w) -> const int64_t* { return r[w] >= 0 ? T + r[w] : e->d_zero; };
        a.A = ptr(W_A_SPACE, W_A_OFF);
        a.B = ptr(W_B_SPACE, W_B_OFF);
        a.C = ptr(W_C_SPACE, W_C_OFF);
        a.soffA = soff(W_A_LEAF);
        a.soffB = soff(W_B_LEAF);
        a.soffC = soff(W_C_LEAF);
        a.R = r[W_R];
        a.Bt = r[W_BT];
        a.K = r[W_K];
        a.N = r[W_N];
        a.row_lo = r[W_ROW_LO];
        a.row_lo_shift = log2_exact(a.row_lo);
        // (rows / k's of a step are numbered from 0 to R - 1 / K - 1; kernels may probe one
        // tile beyond: the reciprocal is exact for any argument below 2^32)
        auto magic = [](int64_t d, int64_t range) -> uint64_t {
            if (d < 2 || (d & (d - 1)) == 0 || range + 4096 >= (1ll << 32)) return 0;
            return (uint64_t)(~0ull / (uint64_t)d) + 1;   // floor(2^64 / d) + 1 (d is not a power of two)
        };
        a.row_lo_magic = magic(a.row_lo, r[W_R]);
        a.rowA = RowTab{tab(W_ROWA_HI), tab(W_ROWA_LO)};
        a.rowB = RowTab{tab(W_ROWB_HI), tab(W_ROWB_LO)};
        a.rowC = RowTab{tab(W_ROWC_HI), tab(W_ROWC_LO)};
        a.k_lo = r[W_K_LO];
        a.k_hi_len = r[W_K_HI_LEN];
        a.k_lo_shift = log2_exact(a.k_lo);
        a.k_lo_magic = magic(a.k_lo, r[W_K]);
        a.kA = RowTab{tab(W_KA_HI), tab(W_KA)};
        a.kB = RowTab{tab(W_KB_HI), tab(W_KB)};
        a.nB = tab(W_NB);
        a.nC = tab(W_NC);
        a.bA = tab(W_BA);
        a.bB = tab(W_BB);
        a.bC = tab(W_BC);
        // slice batching: per-slice operands step through the arena replicas and the
        // rows of d_soff; inputs, slice-invariant intermediates and the result do not
        a.nz = 1;
        a.z0 = 0;
        a.scratch_total = e->scratch_total;
        a.zA = per_slice(r[W_A_SPACE], r[W_A_OFF]);
        a.zB = per_slice(r[W_B_SPACE], r[W_B_OFF]);
        a.zC = per_slice(r[W_C_SPACE], r[W_C_OFF]);
        a.zsA = r[W_A_LEAF] >= 0 ? p->n_inputs + 1 : 0;
        a.zsB = r[W_B_LEAF] >= 0 ? p->n_inputs + 1 : 0;
        a.zsC = r[W_C_LEAF] >= 0 ? p->n_inputs + 1 : 0;
        if (e->group_d > 1) {
            auto shared = [&](int64_t st) {
                return st >= 0 && st < p->n_steps && p->steps[st * STEP_WORDS + W_INVARIANT] == 2 &&
                       p->steps[st * STEP_WORDS + W_KIND] != KIND_ACCUM;
            };
            const int64_t d = e->group_d;
            if (shared(s)) {
                // launched once per group: group g of the launch is slice-in-batch g * d
                a.zA *= d; a.zB *= d; a.zC *= d;
                a.zsA *= d; a.zsB *= d; a.zsC *= d;
            } else if (r[W_KIND] == KIND_PAIR || r[W_KIND] == KIND_SINGLE) {
                // Which step wrote an operand is read off the arena, not off W_A_PROD / W_B_PROD (those
                // name PAIR / STEM2 producers only -- they drive the strip_exponent factors): the latest
                // earlier step whose output starts where the operand starts.  A shared leaf-preprocessing
                // step (KIND_SINGLE) feeding a per-slice step is found this way too.
                auto writer = [&](int sw, int ow) -> int64_t {
                    if (r[sw] != SPACE_ARENA) return -1;
                    for (int64_t t = s - 1; t >= 0; --t) {
                        const int64_t* w = &p->steps[t * STEP_WORDS];
                        if (w[W_KIND] != KIND_ACCUM && w[W_C_SPACE] == SPACE_ARENA && w[W_C_OFF] == r[ow]) return t;
                    }
                    return -1;
                };
                if (shared(writer(W_A_SPACE, W_A_OFF))) a.zqA = (int32_t)d;
                if (r[W_KIND] == KIND_PAIR && shared(writer(W_B_SPACE, W_B_OFF))) a.zqB = (int32_t)d;
            }
        }
        a.facA = a.facB = nullptr;
        a.check_zero = e->check_zero;
        a.bf3 = e->stem_bf16x3;
        if (e->strip && r[W_KIND] == KIND_PAIR) {
            auto fac = [&](int w) -> const double* {
                return (r[w] >= 0 && r[w] < p->n_steps) ? e->d_fac + r[w] : e->d_fac + p->n_steps;
            };
            a.facA = fac(W_A_PROD);
            a.facB = fac(W_B_PROD);
        }
        if (r[W_KIND] == KIND_STEM2) {
            const int64_t* h = &p->tables[r[W_STEM]];
            StemArgs& q = e->stem_args[s];
            memset(&q, 0, sizeof(q));
            q.A = a.A;
            q.B1 = a.B;
            q.C = a.C;
            q.one = h[SW_ONE] == 1 ? 1 : 0;
            q.B2 = q.one ? nullptr : (char*)space_ptr(e, h[SW_B2_SPACE]) + h[SW_B2_OFF] * isz;
            q.soffA = a.soffA;
            q.soffB1 = a.soffB;
            q.soffC = a.soffC;
            q.soffB2 = h[SW_B2_LEAF] >= 0 ? e->d_soff + h[SW_B2_LEAF] : e->d_zero;
            q.K1 = (int)h[SW_K1]; q.N1 = (int)h[SW_N1]; q.K2 = (int)h[SW_K2]; q.N2 = (int)h[SW_N2];
            q.nr1 = (int)h[SW_NR1]; q.rows2 = (int)h[SW_ROWS2]; q.ng2 = (int)h[SW_NG2]; q.ld2 = (int)h[SW_LD2];
            q.n_tiles = h[SW_NTILES];
            q.g_lo = h[SW_GLO];
            q.g_lo_shift = log2_exact(q.g_lo);
            q.check_zero = e->check_zero;
            q.vec = (int)h[SW_VEC];
            q.bf3 = e->stem_bf16x3;
            q.a_elems = r[W_A_SIZE];
            q.c_elems = r[W_C_SIZE];
            const int64_t** tabs[ST_COUNT] = {&q.gA_hi, &q.gA_lo, &q.gC_hi, &q.gC_lo, &q.kj_a, &q.lane_a, &q.rt_a,
                                              &q.chunk_a, &q.b1_off, &q.b2_off, &q.mid_row, &q.mid_col,
                                              &q.out_row, &q.out_col};
            for (int t = 0; t < ST_COUNT; ++t) *tabs[t] = T + h[SW_TABS + t];
            q.nz = 1;
            q.z0 = 0;
            q.zA = a.zA; q.zB1 = a.zB; q.zC = a.zC;
            q.zB2 = per_slice(h[SW_B2_SPACE], h[SW_B2_OFF]);
            q.zsA = a.zsA; q.zsB1 = a.zsB; q.zsC = a.zsC;
            q.zsB2 = h[SW_B2_LEAF] >= 0 ? p->n_inputs + 1 : 0;
            if (e->strip) {
                auto fac = [&](int64_t w) -> const double* {
                    return (w >= 0 && w < p->n_steps) ? e->d_fac + w : e->d_fac + p->n_steps;
                };
                q.facA = fac(r[W_A_PROD]);
                q.facB1 = fac(r[W_B_PROD]);
                q.facB2 = q.one ? e->d_fac + p->n_steps : fac(h[SW_B2_PROD]);   // (one step: the constant 1)
                q.facBM = h[SW_TRI] == 1 ? fac(h[SW_BM_PROD]) : e->d_fac + p->n_steps;
            }
            if (h[SW_TRI] == 1) {
                q.tri = 1;
                q.KM = (int)h[SW_KM]; q.NM = (int)h[SW_NM]; q.rowsM = (int)h[SW_ROWSM]; q.ngM = (int)h[SW_NGM];
                q.ldM = (int)h[SW_LDM];
                q.BM = (char*)space_ptr(e, h[SW_BM_SPACE]) + h[SW_BM_OFF] * isz;
                q.soffBM = h[SW_BM_LEAF] >= 0 ? e->d_soff + h[SW_BM_LEAF] : e->d_zero;
                q.bm_off = T + h[SW_TABS_M];
                q.mid2_row = T + h[SW_TABS_M + 1];
                q.mid2_col = T + h[SW_TABS_M + 2];
                q.zBM = per_slice(h[SW_BM_SPACE], h[SW_BM_OFF]);
                q.zsBM = h[SW_BM_LEAF] >= 0 ? p->n_inputs + 1 : 0;
            }
        }
    }
}

// ---- MFMA gather order tables -------------------------------------------- //

// offset of entry i of a two-level table (host copy)
int64_t tab2(const ctg_plan* p, int64_t hi_w, int64_t lo_w, int64_t lo_size, int64_t i) {
    return p->tables[hi_w + i / lo_size] + p->tables[lo_w + i % lo_size];
}

// true if lo[t*B + r] - lo[t*B] == lo[r] for every tile t: tile-local offsets
// do not depend on the tile, so the order found on tile 0 holds everywhere
bool tile_additive(const ctg_plan* p, int64_t lo_w, int64_t lo_size, int64_t total, int B) {
    if (total % B) return false;
    if (lo_size % B) return false;
    for (int64_t t = 0; t < lo_size / B; ++t)
        for (int r = 0; r < B; ++r)
            if (p->tables[lo_w + t * B + r] - p->tables[lo_w + t * B] != p->tables[lo_w + r])
                return false;
    return true;
}

bool all_even(const ctg_plan* p, int64_t w, int64_t len, int64_t stride = 1) {
    for (int64_t i = 0; i < len; i += stride)
        if (p->tables[w + i] & 1) return false;
    return true;
}

// Build the order tables of one MFMA step; appends to `blob`, returns offsets.
void build_mfma_order(const ctg_plan* p, const int64_t* r, int bn, int BM,
                      std::vector<uint16_t>& blob, size_t* offA, size_t* offB, int* vecA,
                      int TB = 256) {
    const int BK = MFMA_BK;
    const int T = BM == 32 ? 64 : 256;  // threads sharing one tile gather
    const int64_t R = r[W_R], K = r[W_K], N = r[W_N];
    const int64_t BIG = INT64_MAX / 4;
    // ---- A tile ----
    {
        const int n_el = BM * BK, per_t = n_el / T;
        std::vector<int64_t> off(n_el);
        std::vector<int> idx(n_el);
        for (int rr = 0; rr < BM; ++rr)
            for (int c = 0; c < BK; ++c) {
                const int i = rr * BK + c;
                idx[i] = i;
                off[i] = (rr < R && c < K)
                             ? tab2(p, r[W_ROWA_HI], r[W_ROWA_LO], r[W_ROW_LO], rr) +
                                   tab2(p, r[W_KA_HI], r[W_KA], r[W_K_LO], c)
                             : BIG + i;
            }
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return off[x] < off[y]; });
        // streaming kernel with a short contraction (K < 16, one chunk): only the
        // 32*K real elements are gathered (they sort first), all lanes stay busy
        const bool short_k = BM == 32 && K < BK;
        int n_valid = 0;
        for (int i = 0; i < n_el; ++i) n_valid += off[i] < BIG ? 1 : 0;
        bool vec = (R % BM == 0) && (short_k ? (K % 2 == 0) : (K % BK == 0));
        for (int q = 0; vec && q < n_valid / 2; ++q) {
            const int64_t o0 = off[idx[2 * q]], o1 = off[idx[2 * q + 1]];
            if (o1 != o0 + 1 || (o0 & 1)) vec = false;
        }
        // the pairing must hold for every tile / k-step and keep 16-byte alignment
        vec = vec && tile_additive(p, r[W_ROWA_LO], r[W_ROW_LO], R, BM) &&
              (short_k || tile_additive(p, r[W_KA], r[W_K_LO], K, BK));
        vec = vec && all_even(p, r[W_ROWA_HI], r[W_ROW_HI_LEN]) &&
              all_even(p, r[W_ROWA_LO], r[W_ROW_LO], BM) && all_even(p, r[W_KA_HI], r[W_K_HI_LEN]) &&
              all_even(p, r[W_KA], r[W_K_LO], BK) && all_even(p, r[W_BA], r[W_BT]) &&
              (r[W_A_OFF] % 2 == 0);
        if (vec && r[W_A_LEAF] >= 0)
            for (int64_t j = 0; j < p->n_sliced; ++j)
                if (p->slice_strides[r[W_A_LEAF] * p->n_sliced + j] & 1) vec = false;
        *vecA = vec ? 1 : 0;
        *offA = blob.size();
        blob.resize(blob.size() + n_el);
        uint16_t* out = blob.data() + *offA;
        auto pack = [&](int i) { return (uint16_t)(((i / BK) << 4) | (i % BK)); };
        const uint16_t kSkip = 0x8000;  // entry the kernel must not gather (short-K padding)
        for (int tid = 0; tid < T; ++tid) {
            if (vec) {
                for (int jj = 0; jj < per_t / 2; ++jj) {
                    const int q = jj * T + tid;
                    const bool live = 2 * q + 1 < n_valid;
                    out[tid * per_t + 2 * jj] = live ? pack(idx[2 * q]) : kSkip;
                    out[tid * per_t + 2 * jj + 1] = live ? pack(idx[2 * q + 1]) : kSkip;
                }
            } else {
                for (int j = 0; j < per_t; ++j) out[tid * per_t + j] = pack(idx[j * T + tid]);
            }
        }
    }
    // ---- B tile ----
    {
        const int n_el = BK * bn, per_t = (n_el + TB - 1) / TB;   // TB threads share the B tile
        std::vector<int64_t> off(n_el);
        std::vector<int> idx(n_el);
        for (int n = 0; n < bn; ++n)
            for (int c = 0; c < BK; ++c) {
                const int i = n * BK + c;
                idx[i] = i;
                off[i] = (n < N && c < K)
                             ? p->tables[r[W_NB] + n] + tab2(p, r[W_KB_HI], r[W_KB], r[W_K_LO], c)
                             : BIG + i;
            }
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return off[x] < off[y]; });
        *offB = blob.size();
        blob.resize(blob.size() + TB * per_t, 0);
        uint16_t* out = blob.data() + *offB;
        for (int tid = 0; tid < TB; ++tid)
            for (int j = 0; j < per_t; ++j) {
                const int e = j * TB + tid;
                out[tid * per_t + j] =
                    e < n_el ? (uint16_t)(((idx[e] / BK) << 4) | (idx[e] % BK)) : (uint16_t)0;
            }
    }
}

// flat-table variant of tile_additive
bool flat_additive(const ctg_plan* p, int64_t w, int64_t total, int B) {
    if (total % B) return false;
    for (int64_t t = 0; t < total / B; ++t)
        for (int r = 0; r < B; ++r)
            if (p->tables[w + t * B + r] - p->tables[w + t * B] != p->tables[w + r]) return false;
    return true;
}

// conditions of pair_mfma_fast_kernel: full tiles, tile-additive tables, and
// tile-local offsets that fit 32 bits
bool mfma_fast_ok(const ctg_plan* p, const int64_t* r, int bn) {
    const int BM = MFMA_BM, BK = MFMA_BK;
    const int64_t R = r[W_R], K = r[W_K], N = r[W_N];
    if (R % BM || K % BK || N % bn) return false;
    if (r[W_K_LO] & (r[W_K_LO] - 1)) return false;  // the kernel splits k with shift / mask
    if (!tile_additive(p, r[W_ROWA_LO], r[W_ROW_LO], R, BM)) return false;
    if (!tile_additive(p, r[W_ROWC_LO], r[W_ROW_LO], R, BM)) return false;
    if (!tile_additive(p, r[W_KA], r[W_K_LO], K, BK)) return false;
    if (!tile_additive(p, r[W_KB], r[W_K_LO], K, BK)) return false;
    if (!flat_additive(p, r[W_NB], N, bn) || !flat_additive(p, r[W_NC], N, bn)) return false;
    auto mx = [&](int64_t w, int n) {
        int64_t m = 0;
        for (int i = 0; i < n; ++i) {
            if (p->tables[w + i] < 0) return (int64_t)INT64_MAX / 4;
            m = std::max(m, p->tables[w + i]);
        }
        return m;
    };
    const int64_t lim = (int64_t)1 << 31;
    if (mx(r[W_ROWA_LO], BM) + mx(r[W_KA], BK) >= lim) return false;
    if (mx(r[W_NB], bn) + mx(r[W_KB], BK) >= lim) return false;
    if (mx(r[W_ROWC_LO], BM) + mx(r[W_NC], bn) >= lim) return false;
    return true;
}

// conditions of pair_mfma_kstream_kernel: a tiny result reduced over a huge K
bool kstream_ok(const ctg_plan* p, const int64_t* r) {
    const int BK = MFMA_BK;
    const int64_t R = r[W_R], K = r[W_K], N = r[W_N];
    if (r[W_BT] != 1 || R > 32 || N > 32 || N < 1 || K < (1 << 16) || K % BK) return false;
    if (R > r[W_ROW_LO]) return false;                       // rows = one low-table lookup
    if (r[W_K_LO] % BK || (r[W_K_LO] & (r[W_K_LO] - 1))) return false;
    if (!tile_additive(p, r[W_KA], r[W_K_LO], K, BK)) return false;
    if (!tile_additive(p, r[W_KB], r[W_K_LO], K, BK)) return false;
    const int64_t lim = (int64_t)1 << 31;
    int64_t ma = 0, mk = 0, mn = 0, mkb = 0;
    for (int64_t i = 0; i < R; ++i) ma = std::max(ma, p->tables[r[W_ROWA_LO] + i]);
    for (int i = 0; i < BK; ++i) {
        const int64_t da = p->tables[r[W_KA] + i] - p->tables[r[W_KA]];
        const int64_t db = p->tables[r[W_KB] + i] - p->tables[r[W_KB]];
        if (da < 0 || db < 0) return false;
        mk = std::max(mk, da);
        mkb = std::max(mkb, db);
    }
    for (int64_t i = 0; i < N; ++i) mn = std::max(mn, p->tables[r[W_NB] + i]);
    for (int64_t i = 0; i < R; ++i)
        if (p->tables[r[W_ROWA_LO] + i] < 0) return false;
    for (int64_t i = 0; i < N; ++i)
        if (p->tables[r[W_NB] + i] < 0) return false;
    return ma + mk < lim && mn + mkb < lim;
}

// conditions of pair_skinny_kernel: a huge number of rows, a handful of
// multiply-adds per row; row pairs (2i, 2i+1) contiguous and 16-byte aligned in
// A and in C, the N output columns contiguous in C
bool skinny_ok(const ctg_plan* p, const int64_t* r) {
    const int64_t R = r[W_R], K = r[W_K], N = r[W_N], L = r[W_ROW_LO];
    if (r[W_BT] != 1 || R < (1 << 16) || (R & 1) || (L & 1) || K < 2 || K > 16) return false;
    if (N != 1 && N != 2 && N != 4) return false;
    if (L & (L - 1)) return false;   // rows are split with shift / mask
    if (K > r[W_K_LO]) return false; // single-level k tables
    if ((K & (K - 1)) || K * N > 16) return false;
    for (int64_t i = 0; i < N; ++i)
        if (p->tables[r[W_NC] + i] != i) return false;
    for (int64_t lo = 0; lo < L; lo += 2) {
        const int64_t a0 = p->tables[r[W_ROWA_LO] + lo], c0 = p->tables[r[W_ROWC_LO] + lo];
        if ((a0 & 1) || p->tables[r[W_ROWA_LO] + lo + 1] != a0 + 1) return false;
        if ((c0 & 1) || p->tables[r[W_ROWC_LO] + lo + 1] != c0 + N) return false;
    }
    if (!all_even(p, r[W_ROWA_HI], r[W_ROW_HI_LEN]) || !all_even(p, r[W_ROWC_HI], r[W_ROW_HI_LEN]))
        return false;
    if (!all_even(p, r[W_KA_HI], r[W_K_HI_LEN]) || !all_even(p, r[W_KA], r[W_K_LO])) return false;
    if ((r[W_A_OFF] & 1) || (r[W_C_OFF] & 1)) return false;
    if (r[W_A_LEAF] >= 0)
        for (int64_t j = 0; j < p->n_sliced; ++j)
            if (p->slice_strides[r[W_A_LEAF] * p->n_sliced + j] & 1) return false;
    return true;
}

// Can operand A (which = 0) / B (which = 1) of a real-valued matrix-core step be gathered in
// pieces of V elements (16 bytes) along its fastest index -- k when `kfast`, else the rows
// (A) / columns (B)?  Every piece must be contiguous and aligned: the fast table advances by 1
// inside a piece and starts pieces at multiples of V, every other offset that enters the
// address (the other groups' tables, the batch table, the operand's base, its slice strides)
// is a multiple of V, and the extents are whole numbers of pieces.
bool real_vec_ok(const ctg_plan* p, const int64_t* r, int which, bool kfast, int64_t V) {
    const auto& T = p->tables;
    auto all_mult = [&](int64_t off, int64_t len) {
        if (off < 0) return true;
        for (int64_t i = 0; i < len; ++i)
            if (T[off + i] % V) return false;
        return true;
    };
    auto pieces = [&](int64_t off, int64_t len) {   // contiguous, aligned runs of V
        if (off < 0 || len % V) return false;
        for (int64_t i = 0; i < len; i += V) {
            if (T[off + i] % V) return false;
            for (int64_t j = 1; j < V; ++j)
                if (T[off + i + j] != T[off + i] + j) return false;
        }
        return true;
    };
    const int64_t k_lo = r[W_K_LO], k_hi = r[W_K_HI_LEN], row_lo = r[W_ROW_LO], row_hi = r[W_ROW_HI_LEN];
    const int off_w = which ? W_B_OFF : W_A_OFF, leaf_w = which ? W_B_LEAF : W_A_LEAF;
    if (r[off_w] % V) return false;
    if (r[leaf_w] >= 0)
        for (int64_t j = 0; j < p->n_sliced; ++j)
            if (p->slice_strides[r[leaf_w] * p->n_sliced + j] % V) return false;
    if (!all_mult(which ? r[W_BB] : r[W_BA], r[W_BT])) return false;
    const int64_t klo_w = which ? r[W_KB] : r[W_KA], khi_w = which ? r[W_KB_HI] : r[W_KA_HI];
    if (!all_mult(khi_w, k_hi)) return false;
    if (which == 0) {
        if (!all_mult(r[W_ROWA_HI], row_hi)) return false;
        if (kfast) return pieces(klo_w, k_lo) && all_mult(r[W_ROWA_LO], row_lo);
        return pieces(r[W_ROWA_LO], row_lo) && all_mult(klo_w, k_lo);
    }
    if (kfast) return pieces(klo_w, k_lo) && all_mult(r[W_NB], r[W_N]);
    return pieces(r[W_NB], r[W_N]) && all_mult(klo_w, k_lo);
}

// zmult: how many slices a launch with these hints carries at most (1, or the
// executor's batch size); `like`: hints already built for zmult = 1, whose k-splits
// are kept
int build_hints_into(ctg_exec* e, std::vector<MfmaHints>& hints, int64_t zmult,
                     const std::vector<MfmaHints>* like, uint16_t** d_ord_out, char** d_lane_out) {
    const ctg_plan* p = e->plan;
    std::vector<uint16_t> blob;
    std::vector<size_t> offA(p->n_steps, 0), offB(p->n_steps, 0);
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        if (r[W_KIND] != KIND_PAIR || r[W_KERNEL] != KERNEL_MFMA) continue;
        MfmaHints& h = hints[s];
        if (p->dtype != CTG_C64) {
            // FP64 / real kernels: only need to know which group holds each operand's
            // fastest-varying memory index (kept in h.vecA: bit0 A, bit1 B)
            auto stride1 = [&](int w, int64_t len) -> int64_t {
                if (len < 2 || r[w] < 0) return INT64_MAX;
                const int64_t v = p->tables[r[w] + 1] - p->tables[r[w]];
                return v < 0 ? -v : (v == 0 ? INT64_MAX : v);
            };
            const int64_t ak = stride1(W_KA, r[W_K_LO]), am = stride1(W_ROWA_LO, r[W_ROW_LO]);
            const int64_t bk = stride1(W_KB, r[W_K_LO]), bnn = stride1(W_NB, r[W_N]);
            h.vecA = (ak < am ? 1 : 0) | (bk < bnn ? 2 : 0);
            // float32 / float64: bit 2 / bit 3 = operand A / B can be gathered in 16-byte
            // pieces along that fastest index (every piece contiguous and aligned, every
            // other offset a multiple of the piece)
            if (p->dtype == CTG_F32 || p->dtype == CTG_F64) {
                const int64_t V = p->dtype == CTG_F32 ? 4 : 2;
                if (real_vec_ok(p, r, 0, (h.vecA & 1) != 0, V)) h.vecA |= 4;
                if (real_vec_ok(p, r, 1, (h.vecA & 2) != 0, V)) h.vecA |= 8;
            }
            continue;
        }
        h.bn = mfma_pick_bn(r[W_N]);
        if (kstream_ok(p, r)) {
            h.stream = 2;
            h.bn = r[W_N] <= 16 ? 16 : 32;
            build_mfma_order(p, r, h.bn, 32, blob, &offA[s], &offB[s], &h.vecA, 64);
            continue;
        }
        if (skinny_ok(p, r)) {
            h.stream = 3;
            continue;
        }
        h.stream = mfma_use_stream(r[W_R], r[W_BT], r[W_K], r[W_N]) ? 1 : 0;
        // big square-ish GEMMs: 128x128 tiles (fast path only) halve the LDS
        // traffic and barriers per flop
        // (a small result under a very long contraction gets its parallelism from
        // split-K: widest tile there too)
        const int64_t splits = r[W_K] / MFMA_BK / 4;   // k-splits launch_cfg may use
        const int64_t tiles128 = ((r[W_R] + 127) / 128) * ((r[W_N] + 127) / 128) * r[W_BT];
        if (!h.stream && r[W_N] % 128 == 0 && r[W_K] >= 256 &&
            (r[W_R] * r[W_N] * zmult >= (1ll << 22) || tiles128 * splits * zmult >= 1024) &&
            mfma_fast_ok(p, r, 128))
            h.bn = 128;
        // small problems: narrower column tiles until the output alone gives every
        // CU a block -- cheaper than split-K (no slabs to write and reduce); long
        // contractions (K >= 1024) count the k-splits as blocks
        // long contractions on full 64-column tiles: candidates for bf16 x 3 products (MfmaHints::bf3)
        // (... of a size that fills the chip with 64-column tiles in a launch of the plan's nominal batch -- a
        // function of the plan, like the k-splits: a small tree's steps keep their narrow tiles and shared launches)
        h.bf3 = (!h.stream && r[W_K] >= 64 && r[W_N] % 64 == 0 && r[W_R] % MFMA_BM == 0 && r[W_K] % MFMA_BK == 0 &&
                 r[W_BT] == 1 && (r[W_R] / MFMA_BM) * (r[W_N] / 64) * std::max<int64_t>(e->batch_nominal, 1) >= 512 &&
                 mfma_fast_ok(p, r, 64) && !env_on("CTG_NO_PAIR_BF3")) ? 1 : 0;
        // (such a step runs on 64-column tiles: 74 KB of limb planes, two workgroups per CU -- one workgroup's
        // staging under the other's MFMAs; the 128-column tile the fp32 kernel prefers for K >= 256 was measured
        // 16 % slower per launch here: 98 KB, one workgroup of four waves per CU)
        if (h.bf3) h.bn = 64;
        // (experiment, CTG_PAIR_BF3_BN=128: the 128-column tile for such steps -- two limbs leave it 64 KB of planes)
        if (h.bf3 && r[W_N] % 128 == 0 && env_on("CTG_PAIR_BF3_BN128") && mfma_fast_ok(p, r, 128) &&
            (r[W_R] / MFMA_BM) * (r[W_N] / 128) * std::max<int64_t>(e->batch_nominal, 1) >= 512)
            h.bn = 128;
        if (!h.stream) {
            const int64_t tiles_m = (r[W_R] + MFMA_BM - 1) / MFMA_BM;
            const int64_t per_tile = r[W_K] >= 1024 ? splits : 1;
            // (two blocks per CU: a CU with a single block has nothing to overlap its
            // gather latency with -- 8x8 lattice, 4096 x 256 x 256 step: 49 -> 20 us)
            static const int64_t fill = getenv("CTG_TILE_FILL") ? atoll(getenv("CTG_TILE_FILL")) : 512;
            while (h.bn > 16 && !h.bf3 && tiles_m * ((r[W_N] + h.bn - 1) / h.bn) * r[W_BT] * per_tile * zmult < fill)
                h.bn /= 2;
            // the number of k-splits belongs to the step: taken from the single-slice
            // hints; a wider tile whose slabs would not fit the scratch is given up
            h.splitk = (int)(like ? (*like)[s].splitk
                                  : mfma_split_count(r[W_R], r[W_N], r[W_K], r[W_BT], h.bn, kScratchBytes,
                                                     e->batch_nominal));
            if (like) {
                const int64_t slab = tiles_m * MFMA_BM * ((r[W_N] + h.bn - 1) / h.bn) * h.bn * 8 * r[W_BT];
                if ((int64_t)h.splitk * slab > kScratchBytes && !h.bf3) h.bn = (*like)[s].bn;
            }
        }
        h.additive32 = (h.stream && tile_additive(p, r[W_ROWA_LO], r[W_ROW_LO], r[W_R], 32) &&
                        tile_additive(p, r[W_ROWC_LO], r[W_ROW_LO], r[W_R], 32))
                           ? 1
                           : 0;
        // the kernel keeps the 32 tile-local row offsets in 32-bit registers
        for (int rr = 0; h.additive32 && rr < 32; ++rr) {
            const int64_t a = p->tables[r[W_ROWA_LO] + rr], c = p->tables[r[W_ROWC_LO] + rr];
            if (a < 0 || a > INT32_MAX || c < 0 || c > INT32_MAX) h.additive32 = 0;
        }
        {
            // Tall steps with a handful of multiply-adds per row go to the row-wise FMA
            // kernel instead of 1/8-full MFMA tiles when (a) K, N <= 8: fewer instructions
            // per byte than staging 32-row groups through LDS (200-tensor hyper network:
            // 2.6-3.0 -> 3.2-3.9 TB/s), (b) their 32-row groups are not base + constant
            // (extents that are not powers of two) and the contraction is shorter than one
            // MFMA k-step (1.7 -> 3.5-4.6 TB/s), (c) they carry a batch index, which the
            // streaming kernel does not take.  CTG_ROWWISE=0 turns the kernel off, 2 sends
            // every step of its shape there (experiments).
            static const int rw = getenv("CTG_ROWWISE") ? atoi(getenv("CTG_ROWWISE")) : 1;
            // (the shapes rowwise_ok() takes: the batch index rides in gridDim.z)
            const bool shape = r[W_K] <= 32 && r[W_N] <= 32 && r[W_R] >= 8192 && r[W_BT] <= 65535;
            const bool pick = (h.stream == 1 && ((r[W_K] <= 8 && r[W_N] <= 8) ||
                                                 (!h.additive32 && r[W_K] < MFMA_BK) || rw >= 2)) ||
                              (h.stream == 0 && r[W_BT] > 1);
            if (rw > 0 && shape && pick) {
                h.stream = 4;
                // h.vecA bit 0: the output columns are the fastest-varying index of C
                h.vecA = (r[W_N] >= 2 && p->tables[r[W_NC] + 1] - p->tables[r[W_NC]] == 1) ? 1 : 0;
                if (getenv("CTG_ROWWISE_T")) h.vecA &= atoi(getenv("CTG_ROWWISE_T"));
                continue;
            }
        }
        build_mfma_order(p, r, h.bn, h.stream ? 32 : MFMA_BM, blob, &offA[s], &offB[s], &h.vecA);
        h.fast = (!h.stream && mfma_fast_ok(p, r, h.bn)) ? 1 : 0;
        if (!h.fast || h.bn < 64) h.bf3 = 0;
    }
    if (blob.empty()) return CTG_OK;
    HIP_TRY(hipMalloc((void**)d_ord_out, blob.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(*d_ord_out, blob.data(), blob.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    int64_t n_fast = 0;
    for (int64_t s = 0; s < p->n_steps; ++s) {
        if (hints[s].bn == 0) continue;
        hints[s].ordA = *d_ord_out + offA[s];
        hints[s].ordB = *d_ord_out + offB[s];
        if (hints[s].fast) ++n_fast;
    }
    // lane-constant tables of the fast tiled steps: built on the device once, read by
    // every block of every launch (the tables they derive from never change)
    if (n_fast > 0 && !env_on("CTG_NO_LANE_TABLES")) {
        const int64_t each = fast_lane_table_bytes();
        HIP_TRY(hipMalloc((void**)d_lane_out, n_fast * each));
        int64_t i = 0;
        for (int64_t s = 0; s < p->n_steps; ++s) {
            MfmaHints& h = hints[s];
            if (h.bn == 0 || !h.fast) continue;
            void* out = *d_lane_out + i * each;
            ++i;
            hipError_t err = launch_fast_lane_consts(e->args[s], h, out, e->stream);
            if (err != hipSuccess)
                return fail(CTG_E_HIP, "lane-constant kernel of step %lld failed: %s", (long long)s,
                            hipGetErrorString(err));
            h.lane = out;
        }
    }
    return CTG_OK;
}

int build_hints(ctg_exec* e) {
    int rc = build_hints_into(e, e->hints, 1, nullptr, &e->d_ord, &e->d_lane);
    // whose record of its largest |component| has a reader: the producers of a stem launch's big operand and of a
    // long tiled step's operands (fp16 x 2 splits them under it) -- the fp32 tiled / streaming kernels record only then
    e->rec_wanted.assign(e->plan->n_steps, 0);
    if (rc == CTG_OK && e->plan->dtype == CTG_C64)
        for (int64_t t = 0; t < e->plan->n_steps; ++t) {
            const int64_t* r = &e->plan->steps[t * STEP_WORDS];
            auto want = [&](int64_t prod) { if (prod >= 0 && prod < e->plan->n_steps) e->rec_wanted[prod] = 1; };
            if (r[W_KIND] == KIND_STEM2) want(r[W_A_PROD]);
            if (r[W_KIND] == KIND_PAIR && r[W_KERNEL] == KERNEL_MFMA && e->hints[t].bf3) {
                want(r[W_A_PROD]);
                want(r[W_B_PROD]);
            }
        }
    if (rc != CTG_OK || e->batch <= 1 || e->plan->dtype != CTG_C64) return rc;
    e->hints_b.assign(e->plan->n_steps, MfmaHints{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0});
    return build_hints_into(e, e->hints_b, e->batch, &e->hints, &e->d_ord_b, &e->d_lane_b);
}

// does step s write partial results to the scratch buffer?
bool step_needs_scratch(const ctg_exec* e, int64_t s) {
    const ctg_plan* p = e->plan;
    const int64_t* r = &p->steps[s * STEP_WORDS];
    if (r[W_KIND] != KIND_PAIR) return false;
    if (r[W_KERNEL] != KERNEL_MFMA) return !valu_thread_per_output(e->args[s]);
    if (p->dtype != CTG_C64) return false;   // (the real / double kernels have no k-split)
    auto needs = [](const MfmaHints& h) {
        if (h.stream == 2) return true;                   // k-streaming: per-wave partial tiles
        if (h.stream != 0) return false;                  // streaming / skinny / row-wise: none
        return h.splitk != 1;                             // tiled: slabs of the k-splits
    };
    return needs(e->hints[s]) || (!e->hints_b.empty() && needs(e->hints_b[s]));
}

int ensure_scratch(ctg_exec* e) {
    if (e->d_scratch) return CTG_OK;
    if (hipMalloc(&e->d_scratch, e->scratch_total) != hipSuccess) {
        (void)hipGetLastError();
        e->d_scratch = nullptr;
        return fail(CTG_E_NOMEM, "out of device memory allocating %lld bytes of scratch",
                    (long long)e->scratch_total);
    }
    return CTG_OK;
}

// the double-precision running sum an accumulate step adds into (float / complex64 results; null: none):
// the same element offset as the step's result operand
void* wide_of_step(const ctg_exec* e, int64_t s) {
    if (!e->d_wide) return nullptr;
    const int64_t* r = &e->plan->steps[s * STEP_WORDS];
    if (r[W_C_SPACE] != SPACE_RESULT) return nullptr;
    return e->d_wide + r[W_C_OFF] * 2 * kItemSize[e->plan->dtype];
}



// Where the arena lies in physical HBM is worth 2-3 % of a slice of a big tree (profiles/r6_process_alternation.txt: the
// memory of this part is not uniform, and consecutive allocations of an arena that fills more than a third of it alternate
// between two physical regions).  A big single-slice arena is therefore allocated TWICE where the memory allows, the
// ranges of the plan's largest tensors are read once in either copy (a streaming read, timed), and the copy that reads them
// faster is kept.  CTG_ARENA_PLACE=0: the first allocation, as before.  Results do not depend on it.
static void place_arena(ctg_exec* e) {
    const ctg_plan* p = e->plan;
    const int64_t isz = kItemSize[p->dtype];
    const int64_t bytes = p->arena_elems * isz * e->batch;
    // (tests: CTG_ARENA_PLACE_MIN = smallest arena in bytes, CTG_ARENA_PLACE_SEQ=1 = one allocation after the other)
    const int64_t min_bytes = getenv("CTG_ARENA_PLACE_MIN") ? atoll(getenv("CTG_ARENA_PLACE_MIN")) : ((int64_t)32 << 30);
    if (p->dtype != CTG_C64 || e->batch != 1 || bytes < min_bytes) return;
    if (const char* v = getenv("CTG_ARENA_PLACE"))
        if (v[0] == '\0' || (v[0] == '0' && v[1] == '\0')) return;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    // (an arena that does not fit twice -- 169 GiB of 288 -- is tried one allocation after the other: a freed arena is not
    // what the next hipMalloc of its size gets, the allocator alternates between the two regions)
    const bool side_by_side = (int64_t)free_b >= bytes + ((int64_t)8 << 30) && !env_on("CTG_ARENA_PLACE_SEQ");
    // the operands and results of the steps that move the most data
    std::vector<std::pair<int64_t, std::pair<int64_t, int64_t>>> cand;   // (bytes moved, (offset, elements))
    for (int64_t s = 0; s < p->n_steps; ++s) {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        if (r[W_KIND] != KIND_PAIR && r[W_KIND] != KIND_STEM2) continue;
        const int64_t moved = r[W_A_SIZE] + r[W_C_SIZE];
        if (r[W_A_SPACE] == SPACE_ARENA) cand.push_back({moved, {r[W_A_OFF], r[W_A_SIZE]}});
        if (r[W_C_SPACE] == SPACE_ARENA) cand.push_back({moved, {r[W_C_OFF], r[W_C_SIZE]}});
    }
    if (cand.empty()) return;
    std::sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    if (cand.size() > 8) cand.resize(8);
    float* slot = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ok = hipMalloc((void**)&slot, sizeof(float) * kMaxSub) == hipSuccess && hipEventCreate(&ev0) == hipSuccess &&
              hipEventCreate(&ev1) == hipSuccess;
    // one timed reading of the ranges in the copy at `base` (ms; < 0: failed)
    auto probe = [&](char* base) -> float {
        bool good = hipMemsetAsync(slot, 0, sizeof(float) * kMaxSub, e->stream) == hipSuccess &&
                    hipEventRecord(ev0, e->stream) == hipSuccess;
        for (const auto& q : cand)
            if (good) good = launch_maxabs_f32(base + q.second.first * isz, nullptr, 0, 0, 0, q.second.second, slot, e->stream) == hipSuccess;
        good = good && hipEventRecord(ev1, e->stream) == hipSuccess && hipEventSynchronize(ev1) == hipSuccess;
        float t = 0.f;
        if (good) good = hipEventElapsedTime(&t, ev0, ev1) == hipSuccess;
        return good ? t : -1.f;
    };
    float ms[2] = {0.f, 0.f};
    const char* verdict = "first";
    if (ok && side_by_side) {
        char* other = nullptr;
        if (hipMalloc((void**)&other, (size_t)bytes) == hipSuccess) {
            char* base[2] = {e->d_arena, other};
            for (int pass = 0; ok && pass < 3; ++pass)          // (first pass: untimed; then the two copies in turn, twice)
                for (int c = 0; ok && c < 2; ++c) {
                    const float t = probe(base[c]);
                    ok = t >= 0.f;
                    if (pass > 0) ms[c] += t;
                }
            if (ok && ms[1] < 0.995f * ms[0]) {
                std::swap(e->d_arena, other);
                verdict = "second";
            }
            (void)hipFree(other);
        } else {
            (void)hipGetLastError();
        }
    } else if (ok) {
        // one after the other: first (probe), free, second (probe); back to the first's region by a third allocation if
        // that one read faster
        auto timed = [&](char* base) { float t = probe(base); if (t >= 0.f) { const float a = probe(base), b = probe(base); t = (a >= 0.f && b >= 0.f) ? a + b : -1.f; } return t; };
        ms[0] = timed(e->d_arena);
        char* again = nullptr;
        if (ms[0] >= 0.f && hipFree(e->d_arena) == hipSuccess) {
            e->d_arena = nullptr;
            if (hipMalloc((void**)&again, (size_t)bytes) == hipSuccess) {
                e->d_arena = again;
                ms[1] = timed(e->d_arena);
                verdict = "second";
                if (ms[1] >= 0.f && ms[0] < 0.995f * ms[1] && hipFree(e->d_arena) == hipSuccess) {
                    e->d_arena = nullptr;
                    if (hipMalloc((void**)&again, (size_t)bytes) == hipSuccess) e->d_arena = again;
                    verdict = "third (the first's region)";
                }
            }
        }
    }
    if (env_on("CTG_ARENA_DEBUG"))
        fprintf(stderr, "arena placement (%s): %d ranges, first %.3f ms, second %.3f ms -> %s, arena %p\n",
                side_by_side ? "side by side" : "one after the other", (int)cand.size(), ms[0], ms[1], ok ? verdict : "probe failed",
                e->d_arena);
    if (slot) (void)hipFree(slot);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
}

// a long tiled complex64 step that multiplies on the 16-bit matrix cores right now (MfmaHints::bf3; round 6: such a
// launch records its result's largest |component| and may run in the fp16 x 2 arithmetic, see launch_step)
static bool tiled16_step(const ctg_exec* e, int64_t s) {
    const ctg_plan* p = e->plan;
    const int64_t* r = &p->steps[s * STEP_WORDS];
    if (r[W_KIND] != KIND_PAIR || r[W_KERNEL] != KERNEL_MFMA || p->dtype != CTG_C64) return false;
    const MfmaHints& h = e->hints[s];
    return !h.stream && h.fast && h.bf3 && e->stem_arith != 0 && !e->strip && e->d_stem_max != nullptr &&
           pair_bf16x3_on(e->args[s]);
}

// does a launch of pair step s record its result's largest |component| (what launch_step decides, on the shapes)?
static bool pair_records(const ctg_exec* e, int64_t s) {
    const ctg_plan* p = e->plan;
    const int64_t* r = &p->steps[s * STEP_WORDS];
    if (r[W_KIND] != KIND_PAIR || r[W_KERNEL] != KERNEL_MFMA || p->dtype != CTG_C64) return false;
    if (e->stem_arith == 0 || e->strip || e->d_stem_max == nullptr || e->grouped[s]) return false;
    const MfmaHints& h = e->hints[s];
    if (tiled16_step(e, s)) return h.splitk <= 1 && e->args[s].zqA <= 1 && e->args[s].zqB <= 1;
    if (e->stem_arith != 2 || e->wave_member[s] || !e->rec_wanted[s]) return false;
    return h.stream == 1 || (!h.stream && h.fast && !h.bf3 && h.splitk <= 1);
}

int launch_step(ctg_exec* e, int64_t s, hipStream_t stream) {
    const ctg_plan* p = e->plan;
    const int64_t* r = &p->steps[s * STEP_WORDS];
    hipError_t err = hipSuccess;
    // (allocated by ctg_exec_create for every plan with such a step; a kernel that takes
    // scratch must never see a null pointer: keep the check next to the launch)
    if (!e->d_scratch && step_needs_scratch(e, s)) {
        const int rc = ensure_scratch(e);
        if (rc != CTG_OK) return rc;
    }
    switch (r[W_KIND]) {
        case KIND_SINGLE: err = launch_single(p->dtype, e->args[s], stream); break;
        case KIND_ACCUM:
            if (e->strip) {
                err = launch_strip_prepare(e->d_fac, e->d_counted, p->n_steps, e->root_step,
                                           e->check_zero, e->d_strip, e->d_inscale, stream);
                if (err == hipSuccess)
                    err = launch_rescale(p->dtype, e->d_result, p->result_elems, e->d_strip, stream);
                if (err == hipSuccess && e->d_wide)
                    err = launch_rescale(p->dtype + 1, e->d_wide, p->result_elems, e->d_strip, stream);
                if (err == hipSuccess) err = launch_accum(p->dtype, e->args[s], e->d_strip, wide_of_step(e, s), nullptr, stream);
            } else {
                err = launch_accum(p->dtype, e->args[s], nullptr, wide_of_step(e, s), e->d_inscale, stream);
            }
            break;
        case KIND_STEM2: {
            // (slices of a batch one after the other: the workgroups are persistent)
            const int nz = e->args[s].nz;
            // fp16 x 2 (round 6): the executor's arithmetic, not under strip_exponent (a scale per step there), for a
            // shape with a 16-bit-pipe kernel whose big operand was produced by a 16-bit-pipe stem launch (that launch
            // recorded the operand's largest element: the power of two the split needs).  A big operand of any other
            // origin -- the first pair of a stem -- is multiplied in bf16 x 3, which needs no scale and records its
            // result's largest element as well.  CTG_STEM_H2=0 in the environment (read per launch) says no.
            const bool rec = e->stem_arith != 0 && !e->strip && e->d_stem_max != nullptr;
            const int64_t prod = r[W_A_PROD];
            // (a recording producer: a stem launch of the 16-bit pipe or a long tiled step of it, see KIND_PAIR below)
            const bool prod_rec = prod >= 0 && prod < p->n_steps && e->stem_h2_ran[prod];
            // (CTG_STEM_H2_ALL=1, tests and diagnostics: fp16 x 2 for EVERY capable pair -- a max-abs pass over the
            // big operand supplies the scale where no producer recorded it)
            const bool h2_all = env_on("CTG_STEM_H2_ALL");
            bool h2 = e->stem_arith == 2 && rec && (prod_rec || h2_all) && stem2h_uses_h2(e->stem_args[s]);
            if (h2)
                if (const char* v = getenv("CTG_STEM_H2")) h2 = !(v[0] == '\0' || (v[0] == '0' && v[1] == '\0'));
            const bool records = h2 || (rec && stem2_uses_bf3(e->stem_args[s]));
            e->stem_h2_ran[s] = records ? 1 : 0;
            if (env_on("CTG_STEM_DEBUG"))
                fprintf(stderr, "stem step %lld: arith %d rec %d prod %lld prod_rec %d uses_h2 %d uses_bf3 %d -> h2 %d records %d\n",
                        (long long)s, e->stem_arith, (int)rec, (long long)prod, (int)prod_rec, (int)stem2h_uses_h2(e->stem_args[s]),
                        (int)stem2_uses_bf3(e->stem_args[s]), (int)h2, (int)records);
            for (int z = 0; z < nz && err == hipSuccess; ++z) {
                StemArgs q = e->stem_args[s];
                q.z0 = e->args[s].z0 + z;
                q.nz = 1;
                // (a slot per slice of the batch; a slice-invariant tensor's record lives in slot 0)
                const int64_t zc = (e->invariant[s] || e->grouped[s]) ? 0 : q.z0;
                const int64_t za = (prod_rec && (e->invariant[prod] || e->grouped[prod])) ? 0 : q.z0;
                q.amax = h2 ? e->smax_slot(0, prod_rec ? prod : 0, za) : nullptr;
                q.cmax = records ? e->smax_slot(0, s, zc) : nullptr;
                if (h2 && !prod_rec) {
                    float* slot = e->smax_slot(1, s, q.z0);
                    err = hipMemsetAsync(slot, 0, sizeof(float) * kMaxSub, stream);
                    // (an input tensor read in place: the whole leaf, see the tiled steps below)
                    if (err == hipSuccess)
                        err = r[W_A_LEAF] >= 0 ? launch_maxabs_f32(q.A, nullptr, 0, 0, 0, q.a_elems, slot, stream)
                                               : launch_maxabs_f32(q.A, q.soffA, q.z0, q.zsA, q.zA, q.a_elems, slot, stream);
                    q.amax = slot;
                }
                if (err == hipSuccess) err = h2 ? launch_stem2h(q, stream) : launch_stem2(q, stream);
            }
            break;
        }
        case KIND_PAIR:
            if (r[W_KERNEL] == KERNEL_MFMA && p->dtype == CTG_C128)
                err = launch_pair_mfma_c128(e->args[s], e->hints[s].vecA /* = stride flags */, stream);
            else if (r[W_KERNEL] == KERNEL_MFMA && p->dtype != CTG_C64)
                err = launch_pair_mfma_real(p->dtype, e->args[s], e->hints[s].vecA, stream);
            else if (r[W_KERNEL] == KERNEL_MFMA) {
                const MfmaHints& h0 = (e->args[s].nz > 1 && !e->hints_b.empty()) ? e->hints_b[s] : e->hints[s];
                // Long tiled steps on the 16-bit matrix cores (MfmaHints::bf3, pair_mfma_bf3_kernel), round 6: a launch of
                // one slice without k-splits records the largest |component| it stores, like a stem launch does, and --
                // the executor's arithmetic being fp16 x 2 -- multiplies with two fp16 limbs (pair_mfma_h2_kernel) under
                // the operands' recorded maxima; an operand whose producer recorded nothing gets a max-abs pass (its
                // bytes once more, against K >= 64 products per element).  CTG_PAIR_H2=0 in the environment says no.
                const bool tiled16 = tiled16_step(e, s) && !h0.stream && h0.fast && h0.bf3;
                if (tiled16) {
                    MfmaHints h = h0;
                    // (no k-splits: the slabs hold unscaled sums; no operand that a slice group shares: its replica
                    // is not the slice's own)
                    const StepArgs& a = e->args[s];
                    const bool single = h0.splitk <= 1 && a.zqA <= 1 && a.zqB <= 1 && !e->grouped[s] &&
                                        a.z0 + a.nz <= std::max(e->batch, 1);
                    bool h2 = e->stem_arith == 2 && single;
                    if (h2)
                        if (const char* v = getenv("CTG_PAIR_H2")) h2 = !(v[0] == '\0' || (v[0] == '0' && v[1] == '\0'));
                    h.h2 = h2 ? 1 : 0;
                    // (the kernel adds z0 + blockIdx.y times the stride: pointers to slot 0 of the step)
                    h.cmax = single ? e->smax_slot(0, s, 0) : nullptr;
                    h.cmax_zs = e->invariant[s] ? 0 : 1;
                    e->stem_h2_ran[s] = single ? 1 : 0;
                    for (int side = 0; side < 2 && h2 && err == hipSuccess; ++side) {
                        const int64_t prod = r[side == 0 ? W_A_PROD : W_B_PROD];
                        const float* mx = nullptr;
                        int zs = 1;
                        if (prod >= 0 && prod < p->n_steps && e->stem_h2_ran[prod] && !e->grouped[prod]) {
                            mx = e->smax_slot(0, prod, 0);
                            zs = e->invariant[prod] ? 0 : 1;
                        } else if (r[side == 0 ? W_A_LEAF : W_B_LEAF] >= 0) {
                            // an input tensor read in place: W_x_SIZE is the whole leaf, a slice of it need not be
                            // contiguous -- the largest element of the WHOLE leaf bounds every slice's (one record for
                            // all slices of the launch)
                            float* slot = e->smax_slot(1 + side, s, 0);
                            err = hipMemsetAsync(slot, 0, sizeof(float) * kMaxSub, stream);
                            if (err == hipSuccess)
                                err = launch_maxabs_f32(side == 0 ? a.A : a.B, nullptr, 0, 0, 0, r[side == 0 ? W_A_SIZE : W_B_SIZE], slot, stream);
                            mx = slot;
                            zs = 0;
                        } else {
                            float* slot = e->smax_slot(1 + side, s, 0);
                            float* at = slot + (int64_t)a.z0 * kMaxSub;
                            err = hipMemsetAsync(at, 0, sizeof(float) * kMaxSub * (size_t)a.nz, stream);
                            if (err == hipSuccess)
                                err = side == 0 ? launch_maxabs_f32(a.A, a.soffA, a.z0, a.zsA, a.zA, r[W_A_SIZE], at, stream, a.nz, kMaxSub)
                                                : launch_maxabs_f32(a.B, a.soffB, a.z0, a.zsB, a.zB, r[W_B_SIZE], at, stream, a.nz, kMaxSub);
                            mx = slot;
                        }
                        (side == 0 ? h.amax : h.bmax) = mx;
                        (side == 0 ? h.amax_zs : h.bmax_zs) = zs;
                    }
                    if (env_on("CTG_STEM_DEBUG"))
                        fprintf(stderr, "tiled step %lld: arith %d single %d -> h2 %d records %d (A prod %lld, B prod %lld)\n",
                                (long long)s, e->stem_arith, (int)single, (int)h2, (int)single, (long long)r[W_A_PROD],
                                (long long)r[W_B_PROD]);
                    if (err == hipSuccess) err = launch_pair_mfma(p->dtype, e->args[s], h, e->d_scratch, kScratchBytes, stream);
                } else {
                    // (the fp32 tiled kernel without k-splits and the streaming kernel record as well, so that a tiled
                    // 16-bit step or a stem pair behind them needs no max-abs pass -- not inside a wave-front group,
                    // not what a slice group shares)
                    const StepArgs& a = e->args[s];
                    const bool can = e->stem_arith == 2 && !e->strip && e->d_stem_max != nullptr && !e->grouped[s] &&
                                     !env_on("CTG_NO_PAIR_RECORD") &&
                                     !e->wave_member[s] && e->rec_wanted[s] && a.z0 + a.nz <= std::max(e->batch, 1) &&
                                     ((h0.stream == 1) || (!h0.stream && h0.fast && !h0.bf3 && h0.splitk <= 1));
                    if (!e->stem_h2_ran.empty()) e->stem_h2_ran[s] = can ? 1 : 0;
                    if (can) {
                        MfmaHints h = h0;
                        h.cmax = e->smax_slot(0, s, 0);
                        h.cmax_zs = e->invariant[s] ? 0 : 1;
                        err = launch_pair_mfma(p->dtype, a, h, e->d_scratch, kScratchBytes, stream);
                    } else {
                        err = launch_pair_mfma(p->dtype, a, h0, e->d_scratch, kScratchBytes, stream);
                    }
                }
            } else
                err = launch_pair_valu(p->dtype, e->args[s], e->d_scratch, kScratchBytes, stream);
            break;
    }
    if (err == hipSuccess && e->strip && (r[W_KIND] == KIND_PAIR || r[W_KIND] == KIND_STEM2)) {
        // factor = max|p| of the freshly written intermediate (contiguous in the arena)
        const char* c = (const char*)space_ptr(e, r[W_C_SPACE]) + r[W_C_OFF] * kItemSize[p->dtype];
        err = launch_maxabs(p->dtype, c, r[W_C_SIZE], e->d_fac + s, stream);
    }
    if (err != hipSuccess)
        return fail(CTG_E_HIP, "launch of step %lld failed: %s", (long long)s, hipGetErrorString(err));
    return CTG_OK;
}

// Wave-front groups.  Small trees (and the early levels of any tree) are chains
// of launches of a few microseconds each, every one waiting for the previous to
// drain; the planner emits such trees level by level (plan.py: compile_tree) and
// runs of consecutive small steps that neither read nor overwrite each other's
// tensors go out as ONE launch per kernel shape.  The test is on the memory
// intervals of the plan itself, so any step order is safe: an order that
// interleaves dependent steps simply forms no groups.  Within a run the steps
// are mutually independent, so issuing them shape by shape is a legal reorder.
int build_groups(ctg_exec* e) {
    std::fill(e->wave_member.begin(), e->wave_member.end(), 0);
    const ctg_plan* p = e->plan;
    const int64_t n = p->n_steps;
    e->issue.clear();
    if (e->d_group_items) (void)hipFree(e->d_group_items);
    if (e->d_fast_items) (void)hipFree(e->d_fast_items);
    e->d_group_items = nullptr;
    e->d_fast_items = nullptr;
    // LDS-resident subtrees: which steps a component's workgroup runs (none under strip_exponent)
    {
        const int rc = ctg_lds_build(e);
        if (rc != CTG_OK) return rc;
    }
    bool lds_issued[2] = {false, false};
    auto lds_member = [&](int64_t s) { return e->lds_comp_of[s] >= 0; };
    // (strip_exponent measures every intermediate right after its step)
    const bool off = e->strip || env_on("CTG_NO_GROUPS");
    const bool fast_off = env_on("CTG_NO_FAST_GROUPS");
    // class of a step: -1 launches alone, 0 thread-per-output, 1 + key tiled fast kernel
    auto class_of = [&](int64_t s) -> int {
        const int64_t* r = &p->steps[s * STEP_WORDS];
        if (off || r[W_KIND] != KIND_PAIR || e->invariant[s] || lds_member(s)) return -1;
        if (r[W_KERNEL] != KERNEL_MFMA) {
            if (!valu_thread_per_output(e->args[s])) return -1;
            ValuGroupItem it;
            valu_group_fill(e->args[s], &it, 0);
            return it.n_tiles <= kValuGroupMaxTiles ? 0 : -1;
        }
        if (p->dtype != CTG_C64 || fast_off) return -1;
        if (e->hints[s].bf3) return -1;   // (its arithmetic is its own kernel's: never inside a shared launch)
        const int key = fast_group_key(e->args[s], e->hints[s]);
        // (a launch that carries several slices may prefer a wider tile: such steps stay alone)
        if (key < 0 || (!e->hints_b.empty() && fast_group_key(e->args[s], e->hints_b[s]) != key)) return -1;
        return 1 + key;
    };
    auto overlap = [&](const int64_t* x, int xs, int xo, int xn, const int64_t* y, int ys, int yo, int yn) {
        return x[xs] == y[ys] && x[xo] < y[yo] + y[yn] && y[yo] < x[xo] + x[xn];
    };
    auto independent = [&](int64_t j, int64_t i) {
        const int64_t* a = &p->steps[i * STEP_WORDS];
        const int64_t* b = &p->steps[j * STEP_WORDS];
        return !overlap(b, W_A_SPACE, W_A_OFF, W_A_SIZE, a, W_C_SPACE, W_C_OFF, W_C_SIZE) &&
               !overlap(b, W_B_SPACE, W_B_OFF, W_B_SIZE, a, W_C_SPACE, W_C_OFF, W_C_SIZE) &&
               !overlap(b, W_C_SPACE, W_C_OFF, W_C_SIZE, a, W_A_SPACE, W_A_OFF, W_A_SIZE) &&
               !overlap(b, W_C_SPACE, W_C_OFF, W_C_SIZE, a, W_B_SPACE, W_B_OFF, W_B_SIZE) &&
               !overlap(b, W_C_SPACE, W_C_OFF, W_C_SIZE, a, W_C_SPACE, W_C_OFF, W_C_SIZE);
    };
    std::vector<ValuGroupItem> vitems;
    std::vector<FastGroupItem> fitems;
    for (int64_t s = 0; s < n;) {
        if (e->invariant[s]) {
            ++s;
            continue;
        }
        if (lds_member(s)) {
            // all components of the step's sharing class go out as ONE launch, at the first member (members
            // read leaves, slice-invariant results and -- per-slice ones -- group-shared results only, and the
            // planner puts them before every other step of their class: cotengra_amd/plan.py)
            // (a member that is a leaf's preprocessing stands with the other preprocessing steps, before the
            // pair steps of EVERY class: the launch goes where the first member PAIR stands -- behind all
            // steps of the classes that run less often)
            const int cls = e->grouped[s] ? 0 : 1;
            if (!lds_issued[cls] && p->steps[s * STEP_WORDS + W_KIND] == KIND_PAIR) {
                lds_issued[cls] = true;
                e->issue.push_back(ctg_exec::Issue{s, -2, e->lds_first[cls], e->lds_count[cls], (uint32_t)e->lds_count[cls],
                                                   cls == 0});
            }
            ++s;
            continue;
        }
        if (class_of(s) < 0) {
            e->issue.push_back(ctg_exec::Issue{s, -1, 0, 1, 0, e->grouped[s] != 0});
            ++s;
            continue;
        }
        int64_t j = s + 1;
        // (a wave front holds steps the slices of a group share, or steps they do not: never both)
        for (; j < n && j - s < 64 && !e->invariant[j] && class_of(j) >= 0 && e->grouped[j] == e->grouped[s]; ++j) {
            bool ok = true;
            for (int64_t i = s; i < j && ok; ++i) ok = independent(j, i);
            if (!ok) break;
        }
        std::vector<char> done((size_t)(j - s), 0);
        for (int64_t a = s; a < j; ++a) {
            if (done[a - s]) continue;
            const int cls = class_of(a);
            std::vector<int64_t> members;
            for (int64_t b = a; b < j; ++b)
                if (!done[b - s] && class_of(b) == cls) {
                    members.push_back(b);
                    done[b - s] = 1;
                }
            if (members.size() == 1) {
                e->issue.push_back(ctg_exec::Issue{a, -1, 0, 1, 0, e->grouped[a] != 0});
                continue;
            }
            uint32_t blocks = 0;
            const int32_t item0 = (int32_t)(cls == 0 ? vitems.size() : fitems.size());
            for (int64_t m : members) {
                if ((size_t)m < e->wave_member.size()) e->wave_member[m] = 1;
                if (cls == 0) {
                    vitems.emplace_back();
                    blocks += valu_group_fill(e->args[m], &vitems.back(), blocks);
                } else {
                    fitems.emplace_back();
                    blocks += fast_group_fill(e->args[m], e->hints[m], &fitems.back(), blocks);
                }
            }
            e->issue.push_back(ctg_exec::Issue{a, cls, item0, (int32_t)members.size(), blocks, e->grouped[a] != 0});
        }
        s = j;
    }
    // slice groups: the launch list of a slice whose group's shared steps are done (they launch alone)
    e->issue_reuse.clear();
    for (const ctg_exec::Issue& q : e->issue)
        if (!q.shared) e->issue_reuse.push_back(q);
    e->group_key = -1;
    if (!vitems.empty()) {
        HIP_TRY(hipMalloc((void**)&e->d_group_items, vitems.size() * sizeof(ValuGroupItem)));
        HIP_TRY(hipMemcpy(e->d_group_items, vitems.data(), vitems.size() * sizeof(ValuGroupItem),
                          hipMemcpyHostToDevice));
    }
    if (!fitems.empty()) {
        HIP_TRY(hipMalloc((void**)&e->d_fast_items, fitems.size() * sizeof(FastGroupItem)));
        HIP_TRY(hipMemcpy(e->d_fast_items, fitems.data(), fitems.size() * sizeof(FastGroupItem),
                          hipMemcpyHostToDevice));
    }
    return CTG_OK;
}

// one entry of the per-slice launch list, for a batch of nb slices
int launch_issue(ctg_exec* e, const ctg_exec::Issue& q, int nb, hipStream_t stream) {
    // (batched slice groups: what a group shares goes out once per group of the launch)
    if (e->group_d > 1 && q.shared && nb > 1) nb /= e->group_d;
    if (q.cls == -2) {
        const int cls = q.shared ? 0 : 1;
        const hipError_t err = launch_lds_run(e->plan->dtype, e->d_lds_comps + q.item0, q.n, nb, 0, e->lds_bytes[cls], stream);
        if (err != hipSuccess)
            return fail(CTG_E_HIP, "launch of the %d LDS-resident subtrees at step %lld failed: %s", (int)q.n,
                        (long long)q.step, hipGetErrorString(err));
        return CTG_OK;
    }
    if (q.cls < 0) {
        e->args[q.step].nz = nb;
        const int rc = launch_step(e, q.step, stream);
        e->args[q.step].nz = 1;
        return rc;
    }
    const hipError_t err =
        q.cls == 0 ? launch_pair_valu_group(e->plan->dtype, e->d_group_items + q.item0, q.n, q.blocks, nb, stream)
                   : launch_pair_mfma_fast_group(q.cls - 1, e->d_fast_items + q.item0, q.n, q.blocks, nb, stream);
    if (err != hipSuccess)
        return fail(CTG_E_HIP, "launch of the %d steps grouped at step %lld failed: %s", (int)q.n,
                    (long long)q.step, hipGetErrorString(err));
    return CTG_OK;
}

// the slice id with the digits of the group indices set to zero: what the slices of a group share
int64_t slice_group_key(const ctg_plan* p, int64_t sid) {
    int64_t key = 0, rem = sid, stride = 1;
    for (int64_t j = p->n_sliced - 1; j >= 0; --j) {
        if (p->slice_fixed[j] >= 0) continue;
        const int64_t d = rem % p->slice_sizes[j];
        rem /= p->slice_sizes[j];
        if (!p->slice_group[j]) key += d * stride;
        stride *= p->slice_sizes[j];
    }
    return key;
}

// The digits of a slice id, least significant first (the last sliced index varies fastest), without
// the projected ones: (stride in the slice id, extent, is a group index).
struct SliceDigit { int64_t stride, size; bool group; };
std::vector<SliceDigit> slice_digits(const ctg_plan* p) {
    std::vector<SliceDigit> out;
    int64_t stride = 1;
    for (int64_t j = p->n_sliced - 1; j >= 0; --j) {
        if (p->slice_fixed[j] >= 0) continue;
        out.push_back({stride, p->slice_sizes[j], p->slice_group[j] == 1});
        stride = (stride > INT64_MAX / p->slice_sizes[j]) ? INT64_MAX : stride * p->slice_sizes[j];
    }
    return out;
}

int64_t plan_group_size(const ctg_plan* p) {
    int64_t d = 1;
    for (int64_t j = 0; j < p->n_sliced; ++j)
        if (p->slice_group[j] == 1 && p->slice_fixed[j] < 0) d *= p->slice_sizes[j];
    return d;
}

// the slices of group g (g in [0, nslices / group size): its digits are the values of the sliced
// indices that are not group indices), ascending, appended to `out`
void group_members(const std::vector<SliceDigit>& digits, int64_t g, std::vector<int64_t>& out) {
    int64_t base = 0, rem = g;
    for (const SliceDigit& d : digits)
        if (!d.group) {
            base += (rem % d.size) * d.stride;
            rem /= d.size;
        }
    const size_t at = out.size();
    out.push_back(base);
    for (const SliceDigit& d : digits)
        if (d.group) {
            const size_t n = out.size();
            for (int64_t v = 1; v < d.size; ++v)
                for (size_t i = at; i < n; ++i) out.push_back(out[i] + v * d.stride);
        }
    std::sort(out.begin() + at, out.end());
}

// A rank's share of the slices (ABI 6): the UNITS rank, rank + world, ... -- a unit is a whole slice
// group (a single slice for a plan without group indices, which makes this core.py:4070's round-robin).
struct Share { int64_t gsize, n_units_all, units; };
int plan_share(const ctg_plan* p, int64_t rank, int64_t world, Share* out) {
    if (world < 1 || rank < 0 || rank >= world) return fail(CTG_E_INVALID, "rank %lld of %lld", (long long)rank, (long long)world);
    Share sh;
    sh.gsize = plan_group_size(p);
    sh.n_units_all = p->nslices / sh.gsize;
    sh.units = rank < sh.n_units_all ? (sh.n_units_all - rank + world - 1) / world : 0;
    *out = sh;
    return CTG_OK;
}

// Slice groups: the given slices group by group -- same key (= all sliced indices but the group ones)
// one after the other, in slice order within a group --, the steps the slices of a group share launched
// when the key changes.  (Under strip_exponent every step keeps a scale per slice: the callers then take
// the ordinary path, on which the shared steps are simply computed for every slice.)
int run_grouped(ctg_exec* e, const int64_t* ids, size_t n_ids) {
    const ctg_plan* p = e->plan;
    std::vector<std::pair<int64_t, int64_t>> order(n_ids);
    for (size_t k = 0; k < n_ids; ++k) order[k] = {slice_group_key(p, ids[k]), ids[k]};
    std::sort(order.begin(), order.end());
    if (e->group_d > 1) {
        // Batched launches of whole groups: slice-in-batch z = group * d + member.  A group that is not
        // complete among the ids (or has a slice twice) goes slice by slice -- a launch of one slice is
        // consistent as it is (z = 0: every quantum and multiplier drops out), its shared steps are
        // computed for it alone.
        const size_t d = (size_t)e->group_d;
        std::vector<int64_t> rest;
        // the ids of the batched launches are staged in pinned memory the executor keeps (the copies to
        // the device are asynchronous; the event says when the previous call's have been consumed)
        if (e->ev_ids) HIP_TRY(hipEventSynchronize(e->ev_ids));
        if (e->h_ids_cap < n_ids) {
            if (e->h_ids) (void)hipHostFree(e->h_ids);
            e->h_ids = nullptr;
            e->h_ids_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&e->h_ids, std::max<size_t>(n_ids, 4096) * sizeof(int64_t), hipHostMallocDefault));
            e->h_ids_cap = std::max<size_t>(n_ids, 4096);
        }
        if (!e->ev_ids) HIP_TRY(hipEventCreateWithFlags(&e->ev_ids, hipEventDisableTiming));
        size_t n_full = 0;
        for (size_t k = 0; k < order.size();) {
            size_t j = k;
            while (j < order.size() && order[j].first == order[k].first) ++j;
            bool whole = j - k == d;
            for (size_t i = k + 1; whole && i < j; ++i) whole = order[i].second != order[i - 1].second;
            for (size_t i = k; i < j; ++i) {
                if (whole) e->h_ids[n_full++] = order[i].second;
                else rest.push_back(order[i].second);
            }
            k = j;
        }
        e->group_key = -1;
        for (size_t k = 0; k < n_full; k += (size_t)e->batch) {
            const int nb = (int)std::min<size_t>((size_t)e->batch, n_full - k);
            HIP_TRY(hipMemcpyAsync(e->d_batch_ids, e->h_ids + k, nb * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
            hipError_t err = launch_prologue(e->meta, e->d_state, e->d_soff, 0, e->stream, nb, 1, e->d_batch_ids);
            if (err != hipSuccess) return fail(CTG_E_HIP, "prologue launch failed: %s", hipGetErrorString(err));
            for (const ctg_exec::Issue& q : e->issue) {
                const int rc = launch_issue(e, q, nb, e->stream);
                if (rc != CTG_OK) return rc;
            }
        }
        if (n_full) HIP_TRY(hipEventRecord(e->ev_ids, e->stream));
        for (int64_t sid : rest) {
            hipError_t err = launch_prologue(e->meta, e->d_state, e->d_soff, sid, e->stream, 1, 1);
            if (err != hipSuccess) return fail(CTG_E_HIP, "prologue launch failed: %s", hipGetErrorString(err));
            for (const ctg_exec::Issue& q : e->issue) {
                const int rc = launch_issue(e, q, 1, e->stream);
                if (rc != CTG_OK) return rc;
            }
        }
        e->warm = true;
        return CTG_OK;
    }
    for (const auto& ks : order) {
        const bool fresh = ks.first != e->group_key;
        hipError_t err = launch_prologue(e->meta, e->d_state, e->d_soff, ks.second, e->stream, 1, 1);
        if (err != hipSuccess) return fail(CTG_E_HIP, "prologue launch failed: %s", hipGetErrorString(err));
        e->group_key = -1;   // (until the shared steps of this key are all launched)
        for (const ctg_exec::Issue& q : fresh ? e->issue : e->issue_reuse) {
            const int rc = launch_issue(e, q, 1, e->stream);
            if (rc != CTG_OK) return rc;
        }
        e->group_key = ks.first;
    }
    e->warm = true;
    return CTG_OK;
}
int run_grouped(ctg_exec* e, const std::vector<int64_t>& ids) { return run_grouped(e, ids.data(), ids.size()); }

// Slice-invariant steps (no sliced input below them): once per upload.  Their
// strip_exponent factors are computed here too and kept across slices.
int run_invariants(ctg_exec* e) {
    if (e->invariants_ready) return CTG_OK;
    const ctg_plan* p = e->plan;
    bool any = false;
    for (int64_t s = 0; s < p->n_steps; ++s) any = any || e->invariant[s];
    if (any) {
        // invariant operands are never slice dependent, but kernels read *soff
        hipError_t err = launch_prologue(e->meta, e->d_state, e->d_soff, 0, e->stream);
        if (err != hipSuccess)
            return fail(CTG_E_HIP, "prologue launch failed: %s", hipGetErrorString(err));
        if (e->strip)
            HIP_TRY(hipMemsetAsync(e->d_fac, 0, p->n_steps * sizeof(double), e->stream));
        for (int64_t s = 0; s < p->n_steps; ++s) {
            if (!e->invariant[s]) continue;
            const int rc = launch_step(e, s, e->stream);
            if (rc != CTG_OK) return rc;
        }
    }
    e->invariants_ready = true;
    return CTG_OK;
}

}  // namespace

extern "C" {

int ctg_abi_version(void) { return CTG_ABI_VERSION; }

const char* ctg_last_error(void) { return g_err.c_str(); }

// (shared with the host-only sources of the library; not part of the ABI)
__attribute__((visibility("hidden"))) void ctg_set_error_(const char* msg) { g_err = msg ? msg : ""; }

int ctg_plan_create(const ctg_plan_desc* d, ctg_plan** out) {
    if (!d || !out) return fail(CTG_E_INVALID, "null argument");
    if (d->n_inputs < 1 || d->n_steps < 0 || d->n_table_words < 1 || d->n_sliced < 0)
        return fail(CTG_E_INVALID, "bad plan sizes");
    if (!d->input_sizes || !d->input_offsets || !d->steps || !d->tables)
        return fail(CTG_E_INVALID, "null plan array");
    if (d->n_sliced > 0 && (!d->slice_sizes || !d->slice_fixed || !d->slice_strides))
        return fail(CTG_E_INVALID, "null slice array");
    ctg_plan* p = new ctg_plan();
    p->dtype = d->dtype;
    p->n_inputs = d->n_inputs;
    p->input_sizes.assign(d->input_sizes, d->input_sizes + d->n_inputs);
    p->input_offsets.assign(d->input_offsets, d->input_offsets + d->n_inputs);
    p->inputs_elems = d->inputs_elems;
    p->arena_elems = d->arena_elems;
    p->result_elems = d->result_elems;
    p->n_steps = d->n_steps;
    p->steps.assign(d->steps, d->steps + d->n_steps * STEP_WORDS);
    p->tables.assign(d->tables, d->tables + d->n_table_words);
    p->n_sliced = d->n_sliced;
    if (d->n_sliced) {
        p->slice_sizes.assign(d->slice_sizes, d->slice_sizes + d->n_sliced);
        p->slice_fixed.assign(d->slice_fixed, d->slice_fixed + d->n_sliced);
        p->slice_strides.assign(d->slice_strides,
                                d->slice_strides + (d->n_inputs + 1) * d->n_sliced);
        if (d->slice_group) p->slice_group.assign(d->slice_group, d->slice_group + d->n_sliced);
    }
    p->slice_group.resize(p->n_sliced, 0);
    if (p->inputs_elems < 1 || p->arena_elems < 1 || p->result_elems < 1) {
        delete p;
        return fail(CTG_E_INVALID, "empty buffer in plan");
    }
    int rc = validate_plan(p);
    if (rc == CTG_OK) rc = ctg_lds_validate(p);
    if (rc != CTG_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return CTG_OK;
}

int ctg_plan_destroy(ctg_plan* plan) {
    delete plan;
    return CTG_OK;
}

int ctg_plan_nslices(const ctg_plan* plan, int64_t* nslices) {
    if (!plan || !nslices) return fail(CTG_E_INVALID, "null argument");
    *nslices = plan->nslices;
    return CTG_OK;
}

int ctg_plan_workspace_bytes(const ctg_plan* p, int64_t bytes[4]) {
    if (!p || !bytes) return fail(CTG_E_INVALID, "null argument");
    const int64_t isz = kItemSize[p->dtype];
    bytes[0] = p->inputs_elems * isz;
    bytes[1] = p->arena_elems * isz;
    bytes[2] = p->result_elems * isz;
    bytes[3] = (int64_t)p->tables.size() * 8 + kScratchBytes +   /* (x up to 8 when slices are batched) */
               8 * (3 + (p->n_inputs + 1) * (1 + p->n_sliced) + 2 * p->n_sliced);
    return CTG_OK;
}

int ctg_exec_destroy(ctg_exec* e) {
    if (!e) return CTG_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    if (e->gstream) (void)hipStreamSynchronize(e->gstream);
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->gstream) (void)hipStreamDestroy(e->gstream);
    if (e->d_inputs) (void)hipFree(e->d_inputs);
    if (e->d_arena) (void)hipFree(e->d_arena);
    if (e->d_result && e->owns_result) (void)hipFree(e->d_result);
    if (e->d_wide) (void)hipFree(e->d_wide);
    if (e->d_in_tab) (void)hipFree(e->d_in_tab);
    if (e->d_inscale) (void)hipFree(e->d_inscale);
    if (e->d_tables) (void)hipFree(e->d_tables);
    if (e->d_misc) (void)hipFree(e->d_misc);
    if (e->d_scratch) (void)hipFree(e->d_scratch);
    if (e->d_ord) (void)hipFree(e->d_ord);
    if (e->d_lane) (void)hipFree(e->d_lane);
    if (e->d_ord_b) (void)hipFree(e->d_ord_b);
    if (e->d_lane_b) (void)hipFree(e->d_lane_b);
    if (e->d_group_items) (void)hipFree(e->d_group_items);
    if (e->d_fast_items) (void)hipFree(e->d_fast_items);
    if (e->d_stem_max) (void)hipFree(e->d_stem_max);
    if (e->d_smax_zero) (void)hipFree(e->d_smax_zero);
    if (e->d_lds_comps) (void)hipFree(e->d_lds_comps);
    if (e->d_lds_blob) (void)hipFree(e->d_lds_blob);
    if (e->h_ids) (void)hipHostFree(e->h_ids);
    if (e->ev_ids) (void)hipEventDestroy(e->ev_ids);
    if (e->d_fac) (void)hipFree(e->d_fac);
    if (e->d_counted) (void)hipFree(e->d_counted);
    if (e->d_fac_zero) (void)hipFree(e->d_fac_zero);
    if (e->d_strip) (void)hipFree(e->d_strip);
    delete e;
    return CTG_OK;
}

int ctg_exec_create(const ctg_plan* p, int device, void* stream, void* ext_result, ctg_exec** out) {
    if (!p || !out) return fail(CTG_E_INVALID, "null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(CTG_E_INVALID, "device %d not available (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    ctg_exec* e = new ctg_exec();
    e->plan = p;
    e->device = device;
    e->stream = (hipStream_t)stream;
    const int64_t isz = kItemSize[p->dtype];
    auto bail = [&](int rc) {
        std::string keep = g_err;
        ctg_exec_destroy(e);
        g_err = keep;
        return rc;
    };
#define HIP_TRY_E(expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return bail(fail(_e == hipErrorOutOfMemory ? CTG_E_NOMEM : CTG_E_HIP, "%s failed: %s", \
                             #expr, hipGetErrorString(_e)));                                    \
    } while (0)
    // Slice batching: small slices are launch-bound (a slice of the Sycamore m10 tree is
    // 170 launches of a few microseconds), so up to `batch` slices of a run go through
    // every launch together, each in its own replica of the arena.  Wide trees (one slice
    // = tens of GiB) keep batch = 1.  CTG_SLICE_BATCH caps the count (1 = off),
    // CTG_SLICE_BATCH_MIB the memory spent on replicas.
    {
        int64_t cap = 64, mib = 8192;
        {
            // (at most a quarter of what the device has free right now)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                mib = std::min<int64_t>(mib, (int64_t)(free_b >> 22));
            else
                (void)hipGetLastError();
        }
        if (const char* v = getenv("CTG_SLICE_BATCH")) cap = atoll(v);
        if (const char* v = getenv("CTG_SLICE_BATCH_MIB")) mib = atoll(v);
        int64_t b = std::min<int64_t>(cap, p->nslices);
        const int64_t per = std::max<int64_t>(p->arena_elems * isz, 1);
        b = std::min<int64_t>(b, (mib << 20) / per);
        e->batch = (int)std::max<int64_t>(b, 1);
        e->group_d = 0;
        if (p->has_groups) {
            // Slice groups in batched launches: a launch carries whole groups, z = group * d + member;
            // the shared steps go out once per group (nz / d), what the others read of them sits with
            // the group's first slice (StepArgs.zqA / zqB).  Fused stem steps do not take part (their
            // kernel has no z quantum): such plans contract slice by slice, and so does one whose group
            // does not fit a launch.
            int64_t d = 1;
            for (int64_t j = 0; j < p->n_sliced; ++j)
                if (p->slice_group[j] && p->slice_fixed[j] < 0) d *= p->slice_sizes[j];
            bool stems = false;
            for (int64_t s = 0; s < p->n_steps; ++s) stems = stems || p->steps[s * STEP_WORDS + W_KIND] == KIND_STEM2;
            if (stems || d > e->batch || d < 2 || env_on("CTG_NO_BATCHED_GROUPS")) {
                e->batch = 1;
            } else {
                e->batch = (int)(e->batch / d * d);
                e->group_d = (int)d;
            }
        }
        // what the plan alone says about batching (no environment, no free-memory
        // query): the k-splits of its steps are chosen for launches of this many
        // slices, so that they are a function of the plan and a result never depends
        // on how a run is cut into launches
        const int64_t nominal = std::min<int64_t>({(int64_t)64, p->nslices, ((int64_t)8192 << 20) / per});
        e->batch_nominal = (int)std::max<int64_t>(nominal, 1);
        if (env_on("CTG_SPLITK_PER_SLICE")) e->batch_nominal = 1;   // (experiments: round-2 rule)
    }
    HIP_TRY_E(hipMalloc((void**)&e->d_inputs, p->inputs_elems * isz));
    HIP_TRY_E(hipMalloc((void**)&e->d_arena, p->arena_elems * isz * e->batch));
    if (env_on("CTG_ARENA_DEBUG"))
        fprintf(stderr, "arena %p (%lld bytes)\n", e->d_arena, (long long)(p->arena_elems * isz * e->batch));
    place_arena(e);
    if (e->d_arena == nullptr) HIP_TRY_E(hipMalloc((void**)&e->d_arena, p->arena_elems * isz * e->batch));   // (a failed re-allocation)
    if (ext_result) {
        e->d_result = (char*)ext_result;
    } else {
        HIP_TRY_E(hipMalloc((void**)&e->d_result, p->result_elems * isz));
        e->owns_result = true;
    }
    if ((p->dtype == CTG_F32 || p->dtype == CTG_C64) && !env_on("CTG_NO_PRESCALE")) {
        // single-precision trees: inputs far from 1 lose an exact power of two at upload (prescale_inputs_kernel)
        std::vector<int64_t> tab(2 * p->n_inputs);
        for (int64_t i = 0; i < p->n_inputs; ++i) {
            tab[i] = p->input_offsets[i];
            tab[p->n_inputs + i] = p->input_sizes[i];
        }
        HIP_TRY_E(hipMalloc((void**)&e->d_in_tab, tab.size() * 8));
        HIP_TRY_E(hipMemcpy(e->d_in_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY_E(hipMalloc((void**)&e->d_inscale, 2 * sizeof(double) + sizeof(int64_t)));
        const double init[3] = {1.0, 0.0, 0.0};
        HIP_TRY_E(hipMemcpy(e->d_inscale, init, sizeof(init), hipMemcpyHostToDevice));
    }
    {
        // float / complex64 results of a tree with more than one slice: the slices are summed in double
        // precision (accum_kernel)
        bool sums = false;
        for (int64_t s = 0; s < p->n_steps; ++s) sums = sums || p->steps[s * STEP_WORDS + W_KIND] == KIND_ACCUM;
        sums = sums && p->nslices > 1;
        if (sums && (p->dtype == CTG_F32 || p->dtype == CTG_C64) && !env_on("CTG_NO_WIDE_SUM")) {
            HIP_TRY_E(hipMalloc((void**)&e->d_wide, p->result_elems * 2 * isz));
            HIP_TRY_E(hipMemsetAsync(e->d_wide, 0, p->result_elems * 2 * isz, e->stream));
        }
    }
    HIP_TRY_E(hipMalloc((void**)&e->d_tables, p->tables.size() * 8));
    // (split heuristics are always computed with kScratchBytes; a batching executor gets
    // more room so that more slices of a split-K / k-reduction step fit one launch)
    // (allocated by the first launch that needs it -- split-K slabs, k-reduction partials,
    // k-streaming tiles: an executor of a small one-shot expression never does, and 64 of
    // those in the expression cache would pin 4 GiB of scratch for nothing)
    e->scratch_total = kScratchBytes * std::min<int64_t>(std::max(e->batch, 1), 8);
    const int64_t n_leaves = p->n_inputs + 1;
    const int64_t misc_words = 3 + n_leaves * e->batch + 2 * p->n_sliced + n_leaves * p->n_sliced + e->batch;
    HIP_TRY_E(hipMalloc((void**)&e->d_misc, misc_words * 8));
    std::vector<int64_t> misc(misc_words, 0);
    int64_t* cur = e->d_misc;
    e->d_state = cur;
    cur += 2;
    e->d_zero = cur;
    cur += 1;
    e->d_soff = cur;
    cur += n_leaves * e->batch;
    int64_t* d_sizes = cur;
    cur += p->n_sliced;
    int64_t* d_fixed = cur;
    cur += p->n_sliced;
    int64_t* d_strides = cur;
    e->d_batch_ids = d_strides + n_leaves * p->n_sliced;   // (the last `batch` words)
    for (int64_t j = 0; j < p->n_sliced; ++j) {
        misc[(d_sizes - e->d_misc) + j] = p->slice_sizes[j];
        misc[(d_fixed - e->d_misc) + j] = p->slice_fixed[j];
    }
    for (int64_t i = 0; i < n_leaves * p->n_sliced; ++i)
        misc[(d_strides - e->d_misc) + i] = p->slice_strides[i];
    misc[1] = 1;
    HIP_TRY_E(hipMemcpy(e->d_misc, misc.data(), misc_words * 8, hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(e->d_tables, p->tables.data(), p->tables.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemsetAsync(e->d_inputs, 0, p->inputs_elems * isz, e->stream));
    HIP_TRY_E(hipMemsetAsync(e->d_result, 0, p->result_elems * isz, e->stream));
    e->meta = SliceMeta{n_leaves, p->n_sliced, d_sizes, d_fixed, d_strides, nullptr, nullptr, 0, nullptr, nullptr, 0};
    {
        std::vector<double> fac(p->n_steps + 1, 0.0);
        fac[p->n_steps] = 1.0;
        std::vector<int32_t> counted(std::max<int64_t>(p->n_steps, 1), 0);
        std::vector<int32_t> fac_zero(std::max<int64_t>(p->n_steps, 1), 0);
        e->invariant.assign(p->n_steps, 0);
        e->grouped.assign(p->n_steps, 0);
        for (int64_t st = 0; st < p->n_steps; ++st)
            e->grouped[st] = p->has_groups && p->steps[st * STEP_WORDS + W_INVARIANT] == 2 &&
                             p->steps[st * STEP_WORDS + W_KIND] != KIND_ACCUM;
        for (int64_t st = 0; st < p->n_steps; ++st) {
            const int64_t* r = &p->steps[st * STEP_WORDS];
            e->invariant[st] = r[W_INVARIANT] == 1 && r[W_KIND] != KIND_ACCUM;
            if (r[W_KIND] == KIND_PAIR || r[W_KIND] == KIND_STEM2) {
                counted[st] = 1;
                fac_zero[st] = e->invariant[st] ? 0 : 1;
                e->root_step = st;  // the last pair step produces the slice output
            }
        }
        {
            // fp16 x 2 stem kernels: per step the largest element it recorded | of its big operand
            bool stems = false;
            // (three banks of n_steps: recorded by the step | max-abs pass over its operand A | ... B)
            const int64_t nb = std::max<int64_t>(e->batch, 1);
            e->plan_steps = p->n_steps;
            // (what the slice prologue resets: the records of bank 0 of the steps that can record -- stem launches and
            // matrix-core pair steps --; the max-abs banks are cleared where a pass is launched)
            std::vector<int32_t> sz((size_t)(3 * std::max<int64_t>(p->n_steps, 1) * nb * kMaxSub), 0);
            for (int64_t st = 0; st < p->n_steps; ++st) {
                const int64_t* rs = &p->steps[st * STEP_WORDS];
                stems = stems || rs[W_KIND] == KIND_STEM2;
                const bool can = rs[W_KIND] == KIND_STEM2 || (rs[W_KIND] == KIND_PAIR && rs[W_KERNEL] == KERNEL_MFMA);
                // (slice-invariant steps and what a slice group shares keep their record across slices)
                if (can && !e->invariant[st] && !e->grouped[st])
                    for (int64_t z = 0; z < nb * kMaxSub; ++z) sz[(size_t)(st * nb * kMaxSub + z)] = 1;
            }
            e->stem_h2_ran.assign(p->n_steps, 0);
            if (e->wave_member.size() != (size_t)p->n_steps) e->wave_member.assign(p->n_steps, 0);
            (void)stems;
            if (p->dtype == CTG_C64) {   // (plans without a stem may still have long tiled steps)
                HIP_TRY_E(hipMalloc((void**)&e->d_stem_max, sz.size() * sizeof(float)));
                HIP_TRY_E(hipMemset(e->d_stem_max, 0, sz.size() * sizeof(float)));
                HIP_TRY_E(hipMalloc((void**)&e->d_smax_zero, sz.size() * sizeof(int32_t)));
                HIP_TRY_E(hipMemcpy(e->d_smax_zero, sz.data(), sz.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                e->meta.smax = e->d_stem_max;
                e->meta.smax_zero = e->d_smax_zero;
                e->meta.n_smax = (int64_t)(sz.size() / 3);   // (bank 0 only: see above)
            }
        }
        HIP_TRY_E(hipMalloc((void**)&e->d_fac_zero, fac_zero.size() * sizeof(int32_t)));
        HIP_TRY_E(hipMemcpy(e->d_fac_zero, fac_zero.data(), fac_zero.size() * sizeof(int32_t),
                            hipMemcpyHostToDevice));
        HIP_TRY_E(hipMalloc((void**)&e->d_fac, fac.size() * sizeof(double)));
        HIP_TRY_E(hipMalloc((void**)&e->d_counted, counted.size() * sizeof(int32_t)));
        HIP_TRY_E(hipMalloc((void**)&e->d_strip, sizeof(StripState)));
        HIP_TRY_E(hipMemcpy(e->d_fac, fac.data(), fac.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY_E(hipMemcpy(e->d_counted, counted.data(), counted.size() * sizeof(int32_t),
                            hipMemcpyHostToDevice));
    }
    // Replaying a captured slice graph measured SLOWER than eager launches on
    // ROCm 7.2 / MI355X (C2: 622 vs 523 us per contraction in round 1, 243 vs 183 us
    // with the 18 shared launches of round 2; m20: 92.0 vs 91.4
    // ms per slice): the host already runs ahead of the device and the tiny
    // kernels are bound by their dependent-load latency, not by launch cost.
    // The path stays available behind CTG_GRAPH=1.
    e->graph_off = !env_on("CTG_GRAPH");
    // arithmetic of the stem kernels: fp16 x 2 (round 6) unless CTG_STEM_ARITH names another one
    // (fp32 | bf16x3 | fp16x2, or 0 | 1 | 2); ctg_exec_set_stem_arithmetic changes it later
    if (const char* v = getenv("CTG_STEM_ARITH")) {
        const std::string a(v);
        if (a == "fp32" || a == "0") e->stem_arith = 0;
        else if (a == "bf16x3" || a == "1") e->stem_arith = 1;
        else if (a == "fp16x2" || a == "2") e->stem_arith = 2;
        e->stem_bf16x3 = e->stem_arith ? 1 : 0;
    }
    resolve_args(e);
    {
        int rc = build_hints(e);
        if (rc == CTG_OK) rc = build_groups(e);
        // the scratch buffer (split-K slabs, k-reduction partials, k-streaming tiles) is
        // allocated HERE when any step of the plan writes to it -- not inside the first run,
        // where the arena already holds the device memory and an out-of-memory error would
        // bypass the caller's evict-and-retry around executor creation; a plan without such
        // a step (every small one-shot expression) never allocates it
        for (int64_t st = 0; rc == CTG_OK && st < p->n_steps; ++st)
            if (step_needs_scratch(e, st)) {
                rc = ensure_scratch(e);
                break;
            }
        if (rc != CTG_OK) return bail(rc);
    }
    *out = e;
    return CTG_OK;
#undef HIP_TRY_E
}

int ctg_exec_set_stream(ctg_exec* e, void* stream) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    if ((hipStream_t)stream == e->stream) return CTG_OK;
    HIP_TRY(hipSetDevice(e->device));
    // work already enqueued on the old stream must be ordered before anything the
    // new stream does with the executor's buffers
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stream = (hipStream_t)stream;
    return CTG_OK;
}

int ctg_exec_upload_inputs_host(ctg_exec* e, const void* const* ptrs) {
    if (!e || !ptrs) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const ctg_plan* p = e->plan;
    const int64_t isz = kItemSize[p->dtype];
    // stage all inputs in one pinned-free contiguous host buffer -> one copy
    std::vector<char> staging((size_t)(p->inputs_elems * isz), 0);
    for (int64_t i = 0; i < p->n_inputs; ++i) {
        if (!ptrs[i]) return fail(CTG_E_INVALID, "input %lld is null", (long long)i);
        memcpy(staging.data() + p->input_offsets[i] * isz, ptrs[i], p->input_sizes[i] * isz);
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->d_inputs, staging.data(), staging.size(), hipMemcpyHostToDevice));
    if (e->d_inscale)
        HIP_TRY(launch_prescale_inputs(p->dtype, e->d_inputs, e->d_in_tab, e->d_in_tab + p->n_inputs, p->n_inputs,
                                       (int*)(e->d_inscale + 2), e->d_inscale, e->stream));
    e->invariants_ready = false;
    e->group_key = -1;
    return CTG_OK;
}

int ctg_exec_upload_inputs_device(ctg_exec* e, const void* const* ptrs) {
    if (!e || !ptrs) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const ctg_plan* p = e->plan;
    const int64_t isz = kItemSize[p->dtype];
    for (int64_t i = 0; i < p->n_inputs; ++i) {
        if (!ptrs[i]) return fail(CTG_E_INVALID, "input %lld is null", (long long)i);
        HIP_TRY(hipMemcpyAsync(e->d_inputs + p->input_offsets[i] * isz, ptrs[i],
                               p->input_sizes[i] * isz, hipMemcpyDeviceToDevice, e->stream));
    }
    if (e->d_inscale)
        HIP_TRY(launch_prescale_inputs(p->dtype, e->d_inputs, e->d_in_tab, e->d_in_tab + p->n_inputs, p->n_inputs,
                                       (int*)(e->d_inscale + 2), e->d_inscale, e->stream));
    e->invariants_ready = false;
    e->group_key = -1;
    return CTG_OK;
}

int ctg_exec_zero_result(ctg_exec* e) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemsetAsync(e->d_result, 0, e->plan->result_elems * kItemSize[e->plan->dtype],
                           e->stream));
    if (e->d_wide)
        HIP_TRY(hipMemsetAsync(e->d_wide, 0, e->plan->result_elems * 2 * kItemSize[e->plan->dtype], e->stream));
    StripState init{};
    init.E = -HUGE_VAL;
    init.e_slice = -HUGE_VAL;
    init.coefM = 1.0;
    init.coefm = 0.0;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->d_strip, &init, sizeof(init), hipMemcpyHostToDevice));
    return CTG_OK;
}

int ctg_exec_set_stem_arithmetic(ctg_exec* e, int mode) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    if (mode < 0 || mode > 2) return fail(CTG_E_INVALID, "stem arithmetic %d (0 fp32, 1 bf16 x 3, 2 fp16 x 2)", mode);
    const int bf16x3 = mode ? 1 : 0;
    if (bf16x3 == e->stem_bf16x3 && mode == e->stem_arith) return CTG_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stem_arith = mode;
    e->stem_bf16x3 = bf16x3;
    for (auto& q : e->stem_args) q.bf3 = bf16x3;
    for (auto& a : e->args) a.bf3 = bf16x3;
    // (a captured slice graph holds the kernels of the old arithmetic, and the slice-invariant steps were computed
    // with it)
    if (e->gexec) {
        (void)hipGraphExecDestroy(e->gexec);
        e->gexec = nullptr;
    }
    e->invariants_ready = false;
    e->group_key = -1;
    return CTG_OK;
}

int ctg_exec_set_strip_exponent(ctg_exec* e, int strip, int check_zero) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    strip = strip ? 1 : 0;
    check_zero = check_zero ? 1 : 0;
    if (strip == e->strip && check_zero == e->check_zero) return CTG_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->strip = strip;
    e->check_zero = check_zero;
    e->meta.fac = strip ? e->d_fac : nullptr;
    e->meta.fac_zero = e->d_fac_zero;
    e->meta.n_fac = strip ? e->plan->n_steps : 0;
    e->invariants_ready = false;  // their stored scale changes with the option
    e->group_key = -1;
    resolve_args(e);
    {
        const int rc = build_groups(e);
        if (rc != CTG_OK) return rc;
    }
    // a captured slice graph embeds the old arguments
    if (e->gexec) {
        (void)hipGraphExecDestroy(e->gexec);
        e->gexec = nullptr;
    }
    return CTG_OK;
}

int ctg_exec_get_exponent(ctg_exec* e, double* exponent, int* zero) {
    if (!e || !exponent) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    StripState st{};
    HIP_TRY(hipMemcpy(&st, e->d_strip, sizeof(st), hipMemcpyDeviceToHost));
    *exponent = st.E;
    if (zero) *zero = st.zero;
    return CTG_OK;
}

int ctg_exec_run_slices(ctg_exec* e, int64_t first, int64_t count, int64_t stride) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    const ctg_plan* p = e->plan;
    if (count < 0 || stride < 1 || first < 0 ||
        (count > 0 && first + (count - 1) * stride >= p->nslices))
        return fail(CTG_E_INVALID, "slice range [%lld + i*%lld, i<%lld) outside [0, %lld)",
                    (long long)first, (long long)stride, (long long)count, (long long)p->nslices);
    if (count == 0) return CTG_OK;
    HIP_TRY(hipSetDevice(e->device));
    {
        const int rc = run_invariants(e);
        if (rc != CTG_OK) return rc;
    }
    // slices sid, sid + stride, ... (nb of them) through one launch sequence
    auto eager = [&](int64_t sid, int nb = 1) -> int {
        // (an unsliced tree has one set of leaf offsets -- all zero: computed once)
        if (p->n_sliced > 0 || e->strip || !e->soff_static) {
            hipError_t err = launch_prologue(e->meta, e->d_state, e->d_soff, sid, e->stream, nb, stride);
            if (err != hipSuccess)
                return fail(CTG_E_HIP, "prologue launch failed: %s", hipGetErrorString(err));
            e->soff_static = p->n_sliced == 0 && !e->strip;
        }
        for (const ctg_exec::Issue& q : e->issue) {
            const int rc = launch_issue(e, q, nb, e->stream);
            if (rc != CTG_OK) return rc;
        }
        return CTG_OK;
    };
    if (p->has_groups && !e->strip) {
        // every slice of the tree: group by group (host memory bounded by the chunk, not by nslices)
        if (first == 0 && stride == 1 && count == p->nslices) return ctg_exec_run_share(e, 0, 1, 0, -1);
        // any other range: in chunks of ids (a group cut by a chunk boundary is computed slice by slice)
        const int64_t chunk = (int64_t)1 << 20;
        std::vector<int64_t> ids;
        for (int64_t k0 = 0; k0 < count; k0 += chunk) {
            const int64_t n = std::min(chunk, count - k0);
            ids.resize((size_t)n);
            for (int64_t k = 0; k < n; ++k) ids[(size_t)k] = first + (k0 + k) * stride;
            const int rc = run_grouped(e, ids);
            if (rc != CTG_OK) return rc;
        }
        return CTG_OK;
    }
    int64_t i = 0;
    // (strip_exponent keeps one scale per step and slice: no batching there)
    const int64_t batch = e->strip ? 1 : e->batch;
    if (batch > 1 && count > 1) {
        for (; i < count; i += batch) {
            const int rc = eager(first + i * stride, (int)std::min<int64_t>(batch, count - i));
            if (rc != CTG_OK) return rc;
        }
        e->warm = true;
        return CTG_OK;
    }
    if (!e->graph_off && (count >= 2 || e->warm)) {
        if (!e->warm) {  // first slice eagerly: lets the launchers do their one-time setup
            const int rc = eager(first);
            if (rc != CTG_OK) return rc;
            e->warm = true;
            i = 1;
        }
        if (!e->gexec) {
            // capture one slice; any failure just disables the graph path
            bool ok = hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming) == hipSuccess;
            hipGraph_t graph = nullptr;
            if (ok) ok = hipStreamBeginCapture(e->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                bool launched = launch_prologue(e->meta, e->d_state, e->d_soff, -1, e->gstream) == hipSuccess;
                for (size_t q = 0; launched && q < e->issue.size(); ++q)
                    launched = launch_issue(e, e->issue[q], 1, e->gstream) == CTG_OK;
                ok = hipStreamEndCapture(e->gstream, &graph) == hipSuccess && launched && graph;
            }
            if (ok) ok = hipGraphInstantiate(&e->gexec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (!ok) {
                (void)hipGetLastError();
                e->gexec = nullptr;
                e->graph_off = true;
            }
        }
        if (e->gexec && i < count) {
            HIP_TRY(hipEventRecord(e->ev_in, e->stream));
            HIP_TRY(hipStreamWaitEvent(e->gstream, e->ev_in, 0));
            hipError_t err = launch_set_state(e->d_state, first + i * stride, stride, e->gstream);
            if (err != hipSuccess)
                return fail(CTG_E_HIP, "state launch failed: %s", hipGetErrorString(err));
            for (; i < count; ++i) HIP_TRY(hipGraphLaunch(e->gexec, e->gstream));
            HIP_TRY(hipEventRecord(e->ev_out, e->gstream));
            HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_out, 0));
            return CTG_OK;
        }
    }
    for (; i < count; ++i) {
        const int rc = eager(first + i * stride);
        if (rc != CTG_OK) return rc;
    }
    e->warm = true;
    return CTG_OK;
}

int ctg_exec_run_slice_list(ctg_exec* e, const int64_t* ids, int64_t n) {
    if (!e || (n > 0 && !ids)) return fail(CTG_E_INVALID, "null argument");
    const ctg_plan* p = e->plan;
    if (n < 0) return fail(CTG_E_INVALID, "negative slice count");
    for (int64_t k = 0; k < n; ++k)
        if (ids[k] < 0 || ids[k] >= p->nslices)
            return fail(CTG_E_INVALID, "slice id %lld outside [0, %lld)", (long long)ids[k], (long long)p->nslices);
    if (n == 0) return CTG_OK;
    if (p->has_groups && !e->strip) {
        HIP_TRY(hipSetDevice(e->device));
        const int rc = run_invariants(e);
        if (rc != CTG_OK) return rc;
        return run_grouped(e, std::vector<int64_t>(ids, ids + n));
    }
    for (int64_t k = 0; k < n; ++k) {
        const int rc = ctg_exec_run_slices(e, ids[k], 1, 1);
        if (rc != CTG_OK) return rc;
    }
    return CTG_OK;
}

int ctg_plan_share_units(const ctg_plan* p, int64_t rank, int64_t world, int64_t* units, int64_t* slices_per_unit) {
    if (!p || !units) return fail(CTG_E_INVALID, "null argument");
    Share sh;
    const int rc = plan_share(p, rank, world, &sh);
    if (rc != CTG_OK) return rc;
    *units = sh.units;
    if (slices_per_unit) *slices_per_unit = sh.gsize;
    return CTG_OK;
}

int ctg_plan_share_slice_ids(const ctg_plan* p, int64_t rank, int64_t world, int64_t unit_first, int64_t unit_count,
                             int64_t* ids) {
    if (!p || (unit_count > 0 && !ids)) return fail(CTG_E_INVALID, "null argument");
    Share sh;
    const int rc = plan_share(p, rank, world, &sh);
    if (rc != CTG_OK) return rc;
    if (unit_count < 0) unit_count = sh.units - unit_first;
    if (unit_first < 0 || unit_count < 0 || unit_first + unit_count > sh.units)
        return fail(CTG_E_INVALID, "units [%lld, +%lld) outside the %lld of rank %lld", (long long)unit_first,
                    (long long)unit_count, (long long)sh.units, (long long)rank);
    const std::vector<SliceDigit> digits = slice_digits(p);
    std::vector<int64_t> tmp;
    for (int64_t u = 0; u < unit_count; ++u) {
        tmp.clear();
        group_members(digits, rank + (unit_first + u) * world, tmp);
        for (size_t i = 0; i < tmp.size(); ++i) ids[u * sh.gsize + (int64_t)i] = tmp[i];
    }
    return CTG_OK;
}

int ctg_exec_run_share(ctg_exec* e, int64_t rank, int64_t world, int64_t unit_first, int64_t unit_count) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    const ctg_plan* p = e->plan;
    Share sh;
    {
        const int rc = plan_share(p, rank, world, &sh);
        if (rc != CTG_OK) return rc;
    }
    if (unit_count < 0) unit_count = sh.units - unit_first;
    if (unit_first < 0 || unit_count < 0 || unit_first + unit_count > sh.units)
        return fail(CTG_E_INVALID, "units [%lld, +%lld) outside the %lld of rank %lld", (long long)unit_first,
                    (long long)unit_count, (long long)sh.units, (long long)rank);
    if (unit_count == 0) return CTG_OK;
    // no group indices: the round-robin of contract_mpi (core.py:4070) as one strided range
    if (sh.gsize == 1) return ctg_exec_run_slices(e, rank + unit_first * world, unit_count, world);
    HIP_TRY(hipSetDevice(e->device));
    {
        const int rc = run_invariants(e);
        if (rc != CTG_OK) return rc;
    }
    // whole groups, a chunk of them at a time (a multiple of the launch batch when launches are batched)
    const std::vector<SliceDigit> digits = slice_digits(p);
    const int64_t want = std::max<int64_t>((int64_t)e->batch * 64, 8192);
    const int64_t per_chunk = std::max<int64_t>(want / sh.gsize, 1);
    std::vector<int64_t> ids;
    for (int64_t u0 = 0; u0 < unit_count; u0 += per_chunk) {
        const int64_t nu = std::min(per_chunk, unit_count - u0);
        ids.clear();
        for (int64_t u = 0; u < nu; ++u) group_members(digits, rank + (unit_first + u0 + u) * world, ids);
        if (p->has_groups && !e->strip) {
            const int rc = run_grouped(e, ids);
            if (rc != CTG_OK) return rc;
        } else {
            // (strip_exponent keeps a scale per step and slice, or nothing is shared: slice by slice)
            for (int64_t sid : ids) {
                const int rc = ctg_exec_run_slices(e, sid, 1, 1);
                if (rc != CTG_OK) return rc;
            }
        }
    }
    return CTG_OK;
}

int ctg_exec_launch_count(ctg_exec* e, int64_t* steps, int64_t* launches) {
    if (!e || !steps || !launches) return fail(CTG_E_INVALID, "null argument");
    int64_t ns = 0, nl = 0;
    for (const ctg_exec::Issue& q : e->issue) {
        if (q.cls == -2) {
            // (one launch of q.n LDS-resident subtrees: its steps are their members)
            for (int32_t c : e->lds_comp_of) ns += c >= q.item0 && c < q.item0 + q.n;
        } else {
            ns += q.n;
        }
        nl += 1;
    }
    *steps = ns;
    *launches = nl;
    return CTG_OK;
}

int ctg_exec_device_bytes(ctg_exec* e, int64_t* bytes) {
    if (!e || !bytes) return fail(CTG_E_INVALID, "null argument");
    const ctg_plan* p = e->plan;
    const int64_t isz = kItemSize[p->dtype];
    int64_t n = p->inputs_elems * isz + p->arena_elems * isz * std::max(e->batch, 1) + (int64_t)p->tables.size() * 8;
    if (e->owns_result) n += p->result_elems * isz;
    if (e->d_wide) n += p->result_elems * 2 * isz;
    if (e->d_scratch) n += e->scratch_total;
    *bytes = n;
    return CTG_OK;
}

int ctg_exec_slice_batch(ctg_exec* e, int64_t* batch) {
    if (!e || !batch) return fail(CTG_E_INVALID, "null argument");
    *batch = e->strip ? 1 : e->batch;
    return CTG_OK;
}

int ctg_exec_profile_slice(ctg_exec* e, int64_t slice_id, float* ms) {
    if (!e || !ms) return fail(CTG_E_INVALID, "null argument");
    const ctg_plan* p = e->plan;
    if (slice_id < 0 || slice_id >= p->nslices) return fail(CTG_E_INVALID, "slice id out of range");
    e->group_key = -1;   // (every step of this slice is launched: whatever a group shared before is overwritten)
    HIP_TRY(hipSetDevice(e->device));
    while ((int64_t)e->events.size() < p->n_steps + 1) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->events.push_back(ev);
    }
    {
        const int rc = run_invariants(e);
        if (rc != CTG_OK) return rc;
    }
    // (development: CTG_PROFILE_SLICES=n times every step with n slices in its launch,
    // as a batched run issues it -- step by step, without the wave-front groups)
    int nb = 1;
    if (const char* v = getenv("CTG_PROFILE_SLICES"))
        nb = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)atoll(v), (int64_t)(e->strip ? 1 : e->batch),
                                                          p->nslices - slice_id}));
    // (an executor that batches whole slice groups: the batch is the groups from slice_id's on, slice-in-batch
    // z = group * d + member as in run_grouped, and what a group shares goes out once per group)
    const int d = e->group_d > 1 ? e->group_d : 1;
    std::vector<int64_t> ids;
    if (d > 1 && nb > 1) {
        const std::vector<SliceDigit> digits = slice_digits(p);
        int64_t g = 0, mul = 1;
        for (const SliceDigit& dg : digits)
            if (!dg.group) {
                g += ((slice_id / dg.stride) % dg.size) * mul;
                mul *= dg.size;
            }
        for (const int64_t n_groups = p->nslices / d; g < n_groups && (int64_t)ids.size() + d <= nb; ++g)
            group_members(digits, g, ids);
        nb = ids.empty() ? 1 : (int)ids.size();
    }
    hipError_t err;
    if (nb > 1 && d > 1) {
        HIP_TRY(hipMemcpyAsync(e->d_batch_ids, ids.data(), nb * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
        err = launch_prologue(e->meta, e->d_state, e->d_soff, 0, e->stream, nb, 1, e->d_batch_ids);
    } else {
        err = launch_prologue(e->meta, e->d_state, e->d_soff, slice_id, e->stream, nb, 1);
    }
    if (err != hipSuccess) return fail(CTG_E_HIP, "prologue launch failed: %s", hipGetErrorString(err));
    HIP_TRY(hipEventRecord(e->events[0], e->stream));
    bool lds_done[2] = {false, false};
    for (int64_t s = 0; s < p->n_steps; ++s) {
        if (!e->invariant[s] && !e->lds_comp_of.empty() && e->lds_comp_of[s] >= 0) {
            // a member of an LDS-resident subtree: all subtrees of its class are ONE launch, timed under
            // the first member; the others read 0 ms
            const int cls = e->grouped[s] ? 0 : 1;
            if (!lds_done[cls] && p->steps[s * STEP_WORDS + W_KIND] == KIND_PAIR) {
                lds_done[cls] = true;
                const ctg_exec::Issue q{s, -2, e->lds_first[cls], e->lds_count[cls], (uint32_t)e->lds_count[cls], cls == 0};
                const int rc = launch_issue(e, q, nb, e->stream);
                if (rc != CTG_OK) return rc;
            }
        } else if (!e->invariant[s]) {  // invariant steps cost nothing per slice: 0 ms
            e->args[s].nz = (nb > 1 && d > 1 && e->grouped[s]) ? nb / d : nb;
            const int rc = launch_step(e, s, e->stream);
            e->args[s].nz = 1;
            if (rc != CTG_OK) return rc;
        }
        HIP_TRY(hipEventRecord(e->events[s + 1], e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int64_t s = 0; s < p->n_steps; ++s)
        HIP_TRY(hipEventElapsedTime(&ms[s], e->events[s], e->events[s + 1]));
    return CTG_OK;
}

int ctg_exec_step_kernel(ctg_exec* e, int64_t step, char* buf, int64_t buflen) {
    if (!e || !buf || buflen < 1) return fail(CTG_E_INVALID, "null argument");
    const ctg_plan* p = e->plan;
    if (step < 0 || step >= p->n_steps) return fail(CTG_E_INVALID, "step out of range");
    const int64_t* r = &p->steps[step * STEP_WORDS];
    char name[128];
    if (!e->lds_comp_of.empty() && e->lds_comp_of[step] >= 0) {
        snprintf(name, sizeof(name), "lds_run_kernel[%d]", (int)e->lds_comp_of[step]);
    } else if (r[W_KIND] == KIND_SINGLE) {
        snprintf(name, sizeof(name), "single_kernel");
    } else if (r[W_KIND] == KIND_ACCUM) {
        snprintf(name, sizeof(name), "accum_kernel");
    } else if (r[W_KIND] == KIND_STEM2) {
        // (the arithmetic of the step's NEXT launch: fp16 x 2 needs the producer of its big operand to be a stem
        // launch of the 16-bit pipe -- decided on the shapes here, as launch_step decides it on what ran)
        const int64_t prod = r[W_A_PROD];
        const bool prod16 = prod >= 0 && prod < p->n_steps &&
                            ((p->steps[prod * STEP_WORDS + W_KIND] == KIND_STEM2 &&
                              (stem2_uses_bf3(e->stem_args[prod]) || stem2h_uses_h2(e->stem_args[prod]))) ||
                             pair_records(e, prod));
        bool h2 = e->stem_arith == 2 && !e->strip && e->d_stem_max != nullptr && (prod16 || env_on("CTG_STEM_H2_ALL")) &&
                  stem2h_uses_h2(e->stem_args[step]);
        if (h2)
            if (const char* v = getenv("CTG_STEM_H2")) h2 = !(v[0] == '\0' || (v[0] == '0' && v[1] == '\0'));
        if (h2) stem2h_kernel_name(e->stem_args[step], name, sizeof(name));
        else stem2_kernel_name(e->stem_args[step], name, sizeof(name));
    } else if (r[W_KERNEL] == KERNEL_MFMA && p->dtype == CTG_C128) {
        snprintf(name, sizeof(name), "pair_mfma_c128_kernel");
    } else if (r[W_KERNEL] == KERNEL_MFMA && p->dtype != CTG_C64) {
        snprintf(name, sizeof(name), "pair_mfma_real_kernel<%s>", p->dtype == CTG_F32 ? "float" : "double");
    } else if (r[W_KERNEL] == KERNEL_MFMA) {
        const MfmaHints& h = e->hints[step];
        if (h.stream == 4)
            snprintf(name, sizeof(name), "pair_rowwise_kernel<%d>",
                     r[W_N] <= 4 ? 4 : (r[W_N] <= 8 ? 8 : (r[W_N] <= 12 ? 12 : (r[W_N] <= 16 ? 16 : (r[W_N] <= 24 ? 24 : 32)))));
        else if (h.stream == 3)
            snprintf(name, sizeof(name), "pair_skinny_kernel<%d,%d>", (int)r[W_K], (int)r[W_N]);
        else if (h.stream == 2)
            snprintf(name, sizeof(name), "pair_mfma_kstream_kernel<%d,%s>", h.bn / 16, h.vecA ? "true" : "false");
        else if (h.stream)
            snprintf(name, sizeof(name), "pair_mfma_stream_kernel<%d,%s,%s,%s,%d>", h.bn / 16,
                     (h.vecA && h.additive32) ? "true" : "false", h.additive32 ? "true" : "false",
                     r[W_K] < MFMA_BK ? "true" : "false", r[W_K] <= 4 ? 2 : (r[W_K] <= 8 ? 4 : 8));
        else if (h.bf3 && pair_bf16x3_on(e->args[step])) {
            bool h2 = tiled16_step(e, step) && e->stem_arith == 2 && h.splitk <= 1 && !e->grouped[step] &&
                      e->args[step].zqA <= 1 && e->args[step].zqB <= 1;
            if (h2)
                if (const char* v = getenv("CTG_PAIR_H2")) h2 = !(v[0] == '\0' || (v[0] == '0' && v[1] == '\0'));
            snprintf(name, sizeof(name), "%s<128,%d,16>,%s", h2 ? "pair_mfma_h2_kernel" : "pair_mfma_bf3_kernel", h.bn,
                     h.vecA ? "true" : "false");
        }
        else
            snprintf(name, sizeof(name), "%s<128,%d,16>,%s",
                     h.fast ? "pair_mfma_fast_kernel" : "pair_mfma_c64_kernel", h.bn,
                     h.vecA ? "true" : "false");
    } else {
        snprintf(name, sizeof(name), "pair_valu_kernel");
    }
    snprintf(buf, (size_t)buflen, "%s", name);
    return CTG_OK;
}

int ctg_exec_sync(ctg_exec* e) {
    if (!e) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CTG_OK;
}

int ctg_exec_result_ptr(ctg_exec* e, void** dev_ptr) {
    if (!e || !dev_ptr) return fail(CTG_E_INVALID, "null argument");
    *dev_ptr = e->d_result;
    return CTG_OK;
}

int ctg_exec_download_result(ctg_exec* e, void* host_out) {
    if (!e || !host_out) return fail(CTG_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_out, e->d_result, e->plan->result_elems * kItemSize[e->plan->dtype],
                      hipMemcpyDeviceToHost));
    return CTG_OK;
}

int ctg_exec_download_arena(ctg_exec* e, int64_t offset, int64_t n, void* host_out) {
    if (!e || !host_out) return fail(CTG_E_INVALID, "null argument");
    if (offset < 0 || n < 0 || offset + n > e->plan->arena_elems)
        return fail(CTG_E_BOUNDS, "arena range out of bounds");
    const int64_t isz = kItemSize[e->plan->dtype];
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_out, e->d_arena + offset * isz, n * isz, hipMemcpyDeviceToHost));
    return CTG_OK;
}

}  // extern "C"
