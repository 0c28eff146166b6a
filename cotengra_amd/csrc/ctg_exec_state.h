// ctg_exec_state.h -- the opaque handles of include/ctg_hip.h as seen by the
// host-side translation units of the library (ctg_runtime.hip,
// ctg_collective.hip).  Not installed; not part of the ABI.
#pragma once

#include <string>
#include <vector>

#include "../../include/ctg_hip.h"
#include "ctg_common.h"
#include "ctg_lds.h"

struct ctg_plan {
    int dtype = 0;
    int64_t n_inputs = 0;
    std::vector<int64_t> input_sizes, input_offsets;
    int64_t inputs_elems = 0, arena_elems = 0, result_elems = 0;
    int64_t n_steps = 0;
    std::vector<int64_t> steps;
    std::vector<int64_t> tables;
    int64_t n_sliced = 0;
    std::vector<int64_t> slice_sizes, slice_fixed, slice_strides;
    std::vector<int64_t> slice_group;   // [n_sliced] 1: a group index (ctg_plan_desc.slice_group)
    bool has_groups = false;            // some step is shared by the slices of a group (word 42 = 2)
    int64_t nslices = 1;
    // per-leaf maximum slice offset (for bounds validation)
    std::vector<int64_t> max_soff;
};

struct ctg_exec {
    const ctg_plan* plan = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    char* d_inputs = nullptr;
    char* d_arena = nullptr;
    char* d_result = nullptr;
    bool owns_result = false;
    // float / complex64 results of a sliced tree: the running sum of the slices in double precision (same
    // element offsets as the result, which always holds this sum rounded once); null otherwise
    char* d_wide = nullptr;
    // single-precision trees: [n_inputs offsets | n_inputs sizes] of the inputs space, and {2^S, S log10(2), (int) S}
    // -- the power of two taken out of the inputs at upload (prescale_inputs_kernel); null otherwise
    int64_t* d_in_tab = nullptr;
    double* d_inscale = nullptr;
    int64_t* d_tables = nullptr;
    int64_t* d_misc = nullptr;  // [state(2) | zero(1) | soff(n_leaves) | sizes | fixed | strides]
    int64_t* d_state = nullptr;
    int64_t* d_zero = nullptr;
    int64_t* d_soff = nullptr;
    void* d_scratch = nullptr;
    ctg::SliceMeta meta{};
    std::vector<ctg::StepArgs> args;  // resolved per step
    std::vector<ctg::StemArgs> stem_args;  // KIND_STEM2 steps (fused stem pairs)
    int stem_bf16x3 = 1;                   // ctg_exec_set_stem_arithmetic (default since round 4: bf16 x 3)
    // (round 6) 2: the stem kernels multiply with two fp16 limbs and three products (ctg_stem.hip built with
    // -DCTG_STEM_H2) where they took three bf16 limbs and six; stem_bf16x3 stays 1 then (the bf16-pipe kernels are on)
    int stem_arith = 2;
    // [3 banks][n_steps][batch][kMaxSub]: largest |component| recorded by step s in slice z of a launch sequence | of step s's
    // operand A | B (max-abs pass): smax_slot().  A slice-invariant step records into z = 0.
    float* d_stem_max = nullptr;
    int32_t* d_smax_zero = nullptr;        // (same shape) which of them start a slice at zero
    float* smax_slot(int bank, int64_t s, int64_t z) const {
        return d_stem_max + (((int64_t)bank * plan_steps + s) * (batch > 1 ? batch : 1) + z) * ctg::kMaxSub;
    }
    int64_t plan_steps = 0;                // (= plan->n_steps, for smax_slot)
    std::vector<char> rec_wanted;          // the record of step s's largest |component| has a reader (build_hints)
    std::vector<char> wave_member;         // step s is launched inside a wave-front group (it never records its maximum)
    std::vector<char> stem_h2_ran;         // step s last ran in the fp16 x 2 arithmetic (its record is valid)
    std::vector<ctg::MfmaHints> hints;  // per step kernel hints (MFMA steps)
    // the same for launches that carry several slices (batch > 1): wider column tiles
    // where one slice alone would not fill the chip; same k-splits, same kernels otherwise
    std::vector<ctg::MfmaHints> hints_b;
    uint16_t* d_ord_b = nullptr;
    char* d_lane_b = nullptr;
    uint16_t* d_ord = nullptr;     // order tables of all MFMA steps
    char* d_lane = nullptr;        // lane-constant tables of the fast tiled steps
    // slice batching: up to `batch` slices of a run share every launch (gridDim.y);
    // the arena holds `batch` replicas of itself, d_soff `batch` rows of leaf offsets
    int batch = 1;
    // slice groups in batched launches: slices per group (0: the plan has no groups or slices go one by
    // one), the ids of the slices of the launch at hand (device, `batch` entries)
    int group_d = 0;
    int64_t* d_batch_ids = nullptr;
    int64_t* h_ids = nullptr;      // pinned staging of the ids of batched group launches (run_grouped)
    size_t h_ids_cap = 0;
    hipEvent_t ev_ids = nullptr;   // ... recorded after the last copy out of it
    int batch_nominal = 1;         // min(64, nslices, 8 GiB / arena): what the k-splits are chosen for
    int64_t scratch_total = 0;     // bytes of d_scratch (64 MiB x up to 8 for batching executors)
    // the launch list of one slice: steps that launch alone (cls < 0) and wave-front
    // groups -- n independent small steps of one kernel shape sharing a launch (cls 0:
    // thread-per-output items, 1 + key: tiled fast-kernel items, starting at item0)
    struct Issue {
        int64_t step;
        int cls;
        int32_t item0, n;
        uint32_t blocks;
        bool shared = false;   // steps the slices of a group share (all members of a wave-front group alike)
    };
    std::vector<Issue> issue;
    std::vector<Issue> issue_reuse;   // ... of a slice that finds the shared steps of its group done
    // LDS-resident subtrees (round 6; ctg_lds_host.hip / ctg_lds_run.hip): per step the packed component that
    // runs it (-1: none, the step launches as usual); components sorted [group-shared..., per-slice...]
    std::vector<int32_t> lds_comp_of;
    int lds_first[2] = {0, 0}, lds_count[2] = {0, 0}, lds_bytes[2] = {0, 0};   // [0] group-shared, [1] per slice
    ctg::LdsCompDev* d_lds_comps = nullptr;
    char* d_lds_blob = nullptr;
    ctg::ValuGroupItem* d_group_items = nullptr;
    ctg::FastGroupItem* d_fast_items = nullptr;
    std::vector<hipEvent_t> events;
    // slice graph: the launch sequence of one slice captured once and replayed,
    // the slice id advancing on the device (prologue kernel)
    hipStream_t gstream = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bool warm = false;
    bool soff_static = false;      // unsliced tree: the (all-zero) leaf offsets are in place
    bool graph_off = false;
    // strip_exponent state
    int strip = 0, check_zero = 0;
    double* d_fac = nullptr;        // [n_steps + 1] max|.| per pair step; last = constant 1.0
    int32_t* d_counted = nullptr;   // [n_steps] 1 for pair steps
    int32_t* d_fac_zero = nullptr;  // [n_steps] 1 for per-slice pair steps
    // slice-invariant steps: executed once per upload / option change
    std::vector<char> invariant;
    bool invariants_ready = false;
    // steps shared by the slices of a group: executed when the group key changes
    std::vector<char> grouped;
    int64_t group_key = -1;
    ctg::StripState* d_strip = nullptr;
    int64_t root_step = -1;
};


// element size in bytes by CTG_* dtype code
inline int64_t ctg_item_size(int dtype) {
    static const int64_t k[4] = {4, 8, 8, 16};
    return k[dtype];
}

// records `msg` as the calling thread's ctg_last_error() (ctg_runtime.hip)
extern "C" __attribute__((visibility("hidden"))) void ctg_set_error_(const char* msg);

namespace ctg {
// (ctg_kernels_valu.hip) largest |re|, |im| of n complex64 values as a float; *out zeroed by the caller
// (nz slices from z on: the record (kMaxSub floats, zeroed by the caller) at out + i * out_zs for slice z + i)
hipError_t launch_maxabs_f32(const void* base, const int64_t* soff, int64_t z, int64_t zs, int64_t zstride, int64_t n,
                             float* out, hipStream_t stream, int nz = 1, int out_zs = 0);
}

// LDS-resident subtrees (ctg_lds_host.hip): descriptor validation (host only) and per-executor packing
__attribute__((visibility("hidden"))) int ctg_lds_validate(const ctg_plan* p);
__attribute__((visibility("hidden"))) int ctg_lds_build(ctg_exec* e);
