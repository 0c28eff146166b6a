/* ctg_hip.h -- C ABI of the MI355X-native contraction-tree executor.
 *
 * This is the drop-in boundary for the execution path of jcmgray/cotengra
 * (SURVEY.md section 8b).  cotengra itself is pure Python and has no FFI; the
 * slot this library fills is the one the reference gives to a whole-tree
 * native backend, `CuQuantumContractor` (reference cotengra/contract.py:840-922,
 * selected in `make_contractor`, contract.py:986-992): built once from a tree,
 * called with the input arrays, returns the contracted output.  Each entry
 * point below cites the reference behaviour it replaces.
 *
 * Conventions: every function returns 0 on success and a negative CTG_E_*
 * code on failure; `ctg_last_error()` returns a thread-local message for the
 * last failure.  Handles are opaque.  Host buffers are borrowed for the
 * duration of a call only and never written unless documented (the reference
 * never mutates its inputs: contract.py:779-807 only drops references).  All
 * device work is enqueued on the `stream` given at exec creation (a
 * `hipStream_t` passed as `void*`; NULL = the default stream) and is
 * asynchronous unless documented otherwise.  A `ctg_exec` is confined to one
 * host thread at a time (one per GPU); plans are immutable and shareable.
 *
 * No torch / numpy / Python types appear anywhere in this interface.
 */
#ifndef CTG_HIP_H
#define CTG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTG_ABI_VERSION 7

/* element types of the tensors (reference tests cover all four:
 * tests/test_compute.py:102-115) */
enum { CTG_F32 = 0, CTG_F64 = 1, CTG_C64 = 2, CTG_C128 = 3 };

enum {
    CTG_OK = 0,
    CTG_E_INVALID = -1, /* malformed plan / argument (ValueError in the reference) */
    CTG_E_HIP = -2,     /* a HIP runtime call failed */
    CTG_E_NOMEM = -3,   /* device allocation failed */
    CTG_E_BOUNDS = -4,  /* plan addresses outside a declared buffer */
    CTG_E_COMM = -5     /* RCCL missing, or a collective call failed */
};

#define CTG_STEP_WORDS 48

/* Flat description of a compiled plan (produced by cotengra_amd/plan.py).
 * It encodes what the reference keeps as the op list of
 * `extract_contractions` (contract.py:573-651) plus the per-step index
 * classification of `_parse_eq_to_batch_matmul` (contract.py:168-329), lowered
 * to offset tables; and the slice bookkeeping of `SliceInfo` /
 * `get_slice_strides` / `slice_key` (core.py:99-122, 3775-3800).
 *
 * Step kinds (word 0 of a step record): 0 single-term einsum (contract.py:62-119,
 * 332-361), 1 pairwise contraction (contract.py:364-411), 2 accumulate the slice
 * into the result (core.py:3842-3876), and -- ABI 3 -- 3: TWO consecutive pairwise
 * contractions of a stem executed as one launch whose intermediate never exists in
 * memory (two turns of the loop contract.py:788-832; complex64 only).  A kind-3
 * record names the big operand, the first small operand and the result like a
 * pair step; word 43 points at a descriptor in the table blob (geometry of the
 * tile decomposition, the second small operand, 14 offset tables: layout in
 * cotengra_amd/stem.py: serialise_stem, csrc/ctg_common.h: StemWord).  Like every
 * step it is validated by ctg_plan_create: tables inside the blob, every operand
 * address inside its buffer, a tile shape the kernel takes.
 *
 * ABI 7 -- LDS-resident subtrees (the small-tree execution model; the reference's step
 * loop contract.py:788-832 for a whole subtree at once).  Word 44 of a pair / single
 * step record: component id + 1 (0: none), word 45: word offset of the component's
 * descriptor in the table blob (cotengra_amd/ldsrun.py: serialise_run, csrc/ctg_common.h:
 * LdsHeadWord / LdsRecWord) -- a second lowering of the component's member steps on
 * tensors that live in one compute unit's LDS.  The member records stay complete and
 * come first in the step order; an executor runs all components of a plan as ONE launch
 * (one workgroup per component and slice of the batch) at the position of their first
 * member, or -- under strip_exponent, CTG_NO_LDS_RUNS=1, or when a component does not
 * fit the LDS of the device -- the member steps one by one.  ctg_plan_create validates
 * the descriptors like everything else (tables inside the blob, LDS addresses inside
 * the component's data area, global addresses inside their space). */
typedef struct ctg_plan_desc {
    int32_t dtype;                /* CTG_F32 .. CTG_C128 */
    int64_t n_inputs;
    const int64_t* input_sizes;   /* [n_inputs] elements of each unsliced input */
    const int64_t* input_offsets; /* [n_inputs] element offset inside the inputs space */
    int64_t inputs_elems;         /* size of the inputs space */
    int64_t arena_elems;          /* size of the intermediates arena */
    int64_t result_elems;         /* size of the full output tensor */
    int64_t n_steps;
    const int64_t* steps;         /* [n_steps * CTG_STEP_WORDS], layout in plan.py */
    int64_t n_table_words;
    const int64_t* tables;        /* offset-table blob addressed by the step records */
    int64_t n_sliced;
    const int64_t* slice_sizes;   /* [n_sliced] extent (1 when projected) */
    const int64_t* slice_fixed;   /* [n_sliced] projected value or -1 */
    const int64_t* slice_strides; /* [(n_inputs+1) * n_sliced] element strides; row
                                     n_inputs is the output tensor (outer slices) */
    /* ABI 5 (may be NULL = none).  [n_sliced] 1: a GROUP index.  Slices that differ only in the group
     * indices form a group; a step whose record says so (word 42 = 2) depends on none of them and is
     * computed for the first slice of a group only, what later steps read of it living in arena ranges
     * no per-slice step writes (the planner's job, cotengra_amd/plan.py: choose_slice_group).
     * ctg_exec_run_slices visits the slices it is given group by group.  The reference recomputes every
     * slice from the leaves (core.py:3802-3834). */
    const int64_t* slice_group;
} ctg_plan_desc;

typedef struct ctg_plan ctg_plan;
typedef struct ctg_exec ctg_exec;
typedef struct ctg_comm ctg_comm;

/* Library / error --------------------------------------------------------- */
int ctg_abi_version(void);
const char* ctg_last_error(void);

/* Plan: host-only, needs no GPU.  Replaces building a `Contractor` from
 * `extract_contractions(tree)` (contract.py:994-1000).  The descriptor is
 * validated (kinds, table ranges, that every address stays inside its
 * buffer) and deep-copied. */
int ctg_plan_create(const ctg_plan_desc* desc, ctg_plan** out);
int ctg_plan_destroy(ctg_plan* plan);
/* number of independent slices = prod(slice_sizes) (core.py:403-408) */
int ctg_plan_nslices(const ctg_plan* plan, int64_t* nslices);
/* device bytes an exec will allocate: [0] inputs [1] arena [2] result
 * [3] tables+misc */
int ctg_plan_workspace_bytes(const ctg_plan* plan, int64_t bytes[4]);

/* Exec: one per GPU.  Owns inputs space, arena, tables; the result buffer is
 * either owned or caller-provided device memory (`ext_result`, e.g. memory a
 * collective library will reduce in place).  Replaces the lazy `setup(*arrays)`
 * of the whole-tree backend (contract.py:883-899).
 * A single-slice arena of 32 GiB or more is allocated twice where the free device memory allows
 * (peak: twice its size for a moment), the plan's largest tensors' ranges are read in both copies and
 * the faster copy is kept -- physical placement in HBM is worth 2-3 % of a slice on MI355X
 * (profiles/r6_process_alternation.txt); results do not depend on it.  CTG_ARENA_PLACE=0: off. */
int ctg_exec_create(const ctg_plan* plan, int device, void* stream,
                    void* ext_result, ctg_exec** out);
int ctg_exec_destroy(ctg_exec* exec);
/* Move the executor to another stream of its device (a caller that works under
 * changing stream contexts, e.g. `torch.cuda.stream(...)`): waits for the work
 * already enqueued on the old stream, then every later call enqueues on
 * `stream`. */
int ctg_exec_set_stream(ctg_exec* exec, void* stream);

/* Make the (unsliced) input tensors resident: `ptrs[i]` points to
 * input_sizes[i] contiguous row-major elements.  Replaces passing `*arrays`
 * to the contractor (contract.py:718, 779-780) / `reset_operands`
 * (contract.py:916).  `_host` synchronises the stream before returning;
 * `_device` enqueues device-to-device copies. */
int ctg_exec_upload_inputs_host(ctg_exec* exec, const void* const* ptrs);
int ctg_exec_upload_inputs_device(ctg_exec* exec, const void* const* ptrs);

/* result <- 0 (start of a `gather_slices` reduction, core.py:3842-3844) */
int ctg_exec_zero_result(ctg_exec* exec);

/* strip_exponent / check_zero of the reference's Contractor (contract.py:
 * 674-683, 816-829): after every pairwise step the intermediate is normalised
 * by factor = max|p| and log10(factor) accumulated.  On the device the
 * normalisation is lazy (the consumer's epilogue multiplies by 1/(fac_l fac_r))
 * and slices are combined with the exponent-aware adder of core.py:125-172.
 * With strip on, the result tensor holds the mantissa and ctg_exec_get_exponent
 * returns the base-10 exponent E (result = mantissa * 10^E; E = -inf and
 * *zero = 1 when check_zero met a zero intermediate in every slice). */
int ctg_exec_set_strip_exponent(ctg_exec* exec, int strip_exponent, int check_zero);

/* Arithmetic of the fused stem pairs (step kind 3) of this executor.  1 (the default since ABI 4):
 * every fp32 operand of a pair is split EXACTLY into three bfloat16 values and the six significant
 * cross terms are accumulated in fp32 on the bf16 matrix cores (error bound,
 * domain -- operands above 2^-110, small operands rescaled by a power of two inside the kernel --
 * and the adversarial tests that hold it to the fp32 kernel's own error); the same accuracy against
 * a double-precision reference as 0, not the same bits, 11-13 % less time per slice.  0: complex64 on
 * the fp32 matrix cores, an exact-fp32 multiply-add chain like every other step.  The environment
 * variable CTG_STEM_BF16X3, when set, overrides the option ("0" = fp32).  No reference counterpart
 * (the reference computes in whatever its array library does).  Takes effect from the next run.
 * 2 (ABI 7, the default of a new executor unless CTG_STEM_ARITH = fp32 | bf16x3 | fp16x2 says otherwise):
 * two ROUNDED fp16 limbs per operand (22 bits) under per-tensor power-of-two scales and THREE products --
 * half the matrix work of 1, the stem pairs at 0.7 instead of 0.45 of the HBM roofline.  The scales: the
 * small operands' largest elements (found in-kernel), the big operand's as its producer recorded it (every
 * such kernel tracks the largest element it stores; a big operand of other origin gets a max-abs pass), the
 * intermediate tile's own; an element more than 2^-14 below its tensor's largest loses low bits gradually
 * (absolute error <= 2^-24 of the largest: norm-wise accuracy).  strip_exponent runs
 * and CTG_STEM_H2=0 fall back to 1.  The LONG TILED steps (K >= 64 on full 64-column tiles: the GEMM-like steps of
 * a tree) follow the same mode: 1 = six bf16 products, 2 = two fp16 limbs per value under ONE power of two per
 * operand tensor and slice, three products -- for a launch without k-splits (its operands' largest elements as
 * their producers recorded them per slice of the batch, else a max-abs pass); k-split launches and CTG_PAIR_H2=0
 * keep 1. */
int ctg_exec_set_stem_arithmetic(ctg_exec* exec, int mode);
int ctg_exec_get_exponent(ctg_exec* exec, double* exponent, int* zero);

/* Contract slices first, first+stride, ... (count of them) and ACCUMULATE each
 * into the result tensor at its chunk position: the slice loop of
 * `ContractionTree.contract` (core.py:4015-4030) fused with `gather_slices`
 * (core.py:3825-3882), or with count=1 `contract_slice` (core.py:3821-3823);
 * `stride = world_size` gives the round-robin of `contract_mpi`
 * (core.py:4068-4076).  Slice-to-leaf indexing (`slice_arrays`,
 * core.py:3802-3819) happens on the device. */
int ctg_exec_run_slices(ctg_exec* exec, int64_t first, int64_t count, int64_t stride);

/* How many slices of a run_slices call share one launch sequence.  A slice of a
 * narrow tree is launch-bound (Sycamore m10: 170 launches of a few microseconds),
 * so the executor carries up to `batch` consecutive slices of a run through every
 * launch (gridDim.y), each in its own replica of the arena; the reference's slice
 * loop (core.py:4015-4028) has no counterpart -- it is one Python iteration per
 * slice.  Chosen when the executor is built: min(64, nslices), bounded by 8 GiB
 * (and a quarter of the free device memory) of arena replicas (environment:
 * CTG_SLICE_BATCH, CTG_SLICE_BATCH_MIB); wide trees get 1.  The result does not
 * depend on it bit for bit: the k-split of every step is a function of the plan
 * (the step's shape and the plan's nominal batch min(64, nslices, 8 GiB / arena)
 * -- not of the environment, the free memory or the launch at hand), the kernels
 * and the order in which slices are added are those of one launch sequence per
 * slice. */
int ctg_exec_slice_batch(ctg_exec* exec, int64_t* batch);
/* ABI 5.  The same for an arbitrary list of slice ids (each in [0, nslices); repetitions allowed: a slice
 * given twice is added twice).  With slice groups in the plan (ctg_plan_desc.slice_group) the slices are
 * visited group by group and the steps a group shares are computed once per group among the ids given --
 * pass whole groups to get the saving; the sum does not depend on the grouping beyond the order of
 * its floating-point additions. */
int ctg_exec_run_slice_list(ctg_exec* exec, const int64_t* ids, int64_t n);
/* ABI 6.  A rank's SHARE of the slices, as the library deals them: the units rank, rank + world, ... where a
 * unit is a whole slice group (ctg_plan_desc.slice_group; what a group shares is then computed once per
 * group on exactly one rank) and, for a plan without group indices, a single slice -- which is the
 * round-robin `range(rank, nslices, size)` of contract_mpi (core.py:4068-4076).  The shares of the ranks
 * are disjoint, cover every slice and differ by at most one unit.
 *   ctg_plan_share_units      -> how many units rank holds and the slices in a unit (host only);
 *   ctg_plan_share_slice_ids  -> the slice ids of units [unit_first, unit_first + unit_count) of the
 *                                share, unit after unit, ascending inside a unit (unit_count < 0: to the
 *                                end; `ids` holds unit_count * slices_per_unit words) (host only);
 *   ctg_exec_run_share        -> contract those units and accumulate them like ctg_exec_run_slices; host
 *                                memory stays bounded by a chunk of groups whatever nslices is.
 * A checkpointing caller counts finished UNITS of its share and resumes with unit_first = that count. */
int ctg_plan_share_units(const ctg_plan* plan, int64_t rank, int64_t world, int64_t* units, int64_t* slices_per_unit);
int ctg_plan_share_slice_ids(const ctg_plan* plan, int64_t rank, int64_t world, int64_t unit_first,
                             int64_t unit_count, int64_t* ids);
int ctg_exec_run_share(ctg_exec* exec, int64_t rank, int64_t world, int64_t unit_first, int64_t unit_count);
/* ABI 4.  Device memory this executor holds right now: inputs space, arena x slice batch,
 * tables, the result if it owns it, the scratch buffer if the plan has a step that needs one
 * (allocated by ctg_exec_create).  What a cache of contractors -- the reference keeps them
 * on the tree, core.py:3708-3722 -- has to count against its budget. */
int ctg_exec_device_bytes(ctg_exec* exec, int64_t* bytes);
/* ABI 5.  Is there a kernel instantiation for a three-step tile of this shape (stem steps fused
 * three at a time, cotengra_amd/stem.py: build_stem_triple; opt-in, CTG_STEM_TRIPLES)?  A pure
 * function of the shape -- 16 output columns in the first / middle / last step (0 / 1), units of
 * step 1 per wave, its column groups, 16-deep chunks of its contraction, work items per wave of the
 * middle and of the last step, 16-byte gathers (0 / 1) -- that needs no device: the planner asks
 * before it emits such a record (there is no run-time-count variant; ctg_plan_create rejects a
 * record without one).  Returns 1 or 0.  (The reference has no counterpart: it contracts step by
 * step, contract.py:788-832.) */
int ctg_stem_triple_instantiated(int p1, int pm, int p2, int rt1, int cs1, int nch, int itm, int it2, int vec);

/* Steps of one slice and the kernel launches they take.  Independent small steps
 * (the leaves-upward wave fronts of a tree; the reference contracts them one
 * einsum call after the other, contract.py:788-829) share launches: `steps` pair /
 * single / accumulate steps per slice go out as `launches` launches.  Grouping
 * never changes a result bit (every output element is computed by the same
 * thread in the same order); it is off under strip_exponent and with the
 * environment variable CTG_NO_GROUPS. */
int ctg_exec_launch_count(ctg_exec* exec, int64_t* steps, int64_t* launches);

/* Same as run_slices(slice_id, 1, 1) but brackets every step with events and
 * returns its duration in milliseconds (`ms[n_steps]`); synchronous.  Serves
 * the role of `tree.print_contractions` + `tree.benchmark`
 * (core.py:3508, 4092-4164) for per-step rooflines. */
int ctg_exec_profile_slice(ctg_exec* exec, int64_t slice_id, float* ms);

/* Name of the kernel that executes plan step `step` on this executor (e.g.
 * "pair_mfma_fast_kernel<128,128,16>"), NUL-terminated into `buf`; lets a
 * profile attribute per-step timings to rocprof kernel names. */
int ctg_exec_step_kernel(ctg_exec* exec, int64_t step, char* buf, int64_t buflen);

int ctg_exec_sync(ctg_exec* exec);
/* device address of the result tensor (result_elems elements, row-major in
 * the tree's output index order) */
int ctg_exec_result_ptr(ctg_exec* exec, void** dev_ptr);
/* synchronous copy of the result to host memory */
int ctg_exec_download_result(ctg_exec* exec, void* host_out);
/* debugging aid: copy `n` elements of the arena starting at element `offset`
 * to the host (synchronous) */
int ctg_exec_download_arena(ctg_exec* exec, int64_t offset, int64_t n, void* host_out);

/* Checkpoint of a sliced run.  The reference sums slices into `result` one at
 * a time (`gather_slices`, core.py:3842-3844; `contract_mpi`, core.py:4073-4076),
 * so the whole state of an interrupted run is (slices done, partial sum[,
 * exponent]); which slices are done is the caller's cursor (it chose first /
 * count / stride).  `get_state` synchronises and copies the partial result
 * (result_elems elements) to the host together with the strip_exponent
 * exponent (0 / not-zero when stripping is off); `set_state` on a fresh
 * executor (same plan, same strip_exponent setting) replaces
 * ctg_exec_zero_result, after which run_slices continues the sum exactly where
 * it stopped: resuming is bit-identical to an uninterrupted run. */
int ctg_exec_get_state(ctg_exec* exec, void* host_result, double* exponent, int* zero);
int ctg_exec_set_state(ctg_exec* exec, const void* host_result, double exponent, int zero);
/* ABI 6.  Single-precision results (CTG_F32 / CTG_C64) of a sliced tree are summed in DOUBLE precision: the
 * executor keeps the running sum of the slices as CTG_F64 / CTG_C128 next to the result tensor, which always
 * holds that sum rounded once (a left fold of 2^20 slices in fp32 -- what the reference does, core.py:3842-3844
 * -- loses more than the 1e-5 this library promises); ctg_exec_reduce sums the ranks' double-precision sums.
 * The state of an interrupted run therefore is that sum: `ctg_exec_state_dtype` says in which element type
 * (the plan's own when there is no wider sum), `_wide` get / set move it (result_elems elements of that type);
 * the narrow pair above still works -- `ctg_exec_set_state` restarts the sum from the rounded values, which
 * is exact but not the bits an uninterrupted run would have carried. */
int ctg_exec_state_dtype(ctg_exec* exec, int* dtype);
int ctg_exec_get_state_wide(ctg_exec* exec, void* host_sum, double* exponent, int* zero);
int ctg_exec_set_state_wide(ctg_exec* exec, const void* host_sum, double exponent, int zero);

/* Multi-GPU: one process (or thread) per GPU, each with its own exec running
 * ctg_exec_run_share(exec, rank, world, 0, -1) -- ITS SHARE of the slices, dealt by the
 * library (ABI 6): the units rank, rank + world, ... where a unit is a whole slice group
 * (what a group shares is then computed once per group, on one rank) and a single slice
 * for a plan without group indices, which is the round-robin of `contract_mpi`
 * (core.py:4068-4076) -- followed by ONE collective over the result tensor: `comm.Allreduce` / `comm.Reduce` there (core.py:4081, 4089),
 * RCCL over xGMI here.  The communicator is created from a 128-byte unique id
 * made on one rank and handed to the others by whatever channel the caller has
 * (MPI bcast, a file, torch.distributed's store ...) -- the role of mpi4py's
 * COMM_WORLD in the reference (core.py:4057-4060).  RCCL is bound with dlopen
 * at the first call (librccl.so.1, librccl.so; CTG_RCCL_LIB names the one library to
 * bind instead -- no fallback when it cannot be loaded); CTG_E_COMM if it is absent. */
#define CTG_UNIQUE_ID_BYTES 128
int ctg_comm_get_unique_id(void* id_out /* [CTG_UNIQUE_ID_BYTES] */);
/* collective over all `world` ranks; `device` is this rank's GPU ordinal */
int ctg_comm_init(const void* id, int rank, int world, int device, ctg_comm** out);
int ctg_comm_info(const ctg_comm* comm, int* rank, int* world, int* device);
int ctg_comm_destroy(ctg_comm* comm);
/* Sum the ranks' result tensors in place, enqueued on the exec's stream behind
 * its slices: root < 0 -> every rank ends with the total (Allreduce,
 * core.py:4081); root >= 0 -> only `root` does, the others' result tensors are
 * left undefined (Reduce, core.py:4089).  With strip_exponent the partials are
 * (mantissa, exponent) pairs: the ranks first agree on the largest exponent,
 * rescale their mantissas to it (the adder of core.py:163-172 across ranks) and
 * then sum; ctg_exec_get_exponent returns the common exponent afterwards. */
int ctg_exec_reduce(ctg_exec* exec, ctg_comm* comm, int root);

/* Host-side tree tools (no GPU): the inner loops of path search and slicing.
 * The reference runs them in Python, or in its optional Rust accelerator
 * `cotengrust` when installed (pathfinders/path_basic.py:1351-1383); the
 * hyper-optimizers that call them stay in the reference unchanged.
 *
 * A network is given in CSR form: tensor t carries the index ids
 * inds[offsets[t] .. offsets[t+1]) (ids in [0, n_inds), repeats allowed),
 * `out_inds` are the output indices, `sizes[ix]` the extent of index ix. */

/* Greedy pairwise path (reference `optimize_greedy`, path_basic.py:616-700,
 * 1038-1106): score = size(ab)/costmod - (size(a)+size(b))*costmod, optional
 * Boltzmann sampling (temperature, seed), indices shared by more than
 * `max_neighbors` tensors generate no candidates (0 = no limit), disconnected
 * remainder combined smallest first.  Writes n_inputs-1 pairs of SSA ids
 * (inputs 0..n-1, intermediates n, n+1, ...) to ssa_path[2*(n_inputs-1)]. */
int ctg_path_greedy(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                    const int64_t* out_inds, int64_t n_inds, const double* sizes, double costmod,
                    double temperature, int64_t max_neighbors, uint64_t seed, int64_t* ssa_path);

/* Greedy choice of indices to slice until the largest intermediate of the tree
 * `ssa_path` has at most 2^target_log2_size elements (reference `SliceFinder`
 * over `ContractionCosts`, slicer.py:17-201, 204-430; cost model: removing an
 * index of extent d divides the flops / sizes it takes part in by d and
 * multiplies the slice count by d).  Output indices are only used when
 * `allow_outer` (core.py:2632-2719).  Writes at most `max_sliced` index ids. */
int ctg_slice_greedy(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                     const int64_t* out_inds, int64_t n_inds, const double* sizes,
                     const int64_t* ssa_path, double target_log2_size, int allow_outer,
                     int64_t max_sliced, int64_t* sliced, int64_t* n_sliced);

/* Subtree reconfiguration (reference `ContractionTree.subtree_reconfigure`,
 * core.py:2316-2449, with its dynamic-programming sub-optimizer): optimal
 * re-ordering of subtrees of `subtree_size` (2..16) leaves, most expensive node
 * first, at most `maxiter` subtree optimisations (<= 0: min(n_inputs, 1024)).
 * Cost of a contraction = flops + write_factor * size(result) -- the reference's
 * `combo-<write_factor>` objective (0: flops).  Give sliced indices size 1.
 * Reads ssa_path_in, writes ssa_path_out (both 2*(n_inputs-1) ids). */
int ctg_subtree_reconfigure(int64_t n_inputs, const int64_t* offsets, const int64_t* inds, int64_t n_out,
                            const int64_t* out_inds, int64_t n_inds, const double* sizes,
                            const int64_t* ssa_path_in, int64_t subtree_size, int64_t maxiter,
                            double write_factor, int64_t* ssa_path_out);

/* The same search with a machine model as the objective (no reference
 * counterpart; the reference's objectives are the `scoring.py` closed forms): a
 * contraction costs max(MACs / mac_rate_by_log2k[floor(log2 K)], (size_a + size_b
 * + size_out) / elem_rate) seconds, K = its contracted extent (the last table
 * entry serves every larger K), the MAC rate scaled by N/16 when the narrower
 * kept side has N < 16 columns and by 0.8 when it has 16..63 and K >= 64.  The table holds the caller's measurements of
 * the executor's kernels (cotengra_amd.pathfind.MI355X_C64). */
int ctg_subtree_reconfigure_timed(int64_t n_inputs, const int64_t* offsets, const int64_t* inds,
                                  int64_t n_out, const int64_t* out_inds, int64_t n_inds,
                                  const double* sizes, const int64_t* ssa_path_in, int64_t subtree_size,
                                  int64_t maxiter, const double* mac_rate_by_log2k, int64_t n_rates,
                                  double elem_rate, int64_t* ssa_path_out);

#ifdef __cplusplus
}
#endif
#endif /* CTG_HIP_H */
