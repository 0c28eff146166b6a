// ctg_lds.h -- LDS-resident subtrees: what the host packer (ctg_lds_host.hip), the runtime and the kernel
// (ctg_lds_run.hip) share.  Kept out of ctg_common.h so that the large kernel translation units do not
// depend on it.
#pragma once

#include "ctg_common.h"

namespace ctg {

// LDS-resident subtrees (plan: cotengra_amd/ldsrun.py, kernel: ctg_lds_run.hip).  A component's descriptor in
// the table blob: LR_HEAD_WORDS of header, then one record of LR_WORDS per "shadow" step -- the member steps
// lowered a second time, on tensors that live in LDS.
enum LdsHeadWord { LH_MAGIC = 0, LH_NREC = 1, LH_ELEMS = 2, LH_PHASES = 3, LH_ID = 4, LR_HEAD_WORDS = 8 };
enum LdsRecWord {
    LR_KIND = 0,       // 0: load (global -> LDS gather, optional sum over k), 1: pair
    LR_PHASE = 1,      // loads 0, pairs >= 1; a workgroup barrier separates phases
    LR_MAIN = 2,       // the ordinary step the record belongs to
    LR_GOP = 3,        // which operand of that step is the record's GLOBAL side: 0 A, 1 B, 2 C, -1 none
    LR_A_LDS = 4, LR_A_OFF = 5, LR_B_LDS = 6, LR_B_OFF = 7, LR_C_LDS = 8, LR_C_OFF = 9,
    LR_R = 10, LR_K = 11, LR_N = 12, LR_ROW_LO = 13, LR_ROW_HI_LEN = 14, LR_K_LO = 15, LR_K_HI_LEN = 16,
    LR_ROWA_HI = 17, LR_ROWA_LO = 18, LR_ROWB_HI = 19, LR_ROWB_LO = 20, LR_ROWC_HI = 21, LR_ROWC_LO = 22,
    LR_KA_HI = 23, LR_KA_LO = 24, LR_KB_HI = 25, LR_KB_LO = 26, LR_NB = 27, LR_NC = 28,
    LR_A_SIZE = 29, LR_B_SIZE = 30, LR_C_SIZE = 31, LR_MACS = 32,
    LR_WORDS = 40
};
constexpr int64_t LR_MAGIC = 0x4C44535231;
constexpr int LDS_RUN_THREADS = 1024;           // one workgroup of 16 waves per component
constexpr int LDS_RUN_MAX_BYTES = 160 * 1024;   // gfx950: LDS per workgroup

// One shadow step as the kernel reads it from LDS (the component's blob: records, then tables).
// Offsets of tables are BYTES from the start of the blob.
struct LdsStepDev {
    // (the first 16 bytes decide whether a wave takes part in the step at all)
    int32_t kind, phase;
    int32_t wave;                   // -1: the whole workgroup shares the step; else the wave that runs it alone
    int32_t mfma;                   // complex64 pair step on the matrix cores (32 x 16 tiles): 1 columns on the lanes, 2 rows
    int32_t R, K, N;
    int32_t row_lo, row_shift;      // rows are two-level: r -> (r / row_lo, r % row_lo); shift = log2 or -1
    uint32_t row_magic;             // floor(2^32 / row_lo) + 1 when row_lo is not a power of two (r < 2^16)
    int32_t rot;                    // matrix-core steps shared by the workgroup: task t goes to wave (t + rot) % waves
    int32_t a_off, b_off, c_off;    // LDS element offsets of the operands (-1: the global side)
    uint32_t t_row_hi, t_row_lo;    // row entries, 8 bytes each: pair {a | b << 16, c}, load {global a, c}
    uint32_t t_k;                   // K entries, 4 bytes: pair ka | kb << 16, load global ka
    uint32_t t_n;                   // N entries, 4 bytes: nb | nc << 16 (pairs)
    int32_t pad1, pad2;
    // the global side (loads: the source; a component's root: the result): base pointer, per-slice offsets
    const char* gptr;
    const int64_t* gsoff;
    int64_t gz;                     // arena replica stride (elements) of slice-in-batch z
    int32_t gzs;                    // stride of gsoff (entries)
    int32_t gzq;                    // > 1: the operand lives with the first slice of its group of gzq
};
static_assert(sizeof(LdsStepDev) % 16 == 0, "records are copied and read in 16-byte pieces");

struct LdsCompDev {
    const char* blob;       // records + tables of the component (device memory)
    uint32_t blob_bytes;    // multiple of 16
    uint32_t n_steps;
    uint32_t data_off;      // byte offset of the data area in LDS (after the blob)
    uint32_t pad;
};

// n_comps workgroups x nz slices of the batch (ctg_lds_run.hip)
hipError_t launch_lds_run(int dtype, const LdsCompDev* d_comps, int n_comps, int nz, int z0, int lds_bytes, hipStream_t stream);

}  // namespace ctg
