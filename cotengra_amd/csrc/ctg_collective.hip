// ctg_collective.hip -- the exchange step of a slice-parallel contraction and the
// checkpoint state of a sliced run, behind the C ABI.
//
// Reference: `ContractionTree.contract_mpi` (cotengra/core.py:4032-4090) ends in
// ONE `comm.Allreduce` / `comm.Reduce` of the locally summed output (:4081, :4089).
// Here the ranks are one process per MI355X and the collective is RCCL over xGMI,
// enqueued on the executor's own stream right behind its last slice, reducing the
// resident result tensor in place -- no host copy, no extra synchronisation.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already
// carries PyTorch-ROCm gets the RCCL torch loaded (one HIP runtime per process),
// a plain C/C++ caller gets the one of the ROCm installation; a single-GPU user
// needs neither.
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <rccl/rccl.h>  // types and enums only; no link-time dependency

#include "ctg_exec_state.h"

using namespace ctg;

namespace {

int cfail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctg_set_error_(buf);
    return code;
}

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t,
                           hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;  // reason the binding failed
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

void bind_rccl() {
    // (an explicit CTG_RCCL_LIB is the ONLY candidate: a library named by the caller that cannot be loaded is an
    // error -- CTG_E_COMM with dlerror's text --, not a reason to bind some other RCCL silently)
    const char* named = getenv("CTG_RCCL_LIB");
    const bool explicit_lib = named != nullptr && *named != '\0';
    const char* names[3] = {explicit_lib ? named : nullptr, explicit_lib ? nullptr : "librccl.so.1",
                            explicit_lib ? nullptr : "librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
        g_rccl.why = dlerror();
    }
    if (!g_rccl.handle) return;
    bool ok = true;
    auto sym = [&](const char* s) -> void* {
        void* p = dlsym(g_rccl.handle, s);
        if (!p) {
            ok = false;
            g_rccl.why = std::string("missing symbol ") + s;
        }
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.Reduce = (decltype(g_rccl.Reduce))sym("ncclReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
    }
}

const RcclApi* rccl() {
    std::call_once(g_rccl_once, bind_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define RCCL_TRY(api, expr)                                                          \
    do {                                                                             \
        ncclResult_t _r = (expr);                                                    \
        if (_r != ncclSuccess)                                                       \
            return cfail(CTG_E_COMM, "%s failed: %s", #expr, (api)->GetErrorString(_r)); \
    } while (0)

#define HIP_TRY_C(expr)                                                              \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess)                                                        \
            return cfail(CTG_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));  \
    } while (0)

// Exponent-aware merge of per-rank partials (AdderWithMaybeExponentStripped,
// core.py:163-172, across ranks): given the maximum exponent of all ranks, set
// the rescale coefficient of the local mantissa and adopt the common exponent.
__global__ void merge_prepare_kernel(StripState* st, const double* e_max) {
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const double E = st->E, En = *e_max;
    st->coefM = (En == -inf) ? 1.0 : (E == -inf ? 0.0 : pow(10.0, E - En));
    st->E = En;
    st->zero = En == -inf ? 1 : 0;
}

}  // namespace

struct ctg_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    double* d_emax = nullptr;  // exponent exchange of strip_exponent runs
};

extern "C" {

int ctg_comm_get_unique_id(void* id_out) {
    if (!id_out) return cfail(CTG_E_INVALID, "null argument");
    const RcclApi* api = rccl();
    if (!api) return cfail(CTG_E_COMM, "RCCL is not available: %s", g_rccl.why.c_str());
    static_assert(sizeof(ncclUniqueId) == CTG_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return CTG_OK;
}

int ctg_comm_destroy(ctg_comm* c) {
    if (!c) return CTG_OK;
    (void)hipSetDevice(c->device);
    if (c->d_emax) (void)hipFree(c->d_emax);
    const RcclApi* api = rccl();
    if (c->comm && api) (void)api->CommDestroy(c->comm);
    delete c;
    return CTG_OK;
}

int ctg_comm_init(const void* id_in, int rank, int world, int device, ctg_comm** out) {
    if (!id_in || !out) return cfail(CTG_E_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world)
        return cfail(CTG_E_INVALID, "rank %d outside a world of %d", rank, world);
    const RcclApi* api = rccl();
    if (!api) return cfail(CTG_E_COMM, "RCCL is not available: %s", g_rccl.why.c_str());
    int ndev = 0;
    HIP_TRY_C(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return cfail(CTG_E_INVALID, "device %d not available (%d visible)", device, ndev);
    HIP_TRY_C(hipSetDevice(device));
    ctg_comm* c = new ctg_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        ctg_comm_destroy(c);
        return cfail(CTG_E_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world,
                     device, api->GetErrorString(r));
    }
    if (hipMalloc((void**)&c->d_emax, sizeof(double)) != hipSuccess) {
        ctg_comm_destroy(c);
        return cfail(CTG_E_NOMEM, "hipMalloc failed");
    }
    *out = c;
    return CTG_OK;
}

int ctg_comm_info(const ctg_comm* c, int* rank, int* world, int* device) {
    if (!c) return cfail(CTG_E_INVALID, "null argument");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return CTG_OK;
}

int ctg_exec_reduce(ctg_exec* e, ctg_comm* c, int root) {
    if (!e || !c) return cfail(CTG_E_INVALID, "null argument");
    if (root >= c->world) return cfail(CTG_E_INVALID, "root %d outside a world of %d", root, c->world);
    if (c->device != e->device)
        return cfail(CTG_E_INVALID, "communicator lives on device %d, executor on device %d", c->device,
                     e->device);
    const RcclApi* api = rccl();
    if (!api) return cfail(CTG_E_COMM, "RCCL is not available: %s", g_rccl.why.c_str());
    HIP_TRY_C(hipSetDevice(e->device));
    const ctg_plan* p = e->plan;
    if (e->strip) {
        // partial = mantissa * 10^E on every rank: agree on E' = max E, bring the local
        // mantissa to it, then sum (one extra 8-byte all-reduce)
        RCCL_TRY(api, api->AllReduce(&e->d_strip->E, c->d_emax, 1, ncclDouble, ncclMax, c->comm, e->stream));
        hipLaunchKernelGGL(merge_prepare_kernel, dim3(1), dim3(1), 0, e->stream, e->d_strip, c->d_emax);
        HIP_TRY_C(hipGetLastError());
        HIP_TRY_C(launch_rescale(p->dtype, e->d_result, p->result_elems, e->d_strip, e->stream));
        if (e->d_wide) HIP_TRY_C(launch_rescale(p->dtype + 1, e->d_wide, p->result_elems, e->d_strip, e->stream));
    }
    // complex tensors travel as pairs of reals; single-precision results of a sliced tree travel as their
    // double-precision running sums (16 bytes per complex element) and are rounded once, after the sum
    const bool dbl = p->dtype == CTG_F64 || p->dtype == CTG_C128 || e->d_wide != nullptr;
    const size_t count = (size_t)p->result_elems * ((p->dtype == CTG_C64 || p->dtype == CTG_C128) ? 2 : 1);
    const ncclDataType_t dt = dbl ? ncclDouble : ncclFloat;
    void* buf = e->d_wide ? (void*)e->d_wide : (void*)e->d_result;
    if (root < 0)
        RCCL_TRY(api, api->AllReduce(buf, buf, count, dt, ncclSum, c->comm, e->stream));
    else
        RCCL_TRY(api, api->Reduce(buf, buf, count, dt, ncclSum, root, c->comm, e->stream));
    if (e->d_wide && (root < 0 || root == c->rank))
        HIP_TRY_C(launch_narrow(p->dtype, e->d_result, e->d_wide, p->result_elems, e->stream));
    return CTG_OK;
}

// ---- checkpoint state of a sliced run ------------------------------------- //

int ctg_exec_get_state(ctg_exec* e, void* host_result, double* exponent, int* zero) {
    if (!e || !host_result) return cfail(CTG_E_INVALID, "null argument");
    HIP_TRY_C(hipSetDevice(e->device));
    HIP_TRY_C(hipStreamSynchronize(e->stream));
    HIP_TRY_C(hipMemcpy(host_result, e->d_result, e->plan->result_elems * ctg_item_size(e->plan->dtype),
                        hipMemcpyDeviceToHost));
    StripState st{};
    HIP_TRY_C(hipMemcpy(&st, e->d_strip, sizeof(st), hipMemcpyDeviceToHost));
    if (exponent) *exponent = e->strip ? st.E : 0.0;
    if (zero) *zero = e->strip ? st.zero : 0;
    return CTG_OK;
}

static int set_strip_state(ctg_exec* e, double exponent, int zero);

int ctg_exec_state_dtype(ctg_exec* e, int* dtype) {
    if (!e || !dtype) return cfail(CTG_E_INVALID, "null argument");
    *dtype = e->d_wide ? e->plan->dtype + 1 : e->plan->dtype;   // (CTG_F32 -> CTG_F64, CTG_C64 -> CTG_C128)
    return CTG_OK;
}

int ctg_exec_get_state_wide(ctg_exec* e, void* host_sum, double* exponent, int* zero) {
    if (!e || !host_sum) return cfail(CTG_E_INVALID, "null argument");
    if (!e->d_wide) return ctg_exec_get_state(e, host_sum, exponent, zero);
    HIP_TRY_C(hipSetDevice(e->device));
    HIP_TRY_C(hipStreamSynchronize(e->stream));
    HIP_TRY_C(hipMemcpy(host_sum, e->d_wide, e->plan->result_elems * 2 * ctg_item_size(e->plan->dtype), hipMemcpyDeviceToHost));
    StripState st{};
    HIP_TRY_C(hipMemcpy(&st, e->d_strip, sizeof(st), hipMemcpyDeviceToHost));
    if (exponent) *exponent = e->strip ? st.E : 0.0;
    if (zero) *zero = e->strip ? st.zero : 0;
    return CTG_OK;
}

int ctg_exec_set_state_wide(ctg_exec* e, const void* host_sum, double exponent, int zero) {
    if (!e || !host_sum) return cfail(CTG_E_INVALID, "null argument");
    if (!e->d_wide) return ctg_exec_set_state(e, host_sum, exponent, zero);
    HIP_TRY_C(hipSetDevice(e->device));
    HIP_TRY_C(hipStreamSynchronize(e->stream));
    HIP_TRY_C(hipMemcpy(e->d_wide, host_sum, e->plan->result_elems * 2 * ctg_item_size(e->plan->dtype), hipMemcpyHostToDevice));
    HIP_TRY_C(launch_narrow(e->plan->dtype, e->d_result, e->d_wide, e->plan->result_elems, e->stream));
    HIP_TRY_C(hipStreamSynchronize(e->stream));
    return set_strip_state(e, exponent, zero);
}

int ctg_exec_set_state(ctg_exec* e, const void* host_result, double exponent, int zero) {
    if (!e || !host_result) return cfail(CTG_E_INVALID, "null argument");
    HIP_TRY_C(hipSetDevice(e->device));
    HIP_TRY_C(hipStreamSynchronize(e->stream));
    HIP_TRY_C(hipMemcpy(e->d_result, host_result, e->plan->result_elems * ctg_item_size(e->plan->dtype),
                        hipMemcpyHostToDevice));
    if (e->d_wide) {   // (a state in the result's own precision: the running sum starts from it, exactly)
        HIP_TRY_C(launch_widen(e->plan->dtype, e->d_wide, e->d_result, e->plan->result_elems, e->stream));
        HIP_TRY_C(hipStreamSynchronize(e->stream));
    }
    return set_strip_state(e, exponent, zero);
}

static int set_strip_state(ctg_exec* e, double exponent, int zero) {
    StripState st{};
    st.E = e->strip ? exponent : -HUGE_VAL;
    st.e_slice = -HUGE_VAL;
    st.coefM = 1.0;
    st.coefm = 0.0;
    st.zero = zero ? 1 : 0;
    HIP_TRY_C(hipMemcpy(e->d_strip, &st, sizeof(st), hipMemcpyHostToDevice));
    return CTG_OK;
}

}  // extern "C"
