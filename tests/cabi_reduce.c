/* cabi_reduce.c -- the multi-GPU sequence of the C ABI driven from plain C, no Python:
 *
 *   ctg_plan_create -> ctg_exec_create -> ctg_exec_upload_inputs_host
 *   -> ctg_comm_get_unique_id (rank 0) -> ctg_comm_init
 *   -> ctg_exec_run_slices(first = rank, count = my share, stride = world)
 *   -> ctg_exec_reduce(root = -1: every rank gets the total) -> ctg_exec_download_result
 *
 * which is the reference's `contract_mpi` (cotengra/core.py:4057-4090: round-robin slices,
 * eager local sum, Allreduce) behind include/ctg_hip.h.  Built by __graft_entry__.build()
 * into tests/cabi_reduce (gcc, linked against libctg_hip.so only).
 *
 *   cabi_reduce <plan.bin> [rank world idfile [device]]
 *
 * One process per GPU.  The 128-byte unique id travels through `idfile`: rank 0 writes
 * it (to idfile.tmp, then renames), the others wait for it -- any channel will do
 * (MPI_Bcast under `mpirun -n 8 sh -c 'cabi_reduce plan.bin $OMPI_COMM_WORLD_RANK 8 /tmp/id'`).
 * Without rank / world it runs alone (world = 1; the collective is still executed).
 * The plan file is tests/golden/cabi_plan.bin (tests/golden/gen/make_cabi_plan.py: a
 * sliced 4x4 lattice in complex128 with its reference-frozen result); exit status 0 iff
 * the reduced result matches it to 1e-10.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../include/ctg_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != CTG_OK) {                                                               \
            fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, rc_, ctg_last_error()); \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

static int rank = 0;

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s plan.bin [rank world idfile [device]]\n", argv[0]);
        return 2;
    }
    int world = 1, device = 0;
    const char* idfile = NULL;
    if (argc >= 5) {
        rank = atoi(argv[2]);
        world = atoi(argv[3]);
        idfile = argv[4];
        device = argc >= 6 ? atoi(argv[5]) : rank;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    long nbytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* blob = (char*)malloc((size_t)nbytes);
    if (fread(blob, 1, (size_t)nbytes, f) != (size_t)nbytes) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    const int64_t* w = (const int64_t*)blob;
    if (w[0] != 0x43544750) { fprintf(stderr, "not a plan file\n"); return 2; }
    const int64_t n_inputs = w[2], n_steps = w[6], n_tab = w[7], n_sliced = w[8], nslices = w[9];
    ctg_plan_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = (int32_t)w[1];
    d.n_inputs = n_inputs;
    d.inputs_elems = w[3];
    d.arena_elems = w[4];
    d.result_elems = w[5];
    d.n_steps = n_steps;
    d.n_table_words = n_tab;
    d.n_sliced = n_sliced;
    const int64_t* cur = w + 10;
    d.input_sizes = cur;   cur += n_inputs;
    d.input_offsets = cur; cur += n_inputs;
    d.steps = cur;         cur += n_steps * CTG_STEP_WORDS;
    d.tables = cur;        cur += n_tab;
    d.slice_sizes = cur;   cur += n_sliced;
    d.slice_fixed = cur;   cur += n_sliced;
    d.slice_strides = cur; cur += (n_inputs + 1) * n_sliced;
    if (d.dtype != CTG_C128) { fprintf(stderr, "expected a complex128 plan\n"); return 2; }
    const double* data = (const double*)cur;
    const void** ptrs = (const void**)malloc((size_t)n_inputs * sizeof(void*));
    for (int64_t i = 0; i < n_inputs; ++i) {
        ptrs[i] = data;
        data += 2 * d.input_sizes[i];
    }
    const double* expected = data;

    ctg_plan* plan = NULL;
    ctg_exec* ex = NULL;
    ctg_comm* comm = NULL;
    CHECK(ctg_plan_create(&d, &plan));
    int64_t ns = 0;
    CHECK(ctg_plan_nslices(plan, &ns));
    if (ns != nslices || ns < world) { fprintf(stderr, "%lld slices for %d ranks\n", (long long)ns, world); return 2; }
    CHECK(ctg_exec_create(plan, device, NULL, NULL, &ex));
    CHECK(ctg_exec_upload_inputs_host(ex, ptrs));

    /* the unique id: made on rank 0, handed over through a file */
    unsigned char id[CTG_UNIQUE_ID_BYTES];
    if (rank == 0) {
        CHECK(ctg_comm_get_unique_id(id));
        if (idfile) {
            char tmp[4096];
            snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
            FILE* g = fopen(tmp, "wb");
            if (!g || fwrite(id, 1, sizeof(id), g) != sizeof(id)) { perror(tmp); return 2; }
            fclose(g);
            if (rename(tmp, idfile) != 0) { perror(idfile); return 2; }
        }
    } else {
        FILE* g = NULL;
        for (int tries = 0; tries < 6000 && !(g = fopen(idfile, "rb")); ++tries) usleep(10000);
        if (!g || fread(id, 1, sizeof(id), g) != sizeof(id)) { fprintf(stderr, "rank %d: no unique id in %s\n", rank, idfile); return 2; }
        fclose(g);
    }
    CHECK(ctg_comm_init(id, rank, world, device, &comm));

    /* my share of the slices as the library deals them -- whole slice groups rank, rank + world, ...; single
     * slices round-robin (core.py:4070) for a plan without group indices --, accumulated on the device; then
     * ONE collective */
    int64_t units = 0, per_unit = 1;
    CHECK(ctg_plan_share_units(plan, rank, world, &units, &per_unit));
    const int64_t mine = units * per_unit;
    CHECK(ctg_exec_zero_result(ex));
    CHECK(ctg_exec_run_share(ex, rank, world, 0, units));
    CHECK(ctg_exec_reduce(ex, comm, -1));
    double* out = (double*)malloc((size_t)d.result_elems * 2 * sizeof(double));
    CHECK(ctg_exec_download_result(ex, out));

    double err = 0.0, scale = 0.0;
    for (int64_t i = 0; i < 2 * d.result_elems; ++i) {
        err = fmax(err, fabs(out[i] - expected[i]));
        scale = fmax(scale, fabs(expected[i]));
    }
    printf("rank %d of %d on device %d: %lld of %lld slices, result (%.12g, %.12g), rel err %.2e\n", rank, world,
           device, (long long)mine, (long long)nslices, out[0], out[1], err / scale);
    CHECK(ctg_comm_destroy(comm));
    CHECK(ctg_exec_destroy(ex));
    CHECK(ctg_plan_destroy(plan));
    return err <= 1e-10 * scale ? 0 : 1;
}
