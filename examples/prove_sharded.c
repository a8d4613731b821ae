/* C host for ONE proof over several GPUs, no Python / torch: the collectives are issued by libdistaff_hip.so (dst_prove_sharded).
 *
 *   cc -O2 -Iinclude examples/prove_sharded.c -Ldistaff_amd -ldistaff_hip -Wl,-rpath,$PWD/distaff_amd -o prove_sharded
 *
 *   one process per GPU over RCCL (rank 0 writes the 128-byte unique id to <idfile>, the others wait for it):
 *       for r in 0 1 2 3 4 5 6 7; do ./prove_sharded rccl $r 8 20 /tmp/distaff.id proof_$r.bin & done; wait
 *   one process driving every GPU itself (one thread per rank inside the library, device = rank modulo <devices>):
 *       ./prove_sharded local 8 20 8 proof.bin
 *
 * Every rank generates the same trace (standing in for the reference VM, processor::execute) and uploads only the registers it
 * interpolates, r = rank (mod world) (dst_trace_upload_owned): the library all-gathers the coefficient vectors and every rank extends
 * all registers over its own cosets. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "distaff_hip.h"

static dst_ctx* make_ctx(unsigned log_n, unsigned rank, unsigned world, int device, const uint8_t* cols) {
    dst_params p;
    memset(&p, 0, sizeof p);
    p.log_trace_length = log_n; p.log_blowup = 5; p.width = 20; p.ctx_depth = 1; p.loop_depth = 0;
    p.num_queries = 50; p.grinding_factor = 20; p.device = device; p.rank = rank; p.world = world;      /* ProofOptions::default() */
    dst_ctx* ctx = NULL;
    int rc = dst_ctx_create(&p, &ctx);
    if (rc != DST_OK) { fprintf(stderr, "rank %u: dst_ctx_create: %d %s\n", rank, rc, dst_last_error(ctx)); exit(1); }
    const uint8_t* ptrs[20];
    for (unsigned r = 0; r < 20; r++) ptrs[r] = (r % world == rank) ? cols + ((size_t)r << log_n) * 16 : NULL;      /* the others are not read */
    if ((rc = dst_trace_upload_owned(ctx, ptrs)) != DST_OK) { fprintf(stderr, "rank %u: upload: %s\n", rank, dst_last_error(ctx)); exit(1); }
    return ctx;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s rccl <rank> <world> <log_n> <idfile> [out]  |  %s local <world> <log_n> <devices> [out]\n", argv[0], argv[0]); return 2; }
    const int rccl = strcmp(argv[1], "rccl") == 0;
    const unsigned rank = rccl ? (unsigned)atoi(argv[2]) : 0, world = (unsigned)atoi(argv[rccl ? 3 : 2]), log_n = (unsigned)atoi(argv[rccl ? 4 : 3]);
    const char* out_path = argc > (rccl ? 6 : 5) ? argv[rccl ? 6 : 5] : NULL;
    const size_t n = (size_t)1 << log_n, W = 20, cap = (size_t)1 << 22;
    uint8_t* cols = (uint8_t*)malloc(W * n * 16);
    uint8_t program_hash[32], result[16];
    if (!cols || dst_fibonacci_trace(log_n, cols, program_hash, result) != DST_OK) { fprintf(stderr, "trace generation failed\n"); return 2; }
    dst_public pub;
    memset(&pub, 0, sizeof pub);
    pub.num_inputs = 2; pub.num_outputs = 1;
    pub.inputs[0][0] = 1;
    memcpy(pub.outputs[0], result, 16);
    uint8_t* proof = (uint8_t*)malloc(cap);
    size_t len = 0;
    int rc;
    if (rccl) {
        const char* idfile = argv[5];
        uint8_t id[128];
        if (rank == 0) {
            if (dst_comm_unique_id(id) != DST_OK) { fprintf(stderr, "dst_comm_unique_id: %s\n", dst_comm_last_error(NULL)); return 1; }
            char tmp[1024]; snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, 128, f) != 128) { fprintf(stderr, "cannot write %s\n", tmp); return 1; }
            fclose(f);
            rename(tmp, idfile);                               /* appears atomically */
        } else {
            FILE* f = NULL;
            for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); tries++) usleep(100000);
            if (!f || fread(id, 1, 128, f) != 128) { fprintf(stderr, "rank %u: no unique id in %s\n", rank, idfile); return 1; }
            fclose(f);
        }
        dst_ctx* ctx = make_ctx(log_n, rank, world, (int)rank, cols);
        dst_comm* comm = NULL;
        if (dst_comm_init(id, rank, world, (int)rank, &comm) != DST_OK) { fprintf(stderr, "rank %u: dst_comm_init: %s\n", rank, dst_comm_last_error(NULL)); return 1; }
        dst_comm_info info;                                    /* what RCCL itself says about the communicator it built */
        if (dst_comm_describe(comm, &info) == DST_OK)
            fprintf(stderr, "rank %u: RCCL %u connected %u ranks, this rank is %u on device %d\n", rank, info.rccl_version, info.rccl_ranks, info.rccl_rank, (int)info.device);
        /* no rank waits for ever: after 120 s without completion behind a collective (a peer that died or never started) the rank aborts its
           communicator and dst_prove_sharded returns DST_ERR_COMM; dst_comm_last_error names the collective it was stuck behind */
        dst_comm_set_timeout(comm, 120.0);
        rc = dst_prove_sharded(ctx, comm, &pub, proof, cap, &len);
        if (rc != DST_OK) { fprintf(stderr, "rank %u: dst_prove_sharded: %d %s%s%s\n", rank, rc, dst_last_error(ctx), rc == DST_ERR_COMM ? " | " : "", rc == DST_ERR_COMM ? dst_comm_last_error(comm) : ""); return 1; }
        dst_comm_destroy(comm);
        dst_ctx_destroy(ctx);
    } else {
        const unsigned devices = (unsigned)atoi(argv[4]);
        dst_ctx* ctxs[8];
        if (world > 8 || devices == 0) { fprintf(stderr, "world <= 8, devices >= 1\n"); return 2; }
        for (unsigned r = 0; r < world; r++) ctxs[r] = make_ctx(log_n, r, world, (int)(r % devices), cols);
        rc = dst_prove_sharded_local(ctxs, world, &pub, proof, cap, &len);
        if (rc != DST_OK) { fprintf(stderr, "dst_prove_sharded_local: %d %s\n", rc, dst_last_error(ctxs[0])); return 1; }
        for (unsigned r = 0; r < world; r++) dst_ctx_destroy(ctxs[r]);
    }
    printf("rank %u of %u: 2^%u steps, proof %zu bytes\n", rank, world, log_n, len);
    if (out_path) {
        FILE* f = fopen(out_path, "wb");
        if (!f || fwrite(proof, 1, len, f) != len) { fprintf(stderr, "cannot write %s\n", out_path); return 1; }
        fclose(f);
    }
    free(proof); free(cols);
    return 0;
}
