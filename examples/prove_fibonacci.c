/* Minimal C host for libdistaff_hip.so: what a binding in the reference's own language does, without Python.
 *
 *   cc -O2 -Iinclude examples/prove_fibonacci.c -Ldistaff_amd -ldistaff_hip -Wl,-rpath,$PWD/distaff_amd -o prove_fibonacci
 *   ./prove_fibonacci 16 proof.bin          # 2^16-step Fibonacci trace, default ProofOptions, writes bincode(StarkProof)
 *
 * Variant B of INTEGRATION.md: dst_ctx_create, dst_trace_upload_contiguous, dst_prove.  The trace comes from the library's own
 * host-side generator (dst_fibonacci_trace), standing in for the reference VM (processor::execute). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "distaff_hip.h"

int main(int argc, char** argv) {
    unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 10;
    const char* out_path = argc > 2 ? argv[2] : NULL;
    size_t n = (size_t)1 << log_n, W = 20;
    uint8_t* cols = (uint8_t*)malloc(W * n * 16);
    uint8_t program_hash[32], result[16];
    if (!cols || dst_fibonacci_trace(log_n, cols, program_hash, result) != DST_OK) { fprintf(stderr, "trace generation failed\n"); return 2; }

    dst_params p;
    memset(&p, 0, sizeof p);
    p.log_trace_length = log_n; p.log_blowup = 5; p.width = (uint32_t)W; p.ctx_depth = 1; p.loop_depth = 0;
    p.num_queries = 50; p.grinding_factor = 20; p.device = 0; p.rank = 0; p.world = 1;        /* ProofOptions::default() */
    dst_ctx* ctx = NULL;
    int rc = dst_ctx_create(&p, &ctx);
    if (rc != DST_OK) { fprintf(stderr, "dst_ctx_create: %d %s\n", rc, ctx ? dst_last_error(ctx) : "(no context)"); return 1; }
    if ((rc = dst_trace_upload_contiguous(ctx, cols)) != DST_OK) { fprintf(stderr, "upload: %s\n", dst_last_error(ctx)); return 1; }

    dst_public pub;
    memset(&pub, 0, sizeof pub);
    pub.num_inputs = 2; pub.num_outputs = 1;
    pub.inputs[0][0] = 1;                                   /* public inputs [1, 0] */
    memcpy(pub.outputs[0], result, 16);
    size_t cap = (size_t)1 << 22, len = 0;
    uint8_t* proof = (uint8_t*)malloc(cap);
    if ((rc = dst_prove(ctx, &pub, proof, cap, &len)) != DST_OK) { fprintf(stderr, "dst_prove: %d %s\n", rc, dst_last_error(ctx)); return 1; }
    double ms[9];
    dst_phase_ms(ctx, ms);
    double total = 0;
    for (int i = 0; i < 9; i++) total += ms[i];
    printf("2^%u steps: proof %zu bytes, phases %.2f ms (lde %.2f, merkle %.2f, constraints %.2f, combine %.2f, constraint tree %.2f, deep %.2f, fri %.2f, pow %.2f, openings %.2f)\n",
           log_n, len, total, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7], ms[8]);
    if (out_path) {
        FILE* f = fopen(out_path, "wb");
        if (!f || fwrite(proof, 1, len, f) != len) { fprintf(stderr, "cannot write %s\n", out_path); return 1; }
        fclose(f);
    }
    dst_ctx_destroy(ctx);
    free(proof); free(cols);
    return 0;
}
