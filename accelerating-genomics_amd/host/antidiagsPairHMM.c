/*
 * Drop-in for the reference command line `antidiagsPairHMM <input_file_r> <output_file>`
 * (pairHMM/antidiagsPairHMM.c:307-500): same arguments, same input format, same output file
 * (one "%f\n" per (read, haplotype), read-major inside a batch), same stdout chatter
 * (`#batch: k` at the top of every loop turn, including the one that meets EOF, and every
 * value), same failure behaviour on a truncated batch.  pairHMM() (:120-267) runs on the GPU
 * through libagx (include/agx.h); there is no CPU path.
 * Built a second time as `pairHMMmatrix` (-DAGX_PHMM_MATRIX_STDOUT): the reference's row-major
 * program pairHMM/pairHMMmatrix.c has the same command line and output file but prints only the
 * `#batch:` lines on stdout (:171 vs its fprintf at :258).
 *   AGX_PHMM_PRECISION = f64 (default; raw sums bit-identical to the reference) | f64fma | f32 | f32fma
 *   AGX_NUM_DEVICES    = n GPUs to shard whole batches over (default 1, 0 = all)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "agx.h"

int main(int argc, const char *argv[])
{
    if (argc != 3) {
        fprintf(stderr, "Usage: %s <input_file_r> <output_file>\n", argv[0]); /* :314-317 */
        return EXIT_FAILURE;
    }
    agx_phmm_text *t = NULL;
    int rc = agx_phmm_text_read(argv[1], &t);
    if (rc != AGX_OK) {
        fprintf(stderr, "%s\n", agx_last_error()); /* perror("Error opening input file_r"), :320-324 */
        return EXIT_FAILURE;
    }
    FILE *out = fopen(argv[2], "w");
    if (!out) {
        perror("Error opening output file"); /* :333-339 */
        return EXIT_FAILURE;
    }
    int precision = AGX_PHMM_F64;
    const char *pe = getenv("AGX_PHMM_PRECISION");
    if (pe && strcmp(pe, "f32") == 0) precision = AGX_PHMM_F32;
    else if (pe && strcmp(pe, "f32fma") == 0) precision = AGX_PHMM_F32_FMA;
    else if (pe && strcmp(pe, "f64fma") == 0) precision = AGX_PHMM_F64_FMA;
    const char *nd = getenv("AGX_NUM_DEVICES");
    int n_dev = nd ? atoi(nd) : 1;

    double *lh = (double *)malloc(sizeof(double) * (size_t)(t->n_pairs ? t->n_pairs : 1));
    if (!lh) {
        fprintf(stderr, "Error allocating memory for matrices.\n");
        return EXIT_FAILURE;
    }
    rc = agx_phmm_forward_multi(n_dev, &t->desc, precision, lh);
    if (rc != AGX_OK) {
        fprintf(stderr, "antidiagsPairHMM: %s\n", agx_last_error());
        return EXIT_FAILURE;
    }
    int64_t k = 0;
    const agx_phmm_desc *d = &t->desc;
    for (uint32_t g = 0; g < d->n_regions; g++) {
        printf("#batch: %u\n", g + 1); /* :372 */
        int64_t n = (int64_t)(d->region_read[g + 1] - d->region_read[g]) * (d->region_hap[g + 1] - d->region_hap[g]);
        for (int64_t i = 0; i < n; i++, k++) {
#ifndef AGX_PHMM_MATRIX_STDOUT
            printf("%f\n", lh[k]);       /* :459 */
#endif
            fprintf(out, "%f\n", lh[k]); /* :461 */
        }
    }
    printf("#batch: %u\n", d->n_regions + 1); /* the turn that meets EOF, or the truncated batch */
    int status = EXIT_SUCCESS;
    if (t->truncated) {
        fprintf(stderr, "Error reading haplotypes.\n"); /* :394-398 */
        status = EXIT_FAILURE;
    }
    fclose(out);
    free(lh);
    agx_phmm_text_free(t);
    return status;
}
