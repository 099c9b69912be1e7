/*
 * Drop-in for the reference command line `antidiagonalSmithWaterman <file_path>`
 * (smithWaterman/antidiagonalSmithWaterman.c:189-358): same argument, same input format, same
 * stdout (`line_num: N`, one `Score: s` per pair in file order, `elapsed t`), same exit codes.
 * The per-pair anti-diagonal fill (:254-347) runs on the GPU through libagx (include/agx.h);
 * there is no CPU path.  AGX_NUM_DEVICES=n shards the pairs over n GPUs (default 1, 0 = all).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "agx.h"

static double seconds(void)
{
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return (double)tp.tv_sec + (double)tp.tv_usec * 1.e-6;
}

int main(int argc, char *argv[])
{
    if (argc != 2) {
        fprintf(stderr, "Usage: %s <file_path>\n", argv[0]); /* :190-193 */
        return 1;
    }
    const int trace = getenv("AGX_TRACE_CLI") != NULL;
    double tr0 = seconds();
    agx_sw_text *t = NULL;
    int rc = agx_sw_text_read(argv[1], 0, &t);
    if (trace) fprintf(stderr, "[cli] read+parse %.3f s\n", seconds() - tr0);
    if (rc != AGX_OK) {
        if (strcmp(agx_last_error(), "file is empty") == 0) { /* :205-208 */
            printf("file is empty");
            return 1;
        }
        fprintf(stderr, "%s\n", agx_last_error()); /* perror("Error opening file"), :196-199 */
        exit(EXIT_FAILURE);
    }
    printf("line_num: %d\n", t->line_num); /* :210 */
    double t0 = seconds();                 /* the reference's clock also spans reading + scoring + printing */
    int32_t *scores = (int32_t *)malloc(sizeof(int32_t) * (size_t)(t->n_pairs ? t->n_pairs : 1));
    if (!scores) {
        fprintf(stderr, "out of memory\n");
        return EXIT_FAILURE;
    }
    const char *nd = getenv("AGX_NUM_DEVICES");
    int n_dev = nd ? atoi(nd) : 1;
    rc = agx_sw_score_multi(n_dev, t->bases, t->off, t->len, t->n_pairs, scores);
    if (rc != AGX_OK) {
        fprintf(stderr, "antidiagonalSmithWaterman: %s\n", agx_last_error());
        return EXIT_FAILURE;
    }
    if (trace) fprintf(stderr, "[cli] score %.3f s\n", seconds() - t0);
    tr0 = seconds();
    for (int64_t p = 0; p < t->n_pairs; p++) printf("Score: %d\n", scores[p]); /* :348 */
    if (trace) fprintf(stderr, "[cli] print %.3f s\n", seconds() - tr0);
    if (t->dangling) printf("%s", t->dangling);                                /* :225 */
    printf("elapsed %f\n", seconds() - t0);                                    /* :351-352 */
    free(scores);
    agx_sw_text_free(t);
    return 0;
}
