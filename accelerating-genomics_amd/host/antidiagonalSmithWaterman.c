/*
 * Drop-in for the reference command line `antidiagonalSmithWaterman <file_path>`
 * (smithWaterman/antidiagonalSmithWaterman.c:189-358): same argument, same input format, same
 * stdout (`line_num: N`, one `Score: s` per pair in file order, `elapsed t`), same exit codes.
 * The per-pair anti-diagonal fill (:254-347) runs on the GPU through libagx (include/agx.h);
 * there is no CPU path.  AGX_NUM_DEVICES=n shards the pairs over n GPUs (default 1, 0 = all).
 *
 * Streaming: a parser thread reads the file in chunks of AGX_CLI_CHUNK_PAIRS
 * pairs (default 262144) while the main thread packs, scores and prints the previous chunk, and
 * the HIP runtime comes up on a third thread while the first chunk is parsed.
 */
#include <pthread.h>
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

/* two-slot queue between the parser thread and main */
typedef struct {
    agx_sw_reader *reader;
    int64_t chunk_pairs;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    agx_sw_text *slot[2];
    int count, closed, rc;
    char err[512];
} pipe_t;

static void *parser_main(void *arg)
{
    pipe_t *q = (pipe_t *)arg;
    int rc = AGX_OK;
    while (rc == AGX_OK && !agx_sw_reader_done(q->reader)) {
        agx_sw_text *t = NULL;
        rc = agx_sw_reader_next(q->reader, q->chunk_pairs, &t);
        pthread_mutex_lock(&q->mu);
        if (rc != AGX_OK) {
            q->rc = rc;
            snprintf(q->err, sizeof q->err, "%s", agx_last_error());
        } else {
            while (q->count == 2) pthread_cond_wait(&q->cv, &q->mu);
            q->slot[q->count++] = t;
            pthread_cond_broadcast(&q->cv);
        }
        pthread_mutex_unlock(&q->mu);
    }
    pthread_mutex_lock(&q->mu);
    q->closed = 1;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
    return NULL;
}

static agx_sw_text *pipe_pop(pipe_t *q)
{
    pthread_mutex_lock(&q->mu);
    while (q->count == 0 && !q->closed) pthread_cond_wait(&q->cv, &q->mu);
    agx_sw_text *t = NULL;
    if (q->count) {
        t = q->slot[0];
        q->slot[0] = q->slot[1];
        q->count--;
        pthread_cond_broadcast(&q->cv);
    }
    pthread_mutex_unlock(&q->mu);
    return t;
}

typedef struct {
    agx_ctx *ctx;
    int rc;
    char err[512];
} warm_t;

static void *warm_main(void *arg)
{
    warm_t *w = (warm_t *)arg;
    w->rc = agx_ctx_create(0, &w->ctx); /* brings the HIP runtime up */
    if (w->rc != AGX_OK) snprintf(w->err, sizeof w->err, "%s", agx_last_error());
    return NULL;
}

int main(int argc, char *argv[])
{
    if (argc != 2) {
        fprintf(stderr, "Usage: %s <file_path>\n", argv[0]); /* :190-193 */
        return 1;
    }
    const int trace = getenv("AGX_TRACE_CLI") != NULL;
    const double tr0 = seconds();
    const char *nd = getenv("AGX_NUM_DEVICES");
    const int n_dev = nd ? atoi(nd) : 1;
    const char *cp = getenv("AGX_CLI_CHUNK_PAIRS");
    pipe_t q;
    memset(&q, 0, sizeof q);
    q.chunk_pairs = cp && atoll(cp) > 0 ? atoll(cp) : 262144;
    int rc = agx_sw_reader_open(argv[1], 0, &q.reader);
    if (rc != AGX_OK) {
        if (strcmp(agx_last_error(), "file is empty") == 0) { /* :205-208 */
            printf("file is empty");
            return 1;
        }
        fprintf(stderr, "%s\n", agx_last_error()); /* perror("Error opening file"), :196-199 */
        exit(EXIT_FAILURE);
    }
    pthread_mutex_init(&q.mu, NULL);
    pthread_cond_init(&q.cv, NULL);
    pthread_t parser, warmer;
    warm_t warm;
    memset(&warm, 0, sizeof warm);
    if (pthread_create(&parser, NULL, parser_main, &q) || pthread_create(&warmer, NULL, warm_main, &warm)) {
        fprintf(stderr, "antidiagonalSmithWaterman: cannot start threads\n");
        return EXIT_FAILURE;
    }
    printf("line_num: %d\n", agx_sw_reader_line_num(q.reader)); /* :210 */
    const double t0 = seconds(); /* the reference's clock also spans reading + scoring + printing */
    int warm_joined = 0;
    double t_wait = 0, t_score = 0, t_print = 0;
    int64_t n_chunks = 0;
    for (;;) {
        double ta = seconds();
        agx_sw_text *t = pipe_pop(&q);
        t_wait += seconds() - ta;
        if (!t) break;
        if (!warm_joined) {
            pthread_join(warmer, NULL);
            warm_joined = 1;
            if (warm.rc != AGX_OK) {
                fprintf(stderr, "antidiagonalSmithWaterman: %s\n", warm.err);
                return EXIT_FAILURE;
            }
        }
        ta = seconds();
        int32_t *scores = (int32_t *)malloc(sizeof(int32_t) * (size_t)(t->n_pairs ? t->n_pairs : 1));
        if (!scores) {
            fprintf(stderr, "out of memory\n");
            return EXIT_FAILURE;
        }
        rc = n_dev == 1 ? agx_sw_score(warm.ctx, t->bases, t->off, t->len, t->n_pairs, scores)
                        : agx_sw_score_multi(n_dev, t->bases, t->off, t->len, t->n_pairs, scores);
        if (rc != AGX_OK) {
            fprintf(stderr, "antidiagonalSmithWaterman: %s\n", agx_last_error());
            return EXIT_FAILURE;
        }
        t_score += seconds() - ta;
        ta = seconds();
        for (int64_t p = 0; p < t->n_pairs; p++) printf("Score: %d\n", scores[p]); /* :348 */
        if (t->dangling) printf("%s", t->dangling);                                /* :225 */
        t_print += seconds() - ta;
        free(scores);
        agx_sw_text_free(t);
        n_chunks++;
    }
    pthread_join(parser, NULL);
    if (q.rc != AGX_OK) {
        fprintf(stderr, "antidiagonalSmithWaterman: %s\n", q.err);
        return EXIT_FAILURE;
    }
    if (!warm_joined) pthread_join(warmer, NULL); /* no pair in the file: nothing needed the device */
    printf("elapsed %f\n", seconds() - t0); /* :351-352 */
    if (trace)
        fprintf(stderr, "[cli] %lld chunk(s): waited for the parser %.3f s, score %.3f s, print %.3f s, total %.3f s\n",
                (long long)n_chunks, t_wait, t_score, t_print, seconds() - tr0);
    if (warm.ctx) agx_ctx_destroy(warm.ctx);
    agx_sw_reader_close(q.reader);
    return 0;
}
