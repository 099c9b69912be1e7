/*
 * Drop-in for the reference command line `antidiagonalSmithWaterman <file_path>`
 * (smithWaterman/antidiagonalSmithWaterman.c:189-358): same argument, same input format, same
 * stdout (`line_num: N`, one `Score: s` per pair in file order, `elapsed t`), same exit codes.
 * The per-pair anti-diagonal fill (:254-347) runs on the GPU through libagx (include/agx.h);
 * there is no CPU path.
 *   AGX_NUM_DEVICES = n   shard the pairs over GPUs 0..n-1 (default 1, 0 = all visible)
 *   AGX_DEVICES = 0,0,1   explicit list, one shard per entry (an ordinal may repeat)
 *   AGX_CLI_CHUNK_PAIRS   pairs per pipeline step (default 262144)
 *
 * Streaming: three stages on three threads -- the parser reads chunk k+1 while the
 * main thread has chunk k scored on the device and the printer formats and writes chunk k-1 -- and the
 * HIP runtime comes up on a fourth while the first chunk is parsed.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#include "agx.h"
#include "agx_pipe.h"

static double seconds(void)
{
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return (double)tp.tv_sec + (double)tp.tv_usec * 1.e-6;
}

typedef struct {
    agx_sw_reader *reader;
    int64_t chunk_pairs;
    agx_pipe q;
    int rc;
    char err[512];
} parse_stage;

static void *parser_main(void *arg)
{
    parse_stage *s = (parse_stage *)arg;
    while (!agx_sw_reader_done(s->reader)) {
        agx_sw_text *t = NULL;
        const int rc = agx_sw_reader_next(s->reader, s->chunk_pairs, &t);
        if (rc != AGX_OK) {
            s->rc = rc;
            snprintf(s->err, sizeof s->err, "%s", agx_last_error());
            break;
        }
        agx_pipe_push(&s->q, t);
    }
    agx_pipe_close(&s->q);
    return NULL;
}

typedef struct {
    agx_sw_text *text; /* for n_pairs and the dangling line */
    int32_t *scores;
} scored_chunk;

typedef struct {
    agx_pipe q;
    double t_print;
} print_stage;

static void *printer_main(void *arg)
{
    print_stage *s = (print_stage *)arg;
    for (;;) {
        scored_chunk *c = (scored_chunk *)agx_pipe_pop(&s->q);
        if (!c) break;
        const double ta = seconds();
        /* "Score: " + at most 11 characters + newline per pair, formatted into one block */
        const int64_t n = c->text->n_pairs;
        char *buf = (char *)malloc((size_t)n * 20 + 16);
        if (buf) {
            size_t at = 0;
            for (int64_t p = 0; p < n; p++) at += (size_t)sprintf(buf + at, "Score: %d\n", c->scores[p]); /* :348 */
            fwrite(buf, 1, at, stdout);
            free(buf);
        } else
            for (int64_t p = 0; p < n; p++) printf("Score: %d\n", c->scores[p]);
        if (c->text->dangling) printf("%s", c->text->dangling); /* :225 */
        s->t_print += seconds() - ta;
        free(c->scores);
        agx_sw_text_free(c->text);
        free(c);
    }
    return NULL;
}

typedef struct {
    const int *devices; /* NULL: devices 0 .. n-1 */
    int n;
    int rc;
    char err[512];
} warm_t;

static void *warm_main(void *arg)
{
    warm_t *w = (warm_t *)arg;
    w->rc = agx_warmup_devices(w->devices, w->n); /* brings the HIP runtime and the contexts up */
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
    int devices[64];
    int n_dev = agx_parse_devices(getenv("AGX_DEVICES"), devices, 64);
    const char *nd = getenv("AGX_NUM_DEVICES");
    const int n_multi = nd ? atoi(nd) : 1; /* used when AGX_DEVICES is not set */
    const char *cp = getenv("AGX_CLI_CHUNK_PAIRS");
    parse_stage ps;
    memset(&ps, 0, sizeof ps);
    ps.chunk_pairs = cp && atoll(cp) > 0 ? atoll(cp) : 262144;
    int rc = agx_sw_reader_open(argv[1], 0, &ps.reader);
    if (rc != AGX_OK) {
        if (strcmp(agx_last_error(), "file is empty") == 0) { /* :205-208 */
            printf("file is empty");
            return 1;
        }
        fprintf(stderr, "%s\n", agx_last_error()); /* perror("Error opening file"), :196-199 */
        exit(EXIT_FAILURE);
    }
    if (getenv("AGX_CLI_PARSE_THREADS")) agx_sw_reader_set_threads(ps.reader, atoi(getenv("AGX_CLI_PARSE_THREADS")));
    agx_pipe_init(&ps.q, 2);
    print_stage pr;
    memset(&pr, 0, sizeof pr);
    agx_pipe_init(&pr.q, 2);
    pthread_t parser, warmer, printer;
    warm_t warm;
    memset(&warm, 0, sizeof warm);
    warm.devices = n_dev ? devices : NULL;
    warm.n = n_dev ? n_dev : n_multi;
    printf("line_num: %d\n", agx_sw_reader_line_num(ps.reader)); /* :210 */
    fflush(stdout);
    if (pthread_create(&parser, NULL, parser_main, &ps) || pthread_create(&printer, NULL, printer_main, &pr) ||
        pthread_create(&warmer, NULL, warm_main, &warm)) {
        fprintf(stderr, "antidiagonalSmithWaterman: cannot start threads\n");
        return EXIT_FAILURE;
    }
    const double t0 = seconds(); /* the reference's clock also spans reading + scoring + printing */
    int warm_joined = 0;
    double t_wait = 0, t_score = 0;
    int64_t n_chunks = 0;
    int status = 0;
    for (;;) {
        double ta = seconds();
        agx_sw_text *t = (agx_sw_text *)agx_pipe_pop(&ps.q);
        t_wait += seconds() - ta;
        if (!t) break;
        if (!warm_joined) {
            pthread_join(warmer, NULL);
            warm_joined = 1;
            if (warm.rc != AGX_OK) {
                fprintf(stderr, "antidiagonalSmithWaterman: %s\n", warm.err);
                status = EXIT_FAILURE;
            }
        }
        ta = seconds();
        scored_chunk *c = (scored_chunk *)calloc(1, sizeof *c);
        int32_t *scores = (int32_t *)malloc(sizeof(int32_t) * (size_t)(t->n_pairs ? t->n_pairs : 1));
        if (!c || !scores) {
            fprintf(stderr, "out of memory\n");
            status = EXIT_FAILURE;
        }
        if (!status) {
            if (n_dev) rc = agx_sw_score_devices(devices, n_dev, t->bases, t->off, t->len, t->n_pairs, scores);
            else rc = agx_sw_score_multi(n_multi, t->bases, t->off, t->len, t->n_pairs, scores);
            if (rc != AGX_OK) {
                fprintf(stderr, "antidiagonalSmithWaterman: %s\n", agx_last_error());
                status = EXIT_FAILURE;
            }
        }
        t_score += seconds() - ta;
        if (status) { /* drain the parser so it can finish, print nothing more */
            free(scores);
            free(c);
            agx_sw_text_free(t);
            while ((t = (agx_sw_text *)agx_pipe_pop(&ps.q)) != NULL) agx_sw_text_free(t);
            break;
        }
        c->text = t;
        c->scores = scores;
        agx_pipe_push(&pr.q, c);
        n_chunks++;
    }
    pthread_join(parser, NULL);
    agx_pipe_close(&pr.q);
    pthread_join(printer, NULL);
    if (status) return status;
    if (ps.rc != AGX_OK) {
        fprintf(stderr, "antidiagonalSmithWaterman: %s\n", ps.err);
        return EXIT_FAILURE;
    }
    if (!warm_joined) pthread_join(warmer, NULL); /* no pair in the file: nothing needed the device */
    printf("elapsed %f\n", seconds() - t0); /* :351-352 */
    if (trace)
        fprintf(stderr, "[cli] %lld chunk(s): waited for the parser %.3f s, score %.3f s, printer busy %.3f s, total %.3f s\n",
                (long long)n_chunks, t_wait, t_score, pr.t_print, seconds() - tr0);
    agx_sw_reader_close(ps.reader);
    /* everything is written: leave without tearing the HIP runtime down (tens of milliseconds of destructors
     * that free what the driver frees anyway) */
    fflush(NULL);
    _exit(0);
}
