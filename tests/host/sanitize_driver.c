/*
 * Host-only driver for the sanitizer build (tests/test_host_sanitizers.py): exercises the text
 * readers and the plan-only batch constructors of libagx under AddressSanitizer + UBSan.
 * No device is touched (ctx == NULL everywhere).  usage: sanitize_driver <golden dir>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "agx.h"

static int fails = 0;
#define EXPECT(c)                                                   \
    do {                                                            \
        if (!(c)) {                                                 \
            fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            fails++;                                                \
        }                                                           \
    } while (0)

static void sw_file(const char *dir, const char *name, int line_buf)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    agx_sw_text *t = NULL;
    int rc = agx_sw_text_read(path, line_buf, &t);
    EXPECT(rc == AGX_OK && t);
    if (!t) return;
    agx_sw_batch *b = NULL;
    rc = agx_sw_batch_create(NULL, t->bases, t->off, t->len, t->n_pairs, &b);
    EXPECT(rc == AGX_OK && b);
    agx_sw_info info;
    if (b) {
        EXPECT(agx_sw_batch_info(b, &info) == AGX_OK && info.n_pairs == t->n_pairs && info.padded_cells >= info.cells);
        EXPECT(agx_sw_batch_launch(b) == AGX_E_NODEVICE);
        agx_sw_batch_destroy(b);
    }
    agx_sw_scoring sc = {3, -2, -6, -1};
    rc = agx_sw_batch_create_scored(NULL, &sc, t->bases, t->off, t->len, t->n_pairs, &b);
    EXPECT(rc == AGX_OK);
    agx_sw_batch_destroy(b);
    b = NULL;
    /* substitution-matrix mode over every byte the file holds (newline included) */
    agx_sw_matrix m;
    memset(&m, 0, sizeof m);
    m.n_symbols = 6;
    m.gap_open = -4;
    m.gap_extend = -1;
    memset(m.code, 0xff, sizeof m.code);
    const char *alpha = "ACGTN\n";
    for (int k = 0; k < 6; k++) m.code[(unsigned char)alpha[k]] = (uint8_t)k;
    for (int a = 0; a < 6; a++)
        for (int c = 0; c < 6; c++) m.score[a][c] = (int8_t)(a == c ? 3 : -2);
    rc = agx_sw_batch_create_matrix(NULL, &m, t->bases, t->off, t->len, t->n_pairs, &b);
    EXPECT(rc == AGX_OK || rc == AGX_E_SYMBOL || rc == AGX_E_LIMIT); /* split lines / other bytes / long lines */
    agx_sw_batch_destroy(b);
    /* the same file through the chunked reader: same number of pairs */
    agx_sw_reader *r = NULL;
    EXPECT(agx_sw_reader_open(path, line_buf, &r) == AGX_OK && r);
    int64_t n = 0;
    while (r && !agx_sw_reader_done(r)) {
        agx_sw_text *c = NULL;
        EXPECT(agx_sw_reader_next(r, 3, &c) == AGX_OK && c);
        if (!c) break;
        EXPECT(c->n_pairs <= 3 && c->line_num == t->line_num);
        n += c->n_pairs;
        agx_sw_text_free(c);
    }
    EXPECT(n == t->n_pairs);
    agx_sw_reader_close(r);
    agx_sw_text_free(t);
}

static void phmm_file(const char *dir, const char *name)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    agx_phmm_text *t = NULL;
    int rc = agx_phmm_text_read(path, &t);
    EXPECT(rc == AGX_OK && t);
    if (!t) return;
    for (int prec = 0; prec < 4; prec++) {
        agx_phmm_batch *b = NULL;
        rc = agx_phmm_batch_create(NULL, &t->desc, prec | (prec == 1 ? AGX_PHMM_GATK_PRIOR : 0), &b);
        EXPECT(rc == AGX_OK && b);
        agx_phmm_info info;
        if (b) {
            EXPECT(agx_phmm_batch_info(b, &info) == AGX_OK && info.n_pairs == t->n_pairs);
            agx_phmm_batch_destroy(b);
        }
    }
    agx_phmm_text_free(t);
}

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
    const char *sw[] = {"sw_kat.in", "sw_nofinalnl.in", "sw_150.in", "sw_mixed.in", "sw_long.in", "sw_short.in",
                        "sw_hdr_half.in", "sw_hdr_odd.in", "sw_hdr_big.in", "sw_oddlines.in"};
    for (size_t i = 0; i < sizeof sw / sizeof sw[0]; i++) {
        sw_file(argv[1], sw[i], 0);
        sw_file(argv[1], sw[i], 10000);
        sw_file(argv[1], sw[i], 16); /* tiny buffer: every line splits many times */
    }
    const char *ph[] = {"phmm_test.in", "phmm_10s.in", "phmm_synth.in", "phmm_far.in", "phmm_long.in"};
    for (size_t i = 0; i < sizeof ph / sizeof ph[0]; i++) phmm_file(argv[1], ph[i]);
    /* wrong formats fed to each reader must fail or parse without touching invalid memory */
    agx_phmm_text *pt = NULL;
    char path[1024];
    snprintf(path, sizeof path, "%s/sw_kat.in", argv[1]);
    int rc = agx_phmm_text_read(path, &pt);
    EXPECT(rc == AGX_OK || rc == AGX_E_IO);
    agx_phmm_text_free(pt);
    agx_sw_text *st = NULL;
    snprintf(path, sizeof path, "%s/phmm_10s.in", argv[1]);
    rc = agx_sw_text_read(path, 0, &st);
    EXPECT(rc == AGX_OK);
    if (st) {
        agx_sw_batch *b = NULL;
        rc = agx_sw_batch_create(NULL, st->bases, st->off, st->len, st->n_pairs, &b);
        EXPECT(rc == AGX_OK);
        agx_sw_batch_destroy(b);
    }
    agx_sw_text_free(st);
    EXPECT(agx_sw_text_read("/nonexistent/file", 0, &st) == AGX_E_IO);
    EXPECT(agx_sw_batch_create(NULL, NULL, NULL, NULL, -1, NULL) == AGX_E_ARG);
    printf(fails ? "SANITIZE_DRIVER_FAILED %d\n" : "SANITIZE_DRIVER_OK\n", fails);
    return fails ? 1 : 0;
}
