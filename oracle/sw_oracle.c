/*
 * oracle/sw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's Smith-Waterman affine-gap score-only fill
 * (reference: smithWaterman/antidiagonalSmithWaterman.c).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this
 * file; the product (libagx.so, the drop-in CLIs) never links or loads it.
 *
 * Parity status: PINNED.  tests/test_oracle_sw.py checks both functions below
 * against tests/golden/sw_*.expect, which were produced in the authoring
 * container by oracle/_ref/sw_ref = `gcc -O3` of the unmodified reference
 * source (recipe: oracle/Makefile, generator: tests/golden/make_golden.py).
 * The reference repo itself holds no SW fixture.
 *
 * Two restatements of the same recurrence:
 *   oracle_sw_score_antidiag  -- same cell order and storage discipline as the
 *       reference: anti-diagonal sweep, three live anti-diagonals per matrix,
 *       slot = (ix+iy) % 3, position = ix while the diagonal still touches
 *       row..ny-1 from the left edge, ny-1-iy afterwards
 *       (antidiagonalSmithWaterman.c:96-184 accessor rule, :254-347 sweep).
 *       This is the "port" timed as cpu_baseline when oracle/_ref is absent.
 *   oracle_sw_score_rowmajor  -- textbook Gotoh local alignment, two rolling
 *       rows; independent check of the first one.
 *
 * Scoring is fixed exactly as the reference's macros
 * (antidiagonalSmithWaterman.c:40-43): match +1, mismatch -1, first gap cell
 * open+extend = -4, every further gap cell -1.  Sequences are raw byte strings:
 * the caller decides whether the trailing '\n' is part of them (the reference
 * CLI keeps it); oracle_sw_file() below reproduces the CLI rule.
 */
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SW_MATCH 1
#define SW_MISMATCH (-1)
#define SW_GAP_FIRST (-4) /* SCORE_OPEN_GAP + SCORE_EXTEND_GAP, ref :313,:321 */
#define SW_GAP_NEXT (-1)  /* SCORE_EXTEND_GAP */
#define SW_NEG_INF INT_MIN

static inline int imax(int a, int b) { return a > b ? a : b; }

/* guarded add, reference sum_with_infinity (:86-88) */
static inline int add_inf(int a, int b) { return a == SW_NEG_INF ? SW_NEG_INF : a + b; }

/* position of cell (iy,ix) inside its anti-diagonal's nx-wide slot, ref :128-133 */
static inline int slot_pos(int iy, int ix, int nx, int ny)
{
    int d = ix + iy;
    return (d % 3) * nx + (d > ny - 1 ? ny - 1 - iy : ix);
}

/*
 * sx/lx: the sequence laid along columns, sy/ly along rows.  The reference
 * puts the shorter line on x (:229-244); scoring is symmetric so callers need
 * not, but oracle_sw_file() does, to walk the cells in the reference's order.
 */
int oracle_sw_score_antidiag(const unsigned char *sx, int lx, const unsigned char *sy, int ly)
{
    if (lx > ly) { /* keep nx <= ny like the reference: the slot rule assumes it */
        const unsigned char *t = sx; sx = sy; sy = t;
        int tl = lx; lx = ly; ly = tl;
    }
    int nx = lx + 1, ny = ly + 1;
    int *buf = (int *)malloc((size_t)nx * 9 * sizeof(int)); /* ref :250 */
    if (!buf) return INT_MIN;
    int *Pm = buf, *Qm = buf + 3 * nx, *Dm = buf + 6 * nx;
    int best = 0;
    int ndiag = nx + ny - 1;
    for (int d = 0; d < ndiag; d++) {
        /* cells of diagonal d: iy from min(d,ny-1) down, ix = d - iy, ref :270-287 */
        int iy = d < ny ? d : ny - 1;
        int ix = d - iy;
        for (; iy >= 0 && ix < nx; iy--, ix++) {
            int here = slot_pos(iy, ix, nx, ny);
            if (iy == 0) { /* first row, ref :290-297 */
                Pm[here] = SW_NEG_INF; Qm[here] = 0; Dm[here] = 0;
            } else if (ix == 0) { /* first column, ref :299-306 */
                Pm[here] = 0; Qm[here] = SW_NEG_INF; Dm[here] = 0;
            } else {
                int up = slot_pos(iy - 1, ix, nx, ny);
                int left = slot_pos(iy, ix - 1, nx, ny);
                int diag = slot_pos(iy - 1, ix - 1, nx, ny);
                int p = imax(add_inf(Dm[up], SW_GAP_FIRST), add_inf(Pm[up], SW_GAP_NEXT));     /* :313 */
                int q = imax(add_inf(Dm[left], SW_GAP_FIRST), add_inf(Qm[left], SW_GAP_NEXT)); /* :321 */
                int s = Dm[diag] + (sy[iy - 1] == sx[ix - 1] ? SW_MATCH : SW_MISMATCH);         /* :332 */
                int v = imax(imax(p, q), imax(s, 0));                                           /* :333 */
                Pm[here] = p; Qm[here] = q; Dm[here] = v;
                if (v > best) best = v; /* :335 */
            }
        }
    }
    free(buf);
    return best;
}

int oracle_sw_score_rowmajor(const unsigned char *sx, int lx, const unsigned char *sy, int ly)
{
    /* H[j], E[j] hold row iy-1 while row iy is produced; F is carried along the row. */
    int *H = (int *)malloc((size_t)(lx + 1) * 2 * sizeof(int));
    if (!H) return INT_MIN;
    int *E = H + lx + 1;
    for (int j = 0; j <= lx; j++) { H[j] = 0; E[j] = SW_NEG_INF; }
    int best = 0;
    for (int iy = 1; iy <= ly; iy++) {
        int hdiag = 0, hleft = 0, f = SW_NEG_INF;
        for (int ix = 1; ix <= lx; ix++) {
            int e = imax(add_inf(H[ix], SW_GAP_FIRST), add_inf(E[ix], SW_GAP_NEXT));
            f = imax(add_inf(hleft, SW_GAP_FIRST), add_inf(f, SW_GAP_NEXT));
            int s = hdiag + (sy[iy - 1] == sx[ix - 1] ? SW_MATCH : SW_MISMATCH);
            int v = imax(imax(e, f), imax(s, 0));
            hdiag = H[ix];
            H[ix] = v; E[ix] = e; hleft = v;
            if (v > best) best = v;
        }
    }
    free(H);
    return best;
}

/*
 * Parametrised Gotoh (runtime scoring parameters).  The reference has no such
 * mode -- its kernel arguments are ignored (hipvers.cpp:214) -- so for settings other than
 * (+1, -1, -3, -1) this is "parity unpinned": it pins the product against this restatement only.
 * At the reference setting it is checked to equal oracle_sw_score_rowmajor (and thus the goldens).
 * Values are added to the score; first gap cell = gap_open + gap_extend, further cells gap_extend.
 */
int oracle_sw_score_scored(const unsigned char *sx, int lx, const unsigned char *sy, int ly, int match, int mismatch,
                           int gap_open, int gap_extend)
{
    int *H = (int *)malloc((size_t)(lx + 1) * 2 * sizeof(int));
    if (!H) return INT_MIN;
    int *E = H + lx + 1;
    const int gf = gap_open + gap_extend;
    for (int j = 0; j <= lx; j++) { H[j] = 0; E[j] = SW_NEG_INF; }
    int best = 0;
    for (int iy = 1; iy <= ly; iy++) {
        int hdiag = 0, hleft = 0, f = SW_NEG_INF;
        for (int ix = 1; ix <= lx; ix++) {
            int e = imax(add_inf(H[ix], gf), add_inf(E[ix], gap_extend));
            f = imax(add_inf(hleft, gf), add_inf(f, gap_extend));
            int s = hdiag + (sy[iy - 1] == sx[ix - 1] ? match : mismatch);
            int v = imax(imax(e, f), imax(s, 0));
            hdiag = H[ix];
            H[ix] = v; E[ix] = e; hleft = v;
            if (v > best) best = v;
        }
    }
    free(H);
    return best;
}

int oracle_sw_batch_scored(const unsigned char *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                           int32_t *scores, int match, int mismatch, int gap_open, int gap_extend)
{
    for (int64_t p = 0; p < n_pairs; p++) {
        int s = oracle_sw_score_scored(bases + off[2 * p], (int)len[2 * p], bases + off[2 * p + 1], (int)len[2 * p + 1],
                                       match, mismatch, gap_open, gap_extend);
        if (s == INT_MIN) return -1;
        scores[p] = s;
    }
    return 0;
}

/*
 * Gotoh with a substitution matrix.  The reference has no such mode: "parity
 * unpinned", this restatement is the only check of the product's matrix mode.  code[] maps a byte to
 * a symbol number (every byte of the input must map below n, the caller checks); score is
 * [32][32] row-major; gaps as in oracle_sw_score_scored.  With score[a][b] = (a == b ? match :
 * mismatch) it equals oracle_sw_score_scored (tests/test_oracle_sw.py).
 */
int oracle_sw_score_matrix(const unsigned char *sx, int lx, const unsigned char *sy, int ly, const unsigned char *code,
                           const signed char *score, int gap_open, int gap_extend)
{
    int *H = (int *)malloc((size_t)(lx + 1) * 2 * sizeof(int));
    if (!H) return INT_MIN;
    int *E = H + lx + 1;
    const int gf = gap_open + gap_extend;
    for (int j = 0; j <= lx; j++) { H[j] = 0; E[j] = SW_NEG_INF; }
    int best = 0;
    for (int iy = 1; iy <= ly; iy++) {
        int hdiag = 0, hleft = 0, f = SW_NEG_INF;
        const signed char *row = score + 32 * code[sy[iy - 1]];
        for (int ix = 1; ix <= lx; ix++) {
            int e = imax(add_inf(H[ix], gf), add_inf(E[ix], gap_extend));
            f = imax(add_inf(hleft, gf), add_inf(f, gap_extend));
            int s = hdiag + row[code[sx[ix - 1]]];
            int v = imax(imax(e, f), imax(s, 0));
            hdiag = H[ix];
            H[ix] = v; E[ix] = e; hleft = v;
            if (v > best) best = v;
        }
    }
    free(H);
    return best;
}

int oracle_sw_batch_matrix(const unsigned char *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                           int32_t *scores, const unsigned char *code, const signed char *score, int gap_open,
                           int gap_extend)
{
    for (int64_t p = 0; p < n_pairs; p++) {
        int s = oracle_sw_score_matrix(bases + off[2 * p], (int)len[2 * p], bases + off[2 * p + 1], (int)len[2 * p + 1], code,
                                       score, gap_open, gap_extend);
        if (s == INT_MIN) return -1;
        scores[p] = s;
    }
    return 0;
}

/*
 * Batch form used by the tests and by bench.py's cpu_baseline leg:
 * seq k lives at bases+off[k], len[k] bytes; pair p = (2p, 2p+1).
 * variant 0 = antidiag port, 1 = rowmajor.
 */
int oracle_sw_batch(const unsigned char *bases, const uint64_t *off, const uint32_t *len,
                    int64_t n_pairs, int32_t *scores, int variant)
{
    for (int64_t p = 0; p < n_pairs; p++) {
        const unsigned char *a = bases + off[2 * p], *b = bases + off[2 * p + 1];
        int la = (int)len[2 * p], lb = (int)len[2 * p + 1];
        int s = variant ? oracle_sw_score_rowmajor(a, la, b, lb) : oracle_sw_score_antidiag(a, la, b, lb);
        if (s == INT_MIN) return -1;
        scores[p] = s;
    }
    return 0;
}

/*
 * File front end, restating the reference CLI's reading rule
 * (antidiagonalSmithWaterman.c:201-247): line 1 = atoi(number of sequence
 * LINES); fgets into a 1000-byte buffer so longer lines split; the newline
 * fgets keeps is part of the sequence; stop at the first missing line.
 * Returns the number of scores written (<= max_scores), or -1 if the file
 * cannot be opened, -2 if it is empty.  *line_num receives the header value.
 */
#define ORACLE_SW_LINE 1000
long oracle_sw_file(const char *path, int32_t *scores, long max_scores, int *line_num, int variant)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char l1[ORACLE_SW_LINE], l2[ORACLE_SW_LINE];
    if (!fgets(l1, sizeof l1, f)) { fclose(f); return -2; }
    int n = atoi(l1);
    if (line_num) *line_num = n;
    long k = 0;
    for (int i = 0; i < n; i += 2) {
        if (!fgets(l1, sizeof l1, f)) break;
        if (!fgets(l2, sizeof l2, f)) break;
        int a = (int)strlen(l1), b = (int)strlen(l2);
        const unsigned char *sx = (const unsigned char *)l1, *sy = (const unsigned char *)l2;
        int lx = a, ly = b;
        if (a > b) { sx = (const unsigned char *)l2; sy = (const unsigned char *)l1; lx = b; ly = a; }
        int s = variant ? oracle_sw_score_rowmajor(sx, lx, sy, ly) : oracle_sw_score_antidiag(sx, lx, sy, ly);
        if (k < max_scores) scores[k] = s;
        k++;
    }
    fclose(f);
    return k;
}
