/*
 * Readers for the reference's two text formats (include/agx.h, "text front end").
 * Plain C99; no device code.  They restate the reading rules of
 *   smithWaterman/antidiagonalSmithWaterman.c:201-247  and
 *   pairHMM/antidiagsPairHMM.c:353-418,484-489
 * so the drop-in command lines see exactly the pairs the reference programs see.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "agx.h"

/* implemented in agx_runtime.cpp */
extern void agx_set_error(const char *fmt, ...);
extern int agx_host_threads_c(void);                                               /* the process-wide host thread pool */
extern void agx_pool_run_c(int parts, void (*task)(int part, void *arg), void *arg);

/* ------------------------------------------------------------ growable byte/array helpers */

typedef struct {
    unsigned char *p;
    size_t n, cap;
} buf_t;

static int buf_reserve(buf_t *b, size_t extra)
{
    if (b->n + extra <= b->cap) return 0;
    size_t cap = b->cap ? b->cap : 4096;
    while (cap < b->n + extra) cap *= 2;
    unsigned char *q = (unsigned char *)realloc(b->p, cap);
    if (!q) return -1;
    b->p = q;
    b->cap = cap;
    return 0;
}

static int buf_put(buf_t *b, const void *src, size_t n)
{
    if (buf_reserve(b, n ? n : 1)) return -1;
    if (n) memcpy(b->p + b->n, src, n);
    b->n += n;
    return 0;
}

/* -------------------------------------------------------------------------- Smith-Waterman */

void agx_sw_text_free(agx_sw_text *t)
{
    if (!t) return;
    free(t->bases);
    free(t->off);
    free(t->len);
    free(t->dangling);
    free(t);
}

/*
 * Block reader with fgets() semantics: a "line" is the bytes up to and including the next '\n', at
 * most line_buf-1 of them (longer lines split there), or what is left at end of file.  The block
 * is scanned with memchr instead of byte by byte; the strlen() the reference applies to the fgets
 * buffer (a NUL byte in the file hides the rest of the line) is applied to the line found.
 */
struct agx_sw_reader {
    FILE *f;
    int line_buf;
    int32_t line_num;
    int64_t lines_taken; /* the reference's loop variable i (:216): sequence lines consumed so far */
    int finished;
    char *buf;
    size_t cap, lo, hi;
    int eof;
    size_t hint_bases, hint_pairs; /* size of the previous chunk: the next one reserves that much up front */
    int regular;                   /* a regular file: chunks are read with pread() by several threads */
    int threads;                   /* ... at most this many (0 = the pool's size) */
    off_t pos, size;               /* next file byte not yet read into a chunk or the buffer; the file's size */
};

/* next line -> *p (valid until the next call), its strlen() in *n; 0 = no line left */
static int reader_line(agx_sw_reader *r, const char **p, size_t *n)
{
    const size_t max = (size_t)r->line_buf - 1;
    for (;;) {
        const size_t have = r->hi - r->lo;
        const size_t look = have < max ? have : max;
        const char *nl = look ? (const char *)memchr(r->buf + r->lo, '\n', look) : NULL;
        size_t len = 0;
        if (nl) len = (size_t)(nl - (r->buf + r->lo)) + 1;
        else if (look == max) len = max;       /* split an over-long line where fgets would */
        else if (r->eof) len = have;           /* last line without a newline */
        else {                                 /* need more bytes */
            if (r->lo) {
                memmove(r->buf, r->buf + r->lo, have);
                r->lo = 0;
                r->hi = have;
            }
            const size_t got = fread(r->buf + r->hi, 1, r->cap - r->hi, r->f);
            r->hi += got;
            if (got == 0) r->eof = 1;
            continue;
        }
        if (len == 0) return 0;
        *p = r->buf + r->lo;
        const char *z = (const char *)memchr(*p, 0, len);
        *n = z ? (size_t)(z - *p) : len;
        r->lo += len;
        return 1;
    }
}

void agx_sw_reader_close(agx_sw_reader *r)
{
    if (!r) return;
    if (r->f) fclose(r->f);
    free(r->buf);
    free(r);
}

int agx_sw_reader_open(const char *path, int line_buf, agx_sw_reader **out)
{
    if (!out || !path) {
        agx_set_error("agx_sw_reader_open: null argument");
        return AGX_E_ARG;
    }
    *out = NULL;
    if (line_buf <= 0) line_buf = 1000; /* MAX_LINE_LENGTH, antidiagonalSmithWaterman.c:44 */
    if (line_buf < 2) {
        agx_set_error("agx_sw_reader_open: line buffer too small");
        return AGX_E_ARG;
    }
    agx_sw_reader *r = (agx_sw_reader *)calloc(1, sizeof *r);
    if (!r) {
        agx_set_error("agx_sw_reader_open: out of memory");
        return AGX_E_NOMEM;
    }
    r->line_buf = line_buf;
    r->cap = (size_t)4 << 20;
    if (r->cap < 2 * (size_t)line_buf) r->cap = 2 * (size_t)line_buf;
    r->buf = (char *)malloc(r->cap);
    r->f = fopen(path, "r");
    if (!r->f) {
        agx_set_error("Error opening file: %s", strerror(errno));
        agx_sw_reader_close(r);
        return AGX_E_IO;
    }
    if (!r->buf) {
        agx_set_error("agx_sw_reader_open: out of memory");
        agx_sw_reader_close(r);
        return AGX_E_NOMEM;
    }
    setvbuf(r->f, NULL, _IONBF, 0); /* the reader has its own block buffer */
    const char *p;
    size_t n;
    if (!reader_line(r, &p, &n)) { /* :205-208 */
        agx_set_error("file is empty");
        agx_sw_reader_close(r);
        return AGX_E_IO;
    }
    char head[32];
    const size_t hn = n < sizeof head - 1 ? n : sizeof head - 1;
    memcpy(head, p, hn);
    head[hn] = 0;
    r->line_num = atoi(head); /* number of sequence LINES, :209 */
    {
        struct stat st;
        const long at = ftell(r->f);
        if (at >= 0 && fstat(fileno(r->f), &st) == 0 && S_ISREG(st.st_mode)) {
            r->regular = 1;
            r->pos = (off_t)at; /* the block buffer holds the bytes before it */
            r->size = st.st_size;
        }
    }
    *out = r;
    return AGX_OK;
}

int32_t agx_sw_reader_line_num(const agx_sw_reader *r) { return r ? r->line_num : -1; }

void agx_sw_reader_set_threads(agx_sw_reader *r, int n_threads)
{
    if (r) r->threads = n_threads > 0 ? n_threads : 0;
}

/* ------------------------------------------------------------------ parallel chunk scan (regular files)
 * A chunk's bytes are read by T threads (pread of one slice each) and scanned by T threads: a thread owns the physical
 * lines -- newline to newline -- that START in its slice, follows the last of them into the next slices, and cuts every
 * physical line into fgets() lines from its start (at most line_buf - 1 bytes each).  After a newline the reader's state
 * is independent of what came before, which is what makes the slices independent.
 */
typedef struct {
    uint64_t off;
    uint32_t flen, slen; /* bytes fgets() consumes / strlen() of what it returns */
} sw_line;

typedef struct {
    unsigned char *base; /* the chunk's bases array */
    size_t carry, n;     /* bytes present before the read; bytes present after it */
    size_t slice;        /* bytes per part (scan: of [0, n); read: of [carry, n)) */
    int parts, fd, eof, has_nul, oom, io_error;
    off_t pos;
    size_t max;          /* line_buf - 1 */
    sw_line **lines;
    size_t *n_lines;
    size_t tail;         /* first byte of an incomplete last line, or n */
    uint64_t *off;       /* fill pass: destination arrays, lines to write, first global line of every part */
    uint32_t *len;
    size_t used_lines, *first_line;
} sw_scan;

static void sw_read_part(int part, void *arg)
{
    sw_scan *j = (sw_scan *)arg;
    const size_t a = j->carry + (size_t)part * j->slice;
    size_t b = a + j->slice;
    if (b > j->n || part == j->parts - 1) b = j->n;
    size_t at = a;
    while (at < b) { /* pread may return short counts */
        const ssize_t got = pread(j->fd, j->base + at, b - at, j->pos + (off_t)(at - j->carry));
        if (got <= 0) {
            j->io_error = 1;
            return;
        }
        at += (size_t)got;
    }
    if (b > a && memchr(j->base + a, 0, b - a)) j->has_nul = 1;
}

static void sw_scan_part(int part, void *arg)
{
    sw_scan *j = (sw_scan *)arg;
    const unsigned char *base = j->base;
    const size_t n = j->n, max = j->max;
    const size_t a = (size_t)part * j->slice;
    size_t b = a + j->slice;
    if (b > n || part == j->parts - 1) b = n;
    size_t pos = a;
    if (part > 0 && base[a - 1] != '\n') { /* the slice begins inside a physical line owned by an earlier part */
        const unsigned char *q = a < b ? (const unsigned char *)memchr(base + a, '\n', b - a) : NULL;
        if (!q) return;
        pos = (size_t)(q - base) + 1;
    }
    size_t cap = (b - a) / 96 + 64, cnt = 0;
    sw_line *out = (sw_line *)malloc(cap * sizeof *out);
    if (!out) {
        j->oom = 1;
        return;
    }
    while (pos < b) { /* one physical line per trip */
        for (;;) {
            const size_t have = n - pos, look = have < max ? have : max;
            const unsigned char *q = base + pos;
            const unsigned char *nl = look ? (const unsigned char *)memchr(q, '\n', look) : NULL;
            size_t flen;
            int ends;
            if (nl) flen = (size_t)(nl - q) + 1, ends = 1;
            else if (look == max) flen = max, ends = 0; /* an over-long line splits where fgets would */
            else if (j->eof && have) flen = have, ends = 1; /* last line without a newline */
            else {
                j->tail = pos; /* incomplete (or nothing left): only the part that reaches n gets here */
                goto done;
            }
            size_t slen = flen;
            if (j->has_nul) {
                const unsigned char *z = (const unsigned char *)memchr(q, 0, flen);
                if (z) slen = (size_t)(z - q);
            }
            if (cnt == cap) {
                sw_line *g = (sw_line *)realloc(out, 2 * cap * sizeof *out);
                if (!g) {
                    free(out);
                    j->oom = 1;
                    return;
                }
                out = g;
                cap *= 2;
            }
            out[cnt].off = pos;
            out[cnt].flen = (uint32_t)flen;
            out[cnt].slen = (uint32_t)slen;
            ++cnt;
            pos += flen;
            if (ends) break;
        }
    }
done:
    j->lines[part] = out;
    j->n_lines[part] = cnt;
}

static void sw_fill_part(int part, void *arg)
{
    sw_scan *j = (sw_scan *)arg;
    const sw_line *l = j->lines[part];
    size_t g = j->first_line[part];
    for (size_t k = 0; k < j->n_lines[part] && g < j->used_lines; ++k, ++g) {
        j->off[g] = l[k].off;
        j->len[g] = l[k].slen;
    }
}

/*
 * One chunk of pairs.  The lines of a chunk are consecutive bytes of the file, newline included (:229-247 keep it), so the
 * chunk's `bases` array IS a piece of the file: it is read straight into that array and only scanned -- one memchr per
 * line for the newline, one per block for a NUL byte (the strlen() rule; per line only when the block has one) -- where
 * the first version copied every line out of a block buffer (fread, two memchr, memcpy: 2.3 GB/s; now about twice that).
 * Bytes read beyond the chunk's last line go back into the reader's buffer for the next chunk.
 */
int agx_sw_reader_next(agx_sw_reader *r, int64_t max_pairs, agx_sw_text **out)
{
    if (!r || !out) {
        agx_set_error("agx_sw_reader_next: null argument");
        return AGX_E_ARG;
    }
    *out = NULL;
    agx_sw_text *t = (agx_sw_text *)calloc(1, sizeof *t);
    buf_t bases = {0}, off = {0}, len = {0};
    int rc = AGX_E_NOMEM;
    if (!t) goto done;
    t->line_num = r->line_num;
    {
        /* how many bytes this chunk will probably take: the previous chunk's bytes per pair, else the file's size */
        const int64_t lines_left = (int64_t)r->line_num - r->lines_taken;
        int64_t want = lines_left > 0 ? (lines_left + 1) / 2 : 0;
        if (want > max_pairs) want = max_pairs;
        size_t est = (size_t)64 << 20;
        if (r->hint_pairs)
            est = (size_t)((double)r->hint_bases / (double)r->hint_pairs * (double)want * 1.03) + 65536;
        else if (r->regular && (size_t)(r->size - r->pos) < est)
            est = (size_t)(r->size - r->pos);
        /* an unbounded request (agx_sw_text_read) takes the rest of the file in one piece: the threaded branch below
           reads `est` bytes and no more, so a smaller estimate would end the chunk early */
        if (max_pairs == INT64_MAX && r->regular && r->size > r->pos) est = (size_t)(r->size - r->pos);
        const size_t carry = r->hi - r->lo;
        if (buf_reserve(&bases, carry + est + 4096)) goto done;
        if (carry) memcpy(bases.p, r->buf + r->lo, carry);
        bases.n = carry; /* bytes present in the chunk's array */
        r->lo = r->hi = 0;
        if (want > 0 && (buf_reserve(&off, 2 * (size_t)want * sizeof(uint64_t)) || buf_reserve(&len, 2 * (size_t)want * sizeof(uint32_t)))) goto done;

        /* ---- large chunks of a regular file: read and scanned by the host thread pool */
        size_t to_read = r->regular && r->size > r->pos ? (size_t)(r->size - r->pos) : 0;
        if (to_read > est) to_read = est;
        int parts = agx_host_threads_c();
        if (r->threads && parts > r->threads) parts = r->threads;
        if ((size_t)parts > to_read / ((size_t)4 << 20)) parts = (int)(to_read / ((size_t)4 << 20));
        if (want > 0 && parts > 1) {
            sw_scan j;
            memset(&j, 0, sizeof j);
            j.base = bases.p;
            j.carry = carry;
            j.n = carry + to_read;
            j.parts = parts;
            j.fd = fileno(r->f);
            j.pos = r->pos;
            j.max = (size_t)r->line_buf - 1;
            j.slice = (to_read + (size_t)parts - 1) / (size_t)parts;
            j.has_nul = carry && memchr(bases.p, 0, carry) != NULL;
            sw_line **lines = (sw_line **)calloc((size_t)parts, sizeof *lines);
            size_t *counts = (size_t *)calloc(2 * (size_t)parts, sizeof *counts);
            if (!lines || !counts) {
                free(lines);
                free(counts);
                goto done;
            }
            j.lines = lines;
            j.n_lines = counts;
            j.first_line = counts + parts;
            agx_pool_run_c(parts, sw_read_part, &j);
            int failed = j.io_error;
            if (!failed) {
                r->pos += (off_t)to_read;
                bases.n = j.n;
                j.eof = r->eof = r->pos >= r->size;
                j.tail = j.n;
                j.slice = (j.n + (size_t)parts - 1) / (size_t)parts;
                agx_pool_run_c(parts, sw_scan_part, &j);
                failed = j.oom;
            }
            size_t avail = 0;
            for (int k = 0; k < parts; ++k) {
                j.first_line[k] = avail;
                avail += j.n_lines[k];
            }
            /* the pairs the reference's loop (:216-227) takes from these lines */
            size_t pairs = avail / 2;
            if ((int64_t)pairs > want) pairs = (size_t)want;
            if (!failed && (buf_reserve(&off, 2 * pairs * sizeof(uint64_t) + 8) || buf_reserve(&len, 2 * pairs * sizeof(uint32_t) + 4))) failed = 1;
            size_t used_end = j.tail; /* first byte that belongs to the next chunk */
            if (!failed) {
                j.off = (uint64_t *)off.p;
                j.len = (uint32_t *)len.p;
                j.used_lines = 2 * pairs;
                agx_pool_run_c(parts, sw_fill_part, &j);
                off.n = 2 * pairs * sizeof(uint64_t);
                len.n = 2 * pairs * sizeof(uint32_t);
                /* the line after the last one used, if this chunk holds it */
                const sw_line *next = NULL;
                for (int k = 0; k < parts && !next; ++k)
                    if (j.used_lines >= j.first_line[k] && j.used_lines < j.first_line[k] + j.n_lines[k]) next = &j.lines[k][j.used_lines - j.first_line[k]];
                if (next) used_end = (size_t)next->off;
                t->n_pairs = (int64_t)pairs;
                r->lines_taken += 2 * (int64_t)pairs;
                if (r->lines_taken >= r->line_num) r->finished = 1; /* loop condition of :216 */
                else if ((int64_t)pairs < want && r->eof) {          /* the file ends before the count does */
                    if (next) {                                      /* :223-227: a first line without a second is echoed */
                        t->dangling = (char *)malloc((size_t)next->slen + 1);
                        if (!t->dangling) failed = 1;
                        else {
                            memcpy(t->dangling, bases.p + next->off, next->slen);
                            t->dangling[next->slen] = 0;
                        }
                    }
                    r->finished = 1;
                }
            }
            for (int k = 0; k < parts; ++k) free(lines[k]);
            free(lines);
            free(counts);
            if (j.io_error) {
                agx_set_error("agx_sw_reader_next: read error: %s", strerror(errno));
                rc = AGX_E_IO;
                goto done_keep_error;
            }
            if (failed) goto done;
            const size_t left = r->finished ? 0 : bases.n - used_end;
            if (left) {
                if (left > r->cap) {
                    char *nb = (char *)realloc(r->buf, left);
                    if (!nb) goto done;
                    r->buf = nb;
                    r->cap = left;
                }
                memcpy(r->buf, bases.p + used_end, left);
            }
            r->lo = 0;
            r->hi = left;
            r->hint_bases = used_end;
            goto publish;
        }
    }
    {
        const size_t max = (size_t)r->line_buf - 1;
        size_t lo = 0;        /* first byte not yet given to a pair */
        size_t nul_seen = 0;  /* [0, nul_seen) has been searched for NUL bytes */
        int has_nul = 0;
        /* one line at `at`: its fgets length in *flen, its strlen in *slen; 1 = found, 0 = no line left, -1 = needs more bytes */
#define AGX_LINE_AT(at, flen, slen)                                                                                   \
    do {                                                                                                              \
        const size_t have_ = bases.n - (at);                                                                          \
        const size_t look_ = have_ < max ? have_ : max;                                                               \
        const unsigned char *q_ = bases.p + (at);                                                                     \
        const unsigned char *nl_ = look_ ? (const unsigned char *)memchr(q_, '\n', look_) : NULL;                     \
        if (nl_) (flen) = (size_t)(nl_ - q_) + 1, got_ = 1;                                                           \
        else if (look_ == max) (flen) = max, got_ = 1;      /* an over-long line splits where fgets would */          \
        else if (r->eof) (flen) = have_, got_ = have_ ? 1 : 0; /* last line without a newline */                      \
        else got_ = -1;                                                                                               \
        if (got_ == 1) {                                                                                              \
            (slen) = (flen);                                                                                          \
            if (has_nul) {                                                                                            \
                const unsigned char *z_ = (const unsigned char *)memchr(q_, 0, (flen));                               \
                if (z_) (slen) = (size_t)(z_ - q_);                                                                   \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
        while (!r->finished && t->n_pairs < max_pairs) {
            if (r->lines_taken >= r->line_num) { /* loop condition of :216 */
                r->finished = 1;
                break;
            }
            if (nul_seen < bases.n) {
                if (!has_nul && memchr(bases.p + nul_seen, 0, bases.n - nul_seen)) has_nul = 1;
                nul_seen = bases.n;
            }
            int got_;
            size_t f1 = 0, s1 = 0, f2 = 0, s2 = 0;
            AGX_LINE_AT(lo, f1, s1);
            int need_more = got_ < 0;
            if (got_ == 0) { /* :219-221 */
                r->finished = 1;
                break;
            }
            if (!need_more) {
                AGX_LINE_AT(lo + f1, f2, s2);
                need_more = got_ < 0;
                if (got_ == 0) { /* :223-227: the first line is echoed, loop ends */
                    t->dangling = (char *)malloc(s1 + 1);
                    if (!t->dangling) goto done;
                    memcpy(t->dangling, bases.p + lo, s1);
                    t->dangling[s1] = 0;
                    r->finished = 1;
                    break;
                }
            }
            if (need_more) { /* the pair at `lo` is not complete: more of the file, then look at it again */
                if (bases.n + ((size_t)1 << 20) > bases.cap && buf_reserve(&bases, bases.cap / 2 + ((size_t)16 << 20))) goto done;
                /* in slices that stay in the cache while they are scanned (one read() of the whole chunk streamed it
                   through memory three times: 1.3 instead of 0.45 s for 573 MB) */
                size_t slice = bases.cap - bases.n;
                if (slice > ((size_t)4 << 20)) slice = (size_t)4 << 20;
                size_t got;
                if (r->regular) {
                    const ssize_t g = pread(fileno(r->f), bases.p + bases.n, slice, r->pos);
                    got = g > 0 ? (size_t)g : 0;
                    r->pos += (off_t)got;
                } else
                    got = fread(bases.p + bases.n, 1, slice, r->f);
                bases.n += got;
                if (got == 0) r->eof = 1;
                continue;
            }
            const uint64_t o1 = lo, o2 = lo + f1;
            const uint32_t l1 = (uint32_t)s1, l2 = (uint32_t)s2; /* newline included, :229-247 */
            if (buf_put(&off, &o1, sizeof o1) || buf_put(&len, &l1, sizeof l1) || buf_put(&off, &o2, sizeof o2) || buf_put(&len, &l2, sizeof l2))
                goto done;
            lo += f1 + f2;
            t->n_pairs++;
            r->lines_taken += 2;
        }
#undef AGX_LINE_AT
        /* what was read beyond this chunk belongs to the next one */
        const size_t left = r->finished ? 0 : bases.n - lo;
        if (left) {
            if (left > r->cap) {
                char *nb = (char *)realloc(r->buf, left);
                if (!nb) goto done;
                r->buf = nb;
                r->cap = left;
            }
            memcpy(r->buf, bases.p + lo, left);
        }
        r->lo = 0;
        r->hi = left;
        r->hint_bases = lo;
    }
publish:
    t->bases = bases.p;
    t->off = (uint64_t *)off.p;
    t->len = (uint32_t *)len.p;
    r->hint_pairs = (size_t)t->n_pairs;
    bases.p = off.p = len.p = NULL;
    rc = AGX_OK;
done:
    free(bases.p);
    free(off.p);
    free(len.p);
    if (rc != AGX_OK) {
        agx_set_error("agx_sw_reader_next: out of memory");
        agx_sw_text_free(t);
        t = NULL;
    }
    *out = t;
    return rc;
done_keep_error:
    free(bases.p);
    free(off.p);
    free(len.p);
    agx_sw_text_free(t);
    return rc;
}

int agx_sw_reader_done(const agx_sw_reader *r) { return !r || r->finished; }

int agx_sw_text_read(const char *path, int line_buf, agx_sw_text **out)
{
    if (!out || !path) {
        agx_set_error("agx_sw_text_read: null argument");
        return AGX_E_ARG;
    }
    *out = NULL;
    agx_sw_reader *r = NULL;
    int rc = agx_sw_reader_open(path, line_buf, &r);
    if (rc != AGX_OK) return rc;
    rc = agx_sw_reader_next(r, INT64_MAX, out);
    /* An unbounded request takes the whole file (the chunk's size estimate above is the rest of the file); should a
       chunk ever end before the reference's loop would have, that is an error, never a silently shorter batch. */
    if (rc == AGX_OK && !agx_sw_reader_done(r)) {
        agx_sw_text_free(*out);
        *out = NULL;
        agx_set_error("agx_sw_text_read: the reader stopped before the end of the input");
        rc = AGX_E_IO;
    }
    agx_sw_reader_close(r);
    return rc;
}

/* --------------------------------------------------------------------------------- PairHMM */

typedef struct {
    agx_phmm_text pub; /* must be first */
    buf_t rb, qb, qi, qd, qg, roff, hb, hoff, rreg, hreg;
} phmm_text_impl;

void agx_phmm_text_free(agx_phmm_text *t)
{
    if (!t) return;
    phmm_text_impl *m = (phmm_text_impl *)t;
    free(m->rb.p);
    free(m->qb.p);
    free(m->qi.p);
    free(m->qd.p);
    free(m->qg.p);
    free(m->roff.p);
    free(m->hb.p);
    free(m->hoff.p);
    free(m->rreg.p);
    free(m->hreg.p);
    free(m);
}

#define PHMM_LINE (1000 * 5 + 1) /* MAX_READ_LEN*5+1, antidiagsPairHMM.c:8,353 */

/* next whitespace-delimited token of [*s, end) (sscanf "%s" rule); returns its length, *tok its start */
static size_t next_token(const char **s, const char *end, const char **tok)
{
    const char *p = *s;
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) p++;
    *tok = p;
    while (p < end && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) p++;
    *s = p;
    return (size_t)(p - *tok);
}

/*
 * The reference's batch loop (antidiagsPairHMM.c:371-433,484-489) as a reader that hands out whole
 * regions: every next() returns the regions up to the first one that brings the chunk to max_pairs
 * pairs (at least one region), as a fresh agx_phmm_text the caller frees.  State that the reference
 * keeps across loop turns stays in the reader: nr / nh survive a malformed header line (sscanf leaves
 * them untouched, :378).
 *
 * Lines come from a block buffer with the reference's fgets(line, 5001, f) rule (at most 5000 bytes a call, a line
 * ends behind its newline; what a line holds is what strcspn(line, "\n") / strlen leave of it: up to the first
 * newline or NUL).  One pass walks the region structure -- headers, line counts, the region cut short at the end of
 * the file -- and only notes where every read line lies (its bytes go to a store of the chunk); the read lines are
 * then cut into their five fields and copied into the tracks by the host thread pool, every line at the offset the
 * prefix sum of the read lengths gives it (round 2c: the one-threaded reader, two copies and four scans per byte,
 * fed 8 M pairs/s of config 5's shape where the device takes 21 M in double).
 */
#define PHMM_BLOCK ((size_t)4 << 20)
struct agx_phmm_reader {
    FILE *f;
    /* the text of the chunk being read: [pos, end) of buf is unread; next() takes the file in blocks of PHMM_BLOCK
     * bytes behind `end`, notes the read lines where they lie (no copy) and moves what it did not consume to the front
     * for the next chunk */
    char *buf;
    size_t pos, end, cap;
    int eof;
    int nr, nh;
    int finished;
    uint32_t regions_done; /* complete regions handed out so far (for error messages) */
};

void agx_phmm_reader_close(agx_phmm_reader *r)
{
    if (!r) return;
    if (r->f) fclose(r->f);
    free(r->buf);
    free(r);
}

int agx_phmm_reader_open(const char *path, agx_phmm_reader **out)
{
    if (!out || !path) {
        agx_set_error("agx_phmm_reader_open: null argument");
        return AGX_E_ARG;
    }
    *out = NULL;
    agx_phmm_reader *r = (agx_phmm_reader *)calloc(1, sizeof *r);
    if (!r) {
        agx_set_error("agx_phmm_reader_open: out of memory");
        return AGX_E_NOMEM;
    }
    r->cap = 2 * PHMM_BLOCK;
    r->buf = (char *)malloc(r->cap);
    r->f = fopen(path, "r");
    if (!r->f) {
        agx_set_error("Error opening input file_r: %s", strerror(errno));
        agx_phmm_reader_close(r);
        return AGX_E_IO;
    }
    if (!r->buf) {
        agx_set_error("agx_phmm_reader_open: out of memory");
        agx_phmm_reader_close(r);
        return AGX_E_NOMEM;
    }
    setvbuf(r->f, NULL, _IONBF, 0); /* the reader has its own block buffer */
    *out = r;
    return AGX_OK;
}

int agx_phmm_reader_done(const agx_phmm_reader *r) { return !r || r->finished; }

/* fgets(line, PHMM_LINE, f): the next at most PHMM_LINE - 1 bytes up to and including a newline; 0 at the end of the
 * file, (size_t)-1 out of memory.  The line starts at buf + *off (the buffer may move when it grows: offsets, not
 * pointers, are what a chunk keeps). */
static size_t phmm_next_line(agx_phmm_reader *r, size_t *off)
{
    while (r->end - r->pos < (size_t)PHMM_LINE && !r->eof) {
        if (r->cap - r->end < PHMM_BLOCK) {
            size_t cap = r->cap;
            while (cap - r->end < PHMM_BLOCK) cap *= 2;
            char *q = (char *)realloc(r->buf, cap);
            if (!q) return (size_t)-1;
            r->buf = q;
            r->cap = cap;
        }
        const size_t got = fread(r->buf + r->end, 1, PHMM_BLOCK, r->f);
        if (got == 0) r->eof = 1;
        r->end += got;
    }
    size_t avail = r->end - r->pos;
    if (avail == 0) return 0;
    if (avail > (size_t)PHMM_LINE - 1) avail = (size_t)PHMM_LINE - 1;
    const char *s = r->buf + r->pos;
    const char *nl = (const char *)memchr(s, '\n', avail);
    const size_t n = nl ? (size_t)(nl - s) + 1 : avail;
    *off = r->pos;
    r->pos += n;
    return n;
}

/* what strcspn(line, "\n") leaves of a line fgets returned: the bytes before the first newline or NUL */
static size_t phmm_line_text(const char *s, size_t n)
{
    if (n && s[n - 1] == '\n') n--;
    const char *z = (const char *)memchr(s, 0, n);
    return z ? (size_t)(z - s) : n;
}

typedef struct {
    const char *raw;
    const uint64_t *line_off; /* read line i = raw[line_off[i] .. line_off[i] + line_len[i]) */
    const uint32_t *line_len;
    const uint64_t *roff; /* n_lines + 1: where its bases go in the tracks */
    unsigned char *trk[5];
    uint32_t n_lines;
    int parts;
    int64_t first_bad; /* smallest line index with an error; -1 = none */
    int bad_kind[1];   /* 1: line too short; 2 + k: field k too short */
    pthread_mutex_t mu;
} phmm_cut_job;

static void phmm_cut_part(int part, void *arg)
{
    phmm_cut_job *j = (phmm_cut_job *)arg;
    const uint32_t per = (j->n_lines + (uint32_t)j->parts - 1) / (uint32_t)j->parts;
    const uint32_t lo = (uint32_t)part * per, hi = lo + per < j->n_lines ? lo + per : j->n_lines;
    for (uint32_t i = lo; i < hi; i++) {
        const char *s = j->raw + j->line_off[i], *end = s + j->line_len[i];
        const size_t sl = j->line_len[i];
        int kind = 0;
        if (sl < 4) /* (strlen-4)/5 underflows in the reference (:418) */
            kind = 1;
        else {
            const size_t n = (sl - 4) / 5;
            const char *p = s, *tok[5];
            size_t tl[5];
            for (int k = 0; k < 5; k++) tl[k] = next_token(&p, end, &tok[k]);
            for (int k = 0; k < 5 && !kind; k++)
                if (tl[k] < n) kind = 2 + k;
            if (!kind)
                for (int k = 0; k < 5; k++) memcpy(j->trk[k] + j->roff[i], tok[k], n);
        }
        if (kind) {
            pthread_mutex_lock(&j->mu);
            if (j->first_bad < 0 || (int64_t)i < j->first_bad) {
                j->first_bad = (int64_t)i;
                j->bad_kind[0] = kind;
            }
            pthread_mutex_unlock(&j->mu);
            return; /* lines behind an error of this part cannot come first */
        }
    }
}

int agx_phmm_reader_next(agx_phmm_reader *r, int64_t max_pairs, agx_phmm_text **out)
{
    if (!r || !out) {
        agx_set_error("agx_phmm_reader_next: null argument");
        return AGX_E_ARG;
    }
    *out = NULL;
    int rc = AGX_E_NOMEM;
    phmm_text_impl *m = (phmm_text_impl *)calloc(1, sizeof *m);
    buf_t line_off = {0}, line_len = {0}; /* where every read line of the chunk lies in the reader's buffer */
    int nr = r->nr, nh = r->nh;
    uint64_t z64 = 0;
    uint32_t z32 = 0;
    uint32_t n_reads = 0, n_haps = 0, n_regions = 0;
    size_t line = 0, ln = 0;
    if (!m) goto done;
    if (r->pos) { /* what the last chunk left unread moves to the front */
        memmove(r->buf, r->buf + r->pos, r->end - r->pos);
        r->end -= r->pos;
        r->pos = 0;
    }
    if (buf_put(&m->hoff, &z64, 8) || buf_put(&m->rreg, &z32, 4) || buf_put(&m->hreg, &z32, 4)) goto done;
    while (!r->finished && (n_regions == 0 || m->pub.n_pairs < max_pairs)) {
        if ((ln = phmm_next_line(r, &line)) == (size_t)-1) goto done;
        if (!ln) { /* :375 */
            r->finished = 1;
            break;
        }
        m->pub.n_regions_seen++;
        {
            char head[PHMM_LINE]; /* the line as the C string fgets would have left */
            const size_t hl = phmm_line_text(r->buf + line, ln);
            memcpy(head, r->buf + line, hl);
            head[hl] = 0;
            sscanf(head, "%d %d", &nr, &nh);
        }
        if (nh < 0) {
            /* :381-385: malloc(num_haplotypes * sizeof(char *)) of a negative count fails, the reference prints "Memory
             * allocation failed for haplotypes array" and exits with failure after the `#batch:` line of this turn */
            m->pub.truncated = 2;
            r->finished = 1;
            break;
        }
        if (nr < 0) nr = 0; /* a negative read count runs none of the reference's loops over reads */
        /* the reference reads the haplotypes first through a second stream (:389-407) and fails
         * with "Error reading haplotypes." when the region is cut short: nothing of it is output */
        int got_r = 0, got_h = 0;
        const size_t lo_mark = line_off.n, ll_mark = line_len.n;
        for (; got_r < nr; got_r++) {
            if ((ln = phmm_next_line(r, &line)) == (size_t)-1) goto done;
            if (!ln) break;
            const uint64_t o = line;
            const uint32_t l = (uint32_t)phmm_line_text(r->buf + line, ln); /* :417 */
            if (buf_put(&line_off, &o, 8) || buf_put(&line_len, &l, 4)) goto done;
        }
        const size_t hb_mark = m->hb.n, hoff_mark = m->hoff.n;
        if (got_r == nr) {
            for (; got_h < nh; got_h++) {
                if ((ln = phmm_next_line(r, &line)) == (size_t)-1) goto done;
                if (!ln) break;
                if (buf_put(&m->hb, r->buf + line, phmm_line_text(r->buf + line, ln))) goto done; /* :399 */
                uint64_t o = m->hb.n;
                if (buf_put(&m->hoff, &o, 8)) goto done;
            }
        }
        if (got_r < nr || got_h < nh) {
            m->hb.n = hb_mark;
            m->hoff.n = hoff_mark;
            line_off.n = lo_mark;
            line_len.n = ll_mark;
            m->pub.truncated = 1;
            r->finished = 1;
            break;
        }
        n_reads += (uint32_t)nr;
        n_haps += (uint32_t)nh;
        n_regions++;
        if (buf_put(&m->rreg, &n_reads, 4) || buf_put(&m->hreg, &n_haps, 4)) goto done;
        m->pub.n_pairs += (int64_t)nr * nh;
    }
    /* ---- the read lines -> five tracks: lengths and offsets first, then the cutting and copying, threaded */
    {
        const uint64_t *lo = (const uint64_t *)line_off.p;
        const uint32_t *ll = (const uint32_t *)line_len.p;
        if (buf_reserve(&m->roff, ((size_t)n_reads + 1) * 8)) goto done;
        uint64_t *roff = (uint64_t *)m->roff.p;
        roff[0] = 0;
        for (uint32_t i = 0; i < n_reads; i++) {
            const uint64_t sl = ll[i];
            roff[i + 1] = roff[i] + (sl < 4 ? 0 : (sl - 4) / 5);
        }
        m->roff.n = ((size_t)n_reads + 1) * 8;
        const size_t total = (size_t)roff[n_reads];
        buf_t *trk[5] = {&m->rb, &m->qb, &m->qi, &m->qd, &m->qg};
        for (int k = 0; k < 5; k++) {
            if (buf_reserve(trk[k], total ? total : 1)) goto done;
            trk[k]->n = total;
        }
        phmm_cut_job j;
        memset(&j, 0, sizeof j);
        j.raw = r->buf;
        j.line_off = lo;
        j.line_len = ll;
        j.roff = roff;
        for (int k = 0; k < 5; k++) j.trk[k] = trk[k]->p;
        j.n_lines = n_reads;
        j.first_bad = -1;
        j.parts = agx_host_threads_c();
        if ((uint32_t)j.parts > n_reads / 256u + 1u) j.parts = (int)(n_reads / 256u + 1u);
        pthread_mutex_init(&j.mu, NULL);
        if (j.parts <= 1) {
            j.parts = 1;
            phmm_cut_part(0, &j);
        } else
            agx_pool_run_c(j.parts, phmm_cut_part, &j);
        pthread_mutex_destroy(&j.mu);
        if (j.first_bad >= 0) {
            const uint32_t *rreg = (const uint32_t *)m->rreg.p;
            uint32_t g = 0;
            while (g + 1 < n_regions && rreg[g + 1] <= (uint32_t)j.first_bad) g++;
            const int i = (int)((uint32_t)j.first_bad - rreg[g]);
            if (j.bad_kind[0] == 1)
                agx_set_error("region %u, read %d: line too short to hold five fields", r->regions_done + g + 1, i);
            else {
                const uint64_t sl = ll[j.first_bad];
                agx_set_error("region %u, read %d: field %d shorter than the read length %zu", r->regions_done + g + 1, i, j.bad_kind[0] - 2,
                              (size_t)((sl - 4) / 5));
            }
            rc = AGX_E_IO;
            goto done;
        }
    }
    m->pub.desc.read_bases = m->rb.p;
    m->pub.desc.q_base = m->qb.p;
    m->pub.desc.q_ins = m->qi.p;
    m->pub.desc.q_del = m->qd.p;
    m->pub.desc.q_gcp = m->qg.p;
    m->pub.desc.read_off = (const uint64_t *)m->roff.p;
    m->pub.desc.n_reads = n_reads;
    m->pub.desc.hap_bases = m->hb.p;
    m->pub.desc.hap_off = (const uint64_t *)m->hoff.p;
    m->pub.desc.n_haps = n_haps;
    m->pub.desc.region_read = (const uint32_t *)m->rreg.p;
    m->pub.desc.region_hap = (const uint32_t *)m->hreg.p;
    m->pub.desc.n_regions = n_regions;
    r->regions_done += n_regions;
    rc = AGX_OK;
done:
    r->nr = nr;
    r->nh = nh;
    free(line_off.p);
    free(line_len.p);
    if (rc == AGX_E_NOMEM) agx_set_error("agx_phmm_reader_next: out of memory");
    if (rc != AGX_OK) {
        agx_phmm_text_free(m ? &m->pub : NULL);
        m = NULL;
        r->finished = 1;
    }
    *out = m ? &m->pub : NULL;
    return rc;
}

int agx_phmm_text_read(const char *path, agx_phmm_text **out)
{
    if (!out || !path) {
        agx_set_error("agx_phmm_text_read: null argument");
        return AGX_E_ARG;
    }
    *out = NULL;
    agx_phmm_reader *r = NULL;
    int rc = agx_phmm_reader_open(path, &r);
    if (rc != AGX_OK) return rc;
    rc = agx_phmm_reader_next(r, INT64_MAX, out);
    agx_phmm_reader_close(r);
    return rc;
}
