/*
 * "%f\n" of a double, byte for byte what printf writes (antidiagsPairHMM.c:459,461 prints every likelihood twice;
 * glibc's general-purpose conversion takes 100-150 ns a value -- 7 M values/s, a third of what the device delivers).
 * Values below 9e9 in magnitude take the exact short way: the integer part, then the fraction times 10^6 as a rounded
 * product p and its exact remainder e = fma(fraction, 1e6, -p); p's own fraction against one half decides the rounding,
 * e breaks what looks like a tie, a true tie goes to the even neighbour (printf rounds the exact binary value to
 * nearest-even).  Everything else (inf, nan, huge) goes through snprintf.  tests/host/fmt_check.c compares the two on
 * hundreds of millions of values, ties included.
 */
#ifndef AGX_FMT_H
#define AGX_FMT_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define AGX_FMT_F6_MAX 336 /* "%f\n" of -DBL_MAX: 1 + 309 + 1 + 6 + 1 bytes and the NUL */
/* writes "<value>\n" to buf (AGX_FMT_F6_MAX bytes), returns its length */
static inline int agx_fmt_f6_line(char *buf, double x)
{
    const double a = fabs(x);
    if (!(a < 9.0e9)) return snprintf(buf, AGX_FMT_F6_MAX, "%f\n", x); /* also NaN */
    uint64_t ip = (uint64_t)a;
    const double fr = a - (double)ip; /* exact: a < 2^53 */
    const double p = fr * 1e6, e = fma(fr, 1e6, -p);
    uint64_t n = (uint64_t)p;
    const double d = p - (double)n; /* exact */
    if (d > 0.5 || (d == 0.5 && (e > 0 || (e == 0 && (n & 1u))))) n++;
    if (n >= 1000000u) {
        n -= 1000000u;
        ip++;
    }
    char *w = buf;
    if (signbit(x)) *w++ = '-';
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + ip % 10u);
        ip /= 10u;
    } while (ip);
    while (k) *w++ = tmp[--k];
    *w++ = '.';
    for (int i = 5; i >= 0; i--) {
        w[i] = (char)('0' + n % 10u);
        n /= 10u;
    }
    w += 6;
    *w++ = '\n';
    return (int)(w - buf);
}
#endif
