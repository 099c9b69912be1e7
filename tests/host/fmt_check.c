/* agx_fmt_f6_line against snprintf("%f\n") -- tests/test_host_sanitizers.py::test_fast_f6_formatter_is_printf. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "agx_fmt.h"

static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static long bad = 0, total = 0;
static void check(double x)
{
    char a[AGX_FMT_F6_MAX], b[AGX_FMT_F6_MAX];
    const int la = agx_fmt_f6_line(a, x), lb = snprintf(b, sizeof b, "%f\n", x);
    total++;
    if (la != lb || memcmp(a, b, (size_t)la)) {
        if (bad++ < 10) fprintf(stderr, "MISMATCH %a: got %.*s want %s", x, la - 1, a, b);
    }
}
int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    const double special[] = {0.0, -0.0, 1.0, -1.0, 0.5e-6, 1.5e-6, 2.5e-6, -0.5e-6, 0.0078125, 0.0234375, 1.0 / 128, 3.0 / 128, 5.0 / 128,
                              0.9999995, 0.99999949999999, 0.9999994999999999, 999999.9999995, 9.0e9, 8.9999999999e9, 1e10, 1e300, -1e300,
                              1e-300, -1e-300, 4.9e-324, INFINITY, -INFINITY, NAN, -NAN, -4.485565, 123456.7890125, 0.1, 0.2, 0.3};
    for (size_t i = 0; i < sizeof special / sizeof special[0]; i++) check(special[i]);
    for (long k = -70000; k <= 70000; k++) {  /* every multiple of 2^-7 and 2^-12 near zero: exact ties and near-ties */
        check((double)k / 128.0);
        check((double)k / 4096.0);
        check(nextafter((double)k / 128.0, 1e9));
        check(nextafter((double)k / 128.0, -1e9));
    }
    for (long i = 0; i < n; i++) {
        const uint64_t r = rnd();
        const double u = (double)(r >> 11) / 9007199254740992.0; /* [0, 1) */
        check(-u * 700.0);                        /* log10 likelihoods */
        check((u - 0.5) * 2e4);
        check(u * 1e-5);
        check((double)(int64_t)(rnd() % 2000000001ull - 1000000000ll) / 2e6); /* multiples of 0.5e-6 as decimals: near-ties */
        check(((double)(rnd() % 9000000000ull) + u) * (rnd() & 1 ? 1 : -1));
        double any;
        const uint64_t bits = rnd();
        memcpy(&any, &bits, 8);
        check(any); /* any bit pattern */
    }
    printf(bad ? "FMT_CHECK_FAILED %ld of %ld\n" : "FMT_CHECK_OK %ld values\n", bad ? bad : total, total);
    return bad ? 1 : 0;
}
