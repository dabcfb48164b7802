/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the convolutional matching-pursuit hot path.
 * See hsc_oracle.h for scope, pinning and the rules on who may call this.
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off -fPIC -shared).
 */
#include "hsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int hsco_version(void) { return 1; }

/* utils.py:76-161 -- shared index rule of peek / overlapAdd / overlapReplace.
 * Even W: element covers t-(W/2-1) .. t+W/2; odd W: t-W//2 .. t+W//2 (utils.py:84-99). */
int hsco_span(int T, int W, int t, int* start, int* end, int* estart, int* eend)
{
    const int s = t - (W - 1) / 2;
    const int e = t + W / 2 + 1; /* exclusive */
    const int cs = s < 0 ? 0 : s;
    const int ce = e > T ? T : e;
    *start = cs; *end = ce;
    *estart = cs - s;
    *eend = W - (e - ce);
    return ce - cs > 0 ? ce - cs : 0;
}

#define REAL float
#define SFX(n) n##_f32
#define RFMA fmaf
#define RABS fabsf
#include "hsc_oracle_impl.h"
#undef REAL
#undef SFX
#undef RFMA
#undef RABS

#define REAL double
#define SFX(n) n##_f64
#define RFMA fma
#define RABS fabs
#include "hsc_oracle_impl.h"
#undef REAL
#undef SFX
#undef RFMA
#undef RABS
