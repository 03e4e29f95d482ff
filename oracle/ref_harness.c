/* oracle/ref_harness.c -- thin exports around the REFERENCE's own header-inline
 * vector math (src/util.h:170-198 over src/sse_mathfun.h), compiled against the
 * headers where they lie under /root/reference/src.  TEST INFRASTRUCTURE ONLY:
 * built only in the build container (the GPU box has no /root/reference), output
 * only under oracle/_ref/.  No reference source is copied: this file contains
 * only loops calling the reference's inline functions.
 */
#include <stddef.h>
#include "util.h"   /* -I/root/reference/src */

#define LANEWISE(NAME, FN)                                                   \
    void NAME(const float *in, float *out, size_t n) {                       \
        for (size_t i = 0; i < n; i += 4) {                                  \
            float a[4] = {0, 0, 0, 0}, r[4];                                 \
            for (size_t l = 0; l < 4 && i + l < n; l++) a[l] = in[i + l];    \
            __m128 v = FN(_mm_loadu_ps(a));                                  \
            _mm_storeu_ps(r, v);                                             \
            for (size_t l = 0; l < 4 && i + l < n; l++) out[i + l] = r[l];   \
        }                                                                    \
    }

LANEWISE(ref_expfv, expfv)
LANEWISE(ref_logfv, logfv)
LANEWISE(ref_logisticfv, logisticfv)
LANEWISE(ref_tanhfv, tanhfv)
LANEWISE(ref_elufv, elufv)

float ref_logsumexpf(float x, float y) { return logsumexpf(x, y); }
