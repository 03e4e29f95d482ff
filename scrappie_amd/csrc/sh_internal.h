/* sh_internal.h -- symbols shared between the host C (sh_host.c) and the HIP
 * translation unit (scrappie_hip.hip); not part of the public ABI. */
#ifndef SH_INTERNAL_H
#define SH_INTERNAL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
float sh_medianf(const float *x, size_t n, float *scratch);
float sh_madf(const float *x, size_t n, const float *med, float *scratch);
int sh_kmerlength(int nstate);
/* homopolymer correction from the 5-row side buffer [nblock][5] = {A,C,G,T homopolymer k-mer, stay} */
int sh_homopolymer_side(const float *side, int *path, int nblock, int nstate);
#ifdef __cplusplus
}
#endif
#endif
