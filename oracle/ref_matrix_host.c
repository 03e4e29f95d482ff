/* oracle/ref_matrix_host.c -- HOSTING SHIM for oracle/_ref/libref_decode.so.
 *
 * The reference's decode.c (compiled unmodified from /root/reference/src) imports
 * four allocator symbols from scrappie_matrix.c.  scrappie_matrix.c itself cannot
 * be compiled in this image (its first include is <cblas.h>, which the image
 * lacks, and no stand-in header is written).  The four allocators below are
 * therefore provided by the oracle's own container (oracle.c: orc_make_mat etc.,
 * ABI-identical to _Mat/_iMat).  They contain no arithmetic: zeroed, 16-byte
 * aligned, 4-row-padded buffers.  Everything decode.c computes runs as shipped.
 * DESIGN.md states this openly; libref_pure.so has no such hosting.
 */
#include "scrappie_matrix.h"   /* the reference's own header: prototypes must match */
#include "oracle.h"

scrappie_matrix make_scrappie_matrix(size_t nr, size_t nc) {
    return (scrappie_matrix)orc_make_mat(nr, nc);
}
scrappie_matrix free_scrappie_matrix(scrappie_matrix mat) {
    return (scrappie_matrix)orc_free_mat((orc_mat *)mat);
}
scrappie_imatrix make_scrappie_imatrix(size_t nr, size_t nc) {
    return (scrappie_imatrix)orc_make_imat(nr, nc);
}
scrappie_imatrix free_scrappie_imatrix(scrappie_imatrix mat) {
    return (scrappie_imatrix)orc_free_imat((orc_imat *)mat);
}
