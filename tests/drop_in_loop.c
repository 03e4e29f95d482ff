/* The reference's per-read loop (scrappie_raw.c:265-315 inside the OpenMP loop of :355,387), written against this library's header with the
 * reference's own function names only: read -> trim -> normalise -> get_posterior_function(model)(...) -> decode_transducer -> homopolymer_path ->
 * overlapper, one read per iteration, iterations spread over OpenMP threads.  What a maintainer gets by linking libscrappie_hip.so under the
 * reference's host code unchanged.  Weights: $SCRAPPIE_MODEL_DIR/<model>.scrm.  Output: one line per read, "<index> <score> <nblock> <bases>",
 * in index order.
 *     drop_in_loop <model> <local_pen> <repeat> file...            (built by tests/test_cli.py) */
#include "scrappie_hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const char *model = argv[1];
    const float local_pen = (float)atof(argv[2]);
    const int repeat = atoi(argv[3]), nfile = argc - 4, n = nfile * repeat;
    const enum raw_model_type mt = get_raw_model(model);
    posterior_function_ptr calcpost = get_posterior_function(mt);
    if (!calcpost) return 3;
    char **bases = calloc((size_t)n, sizeof(char *));
    float *score = calloc((size_t)n, sizeof(float));
    size_t *nblk = calloc((size_t)n, sizeof(size_t));
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n; i++) {
        raw_table rt = scrappie_hip_read_raw(argv[4 + i % nfile], true);
        if (!rt.raw) continue;
        /* every copy of a file a little shorter than the one before, so that the reads of a launch differ */
        rt.n -= (size_t)(i / nfile) * 37; rt.end = rt.n;
        rt = trim_and_segment_raw(rt, 200, 10, 100, 0.0f);
        if (!rt.raw) continue;
        medmad_normalise_array(rt.raw + rt.start, rt.end - rt.start);
        scrappie_matrix post = calcpost(rt, 1e-5f, 1.0f, 1.0f, true);
        if (post) {
            const size_t nblock = post->nc;
            int *path = calloc(nblock + 1, sizeof(int)), *pos = calloc(nblock + 1, sizeof(int));
            score[i] = decode_transducer(post, 0.0f, 0.0f, local_pen, path, false);
            homopolymer_path(post, path, HOMOPOLYMER_MEAN);
            bases[i] = overlapper(path, nblock + 1, (int)post->nr - 1, pos);
            nblk[i] = nblock;
            free(path); free(pos);
            post = free_scrappie_matrix(post);
        }
        free(rt.raw); free(rt.uuid);
    }
    for (int i = 0; i < n; i++) printf("%d %.6f %zu %s\n", i, (double)score[i], nblk[i], bases[i] ? bases[i] : "-");
    if (getenv("DROP_IN_STATS")) {          /* how the calls were coalesced */
        unsigned long long a[3], b[3];
        scrappie_hip_coalescer_stats(a);
        scrappie_hip_decode_coalescer_stats(b);
        fprintf(stderr, "%d reads: %llu network launch groups (largest %llu reads), %llu decode launches (largest %llu)\n", n, a[0], a[2], b[0], b[2]);
    }
    return 0;
}
