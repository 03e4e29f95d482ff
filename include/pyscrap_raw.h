/* include/pyscrap_raw.h -- the prototypes of the reference's python/pyscrap.h that
 * libscrappie_hip.so provides: python/pyscrap.h:2-26 (signal preparation, matrix helpers, the four raw
 * network entry points, transducer and CRF decoding) and :62 (get_raw_model_stride_from_string).
 *
 * This is the `cdef` text for the reference's cffi build when ONLY the raw basecalling path is routed
 * to the GPU library (option A): plain prototypes, no preprocessor lines, the
 * types (raw_table, _Mat, scrappie_matrix, const_scrappie_matrix) as python/build.py:68-109 declares them.
 *
 * NOT provided (they stay with the reference's own sources, option B): squiggle_r94, squiggle_r94_rna,
 * squiggle_r10 (pyscrap.h:28-30), squiggle_match_viterbi, squiggle_match_forward (:33-38),
 * are_bounds_sane, map_to_sequence_forward, map_to_sequence_forward_banded, map_to_sequence_viterbi,
 * map_to_sequence_viterbi_banded (:41-58), encode_bases_to_integers (:61), detect_events (:65).
 */
void medmad_normalise_array(float *x, size_t n);
raw_table trim_and_segment_raw(raw_table rt, size_t trim_start, size_t trim_end,
                               size_t varseg_chunk, float varseg_thresh);
raw_table trim_raw_by_mad(raw_table rt, size_t chunk_size, float perc);

scrappie_matrix mat_from_array(const float * x, size_t nr, size_t nc);
scrappie_matrix free_scrappie_matrix(scrappie_matrix mat);

scrappie_matrix nanonet_rgrgr_r94_posterior(const raw_table signal, float min_prob,
                                            float tempW, float tempb, bool return_log);
scrappie_matrix nanonet_rgrgr_r941_posterior(const raw_table signal, float min_prob,
                                             float tempW, float tempb, bool return_log);
scrappie_matrix nanonet_rgrgr_r10_posterior(const raw_table signal, float min_prob,
                                            float tempW, float tempb, bool return_log);
float decode_transducer(const_scrappie_matrix logpost, float stay_pen, float skip_pen,
                        float local_pen, int *seq, bool allow_slip);
char *overlapper(const int *seq, size_t n, int nkmer, int *pos);

scrappie_matrix nanonet_rnnrf_r94_transitions(const raw_table signal, float min_prob,
                                              float tempW, float tempb, bool return_log);
float decode_crf(const_scrappie_matrix trans, int * path);
char * crfpath_to_basecall(int const * path, size_t npos, int * pos);
scrappie_matrix posterior_crf(const_scrappie_matrix trans);

int get_raw_model_stride_from_string(const char * modelstr);

/* additions of the GPU library the binding needs (weights are data, not compiled in) */
int scrappie_hip_register_model(const char *name, const char *path);
const char *scrappie_hip_last_error(void);
