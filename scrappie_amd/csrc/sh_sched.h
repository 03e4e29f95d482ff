/* sh_sched.h -- lane schedule of the kernels that are recurrences over a read's
 * blocks: the GRU (two lanes per workgroup) and the Viterbi decoder (one).
 *
 * A "lane" is the part of a workgroup that steps one tile (16 reads) through its
 * blocks: a group of S/16 waves of a 2-lane GRU workgroup, or a whole decoder workgroup.  A launch group of 10 000 reads is 625
 * tiles for 512 lanes (256 CUs): whole tiles per lane would leave 2- and
 * 3-tile CUs and the launch would last as long as the 3-tile ones.  Tiles
 * are therefore laid end to end over the lanes and cut at the lane capacity
 * M = max(longest tile, ceil(total blocks / lanes)) (McNaughton's wrap-around
 * rule): a tile that does not fit the rest of a lane runs its LAST blocks
 * there and its FIRST blocks at the start of the next lane, and hands its
 * state over through HBM.  Since no tile is longer than M the two pieces do not
 * overlap in time; the consumer still checks an arrival flag.  Lanes are
 * numbered so that the producing piece sits in the lower-numbered workgroup.
 *
 * Plain host C++ (no HIP): unit-tested on CPU through scrappie_hip_gru_schedule.
 */
#ifndef SH_SCHED_H
#define SH_SCHED_H
#include <algorithm>
#include <vector>

struct ShGruSeg { int tile, s0, s1, pad; };      /* steps [s0, s1) of `tile`; pad: piece ordinal (decoder) */

struct ShGruSchedule {
    int nwg = 0;                                 /* workgroups (lpw lanes each) */
    int capacity = 0;                            /* M */
    std::vector<int> lane_off;                   /* [lpw nwg + 1] */
    std::vector<ShGruSeg> seg;
    std::vector<int> wg_iter;                    /* [nwg] steps of the longer lane */
};

static inline void sh_lane_schedule(const int *tile_T, size_t ntile, int ncu, int lpw, ShGruSchedule &out,
                                    bool allow_split = true) {
    out = ShGruSchedule();
    std::vector<int> live;
    long long W = 0;
    int maxT = 0;
    for (size_t t = 0; t < ntile; t++) if (tile_T[t] > 0) { live.push_back((int)t); W += tile_T[t]; maxT = std::max(maxT, tile_T[t]); }
    if (live.empty()) { out.lane_off.assign(1, 0); return; }
    const long long nlive = (long long)live.size();
    /* as many workgroups as there are tiles (up to one per CU) before any workgroup gets a second lane:
     * a lane that has its CU to itself steps faster, and with few, long tiles the launch lasts as long
     * as the longest tile's steps */
    const int nwg = (int)std::min<long long>(ncu, nlive);
    const int L = lpw * nwg;
    std::vector<std::vector<ShGruSeg>> pos(L);   /* position p -> lane L-1-p */
    int M;
    if (nlive <= L) {                            /* enough lanes: whole tiles, nothing to hand over */
        M = maxT;
        /* first lanes in order of length; further lanes filled from the other end, so that the longest
         * tiles (callers sort by length) share their workgroup with the shortest ones, or with nothing */
        std::vector<int> bylen(live);
        std::stable_sort(bylen.begin(), bylen.end(), [&](int x, int y) { return tile_T[x] > tile_T[y]; });
        for (long long i = 0; i < nlive; i++) {
            const int sub = (int)(i / nwg), j = (int)(i % nwg);
            const int wg = (sub & 1) ? (int)std::min<long long>(nwg, nlive - (long long)sub * nwg) - 1 - j : j;
            const int t = bylen[i];
            pos[L - 1 - (wg * lpw + sub)].push_back({t, 0, tile_T[t], 0});
        }
    } else if (!allow_split) {                   /* whole tiles only (no hand-over): longest first onto the least loaded lane */
        std::vector<int> load(L, 0);
        for (long long i = 0; i < nlive; i++) {
            const int p = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            pos[p].push_back({live[i], 0, tile_T[live[i]], 0});
            load[p] += tile_T[live[i]];
        }
        M = *std::max_element(load.begin(), load.end());
    } else {
        M = (int)std::max<long long>(maxT, (W + L - 1) / L);
        int p = 0, used = 0;
        for (int t : live) {
            const int T = tile_T[t];
            const int room = M - used;
            if (T <= room) {
                pos[p].push_back({t, 0, T, 0});
                used += T;
            } else {
                if (room > 0) pos[p].push_back({t, T - room, T, 0});          /* last blocks, end of this lane */
                p++;
                pos[p].push_back({t, 0, T - room, 0});                       /* first blocks, start of the next */
                used = T - room;
            }
            if (used == M && p + 1 < L) { p++; used = 0; }
        }
    }
    out.nwg = nwg; out.capacity = M;
    out.lane_off.assign(L + 1, 0);
    out.wg_iter.assign(nwg, 0);
    for (int ln = 0; ln < L; ln++) {
        const std::vector<ShGruSeg> &v = pos[L - 1 - ln];
        int steps = 0;
        for (const ShGruSeg &sg : v) { out.seg.push_back(sg); steps += sg.s1 - sg.s0; }
        out.lane_off[ln + 1] = (int)out.seg.size();
        out.wg_iter[ln / lpw] = std::max(out.wg_iter[ln / lpw], steps);
    }
}
static inline void sh_gru_schedule(const int *tile_T, size_t ntile, int ncu, ShGruSchedule &out) {
    sh_lane_schedule(tile_T, ntile, ncu, 2, out);
}

/* Decoder pieces.  The Viterbi kernel keeps a tile's 64 KB of scores in LDS, one workgroup
 * per CU, so 625 tiles on 256 CUs would run as three rounds of whole tiles (2.44 needed).
 * Each tile is cut into K pieces of equal length instead, numbered piece-major (all first
 * pieces, then all second pieces, ...): the hardware hands workgroups to CUs in index
 * order, a tile's earlier piece is therefore always dispatched before the later one, and
 * the later one waits on an arrival flag only if it must.  K minimises
 * ceil(ntile * K / ncu) / K over 1..4. */
static inline int sh_pieces_per_tile(long long nlive, int ncu) {
    if (nlive <= ncu) return 1;
    int best = 1;
    double bestv = 1e30;
    for (int k = 1; k <= 4; k++) {
        const double v = (double)((nlive * k + ncu - 1) / ncu) / k;
        if (v < bestv - 1e-9) { bestv = v; best = k; }
    }
    return best;
}
static inline void sh_piece_schedule(const int *tile_T, size_t ntile, int ncu, std::vector<ShGruSeg> &seg,
                                     bool allow_split = true) {
    seg.clear();
    long long nlive = 0;
    for (size_t t = 0; t < ntile; t++) nlive += tile_T[t] > 0;
    const int K = allow_split ? sh_pieces_per_tile(nlive, ncu) : 1;
    for (int k = 0; k < K; k++)
        for (size_t t = 0; t < ntile; t++) {
            const int T = tile_T[t];
            if (T <= 0) continue;
            const int len = (T + K - 1) / K;
            const int s0 = std::min(T, k * len), s1 = std::min(T, (k + 1) * len);
            if (s1 > s0) seg.push_back({(int)t, s0, s1, k});          /* pad = ordinal of the piece within its tile */
        }
}
#endif
