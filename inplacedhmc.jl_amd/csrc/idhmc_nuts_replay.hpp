// idhmc_nuts_replay.hpp -- the tree's scalar bookkeeping of a NUTS transition, evaluated AFTER the tree (nuts_defer): acceptance statistic
// and the winner of the biased progressive sampling from the log of the leaves' Delta, one level of up to 64 leaves per pass.
// Included by idhmc_nuts_kernel.hpp (after AccStat, MergeScalars, nuts_merge_scalars, nuts_logaddexp, nuts_randexp_batch, uni / usi);
// replaces, for the separable densities, what reference adjacent_tree / sample_trajectory compute on the way
// (src/tree.jl:238-263, 335-363, 410-434; src/NUTS.jl:32-45, 68-84).
#pragma once

namespace idhmc {

// ---- deferred tree bookkeeping (nuts_defer) -------------------------------------------------------------------------------
// What the tree loop leaves behind: Delta of every leaf (dl: the leaf n of the doubling of depth d at index 2^d - 1 + n, the order
// in which the leaves were made), how many doublings completed and were merged into the tree (ndone), and where the doubling after
// them stopped, if it did: stop_kind 1 = divergent leaf stop_n (src/tree.jl:332), 2 = the sub-tree completed by leaf stop_n turned
// in its merge at level stop_k (:358).  nuts_replay evaluates from that record what adjacent_tree / sample_trajectory compute on the
// way (src/tree.jl:335-363, 410-434; src/NUTS.jl:32-45, 68-70): the acceptance statistic of the visited nodes and the winner of the
// biased progressive sampling -- same merges, same operands, same association, same exponential draws at the same addresses, so
// the same bits -- but the merges of one level of up to 64 leaves in ONE pass, lane l standing for leaf l.
struct ReplayIn {
    const double *dl;
    int ndone, stop_kind, stop_n, stop_k;
    uint32_t k0, k1, chain, iter;
};
struct ReplayOut {
    double lsa;          // log of the summed acceptance probabilities of the visited nodes
    int steps;           // ... and their number
    int win_d, win_n;    // the proposal: leaf win_n of the doubling of depth win_d; win_d < 0: the starting point
};
// Levels 0 .. maxlev-1 for the leaves the lanes hold.  Per lane: valid (holds a leaf that was made), nloc (index of the leaf inside
// its sub-tree), nlev (levels the sub-tree has), capk (highest level at which this lane may still merge: the leaf at which the tree
// stopped takes part up to the merge that turned; -1 for a divergent leaf).  A merge at level k lives on the lane of its last leaf
// (k + 1 trailing one bits in nloc) and takes its left operand from 2^k lanes below.  The lane below an active lane holds a leaf
// with an even index, which never merges: it computes the second log-sum-exp of the pair (the acceptance statistic) while the
// active lane computes the first (the weights), so one call serves both (as nuts_merge_scalars does with lane parity).
// Out: w, a = log weight and log acceptance sum of the largest complete sub-tree ending at the lane; lp[k] = logprob2 of the merge
// the lane made at level k (src/tree.jl:261-263); actbits = the levels at which it merged.
IDHMC_DEV void replay_passes(int maxlev, bool valid, int nloc, int nlev, int capk, double &w, double &a, double (&lp)[6], uint32_t &actbits)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k < maxlev) {
            const int str = 1 << k, m2 = 2 * str - 1;
            const bool act = valid && k < nlev && k <= capk && ((nloc & m2) == m2);
            const double wL = __shfl_up(w, str), aL = __shfl_up(a, str);
            const double aLn = __shfl_down(aL, 1), an = __shfl_down(a, 1);
            const double x = act ? wL : aLn, y = act ? w : an;
            const double r = nuts_logaddexp(x, y);
            const double ra = __shfl_up(r, 1);
            if (act) { lp[k] = w - r; w = r; a = ra; actbits |= 1u << k; }
        }
    }
}
// The multinomial picks of one complete sub-tree whose leaves are the lanes with `inrange` (src/NUTS.jl:32-45): a merge draws only
// when logprob2 < 0, in the order the reference makes the merges -- leaf by leaf, and level by level at a leaf -- so the address of a
// merge's draw is `draw` plus the number of drawing merges before it.  win: in = every lane its own leaf, out (last lane) = the winner.
IDHMC_DEV void replay_picks(int nlevels, bool inrange, int lane, const double (&lp)[6], uint32_t actbits, int &win,
                            uint32_t &draw, uint32_t &ebase, double &ebatch, const ReplayIn &in)
{
    uint32_t nb = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (inrange && ((actbits >> k) & 1u) && !(lp[k] >= 0.0)) nb |= 1u << k;
    const int cnt = __builtin_popcount(nb);
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    const int excl = inc - cnt;
    const int total = __builtin_amdgcn_readlane(inc, 63);
    if (total == 0) return;             // every merge keeps the later sub-tree's pick: the last leaf wins
    if (draw < ebase || draw + (uint32_t)total > ebase + 64u) {
        ebase = draw;
        ebatch = nuts_randexp_batch(in.k0, in.k1, in.chain, in.iter, ebase);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (k < nlevels) {
            const bool act = inrange && ((actbits >> k) & 1u);
            const int idx = (int)(draw - ebase) + excl + __builtin_popcount(nb & ((1u << k) - 1u));
            const double e = __shfl(ebatch, idx & 63);
            const bool pick2 = !((nb >> k) & 1u) || (e > -lp[k]);
            const int winL = __shfl_up(win, 1 << k);
            if (act && !pick2) win = winL;
        }
    }
    draw += (uint32_t)total;
}
// S: the wavefront's LevelScalars; its omega / lsa / zeta entries 6.. serve as the stack of 64-leaf blocks of a long doubling
template <class LS>
IDHMC_DEV ReplayOut nuts_replay(const ReplayIn &in, LS &S)
{
    const int lane = threadIdx.x & 63;
    uint32_t draw = 0, ebase = 0;
    double ebatch = nuts_randexp_batch(in.k0, in.k1, in.chain, in.iter, 0u);
    auto take_draw = [&]() -> double {
        if (draw < ebase || draw >= ebase + 64u) {
            ebase = draw;
            ebatch = nuts_randexp_batch(in.k0, in.k1, in.chain, in.iter, ebase);
        }
        const double e = read_lane(ebatch, usi((int)(draw - ebase)));
        ++draw;
        return e;
    };
    const int nd = in.ndone + (in.stop_kind ? 1 : 0);      // doublings with leaves on record
    // ---- the doublings of up to 32 leaves share one set of passes: the leaf n of the doubling d sits on lane 2^d - 1 + n ----------
    const int g = lane + 1;
    const int dl_ = 31 - __builtin_clz(g);                  // this lane's doubling
    const int nl = g - (1 << dl_);                          // ... and leaf
    const bool isstop = in.stop_kind != 0 && dl_ == in.ndone;
    const bool valid = dl_ <= 5 && (dl_ < in.ndone || (isstop && nl <= in.stop_n));
    const int capk = (isstop && nl == in.stop_n) ? (in.stop_kind == 2 ? in.stop_k : -1) : 99;
    const double dlt = valid ? __hip_atomic_load(in.dl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    double w = dlt, a = dlt < 0.0 ? dlt : 0.0;              // leaf: omega = Delta, log alpha = min(Delta, 0) (src/NUTS.jl:76-78, 179)
    double lp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    uint32_t actbits = 0;
    int win = nl;
    replay_passes(nd > 5 ? 5 : nd - 1, valid, nl, dl_, capk, w, a, lp, actbits);

    ReplayOut out;
    out.win_d = -1; out.win_n = 0;
    AccStat v{-kInf, 0};
    double top_omega = 0.0;
    // ---- sample_trajectory's loop over the doublings (src/tree.jl:395-441) ------------------------------------------------------
    for (int d = 0; d < nd; ++d) {
        const bool partial = d == in.ndone;                 // the doubling that stopped: only its acceptance statistic counts (:414-417)
        const int m = 1 << d;
        double sub_w, sub_a;
        int sub_win;
        double vres = 0.0;                                  // partial: log acceptance sum of what was visited
        if (d <= 5) {
            const int last = 2 * m - 2;                     // lane of the doubling's last leaf
            if (!partial) {
                replay_picks(d, valid && dl_ == d, lane, lp, actbits, win, draw, ebase, ebatch, in);
                sub_w = read_lane(w, last); sub_a = read_lane(a, last);
                sub_win = __builtin_amdgcn_readlane(win, last);
            } else {
                const int base = m - 1, ns = in.stop_n;
                vres = read_lane(a, base + ns);             // the divergent leaf, or the merge that turned (NUTS.jl:76-78, tree.jl:347)
                for (int j = in.stop_kind == 2 ? in.stop_k + 1 : 0; j < d; ++j)
                    if ((ns >> j) & 1) {                    // the complete left siblings, bottom up (tree.jl:347-348)
                        const int e = ((ns >> (j + 1)) << (j + 1)) + (1 << j) - 1;
                        vres = nuts_logaddexp(read_lane(a, base + e), vres);
                    }
            }
        } else {
            // ---- a doubling of 2^d >= 64 leaves: 64-leaf blocks, one set of passes each, and the reference's own cascade above -----
            const int ns = partial ? in.stop_n : m - 1;     // last leaf on record
            const int nblk = (ns >> 6) + 1;
            double cw = 0.0, ca = 0.0;                      // the sub-tree in hand after a block: omega, log alpha sum, pick
            int cwin = 0;
            bool turned_high = false;
            for (int b = 0; b < nblk; ++b) {
                const bool lastb = partial && b == nblk - 1;
                const int nsl = ns & 63;
                const bool bvalid = !lastb || lane <= nsl;
                const int bcap = (lastb && lane == nsl) ? (in.stop_kind == 2 ? (in.stop_k < 6 ? in.stop_k : 5) : -1) : 99;
                const double bd = bvalid ? __hip_atomic_load(in.dl + ((m - 1) + 64 * b + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                double bw = bd, ba = bd < 0.0 ? bd : 0.0;
                double blp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                uint32_t bact = 0;
                int bwin = 64 * b + lane;
                replay_passes(6, bvalid, lane, 6, bcap, bw, ba, blp, bact);
                if (lastb && !(in.stop_kind == 2 && in.stop_k >= 6)) {
                    // stopped inside this block: fold as above, then the parked blocks
                    vres = read_lane(ba, nsl);
                    for (int j = in.stop_kind == 2 ? in.stop_k + 1 : 0; j < d; ++j)
                        if ((ns >> j) & 1) {
                            const int e = ((nsl >> (j + 1)) << (j + 1)) + (1 << j) - 1;
                            vres = nuts_logaddexp(j < 6 ? read_lane(ba, e) : S.lsa[j], vres);
                        }
                    break;
                }
                if (!partial) replay_picks(6, true, lane, blp, bact, bwin, draw, ebase, ebatch, in);
                cw = read_lane(bw, 63); ca = read_lane(ba, 63);
                cwin = __builtin_amdgcn_readlane(bwin, 63);
                int kk = 6;
                while ((b >> (kk - 6)) & 1) {               // complete pairs of blocks: the reference's merge, one at a time
                    const MergeScalars ms = nuts_merge_scalars(S.lsa[kk], ca, S.omega[kk], cw);
                    if (lastb && kk == in.stop_k) {         // this merge turned (tree.jl:358): its acceptance statistic, then the parked ones
                        vres = ms.lsa;
                        for (int j = kk + 1; j < d; ++j)
                            if ((ns >> j) & 1) vres = nuts_logaddexp(S.lsa[j], vres);
                        turned_high = true;
                        break;
                    }
                    if (!partial) {
                        const double logprob2 = cw - ms.omega;
                        bool pick2 = uni(logprob2 >= 0.0);
                        if (!pick2) pick2 = uni(take_draw() > -logprob2);
                        if (!pick2) cwin = usi(S.zeta[kk]);
                    }
                    cw = ms.omega; ca = ms.lsa;
                    ++kk;
                }
                if (turned_high) break;
                S.omega[kk] = cw; S.lsa[kk] = ca; S.zeta[kk] = cwin;      // park until the right sibling is complete
            }
            sub_w = cw; sub_a = ca; sub_win = cwin;
        }
        if (partial) {
            v = AccStat{nuts_logaddexp(v.lsa, vres), v.steps + in.stop_n + 1};                        // tree.jl:414, :417
            break;
        }
        // combine_proposals_and_logweights(is_doubling = true) and the acceptance statistic, tree.jl:414, 431-433
        const MergeScalars mt = nuts_merge_scalars(v.lsa, sub_a, top_omega, sub_w);
        v = AccStat{mt.lsa, v.steps + m};
        const double logprob2 = sub_w - top_omega;
        bool pick2 = uni(logprob2 >= 0.0);
        if (!pick2) pick2 = uni(take_draw() > -logprob2);
        if (pick2) { out.win_d = d; out.win_n = sub_win; }
        top_omega = mt.omega;
    }
    out.lsa = v.lsa;
    out.steps = v.steps;
    return out;
}

}  // namespace idhmc
