// bioik_kernels.h — workgroup-level bodies of the gfx950 kernels.
//
// Mapping: one workgroup owns one (query, island).  Its lanes are the individuals of the current
// generation: lane r reproduces child r (and r + nthreads, ...) from the two elites held in LDS into its own column
// of the [op][lane] genotype array in LDS, walks the joint program for it (exact FK) or evaluates the linearised
// model, and the elitist top-2 selection is a wave64 butterfly of (fitness, position) pairs (+ one LDS hop across
// waves).  Winners are re-derived from the counter RNG instead of being stored.  The memetic phase spreads its D
// finite differences and the two support points of the line search over lanes.  Everything of a query (seed, goal
// parameters, elites, genotype columns, linear model, per-joint frame chain) lives in LDS; HBM sees the query once
// on the way in and the result once on the way out.
//
// Reference behaviour restated here: src/ik_evolution_2.cpp:111-230 (initialize), :328-646 (step),
// src/ik_parallel.h:148-190 (island loop, budget form), :220-269 (best island).
#pragma once
#include <type_traits>

#include "bioik_device.h"


#ifndef BIOIK_CANDIDATE_BOUND
#define BIOIK_CANDIDATE_BOUND 1e300  // a candidate of the memetic line search with a gene of this magnitude or more is no candidate (quirk Q7; the oracle's default mode has the same bound)
#endif
#ifndef BIOIK_COOP_WALKS
#define BIOIK_COOP_WALKS 1  // single-individual walks of the solver share the joints' trigonometry over the lanes (fk_walk<COOP>); 0: every lane repeats it
#endif

struct LdsLayout {  // offsets in doubles; "g_" regions exist once per species group (stride g_stride)
    int seed, par, pop, sol, prefix, state, clip, xcol, slots, g_first, g_stride, total;
    int xn, gv, frames, tips, delta, base, grad, red, sec, order, bc;  // offsets inside a group region
    int xm, xp, dv, fc;  // memetic phase (per group): support points x -+ g, gene displacements [4][m], tip-frame components [4][T*8]
    int fitp;            // fit_park: the children's fitness values of one generation [lambda], on the space of the memetic phase's vectors
    int help;            // the helped kernel (solve_body<.., FIXED = 5>): sixteen 32-bit words -- the two main wavefronts' barrier counts, "go" and "done" per species,
                         // a four-word mailbox per species (offsets of the parent's genes and of the momentum table, the stream's counter, the children to walk)
};
BIOIK_HD LdsLayout make_layout(int n_ops, int V, int P, int T, int n_slots, int nthreads, int lambda, int has_secondary, int child_cols = 1,
                               int groups = 1, int slot_sets = 1, int fit_park = 0, int fc_in_pop = 0, int helped = 0) {
    LdsLayout L;
    const int m = n_ops > 0 ? n_ops : 1;
    int o = 0;
    L.seed = o, o += V;
    L.par = o, o += P > 0 ? P : 1;
    L.pop = o, o += 2 * 2 * 2 * 2 * m;  // [species][buffer][individual][genes|momentum][op]
    L.sol = o, o += m;
    L.prefix = o, o += 8;               // frame behind the leading non-gene joints (DevProblem::n_prefix), per query
    L.state = o, o += 2 * 8 + 4 + 4;    // species bookkeeping [2][8], workgroup broadcast slots [4], fitness / success flag of the solution [2] (+2 spare)
    L.clip = o, o += 2 * m;             // RobotInfo clip_min | clip_max per op (robot_info.h:109-113), staged once per query
    L.help = o, o += helped ? 8 : 0;
    L.xcol = o, o += m * nthreads * (child_cols >= 0 ? child_cols : 1);  // genotype columns: [col][op][lane]; none when children are computed where they are read
    L.slots = o, o += n_slots * 7 * nthreads * (slot_sets > 0 ? slot_sets : 1);  // parked branch frames, one set per child a lane walks at once
    int g = 0;  // per species group: line-search vectors, linear model, reduction and pre-selection scratch
    // The per-joint frame chain [m][7] is read by the Jacobian columns right after the walk that publishes it and never again (build_approximator);
    // the line-search vectors (9 m doubles) are written after that: the chain lies on top of them (round 4: 7 m doubles per group less, which is
    // what lets a CU hold 15 instead of 13 queries of the two-armed problem)
    L.frames = g;
    L.xn = g, g += m;
    L.gv = g, g += m;
    L.grad = g, g += m;
    L.xm = g, g += m;
    L.xp = g, g += m;
    L.dv = g, g += 4 * m;
    g += g & 1;  // (two-double alignment of the component blocks)
    // The four frame sets of the line search (fc, 32 T doubles) are written and read inside the memetic phase, when the species' OTHER elite buffer (4 m doubles)
    // holds nothing: the next generation's winners overwrite it.  Where they fit they lie there (fc_in_pop: the solver's layouts; fc = -1) -- for the two-armed
    // problem 1 KiB per query, which is what lets a CU's LDS hold sixteen of its queries instead of fifteen
    L.fc = -1;
    if (!(fc_in_pop && 4 * 8 * (T > 0 ? T : 1) <= 4 * m)) L.fc = g, g += 4 * 8 * (T > 0 ? T : 1);
    L.tips = g, g += T * 7;
    L.delta = g, g += T * m * 7;
    L.base = g, g += m;
    // has_secondary == 2: the pre-selection scratch of the generation loop shares the space of the memetic phase's vectors and
    // linear model (exact-FK generations never read the linear model, and both are rebuilt before every use)
    int n_sort = 2;  // the pre-selection sorts its lambda children in a power-of-two array (solve_body)
    while (n_sort < lambda) n_sort <<= 1;
    const int n_sec = has_secondary ? n_sort : 0, n_order = has_secondary ? n_sort / 2 : 0;
    if (has_secondary == 2) {
        L.sec = 0, L.order = n_sec;
        if (n_sec + n_order > g) g = n_sec + n_order;
    }
    // fit_park (exact-FK generations only, like has_secondary == 2): the fitness of every child of the running generation, written by the lane
    // that scored it and read back by the same lane when the generation's walks are over -- so that no lane carries a best-two across them
    L.fitp = 0;
    if (fit_park && g < lambda) g = lambda;
    L.red = g, g += 4 * (nthreads / 64) + 4;
    L.bc = g, g += 4;  // values broadcast from the group's leading wavefront
    if (has_secondary != 2) {
        L.sec = g, g += n_sec;
        L.order = g, g += n_order;
    }
    L.g_first = o;
    L.g_stride = g;
    o += g * (groups > 0 ? groups : 1);
    L.total = o;
    return L;
}

struct Cand {
    double f;
    int pos;  // position in the reference's child_indices order (ties go to the lower position)
    int id;   // 0/1 = parent, >=2 = evaluated child at sorted position id-2
};
BIOIK_DEV bool cand_better(double f, int pos, double of, int opos) { return (f < of) || (f == of && pos < opos); }

// all-lanes top-2 of (f,pos) over the workgroup: every lane brings its own best two (b1 <= b2); wave64 xor-butterfly
// merging sorted pairs, then one LDS hop across waves.  Positions are unique, so (f,pos) is a total order.
// merge another sorted pair (o1 <= o2) into (b1 <= b2); positions are unique, so (f, pos) is a total order and the top-2 of
// a union does not depend on the order of the merges
// one more candidate into the sorted pair (b1 <= b2).  Selects, no branches: written as conditional stores the compiler turns b1 / b2
// into an address select and keeps them in scratch memory (two scratch round trips per child in the hottest loop).
BIOIK_DEV void top2_insert(double& b1f, int& b1p, double& b2f, int& b2p, double f, int pos) {
    const bool w1 = cand_better(f, pos, b1f, b1p), w2 = cand_better(f, pos, b2f, b2p);
    b2f = w1 ? b1f : (w2 ? f : b2f);
    b2p = w1 ? b1p : (w2 ? pos : b2p);
    b1f = w1 ? f : b1f;
    b1p = w1 ? pos : b1p;
}
BIOIK_DEV void top2_merge(double& b1f, int& b1p, double& b2f, int& b2p, double o1f, int o1p, double o2f, int o2p) {
    const bool w1 = cand_better(o1f, o1p, b1f, b1p);            // the other pair's best beats ours
    const bool k2 = cand_better(b1f, b1p, o2f, o2p);            // then our best against their second
    const bool w2 = cand_better(o1f, o1p, b2f, b2p);            // else their best against our second
    const double n2f = w1 ? (k2 ? b1f : o2f) : (w2 ? o1f : b2f);
    const int n2p = w1 ? (k2 ? b1p : o2p) : (w2 ? o1p : b2p);
    b1f = w1 ? o1f : b1f;
    b1p = w1 ? o1p : b1p;
    b2f = n2f, b2p = n2p;
}
// wave64 xor-butterfly: afterwards every lane holds the two best of the wavefront.  (A DPP reduction with row broadcasts and
// scalar read-back was measured 30 % slower than ds_bpermute rounds on gfx950; the two steps inside a quad are DPP moves.)
// G: lanes of the group that is reduced (>= 64: the whole wavefront; 32: one half of it, the other half is the other species)
// The same result for a whole wavefront (G >= 64) from two wavefront minima.  Every lane's pair is sorted, so the best candidate of
// the wavefront is the least of the lanes' FIRST entries: a minimum over 64 doubles (four DPP steps inside the rows of 16, then the
// four row results through scalar registers), one ballot for the lane that holds it, one v_readlane for its position.  The
// runner-up is the least of what is left: the winner's lane offers its second entry, every other lane its first.  Equal fitness in
// several lanes (clipped clones; +inf when fewer children than lanes were scored) is decided by position, as cand_better does --
// the rare path, one more minimum over the tied lanes' positions.  About a third of the instructions of the merging butterfly.
BIOIK_DEV double wave_min_f64(double v) {
    v = fmin(v, p_quad_xor<1>(v));
    v = fmin(v, p_quad_xor<2>(v));
    v = fmin(v, p_row_mirror<1>(v));
    v = fmin(v, p_row_mirror<0>(v));  // every lane: the minimum of its row of 16
    return fmin(fmin(p_read_lane(v, 0), p_read_lane(v, 16)), fmin(p_read_lane(v, 32), p_read_lane(v, 48)));
}
BIOIK_DEV int imin(int a, int b) { return a < b ? a : b; }
BIOIK_DEV int wave_min_i32(int v) {
    v = imin(v, p_quad_xor<1>(v));
    v = imin(v, p_quad_xor<2>(v));
    v = imin(v, p_row_mirror<1>(v));
    v = imin(v, p_row_mirror<0>(v));
    return imin(imin(p_read_lane(v, 0), p_read_lane(v, 16)), imin(p_read_lane(v, 32), p_read_lane(v, 48)));
}
// lane (wavefront-uniform) of the least (f, pos) among the lanes' candidates, and that candidate
BIOIK_DEV int wave_argmin(double f, int pos, double& mf, int& mp) {
    mf = wave_min_f64(f);
    const unsigned long long tied = p_ballot(f == mf);  // (never empty: the candidates are never NaN, top2_insert)
    int lane;
    if ((tied & (tied - 1ull)) == 0ull) {
        lane = __builtin_ctzll(tied);
    } else {
        const int pm = wave_min_i32(f == mf ? pos : 0x7fffffff);
        lane = __builtin_ctzll(p_ballot(f == mf && pos == pm));
    }
    mp = p_read_lane(pos, lane);
    return lane;
}
BIOIK_DEV void top2_wave64_minima(double& b1f, int& b1p, double& b2f, int& b2p) {
    double w1f, w2f;
    int w1p, w2p;
    const int l1 = wave_argmin(b1f, b1p, w1f, w1p);
    const bool mine = (p_tid() & 63) == l1;
    wave_argmin(mine ? b2f : b1f, mine ? b2p : b1p, w2f, w2p);
    b1f = w1f, b1p = w1p, b2f = w2f, b2p = w2p;
}
BIOIK_DEV void top2_wave(double& b1f, int& b1p, double& b2f, int& b2p, int G) {
#if !defined(BIOIK_TOP2_BUTTERFLY)
    if (G >= 64) {
        top2_wave64_minima(b1f, b1p, b2f, b2p);
        return;
    }
#endif
#pragma unroll
    for (int m = 32; m >= 16; m >>= 1) {  // across the rows of 16 lanes: LDS-crossbar permutes
        if (m >= G) continue;
        double o1f = p_shfl_xor(b1f, m), o2f = p_shfl_xor(b2f, m);
        int o1p = p_shfl_xor(b1p, m), o2p = p_shfl_xor(b2p, m);
        top2_merge(b1f, b1p, b2f, b2p, o1f, o1p, o2f, o2p);
    }
    // inside a row: DPP moves (mirror of the row, mirror of the half row, then the two quad permutations) -- every lane has met
    // every other lane's pair after these four steps
#define TOP2_DPP_STEP(MOVE)                                          \
    {                                                                \
        const double o1f = MOVE(b1f), o2f = MOVE(b2f);               \
        const int o1p = MOVE(b1p), o2p = MOVE(b2p);                  \
        top2_merge(b1f, b1p, b2f, b2p, o1f, o1p, o2f, o2p);          \
    }
    TOP2_DPP_STEP(p_row_mirror<0>)
    TOP2_DPP_STEP(p_row_mirror<1>)
    TOP2_DPP_STEP(p_quad_xor<2>)
    TOP2_DPP_STEP(p_quad_xor<1>)
#undef TOP2_DPP_STEP
}
// Bitonic sort of G * E (fitness, child index) pairs held in REGISTERS, E per lane, ascending by (fitness, index) -- the order of a stable sort by
// fitness.  Sorted position p lives in lane p / E, register p % E.  The compare-exchange partners of the network's step (k, j) sit at positions that
// differ in bit j: for j < E that is another register of the same lane (no data moves), else the same register of lane ^ (j / E) (one ds_bpermute per
// dword: three per pair).  Against the same network over two arrays in LDS (four reads, up to four writes per exchange, a dependent write -> read
// between any two of its log2(n)(log2(n)+1)/2 rounds) the 512 children of the 31-joint chain take 21 rounds of 24 permutes instead of 45 rounds of 24
// reads and writes with their address arithmetic.  G: lanes of the group, a power of two <= 64 (a half or a whole wavefront).
template <int E>
BIOIK_DEV void sort_pairs_in_registers(double (&f)[E], int (&c)[E], int gtid, int G) {
    const int n = G * E;
    for (int k = 2; k <= n; k <<= 1) {
        const bool ascending = ((gtid * E) & k) == 0;  // (k >= E: the direction of a block is the same for all registers of a lane)
        for (int j = k >> 1; j >= E; j >>= 1) {
            const int d = j / E;  // (E: a power of two, known at compile time)
            const bool keep_low = ((gtid & d) == 0) == ascending;
#pragma unroll
            for (int i = 0; i < E; i++) {
                const double of = p_shfl_xor(f[i], d);
                const int oc = p_shfl_xor(c[i], d);
                const bool mine_after = (f[i] > of) || (f[i] == of && c[i] > oc);
                const bool take = mine_after == keep_low;
                f[i] = take ? of : f[i], c[i] = take ? oc : c[i];
            }
        }
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
            if (j >= k) continue;
#pragma unroll
            for (int i = 0; i < E; i++) {
                if (i & j) continue;
                const int i2 = i | j;
                const bool asc = k >= E ? ascending : ((i & k) == 0);
                const bool a_after_b = (f[i] > f[i2]) || (f[i] == f[i2] && c[i] > c[i2]);
                const bool swap = a_after_b == asc;
                const double fa = f[i], fb = f[i2];
                const int ca = c[i], cb = c[i2];
                f[i] = swap ? fb : fa, f[i2] = swap ? fa : fb, c[i] = swap ? cb : ca, c[i2] = swap ? ca : cb;
            }
        }
    }
}
// The same network over 64-bit KEYS: one unsigned compare per exchange and two dwords per element instead of three compares and three dwords.  The caller
// builds a key from a non-negative fitness and its child index (sort_key) -- unique, so every order is strict.
template <int E>
BIOIK_DEV void sort_keys_in_registers(unsigned long long (&key)[E], int gtid, int G) {
    const int n = G * E;
    for (int k = 2; k <= n; k <<= 1) {
        const bool ascending = ((gtid * E) & k) == 0;
        for (int j = k >> 1; j >= E; j >>= 1) {
            const int d = j / E;
            const bool keep_low = ((gtid & d) == 0) == ascending;
#pragma unroll
            for (int i = 0; i < E; i++) {
                const unsigned long long other = p_shfl_xor(key[i], d);
                key[i] = ((key[i] > other) == keep_low) ? other : key[i];
            }
        }
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
            if (j >= k) continue;
#pragma unroll
            for (int i = 0; i < E; i++) {
                if (i & j) continue;
                const int i2 = i | j;
                const bool asc = k >= E ? ascending : ((i & k) == 0);
                const unsigned long long a = key[i], b = key[i2];
                const bool swap = (a > b) == asc;
                key[i] = swap ? b : a, key[i2] = swap ? a : b;
            }
        }
    }
}
// the least key of a half-wavefront group, known to all of its lanes: one permute across the two rows of 16, then DPP moves inside a row (as top2_wave)
BIOIK_DEV unsigned long long half_min_u64(unsigned long long k) {
    unsigned long long o = p_shfl_xor(k, 16);
    k = o < k ? o : k;
    o = p_row_mirror<0>(k), k = o < k ? o : k;
    o = p_row_mirror<1>(k), k = o < k ? o : k;
    o = p_quad_xor<2>(k), k = o < k ? o : k;
    o = p_quad_xor<1>(k), k = o < k ? o : k;
    return k;
}
// the least key of a lane group of 32 or 64 lanes (half a wavefront or a whole one), known to all of its lanes
BIOIK_DEV unsigned long long group_min_u64(unsigned long long k, int G) {
    if (G >= 64) {
        const unsigned long long o = p_shfl_xor(k, 32);
        k = o < k ? o : k;
    }
    return half_min_u64(k);
}
// The survivors of the pre-selection WITHOUT the sort.  ik_evolution_2.cpp:366-378 scores every child on the secondary goals, sorts the children and keeps a
// prefix whose length was drawn before the sort: WHICH children survive is a question to the k-th order statistic of the keys, and their order among
// themselves reaches the result only where two of them tie on their whole fitness (the selection takes the earlier one; solve_body settles those ties
// where it finds them).  The threshold comes from a bisection of the VALUE range [least key, greatest key]: a round counts the keys <= mid -- a compare,
// a ballot and a population count per register -- and the search is over as soon as exactly k are, because mid then lies in the gap behind the k-th key.
// Keys spread like a generation's reach that after about log2(n) + 2 rounds (unique keys: 64 at most), against the log2(n)(log2(n)+1)/2 = 28 / 45 rounds
// of permutes and compare-exchanges of the network (128 / 512 keys).  Returns the threshold: the survivors are the keys <= it.
// n_mine: this lane's first n_mine keys belong to children (the rest is padding: +inf); gmask: the lanes of the caller's group inside its wavefront.
// found: false if 66 rounds have not produced the threshold -- with distinct keys and 1 <= k < the number of children the interval shrinks by at least one
// key value per round and that cannot happen; the bound is there so that no input, whatever it is, can keep a wavefront in the loop (the caller sorts then).
template <int E>
BIOIK_DEV unsigned long long select_threshold(const unsigned long long (&key)[E], int n_mine, int k, int G, unsigned long long gmask, bool& found) {
    unsigned long long lo = ~0ull, nhi = ~0ull;  // (nhi: the complement of the greatest key, so that both ends are minima)
#pragma unroll
    for (int i = 0; i < E; i++)
        if (i < n_mine) lo = key[i] < lo ? key[i] : lo, nhi = ~key[i] < nhi ? ~key[i] : nhi;
    lo = group_min_u64(lo, G);
    unsigned long long hi = ~group_min_u64(nhi, G);
    unsigned long long T = hi;
    bool done = false;
    for (int round = 0; round < 66; round++) {  // (the halves of a wavefront that carries two species search side by side: the loop ends when both have found theirs)
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < E; i++) cnt += p_popc64(p_ballot(key[i] <= mid) & gmask);
        if (!done) {
            if (cnt == k) T = mid, done = true;
            else if (cnt < k) lo = mid + 1ull;
            else hi = mid;
        }
        if (p_ballot(!done) == 0ull) break;
    }
    found = done;
    return T;
}
// A non-negative double orders like its bit pattern.  The key keeps the upper 54 bits of the pattern and carries the child index (< 1024) in the lower ten:
// keys order like (fitness, index) wherever two fitness values differ above their lowest ten mantissa bits or not at all.  Pairs that differ ONLY there
// (relative difference below 2.3e-13) are ordered by index, possibly wrongly -- the caller finds them among the sorted neighbours and sorts exactly then.
// (drop: the low bits of the pattern the key gives up, ten in the product; the parity suites raise it so that the exact path is taken often)
BIOIK_DEV unsigned long long sort_key(double f, int index, int drop) {
    unsigned long long b;
    __builtin_memcpy(&b, &f, 8);
    return (b & ~((1ull << drop) - 1ull)) | (unsigned long long)index;
}
// rendezvous of one lane group: a single wavefront needs no s_barrier (p_wave_sync), several wavefronts take the workgroup
// barrier -- every group of the workgroup then executes the same number of them
BIOIK_DEV void group_sync(int G) {
    if (G > 64) p_barrier(); else p_wave_sync();
}
BIOIK_DEV void top2_xwave(double& b1f, int& b1p, double& b2f, int& b2p, double* s_red, int gtid, int G) {
    const int nw = G >> 6;  // wavefronts of this species group (s_red is the group's own scratch)
    if (G > 64) {
        if ((gtid & 63) == 0) {
            double* d = s_red + 4 * (gtid >> 6);
            d[0] = b1f, d[1] = (double)b1p, d[2] = b2f, d[3] = (double)b2p;
        }
        p_barrier();
        b1f = b2f = P_INF;
        b1p = b2p = 0x7fffffff;
        for (int w = 0; w < nw; w++) {
            const double* d = s_red + 4 * w;
            for (int i = 0; i < 2; i++) {
                double of = d[2 * i];
                int op = (int)d[2 * i + 1];
                top2_insert(b1f, b1p, b2f, b2p, of, op);
            }
        }
        p_barrier();
    }
}

// RobotFK::applyConfiguration + initializeMutationApproximator at the (workgroup-shared) individual x:
// the joint frames are published to LDS by lane 0 (the per-joint frame chain), then lanes fan out over (tip, op).
// (gtid, G): index and size of the cooperating lane group (the whole workgroup, or one species group of it)
// COOP != 0: the lanes of the group walk ONE individual x (the solver); 0: every lane its own or no full wavefront (the function-level kernels)
template <int COOP = 0, class PB>
BIOIK_NOINLINE void build_approximator(PB pb, XV x, double* slots, double* s_frames, double* s_tips, double* s_delta, double* s_base, int gtid, int G,
                                       const double* prefix = nullptr) {
    const int n_ops = pb->n_ops, T = pb->T;
    if (gtid < 64) {  // one wavefront walks the chain (lane 0 publishes), its lanes sharing the joints' trigonometry; the others wait at the barrier
        auto publish = [&](int t, const F7& f) {
            if (gtid == 0) f7_store(s_tips + t * 7, f);
        };
        if (COOP == 0) fk_walk(pb, x, slots, gtid == 0 ? s_frames : nullptr, publish, prefix);
        else if (G >= 64) fk_walk<64>(pb, x, slots, gtid == 0 ? s_frames : nullptr, publish, prefix);
        else fk_walk<32>(pb, x, slots, gtid == 0 ? s_frames : nullptr, publish, prefix);  // half-wavefront groups: one individual per half
    }
    for (int k = gtid; k < n_ops; k += G) s_base[k] = x(k);
    group_sync(G);
    for (int t = 0; t < T; t++)  // (no idx / n_ops: the reciprocal of a division by a uniform would be set up in front of the step loop and kept)
        for (int k = gtid; k < n_ops; k += G) {
            double o[7];
            approximator_entry(pb, t, k, s_frames, s_tips, o, s_base, prefix);
            double* d = s_delta + ((size_t)t * n_ops + k) * 7;
            for (int c = 0; c < 7; c++) d[c] = o[c];
        }
    group_sync(G);
}

// best island per query (ik_parallel.h:220-269); one lane per query
struct SelectArgs {
    int islands, V;
    int sync, pad;  // sync: only the islands that passed after the LEAST number of steps are candidates (bioik_solve_params::island_sync)
    uint64_t n;
    const double* isl_solutions;
    const double* isl_fitness;
    const int32_t* isl_success;
    const int32_t* isl_steps;
    double* solutions;
    double* fitness;
    int32_t* success;
    int32_t* steps;
};
// (the islands' results were written by the launches in front of this one: device-scope loads, p_load_device)
BIOIK_DEV void select_body(const SelectArgs& a, uint64_t q) {
    if (q >= a.n) return;
    int best = 0;
    double best_fit = BIOIK_DBL_MAX;
    int least_steps = 0x7fffffff;
    if (a.sync)
        for (int i = 0; i < a.islands; i++) {
            uint64_t u = q * (uint64_t)a.islands + i;
            if (p_load_device(a.isl_success + u) && p_load_device(a.isl_steps + u) < least_steps) least_steps = p_load_device(a.isl_steps + u);
        }
    for (int i = 0; i < a.islands; i++) {
        uint64_t u = q * (uint64_t)a.islands + i;
        if (p_load_device(a.isl_success + u) && (!a.sync || p_load_device(a.isl_steps + u) == least_steps) && p_load_device(a.isl_fitness + u) < best_fit) best_fit = p_load_device(a.isl_fitness + u), best = i;
    }
    if (best_fit == BIOIK_DBL_MAX) {
        for (int i = 0; i < a.islands; i++) {
            uint64_t u = q * (uint64_t)a.islands + i;
            if (p_load_device(a.isl_fitness + u) < best_fit) best_fit = p_load_device(a.isl_fitness + u), best = i;
        }
    }
    uint64_t u = q * (uint64_t)a.islands + best;
    for (int v = 0; v < a.V; v++) a.solutions[q * a.V + v] = p_load_device(a.isl_solutions + u * a.V + v);
    a.fitness[q] = best_fit;
    a.success[q] = p_load_device(a.isl_success + u);
    a.steps[q] = p_load_device(a.isl_steps + u);
}

// select_body by ONE WAVEFRONT for one query (all 64 lanes call it): lane i reads the islands i, i + 64, ... and three butterflies pick what the loop above
// picks -- the least step count of a passing island, then the least fitness among the islands that count, the LOWEST island among equal values (the loop's
// strict comparison keeps the first), else the least fitness of all -- and the lanes copy the winner's solution.  A single lane walks 64 islands in ~22 us
// (dependent device-scope loads: k_select on one query with 64 islands, profiles/r06_select_by_wavefront.log), the wavefront in ~3: what MoveIt's one pose per call
// pays once per call.
BIOIK_DEV void select_coop(const SelectArgs& a, uint64_t q, int lane) {
    const uint64_t u0 = q * (uint64_t)a.islands;
    int least = 0x7fffffff;
    if (a.sync) {
        for (int i = lane; i < a.islands; i += 64) {
            const int st = p_load_device(a.isl_steps + u0 + i);
            if (p_load_device(a.isl_success + u0 + i) && st < least) least = st;
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = p_shfl_xor(least, m);
            least = o < least ? o : least;
        }
    }
    double bf = BIOIK_DBL_MAX, af = BIOIK_DBL_MAX;  // the least fitness among the islands that count / among all, of this lane's islands
    int bi = 0x7fffffff, ai = 0x7fffffff;
    for (int i = lane; i < a.islands; i += 64) {
        const double f = p_load_device(a.isl_fitness + u0 + i);
        const bool counts = p_load_device(a.isl_success + u0 + i) && (!a.sync || p_load_device(a.isl_steps + u0 + i) == least);
        if (counts && f < bf) bf = f, bi = i;
        if (f < af) af = f, ai = i;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const double obf = p_shfl_xor(bf, m), oaf = p_shfl_xor(af, m);
        const int obi = p_shfl_xor(bi, m), oai = p_shfl_xor(ai, m);
        if (obf < bf || (obf == bf && obi < bi)) bf = obf, bi = obi;
        if (oaf < af || (oaf == af && oai < ai)) af = oaf, ai = oai;
    }
    double best_fit = bf;
    int best = bi;
    if (!(bf < BIOIK_DBL_MAX)) best_fit = af, best = ai;               // no island counts (or none with a fitness below DBL_MAX): the least fitness of all
    if (!(best_fit < BIOIK_DBL_MAX)) best_fit = BIOIK_DBL_MAX, best = 0;  // ... and if there is none either, the first island
    const uint64_t u = u0 + (uint64_t)best;
    for (int v = lane; v < a.V; v += 64) a.solutions[q * a.V + v] = p_load_device(a.isl_solutions + u * a.V + v);
    if (lane == 0) {
        a.fitness[q] = best_fit;
        a.success[q] = p_load_device(a.isl_success + u);
        a.steps[q] = p_load_device(a.isl_steps + u);
    }
}

struct SolveArgs {
    ProbPtr pb;
    DevSolveParams sp;
    const double* seeds;    // [n][V]
    const double* params;   // [n][P]
    double* solutions;      // [n*islands][V]
    double* fitness;        // [n*islands]   ranking fitness of ik_parallel.h:229-246
    int32_t* success;       // [n*islands]
    int32_t* steps;         // [n*islands]
    unsigned long long* phase_cycles;  // [n*islands][8] or null: per-phase shader cycles (builds with -DBIOIK_PHASE_TIMING)
    unsigned long long* launch_clock;  // captured calls only: one zeroed word per launch (timeout_ticks != 0), the first workgroup's start on the device clock
    unsigned long long deadline = 0ull;  // eager calls (timeout_ticks != 0, launch_clock null): the call's deadline on the device clock, counted from its SUBMISSION
    // A solve in two launches (the launcher's choice, bioik_hip.hip): the first runs the steps [0, step_end) of every unit under the lane
    // mapping that fills the chip best and hands the units that are neither solved nor out of time to the second, which runs them to the
    // end under the mapping with the fastest lone step.  What a unit is between two steps: the species' elites, the solution and
    // 24 numbers of bookkeeping (three LDS arrays); the RNG is a function of the step index.
    int32_t step_begin = 0, step_end = 0x7fffffff;  // this launch runs the steps [step_begin, min(step_end, sp.max_steps))
    double* carry = nullptr;                  // [units][9 M + 24] state of the handed-over units (first launch writes, second reads)
    int32_t* carry_list = nullptr;            // first launch: the units handed over, in the order they finish ...
    unsigned int* carry_count = nullptr;      // ... and how many (zeroed by the host)
    const int32_t* unit_list = nullptr;       // second launch: workgroup b continues unit_list[b] ...
    const unsigned int* unit_count = nullptr; // ... for b < *unit_count (the grid is an upper bound)
    // sp.island_sync ("any island succeeds => all stop", ik_parallel.h:102, 160-178): one word per query, the least number of steps after which
    // an island of the query has passed the success test (the host fills it with 0xffffffff).  An island that passes files its step count (atomic
    // minimum); an island that finds a count <= its own leaves, because k_select only considers the islands that passed at the least count.
    unsigned int* first_success = nullptr;    // [n]
    // One launch for a call with islands (round 6; calls that cannot fill the chip: MoveIt's one pose per call): an island that has written its result counts itself
    // in a word of its query, and the island that completes the count picks the query's best island itself (select_body: what k_select does in a launch of its own)
    // and puts the query's two words back -- the count to 0, first_success to 0xffffffff --, so that the next call finds them as it needs them and no launch has
    // to set them up.  Null: the islands' results are reduced by k_select (solves in several launches, the gradient family).
    unsigned int* island_done = nullptr;      // [n]
    double* final_solutions = nullptr;        // the caller's arrays: [n][V], [n], [n], [n]
    double* final_fitness = nullptr;
    int32_t* final_success = nullptr;
    int32_t* final_steps = nullptr;
    // Hand-over when the chip runs empty (the throughput schedule's straggler tail): the workgroups of every launch that carries this word count
    // their wavefronts in it while they run (the launches that take the stragglers over too: their work keeps the chip busy just as well).  New workgroups start as fast as old ones leave while any launch has work queued, so a count below `drain_below`
    // means the queue is empty and the chip is emptying: a unit that has run `drain_min_steps` steps then leaves for the next launch (the mapping with
    // the faster lone step) whatever step it is at, its step count travelling with its state; that launch has step_begin < 0 and reads it there.
    // Which units leave when depends on timing; their results do not (every mapping computes the same trajectory).
    int32_t sort_key_drop = 10;                    // the pre-selection's sort keys give up this many low bits of a fitness for the child index (sort_key)
    int32_t preselect = 1;                         // bit 0: the pre-selection's survivors by selection (select_threshold) instead of the sort; bits 8...: parity suites -- the parked fitness values lose that many low bits, so that children tie
    // A rendezvous between wavefronts through words in LDS (the helped kernel) that gives up -- its partner did not answer within 2^22 polls: a debugger, a
    // context switch, a wavefront that died -- sets this word of the handle (page-locked host memory, mapped): the host turns it into BIOIK_ERR_HIP for the call.
    // The reference's boost::barrier cannot time out (ik_parallel.h:64-67); a solve that went on unsynchronised must not pass for a result.
    unsigned int* error = nullptr;
    int32_t debug_flags = 0;                       // tests: bit 0 -- the helper wavefront of species 0 never answers (the host simulator's rendezvous test)
    unsigned int* resident = nullptr;              // [16][32]: word 32 x of XCD x
    int32_t drain_below = 0, drain_min_steps = 0;  // (wavefronts per XCD)  // (drain_below < 0: test pattern -- unit u leaves after 1 + hash(u) % -drain_below steps)
};

struct SpeciesState {
    double fit;       // Species::fitness (exact FK, primary)
    double pf0, pf1;  // fitness of the two elites under the evaluation currently in use
    int id;           // persistent id (RNG stream)
    int slot;         // LDS slot of the species' elites
    int cur;          // which of the two LDS buffers holds the elites
    int improved;
    int ok;           // success test of the first elite (it is what becomes the solution when the species leads)
};

// The lane numbers of one phase of solve_body and everything derived from them -- the species group's scratch pointers, the lane's
// genotype column -- declared from FRESH copies of the lane numbers (p_fresh, bioik_platform.h): a phase that opens a scope with this
// computes its LDS addresses where it uses them instead of inheriting them from in front of the step loop, where the compiler had
// parked them in scratch memory (r02: 36 spilled VGPRs, every one of them an address of this kind).
#define BIOIK_LANE_SCOPE                                                                                                            \
    const int tid = HALVES ? p_lane_fresh() : (SLIM ? p_tid_fresh() : p_fresh(tid0)); /* (the lane number, from a copy kept across the phases or (SLIM) computed afresh -- HALVES: the workgroup is one wavefront; group and index inside it follow from it) */ \
    const int grp = g_shift >= 0 ? tid >> g_shift : tid / G, gtid = tid - grp * G;                                                  \
    double* const gbase = lds + L.g_first + grp * L.g_stride; /* this group's scratch */                                            \
    double* const s_xn = gbase + L.xn;                                                                                              \
    double* const s_gv = gbase + L.gv;                                                                                              \
    double* const s_frames = gbase + L.frames;                                                                                      \
    double* const s_tips = gbase + L.tips;                                                                                          \
    double* const s_delta = gbase + L.delta;                                                                                        \
    double* const s_base = gbase + L.base;                                                                                          \
    double* const s_grad = gbase + L.grad;                                                                                          \
    double* const s_xm = gbase + L.xm;                                                                                              \
    double* const s_xp = gbase + L.xp;                                                                                              \
    double* const s_dv = gbase + L.dv;                                                                                              \
    double* const s_fc = gbase + L.fc;                                                                                              \
    double* const s_red = gbase + L.red;                                                                                            \
    double* const s_sec = gbase + L.sec;                                                                                            \
    int32_t* const s_order = (int32_t*)(gbase + L.order);                                                                           \
    double* const s_bc = gbase + L.bc; /* values broadcast from the group's leading wavefront */                                    \
    double* const xcol = lds + L.xcol + tid; /* this lane's genotype column(s): [col][op][lane], stride nth */                      \
    const XV xl{xcol, nth};                                                                                                         \
    const LinModel lm{s_tips, s_delta, s_base};                                                                                     \
    const bool glead = gtid < 64, wlead = tid < 64
#define BIOIK_EPILOGUE_SCOPE_BEGIN { BIOIK_LANE_SCOPE;
#define BIOIK_EPILOGUE_SCOPE_END }

// ---------------------------------------------------------------------------------------------------------
// solve_body in parts (round 6).  What the phases of a step share -- the launch's constants, the lane mapping, the LDS pointers of the unit, the rendezvous of the
// workgroup, the step loop's state -- is a SolveFrame; a phase is a function over it (BIOIK_FRAME_NAMES brings its members in under the names the code has always
// used).  Everything is inlined into the one kernel body, as before; the parts exist for the reader and the reviewer.
//
// LEAN: the flavour without floating / planar joints (see pb_flavour); the launcher picks it whenever the problem allows.
// CL: children computed where they are read (no genotype columns, sp.columnless) — a kernel of its own, so that the accessor-templated
// copies of the chain walk do not weigh on the register allocation of the column kernels (with both in one kernel the lean flavour
// spilled 74 instead of 32 VGPRs and lost 4 % on C2)
// JOINT: both species of a query on the halves of ONE wavefront (64 lanes) and a problem with secondary goals -- every species then walks a random
// prefix of its pre-selected children (ik_evolution_2.cpp:366-378), and with one half per species the wavefront waits for the longer of
// the two prefixes (2/3 of the children on average, against 1/2).  In this instantiation the 64 lanes walk the children of BOTH species as
// one list, and each species' two best are found by a reduction inside its half.
// SLIM: the instantiation for the 128-register budget (four wavefronts per SIMD): the species record is read from LDS where a generation begins
// and filed where it ends, so that nothing of it lives in registers -- or, under that budget, in scratch memory -- across the chain walks (the
// record's reads are then LDS reads; a species group is one wavefront or half of one in the mappings that run under this budget, so the
// hand-over is a wavefront-level rendezvous, not a barrier)
// FIXED: what the launcher guarantees for the kernels under the smaller register budgets is known at compile time (the runtime switches of the other
// instantiations cost them registers and code):  1 = 64 lanes, the species on the halves of ONE wavefront, exact FK, children computed where they are
// read and walked in pairs, no secondary goal (k_solve_lean_cl64w4);  2 = 128 lanes, a wavefront per species, exact FK, computed children in pairs,
// secondary goals allowed (k_solve_lean_cl4);  3 = 64 lanes, the species on the halves of one wavefront, LINEARISED phenotypes, one computed child per
// lane and trip (k_solve_lean_lin: populations of up to 32 children per species -- the reference's own parameters);  4 = 64 lanes, halves, exact FK,
// secondary goals, the pre-selected children of both species walked as one list (JOINT; k_solve_lean_clj4);  5 = FIXED 2 with two helper wavefronts (k_solve_lean_cl4h)
// ---------------------------------------------------------------------------------------------------------
template <bool LEAN_, bool CL_, bool JOINT_, bool SLIM_, int FIXED_>
struct SolveFrame {
    static constexpr bool LEAN = LEAN_, CL = CL_, JOINT = JOINT_, SLIM = SLIM_;
    static constexpr int FIXED = FIXED_;
    static constexpr bool DENSE = FIXED == 1, HELPED = FIXED == 5, WAVE2 = FIXED == 2 || HELPED, LIN = FIXED == 3, JH = FIXED == 4, HALVES = DENSE || LIN || JH;
    // The pre-selection's survivors by selection instead of the sort (select_threshold) where a lane holds eight keys -- a wavefront per species, C4's 512 children:
    // 45 rounds of the network against ~14 of the bisection, +5 % on the 31-joint chain.  With four keys per lane on half a wavefront (the joint walk, C3's 128
    // children) the 28-round network is the cheaper of the two (measured: -2.5 % with the selection, profiles/r05_preselect_by_selection.log), so those sort.
    static constexpr bool SELECTS = WAVE2;
    static_assert((FIXED == 4) == JOINT, "the joint walk of both species' children exists as FIXED = 4 only (64 lanes, halves, exact FK, secondary goals: k_solve_lean_clj4)");
    static_assert(FIXED == 0 || (SLIM && CL), "the fixed mappings are builds of the computed-children kernel for the 128-register budget");
    static_assert(LEAN || !CL, "computed children: lean flavour only (quaternion genes are renormalised in place)");
    static_assert(CL || !JOINT, "the joint walk of both species' children exists for computed children only");
    typedef typename std::conditional<LEAN, LeanProbPtr, ProbPtr>::type PB;
    const SolveArgs& a;
    const DevSolveParams& sp;
    double* const lds;
    PB pb;
    uint64_t unit, q;
    uint32_t island, key;
    bool resume;
    int tid0, nth, V, P, T, n_ops, D, lambda, n_cols, groups, G, g_shift;
    uint64_t active_mask;
    bool has_sec, exact, child_pairs;
    LdsLayout L;
    double *s_seed, *s_par, *s_pop, *s_sol, *s_prefix, *s_state, *s_slots, *s_clip;
    // the rendezvous of the workgroup: the hardware barrier, or (helped kernel: its helpers are elsewhere) a count per main wavefront in LDS that the other one waits for
    unsigned int bar_count = 0u, gen_count = 0u;  // (gen_count: generations this main wavefront has published to its helper)
    // (helped kernel) a wait of this wavefront has given up: its partner -- the other main wavefront, or its helper -- did not answer.  The handle's error word is
    // set (the host reports BIOIK_ERR_HIP for the call), this wavefront waits for nobody any more (it keeps raising its own counts, so that a partner that is
    // merely late does not wait for IT) and leaves the step loop at the end of the step; whatever it computes from here on is not a result.
    bool lost = false;
    // the step loop's state
    int step_first = 0, step_end = 0, steps = 0;
    bool success = false, expired = false, overtaken_out = false, drained = false;
    double final_fit = BIOIK_DBL_MAX;
#if defined(BIOIK_PHASE_TIMING)
    unsigned long long ph_t_[PHASE_N], ph_last_, ph_start_;
#endif
    BIOIK_DEV SolveFrame(const SolveArgs& args, double* lds_base) : a(args), sp(args.sp), lds(lds_base) {}
    // (counted in wavefronts: the launches that share the words differ in theirs.  One word per XCD, 128 bytes apart, each touched by the workgroups of
    // ITS XCD only: a word all eight L2s fight over cost 15 % of a stream's throughput, profiles/r04_drain_handover.log)
    BIOIK_DEV unsigned int* my_resident() const { return a.resident + 32 * p_xcc_id(); }  // (computed where it is used: no register carries it through the kernel)
    BIOIK_DEV void rendezvous_lost() {
        lost = true;
        if (a.error) p_store_device(a.error, 1u);
    }
    BIOIK_DEV void wg_barrier() {
        if constexpr (HELPED) {
            unsigned int* const hw = (unsigned int*)(lds + L.help);
            const int w = p_wave_index();
            p_wave_sync();
            bar_count++;
            p_flag_store(hw + w, bar_count);
            if (!lost && p_flag_wait_ge(hw + (w ^ 1), bar_count) == 0xffffffffu) rendezvous_lost();
        } else {
            p_barrier();
        }
    }
    // Values every lane needs but one wavefront can compute (fitness of an elite, of the solution ...): the leading
    // wavefront of the species group / of the workgroup evaluates and publishes through LDS; the other wavefronts sleep
    // at the barrier instead of spending issue slots of their SIMDs on identical copies.
    // (the lane numbers and the group's broadcast slot are arguments: the caller's phase passes its own, see BIOIK_LANE_SCOPE)
    template <class Fn>
    BIOIK_DEV double group_value(bool glead, int gtid, double* s_bc, Fn&& fn) {
        double v = 0.0;
        if (glead) v = fn();
        if (G > 64) {
            if (gtid == 0) s_bc[0] = v;
            wg_barrier();
            v = s_bc[0];
            wg_barrier();
        }
        return v;
    }
    BIOIK_DEV FitCheck group_check(bool glead, int gtid, double* s_bc, const XV& x) {  // exact fitness + success test of a group's vector, known to the whole group
        const QueryCtx qc{s_seed, s_par};
        FitCheck fc{0.0, 0};
        if (glead) {  // (the group's leading wavefront, or its half of the wavefront: the lanes share the walk of x)
            if (BIOIK_COOP_WALKS == 0) fc = exact_fitness_check(pb, x, qc, s_slots, sp.dpos, sp.drot, sp.dtwist, 1, s_prefix);
            else if (G >= 64) fc = exact_fitness_check<64>(pb, x, qc, s_slots, sp.dpos, sp.drot, sp.dtwist, 1, s_prefix);
            else fc = exact_fitness_check<32>(pb, x, qc, s_slots, sp.dpos, sp.drot, sp.dtwist, 1, s_prefix);
        }
        if (G > 64) {
            if (gtid == 0) s_bc[0] = fc.fitness, s_bc[2] = (double)fc.ok;
            wg_barrier();
            fc.fitness = s_bc[0];
            fc.ok = (int)s_bc[2];
            wg_barrier();
        }
        return fc;
    }
    BIOIK_DEV FitCheck wg_check(bool wlead, int tid, const XV& x, double dpos, double drot, double dtwist, int do_check) {
        const QueryCtx qc{s_seed, s_par};
        FitCheck fc{0.0, 0};
        if (wlead) fc = exact_fitness_check<BIOIK_COOP_WALKS ? 64 : 0>(pb, x, qc, s_slots, dpos, drot, dtwist, do_check, s_prefix);  // (a vector of the whole workgroup)
        if (nth > 64) {
            double* const s_wbc = s_state + 16;  // (the workgroup's broadcast slots)
            if (tid == 0) s_wbc[0] = fc.fitness, s_wbc[1] = (double)fc.ok;
            wg_barrier();
            fc.fitness = s_wbc[0];
            fc.ok = (int)s_wbc[1];
            wg_barrier();
        }
        return fc;
    }
    // The bookkeeping of the two species lives in LDS between the phases of a step (s_state[rank][8], rank 0 = the leading species of
    // the last ranking): it is a handful of numbers read a few times per step, and as per-lane registers it was what the register
    // allocator spilled to scratch around every step.
    BIOIK_DEV SpeciesState species_load(int r) const {
        const double* d = s_state + r * 8;
        return SpeciesState{d[0], d[1], d[2], (int)d[3], (int)d[4], (int)d[5], (int)d[6], (int)d[7]};
    }
    BIOIK_DEV void species_store(int r, const SpeciesState& S) const {
        double* d = s_state + r * 8;
        d[0] = S.fit, d[1] = S.pf0, d[2] = S.pf1, d[3] = (double)S.id, d[4] = (double)S.slot, d[5] = (double)S.cur, d[6] = (double)S.improved, d[7] = (double)S.ok;
    }
    // the rank (0 / 1: the order of the last ranking) of the species this lane works for.  (DENSE: a half-wavefront runs the species of its own number, read off the
    // lane number wherever it is needed: no register carries it)
    BIOIK_DEV int rank_now(int rank_it) const {
        return (SLIM && groups == 2) ? (HALVES ? p_lane_fresh() >> 5 : (WAVE2 ? p_wave_index() : (g_shift >= 0 ? p_tid_fresh() >> g_shift : p_tid_fresh() / G))) : rank_it;
    }
};

// the members of a SolveFrame under the names the phases use (copies of scalars and pointers: the optimiser sees through them)
#if defined(BIOIK_PHASE_TIMING)
#define BIOIK_FRAME_PHASE_NAMES(F) auto& ph_t_ = (F).ph_t_; auto& ph_last_ = (F).ph_last_; auto& ph_start_ = (F).ph_start_;
#else
#define BIOIK_FRAME_PHASE_NAMES(F)
#endif
#define BIOIK_FRAME_NAMES(F)                                                                                                                                          \
    typedef typename Frame::PB PB;                                                                                                                                    \
    constexpr bool LEAN = Frame::LEAN, CL = Frame::CL, JOINT = Frame::JOINT, SLIM = Frame::SLIM, DENSE = Frame::DENSE, HELPED = Frame::HELPED, WAVE2 = Frame::WAVE2,   \
                   LIN = Frame::LIN, JH = Frame::JH, HALVES = Frame::HALVES, SELECTS = Frame::SELECTS, columnless = Frame::CL;                                        \
    constexpr int FIXED = Frame::FIXED;                                                                                                                               \
    const SolveArgs& a = (F).a;                                                                                                                                       \
    const DevSolveParams& sp = (F).sp;                                                                                                                                \
    double* const lds = (F).lds;                                                                                                                                      \
    const PB pb = (F).pb;                                                                                                                                             \
    const uint64_t unit = (F).unit, q = (F).q, active_mask = (F).active_mask;                                                                                         \
    const uint32_t key = (F).key;                                                                                                                                     \
    const bool resume = (F).resume, has_sec = (F).has_sec, exact = (F).exact, child_pairs = (F).child_pairs;                                                          \
    const int tid0 = (F).tid0, nth = (F).nth, V = (F).V, P = (F).P, T = (F).T, n_ops = (F).n_ops, D = (F).D, lambda = (F).lambda,                                     \
              n_cols = (F).n_cols, groups = (F).groups, G = (F).G, g_shift = (F).g_shift;                                                                             \
    const int M = n_ops > 0 ? n_ops : 1;                                                                                                                              \
    const int SP = 2 * 2 * 2 * M; /* doubles per species in s_pop */                                                                                                  \
    const int BF = 4 * M;         /* doubles per buffer: [ind0 genes][ind0 momentum][ind1 genes][ind1 momentum] */                                                    \
    /* doubles of a unit's state between two steps: per species (in ranking order) the elite buffer in use -- two individuals, genes and momentum --, the solution, */  \
    /* the bookkeeping block (the other elite buffer is written before it is read: it does not travel) */                                                             \
    const int carry_n = 2 * BF + M + 24;                                                                                                                              \
    const LdsLayout& L = (F).L;                                                                                                                                       \
    double* const s_seed = (F).s_seed;                                                                                                                                \
    double* const s_par = (F).s_par;                                                                                                                                  \
    double* const s_pop = (F).s_pop;                                                                                                                                  \
    double* const s_sol = (F).s_sol;                                                                                                                                  \
    double* const s_prefix = (F).s_prefix;                                                                                                                            \
    double* const s_state = (F).s_state;                                                                                                                              \
    double* const s_slots = (F).s_slots;                                                                                                                              \
    double* const s_clip = (F).s_clip;                                                                                                                                \
    double* const s_wbc = s_state + 16;      /* broadcast slots of the workgroup */                                                                                    \
    double* const s_solst = s_state + 20;    /* [0] fitness, [1] success flag of the current solution */                                                               \
    double* const s_deadline = s_state + 22; /* the deadline on the device clock as two exact halves (the slots are doubles); only lane 0 reads it */                  \
    const QueryCtx qc{s_seed, s_par};                                                                                                                                 \
    auto wg_barrier = [&]() { (F).wg_barrier(); };                                                                                                                    \
    auto species_load = [&](int r_) { return (F).species_load(r_); };                                                                                                 \
    auto species_store = [&](int r_, const SpeciesState& S_) { (F).species_store(r_, S_); };                                                                          \
    auto my_resident = [&]() { return (F).my_resident(); };                                                                                                           \
    (void)wg_barrier, (void)species_load, (void)species_store, (void)my_resident; /* (not every phase uses every one) */                                              \
    BIOIK_FRAME_PHASE_NAMES(F)
// the launch's constants, the lane mapping and the unit's LDS block; seed and goal parameters staged (the workgroup's first barrier).  false: nothing to do for this
// workgroup (a launch that continues handed-over units has a grid as large as the list can get)
template <class Frame>
BIOIK_DEV bool solve_setup(Frame& F, uint64_t unit_in) {
    typedef typename Frame::PB PB;
    constexpr bool LEAN = Frame::LEAN, CL = Frame::CL, SLIM = Frame::SLIM, DENSE = Frame::DENSE, HELPED = Frame::HELPED, WAVE2 = Frame::WAVE2, LIN = Frame::LIN, JH = Frame::JH, HALVES = Frame::HALVES;
    constexpr bool columnless = CL;
    constexpr int FIXED = Frame::FIXED;
    const SolveArgs& a = F.a;
    const DevSolveParams& sp = F.sp;
    double* const lds = F.lds;
    uint64_t unit = unit_in;
    const bool resume = a.unit_list != nullptr;
    if (resume) {  // (uniform over the workgroup, before the first barrier)
        if (unit >= (uint64_t)p_atomic_load(a.unit_count)) return false;
        unit = (uint64_t)p_load_device(a.unit_list + unit);  // (device-scope loads of everything the last launch left: bioik_platform.h, p_load_device)
    }
    const PB pb = (PB)a.pb;
    const int tid0 = p_tid(), nth = HALVES ? 64 : (WAVE2 ? 128 : p_nthreads());
    if (a.resident && tid0 == 0) p_atomic_add(F.my_resident(), (unsigned int)(HELPED ? 4 : nth >> 6));
    const int V = pb->V, P = pb->P, T = pb->T, n_ops = pb->n_ops, D = pb->D;
    const int lambda = sp.lambda;
    const uint64_t active_mask = pb->active_mask;  // bit k: op k is a gene
    const bool has_sec = DENSE ? false : (JH ? true : pb->n_secondary > 0);
    const bool exact = LIN ? false : (FIXED ? true : sp.fk_mode == FK_EXACT);
    const bool child_pairs = LIN ? false : (FIXED ? true : sp.child_pairs != 0);
    const int n_cols = sp.child_cols > 0 ? sp.child_cols : 1;
    // The two species of bio2 only meet in the species management at the end of a step, so with >= 2 wavefronts the
    // workgroup splits into two lane groups that run one species each, concurrently (on different SIMDs of the CU).
    const int groups = FIXED ? 2 : (sp.species_parallel ? 2 : 1);
    const int G = HALVES ? 32 : (WAVE2 ? 64 : nth / groups);        // lanes per species group (a multiple of 64, or half a wavefront)
    const int g_shift = HALVES ? 5 : (WAVE2 ? 6 : ((G & (G - 1)) == 0 ? 31 - __builtin_clz((unsigned)G) : -1));  // the group sizes the launcher produces are powers of two: no integer division
    const LdsLayout L = make_layout(n_ops, V, P, T, pb->n_slots, nth, lambda, has_sec ? (exact ? 2 : 1) : 0, columnless ? 0 : n_cols, groups, child_pairs ? 2 : 1, (CL && exact) ? 1 : 0, 1, HELPED ? 1 : 0);
    double* s_seed = lds + L.seed;
    double* s_par = lds + L.par;
    double* s_pop = lds + L.pop;
    double* s_sol = lds + L.sol;
    double* s_prefix = lds + L.prefix;
    double* s_state = lds + L.state;
    double* s_slots = lds + L.slots;
    double* s_clip = lds + L.clip;
    const int M = n_ops > 0 ? n_ops : 1;
    BIOIK_LANE_SCOPE;  // (the initialisation's own; every phase of the step loop opens a new one)
    const int SP = 2 * 2 * 2 * M;  // doubles per species in s_pop
    const int BF = 4 * M;          // doubles per buffer: [ind0 genes][ind0 momentum][ind1 genes][ind1 momentum]

#if defined(BIOIK_PHASE_TIMING)
    for (int i_ = 0; i_ < PHASE_N; i_++) F.ph_t_[i_] = 0ull;
    F.ph_last_ = __builtin_readcyclecounter(), F.ph_start_ = wall_clock64();
#endif
    const uint64_t q = unit / (uint64_t)sp.islands;
    const uint32_t island = (uint32_t)(unit % (uint64_t)sp.islands);
    const bool is_helper = HELPED && tid0 >= 128;  // (the helped kernel's wavefronts 2 and 3: they walk half of a generation's children for wavefronts 0 and 1 and do nothing else)
    if (!is_helper) {
        for (int i = tid; i < V; i += nth) s_seed[i] = a.seeds[q * V + i];
        for (int i = tid; i < P; i += nth) s_par[i] = a.params[q * P + i];
        if constexpr (HELPED)
            if (tid < 16) ((unsigned int*)(lds + L.help))[tid] = 0u;
    }
    p_barrier();  // (the ONE hardware barrier of the helped kernel: its helper wavefronts never reach another, so from here on its two main wavefronts meet at wg_barrier's words)
    const QueryCtx qc{s_seed, s_par};
    const uint32_t key = rng_query_key(sp.random_seed, sp.first_query + q, island);
    F.pb = pb, F.unit = unit, F.q = q, F.island = island, F.key = key, F.resume = resume;
    F.tid0 = tid0, F.nth = nth, F.V = V, F.P = P, F.T = T, F.n_ops = n_ops, F.D = D, F.lambda = lambda, F.n_cols = n_cols, F.groups = groups, F.G = G, F.g_shift = g_shift;
    F.active_mask = active_mask, F.has_sec = has_sec, F.exact = exact, F.child_pairs = child_pairs;
    F.L = L;
    F.s_seed = s_seed, F.s_par = s_par, F.s_pop = s_pop, F.s_sol = s_sol, F.s_prefix = s_prefix, F.s_state = s_state, F.s_slots = s_slots, F.s_clip = s_clip;
    (void)SLIM, (void)DENSE, (void)WAVE2, (void)LIN, (void)JH, (void)LEAN, (void)qc;
    return true;
}
// A helper wavefront of the helped kernel (FIXED = 5: wavefronts 2 and 3): for the species of main wavefront `hs`, generation after generation, the children
// 64 ... 127 (+ 128 j) of what that wavefront published, into the species' parked fitness values; it reaches no barrier behind the kernel's first
template <class Frame>
BIOIK_DEV void solve_helper(Frame& F) {
    BIOIK_FRAME_NAMES(F);
    // A helper wavefront: for the species of main wavefront `hs`, generation after generation, the children 64 ... 127 (+ 128 j) of what that wavefront
    // published -- the same accessor, the same walk, the same sum as the main wavefront's own share -- into the species' parked fitness values.
    // Hand-overs are words in LDS (p_flag_*): "go" carries the generation's number, 0xffffffff means leave.
    const int hs = p_wave_index() - 2;
    unsigned int* const hw = (unsigned int*)(lds + L.help);
    double* const s_fit = lds + L.g_first + hs * L.g_stride + L.fitp;
    if ((a.debug_flags & 1) != 0 && hs == 0) return;  // (tests: a helper that never answers)
    for (unsigned int expect = 1u;; expect++) {
        if (p_flag_wait_ge(hw + 2 + hs, expect) == 0xffffffffu) break;
        const int off_p0g = (int)hw[6 + 4 * hs], off_pgt = (int)hw[7 + 4 * hs], n_walk = (int)hw[9 + 4 * hs];
        const uint32_t hctr1 = hw[8 + 4 * hs];
        const int32_t* const h_order = (const int32_t*)(lds + L.g_first + hs * L.g_stride + L.order);  // (secondary goals: the species' children in pre-selected order)
        for (int r0 = 64; r0 < n_walk; r0 += 128) {
            const int r = r0 + p_lane_fresh(), ra = r < n_walk ? r : r0;  // (a lane without a child walks a copy and drops it)
            const int c = has_sec ? h_order[ra] : ra;
            const double* const hp0 = lds + off_p0g;
            const ChildT<PB> cx[1] = {make_child_t(pb, key, hctr1, (uint32_t)c + 2u, hp0, lds + off_pgt, M)};
            double f[1];
            eval_exact_primary_n<1, true, true>(pb, cx, qc, s_slots, 0, f, s_prefix);
            if (pb->n_link_primary < pb->n_primary) f[0] = nonlink_primary(pb, make_child_x(pb, key, hctr1, (uint32_t)c + 2u, hp0, hp0 + M, hp0 + 3 * M), qc, f[0]);
            else f[0] += 0.0;
            f[0] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc);
            if (r < n_walk) s_fit[r] = f[0];
        }
        p_wave_sync();
        p_flag_store(hw + 4 + hs, expect);
    }
}
// ik_evolution_2.cpp:111-230 for the unit (or its state as the last launch left it), the launch's deadline, the bounds of the step loop
template <class Frame>
BIOIK_DEV void solve_init(Frame& F) {
    BIOIK_FRAME_NAMES(F);
    BIOIK_LANE_SCOPE;
    // ik_evolution_2.cpp:129-179: solution = seed, 2 species x 2 clones of the seed, zero momentum.
    // Inactive ops carry the seed's value in every vector, so the chain walk never distinguishes them.
    // doubles of a unit's state between two steps: per species (in ranking order) the elite buffer in use -- two individuals, genes and
    // momentum --, the solution, the bookkeeping block (the other elite buffer is written before it is read: it does not travel)
    if (!resume) {
        for (int k = tid; k < n_ops; k += nth) {
            double v = s_seed[pb->ops[k].var];
            for (int s = 0; s < 2; s++)
                for (int i = 0; i < 2; i++) {
                    double* d = s_pop + s * SP + i * 2 * M;
                    d[k] = v, d[M + k] = 0.0;
                }
            s_sol[k] = v;
        }
    } else {
        const double* c = a.carry + unit * (uint64_t)carry_n;
        for (int i = tid; i < M + 24; i += nth) {
            const double v = p_load_device(c + 2 * BF + i);
            if (i < M) s_sol[i] = v;
            else s_state[i - M] = v;
        }
        for (int i = tid; i < 2 * BF; i += nth) {  // species of rank r: into buffer 0 of its slot (the record travels with cur = 0)
            const int r = i >= BF ? 1 : 0;
            s_pop[(int)p_load_device(c + 2 * BF + M + r * 8 + 4) * SP + (i - r * BF)] = p_load_device(c + i);
        }
    }
    for (int k = tid; k < n_ops; k += nth) s_clip[k] = pb->ops[k].clip_min, s_clip[M + k] = pb->ops[k].clip_max;
    wg_barrier();
    if (pb->n_prefix > 0) {  // the joints in front of the first gene see the seed in every individual: walk them once per query
        if (tid == 0) f7_store(s_prefix, fk_prefix(pb, XV{s_sol, 1}));
        wg_barrier();
    }
    // the seed is the first solution; whether it already satisfies the goals is what the first success test will find
    FitCheck fc0{0.0, 0};
    if (!resume) fc0 = F.wg_check(wlead, tid, XV{s_sol, 1}, sp.dpos, sp.drot, sp.dtwist, 1);
    const double sol_fit = fc0.fitness;
    if (tid == 0 && !resume) {
        species_store(0, SpeciesState{P_INF, sol_fit, sol_fit, 0, 0, 0, 0, 0});
        species_store(1, SpeciesState{P_INF, sol_fit, sol_fit, 1, 1, 0, 0, 0});
        s_solst[0] = sol_fit, s_solst[1] = (double)fc0.ok;
    }
    wg_barrier();
    // (species-parallel: a lane group runs the species of its own number; its loop below is one trip with a per-lane index)
    PHASE_MARK(PH_INIT);

    // ik_parallel.h:160: the caller's timeout bounds the call, not the query.  The launch's clock starts when its first workgroup
    // does (one compare-and-swap per workgroup on a word the host zeroed); lane 0 reads the clock once per step and the verdict
    // crosses LDS, so that every wavefront of the workgroup leaves the loop in the same step.
    if (sp.timeout_ticks != 0ull) {
        if (tid == 0) {
            // (ik_parallel.h:160, 200: the reference's timeout is a point in time fixed when the call comes in.  The host turns it into device ticks
            // -- launch_solve --; a call captured into a hipGraph cannot know when it will be replayed and counts from its first workgroup's start)
            const unsigned long long t1 = a.launch_clock ? p_stamp_once(a.launch_clock, p_wall_clock()) + sp.timeout_ticks : a.deadline;
            s_deadline[0] = (double)(t1 >> 32), s_deadline[1] = (double)(t1 & 0xffffffffull);
        }
    }
    F.step_first = a.step_begin >= 0 ? a.step_begin : (int)s_state[16];  // (< 0: the step this unit left its last launch at, SolveArgs::resident)
    F.steps = F.step_first;
    F.step_end = a.step_end < sp.max_steps ? a.step_end : sp.max_steps;
}

// :341-346 (linearised phenotypes): linearise at the elite, both elites re-scored under the new linear model; under SLIM the species record goes to LDS for the generations
template <class Frame>
BIOIK_DEV void solve_linearise(Frame& F, SpeciesState& S, double* popS, int rank_it) {
    BIOIK_FRAME_NAMES(F);
    auto rank_now = [&]() { return F.rank_now(rank_it); };
    auto group_value = [&](bool glead_, int gtid_, double* s_bc_, auto&& fn) { return F.group_value(glead_, gtid_, s_bc_, fn); };
    if (!exact) {
        // :341-346 linearise at the elite; both elites are re-scored under the new linear model
        BIOIK_LANE_SCOPE;
        const double* cb = popS + S.cur * BF;
        build_approximator<BIOIK_COOP_WALKS>(pb, XV{cb, 1}, s_slots, s_frames, s_tips, s_delta, s_base, gtid, G, s_prefix);
        S.pf0 = group_value(glead, gtid, s_bc, [&]() { return eval_linear_primary(pb, XV{cb, 1}, qc, lm); });
        S.pf1 = group_value(glead, gtid, s_bc, [&]() { return eval_linear_primary(pb, XV{cb + 2 * M, 1}, qc, lm); });
    }
    if constexpr (SLIM) {
        BIOIK_LANE_SCOPE;
        if (gtid == 0) species_store(rank_now(), S);
        group_sync(G);
    }
}

// :366-378 pre-selection: children ordered by secondary fitness (stable), a random prefix survives -- n_eval children, in s_order
template <class Frame>
BIOIK_DEV void solve_preselect(Frame& F, const SpeciesState& S, double* popS, int step, int gen, uint32_t ctr1, uint64_t inside_mask, int& n_eval) {
    BIOIK_FRAME_NAMES(F);
    BIOIK_LANE_SCOPE;
    const double* cb = popS + S.cur * BF;
    const double *p0g = cb, *p0d = cb + M, *p1d = cb + 3 * M;  // (pointers to const: re-derived after the walks under SLIM)
    const uint32_t gctr = (uint32_t)step * 16u + (uint32_t)gen;
    int n_sort = 2;  // the pre-selection sorts lambda children: next power of two
    while (n_sort < lambda) n_sort <<= 1;
        // :366-378 pre-selection: children ordered by secondary fitness (stable), a random prefix survives.  Its length (:367) is drawn first -- here
        // from the counter RNG, so it is known before the children are scored and a selection can stand in for the sort
        {
            uint32_t o0, o1;
            philox2x32_10(key, rng_ctr0(0, 0), rng_ctr1(gctr, (uint32_t)S.id, RNG_PRESELECT), o0, o1);
            n_eval = (int)(o0 % (uint32_t)(lambda - 1)) + 1;
            if constexpr (WAVE2) n_eval = p_uniform(n_eval);  // (a group is a whole wavefront: the count is the same in all its lanes, so it can live in a scalar register)
        }
        bool sorted = false;
        if constexpr (CL) {
            // one wavefront (or half of one) per species and four or eight children per lane: lane l scores the children l E ... l E + E - 1 and the
            // pairs (fitness, child) are sorted where they are, in registers (sort_pairs_in_registers); only the order reaches LDS
            auto presort = [&](auto e_tag) {
                constexpr int E = decltype(e_tag)::value;
                double sf[E];
                int sc[E];
#pragma unroll
                for (int i0 = 0; i0 < E; i0 += 4) {
                    int cj[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) cj[j] = gtid * E + i0 + j < lambda ? gtid * E + i0 + j : 0;  // (padding scores child 0 and drops it)
                    double e[4];
                    if constexpr (DENSE || WAVE2 || JH) {
                        const double* const pgt = popS + (S.cur ^ 1) * BF;
                        const ChildT<PB> cx[4] = {make_child_t(pb, key, ctr1, (uint32_t)cj[0] + 2u, p0g, pgt, M), make_child_t(pb, key, ctr1, (uint32_t)cj[1] + 2u, p0g, pgt, M),
                                                  make_child_t(pb, key, ctr1, (uint32_t)cj[2] + 2u, p0g, pgt, M), make_child_t(pb, key, ctr1, (uint32_t)cj[3] + 2u, p0g, pgt, M)};
                        secondary_fitness_n<4>(pb, cx, qc, e, inside_mask);
                    } else {
                        const ChildX<PB> cx[4] = {make_child_x(pb, key, ctr1, (uint32_t)cj[0] + 2u, p0g, p0d, p1d), make_child_x(pb, key, ctr1, (uint32_t)cj[1] + 2u, p0g, p0d, p1d),
                                                  make_child_x(pb, key, ctr1, (uint32_t)cj[2] + 2u, p0g, p0d, p1d), make_child_x(pb, key, ctr1, (uint32_t)cj[3] + 2u, p0g, p0d, p1d)};
                        secondary_fitness_n<4>(pb, cx, qc, e);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) sf[i0 + j] = gtid * E + i0 + j < lambda ? e[j] : P_INF, sc[i0 + j] = gtid * E + i0 + j;
                }
                PHASE_MARK(PH_SELECTION);
                // The survivors by selection (select_threshold): which children pass is decided by the k-th least key, not by the order of all of them.
                // The keys order like (fitness, index) except among values that differ in the bits the key gave up; only the class of the threshold
                // can put a child on the wrong side of it, and only if it lies on both sides: then its exact values must all be the same (as the
                // zeros of AvoidJointLimitsGoal for every child inside its free zone are -- the index order is the stable order then), else the sort
                // below decides.  The survivors are written in lane order, not in sorted order: a walk does not care, and the selection settles
                // ties of the whole fitness by the stable order where it meets them.
                if constexpr (SELECTS && E * 64 <= 1024) {
                    if ((a.preselect & 1) != 0 && (G == 32 || G == 64)) {
                        const int drop = a.sort_key_drop, lane = tid & 63;
                        const unsigned long long gmask = G >= 64 ? ~0ull : 0xffffffffull << (lane & 32);
                        unsigned long long sk[E];
#pragma unroll
                        for (int i = 0; i < E; i++) sk[i] = sort_key(sf[i], sc[i], drop), s_sec[gtid * E + i] = sf[i];
                        bool found;
                        const unsigned long long T = select_threshold<E>(sk, lambda - gtid * E, n_eval, G, gmask, found);
                        bool in_s = false, in_n = false;
#pragma unroll
                        for (int i = 0; i < E; i++) {
                            const bool cls = ((sk[i] ^ T) >> drop) == 0ull;
                            in_s = in_s || (cls && sk[i] <= T), in_n = in_n || (cls && sk[i] > T);
                        }
                        const unsigned long long any_s = p_ballot(in_s) & gmask, any_n = p_ballot(in_n) & gmask;  // (both asked by every lane: no short circuit)
                        const bool straddle = any_s != 0ull && any_n != 0ull;
                        bool bad = !found;
                        if (p_ballot(straddle) != 0ull) {
                            // (any member's value through a word of the group's scratch: if all are the same it does not matter whose arrives)
                            unsigned long long mine = 0ull;
                            bool have = false, mixed = false;
#pragma unroll
                            for (int i = 0; i < E; i++) {
                                if (((sk[i] ^ T) >> drop) != 0ull) continue;
                                unsigned long long v;
                                const double e = s_sec[gtid * E + i];
                                __builtin_memcpy(&v, &e, 8);
                                if (!have) mine = v, have = true;
                                else mixed = mixed || v != mine;
                            }
                            if (straddle && have) __builtin_memcpy(&s_bc[0], &mine, 8);
                            group_sync(G);
                            unsigned long long ref;
                            __builtin_memcpy(&ref, &s_bc[0], 8);
                            bad = bad || (straddle && have && (mixed || mine != ref));
                            group_sync(G);
                        }
                        if (p_ballot(bad) == 0ull) {  // (the halves of a wavefront decide together: one path through the code)
                            const unsigned long long below = (1ull << lane) - 1ull;
                            int base = 0;
#pragma unroll
                            for (int i = 0; i < E; i++) {
                                const bool sv = sk[i] <= T;
                                const unsigned long long b = p_ballot(sv) & gmask;
                                if (sv) s_order[base + p_popc64(b & below)] = sc[i];
                                base += p_popc64(b);
                            }
                            group_sync(G);
                            sorted = true;
                            return;
                        }
                    }
                }
                // Keys first (sort_key: one compare and two dwords per exchange).  The exact values wait in LDS for the check behind the sort: sorted
                // neighbours whose keys agree above the index bits are equal (then the index order is the stable order) or differ in their lowest
                // ten mantissa bits only -- in that case, which a generation meets about once in a billion, the pairs are sorted again, exactly.
                bool exact_order = true;
                if constexpr (E * 64 <= 1024) {
                    unsigned long long sk[E];
#pragma unroll
                    for (int i = 0; i < E; i++) sk[i] = sort_key(sf[i], sc[i], a.sort_key_drop), s_sec[gtid * E + i] = sf[i];
                    sort_keys_in_registers<E>(sk, gtid, G);
                    group_sync(G);  // (every lane's values are in LDS)
                    const unsigned long long next_lane = p_shfl(sk[0], (tid & 63) + 1);  // (the first key of the lane behind this one; the group's last lane has none)
                    bool wrong = false;
#pragma unroll
                    for (int i = 0; i < E; i++) {
                        const unsigned long long a = sk[i], b2 = i + 1 < E ? sk[i + 1 < E ? i + 1 : i] : next_lane;
                        const bool has_next = i + 1 < E || gtid + 1 < G;
                        if (has_next && ((a ^ b2) >> 10) == 0ull) wrong = wrong || s_sec[(int)(a & 0x3ffull)] > s_sec[(int)(b2 & 0x3ffull)];
                    }
                    exact_order = p_ballot(wrong) == 0ull;  // (both halves of a wavefront that carries two species decide together: the network is the same code)
#pragma unroll
                    for (int i = 0; i < E; i++) sc[i] = (int)(sk[i] & 0x3ffull);
                    if (!exact_order) {
#pragma unroll
                        for (int i = 0; i < E; i++) sc[i] = gtid * E + i, sf[i] = s_sec[gtid * E + i];
                    }
                } else {
                    exact_order = false;
                }
                if (!exact_order) sort_pairs_in_registers<E>(sf, sc, gtid, G);
#pragma unroll
                for (int i = 0; i < E; i++) s_order[gtid * E + i] = sc[i];
                group_sync(G);
                sorted = true;
            };
            const int per_lane = (G <= 64 && (G & (G - 1)) == 0) ? n_sort / G : 0;
            if constexpr (!WAVE2 && !LIN && !DENSE)
                if (per_lane == 4) presort(std::integral_constant<int, 4>{});
            if constexpr (!JH && !LIN && !DENSE)
                if (per_lane == 8) presort(std::integral_constant<int, 8>{});
        }
        if (!sorted) {
        if (columnless && lambda >= 4 * G) {  // four children per lane and trip: four independent hash -> Gaussian -> clip -> cost chains
            for (int c = gtid; c < lambda; c += 4 * G) {
                int cj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) cj[j] = c + j * G < lambda ? c + j * G : c;  // (a tail repeats the first child and drops it)
                double e[4];
                if constexpr (DENSE || WAVE2 || JH) {
                    const double* const pgt = popS + (S.cur ^ 1) * BF;
                    const ChildT<PB> cx[4] = {make_child_t(pb, key, ctr1, (uint32_t)cj[0] + 2u, p0g, pgt, M), make_child_t(pb, key, ctr1, (uint32_t)cj[1] + 2u, p0g, pgt, M),
                                              make_child_t(pb, key, ctr1, (uint32_t)cj[2] + 2u, p0g, pgt, M), make_child_t(pb, key, ctr1, (uint32_t)cj[3] + 2u, p0g, pgt, M)};
                    secondary_fitness_n<4>(pb, cx, qc, e, inside_mask);
                } else {
                    const ChildX<PB> cx[4] = {make_child_x(pb, key, ctr1, (uint32_t)cj[0] + 2u, p0g, p0d, p1d), make_child_x(pb, key, ctr1, (uint32_t)cj[1] + 2u, p0g, p0d, p1d),
                                              make_child_x(pb, key, ctr1, (uint32_t)cj[2] + 2u, p0g, p0d, p1d), make_child_x(pb, key, ctr1, (uint32_t)cj[3] + 2u, p0g, p0d, p1d)};
                    secondary_fitness_n<4>(pb, cx, qc, e);
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (c + j * G < lambda) s_sec[c + j * G] = e[j];
            }
        } else if (columnless && lambda >= 2 * G) {
            for (int c = gtid; c < lambda; c += 2 * G) {
                const int c1 = c + G < lambda ? c + G : c;
                const ChildX<PB> cx[2] = {make_child_x(pb, key, ctr1, (uint32_t)c + 2u, p0g, p0d, p1d), make_child_x(pb, key, ctr1, (uint32_t)c1 + 2u, p0g, p0d, p1d)};
                double e[2];
                secondary_fitness_n<2>(pb, cx, qc, e);
                s_sec[c] = e[0];
                if (c + G < lambda) s_sec[c + G] = e[1];
            }
        } else {
            for (int c = gtid; c < lambda; c += G) {
                if (columnless) {
                    s_sec[c] = secondary_fitness<true>(pb, make_child_x(pb, key, ctr1, (uint32_t)c + 2u, p0g, p0d, p1d), qc);
                } else {
                    reproduce_child(pb, key, ctr1, (uint32_t)c + 2u, p0g, p0d, p1d, xcol, nth, nullptr, 0);
                    s_sec[c] = secondary_fitness<true>(pb, xl, qc);
                }
            }
        }
        // ascending by (secondary fitness, child index) -- the order of a stable sort -- with a bitonic network over the next power
        // of two (padding: +inf): log2(n)(log2(n)+1)/2 rounds of n/2 compare-exchanges shared by the group's lanes, instead of
        // lambda comparisons per child (512 children: 45 x 4 exchanges per lane instead of 4096 comparisons)
        PHASE_MARK(PH_SELECTION);  // (profiling build: the children's secondary fitness | the sort | the draw of the survivors' count)
        for (int i = gtid; i < n_sort; i += G) {
            if (i >= lambda) s_sec[i] = P_INF;
            s_order[i] = i;
        }
        group_sync(G);
        for (int k = 2; k <= n_sort; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = gtid; t < (n_sort >> 1); t += G) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const double fa = s_sec[lo], fb = s_sec[hi];
                    const int ca = s_order[lo], cb = s_order[hi];
                    const bool a_after_b = (fa > fb) || (fa == fb && ca > cb);
                    if (a_after_b == ((lo & k) == 0)) s_sec[lo] = fb, s_sec[hi] = fa, s_order[lo] = cb, s_order[hi] = ca;
                }
                group_sync(G);
            }
        }
        PHASE_MARK(PH_MEMETICS);
        PHASE_MARK(PH_PRESELECT);
}

// genotype -> phenotype -> fitness (:391-407): lane r of the group scores the child at sorted position r; the lane's best two in (b1, b2), or -- the kernels of the
// 128-register budget -- every child's fitness parked in LDS
template <class Frame>
BIOIK_DEV void solve_walks(Frame& F, SpeciesState& S, double*& popS, int rank_it, int step, int gen, uint32_t ctr1, int n_eval, double& b1f, int& b1p, double& b2f, int& b2p) {
    BIOIK_FRAME_NAMES(F);
    auto rank_now = [&]() { return F.rank_now(rank_it); };
    unsigned int& gen_count = F.gen_count;
    bool& lost = F.lost;
    auto rendezvous_lost = [&]() { F.rendezvous_lost(); };
    BIOIK_LANE_SCOPE;
    const double* cb = popS + S.cur * BF;
    const double *p0g = cb, *p0d = cb + M, *p1d = cb + 3 * M;  // (pointers to const: re-derived after the walks under SLIM)
    const uint32_t gctr = (uint32_t)step * 16u + (uint32_t)gen;
    // every child keeps its own column until selection; not with quaternion genes: a winner's momentum is taken from the
    // gene before its renormalisation (:299 vs :320-324), so those winners are re-derived from the RNG
    const bool stored = !columnless && n_cols * G >= lambda && (LEAN || pb->n_quat == 0);
    auto offer = [&](double f, int pos) { top2_insert(b1f, b1p, b2f, b2p, f, pos); };  // the lane's best two so far
    // (parity suites, SolveArgs::preselect: the parked fitness values made coarse where the walks are over, so that children tie and the tie order is exercised)
    auto coarse_parked = [&](int gt) {
        if constexpr (SLIM && CL) {
            if (const int tie_bits = a.preselect >> 8) {
                double* const s_fq = gbase + L.fitp;
                for (int r = gt; r < n_eval; r += G) {
                    unsigned long long v;
                    __builtin_memcpy(&v, &s_fq[r], 8);
                    v &= ~((1ull << tie_bits) - 1ull);
                    __builtin_memcpy(&s_fq[r], &v, 8);
                }
                group_sync(G);
            }
        }
    };
    if (!JOINT && stored && child_pairs && exact) {
        // two children per trip: columns j and j+1 of this lane (an odd tail repeats the first child and drops it)
        for (int r = gtid, j = 0; r < n_eval; r += 2 * G, j += 2) {
            const int r1 = r + G;
            const bool two = r1 < n_eval;
            const int c0 = has_sec ? s_order[r] : r;
            const int c1 = two ? (has_sec ? s_order[r1] : r1) : c0;
            double* const xc[2] = {xcol + (size_t)j * M * nth, two ? xcol + (size_t)(j + 1) * M * nth : xcol + (size_t)j * M * nth};
            const uint32_t ci[2] = {(uint32_t)c0 + 2u, (uint32_t)c1 + 2u};
            reproduce_children<2>(pb, key, ctr1, ci, p0g, p0d, p1d, xc, nth);
            PHASE_MARK(PH_REPRODUCE);
            const XV xv[2] = {XV{xc[0], nth}, XV{xc[1], nth}};
            double f[2];
            eval_exact_primary_n<2>(pb, xv, qc, s_slots, pb->n_slots * 7 * nth, f, s_prefix);
            PHASE_MARK(PH_FITNESS);
            offer(f[0], r + 2);
            if (two) offer(f[1], r1 + 2);
        }
    } else if (JOINT) {
        if constexpr (JOINT) {
            // the pre-selected children of both species as one list over the 64 lanes: item i < n0 is species 0's child at sorted position i,
            // item n0 + i species 1's; two items per lane and trip.  What a lane needs of the OTHER half's species -- its elites,
            // its random stream, its sorted order -- is uniform inside that half: one v_readlane each.
            const int cb_off = (int)(cb - lds);
            const int cbo[2] = {p_read_lane(cb_off, 0), p_read_lane(cb_off, 32)};
            const int ob_off = (int)(popS + (S.cur ^ 1) * BF - lds);  // (the species' other elite buffer: under SLIM the table of the parents' mixed momentum)
            const int obo[2] = {p_read_lane(ob_off, 0), p_read_lane(ob_off, 32)};
            const int ct[2] = {p_read_lane((int)ctr1, 0), p_read_lane((int)ctr1, 32)};
            const int ne0 = p_read_lane(n_eval, 0), total = ne0 + p_read_lane(n_eval, 32);
            if constexpr (SLIM) {
                // (fit_park, as in the other kernels of the 128-register budget: every item's fitness goes to ITS species' array in LDS; when the
                // walks are over a half reads its own species' entries back and reduces them inside the half)
                for (int t0 = 0; t0 < total; t0 += 128) {
                    double f[2];
                    {
                        BIOIK_LANE_SCOPE;
                        const int i0r = t0 + tid, i0 = i0r < total ? i0r : 0, i1 = i0r + 64 < total ? i0r + 64 : i0;
                        const int sp0 = i0 >= ne0 ? 1 : 0, sp1 = i1 >= ne0 ? 1 : 0;
                        const int r0 = i0 - (sp0 ? ne0 : 0), r1 = i1 - (sp1 ? ne0 : 0);
                        // (the other species' order array: base + species x stride, not a select between two addresses -- those would be two
                        // registers that live as long as the kernel)
                        const int32_t* const ord0 = (const int32_t*)(lds + L.g_first + L.order);
                        const int c0 = ord0[sp0 * 2 * L.g_stride + r0], c1 = ord0[sp1 * 2 * L.g_stride + r1];
                        const double *pa = lds + (sp0 ? cbo[1] : cbo[0]), *pc = lds + (sp1 ? cbo[1] : cbo[0]);
                        // (the parents' mixed momentum from the item's species' table, built where the generation began: the species' other elite buffer)
                        const ChildT<PB> cx[2] = {make_child_t(pb, key, (uint32_t)(sp0 ? ct[1] : ct[0]), (uint32_t)c0 + 2u, pa, lds + (sp0 ? obo[1] : obo[0]), M),
                                                  make_child_t(pb, key, (uint32_t)(sp1 ? ct[1] : ct[0]), (uint32_t)c1 + 2u, pc, lds + (sp1 ? obo[1] : obo[0]), M)};
                        PHASE_MARK(PH_REPRODUCE);
                        eval_exact_primary_n<2, true>(pb, cx, qc, s_slots, pb->n_slots * 7 * nth, f, s_prefix);
                        PHASE_MARK(PH_FITNESS);
                    }
                    BIOIK_LANE_SCOPE;
                    const int i0r = t0 + tid, i0 = i0r < total ? i0r : 0, i1 = i0r + 64 < total ? i0r + 64 : i0;
                    const int sp0 = i0 >= ne0 ? 1 : 0, sp1 = i1 >= ne0 ? 1 : 0;
                    const int r0 = i0 - (sp0 ? ne0 : 0), r1 = i1 - (sp1 ? ne0 : 0);
                    if (pb->n_link_primary < pb->n_primary) {  // (primary goals over the joint values: the accessors are built again behind the walk)
                        const int32_t* const ord0 = (const int32_t*)(lds + L.g_first + L.order);
                        const int c0 = ord0[sp0 * 2 * L.g_stride + r0], c1 = ord0[sp1 * 2 * L.g_stride + r1];
                        const double *pa = lds + (sp0 ? cbo[1] : cbo[0]), *pc = lds + (sp1 ? cbo[1] : cbo[0]);
                        const ChildX<PB> cx[2] = {make_child_x(pb, key, (uint32_t)(sp0 ? ct[1] : ct[0]), (uint32_t)c0 + 2u, pa, pa + M, pa + 3 * M),
                                                  make_child_x(pb, key, (uint32_t)(sp1 ? ct[1] : ct[0]), (uint32_t)c1 + 2u, pc, pc + M, pc + 3 * M)};
                        f[0] = nonlink_primary(pb, cx[0], qc, f[0]), f[1] = nonlink_primary(pb, cx[1], qc, f[1]);
                    } else {
                        f[0] += 0.0, f[1] += 0.0;
                    }
                    f[0] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc), f[1] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc);
                    if (i0r < total) (lds + L.g_first + sp0 * L.g_stride + L.fitp)[r0] = f[0];
                    if (i0r + 64 < total) (lds + L.g_first + sp1 * L.g_stride + L.fitp)[r1] = f[1];
                }
                p_wave_sync();  // (the selection below reads the parked values of its own species back, as keys)
            }
        }
    } else if (!JOINT && columnless && child_pairs && exact) {
        // two children per trip, both computed where they are read: two independent dependency chains per lane
        if constexpr (HELPED) {
            // The helped kernel: this wavefront walks the children 0 ... 63 (+ 128 j) ONE at a time, its helper wavefront the children 64 ... 127
            // (+ 128 j) -- half the instructions of the pair walk per wavefront, on SIMDs a launch that cannot fill the chip leaves idle.  The
            // generation is published in the species' mailbox and "go" raised before the own walk starts; "done" is waited for behind it.
            unsigned int* const hw = (unsigned int*)(lds + L.help);
            {
                BIOIK_LANE_SCOPE;
                const int w = p_wave_index();
                if (gtid == 0) {
                    hw[6 + 4 * w] = (unsigned int)(int)(p0g - lds), hw[7 + 4 * w] = (unsigned int)(int)(popS + (S.cur ^ 1) * BF - lds);
                    hw[8 + 4 * w] = ctr1, hw[9 + 4 * w] = (unsigned int)n_eval;
                }
                p_wave_sync();
                gen_count++;
                p_flag_store(hw + 2 + w, gen_count);
            }
            for (int r0 = 0; r0 < n_eval; r0 += 128) {
                double f[1];
                {
                    BIOIK_LANE_SCOPE;
                    const int r = r0 + gtid, ra = r < n_eval ? r : 0;
                    const int c = has_sec ? s_order[ra] : ra;
                    const uint32_t ctr1t = rng_ctr1(gctr, (uint32_t)species_load(rank_now()).id, RNG_REPRODUCE);
                    const double* const pgt = popS + (S.cur ^ 1) * BF;
                    const ChildT<PB> cx[1] = {make_child_t(pb, key, ctr1t, (uint32_t)c + 2u, p0g, pgt, M)};
                    PHASE_MARK(PH_REPRODUCE);
                    eval_exact_primary_n<1, true, true>(pb, cx, qc, s_slots, 0, f, s_prefix);
                }
                BIOIK_LANE_SCOPE;
                const int r = r0 + gtid;
                if (pb->n_link_primary < pb->n_primary) {
                    const SpeciesState S2 = species_load(rank_now());
                    const double* cb2 = s_pop + S2.slot * SP + S2.cur * BF;
                    const uint32_t ctr2 = rng_ctr1(gctr, (uint32_t)S2.id, RNG_REPRODUCE);
                    const int ra2 = r < n_eval ? r : 0;
                    f[0] = nonlink_primary(pb, make_child_x(pb, key, ctr2, (uint32_t)(has_sec ? s_order[ra2] : ra2) + 2u, cb2, cb2 + M, cb2 + 3 * M), qc, f[0]);
                } else {
                    f[0] += 0.0;
                }
                f[0] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc);
                PHASE_MARK(PH_FITNESS);
                double* const s_fit = gbase + L.fitp;
                if (r < n_eval) s_fit[r] = f[0];
            }
            {
                BIOIK_LANE_SCOPE;
                p_wave_sync();
                if (!lost && p_flag_wait_ge(hw + 4 + p_wave_index(), gen_count) == 0xffffffffu) rendezvous_lost();  // the helper's share of the fitness values is parked
            }
            {
                BIOIK_LANE_SCOPE;
                const double* const s_fit2 = gbase + L.fitp;
                coarse_parked(gtid);
                for (int r = gtid; r < n_eval; r += G) offer(s_fit2[r], r + 2);
            }
        } else if constexpr (SLIM) {
            // the fitness values go to LDS (fit_park) and come back when the walks are over: nothing but the lane number lives across a walk
            // (so the trip counter is uniform, and what follows a walk -- the goals that read no link, rarely present -- starts from the lane number again)
            for (int r0 = 0; r0 < n_eval; r0 += 2 * G) {
                double f[2];
                {
                    BIOIK_LANE_SCOPE;
                    const int r = r0 + gtid, r1 = r + G;
                    const bool two = r1 < n_eval;
                    const int ra = r < n_eval ? r : 0, rb = two ? r1 : ra;  // (a lane without a child in this trip walks a copy and drops it)
                    const int c0 = has_sec ? s_order[ra] : ra, c1 = has_sec ? s_order[rb] : rb;
                    const uint32_t ctr1t = rng_ctr1(gctr, (uint32_t)species_load(rank_now()).id, RNG_REPRODUCE);  // (= ctr1, from the record: the stream's hash is not carried over the walks)
                    // (FIXED: the launcher hands these kernels serial chains only, DevProblem::serial_chain -- the usual robot arm: one chain, nothing parked)
                    if constexpr (DENSE || WAVE2) {
                        const double* const pgt = popS + (S.cur ^ 1) * BF;  // (the table of the parents' mixed momentum, built where the generation begins)
                        const ChildT<PB> cx[2] = {make_child_t(pb, key, ctr1t, (uint32_t)c0 + 2u, p0g, pgt, M), make_child_t(pb, key, ctr1t, (uint32_t)c1 + 2u, p0g, pgt, M)};
                        PHASE_MARK(PH_REPRODUCE);
                        eval_exact_primary_n<2, true, true>(pb, cx, qc, s_slots, 0, f, s_prefix);
                    } else {
                        const ChildX<PB> cx[2] = {make_child_x(pb, key, ctr1t, (uint32_t)c0 + 2u, p0g, p0d, p1d), make_child_x(pb, key, ctr1t, (uint32_t)c1 + 2u, p0g, p0d, p1d)};
                        PHASE_MARK(PH_REPRODUCE);
                        eval_exact_primary_n<2, true, FIXED != 0>(pb, cx, qc, s_slots, pb->n_slots * 7 * nth, f, s_prefix);
                    }
                }
                BIOIK_LANE_SCOPE;
                const int r = r0 + gtid, r1 = r + G;
                const bool two = r1 < n_eval;
                if (pb->n_link_primary < pb->n_primary) {  // (primary goals over the joint values: the accessors are built again, nothing of them crossed the walk)
                    const SpeciesState S2 = species_load(rank_now());
                    const double* cb2 = s_pop + S2.slot * SP + S2.cur * BF;
                    const uint32_t ctr2 = rng_ctr1(gctr, (uint32_t)S2.id, RNG_REPRODUCE);
                    const int ra = r < n_eval ? r : 0, rb = two ? r1 : ra;
                    const int c0 = has_sec ? s_order[ra] : ra, c1 = has_sec ? s_order[rb] : rb;
                    const ChildX<PB> cx[2] = {make_child_x(pb, key, ctr2, (uint32_t)c0 + 2u, cb2, cb2 + M, cb2 + 3 * M),
                                              make_child_x(pb, key, ctr2, (uint32_t)c1 + 2u, cb2, cb2 + M, cb2 + 3 * M)};
                    f[0] = nonlink_primary(pb, cx[0], qc, f[0]), f[1] = nonlink_primary(pb, cx[1], qc, f[1]);
                } else {
                    f[0] += 0.0, f[1] += 0.0;  // (nonlink_primary of no goal: the sum it returns)
                }
                f[0] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc), f[1] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc);
                PHASE_MARK(PH_FITNESS);
                double* const s_fit = gbase + L.fitp;
                if (r < n_eval) s_fit[r] = f[0];
                if (two) s_fit[r1] = f[1];
            }
            if constexpr (!DENSE) {  // (DENSE: the selection below reads the parked values itself, as keys)
                BIOIK_LANE_SCOPE;
                const double* const s_fit2 = gbase + L.fitp;
                coarse_parked(gtid);
                for (int r = gtid; r < n_eval; r += G) offer(s_fit2[r], r + 2);  // (its own entries: a lane's LDS accesses stay in program order)
            }
        } else
        for (int r = gtid; r < n_eval; r += 2 * G) {
            const int r1 = r + G;
            const bool two = r1 < n_eval;
            const int c0 = has_sec ? s_order[r] : r;
            const int c1 = two ? (has_sec ? s_order[r1] : r1) : c0;
            const ChildX<PB> cx[2] = {make_child_x(pb, key, ctr1, (uint32_t)c0 + 2u, p0g, p0d, p1d),
                                      make_child_x(pb, key, ctr1, (uint32_t)c1 + 2u, p0g, p0d, p1d)};
            PHASE_MARK(PH_REPRODUCE);
            double f[2];
            eval_exact_primary_n<2>(pb, cx, qc, s_slots, pb->n_slots * 7 * nth, f, s_prefix);
            PHASE_MARK(PH_FITNESS);
            offer(f[0], r + 2);
            if (two) offer(f[1], r1 + 2);
        }
    } else if (!JOINT && columnless) {
        for (int r = gtid; r < n_eval; r += G) {
            const int c = has_sec ? s_order[r] : r;
            const auto cx = make_child_x(pb, key, ctr1, (uint32_t)c + 2u, p0g, p0d, p1d);
            PHASE_MARK(PH_REPRODUCE);
            const double f = exact ? eval_exact_primary(pb, cx, qc, s_slots, s_prefix) : eval_linear_primary(pb, cx, qc, lm);
            PHASE_MARK(PH_FITNESS);
            offer(f, r + 2);
        }
    } else {
        for (int r = gtid, j = 0; r < n_eval; r += G, j++) {
            int c = has_sec ? s_order[r] : r;
            double* xc = stored ? xcol + (size_t)j * M * nth : xcol;
            const XV xv{xc, nth};
            reproduce_child(pb, key, ctr1, (uint32_t)c + 2u, p0g, p0d, p1d, xc, nth, nullptr, 0);
            PHASE_MARK(PH_REPRODUCE);
            double f = exact ? eval_exact_primary(pb, xv, qc, s_slots, s_prefix) : eval_linear_primary(pb, xv, qc, lm);
            PHASE_MARK(PH_FITNESS);
            offer(f, r + 2);
        }
    }
}

// elitist top-2 selection (:410-431), including the tie order of the reference's selection sort, and the winners into the species' other elite buffer
template <class Frame>
BIOIK_DEV void solve_select_and_copy(Frame& F, SpeciesState& S, double*& popS, int rank_it, int step, int gen, uint32_t ctr1, int n_eval, double& b1f, int& b1p, double& b2f, int& b2p) {
    BIOIK_FRAME_NAMES(F);
    auto rank_now = [&]() { return F.rank_now(rank_it); };
    BIOIK_LANE_SCOPE;  // (a lane scope of its own: nothing of the lane numbers in front of the walks is used behind them)
    const double* cb = popS + S.cur * BF;
    const double *p0g = cb, *p0d = cb + M, *p1d = cb + 3 * M;  // (pointers to const: re-derived after the walks under SLIM)
    // every child keeps its own column until selection; not with quaternion genes: a winner's momentum is taken from the
    // gene before its renormalisation (:299 vs :320-324), so those winners are re-derived from the RNG
    const bool stored = !columnless && n_cols * G >= lambda && (LEAN || pb->n_quat == 0);
    auto offer = [&](double f, int pos) { top2_insert(b1f, b1p, b2f, b2p, f, pos); };  // the lane's best two so far
    // (parity suites, SolveArgs::preselect: the parked fitness values made coarse where the walks are over, so that children tie and the tie order is exercised)
    auto coarse_parked = [&](int gt) {
        if constexpr (SLIM && CL) {
            if (const int tie_bits = a.preselect >> 8) {
                double* const s_fq = gbase + L.fitp;
                for (int r = gt; r < n_eval; r += G) {
                    unsigned long long v;
                    __builtin_memcpy(&v, &s_fq[r], 8);
                    v &= ~((1ull << tie_bits) - 1ull);
                    __builtin_memcpy(&s_fq[r], &v, 8);
                }
                group_sync(G);
            }
        }
    };
    if constexpr (SLIM) S = species_load(rank_now()), popS = s_pop + S.slot * SP, cb = popS + S.cur * BF, p0g = cb, p0d = cb + M, p1d = cb + 3 * M;
    const uint32_t ctr1w = SLIM ? rng_ctr1((uint32_t)step * 16u + (uint32_t)gen, (uint32_t)S.id, RNG_REPRODUCE) : ctr1;  // (the winners' stream: not carried through the walks under SLIM)
    if constexpr (DENSE || JH) coarse_parked(gtid);  // (the other kernels: in front of their lanes' offers)
    // (the joint walk has reduced over the whole wavefront already; under SLIM it parks its values like the other walks, and the reduction
    // inside the half stands here, between the lanes' reads of the species record above and its update below)
    if constexpr (DENSE || JH) {
        // The two best children of a half-wavefront's species from KEYS (sort_key: the fitness's upper bits and the position): a lane's candidates
        // are its entries of the parked values, the group's least key and the least of the rest are two minimum reductions of one 64-bit number
        // each (against a butterfly that merges sorted (fitness, position) pairs: half the instructions).  Exact unless a second candidate
        // shares the upper bits of a winner's fitness: the runner-up is the second least key, so nobody shares the winner's unless the runner-up
        // does, and the candidates that share the runner-up's are counted; then, or on exact ties, the pairs themselves are reduced.
        const double* const s_fit2 = gbase + L.fitp;
        const int drop = a.sort_key_drop;
        unsigned long long k1 = ~0ull, k2 = ~0ull;
        for (int r = gtid; r < n_eval; r += G) {
            const unsigned long long k = sort_key(s_fit2[r], r + 2, drop);
            const bool w1 = k < k1, w2 = k < k2;
            k2 = w1 ? k1 : (w2 ? k : k2);
            k1 = w1 ? k : k1;
        }
        const unsigned long long B1 = half_min_u64(k1);
        const unsigned long long B2 = half_min_u64(k1 == B1 ? k2 : k1);
        int shares = 0;  // this lane's candidates with the runner-up's upper bits (the runner-up itself is one of the group's)
        for (int r = gtid; r < n_eval; r += G) shares += ((sort_key(s_fit2[r], r + 2, drop) ^ B2) >> drop) == 0ull ? 1 : 0;
        const unsigned long long one = p_ballot(shares > 0), more = p_ballot(shares > 1);
        const uint32_t mine = grp ? (uint32_t)(one >> 32) : (uint32_t)one;
        const bool in_doubt = B2 != ~0ull && ((mine & (mine - 1u)) != 0u);
        if (p_ballot(in_doubt) == 0ull && more == 0ull) {  // (the halves of the wavefront decide together: one path through the code)
            b1p = (int)(B1 & 1023ull), b1f = s_fit2[b1p - 2];
            if (B2 != ~0ull) b2p = (int)(B2 & 1023ull), b2f = s_fit2[b2p - 2];
        } else {
            for (int r = gtid; r < n_eval; r += G) offer(s_fit2[r], r + 2);
            top2_wave(b1f, b1p, b2f, b2p, G);
        }
    } else {
        top2_wave(b1f, b1p, b2f, b2p, G);
    }
    PHASE_MARK(PH_SEL_TOP2);
    top2_xwave(b1f, b1p, b2f, b2p, s_red, gtid, G);  // now the two best children of the whole generation
    PHASE_MARK(PH_SEL_XWAVE);
    if constexpr (SELECTS) {
        // The reference's selection takes, of two children with the same fitness, the one the pre-selection's stable sort put first, and its swap
        // of the first winner with parent 0 puts that parent at the winner's POSITION for the second pass (:410-423).  The survivors of
        // select_threshold stand in lane order, so a position does not say which came first.  It matters where a winner's fitness is shared by
        // another child, or the runner-up's by parent 0 -- children with the same genes, which a generation all but never has --: then the
        // candidates' secondary fitness is computed again, the stable order (secondary fitness, child index) picks among them, and the two
        // winners change places in the list if their positions say the opposite of it.  (After a sort the positions say the same: no word needed
        // on which of the two it was.)
        if (has_sec) {
            double* const s_fv = gbase + L.fitp;
            int t1 = 0, t2 = 0;
            for (int r = gtid; r < n_eval; r += G) {
                const double v = s_fv[r];
                t1 += v == b1f ? 1 : 0, t2 += v == b2f ? 1 : 0;
            }
            const unsigned long long gm = G >= 64 ? ~0ull : 0xffffffffull << (tid & 32);
            const unsigned long long o1 = p_ballot(t1 > 0) & gm, o2 = p_ballot(t2 > 0) & gm;
            const bool tie = (p_ballot(t1 > 1 || t2 > 1) & gm) != 0ull || (o1 & (o1 - 1ull)) != 0ull || (o2 & (o2 - 1ull)) != 0ull || (b2p != 0x7fffffff && b2f == S.pf0);
            if (p_ballot(tie) != 0ull) {  // (both halves of a wavefront take the path when one of them must)
                // the position of the candidate with this fitness that the stable order puts first, its secondary fitness and its child index
                auto stable_first = [&](double fwant, int skip, double& wf, int& wc) -> int {
                    double bs = P_INF;
                    int bc = 0x7fffffff, br = 0x7fffffff;
                    for (int r = gtid; r < n_eval; r += G) {
                        if (!(s_fv[r] == fwant) || r == skip) continue;
                        const int c = s_order[r];
                        const double sec = secondary_fitness<true>(pb, make_child_x(pb, key, ctr1w, (uint32_t)c + 2u, p0g, p0d, p1d), qc);
                        if (sec < bs || (sec == bs && c < bc)) bs = sec, bc = c, br = r;
                    }
                    double w2f = P_INF;
                    int w2c = 0x7fffffff;
                    wf = bs, wc = bc;
                    top2_wave(wf, wc, w2f, w2c, G);  // (wf, wc): the least (secondary fitness, child) of the group
                    const unsigned long long own = p_ballot(bc == wc && bc != 0x7fffffff) & gm;
                    const int src = own != 0ull ? __builtin_ctzll(own) : (tid & 63);
                    const int rr = p_shfl(br, src);
                    return own != 0ull ? rr : 0x7fffffff;
                };
                double e1, e2;
                int c1, c2;
                const int r1 = stable_first(b1f, -1, e1, c1);
                const int r2 = stable_first(b2f, r1, e2, c2);
                if (tie && r1 != 0x7fffffff) {
                    b1p = r1 + 2;
                    if (b2p != 0x7fffffff && r2 != 0x7fffffff) {
                        b2p = r2 + 2;
                        const bool first_is_first = e1 < e2 || (e1 == e2 && c1 < c2);
                        if (first_is_first != (r1 < r2)) {  // the two change places: their positions then say what the stable order says
                            if (gtid == 0) {
                                const int ca = s_order[r1], cb2 = s_order[r2];
                                const double fa = s_fv[r1], fb = s_fv[r2];
                                s_order[r1] = cb2, s_order[r2] = ca, s_fv[r1] = fb, s_fv[r2] = fa;
                            }
                            b1p = r2 + 2, b2p = r1 + 2;
                        }
                    }
                }
                group_sync(G);
            }
        }
    }
    Cand first{S.pf0, 0, 0};
    if (cand_better(S.pf1, 1, first.f, first.pos)) first = Cand{S.pf1, 1, 1};
    if (cand_better(b1f, b1p, first.f, first.pos)) first = Cand{b1f, b1p, b1p};
    const double c2f = (b1p == first.id) ? b2f : b1f;
    const int c2p = (b1p == first.id) ? b2p : b1p;
    Cand second{P_INF, 0x7fffffff, -1};
    if (first.id != 0) second = Cand{S.pf0, first.pos, 0};  // parent 0 was swapped to the winner's position
    if (first.id != 1 && (second.id < 0 || cand_better(S.pf1, 1, second.f, second.pos))) second = Cand{S.pf1, 1, 1};
    if (second.id < 0 || cand_better(c2f, c2p, second.f, second.pos)) second = Cand{c2f, c2p, c2p};
    // the winners become the elites (written to the species' other buffer)
    // lanes 0..31 of the group write the first winner, lanes 32..63 the second, lane k its ops k, k + 32
    double* nb = popS + (S.cur ^ 1) * BF;
    // (a half-wave group has 32 lanes: sixteen per winner where the ops are no more than that, else one winner per pass)
    const bool both_at_once = G >= 64 || M <= 16;
    for (int pass = 0; pass < (both_at_once ? 1 : 2); pass++) {
        BIOIK_LANE_SCOPE;
        if (gtid >= 64) break;
        const int i = G >= 64 ? gtid >> 5 : (both_at_once ? gtid >> 4 : pass), k0 = G >= 64 || !both_at_once ? gtid & 31 : gtid & 15;
        const int id = i == 0 ? first.id : second.id;
        double* dst = nb + i * 2 * M;
        if (id < 2) {
            const double* src = cb + id * 2 * M;
            if (k0 < M) dst[k0] = src[k0], dst[M + k0] = src[M + k0];
            if (M > 32 && k0 + 32 < M) dst[k0 + 32] = src[k0 + 32], dst[M + k0 + 32] = src[M + k0 + 32];  // (at most 64 ops)
        } else if (stored) {
            // the winner's genes are still in its owner's column; its momentum follows from the genes
            // (ik_evolution_2.cpp:299: gradient = mix(parent_gradient, gene - parent_gene, 0.3))
            BIOIK_FP_STRICT
            const int r = id - 2;
            const int c = has_sec ? s_order[r] : r;
            const double fmix = (((uint32_t)c + 2u) % 2u == 0u) ? 0.2 : 0.0;
            // (column r / G, lane r % G of the group; the group sizes the launcher produces are powers of two: no integer division)
            const int r_col = (G & (G - 1)) == 0 ? r >> (31 - __builtin_clz((unsigned)G)) : r / G, r_lane = r - r_col * G;
            const double* src = (lds + L.xcol) + (size_t)r_col * M * nth + (grp * G + r_lane);
            auto take = [&](int k) {
                double gene = src[(size_t)k * nth];
                double mom = 0.0;
                if ((active_mask >> k) & 1ull) {
                    double parent_gradient = p0d[k] * (1.0 - fmix) + p1d[k] * fmix;
                    mom = parent_gradient * (1.0 - 0.3) + (gene - p0g[k]) * 0.3;
                }
                dst[k] = gene, dst[M + k] = mom;
            };
            if (k0 < n_ops) take(k0);
            if (n_ops > 32 && k0 + 32 < n_ops) take(k0 + 32);  // (at most 64 ops)
        } else if (LEAN) {
            // not stored: the winner is re-derived from the counter RNG, lane k its op k (the operations of reproduce_children per gene;
            // one lane doing all of them was 7 % of a C3 step)
            BIOIK_FP_STRICT
            const int c = has_sec ? s_order[id - 2] : id - 2;
            const ChildX<PB> cx = make_child_x(pb, key, ctr1w, (uint32_t)c + 2u, p0g, p0d, p1d);
            auto derive = [&](int k) {
                const double gene = cx.template value<false>(k);  // (lane k its op k: the clip range differs from lane to lane)
                double mom = 0.0;
                if ((active_mask >> k) & 1ull) {
                    const double parent_gradient = p0d[k] * (1.0 - cx.fmix) + p1d[k] * cx.fmix;
                    mom = parent_gradient * (1.0 - 0.3) + (gene - p0g[k]) * 0.3;
                }
                dst[k] = gene, dst[M + k] = mom;
            };
            if (k0 < n_ops) derive(k0);
            if (n_ops > 32 && k0 + 32 < n_ops) derive(k0 + 32);  // (at most 64 ops)
        } else if (k0 == 0) {  // general flavour: quaternion genes are renormalised over the whole vector (reproduce_children)
            int c = has_sec ? s_order[id - 2] : id - 2;
            reproduce_child(pb, key, ctr1, (uint32_t)c + 2u, p0g, p0d, p1d, dst, 1, dst + M, 1);
        }
    }
    S.cur ^= 1;
    S.pf0 = first.f;
    S.pf1 = second.f;
    if constexpr (SLIM)
        if (gtid == 0) species_store(rank_now(), S);
    group_sync(G);
    PHASE_MARK(PH_SEL_COPY);
    group_sync(G);
    PHASE_MARK(PH_SEL_BAR);
}

// one generation of one species (ik_evolution_2.cpp:348-431)
template <class Frame>
BIOIK_DEV void solve_generation(Frame& F, SpeciesState& S, double*& popS, int rank_it, int step, int gen) {
    BIOIK_FRAME_NAMES(F);
    auto rank_now = [&]() { return F.rank_now(rank_it); };
    BIOIK_LANE_SCOPE;
    if constexpr (SLIM) S = species_load(rank_now()), popS = s_pop + S.slot * SP;
    const double* cb = popS + S.cur * BF;
    const double *p0g = cb, *p0d = cb + M, *p1d = cb + 3 * M;  // (pointers to const: re-derived after the walks under SLIM)
    const uint32_t gctr = (uint32_t)step * 16u + (uint32_t)gen;
    const uint32_t ctr1 = rng_ctr1(gctr, (uint32_t)S.id, RNG_REPRODUCE);
    int n_eval = lambda;
    uint64_t inside_mask = 0ull;  // (WAVE2: the ops no child of this generation can take out of AvoidJointLimitsGoal's free zone)
    if constexpr (DENSE || WAVE2 || JH) {
        // the two forms of the parents' mixed momentum (ChildT), lane k the column of op k, into the species' other elite buffer
        double* const pgt = popS + (S.cur ^ 1) * BF;
        bool inside = false;
        for (int k = gtid; k < n_ops; k += G) {
            const double d0 = p0d[k], d1 = p1d[k];
            const double pg0 = child_parent_gradient(d0, d1, 0), pg1 = child_parent_gradient(d0, d1, 1);
            pgt[k] = pg0, pgt[M + k] = pg1;
            if constexpr (WAVE2)
                inside = pb->ops[k].gene >= 0 && !pb->ops[k].unbounded && avoid_limits_surely_free(p0g[k], pg0, pg1, pb->ops[k].vmin, pb->ops[k].vmax, pb->ops[k].span);
        }
        if constexpr (WAVE2) inside_mask = has_sec ? p_ballot(inside) : 0ull;  // (a group is one wavefront and an op a lane: at most 64 ops)
        group_sync(G);
    }
    if (has_sec) solve_preselect(F, S, popS, step, gen, ctr1, inside_mask, n_eval);
    double b1f = P_INF, b2f = P_INF;
    int b1p = 0x7fffffff, b2p = 0x7fffffff;
    solve_walks(F, S, popS, rank_it, step, gen, ctr1, n_eval, b1f, b1p, b2f, b2p);
    solve_select_and_copy(F, S, popS, rank_it, step, gen, ctr1, n_eval, b1f, b1p, b2f, b2p);
}

template <class Frame>
BIOIK_DEV void solve_memetic(Frame& F, SpeciesState& S, double*& popS, int rank_it, int step) {
    BIOIK_FRAME_NAMES(F);
    auto rank_now = [&]() { return F.rank_now(rank_it); };
    // memetic phase on the elite (:436-570): finite-difference gradient of the linearised fitness, L1 normalisation, three-point
    // line search, clipped candidate, acceptance on primary fitness; up to 8 iterations.
    // One wavefront (the group's leading one; a half-wave group: its half) does the whole phase, and its lanes take three roles:
    //   op lane k        owns op k of the vectors involved (the elite, the support points x -+ g, the candidate) and their
    //                    displacements from the linearisation point, dv[k] = x[k] - base[k]
    //   component lane   (t, c) runs the first-order model for ONE component c of ONE tip frame t: the chain
    //                    F[t][c] = tipbase[t][c] + sum over the genes, in gene order, of delta[t][gene][c] * dv[gene]
    //                    -- the same fused multiply-adds in the same order as linear_tip, 1/7 of them per lane
    //   every lane       then reads the finished frames and evaluates the goals: lane i < D on the frame advanced by
    //                    delta[.][gene i] * dp (its gradient entry), lane D on the frame itself; the support points on even / odd
    //                    lanes; the candidate on all lanes alike (scalars of the line search are carried redundantly)
    // Hand-overs are LDS writes and reads of one wavefront in program order (p_wave_sync): no s_barrier inside the phase, so a
    // species stops as soon as a candidate is rejected, whatever the other species' wavefront is doing.
    if constexpr (SLIM) S = species_load(rank_now()), popS = s_pop + S.slot * SP;
    if (sp.memetic) {
        BIOIK_LANE_SCOPE;
        double* el = popS + S.cur * BF;  // the elite's genes, edited in place
        const XV xe{el, 1};
        if (exact) build_approximator<BIOIK_COOP_WALKS>(pb, xe, s_slots, s_frames, s_tips, s_delta, s_base, gtid, G, s_prefix);  // fresh linearisation at the elite
        PHASE_MARK(PH_MEM_APPROX);
        if (gtid < 64) {
            const int Gw = G < 64 ? G : 64;  // lanes at work
            double dp = 0.0000001;
            {
                uint32_t o0, o1;
                philox2x32_10(key, rng_ctr0(0, 0), rng_ctr1((uint32_t)step * 16u, (uint32_t)S.id, RNG_MEMETIC_SIGN), o0, o1);
                if (rng_uniform(o0, o1) < 0.5) dp = -dp;
            }
            const bool by_op = pb_flavour<PB>::general ? pb->genes_follow_ops != 0 : true;
            const int cnt = by_op ? n_ops : D;
            const int my_op = gtid < D ? pb->op_of_gene[gtid] : -1;  // lane i differentiates gene i, lane D holds the elite itself
            double* s_gop = s_gv;  // gradient in op order (zero for the ops that are not genes), next to the gene-ordered s_grad
            const int FB = 8 * T;
            double* s_x4 = s_xn;
            double* s_ex = s_bc;  // values exchanged between lanes: [0] primary, [1] all goals at the elite, [2] / [3] f(x - g) / f(x + g)
            // component lanes: one or two chains (same delta entries, two displacement vectors), four entries per trip
            auto chains = [&](const double* d0, double* f0, const double* d1, double* f1) {
                for (int idx = gtid; idx < FB; idx += Gw) {
                    const int t = idx >> 3, c = idx & 7;
                    if (c == 7) continue;
                    double a0 = s_tips[t * 7 + c], a1 = a0;
                    const double* dl = s_delta + (size_t)t * n_ops * 7 + c;
                    for (int g0 = 0; g0 < cnt; g0 += 4) {
                        double d[4], v0[4], v1[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int i2 = g0 + j < cnt ? g0 + j : cnt - 1;
                            const int kk = by_op ? i2 : pb->op_of_gene[i2];
                            const bool pad = g0 + j >= cnt;
                            d[j] = dl[(size_t)kk * 7];
                            v0[j] = pad ? 0.0 : d0[kk];
                            v1[j] = (pad || !d1) ? 0.0 : d1[kk];
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            a0 = BK_FMA(d[j], v0[j], a0);
                            if (d1) a1 = BK_FMA(d[j], v1[j], a1);
                        }
                    }
                    f0[idx] = a0;
                    if (d1) f1[idx] = a1;
                }
            };
            auto frame_of = [&](const double* fc, int t) { return f7_load(fc + t * 8); };
            // goal fitness of the lane's frames `fc` (+ its gene's delta * step): (primary, all goals); x: what joint-value goals read
            // (want_all: the sum over the secondary goals is wanted too -- the gradient and the support points; the candidate of round 2 is accepted on
            // its PRIMARY fitness alone (:527-538), and its secondary sum, a loop over every gene for a MinimalDisplacementGoal, was computed and dropped)
            // The secondary goals in the line search (secondary_fitness: the goals in their order, each a weighted sum): the sums over the joint values
            // -- MinimalDisplacementGoal, AvoidJointLimitsGoal and their kind, a term per op -- read vectors the lanes SHARE: the elite with the lane's
            // gene advanced (round 0) or one of the two support points (round 1).  Lane k computes the term of op k once, into `terms` (per vector:
            // tm), and every lane adds the terms up in their order with its own term in its place: the additions of goal_eval_joint_set_x, a
            // read and an add per op instead of the whole term.  (15 / 31 ops per sum: -35 % / -47 % of the phase's instructions on C3 / C4.)
            double* const s_tm = s_dv;  // [0, M): the terms of the first vector, [3 M, 4 M): of the second -- rows the round's chains have consumed / not yet written
            auto secondary_shared = [&](const PerturbX& xown, const PerturbX& x, const double* tm) -> double {  // xown: the lane's own vector; x: the shared ones
                double sum = 0.0;
                const F7 zero = F7{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
                for (int g = 0; g < pb->n_secondary; g++) {
                    const int type = pb->secondary[g].type;
                    double e;
                    if (joint_set_is_sum(type)) {
                        p_wave_sync();  // (the rows are free: the chains of this round, or the last goal's sums, have read them)
                        for (int k = gtid; k < n_ops; k += Gw) {
                            s_tm[k] = joint_set_term(pb, type, k, x.el[k], qc.seed);
                            if (x.el2) s_tm[3 * M + k] = joint_set_term(pb, type, k, x.el2[k], qc.seed);
                        }
                        p_wave_sync();
                        const double own = x.op >= 0 ? joint_set_term(pb, type, x.op, x.el[x.op] + x.step, qc.seed) : 0.0;
                        e = 0.0;
                        if (by_op) {
                            for (int k = 0; k < n_ops; k++) e += k == x.op ? own : tm[k];
                        } else {  // (the reference adds in the order of the genes, goal_eval_joint_set_x)
                            for (int i = 0; i < D; i++) {
                                const int k = pb->op_of_gene[i];
                                e += k == x.op ? own : tm[k];
                            }
                        }
                    } else {
                        e = goal_eval<false, PerturbX>(pb, type, pb->secondary[g].var_op, pb->secondary[g].var_seed, qc.par + pb->secondary[g].param_off, zero, xown, qc);
                    }
                    sum += e * pb->secondary[g].weight_sq;
                }
                return sum;
            };
            auto goals_on = [&](const double* fc, int dop, double dstep, const PerturbX& x, double& prim, double& all, bool want_all, const PerturbX& xsh, const double* tm) {
#if !defined(BIOIK_NO_POSE_ONLY)
                if (pb->pose_only) {  // one PoseGoal on one tip and nothing else: the same operations without the goal tables (0 + w² e = w² e)
                    F7 f = frame_of(fc, 0);
                    if (dstep != 0.0 && dop >= 0) {
                        const double* dl = s_delta + (size_t)dop * 7;
                        f = F7{{BK_FMA(dl[0], dstep, f.p.x), BK_FMA(dl[1], dstep, f.p.y), BK_FMA(dl[2], dstep, f.p.z)},
                               {BK_FMA(dl[3], dstep, f.q.x), BK_FMA(dl[4], dstep, f.q.y), BK_FMA(dl[5], dstep, f.q.z), BK_FMA(dl[6], dstep, f.q.w)}};
                    }
                    const double* P = qc.par + pb->pose_param_off;
                    double e = dist2(f.p, v3(P[0], P[1], P[2]));
                    const Q4 d = Q4{P[3] - f.q.x, P[4] - f.q.y, P[5] - f.q.z, P[6] - f.q.w};
                    const Q4 a = Q4{P[3] + f.q.x, P[4] + f.q.y, P[5] + f.q.z, P[6] + f.q.w};
                    const double rs = P[7];
                    e += fmin(qdot(d, d), qdot(a, a)) * (rs * rs);
                    prim = all = e * pb->pose_weight_sq;
                    return;
                }
#endif
                double acc = 0.0;
                V3 bal = v3(0.0, 0.0, 0.0);
                for (int t = 0; t < T; t++) {
                    F7 f = frame_of(fc, t);
                    if (dstep != 0.0) {
                        double d[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                        if (dop >= 0) {
                            const double* dl = s_delta + ((size_t)t * n_ops + dop) * 7;
                            for (int c = 0; c < 7; c++) d[c] = dl[c];
                        }
                        f = F7{{BK_FMA(d[0], dstep, f.p.x), BK_FMA(d[1], dstep, f.p.y), BK_FMA(d[2], dstep, f.p.z)},
                               {BK_FMA(d[3], dstep, f.q.x), BK_FMA(d[4], dstep, f.q.y), BK_FMA(d[5], dstep, f.q.z), BK_FMA(d[6], dstep, f.q.w)}};
                    }
                    acc = tip_goals(pb, t, f, x, qc, acc);
                    balance_tip(pb, t, f, bal);
                }
                acc = nonlink_primary(pb, x, qc, acc);
                acc += balance_cost(pb, bal, qc);
                prim = acc;
                if constexpr (DENSE) all = acc + 0.0;  // (the launcher gives the dense kernel problems without secondary goals only: the empty sum)
                else all = want_all ? acc + secondary_shared(x, xsh, tm) : acc;
            };
            for (int k = gtid; k < n_ops; k += Gw) s_gop[k] = 0.0;
            const bool odd = gtid & 1;
            bool descending = true;
            double f2p = 0.0, fa = 0.0;  // primary fitness / all goals at the elite, as the last gradient round left them
            for (int it = 0; it < 8 && descending; it++) {
                PHASE_COUNT(PH_N_MEM_ITER);
                // three rounds of the same shape -- op lanes prepare displacement vectors, component lanes run the chains, every
                // lane evaluates the goals on its frames -- written as one loop so that each piece of code exists once:
                //   round 0  gradient (:450-475): D + 1 evaluations, one per lane
                //   round 1  L1 normalisation (:477-482) and the two support points x - g (even lanes), x + g (odd lanes) (:485-495)
                //   round 2  step along the gradient (:498-568), clipped candidate, acceptance on primary fitness
                // Round 6: round 2 IS the next iteration's round 0.  An accepted candidate becomes the elite, and the gradient round that follows evaluates
                // that very vector (lane D) and the D vectors with one gene advanced by dp (lanes i < D) -- on the candidate's frames, which round 2 has
                // just built: the same displacements x4 - base, the same chains, the same goals.  So round 2 evaluates all D + 1 of them (a wavefront
                // instruction costs the same for one lane as for eight), lane D's primary fitness decides, and on acceptance the gradient of the next
                // iteration is already there: every iteration but the first is two rounds instead of three.  A rejected candidate's gradient is dropped
                // (the species stops).  Same operations on the same operands: the same bits.
                double fnorm = 0.0;
                bool nan_gene = false;  // (round 2: a gene of the candidate is not a number)
                for (int round = it == 0 ? 0 : 1; round < 3; round++) {
                    double* dv0 = s_dv + (round == 0 ? 0 : round == 1 ? 1 : 3) * M;
                    double* fc0 = (L.fc >= 0 ? s_fc : popS + (S.cur ^ 1) * BF) + (round == 0 ? 0 : round == 1 ? 1 : 3) * FB;  // (make_layout: fc_in_pop)
                    if (round == 0) {
                        for (int k = gtid; k < n_ops; k += Gw) dv0[k] = ((active_mask >> k) & 1ull) ? el[k] - s_base[k] : 0.0;
                    } else if (round == 1) {
                        double sum = dp * dp;
                        for (int i = 0; i < D; i++) sum += fabs(s_grad[i]);
                        fnorm = 1.0 / sum * dp;
                        PHASE_MARK(PH_MEM_NORM);
                        for (int k = gtid; k < n_ops; k += Gw) {
                            const double e = el[k], g = s_gop[k] * fnorm, b = s_base[k];
                            const bool on = (active_mask >> k) & 1ull;
                            const double xm = e - g, xp = e + g;
                            s_xm[k] = xm, s_xp[k] = xp;
                            dv0[k] = on ? xm - b : 0.0;
                            dv0[M + k] = on ? xp - b : 0.0;
                        }
                    } else {
                        const double f1 = s_ex[2], f3 = s_ex[3], f2 = fa;
                        double step_size;
                        if (sp.memetic == 'q') {  // :498-539
                            double v1 = f2 - f1, v2 = f3 - f2;
                            double v = (v1 + v2) * 0.5, aa = v1 - v2;
                            step_size = v / aa;
                        } else {  // 'l' :545-568
                            double cost_diff = (f3 - f1) * 0.5;
                            step_size = -(f2 / cost_diff);
                        }
                        // A step that is not a number: three equal support values make the quadratic step 0 / 0, the linear one f2 / 0 -- and 0 * inf for
                        // a gene the gradient does not move.  The reference's clip lets a NaN through (utils.h:328-333), its candidate's fitness is NaN and
                        // fails the comparison below: the search stops.  fmin / fmax would make the lower limit of a NaN (-DBL_MAX for a joint without
                        // limits) and the candidate a jump there; so a candidate with a NaN gene is no candidate.  (Where a goal hides the NaN -- max(0, .)
                        // of its error -- the literal reference ACCEPTS the NaN genes and can return them: quirk Q5)
                        // A step without bound -- v / 0 of a model without curvature -- puts a joint WITHOUT limits at its clip range's end, +-DBL_MAX
                        // (robot_info.h:109-113), where the linear model overflows; the literal reference may accept that vector and return it (quirk Q7,
                        // DESIGN.md section 3).  A candidate with a gene of magnitude 1e300 or more is no candidate either: the search stops.
                        bool nan_here = false;
                        for (int k = gtid; k < n_ops; k += Gw) {
                            const double e = el[k], gv = s_gop[k] * fnorm;
                            const bool on = (active_mask >> k) & 1ull;
                            const double raw = e + gv * step_size;
                            const bool is_nan = on && !(raw == raw);
                            const double x4 = (on && !is_nan) ? fmin(fmax(raw, s_clip[k]), s_clip[M + k]) : e;
                            nan_here = nan_here || is_nan || (on && fabs(x4) >= BIOIK_CANDIDATE_BOUND);
                            s_x4[k] = x4;
                            dv0[k] = on ? x4 - s_base[k] : 0.0;
                        }
                        nan_gene = (p_ballot(nan_here) & (G >= 64 ? ~0ull : 0xffffffffull << (tid & 32))) != 0ull;
                    }
                    p_wave_sync();
                    PHASE_MARK(PH_MEM_SUPPORT_COLS);
                    chains(dv0, fc0, round == 1 ? dv0 + M : nullptr, fc0 + FB);
                    p_wave_sync();
                    double vprim, vall;
                    // what joint-value goals read: the lane's own vector -- in the gradient round the elite with its gene advanced by dp
                    // (computed where it is read, no column), else the shared support point / candidate
                    const bool grad_round = round != 1;  // (the candidate's round too: lanes i < D advance gene i on the candidate's frames)
                    const PerturbX xq{round == 0 ? el : (round == 2 ? s_x4 : (odd ? s_xp : s_xm)), grad_round ? my_op : -1, grad_round ? dp : 0.0};
                    // (the vectors whose terms the lanes share: the elite / the candidate, or the two support points -- even lanes read the first's sums, odd lanes the second's)
                    const PerturbX xs{round == 0 ? el : (round == 2 ? s_x4 : s_xm), grad_round ? my_op : -1, grad_round ? dp : 0.0, round == 1 ? s_xp : nullptr};
                    goals_on(fc0 + ((round == 1 && odd) ? FB : 0), grad_round ? my_op : -1, grad_round ? dp : 0.0, xq, vprim, vall, true, xs,
                             s_tm + ((round == 1 && odd) ? 3 * M : 0));
                    PHASE_MARK(PH_MEM_SUPPORT_EVAL);
                    if (round == 0) {
                        if (gtid == D) s_ex[0] = vprim, s_ex[1] = vall;
                        p_wave_sync();
                        f2p = s_ex[0], fa = s_ex[1];
                        if (my_op >= 0) {
                            s_grad[gtid] = vall - fa;
                            s_gop[my_op] = vall - fa;
                        }
                        p_wave_sync();
                        PHASE_MARK(PH_MEM_GRAD);
                    } else if (round == 1) {
                        if (gtid < 2) s_ex[2 + gtid] = vall;
                        p_wave_sync();
                    } else {
                        if (gtid == D) s_ex[0] = vprim, s_ex[1] = vall;  // the candidate itself (the other lanes: the candidate with a gene advanced)
                        p_wave_sync();
                        const double cprim = s_ex[0], call = s_ex[1];
                        const bool accept = !nan_gene && cprim < f2p;  // accept iff the primary fitness improves, else stop (:527-538)
                        // (a half-wave group shares its wavefront with the other species: it stays in step with it and merely
                        // repeats the rejected iteration from its support points on, which changes nothing -- its elite, its gradient and the
                        // fitness values that belong to them stay as they are -- until the other species has stopped as well)
                        if (G >= 64 ? !accept : p_ballot(accept) == 0ull) descending = false;
                        if (accept) {
                            for (int k = gtid; k < n_ops; k += Gw) el[k] = s_x4[k];
                            f2p = cprim, fa = call;  // the candidate is the elite: what the gradient round of the next iteration would compute
                            if (my_op >= 0) {
                                s_grad[gtid] = vall - call;
                                s_gop[my_op] = vall - call;
                            }
                        }
                        p_wave_sync();
                        PHASE_MARK(PH_MEM_ACCEPT);
                    }
                }
            }
        }
        group_sync(G);
        PHASE_MARK(PH_MEM_TAIL);
    }
}

template <class Frame>
BIOIK_DEV void solve_rank(Frame& F, SpeciesState& S, double* popS, int rank_it) {
    BIOIK_FRAME_NAMES(F);
    auto rank_now = [&]() { return F.rank_now(rank_it); };
    auto group_check = [&](bool glead_, int gtid_, double* s_bc_, const XV& x_) { return F.group_check(glead_, gtid_, s_bc_, x_); };
    // species ranking fitness: exact FK of the elite (:607-614).  The same walk decides whether that elite satisfies the goals
    // (problem.cpp:259-341): if the species leads and improves on the solution, this elite IS the new solution, so the
    // island loop's success test (ik_parallel.h:173-181) needs no walk of its own.
    {
        BIOIK_LANE_SCOPE;
        const double* cb = popS + S.cur * BF;
        const FitCheck fc = group_check(glead, gtid, s_bc, XV{cb, 1});
        S.improved = (fc.fitness != S.fit) ? 1 : 0;
        S.fit = fc.fitness;
        S.pf0 = fc.fitness;
        S.ok = fc.ok;
        PHASE_MARK(PH_RANK);
    }
    {
        BIOIK_LANE_SCOPE;
        if (gtid == 0) species_store(rank_now(), S);
    }
}

// species management (:617-645), the solution's update and the island loop's checks at the end of a step (ik_parallel.h:160-181); true: the step loop ends
template <class Frame>
BIOIK_DEV bool solve_species_and_checks(Frame& F, int step) {
    BIOIK_FRAME_NAMES(F);
    int& steps = F.steps;
    bool &success = F.success, &expired = F.expired, &overtaken_out = F.overtaken_out, &drained = F.drained, &lost = F.lost;
    double& final_fit = F.final_fit;
    const int step_first = F.step_first, step_end = F.step_end;
    auto wg_check = [&](bool wlead_, int tid_, const XV& x_, double dpos_, double drot_, double dtwist_, int do_check_) { return F.wg_check(wlead_, tid_, x_, dpos_, drot_, dtwist_, do_check_); };
    wg_barrier();  // both species are ranked and their bookkeeping is in LDS
    BIOIK_LANE_SCOPE;  // species management and the checks at the end of the step

    // species management (:617-645)
    SpeciesState A = species_load(0), B = species_load(1);
    if (B.fit < A.fit) {
        SpeciesState tmp = A;
        A = B;
        B = tmp;
    }
    {
        uint32_t o0, o1;
        philox2x32_10(key, rng_ctr0(0, 0), rng_ctr1((uint32_t)step * 16u, (uint32_t)B.id, RNG_WIPEOUT), o0, o1);
        bool wipe = rng_uniform(o0, o1) < 0.1;
        wipe = wipe || !B.improved;
        if (sp.no_wipeout) wipe = false;
        if (wipe) {
            BIOIK_FP_STRICT
            const uint32_t wc1 = rng_ctr1((uint32_t)step * 16u, (uint32_t)B.id, RNG_WIPEOUT_GENE);
            double* cb = s_pop + B.slot * SP + B.cur * BF;
            wg_barrier();
            for (int k = tid; k < n_ops; k += nth) {
                double v = cb[k];
                if (pb->ops[k].gene >= 0) {
                    philox2x32_10(key, rng_ctr0(0, (uint32_t)pb->ops[k].gene), wc1, o0, o1);
                    v = rng_uniform(o0, o1) * (pb->ops[k].vmax - pb->ops[k].vmin) + pb->ops[k].vmin;
                }
                cb[k] = v, cb[M + k] = 0.0;
                cb[2 * M + k] = v, cb[3 * M + k] = 0.0;
            }
            wg_barrier();
            if (exact) B.pf0 = B.pf1 = wg_check(wlead, tid, XV{cb, 1}, 0.0, 0.0, 0.0, 0).fitness;
        }
    }
    steps++;
    PHASE_COUNT(PH_N_STEPS);
    PHASE_MARK(PH_SPECIES);
    const bool better = A.fit < s_solst[0];
    if (better) {
        const double* cb = s_pop + A.slot * SP + A.cur * BF;
        wg_barrier();
        for (int k = tid; k < n_ops; k += nth) s_sol[k] = cb[k];
    }
    wg_barrier();  // every lane has read the bookkeeping; lane 0 files the new ranking (and the new solution's figures) for the next step
    if (tid == 0) {
        species_store(0, A), species_store(1, B);
        if (better) s_solst[0] = A.fit, s_solst[1] = (double)A.ok;
    }
    wg_barrier();
    // ik_parallel.h:173-181: fitness and success test of the solution = those of the elite it was copied from (or of the seed)
    final_fit = s_solst[0];
    success = s_solst[1] != 0.0;
    PHASE_MARK(PH_CHECK);
    if constexpr (HELPED)
        if (lost) {  // (a rendezvous of this wavefront gave up: nothing it holds is a result)
            success = false, final_fit = BIOIK_DBL_MAX;
            return true;
        }
    if (success) {
        if (a.first_success && tid == 0) p_atomic_min(a.first_success + q, (unsigned int)steps);  // ik_parallel.h:176-177 `finished = 1`
        return true;
    }
    if (sp.timeout_ticks != 0ull || a.first_success || (a.resident && a.carry_list)) {  // at least one step has run (ik_parallel.h:160 `iteration != 0`)
        if (tid == 0) {
            bool stop = false;
            bool leave = false;  // the chip is emptying: on to the launch with the faster lone step (SolveArgs::resident)
            if (a.resident && a.carry_list && steps < step_end) {
                if (a.drain_below < 0) leave = steps - step_first >= 1 + (int)((((uint32_t)unit + 1u) * 2654435761u >> 16) % (uint32_t)(-a.drain_below));
                else leave = steps >= a.drain_min_steps && p_prefetched_word(s_prefix + 7) < (unsigned int)a.drain_below;
            }
            s_wbc[0] = leave ? 1.0 : 0.0;
            if (sp.timeout_ticks != 0ull) {
                const unsigned long long deadline = ((unsigned long long)s_deadline[0] << 32) | (unsigned long long)s_deadline[1];
                stop = p_wall_clock() >= deadline;
            }
            // ik_parallel.h:160 `!finished`: another island of the query has passed after no more steps than this one has run -- whatever this
            // island finds from here on, the selection will not look at it (lane 0 reads the word, the verdict crosses LDS like the clock's)
            const bool overtaken = a.first_success && p_atomic_load(a.first_success + q) <= (unsigned int)steps;
            s_wbc[2] = stop ? 1.0 : 0.0;
            s_wbc[3] = overtaken ? 1.0 : 0.0;
        }
        wg_barrier();
        expired = s_wbc[2] != 0.0;
        const bool overtaken = s_wbc[3] != 0.0;
        drained = s_wbc[0] != 0.0;
        wg_barrier();
        if (overtaken) overtaken_out = true;
        if (expired || overtaken) return true;
        if (drained) return true;
    }
    return false;
}

// the unit's state to the next launch (a hand-over), or its result; the islands' reduction by the query's last island (SolveArgs::island_done)
template <class Frame>
BIOIK_DEV void solve_epilogue(Frame& F) {
    BIOIK_FRAME_NAMES(F);
    int& steps = F.steps;
    bool &success = F.success, &expired = F.expired, &overtaken_out = F.overtaken_out, &drained = F.drained, &lost = F.lost;
    double& final_fit = F.final_fit;
    const int step_first = F.step_first, step_end = F.step_end;
    PHASE_DUMP(a.phase_cycles, unit);
    BIOIK_EPILOGUE_SCOPE_BEGIN
    if (a.resident && tid == 0) p_atomic_sub(my_resident(), (unsigned int)(HELPED ? 4 : nth >> 6));
    const bool handed_over = a.carry_list && !success && !expired && !overtaken_out && !lost && (step_end < sp.max_steps || drained);  // neither solved nor out of time: the next launch goes on
    if (handed_over) {
        double* c = a.carry + unit * (uint64_t)carry_n;
        for (int i = tid; i < 2 * BF; i += nth) {
            const int r = i >= BF ? 1 : 0;
            p_store_device(c + i, s_pop[(int)s_state[r * 8 + 4] * SP + (int)s_state[r * 8 + 5] * BF + (i - r * BF)]);  // (slot, cur of the species of rank r: species_store)
        }
        // (slot 16 of the bookkeeping, a broadcast slot between the steps: the step count, for a launch that continues every unit where it stands)
        for (int i = tid; i < M + 24; i += nth) p_store_device(c + 2 * BF + i, i < M ? s_sol[i] : ((i - M == 5 || i - M == 13) ? 0.0 : (i - M == 16 ? (double)steps : s_state[i - M])));
        if (tid == 0) p_store_device(a.carry_list + p_atomic_inc(a.carry_count), (int32_t)unit);
    }

    // result of this island; ranking fitness of ik_parallel.h:229-246
    double rank_fit = final_fit;
    if (success && has_sec) rank_fit = final_fit + secondary_fitness(pb, XV{s_sol, 1}, qc);
    double* out = a.solutions + unit * (uint64_t)V;
    if (!handed_over)  // (a handed-over unit's results are written by the launch that finishes it)
        for (int i = tid; i < V; i += nth) out[i] = s_seed[i];
    wg_barrier();
    if (steps > 0 && !handed_over)
        for (int k = tid; k < n_ops; k += nth)
            if (pb->ops[k].gene >= 0) out[pb->ops[k].var] = s_sol[k];
    if (tid == 0 && !handed_over) {
        a.fitness[unit] = rank_fit;
        a.success[unit] = success ? 1 : 0;
        a.steps[unit] = steps;
    }
    if constexpr (HELPED) p_flag_store((unsigned int*)(lds + L.help) + 2 + p_wave_index(), 0xffffffffu);  // this wavefront's helper may leave
    if (a.island_done) {  // (SolveArgs::island_done: the query's last island to file its result reduces the islands; set for solves in ONE launch: nothing is handed over)
        p_fence_device();  // this lane's part of the result is visible to the whole device ...
        wg_barrier();      // ... and so is every other lane's
        if (tid == 0) s_wbc[0] = p_atomic_inc(a.island_done + q) + 1u == (unsigned int)sp.islands ? 1.0 : 0.0;
        wg_barrier();
        if (tid < 64 && s_wbc[0] != 0.0) {  // the workgroup's first wavefront, all of its lanes (select_coop)
            p_fence_device();
            SelectArgs sa;
            sa.islands = sp.islands, sa.V = V, sa.sync = sp.island_sync, sa.pad = 0, sa.n = q + 1;
            sa.isl_solutions = a.solutions, sa.isl_fitness = a.fitness, sa.isl_success = a.success, sa.isl_steps = a.steps;
            sa.solutions = a.final_solutions, sa.fitness = a.final_fitness, sa.success = a.final_success, sa.steps = a.final_steps;
            select_coop(sa, q, tid);
            if (tid == 0) {
                p_store_device(a.island_done + q, 0u);
                if (a.first_success) p_store_device(a.first_success + q, 0xffffffffu);
            }
        }
    }
    BIOIK_EPILOGUE_SCOPE_END
}

template <bool LEAN, bool CL = false, bool JOINT = false, bool SLIM = false, int FIXED = 0>
BIOIK_DEV void solve_body(const SolveArgs& a, uint64_t unit_in, double* lds) {
    typedef SolveFrame<LEAN, CL, JOINT, SLIM, FIXED> Frame;
    Frame F(a, lds);
    if (!solve_setup(F, unit_in)) return;
    if constexpr (Frame::HELPED)
        if (F.tid0 >= 128) {  // (the helped kernel's wavefronts 2 and 3: they walk half of a generation's children for wavefronts 0 and 1 and do nothing else)
            solve_helper(F);
            return;
        }
    solve_init(F);
    for (int step = F.step_first; step < F.step_end; step++) {
        // the count of its XCD's wavefronts, asked for HERE and read when the step is over (SolveArgs::resident): the load goes straight to LDS (the spare
        // eighth double of the prefix frame) and nothing waits for it -- read where it is used, every workgroup stalled on it once per step: -13 % on a stream
        // of ten solves in flight (profiles/r04_drain_handover.log)
        if (a.resident && a.carry_list && a.drain_below > 0 && (Frame::HALVES ? p_lane_fresh() : p_tid_fresh()) == 0) p_prefetch_word_to_lds(F.my_resident(), F.s_prefix + 7);
        // (species-parallel: a lane group runs the species of its own number; its loop below is one trip with a per-lane index)
        const int rank_begin = F.groups == 2 ? (SLIM ? 0 : (F.g_shift >= 0 ? p_fresh(F.tid0) >> F.g_shift : p_fresh(F.tid0) / F.G)) : 0, rank_end = F.groups == 2 ? rank_begin + 1 : 2;
        for (int rank_it = rank_begin; rank_it < rank_end; rank_it++) {
            SpeciesState S = F.species_load(F.rank_now(rank_it));
            double* popS = F.s_pop + S.slot * (8 * (F.n_ops > 0 ? F.n_ops : 1));  // (the species' two elite buffers: [2 buffers][2 individuals][genes | momentum][op])
            solve_linearise(F, S, popS, rank_it);
            for (int gen = 0; gen < a.sp.generations; gen++) solve_generation(F, S, popS, rank_it, step, gen);
            solve_memetic(F, S, popS, rank_it, step);
            solve_rank(F, S, popS, rank_it);
        }
        if (solve_species_and_checks(F, step)) break;
    }
    solve_epilogue(F);
}

// ---------------------------------------------------------------------------------------------------------
// function-level kernels: one lane per genotype, same device functions as the solver
// ---------------------------------------------------------------------------------------------------------
struct EvalArgs {
    ProbPtr pb;
    uint64_t n;             // genotypes (eval kernels) / population (reproduce)
    const double* seed;     // [V]  one query
    const double* params;   // [P]
    const double* genes;    // [n][D] problem gene order
    const double* base;     // [D] base genes of the linearisation (linear fitness / approximator)
    double* out0;           // fk: tips [n][T][7]; fitness: primary [n]; approximator: tip frames [T][7]; reproduce: genes [n][D]
    double* out1;           // fitness: secondary [n]; approximator: deltas [T][D][7]; reproduce: momentum [n][D]
    int32_t* outi;          // check: ok [n]
    double dpos, drot, dtwist;
    uint32_t rng_key, rng_ctr1;
    int32_t fk_mode;
    int32_t pad;
};

BIOIK_DEV void load_query(const EvalArgs& a, double* s_seed, double* s_par) {
    const int tid = p_tid(), nth = p_nthreads();
    for (int i = tid; i < a.pb->V; i += nth) s_seed[i] = a.seed[i];
    for (int i = tid; i < a.pb->P; i += nth) s_par[i] = a.params ? a.params[i] : 0.0;
    p_barrier();
}
// op-ordered genotype from problem-ordered genes (or the seed when genes is null) into dst (stride ds)
BIOIK_DEV void load_genotype(ProbPtr pb, const double* s_seed, const double* genes, double* dst, int ds) {
    const int n_ops = pb->n_ops;
    for (int k = 0; k < n_ops; k++) dst[(size_t)k * ds] = (pb->ops[k].gene >= 0 && genes) ? genes[pb->ops[k].gene] : s_seed[pb->ops[k].var];
}

// RobotFK_Fast_Base::applyConfiguration + getTipFrames for n genotypes
BIOIK_DEV void eval_fk_body(const EvalArgs& a, uint64_t block, double* lds) {
    ProbPtr pb = a.pb;
    const int tid = p_tid(), nth = p_nthreads();
    const LdsLayout L = make_layout(pb->n_ops, pb->V, pb->P, pb->T, pb->n_slots, nth, 0, 0);
    load_query(a, lds + L.seed, lds + L.par);
    uint64_t i = block * (uint64_t)nth + tid;
    if (i >= a.n) return;
    double* xcol = lds + L.xcol + tid;
    load_genotype(pb, lds + L.seed, a.genes + i * pb->D, xcol, nth);
    const int T = pb->T;
    fk_walk(pb, XV{xcol, nth}, lds + L.slots, nullptr, [&](int t, const F7& f) { f7_store(a.out0 + (i * T + pb->tips[t].out_index) * 7, f); });
}

// Problem::computeGoalFitness on exact or linear phenotypes; every block of the linear variant rebuilds the tables itself
BIOIK_DEV void eval_fitness_body(const EvalArgs& a, uint64_t block, double* lds) {
    ProbPtr pb = a.pb;
    const int tid = p_tid(), nth = p_nthreads();
    const LdsLayout L = make_layout(pb->n_ops, pb->V, pb->P, pb->T, pb->n_slots, nth, 0, 0);
    load_query(a, lds + L.seed, lds + L.par);
    const QueryCtx qc{lds + L.seed, lds + L.par};
    if (a.fk_mode == FK_LINEAR) {
        // (the base configuration sits in the solution's slot: the line-search vectors lie under the frame chain, make_layout)
        if (tid == 0) load_genotype(pb, lds + L.seed, a.base, lds + L.sol, 1);
        p_barrier();
        build_approximator(pb, XV{lds + L.sol, 1}, lds + L.slots, lds + L.g_first + L.frames, lds + L.g_first + L.tips, lds + L.g_first + L.delta,
                           lds + L.g_first + L.base, tid, nth);
    }
    const LinModel lm{lds + L.g_first + L.tips, lds + L.g_first + L.delta, lds + L.g_first + L.base};
    uint64_t i = block * (uint64_t)nth + tid;
    if (i >= a.n) return;
    double* xcol = lds + L.xcol + tid;
    load_genotype(pb, lds + L.seed, a.genes + i * pb->D, xcol, nth);
    const XV xl{xcol, nth};
    a.out0[i] = a.fk_mode == FK_LINEAR ? eval_linear_primary(pb, xl, qc, lm) : eval_exact_primary(pb, xl, qc, lds + L.slots);
    a.out1[i] = secondary_fitness(pb, xl, qc);
}

// RobotFK_Mutator::initializeMutationApproximator: tables in the public (tip, gene) order; 1 block
BIOIK_DEV void eval_approximator_body(const EvalArgs& a, double* lds) {
    ProbPtr pb = a.pb;
    const int tid = p_tid(), nth = p_nthreads();
    const LdsLayout L = make_layout(pb->n_ops, pb->V, pb->P, pb->T, pb->n_slots, nth, 0, 0);
    load_query(a, lds + L.seed, lds + L.par);
    if (tid == 0) load_genotype(pb, lds + L.seed, a.base, lds + L.sol, 1);
    p_barrier();
    build_approximator(pb, XV{lds + L.sol, 1}, lds + L.slots, lds + L.g_first + L.frames, lds + L.g_first + L.tips, lds + L.g_first + L.delta,
                           lds + L.g_first + L.base, tid, nth);
    const int T = pb->T, D = pb->D, n_ops = pb->n_ops;
    for (int idx = tid; idx < T * 7; idx += nth) {
        int t = idx / 7, c = idx - t * 7;
        a.out0[pb->tips[t].out_index * 7 + c] = lds[L.g_first + L.tips + idx];
    }
    for (int idx = tid; idx < T * D * 7; idx += nth) {
        int t = idx / (D * 7), rem = idx - t * D * 7, g = rem / 7, c = rem - g * 7;
        a.out1[((size_t)pb->tips[t].out_index * D + g) * 7 + c] = lds[L.g_first + L.delta + ((size_t)t * n_ops + pb->op_of_gene[g]) * 7 + c];
    }
}

// IKEvolution2::reproduce for `n` children of one (species, generation)
BIOIK_DEV void eval_reproduce_body(const EvalArgs& a, uint64_t block, double* lds) {
    ProbPtr pb = a.pb;
    const int tid = p_tid(), nth = p_nthreads(), D = pb->D, n_ops = pb->n_ops;
    const int M = n_ops > 0 ? n_ops : 1;
    double* s_par = lds;  // [2][2][M] parents, op-indexed: a.genes is [parent][genes|momentum][gene]
    for (int idx = tid; idx < 4 * M; idx += nth) {
        int k = idx % M, which = idx / M;
        s_par[idx] = (k < n_ops && pb->ops[k].gene >= 0) ? a.genes[which * D + pb->ops[k].gene] : 0.0;
    }
    p_barrier();
    uint64_t c = block * (uint64_t)nth + tid;
    if (c >= a.n) return;
    double* xcol = lds + 4 * M + tid;            // [M][nth]
    double* gcol = lds + 4 * M + M * nth + tid;  // [M][nth]
    reproduce_child(pb, a.rng_key, a.rng_ctr1, (uint32_t)c + 2u, s_par, s_par + M, s_par + 3 * M, xcol, nth, gcol, nth);
    for (int k = 0; k < n_ops; k++)
        if (pb->ops[k].gene >= 0) {
            a.out0[c * D + pb->ops[k].gene] = xcol[(size_t)k * nth];
            a.out1[c * D + pb->ops[k].gene] = gcol[(size_t)k * nth];
        }
}

// Problem::checkSolutionActiveVariables on the exact-FK pose of n genotypes
BIOIK_DEV void eval_check_body(const EvalArgs& a, uint64_t block, double* lds) {
    ProbPtr pb = a.pb;
    const int tid = p_tid(), nth = p_nthreads();
    const LdsLayout L = make_layout(pb->n_ops, pb->V, pb->P, pb->T, pb->n_slots, nth, 0, 0);
    load_query(a, lds + L.seed, lds + L.par);
    const QueryCtx qc{lds + L.seed, lds + L.par};
    uint64_t i = block * (uint64_t)nth + tid;
    if (i >= a.n) return;
    double* xcol = lds + L.xcol + tid;
    load_genotype(pb, lds + L.seed, a.genes + i * pb->D, xcol, nth);
    FitCheck fc = exact_fitness_check(pb, XV{xcol, nth}, qc, lds + L.slots, a.dpos, a.drot, a.dtwist, 1);
    a.outi[i] = fc.ok;
    if (a.out0) a.out0[i] = fc.fitness;
}

// ---------------------------------------------------------------------------------------------------------
// The shared arithmetic headers one function at a time (bioik_sincos.h, bioik_fused.h): what the kernels AND the CPU checker's "device arithmetic"
// mode both include is evaluated here on the device for arbitrary arguments, so that a test can hold it against an independent high-precision
// reference (tests/test_arith_headers.py) -- bit parity between two users of one header cannot show a defect of the header itself.
// ---------------------------------------------------------------------------------------------------------
enum { ARITH_SINCOS = 0, ARITH_QROT = 1, ARITH_QMUL = 2, ARITH_DOT3 = 3, ARITH_DOT4 = 4, ARITH_REVOLUTE = 5, ARITH_ACOS = 6, ARITH_ATAN2 = 7 };
struct ArithArgs {
    int32_t op, pad;
    uint64_t n;
    const double* in;  // [n][arith_in(op)]
    double* out;       // [n][arith_out(op)]
};
BIOIK_HD int arith_in(int op) { return op == ARITH_SINCOS ? 1 : op == ARITH_QROT ? 7 : op == ARITH_QMUL ? 8 : op == ARITH_DOT3 ? 6 : op == ARITH_DOT4 ? 8 : op == ARITH_REVOLUTE ? 22 : op == ARITH_ACOS ? 1 : op == ARITH_ATAN2 ? 2 : 0; }
BIOIK_HD int arith_out(int op) { return op == ARITH_SINCOS ? 2 : op == ARITH_QROT ? 3 : op == ARITH_QMUL ? 4 : op == ARITH_DOT3 ? 1 : op == ARITH_DOT4 ? 1 : op == ARITH_REVOLUTE ? 14 : op == ARITH_ACOS ? 1 : op == ARITH_ATAN2 ? 1 : 0; }
BIOIK_DEV void arith_body(const ArithArgs& a, uint64_t i) {
    if (i >= a.n) return;
    const double* x = a.in + i * (uint64_t)arith_in(a.op);
    double* o = a.out + i * (uint64_t)arith_out(a.op);
    if (a.op == ARITH_SINCOS) {
        p_sincos(x[0], &o[0], &o[1]);
    } else if (a.op == ARITH_ACOS) {
        o[0] = bioik_acos(x[0]);
    } else if (a.op == ARITH_ATAN2) {
        o[0] = bioik_atan2(x[0], x[1]);
    } else if (a.op == ARITH_QROT) {
        const V3 r = qrot(Q4{x[0], x[1], x[2], x[3]}, v3(x[4], x[5], x[6]));
        o[0] = r.x, o[1] = r.y, o[2] = r.z;
    } else if (a.op == ARITH_QMUL) {
        const Q4 r = qmul(Q4{x[0], x[1], x[2], x[3]}, Q4{x[4], x[5], x[6], x[7]});
        o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = r.w;
    } else if (a.op == ARITH_DOT3) {
        o[0] = dot3(v3(x[0], x[1], x[2]), v3(x[3], x[4], x[5]));
    } else if (a.op == ARITH_DOT4) {
        o[0] = qdot(Q4{x[0], x[1], x[2], x[3]}, Q4{x[4], x[5], x[6], x[7]});
    } else if (a.op == ARITH_REVOLUTE) {
        // one revolute joint applied to a frame: in = frame[7], half angle, cpos[3], ca[4], cb[4], pos_kind, rot_kind, (pad); out = the general form's
        // frame [7], then the frame of the sparse form named by (pos_kind, rot_kind) [7] -- equal wherever the struck-out constants are exact zeros
        F7 g[1] = {f7_load(x)}, sp[1] = {f7_load(x)};
        double sn[1], cs[1];
        p_sincos(x[7], &sn[0], &cs[0]);
        const RevConst kg{x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15], x[16], x[17], x[18], BIOIK_POS_GENERAL, BIOIK_ROT_GENERAL};
        const RevConst ks{x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15], x[16], x[17], x[18], (int)x[19], (int)x[20]};
        revolute_apply<1>(g, sn, cs, kg);
        revolute_apply<1>(sp, sn, cs, ks);
        f7_store(o, g[0]);
        f7_store(o + 7, sp[0]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// streamed generation: the population genotype array is resident in HBM, genes [unit][D][population]
// (individual index fastest: a wavefront's 64 loads of one gene are one contiguous 512-byte segment).
// Block b serves `nthreads` individuals of unit b / blocks_per_unit.
// ---------------------------------------------------------------------------------------------------------
struct StreamArgs {
    ProbPtr pb;
    uint64_t n_units;
    int32_t population, blocks_per_unit;
    const double* seeds;   // [n_units][V]
    const double* params;  // [n_units][P]
    const double* genes;   // [n_units][D][population]
    double* fitness;       // [n_units][population]
};
BIOIK_DEV void stream_fitness_body(const StreamArgs& a, uint64_t block, double* lds) {
    ProbPtr pb = a.pb;
    const int tid = p_tid(), nth = p_nthreads();
    const int V = pb->V, P = pb->P, D = pb->D, n_ops = pb->n_ops;
    const LdsLayout L = make_layout(n_ops, V, P, pb->T, pb->n_slots, nth, 0, 0);
    const uint64_t unit = block / (uint64_t)a.blocks_per_unit;
    const int ind = (int)(block % (uint64_t)a.blocks_per_unit) * nth + tid;
    for (int i = tid; i < V; i += nth) lds[L.seed + i] = a.seeds[unit * V + i];
    for (int i = tid; i < P; i += nth) lds[L.par + i] = a.params[unit * P + i];
    p_barrier();
    if (ind >= a.population) return;
    const QueryCtx qc{lds + L.seed, lds + L.par};
    const double* g = a.genes + unit * (uint64_t)D * a.population + ind;
    double* xcol = lds + L.xcol + tid;
    for (int k = 0; k < n_ops; k++) xcol[(size_t)k * nth] = pb->ops[k].gene >= 0 ? g[(uint64_t)pb->ops[k].gene * a.population] : lds[L.seed + pb->ops[k].var];
    a.fitness[unit * (uint64_t)a.population + ind] = eval_exact_primary(pb, XV{xcol, nth}, qc, lds + L.slots);
}
