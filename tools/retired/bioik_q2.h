// bioik_q2.h — k_solve_lean_q2: TWO queries per wavefront, each species on a QUARTER of it (round 6).
//
// RETIRED (round 6; never part of a shipped library): built, bit-identical to the oracle on the host simulator and on the GPU, measured, and taken out again.
// What it measured (profiles/r06_q2_two_queries_per_wavefront.log): on FIXED work with the chip oversubscribed four times (16384 queries in one launch) 20.8e3
// against the dense kernel's 18.7e3 steps per ms (+11 %); with 8192 queries +4 %; with the 4096 queries of a bench batch -30 % (half the wavefronts).  On the
// bench's stream of ten 4096-query batches 1.06 ... 1.21e6 solves/s against 1.40e6: a launch has no more units than the chip has SLOTS, so the persistent
// slots have nothing to pull, a wavefront runs at half its lanes from the step at which the faster of its two queries ends, and handing the slower one to the
// stragglers' kernel moves half of all queries there.  The same holds for any mapping that puts several queries into one persistent workgroup (the four-wavefront
// workgroup with its single-individual phases on one wavefront): it needs a pool of units several times the chip's residency per launch.
// One memory access fault -- in one of the two GPU runs with one wavefront per two units AND the hand-over; nine other runs were clean -- was not chased.
// How it was wired: SolveArgs carried `unsigned int* next_unit; uint64_t n_units;`, bioik_hip.hip had
//     __global__ void __launch_bounds__(64, 4) k_solve_lean_q2(SolveArgs a) { extern __shared__ double lds[]; solve_q2_body<LeanProbPtr>(a, blockIdx.x, lds); }
// and launch_solve's `dense_launch` branch launched it on (units + 2 u - 1) / (2 u) wavefronts with 2 * q2_slot_doubles(L) * 8 bytes of LDS and a zeroed counter
// word of the stream's scratch (127 VGPRs, no spilled register).
//
// The dense mapping of the throughput schedule (k_solve_lean_cl64w4, rounds 3 - 5) gives a query one wavefront: a species per half, the 128 children of a
// generation two per lane and trip.  Its chain walks fill all 64 lanes; everything else -- the two best of a generation, the winners' copy, the memetic
// phase's line search (D + 1 = 8 evaluations per species), the species' ranking walk, the species management -- is work for a handful of lanes that costs the
// wavefront a whole instruction each time: ~30 % of its vector instructions on the 7-joint arm.  This kernel runs TWO queries on the wavefront: lanes
// 0 ... 31 = slot 0, lanes 32 ... 63 = slot 1, a species per row of 16 lanes, eight children per lane and generation (four pair walks).  The walks cost a
// query what they cost before; the phases that evaluate one individual are paid once for two queries.  A row of 16 lanes is a DPP row: the reductions of a
// species group are row operations, nothing crosses the LDS crossbar.
//
// The slots are PERSISTENT: a slot whose query ends takes the launch's next unit from an atomic counter (SolveArgs::next_unit) and initialises it while its
// neighbour goes on -- a wavefront that waited for the slower of its two queries would give back what the sharing gains (queries take 3 ... 64 steps).  The
// two slots step in lock step -- the control flow is the wavefront's; what differs per slot is data: the unit, its step count, its random stream, its block
// of LDS -- and a slot without a unit runs along on whatever its block holds, with its stores to device memory switched off.  When the counter has run dry, a
// unit whose neighbour slot is idle leaves for the next launch once it has run SolveArgs::drain_min_steps steps (the hand-over of solve_body: the stragglers
// continue under k_solve_lean_cl4h, whose lone step is a third shorter) instead of running on at half a wavefront.
//
// Written as phase functions over one lane scope (Q2_SCOPE) -- the structure solve_body never got.  Same arithmetic, same random streams, same results as
// every other mapping, bit for bit (tests/test_hostsim_parity.py, tests/test_gpu_parity.py run it against the oracle and against the other kernels).
// What the launcher guarantees (launch_solve): lean flavour, serial chain, exact FK, no secondary goal, 128 ... 256 children per species, at most 15 genes and
// 16 chain ops (lane D of a row holds the elite, lane k the trigonometry of op k), children computed where they are read.
//
// Reference behaviour restated here: src/ik_evolution_2.cpp:111-230 (initialize), :328-646 (step), src/ik_parallel.h:148-190 (island loop, budget form).
#pragma once
#include "bioik_kernels.h"

// the least key of a row of 16 lanes, known to all of them: four DPP steps (half_min_u64 without the hop across rows)
BIOIK_DEV unsigned long long row_min_u64(unsigned long long k) {
    unsigned long long o = p_row_mirror<0>(k);
    k = o < k ? o : k;
    o = p_row_mirror<1>(k), k = o < k ? o : k;
    o = p_quad_xor<2>(k), k = o < k ? o : k;
    o = p_quad_xor<1>(k), k = o < k ? o : k;
    return k;
}

struct Q2Species {  // SpeciesState of solve_body, in the same eight slots of the bookkeeping block
    double fit, pf0, pf1;
    int id, slot, cur, improved, ok;
};
BIOIK_DEV Q2Species q2_species_load(const double* s_state, int r) {
    const double* d = s_state + r * 8;
    return Q2Species{d[0], d[1], d[2], (int)d[3], (int)d[4], (int)d[5], (int)d[6], (int)d[7]};
}
BIOIK_DEV void q2_species_store(double* s_state, int r, const Q2Species& S) {
    double* d = s_state + r * 8;
    d[0] = S.fit, d[1] = S.pf0, d[2] = S.pf1, d[3] = (double)S.id, d[4] = (double)S.slot, d[5] = (double)S.cur, d[6] = (double)S.improved, d[7] = (double)S.ok;
}

// LDS of one slot: solve_body's layout of the dense mapping (so that a unit's state between two steps is the block SolveArgs::carry describes), and behind it
// eight numbers of the slot's own: [0] the unit, [1] its random key, [2] the steps it has run
BIOIK_HD LdsLayout q2_layout(int n_ops, int V, int P, int T, int lambda) { return make_layout(n_ops, V, P, T, 0, 64, lambda, 0, 0, 2, 2, 1, 1, 0); }
BIOIK_HD int q2_slot_doubles(const LdsLayout& L) { return L.total + 8; }

// The lane numbers of a phase and everything derived from them, from a FRESH copy of the lane number (p_lane_fresh): nothing of it lives across a chain walk.
//   slot = lane / 32 (the query), sg = the row of 16 inside the slot (the species group), gtid = the lane inside its row, tid32 = the lane inside its slot
#define Q2_SCOPE                                                                                   \
    const int lane = p_lane_fresh();                                                               \
    const int slot = lane >> 5, sg = (lane >> 4) & 1, gtid = lane & 15, tid32 = lane & 31;         \
    double* const sl = lds + slot * slot_stride; /* this slot's block */                           \
    double* const s_seed = sl + L.seed;                                                            \
    double* const s_par = sl + L.par;                                                              \
    double* const s_pop = sl + L.pop;                                                              \
    double* const s_sol = sl + L.sol;                                                              \
    double* const s_prefix = sl + L.prefix;                                                        \
    double* const s_state = sl + L.state;                                                          \
    double* const s_clip = sl + L.clip;                                                            \
    double* const s_ctl = sl + L.total;                                                            \
    double* const gbase = sl + L.g_first + sg * L.g_stride; /* this species group's scratch */     \
    const QueryCtx qc{s_seed, s_par};                                                              \
    const unsigned long long rowmask = 0xffffull << (lane & 48) /* the lanes of this row */

template <class PB>
struct Q2 {
    PB pb;
    const SolveArgs& a;
    double* lds;
    LdsLayout L;
    int slot_stride, M, SP, BF, n_ops, D, T, V, P, lambda;
};

// ik_evolution_2.cpp:129-179 for the slots that have just taken a unit (`got`): solution = seed, 2 species x 2 clones of the seed, zero momentum; the seed's
// fitness and whether it already satisfies the goals.  The other slot executes the same instructions with its stores switched off.
template <class PB>
BIOIK_DEV void q2_init(const Q2<PB>& c, bool got, unsigned int unit) {
    const PB pb = c.pb;
    const SolveArgs& a = c.a;
    const DevSolveParams& sp = a.sp;
    double* const lds = c.lds;
    const LdsLayout& L = c.L;
    const int slot_stride = c.slot_stride, M = c.M, SP = c.SP, n_ops = c.n_ops, V = c.V, P = c.P;
    Q2_SCOPE;
    const uint64_t q = (uint64_t)unit / (uint64_t)sp.islands;
    const uint32_t island = (uint32_t)((uint64_t)unit % (uint64_t)sp.islands);
    if (got) {
        for (int i = tid32; i < V; i += 32) s_seed[i] = a.seeds[q * V + i];
        for (int i = tid32; i < P; i += 32) s_par[i] = a.params[q * P + i];
        if (tid32 == 0) s_ctl[0] = (double)unit, s_ctl[1] = (double)rng_query_key(sp.random_seed, sp.first_query + q, island), s_ctl[2] = 0.0;
    }
    p_wave_sync();
    if (got) {
        for (int k = tid32; k < n_ops; k += 32) {
            const double v = s_seed[pb->ops[k].var];
            for (int s = 0; s < 2; s++)
                for (int i = 0; i < 2; i++) {
                    double* d = s_pop + s * SP + i * 2 * M;
                    d[k] = v, d[M + k] = 0.0;
                }
            s_sol[k] = v;
            s_clip[k] = pb->ops[k].clip_min, s_clip[M + k] = pb->ops[k].clip_max;
        }
    }
    p_wave_sync();
    if (pb->n_prefix > 0) {  // the joints in front of the first gene see the seed in every individual: walked once per unit
        if (got && tid32 == 0) f7_store(s_prefix, fk_prefix(pb, XV{s_sol, 1}));
        p_wave_sync();
    }
    const FitCheck fc0 = exact_fitness_check<32>(pb, XV{s_sol, 1}, qc, lds, sp.dpos, sp.drot, sp.dtwist, 1, s_prefix);  // (the slot's 32 lanes share the walk)
    if (got && tid32 == 0) {
        q2_species_store(s_state, 0, Q2Species{P_INF, fc0.fitness, fc0.fitness, 0, 0, 0, 0, 0});
        q2_species_store(s_state, 1, Q2Species{P_INF, fc0.fitness, fc0.fitness, 1, 1, 0, 0, 0});
        s_state[20] = fc0.fitness, s_state[21] = (double)fc0.ok;
    }
    p_wave_sync();
}

// One generation of one species group (ik_evolution_2.cpp:348-431): the table of the parents' mixed momentum, the children's walks two at a time, the two
// best of the generation from keys, the winners into the species' other elite buffer.
template <class PB>
BIOIK_DEV void q2_generation(const Q2<PB>& c, int gen) {
    const PB pb = c.pb;
    const SolveArgs& a = c.a;
    double* const lds = c.lds;
    const LdsLayout& L = c.L;
    const int slot_stride = c.slot_stride, M = c.M, SP = c.SP, BF = c.BF, n_ops = c.n_ops, lambda = c.lambda;
    const uint64_t active_mask = pb->active_mask;
    {
        Q2_SCOPE;
        const Q2Species S = q2_species_load(s_state, sg);
        double* const popS = s_pop + S.slot * SP;
        const double* const cb = popS + S.cur * BF;
        double* const pgt = popS + (S.cur ^ 1) * BF;  // the two forms of the parents' mixed momentum (ChildT), lane k the column of op k
        for (int k = gtid; k < n_ops; k += 16) {
            const double d0 = cb[M + k], d1 = cb[3 * M + k];
            pgt[k] = child_parent_gradient(d0, d1, 0), pgt[M + k] = child_parent_gradient(d0, d1, 1);
        }
        p_wave_sync();
    }
    // genotype -> phenotype -> fitness (:391-407): lane r of the row walks the children r and r + 16 (+ 32 j); the fitness values wait in LDS (fit_park)
    for (int r0 = 0; r0 < lambda; r0 += 32) {
        double f[2];
        {
            Q2_SCOPE;
            const int r = r0 + gtid, r1 = r + 16;
            const int ra = r < lambda ? r : 0, rb = r1 < lambda ? r1 : ra;  // (a lane without a child in this trip walks a copy and drops it)
            const Q2Species S = q2_species_load(s_state, sg);
            const double* const cb = s_pop + S.slot * SP + S.cur * BF;
            const double* const pgt = s_pop + S.slot * SP + (S.cur ^ 1) * BF;
            const uint32_t key = (uint32_t)s_ctl[1];
            const uint32_t ctr1 = rng_ctr1((uint32_t)s_ctl[2] * 16u + (uint32_t)gen, (uint32_t)S.id, RNG_REPRODUCE);
            const ChildT<PB> cx[2] = {make_child_t(pb, key, ctr1, (uint32_t)ra + 2u, cb, pgt, M), make_child_t(pb, key, ctr1, (uint32_t)rb + 2u, cb, pgt, M)};
            eval_exact_primary_n<2, true, true>(pb, cx, qc, lds, 0, f, s_prefix);
        }
        Q2_SCOPE;
        const int r = r0 + gtid, r1 = r + 16;
        if (pb->n_link_primary < pb->n_primary) {  // (primary goals over the joint values: the accessors are built again behind the walk)
            const Q2Species S = q2_species_load(s_state, sg);
            const double* const cb = s_pop + S.slot * SP + S.cur * BF;
            const uint32_t ctr1 = rng_ctr1((uint32_t)s_ctl[2] * 16u + (uint32_t)gen, (uint32_t)S.id, RNG_REPRODUCE);
            const int ra = r < lambda ? r : 0, rb = r1 < lambda ? r1 : ra;
            const ChildX<PB> cx[2] = {make_child_x(pb, (uint32_t)s_ctl[1], ctr1, (uint32_t)ra + 2u, cb, cb + M, cb + 3 * M),
                                      make_child_x(pb, (uint32_t)s_ctl[1], ctr1, (uint32_t)rb + 2u, cb, cb + M, cb + 3 * M)};
            f[0] = nonlink_primary(pb, cx[0], qc, f[0]), f[1] = nonlink_primary(pb, cx[1], qc, f[1]);
        } else {
            f[0] += 0.0, f[1] += 0.0;  // (nonlink_primary of no goal: the sum it returns)
        }
        f[0] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc), f[1] += balance_cost(pb, v3(0.0, 0.0, 0.0), qc);
        double* const s_fit = gbase + L.fitp;
        if (r < lambda) s_fit[r] = f[0];
        if (r1 < lambda) s_fit[r1] = f[1];
    }
    p_wave_sync();
    // elitist top-2 selection (:410-431), including the tie order of the reference's selection sort, and the winners' copy
    Q2_SCOPE;
    double* const s_fit = gbase + L.fitp;
    if (const int tie_bits = a.preselect >> 8) {  // (parity suites: the parked values made coarse, so that children tie and the tie order is exercised)
        for (int r = gtid; r < lambda; r += 16) {
            unsigned long long v;
            __builtin_memcpy(&v, &s_fit[r], 8);
            v &= ~((1ull << tie_bits) - 1ull);
            __builtin_memcpy(&s_fit[r], &v, 8);
        }
        p_wave_sync();
    }
    Q2Species S = q2_species_load(s_state, sg);
    double* const popS = s_pop + S.slot * SP;
    const double* const cb = popS + S.cur * BF;
    const double *p0g = cb, *p0d = cb + M, *p1d = cb + 3 * M;
    double b1f = P_INF, b2f = P_INF;
    int b1p = 0x7fffffff, b2p = 0x7fffffff;
    {
        // The two best children of the row's species from KEYS (sort_key: the fitness's upper bits and the position): two minimum reductions of one 64-bit number
        // each, exact unless a second candidate shares the upper bits of the runner-up's fitness (counted) -- then the pairs themselves are reduced.
        const int drop = a.sort_key_drop;
        unsigned long long k1 = ~0ull, k2 = ~0ull;
        for (int r = gtid; r < lambda; r += 16) {
            const unsigned long long k = sort_key(s_fit[r], r + 2, drop);
            const bool w1 = k < k1, w2 = k < k2;
            k2 = w1 ? k1 : (w2 ? k : k2);
            k1 = w1 ? k : k1;
        }
        const unsigned long long B1 = row_min_u64(k1);
        const unsigned long long B2 = row_min_u64(k1 == B1 ? k2 : k1);
        int shares = 0;  // this lane's candidates with the runner-up's upper bits (the runner-up itself is one of the row's)
        for (int r = gtid; r < lambda; r += 16) shares += ((sort_key(s_fit[r], r + 2, drop) ^ B2) >> drop) == 0ull ? 1 : 0;
        const unsigned long long one = p_ballot(shares > 0) & rowmask, more = p_ballot(shares > 1);
        const bool in_doubt = B2 != ~0ull && (one & (one - 1ull)) != 0ull;
        if (p_ballot(in_doubt) == 0ull && more == 0ull) {  // (the four rows of the wavefront decide together: one path through the code)
            b1p = (int)(B1 & 1023ull), b1f = s_fit[b1p - 2];
            if (B2 != ~0ull) b2p = (int)(B2 & 1023ull), b2f = s_fit[b2p - 2];
        } else {
            for (int r = gtid; r < lambda; r += 16) top2_insert(b1f, b1p, b2f, b2p, s_fit[r], r + 2);
            top2_wave(b1f, b1p, b2f, b2p, 16);
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
    // the winners become the elites (the species' other buffer): lanes 0 ... 7 of the row the first, 8 ... 15 the second where the ops are no more than eight,
    // else one winner per pass; a parent is copied, a child re-derived from the counter RNG, lane k its op k
    double* const nb = popS + (S.cur ^ 1) * BF;
    const uint32_t ctr1w = rng_ctr1((uint32_t)s_ctl[2] * 16u + (uint32_t)gen, (uint32_t)S.id, RNG_REPRODUCE);
    const bool both_at_once = M <= 8;
    // (the table of the parents' mixed momentum lies in the buffer the winners go to: every lane reads what it needs of both parents before any lane writes)
    for (int pass = 0; pass < (both_at_once ? 1 : 2); pass++) {
        const int i = both_at_once ? gtid >> 3 : pass;
        const int id = i == 0 ? first.id : second.id;
        double* const dst = nb + i * 2 * M;
        double gene[2] = {0.0, 0.0}, mom[2] = {0.0, 0.0};
        int kk[2] = {-1, -1};
        {
            BIOIK_FP_STRICT
            const ChildX<PB> cx = make_child_x(pb, (uint32_t)s_ctl[1], ctr1w, (uint32_t)(id >= 2 ? id - 2 : 0) + 2u, p0g, p0d, p1d);
            int n = 0;
            for (int k = both_at_once ? (gtid & 7) : gtid; k < n_ops && n < 2; k += both_at_once ? 8 : 16, n++) {
                kk[n] = k;
                if (id < 2) {
                    const double* src = cb + id * 2 * M;
                    gene[n] = src[k], mom[n] = src[M + k];
                } else {
                    gene[n] = cx.template value<false>(k);  // (lane k its op k: the clip range differs from lane to lane)
                    mom[n] = 0.0;
                    if ((active_mask >> k) & 1ull) {
                        const double parent_gradient = p0d[k] * (1.0 - cx.fmix) + p1d[k] * cx.fmix;
                        mom[n] = parent_gradient * (1.0 - 0.3) + (gene[n] - p0g[k]) * 0.3;
                    }
                }
            }
        }
        p_wave_sync();  // (all reads of the momentum table and of the parents are done)
        for (int n = 0; n < 2; n++)
            if (kk[n] >= 0) dst[kk[n]] = gene[n], dst[M + kk[n]] = mom[n];
        p_wave_sync();
    }
    S.cur ^= 1;
    S.pf0 = first.f;
    S.pf1 = second.f;
    if (gtid == 0) q2_species_store(s_state, sg, S);
    p_wave_sync();
}

// memetic phase on the elite of every species group (:436-570): fresh linearisation at the elite, then up to eight iterations of finite-difference gradient,
// L1 normalisation, three-point line search, clipped candidate, acceptance on primary fitness -- solve_body's three lane roles on a row of 16 lanes, the
// candidate's round doubling as the next iteration's gradient round.  Then the species' ranking walk (:607-614) with the success test of its elite.
template <class PB>
BIOIK_DEV void q2_memetic_and_rank(const Q2<PB>& c) {
    const PB pb = c.pb;
    const SolveArgs& a = c.a;
    const DevSolveParams& sp = a.sp;
    double* const lds = c.lds;
    const LdsLayout& L = c.L;
    const int slot_stride = c.slot_stride, M = c.M, SP = c.SP, BF = c.BF, n_ops = c.n_ops, D = c.D, T = c.T;
    const uint64_t active_mask = pb->active_mask;
    if (sp.memetic) {
        Q2_SCOPE;
        const Q2Species S = q2_species_load(s_state, sg);
        double* const popS = s_pop + S.slot * SP;
        double* const el = popS + S.cur * BF;  // the elite's genes, edited in place
        double* const s_xn = gbase + L.xn;
        double* const s_gv = gbase + L.gv;
        double* const s_frames = gbase + L.frames;
        double* const s_tips = gbase + L.tips;
        double* const s_delta = gbase + L.delta;
        double* const s_base = gbase + L.base;
        double* const s_grad = gbase + L.grad;
        double* const s_xm = gbase + L.xm;
        double* const s_xp = gbase + L.xp;
        double* const s_dv = gbase + L.dv;
        double* const s_ex = gbase + L.bc;  // values exchanged between the lanes of the row: [0] primary, [1] all goals at the elite, [2] / [3] f(x - g) / f(x + g)
        {  // RobotFK::applyConfiguration + initializeMutationApproximator at the elite: the row shares the walk, lane 0 publishes, then the lanes fan out over (tip, op)
            const XV xe{el, 1};
            fk_walk<16>(pb, xe, lds, gtid == 0 ? s_frames : nullptr, [&](int t, const F7& f) {
                if (gtid == 0) f7_store(s_tips + t * 7, f);
            }, s_prefix);
            for (int k = gtid; k < n_ops; k += 16) s_base[k] = xe(k);
            p_wave_sync();
            for (int t = 0; t < T; t++)
                for (int k = gtid; k < n_ops; k += 16) {
                    double o[7];
                    approximator_entry(pb, t, k, s_frames, s_tips, o, s_base, s_prefix);
                    double* d = s_delta + ((size_t)t * n_ops + k) * 7;
                    for (int cc = 0; cc < 7; cc++) d[cc] = o[cc];
                }
            p_wave_sync();
        }
        double dp = 0.0000001;
        {
            uint32_t o0, o1;
            philox2x32_10((uint32_t)s_ctl[1], rng_ctr0(0, 0), rng_ctr1((uint32_t)s_ctl[2] * 16u, (uint32_t)S.id, RNG_MEMETIC_SIGN), o0, o1);
            if (rng_uniform(o0, o1) < 0.5) dp = -dp;
        }
        const int my_op = gtid < D ? pb->op_of_gene[gtid] : -1;  // lane i differentiates gene i, lane D holds the elite itself
        double* const s_gop = s_gv;  // gradient in op order (zero for the ops that are not genes), next to the gene-ordered s_grad
        const int FB = 8 * T;
        double* const s_x4 = s_xn;
        double* const fc_base = L.fc >= 0 ? gbase + L.fc : popS + (S.cur ^ 1) * BF;  // (make_layout: fc_in_pop)
        // component lanes: one or two chains (same delta entries, two displacement vectors), four entries per trip
        auto chains = [&](const double* d0, double* f0, const double* d1, double* f1) {
            for (int idx = gtid; idx < FB; idx += 16) {
                const int t = idx >> 3, cc = idx & 7;
                if (cc == 7) continue;
                double a0 = s_tips[t * 7 + cc], a1 = a0;
                const double* dl = s_delta + (size_t)t * n_ops * 7 + cc;
                for (int g0 = 0; g0 < n_ops; g0 += 4) {
                    double d[4], v0[4], v1[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int kk = g0 + j < n_ops ? g0 + j : n_ops - 1;
                        const bool pad = g0 + j >= n_ops;
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
        // goal fitness of the lane's frames `fc` (+ its gene's delta * step): primary = all goals (no secondary goal under this kernel: the empty sum)
        auto goals_on = [&](const double* fc, int dop, double dstep, const PerturbX& x, double& prim, double& all) {
#if !defined(BIOIK_NO_POSE_ONLY)
            if (pb->pose_only) {  // one PoseGoal on one tip and nothing else: the same operations without the goal tables (0 + w² e = w² e)
                F7 f = f7_load(fc);
                if (dstep != 0.0 && dop >= 0) {
                    const double* dl = s_delta + (size_t)dop * 7;
                    f = F7{{BK_FMA(dl[0], dstep, f.p.x), BK_FMA(dl[1], dstep, f.p.y), BK_FMA(dl[2], dstep, f.p.z)},
                           {BK_FMA(dl[3], dstep, f.q.x), BK_FMA(dl[4], dstep, f.q.y), BK_FMA(dl[5], dstep, f.q.z), BK_FMA(dl[6], dstep, f.q.w)}};
                }
                const double* Pg = qc.par + pb->pose_param_off;
                double e = dist2(f.p, v3(Pg[0], Pg[1], Pg[2]));
                const Q4 d = Q4{Pg[3] - f.q.x, Pg[4] - f.q.y, Pg[5] - f.q.z, Pg[6] - f.q.w};
                const Q4 s4 = Q4{Pg[3] + f.q.x, Pg[4] + f.q.y, Pg[5] + f.q.z, Pg[6] + f.q.w};
                const double rs = Pg[7];
                e += fmin(qdot(d, d), qdot(s4, s4)) * (rs * rs);
                prim = all = e * pb->pose_weight_sq;
                return;
            }
#endif
            double acc = 0.0;
            for (int t = 0; t < T; t++) {
                F7 f = f7_load(fc + t * 8);
                if (dstep != 0.0) {
                    double d[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                    if (dop >= 0) {
                        const double* dl = s_delta + ((size_t)t * n_ops + dop) * 7;
                        for (int cc = 0; cc < 7; cc++) d[cc] = dl[cc];
                    }
                    f = F7{{BK_FMA(d[0], dstep, f.p.x), BK_FMA(d[1], dstep, f.p.y), BK_FMA(d[2], dstep, f.p.z)},
                           {BK_FMA(d[3], dstep, f.q.x), BK_FMA(d[4], dstep, f.q.y), BK_FMA(d[5], dstep, f.q.z), BK_FMA(d[6], dstep, f.q.w)}};
                }
                acc = tip_goals(pb, t, f, x, qc, acc);
            }
            acc = nonlink_primary(pb, x, qc, acc);
            acc += balance_cost(pb, v3(0.0, 0.0, 0.0), qc);
            prim = acc;
            all = acc + 0.0;
        };
        for (int k = gtid; k < n_ops; k += 16) s_gop[k] = 0.0;
        const bool odd = gtid & 1;
        bool descending = true;
        double f2p = 0.0, fa = 0.0;  // primary fitness / all goals at the elite, as the last gradient round left them
        for (int it = 0; it < 8 && descending; it++) {
            // rounds: 0 gradient (:450-475), 1 L1 normalisation and the two support points (:477-495), 2 the step along the gradient (:498-568) with the
            // candidate's gradient computed beside it -- the next iteration's round 0 (solve_body)
            double fnorm = 0.0;
            bool no_candidate = false;
            for (int round = it == 0 ? 0 : 1; round < 3; round++) {
                double* dv0 = s_dv + (round == 0 ? 0 : round == 1 ? 1 : 3) * M;
                double* fc0 = fc_base + (round == 0 ? 0 : round == 1 ? 1 : 3) * FB;
                if (round == 0) {
                    for (int k = gtid; k < n_ops; k += 16) dv0[k] = ((active_mask >> k) & 1ull) ? el[k] - s_base[k] : 0.0;
                } else if (round == 1) {
                    double sum = dp * dp;
                    for (int i = 0; i < D; i++) sum += fabs(s_grad[i]);
                    fnorm = 1.0 / sum * dp;
                    for (int k = gtid; k < n_ops; k += 16) {
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
                        const double v1 = f2 - f1, v2 = f3 - f2;
                        const double v = (v1 + v2) * 0.5, aa = v1 - v2;
                        step_size = v / aa;
                    } else {  // 'l' :545-568
                        const double cost_diff = (f3 - f1) * 0.5;
                        step_size = -(f2 / cost_diff);
                    }
                    // (a candidate with a NaN gene, or with a gene of magnitude BIOIK_CANDIDATE_BOUND and more, is no candidate: quirks Q5 / Q7, solve_body)
                    bool bad_here = false;
                    for (int k = gtid; k < n_ops; k += 16) {
                        const double e = el[k], gv = s_gop[k] * fnorm;
                        const bool on = (active_mask >> k) & 1ull;
                        const double raw = e + gv * step_size;
                        const bool is_nan = on && !(raw == raw);
                        const double x4 = (on && !is_nan) ? fmin(fmax(raw, s_clip[k]), s_clip[M + k]) : e;
                        bad_here = bad_here || is_nan || (on && fabs(x4) >= BIOIK_CANDIDATE_BOUND);
                        s_x4[k] = x4;
                        dv0[k] = on ? x4 - s_base[k] : 0.0;
                    }
                    no_candidate = (p_ballot(bad_here) & rowmask) != 0ull;
                }
                p_wave_sync();
                chains(dv0, fc0, round == 1 ? dv0 + M : nullptr, fc0 + FB);
                p_wave_sync();
                double vprim, vall;
                const bool grad_round = round != 1;
                const PerturbX xq{round == 0 ? el : (round == 2 ? s_x4 : (odd ? s_xp : s_xm)), grad_round ? my_op : -1, grad_round ? dp : 0.0};
                goals_on(fc0 + ((round == 1 && odd) ? FB : 0), grad_round ? my_op : -1, grad_round ? dp : 0.0, xq, vprim, vall);
                if (round == 0) {
                    if (gtid == D) s_ex[0] = vprim, s_ex[1] = vall;
                    p_wave_sync();
                    f2p = s_ex[0], fa = s_ex[1];
                    if (my_op >= 0) {
                        s_grad[gtid] = vall - fa;
                        s_gop[my_op] = vall - fa;
                    }
                    p_wave_sync();
                } else if (round == 1) {
                    if (gtid < 2) s_ex[2 + gtid] = vall;
                    p_wave_sync();
                } else {
                    if (gtid == D) s_ex[0] = vprim, s_ex[1] = vall;
                    p_wave_sync();
                    const double cprim = s_ex[0], call = s_ex[1];
                    const bool accept = !no_candidate && cprim < f2p;  // accept iff the primary fitness improves, else stop (:527-538)
                    // (a row shares its wavefront with three others: it stays in step with them and merely repeats the rejected iteration from its support points on,
                    // which changes nothing, until every row of the wavefront has stopped)
                    if (p_ballot(accept) == 0ull) descending = false;
                    if (accept) {
                        for (int k = gtid; k < n_ops; k += 16) el[k] = s_x4[k];
                        f2p = cprim, fa = call;
                        if (my_op >= 0) {
                            s_grad[gtid] = vall - call;
                            s_gop[my_op] = vall - call;
                        }
                    }
                    p_wave_sync();
                }
            }
        }
        p_wave_sync();
    }
    // species ranking fitness: exact FK of the elite (:607-614); the same walk decides whether that elite satisfies the goals (problem.cpp:259-341)
    {
        Q2_SCOPE;
        Q2Species S = q2_species_load(s_state, sg);
        const double* const cb = s_pop + S.slot * SP + S.cur * BF;
        const FitCheck fc = exact_fitness_check<16>(pb, XV{cb, 1}, qc, lds, sp.dpos, sp.drot, sp.dtwist, 1, s_prefix);
        S.improved = (fc.fitness != S.fit) ? 1 : 0;
        S.fit = fc.fitness;
        S.pf0 = fc.fitness;
        S.ok = fc.ok;
        p_wave_sync();  // (every lane of the row has read the record)
        if (gtid == 0) q2_species_store(s_state, sg, S);
        p_wave_sync();
    }
}

// species management (:617-645), the solution's update and the island loop's checks at the end of a step (ik_parallel.h:160-181), per slot.
// Returns what the slot does next: 0 goes on, 1 its unit has ended (results written), 2 its unit leaves for the next launch (state written).
template <class PB>
BIOIK_DEV int q2_species_and_checks(const Q2<PB>& c, bool busy, bool dry, bool neighbour_idle, unsigned long long deadline) {
    const PB pb = c.pb;
    const SolveArgs& a = c.a;
    const DevSolveParams& sp = a.sp;
    double* const lds = c.lds;
    const LdsLayout& L = c.L;
    const int slot_stride = c.slot_stride, M = c.M, SP = c.SP, BF = c.BF, n_ops = c.n_ops, V = c.V;
    Q2_SCOPE;
    const uint32_t key = (uint32_t)s_ctl[1];
    const uint32_t step = (uint32_t)s_ctl[2];
    const uint64_t unit = (uint64_t)s_ctl[0];
    Q2Species A = q2_species_load(s_state, 0), B = q2_species_load(s_state, 1);
    if (B.fit < A.fit) {
        const Q2Species tmp = A;
        A = B;
        B = tmp;
    }
    {
        uint32_t o0, o1;
        philox2x32_10(key, rng_ctr0(0, 0), rng_ctr1(step * 16u, (uint32_t)B.id, RNG_WIPEOUT), o0, o1);
        bool wipe = rng_uniform(o0, o1) < 0.1;
        wipe = wipe || !B.improved;
        if (sp.no_wipeout) wipe = false;
        if (p_ballot(wipe) != 0ull) {  // (both slots take the path when one must: the walk inside is shared by a slot's lanes)
            BIOIK_FP_STRICT
            const uint32_t wc1 = rng_ctr1(step * 16u, (uint32_t)B.id, RNG_WIPEOUT_GENE);
            double* const cbw = s_pop + B.slot * SP + B.cur * BF;
            p_wave_sync();  // (every lane has read the records)
            if (wipe)
                for (int k = tid32; k < n_ops; k += 32) {
                    double v = cbw[k];
                    if (pb->ops[k].gene >= 0) {
                        philox2x32_10(key, rng_ctr0(0, (uint32_t)pb->ops[k].gene), wc1, o0, o1);
                        v = rng_uniform(o0, o1) * (pb->ops[k].vmax - pb->ops[k].vmin) + pb->ops[k].vmin;
                    }
                    cbw[k] = v, cbw[M + k] = 0.0;
                    cbw[2 * M + k] = v, cbw[3 * M + k] = 0.0;
                }
            p_wave_sync();
            const double wf = exact_fitness_check<32>(pb, XV{cbw, 1}, qc, lds, 0.0, 0.0, 0.0, 0, s_prefix).fitness;
            if (wipe) B.pf0 = B.pf1 = wf;
        }
    }
    const int steps = (int)step + 1;
    const bool better = A.fit < s_state[20];
    p_wave_sync();  // (every lane has read the bookkeeping)
    if (better) {
        const double* const cbs = s_pop + A.slot * SP + A.cur * BF;
        for (int k = tid32; k < n_ops; k += 32) s_sol[k] = cbs[k];
    }
    if (tid32 == 0) {
        q2_species_store(s_state, 0, A), q2_species_store(s_state, 1, B);
        if (better) s_state[20] = A.fit, s_state[21] = (double)A.ok;
        s_ctl[2] = (double)steps;
    }
    p_wave_sync();
    // ik_parallel.h:173-181: fitness and success test of the solution = those of the elite it was copied from (or of the seed)
    const double final_fit = s_state[20];
    const bool success = s_state[21] != 0.0;
    bool expired = false, overtaken = false;
    if (busy && success && a.first_success && tid32 == 0) p_atomic_min(a.first_success + unit / (uint64_t)sp.islands, (unsigned int)steps);  // ik_parallel.h:176-177 `finished = 1`
    if (sp.timeout_ticks != 0ull) expired = p_wall_clock() >= deadline;  // at least one step has run (ik_parallel.h:160 `iteration != 0`)
    // ik_parallel.h:160 `!finished`: another island of the query has passed after no more steps than this one has run
    if (a.first_success) {  // (lane 0 of the slot reads the word, the verdict crosses the slot)
        int ov = 0;
        if (busy && !success && tid32 == 0) ov = p_atomic_load(a.first_success + unit / (uint64_t)sp.islands) <= (unsigned int)steps ? 1 : 0;
        overtaken = p_shfl(ov, lane & 32) != 0;
    }
    const bool ended = success || expired || overtaken || steps >= sp.max_steps;
    // the counter has run dry and the neighbour slot is idle: the unit goes on under the next launch's kernel instead of at half a wavefront
    // (drain_below < 0: the parity suites' test pattern -- unit u leaves after 1 + hash(u) % -drain_below steps, whatever its neighbour does)
    const bool leaves = busy && !ended && a.carry_list != nullptr &&
                        (a.drain_below < 0 ? steps >= 1 + (int)((((uint32_t)unit + 1u) * 2654435761u >> 16) % (uint32_t)(-a.drain_below)) : (dry && neighbour_idle && steps >= a.drain_min_steps));
    if (busy && leaves) {
        const int carry_n = 2 * BF + M + 24;
        double* const cw = a.carry + unit * (uint64_t)carry_n;
        for (int i = tid32; i < 2 * BF; i += 32) {
            const int r = i >= BF ? 1 : 0;
            p_store_device(cw + i, s_pop[(int)s_state[r * 8 + 4] * SP + (int)s_state[r * 8 + 5] * BF + (i - r * BF)]);  // (slot, cur of the species of rank r)
        }
        for (int i = tid32; i < M + 24; i += 32)
            p_store_device(cw + 2 * BF + i, i < M ? s_sol[i] : ((i - M == 5 || i - M == 13) ? 0.0 : (i - M == 16 ? (double)steps : (i - M < 22 ? s_state[i - M] : 0.0))));
        if (tid32 == 0) p_store_device(a.carry_list + p_atomic_inc(a.carry_count), (int32_t)unit);
    }
    if (busy && ended) {  // result of this island; ranking fitness of ik_parallel.h:229-246 (no secondary goal: the primary fitness)
        double* const out = a.solutions + unit * (uint64_t)V;
        for (int i = tid32; i < V; i += 32) out[i] = s_seed[i];
    }
    p_wave_sync();
    if (busy && ended) {
        double* const out = a.solutions + unit * (uint64_t)V;
        for (int k = tid32; k < n_ops; k += 32)
            if (pb->ops[k].gene >= 0) out[pb->ops[k].var] = s_sol[k];
        if (tid32 == 0) {
            a.fitness[unit] = final_fit;
            a.success[unit] = success ? 1 : 0;
            a.steps[unit] = steps;
        }
    }
    return !busy ? 0 : (ended ? 1 : (leaves ? 2 : 0));
}

template <class PB>
BIOIK_DEV void solve_q2_body(const SolveArgs& a, uint64_t /*wave*/, double* lds) {
    const PB pb = (PB)a.pb;
    const DevSolveParams& sp = a.sp;
    Q2<PB> c{pb, a, lds, q2_layout(pb->n_ops, pb->V, pb->P, pb->T, sp.lambda), 0, 0, 0, 0, pb->n_ops, pb->D, pb->T, pb->V, pb->P, sp.lambda};
    c.slot_stride = q2_slot_doubles(c.L);
    c.M = c.n_ops > 0 ? c.n_ops : 1;
    c.SP = 2 * 2 * 2 * c.M, c.BF = 4 * c.M;
    // ik_parallel.h:160, 200: the caller's timeout is a point in time fixed when the call comes in (eager calls: SolveArgs::deadline); a captured call counts
    // from its first workgroup's start
    unsigned long long deadline = 0ull;
    if (sp.timeout_ticks != 0ull) {
        unsigned long long t1 = 0ull;
        if (p_lane_fresh() == 0) t1 = a.launch_clock ? p_stamp_once(a.launch_clock, p_wall_clock()) + sp.timeout_ticks : a.deadline;
        deadline = ((unsigned long long)(unsigned int)p_shfl((int)(t1 >> 32), 0) << 32) | (unsigned long long)(unsigned int)p_shfl((int)(t1 & 0xffffffffull), 0);
    }
    bool busy = false, dry = false;  // of this lane's slot: it has a unit / the launch has no more units to give
    for (;;) {
        if (p_ballot(!busy && !dry) != 0ull) {  // a slot without a unit asks for the next one (SolveArgs::next_unit) and initialises it
            const int lane = p_lane_fresh();
            const bool need = !busy && !dry;
            unsigned int u = 0u;
            if (need && (lane & 31) == 0) u = p_atomic_inc(a.next_unit);
            u = (unsigned int)p_shfl((int)u, lane & 32);
            const bool got = need && (uint64_t)u < a.n_units;
            if (need && !got) dry = true;
            if (p_ballot(got) != 0ull) q2_init(c, got, got ? u : 0u);
            busy = busy || got;
        }
        const unsigned long long busy_lanes = p_ballot(busy);
        if (busy_lanes == 0ull) break;
        for (int gen = 0; gen < sp.generations; gen++) q2_generation(c, gen);
        q2_memetic_and_rank(c);
        const bool neighbour_idle = (p_lane_fresh() & 32) ? (busy_lanes & 1ull) == 0ull : (busy_lanes >> 32) == 0ull;
        const bool counter_dry = p_ballot(dry) != 0ull;  // (a slot learns that the launch has no more units when IT asks: its neighbour learns it here)
        const int next = q2_species_and_checks(c, busy, counter_dry, neighbour_idle, deadline);
        if (next != 0) busy = false;
    }
}
