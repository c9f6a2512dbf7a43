// bioik_gradient.h — workgroup body of the point solvers `gd_c` and `jac` (reference src/ik_gradient.cpp:136-251 and :42-133, 269-292).
//
// Mapping: one wavefront owns one query.  Both solvers carry ONE configuration (no population), so the lanes are not individuals but
// the pieces of a step that are independent:
//   gd_c  lane i < D differentiates gene i (two exact-FK fitness evaluations, x_i -+ 1e-4); lanes 0 / 1 then score the two support
//         points of the linear step estimate, and again the new configuration against the best one so far;
//   jac   one chain walk publishes the per-joint frame chain to LDS, lanes fan out over (tip, gene) Jacobian columns and over the tips'
//         twists towards their pose goals; the minimum-norm least-squares step is a one-sided Jacobi SVD on the 6T x D system in LDS
//         (lane 0: the system is a few dozen numbers), in exactly the operation order of the CPU restatement the tests compare with.
// Everything of the query lives in LDS; HBM sees the query once on the way in and the result once on the way out, as in k_solve.
#pragma once
#include "bioik_kernels.h"

struct PointLayout {  // offsets in doubles
    int seed, par, prefix, sol, best, grad, gg, ex, xcol, slots, frames, tips, jac, A, Vm, sig, b, x, total;
};
BIOIK_HD PointLayout make_point_layout(int n_ops, int V, int P, int T, int n_slots, int D, int nthreads) {
    PointLayout L;
    const int m = n_ops > 0 ? n_ops : 1, d = D > 0 ? D : 1, t6 = 6 * (T > 0 ? T : 1);
    const int q = d < t6 ? d : t6;
    int o = 0;
    L.seed = o, o += V;
    L.par = o, o += P > 0 ? P : 1;
    L.prefix = o, o += 8;
    L.sol = o, o += m;
    L.best = o, o += m;
    L.grad = o, o += m;     // op-indexed gradient (zero for the ops that are not genes)
    L.gg = o, o += d;       // gene-ordered gradient entries (the L1 norm sums them in gene order)
    L.ex = o, o += 8;       // values exchanged between lanes
    L.xcol = o, o += m * nthreads;
    L.slots = o, o += n_slots * 7 * nthreads;
    L.frames = o, o += m * 7;
    L.tips = o, o += (T > 0 ? T : 1) * 7;
    L.jac = o, o += t6 * d;  // row-major [6T][D]
    L.A = o, o += t6 * d;
    L.Vm = o, o += q * q;
    L.sig = o, o += q;
    L.b = o, o += t6;
    L.x = o, o += d;
    L.total = o;
    return L;
}

// minimum-norm least-squares solution of J x = b through a one-sided Jacobi SVD; one lane, arrays in LDS; the loop order is the
// definition the test-suite's CPU restatement follows operation by operation; plain IEEE operations, no contraction
BIOIK_DEV void pinv_solve_lds(const double* J, int rows, int cols, const double* b, double* x, double* A, double* V, double* sigma) {
    BIOIK_FP_STRICT
    const bool transposed = rows < cols;
    const int p = transposed ? cols : rows, q = transposed ? rows : cols;
    for (int i = 0; i < p; i++)
        for (int j = 0; j < q; j++) A[i * q + j] = transposed ? J[j * cols + i] : J[i * cols + j];
    for (int i = 0; i < q * q; i++) V[i] = 0.0;
    for (int j = 0; j < q; j++) V[j * q + j] = 1.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int i = 0; i < q - 1; i++)
            for (int j = i + 1; j < q; j++) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int k = 0; k < p; k++) {
                    const double ai = A[k * q + i], aj = A[k * q + j];
                    alpha += ai * ai, beta += aj * aj, gamma += ai * aj;
                }
                if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < p; k++) {
                    const double ai = A[k * q + i], aj = A[k * q + j];
                    A[k * q + i] = c * ai - s * aj;
                    A[k * q + j] = s * ai + c * aj;
                }
                for (int k = 0; k < q; k++) {
                    const double vi = V[k * q + i], vj = V[k * q + j];
                    V[k * q + i] = c * vi - s * vj;
                    V[k * q + j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    double smax = 0.0;
    for (int k = 0; k < q; k++) {
        double n2 = 0.0;
        for (int i = 0; i < p; i++) n2 += A[i * q + k] * A[i * q + k];
        sigma[k] = sqrt(n2);
        if (sigma[k] > smax) smax = sigma[k];
    }
    const double threshold = 2.220446049250313e-16 * (double)(rows < cols ? rows : cols) * smax;  // Eigen: epsilon * diagSize = min(rows, cols)
    for (int i = 0; i < cols; i++) x[i] = 0.0;
    for (int k = 0; k < q; k++) {
        if (!(sigma[k] > threshold)) continue;
        const double inv = 1.0 / (sigma[k] * sigma[k]);
        double d = 0.0;
        if (!transposed) {
            for (int i = 0; i < rows; i++) d += A[i * q + k] * b[i];
            d *= inv;
            for (int i = 0; i < cols; i++) x[i] += V[i * q + k] * d;
        } else {
            for (int i = 0; i < rows; i++) d += V[i * q + k] * b[i];
            d *= inv;
            for (int i = 0; i < cols; i++) x[i] += A[i * q + k] * d;
        }
    }
}

// one (query, island) per workgroup of ONE wavefront; the island loop in budget / wall-clock form as in solve_body.  Island 0 is solver
// thread 0 of the reference (started at the seed), the islands i > 0 its threads started at random configurations (the _2 / _4 / _8
// factory names, ik_gradient.cpp:157-159, :283-285); solver 4 (gd_r) draws a new configuration after a step that did not improve.
BIOIK_DEV void point_body(const SolveArgs& a, uint64_t unit, double* lds) {
    const ProbPtr pb = a.pb;
    const DevSolveParams& sp = a.sp;
    const int tid = p_tid(), nth = p_nthreads();
    const int V = pb->V, P = pb->P, T = pb->T, n_ops = pb->n_ops, D = pb->D;
    const uint64_t active_mask = pb->active_mask;
    const PointLayout L = make_point_layout(n_ops, V, P, T, pb->n_slots, D, nth);
    double *s_seed = lds + L.seed, *s_par = lds + L.par, *s_prefix = lds + L.prefix, *s_sol = lds + L.sol, *s_best = lds + L.best;
    double *s_grad = lds + L.grad, *s_gg = lds + L.gg, *s_ex = lds + L.ex, *s_slots = lds + L.slots;
    double *s_frames = lds + L.frames, *s_tips = lds + L.tips, *s_jac = lds + L.jac;
    const int M = n_ops > 0 ? n_ops : 1;
    double* xcol = lds + L.xcol + tid;
    const XV xl{xcol, nth};
    const uint64_t q = unit / (uint64_t)sp.islands;
    const uint32_t island = (uint32_t)(unit % (uint64_t)sp.islands);
    const uint32_t key = rng_query_key(sp.random_seed, sp.first_query + q, island);
    // random(modelInfo.getMin(vi), modelInfo.getMax(vi)) for the gene of op k, draw `count` of this island
    auto random_gene = [&](int k, uint32_t count) {
        BIOIK_FP_STRICT
        uint32_t o0, o1;
        philox2x32_10(key, rng_ctr0(0, (uint32_t)pb->ops[k].gene), rng_ctr1(count, 0u, RNG_POINT_RANDOM), o0, o1);
        return rng_uniform(o0, o1) * (pb->ops[k].vmax - pb->ops[k].vmin) + pb->ops[k].vmin;
    };
    for (int i = tid; i < V; i += nth) s_seed[i] = a.seeds[q * V + i];
    for (int i = tid; i < P; i += nth) s_par[i] = a.params[q * P + i];
    p_wave_sync();
    const QueryCtx qc{s_seed, s_par};
    for (int k = tid; k < n_ops; k += nth) {  // solution = problem.initial_guess (ik_gradient.cpp:150, :281); threads > 0: a random one (:157-159, :283-285)
        double v = s_seed[pb->ops[k].var];
        if (island > 0u && ((active_mask >> k) & 1ull)) v = random_gene(k, 0u);
        s_sol[k] = v, s_best[k] = v, s_grad[k] = 0.0;
    }
    p_wave_sync();
    if (pb->n_prefix > 0) {
        if (tid == 0) f7_store(s_prefix, fk_prefix(pb, XV{s_sol, 1}));
        p_wave_sync();
    }
    unsigned long long deadline = 0ull;
    if (sp.timeout_ticks != 0ull) {
        if (tid == 0) {
            const unsigned long long t0 = a.launch_clock ? p_stamp_once(a.launch_clock, p_wall_clock()) + sp.timeout_ticks : a.deadline;  // (SolveArgs::deadline)
            s_ex[6] = (double)(t0 >> 32), s_ex[7] = (double)(t0 & 0xffffffffull);
        }
        p_wave_sync();
        deadline = ((unsigned long long)s_ex[6] << 32) | (unsigned long long)s_ex[7];
        p_wave_sync();
    }
    const int my_op = tid < D ? pb->op_of_gene[tid] : -1;
    auto fitness_of = [&](const XV& x) { return eval_exact_primary(pb, x, qc, s_slots, s_prefix); };
    auto clip_op = [&](double v, int k) {  // RobotInfo::clip, robot_info.h:109-113 (clamp2: max first, then min)
        const double lo = pb->ops[k].clip_min, hi = pb->ops[k].clip_max;
        if (v < lo) v = lo;
        if (v > hi) v = hi;
        return v;
    };
    int steps = 0;
    bool success = false, reset = false;
    double final_fit = BIOIK_DBL_MAX;
    for (int step = 0; step < sp.max_steps; step++) {
        if (sp.solver != 2) {
            // ---- IKGradientDescent<'c'>::step (solver 1) / <' '> (solver 3) / <'r'> (solver 4), ik_gradient.cpp:162-247
            BIOIK_FP_STRICT
            if (reset) {  // random reset if stuck (:165-170)
                reset = false;
                for (int k = tid; k < n_ops; k += nth)
                    if ((active_mask >> k) & 1ull) s_sol[k] = random_gene(k, (uint32_t)step + 1u);
                p_wave_sync();
            }
            const double jd = 0.0001;
            double g = 0.0;
            if (my_op >= 0) {
                for (int k = 0; k < n_ops; k++) xcol[(size_t)k * nth] = (k == my_op) ? s_sol[k] - jd : s_sol[k];
            } else {
                for (int k = 0; k < n_ops; k++) xcol[(size_t)k * nth] = s_sol[k];
            }
            const double p1g = fitness_of(xl);
            if (my_op >= 0) xcol[(size_t)my_op * nth] = s_sol[my_op] + jd;
            const double p3g = fitness_of(xl);
            if (my_op >= 0) {
                g = p3g - p1g;
                s_grad[my_op] = g, s_gg[tid] = g;
            }
            p_wave_sync();
            double sum = 0.0001;
            for (int i = 0; i < D; i++) sum += fabs(s_gg[i]);
            const double f = 1.0 / sum * jd;
            p_wave_sync();
            for (int k = tid; k < n_ops; k += nth) s_grad[k] = ((active_mask >> k) & 1ull) ? s_grad[k] * f : 0.0;
            p_wave_sync();
            {   // the two support points: even lanes x - g, odd lanes x + g
                const bool odd = tid & 1;
                for (int k = 0; k < n_ops; k++) {
                    const bool on = (active_mask >> k) & 1ull;
                    xcol[(size_t)k * nth] = on ? (odd ? s_sol[k] + s_grad[k] : s_sol[k] - s_grad[k]) : s_sol[k];
                }
                const double fl = fitness_of(xl);
                if (tid < 2) s_ex[tid] = fl;
            }
            p_wave_sync();
            const double p1 = s_ex[0], p3 = s_ex[1];
            const double p2 = (p1 + p3) * 0.5;
            const double cost_diff = (p3 - p1) * 0.5;
            double joint_diff = p2 / cost_diff;
            if (!__builtin_isfinite(joint_diff)) joint_diff = 0.0;
            p_wave_sync();
            bool accept = true;  // 'c': always accept and continue (:220-223)
            if (sp.solver >= 3) {  // ' ' / 'r': has the solution improved? (:225-232) even lanes score the candidate, odd lanes the configuration
                const bool odd = tid & 1;
                for (int k = 0; k < n_ops; k++)
                    xcol[(size_t)k * nth] = (!odd && ((active_mask >> k) & 1ull)) ? clip_op(s_sol[k] - s_grad[k] * joint_diff, k) : s_sol[k];
                const double fl = fitness_of(xl);
                if (tid < 2) s_ex[tid] = fl;
                p_wave_sync();
                accept = s_ex[0] < s_ex[1];
                p_wave_sync();
                if (!accept && sp.solver == 4) reset = true;  // 'r': reset if stuck (:233-237)
            }
            if (accept)
                for (int k = tid; k < n_ops; k += nth)
                    if ((active_mask >> k) & 1ull) s_sol[k] = clip_op(s_sol[k] - s_grad[k] * joint_diff, k);
            p_wave_sync();
            {   // update best solution (:246): lane 0 scores the configuration, lane 1 the best one so far
                const double fv = fitness_of(XV{(tid & 1) ? s_best : s_sol, 1});
                if (tid < 2) s_ex[2 + tid] = fv;
            }
            p_wave_sync();
            const bool better = s_ex[2] < s_ex[3];
            p_wave_sync();
            if (better)
                for (int k = tid; k < n_ops; k += nth) s_best[k] = s_sol[k];
            p_wave_sync();
        } else {
            // ---- IKJacobianBase::optimizeJacobian, ik_gradient.cpp:69-132; getSolution() is the configuration itself
            BIOIK_FP_STRICT
            fk_walk(pb, XV{s_sol, 1}, s_slots, tid == 0 ? s_frames : nullptr, [&](int t, const F7& f) {
                if (tid == 0) f7_store(s_tips + t * 7, f);
            }, s_prefix);
            p_wave_sync();
            double* s_b = lds + L.b;
            for (int t = tid; t < T; t += nth) {  // tip_diffs = frameTwist(tip, objective) (:84-93), rows in the PUBLIC tip order
                F7 obj = f7_identity();
                const int ot = pb->tips[t].obj_type, oo = pb->tips[t].obj_param_off;
                if (ot == G_POSITION) obj.p = v3(s_par[oo], s_par[oo + 1], s_par[oo + 2]);
                if (ot == G_ORIENTATION) obj.q = Q4{s_par[oo], s_par[oo + 1], s_par[oo + 2], s_par[oo + 3]};
                if (ot == G_POSE) obj = F7{{s_par[oo], s_par[oo + 1], s_par[oo + 2]}, {s_par[oo + 3], s_par[oo + 4], s_par[oo + 5], s_par[oo + 6]}};
                V3 lin, ang;
                frame_twist(f7_load(s_tips + t * 7), obj, lin, ang);
                double* d = s_b + pb->tips[t].out_index * 6;
                d[0] = lin.x, d[1] = lin.y, d[2] = lin.z, d[3] = ang.x, d[4] = ang.y, d[5] = ang.z;
            }
            for (int idx = tid; idx < T * D; idx += nth) {  // Jacobian columns (:97), [6 * public tip + c][gene]
                const int t = idx / D, gidx = idx - t * D;
                double o6[6];
                jacobian_entry6(pb, t, pb->op_of_gene[gidx], s_frames, s_tips, o6, s_sol, s_prefix);
                for (int c = 0; c < 6; c++) s_jac[(pb->tips[t].out_index * 6 + c) * D + gidx] = o6[c];
            }
            p_wave_sync();
            if (tid == 0) {
                double* s_x = lds + L.x;
                pinv_solve_lds(s_jac, 6 * T, D, s_b, s_x, lds + L.A, lds + L.Vm, lds + L.sig);  // (:117)
                int icol = 0;
                for (int gidx = 0; gidx < D; gidx++) {  // apply joint deltas and clip (:120-131)
                    const int k = pb->op_of_gene[gidx];
                    double v = s_sol[k] + s_x[icol];
                    if (!__builtin_isfinite(v)) continue;  // (the column index is not advanced, as in the reference)
                    s_sol[k] = clip_op(v, k);
                    icol++;
                }
            }
            p_wave_sync();
            for (int k = tid; k < n_ops; k += nth) s_best[k] = s_sol[k];
            p_wave_sync();
        }
        steps++;
        // ik_parallel.h:173-181 on getSolution()
        const FitCheck fc = exact_fitness_check(pb, XV{s_best, 1}, qc, s_slots, sp.dpos, sp.drot, sp.dtwist, 1, s_prefix);
        final_fit = fc.fitness;
        success = fc.ok != 0;
        if (success) {
            if (a.first_success && tid == 0) p_atomic_min(a.first_success + q, (unsigned int)steps);  // ik_parallel.h:176-177 `finished = 1`
            break;
        }
        if (sp.timeout_ticks != 0ull || a.first_success) {
            if (tid == 0) {
                const bool expired_now = sp.timeout_ticks != 0ull && p_wall_clock() >= deadline;
                const bool overtaken = a.first_success && p_atomic_load(a.first_success + q) <= (unsigned int)steps;  // (see solve_body)
                s_ex[6] = (expired_now || overtaken) ? 1.0 : 0.0;
            }
            p_wave_sync();
            const bool stop = s_ex[6] != 0.0;
            p_wave_sync();
            if (stop) break;
        }
    }
    double rank_fit = final_fit;
    if (success && pb->n_secondary > 0) rank_fit = final_fit + secondary_fitness(pb, XV{s_best, 1}, qc);
    double* out = a.solutions + unit * (uint64_t)V;
    for (int i = tid; i < V; i += nth) out[i] = s_seed[i];
    p_wave_sync();
    if (steps > 0)
        for (int k = tid; k < n_ops; k += nth)
            if (pb->ops[k].gene >= 0) out[pb->ops[k].var] = s_best[k];
    if (tid == 0) {
        a.fitness[unit] = rank_fit;
        a.success[unit] = success ? 1 : 0;
        a.steps[unit] = steps;
    }
}
