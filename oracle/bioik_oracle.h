/*
 * bioik_oracle.h — C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a CPU restatement of the reference algorithm
 * (TAMS-Group/bio_ik, bio2_memetic path).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; the product (bio_ik_amd/, include/) never does.
 *
 * PARITY STATUS: PINNED against the reference's own code.  The reference cannot be built with its own build system here
 * (ROS / MoveIt / tf2 / KDL / Eigen are not on disk), but its hot-path sources — include/bio_ik/{frame,goal,goal_types,
 * robot_info}.h, src/forward_kinematics.h, src/problem.{h,cpp}, src/ik_base.h, src/ik_evolution_2.cpp — compile UNMODIFIED,
 * from where they lie, against the minimal stand-in third-party headers of oracle/ref_shim (`make -C oracle ref` ->
 * oracle/_ref/libbioik_ref.so, driver oracle/ref_driver.cpp).  tests/test_oracle_vs_reference.py compares this restatement
 * with that library and with the fixtures generated from it (tests/golden/reference_golden.npz), BIT FOR BIT:
 *   (1) frame algebra, RobotInfo, exact FK, mutation approximator, all goal costs, success test;
 *   (2) whole IKEvolution2 trajectories (bio2, bio2_memetic, bio2_memetic_l; PR2-like arm, two-arm + torso, 31-DOF snake)
 *       driven by the reference's own random sources (std::minstd_rand + tables + XORShift64);
 * plus, independently of the reference tree: the two properties of the reference's test/utest.cpp, a NumPy float64 /
 * longdouble FK, closed-form known answers for the goal costs and Random123 known-answer vectors for Philox.
 * What remains a restatement (the third-party libraries themselves are absent): tf2 vector / quaternion helpers,
 * KDL::diff / Equal / Rotation, MoveIt's RobotModel container — oracle/ref_shim/README.md.
 *
 * Inputs use the PODs of include/bioik_hip.h so that the oracle and the HIP path consume identical data.
 */
#ifndef BIOIK_ORACLE_H
#define BIOIK_ORACLE_H

#include "../include/bioik_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rng back-ends of the solver */
enum {
    ORC_RNG_REFERENCE = 0, /* std::minstd_rand + 8Mi-entry uniform/gauss tables + XORShift64 (reference src/ik_base.h:49-126) */
    ORC_RNG_COUNTER = 1    /* Philox2x32-10 counter RNG shared bit-exactly with the device                      */
};

const char* orc_last_error(void);
/* 0: libm sin/cos as in the reference (default); 1: bioik_sincos shared bit-for-bit with the device (orc_model.h) */
void orc_set_trig_mode(int mode);
/* the sincos shared with the device kernels (bio_ik_amd/csrc/bioik_sincos.h), exposed for its accuracy test */
void orc_shared_sincos(size_t n, const double* x, double* s, double* c);
int orc_get_trig_mode(void);
/* diagnostics (soaks): candidates of the memetic line search with a gene at +-DBL_MAX so far -- an infinite step on a joint without limits (orc_model.h) */
unsigned long long orc_debug_unbounded_candidates(void);
/* 0: quirks Q1 (stale masked tips), Q4 (unstable pre-selection sort) and Q5 (the line search accepts a candidate with NaN genes, orc_evolution.h) fixed, as on the device (default);
 * 1: literal reference behaviour, for the trajectory comparison against oracle/_ref (orc_model.h) */
void orc_set_quirk_mode(int mode);

void* orc_model_create(const bioik_model_desc* desc);
void orc_model_destroy(void* model);
void* orc_problem_create(void* model, const bioik_problem_desc* desc);
void orc_problem_destroy(void* problem);

/* out[0..3] = D (active variables), T (tips), P (params per query), V (robot variables) */
int orc_problem_info(void* problem, int32_t* out4);
int orc_problem_active_variables(void* problem, int32_t* out);
int orc_problem_tip_links(void* problem, int32_t* out);
/* RobotInfo per robot variable: out[v*6 + {0..5}] = clip_min clip_max span min max max_velocity_rcp */
int orc_model_robot_info(void* model, double* out);
/* minimal_displacement_factors per gene (problem.cpp:207-225) */
int orc_problem_velocity_weights(void* problem, double* out);

/* ---- L1: frame.h ---- */
void orc_quat_mul_vec(const double* q4, const double* v3, double* out3);
void orc_quat_mul_quat(const double* p4, const double* q4, double* out4);
void orc_frame_concat(const double* a7, const double* b7, double* out7);
void orc_frame_invert(const double* a7, double* out7);
void orc_frame_change(const double* a7, const double* b7, const double* c7, double* out7);
void orc_normalize_fast(double* q4);
void orc_frame_twist(const double* a7, const double* b7, double* out6);
/* utils.h:348-367 linear_int_distribution<size_t>(n) driven by std::mt19937(seed): histogram of `iters` draws */
void orc_linear_int_distribution_hist(uint32_t seed, uint32_t n, uint32_t iters, double* hist);

/* ---- L2: forward_kinematics.h ---- */
/* exact FK of full variable vectors: tip_frames [n][T][7]; global_frames (optional) [n][n_links][7] */
int orc_fk(void* problem, size_t n, const double* vars, double* tip_frames, double* global_frames);
/* exact FK of genotypes (inactive variables from seed) */
int orc_fk_genes(void* problem, size_t n, const double* seed, const double* genes, double* tip_frames);
/* tip-local Jacobian [6T][D] row-major at base_genes (forward_kinematics.h:600-730) */
int orc_jacobian(void* problem, const double* seed, const double* base_genes, double* jac);
/* mutation approximator tables: tip_frames [T][7], deltas [T][D][7], mask [T][D] */
int orc_approximator(void* problem, const double* seed, const double* base_genes, double* tip_frames, double* deltas,
                     int32_t* mask);
/* linear phenotypes for n genotypes: frames [n][T][7] (forward_kinematics.h:1172-1233) */
int orc_approx_eval(void* problem, const double* seed, const double* base_genes, size_t n, const double* genes,
                    double* frames);

/* ---- L4: problem.cpp / goal_types.h ---- */
/* goal cost on given tip frames [T][7] and genes [D]: primary and secondary sums */
int orc_fitness_frames(void* problem, const double* seed, const double* goal_params, const double* frames,
                       const double* genes, double* primary, double* secondary);
int orc_fitness(void* problem, int fk_mode, size_t n, const double* seed, const double* goal_params,
                const double* base_genes, const double* genes, double* primary, double* secondary);
int orc_check(void* problem, const bioik_solve_params* params, size_t n, const double* seed, const double* goal_params,
              const double* genes, int32_t* ok);
/* pose-goal twist (6 numbers, goal frame) used by the dtwist test: problem.cpp:316-323 */
void orc_pose_twist(const double* goal7, const double* tip7, double* out6);

/* ---- counter RNG ---- */
void orc_philox2x32(uint32_t key, uint32_t c0, uint32_t c1, uint32_t* out2);
void orc_philox4x32(const uint32_t* key2, const uint32_t* ctr4, uint32_t* out4);
uint32_t orc_child_word(uint32_t key, uint32_t ctr1, uint32_t child, uint32_t w); /* random word w of a child of the stream (key, ctr1) */
double orc_counter_gauss32(uint32_t word);  // the Gaussian of one random word (orc_rng.h)
double orc_counter_uniform(uint32_t key, uint32_t c0, uint32_t c1);
uint32_t orc_query_key(uint64_t seed, uint64_t query, uint32_t island);

/* the restated reference random sources (ORC_RNG_REFERENCE), probed in the order reproduce()/step() consume them */
int orc_reference_random_probe(int seed, size_t n_gauss, double* gauss, size_t n_index, uint64_t* index16, size_t n_fast, double* fast, size_t n_rng,
                               double* rng_uniform);

/* ---- L3: ik_evolution_2.cpp ---- */
int orc_reproduce_counter(void* problem, int population, uint32_t rng_key, int species, uint32_t generation,
                          const double* parents, double* children_genes, double* children_gradients);

/* stateful single-island solver, for step-level parity */
void* orc_solver_create(void* problem, const bioik_solve_params* params, int rng_mode, uint32_t rng_key,
                        const double* seed, const double* goal_params);
void orc_solver_destroy(void* solver);
int orc_solver_step(void* solver);
/* state: species_genes [2][2][2][D] (species, individual, {genes,grads}, gene), species_fitness [2],
 * solution [V], solution_fitness [1] */
int orc_solver_state(void* solver, double* species_genes, double* species_fitness, double* solution,
                     double* solution_fitness);
/* exact FK of the current solution + checkSolution + fitness, as ik_parallel.h:173-181 */
int orc_solver_check(void* solver, int32_t* success, double* fitness);

/* batched solve (per query: islands, best-of selection of ik_parallel.h:220-269); n_threads queries in parallel.
 * timeout_s > 0 switches every island to the reference's wall-clock termination (ik_parallel.h:160-168) */
int orc_solve_batch(void* problem, const bioik_solve_params* params, int rng_mode, size_t n, const double* seeds,
                    const double* goal_params, double* solutions, double* fitness, int32_t* success, int32_t* steps,
                    int n_threads, double timeout_s, uint64_t first_query_index);

/* kinematics_plugin.cpp:580-613 angle wrapping (+ clamp into bounds, :616) on a full variable vector */
int orc_wrap_angles(void* problem, const double* seed, double* state);

#ifdef __cplusplus
}
#endif
#endif
