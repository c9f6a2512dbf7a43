// bioik_hip.hip — the C-ABI of include/bioik_hip.h: __global__ entry points for gfx950 and the thin host shim
// that owns device memory and launches them.  No torch, no C++ types across the boundary.
//
// The library is built by hipcc for gfx950 and refuses to do anything without a HIP device; there is no CPU execution path in it.
// (The test-suite compiles this file a second time with a substitute back end that lives under tests/hostsim and arrives through
// BIOIK_BACKEND_HEADER / BIOIK_PLATFORM_HEADER; the product build defines neither.)
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "bioik_compile.h"
#include "bioik_gradient.h"
#include "bioik_kernels.h"

static_assert((int)G_POSITION == (int)BIOIK_GOAL_POSITION && (int)G_POSE == (int)BIOIK_GOAL_POSE && (int)G_CONE == (int)BIOIK_GOAL_CONE &&
                  (int)G_JOINT_VARIABLE == (int)BIOIK_GOAL_JOINT_VARIABLE && (int)G_MINIMAL_DISPLACEMENT == (int)BIOIK_GOAL_MINIMAL_DISPLACEMENT &&
                  (int)G_AVOID_JOINT_LIMITS == (int)BIOIK_GOAL_AVOID_JOINT_LIMITS,
              "goal opcodes out of sync with include/bioik_hip.h");
static_assert((int)FK_LINEAR == (int)BIOIK_FK_LINEAR && (int)FK_EXACT == (int)BIOIK_FK_EXACT, "fk modes out of sync");

using bioik::Error;

static thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------------------
// back end: memory + launch
// ------------------------------------------------------------------------------------------------------------
#if defined(BIOIK_BACKEND_HEADER)
#include BIOIK_BACKEND_HEADER  // test builds only (tests/hostsim): a substitute back end; the product never defines it
#else
typedef hipStream_t stream_t;
#define HIP_CHECK(expr)                                                                                       \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) throw Error(BIOIK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
static int be_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
static void be_set_device(int d) { HIP_CHECK(hipSetDevice(d)); }
// what the launcher sizes its mappings to (MI355X: 256 CUs in 8 XCDs, 160 KiB of LDS per CU)
struct DeviceInfo {
    size_t lds_cu = 160 * 1024;
    int cus = 256, xcds = 8;
};
static DeviceInfo be_device_info(int d) {
    DeviceInfo di;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess) {
        if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) di.lds_cu = prop.maxSharedMemoryPerMultiProcessor;
        if (prop.multiProcessorCount > 0) di.cus = prop.multiProcessorCount;
    }
    int x = 0;
    if (hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, d) == hipSuccess && x > 0 && x <= 16) di.xcds = x;
    return di;
}
static int be_get_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
static void* be_alloc(size_t bytes) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
    return p;
}
static void be_free(void* p) {
    if (p) (void)hipFree(p);
}
static void* be_alloc_pinned(size_t bytes) {  // page-locked host memory: hipMemcpyAsync from / to it is one DMA, no staging copy
    void* p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault));
    return p;
}
static void be_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}
// stream-ordered scratch: launches of one problem handle on different streams never share it
static void* be_alloc_async(size_t bytes, stream_t s) {
    void* p = nullptr;
    HIP_CHECK(hipMallocAsync(&p, bytes ? bytes : 8, s));
    return p;
}
static void be_free_async(void* p, stream_t s) {
    if (p) (void)hipFreeAsync(p, s);
}
static bool be_stream_capturing(stream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) return false;
    return st != hipStreamCaptureStatusNone;
}
static void be_h2d(void* d, const void* h, size_t bytes, stream_t s) {
    if (bytes) HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
}
static void be_d2h(void* h, const void* d, size_t bytes, stream_t s) {
    if (bytes) HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
}
static void be_sync(stream_t s) { HIP_CHECK(hipStreamSynchronize(s)); }
// The words a solve needs (re)set in stream order -- the hand-over's counters, the launch clock, the islands' first-success words -- are written by a KERNEL
// of this library, not by hipMemsetAsync: a hipGraph that holds a memset node in front of these kernels replays correctly once and writes to a wild address
// from its second replay on under this runtime's AQL packet capture (ROCm 7.2; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 cures it, so does having no memset node:
// profiles/r05_graph_replay_root_cause.log, tools/graph_replay_raw_probe.py).  g_memset_nodes (BIOIK_SOLVE_MEMSET_NODES=1) brings the memsets back for that probe.
__global__ void __launch_bounds__(256) k_fill_words(unsigned int* p, size_t n, unsigned int value) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = value;
}
static bool g_memset_nodes = false;
static void be_fill_async(void* p, size_t bytes, int byte_value, stream_t s) {  // bytes: a multiple of four
    if (bytes == 0) return;
    if (g_memset_nodes) {
        HIP_CHECK(hipMemsetAsync(p, byte_value, bytes, s));
        return;
    }
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(k_fill_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned int*)p, n, (unsigned int)(byte_value & 0xff) * 0x01010101u);
    HIP_CHECK(hipGetLastError());
}
__global__ void k_read_clock(unsigned long long* out) { *out = wall_clock64(); }
// the device's constant 100 MHz clock as of "now" (a one-lane kernel on a stream of its own and a wait for it: ~30 us; launch_solve calls it rarely).
// The stream and the word the kernel writes exist once per DEVICE for the life of the process (a map under a mutex) -- whatever thread asks, and however often the
// caller's threads come and go: per-thread copies keyed on the last device leaked a stream and a pinned block whenever a thread alternated between two devices
// (the plugin's shards) or was created per call (bioik_solve_batch_multi's workers).
struct ClockReader {
    hipStream_t stream = nullptr;
    unsigned long long* word = nullptr;  // (page-locked and mapped: the kernel writes it, the host reads it after the wait)
};
static std::mutex g_clock_mtx;
static std::map<int, ClockReader> g_clock_readers;
static unsigned long long be_device_clock_now() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_clock_mtx);
    ClockReader& r = g_clock_readers[dev];
    if (!r.stream) {
        HIP_CHECK(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
        HIP_CHECK(hipHostMalloc((void**)&r.word, 64, hipHostMallocDefault));
    }
    hipLaunchKernelGGL(k_read_clock, dim3(1), dim3(1), 0, r.stream, r.word);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(r.stream));
    return *(volatile unsigned long long*)r.word;
}
// wall time of what `enqueue` puts on stream s, in ms (two events and a wait for the second): the launcher's measured mapping choice (solve_dispatch)
static bool be_can_time() { return true; }
template <class F>
static double be_time_ms(stream_t s, F&& enqueue) {
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    try {
        HIP_CHECK(hipEventRecord(e0, s));
        enqueue();
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    } catch (...) {
        (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
        throw;
    }
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    return (double)ms;
}
static void be_zero_async(void* p, size_t bytes, stream_t s) { be_fill_async(p, bytes, 0, s); }
static void be_fill_ff_async(void* p, size_t bytes, stream_t s) { be_fill_async(p, bytes, 0xff, s); }
static stream_t be_stream_create() {
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
static void be_stream_destroy(stream_t s) {
    if (s) (void)hipStreamDestroy(s);
}

#ifndef BIOIK_SOLVE_WAVES_PER_SIMD
#define BIOIK_SOLVE_WAVES_PER_SIMD 3
#endif
// two flavours of the solver (bioik_platform.h, pb_flavour): k_solve_lean for problems without floating / planar joints whose ops
// meet the genes in gene order (every BASELINE.json configuration), k_solve for everything
__global__ void __launch_bounds__(256, BIOIK_SOLVE_WAVES_PER_SIMD) k_solve(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<false>(a, blockIdx.x, lds);
}
__global__ void __launch_bounds__(256, BIOIK_SOLVE_WAVES_PER_SIMD) k_solve_lean(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true>(a, blockIdx.x, lds);
}
// the lean flavour with the children computed where they are read (no genotype columns in LDS: the LDS-bound problems, C3 / C4)
__global__ void __launch_bounds__(256, BIOIK_SOLVE_WAVES_PER_SIMD) k_solve_lean_cl(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true, true>(a, blockIdx.x, lds);
}
// The same body under the register budget of FOUR wavefronts per SIMD (128 VGPRs instead of 168).  The hot loops of the computed-children
// flavour fit it; its once-per-step code then keeps some values in scratch memory, so a lone step is slower (110 against 102 us) and it
// only pays where the LDS footprint lets the fourth wavefront be resident AND the generation loops dominate a step: the launcher
// picks it when a CU holds at least 16 wavefronts of the problem and a lane walks eight or more children per generation (the
// 31-joint, 512-children configuration: +12 %; the first launch of a 7-joint, 128-children solve: -2 %; profiles/r03_ab_four_waves.log).
// (solve_body<.., SLIM>: the species record is read from LDS per generation instead of living in scratch memory across the chain walks: 67 -> 31 spilled values)
__global__ void __launch_bounds__(128, 4) k_solve_lean_cl4(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true, true, false, true, 2>(a, blockIdx.x, lds);
}
// k_solve_lean_cl4 with two HELPER wavefronts (solve_body<.., FIXED = 5>), for launches that leave most of the chip idle -- a single pose of the plugin, a few
// hundred queries, the stragglers a chip-filling call hands over when the chip runs empty: wavefronts 0 and 1 are k_solve_lean_cl4's (a species each), wavefronts
// 2 and 3 walk half of every generation's children for them, so that a generation's walks take a wavefront half the instructions.  A lone wavefront issues one
// instruction per ~4.3 cycles whatever it does and k_solve_lean_cl4 keeps two of a CU's four SIMDs busy: this keeps four.  Hand-overs between the wavefronts
// are words in LDS, the helpers reach no barrier (profiles/r05_helped_kernel.log)
__global__ void __launch_bounds__(256, 4) k_solve_lean_cl4h(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true, true, false, true, 5>(a, blockIdx.x, lds);
}
// The computed-children kernel for BIOIK_SCHEDULE_THROUGHPUT: ONE wavefront per query (both species on its halves; the compiler knows it and drops the
// barriers) under the register budget of four wavefronts per SIMD
// (solve_body<.., DENSE>: what the launcher guarantees for this kernel -- 64 lanes, the species on the halves of the wavefront, exact FK, children in
// pairs, no secondary goal -- is known at compile time, and the fitness values of a generation cross its walks in LDS instead of in registers)
#ifndef BIOIK_DENSE_WAVES
#define BIOIK_DENSE_WAVES 4
#endif
__global__ void __launch_bounds__(64, BIOIK_DENSE_WAVES) k_solve_lean_cl64w4(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true, true, false, true, 1>(a, blockIdx.x, lds);
}
// Populations of up to 32 children per species with LINEARISED phenotypes (the reference's own parameters: 16 children, RobotFK_Mutator): both species on
// the halves of one wavefront, children computed where they are read, under the register budget of four wavefronts per SIMD (solve_body<.., FIXED = 3>).
// Such solves are bound by the latency of their single-individual phases (linearisation, line search, ranking), not by arithmetic: sixteen queries
// per CU instead of twelve (profiles/r04_ab_small_population_kernel.log)
__global__ void __launch_bounds__(64, 4) k_solve_lean_lin(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true, true, false, true, 3>(a, blockIdx.x, lds);
}
// Computed children with both species of a query on the halves of one wavefront AND secondary goals: the children of the two species are walked as one list over
// the 64 lanes (solve_body<.., JOINT>), so that the wavefront does not wait for the longer of two random prefixes (C3: +7 %, profiles/r03_ab_joint_walk.log) -- under
// the register budget of four wavefronts per SIMD (FIXED = 4: fitness values parked in LDS, accessors rebuilt behind the walks).  (Its 168-register sibling
// k_solve_lean_clj of rounds 3 - 5 served the one LDS band in which a CU holds exactly twelve queries; retired in round 6: this build runs there too.)
__global__ void __launch_bounds__(64, 4) k_solve_lean_clj4(SolveArgs a) {
    extern __shared__ double lds[];
    solve_body<true, true, true, true, 4>(a, blockIdx.x, lds);
}
// the point solvers gd_c / jac (bioik_gradient.h): one wavefront per query
__global__ void __launch_bounds__(64) k_solve_point(SolveArgs a) {
    extern __shared__ double lds[];
    point_body(a, blockIdx.x, lds);
}
__global__ void k_select(SelectArgs a) { select_body(a, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x); }
__global__ void __launch_bounds__(64) k_select_wave(SelectArgs a) { select_coop(a, (uint64_t)blockIdx.x, (int)threadIdx.x); }  // a wavefront per query: calls of few queries with many islands
__global__ void __launch_bounds__(256) k_eval_fk(EvalArgs a) {
    extern __shared__ double lds[];
    eval_fk_body(a, blockIdx.x, lds);
}
__global__ void __launch_bounds__(256) k_eval_fitness(EvalArgs a) {
    extern __shared__ double lds[];
    eval_fitness_body(a, blockIdx.x, lds);
}
__global__ void __launch_bounds__(256) k_eval_approximator(EvalArgs a) {
    extern __shared__ double lds[];
    eval_approximator_body(a, lds);
}
__global__ void __launch_bounds__(256) k_eval_reproduce(EvalArgs a) {
    extern __shared__ double lds[];
    eval_reproduce_body(a, blockIdx.x, lds);
}
__global__ void __launch_bounds__(256) k_eval_check(EvalArgs a) {
    extern __shared__ double lds[];
    eval_check_body(a, blockIdx.x, lds);
}
__global__ void __launch_bounds__(256) k_eval_arith(ArithArgs a) { arith_body(a, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x); }
__global__ void __launch_bounds__(256) k_stream_fitness(StreamArgs a) {
    extern __shared__ double lds[];
    stream_fitness_body(a, blockIdx.x, lds);
}
// a launch: KERNEL with `args`; BODYCALL names the same kernel body as a function of (b_, l_) = (block index, LDS base) for back ends
// that call it directly
#define LAUNCH(KERNEL, BODYCALL, grid, block, lds, stream, args)                                      \
    do {                                                                                              \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)(grid)), dim3(block), lds, stream, args);           \
        HIP_CHECK(hipGetLastError());                                                                 \
    } while (0)
// more than 64 KiB of dynamic LDS per workgroup must be allowed explicitly
static void be_allow_lds(size_t bytes) {
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean_cl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean_cl4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean_cl64w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean_lin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean_clj4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_solve_lean_cl4h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}
#endif

// ------------------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------------------
struct bioik_model {
    bioik::HostModel host;
    int device;
    DeviceInfo dev;  // of `device`: LDS per CU, CUs, XCDs (hipDeviceProp_t)
    bioik_model(const bioik_model_desc& d, int dv) : host(d), device(dv), dev(be_device_info(dv)) {}
};
struct bioik_problem {
    bioik_model* model;
    bioik::HostProblem host;
    DevProblem* d_pb = nullptr;
    // Host-pointer solves (bioik_solve_batch, bioik_solve_batch_submit / _wait) go through one of kIoSlots slots, each with a grow-only
    // device arena, its page-locked host mirror (one DMA in, one DMA out per solve) and its own stream: up to kIoSlots batches of one
    // handle are in flight together, the tail of one solve behind the bulk of the next.  Streams are created at a
    // slot's first use, not with the handle: HIP multiplexes streams onto four hardware queues per device, and a stream that a
    // device-pointer caller never uses would push one of the caller's own streams onto a shared queue.
    static constexpr int kIoSlots = 6;  // (three solves in flight fill the chip under BIOIK_SCHEDULE_LATENCY, six under BIOIK_SCHEDULE_THROUGHPUT)
    struct IoSlot {
        void* dev = nullptr;
        void* host = nullptr;
        size_t bytes = 0;
        stream_t stream = nullptr;
        bool stream_made = false;
        // the solve in flight on this slot, if any: where its results go when it completes
        bool pending = false;
        uint64_t ticket = 0;
        size_t n = 0, o_sol = 0, o_fit = 0, o_suc = 0, o_steps = 0;
        double *solutions = nullptr, *fitness = nullptr;
        int32_t *success = nullptr, *steps = nullptr;
        // a solve of this slot that ended in a device error: remembered for ITS ticket's wait (the slot itself is free again)
        uint64_t failed_ticket = 0;
        int failed_code = 0;
        std::string failed_message;
    } io[kIoSlots];
    uint64_t next_ticket = 1;  // ticket t runs on slot t % kIoSlots
    uint64_t first_query = 0;
    // The caller's timeout counts from the call's SUBMISSION (ik_parallel.h:160, 200): the host reads the device's 100 MHz clock next to its own steady
    // clock (when the handle first needs it, and again when the pair is older than a few seconds: the two clocks drift by parts per million) and hands every
    // eager solve its deadline in device ticks (SolveArgs::deadline).  A solve captured into a hipGraph cannot know when it will be replayed: it counts from
    // its first workgroup's start, on a word of its own that a kernel in the graph zeroes -- one of kCaptureClocks per handle, never reused.
    bool clock_valid = false;
    std::chrono::steady_clock::time_point clock_host0;
    unsigned long long clock_dev0 = 0;
    unsigned long long* d_clocks = nullptr;
    unsigned clock_next = 0;
    static constexpr unsigned kCaptureClocks = 64;
    // SolveArgs::error: one word per host-pointer slot and one for the device-pointer entry, in page-locked host memory the kernels can write: a rendezvous between
    // wavefronts that gave up (k_solve_lean_cl4h) sets the word of its call; the call's wait (or, for the device-pointer entry, the handle's next call) reports it
    unsigned int* h_error = nullptr;
    unsigned int* d_resident = nullptr;  // workgroups of the throughput schedule's launches that are running now (SolveArgs::resident), all streams of this handle
    // Scratch of a solve that needs some (per-island results, the state of handed-over units): one persistent buffer per (stream, purpose), grown when a
    // solve asks for more.  Solves on one stream follow each other, so they may share it; solves on other streams have their own.  (Stream-ordered
    // allocations did this until round 4 -- but a captured graph that contains an allocation and its release aborts on its second replay on this ROCm.)
    // Lifetime under hipGraph capture: a captured solve bakes the buffer's address into its graph, so a buffer a capture has used is PINNED -- it is never
    // released or handed to an eager solve again before the handle goes; the next eager solve on that stream retires it (the graphs keep it) and allocates its own.
    struct Scratch {
        void* base = nullptr;
        size_t capacity = 0;
        bool pinned = false;
        // purpose 0 (the islands' results): the control words of SolveArgs::island_done -- per query a count of the islands that have filed and the first_success word --
        // lie at the buffer's start, `ctl_n` of each, and are back in their resting state (0 / 0xffffffff) whenever no solve is running on them: the kernel that uses
        // them puts them back.  ctl_base != base (or a call of more than ctl_n queries): they have to be set up (a new buffer, a call of the other kind in between)
        void* ctl_base = nullptr;
        size_t ctl_n = 0;
    };
    std::map<std::pair<stream_t, int>, Scratch> scratch;
    // The launcher's MEASURED mapping choice (solve_dispatch): per kind of call -- (children per species, FK mode, islands or not) under the latency schedule --
    // the preset that ran a chip-filling call fastest, and what each eligible preset took (ms; < 0: not eligible)
    struct Tuned {
        int preset = -1;
        double ms[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    };
    std::map<std::tuple<int, int, int, int>, Tuned> tuned;  // (..., floor(log2(units)): a choice holds for calls of about the size it was measured on)
    // pinned buffers that eager solves have moved away from: captured graphs may still replay into them -- and LATER captures on the same stream take them again
    // (graphs captured from one stream share that stream's pinned buffer anyway), so that a caller who alternates eager calls and captures holds two buffers
    // per stream, not one more per alternation
    struct Retired {
        void* base;
        size_t capacity;
        stream_t stream;
        int purpose;
    };
    std::vector<Retired> retired_scratch;
    std::mutex mtx;
    bioik_problem(bioik_model* m, const bioik_problem_desc& d) : model(m), host(&m->host, d) {}
    ProbPtr pb() const { return (ProbPtr)d_pb; }
};

static int fail(const Error& e) {
    g_err = e.what();
    return e.code;
}
static int fail(const std::exception& e) {
    g_err = e.what();
    return BIOIK_ERR_INVALID_ARGUMENT;
}
#define API_BEGIN try {
#define API_END                      \
    return BIOIK_OK;                 \
    }                                \
    catch (const Error& e) {         \
        return fail(e);              \
    }                                \
    catch (const std::exception& e) { \
        return fail(e);              \
    }

// ------------------------------------------------------------------------------------------------------------
// Diagnostic switches (tools/README.md).  The environment is read ONCE, when the library is loaded, into this struct -- and again only when
// a test or a probe asks for it (bioik_debug_reload_switches): nothing on the launch path touches the environment.
// ------------------------------------------------------------------------------------------------------------
struct SolveSwitches {
    int threads = 0;            // BIOIK_SOLVE_THREADS: lanes per (query, island), 0 = the launcher's choice
    int store_children = -1;    // BIOIK_SOLVE_STORE_CHILDREN (-1: not set)
    int child_pairs = -1;       // BIOIK_SOLVE_CHILD_PAIRS
    int species_parallel = -1;  // BIOIK_SOLVE_SPECIES_PARALLEL
    int columnless = -1;        // BIOIK_SOLVE_COLUMNLESS (1: children computed where they are read, 2: ... and scored in pairs)
    bool general = false, general_set = false;  // BIOIK_SOLVE_GENERAL
    bool report = false;        // BIOIK_SOLVE_REPORT
    bool three_waves = false, four_waves = false, no_joint = false;
    int dense_handover = 0;     // BIOIK_SOLVE_DENSE_HANDOVER=K: a throughput solve hands the queries that pass K steps over to the latency mapping (0: never)
    int drain_below = 1024;     // BIOIK_SOLVE_DRAIN_BELOW=N: the dense kernel's stragglers leave for the latency mapping when fewer than N wavefronts of the handle's launches
                                // are left on the chip (0: never): isolated chip-filling calls of the latency schedule (launch_solve: latency_drain)
    bool drain_throughput = true;   // BIOIK_SOLVE_DRAIN_THROUGHPUT=0: the throughput schedule's solves do NOT hand their stragglers over (round 5: they do -- with k_solve_lean_cl4h as the
                                    // stragglers' kernel a stream's tail gains more than the bookkeeping costs: 20 timed steps +3.5 %, 60 steps +-0, profiles/r05_drain_throughput.log)
    int drain_min_units = 1025;  // BIOIK_SOLVE_DRAIN_MIN_UNITS=N: the latency schedule starts a call of N units and more (on 256 CUs; scaled to the chip) under the dense kernel and hands the
                                 // stragglers over; smaller calls run k_solve_lean_cl4 (or its helped build) from the start.  (With four and more islands per query, which stop
                                 // each other after a few steps, the dense kernel's long steps cost more than they bring below 3072 units: profiles/r05_drain_min_units.log)
    int drain_below_throughput = 512;  // BIOIK_SOLVE_DRAIN_BELOW_THROUGHPUT=N: ... when fewer than N wavefronts are left (a stream keeps the chip fuller than an isolated call: later)
    int drain_min_steps = 4;    // BIOIK_SOLVE_DRAIN_MIN_STEPS: ... and have run this many steps
    int drain_test = 0;         // BIOIK_SOLVE_DRAIN_TEST=n (parity suites): any solve, unit u leaves its first launch after 1 + hash(u) % n steps
    int helped = 1024;  // BIOIK_SOLVE_HELPED=N: launches of up to N (query, island) units of a problem k_solve_lean_cl4 covers without a secondary goal run its helped
                        // build (k_solve_lean_cl4h: four wavefronts per unit), as do the stragglers a chip-filling call hands over; 0: never
    int debug_flags = 0;  // BIOIK_SOLVE_DEBUG_FLAGS (tests): SolveArgs::debug_flags
    int fused_select = 1;  // BIOIK_SOLVE_FUSED_SELECT=0: the islands of a one-launch solve are reduced by a kernel of their own (and the first_success words set up by a fill kernel), as until round 5; -1: ... by k_select's lane per query whatever the island count (tests)
    int autotune = 1;  // BIOIK_SOLVE_AUTOTUNE: 1 (default) = the host-pointer entries time the eligible lane mappings on a handle's first chip-filling call of a kind and keep
                       // the fastest (solve_dispatch); 2 = the device-pointer entry does so too (it then waits for its stream once); 0 = the rules alone
    bool memset_nodes = false;  // BIOIK_SOLVE_MEMSET_NODES=1 (probe of the runtime's graph-replay defect): hipMemsetAsync instead of the library's own fill kernel
    bool capture_one_launch = false;  // BIOIK_SOLVE_CAPTURE_ONE_LAUNCH=1: a call on a stream that is being captured gets a one-launch mapping (round 4's rule)
    int sort_key_drop = 10;     // BIOIK_SOLVE_SORT_KEY_DROP=b (parity suites, 10 ... 52): the pre-selection's sort keys give up b low bits of a fitness, so that the exact path runs often
    int preselect = 1;          // BIOIK_SOLVE_PRESELECT: 1 (default) = the pre-selection's survivors by selection of the k-th key (select_threshold), 0 = by the sort of all children
    int tie_test_bits = 0;      // BIOIK_SOLVE_TIE_TEST_BITS=b (parity suites, 0 ... 52): the children's fitness loses b low bits before the selection, so that children tie
    bool two_phase_set = false, two_phase_init = false;
    std::vector<long> two_phase;  // BIOIK_SOLVE_TWO_PHASE: K or K1,K2,... (hand-overs after those steps), "init", 0 = never
    std::string phase_dump;       // BIOIK_PHASE_DUMP (profiling builds)
    std::string handover_dump;    // BIOIK_SOLVE_HANDOVER_DUMP=path (probes): every solve in several launches appends "scratch list_off count_off units" to this file
    bool manual() const { return threads > 0 || store_children >= 0 || child_pairs >= 0 || species_parallel >= 0 || columnless >= 0; }
};
static SolveSwitches parse_switches() {
    SolveSwitches w;
    auto geti = [](const char* name, int unset) {
        const char* e = std::getenv(name);
        return e ? std::atoi(e) : unset;
    };
    w.threads = geti("BIOIK_SOLVE_THREADS", 0);
    if (std::getenv("BIOIK_SOLVE_THREADS") && w.threads <= 0) w.threads = 64;
    w.store_children = geti("BIOIK_SOLVE_STORE_CHILDREN", -1);
    w.child_pairs = geti("BIOIK_SOLVE_CHILD_PAIRS", -1);
    w.species_parallel = geti("BIOIK_SOLVE_SPECIES_PARALLEL", -1);
    w.columnless = geti("BIOIK_SOLVE_COLUMNLESS", -1);
    if (const char* e = std::getenv("BIOIK_SOLVE_GENERAL")) w.general_set = true, w.general = std::atoi(e) != 0;
    w.report = std::getenv("BIOIK_SOLVE_REPORT") != nullptr;
    w.three_waves = std::getenv("BIOIK_SOLVE_THREE_WAVES") != nullptr;
    w.four_waves = std::getenv("BIOIK_SOLVE_FOUR_WAVES") != nullptr;
    w.no_joint = std::getenv("BIOIK_SOLVE_NO_JOINT") != nullptr;
    w.dense_handover = geti("BIOIK_SOLVE_DENSE_HANDOVER", 0);
    w.drain_below = geti("BIOIK_SOLVE_DRAIN_BELOW", 1024);
    w.drain_throughput = geti("BIOIK_SOLVE_DRAIN_THROUGHPUT", 1) != 0;
    w.drain_below_throughput = geti("BIOIK_SOLVE_DRAIN_BELOW_THROUGHPUT", 512);
    w.drain_min_units = geti("BIOIK_SOLVE_DRAIN_MIN_UNITS", 1025);
    w.drain_min_steps = geti("BIOIK_SOLVE_DRAIN_MIN_STEPS", 4);
    w.drain_test = geti("BIOIK_SOLVE_DRAIN_TEST", 0);
    w.sort_key_drop = geti("BIOIK_SOLVE_SORT_KEY_DROP", 10);
    w.preselect = geti("BIOIK_SOLVE_PRESELECT", 1) != 0 ? 1 : 0;
    w.tie_test_bits = geti("BIOIK_SOLVE_TIE_TEST_BITS", 0);
    if (w.tie_test_bits < 0 || w.tie_test_bits > 52) w.tie_test_bits = 0;
    w.capture_one_launch = geti("BIOIK_SOLVE_CAPTURE_ONE_LAUNCH", 0) != 0;
    w.memset_nodes = geti("BIOIK_SOLVE_MEMSET_NODES", 0) != 0;
    w.autotune = geti("BIOIK_SOLVE_AUTOTUNE", 1);
    w.helped = geti("BIOIK_SOLVE_HELPED", 1024);
    w.debug_flags = geti("BIOIK_SOLVE_DEBUG_FLAGS", 0);
    w.fused_select = geti("BIOIK_SOLVE_FUSED_SELECT", 1);
    if (w.sort_key_drop < 10 || w.sort_key_drop > 52) w.sort_key_drop = 10;
    if (const char* e = std::getenv("BIOIK_SOLVE_TWO_PHASE")) {
        w.two_phase_set = true;
        w.two_phase_init = std::strcmp(e, "init") == 0;
        for (const char* c = e; *c && !w.two_phase_init;) {
            char* end = nullptr;
            const long k = std::strtol(c, &end, 10);
            if (end == c) break;
            w.two_phase.push_back(k);
            c = *end == ',' ? end + 1 : end;
        }
    }
    if (const char* e = std::getenv("BIOIK_PHASE_DUMP")) w.phase_dump = e;
    if (const char* e = std::getenv("BIOIK_SOLVE_HANDOVER_DUMP")) w.handover_dump = e;
    return w;
}
static std::mutex g_switch_mtx;
static SolveSwitches apply_switches(SolveSwitches w) {  // (what the back end itself reads of them)
#if !defined(BIOIK_BACKEND_HEADER)
    g_memset_nodes = w.memset_nodes;
#endif
    return w;
}
static SolveSwitches g_switches = apply_switches(parse_switches());
static SolveSwitches switches() {
    std::lock_guard<std::mutex> lock(g_switch_mtx);
    return g_switches;
}

// lanes per (query, island).  The two species run concurrently on two lane groups when the workgroup has >= 2 wavefronts;
// each group gets one lane per child up to 128 lanes (pop=128 -> 256 lanes: 4 wavefronts on the 4 SIMDs of a CU).
static int solve_threads(const DevSolveParams& sp, uint64_t units, const SolveSwitches& sw, uint64_t cus) {
    int t;
    if (sw.threads > 0) {
        t = sw.threads;
    } else {
        // one wavefront per species, two children per lane and trip at pop=128: every wavefront is busy in every phase; with
        // children evaluated in pairs this is also as fast per query as one lane per child (measured: 165 vs 161 us per
        // step for a lone query, 95 k vs 84 k solves/s at batch 1024)
        int per_species = sp.lambda > 32 ? 64 : 32;
        t = 2 * per_species;
        // a launch that cannot fill the chip anyway (a single query of the plugin, a few hundred queries): one lane per child,
        // two wavefronts per species — a lone step takes 131 instead of 141 us, 1024 queries run 6 % faster; beyond ~768
        // queries the 128-lane mapping wins (profiles/r01_batch_sweep.log, r01_lone_workgroup_phases.log)
        if (sp.lambda >= 128 && units <= 3 * cus) t = 256;  // (768 on MI355X)
    }
    if (t < 64) t = 64;
    t = (t + 63) / 64 * 64;
    if (t > 256) t = 256;
    return t;
}

// every entry point runs on its handle's device and leaves the caller's current device as it found it
struct DeviceGuard {
    int prev;
    explicit DeviceGuard(int device) : prev(be_get_device()) {
        if (prev != device) be_set_device(device);
    }
    ~DeviceGuard() {
        if (be_get_device() != prev) {
            try {
                be_set_device(prev);
            } catch (...) {
            }
        }
    }
    DeviceGuard(const DeviceGuard&) = delete;
};

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) : p(be_alloc(bytes)) {}
    ~DevBuf() { be_free(p); }
    DevBuf(const DevBuf&) = delete;
    template <class T>
    T* as() const { return (T*)p; }
};

#ifndef BIOIK_SOLVE_WAVES_PER_SIMD
#define BIOIK_SOLVE_WAVES_PER_SIMD 3  // register budget of k_solve: wavefronts per SIMD (its __launch_bounds__)
#endif
static size_t lds_bytes(const bioik_problem* p, int nthreads, int lambda, int child_cols = 1, int groups = 1, int slot_sets = 1, bool exact = false, bool fit_park = false, bool helped = false) {
    const DevProblem& d = p->host.dev;
    return (size_t)make_layout(d.n_ops, d.V, d.P, d.T, d.n_slots, nthreads, lambda, d.n_secondary > 0 ? (exact ? 2 : 1) : 0, child_cols, groups, slot_sets, fit_park ? 1 : 0, lambda > 0 ? 1 : 0, helped ? 1 : 0).total * 8;  // (lambda > 0: a solve's layout; the function-level kernels keep their own)
}

// the timeout of a solve on the device clock (bioik_problem: clock_*)
static void set_deadline(bioik_problem* p, const DevSolveParams& sp, stream_t stream, SolveArgs& a) {
    a.launch_clock = nullptr, a.deadline = 0ull;
    if (sp.timeout_ticks == 0) return;
    if (be_stream_capturing(stream)) {
        if (p->clock_next >= bioik_problem::kCaptureClocks) throw Error(BIOIK_ERR_UNSUPPORTED, "more than 64 solves with a timeout captured into hipGraphs on one problem handle");
        a.launch_clock = p->d_clocks + p->clock_next++;
        be_zero_async(a.launch_clock, sizeof(unsigned long long), stream);
        return;
    }
    const auto now = std::chrono::steady_clock::now();
    if (!p->clock_valid || now - p->clock_host0 > std::chrono::seconds(4)) {
        try {
            // (the first reading of a process pays for the reader's stream, its page-locked word and the kernel's first launch -- milliseconds, with the kernel at their
            // END: paired with the middle of that interval the deadline of every call of the next four seconds came ~3 ms late, tools/timeout_probe.py.  So: one
            // reading to set things up, then the pairing from the tighter of two)
            if (!p->clock_valid) (void)be_device_clock_now();
            std::chrono::steady_clock::duration best = std::chrono::steady_clock::duration::max();
            for (int attempt = 0; attempt < 2; attempt++) {
                const auto t0 = std::chrono::steady_clock::now();
                const unsigned long long dev = be_device_clock_now();
                const auto t1 = std::chrono::steady_clock::now();
                if (t1 - t0 < best) best = t1 - t0, p->clock_dev0 = dev, p->clock_host0 = t0 + (t1 - t0) / 2;  // (the kernel ran somewhere between the two readings: +- half the ~30 us between them)
            }
            p->clock_valid = true;
        } catch (const Error&) {
            if (!p->clock_valid) throw;  // (a reading that is a few seconds old still serves: the clocks drift by parts per million)
        }
    }
    const double since = std::chrono::duration<double>(std::chrono::steady_clock::now() - p->clock_host0).count();
    a.deadline = p->clock_dev0 + (unsigned long long)(since * 1e8) + sp.timeout_ticks;
}

// ------------------------------------------------------------------------------------------------------------
// One solve = SolveLauncher: what the call was given, where its results go (scratch, result_arrays, select_islands), the lane mapping the rules below pick
// for it (choose_mapping), the kernel that mapping is compiled as (launch), and the launches it is cut into (plan_handovers, run).  launch_solve at the end
// puts them in order.
// ------------------------------------------------------------------------------------------------------------
struct SolveLauncher {
    bioik_problem* p;
    DevSolveParams sp;
    const size_t n;
    const double* d_seeds;
    const double* d_params;
    double* d_solutions;
    double* d_fitness;
    int32_t* d_success;
    int32_t* d_steps;
    stream_t stream;
    const SolveSwitches& sw;
    unsigned int* error_word;
    const DevProblem& dp;
    const size_t kLds;    // LDS of a CU (160 KiB on MI355X)
    const uint64_t kCus;
    uint64_t units = 0;
    void* island_ws = nullptr;  // (a stream-ordered fallback allocation of the per-island results, if any)
    bool fused_select = false;
    // the mapping (choose_mapping)
    int nth = 0, groups = 1;
    size_t lds = 0;
    bool exact = false, quat = false, manual = false, can_columnless = false, prefer_cl4 = false, small_sec_cl4 = false, small_linear = false, throughput = false, dense_ok = false,
         capturing = false, latency_drain = false, dense = false, lean = false, halves_ok = false;
    // the launches (plan_handovers)
    std::vector<int> handovers;  // the steps after which the unsolved units pass to the next launch (ascending)
    bool when_draining = false;  // ... or: whenever the chip runs empty (SolveArgs::resident), every unit from the step it is at

    SolveLauncher(bioik_problem* p_, const DevSolveParams& sp_in, size_t n_, const double* seeds, const double* params, double* solutions, double* fitness, int32_t* success,
                  int32_t* steps, stream_t s, const SolveSwitches& w, unsigned int* err)
        : p(p_), sp(sp_in), n(n_), d_seeds(seeds), d_params(params), d_solutions(solutions), d_fitness(fitness), d_success(success), d_steps(steps), stream(s), sw(w),
          error_word(err), dp(p_->host.dev), kLds(p_->model->dev.lds_cu), kCus((uint64_t)p_->model->dev.cus) {
        if (dp.n_secondary > 0 && sp.lambda < 2) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "population must be >= 2 when secondary goals are present");
        units = (uint64_t)n * sp.islands;
        if (units > 0x7fffffffull) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "too many (query, island) units for one launch: split the batch");
    }
    ~SolveLauncher() { be_free_async(island_ws, stream); }
    SolveLauncher(const SolveLauncher&) = delete;
    struct AsyncFree {
        void*& p;
        stream_t s;
        ~AsyncFree() { be_free_async(p, s); }
    };

    // scratch of this solve: the handle's persistent buffer for (stream, purpose), or -- while the stream is being captured and the buffer would have to
    // grow, or for the sixty-fifth stream of a handle -- a stream-ordered allocation released on every path out (null: nothing to release)
    void* scratch(int purpose, size_t bytes, void*& async_owned) {
        const auto key = std::make_pair(stream, purpose);
        auto it = p->scratch.find(key);
        if (it == p->scratch.end() && p->scratch.size() < 128) it = p->scratch.emplace(key, bioik_problem::Scratch{}).first;
        if (it != p->scratch.end()) {
            bioik_problem::Scratch& sc = it->second;
            if (be_stream_capturing(stream)) {
                for (const auto& r : p->retired_scratch)
                    if (r.stream == stream && r.purpose == purpose && r.capacity >= bytes) return r.base;  // (pinned by an earlier capture on this stream)
                if (sc.capacity >= bytes) {
                    sc.pinned = true;  // (the graph being captured keeps this address: see bioik_problem::Scratch)
                    return sc.base;
                }
                // (a stream-ordered allocation inside the capture would replay once and abort at the second replay on this ROCm: refuse, and say what to do)
                throw Error(BIOIK_ERR_UNSUPPORTED, "a solve captured into a hipGraph needs scratch memory this handle does not hold yet for the stream: run ONE eager call of the same size "
                                                   "(queries, islands) on the stream before capturing");
            } else {
                if (sc.pinned) p->retired_scratch.push_back(bioik_problem::Retired{sc.base, sc.capacity, stream, purpose}), sc = bioik_problem::Scratch{};
                if (sc.capacity >= bytes) return sc.base;
                // (stream-ordered on THIS stream, kept until it has to grow or the handle goes: no call here waits for the device -- hipMalloc / hipFree do, which
                // cost the first solves of a pipeline over the handle's six streams a factor of three)
                be_free_async(sc.base, stream);
                sc.base = nullptr, sc.capacity = 0;
                sc.base = be_alloc_async(bytes + bytes / 2, stream), sc.capacity = bytes + bytes / 2;
                return sc.base;
            }
        }
        async_owned = be_alloc_async(bytes, stream);
        return async_owned;
    }
    // where the launch writes its results: the caller's arrays, or (islands > 1) per-island arrays that are then reduced to them -- by the query's last island
    // itself (`fused`: SolveArgs::island_done; one launch in all) or by k_select behind the solve's launches (select_islands)
    void result_arrays(SolveArgs& args, bool fused) {
        if (sp.islands == 1) {
            args.solutions = d_solutions, args.fitness = d_fitness, args.success = d_success, args.steps = d_steps;
            return;
        }
        const size_t per = (size_t)dp.V * 8 + 8 + 4 + 4;
        const size_t ctl_n = fused ? std::max<size_t>(n, 256) : 0;  // (the control words of the fused form: room for calls of up to this many queries)
        const size_t ctl_bytes = (2 * ctl_n * 4 + 63) / 64 * 64;
        char* w = (char*)scratch(0, ctl_bytes + units * per + 64 + n * 4, island_ws);
        bioik_problem::Scratch* sc = nullptr;
        {
            auto it = p->scratch.find(std::make_pair(stream, 0));
            if (it != p->scratch.end() && it->second.base == (void*)w) sc = &it->second;
        }
        if (fused) {
            size_t have = sc && sc->ctl_base == (void*)w ? sc->ctl_n : 0;
            if (have < n) {  // a new buffer, a larger call, or a call of the other kind in between: set the words up (the kernels keep them from then on)
                be_fill_ff_async(w, ctl_n * 4, stream);
                be_zero_async(w + ctl_n * 4, ctl_n * 4, stream);
                have = ctl_n;
                if (sc) sc->ctl_base = (void*)w, sc->ctl_n = ctl_n;
            }
            if (sp.island_sync) args.first_success = (unsigned int*)w;
            args.island_done = (unsigned int*)(w + have * 4);
            args.final_solutions = d_solutions, args.final_fitness = d_fitness, args.final_success = d_success, args.final_steps = d_steps;
            w += (2 * have * 4 + 63) / 64 * 64;
            fused_select = true;
        } else if (sc) {
            sc->ctl_base = nullptr, sc->ctl_n = 0;  // (this call lays the buffer out its own way)
        }
        args.solutions = (double*)w, w += units * dp.V * 8;
        args.fitness = (double*)w, w += units * 8;
        args.success = (int32_t*)w, w += units * 4;
        args.steps = (int32_t*)w, w += units * 4;
        if (!fused && sp.island_sync) {  // "any island succeeds => all stop": the least step count of a passing island, per query (0xffffffff: none yet)
            args.first_success = (unsigned int*)w;
            be_fill_ff_async(args.first_success, n * 4, stream);
        }
    }
    void select_islands(const SolveArgs& args) {  // ik_parallel.h:220-269: the best island of every query
        if (sp.islands == 1 || fused_select) return;
        SelectArgs s;
        s.islands = sp.islands, s.V = dp.V, s.n = n;
        s.sync = sp.island_sync, s.pad = 0;
        s.isl_solutions = args.solutions, s.isl_fitness = args.fitness, s.isl_success = args.success, s.isl_steps = args.steps;
        s.solutions = d_solutions, s.fitness = d_fitness, s.success = d_success, s.steps = d_steps;
        if (sp.islands >= 8 && n <= 4096 && sw.fused_select >= 0) LAUNCH(k_select_wave, select_coop(s, b_, p_tid()), n, 64, 0, stream, s);
        else LAUNCH(k_select, select_body(s, b_ * 256 + (uint64_t)p_tid()), (n + 255) / 256, 256, 0, stream, s);
    }
    // gd / gd_r / gd_c / jac: one wavefront per (query, island), its own (small) LDS layout
    void solve_point() {
        const size_t lds_point = (size_t)make_point_layout(dp.n_ops, dp.V, dp.P, dp.T, dp.n_slots, dp.D, 64).total * 8;
        if (lds_point > 64 * 1024) throw Error(BIOIK_ERR_UNSUPPORTED, "problem too large for the gd / jac kernels (more than 64 KiB of LDS per query)");
        SolveArgs pa;
        pa.pb = p->pb(), pa.sp = sp, pa.seeds = d_seeds, pa.params = d_params;
        result_arrays(pa, false);
        pa.phase_cycles = nullptr;
        set_deadline(p, sp, stream, pa);
        LAUNCH(k_solve_point, point_body(pa, b_, l_), units, 64, lds_point, stream, pa);
        select_islands(pa);
    }
    void choose_mapping() {
        // Mapping of a (query, island) onto lanes.  Candidates: 128 lanes (one wavefront per species) with every child kept in LDS
        // and evaluated in pairs / kept / re-derived from the RNG, or 64 lanes (one wavefront, the species one after the other).  A CU
        // holds 160 KiB of LDS and, at this kernel's register budget, 12 wavefronts: the candidate that puts most wavefronts on a CU
        // wins; among equals the richer mapping wins when the CU is full (C2: 20 KiB per workgroup) and the leaner one when LDS is
        // what limits residency (C3, C4: measured +7 % and +26 % for 64 lanes, tools/mapping_sweep.py).
        nth = solve_threads(sp, units, sw, kCus);
        exact = sp.fk_mode == BIOIK_FK_EXACT;  // (the LDS layout of exact-FK solves is smaller, make_layout)
        if (sw.threads <= 0 && nth == 256 && lds_bytes(p, 256, sp.lambda, 1, 2, 1, exact) > 48 * 1024) nth = 128;  // LDS-heavy problem
        quat = dp.n_quat > 0;  // winners re-derived: their momentum is taken before the quaternion genes are renormalised
        manual = sw.manual();
        // children computed where they are read (no genotype columns in LDS): the lean flavour can, whenever it is chosen below
        can_columnless = dp.multi_op < 0 && dp.n_quat == 0 && dp.genes_follow_ops != 0 && dp.n_balance == 0 && !sw.general_set;
        sp.columnless = 0;
        // k_solve_lean_cl4's mapping first -- 128 lanes, a wavefront per species, children computed where they are read and walked in pairs, the kernel compiled
        // for exactly that under the budget of four wavefronts per SIMD (no register spills since round 4): for problems without a secondary goal whose
        // lanes get at least one pair of children per generation and whose LDS footprint lets sixteen wavefronts share a CU it beats the kernel with the
        // children kept in columns on every count (C2: lone step 94 -> 91 us, fixed work at 4096 queries +30 %, three solves in flight +26 %, an isolated
        // call +16 %: profiles/r04_ab_latency_schedule_kernel.log)
        const bool cl4_eligible = !manual && !sw.three_waves && can_columnless && exact && dp.serial_chain != 0 && !(dp.n_quat > 0) && dp.n_secondary == 0 &&
                                  sp.lambda >= 128 && lds_bytes(p, 128, sp.lambda, 0, 2, 2, exact, exact) * 8 <= kLds;
        // (... also for the launches that cannot fill the chip, where solve_threads asks for a lane per child: one query 0.928 against 0.942 ms, 256 queries 5.49 against
        // 5.74 ms, profiles/r04_small_batches.log)
        if (cl4_eligible && nth == 256) nth = 128;
        prefer_cl4 = cl4_eligible && nth == 128;
        // ... and, for launches small enough for its helped build (k_solve_lean_cl4h), the same mapping for serial chains WITH secondary goals (a 7-joint arm with a
        // MinimalDisplacementGoal, the 31-joint chain with AvoidJointLimitsGoal: the usual MoveIt configurations of bio_ik): the kernel pre-selects, the helper takes
        // half of the survivors' walks.  Chip-filling batches of such problems keep the mappings chosen below (the joint walk, the 128-register build by residency).
        small_sec_cl4 = !manual && !sw.three_waves && can_columnless && exact && dp.serial_chain != 0 && !(dp.n_quat > 0) && dp.n_secondary > 0 && sp.lambda >= 128 &&
                                   sw.helped > 0 && units <= (uint64_t)sw.helped && lds_bytes(p, 128, sp.lambda, 0, 2, 2, exact, exact) * 8 <= kLds;
        if (small_sec_cl4) nth = 128;
        if (prefer_cl4 || small_sec_cl4) {
            sp.species_parallel = 1, sp.child_cols = 1, sp.child_pairs = 1, sp.columnless = 1;
        } else if (!manual && nth == 128) {
            struct Cand {
                int nth, store, pairs, columnless;
            };
            // richest first: children kept in LDS and scored in pairs / kept / computed where they are read / one reusable column per lane
            const Cand cands[] = {{128, 1, 1, 0}, {128, 1, 0, 0}, {128, 0, 1, 1}, {128, 0, 0, 1}, {128, 0, 0, 0}, {64, 0, 1, 1}, {64, 0, 0, 1}, {64, 0, 0, 0}};
            const int kCuWaves = 4 * BIOIK_SOLVE_WAVES_PER_SIMD;  // wavefronts a CU holds at this kernel's register budget
            int best = -1, best_waves = -1;
            for (int i = 0; i < 8; i++) {
                const Cand& c = cands[i];
                if ((c.store || c.pairs) && quat) continue;
                if (c.columnless && !can_columnless) continue;
                if (c.pairs && sp.fk_mode != BIOIK_FK_EXACT) continue;
                const int groups_c = c.nth % 128 == 0 ? 2 : 1, G_c = c.nth / groups_c;
                const int cols = c.store ? (sp.lambda + G_c - 1) / G_c : (c.columnless ? 0 : 1);
                if (c.pairs && !c.columnless && cols < 2) continue;
                // (computed children in pairs: measured +1.5 % with eight children per lane and generation (C4), -4 % with two (C3))
                if (c.pairs && c.columnless && sp.lambda < 4 * G_c) continue;
                const size_t bytes = lds_bytes(p, c.nth, sp.lambda, cols, groups_c, c.pairs ? 2 : 1, exact, c.columnless && exact);
                if (bytes > kLds) continue;
                int waves = (int)(kLds / bytes) * (c.nth / 64);
                if (waves > kCuWaves) waves = kCuWaves;
                // full CU: first (richest) candidate wins; LDS-limited: a later (leaner) candidate wins ties
                if (waves > best_waves || (waves == best_waves && waves < kCuWaves)) best = i, best_waves = waves;
            }
            if (best < 0) throw Error(BIOIK_ERR_UNSUPPORTED, "problem needs more LDS per workgroup than a CU has");
            nth = cands[best].nth;
            sp.species_parallel = nth % 128 == 0 ? 1 : 0;
            const int G_b = nth / (sp.species_parallel ? 2 : 1);
            sp.child_cols = cands[best].store ? (sp.lambda + G_b - 1) / G_b : 1;
            sp.child_pairs = cands[best].pairs;
            sp.columnless = cands[best].columnless;
            // Problems with secondary goals score only a random prefix of the pre-selected children, so the phases besides the chain walk (the
            // pre-selection itself, selection, the memetic phase) weigh more; with both species of a query on the halves of ONE wavefront those
            // run once for the two.  Measured: C3 (128 children per species) +14 %, C4 (512: sixteen children per lane) -22 % (tools/c34_mapping_probe.sh).
            if (sp.columnless && dp.n_secondary > 0 && exact && sp.lambda >= 128 && sp.lambda <= 256 && dp.D < 32 &&
                lds_bytes(p, 64, sp.lambda, 0, 2, 2, exact, exact) * kCuWaves <= kLds) {
                nth = 64, sp.species_parallel = 1, sp.child_cols = 1, sp.child_pairs = 1;
            }
        } else {
            while (nth > 64 && lds_bytes(p, nth, sp.lambda, 1, 2, 1, exact) > 64 * 1024) nth -= 64;  // genotype columns scale with the lane count
            // two lane groups, one species each: whole wavefronts (128 / 256 lanes), or the two halves of one wavefront (64 lanes)
            // (small populations, <= 32 children per species: +65..80 % measured, tools/halfwave_sweep.sh; at 64 and more children
            // per species the sequential single wavefront or the two-wavefront mapping is as good or better)
            sp.species_parallel = (nth % 128 == 0 || (nth == 64 && sp.lambda <= 32 && dp.D < 32)) ? 1 : 0;  // (the memetic phase wants lane D of a group)
            if (sw.species_parallel >= 0) sp.species_parallel = (sw.species_parallel != 0 && (nth % 128 == 0 || (nth == 64 && dp.D < 32))) ? 1 : 0;
            const int groups_m = sp.species_parallel ? 2 : 1, G_m = nth / groups_m;
            sp.child_cols = (sp.lambda + G_m - 1) / G_m;
            if (quat) {
                sp.child_cols = 1;
            } else if (sw.store_children >= 0) {
                if (sw.store_children == 0) sp.child_cols = 1;
            } else if (lds_bytes(p, nth, sp.lambda, sp.child_cols, groups_m, 1, exact) > 48 * 1024) {
                sp.child_cols = 1;
            }
            // children two at a time per lane (two independent dependency chains): needs both columns of the pair and, for branching
            // trees, a second set of parked frames
            sp.child_pairs = (sp.child_cols >= 2 && sp.fk_mode == BIOIK_FK_EXACT && lds_bytes(p, nth, sp.lambda, sp.child_cols, groups_m, 2, exact) <= 64 * 1024) ? 1 : 0;
            if (sw.child_pairs == 0) sp.child_pairs = 0;
        }
        // small populations, linearised phenotypes (the reference's own parameters), both species on the halves of one wavefront: children computed where
        // they are read, and the kernel compiled for exactly that (k_solve_lean_lin: sixteen queries per CU)
        small_linear = !manual && !sw.three_waves && can_columnless && !exact && nth == 64 && sp.species_parallel && sp.lambda <= 32 && sp.solver == 0;
        if (small_linear) sp.columnless = 1, sp.child_cols = 1, sp.child_pairs = 0;
        // BIOIK_SCHEDULE_THROUGHPUT: the whole solve under the mapping that retires most steps per ms on a full chip -- both species of a query on the
        // halves of one wavefront, children computed where they are read and scored in pairs (the first launch's mapping of the two-launch solve
        // below) -- for callers that keep six or more batches in flight (include/bioik_hip.h; profiles/r03_inflight_and_schedule.log)
        throughput = sp.schedule == BIOIK_SCHEDULE_THROUGHPUT && !manual && can_columnless && exact && sp.lambda >= 128 && sp.lambda <= 256 && dp.D < 32 &&
                                dp.n_secondary == 0;
        // (nth / sp keep the LATENCY mapping: a throughput solve may hand its stragglers over to it, sw.dense_handover; its own launch takes the
        // dense mapping where it is made, `halves` below)
        // k_solve_lean_cl64w4 (solve_body<.., DENSE>): the 128-register build of that mapping, four wavefronts per SIMD
        // BIOIK_SCHEDULE_LATENCY, a batch beyond what the chip holds of k_solve_lean_cl4's workgroups (2048; 2048 queries: 7.27 ms alone, 7.51 ms this way) under its mapping: the dense kernel first -- every query of up to 4096
        // resident from the start, most steps retired per ms while the chip is full -- and, when the chip runs empty, the stragglers on to k_solve_lean_cl4 whose lone
        // step is a third shorter (SolveArgs::resident).  An isolated 4096-query call: 9.2 -> 8.5 ms (profiles/r04_drain_handover.log).  Streams of solves keep the
        // chip full, never see the hand-over and pay for its bookkeeping: the throughput schedule does without (BIOIK_SOLVE_DRAIN_THROUGHPUT=1: with).
        dense_ok = !manual && can_columnless && exact && sp.lambda >= 128 && sp.lambda <= 256 && dp.D < 32 && dp.n_secondary == 0 && !sw.three_waves && dp.multi_op < 0 &&
                              dp.n_quat == 0 && dp.genes_follow_ops != 0 && dp.n_balance == 0 && dp.serial_chain != 0;
        // (BIOIK_SOLVE_CAPTURE_ONE_LAUNCH=1: round 4's rule -- a call on a stream that is being captured gets a one-launch mapping.  The replay defect it worked around
        // was the runtime's memset NODE in front of the kernels, not the hand-over; the library fills its words with a kernel of its own now
        // and captured calls take the same mapping as eager ones)
        capturing = sw.capture_one_launch && be_stream_capturing(stream);
        latency_drain = sp.schedule != BIOIK_SCHEDULE_THROUGHPUT && prefer_cl4 && dense_ok && units >= (uint64_t)(sp.islands <= 2 ? sw.drain_min_units : 3072) * kCus / 256 && sw.drain_below > 0 && sp.max_steps > sw.drain_min_steps + 1 &&
                                   !sw.two_phase_set && !capturing;
        dense = (throughput || latency_drain) && dense_ok;
        if (sw.columnless > 0 && can_columnless) {
            sp.columnless = 1, sp.child_cols = 1;
            sp.child_pairs = (sw.columnless == 2 && sp.fk_mode == BIOIK_FK_EXACT) ? 1 : 0;  // 2: children scored two at a time
        }
        groups = sp.species_parallel ? 2 : 1;
        lds = lds_bytes(p, nth, sp.lambda, sp.columnless ? 0 : sp.child_cols, groups, sp.child_pairs ? 2 : 1, exact, sp.columnless && exact);
        if (lds > kLds) throw Error(BIOIK_ERR_UNSUPPORTED, "problem needs more LDS per workgroup than a CU has");
        lean = dp.multi_op < 0 && dp.n_quat == 0 && dp.genes_follow_ops != 0 && dp.n_balance == 0;
        if (sw.general_set && sw.general) lean = false;
        halves_ok = lean && can_columnless && exact && sp.lambda >= 128 && dp.D < 32;  // the first launch's mapping exists for this problem
    }
    void report_mapping() const {  // diagnostics: the lane mapping and the residency it gives
        const LdsLayout L = make_layout(dp.n_ops, dp.V, dp.P, dp.T, dp.n_slots, nth, sp.lambda, dp.n_secondary > 0 ? (exact ? 2 : 1) : 0, sp.columnless ? 0 : sp.child_cols,
                                        groups, sp.child_pairs ? 2 : 1, (sp.columnless && exact) ? 1 : 0, 1);
        int n_rev = 0, n_pos = 0, n_rot = 0;  // revolute ops and how many of them the walk takes through a sparse form
        for (int k = 0; k < dp.n_chain_ops; k++)
            if (dp.ops[k].type == BIOIK_OP_REVOLUTE) n_rev++, n_pos += dp.ops[k].pos_kind != BIOIK_POS_GENERAL, n_rot += dp.ops[k].rot_kind != BIOIK_ROT_GENERAL;
        std::fprintf(stderr, "[bioik] joint program: %d revolute ops, sparse position %d, sparse rotation %d\n", n_rev, n_pos, n_rot);
        std::string prog;  // op (source, gene): the shape of the tree the walk takes
        for (int k = 0; k < dp.n_chain_ops; k++) {
            char buf[48];
            std::snprintf(buf, sizeof buf, " %d(%d,%d%s%s)", k, (int)dp.ops[k].src, (int)dp.ops[k].gene, dp.ops[k].load_slot >= 0 ? ",load" : "", dp.ops[k].save_slot >= 0 ? ",save" : "");
            prog += buf;
        }
        std::fprintf(stderr, "[bioik] joint program: prefix %d, serial %d, op(source,gene):%s\n", (int)dp.n_prefix, (int)dp.serial_chain, prog.c_str());
        std::fprintf(stderr, "[bioik] solve: ops %d genes %d tips %d slots %d | lanes %d species_parallel %d child_cols %d pairs %d columnless %d | LDS %zu B "
                     "(genotype columns %d, parked frames %d, per-group %d x %d) -> %d workgroups = %d wavefronts per CU\n",
                     dp.n_ops, dp.D, dp.T, dp.n_slots, nth, sp.species_parallel, sp.child_cols, sp.child_pairs, sp.columnless, lds, (L.slots - L.xcol) * 8,
                     (L.g_first - L.slots) * 8, L.g_stride * 8, groups, (int)(kLds / lds), (int)(kLds / lds) * (nth / 64));
    }
    void launch(const SolveArgs& args, int lanes, size_t lds_b) {
        // computed children: the 128-register build when a CU's LDS holds at least the 16 wavefronts it makes room for and a lane walks
        // at least eight children per generation (the generation loops, which fit the smaller budget, are then most of a step)
        const int group_lanes = lanes / (args.sp.species_parallel ? 2 : 1);
        // (k_solve_lean_cl4 is compiled for exactly this mapping -- solve_body<.., FIXED = 2> --: 128 lanes, a wavefront per species, exact FK, children in pairs)
        const bool cl4_mapping = lanes == 128 && args.sp.species_parallel && args.sp.child_pairs && args.sp.fk_mode == BIOIK_FK_EXACT && dp.serial_chain != 0;
        const bool four_waves = cl4_mapping && (((kLds / lds_b) * (size_t)(lanes / 64) >= 16 && (args.sp.lambda >= 8 * group_lanes || prefer_cl4 || small_sec_cl4) && !sw.three_waves) ||
                                                sw.four_waves);  // (diagnostic: the 128-register build wherever its mapping is the one in use)
        // both species on one wavefront, secondary goals, exact FK, children in pairs: the joint walk of the two species' children
        // (with a wavefront per species -- 128 lanes, C4 -- the same walk gains nothing: a wavefront that waits at a barrier costs no issue slots,
        // profiles/r03_ab_joint_walk.log)
        const bool joint = lanes == 64 && args.sp.species_parallel && args.sp.child_pairs && dp.n_secondary > 0 && args.sp.fk_mode == BIOIK_FK_EXACT &&
                           !sw.no_joint;
        const bool joint4 = joint;  // (k_solve_lean_clj4 wherever the joint walk applies)
        const bool dense_launch = lanes == 64 && dense && args.sp.species_parallel && args.sp.child_pairs && args.sp.columnless;  // (what solve_body<.., FIXED = 1> is compiled for)
        const bool lin_launch = small_linear && lanes == 64 && args.sp.species_parallel && args.sp.columnless && !args.sp.child_pairs && args.sp.fk_mode == BIOIK_FK_LINEAR;
        // k_solve_lean_cl4's helped build: launches that leave most of the chip idle (every unit gets four wavefronts instead of two), and the stragglers of a
        // chip-filling call (the list is a fraction of the grid)
        // (since its second version also for problems with secondary goals -- the 31-joint chain with AvoidJointLimitsGoal --: the helper takes the upper half of the
        // pre-selected children's walks)
        const bool helped_launch = lean && four_waves && !manual && !sw.four_waves && args.sp.columnless && sw.helped > 0 &&
                                   (units <= (uint64_t)sw.helped || (args.unit_list != nullptr && args.resident != nullptr));
        if (helped_launch) {
            const size_t lds_h = lds_b + 64;  // (make_layout: the sixteen words of the hand-overs)
            if (lds_h > 64 * 1024) be_allow_lds(lds_h);
            if (sw.report)
                std::fprintf(stderr, "[bioik] launch: k_solve_lean_cl4h, 256 lanes (two of the four wavefronts are helpers), %zu B of LDS, steps [%d, %d)\n", lds_h, (int)args.step_begin,
                             (int)(args.step_end < args.sp.max_steps ? args.step_end : args.sp.max_steps));
            LAUNCH(k_solve_lean_cl4h, (solve_body<true, true, false, true, 5>(args, b_, l_)), units, 256, lds_h, stream, args);
            return;
        }
        if (sw.report)
            std::fprintf(stderr, "[bioik] launch: %s, %d lanes, %zu B of LDS, steps [%d, %d)\n",
                         !lean ? "k_solve" : lin_launch ? "k_solve_lean_lin" : !args.sp.columnless ? "k_solve_lean" : joint ? "k_solve_lean_clj4" : dense_launch ? "k_solve_lean_cl64w4" : four_waves ? "k_solve_lean_cl4" : "k_solve_lean_cl",
                         lanes, lds_b, (int)args.step_begin, (int)(args.step_end < args.sp.max_steps ? args.step_end : args.sp.max_steps));
        if (lean && lin_launch)
            LAUNCH(k_solve_lean_lin, (solve_body<true, true, false, true, 3>(args, b_, l_)), units, lanes, lds_b, stream, args);
        else if (lean && args.sp.columnless && joint && joint4)
            LAUNCH(k_solve_lean_clj4, (solve_body<true, true, true, true, 4>(args, b_, l_)), units, lanes, lds_b, stream, args);
        else if (lean && args.sp.columnless && dense_launch)
            // the whole solve of a stream of batches under the dense mapping: sixteen queries per CU instead of twelve (+11 % with six solves in flight;
            // 30 values -- the lane's best two across the chain walk, a few kernel-lifetime ones -- then live in scratch memory;
            // profiles/r03_ab_dense_four_waves.log)
            LAUNCH(k_solve_lean_cl64w4, (solve_body<true, true, false, true, 1>(args, b_, l_)), units, lanes, lds_b, stream, args);
        else if (lean && args.sp.columnless && four_waves)
            LAUNCH(k_solve_lean_cl4, (solve_body<true, true, false, true, 2>(args, b_, l_)), units, lanes, lds_b, stream, args);
        else if (lean && args.sp.columnless)
            LAUNCH(k_solve_lean_cl, (solve_body<true, true>(args, b_, l_)), units, lanes, lds_b, stream, args);
        else if (lean)
            LAUNCH(k_solve_lean, solve_body<true>(args, b_, l_), units, lanes, lds_b, stream, args);
        else
            LAUNCH(k_solve, solve_body<false>(args, b_, l_), units, lanes, lds_b, stream, args);
    }
    // One launch, or two (SolveArgs::step_begin ...).  Measured on streams of distinct 4096-query PoseGoal batches (profiles/r02_two_launch_sweep.log):
    // split after the first step(), the state of the unsolved queries handed over through HBM, the first launch with both species of a query on
    // the halves of ONE wavefront and the children computed where they are read, a stream of batches runs 2.6 % faster and an isolated call 3.5 %
    // (later hand-overs give the same on uniformly seeded queries and cost up to 11 % on tracking seeds, most of which are solved within six steps).
    // The second launch takes the queries in the order the first finished them: solves in flight whose long-running queries would otherwise
    // coincide (the same batch solved again on another stream) do not lose the 8 % such a coincidence costs.
    // BIOIK_SOLVE_TWO_PHASE=K (or K1,K2,... / init) forces hand-overs after those steps for any problem (0: never) -- the parity suites run every
    // mapping through it.
    void plan_handovers() {
        if (sw.drain_test > 0 && sp.max_steps > 1 && sp.solver == 0) {
            when_draining = true;
            handovers.push_back(sp.max_steps);
        } else if (sw.two_phase_set) {  // "K" or "K1,K2,..." (experiments: more than one hand-over)
            if (sw.two_phase_init) handovers.push_back(0);  // (experiment: the first launch only initialises)
            for (const long k : sw.two_phase)
                if (k > (handovers.empty() ? 0 : handovers.back()) && k < sp.max_steps) handovers.push_back((int)k);
        } else if (throughput) {
            // the dense mapping retires most steps per ms but its steps are 2.5 x as long: the stragglers of a batch may pass to the latency mapping
            if (sw.dense_handover > 0 && sw.dense_handover < sp.max_steps) handovers.push_back(sw.dense_handover);
            else if (dense && prefer_cl4 && units >= 8 * kCus && sw.drain_throughput && sw.drain_below > 0 && sw.drain_below_throughput > 0 && sp.max_steps > sw.drain_min_steps + 1 && !capturing)
                when_draining = true, handovers.push_back(sp.max_steps);
        } else if (latency_drain) {
            when_draining = true, handovers.push_back(sp.max_steps);
        } else if (halves_ok && !manual && !prefer_cl4 && sp.lambda <= 256 && dp.n_secondary == 0 && units >= 8 * kCus && sp.max_steps >= 24 && !capturing) {
            // (not under k_solve_lean_cl4's mapping: there one launch is faster -- three in flight 9.9e5 against 9.3e5, an isolated call 8.9 against 9.3 ms,
            // profiles/r04_ab_latency_schedule_kernel.log)
            handovers.push_back(1);
        }
    }
    // the mapping with both species of a query on the halves of ONE wavefront, children computed where they are read and walked in pairs: the
    // first launch of a solve in several launches, and the whole of a throughput solve
    void halves(SolveArgs& aj, int& lanes, size_t& lds_j) {
        lanes = 64;
        aj.sp.species_parallel = 1, aj.sp.columnless = 1, aj.sp.child_cols = 1, aj.sp.child_pairs = 1;
        lds_j = lds_bytes(p, 64, sp.lambda, 0, 2, 2, exact, exact);
        if (lds_j > 64 * 1024) be_allow_lds(lds_j);
    }
    // the launch(es) of a solve whose arguments `a` are complete
    void run(const SolveArgs& a) {
        if (handovers.empty() && throughput) {
            SolveArgs a0 = a;
            int lanes = nth;
            size_t lds_0 = lds;
            halves(a0, lanes, lds_0);
            launch(a0, lanes, lds_0);
        } else if (handovers.empty()) {
            launch(a, nth, lds);
        } else {
            const size_t nh = handovers.size();
            const size_t carry_n = 9 * (size_t)(dp.n_ops > 0 ? dp.n_ops : 1) + 24;  // (solve_body: carry_n)
            const size_t list_bytes = (units * 4 + 63) / 64 * 64;
            const size_t list_off = (units * carry_n * 8 + 63) / 64 * 64, count_off = list_off + nh * list_bytes;
            void* ws_async = nullptr;
            AsyncFree ws_guard{ws_async, stream};
            void* ws = scratch(1, count_off + nh * 64, ws_async);
            be_zero_async((char*)ws + count_off, nh * 64, stream);
            if (!sw.handover_dump.empty())
                if (FILE* f = std::fopen(sw.handover_dump.c_str(), "a")) {
                    std::fprintf(f, "%llu %zu %zu %llu %zu\n", (unsigned long long)(size_t)ws, list_off, count_off, (unsigned long long)units, carry_n);
                    std::fclose(f);
                }
            for (size_t j = 0; j <= nh; j++) {  // launch j runs the steps [handovers[j-1], handovers[j])
                SolveArgs aj = a;
                int lanes = nth;
                size_t lds_j = lds;
                if (j == 0 && halves_ok && !manual) halves(aj, lanes, lds_j);
                aj.carry = (double*)ws;
                if (when_draining) {  // (the last launch counts its wavefronts too; it has no list to leave for)
                    aj.resident = p->d_resident;
                    {  // (per XCD -- eight on MI355X --, the threshold itself scaled to the chip's CUs: BIOIK_SOLVE_DRAIN_BELOW names it for 256 of them)
                        const int xcds = p->model->dev.xcds > 0 ? p->model->dev.xcds : 1;
                        const long long below = (long long)(throughput ? sw.drain_below_throughput : sw.drain_below) * (long long)kCus / 256;
                        aj.drain_below = sw.drain_test > 0 ? -sw.drain_test : (int32_t)((below + xcds - 1) / xcds);
                    }
                    aj.drain_min_steps = sw.drain_min_steps;
                }
                if (j > 0) {
                    aj.step_begin = when_draining ? -1 : handovers[j - 1];
                    aj.unit_list = (const int32_t*)((char*)ws + list_off + (j - 1) * list_bytes);
                    aj.unit_count = (const unsigned int*)((char*)ws + count_off + (j - 1) * 64);
                }
                if (j < nh) {
                    aj.step_end = handovers[j];
                    aj.carry_list = (int32_t*)((char*)ws + list_off + j * list_bytes);
                    aj.carry_count = (unsigned int*)((char*)ws + count_off + j * 64);
                }
                launch(aj, lanes, lds_j);  // (always a grid of `units` workgroups: those beyond the list's length leave at once)
            }
        }
    }
};

static void launch_solve(bioik_problem* p, const DevSolveParams& sp_in, size_t n, const double* d_seeds, const double* d_params, double* d_solutions,
                         double* d_fitness, int32_t* d_success, int32_t* d_steps, stream_t stream, const SolveSwitches& sw, unsigned int* error_word) {
    if (n == 0) return;
    SolveLauncher s(p, sp_in, n, d_seeds, d_params, d_solutions, d_fitness, d_success, d_steps, stream, sw, error_word);
    if (s.sp.solver != 0) return s.solve_point();
    s.choose_mapping();
    if (sw.report) s.report_mapping();
    if (s.lds > 64 * 1024) be_allow_lds(s.lds);
    SolveArgs a;
    a.pb = p->pb();
    a.sp = s.sp;
    a.seeds = d_seeds;
    a.params = d_params;
    a.phase_cycles = nullptr;
    a.sort_key_drop = sw.sort_key_drop;
    a.preselect = sw.preselect | (sw.tie_test_bits << 8);
    a.error = error_word;
    a.debug_flags = sw.debug_flags;
    set_deadline(p, s.sp, stream, a);
#if defined(BIOIK_PHASE_TIMING)
    DevBuf phase_buf(s.units * PHASE_SLOTS * sizeof(unsigned long long));
    const char* phase_path = sw.phase_dump.empty() ? nullptr : sw.phase_dump.c_str();
    if (phase_path) a.phase_cycles = phase_buf.as<unsigned long long>();
#endif
    s.plan_handovers();
#if defined(BIOIK_PHASE_TIMING)
    if (phase_path) s.handovers.clear();
#endif
    // (a solve in ONE launch reduces its islands itself, SolveArgs::island_done; BIOIK_SOLVE_FUSED_SELECT=0: by k_select as before)
    s.result_arrays(a, s.handovers.empty() && sw.fused_select > 0);
    s.run(a);
#if defined(BIOIK_PHASE_TIMING)
    if (phase_path) {  // profiling build only: synchronous dump of the per-phase cycle counters
        std::vector<unsigned long long> h(s.units * PHASE_SLOTS);
        be_d2h(h.data(), phase_buf.p, h.size() * sizeof(unsigned long long), stream);
        be_sync(stream);
        if (FILE* f = std::fopen(phase_path, "wb")) {
            std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
            std::fclose(f);
        }
    }
#endif
    s.select_islands(a);
}

// ------------------------------------------------------------------------------------------------------------
// The launcher's rules (launch_solve) were fitted to the three robots of BASELINE.json.  For a handle's FIRST chip-filling call of a kind under the latency
// schedule the entry points that may wait for their stream (the host-pointer ones; the device-pointer one under BIOIK_SOLVE_AUTOTUNE=2) run the call once
// per ELIGIBLE lane mapping -- every mapping returns the same bits, so each run is the caller's solve -- time it with events, keep the fastest in the handle
// and use it from then on (BIOIK_SOLVE_REPORT prints the table).  The presets are what the BIOIK_SOLVE_* switches can force; a preset the problem has no
// room or no kernel for is skipped.  Not for captured streams, solves with a timeout (their time is the caller's), the gradient family, or the throughput
// schedule (a lone call says nothing about a stream of ten: its dense mapping is the measured choice of profiles/r03_inflight_and_schedule.log).
// ------------------------------------------------------------------------------------------------------------
static const int kPresets = 5;
static const char* preset_name(int i) {
    static const char* names[kPresets] = {"the rules", "128 lanes, computed children in pairs (128-register build where it exists)", "64 lanes, species on the halves, computed children in pairs",
                                          "128 lanes, children kept in columns", "128 lanes, computed children one at a time"};
    return names[i];
}
static SolveSwitches preset_switches(const SolveSwitches& base, int i) {
    SolveSwitches w = base;
    if (i == 1) w.threads = 128, w.columnless = 2, w.four_waves = true;
    if (i == 2) w.threads = 64, w.species_parallel = 1, w.columnless = 2;
    if (i == 3) w.threads = 128;
    if (i == 4) w.threads = 128, w.columnless = 1;
    return w;
}
static void solve_dispatch(bioik_problem* p, const DevSolveParams& sp, size_t n, const double* d_seeds, const double* d_params, double* d_solutions, double* d_fitness,
                           int32_t* d_success, int32_t* d_steps, stream_t stream, bool may_wait, unsigned int* error_word) {
    const SolveSwitches sw = switches();  // (the diagnostic switches as last parsed: no environment access on the launch path)
    auto run = [&](const SolveSwitches& w) { launch_solve(p, sp, n, d_seeds, d_params, d_solutions, d_fitness, d_success, d_steps, stream, w, error_word); };
    const uint64_t units = (uint64_t)n * (uint64_t)sp.islands;
    const bool kind_ok = sw.autotune > 0 && !sw.manual() && !sw.general_set && !sw.two_phase_set && sw.drain_test == 0 && sw.dense_handover == 0 && sp.solver == 0 &&
                         sp.schedule != BIOIK_SCHEDULE_THROUGHPUT && sp.timeout_ticks == 0 && units >= 8 * (uint64_t)p->model->dev.cus && sp.max_steps >= 8;
    if (!kind_ok || be_stream_capturing(stream)) {
        run(sw);
        return;
    }
    int bucket = 0;  // floor(log2(units)): a mapping measured on 2048 units says little about 65536 (whether the hand-over to the helped kernel pays, ...)
    for (uint64_t u = units; u > 1; u >>= 1) bucket++;
    const auto key = std::make_tuple((int)sp.lambda, (int)sp.fk_mode, sp.islands > 1 ? 1 : 0, bucket);
    auto it = p->tuned.find(key);
    if (it != p->tuned.end() && it->second.preset >= 0) {
        run(preset_switches(sw, it->second.preset));
        return;
    }
    // (not while other solves of the handle are in flight on its other slots: their work would be in the timings, and a submit would block for five solves)
    bool others_pending = false;
    for (const auto& sl : p->io) others_pending = others_pending || sl.pending;
    if (!(may_wait || sw.autotune >= 2) || !be_can_time() || others_pending) {
        run(sw);
        return;
    }
    bioik_problem::Tuned t;
    int best = 0;
    for (int i = 0; i < kPresets; i++) {
        const SolveSwitches w = preset_switches(sw, i);
        try {
            t.ms[i] = be_time_ms(stream, [&]() { run(w); });
        } catch (const Error& e) {
            if (i == 0 || (e.code != BIOIK_ERR_UNSUPPORTED && e.code != BIOIK_ERR_INVALID_ARGUMENT)) throw;  // (a preset this problem has no room for: skipped)
            t.ms[i] = -1.0;
        }
        if (t.ms[i] >= 0.0 && t.ms[i] < t.ms[best]) best = i;
    }
    if (best != 0 && t.ms[best] > 0.97 * t.ms[0]) best = 0;  // (the rules stand unless something beats them by more than the run-to-run spread)
    t.preset = best;
    p->tuned[key] = t;
    if (sw.report) {
        std::fprintf(stderr, "[bioik] measured mapping choice (population %d, fk %d, islands %s; %llu units):\n", (int)sp.lambda, (int)sp.fk_mode, sp.islands > 1 ? "yes" : "no", (unsigned long long)units);
        for (int i = 0; i < kPresets; i++)
            if (t.ms[i] >= 0.0) std::fprintf(stderr, "[bioik]   %c %-90s %9.3f ms\n", i == best ? '*' : ' ', preset_name(i), t.ms[i]);
            else std::fprintf(stderr, "[bioik]     %-90s   not eligible\n", preset_name(i));
    }
    // (the caller's arrays hold the last eligible run's results: every mapping's are the same bits, nothing to redo)
}

// ------------------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

const char* bioik_last_error(void) { return g_err.c_str(); }
int bioik_abi_version(void) { return BIOIK_ABI_VERSION; }
int bioik_device_count(void) { return be_device_count(); }
int bioik_goal_param_count(int goal_type) { return bioik::goal_param_count(goal_type); }
int bioik_debug_reload_switches(void) {  // diagnostics only: the BIOIK_SOLVE_* switches are otherwise read once, when the library is loaded
    SolveSwitches w = parse_switches();
    std::lock_guard<std::mutex> lock(g_switch_mtx);
    g_switches = apply_switches(std::move(w));
    return BIOIK_OK;
}

void bioik_default_solve_params(bioik_solve_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(*p);
    p->mode = BIOIK_MODE_BIO2_MEMETIC;
    p->fk_mode = BIOIK_FK_EXACT;
    p->population = 128;
    p->islands = 1;
    p->max_steps = 64;
    p->random_seed = 0;
    p->dpos = -1.0;
    p->drot = -1.0;
    p->dtwist = 1e-5;
    p->no_wipeout = 0;
    p->timeout = 0.0;
    p->island_sync = 0;
}

int bioik_model_create(const bioik_model_desc* desc, int device, bioik_model** out) {
    API_BEGIN
    if (!desc || !out) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int n = be_device_count();
    if (n <= 0) throw Error(BIOIK_ERR_NO_DEVICE, "no HIP device available: this solver has no CPU path");
    if (device < 0 || device >= n) throw Error(BIOIK_ERR_NO_DEVICE, "HIP device index out of range");
    *out = new bioik_model(*desc, device);
    API_END
}
void bioik_model_destroy(bioik_model* m) { delete m; }

int bioik_problem_create(bioik_model* model, const bioik_problem_desc* desc, bioik_problem** out) {
    API_BEGIN
    if (!model || !desc || !out) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (desc->struct_size != sizeof(bioik_problem_desc)) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_problem_desc: struct_size mismatch");
    if ((desc->n_goals && !desc->goals) || (desc->n_group_joints && !desc->group_joints) || (desc->n_fixed_joints && !desc->fixed_joints))
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_problem_desc: a count is non-zero but its array is null");
    std::unique_ptr<bioik_problem> p(new bioik_problem(model, *desc));
    DeviceGuard on_device(model->device);
    p->d_pb = (DevProblem*)be_alloc(sizeof(DevProblem));
    p->d_clocks = (unsigned long long*)be_alloc(bioik_problem::kCaptureClocks * sizeof(unsigned long long));
    p->d_resident = (unsigned int*)be_alloc(16 * 128);
    be_zero_async(p->d_resident, 16 * 128, 0);
    p->h_error = (unsigned int*)be_alloc_pinned(64);
    std::memset(p->h_error, 0, 64);
    be_h2d(p->d_pb, &p->host.dev, sizeof(DevProblem), 0);
    be_sync(0);
    *out = p.release();
    API_END
}
void bioik_problem_destroy(bioik_problem* p) {
    if (!p) return;
    be_free(p->d_pb);
    be_free(p->d_clocks);
    be_free(p->d_resident);
    be_free_pinned(p->h_error);
    for (auto& kv : p->scratch) be_free(kv.second.base);
    for (const auto& r : p->retired_scratch) be_free(r.base);
    for (auto& sl : p->io) {
        if (sl.pending) {  // (a submitted solve nobody waited for: let it finish before its buffers go)
            try {
                be_sync(sl.stream);
            } catch (...) {
            }
        }
        be_stream_destroy(sl.stream);
        be_free(sl.dev);
        be_free_pinned(sl.host);
    }
    delete p;
}

int bioik_problem_active_variable_count(const bioik_problem* p) { return p ? p->host.dev.D : BIOIK_ERR_INVALID_ARGUMENT; }
int bioik_problem_active_variables(const bioik_problem* p, int32_t* out) {
    if (!p || !out) return BIOIK_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < p->host.active_variables.size(); i++) out[i] = p->host.active_variables[i];
    return BIOIK_OK;
}
int bioik_problem_tip_count(const bioik_problem* p) { return p ? p->host.dev.T : BIOIK_ERR_INVALID_ARGUMENT; }
int bioik_problem_tip_links(const bioik_problem* p, int32_t* out) {
    if (!p || !out) return BIOIK_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < p->host.tip_links.size(); i++) out[i] = p->host.tip_links[i];
    return BIOIK_OK;
}
int bioik_problem_param_count(const bioik_problem* p) { return p ? p->host.dev.P : BIOIK_ERR_INVALID_ARGUMENT; }
int bioik_problem_variable_count(const bioik_problem* p) { return p ? p->host.dev.V : BIOIK_ERR_INVALID_ARGUMENT; }
int bioik_problem_set_first_query(bioik_problem* p, uint64_t first_query) {
    if (!p) return BIOIK_ERR_INVALID_ARGUMENT;
    p->first_query = first_query;
    return BIOIK_OK;
}

int bioik_resolve_islands(const bioik_problem* p, const bioik_solve_params* params, size_t n, int32_t* islands, int32_t* island_sync) {
    API_BEGIN
    if (!p || !params || !islands || !island_sync) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    const DevSolveParams r = bioik::normalize_params(*params, 0, n, 8 * (size_t)p->model->dev.cus);
    *islands = r.islands, *island_sync = r.island_sync;
    API_END
}
int bioik_solve_batch_device(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* d_seeds, const double* d_goal_params,
                             double* d_solutions, double* d_fitness, int32_t* d_success, int32_t* d_steps, void* hip_stream) {
    API_BEGIN
    if (!p || !params) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    if (n && (!d_seeds || !d_solutions || !d_fitness || !d_success || !d_steps || (p->host.dev.P > 0 && !d_goal_params)))
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null array");
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    DevSolveParams sp = bioik::normalize_params(*params, p->first_query, n, 8 * (size_t)p->model->dev.cus);
    // (this entry does not wait for its solve: a rendezvous time-out of an EARLIER solve through it is reported here)
    if (p->h_error[bioik_problem::kIoSlots] != 0u) {
        p->h_error[bioik_problem::kIoSlots] = 0u;
        throw Error(BIOIK_ERR_HIP, "an earlier solve of this handle through bioik_solve_batch_device timed out at a rendezvous between its wavefronts (k_solve_lean_cl4h): its results are not valid");
    }
    solve_dispatch(p, sp, n, d_seeds, d_goal_params, d_solutions, d_fitness, d_success, d_steps, (stream_t)hip_stream, false, p->h_error + bioik_problem::kIoSlots);
    API_END
}

// host arrays in, host arrays out: staged through a slot's page-locked arena, one DMA each way, on the slot's own stream.
// io_finish: the slot's solve is complete and its results are in the caller's arrays (no-op for an idle slot).  Called with p->mtx held.
static void io_finish(bioik_problem* p, bioik_problem::IoSlot& sl) {
    if (!sl.pending) return;
    DeviceGuard on_device(p->model->device);
    try {
        be_sync(sl.stream);
    } catch (const Error& e) {
        // the device failed under THIS slot's solve: its ticket's wait reports it (not whichever call came by to reuse the slot), its result arrays
        // stay untouched, and the slot is free again
        sl.failed_ticket = sl.ticket, sl.failed_code = e.code, sl.failed_message = e.what();
        sl.pending = false;
        for (auto& kv : p->scratch) kv.second.ctl_base = nullptr;  // (a solve that broke off may have left its control words anywhere: set up again)
        return;
    }
    sl.pending = false;
    unsigned int& err = p->h_error[&sl - p->io];
    if (err != 0u) {  // (SolveArgs::error: a wavefront of this solve gave up waiting for its partner and went on unsynchronised)
        err = 0u;
        for (auto& kv : p->scratch) kv.second.ctl_base = nullptr;
        sl.failed_ticket = sl.ticket, sl.failed_code = BIOIK_ERR_HIP;
        sl.failed_message = "a rendezvous between the wavefronts of a workgroup timed out (k_solve_lean_cl4h): the results of this solve are not valid";
        return;
    }
    const size_t V = p->host.dev.V;
    const char* hd = (const char*)sl.host;
    std::memcpy(sl.solutions, hd + sl.o_sol, sl.n * V * 8);
    std::memcpy(sl.fitness, hd + sl.o_fit, sl.n * 8);
    std::memcpy(sl.success, hd + sl.o_suc, sl.n * 4);
    std::memcpy(sl.steps, hd + sl.o_steps, sl.n * 4);
}
// io_begin: copy in, enqueue the transfer in, the solve and the transfer out on the slot's stream; returns without waiting.
// A solve still pending on the slot is completed first (its results reach its caller's arrays).  Called with p->mtx held.
static void io_begin(bioik_problem* p, bioik_problem::IoSlot& sl, uint64_t ticket, const bioik_solve_params& params, uint64_t first_query, size_t n,
                     const double* seeds, const double* goal_params, double* solutions, double* fitness, int32_t* success, int32_t* steps) {
    io_finish(p, sl);
    DeviceGuard on_device(p->model->device);
    const size_t V = p->host.dev.V, P = p->host.dev.P;
    // arena layout: [seeds | goal_params] in, [solutions | fitness | success | steps] out, every block 64-byte aligned
    auto up = [](size_t b) { return (b + 63) / 64 * 64; };
    const size_t o_seeds = 0, o_par = o_seeds + up(n * V * 8), in_bytes = o_par + up(n * P * 8);
    const size_t o_sol = in_bytes, o_fit = o_sol + up(n * V * 8), o_suc = o_fit + up(n * 8), o_steps = o_suc + up(n * 4), total = o_steps + up(n * 4);
    if (sl.bytes < total) {
        be_free(sl.dev);
        be_free_pinned(sl.host);
        sl.dev = sl.host = nullptr, sl.bytes = 0;
        const size_t cap = total + total / 4;
        sl.dev = be_alloc(cap);
        sl.host = be_alloc_pinned(cap);
        sl.bytes = cap;
    }
    char* hd = (char*)sl.host;
    char* dd = (char*)sl.dev;
    if (!sl.stream_made) sl.stream = be_stream_create(), sl.stream_made = true;
    const stream_t st = sl.stream;
    std::memcpy(hd + o_seeds, seeds, n * V * 8);
    if (P) std::memcpy(hd + o_par, goal_params, n * P * 8);
    // (a call of a few queries -- MoveIt's one pose per call -- reads its inputs where they lie: the page-locked arena is mapped into the device's address space, and a
    // few hundred bytes per workgroup over the bus cost less than a DMA transfer in front of the launch)
    const bool direct_inputs = n <= 16;
    if (!direct_inputs) be_h2d(dd, hd, in_bytes, st);
    const char* in_base = direct_inputs ? hd : dd;
    DevSolveParams sp = bioik::normalize_params(params, first_query, n, 8 * (size_t)p->model->dev.cus);
    // The results go from the kernels straight into the page-locked arena (it is mapped into the device's address space; 1.5 MB per 4096 queries,
    // written once per query).  A transfer out enqueued behind the solve would sit at the head of a DMA queue until the solve is over -- 12 ms
    // for a one-launch solve -- with the transfers in of the handle's next solves behind it: nothing would overlap
    // (profiles/r03_inflight_and_schedule.log, host pipeline).
    solve_dispatch(p, sp, n, (const double*)(in_base + o_seeds), (const double*)(in_base + o_par), (double*)(hd + o_sol), (double*)(hd + o_fit), (int32_t*)(hd + o_suc),
                   (int32_t*)(hd + o_steps), st, true, p->h_error + (&sl - p->io));
    sl.pending = true, sl.ticket = ticket, sl.n = n;
    sl.o_sol = o_sol, sl.o_fit = o_fit, sl.o_suc = o_suc, sl.o_steps = o_steps;
    sl.solutions = solutions, sl.fitness = fitness, sl.success = success, sl.steps = steps;
}
static void solve_host(bioik_problem* p, const bioik_solve_params& params, uint64_t first_query, size_t n, const double* seeds, const double* goal_params,
                       double* solutions, double* fitness, int32_t* success, int32_t* steps) {
    if (n == 0) return;
    std::lock_guard<std::mutex> lock(p->mtx);
    const uint64_t ticket = p->next_ticket++;
    // (a synchronous call needs no slot of its own: an idle one whose stream and arenas exist already, so that a loop of searchPositionIK calls does not pay for
    // six streams and six pairs of arenas -- 10 ms each -- during its first six calls)
    size_t pick = ticket % bioik_problem::kIoSlots;
    for (size_t i = 0; i < (size_t)bioik_problem::kIoSlots; i++)
        if (!p->io[i].pending && p->io[i].stream_made) {
            pick = i;
            break;
        }
    bioik_problem::IoSlot& sl = p->io[pick];
    io_begin(p, sl, ticket, params, first_query, n, seeds, goal_params, solutions, fitness, success, steps);
    io_finish(p, sl);
    if (sl.failed_ticket == ticket) throw Error(sl.failed_code, "the solve failed on the device: " + sl.failed_message);  // (the synchronous call is its own ticket's wait)
}

int bioik_solve_batch(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* seeds, const double* goal_params, double* solutions,
                      double* fitness, int32_t* success, int32_t* steps) {
    API_BEGIN
    if (!p || !params) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    if (n && (!seeds || !solutions || !fitness || !success || !steps || (p->host.dev.P > 0 && !goal_params)))
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null array");
    solve_host(p, *params, p->first_query, n, seeds, goal_params, solutions, fitness, success, steps);
    API_END
}

// Asynchronous host-pointer solve: a caller with a stream of batches keeps up to kIoSlots (six) of them in flight on ONE handle.
int bioik_solve_batch_submit(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* seeds, const double* goal_params, double* solutions,
                             double* fitness, int32_t* success, int32_t* steps, uint64_t* ticket) {
    API_BEGIN
    if (!p || !params || !ticket) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    if (n && (!seeds || !solutions || !fitness || !success || !steps || (p->host.dev.P > 0 && !goal_params)))
        throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null array");
    std::lock_guard<std::mutex> lock(p->mtx);
    const uint64_t t = p->next_ticket++;
    *ticket = t;
    if (n == 0) return BIOIK_OK;  // (nothing to do: waiting for this ticket returns at once)
    bioik_solve_params sp = *params;
    if (sp.schedule == BIOIK_SCHEDULE_AUTO) {  // a pipeline three or more solves deep (this one and two before it) pays for the dense mapping (include/bioik_hip.h)
        int in_flight = 0;
        for (auto& sl : p->io) in_flight += (sl.pending && &sl != &p->io[t % bioik_problem::kIoSlots]) ? 1 : 0;
        sp.schedule = in_flight >= 2 ? BIOIK_SCHEDULE_THROUGHPUT : BIOIK_SCHEDULE_LATENCY;
    }
    io_begin(p, p->io[t % bioik_problem::kIoSlots], t, sp, p->first_query, n, seeds, goal_params, solutions, fitness, success, steps);
    API_END
}
int bioik_solve_batch_wait(bioik_problem* p, uint64_t ticket) {
    API_BEGIN
    if (!p) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    bioik_problem::IoSlot& sl = p->io[ticket % bioik_problem::kIoSlots];
    stream_t st = nullptr;
    {
        std::lock_guard<std::mutex> lock(p->mtx);
        if (ticket == 0 || ticket >= p->next_ticket) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_solve_batch_wait: unknown ticket");
        if (sl.failed_ticket == ticket) throw Error(sl.failed_code, "the solve of this ticket failed on the device: " + sl.failed_message);
        if (!sl.pending || sl.ticket != ticket) return BIOIK_OK;  // completed when its slot was taken again, or an empty batch
        st = sl.stream;
    }
    {  // the wait itself happens outside the handle's lock: another thread may submit the next batch meanwhile
        DeviceGuard on_device(p->model->device);
        try {
            be_sync(st);
        } catch (const Error&) {  // (io_finish below meets the same error under the lock and files it with the ticket)
        }
    }
    std::lock_guard<std::mutex> lock(p->mtx);
    if (sl.pending && sl.ticket == ticket) io_finish(p, sl);
    if (sl.failed_ticket == ticket) throw Error(sl.failed_code, "the solve of this ticket failed on the device: " + sl.failed_message);
    API_END
}

// One batch over several problem handles of the SAME template -- one per device of the node, or several on one device: shard r =
// queries [r n / W, (r + 1) n / W), one host thread per handle, each on its handle's stream; no exchange between the shards (the
// queries are independent), and the query-indexed random streams make the result identical to the unsharded solve.
int bioik_solve_batch_multi(bioik_problem* const* problems, int n_problems, const bioik_solve_params* params, size_t n, const double* seeds,
                            const double* goal_params, double* solutions, double* fitness, int32_t* success, int32_t* steps) {
    API_BEGIN
    if (!problems || n_problems <= 0 || !params) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    for (int r = 0; r < n_problems; r++) {
        if (!problems[r]) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null problem handle");
        const DevProblem &a = problems[0]->host.dev, &b = problems[r]->host.dev;
        if (a.V != b.V || a.P != b.P || a.D != b.D || a.T != b.T || a.n_ops != b.n_ops)
            throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_solve_batch_multi: the handles were compiled from different problem templates");
        for (int q = 0; q < r; q++)
            if (problems[q] == problems[r]) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_solve_batch_multi: the same handle twice");
    }
    const size_t V = problems[0]->host.dev.V, P = problems[0]->host.dev.P;
    if (n && (!seeds || !solutions || !fitness || !success || !steps || (P > 0 && !goal_params))) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null array");
    const uint64_t first = problems[0]->first_query;
    const size_t W = (size_t)n_problems;
    // BIOIK_ISLANDS_AUTO: one island count for the whole batch, sized to the largest shard (what one device gets), so that every shard runs the same solve
    bioik_solve_params shard_params = *params;
    if (params->islands <= 0 && n > 0) {
        const DevSolveParams r = bioik::normalize_params(*params, 0, (n + W - 1) / W, 8 * (size_t)problems[0]->model->dev.cus);
        shard_params.islands = r.islands, shard_params.island_sync = r.island_sync;
    }
    std::vector<int> status(W, BIOIK_OK);
    std::vector<std::string> message(W);
    std::vector<std::thread> workers;
    for (size_t r = 0; r < W; r++) {
        const size_t a = r * n / W, b = (r + 1) * n / W;
        if (a == b) continue;
        workers.emplace_back([&, r, a, b]() {
            try {
                solve_host(problems[r], shard_params, first + a, b - a, seeds + a * V, P ? goal_params + a * P : nullptr, solutions + a * V, fitness + a,
                           success + a, steps + a);
            } catch (const Error& e) {
                status[r] = e.code, message[r] = e.what();
            } catch (const std::exception& e) {
                status[r] = BIOIK_ERR_INVALID_ARGUMENT, message[r] = e.what();
            }
        });
    }
    for (auto& w : workers) w.join();
    for (size_t r = 0; r < W; r++)
        if (status[r] != BIOIK_OK) throw Error(status[r], "shard " + std::to_string(r) + ": " + message[r]);
    API_END
}

// ---- function-level entry points -------------------------------------------------------------------------
static EvalArgs eval_args(bioik_problem* p) {
    EvalArgs a;
    std::memset(&a, 0, sizeof(a));
    a.pb = p->pb();
    a.dpos = a.drot = a.dtwist = DBL_MAX;
    return a;
}

int bioik_eval_fk(bioik_problem* p, size_t n, const double* seed, const double* genes, double* tip_frames) {
    API_BEGIN
    if (!p || !seed || (n && (!genes || !tip_frames))) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return BIOIK_OK;
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    const size_t V = p->host.dev.V, D = p->host.dev.D, T = p->host.dev.T;
    DevBuf dseed(V * 8), dgenes(n * D * 8), dout(n * T * 7 * 8);
    be_h2d(dseed.p, seed, V * 8, 0);
    be_h2d(dgenes.p, genes, n * D * 8, 0);
    EvalArgs a = eval_args(p);
    a.n = n, a.seed = dseed.as<double>(), a.genes = dgenes.as<double>(), a.out0 = dout.as<double>();
    const int nth = 64;
    LAUNCH(k_eval_fk, eval_fk_body(a, b_, l_), (n + nth - 1) / nth, nth, lds_bytes(p, nth, 0), 0, a);
    be_d2h(tip_frames, dout.p, n * T * 7 * 8, 0);
    be_sync(0);
    API_END
}

int bioik_eval_fitness(bioik_problem* p, int fk_mode, size_t n, const double* seed, const double* goal_params, const double* base_genes, const double* genes,
                       double* primary, double* secondary) {
    API_BEGIN
    if (!p || !seed || (n && (!genes || !primary || !secondary))) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    if (fk_mode != BIOIK_FK_LINEAR && fk_mode != BIOIK_FK_EXACT) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "unknown fk_mode");
    if (fk_mode == BIOIK_FK_LINEAR && !base_genes) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "base_genes required for the linear model");
    if (p->host.dev.P > 0 && !goal_params) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "goal_params required");
    if (n == 0) return BIOIK_OK;
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    const size_t V = p->host.dev.V, D = p->host.dev.D, P = p->host.dev.P;
    DevBuf dseed(V * 8), dpar(P * 8), dbase(D * 8), dgenes(n * D * 8), d0(n * 8), d1(n * 8);
    be_h2d(dseed.p, seed, V * 8, 0);
    be_h2d(dpar.p, goal_params, P * 8, 0);
    if (base_genes) be_h2d(dbase.p, base_genes, D * 8, 0);
    be_h2d(dgenes.p, genes, n * D * 8, 0);
    EvalArgs a = eval_args(p);
    a.n = n, a.seed = dseed.as<double>(), a.params = dpar.as<double>(), a.genes = dgenes.as<double>(), a.base = dbase.as<double>();
    a.out0 = d0.as<double>(), a.out1 = d1.as<double>(), a.fk_mode = fk_mode;
    const int nth = 64;
    LAUNCH(k_eval_fitness, eval_fitness_body(a, b_, l_), (n + nth - 1) / nth, nth, lds_bytes(p, nth, 0), 0, a);
    be_d2h(primary, d0.p, n * 8, 0);
    be_d2h(secondary, d1.p, n * 8, 0);
    be_sync(0);
    API_END
}

int bioik_eval_approximator(bioik_problem* p, const double* seed, const double* base_genes, double* tip_frames, double* deltas) {
    API_BEGIN
    if (!p || !seed || !base_genes || !tip_frames || !deltas) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    const size_t V = p->host.dev.V, D = p->host.dev.D, T = p->host.dev.T;
    DevBuf dseed(V * 8), dbase(D * 8), d0(T * 7 * 8), d1(T * D * 7 * 8);
    be_h2d(dseed.p, seed, V * 8, 0);
    be_h2d(dbase.p, base_genes, D * 8, 0);
    EvalArgs a = eval_args(p);
    a.n = 1, a.seed = dseed.as<double>(), a.base = dbase.as<double>(), a.out0 = d0.as<double>(), a.out1 = d1.as<double>();
    const int nth = 64;
    LAUNCH(k_eval_approximator, eval_approximator_body(a, l_), 1, nth, lds_bytes(p, nth, 0), 0, a);
    be_d2h(tip_frames, d0.p, T * 7 * 8, 0);
    be_d2h(deltas, d1.p, T * D * 7 * 8, 0);
    be_sync(0);
    API_END
}

int bioik_eval_reproduce(bioik_problem* p, int population, uint32_t rng_key, int species, uint32_t generation, const double* parents, double* children_genes,
                         double* children_gradients) {
    API_BEGIN
    if (!p || !parents || !children_genes || !children_gradients || population <= 0) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bad argument");
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    const size_t D = p->host.dev.D, n = (size_t)population;
    DevBuf dpar(4 * D * 8), d0(n * D * 8), d1(n * D * 8);
    be_h2d(dpar.p, parents, 4 * D * 8, 0);
    EvalArgs a = eval_args(p);
    a.n = n, a.genes = dpar.as<double>(), a.out0 = d0.as<double>(), a.out1 = d1.as<double>();
    a.rng_key = rng_key;
    a.rng_ctr1 = (generation << 4) | ((uint32_t)species << 3) | 0u;
    const int nth = 64;
    const size_t M = p->host.dev.n_ops > 0 ? p->host.dev.n_ops : 1;
    LAUNCH(k_eval_reproduce, eval_reproduce_body(a, b_, l_), (n + nth - 1) / nth, nth, (4 * M + 2 * M * nth) * 8, 0, a);
    be_d2h(children_genes, d0.p, n * D * 8, 0);
    be_d2h(children_gradients, d1.p, n * D * 8, 0);
    be_sync(0);
    API_END
}

int bioik_eval_check(bioik_problem* p, const bioik_solve_params* params, size_t n, const double* seed, const double* goal_params, const double* genes,
                     int32_t* ok) {
    API_BEGIN
    if (!p || !params || !seed || (n && (!genes || !ok))) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null argument");
    if (p->host.dev.P > 0 && !goal_params) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "goal_params required");
    if (n == 0) return BIOIK_OK;
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    DevSolveParams sp = bioik::normalize_params(*params, 0);
    const size_t V = p->host.dev.V, D = p->host.dev.D, P = p->host.dev.P;
    DevBuf dseed(V * 8), dpar(P * 8), dgenes(n * D * 8), dok(n * 4);
    be_h2d(dseed.p, seed, V * 8, 0);
    be_h2d(dpar.p, goal_params, P * 8, 0);
    be_h2d(dgenes.p, genes, n * D * 8, 0);
    EvalArgs a = eval_args(p);
    a.n = n, a.seed = dseed.as<double>(), a.params = dpar.as<double>(), a.genes = dgenes.as<double>(), a.outi = dok.as<int32_t>();
    a.dpos = sp.dpos, a.drot = sp.drot, a.dtwist = sp.dtwist;
    const int nth = 64;
    LAUNCH(k_eval_check, eval_check_body(a, b_, l_), (n + nth - 1) / nth, nth, lds_bytes(p, nth, 0), 0, a);
    be_d2h(ok, dok.p, n * 4, 0);
    be_sync(0);
    API_END
}

int bioik_eval_arith(int device, int op, size_t n, const double* in, double* out) {
    API_BEGIN
    const int ni = arith_in(op), no = arith_out(op);
    if (ni == 0) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bioik_eval_arith: unknown op");
    if (n && (!in || !out)) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null array");
    if (n == 0) return BIOIK_OK;
    const int nd = be_device_count();
    if (nd <= 0) throw Error(BIOIK_ERR_NO_DEVICE, "no HIP device available: this library has no CPU path");
    if (device < 0 || device >= nd) throw Error(BIOIK_ERR_NO_DEVICE, "HIP device index out of range");
    DeviceGuard on_device(device);
    DevBuf din(n * ni * 8), dout(n * no * 8);
    be_h2d(din.p, in, n * ni * 8, 0);
    ArithArgs a;
    a.op = op, a.pad = 0, a.n = n, a.in = din.as<double>(), a.out = dout.as<double>();
    const uint64_t grid = (n + 255) / 256;
    if (grid > 0x7fffffffull) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "too many elements for one launch");
    LAUNCH(k_eval_arith, arith_body(a, b_ * 256 + (uint64_t)p_tid()), grid, 256, 0, 0, a);
    be_d2h(out, dout.p, n * no * 8, 0);
    be_sync(0);
    API_END
}

int bioik_stream_fitness_device(bioik_problem* p, size_t n_units, int population, const double* d_seeds, const double* d_goal_params, const double* d_genes,
                                double* d_fitness, void* hip_stream) {
    API_BEGIN
    if (!p || population <= 0) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "bad argument");
    if (n_units && (!d_seeds || !d_genes || !d_fitness || (p->host.dev.P > 0 && !d_goal_params))) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "null array");
    if (n_units == 0) return BIOIK_OK;
    std::lock_guard<std::mutex> lock(p->mtx);
    DeviceGuard on_device(p->model->device);
    const int nth = population >= 256 ? 256 : (population + 63) / 64 * 64;
    StreamArgs a;
    a.pb = p->pb();
    a.n_units = n_units;
    a.population = population;
    a.blocks_per_unit = (population + nth - 1) / nth;
    a.seeds = d_seeds, a.params = d_goal_params, a.genes = d_genes, a.fitness = d_fitness;
    uint64_t grid = (uint64_t)n_units * a.blocks_per_unit;
    if (grid > 0x7fffffffull) throw Error(BIOIK_ERR_INVALID_ARGUMENT, "too many units for one launch");
    LAUNCH(k_stream_fitness, stream_fitness_body(a, b_, l_), grid, nth, lds_bytes(p, nth, 0), (stream_t)hip_stream, a);
    API_END
}

}  // extern "C"
