// bioik_platform.h — the handful of execution-model primitives the kernels are written against.
//
// They are thin inline wrappers over the CDNA4 wave64 builtins; this is the only execution model the product is built for
// (hipcc, gfx950).  The kernel bodies are written against these names alone, which is what lets the test-suite compile the SAME
// bodies with a substitute set of primitives (a host simulator that lives entirely under tests/hostsim and is injected through
// BIOIK_PLATFORM_HEADER; the product build never defines it and contains no CPU execution path).
#pragma once
#include <stdint.h>

#include "bioik_types.h"

#if defined(BIOIK_PLATFORM_HEADER)
#include BIOIK_PLATFORM_HEADER
#else
// ------------------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define BIOIK_DEV __device__ __forceinline__
// A real function (one copy in the kernel, its own register allocation) for cold, arithmetic-heavy code.  Only for functions
// whose arguments and results are plain values: ROCm 7.2's gfx950 backend rejects LDS pointers that cross a call boundary.
#define BIOIK_CALL __device__ __attribute__((noinline))
// LDS pointers that do cross a call boundary are passed with their address space spelled out (a GENERIC pointer to LDS crossing
// a call trips a backend bug of this toolchain)
typedef __attribute__((address_space(3))) double lds_f64;
// uniform, read-only problem block: the constant address space makes every access a scalar (s_load) candidate
typedef const DevProblem __attribute__((address_space(4))) * ProbPtr;
typedef const DevProblemLean __attribute__((address_space(4))) * LeanProbPtr;

BIOIK_DEV int p_tid() { return (int)threadIdx.x; }
BIOIK_DEV int p_nthreads() { return (int)blockDim.x; }
BIOIK_DEV void p_barrier() { __syncthreads(); }
// LDS hand-over between lanes of ONE wavefront: a wavefront's LDS instructions execute in program order, so all that is
// needed is that the compiler keeps them in program order (wavefront-scope fences cost no instruction)
BIOIK_DEV void p_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class T>
BIOIK_DEV T p_shfl(T v, int src_lane) { return __shfl(v, src_lane, 64); }
template <class T>
BIOIK_DEV T p_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
// lane ^ 1 / lane ^ 2 inside a quad as a DPP move (quad_perm): no LDS crossbar round trip
template <int MASK>
BIOIK_DEV int p_quad_xor(int v) {
    static_assert(MASK == 1 || MASK == 2, "quad_perm covers xor 1 and xor 2");
    return __builtin_amdgcn_mov_dpp(v, MASK == 1 ? 0xB1 : 0x4E, 0xf, 0xf, true);  // [1,0,3,2] / [2,3,0,1]
}
template <int HALF>
BIOIK_DEV int p_row_mirror(int v) { return __builtin_amdgcn_mov_dpp(v, HALF ? 0x141 : 0x140, 0xf, 0xf, true); }  // row_half_mirror / row_mirror
template <int HALF>
BIOIK_DEV double p_row_mirror(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = p_row_mirror<HALF>((int)b), hi = p_row_mirror<HALF>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int MASK>
BIOIK_DEV double p_quad_xor(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = p_quad_xor<MASK>((int)b), hi = p_quad_xor<MASK>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int HALF>
BIOIK_DEV unsigned long long p_row_mirror(unsigned long long v) {
    const int lo = p_row_mirror<HALF>((int)v), hi = p_row_mirror<HALF>((int)(v >> 32));
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
template <int MASK>
BIOIK_DEV unsigned long long p_quad_xor(unsigned long long v) {
    const int lo = p_quad_xor<MASK>((int)v), hi = p_quad_xor<MASK>((int)(v >> 32));
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
BIOIK_DEV int p_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// the value lane `lane` of the wavefront holds (lane: wavefront-uniform), in every lane: v_readlane_b32, a scalar register as the carrier
BIOIK_DEV int p_read_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
BIOIK_DEV double p_read_lane(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// The same value as a NEW value the optimiser cannot see through (no instruction is emitted).  The solver's body is one long
// function whose loops all index LDS by a few lane numbers; the compiler hoists every address it derives from them in front of the
// outermost loop, where they are live across everything and end up in scratch memory.  A phase that starts from a fresh copy of
// the lane numbers computes its addresses where it uses them (one or two integer instructions each) and lets them die with it.
BIOIK_DEV int p_fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
BIOIK_DEV double p_fresh(double v) {
    asm volatile("" : "+v"(v));
    return v;
}
// The lane's number inside its wavefront, computed where it is asked for (v_mbcnt, two instructions) and never kept: for a workgroup that IS one
// wavefront this is p_tid() without a register that lives as long as the kernel.
BIOIK_DEV int p_lane_fresh() {
    int v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v));
    return v;
}
// p_tid() the same way: the wavefront's number inside the workgroup is uniform (a scalar register from the kernel's first instructions on), the lane's
// number comes from v_mbcnt where it is needed -- the kernel then has no use for the vector register it received the thread index in.
BIOIK_DEV int p_wave_index() { return p_uniform(p_tid() >> 6); }  // this wavefront's number inside its workgroup (a scalar register)
BIOIK_DEV int p_tid_fresh() { return p_lane_fresh() + (p_wave_index() << 6); }
// fmin(fmax(x, lo), hi) for wavefront-uniform bounds: the two instructions themselves.  Written with the library functions, the compiler first
// passes each bound through a v_max_f64 of its own (it cannot know that a number loaded from memory is no signalling NaN): four instructions per
// clipped gene instead of two.  Same instruction, same operands: the same bits for every x and every pair of bounds that are numbers.
BIOIK_DEV double p_clamp_uniform(double x, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "s"(hi));
    return r;
}
// the same two instructions for bounds that differ from lane to lane (vector-register operands): the winners' re-derivation, where lane k clamps op k
BIOIK_DEV double p_clamp(double x, double lo, double hi) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(r), "v"(hi));
    return r;
}
BIOIK_DEV unsigned long long p_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }  // lane mask of the wavefront (every lane calls it)
BIOIK_DEV int p_popc(uint32_t v) { return __popc(v); }
// the four bytes of v + addend: v_sad_u8 against zero, the (uniform) addend as a SCALAR operand -- through the builtin the compiler moves the constant into a vector
// register first, once per call (VOP3 takes no literal on gfx9): one more instruction per gene and child in the hottest loop
BIOIK_DEV int p_byte_sum(uint32_t v, int addend) {
    int r;
    asm("v_sad_u8 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(addend));
    return r;
}
BIOIK_DEV int p_popc64(unsigned long long v) { return __popcll(v); }
BIOIK_DEV unsigned long long p_wall_clock() { return wall_clock64(); }  // s_memrealtime: the chip-wide constant 100 MHz clock
BIOIK_DEV unsigned long long p_stamp_once(unsigned long long* word, unsigned long long value) {  // first caller's value wins; returns the winner
    const unsigned long long old = atomicCAS(word, 0ull, value);
    return old ? old : value;
}
BIOIK_DEV unsigned int p_atomic_inc(unsigned int* counter) { return atomicAdd(counter, 1u); }  // returns the value before
// what this lane has written is visible to every workgroup of the device / what they have written is visible to this lane (agent scope: the chip's eight L2s)
BIOIK_DEV void p_fence_device() { __threadfence(); }
BIOIK_DEV int p_xcc_id() { return (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u); }  // the XCD this wavefront runs on (hardware register XCC_ID)
// One 32-bit word from device memory straight into LDS (global_load_lds_dword: no register receives it, so nothing waits for it until p_prefetched_word
// does).  Called by ONE lane (the instruction writes lane l's word at the LDS address + 4 l).  The compiler does not know of the load: its own waits on the
// vector-memory counter only become stricter for it, and the reader waits for the counter itself.
BIOIK_DEV void p_prefetch_word_to_lds(const unsigned int* src, double* lds_slot) {
    const unsigned int lds_off = (unsigned int)(size_t)lds_slot;  // (the low half of a flat LDS address is the LDS offset)
    unsigned int m0_before;  // (M0 carries the LDS address of the instruction; whatever the compiler keeps there is put back)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(m0_before) : "v"(src), "s"(lds_off) : "memory");
}
BIOIK_DEV unsigned int p_prefetched_word(const double* lds_slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return *(const volatile unsigned int*)lds_slot;
}
BIOIK_DEV void p_atomic_add(unsigned int* counter, unsigned int v) { (void)atomicAdd(counter, v); }
BIOIK_DEV void p_atomic_sub(unsigned int* counter, unsigned int v) { (void)atomicSub(counter, v); }
BIOIK_DEV void p_atomic_min(unsigned int* word, unsigned int value) { (void)atomicMin(word, value); }
// (a word other workgroups of this device write: device scope -- the default scope of __atomic_load_n is the system's, a load that no cache may answer:
// 4096 workgroups reading one word once per step that way cost a stream of solves 13 % of its throughput, profiles/r04_drain_handover.log)
BIOIK_DEV unsigned int p_atomic_load(const unsigned int* word) { return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The state a unit hands from one launch to the next (SolveArgs::carry, carry_list): written and read with DEVICE-scope accesses (global_store / global_load
// ... sc1).  The chip has eight XCDs with an L2 each; a plain load may be answered by a line the reader's L2 kept from an EARLIER launch, and what makes
// that impossible between two eager launches -- the acquire at the start of every dispatch -- is not there between the kernel nodes of a replayed hipGraph
// (tools/micro/graph_handover_repro.hip: wrong from the second replay on with plain accesses, right with these).
#if defined(BIOIK_HANDOVER_PLAIN)  // (A/B builds only: the accesses as they were until round 4)
template <class T>
BIOIK_DEV T p_load_device(const T* p) { return *p; }
template <class T>
BIOIK_DEV void p_store_device(T* p, T v) { *p = v; }
#else
template <class T>
BIOIK_DEV T p_load_device(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
BIOIK_DEV void p_store_device(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
// Hand-overs between WAVEFRONTS of one workgroup through 32-bit words in LDS, without s_barrier (the helped kernel, solve_body<.., FIXED = 5>: its helper
// wavefronts never reach the workgroup's barriers).  A wavefront's LDS instructions execute in program order, so "data, then the word" on the writer's side and
// "the word, then data" on the reader's is all the ordering there is to keep; the fences keep the COMPILER from moving accesses across.
// p_flag_store: every lane stores the same value (one broadcast write).  p_flag_wait_ge: spins until *word >= value (s_sleep between the reads) and returns
// what it read; it gives up after ~2^22 reads and returns 0xffffffff -- the "leave" value of the protocol -- so that a wavefront whose partner died ends too.
BIOIK_DEV void p_flag_store(unsigned int* word, unsigned int value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *(volatile unsigned int*)word = value;
}
BIOIK_DEV unsigned int p_flag_wait_ge(const unsigned int* word, unsigned int value) {
    unsigned int v = 0xffffffffu;
    for (int spin = 0; spin < (1 << 22); spin++) {
        const unsigned int r = *(const volatile unsigned int*)word;
        if (r >= value) {
            v = r;
            break;
        }
        if (spin >= 128) __builtin_amdgcn_s_sleep(1);  // (the first polls back to back: a partner that is about to arrive is met within one LDS round trip)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return v;
}
#define P_INF (__builtin_inf())
#define BIOIK_FP_STRICT _Pragma("clang fp contract(off)")
#define BIOIK_HD __host__ __device__ inline
#endif

// Kernel flavour, carried by the TYPE of the problem pointer so that it reaches every device function by argument deduction:
// a LeanProbPtr promises "no floating / planar joint, no quaternion genes, the ops meet the genes in gene order" and the code for
// those cases is not instantiated (k_solve_lean: the robots of BASELINE.json); a plain ProbPtr keeps everything (k_solve).
template <class PB>
struct pb_flavour {
    static constexpr bool general = true;
};
template <>
struct pb_flavour<LeanProbPtr> {
    static constexpr bool general = false;
};

// Phase profiler (the reference's BLOCKPROFILER taxonomy, src/ik_evolution_2.cpp:330-437,605): compiled in only with
// -DBIOIK_PHASE_TIMING; lane 0 of the workgroup accumulates shader-clock cycles per phase.
#if defined(BIOIK_PHASE_TIMING)
#define PHASE_N 24      // phases (PH_*)
#define PHASE_SLOTS 28  // PHASE_N phases, then: start / end of the workgroup on the 100 MHz wall clock, HW_ID | XCC_ID << 32, spare
#define PHASE_DECL unsigned long long ph_t_[PHASE_N] = {0}, ph_last_ = __builtin_readcyclecounter(), ph_start_ = wall_clock64()
#define PHASE_MARK(i)                                          \
    do {                                                       \
        unsigned long long now_ = __builtin_readcyclecounter(); \
        ph_t_[i] += now_ - ph_last_;                           \
        ph_last_ = now_;                                       \
    } while (0)
#define PHASE_COUNT(i) (ph_t_[i] += 1)
#define PHASE_DUMP(ptr, unit)                                                       \
    do {                                                                            \
        if ((ptr) && p_tid() == 0)                                                  \
        {                                                                           \
            for (int i_ = 0; i_ < PHASE_N; i_++) (ptr)[(unit) * PHASE_SLOTS + i_] = ph_t_[i_]; \
            (ptr)[(unit) * PHASE_SLOTS + PHASE_N] = ph_start_;                      \
            (ptr)[(unit) * PHASE_SLOTS + PHASE_N + 1] = wall_clock64();             \
            (ptr)[(unit) * PHASE_SLOTS + PHASE_N + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
            (ptr)[(unit) * PHASE_SLOTS + PHASE_N + 3] = 0;                                   \
        }                                                                           \
    } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_DUMP(ptr, unit)
#define PHASE_COUNT(i)
#endif
enum { PH_INIT = 0, PH_REPRODUCE = 1, PH_FITNESS = 2, PH_SELECTION = 3, PH_MEMETICS = 4, PH_SPECIES = 5, PH_CHECK = 6, PH_PRESELECT = 7,
       // finer marks of the profiling build: selection = top-2 butterfly / cross-wave hop / winner copy / closing barrier;
       // memetics = linearisation / gradient / normalisation / line search / acceptance; counters of iterations
       PH_SEL_TOP2 = 8, PH_SEL_XWAVE = 9, PH_SEL_COPY = 10, PH_SEL_BAR = 11, PH_MEM_APPROX = 12, PH_MEM_GRAD = 13, PH_MEM_NORM = 14,
       PH_MEM_LINE = 15, PH_MEM_ACCEPT = 16, PH_MEM_TAIL = 17, PH_RANK = 18, PH_N_MEM_ITER = 19, PH_N_STEPS = 20, PH_LINEARISE = 21,
       PH_MEM_SUPPORT_COLS = 22, PH_MEM_SUPPORT_EVAL = 23, PH_MEM_CANDIDATE = 7 /* shares the slot of PH_PRESELECT */ };

// sin/cos of the joint half angles: the shared bit-reproducible implementation (bioik_sincos.h)
#define BIOIK_SINCOS_FN BIOIK_DEV
#include "bioik_sincos.h"
// fused forms of the frame algebra, shared with the oracle's device-arithmetic mode (bioik_fused.h)
#define BIOIK_FUSED_FN BIOIK_DEV
#include "bioik_fused.h"
BIOIK_DEV void p_sincos(double x, double* s, double* c) { bioik_sincos(x, s, c); }
// acos / atan2 of the goal costs and of the success test: the shared bit-reproducible implementations (bioik_acos.h)
#define BIOIK_ACOS_FN BIOIK_DEV
#include "bioik_acos.h"

#define BIOIK_DBL_MAX 1.7976931348623157e308
#define BIOIK_PI 3.14159265358979323846
