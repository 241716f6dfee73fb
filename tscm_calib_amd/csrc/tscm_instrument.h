// tscm_instrument.h -- ALL instrumentation of the kernels goes through the macros of this file (round 6: rounds 3-5 had left 59
// #if blocks interleaved through tscm_kernels.h, tscm_eval_gram4.h and tscm_solve_*.h).  A release build defines none of the three
// switches and every macro below expands to nothing: the release kernels read without #if ladders.
//
//   TSCM_WAVE_TIMELINE  (make variant EXTRA=-DTSCM_WAVE_TIMELINE)  per-wave / per-workgroup time stamps into device buffers that
//                       tscm_debug_wave_timeline, _kernel_timeline, _phase_stamps, _wave_views, _wave_phases copy out
//                       (tools/wave_timeline.py, kernel_timeline.py, phase_timeline.py): TL_ONLY(...), TL_STAMP, TL_ADD, KTL, KTLX
//   TSCM_PHASE_PROFILE  (make PHASES=1)                            phase stamps printed from the device (one line per launch from
//                       workgroups 0 and 200 of the fused kernels, the solver workgroups' phases): PH_ONLY(...), PHASE_STAMP
//   TSCM_BIG_PROFILE                                               the same for k_solve_reduced_big: BIG_ONLY(...)
//
// Included twice by tscm_kernels.h: first for the macros, then -- behind CtrlHead and wall_clock64 -- for the timeline buffers.
#ifndef TSCM_INSTRUMENT_MACROS
#define TSCM_INSTRUMENT_MACROS
#ifdef TSCM_WAVE_TIMELINE
#define TL_ONLY(...) __VA_ARGS__
#define TL_STAMP(var) const long long var = (long long)__builtin_readcyclecounter()
#define TL_ADD(k, a, b) tl_ph[k] += (b) - (a)
#define KTL(id) KtlScope ktl_scope(id, S.ctrl)
#define KTLX(i, on) do { if ((on) && threadIdx.x == 0) s_ktlx[i] = wall_clock64(); } while (0)
#define KTLX_FLUSH() do { if (threadIdx.x == 0) for (int q_ = 0; q_ < 32; ++q_) g_ktlx[q_] = s_ktlx[q_]; } while (0)
#else
#define TL_ONLY(...)
#define TL_STAMP(var)
#define TL_ADD(k, a, b)
#define KTL(id)
#define KTLX(i, on)
#define KTLX_FLUSH()
#endif
#ifdef TSCM_PHASE_PROFILE
#define PH_ONLY(...) __VA_ARGS__
#else
#define PH_ONLY(...)
#endif
#ifdef TSCM_BIG_PROFILE
#define BIG_ONLY(...) __VA_ARGS__
#else
#define BIG_ONLY(...)
#endif
// phase stamps of the fused kernels (s_memrealtime, 10 ns ticks)
#if defined(TSCM_PHASE_PROFILE) || defined(TSCM_WAVE_TIMELINE)
#define PHASE_STAMP(var) const long long var = wall_clock64()
#else
#define PHASE_STAMP(var)
#endif
#else   // ---- second inclusion (inside namespace tscm, behind CtrlHead and wall_clock64): the timeline buffers ----
#ifdef TSCM_WAVE_TIMELINE
constexpr int kTimelineWaves = 8192;
__device__ long long g_timeline[4 * kTimelineWaves];     // per wave of k_eval_gram: HW_ID, XCC_ID, start, end (10 ns ticks)
__device__ long long g_phase[5 * kTimelineWaves];        // per wave of k_eval_gram4: shader clocks per phase, summed over its views
constexpr int kTlViews = 12;
__device__ long long g_tlv[(4 + kTlViews) * kTimelineWaves];   // per wave of k_eval_gram4: wall-clock stamps of its head, tail and views (see there)
// per workgroup of the six kernels of an LM iteration (iteration 5): start, end of its thread 0 in 10 ns ticks
// (tscm_debug_kernel_timeline, tools/kernel_timeline.py: launch gaps, dispatch ramps and tails between the kernels)
constexpr int kKtlKernels = 6, kKtlGroups = 2048;
__device__ long long g_ktl[2 * kKtlKernels * kKtlGroups];
struct KtlScope {
    long long t0; int id; bool on; int blk;
    __device__ KtlScope(int id_, const CtrlHead *c) : t0(wall_clock64()), id(id_), on(c->iteration == 5), blk((int)blockIdx.x) {}
    __device__ ~KtlScope()
    {
        if (on && threadIdx.x == 0 && blk < kKtlGroups) {
            g_ktl[2 * (id * kKtlGroups + blk)] = t0;
            g_ktl[2 * (id * kKtlGroups + blk) + 1] = wall_clock64();
        }
    }
};
// per workgroup of k_schur_gram / k_backsub_prep (iteration 5): stamps of its phases (tscm_debug_phase_stamps, tools/phase_timeline.py)
constexpr int kPhStamps = 8;
__device__ long long g_phs[3 * kPhStamps * kKtlGroups];       // [0] k_schur_gram, [1] back-substitution, [2] k_schur_gram<NV, true>'s reduction blocks
__device__ long long g_ktlx[32];         // stamps inside the workgroup that runs the control step (thread 0): kept in LDS
__shared__ long long s_ktlx[32];         // and written out at the end (a global store in front of a barrier is waited for)
#endif
#endif
