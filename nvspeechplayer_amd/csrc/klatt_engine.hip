// klatt_engine.hip -- host side of the MI355X Klatt engine and its C-ABI.
//
// Exports (include/speechPlayer.h, include/speechPlayer_batch.h):
//   the reference's five entry points  (reference src/speechPlayer.cpp:25-53)
//     -- each handle is a GPU-resident stream: queueFrame keeps the producer-side queue on the
//        host (as reference src/frame.cpp:90-115 does), synthesize runs the kernel for that one
//        stream from its saved state and copies the PCM back;
//   the additive batch entry points    (N independent streams per launch): setUtterances classifies the utterances, packs them
//     into wavefronts by length and timing and plans the tracks (plan_tracks); a synthesis call launches up to five groups side by
//     side (batch_launch) -- for speech: klatt_tracks, then the stage-parallel kernel with flat stages.
// There is no CPU synthesis path in this library: without a HIP device every entry point fails.
#include "klatt_device.h"
#include "klatt_systolic.h"
#include "klatt_lanepipe.h"
#include "klatt_tracks.h"
#include "klatt_direct.h"
#include "klatt_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <atomic>
#include <system_error>
#include <thread>
#include <numeric>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/speechPlayer_batch.h"

using namespace klatt;

// int16 PCM -> float in [-1, 1]: sample / 32767, the scaling of the reference's audio sink
// (reference lavPlayer.py:17).  HBM-bound elementwise pass: 8 samples (16 B) in, 32 B out per lane.
__global__ void __launch_bounds__(256) pcm_to_float(const int16_t* __restrict__ in, float* __restrict__ out, long long n8)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float4 lo, hi;
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = (float)(int16_t)(w[k] & 0xFFFFu) / 32767.0f;
            f[2 * k + 1] = (float)(int16_t)(w[k] >> 16) / 32767.0f;
        }
        lo = make_float4(f[0], f[1], f[2], f[3]); hi = make_float4(f[4], f[5], f[6], f[7]);
        reinterpret_cast<float4*>(out)[2 * i] = lo;
        reinterpret_cast<float4*>(out)[2 * i + 1] = hi;
    }
}

// The PCM pool pads every utterance to a multiple of 32 samples (whole 64-byte row segments for the synthesis kernels' tile
// flushes); a consumer wants the utterances back to back.  pcm_compact writes that DENSE order in HBM, so that what crosses the
// link afterwards is ONE contiguous copy at the DMA engines' rate (until round 5 a host thread compacted 16-MB bounce buffers:
// 16 GB/s of a 63 GB/s link).  A thread owns 8 dense samples (one aligned 16-byte store): it finds its utterance by bisection
// over the dense starts, and -- when the 8 samples lie in one utterance, i.e. practically always -- reads the two aligned 16-byte
// vectors that cover them in the pool and shifts (the dense start of an utterance is any sample, so source and destination are
// misaligned by an even number of bytes that is the same for every thread of the utterance).  HBM-bound: 2 B read + 2 B written per sample.
__global__ void __launch_bounds__(256) pcm_compact(const int16_t* __restrict__ pool, int16_t* __restrict__ dense, const UttDesc* __restrict__ utt,
                                                   const long long* __restrict__ denseStart, long long nUtt, long long total)
{
    const long long n8 = (total + 7) / 8;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += stride) {
        const long long p = t * 8;
        long long lo = 0, hi = nUtt;                     // the last utterance whose dense start is <= p
        while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if (denseStart[mid] <= p) lo = mid; else hi = mid; }
        long long u = lo;
        const long long d0 = denseStart[u], d1 = denseStart[u + 1];
        uint32_t w[4];
        if (p + 8 <= d1) {
            const long long src = utt[u].outStart + (p - d0);            // in samples; the pool start of an utterance is a multiple of 32
            const int k = (int)(src & 7) * 2;                               // byte offset within the aligned 16-byte vector
            const uint4* a = reinterpret_cast<const uint4*>(pool + (src - (src & 7)));
            const uint4 v0 = a[0];
            uint32_t x[8] = {v0.x, v0.y, v0.z, v0.w, 0u, 0u, 0u, 0u};
            if (k) { const uint4 v1 = a[1]; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w; }
            const int dw = k >> 2, half = k & 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t a0 = 0, a1 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (dw == q) { a0 = x[j + q]; a1 = x[j + q + 1 < 8 ? j + q + 1 : 7]; }
                w[j] = half ? (a0 >> 16) | (a1 << 16) : a0;
            }
        } else {
            // the 8 samples straddle utterances (or run past the end): sample by sample
            uint32_t s16[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long q = p + i;
                uint32_t v = 0;
                if (q < total) {
                    while (q >= denseStart[u + 1]) ++u;   // (utterances of zero samples are stepped over)
                    v = (uint16_t)pool[utt[u].outStart + (q - denseStart[u])];
                }
                s16[i] = v;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = s16[2 * j] | (s16[2 * j + 1] << 16);
        }
        reinterpret_cast<uint4*>(dense)[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// SourceRef of every frame (klatt_device.h): what a flat or direct source stage reads when the frame is dequeued.  The division is
// the reference's (endVoicePitch - voicePitch) / minFrameDuration (src/frame.cpp:98), an IEEE double division here as there; a NULL
// frame carries no pitch of its own.
__global__ void __launch_bounds__(256) klatt_source_refs(const double* __restrict__ frames, const FrameMeta* __restrict__ meta, SourceRef* __restrict__ out, long long nFrames)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nFrames; k += stride) {
        const FrameMeta m = meta[k];
        const bool isNullFrame = (m.flags & FRAME_NULL) != 0;
        const double g0 = isNullFrame ? 0.0 : frames[k * kNumParams], g46 = isNullFrame ? 0.0 : frames[k * kNumParams + 46];
        SourceRef r;
        r.pitch = g0;
        r.pitchInc = isNullFrame ? 0.0 : __ddiv_rn(g46 - g0, (double)m.minSamples);
        r.invFade = __ddiv_rn(1.0, (double)m.fadeSamples);
        r.userIndex = m.userIndex;
        r.flags = m.flags & FRAME_NULL;
        out[k] = r;
    }
}

// 64-bit digest of every utterance's PCM, computed where the PCM lives: sum over the utterance's samples of
// mix64(position, value).  A sum, so lanes and wavefronts may add their shares in any order; one wavefront per utterance,
// 16 B (8 samples) per lane and step.  HBM-bound read of the pool.  (Checks of full-size configurations compare digests
// instead of copying tens of GB of PCM to the host; tests/test_gpu_parity.py restates the formula in numpy.)
__host__ __device__ inline unsigned long long digest_mix(unsigned long long pos, unsigned int value16)
{
    unsigned long long x = (pos + 1ull) * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)value16 + 1ull) * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}
__global__ void __launch_bounds__(256) pcm_digest(const int16_t* __restrict__ pcm, const UttDesc* __restrict__ utt,
                                                  const UttResult* __restrict__ result, unsigned long long* __restrict__ out, long long nUtt)
{
    const int lane = threadIdx.x & 63;
    const long long wavesPerGrid = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long u = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < nUtt; u += wavesPerGrid) {
        const long long base = utt[u].outStart;          // multiple of 32 samples: 16-B aligned
        const unsigned int n = result[u].produced;
        unsigned long long acc = 0;
        for (unsigned int i = (unsigned int)lane * 8u; i < n; i += 64u * 8u) {
            const uint4 v = *reinterpret_cast<const uint4*>(pcm + base + i);
            const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i + (unsigned int)k < n) acc += digest_mix(i + (unsigned int)k, (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)acc, m, 64);
            const unsigned int hi = (unsigned int)__shfl_xor((int)(unsigned int)(acc >> 32), m, 64);
            acc += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) out[u] = acc;
    }
}

namespace {

thread_local std::string g_lastError;
thread_local int g_lastErrorCode = 0;      // speechPlayer_lastErrorCode(): 0 after a call that succeeded

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_lastError = buf;
    if (g_lastErrorCode == 0) g_lastErrorCode = SPEECHPLAYER_ERR_ARGUMENT;
    fprintf(stderr, "[speechPlayer/hip] %s\n", buf);
}
void set_error_code(int code) { g_lastErrorCode = code; }
// every entry point that can fail starts with this: the code then describes THIS call
void begin_call() { g_lastErrorCode = 0; }

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_error_code(SPEECHPLAYER_ERR_HIP);                                           \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

template <typename T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
        cap = n;
        return 0;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
    }
};

// Is this host memory page-locked (from speechPlayer_hostAlloc, or registered by the caller with hipHostRegister)?  Copies to and
// from such memory are one DMA transfer at the link's rate and truly asynchronous; pageable memory goes through bounce buffers.
static bool is_pinned(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// one page-locked host block, grown on demand
struct PinnedBlock {
    void* ptr = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        release();
        HIP_TRY(hipHostMalloc(&ptr, bytes + bytes / 8, hipHostMallocDefault));
        cap = bytes + bytes / 8;
        return 0;
    }
    void release()
    {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr; cap = 0;
    }
};

// two pinned host buffers + events for pipelined device-to-host copies
struct PinnedPair {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        release();
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipHostMalloc(&buf[i], bytes, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        cap = bytes;
        return 0;
    }
    void release()
    {
        for (int i = 0; i < 2; ++i) {
            if (buf[i]) (void)hipHostFree(buf[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            buf[i] = nullptr; ev[i] = nullptr;
        }
        cap = 0;
    }
};

// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute is per device, and a process may
// drive several devices (one BatchPlayer each).
int ensure_lds_limit(const void* kernel, int ldsBytes)
{
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    for (const auto& e : done)
        if (e.first == kernel && e.second == dev) return 0;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
    done.emplace_back(kernel, dev);
    return 0;
}

KernelArgs base_args(int sampleRate)
{
    KernelArgs a;
    memset(&a, 0, sizeof a);
    a.sampleRate = sampleRate;
    a.sampleRateF = (double)sampleRate;
    a.invSampleRate = 1.0 / (double)sampleRate;
    a.negPiOverSr = -M_PI / sampleRate;          // reference src/speechWaveGenerator.cpp:116
    a.twoPiOverSr = (M_PI * 2) / sampleRate;     // reference src/speechWaveGenerator.cpp:31,118
    a.maxSamples = 0xFFFFFFFFu;
    return a;
}

template <bool NOISE, int CH, int WPS = 1, bool NASAL = true, bool STREAM = false, bool FLAT = false, bool LONE = false>
int launch_systolic(const KernelArgs& a, int mode, long long nGroups, hipStream_t stream)
{
    if (nGroups <= 0) return 0;
    if (nGroups > 0x7FFFFFFF) { set_error("too many workgroups: %lld", nGroups); return -1; }
    constexpr int ldsBytes = SysLds<NOISE, CH, FLAT, stream_split<STREAM, WPS, LONE>()>::kBytes + (LONE ? kStages * kLoneLdsPerStage : 0);     // LONE: a stage's fade chunks pass through LDS
    static_assert(ldsBytes <= 160 * 1024, "a workgroup's LDS");
    auto go = [&](auto kernel) -> int {
        if (ensure_lds_limit(reinterpret_cast<const void*>(kernel), ldsBytes)) return -1;
        hipLaunchKernelGGL(kernel, dim3((unsigned)nGroups), dim3(kLanes * kStages), ldsBytes, stream, a);
        return 0;
    };
    int rc;
    switch (mode) {
    case MODE_EXACT: rc = go(klatt_systolic<MODE_EXACT, NOISE, CH, WPS, NASAL, STREAM, FLAT, LONE>); break;
    case MODE_FAST: rc = go(klatt_systolic<MODE_FAST, NOISE, CH, WPS, NASAL, STREAM, FLAT, LONE>); break;
    default: set_error("unknown arithmetic mode %d", mode); return -1;
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

// the direct stages (klatt_direct.h): eight wavefronts per workgroup; `lean`: two workgroups per CU (8-sample hand-overs, 77 KB of pipes,
// 128 registers per stage), else one (16-sample hand-overs, 152 KB, 256 registers)
constexpr int kDirectChunk = 16, kDirectLeanChunk = 8;
template <int CH, int WPE>
int launch_direct_as(const KernelArgs& a, int mode, long long nGroups, hipStream_t stream)
{
    constexpr int ldsBytes = DirectLds<CH>::kBytes;
    auto go = [&](auto kernel) -> int {
        if (ensure_lds_limit(reinterpret_cast<const void*>(kernel), ldsBytes)) return -1;
        hipLaunchKernelGGL(kernel, dim3((unsigned)nGroups), dim3(kLanes * kDirectStages), ldsBytes, stream, a);
        return 0;
    };
    int rc;
    switch (mode) {
    case MODE_EXACT: rc = go(klatt_direct<MODE_EXACT, CH, WPE>); break;
    case MODE_FAST: rc = go(klatt_direct<MODE_FAST, CH, WPE>); break;
    default: set_error("unknown arithmetic mode %d", mode); return -1;
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_direct(const KernelArgs& a, int mode, bool lean, long long nGroups, hipStream_t stream)
{
    if (nGroups <= 0) return 0;
    if (nGroups > 0x7FFFFFFF) { set_error("too many workgroups: %lld", nGroups); return -1; }
    return lean ? launch_direct_as<kDirectLeanChunk, 4>(a, mode, nGroups, stream) : launch_direct_as<kDirectChunk, 2>(a, mode, nGroups, stream);
}

#ifndef KLATT_LP_CH
#define KLATT_LP_CH 32     // hand-over size of the lane-pipelined kernel with one workgroup per CU
#endif
template <int CH, int WPS>
int launch_lanepipe(const KernelArgs& a, int mode, long long nGroups, hipStream_t stream)
{
    if (nGroups <= 0) return 0;
    if (nGroups > 0x7FFFFFFF) { set_error("too many workgroups: %lld", nGroups); return -1; }
    constexpr int ldsBytes = LpLds<CH>::kBytes;
    auto go = [&](auto kernel) -> int {
        if (ensure_lds_limit(reinterpret_cast<const void*>(kernel), ldsBytes)) return -1;
        hipLaunchKernelGGL(kernel, dim3((unsigned)nGroups), dim3(kLanes * kStages), ldsBytes, stream, a);
        return 0;
    };
    int rc;
    switch (mode) {
    case MODE_EXACT: rc = go(klatt_lanepipe<MODE_EXACT, CH, WPS>); break;
    case MODE_FAST: rc = go(klatt_lanepipe<MODE_FAST, CH, WPS>); break;
    default: set_error("unknown arithmetic mode %d", mode); return -1;
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

// Which kernel runs the batch's groups (64 utterances per group)?  Measured on MI355X:
// stage-parallel workgroups (klatt_systolic.h) win at every batch size and for both groups -- 2.8e11 samples/s
// at 65 536 steady vowels against 1.3e11 for the lane kernel, 28.8 ms against 37.1 ms on cfg2 -- so "auto"
// always takes them; the lane kernel stays selectable (layout 0) and runs the streaming handles.
// Quiet batches that fit one workgroup per CU use 32-sample hand-overs (fewer barriers), else 16.
#ifndef KLATT_NOISY_CH
#define KLATT_NOISY_CH 8
#endif
struct GroupPlan { bool systolic; int chunk; };
GroupPlan plan_group(int layout, bool noisy, long long nUttBatch, long long nNoisyBatch, int cus)
{
    const long long groups = (nUttBatch + kLanes - 1) / kLanes + 1;
    GroupPlan p;
    p.systolic = layout != 0;
    (void)nNoisyBatch;
    p.chunk = noisy ? (groups <= cus ? 16 : 8)      // noisy: 8-sample hand-overs fit two workgroups per CU (LDS 80 KB each)
                    : (groups <= cus ? 32 : 16);
    return p;
}

template <bool STREAM, bool NOISE>
int launch(const KernelArgs& a, int mode, long long nWaves, hipStream_t stream)
{
    if (nWaves <= 0) return 0;
    if (nWaves > 0x7FFFFFFF) { set_error("too many wavefronts: %lld", nWaves); return -1; }
    constexpr int ldsBytes = LdsLayout<STREAM>::kBytes;
    auto go = [&](auto kernel) -> int {
        if (ensure_lds_limit(reinterpret_cast<const void*>(kernel), ldsBytes)) return -1;
        hipLaunchKernelGGL(kernel, dim3((unsigned)nWaves), dim3(kLanes), ldsBytes, stream, a);
        return 0;
    };
    int rc;
    switch (mode) {
    case MODE_EXACT: rc = go(klatt_synthesize<MODE_EXACT, STREAM, NOISE>); break;
    case MODE_FAST: rc = go(klatt_synthesize<MODE_FAST, STREAM, NOISE>); break;
    default:
        set_error("unknown arithmetic mode %d", mode);
        return -1;
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int pick_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error_code(SPEECHPLAYER_ERR_NO_DEVICE);
        set_error("no HIP device available (%s); this engine has no CPU path", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return -1;
    }
    if (device < 0) {
        const char* env = getenv("SPEECHPLAYER_DEVICE");
        if (env) device = atoi(env);
        else if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) { set_error("device %d out of range (%d devices)", device, n); return -1; }
    return device;
}

// ------------------------------------------------------------------------------------------
// Batch
// ------------------------------------------------------------------------------------------
struct Batch {
    int sampleRate = 0;
    int device = 0;
    int mode = MODE_EXACT;
    int sortByLength = 1;
    int layout = -1;                       // -1: choose per group (plan_group); 2: lane-pipelined where eligible; 1: stage-parallel workgroups; 0: one wave per 64 utterances
    int cus = 256;
    hipStream_t stream = nullptr;
    int tracks = 1;                        // 1: noisy utterances with finite parameters run on flat stages fed by tracks (klatt_tracks.h)
    long long trackBudgetMB = 4096;        // the tracks of a batch may take this much device memory (at most 4 GB: the flat stages address them with 32-bit byte offsets); utterances beyond it run untracked
    hipStream_t side[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // the other groups run beside the last one (batch_launch)
    hipEvent_t forkEvent = nullptr, join[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int quietLast = 1;                     // option "quiet_last": the quiet groups are launched after the noisy ones (batch_launch)
    long long nUtt = 0, nFrames = 0, nSlots = 0;   // nFrames: frames resident in HBM (the lists')
    long long nLists = 0, nFramesSpoken = 0;       // frame lists of the batch; frames the utterances queue (sum over utterances of their list's)
    std::vector<long long> uttFrameStart;          // [nUtt] first frame of the utterance's list
    std::vector<uint32_t> uttFrames;               // [nUtt] its number of frames
    bool launched = false;                         // a synthesis launch has been queued since the batch was set
    long long nQuiet = 0;                  // order[0..nQuiet) = utterances without noise, the rest with
    long long nTracked = 0;                // order[nQuiet..nQuiet + nTracked) = noisy utterances with tracks (slots: utterances + padding)
    long long nTrackedUtt = 0;             // the utterances among them
    long long nJobs = 0, trackEntries = 0; // distinct tracks of the batch, their entries (16 B each)
    int directLean = -1;                   // the direct stages' residency: 1 two workgroups per CU (the lean stages), 0 one, -1 the engine's choice (direct_lean())
    bool directAligned = false;            // setUtterances: half or more of the direct candidates sit in runs of 32 or more equally long, equally timed utterances
    int direct = 1;                        // noisy utterances with finite, bounded parameters and no tracks: 1 the direct stages (klatt_direct.h) unless their
                                           // lanes are time-aligned and the mode is MODE_EXACT (setUtterances), 2 the direct stages always, 0 the stages with the frame state machine
    long long nDirect = 0;                 // order[nQuiet + nTracked .. + nDirect) = such utterances (slots)
    long long nDirectUtt = 0, nDirectFrames = 0;
    long long nNoNasal = 0;                // order[0..nNoNasal) = quiet utterances that never couple the nasal pair (UTT_NO_NASAL); slots
    long long nNoNasalUtt = 0;             // the utterances among them (the rest: replicas that complete a sparse last wavefront)
    long long totalSamples = 0, poolSamples = 0;
    std::vector<uint32_t> lens;
    std::vector<long long> outStart;   // padded offsets in the device pool
    std::vector<UttResult> results;
    bool resultsFresh = false;
    DeviceBuffer<double> dFrames;
    DeviceBuffer<FrameMeta> dMeta;
    DeviceBuffer<UttDesc> dUtt;
    DeviceBuffer<uint32_t> dOrder;
    DeviceBuffer<int16_t> dPcm;
    DeviceBuffer<UttResult> dResult;
    DeviceBuffer<FlatRef> dFlatRef;            // [nFrames] the same for the flat filter stages: one 16-byte load per fade start
    DeviceBuffer<SourceRef> dSourceRef;        // [nFrames] what the flat source stage reads at a dequeue
    DeviceBuffer<TrackJob> dJobs;
    DeviceBuffer<double> dShapes;              // [nShapes][kShapeStride]
    DeviceBuffer<double2> dTrack;
    DeviceBuffer<DirectJob> dDirectJobs;       // [nDirectFrames] the end points of every direct frame's fade (host walk of reference src/frame.cpp:55-72)
    DeviceBuffer<uint32_t> dDirectFirst;       // [nUtt] record number of an utterance's first frame
    DeviceBuffer<DirectHdr> dDirectHdr;        // [8][nDirectFrames] written by klatt_seeds in every launch
    DeviceBuffer<double2> dDirectRec;          // [kDirectEntries][nDirectFrames]
    DeviceBuffer<unsigned long long> dDebug;   // KLATT_STAMPS builds
    DeviceBuffer<float> dFloat;                // float copy of the PCM pool (speechPlayer_batch_readFloat)
    DeviceBuffer<unsigned long long> dDigest;  // per-utterance digests (speechPlayer_batch_digest)
    PinnedPair bounce;                         // speechPlayer_batch_readAll into pageable memory
    DeviceBuffer<FrameRecord> dRecords;        // speechPlayer_batch_setRecords: the records as they crossed the link (klatt_expand_frames reads them)
    DeviceBuffer<double> dShapeTable;          // ... and the call's shape table
    DeviceBuffer<uint32_t> dRep;               // klatt_verify_shared: per frame the frame the planner took its shape values from
    DeviceBuffer<unsigned long long> dMismatch;
    DeviceBuffer<FrameFacts> dFacts;           // klatt_frame_facts' output (frames that arrived by DMA are classified and hashed on the device)
    PinnedBlock hFacts;                        // ... and where it lands on the host
    DeviceBuffer<int16_t> dDense;              // the utterances back to back (pcm_compact): what speechPlayer_batch_readAll copies out
    DeviceBuffer<long long> dDenseStart;       // [nUtt + 1] dense start of every utterance (prefix sum of the closed-form lengths)
    std::vector<long long> denseStart;         // the same on the host
    hipStream_t copyStream = nullptr;          // device-to-host copies of dDense run here, beside the next launch on `stream`
    hipEvent_t denseReady = nullptr, copyDone = nullptr;
    bool copyPending = false;                  // a speechPlayer_batch_readAllAsync copy is in flight (speechPlayer_batch_readWait)
    bool floatFresh = false;
};

// How many of the quiet, nasal-free utterances (the head of `order`) the lane-pipelined kernel takes.
// layout 2 forces it; "auto" takes it where it measured faster than the stage-parallel kernel.
long long lanepipe_count(const Batch* b)
{
    if (b->layout == 2) return b->nNoNasal;
    if (b->layout != -1 || b->nNoNasal == 0) return 0;
    // Its workgroups hold 16 utterances against 64, at ~30 ns per sample against 46 for the stage-parallel kernel's
    // slowest stage (MI355X, tools/len_probe.py): it wins while every workgroup of the launch has a CU of its own --
    // 0.75 against 1.2 ms at 4096 vowels -- and loses beyond (8192 vowels: two workgroups per CU).
    const long long groups = (b->nNoNasal + kLpUPG - 1) / kLpUPG + (b->nSlots - b->nNoNasal + kLanes - 1) / kLanes;
    return groups <= b->cus ? b->nNoNasal : 0;
}

// How many of the noisy utterances (the head of the noisy part of `order`) take their coefficients from tracks: those the
// host planned tracks for, under the stage-parallel layouts.
#ifndef KLATT_FLAT_CH
#define KLATT_FLAT_CH 16
#endif
#ifndef KLATT_FLAT_WPS
#define KLATT_FLAT_WPS 2
#endif
constexpr size_t kTrackPad = 64;   // slack past the last track
// ---- planning the tracks of a batch (host only; klatt_tracks.h, klatt_device.h for the track layout) --------------------
// Per frame of an eligible utterance: which entry kinds its fade moves and where the fade's track will be.  The state walked
// here is the part of the frame state machine that decides a fade's end points (reference src/frame.cpp:55-72, restated by
// stage_event): silence keeps the last request's values with the gain gated off, the first frame after silence starts from its
// own values with the gain gated off, any other frame fades from the previous request's values.  A SHAPE is the vector of
// the 39 parameter values a track depends on (klatt_device.h); fades with bitwise equal shapes at both ends and the same
// length share one track.
// Tracks pay when (nearly) the whole noisy group has them: a batch whose fades share nothing may need more memory than the
// budget, and splitting such a batch into a tracked and an untracked launch measured slower than either kernel alone
// (tools/track_probe.py +distinct).  So once more than a tenth of the eligible utterances did not fit, nothing is tracked.
// hash of a shape's 45 values: four independent multiply-xor chains (one chain of 45 dependent steps was half the planning's time)
inline unsigned long long hash_shape(const double* v)
{
    unsigned long long h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    int i = 0;
    for (; i + 4 <= kShapeValues; i += 4)
        for (int j = 0; j < 4; ++j) { unsigned long long w; memcpy(&w, &v[i + j], 8); h[j] = (h[j] ^ w) * 0xFF51AFD7ED558CCDull; h[j] ^= h[j] >> 29; }
    for (; i < kShapeValues; ++i) { unsigned long long w; memcpy(&w, &v[i], 8); h[0] = (h[0] ^ w) * 0xFF51AFD7ED558CCDull; h[0] ^= h[0] >> 29; }
    unsigned long long x = h[0] ^ (h[1] * 0x9E3779B97F4A7C15ull) ^ (h[2] << 21 | h[2] >> 43) ^ (h[3] * 0xC2B2AE3D27D4EB4Full);
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return x;
}

// where a frame's parameter values are on the host: in the caller's frames, or -- records -- in the row of the shape table a record names
// (the planner reads parameters 1..45 only: a record's own pitches are not among them)
struct FrameSource {
    const speechPlayer_frame_t* frames;
    const speechPlayer_frameRecord_t* records;
    const speechPlayer_frame_t* shapes;
    const double* values(long long k) const { return reinterpret_cast<const double*>(records ? shapes + records[k].shape : frames + k); }
};
unsigned long long g_planHashMask[2] = {~0ull, ~0ull};      // speechPlayer_setGlobalOption("plan_hash_bits")
struct TrackPlan {
    std::vector<TrackRef> ref;              // [nFrames]
    std::vector<uint32_t> rep;              // [nFrames] the frame whose shape values stand for this frame's (the first one seen with its hash); 0xFFFFFFFF: itself / none
    std::vector<TrackJob> jobs;             // one per distinct track
    std::vector<double> shapes;             // [nShapes][kShapeStride]
    std::vector<unsigned char> tracked;     // [nUtterances]
    std::vector<uint32_t> kinds;            // [nUtterances] the entry kinds whose values change after the first sample of the utterance's first fade (UttDesc.flags)
    unsigned long long entries = 0;
    long long eligible = 0, missedSize = 0, missedBudget = 0;   // utterances that may be tracked; that a fade of 2^27 entries or the budget kept out
};
// One pass over utterances [0, nUtterances) of the arrays given (frameStart may start anywhere: ref is indexed from frameStart[0]).
// `whole`: this is the whole batch -- apply the all-or-nothing rule; a part of it (plan_tracks below) leaves that to the merge,
// and gives up (missedBudget != 0) as soon as the parts' tracks together (`sum`) pass a quarter of the budget: that batch is
// planned in one pass (its fades are shared by few utterances, if at all: the parts would each make most of the tracks again, or
// fill the budget with 400 MB of shapes and map nodes only for the one pass to find that nothing fits -- 0.4 s for 65 536
// utterances with nothing in common, which is what such a batch costs without the parts).
// `facts`: klatt_plan.h's 128-bit hash of every frame's shape values (indexed like `frames`): a frame seen before is recognised by
// it, without gathering, hashing or comparing its 45 values again.
void plan_tracks_pass(long long nUtterances, const long long* frameStart, const FrameSource& frames, const FrameFacts* facts, const FrameMeta* meta,
                      const unsigned char* eligible, long long budgetMB, bool whole, std::atomic<unsigned long long>* sum, TrackPlan& out)
{
    const long long frame0 = frameStart[0], nF = frameStart[nUtterances] - frame0;
    out.ref.assign((size_t)nF, TrackRef{0, 0, 0});
    const bool wantRep = frameStart[nUtterances] < 0xFFFFFFFFll;
    out.rep.assign(wantRep ? (size_t)nF : 0, 0xFFFFFFFFu);
    out.jobs.clear();
    out.shapes.clear();
    out.tracked.assign((size_t)nUtterances, 0);
    out.kinds.assign((size_t)nUtterances, 0);
    out.entries = 0;
    // the entry kinds whose values differ between two shapes (the comparisons stage_event makes on the device)
    auto diff_kinds = [](const double* a, const double* c) -> uint32_t {
        uint32_t mask = 0;
        for (int r = 0; r < kNumRes; ++r)
            if (!(c[2 * r] == a[2 * r]) || !(c[2 * r + 1] == a[2 * r + 1])) mask |= 1u << r;
        for (int e = kNumRes; e < kTrackEntries; ++e) {
            const int x = entry_value(e, 0), y = entry_value(e, 1);
            if (!(c[x] == a[x]) || (y >= 0 && !(c[y] == a[y]))) mask |= 1u << e;
        }
        return mask;
    };
    struct Shape { double v[kShapeValues]; bool operator==(const Shape& o) const { return !memcmp(v, o.v, sizeof v); } };
    struct ShapeHash { size_t operator()(const Shape& k) const { return (size_t)hash_shape(k.v); } };
    struct Fade { uint32_t from, to, len; bool operator==(const Fade& o) const { return from == o.from && to == o.to && len == o.len; } };
    struct FadeHash { size_t operator()(const Fade& k) const { unsigned long long h = ((unsigned long long)k.from << 32 | k.to) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h += k.len; h *= 0xFF51AFD7ED558CCDull; return (size_t)(h ^ (h >> 32)); } };
    std::unordered_map<Shape, uint32_t, ShapeHash> shapes;          // values -> id (row of out.shapes)
    std::unordered_map<Fade, TrackRef, FadeHash> fades;             // (from, to, length) -> the fade's track
    auto shape_id = [&](const Shape& sh) -> uint32_t {
        auto it = shapes.find(sh);
        if (it != shapes.end()) return it->second;
        const uint32_t id = (uint32_t)shapes.size();
        shapes.emplace(sh, id);
        out.shapes.resize((size_t)(id + 1) * kShapeStride, 0.0);
        memcpy(&out.shapes[(size_t)id * kShapeStride], sh.v, sizeof sh.v);
        return id;
    };
    Shape zero; memset(&zero, 0, sizeof zero);
    shape_id(zero);
    // frames seen before, by their hash: the id of their shape and (once somebody needed it) of the same values with the gain gated off
    struct Key128 { unsigned long long a, b; bool operator==(const Key128& o) const { return a == o.a && b == o.b; } };
    struct Key128Hash { size_t operator()(const Key128& k) const { return (size_t)k.a; } };
    struct FrameIds { uint32_t plain, gated; long long first; };
    std::unordered_map<Key128, FrameIds, Key128Hash> seen;
    std::vector<uint32_t> gatedOf;          // shape id -> the id of its values with the gain gated off (0xFFFFFFFF: not asked for yet)
    auto gated_id = [&](uint32_t id) -> uint32_t {
        if (gatedOf.size() <= id) gatedOf.resize((size_t)id + 64, 0xFFFFFFFFu);
        if (gatedOf[id] == 0xFFFFFFFFu) {
            Shape g;
            memcpy(g.v, &out.shapes[(size_t)id * kShapeStride], sizeof g.v);
            g.v[kShapePreGain] = 0.0;
            const uint32_t gid = shape_id(g);
            if (gatedOf.size() <= id) gatedOf.resize((size_t)id + 64, 0xFFFFFFFFu);
            gatedOf[id] = gid;
        }
        return gatedOf[id];
    };
    // (the flat stages address the tracks with 32-bit byte offsets: below 4 GB, 2^28 entries)
    const unsigned long long budget = std::min((unsigned long long)std::max(budgetMB, 0ll) * (1ull << 20) / sizeof(double2), (1ull << 28) - (1ull << 21) - kTrackPad);
    std::vector<Fade> added;
    long long nEligible = 0, nMissed = 0;
    out.missedSize = 0; out.missedBudget = 0;
    for (long long u = 0; u < nUtterances; ++u) nEligible += eligible[u] ? 1 : 0;
    out.eligible = nEligible;
    // Give up early on a batch whose fades are (nearly) all different: its tracks grow in proportion to the utterances walked and
    // would pass the budget long before the end -- walking on until they do, hashing 45 values per frame and filling the maps,
    // was 0.6 s for 65 536 such utterances (30 launches' worth) for nothing.  Two look-outs, after 1/32 and 1/16 of the
    // utterances: tracks that doubled in between (no sharing yet: a batch that shares its fades saturates early) and that at this
    // rate end beyond the budget.
    const long long look1 = nUtterances / 32, look2 = nUtterances / 16;
    unsigned long long entriesAt1 = 0;
    for (long long u = 0; u < nUtterances && (!whole || nMissed * 10 <= nEligible); ++u) {
        if (whole && look1 >= 256) {
            if (u == look1) entriesAt1 = out.entries;
            if (u == look2 && out.entries - entriesAt1 >= entriesAt1 - entriesAt1 / 8 &&
                (long double)out.entries * ((long double)nUtterances / (long double)look2) > (long double)budget) {
                nMissed = nEligible + 1;      // the all-or-nothing rule below: nothing is tracked
                out.missedBudget = nEligible;
                break;
            }
        }
        if (!eligible[u]) continue;
        if (sum && sum->load(std::memory_order_relaxed) > budget / 4) { ++out.missedBudget; return; }
        added.clear();
        const unsigned long long before = out.entries;
        const size_t jobsBefore = out.jobs.size();
        bool fits = true, prevNull = true;
        uint32_t prevId = 0;       // the previous request's values (a fresh handle: all zero = shape 0)
        uint32_t kinds = 0;
        for (long long k = frameStart[u]; k < frameStart[u + 1] && fits; ++k) {
            uint32_t fromId = prevId, toId;
            const uint32_t lastTo = prevId;
            if (meta[k].flags & FRAME_NULL) {
                toId = gated_id(prevId);                       // silence: the old values, the gain gated off (:59-63)
                prevNull = true;
            } else {
                // (records: the key is the shape's number, exact -- never masked)
                const Key128 fk = frames.records ? Key128{facts[k].h0, facts[k].h1} : Key128{facts[k].h0 & g_planHashMask[0], facts[k].h1 & g_planHashMask[1]};
                auto it = seen.find(fk);
                if (it == seen.end()) {
                    Shape to;
                    const double* p = frames.values(k);
                    for (int i = 0; i < kShapeValues; ++i) to.v[i] = p[shape_param(i)];
                    it = seen.emplace(fk, FrameIds{shape_id(to), 0xFFFFFFFFu, k}).first;
                } else if (wantRep) {
                    out.rep[k - frame0] = (uint32_t)it->second.first;      // taken on trust here; compared where the frames are (klatt_verify_shared)
                }
                toId = it->second.plain;
                if (prevNull) {                                // out of silence: the new values, from gain 0 (:64-67)
                    if (it->second.gated == 0xFFFFFFFFu) it->second.gated = gated_id(toId);
                    fromId = it->second.gated;
                }
                prevNull = false;
            }
            prevId = toId;
            const Fade key{fromId, toId, meta[k].fadeSamples};
            auto f = fades.find(key);
            if (f == fades.end()) {
                // a new fade: what moves in it (the comparisons stage_event makes on the device), its size, its place
                const uint32_t mask = diff_kinds(&out.shapes[(size_t)fromId * kShapeStride], &out.shapes[(size_t)toId * kShapeStride]);
                const uint32_t nSlots = track_slots(mask);
                const unsigned long long n = (unsigned long long)kTrackFirst + (unsigned long long)(meta[k].fadeSamples - 1u) * nSlots;
                if (n >= (1ull << 27)) { fits = false; ++out.missedSize; break; }
                if (out.entries + n > budget) { fits = false; ++out.missedBudget; break; }
                if (sum && sum->fetch_add(n, std::memory_order_relaxed) + n > budget / 4) { ++out.missedBudget; return; }
                f = fades.emplace(key, TrackRef{out.entries, mask, nSlots}).first;
                added.push_back(key);
                out.jobs.push_back(TrackJob{out.entries, key.from, key.to, meta[k].fadeSamples, mask});
                out.entries += n;
            }
            out.ref[k - frame0] = f->second;
            // what changes after the utterance's first fade sample: what a fade moves, and what a later fade's first sample re-sets
            // to other values than the previous fade ended on (the frame after a silence starts from ITS values, gain gated off)
            kinds |= f->second.mask;
            if (k > frameStart[u] && fromId != lastTo)
                kinds |= diff_kinds(&out.shapes[(size_t)lastTo * kShapeStride], &out.shapes[(size_t)fromId * kShapeStride]);
        }
        out.kinds[u] = kinds;
        if (fits) out.tracked[u] = 1;
        else {
            for (const Fade& key : added) fades.erase(key);
            out.jobs.resize(jobsBefore);
            out.entries = before;
            ++nMissed;
        }
    }
    if (whole && nMissed * 10 > nEligible) {
        out.tracked.assign((size_t)nUtterances, 0);
        out.jobs.clear();
        out.entries = 0;
    }
}

// The host's worker threads.  Until round 5 every parallel section started and joined its own std::threads: eight sections of a
// setUtterances call are ~56 thread starts, 2-4 ms of a 22 ms call on the GPU box.  Now a pool that lives as long as the process:
// workers sleep on a condition variable, a section queues its parts and its caller works on queued parts (its own or anybody's) until
// its own are done -- so sections of several callers (the setter threads of a pipeline, the device threads of a node) share the workers
// without waiting for each other, and a section whose parts all queue behind others still finishes on its caller's thread.
// Parts never start sections of their own.  The pool is never destroyed (its threads sleep at process exit).
class WorkerPool {
public:
    struct Section { std::atomic<unsigned> left{0}; };
    static WorkerPool& get() { static WorkerPool* p = new WorkerPool(); return *p; }
    void submit(Section* sec, std::function<void()> fn)
    {
        { std::lock_guard<std::mutex> lg(mu_); queue_.emplace_back(sec, std::move(fn)); }
        cv_.notify_one();
    }
    // run queued parts until `sec` has none left
    void help_until_done(Section* sec)
    {
        std::unique_lock<std::mutex> lk(mu_);
        while (sec->left.load(std::memory_order_acquire) != 0) {
            if (!queue_.empty()) {
                auto job = std::move(queue_.front());
                queue_.pop_front();
                lk.unlock();
                job.second();
                job.first->left.fetch_sub(1, std::memory_order_acq_rel);
                done_.notify_all();
                lk.lock();
            } else {
                done_.wait_for(lk, std::chrono::microseconds(200));
            }
        }
    }
    unsigned workers() const { return nWorkers_; }
private:
    WorkerPool()
    {
        // as many workers as the process may really use: CPU affinity, capped by the cgroup quota (the GPU box shows 256 hardware threads
        // and grants 16 CPUs) and by 32
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0}; long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, atoll(q) / period));
            fclose(f);
        }
        n = std::min(n, 32u);
        for (unsigned i = 0; i + 1 < n; ++i) {
            try { std::thread([this] { loop(); }).detach(); ++nWorkers_; } catch (const std::system_error&) { break; }
        }
    }
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [this] { return !queue_.empty(); });
            auto job = std::move(queue_.front());
            queue_.pop_front();
            lk.unlock();
            job.second();
            job.first->left.fetch_sub(1, std::memory_order_acq_rel);
            done_.notify_all();
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<std::pair<Section*, std::function<void()>>> queue_;
    unsigned nWorkers_ = 0;
};

// fn(0) .. fn(n - 1) side by side: n - 1 of them offered to the pool, the last one on the caller's thread, which then helps until all are done.
template <class F>
void run_parts(unsigned n, F fn)
{
    if (n <= 1) { if (n) fn(0u); return; }
    WorkerPool& pool = WorkerPool::get();
    if (pool.workers() == 0) { for (unsigned t = 0; t < n; ++t) fn(t); return; }
    // A part that throws (std::bad_alloc in a planner's maps) must neither unwind this frame while queued parts still point at it nor
    // leave the section's count standing: every part runs under a guard that keeps the first exception, the caller helps until all
    // parts are done, and only then is the exception thrown on -- on the caller's thread.
    WorkerPool::Section sec;
    std::mutex emu;
    std::exception_ptr first;
    auto guarded = [&](unsigned t) {
        try { fn(t); } catch (...) { std::lock_guard<std::mutex> g(emu); if (!first) first = std::current_exception(); }
    };
    sec.left.store(n - 1, std::memory_order_release);
    for (unsigned t = 0; t + 1 < n; ++t) pool.submit(&sec, [&guarded, t] { guarded(t); });
    guarded(n - 1);
    pool.help_until_done(&sec);
    if (first) std::rethrow_exception(first);
}

// Host threads of setUtterances (the planner's setting: SPEECHPLAYER_PLAN_THREADS, else up to 8), and a loop over [0, n) cut into
// one contiguous range per thread (small loops stay on the caller's thread).
unsigned host_threads()
{
    unsigned n = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("SPEECHPLAYER_PLAN_THREADS")) n = (unsigned)std::max(1, atoi(e));
    return n;
}
template <class F>
void parallel_ranges(long long n, long long grain, F fn)
{
    const unsigned t = (unsigned)std::min<long long>(host_threads(), std::max<long long>(1, n / std::max<long long>(grain, 1)));
    if (t < 2) { fn(0ll, n); return; }
    run_parts(t, [&](unsigned k) { fn(n * (long long)k / t, n * (long long)(k + 1) / t); });
}

// The plan of a batch.  Large batches are planned in parts, one host thread each (the walk is a hash of 45 values and two map
// look-ups per frame: 0.2 s for BASELINE configs[2] on one thread), and the parts' shapes and fades merged: equal fades of
// different parts end up with one track.  If the merged tracks fit the budget that is the plan; if not -- or if the parts gave
// up (plan_tracks_pass) -- the batch is planned again in one pass, whose order decides which utterances stay in.
void plan_tracks(long long nUtterances, const long long* frameStart, const FrameSource& frames, const FrameFacts* facts, const FrameMeta* meta,
                 const unsigned char* eligible, long long budgetMB, TrackPlan& out)
{
    const long long nF = frameStart[nUtterances];
    static const bool trace = getenv("SPEECHPLAYER_PLAN_TRACE") != nullptr;      // where the planning's time goes, on stderr
    const auto tStart = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace) fprintf(stderr, "[speechPlayer/plan] %s: %.1f ms since the start\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count());
    };
    const unsigned nThreads = host_threads();
    if (nThreads < 2 || nF < 200000 || nUtterances < (long long)nThreads * 64) {
        plan_tracks_pass(nUtterances, frameStart, frames, facts, meta, eligible, budgetMB, true, nullptr, out);
        return;
    }
    // A look at the first 128th of the batch before the threads start (the look-outs of plan_tracks_pass, whole = true): a batch
    // whose fades are all different is given up after a few milliseconds instead of after the parts have filled a quarter of the budget.
    {
        TrackPlan probe;
        // (never past the batch: the threaded path is entered with as few as 2 x 64 utterances -- ADVICE r3)
        const long long nProbe = std::min<long long>(nUtterances, std::max<long long>(nUtterances / 128, 256));
        std::vector<unsigned char> none;
        plan_tracks_pass(nProbe, frameStart, frames, facts, meta, eligible, budgetMB, false, nullptr, probe);
        TrackPlan half;
        plan_tracks_pass(nProbe / 2, frameStart, frames, facts, meta, eligible, budgetMB, false, nullptr, half);
        const unsigned long long cap = std::min((unsigned long long)std::max(budgetMB, 0ll) * (1ull << 20) / sizeof(double2), (1ull << 28) - (1ull << 21) - kTrackPad);
        if (trace) fprintf(stderr, "[speechPlayer/plan] look: %llu entries after %lld utterances, %llu after %lld; budget %llu\n", half.entries, nProbe / 2, probe.entries, nProbe, cap);
        const long double scale = (long double)nUtterances / (long double)nProbe;
        // (tracks that double with the utterances and at that rate end more than a ninth beyond the budget: then more than a tenth of
        // the utterances would not fit, and the rule is all or nothing)
        if (half.entries > 0 && probe.missedBudget == 0 && probe.entries - half.entries >= half.entries - half.entries / 8 &&
            (long double)probe.entries * scale * 0.9L > (long double)cap) {
            out.ref.assign((size_t)nF, TrackRef{0, 0, 0});
            out.rep.clear();
            out.jobs.clear(); out.shapes.clear(); out.entries = 0;
            out.tracked.assign((size_t)nUtterances, 0);
            out.kinds.assign((size_t)nUtterances, 0);
            out.eligible = 0; out.missedSize = 0; out.missedBudget = 0;
            for (long long u = 0; u < nUtterances; ++u) out.eligible += eligible[u] ? 1 : 0;
            out.missedBudget = out.eligible;
            lap("given up after the look at the first utterances");
            return;
        }
    }
    lap("looked at the first utterances");
    // parts of about equal frame counts
    std::vector<long long> cut(nThreads + 1, nUtterances);
    cut[0] = 0;
    for (unsigned t = 1; t < nThreads; ++t)
        cut[t] = std::lower_bound(frameStart, frameStart + nUtterances, nF * (long long)t / nThreads) - frameStart;
    std::vector<TrackPlan> part(nThreads);
    std::atomic<unsigned long long> sum{0};
    run_parts(nThreads, [&](unsigned t) {
        plan_tracks_pass(cut[t + 1] - cut[t], frameStart + cut[t], frames, facts, meta, eligible + cut[t], budgetMB, false, &sum, part[t]);
    });
    lap("parts planned");
    // merge: shapes by value, fades by (from, to, length)
    struct ShapeKey { const double* v; bool operator==(const ShapeKey& o) const { return !memcmp(v, o.v, kShapeValues * sizeof(double)); } };
    struct ShapeKeyHash { size_t operator()(const ShapeKey& k) const { return (size_t)hash_shape(k.v); } };
    struct FadeKey { uint32_t from, to, len; bool operator==(const FadeKey& o) const { return from == o.from && to == o.to && len == o.len; } };
    struct FadeKeyHash { size_t operator()(const FadeKey& k) const { unsigned long long h = ((unsigned long long)k.from << 32 | k.to) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h += k.len; h *= 0xFF51AFD7ED558CCDull; return (size_t)(h ^ (h >> 32)); } };
    long long eligibleAll = 0, missedSize = 0, missedBudget = 0;
    size_t nShapesUpper = 0;
    for (const TrackPlan& p : part) { eligibleAll += p.eligible; missedSize += p.missedSize; missedBudget += p.missedBudget; nShapesUpper += p.shapes.size() / kShapeStride; }
    if (missedBudget > 0) {
        part.clear();
        plan_tracks_pass(nUtterances, frameStart, frames, facts, meta, eligible, budgetMB, true, nullptr, out);
        return;
    }
    out.shapes.clear();
    out.shapes.reserve(nShapesUpper * kShapeStride);     // no reallocation below: the keys point into it
    out.jobs.clear();
    out.entries = 0;
    std::unordered_map<ShapeKey, uint32_t, ShapeKeyHash> shapes;
    std::unordered_map<FadeKey, unsigned long long, FadeKeyHash> fades;
    std::vector<std::unordered_map<unsigned long long, unsigned long long>> offOf(nThreads);   // part, its track's first entry -> the merged track's
    for (unsigned t = 0; t < nThreads; ++t) {
        const TrackPlan& p = part[t];
        std::vector<uint32_t> gid(p.shapes.size() / kShapeStride);
        for (size_t i = 0; i < gid.size(); ++i) {
            const ShapeKey key{&p.shapes[i * kShapeStride]};
            auto it = shapes.find(key);
            if (it == shapes.end()) {
                const uint32_t id = (uint32_t)(out.shapes.size() / kShapeStride);
                out.shapes.insert(out.shapes.end(), key.v, key.v + kShapeStride);
                it = shapes.emplace(ShapeKey{&out.shapes[(size_t)id * kShapeStride]}, id).first;
            }
            gid[i] = it->second;
        }
        for (const TrackJob& j : p.jobs) {
            const FadeKey key{gid[j.fromShape], gid[j.toShape], j.fadeSamples};
            auto it = fades.find(key);
            if (it == fades.end()) {
                it = fades.emplace(key, out.entries).first;
                out.jobs.push_back(TrackJob{out.entries, key.from, key.to, j.fadeSamples, j.mask});
                out.entries += (unsigned long long)kTrackFirst + (unsigned long long)(j.fadeSamples - 1u) * track_slots(j.mask);
            }
            offOf[t].emplace(j.off, it->second);
        }
    }
    const unsigned long long budget = std::min((unsigned long long)std::max(budgetMB, 0ll) * (1ull << 20) / sizeof(double2), (1ull << 28) - (1ull << 21) - kTrackPad);
    if (out.entries > budget) {
        plan_tracks_pass(nUtterances, frameStart, frames, facts, meta, eligible, budgetMB, true, nullptr, out);
        return;
    }
    out.eligible = eligibleAll; out.missedSize = missedSize; out.missedBudget = 0;
    out.ref.assign((size_t)nF, TrackRef{0, 0, 0});
    out.rep.assign(nF < 0xFFFFFFFFll ? (size_t)nF : 0, 0xFFFFFFFFu);
    out.tracked.assign((size_t)nUtterances, 0);
    out.kinds.assign((size_t)nUtterances, 0);
    if (missedSize * 10 > eligibleAll) { out.jobs.clear(); out.entries = 0; return; }   // all or nothing
    run_parts(nThreads, [&](unsigned t) {
        const TrackPlan& p = part[t];
        const long long f0 = frameStart[cut[t]];
        for (long long u = cut[t]; u < cut[t + 1]; ++u) {
            if (!p.tracked[u - cut[t]]) continue;
            out.tracked[u] = 1;
            out.kinds[u] = p.kinds[u - cut[t]];
            for (long long k = frameStart[u]; k < frameStart[u + 1]; ++k) {
                TrackRef r = p.ref[k - f0];
                r.off = offOf[t].find(r.off)->second;
                out.ref[k] = r;
                if (!out.rep.empty() && !p.rep.empty()) out.rep[k] = p.rep[k - f0];
            }
        }
    });
    lap("merged");
}

// (Timing-only experiment builds -- wrong PCM, kept as a patch under tools/variants/ since round 5 -- define KLATT_TIMING_ONLY_BUILD: such a
// library refuses to hand PCM out, it only times.  ADVICE r3.)
#ifdef KLATT_TIMING_ONLY_BUILD
constexpr bool kTimingOnlyBuild = true;
#else
constexpr bool kTimingOnlyBuild = false;
#endif
bool refuse_timing_only(const char* what)
{
    if (!kTimingOnlyBuild || getenv("SPEECHPLAYER_ALLOW_TIMING_ONLY_PCM")) return false;
    set_error_code(SPEECHPLAYER_ERR_ARGUMENT);
    set_error("%s: this library was built with a timing-only experiment switch; its PCM is not valid", what);
    return true;
}

// The end points of the fades of one utterance's frames [k0, k1), appended to `out` (reference src/frame.cpp:55-72: a NULL request
// -- silence -- keeps the previous request's values with the gain gated off; the first frame after silence starts from its own
// values with the gain gated off; any other frame fades from the previous request's values).  klatt_seeds reads the values
// themselves on the device.  Host-only view for the tests: speechPlayer_planDirect.
template <class V>
void walk_fade_ends(long long k0, long long k1, const FrameMeta* meta, V& out)
{
    uint32_t prevReal = kNoFrame;
    bool prevNull = true;
    for (long long k = k0; k < k1; ++k) {
        DirectJob j;
        j.frame = (uint32_t)k;
        if (meta[k].flags & FRAME_NULL) {
            j.from = prevReal; j.to = prevReal;
            j.flags = (prevNull ? 1u : 0u) | 2u;
            prevNull = true;
        } else {
            j.to = (uint32_t)k;
            j.from = prevNull ? (uint32_t)k : prevReal;
            j.flags = prevNull ? 1u : 0u;
            prevReal = (uint32_t)k;
            prevNull = false;
        }
        out.push_back(j);
    }
}

// A vector whose resize() does not value-initialise (trivial element types only): the per-frame work arrays of setUtterances are
// written in full, in parallel, right after they are sized -- a zero-fill by the calling thread first costs ~1 ms per 6 MB.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using RawVector = std::vector<T, NoInitAlloc<T>>;

long long tracked_count(const Batch* b) { return (b->tracks && b->layout != 0) ? b->nTracked : 0; }
// (setUtterances only forms the direct group under the stage-parallel layouts)
long long direct_count(const Batch* b) { return b->nDirect; }
// The direct stages' residency for a launch of nGroups workgroups (option "direct_lean").  The engine's choice (tools/direct_size_probe.py,
// all-different batches of 8192 .. 65 536 utterances): MODE_FAST takes the lean stages -- two workgroups per CU -- at every size (12.4
// against 12.8 ms at 128 workgroups, 13.4 / 14.5 at 512, 18.1 / 27.3 at 1024); MODE_EXACT once the launch has more than TWO workgroups
// per CU (a launch of up to 512 workgroups lasts as long as its longest utterance takes through one workgroup's pipeline, which is
// shorter with the CU to itself: 26.4 against 27.7 ms at 512 workgroups; 31.7 / 30.0 at 576, 50.6 / 37.7 at 1024) -- and neither when
// the lanes are time-aligned: the lean stages have no steady path of their own (an aligned batch spends most chunks there: cfg2
// without its tracks 13.9 ms against 16.4, MODE_FAST; "distinct" 14.3 / 15.2).
bool direct_lean(const Batch* b, long long nGroups)
{
    if (b->directLean >= 0) return b->directLean != 0;
    if (b->directAligned) return false;
    return b->mode == MODE_FAST ? true : nGroups > 2ll * b->cus;
}

int batch_launch(Batch* b)
{
    KernelArgs a = base_args(b->sampleRate);
    a.frames = b->dFrames.ptr; a.meta = b->dMeta.ptr; a.utt = b->dUtt.ptr;
    a.pcm = b->dPcm.ptr; a.result = b->dResult.ptr; a.state = nullptr; a.control = nullptr;
    b->resultsFresh = false;
    b->floatFresh = false;
    b->launched = true;
#ifdef KLATT_STAMPS
    if (b->dDebug.reserve((size_t)(b->nSlots / kLpUPG + 2) * 32 * 2)) return -1;
    HIP_TRY(hipMemsetAsync(b->dDebug.ptr, 0, b->dDebug.cap * 8, b->stream));
    a.debug = b->dDebug.ptr;
#endif
    // up to four groups, each with its own kernel, side by side on their own streams:
    //   order[0, nLp)              quiet, nasal-free -> lane-pipelined kernel (klatt_lanepipe.h), when the plan takes it
    //   order[nLp, nNoNasal)       quiet, nasal-free -> stage-parallel kernel without the nasal pair (NASAL = false)
    //   order[nNoNasal, nQuiet)    quiet             -> stage-parallel (or lane) kernel, NOISE = false
    //   order[nQuiet, nSlots)      noisy             -> stage-parallel (or lane) kernel, NOISE = true
    //   order[nQuiet, nQuiet + nTr)  noisy, tracked  -> klatt_tracks, then the stage-parallel kernel with flat stages (FLAT = true)
    const long long nLp = lanepipe_count(b);
    const bool laneKernel = b->layout == 0;
    const long long nTr = tracked_count(b);
    const long long nDir = direct_count(b);
    // the noisy utterances that run on the stages with the frame state machine: the tail of `order`, and the tracked group's slots
    // when the tracks were planned but have been switched off since (options are read by setUtterances; "tracks" also here)
    const long long nNoisyHead = b->nTracked - nTr;
    const long long nNoisy = b->nSlots - b->nQuiet - b->nTracked - nDir;
    const long long nNn = laneKernel ? 0 : b->nNoNasal - nLp;
    const long long nQ = b->nQuiet - nLp - nNn;
    const int parts = (nLp > 0) + (nNn > 0) + (nQ > 0) + (nNoisy > 0) + (nNoisyHead > 0) + (nTr > 0) + (nDir > 0);
    const bool fork = parts > 1;
    if (fork) HIP_TRY(hipEventRecord(b->forkEvent, b->stream));
    int sideUsed = 0, seen = 0;
    // The last group launched runs on the main stream, the others beside it.  The QUIET groups are launched AFTER the noisy ones (option
    // "quiet_last", default 1): the noisy workgroups are few, long jobs (configs[2]: 896 workgroups of 8 273 .. 41 238 steps on 512
    // slots, whose best schedule ends with half the chip idle), the quiet ones cost a third per step -- dispatched behind the noisy
    // grid they take the slots it leaves free instead of delaying its longest jobs: configs[2] 8.44 -> 8.39 ms, MODE_FAST 7.88 -> 7.68,
    // configs[4]'s share 62.6 -> 62.2 (tools/quiet_priority_probe.py; low-PRIORITY streams for them were measured too and lost: 8.9 ms).
    auto next_stream = [&](bool = false) -> hipStream_t {
        if (!fork || ++seen == parts) return b->stream;
        hipStream_t st = b->side[sideUsed++];
        (void)hipStreamWaitEvent(st, b->forkEvent, 0);
        return st;
    };
    const GroupPlan plq = plan_group(b->layout, false, b->nSlots, nNoisy, b->cus);
    auto launch_quiet = [&]() -> int {
    if (nLp > 0) {
        hipStream_t st = next_stream(true);
        a.order = b->dOrder.ptr; a.nSlots = nLp;
        const long long g = (nLp + kLpUPG - 1) / kLpUPG;
        if (g <= b->cus ? launch_lanepipe<KLATT_LP_CH, 1>(a, b->mode, g, st) : launch_lanepipe<16, 2>(a, b->mode, g, st)) return -1;
    }
    if (nNn > 0) {
        hipStream_t st = next_stream(true);
        a.order = b->dOrder.ptr + nLp; a.nSlots = nNn;
        const long long g = (nNn + kLanes - 1) / kLanes;
        if (plq.chunk == 32 ? launch_systolic<false, 32, 1, false>(a, b->mode, g, st) : launch_systolic<false, 16, 2, false>(a, b->mode, g, st)) return -1;
    }
    if (nQ > 0) {
        hipStream_t st = next_stream(true);
        a.order = b->dOrder.ptr + nLp + nNn; a.nSlots = nQ;
        const long long g = (nQ + kLanes - 1) / kLanes;
        if (plq.systolic ? (plq.chunk == 32 ? launch_systolic<false, 32>(a, b->mode, g, st) : launch_systolic<false, 16>(a, b->mode, g, st))
                         : launch<false, false>(a, b->mode, g, st)) return -1;
    }
        return 0;
    };
    // (behind FLAT or direct stages only: the stages with the frame state machine fill the CU's LDS two workgroups at a time, and quiet
    // workgroups queued behind them wait for the launch's tail -- the benchmarked batch without its tracks 13.8 -> 14.7 ms)
    const bool quietLast = b->quietLast && (nTr > 0 || nDir > 0) && nNoisy == 0 && nNoisyHead == 0;
    if (!quietLast && launch_quiet()) return -1;
    if (nTr > 0) {
        hipStream_t st = next_stream();
        TrackArgs t;
        t.jobs = b->dJobs.ptr; t.nJobs = b->nJobs; t.shapes = b->dShapes.ptr; t.track = b->dTrack.ptr;
        t.negPiOverSr = a.negPiOverSr; t.twoPiOverSr = a.twoPiOverSr;
        const long long tg = b->nJobs;     // one workgroup per track
        if (tg > 0x7FFFFFFF) { set_error("too many tracks: %lld", b->nJobs); return -1; }
        hipLaunchKernelGGL(klatt_tracks, dim3((unsigned)tg), dim3(kLanes * kTrackWaves), 0, st, t);
        HIP_TRY(hipGetLastError());
        a.order = b->dOrder.ptr + b->nQuiet; a.nSlots = nTr;
        a.flatRef = b->dFlatRef.ptr; a.sourceRef = b->dSourceRef.ptr; a.track = b->dTrack.ptr;
        a.trackBytes = (uint32_t)std::min<unsigned long long>(((unsigned long long)b->trackEntries + kTrackPad) * sizeof(double2), 0xFFFFFFFFull);
        const GroupPlan pl = plan_group(b->layout, true, b->nSlots, nTr + nNoisy, b->cus);
        const long long g = (nTr + kLanes - 1) / kLanes;
        // flat stages keep nothing but the pipes and the PCM tile in LDS: 16-sample hand-overs fit two workgroups per CU (70 KB each)
        if (pl.chunk == 8 ? launch_systolic<true, KLATT_FLAT_CH, KLATT_FLAT_WPS, true, false, true>(a, b->mode, g, st)
                          : launch_systolic<true, 16, 1, true, false, true>(a, b->mode, g, st)) return -1;
        a.flatRef = nullptr; a.sourceRef = nullptr; a.track = nullptr;
    }
    if (nDir > 0) {
        hipStream_t st = next_stream();
        SeedArgs sa;
        sa.jobs = b->dDirectJobs.ptr; sa.nJobs = (uint32_t)b->nDirectFrames; sa.frames = b->dFrames.ptr; sa.meta = b->dMeta.ptr;
        sa.hdr = b->dDirectHdr.ptr; sa.rec = b->dDirectRec.ptr; sa.negPiOverSr = a.negPiOverSr; sa.twoPiOverSr = a.twoPiOverSr;
        const dim3 sg((unsigned)((b->nDirectFrames + 255) / 256), (unsigned)kDirectStages);
        constexpr int kSeedLds = 14 * 257 * (int)sizeof(double2);      // the largest record (14 entries) of 256 frames, pitch 257 (seed_stage)
        if (b->mode == MODE_FAST) hipLaunchKernelGGL(klatt_seeds<MODE_FAST>, sg, dim3(256), kSeedLds, st, sa);
        else hipLaunchKernelGGL(klatt_seeds<MODE_EXACT>, sg, dim3(256), kSeedLds, st, sa);
        HIP_TRY(hipGetLastError());
        a.order = b->dOrder.ptr + b->nQuiet + b->nTracked; a.nSlots = nDir;
        a.directHdr = b->dDirectHdr.ptr; a.directRec = b->dDirectRec.ptr; a.directFirst = b->dDirectFirst.ptr; a.nDirect = (uint32_t)b->nDirectFrames;
        a.sourceRef = b->dSourceRef.ptr;
        const long long g = (nDir + kLanes - 1) / kLanes;
        if (launch_direct(a, b->mode, direct_lean(b, g), g, st)) return -1;
        a.directHdr = nullptr; a.directRec = nullptr; a.directFirst = nullptr; a.nDirect = 0; a.sourceRef = nullptr;
    }
    for (int half = 0; half < 2; ++half) {
        const long long n = half ? nNoisy : nNoisyHead;
        if (n <= 0) continue;
        hipStream_t st = next_stream();
        a.order = b->dOrder.ptr + b->nQuiet + (half ? b->nTracked + nDir : 0); a.nSlots = n;
        const GroupPlan pl = plan_group(b->layout, true, b->nSlots, nTr + nNoisy + nNoisyHead, b->cus);
        const long long g = (n + kLanes - 1) / kLanes;
        if (pl.systolic ? (pl.chunk == 8 ? launch_systolic<true, KLATT_NOISY_CH, 2>(a, b->mode, g, st) : launch_systolic<true, 16>(a, b->mode, g, st))
                        : launch<false, true>(a, b->mode, g, st)) return -1;
    }
    if (quietLast && launch_quiet()) return -1;
    for (int i = 0; i < sideUsed; ++i) {
        HIP_TRY(hipEventRecord(b->join[i], b->side[i]));
        HIP_TRY(hipStreamWaitEvent(b->stream, b->join[i], 0));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Streams behind the reference's five entry points
// ------------------------------------------------------------------------------------------
// What a live handle owns lives in ONE arena per device, indexed by the handle's slot: its saved synthesiser state
// (kStateDoubles doubles) and a ring of kRing frames.  A queued frame is written once, into a pinned log
// (speechPlayer_queueFrame), the log travels in one copy and a scatter kernel drops its entries into the rings; a pull
// uploads one control block (what to do for each handle of the call) and whatever part of the log has not travelled yet.
// Nothing a handle queued is staged again on a later pull (reference contract: src/frame.cpp:90-115 queue, :54-75 dequeue).
constexpr uint32_t kRing = 256;                  // frames of a handle on the device; further ones wait on the host (Stream::overflow)
constexpr size_t kLogEntries = 40960;            // 16 MB of pinned log
constexpr size_t kLogEager = 2560;               // queueFrame sends the log on its way once 1 MB waits (if no pull is running)
constexpr uint32_t kNoTarget = 0xFFFFFFFFu;

struct LogEntry {            // 400 B
    double p[kNumParams];
    FrameMeta meta;
    uint32_t target;         // slot * kRing + ring position, or kNoTarget (purged / handle closed before the entry travelled)
    uint32_t pad;
};
static_assert(sizeof(LogEntry) == 400, "LogEntry layout");

struct PendingFrame {
    double p[kNumParams];
    FrameMeta meta;
};

__global__ void __launch_bounds__(256) live_scatter(const LogEntry* __restrict__ log, uint32_t n, double* __restrict__ ringFrames,
                                                    FrameMeta* __restrict__ ringMeta)
{
    const uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6), j = threadIdx.x & 63u;     // one wavefront per entry
    if (e >= n) return;
    const uint32_t at = log[e].target;
    if (at == kNoTarget) return;
    if (j < (uint32_t)kNumParams) ringFrames[(size_t)at * kNumParams + j] = log[e].p[j];
    else if (j == (uint32_t)kNumParams) ringMeta[at] = log[e].meta;
}

inline uint32_t frame_span(const FrameMeta& m)      // samples from this frame's dequeue to the next one's, the closed form of speechPlayer_batch_setUtterances
{
    const unsigned long long v = std::max<unsigned long long>(m.minSamples, (unsigned long long)m.fadeSamples + 1) + 1;
    return (uint32_t)std::min<unsigned long long>(v, 0xFFFFFFFFull);
}

struct Stream {
    int sampleRate = 0;
    int device = 0;
    int mode = MODE_EXACT;
    uint32_t seed = 0;
    uintptr_t id = 0;
    std::mutex mu;                       // per call, not per sample (reference locks per sample: src/frame.cpp:122)
    struct LiveContext* context = nullptr;   // the device's live-handle context
    uint32_t slot = 0;                   // its place in the device's arena
    uint32_t ringHead = 0;               // ring position of the oldest frame the kernel has not taken
    uint32_t ringCount = 0;              // frames in the ring from there on (travelled, or waiting in the log)
    uint32_t span[kRing];                // frame_span of each ring position; read only while frames wait in `overflow`
    std::vector<PendingFrame> overflow;  // queued beyond the ring: overflow[overHead ..], in order
    size_t overHead = 0;
    uint32_t logEpoch = 0;               // logIdx is valid for this upload epoch of the log only
    std::vector<uint32_t> logIdx;        // this handle's entries in the part of the log that has not travelled
    bool purgePending = false;
    int lastIndex = -1;
};

std::mutex g_tableMutex;
std::vector<Stream*> g_streams;   // handle = index + 1

Stream* lookup_locked(speechPlayer_handle_t h)
{
    uintptr_t id = reinterpret_cast<uintptr_t>(h);
    // a prototype-less 32-bit ctypes call hands the id back sign-/zero-extended; ids are small
    id &= 0xFFFFFFFFu;
    if (id == 0 || id > g_streams.size()) return nullptr;
    return g_streams[id - 1];
}
Stream* lookup(speechPlayer_handle_t h)
{
    std::lock_guard<std::mutex> g(g_tableMutex);
    return lookup_locked(h);
}

template <typename T>
struct PinnedBuffer {
    T* ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        n = std::max(n, cap * 2);          // contents are not kept
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr; cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T), hipHostMallocDefault));
        cap = n;
        return 0;
    }
};

// The live handles of one device: arena, log, launch buffers.  Lock order: Stream::mu, then LiveContext::mu, then logMu.
// `mu` serialises everything that enqueues on `stream` or moves the arena; `logMu` only guards the log's indices, so that
// queueFrame on one handle does not wait for a pull of other handles.
struct LiveContext {
    std::mutex mu;
    hipStream_t stream = nullptr;
    // arena
    uint32_t slots = 0, nextSlot = 0;
    std::vector<uint32_t> freeSlots;
    DeviceBuffer<double> dState;         // [slots][kStateDoubles]
    DeviceBuffer<double> dRingFrames;    // [slots][kRing][kNumParams]
    DeviceBuffer<FrameMeta> dRingMeta;   // [slots][kRing]
    // log
    std::mutex logMu;
    PinnedBuffer<LogEntry> hLog;
    DeviceBuffer<LogEntry> dLog;
    size_t logTail = 0, logSent = 0;     // entries [logSent, logTail) have not travelled
    uint32_t logEpoch = 1;
    // one pull
    PinnedBuffer<unsigned char> hCtl;    // UttDesc[n] | state pointer[n] | control[n], one copy
    DeviceBuffer<unsigned char> dCtl;
    PinnedBuffer<UttResult> hResult;
    DeviceBuffer<UttResult> dResult;
    DeviceBuffer<uint32_t> dOrder;       // 0, 1, 2, ...: written once
    size_t orderFilled = 0;
    DeviceBuffer<int16_t> dPcm, dPcmJoin;
    std::vector<uint32_t> done;
    PinnedPair bounce;
    hipEvent_t kernelStart = nullptr, kernelStop = nullptr;
    float lastKernelMs = 0.0f;           // the last call's kernel time (speechPlayer_lastLiveKernelMs)
    int lastLaunches = 0;
};
std::mutex g_liveMutex;
std::vector<LiveContext*> g_live;   // per device
// Which kernel advances live handles: 1 = the stage-parallel kernel's STREAM instantiation (four wavefronts per 64 handles;
// default), 0 = the lane kernel's (one wavefront per 64 handles).  Same saved state, same PCM; speechPlayer_setGlobalOption.
int g_liveLayout = [] { const char* e = getenv("SPEECHPLAYER_LIVE_LAYOUT"); return e ? atoi(e) : 1; }();
int g_liveCus = 0;
bool g_liveCusForced = false;
int g_liveReplicate = [] { const char* e = getenv("SPEECHPLAYER_LIVE_REPLICATE"); return e ? atoi(e) : 1; }();   // a lone handle fills its wavefront (streams_synthesize)
int g_liveMode = [] { const char* e = getenv("SPEECHPLAYER_LIVE_MODE"); return e && atoi(e) == 1 ? MODE_FAST : MODE_EXACT; }();   // arithmetic mode of handles created from now on
int g_liveAlone = [] { const char* e = getenv("SPEECHPLAYER_LIVE_ALONE"); return e ? atoi(e) : 1536; }();   // pulls of up to this many handles: a wavefront per handle (streams_synthesize)
int g_liveTrim = 0;                 // speechPlayer_setGlobalOption("live_trim"): release a device's arena when its last handle is terminated

// c->mu held.  No handle lives on this device: give its arena (state blocks, rings) and the pull buffers back.  The next
// speechPlayer_initialize starts a new arena of 64 slots.
void arena_trim(LiveContext* c)
{
    if (c->slots == 0 || c->freeSlots.size() != (size_t)c->nextSlot) return;
    (void)hipStreamSynchronize(c->stream);
    c->dState.release(); c->dRingFrames.release(); c->dRingMeta.release();
    c->dPcm.release(); c->dPcmJoin.release(); c->dCtl.release(); c->dResult.release(); c->dOrder.release();
    c->orderFilled = 0;
    c->slots = 0; c->nextSlot = 0;
    c->freeSlots.clear();
}

LiveContext* live_context(int device)
{
    std::lock_guard<std::mutex> g(g_liveMutex);
    if ((int)g_live.size() <= device) g_live.resize(device + 1, nullptr);
    if (!g_live[device]) {
        LiveContext* c = new LiveContext;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&c->kernelStart) != hipSuccess || hipEventCreate(&c->kernelStop) != hipSuccess ||
            c->hLog.reserve(kLogEntries) || c->dLog.reserve(kLogEntries)) {
            set_error("cannot create the live-handle context of device %d", device);
            delete c;
            return nullptr;
        }
        g_live[device] = c;
    }
    return g_live[device];
}

// c->mu held.  Room for `want` slots: the arena doubles, what the handles saved moves along (device to device).
int arena_reserve(LiveContext* c, uint32_t want)
{
    if (want <= c->slots) return 0;
    uint32_t cap = std::max<uint32_t>(64, c->slots);
    while (cap < want) cap *= 2;
    DeviceBuffer<double> st, fr;
    DeviceBuffer<FrameMeta> me;
    if (st.reserve((size_t)cap * kStateDoubles) || fr.reserve((size_t)cap * kRing * kNumParams) || me.reserve((size_t)cap * kRing)) {
        st.release(); fr.release(); me.release();
        return -1;
    }
    if (c->slots) {
        HIP_TRY(hipMemcpyAsync(st.ptr, c->dState.ptr, (size_t)c->slots * kStateDoubles * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(fr.ptr, c->dRingFrames.ptr, (size_t)c->slots * kRing * kNumParams * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(me.ptr, c->dRingMeta.ptr, (size_t)c->slots * kRing * sizeof(FrameMeta), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->dState.release(); c->dRingFrames.release(); c->dRingMeta.release();
    c->dState = st; c->dRingFrames = fr; c->dRingMeta = me;
    c->slots = cap;
    return 0;
}

int stream_init_device(Stream* s)
{
    LiveContext* c = live_context(s->device);
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(s->device));
    uint32_t slot;
    if (!c->freeSlots.empty()) { slot = c->freeSlots.back(); c->freeSlots.pop_back(); }
    else {
        if (arena_reserve(c, c->nextSlot + 1)) return -1;
        slot = c->nextSlot++;
    }
    s->slot = slot; s->context = c;
    HIP_TRY(hipMemsetAsync(c->dState.ptr + (size_t)slot * kStateDoubles, 0, kStateDoubles * sizeof(double), c->stream));   // ordered before the handle's first pull
    return 0;
}

// c->mu and c->logMu held.  What waits in the log goes on its way: one copy, one scatter launch; not waited for.
int log_send(LiveContext* c)
{
    if (c->logSent == c->logTail) return 0;
    const size_t n = c->logTail - c->logSent;
    HIP_TRY(hipMemcpyAsync(c->dLog.ptr + c->logSent, c->hLog.ptr + c->logSent, n * sizeof(LogEntry), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(live_scatter, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, c->dLog.ptr + c->logSent, (uint32_t)n,
                       c->dRingFrames.ptr, c->dRingMeta.ptr);
    HIP_TRY(hipGetLastError());
    c->logSent = c->logTail;
    c->logEpoch++;                       // every Stream::logIdx is stale from here on
    return 0;
}

// s->mu held.  One frame into the handle's ring, by way of the log.  The caller has checked ringCount < kRing.
int log_append(LiveContext* c, Stream* s, const double* p, const FrameMeta& meta, bool holdsContext)
{
    std::unique_lock<std::mutex> lg(c->logMu);
    if (c->logTail == c->hLog.cap) {     // full: send it, wait until it has arrived, start over (lock order: mu before logMu)
        std::unique_lock<std::mutex> g(c->mu, std::defer_lock);
        if (!holdsContext) { lg.unlock(); g.lock(); lg.lock(); }
        if (c->logTail == c->hLog.cap) {
            HIP_TRY(hipSetDevice(s->device));
            if (log_send(c)) return -1;
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->logTail = c->logSent = 0;
        }
    }
    const uint32_t pos = (s->ringHead + s->ringCount) & (kRing - 1);
    LogEntry& e = c->hLog.ptr[c->logTail];
    if (p) memcpy(e.p, p, sizeof e.p); else memset(e.p, 0, sizeof e.p);
    e.meta = meta;
    e.target = s->slot * kRing + pos;
    e.pad = 0;
    if (s->logEpoch != c->logEpoch) { s->logIdx.clear(); s->logEpoch = c->logEpoch; }
    s->logIdx.push_back((uint32_t)c->logTail);
    c->logTail++;
    s->span[pos] = frame_span(meta);
    s->ringCount++;
    return 0;
}

// s->mu held.  Forget what the handle has in the ring and in the log (purge, close).
void ring_drop(LiveContext* c, Stream* s)
{
    std::lock_guard<std::mutex> lg(c->logMu);
    if (s->logEpoch == c->logEpoch)
        for (uint32_t i : s->logIdx) c->hLog.ptr[i].target = kNoTarget;
    s->logIdx.clear();
    s->ringHead = (s->ringHead + s->ringCount) & (kRing - 1);
    s->ringCount = 0;
}

// Advance n live streams by up to `count` samples each (one stream per wavefront lane; one launch, unless some handle has
// more frames queued than its ring holds AND the ring's frames end before `count` samples: then the call proceeds in pieces,
// refilling the rings in between -- a pull is the same as several shorter pulls, reference src/speechPlayer.cpp:39-42).
// The callers hold every stream's mutex.  produced[i] receives speechPlayer_synthesize's return value.
// outs == nullptr: the PCM stays on the device (row i at devicePcm + i * stride), for consumers on the GPU.
#ifdef KLATT_STAMPS
unsigned long long g_streamStamps[32];
extern "C" __attribute__((visibility("default"))) void speechPlayer_debugStreamStamps(unsigned long long* out) { for (int k = 0; k < 32; ++k) { out[k] = g_streamStamps[k]; g_streamStamps[k] = 0; } }
#endif
int streams_synthesize(Stream* const* ss, int n, unsigned int count, sample* const* outs, int* produced,
                       const int16_t** devicePcm = nullptr, long long* deviceStride = nullptr)
{
    for (int i = 0; i < n; ++i) produced[i] = 0;
    if (count == 0 || n <= 0) return 0;
    static const bool trace = getenv("SPEECHPLAYER_LIVE_TRACE") != nullptr;     // host-side breakdown of a call on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    const int device = ss[0]->device, rate = ss[0]->sampleRate, mode = ss[0]->mode;
    for (int i = 1; i < n; ++i)
        if (ss[i]->device != device || ss[i]->sampleRate != rate || ss[i]->mode != mode) {
            set_error("streams advanced together must share device, sample rate and mode");
            return -1;
        }
    LiveContext* c = live_context(device);
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(device));
    const size_t padded = ((size_t)count + kTile - 1) / kTile * kTile;
    // A LONE handle -- the reference's own use (one stream pulled 8192 samples at a time, nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:62-81)
    // -- is advanced in ALL 64 lanes of its wavefront: every lane is given the same control entry (same state block, same ring, same PCM
    // row) and computes the same samples; they store the same values to the same places.  A wavefront with one active lane runs the same
    // instructions but, measured, takes 1.0 / 1.4 / 1.7 times as long from launch to launch (4.3 / 6.1 / 7.4 ms for one cfg2 utterance
    // against a steady 4.13 ms with 16 or more lanes active: tools/lone_probe2.py); with all lanes active a pull costs what 64 handles cost:
    // thirty 8192-sample pulls of one handle 3.05 -> 2.04 ms of kernel time on average (tools/single_stream_ab.sh).  The launch then takes
    // the kernel's LONE instantiation, whose fade chunks are computed side by side across the identical lanes (klatt_systolic.h,
    // stage_loop: 2.04 -> 1.84 ms); control bit 1 marks the entries.
    // The same goes for a pull of SEVERAL handles, up to "live_alone" of them (default 1536: six rounds of 256 workgroups; the two policies meet near 1850 unrelated handles): a workgroup per handle, 64 replicas each.  Handles
    // that share a wavefront pay for one another -- with unrelated handles every chunk has some lane at an event or in a fade and runs sample
    // by sample: 11.7 ms per 8192-sample pull however few they are -- while a handle alone in its wavefront costs 1.3-1.7 ms and 256 of them
    // run side by side, one per CU (the LONE instantiation's LDS allows one workgroup per CU): n handles take ceil(n / CUs) rounds of that.
    const bool replicate = n >= 1 && n <= std::max(1, g_liveAlone) && g_liveLayout != 0 && g_liveReplicate;
    const int rep = replicate ? kLanes : 1;              // control entries (lanes) per handle
    // (a FEW handles pulled together -- fewer than half a wavefront -- get their empty lanes filled with replicas of themselves too, on the
    // ordinary kernel: the same effect, and the same remedy as for the sparse wavefronts of a batch)
    const bool fillSparse = !replicate && n > 1 && n < kLanes / 2 && g_liveLayout != 0 && g_liveReplicate;
    const int nCtl = replicate ? n * kLanes : (fillSparse ? kLanes : n);
    const size_t ctlBytes = (size_t)nCtl * (sizeof(UttDesc) + sizeof(double*) + sizeof(uint32_t));
    if (c->hCtl.reserve(ctlBytes) || c->dCtl.reserve(c->hCtl.cap) || c->hResult.reserve(nCtl) || c->dResult.reserve(c->hResult.cap) ||
        c->dPcm.reserve(padded * n))
        return -1;
    if (c->orderFilled < (size_t)nCtl) {
        const size_t m = std::max<size_t>(nCtl, 1024);
        std::vector<uint32_t> iota(m);
        std::iota(iota.begin(), iota.end(), 0u);
        if (c->dOrder.reserve(m)) return -1;
        HIP_TRY(hipMemcpy(c->dOrder.ptr, iota.data(), m * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->orderFilled = m;
    }
    UttDesc* const hUtt = reinterpret_cast<UttDesc*>(c->hCtl.ptr);
    double** const hState = reinterpret_cast<double**>(c->hCtl.ptr + (size_t)nCtl * sizeof(UttDesc));
    uint32_t* const hControl = reinterpret_cast<uint32_t*>(c->hCtl.ptr + (size_t)nCtl * (sizeof(UttDesc) + sizeof(double*)));
    if (g_liveCus == 0 && !g_liveCusForced) {
        hipDeviceProp_t prop;
        g_liveCus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }

    KernelArgs a = base_args(rate);
    a.frames = c->dRingFrames.ptr; a.meta = c->dRingMeta.ptr; a.ringMask = kRing - 1;
    a.utt = reinterpret_cast<const UttDesc*>(c->dCtl.ptr);
    a.statePtrs = reinterpret_cast<double* const*>(c->dCtl.ptr + (size_t)nCtl * sizeof(UttDesc));
    a.control = reinterpret_cast<const uint32_t*>(c->dCtl.ptr + (size_t)nCtl * (sizeof(UttDesc) + sizeof(double*)));
    a.order = c->dOrder.ptr; a.pcm = c->dPcm.ptr; a.result = c->dResult.ptr; a.state = nullptr;
    a.nSlots = nCtl;
    const long long groups = (nCtl + kLanes - 1) / kLanes;

    c->done.assign(n, 0u);
    c->lastKernelMs = 0.0f; c->lastLaunches = 0;
    double tFill = 0, tRun = 0;
    unsigned int at = 0;                 // samples of the call behind us
    bool joined = false;                 // the pieces of a call in pieces are put together in dPcmJoin
    while (at < count) {
        const auto ta = now();
        // rings take what waited on the host; how far may this launch go without meeting a frame that is not in a ring?
        unsigned int piece = count - at;
        for (int i = 0; i < n; ++i) {
            Stream* s = ss[i];
            if (s->overHead < s->overflow.size()) {
                while (s->ringCount < kRing && s->overHead < s->overflow.size()) {
                    const PendingFrame& f = s->overflow[s->overHead];
                    if (log_append(c, s, (f.meta.flags & FRAME_NULL) ? nullptr : f.p, f.meta, true)) return -1;
                    s->overHead++;
                }
                if (s->overHead == s->overflow.size()) { s->overflow.clear(); s->overHead = 0; }
                else {
                    if (s->overHead >= 64 && s->overHead * 2 >= s->overflow.size()) { s->overflow.erase(s->overflow.begin(), s->overflow.begin() + s->overHead); s->overHead = 0; }
                    // the ring's last frame may be dequeued only when its successor is there to follow it
                    unsigned long long safe = 0;
                    for (uint32_t k = 0; k + 1 < s->ringCount && safe < piece; ++k) safe += s->span[(s->ringHead + k) & (kRing - 1)];
                    piece = (unsigned int)std::min<unsigned long long>(piece, safe);
                }
            }
        }
        for (int i = 0; i < n; ++i) {
            const Stream* s = ss[i];
            UttDesc& d = hUtt[(size_t)i * rep];
            d.frameStart = (long long)s->slot * kRing + s->ringHead;
            d.outStart = (long long)(i * padded);
            d.nFrames = s->ringCount;
            d.seed = s->seed; d.flags = UTT_NEEDS_NOISE;
            d.length = piece;                // the stage-parallel kernel runs exactly `piece` steps (klatt_systolic.h, STREAM)
            hState[(size_t)i * rep] = c->dState.ptr + (size_t)s->slot * kStateDoubles;
            hControl[(size_t)i * rep] = s->purgePending ? 1u : 0u;
        }
        if (replicate) {
            for (int i = 0; i < n; ++i) {
                const size_t j0 = (size_t)i * kLanes;
                hControl[j0] |= 2u;              // every lane of the wavefront advances THIS handle
                for (size_t j = j0 + 1; j < j0 + kLanes; ++j) { hUtt[j] = hUtt[j0]; hState[j] = hState[j0]; hControl[j] = hControl[j0]; }
            }
        } else if (fillSparse) {
            for (int i = n; i < nCtl; ++i) { hUtt[i] = hUtt[i % n]; hState[i] = hState[i % n]; hControl[i] = hControl[i % n]; }
        }
        const auto tb = now();
        tFill += ms(ta, tb);
        {
            std::lock_guard<std::mutex> lg(c->logMu);
            if (log_send(c)) return -1;
        }
        HIP_TRY(hipMemcpyAsync(c->dCtl.ptr, c->hCtl.ptr, ctlBytes, hipMemcpyHostToDevice, c->stream));
        a.maxSamples = piece;
#ifdef KLATT_STAMPS
        static unsigned long long* dStamps = nullptr;
        if (!dStamps) HIP_TRY(hipMalloc(&dStamps, 32 * 8));
        HIP_TRY(hipMemsetAsync(dStamps, 0, 32 * 8, c->stream));
        a.debug = groups == 1 ? dStamps : nullptr;
#endif
        HIP_TRY(hipEventRecord(c->kernelStart, c->stream));
        if (g_liveLayout == 0) {
            if (launch<true, true>(a, mode, groups, c->stream)) return -1;
        } else {
            // one workgroup per CU while they all fit (16-sample hand-overs), else two per CU (8-sample hand-overs), as for batches
            if (replicate ? launch_systolic<true, 16, 1, true, true, false, true>(a, mode, groups, c->stream)      // the LONE instantiation: 64 replicas of one handle
                : (groups <= g_liveCus ? launch_systolic<true, 16, 1, true, true>(a, mode, groups, c->stream)
                                       : launch_systolic<true, KLATT_NOISY_CH, 2, true, true>(a, mode, groups, c->stream))) return -1;
        }
        HIP_TRY(hipEventRecord(c->kernelStop, c->stream));
#ifdef KLATT_STAMPS
        if (a.debug) {
            unsigned long long h[32];
            HIP_TRY(hipMemcpyAsync(h, dStamps, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (int k = 0; k < 32; ++k) g_streamStamps[k] += h[k];
        }
#endif
        HIP_TRY(hipMemcpyAsync(c->hResult.ptr, c->dResult.ptr, (size_t)(replicate ? nCtl : n) * sizeof(UttResult), hipMemcpyDeviceToHost, c->stream));
        if (piece < count || joined) {       // a call in pieces: this piece's columns into the joined rows
            if (c->dPcmJoin.reserve(padded * n)) return -1;
            HIP_TRY(hipMemcpy2DAsync(c->dPcmJoin.ptr + at, padded * sizeof(int16_t), c->dPcm.ptr, padded * sizeof(int16_t), (size_t)piece * sizeof(int16_t), n,
                                     hipMemcpyDeviceToDevice, c->stream));
            joined = true;
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        {
            std::lock_guard<std::mutex> lg(c->logMu);      // everything sent has arrived: an empty log starts over
            if (c->logSent == c->logTail) c->logSent = c->logTail = 0;
        }
        float kms = 0.0f;
        (void)hipEventElapsedTime(&kms, c->kernelStart, c->kernelStop);
        c->lastKernelMs += kms; c->lastLaunches++;
        for (int i = 0; i < n; ++i) {
            Stream* s = ss[i];
            const UttResult& r = c->hResult.ptr[(size_t)i * rep];
            if (r.produced > piece || r.framesTaken > s->ringCount) { set_error("kernel produced %u > %u (took %u frames of %u)", r.produced, piece, r.framesTaken, s->ringCount); return -1; }
            s->purgePending = false;
            s->ringHead = (s->ringHead + r.framesTaken) & (kRing - 1);
            s->ringCount -= r.framesTaken;
            s->lastIndex = r.lastIndex;
            c->done[i] += r.produced;
        }
        at += piece;
        tRun += ms(tb, now());
    }
    const auto t2 = now();
    const int16_t* const pcm = joined ? c->dPcmJoin.ptr : c->dPcm.ptr;
    if (devicePcm) *devicePcm = pcm;
    if (deviceStride) *deviceStride = (long long)padded;
    size_t lastWithData = 0;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        produced[i] = (int)c->done[i];
        if (c->done[i]) { lastWithData = i; any = true; }
    }
    if (any && outs) {
        if (n == 1) {
            HIP_TRY(hipMemcpy(outs[0], pcm, (size_t)c->done[0] * sizeof(int16_t), hipMemcpyDeviceToHost));
        } else {
            // whole rows in pieces of about 16 MB through two pinned buffers; piece k + 1 is in flight while the
            // rows of piece k are handed to their callers
            const size_t rowsPerPiece = std::max<size_t>(1, (8u << 20) / padded);
            if (c->bounce.ensure(rowsPerPiece * padded * sizeof(int16_t))) return -1;
            const size_t rows = lastWithData + 1, nPieces = (rows + rowsPerPiece - 1) / rowsPerPiece;
            auto issue = [&](size_t k) -> int {
                const size_t r0 = k * rowsPerPiece, r1 = std::min(rows, r0 + rowsPerPiece);
                HIP_TRY(hipMemcpyAsync(c->bounce.buf[k & 1], pcm + r0 * padded, (r1 - r0) * padded * sizeof(int16_t), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipEventRecord(c->bounce.ev[k & 1], c->stream));
                return 0;
            };
            if (issue(0)) return -1;
            for (size_t k = 0; k < nPieces; ++k) {
                if (k + 1 < nPieces && issue(k + 1)) return -1;
                HIP_TRY(hipEventSynchronize(c->bounce.ev[k & 1]));
                const size_t r0 = k * rowsPerPiece, r1 = std::min(rows, r0 + rowsPerPiece);
                const int16_t* src = static_cast<const int16_t*>(c->bounce.buf[k & 1]);
                auto rows_out = [&](size_t b0, size_t b1) {
                    for (size_t i = b0; i < b1; ++i)
                        if (c->done[i]) memcpy(outs[i], src + (i - r0) * padded, (size_t)c->done[i] * sizeof(int16_t));
                };
                // a piece of several megabytes is handed out by four threads: one thread's memcpy (about 20 GB/s) is slower than the copy over PCIe
                constexpr int kHands = 4;
                const size_t share = (r1 - r0 + kHands - 1) / kHands;
                std::thread hands[kHands - 1];
                int started = 0;
                if ((r1 - r0) * padded * sizeof(int16_t) >= (4u << 20)) {
                    try {
                        for (; started < kHands - 1; ++started) {
                            const size_t b0 = std::min(r1, r0 + (started + 1) * share), b1 = std::min(r1, b0 + share);
                            hands[started] = std::thread(rows_out, b0, b1);
                        }
                    } catch (const std::system_error&) {}
                }
                rows_out(r0, started ? std::min(r1, r0 + share) : r1);
                for (int t = 0; t < started; ++t) hands[t].join();
                if (started && started < kHands - 1) rows_out(std::min(r1, r0 + (started + 1) * share), r1);     // what the threads that did not start would have done
            }
        }
    }
    if (trace)
        fprintf(stderr, "[speechPlayer/live] %d handles x %u in %d launch(es): control block + ring refill %.2f ms | log + upload + kernel (%.2f ms) + results %.2f ms | PCM to host %.2f ms | before %.2f ms\n",
                n, count, c->lastLaunches, tFill, c->lastKernelMs, tRun, ms(t2, now()), ms(t0, t2) - tFill - tRun);
    return 0;
}

}  // namespace

// ==========================================================================================
// C-ABI: the reference's five entry points
// ==========================================================================================
extern "C" {

const char* speechPlayer_lastError(void) { return g_lastError.c_str(); }
// for the library's other translation units (frame_producer.cpp); not in the public headers
void speechPlayer_internal_setError(int code, const char* message)
{
    g_lastErrorCode = 0;
    set_error_code(code);
    set_error("%s", message ? message : "");
}
int speechPlayer_lastErrorCode(void) { return g_lastErrorCode; }

speechPlayer_handle_t speechPlayer_initialize(int sampleRate)
{
    begin_call();
    int dev = pick_device(-1);
    if (dev < 0) return nullptr;
    Stream* s = new Stream;
    s->sampleRate = sampleRate;
    s->mode = g_liveMode;
    s->device = dev;
    if (stream_init_device(s)) { delete s; return nullptr; }
    std::lock_guard<std::mutex> g(g_tableMutex);
    for (size_t i = 0; i < g_streams.size(); ++i)
        if (!g_streams[i]) { g_streams[i] = s; s->id = i + 1; return reinterpret_cast<speechPlayer_handle_t>(i + 1); }
    g_streams.push_back(s);
    s->id = g_streams.size();
    return reinterpret_cast<speechPlayer_handle_t>(g_streams.size());
}

void speechPlayer_queueFrame(speechPlayer_handle_t playerHandle, speechPlayer_frame_t* framePtr, unsigned int minFrameDuration,
                             unsigned int fadeDuration, int userIndex, bool purgeQueue)
{
    begin_call();
    Stream* s = lookup(playerHandle);
    if (!s) { set_error("speechPlayer_queueFrame: invalid handle"); return; }
    FrameMeta meta;
    meta.minSamples = minFrameDuration;
    meta.fadeSamples = std::max(fadeDuration, 1u);       // reference src/speechPlayer.cpp:36
    meta.userIndex = userIndex;
    meta.flags = framePtr ? 0u : FRAME_NULL;
    LiveContext* const c = s->context;
    std::lock_guard<std::mutex> g(s->mu);
    if (purgeQueue) {                                    // reference src/frame.cpp:103-112; the state half of
        ring_drop(c, s);                                 // the purge runs in the kernel before the next sample
        s->overflow.clear(); s->overHead = 0;
        s->purgePending = true;
    }
    // copied: the caller may reuse its frame (reference src/frame.cpp:97) -- into the pinned log while the handle's ring has
    // room, else into the handle's host queue, from where the next pulls refill the ring
    if (s->overHead == s->overflow.size() && s->ringCount < kRing) {
        if (log_append(c, s, reinterpret_cast<const double*>(framePtr), meta, false)) return;
        bool eager;
        { std::lock_guard<std::mutex> lg(c->logMu); eager = c->logTail - c->logSent >= kLogEager; }
        if (eager && c->mu.try_lock()) {                 // no pull is running: the log need not wait for the next one
            std::lock_guard<std::mutex> lg(c->logMu);
            if (hipSetDevice(s->device) == hipSuccess) (void)log_send(c);
            c->mu.unlock();
        }
    } else {
        PendingFrame f;
        memset(&f, 0, sizeof f);
        f.meta = meta;
        if (framePtr) memcpy(f.p, framePtr, sizeof f.p);
        s->overflow.push_back(f);
    }
}

int speechPlayer_synthesize(speechPlayer_handle_t playerHandle, unsigned int sampleCount, sample* sampleBuf)
{
    begin_call();
    Stream* s = lookup(playerHandle);
    if (!s) { set_error("speechPlayer_synthesize: invalid handle"); return 0; }
    std::lock_guard<std::mutex> g(s->mu);
    int produced = 0;
    sample* out = sampleBuf;
    if (streams_synthesize(&s, 1, sampleCount, &out, &produced)) return 0;
    return produced;
}

int speechPlayer_getLastIndex(speechPlayer_handle_t playerHandle)
{
    Stream* s = lookup(playerHandle);
    if (!s) return -1;
    return s->lastIndex;   // unlocked, as reference src/frame.cpp:117-119
}

void speechPlayer_terminate(speechPlayer_handle_t playerHandle)
{
    begin_call();
    Stream* s = nullptr;
    {
        uintptr_t id = reinterpret_cast<uintptr_t>(playerHandle) & 0xFFFFFFFFu;
        std::lock_guard<std::mutex> g(g_tableMutex);
        if (id == 0 || id > g_streams.size()) return;
        s = g_streams[id - 1];
        g_streams[id - 1] = nullptr;
    }
    if (!s) return;
    { std::lock_guard<std::mutex> g(s->mu); }   // let a call in flight finish
    if (LiveContext* c = s->context) {
        std::lock_guard<std::mutex> g(c->mu);
        ring_drop(c, s);                        // its entries in the log go nowhere
        c->freeSlots.push_back(s->slot);        // state block and ring are the next handle's (zeroed at its initialize, in stream order)
        if (g_liveTrim && hipSetDevice(s->device) == hipSuccess) arena_trim(c);
    }
    delete s;
}

// Additive: choose the handle's noise stream (default 0) and arithmetic mode.
int speechPlayer_setNoiseSeed(speechPlayer_handle_t playerHandle, unsigned int seed)
{
    begin_call();
    Stream* s = lookup(playerHandle);
    if (!s) return -1;
    std::lock_guard<std::mutex> g(s->mu);
    s->seed = seed;
    return 0;
}

// Additive: process-wide options.  "live_layout": which kernel advances live handles (1: stage-parallel, default; 0: lane kernel).
int speechPlayer_setGlobalOption(const char* name, int value)
{
    begin_call();
    if (name && !strcmp(name, "live_layout")) { g_liveLayout = value ? 1 : 0; return 0; }
    // "live_cus": how many workgroups of live handles count as one per CU (0: the device's CU count).  Beyond it a pull of live handles
    // takes the two-workgroups-per-CU instantiation of the stream kernel -- on a 256-CU device from 16 385 handles on; a small value
    // lets a test (or a small device) reach that kernel with a few hundred handles.
    if (name && !strcmp(name, "live_cus")) { g_liveCus = value < 0 ? 0 : value; g_liveCusForced = value > 0; return 0; }
    // "live_replicate": 1 (default) a handle pulled alone is advanced in all 64 lanes of its wavefront (streams_synthesize); 0: in one lane
    // "live_mode": the arithmetic mode of handles created from now on -- 0 (default) MODE_EXACT, 1 MODE_FAST (fused multiply-adds in the filters)
    if (name && !strcmp(name, "live_mode")) { if (value != MODE_EXACT && value != MODE_FAST) { set_error("live_mode: 0 or 1"); return -1; } g_liveMode = value; return 0; }
    if (name && !strcmp(name, "live_replicate")) { g_liveReplicate = value ? 1 : 0; return 0; }
    // "live_alone": pulls of up to this many handles give every handle a wavefront of its own (default 1536; 1: only a handle pulled alone)
    if (name && !strcmp(name, "live_alone")) { g_liveAlone = value < 1 ? 1 : (value > 65536 ? 65536 : value); return 0; }
    // "live_trim": 1 = a device's arena of live handles (~100 KB of HBM per slot, grown by doubling) is released when the last handle on
    // that device is terminated -- and now, on devices where none lives; 0 (default): it stays for the next handles.
    if (name && !strcmp(name, "live_trim")) {
        g_liveTrim = value ? 1 : 0;
        if (g_liveTrim) {
            std::vector<LiveContext*> all;
            { std::lock_guard<std::mutex> g(g_liveMutex); all = g_live; }
            for (size_t dev = 0; dev < all.size(); ++dev)
                if (all[dev] && hipSetDevice((int)dev) == hipSuccess) { std::lock_guard<std::mutex> g(all[dev]->mu); arena_trim(all[dev]); }
        }
        return 0;
    }
    // "plan_hash_bits" (tests): the track planner looks at this many bits of a frame's 128-bit shape hash (default 128).  With few bits
    // different frames collide for certain, which is how the tests reach the verification of hashed shapes (klatt_verify_shared) and the
    // fall-back behind it; the PCM must not change.
    if (name && !strcmp(name, "plan_hash_bits")) {
        const int bits = value < 0 ? 0 : (value > 128 ? 128 : value);
        g_planHashMask[0] = bits >= 64 ? ~0ull : (bits == 0 ? 0ull : ((1ull << bits) - 1ull));
        g_planHashMask[1] = bits >= 128 ? ~0ull : (bits <= 64 ? 0ull : ((1ull << (bits - 64)) - 1ull));
        return 0;
    }
    set_error("unknown global option %s", name ? name : "(null)");
    return -1;
}

// Additive: advance many live handles together -- one kernel launch, one handle per wavefront lane.
// Equivalent to calling speechPlayer_synthesize(handles[i], sampleCount, sampleBufs[i]) for every i;
// produced[i] receives each call's return value.  Handles must be distinct and share a sample rate.
static int synthesize_many(speechPlayer_handle_t* handles, int nHandles, unsigned int sampleCount, sample** sampleBufs, int* produced,
                           const int16_t** devicePcm, long long* deviceStride);

int speechPlayer_synthesizeMany(speechPlayer_handle_t* handles, int nHandles, unsigned int sampleCount, sample** sampleBufs, int* produced)
{
    begin_call();
    if (nHandles > 0 && !sampleBufs) { set_error("speechPlayer_synthesizeMany: bad arguments"); return -1; }
    return synthesize_many(handles, nHandles, sampleCount, sampleBufs, produced, nullptr, nullptr);
}

// The same with the PCM left in HBM: handle i's samples start at *devicePcm + i * *rowStride (valid until the next live call
// on that device).  For consumers on the GPU, and for measuring the engine without the PCIe copy of the PCM.
int speechPlayer_synthesizeManyDevice(speechPlayer_handle_t* handles, int nHandles, unsigned int sampleCount, const sample** devicePcm,
                                      long long* rowStride, int* produced)
{
    begin_call();
    if (!devicePcm || !rowStride) { set_error("speechPlayer_synthesizeManyDevice: bad arguments"); return -1; }
    const int16_t* p = nullptr;
    const int rc = synthesize_many(handles, nHandles, sampleCount, nullptr, produced, &p, rowStride);
    *devicePcm = reinterpret_cast<const sample*>(p);
    return rc;
}

// Duration in milliseconds of the last live-handle launch on `device` (HIP events around the kernel on its stream).
float speechPlayer_lastLiveKernelMs(int device)
{
    std::lock_guard<std::mutex> g(g_liveMutex);
    return (device >= 0 && device < (int)g_live.size() && g_live[device]) ? g_live[device]->lastKernelMs : -1.0f;
}

// Kernel launches the last live call on `device` took: 1, unless it went in pieces (a handle with more frames queued than its ring holds).
int speechPlayer_lastLiveLaunches(int device)
{
    std::lock_guard<std::mutex> g(g_liveMutex);
    return (device >= 0 && device < (int)g_live.size() && g_live[device]) ? g_live[device]->lastLaunches : -1;
}

static int synthesize_many(speechPlayer_handle_t* handles, int nHandles, unsigned int sampleCount, sample** sampleBufs, int* produced,
                           const int16_t** devicePcm, long long* deviceStride)
{
    if (nHandles < 0 || (nHandles > 0 && (!handles || !produced))) { set_error("speechPlayer_synthesizeMany: bad arguments"); return -1; }
    std::vector<Stream*> ss((size_t)nHandles);
    bool ascending = true;
    {
        std::lock_guard<std::mutex> g(g_tableMutex);
        for (int i = 0; i < nHandles; ++i) {
            ss[i] = lookup_locked(handles[i]);
            if (!ss[i]) { set_error("speechPlayer_synthesizeMany: invalid handle at %d", i); return -1; }
            if (i && ss[i]->id <= ss[i - 1]->id) ascending = false;
        }
    }
    // lock in handle order (no deadlock between concurrent calls); duplicates are an error
    std::vector<Stream*> order(ss);
    if (!ascending) std::sort(order.begin(), order.end(), [](Stream* x, Stream* y) { return x->id < y->id; });
    for (size_t i = 1; i < order.size(); ++i)
        if (order[i] == order[i - 1]) { set_error("speechPlayer_synthesizeMany: handle listed twice"); return -1; }
    for (Stream* s : order) s->mu.lock();
    const int rc = streams_synthesize(ss.data(), nHandles, sampleCount, sampleBufs, produced, devicePcm, deviceStride);
    for (Stream* s : order) s->mu.unlock();
    return rc;
}

// ==========================================================================================
// C-ABI: batch
// ==========================================================================================
speechPlayer_batch_t speechPlayer_batch_create(int sampleRate, int device)
{
    begin_call();
    int dev = pick_device(device);
    if (dev < 0) return nullptr;
    Batch* b = new Batch;
    b->sampleRate = sampleRate;
    b->device = dev;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) b->cus = prop.multiProcessorCount; }
    bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&b->forkEvent, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&b->copyStream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&b->denseReady, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&b->copyDone, hipEventDisableTiming) == hipSuccess;
    { const char* e = getenv("SPEECHPLAYER_TRACKS"); if (e) b->tracks = atoi(e) ? 1 : 0; }
    { const char* e = getenv("SPEECHPLAYER_DIRECT"); if (e) b->direct = std::min(2, std::max(0, atoi(e))); }
    { const char* e = getenv("SPEECHPLAYER_DIRECT_LEAN"); if (e) b->directLean = std::min(1, std::max(-1, atoi(e))); }
    for (int i = 0; i < 6 && ok; ++i)
        ok = hipStreamCreateWithFlags(&b->side[i], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&b->join[i], hipEventDisableTiming) == hipSuccess;
    { const char* e = getenv("SPEECHPLAYER_QUIET_LAST"); if (e) b->quietLast = atoi(e) ? 1 : 0; }
    if (!ok) {
        set_error("cannot create a stream on device %d", dev);
        speechPlayer_batch_destroy(b);
        return nullptr;
    }
    return b;
}

void speechPlayer_batch_destroy(speechPlayer_batch_t batch)
{
    Batch* b = static_cast<Batch*>(batch);
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    for (int i = 0; i < 6; ++i) {
        if (b->side[i]) { (void)hipStreamSynchronize(b->side[i]); (void)hipStreamDestroy(b->side[i]); }
        if (b->join[i]) (void)hipEventDestroy(b->join[i]);
    }
    if (b->forkEvent) (void)hipEventDestroy(b->forkEvent);
    if (b->copyStream) { (void)hipStreamSynchronize(b->copyStream); (void)hipStreamDestroy(b->copyStream); }
    if (b->denseReady) (void)hipEventDestroy(b->denseReady);
    if (b->copyDone) (void)hipEventDestroy(b->copyDone);
    b->dDense.release(); b->dDenseStart.release(); b->dFacts.release(); b->hFacts.release();
    b->dRecords.release(); b->dShapeTable.release(); b->dRep.release(); b->dMismatch.release();
    b->dFrames.release(); b->dMeta.release(); b->dUtt.release(); b->dOrder.release(); b->dPcm.release(); b->dResult.release();
    b->dFloat.release(); b->dDebug.release(); b->dDigest.release(); b->bounce.release();
    b->dFlatRef.release(); b->dSourceRef.release(); b->dJobs.release(); b->dShapes.release(); b->dTrack.release();
    b->dDirectJobs.release(); b->dDirectFirst.release(); b->dDirectHdr.release(); b->dDirectRec.release();
    delete b;
}

int speechPlayer_batch_setOption(speechPlayer_batch_t batch, const char* name, int value)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b || !name) return -1;
    if (!strcmp(name, "mode")) {
        if (value != MODE_EXACT && value != MODE_FAST) { set_error("mode %d not available", value); return -1; }
        b->mode = value;
        return 0;
    }
    if (!strcmp(name, "sort")) { b->sortByLength = value ? 1 : 0; return 0; }
    if (!strcmp(name, "quiet_last")) { b->quietLast = value ? 1 : 0; return 0; }      // read by every launch (batch_launch)
    if (!strcmp(name, "layout")) { b->layout = value < 0 ? -1 : (value > 2 ? 1 : value); return 0; }
    // tracks: planned by setUtterances (set the option before it), used by the stage-parallel layouts
    if (!strcmp(name, "tracks")) { b->tracks = value ? 1 : 0; return 0; }
    if (!strcmp(name, "track_budget_mb")) { b->trackBudgetMB = value < 0 ? 0 : value; return 0; }
    // direct: read by setUtterances (set the option before it); 0 never, 1 unless the lanes are time-aligned (default), 2 always
    if (!strcmp(name, "direct")) { b->direct = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
    // direct_lean: the direct stages two workgroups to a CU (1), one (0), or the engine's choice (-1, default); same PCM in MODE_EXACT
    if (!strcmp(name, "direct_lean")) { b->directLean = value < 0 ? -1 : (value ? 1 : 0); return 0; }
    set_error("unknown option %s", name);
    return -1;
}

// What a set call describes: `nLists` frame lists -- as full frames (frames, durations, marks, silences) or as records over a shape table
// -- and `nUtt` utterances, each speaking one list (listOf == nullptr: utterance u speaks list u).  speechPlayer_batch_setUtterances is
// the case "every utterance its own list, full frames".
struct SetInput {
    long long nLists = 0;
    const long long* listStart = nullptr;
    const speechPlayer_frame_t* frames = nullptr;
    const unsigned int* minDur = nullptr;
    const unsigned int* fadeDur = nullptr;
    const int* userIndex = nullptr;
    const unsigned char* isNull = nullptr;
    const speechPlayer_frameRecord_t* records = nullptr;
    long long nShapes = 0;
    const speechPlayer_frame_t* shapes = nullptr;
    long long nUtt = 0;
    const unsigned int* listOf = nullptr;
    const unsigned int* seeds = nullptr;
    bool noTracks = false;       // the second attempt of a batch whose shared shapes failed their verification
};
static_assert(sizeof(speechPlayer_frameRecord_t) == sizeof(FrameRecord), "record layout");

static void batch_clear(Batch* b)
{
    b->nUtt = 0; b->nFrames = 0; b->nFramesSpoken = 0; b->nLists = 0; b->nSlots = 0; b->nQuiet = 0; b->nNoNasal = 0; b->nNoNasalUtt = 0; b->totalSamples = 0; b->poolSamples = 0;
    b->nTracked = 0; b->nTrackedUtt = 0; b->nJobs = 0; b->trackEntries = 0; b->nDirect = 0; b->nDirectUtt = 0; b->nDirectFrames = 0;
    b->lens.clear(); b->outStart.assign(1, 0); b->results.clear(); b->resultsFresh = false; b->floatFresh = false;
    b->uttFrameStart.clear(); b->uttFrames.clear();
}

static int batch_set(Batch* b, const SetInput& in);

static int batch_set_guarded(const char* what, speechPlayer_batch_t batch, const SetInput& in)
{
    begin_call();
    try {       // nothing may throw across the C ABI (host allocations of a large batch; the planning threads)
        return batch_set(static_cast<Batch*>(batch), in);
    } catch (const std::exception& e) {
        set_error("%s: %s", what, e.what());
        return -1;
    }
}

int speechPlayer_batch_setUtterances(speechPlayer_batch_t batch, long long nUtterances, const long long* frameStart,
                                     const speechPlayer_frame_t* frames, const unsigned int* minFrameDuration,
                                     const unsigned int* fadeDuration, const int* userIndex, const unsigned char* isNull,
                                     const unsigned int* noiseSeed)
{
    SetInput in;
    in.nLists = nUtterances; in.listStart = frameStart; in.frames = frames; in.minDur = minFrameDuration; in.fadeDur = fadeDuration;
    in.userIndex = userIndex; in.isNull = isNull; in.nUtt = nUtterances; in.seeds = noiseSeed;
    return batch_set_guarded("setUtterances", batch, in);
}

int speechPlayer_batch_setUtterancesShared(speechPlayer_batch_t batch, long long nLists, const long long* listStart,
                                           const speechPlayer_frame_t* frames, const unsigned int* minFrameDuration, const unsigned int* fadeDuration,
                                           const int* userIndex, const unsigned char* isNull, long long nUtterances, const unsigned int* listOf,
                                           const unsigned int* noiseSeed)
{
    SetInput in;
    in.nLists = nLists; in.listStart = listStart; in.frames = frames; in.minDur = minFrameDuration; in.fadeDur = fadeDuration;
    in.userIndex = userIndex; in.isNull = isNull; in.nUtt = nUtterances; in.listOf = listOf; in.seeds = noiseSeed;
    return batch_set_guarded("setUtterancesShared", batch, in);
}

int speechPlayer_batch_setRecords(speechPlayer_batch_t batch, long long nShapes, const speechPlayer_frame_t* shapes,
                                  long long nLists, const long long* listStart, const speechPlayer_frameRecord_t* records,
                                  long long nUtterances, const unsigned int* listOf, const unsigned int* noiseSeed)
{
    SetInput in;
    in.nLists = nLists; in.listStart = listStart; in.records = records; in.nShapes = nShapes; in.shapes = shapes;
    in.nUtt = nUtterances; in.listOf = listOf; in.seeds = noiseSeed;
    if (nShapes < 0 || (nShapes > 0 && !shapes)) { begin_call(); set_error("setRecords: bad shape table"); return -1; }
    if (!records && nLists > 0 && listStart && listStart[nLists] > 0) { begin_call(); set_error("setRecords: no records"); return -1; }
    static const speechPlayer_frameRecord_t none = {0.0, 0.0, SPEECHPLAYER_RECORD_SILENCE, 0u, 0u, -1};
    if (!in.records) in.records = &none;      // (an empty batch: the record path all the same)
    return batch_set_guarded("setRecords", batch, in);
}

static int batch_set(Batch* b, const SetInput& in)
{
    const long long nL = in.nLists, nU = in.nUtt;
    const long long* const listStart = in.listStart;
    const unsigned int* const listOf = in.listOf;
    const bool byRecords = in.records != nullptr;
    if (!b || nL < 0 || nU < 0 || !listStart) { set_error("set: bad arguments"); return -1; }
    if (nU >= 0xFFFFFFFFll || nL >= 0xFFFFFFFFll) { set_error("set: too many utterances"); return -1; }
    if (!listOf && nU != nL) { set_error("set: %lld utterances for %lld lists and no listOf", nU, nL); return -1; }
    static const bool setTrace = getenv("SPEECHPLAYER_SET_TRACE") != nullptr;      // where the call's time goes, on stderr
    const auto tSet = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (setTrace) fprintf(stderr, "[speechPlayer/set] %s: %.1f ms since the start\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tSet).count());
    };
    HIP_TRY(hipSetDevice(b->device));
    // validate the index arrays before anything reads through them
    if (listStart[0] != 0) { set_error("set: frameStart[0] must be 0"); return -1; }
    for (long long l = 0; l < nL; ++l)
        if (listStart[l + 1] < listStart[l]) { set_error("set: frameStart not monotone at %lld", l); return -1; }
    const long long nF = listStart[nL];
    if (nF > 0 && !byRecords && (!in.frames || !in.minDur || !in.fadeDur)) { set_error("set: bad frame arrays"); return -1; }
    if (listOf) {
        std::atomic<long long> bad{-1};
        parallel_ranges(nU, 1 << 16, [&](long long a, long long e) {
            for (long long u = a; u < e; ++u) if ((long long)listOf[u] >= nL) { long long none = -1; bad.compare_exchange_strong(none, u); }
        });
        if (bad.load() >= 0) { set_error("set: listOf[%lld] is not a list", bad.load()); return -1; }
    }
    if (byRecords) {
        std::atomic<long long> bad{-1};
        parallel_ranges(nF, 1 << 16, [&](long long a, long long e) {
            for (long long k = a; k < e; ++k)
                if (in.records[k].shape != SPEECHPLAYER_RECORD_SILENCE && (long long)in.records[k].shape >= in.nShapes) { long long none = -1; bad.compare_exchange_strong(none, k); }
        });
        if (bad.load() >= 0) { set_error("setRecords: record %lld names shape %u of %lld", bad.load(), in.records[bad.load()].shape, in.nShapes); return -1; }
    }
    auto list_of = [&](long long u) -> long long { return listOf ? (long long)listOf[u] : u; };
    // everything below is built in locals and committed to the Batch only after the uploads succeeded: a call that fails
    // validation leaves the previous batch in place, one that fails while uploading leaves an empty batch
    // Per-frame work arrays are kept between calls (per calling thread): a fresh 25-50 MB vector costs its page faults on first touch
    // and its unmapping on release -- together ~15 ms of a 65 ms call for BASELINE configs[2].  Every element is written below before it
    // is read.  (Released again when a batch was very large: kScratchKeepFrames.)
    constexpr long long kScratchKeepFrames = 4000000;
    // (RawVector: resize() leaves new elements uninitialised, so a fresh array is first touched by the threads that fill it)
    static thread_local RawVector<FrameMeta> metaScratch;
    static thread_local RawVector<FlatRef> flatRefScratch;
    static thread_local RawVector<DirectJob> directJobsScratch;
    static thread_local RawVector<FrameFacts> factsScratch;
    struct ScratchRelease {
        long long nF;
        ~ScratchRelease()
        {
            if (nF > kScratchKeepFrames) {
                RawVector<FrameMeta>().swap(metaScratch); RawVector<FlatRef>().swap(flatRefScratch); RawVector<FrameFacts>().swap(factsScratch);
                RawVector<DirectJob>().swap(directJobsScratch);
            }
        }
    } scratchRelease{nF};
    RawVector<FrameMeta>& meta = metaScratch;
    meta.resize((size_t)nF);
    parallel_ranges(nF, 1 << 16, [&](long long a, long long e) {
        if (byRecords)
            for (long long k = a; k < e; ++k) {
                const speechPlayer_frameRecord_t& r = in.records[k];
                meta[k].minSamples = r.minFrameDuration;
                meta[k].fadeSamples = std::max(r.fadeDuration, 1u);      // reference src/speechPlayer.cpp:36
                meta[k].userIndex = r.userIndex;
                meta[k].flags = r.shape == SPEECHPLAYER_RECORD_SILENCE ? FRAME_NULL : 0u;
            }
        else
            for (long long k = a; k < e; ++k) {
                meta[k].minSamples = in.minDur[k];
                meta[k].fadeSamples = std::max(in.fadeDur[k], 1u);       // reference src/speechPlayer.cpp:36
                meta[k].userIndex = in.userIndex ? in.userIndex[k] : -1;
                meta[k].flags = (in.isNull && in.isNull[k]) ? FRAME_NULL : 0u;
            }
    });
    // per list, from the durations alone: its length (closed form, reference src/frame.cpp:41-80)
    std::vector<uint32_t> lensL((size_t)nL, 0);
    std::atomic<long long> tooLong{-1};
    parallel_ranges(nL, 4096, [&](long long la, long long le) {
        for (long long l = la; l < le; ++l) {
            unsigned long long len = 0;
            for (long long k = listStart[l]; k < listStart[l + 1]; ++k) {
                const unsigned long long m = meta[k].minSamples, f = meta[k].fadeSamples;
                len += std::max(m, f + 1) + 1;   // samples one request spans
            }
            if (len >= 0xFFFFFFFFull) { long long none = -1; tooLong.compare_exchange_strong(none, l); len = 0; }
            lensL[l] = (uint32_t)len;
        }
    });
    if (tooLong.load() >= 0) { set_error("utterance (frame list %lld) too long (4294967295 samples or more)", tooLong.load()); return -1; }
    // From here on nothing but an allocation or a copy can fail (which leaves an empty batch).  The frames -- by far the largest upload
    // of a batch of full frames, 0.6 GB for BASELINE configs[2] -- start on their way now: on a thread of their own from pageable memory
    // (such a copy keeps its caller until it is done), as one asynchronous DMA from page-locked memory (speechPlayer_hostAlloc); records
    // travel the same way (32 bytes per frame) with their shape table, and klatt_expand_frames builds frames and meta words behind them.
    int earlyRc = 0;
    std::string earlyErr;
    std::thread early;
    bool earlyStarted = false;
    // while a DMA reads the caller's arrays no exit may leave it running: every return below passes this guard
    struct CopyGuard { Batch* b; bool armed = false; ~CopyGuard() { if (armed) (void)hipStreamSynchronize(b->copyStream); } } copyGuard{b};
    const bool framesPinned = !byRecords && nF > 0 && is_pinned(in.frames);
    if (byRecords) {
        if (b->dFrames.reserve(std::max<size_t>((size_t)nF * kNumParams, 1)) || b->dMeta.reserve(std::max<size_t>((size_t)nF, 1)) ||
            b->dRecords.reserve(std::max<size_t>((size_t)nF, 1)) || b->dShapeTable.reserve(std::max<size_t>((size_t)in.nShapes * kNumParams, 1))) { batch_clear(b); return -1; }
        batch_clear(b);      // the batch on the device is the old frames with the new ones over them from now on: it no longer exists
        copyGuard.armed = true;
        if (in.nShapes) HIP_TRY(hipMemcpyAsync(b->dShapeTable.ptr, in.shapes, (size_t)in.nShapes * sizeof(speechPlayer_frame_t), hipMemcpyHostToDevice, b->copyStream));
        if (nF) {
            HIP_TRY(hipMemcpyAsync(b->dRecords.ptr, in.records, (size_t)nF * sizeof(FrameRecord), hipMemcpyHostToDevice, b->copyStream));
            const unsigned grid = (unsigned)std::min<long long>((nF + 63) / 64, 8192);      // tiles of 64 frames
            hipLaunchKernelGGL(klatt_expand_frames, dim3(grid), dim3(256), 0, b->copyStream, b->dRecords.ptr, b->dShapeTable.ptr, b->dFrames.ptr, b->dMeta.ptr, nF);
            HIP_TRY(hipGetLastError());
        }
        earlyStarted = true;
    } else if (nF > 0 && (framesPinned || (size_t)nF * kNumParams * sizeof(double) >= (32u << 20))) {
        if (b->dFrames.reserve((size_t)nF * kNumParams)) { batch_clear(b); return -1; }
        batch_clear(b);
        if (framesPinned) {
            copyGuard.armed = true;
            HIP_TRY(hipMemcpyAsync(b->dFrames.ptr, in.frames, (size_t)nF * kNumParams * sizeof(double), hipMemcpyHostToDevice, b->copyStream));
            earlyStarted = true;
        } else
        try {
            early = std::thread([&, dev = b->device, dst = b->dFrames.ptr]() {
                hipError_t e = hipSetDevice(dev);
                if (e == hipSuccess) e = hipMemcpy(dst, in.frames, (size_t)nF * kNumParams * sizeof(double), hipMemcpyHostToDevice);
                if (e != hipSuccess) { earlyRc = -1; earlyErr = hipGetErrorString(e); }
            });
            earlyStarted = true;
        } catch (const std::system_error&) {}
    }
    struct JoinEarly { std::thread& t; ~JoinEarly() { if (t.joinable()) t.join(); } } joinEarly{early};
    // What the planning needs to know of every frame (klatt_plan.h: a word of flags, and something that stands for its 45 shape values),
    // in ONE pass.  Records: the flags of their shape (evaluated once per row of the table) and of their pitches; the shape's number
    // stands for the values, exactly.  Frames on their way to the device as a DMA (page-locked memory) are looked at THERE, behind the
    // copy on the same stream, and 24 bytes per frame come back: the host never reads them.  Pageable frames are read here, by the
    // host's threads, while the staging thread copies them.
    const double maxBwDirect = 690.0 * b->sampleRate / M_PI, maxFDirect = 9900.0 * b->sampleRate / (2.0 * M_PI);
    const FrameFacts* facts = nullptr;
    const bool factsOnDevice = framesPinned && earlyStarted;
    if (byRecords) {
        std::vector<uint32_t> shapeFl((size_t)in.nShapes);
        for (long long s = 0; s < in.nShapes; ++s) shapeFl[(size_t)s] = shape_flags(reinterpret_cast<const double*>(in.shapes + s), maxFDirect, maxBwDirect);
        RawVector<FrameFacts>& fv = factsScratch;
        fv.resize((size_t)nF);
        parallel_ranges(nF, 1 << 15, [&](long long a, long long e) {
            for (long long k = a; k < e; ++k) {
                const speechPlayer_frameRecord_t& r = in.records[k];
                FrameFacts f;
                f.h0 = r.shape; f.h1 = ~0ull; f.pad = 0;
                f.flags = r.shape == SPEECHPLAYER_RECORD_SILENCE ? 0u : (shapeFl[r.shape] | pitch_flags(r.voicePitch, r.endVoicePitch));
                fv[k] = f;
            }
        });
        facts = fv.data();
    } else if (factsOnDevice) {
        if (b->dFacts.reserve((size_t)nF) || b->hFacts.ensure((size_t)nF * sizeof(FrameFacts))) return -1;
        const unsigned grid = (unsigned)std::min<long long>((nF + 255) / 256, 1 << 16);
        hipLaunchKernelGGL(klatt_frame_facts, dim3(grid), dim3(256), 0, b->copyStream, b->dFrames.ptr, b->dFacts.ptr, nF, maxFDirect, maxBwDirect);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(b->hFacts.ptr, b->dFacts.ptr, (size_t)nF * sizeof(FrameFacts), hipMemcpyDeviceToHost, b->copyStream));
        facts = static_cast<const FrameFacts*>(b->hFacts.ptr);
    } else if (nF > 0) {
        RawVector<FrameFacts>& fv = factsScratch;      // (a reference: the worker threads must write the CALLER's scratch, not their own thread_local one)
        fv.resize((size_t)nF);
        parallel_ranges(nF, 1 << 13, [&](long long a, long long e) {
            for (long long k = a; k < e; ++k) fv[k] = frame_facts(reinterpret_cast<const double*>(in.frames + k), maxFDirect, maxBwDirect);
        });
        facts = fv.data();
    }
    // per utterance: its length, where its PCM goes
    std::vector<uint32_t> lens((size_t)nU, 0);
    std::vector<long long> outStart((size_t)nU + 1, 0);
    std::vector<long long> denseStart((size_t)nU + 1);      // the utterances back to back: what speechPlayer_batch_readAll hands out
    long long total = 0, pool = 0, spoken = 0;
    for (long long u = 0; u < nU; ++u) {
        const long long l = list_of(u);
        lens[u] = lensL[l];
        outStart[u] = pool;
        denseStart[u] = total;
        total += (long long)lens[u];
        pool += ((long long)lens[u] + kTile - 1) / kTile * kTile;
        spoken += listStart[l + 1] - listStart[l];
    }
    outStart[nU] = pool;
    denseStart[nU] = total;
    // how many utterances speak each list
    std::vector<uint32_t> weight((size_t)nL, listOf ? 0u : 1u);
    if (listOf) for (long long u = 0; u < nU; ++u) ++weight[listOf[u]];
    lap("meta, lengths, frame facts started");
    // A list's TIMING: a hash of its sequence of frame durations, fades and silences -- the same text at the same speed, whatever
    // the pitch, the voice or the noise seed.  Lanes with one timing dequeue and fade on the same samples (lane packing, below).
    std::vector<unsigned long long> timingL((size_t)nL);
    parallel_ranges(nL, 4096, [&](long long la, long long le) {
        for (long long l = la; l < le; ++l) {
            unsigned long long h = 0x9E3779B97F4A7C15ull ^ (unsigned long long)(listStart[l + 1] - listStart[l]);
            for (long long k = listStart[l]; k < listStart[l + 1]; ++k) {
                h ^= ((unsigned long long)meta[k].minSamples << 32) ^ meta[k].fadeSamples ^ ((unsigned long long)(meta[k].flags & FRAME_NULL) << 63);
                h *= 0xFF51AFD7ED558CCDull; h ^= h >> 29;
            }
            timingL[l] = h;
        }
    });
    // the facts are back (device path): per list, whether it needs its noise sources, whether it may skip the nasal pair, and -- for
    // the tracks and the direct stages below -- whether all its parameters are finite (bit 0 of `shape`) and within the range of
    // klatt_math.h (bit 1).  NULL frames carry no parameters of their own.
    if (factsOnDevice) HIP_TRY(hipStreamSynchronize(b->copyStream));
    std::vector<unsigned char> shapeL((size_t)nL, 0);
    std::vector<uint32_t> flagsL((size_t)nL, 0);
    parallel_ranges(nL, 4096, [&](long long la, long long le) {
        for (long long l = la; l < le; ++l) {
            uint32_t fl = 0;
            for (long long k = listStart[l]; k < listStart[l + 1]; ++k)
                if (!(meta[k].flags & FRAME_NULL)) fl |= facts[k].flags;
            const bool finite = !(fl & FACT_NONFINITE), needsNoise = (fl & FACT_NOISE) || !finite;
            flagsL[l] = needsNoise ? UTT_NEEDS_NOISE : (!(fl & FACT_NASAL) ? UTT_NO_NASAL : 0u);
            shapeL[l] = finite ? (!(fl & FACT_UNBOUNDED) ? 3 : 1) : 0;
        }
    });
    lap("classification");
    // how many utterances have a given (timing, length): lanes of such a run fade together
    auto run_key = [&](long long l) { return timingL[l] ^ ((unsigned long long)lensL[l] * 0x9E3779B97F4A7C15ull); };
    // A quiet utterance whose timing too few others share cannot fill a wavefront of the quiet kernels with lanes that fade together:
    // its wavefront would run every chunk sample by sample, evaluating exp / cos for whichever lane is fading (a few workgroups that
    // take longer than the whole flat launch: 24 ms for the 8192 quiet utterances of a batch with 65 536 different timings).  Such an
    // utterance goes with the noisy ones instead -- same PCM (its noise gains are zero: the sources add exactly 0), flat stages.
    constexpr long long kQuietRunMin = 32;
    std::vector<std::pair<long long, uint32_t>> rerouted;      // (list, its flags as a quiet one): back to the quiet kernels if it gets no tracks
    const bool wantTracks = b->tracks && !in.noTracks;
    const bool wantDirect = b->direct && b->layout != 0 && nF > 0 && nF < 0xFFFFFFFFll;
    if ((wantTracks || wantDirect) && nF > 0 && b->layout == -1) {      // (an explicit layout is taken at its word)
        std::unordered_map<unsigned long long, long long> runOf;
        for (long long l = 0; l < nL; ++l)
            if (!(flagsL[l] & UTT_NEEDS_NOISE)) runOf[run_key(l)] += weight[l];
        // (quiet utterances that ALL share one timing fade together however few they are: a single sentence, a handful of copies)
        for (long long l = 0; l < nL && runOf.size() > 1; ++l)
            if (!(flagsL[l] & UTT_NEEDS_NOISE) && runOf[run_key(l)] < kQuietRunMin) {
                rerouted.emplace_back(l, flagsL[l]);
                flagsL[l] = (flagsL[l] | UTT_NEEDS_NOISE) & ~UTT_NO_NASAL;
            }
    }
    lap("timing hashes, quiet runs");
    // ---- tracks (klatt_tracks.h) for the noisy lists whose parameters are all finite: plan_tracks ---------
    // ---- and the direct stages (klatt_direct.h) for those among them that get none ------------------------
    TrackPlan plan;
    std::vector<unsigned char> eligible;
    if ((wantTracks || wantDirect) && nF > 0) {
        // A NaN anywhere ("hold" targets, reference src/utils.h:21) or an infinite parameter keeps an utterance with the untracked kernel.
        // Finite parameters whose COEFFICIENTS overflow (a huge bandwidth or frequency) are tracked all the same: klatt_tracks evaluates
        // the same expressions as the kernels' own coefficient code, so the track holds the same inf / NaN the kernel would have computed.
        eligible.assign((size_t)nL, 0);
        for (long long l = 0; l < nL; ++l)
            if ((flagsL[l] & UTT_NEEDS_NOISE) && weight[l]) eligible[l] = shapeL[l];
    }
    lap("eligibility");
    const FrameSource source{in.frames, in.records, in.shapes};
    if (wantTracks && nF > 0) {
        plan_tracks(nL, listStart, source, facts, meta.data(), eligible.data(), b->trackBudgetMB, plan);
        lap("tracks planned");
        // MODE_FAST, lanes that fade at unrelated times, tracks far beyond the caches (every fading lane streams through a track of its
        // own: the jittered batch's 445 MB): the lean direct stages, whose pole recurrences compute what the tracks would deliver, are
        // faster than the flat stages waiting for rows -- 18.4 against 21.8 ms (bench.py, jittered_durations) -- so such a batch is
        // not tracked.  (MODE_EXACT keeps its tracks: the polynomials cost more than the rows' latency, 22 ms against ~35.)
        bool useTracks = true;
        if (wantDirect && b->direct == 1 && b->mode == MODE_FAST && plan.entries * sizeof(double2) > (128ull << 20)) {
            std::unordered_map<unsigned long long, long long> runOf;
            long long tracked = 0, direct = 0, inRuns = 0;
            for (long long l = 0; l < nL; ++l)
                if (plan.tracked[l]) { tracked += weight[l]; direct += (eligible[l] & 2) ? weight[l] : 0; runOf[run_key(l)] += weight[l]; }
            for (const auto& kv : runOf) if (kv.second >= 32) inRuns += kv.second;
            const long long groups = (tracked + kLanes - 1) / kLanes;
            if (b->sortByLength && inRuns * 2 <= tracked && direct == tracked && groups > b->cus) useTracks = false;
        }
        if (useTracks)
            for (long long l = 0; l < nL; ++l)
                if (plan.tracked[l]) flagsL[l] |= UTT_TRACKED | (plan.kinds[l] << kUttKindShift);
        if (!useTracks) { plan.jobs.clear(); plan.entries = 0; plan.tracked.assign((size_t)nL, 0); }
    }
    if (wantDirect) {
        // The direct stages are for lanes that fade at unrelated times.  A group whose wavefronts hold equally timed utterances (the
        // BASELINE recipes without their tracks, a batch of few sentences in many voices) runs whole chunks on the uniform paths of the
        // stages with the frame state machine, two workgroups per CU, and is faster there (cfg2 without tracks 13.9 against 19.1 ms,
        // DESIGN.md section 4.7): "direct" = 1 decides by the share of the candidates that sit in runs of 32 or more equally long,
        // equally timed utterances; 2 takes the direct stages whatever the timing.
        bool take = true;
        b->directAligned = false;
        {
            std::unordered_map<unsigned long long, long long> runOf;
            long long candidates = 0, inRuns = 0;
            for (long long l = 0; l < nL; ++l)
                if ((eligible[l] & 2) && !(flagsL[l] & UTT_TRACKED)) { candidates += weight[l]; runOf[run_key(l)] += weight[l]; }
            for (const auto& kv : runOf) if (kv.second >= 32) inRuns += kv.second;
            // (without the sort by length and timing nothing is side by side; in MODE_FAST the direct stages advance coefficients by
            // recurrences and win on the aligned batches whose fades move everything too -- "distinct" 19.2 -> 15.9 ms -- while a batch
            // of few moving kinds loses 8 % there: cfg2 without its tracks 12.7 -> 13.7)
            b->directAligned = b->sortByLength && inRuns * 2 > candidates;
            if (b->direct == 1) take = !b->directAligned || b->mode == MODE_FAST;
        }
        if (take)
            for (long long l = 0; l < nL; ++l)
                if ((eligible[l] & 2) && !(flagsL[l] & UTT_TRACKED)) flagsL[l] |= UTT_DIRECT;
    }
    // (a re-routed quiet list that got neither tracks nor the direct stages goes back to the quiet kernels)
    for (const auto& r : rerouted)
        if (!(flagsL[r.first] & (UTT_TRACKED | UTT_DIRECT))) flagsL[r.first] = r.second;
    // the direct lists' fades: per frame where its fade starts from and ends on (reference src/frame.cpp:55-72: silence keeps
    // the previous request's values with the gain gated off; the first frame after silence starts from its own values with the gain
    // gated off; any other frame fades from the previous request's values) -- klatt_seeds reads the values themselves on the device
    RawVector<DirectJob>& directJobs = directJobsScratch;
    directJobs.clear();
    std::vector<uint32_t> directFirstL, directFirst;
    long long nDirectUtt = 0;
    if (wantDirect) {
        directFirstL.assign((size_t)nL, 0u);
        for (long long l = 0; l < nL; ++l) {
            if (!(flagsL[l] & UTT_DIRECT)) continue;
            nDirectUtt += weight[l];
            directFirstL[l] = (uint32_t)directJobs.size();
            walk_fade_ends(listStart[l], listStart[l + 1], meta.data(), directJobs);
        }
        directFirst.resize((size_t)nU);
        parallel_ranges(nU, 1 << 16, [&](long long a, long long e) { for (long long u = a; u < e; ++u) directFirst[u] = directFirstL[list_of(u)]; });
    }
    RawVector<FlatRef>& flatRef = flatRefScratch;
    if (!plan.jobs.empty()) {
        flatRef.resize((size_t)nF);
        parallel_ranges(nF, 1 << 16, [&](long long ka, long long ke) {
        for (long long k = ka; k < ke; ++k) {
            const unsigned long long m = meta[k].minSamples, f = meta[k].fadeSamples;
            flatRef[k] = FlatRef{(uint32_t)plan.ref[k].off, plan.ref[k].mask, meta[k].fadeSamples, (uint32_t)std::min<unsigned long long>(std::max(m, f + 1) + 1, 0xFFFFFFFFull)};
        }
        });
    }
    lap("direct jobs, flat / source references");
    // the utterances, each with its list's frames, flags and length
    std::vector<UttDesc> utt((size_t)nU);
    std::vector<long long> uttFrameStart((size_t)nU);
    std::vector<uint32_t> uttFrames((size_t)nU);
    parallel_ranges(nU, 1 << 15, [&](long long a, long long e) {
        for (long long u = a; u < e; ++u) {
            const long long l = list_of(u);
            UttDesc d;
            d.frameStart = listStart[l];
            d.outStart = outStart[u];
            d.nFrames = (uint32_t)(listStart[l + 1] - listStart[l]);
            d.seed = in.seeds ? in.seeds[u] : (uint32_t)u;
            d.flags = flagsL[l];
            d.length = lensL[l];
            utt[u] = d;
            uttFrameStart[u] = d.frameStart; uttFrames[u] = d.nFrames;
        }
    });
    std::vector<TrackJob>& jobs = plan.jobs;
    const unsigned long long trackEntries = plan.entries;
    // lane packing: similar lengths share a wavefront (longest first), so lanes finish together
    // (within the quiet group and within the noisy group, which are launched as separate kernels)
    std::vector<unsigned long long> timing;      // per utterance (its list's)
    if (listOf) { timing.resize((size_t)nU); for (long long u = 0; u < nU; ++u) timing[u] = timingL[listOf[u]]; }
    else timing.swap(timingL);
    std::vector<uint32_t> order((size_t)nU);
    std::iota(order.begin(), order.end(), 0u);
    auto quietEnd = std::stable_partition(order.begin(), order.end(), [&](uint32_t x) { return !(utt[x].flags & UTT_NEEDS_NOISE); });
    const long long nQuiet = quietEnd - order.begin();
    auto noNasalEnd = std::stable_partition(order.begin(), quietEnd, [&](uint32_t x) { return (utt[x].flags & UTT_NO_NASAL) != 0; });
    const long long nNoNasal = noNasalEnd - order.begin();
    auto trackedEnd = std::stable_partition(quietEnd, order.end(), [&](uint32_t x) { return (utt[x].flags & UTT_TRACKED) != 0; });
    const long long nTrackedUtt = trackedEnd - quietEnd;
    long long nTracked = nTrackedUtt;      // slots of the tracked group (utterances + padding, below)
    auto directEnd = std::stable_partition(trackedEnd, order.end(), [&](uint32_t x) { return (utt[x].flags & UTT_DIRECT) != 0; });
    long long nDirectSlots = directEnd - trackedEnd;      // slots of the direct group (utterances + padding to whole wavefronts, below)
    if (b->sortByLength) {
        // Within a group: longest first, and utterances with the same TIMING (the same sequence of frame durations, fades and
        // silences: the same text at the same speed, whatever the pitch, the voice or the noise seed) side by side.  Lanes
        // with one timing dequeue and fade on the same samples, so their wavefront runs whole chunks on the uniform paths.
        auto before = [&](uint32_t x, uint32_t y) { return lens[x] != lens[y] ? lens[x] > lens[y] : timing[x] < timing[y]; };
        std::stable_sort(order.begin(), noNasalEnd, before);
        std::stable_sort(noNasalEnd, quietEnd, before);
        std::stable_sort(quietEnd, trackedEnd, before);
        std::stable_sort(trackedEnd, directEnd, before);
        std::stable_sort(directEnd, order.end(), before);
        // The noisy groups (64 utterances per wavefront): a wavefront that holds two timings runs every chunk on the general
        // path -- ~2.6 times the time of a pure one for the whole length of its utterances, and it is the last to finish.  So a
        // run of equally timed utterances that filled at least a quarter of its last wavefront, or that is followed by a run
        // of 64 or more, ends its wavefront there: the remaining lanes stay empty (order slot 0xFFFFFFFF).  Batches of
        // utterances that are all different (runs of 1) are packed densely, as before.
        auto pad_runs = [&](std::vector<uint32_t>::iterator first, std::vector<uint32_t>::iterator last, std::vector<uint32_t>& out) {
            size_t lanesOfRun = 0;     // lanes the current run occupies in the wavefront being filled
            for (auto it = first; it != last;) {
                auto runEnd = it;
                while (runEnd != last && timing[*runEnd] == timing[*it] && lens[*runEnd] == lens[*it]) ++runEnd;
                const size_t runSize = (size_t)(runEnd - it), fill = out.size() % kLanes;
                if (fill != 0 && (lanesOfRun >= (size_t)kLanes / 4 || runSize >= (size_t)kLanes))
                    out.insert(out.end(), kLanes - fill, 0xFFFFFFFFu);
                out.insert(out.end(), it, runEnd);
                const size_t tail = out.size() % kLanes;
                lanesOfRun = tail == 0 ? 0 : std::min(runSize, tail);
                it = runEnd;
            }
        };
        std::vector<uint32_t> tracked, direct(trackedEnd, directEnd), untracked;
        pad_runs(quietEnd, trackedEnd, tracked);
        if (!tracked.empty() && trackedEnd != order.end()) tracked.insert(tracked.end(), (kLanes - tracked.size() % kLanes) % kLanes, 0xFFFFFFFFu);   // the next group starts its own wavefront
        // (the direct stages do not care whether their lanes fade together: packed densely, longest first)
        if (!direct.empty() && directEnd != order.end()) direct.insert(direct.end(), (kLanes - direct.size() % kLanes) % kLanes, 0xFFFFFFFFu);
        pad_runs(directEnd, order.end(), untracked);
        nTracked = (long long)tracked.size();
        nDirectSlots = (long long)direct.size();
        order.resize((size_t)nQuiet);
        order.insert(order.end(), tracked.begin(), tracked.end());
        order.insert(order.end(), direct.begin(), direct.end());
        order.insert(order.end(), untracked.begin(), untracked.end());
    } else if (nDirectSlots > 0 && directEnd != order.end()) {
        // unsorted: the groups still start on wavefront boundaries (each is a launch of its own)
        std::vector<uint32_t> head(order.begin(), directEnd), tail(directEnd, order.end());
        const size_t padT = nTrackedUtt > 0 ? (size_t)((kLanes - nTrackedUtt % kLanes) % kLanes) : 0;
        head.insert(head.begin() + nQuiet + nTrackedUtt, padT, 0xFFFFFFFFu);
        nTracked = nTrackedUtt + (long long)padT;
        const size_t padD = (size_t)((kLanes - nDirectSlots % kLanes) % kLanes);
        head.insert(head.end(), padD, 0xFFFFFFFFu);
        nDirectSlots += (long long)padD;
        order.swap(head);
        order.insert(order.end(), tail.begin(), tail.end());
    } else if (nDirectSlots > 0 && nTrackedUtt > 0) {
        const size_t padT = (size_t)((kLanes - nTrackedUtt % kLanes) % kLanes);
        order.insert(order.begin() + nQuiet + nTrackedUtt, padT, 0xFFFFFFFFu);
        nTracked = nTrackedUtt + (long long)padT;
    }
    // A wavefront with FEW live lanes takes up to 1.7 times as long as a full one for the same instructions (measured:
    // streams_synthesize, tools/lone_probe2.py: 1 .. 8 live lanes 4.3 / 6.1 / 7.4 ms from launch to launch, 16 or more a steady 4.13).  Its
    // empty slots are given its own utterances again: those lanes compute the same samples and store the same bytes to the same places.
    // (What a batch of a handful of sentences -- or the tail of a large one -- costs in latency; nothing for full wavefronts.)
    constexpr int kSparse = 32;      // (16 live lanes still wavered a little: 4.32 against 4.15 ms)
    long long nQuietSlots = nQuiet, nNoNasalSlots = nNoNasal;
    {
        // the two quiet groups are packed densely: only their LAST wavefront can be sparse (one vowel alone: 46 ns per sample against 31)
        auto fill_tail = [&](long long begin, long long end) -> long long {
            const long long n = end - begin, tail = n % kLanes;
            if (n == 0 || tail == 0 || tail >= kSparse) return 0;
            const long long ext = kLanes - tail;
            std::vector<uint32_t> rep((size_t)ext);
            for (long long j = 0; j < ext; ++j) rep[(size_t)j] = order[(size_t)(end - tail + j % tail)];
            order.insert(order.begin() + end, rep.begin(), rep.end());
            return ext;
        };
        const long long e1 = fill_tail(0, nNoNasal);
        nNoNasalSlots += e1; nQuietSlots += e1;
        nQuietSlots += fill_tail(nNoNasalSlots, nQuietSlots);
        const long long noisy0 = nQuietSlots;
        const bool tailNoisy = (long long)order.size() > noisy0;
        if (tailNoisy && ((long long)order.size() - noisy0) % kLanes != 0) {
            // the last wavefront's dead lanes become slots of the group that ends there
            const long long ext = kLanes - ((long long)order.size() - noisy0) % kLanes;
            const long long legacy = (long long)order.size() - nQuietSlots - nTracked - nDirectSlots;
            if (legacy <= 0) { if (nDirectSlots > 0) nDirectSlots += ext; else nTracked += ext; }
            order.insert(order.end(), (size_t)ext, 0xFFFFFFFFu);
        }
        for (long long w = noisy0; w + kLanes <= (long long)order.size(); w += kLanes) {
            uint32_t liveU[kSparse];
            int nLive = 0;
            for (int i = 0; i < kLanes && nLive < kSparse; ++i)
                if (order[(size_t)(w + i)] != 0xFFFFFFFFu) liveU[nLive++] = order[(size_t)(w + i)];
            if (nLive == 0 || nLive >= kSparse) continue;
            int next = 0;
            for (int i = 0; i < kLanes; ++i)
                if (order[(size_t)(w + i)] == 0xFFFFFFFFu) { order[(size_t)(w + i)] = liveU[next]; next = (next + 1) % nLive; }
        }
    }
    const long long nSlotsAll = (long long)order.size();
    lap("lane packing");

    // Frames the planner recognised by their HASH are compared with the frame that first carried it, where the frames are (records
    // need none of this: a shape number IS the values).  The verdict comes back with the upload's last synchronisation.
    const bool verify = !byRecords && nTrackedUtt > 0 && !plan.rep.empty();
    unsigned long long mismatchAt = ~0ull;

    auto upload = [&]() -> int {
        if (b->dFrames.reserve(std::max<size_t>((size_t)nF * kNumParams, 1)) || b->dMeta.reserve(std::max<size_t>(nF, 1)) ||
            b->dUtt.reserve(std::max<size_t>(nU, 1)) || b->dOrder.reserve(std::max<size_t>(nSlotsAll, 1)) ||
            b->dResult.reserve(std::max<size_t>(nU, 1)) || b->dPcm.reserve(std::max<size_t>(pool, 1)))
            return -1;
        if (nTrackedUtt > 0) {
            if (b->dFlatRef.reserve((size_t)nF) || b->dJobs.reserve(jobs.size()) || b->dShapes.reserve(plan.shapes.size()) ||
                b->dTrack.reserve((size_t)trackEntries + kTrackPad)) return -1;
            HIP_TRY(hipMemcpyAsync(b->dShapes.ptr, plan.shapes.data(), plan.shapes.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->dFlatRef.ptr, flatRef.data(), (size_t)nF * sizeof(FlatRef), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->dJobs.ptr, jobs.data(), jobs.size() * sizeof(TrackJob), hipMemcpyHostToDevice, b->stream));
        }
        if (nTrackedUtt > 0 || nDirectUtt > 0) {
            if (b->dSourceRef.reserve((size_t)nF)) return -1;
        }
        if (nDirectUtt > 0) {
            const size_t nD = directJobs.size();
            if (b->dDirectJobs.reserve(nD) || b->dDirectFirst.reserve((size_t)nU) || b->dDirectHdr.reserve((size_t)kDirectStages * nD) ||
                b->dDirectRec.reserve((size_t)kDirectEntries * nD)) return -1;
            HIP_TRY(hipMemcpyAsync(b->dDirectJobs.ptr, directJobs.data(), nD * sizeof(DirectJob), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->dDirectFirst.ptr, directFirst.data(), (size_t)nU * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
        }
        if (nF) {
            if (!earlyStarted) HIP_TRY(hipMemcpyAsync(b->dFrames.ptr, in.frames, (size_t)nF * kNumParams * sizeof(double), hipMemcpyHostToDevice, b->stream));
            if (!byRecords) HIP_TRY(hipMemcpyAsync(b->dMeta.ptr, meta.data(), (size_t)nF * sizeof(FrameMeta), hipMemcpyHostToDevice, b->stream));
        }
        if (nU) {
            HIP_TRY(hipMemcpyAsync(b->dUtt.ptr, utt.data(), (size_t)nU * sizeof(UttDesc), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemcpyAsync(b->dOrder.ptr, order.data(), (size_t)nSlotsAll * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemsetAsync(b->dResult.ptr, 0, (size_t)nU * sizeof(UttResult), b->stream));
            if (b->dDenseStart.reserve((size_t)nU + 1)) return -1;
            HIP_TRY(hipMemcpyAsync(b->dDenseStart.ptr, denseStart.data(), ((size_t)nU + 1) * sizeof(long long), hipMemcpyHostToDevice, b->stream));
        }
        if (verify) {
            if (b->dRep.reserve((size_t)nF) || b->dMismatch.reserve(1)) return -1;
            HIP_TRY(hipMemcpyAsync(b->dRep.ptr, plan.rep.data(), (size_t)nF * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipMemsetAsync(b->dMismatch.ptr, 0xFF, sizeof(unsigned long long), b->stream));
        }
        HIP_TRY(hipStreamSynchronize(b->stream));
        if (copyGuard.armed) { HIP_TRY(hipStreamSynchronize(b->copyStream)); copyGuard.armed = false; }
        if (early.joinable()) early.join();
        if (earlyRc) { set_error_code(SPEECHPLAYER_ERR_HIP); set_error("set: uploading the frames failed: %s", earlyErr.c_str()); return -1; }
        const unsigned grid = (unsigned)std::min<long long>((nF + 255) / 256, 1 << 16);
        if (verify) {
            hipLaunchKernelGGL(klatt_verify_shared, dim3(grid), dim3(256), 0, b->stream, b->dFrames.ptr, b->dRep.ptr, nF, b->dMismatch.ptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&mismatchAt, b->dMismatch.ptr, sizeof mismatchAt, hipMemcpyDeviceToHost, b->stream));
        }
        if (nTrackedUtt > 0 || nDirectUtt > 0) {
            // what the flat and direct source stages read at a dequeue (SourceRef: pitch, pitch increment, 1 / fade, index mark), from the
            // frames and durations now resident: 32 bytes per frame that no longer cross the link, and no third pass of the host over the frames
            hipLaunchKernelGGL(klatt_source_refs, dim3(grid), dim3(256), 0, b->stream, b->dFrames.ptr, b->dMeta.ptr, b->dSourceRef.ptr, nF);
            HIP_TRY(hipGetLastError());
        }
        if (verify || nTrackedUtt > 0 || nDirectUtt > 0) HIP_TRY(hipStreamSynchronize(b->stream));
        return 0;
    };
    if (upload()) {
        // the device buffers may hold a mix of the old and the new batch now: the object becomes an empty batch
        batch_clear(b);
        return -1;
    }
    lap("uploads");
    if (verify && mismatchAt != ~0ull) {
        // two frames with one hash and different values: the plan built on their equality is void.  The batch is planned again
        // without tracks (nothing else rests on the hash) and the caller is told -- the call succeeds, the message stays.
        const unsigned long long k = mismatchAt;
        if (early.joinable()) early.join();
        SetInput again = in;
        again.noTracks = true;
        const uint32_t first = plan.rep[(size_t)k];
        const int rc = batch_set(b, again);
        if (rc == 0) {
            set_error("set: frames %llu and %u carry one 128-bit shape hash and different values; the batch runs without tracks", k, first);
            set_error_code(SPEECHPLAYER_OK);
        }
        return rc;
    }
    b->nUtt = nU; b->nFrames = nF; b->nFramesSpoken = spoken; b->nLists = nL; b->nSlots = nSlotsAll;
    b->nNoNasalUtt = nNoNasal;
    b->nQuiet = nQuietSlots; b->nNoNasal = nNoNasalSlots;      // (slots: the groups' utterances and the replicas that complete their sparse last wavefronts)
    b->nTracked = nTrackedUtt > 0 ? nTracked : 0; b->nTrackedUtt = nTrackedUtt;
    b->nJobs = nTrackedUtt > 0 ? (long long)jobs.size() : 0; b->trackEntries = nTrackedUtt > 0 ? (long long)trackEntries : 0;
    b->nDirect = nDirectUtt > 0 ? nDirectSlots : 0; b->nDirectUtt = nDirectUtt; b->nDirectFrames = (long long)directJobs.size();
    b->totalSamples = total; b->poolSamples = pool;
    b->lens.swap(lens); b->outStart.swap(outStart); b->denseStart.swap(denseStart);
    b->uttFrameStart.swap(uttFrameStart); b->uttFrames.swap(uttFrames);
    b->results.clear();
    b->resultsFresh = false;
    b->floatFresh = false;
    b->launched = false;
    return 0;
}

long long speechPlayer_batch_frames(speechPlayer_batch_t batch, long long u, speechPlayer_frame_t* frames, unsigned int* minFrameDuration,
                                    unsigned int* fadeDuration, int* userIndex, unsigned char* isNull, long long capacity)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b || u < 0 || u >= b->nUtt) { set_error("batch_frames: no such utterance"); return -1; }
    HIP_TRY(hipSetDevice(b->device));
    const long long n = b->uttFrames[(size_t)u], k0 = b->uttFrameStart[(size_t)u];
    if (n > capacity || n == 0) return n;
    if (frames) HIP_TRY(hipMemcpy(frames, b->dFrames.ptr + k0 * kNumParams, (size_t)n * sizeof(speechPlayer_frame_t), hipMemcpyDeviceToHost));
    if (minFrameDuration || fadeDuration || userIndex || isNull) {
        std::vector<FrameMeta> m((size_t)n);
        HIP_TRY(hipMemcpy(m.data(), b->dMeta.ptr + k0, (size_t)n * sizeof(FrameMeta), hipMemcpyDeviceToHost));
        for (long long k = 0; k < n; ++k) {
            if (minFrameDuration) minFrameDuration[k] = m[(size_t)k].minSamples;
            if (fadeDuration) fadeDuration[k] = m[(size_t)k].fadeSamples;
            if (userIndex) userIndex[k] = m[(size_t)k].userIndex;
            if (isNull) isNull[k] = (m[(size_t)k].flags & FRAME_NULL) ? 1 : 0;
        }
    }
    return n;
}

long long speechPlayer_batch_utteranceSamples(speechPlayer_batch_t batch, long long u)
{
    Batch* b = static_cast<Batch*>(batch);
    if (!b || u < 0 || u >= b->nUtt) return -1;
    return b->lens[u];
}
long long speechPlayer_batch_totalSamples(speechPlayer_batch_t batch) { return batch ? static_cast<Batch*>(batch)->totalSamples : -1; }
long long speechPlayer_batch_totalFrames(speechPlayer_batch_t batch) { return batch ? static_cast<Batch*>(batch)->nFramesSpoken : -1; }
int speechPlayer_batch_sampleRate(speechPlayer_batch_t batch) { return batch ? static_cast<Batch*>(batch)->sampleRate : -1; }

int speechPlayer_batch_synthesize(speechPlayer_batch_t batch)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b) return -1;
    HIP_TRY(hipSetDevice(b->device));
    return batch_launch(b);
}

int speechPlayer_batch_wait(speechPlayer_batch_t batch)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b) return -1;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
}

static int fetch_results(Batch* b)
{
    if (b->resultsFresh) return 0;
    b->results.resize((size_t)b->nUtt);
    if (b->nUtt) HIP_TRY(hipMemcpy(b->results.data(), b->dResult.ptr, (size_t)b->nUtt * sizeof(UttResult), hipMemcpyDeviceToHost));
    b->resultsFresh = true;
    return 0;
}

long long speechPlayer_batch_read(speechPlayer_batch_t batch, long long u, sample* sampleBuf, long long capacity)
{
    begin_call();
    if (refuse_timing_only("speechPlayer_batch_read")) return -1;
    Batch* b = static_cast<Batch*>(batch);
    if (!b || u < 0 || u >= b->nUtt || !sampleBuf) return -1;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (fetch_results(b)) return -1;
    long long n = std::min<long long>(b->results[u].produced, capacity);
    if (n > 0) HIP_TRY(hipMemcpy(sampleBuf, b->dPcm.ptr + b->outStart[u], (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost));
    return n;
}

// The utterances back to back in HBM (pcm_compact), queued on the batch's stream behind the synthesis; `denseReady` marks the end.
static int dense_prepare(Batch* b)
{
    const long long total = b->totalSamples;
    if (b->copyPending) { HIP_TRY(hipEventSynchronize(b->copyDone)); b->copyPending = false; }      // the previous copy still reads dDense (before it may be re-allocated)
    if (b->dDense.reserve((size_t)((total + 7) / 8 * 8 + 8))) return -1;
    const long long n8 = (total + 7) / 8;
    if (n8 > 0) {
        const unsigned grid = (unsigned)std::min<long long>((n8 + 255) / 256, 1 << 16);
        hipLaunchKernelGGL(pcm_compact, dim3(grid), dim3(256), 0, b->stream, b->dPcm.ptr, b->dDense.ptr, b->dUtt.ptr, b->dDenseStart.ptr, b->nUtt, total);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(b->denseReady, b->stream));
    return 0;
}

// Before round 5 (and still, for a batch that has not been synthesised to the end: produced != length somewhere): the padded pool comes
// over in 16 MB pieces through two pinned buffers and one host thread compacts them into the caller's buffer.
static long long read_all_padded(Batch* b, sample* sampleBuf, long long capacity, long long* outStart)
{
    std::vector<long long> dstStart((size_t)b->nUtt + 1);
    long long pos = 0;
    for (long long u = 0; u < b->nUtt; ++u) { dstStart[u] = pos; pos += b->results[u].produced; }
    dstStart[b->nUtt] = pos;
    if (pos > capacity) { set_error("readAll: capacity %lld too small for %lld samples", capacity, pos); return -1; }
    if (outStart) memcpy(outStart, dstStart.data(), sizeof(long long) * ((size_t)b->nUtt + 1));
    constexpr long long kPiece = 8ll << 20;   // samples
    if (b->bounce.ensure((size_t)kPiece * sizeof(int16_t))) return -1;
    const long long nPieces = (b->poolSamples + kPiece - 1) / kPiece;
    auto issue = [&](long long k) -> int {
        const long long off = k * kPiece, n = std::min(kPiece, b->poolSamples - off);
        HIP_TRY(hipMemcpyAsync(b->bounce.buf[k & 1], b->dPcm.ptr + off, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipEventRecord(b->bounce.ev[k & 1], b->stream));
        return 0;
    };
    if (nPieces > 0 && issue(0)) return -1;
    int16_t* const dst = reinterpret_cast<int16_t*>(sampleBuf);
    long long u = 0;
    for (long long k = 0; k < nPieces; ++k) {
        if (k + 1 < nPieces && issue(k + 1)) return -1;
        HIP_TRY(hipEventSynchronize(b->bounce.ev[k & 1]));
        const long long c0 = k * kPiece, c1 = std::min(c0 + kPiece, b->poolSamples);
        const int16_t* const src = static_cast<const int16_t*>(b->bounce.buf[k & 1]);
        while (u < b->nUtt) {
            const long long s0 = b->outStart[u], e0 = s0 + b->results[u].produced;
            if (s0 >= c1) break;
            const long long lo = std::max(s0, c0), hi = std::min(e0, c1);
            if (hi > lo) memcpy(dst + dstStart[u] + (lo - s0), src + (lo - c0), (size_t)(hi - lo) * sizeof(int16_t));
            if (e0 > c1) break;       // continues in the next piece
            ++u;
        }
    }
    return pos;
}

long long speechPlayer_batch_readAll(speechPlayer_batch_t batch, sample* sampleBuf, long long capacity, long long* outStart)
{
    begin_call();
    if (refuse_timing_only("speechPlayer_batch_readAll")) return -1;
    Batch* b = static_cast<Batch*>(batch);
    if (!b || !sampleBuf) return -1;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (fetch_results(b)) return -1;
    bool whole = true;      // every utterance synthesised to its end (a batch launch always does; a batch never launched has produced nothing)
    for (long long u = 0; u < b->nUtt && whole; ++u) whole = b->results[u].produced == b->lens[u];
    if (!whole) return read_all_padded(b, sampleBuf, capacity, outStart);
    const long long total = b->totalSamples;
    if (total > capacity) { set_error("readAll: capacity %lld too small for %lld samples", capacity, total); return -1; }
    if (outStart) memcpy(outStart, b->denseStart.data(), sizeof(long long) * ((size_t)b->nUtt + 1));
    if (total == 0) return 0;
    // dense order on the device first; then the bytes cross the link as they will lie in the caller's buffer
    if (dense_prepare(b)) return -1;
    HIP_TRY(hipStreamWaitEvent(b->copyStream, b->denseReady, 0));
    if (is_pinned(sampleBuf)) {
        HIP_TRY(hipMemcpyAsync(sampleBuf, b->dDense.ptr, (size_t)total * sizeof(int16_t), hipMemcpyDeviceToHost, b->copyStream));
        HIP_TRY(hipStreamSynchronize(b->copyStream));
        return total;
    }
    // pageable destination: 32 MB pieces through two pinned buffers, piece k + 1 in flight while piece k is copied out (a plain memcpy now)
    constexpr long long kPiece = 16ll << 20;   // samples
    if (b->bounce.ensure((size_t)kPiece * sizeof(int16_t))) return -1;
    const long long nPieces = (total + kPiece - 1) / kPiece;
    auto issue = [&](long long k) -> int {
        const long long off = k * kPiece, n = std::min(kPiece, total - off);
        HIP_TRY(hipMemcpyAsync(b->bounce.buf[k & 1], b->dDense.ptr + off, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost, b->copyStream));
        HIP_TRY(hipEventRecord(b->bounce.ev[k & 1], b->copyStream));
        return 0;
    };
    if (issue(0)) return -1;
    int16_t* const dst = reinterpret_cast<int16_t*>(sampleBuf);
    for (long long k = 0; k < nPieces; ++k) {
        HIP_TRY(hipEventSynchronize(b->bounce.ev[k & 1]));
        const long long off = k * kPiece, n = std::min(kPiece, total - off);
        // (the other buffer's copy is issued only after this one has been emptied two pieces ago: buffers alternate)
        if (k + 1 < nPieces && issue(k + 1)) return -1;
        const int16_t* const src = static_cast<const int16_t*>(b->bounce.buf[k & 1]);
        parallel_ranges(n, 1 << 21, [&](long long a, long long e) { memcpy(dst + off + a, src + a, (size_t)(e - a) * sizeof(int16_t)); });
    }
    return total;
}

// readAll without waiting: the compaction and ONE copy into page-locked memory (speechPlayer_hostAlloc, or registered by the caller) are
// queued behind the synthesis and the call returns; the copy runs on a stream of its own, beside whatever is launched next (another
// batch's synthesis, this batch's next launch: the pool is free again as soon as the compaction has run).  speechPlayer_batch_readWait
// waits for it.  Returns the number of samples that will arrive, -1 when the buffer is not page-locked or too small.
long long speechPlayer_batch_readAllAsync(speechPlayer_batch_t batch, sample* sampleBuf, long long capacity, long long* outStart)
{
    begin_call();
    if (refuse_timing_only("speechPlayer_batch_readAllAsync")) return -1;
    Batch* b = static_cast<Batch*>(batch);
    if (!b || !sampleBuf) return -1;
    HIP_TRY(hipSetDevice(b->device));
    if (!is_pinned(sampleBuf)) { set_error("readAllAsync: the buffer is not page-locked (speechPlayer_hostAlloc)"); return -1; }
    // (readAll checks what every utterance produced and falls back to the padded copy; this call cannot wait for that -- it needs a launch
    // queued since the batch was set: a batch launch always synthesises every utterance to its end.  ADVICE r5)
    if (!b->launched && b->nUtt > 0) { set_error("readAllAsync: the batch has not been synthesised since it was set"); return -1; }
    const long long total = b->totalSamples;
    if (total > capacity) { set_error("readAllAsync: capacity %lld too small for %lld samples", capacity, total); return -1; }
    if (outStart) memcpy(outStart, b->denseStart.data(), sizeof(long long) * ((size_t)b->nUtt + 1));
    if (total == 0) return 0;
    if (dense_prepare(b)) return -1;
    HIP_TRY(hipStreamWaitEvent(b->copyStream, b->denseReady, 0));
    HIP_TRY(hipMemcpyAsync(sampleBuf, b->dDense.ptr, (size_t)total * sizeof(int16_t), hipMemcpyDeviceToHost, b->copyStream));
    HIP_TRY(hipEventRecord(b->copyDone, b->copyStream));
    b->copyPending = true;
    return total;
}

int speechPlayer_batch_readWait(speechPlayer_batch_t batch)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b) return -1;
    HIP_TRY(hipSetDevice(b->device));
    if (b->copyPending) { HIP_TRY(hipEventSynchronize(b->copyDone)); b->copyPending = false; }
    return 0;
}

// Page-locked host memory for the two large transfers of a batch -- frames in (speechPlayer_batch_setUtterances), PCM out
// (speechPlayer_batch_readAll / readAllAsync): a buffer from here travels as one DMA at the link's rate.  NULL when it cannot be had.
void* speechPlayer_hostAlloc(long long bytes)
{
    begin_call();
    if (bytes <= 0) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error_code(SPEECHPLAYER_ERR_HIP);
        set_error("hostAlloc: cannot page-lock %lld bytes", bytes);
        return nullptr;
    }
    return p;
}
void speechPlayer_hostFree(void* p)
{
    if (p) (void)hipHostFree(p);
}

long long speechPlayer_batch_readFloat(speechPlayer_batch_t batch, long long u, float* sampleBuf, long long capacity)
{
    begin_call();
    if (refuse_timing_only("speechPlayer_batch_readFloat")) return -1;
    Batch* b = static_cast<Batch*>(batch);
    if (!b || u < 0 || u >= b->nUtt || !sampleBuf) return -1;
    HIP_TRY(hipSetDevice(b->device));
    if (!b->floatFresh) {
        if (b->dFloat.reserve(std::max<size_t>(b->poolSamples, 8))) return -1;
        const long long n8 = b->poolSamples / 8;     // the pool is a multiple of 32 samples
        if (n8 > 0) {
            const unsigned grid = (unsigned)std::min<long long>((n8 + 255) / 256, 2048);
            hipLaunchKernelGGL(pcm_to_float, dim3(grid), dim3(256), 0, b->stream, b->dPcm.ptr, b->dFloat.ptr, n8);
            HIP_TRY(hipGetLastError());
        }
        b->floatFresh = true;
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (fetch_results(b)) return -1;
    long long n = std::min<long long>(b->results[u].produced, capacity);
    if (n > 0) HIP_TRY(hipMemcpy(sampleBuf, b->dFloat.ptr + b->outStart[u], (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
}

int speechPlayer_batch_digest(speechPlayer_batch_t batch, unsigned long long* perUtterance, unsigned long long* whole)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b) { set_error("digest: no batch"); return -1; }
    HIP_TRY(hipSetDevice(b->device));
    std::vector<unsigned long long> host((size_t)b->nUtt);
    if (b->nUtt) {
        if (b->dDigest.reserve((size_t)b->nUtt)) return -1;
        const unsigned grid = (unsigned)std::min<long long>((b->nUtt + 3) / 4, 16384);
        hipLaunchKernelGGL(pcm_digest, dim3(grid), dim3(256), 0, b->stream, b->dPcm.ptr, b->dUtt.ptr, b->dResult.ptr, b->dDigest.ptr, b->nUtt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host.data(), b->dDigest.ptr, (size_t)b->nUtt * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    if (perUtterance && b->nUtt) memcpy(perUtterance, host.data(), (size_t)b->nUtt * sizeof(unsigned long long));
    if (whole) {
        unsigned long long acc = 0;
        for (long long u = 0; u < b->nUtt; ++u) acc += digest_mix((unsigned long long)u, 0u) ^ host[u] * 0x9E3779B97F4A7C15ull;
        *whole = acc;
    }
    return 0;
}

int speechPlayer_batch_getLastIndex(speechPlayer_batch_t batch, long long u)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b || u < 0 || u >= b->nUtt) return -1;
    if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) return -1;
    if (fetch_results(b)) return -1;
    return b->results[u].lastIndex;
}

const sample* speechPlayer_batch_devicePcm(speechPlayer_batch_t batch)
{
    Batch* b = static_cast<Batch*>(batch);
    return b ? reinterpret_cast<const sample*>(b->dPcm.ptr) : nullptr;
}

long long speechPlayer_batch_deviceOffset(speechPlayer_batch_t batch, long long u)
{
    Batch* b = static_cast<Batch*>(batch);
    if (!b || u < 0 || u > b->nUtt) return -1;
    return b->outStart[u];
}

int speechPlayer_batch_time(speechPlayer_batch_t batch, int launches, float* msPerLaunch)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b || launches <= 0 || !msPerLaunch) return -1;
    HIP_TRY(hipSetDevice(b->device));
    struct Events {
        std::vector<hipEvent_t> v;
        ~Events() { for (auto e : v) if (e) (void)hipEventDestroy(e); }
    } events;
    events.v.assign((size_t)launches * 2, nullptr);
    std::vector<hipEvent_t>& ev = events.v;
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    int rc = 0;
    for (int i = 0; i < launches && !rc; ++i) {
        HIP_TRY(hipEventRecord(ev[2 * i], b->stream));
        rc = batch_launch(b);
        HIP_TRY(hipEventRecord(ev[2 * i + 1], b->stream));
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (int i = 0; i < launches && !rc; ++i) HIP_TRY(hipEventElapsedTime(&msPerLaunch[i], ev[2 * i], ev[2 * i + 1]));
    return rc;
}

// ==========================================================================================
// C-ABI: one batch over several devices of a node
// ==========================================================================================
// Utterances are independent, so a node's batch is cut into contiguous shards with near-equal total SAMPLE counts
// (closed-form lengths; what balances the kernels is samples, not utterance counts), one shard per device, each shard a
// Batch of its own with its own stream.  One host thread per device uploads its shard; launches are asynchronous, so
// every device synthesises at the same time; nothing is exchanged between devices (no collective).
struct Node {
    int sampleRate = 0;
    std::vector<Batch*> parts;
    std::vector<long long> bounds;       // contiguous deal: shard d = utterances bounds[d] .. bounds[d + 1] - 1
    int deal = 0;                        // option "deal": 0 contiguous shards of near-equal sample count, 1 the sorted deal
    bool sorted = false;                 // the batch that is set was dealt sorted: shard d = members[d]
    std::vector<std::vector<long long>> members;   // sorted deal: the utterances of every shard, in the shard's own order
    std::vector<int> shardOf;            // sorted deal: [nUtterances]
    std::vector<long long> localOf;      // ... and the utterance's number within its shard
    long long nUtt = 0;
};

speechPlayer_node_t speechPlayer_node_create(int sampleRate, int nDevices, const int* devices)
{
    begin_call();
    if (nDevices <= 0) { set_error("node_create: need at least one device"); return nullptr; }
    Node* n = new Node;
    n->sampleRate = sampleRate;
    for (int d = 0; d < nDevices; ++d) {
        Batch* b = static_cast<Batch*>(speechPlayer_batch_create(sampleRate, devices ? devices[d] : d));
        if (!b) { speechPlayer_node_destroy(n); return nullptr; }
        n->parts.push_back(b);
    }
    n->bounds.assign((size_t)nDevices + 1, 0);
    return n;
}

void speechPlayer_node_destroy(speechPlayer_node_t node)
{
    Node* n = static_cast<Node*>(node);
    if (!n) return;
    for (Batch* b : n->parts) speechPlayer_batch_destroy(b);
    delete n;
}

int speechPlayer_node_devices(speechPlayer_node_t node) { return node ? (int)static_cast<Node*>(node)->parts.size() : -1; }

speechPlayer_batch_t speechPlayer_node_part(speechPlayer_node_t node, int shard)
{
    Node* n = static_cast<Node*>(node);
    return (n && shard >= 0 && shard < (int)n->parts.size()) ? n->parts[shard] : nullptr;
}

int speechPlayer_node_setOption(speechPlayer_node_t node, const char* name, int value)
{
    begin_call();
    Node* n = static_cast<Node*>(node);
    if (!n || !name) return -1;
    // "deal" (read by speechPlayer_node_setUtterances): 0 contiguous shards of near-equal total sample count (default), 1 the SORTED deal
    // of SURVEY 8(e): utterances sorted by length, blocks of 64 (one wavefront) dealt round-robin, so that every device sees the same
    // length distribution -- a batch whose long utterances cluster gives every device its share of them.  Results are addressed by the
    // batch's own utterance numbers either way.
    if (!strcmp(name, "deal")) { n->deal = value ? 1 : 0; return 0; }
    for (Batch* b : n->parts)
        if (speechPlayer_batch_setOption(b, name, value)) return -1;
    return 0;
}

// The deal of a node's batch: before[u] = samples before utterance u (closed-form lengths).  Contiguous shards of near-equal
// sample count, or -- option "deal" -- the sorted deal: longest first (ties in the batch's order), blocks of 64 dealt round-robin
// (nvspeechplayer_amd.sharding.shard_deal states the same rules).
static void node_deal(Node* n, long long nUtterances, const std::vector<unsigned long long>& before)
{
    const int nd = (int)n->parts.size();
    const unsigned long long total = before[nUtterances];
    n->nUtt = nUtterances;
    n->sorted = n->deal == 1 && nd > 1;
    n->members.clear(); n->shardOf.clear(); n->localOf.clear();
    n->bounds[0] = 0;
    for (int d = 1; d < nd; ++d) {
        // first utterance whose start is at or beyond d/nd of the samples (the rule of nvspeechplayer_amd.sharding.shard_bounds)
        const double target = (double)total * d / (double)nd;
        long long cut = std::lower_bound(before.begin(), before.end(), target, [](unsigned long long x, double t) { return (double)x < t; }) - before.begin();
        n->bounds[d] = std::max(n->bounds[d - 1], std::min(cut, nUtterances));
    }
    n->bounds[nd] = nUtterances;
    if (n->sorted) {
        // longest first (ties in the batch's order), blocks of 64 dealt round-robin (nvspeechplayer_amd.sharding.shard_deal states the same rule)
        std::vector<long long> order((size_t)nUtterances);
        std::iota(order.begin(), order.end(), 0ll);
        std::stable_sort(order.begin(), order.end(), [&](long long x, long long y) { return before[x + 1] - before[x] > before[y + 1] - before[y]; });
        n->members.assign((size_t)nd, {});
        n->shardOf.assign((size_t)nUtterances, 0); n->localOf.assign((size_t)nUtterances, 0);
        for (long long i = 0; i < nUtterances; ++i) {
            const int d = (int)((i / kLanes) % nd);
            n->shardOf[(size_t)order[i]] = d;
            n->localOf[(size_t)order[i]] = (long long)n->members[(size_t)d].size();
            n->members[(size_t)d].push_back(order[i]);
        }
    }
}

int speechPlayer_node_setUtterances(speechPlayer_node_t node, long long nUtterances, const long long* frameStart,
                                    const speechPlayer_frame_t* frames, const unsigned int* minFrameDuration,
                                    const unsigned int* fadeDuration, const int* userIndex, const unsigned char* isNull,
                                    const unsigned int* noiseSeed)
{
    begin_call();
    Node* n = static_cast<Node*>(node);
    if (!n || nUtterances < 0 || !frameStart) { set_error("node_setUtterances: bad arguments"); return -1; }
    if (frameStart[0] != 0) { set_error("node_setUtterances: frameStart[0] must be 0"); return -1; }
    for (long long u = 0; u < nUtterances; ++u)
        if (frameStart[u + 1] < frameStart[u]) { set_error("node_setUtterances: frameStart not monotone at %lld", u); return -1; }
    if (frameStart[nUtterances] > 0 && (!minFrameDuration || !fadeDuration)) { set_error("node_setUtterances: bad frame arrays"); return -1; }
    // samples before each utterance (closed form: a request spans max(M, F + 1) + 1 samples, F clamped to >= 1)
    std::vector<unsigned long long> before((size_t)nUtterances + 1, 0);
    for (long long u = 0; u < nUtterances; ++u) {
        unsigned long long len = 0;
        for (long long k = frameStart[u]; k < frameStart[u + 1]; ++k) {
            const unsigned long long m = minFrameDuration[k], f = std::max(fadeDuration[k], 1u);
            len += std::max(m, f + 1) + 1;
        }
        before[u + 1] = before[u] + len;
    }
    node_deal(n, nUtterances, before);
    const int nd = (int)n->parts.size();
    // one host thread per device: rebase the shard's index array, give every utterance its GLOBAL default seed, upload
    std::vector<int> rc((size_t)nd, 0), codes((size_t)nd, 0);
    std::vector<std::string> errors((size_t)nd);
    // (run_parts: a thread that cannot be started leaves its shard to the caller's thread; nothing may throw across the C ABI)
    try {
    run_parts((unsigned)nd, [&](unsigned ud) {
        const int d = (int)ud;
        try {
            if (n->sorted) {
                // the shard's utterances are scattered over the batch: their frames are gathered (one copy on the host, 392 bytes per frame)
                const std::vector<long long>& mem = n->members[(size_t)d];
                std::vector<long long> fs(mem.size() + 1, 0);
                for (size_t j = 0; j < mem.size(); ++j) fs[j + 1] = fs[j] + (frameStart[mem[j] + 1] - frameStart[mem[j]]);
                const size_t nf = (size_t)fs[mem.size()];
                std::vector<speechPlayer_frame_t> fr(frames ? nf : 0);
                std::vector<unsigned int> mins(nf), fades(nf), seeds(mem.size());
                std::vector<int> idx(userIndex ? nf : 0);
                std::vector<unsigned char> nul(isNull ? nf : 0);
                for (size_t j = 0; j < mem.size(); ++j) {
                    const long long a = frameStart[mem[j]], cnt = frameStart[mem[j] + 1] - a, at = fs[j];
                    if (cnt > 0) {
                        if (frames) memcpy(&fr[(size_t)at], frames + a, (size_t)cnt * sizeof(speechPlayer_frame_t));
                        memcpy(&mins[(size_t)at], minFrameDuration + a, (size_t)cnt * sizeof(unsigned int));
                        memcpy(&fades[(size_t)at], fadeDuration + a, (size_t)cnt * sizeof(unsigned int));
                        if (userIndex) memcpy(&idx[(size_t)at], userIndex + a, (size_t)cnt * sizeof(int));
                        if (isNull) memcpy(&nul[(size_t)at], isNull + a, (size_t)cnt);
                    }
                    seeds[j] = noiseSeed ? noiseSeed[mem[j]] : (unsigned int)mem[j];
                }
                rc[d] = speechPlayer_batch_setUtterances(n->parts[d], (long long)mem.size(), fs.data(), frames ? fr.data() : nullptr, mins.data(), fades.data(),
                                                         userIndex ? idx.data() : nullptr, isNull ? nul.data() : nullptr, seeds.data());
                if (rc[d]) { errors[d] = g_lastError; codes[d] = g_lastErrorCode; }
                return;
            }
            const long long u0 = n->bounds[d], u1 = n->bounds[d + 1], f0 = frameStart[u0];
            std::vector<long long> fs((size_t)(u1 - u0) + 1);
            for (long long u = u0; u <= u1; ++u) fs[u - u0] = frameStart[u] - f0;
            std::vector<unsigned int> seeds((size_t)(u1 - u0));
            for (long long u = u0; u < u1; ++u) seeds[u - u0] = noiseSeed ? noiseSeed[u] : (unsigned int)u;
            rc[d] = speechPlayer_batch_setUtterances(n->parts[d], u1 - u0, fs.data(), frames ? frames + f0 : nullptr,
                                                     minFrameDuration ? minFrameDuration + f0 : nullptr, fadeDuration ? fadeDuration + f0 : nullptr,
                                                     userIndex ? userIndex + f0 : nullptr, isNull ? isNull + f0 : nullptr, seeds.data());
            if (rc[d]) { errors[d] = g_lastError; codes[d] = g_lastErrorCode; }
        } catch (const std::exception& e) {
            rc[d] = -1; errors[d] = e.what(); codes[d] = SPEECHPLAYER_ERR_ARGUMENT;
        }
    });
    } catch (const std::exception& e) {
        set_error("node_setUtterances: %s", e.what());
        return -1;
    }
    for (int d = 0; d < nd; ++d)
        if (rc[d]) { set_error_code(codes[d]); set_error("node_setUtterances: shard %d: %s", d, errors[d].c_str()); return -1; }
    return 0;
}

// The node's batch in compact form (speechPlayer_batch_setRecords): the lists, their records and the shape table go to EVERY shard (they
// are small), each shard gets the utterances the deal gives it -- their list numbers and noise seeds.
int speechPlayer_node_setRecords(speechPlayer_node_t node, long long nShapes, const speechPlayer_frame_t* shapes, long long nLists,
                                 const long long* listStart, const speechPlayer_frameRecord_t* records, long long nUtterances,
                                 const unsigned int* listOf, const unsigned int* noiseSeed)
{
    begin_call();
    Node* n = static_cast<Node*>(node);
    if (!n || nUtterances < 0 || nLists < 0 || !listStart) { set_error("node_setRecords: bad arguments"); return -1; }
    if (!listOf && nUtterances != nLists) { set_error("node_setRecords: %lld utterances for %lld lists and no listOf", nUtterances, nLists); return -1; }
    if (listStart[0] != 0) { set_error("node_setRecords: listStart[0] must be 0"); return -1; }
    for (long long l = 0; l < nLists; ++l)
        if (listStart[l + 1] < listStart[l]) { set_error("node_setRecords: listStart not monotone at %lld", l); return -1; }
    if (listStart[nLists] > 0 && !records) { set_error("node_setRecords: no records"); return -1; }
    for (long long u = 0; listOf && u < nUtterances; ++u)
        if ((long long)listOf[u] >= nLists) { set_error("node_setRecords: listOf[%lld] is not a list", u); return -1; }
    try {
        std::vector<unsigned long long> lenL((size_t)nLists, 0), before((size_t)nUtterances + 1, 0);
        for (long long l = 0; l < nLists; ++l)
            for (long long k = listStart[l]; k < listStart[l + 1]; ++k) {
                const unsigned long long m = records[k].minFrameDuration, f = std::max(records[k].fadeDuration, 1u);
                lenL[(size_t)l] += std::max(m, f + 1) + 1;
            }
        for (long long u = 0; u < nUtterances; ++u) before[(size_t)u + 1] = before[(size_t)u] + lenL[listOf ? listOf[u] : (size_t)u];
        node_deal(n, nUtterances, before);
        const int nd = (int)n->parts.size();
        std::vector<int> rc((size_t)nd, 0), codes((size_t)nd, 0);
        std::vector<std::string> errors((size_t)nd);
        run_parts((unsigned)nd, [&](unsigned ud) {
            const int d = (int)ud;
            try {
                const long long cnt = n->sorted ? (long long)n->members[(size_t)d].size() : n->bounds[d + 1] - n->bounds[d];
                std::vector<unsigned int> lo((size_t)cnt), seeds((size_t)cnt);
                for (long long j = 0; j < cnt; ++j) {
                    const long long u = n->sorted ? n->members[(size_t)d][(size_t)j] : n->bounds[d] + j;
                    lo[(size_t)j] = listOf ? listOf[u] : (unsigned int)u;
                    seeds[(size_t)j] = noiseSeed ? noiseSeed[u] : (unsigned int)u;
                }
                rc[d] = speechPlayer_batch_setRecords(n->parts[d], nShapes, shapes, nLists, listStart, records, cnt, lo.data(), seeds.data());
                if (rc[d]) { errors[d] = g_lastError; codes[d] = g_lastErrorCode; }
            } catch (const std::exception& e) {
                rc[d] = -1; errors[d] = e.what(); codes[d] = SPEECHPLAYER_ERR_ARGUMENT;
            }
        });
        for (int d = 0; d < nd; ++d)
            if (rc[d]) { set_error_code(codes[d]); set_error("node_setRecords: shard %d: %s", d, errors[d].c_str()); return -1; }
    } catch (const std::exception& e) {
        set_error("node_setRecords: %s", e.what());
        return -1;
    }
    return 0;
}

int speechPlayer_node_synthesize(speechPlayer_node_t node)
{
    begin_call();
    Node* n = static_cast<Node*>(node);
    if (!n) return -1;
    for (Batch* b : n->parts) {       // asynchronous launches: the devices run side by side
        HIP_TRY(hipSetDevice(b->device));
        if (batch_launch(b)) return -1;
    }
    return 0;
}

int speechPlayer_node_wait(speechPlayer_node_t node)
{
    begin_call();
    Node* n = static_cast<Node*>(node);
    if (!n) return -1;
    for (Batch* b : n->parts) {
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    return 0;
}

int speechPlayer_node_shardInfo(speechPlayer_node_t node, int shard, long long* firstUtterance, long long* nUtterances, long long* samples, int* device)
{
    Node* n = static_cast<Node*>(node);
    if (!n || shard < 0 || shard >= (int)n->parts.size()) return -1;
    if (firstUtterance) *firstUtterance = n->sorted ? -1 : n->bounds[shard];      // (the sorted deal's shards are not ranges: speechPlayer_node_shardUtterances)
    if (nUtterances) *nUtterances = n->sorted ? (long long)n->members[(size_t)shard].size() : n->bounds[shard + 1] - n->bounds[shard];
    if (samples) *samples = n->parts[shard]->totalSamples;
    if (device) *device = n->parts[shard]->device;
    return 0;
}

long long speechPlayer_node_totalSamples(speechPlayer_node_t node)
{
    Node* n = static_cast<Node*>(node);
    if (!n) return -1;
    long long t = 0;
    for (Batch* b : n->parts) t += b->totalSamples;
    return t;
}

// the shard that holds utterance u and its number there
static int node_locate(Node* n, long long u, long long* local)
{
    if (!n || u < 0 || u >= n->nUtt) return -1;
    if (n->sorted) { *local = n->localOf[(size_t)u]; return n->shardOf[(size_t)u]; }
    const int d = (int)(std::upper_bound(n->bounds.begin(), n->bounds.end(), u) - n->bounds.begin()) - 1;
    *local = u - n->bounds[d];
    return d;
}

long long speechPlayer_node_shardUtterances(speechPlayer_node_t node, int shard, long long* utterances, long long capacity)
{
    Node* n = static_cast<Node*>(node);
    if (!n || shard < 0 || shard >= (int)n->parts.size()) return -1;
    const long long cnt = n->sorted ? (long long)n->members[(size_t)shard].size() : n->bounds[shard + 1] - n->bounds[shard];
    if (utterances && cnt <= capacity)
        for (long long j = 0; j < cnt; ++j) utterances[j] = n->sorted ? n->members[(size_t)shard][(size_t)j] : n->bounds[shard] + j;
    return cnt;
}

long long speechPlayer_node_read(speechPlayer_node_t node, long long u, sample* sampleBuf, long long capacity)
{
    Node* n = static_cast<Node*>(node);
    long long local = 0;
    const int d = node_locate(n, u, &local);
    if (d < 0) { begin_call(); set_error("node_read: utterance %lld out of range", u); return -1; }
    return speechPlayer_batch_read(n->parts[d], local, sampleBuf, capacity);
}

int speechPlayer_node_getLastIndex(speechPlayer_node_t node, long long u)
{
    Node* n = static_cast<Node*>(node);
    long long local = 0;
    const int d = node_locate(n, u, &local);
    return d < 0 ? -1 : speechPlayer_batch_getLastIndex(n->parts[d], local);
}

// `launches` passes over the node's batch, every device launched before any is waited for; wall-clock milliseconds per pass
int speechPlayer_node_time(speechPlayer_node_t node, int launches, float* msPerLaunch)
{
    begin_call();
    Node* n = static_cast<Node*>(node);
    if (!n || launches <= 0 || !msPerLaunch) return -1;
    if (speechPlayer_node_wait(n)) return -1;
    for (int i = 0; i < launches; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        if (speechPlayer_node_synthesize(n) || speechPlayer_node_wait(n)) return -1;
        msPerLaunch[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// diagnostic builds (-DKLATT_STAMPS): per workgroup and stage {work cycles, barrier-wait cycles} of the last launch
int speechPlayer_batch_debugStamps(speechPlayer_batch_t batch, unsigned long long* out, int capacity)
{
    Batch* b = static_cast<Batch*>(batch);
    if (!b || !out) return -1;
    int n = (int)std::min<size_t>(b->dDebug.cap, (size_t)capacity);
    if (n > 0) HIP_TRY(hipMemcpy(out, b->dDebug.ptr, (size_t)n * 8, hipMemcpyDeviceToHost));
    return n;
}

int speechPlayer_batch_kernelInfo(speechPlayer_batch_t batch, int* info, int nInfo)
{
    begin_call();
    Batch* b = static_cast<Batch*>(batch);
    if (!b || !info || nInfo < 6) return -1;
    HIP_TRY(hipSetDevice(b->device));
    hipFuncAttributes fa;
    // report the kernel of the largest of the three groups (batch_launch)
    const long long nNoisy = b->nSlots - b->nQuiet, nLp = lanepipe_count(b);
    const long long nNn = b->layout == 0 ? 0 : b->nNoNasal - nLp, nQ = b->nQuiet - nLp - nNn;
    const bool lanepipe = nLp > 0 && nLp >= nQ && nLp >= nNoisy && nLp >= nNn;
    const bool nasalFree = !lanepipe && nNn > 0 && nNn >= nQ && nNn >= nNoisy;
    const bool noisy = !lanepipe && !nasalFree && nNoisy >= nQ;
    const GroupPlan pl = plan_group(b->layout, noisy, b->nSlots, nNoisy, b->cus);
    const bool fast = b->mode == MODE_FAST;
    const long long nTr = tracked_count(b);
    const long long nDir = direct_count(b);
    const bool directG = noisy && nDir > 0 && nDir >= nTr && nDir >= nNoisy - nTr - nDir;
    const bool tracked = noisy && !directG && nTr > 0 && nTr >= nNoisy - nTr - nDir;
    const void* fn;
    int ldsBytes = LdsLayout<false>::kBytes, chunk = 0, wavesPerGroup = 1;
    long long groups = (nNn + kLanes - 1) / kLanes + (nQ + kLanes - 1) / kLanes + (nNoisy + kLanes - 1) / kLanes;
    if (nasalFree) {
        if (pl.chunk == 32) { fn = fast ? (const void*)klatt_systolic<MODE_FAST, false, 32, 1, false> : (const void*)klatt_systolic<MODE_EXACT, false, 32, 1, false>; ldsBytes = SysLds<false, 32>::kBytes; }
        else { fn = fast ? (const void*)klatt_systolic<MODE_FAST, false, 16, 2, false> : (const void*)klatt_systolic<MODE_EXACT, false, 16, 2, false>; ldsBytes = SysLds<false, 16>::kBytes; }
        chunk = pl.chunk;
        wavesPerGroup = kStages;
    } else if (lanepipe) {
        const long long g = (nLp + kLpUPG - 1) / kLpUPG;
        if (g <= b->cus) { fn = fast ? (const void*)klatt_lanepipe<MODE_FAST, KLATT_LP_CH, 1> : (const void*)klatt_lanepipe<MODE_EXACT, KLATT_LP_CH, 1>; ldsBytes = LpLds<KLATT_LP_CH>::kBytes; chunk = KLATT_LP_CH; }
        else { fn = fast ? (const void*)klatt_lanepipe<MODE_FAST, 16, 2> : (const void*)klatt_lanepipe<MODE_EXACT, 16, 2>; ldsBytes = LpLds<16>::kBytes; chunk = 16; }
        wavesPerGroup = kStages;
        groups = g;
    } else if (directG) {
        if (direct_lean(b, (nDir + kLanes - 1) / kLanes)) {
            fn = fast ? (const void*)klatt_direct<MODE_FAST, kDirectLeanChunk, 4> : (const void*)klatt_direct<MODE_EXACT, kDirectLeanChunk, 4>;
            ldsBytes = DirectLds<kDirectLeanChunk>::kBytes; chunk = kDirectLeanChunk;
        } else {
            fn = fast ? (const void*)klatt_direct<MODE_FAST, kDirectChunk, 2> : (const void*)klatt_direct<MODE_EXACT, kDirectChunk, 2>;
            ldsBytes = DirectLds<kDirectChunk>::kBytes; chunk = kDirectChunk;
        }
        wavesPerGroup = kDirectStages;
    } else if (tracked) {
        if (pl.chunk == 8) { fn = fast ? (const void*)klatt_systolic<MODE_FAST, true, KLATT_FLAT_CH, KLATT_FLAT_WPS, true, false, true> : (const void*)klatt_systolic<MODE_EXACT, true, KLATT_FLAT_CH, KLATT_FLAT_WPS, true, false, true>; ldsBytes = SysLds<true, KLATT_FLAT_CH, true>::kBytes; chunk = KLATT_FLAT_CH; }
        else { fn = fast ? (const void*)klatt_systolic<MODE_FAST, true, 16, 1, true, false, true> : (const void*)klatt_systolic<MODE_EXACT, true, 16, 1, true, false, true>; ldsBytes = SysLds<true, 16, true>::kBytes; chunk = 16; }
        wavesPerGroup = kStages;
    } else if (pl.systolic) {
        if (noisy && pl.chunk == 8) { fn = fast ? (const void*)klatt_systolic<MODE_FAST, true, 8, 2> : (const void*)klatt_systolic<MODE_EXACT, true, 8, 2>; ldsBytes = SysLds<true, 8>::kBytes; }
        else if (noisy) { fn = fast ? (const void*)klatt_systolic<MODE_FAST, true, 16> : (const void*)klatt_systolic<MODE_EXACT, true, 16>; ldsBytes = SysLds<true, 16>::kBytes; }
        else if (pl.chunk == 32) { fn = fast ? (const void*)klatt_systolic<MODE_FAST, false, 32> : (const void*)klatt_systolic<MODE_EXACT, false, 32>; ldsBytes = SysLds<false, 32>::kBytes; }
        else { fn = fast ? (const void*)klatt_systolic<MODE_FAST, false, 16> : (const void*)klatt_systolic<MODE_EXACT, false, 16>; ldsBytes = SysLds<false, 16>::kBytes; }
        chunk = pl.chunk;
        wavesPerGroup = kStages;
    } else {
        fn = fast ? (noisy ? (const void*)klatt_synthesize<MODE_FAST, false, true> : (const void*)klatt_synthesize<MODE_FAST, false, false>)
                  : (noisy ? (const void*)klatt_synthesize<MODE_EXACT, false, true> : (const void*)klatt_synthesize<MODE_EXACT, false, false>);
    }
    HIP_TRY(hipFuncGetAttributes(&fa, fn));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, b->device));
    info[0] = fa.numRegs;
    info[1] = ldsBytes;
    info[2] = (int)(groups * wavesPerGroup);
    info[3] = prop.multiProcessorCount;
    info[4] = (int)(prop.sharedMemPerMultiprocessor / ldsBytes);
    info[5] = (int)fa.localSizeBytes;   // scratch
    if (nInfo >= 8) { info[6] = chunk; info[7] = noisy ? 1 : 0; }
    // (utterances, not slots: the lane-pipelined kernel takes all of the nasal-free group or none of it)
    if (nInfo >= 10) { info[8] = lanepipe ? 1 : 0; info[9] = (int)std::min<long long>(nLp > 0 ? b->nNoNasalUtt : 0, 0x7FFFFFFF); }
    if (nInfo >= 12) { info[10] = nasalFree ? 1 : 0; info[11] = (int)std::min<long long>(nNn > 0 ? b->nNoNasalUtt : 0, 0x7FFFFFFF); }
    if (nInfo >= 16) {
        info[12] = (int)std::min<long long>(nTr > 0 ? b->nTrackedUtt : 0, 0x7FFFFFFF); info[13] = (int)std::min<long long>(b->nJobs, 0x7FFFFFFF);
        info[14] = (int)std::min<long long>(b->trackEntries * (long long)sizeof(double2) >> 20, 0x7FFFFFFF); info[15] = tracked ? 1 : 0;
    }
    if (nInfo >= 20) {
        info[16] = (int)std::min<long long>(b->nDirectUtt, 0x7FFFFFFF); info[17] = directG ? 1 : 0;
        info[18] = (int)std::min<long long>(b->nDirectFrames * (long long)(kDirectEntries * sizeof(double2) + kDirectStages * sizeof(DirectHdr)) >> 20, 0x7FFFFFFF);
        info[19] = 0;
    }
    return 0;
}


// Host-only view of the track planning (tests, tools): the plan speechPlayer_batch_setUtterances would make for these
// utterances (eligible[u] != 0: utterance u may be tracked; NULL: all).  Per frame: first entry and resonator mask of its
// fade's track (0 / 0 in utterances that are not tracked); per utterance: tracked or not.  Returns the number of distinct
// tracks, *nEntries the 16-byte entries they hold; -1 on bad arguments.  Touches no device.
static long long plan_tracks_view(long long nUtterances, const long long* frameStart, const speechPlayer_frame_t* frames,
                                  const unsigned int* fadeDuration, const unsigned char* isNull, const unsigned char* eligible,
                                  long long budgetMB, const void* facts24, unsigned long long* trackOff, unsigned int* trackMask, unsigned char* tracked,
                                  unsigned long long* nEntries, long long* collisionAt)
{
    begin_call();
    if (nUtterances < 0 || !frameStart || frameStart[0] != 0) { set_error("planTracks: bad arguments"); return -1; }
    for (long long u = 0; u < nUtterances; ++u)
        if (frameStart[u + 1] < frameStart[u]) { set_error("planTracks: frameStart not monotone at %lld", u); return -1; }
    const long long nF = frameStart[nUtterances];
    if (nF > 0 && (!frames || !fadeDuration)) { set_error("planTracks: bad frame arrays"); return -1; }
    std::vector<FrameMeta> meta((size_t)nF);
    for (long long k = 0; k < nF; ++k) {
        meta[k].minSamples = 0; meta[k].userIndex = -1;
        meta[k].fadeSamples = std::max(fadeDuration[k], 1u);
        meta[k].flags = (isNull && isNull[k]) ? FRAME_NULL : 0u;
    }
    std::vector<unsigned char> all;
    if (!eligible) { all.assign((size_t)nUtterances, 1); eligible = all.data(); }
    TrackPlan plan;
    std::vector<FrameFacts> own;
    const FrameFacts* facts = static_cast<const FrameFacts*>(facts24);
    if (!facts) {
        own.resize((size_t)nF);
        parallel_ranges(nF, 1 << 14, [&](long long a, long long e) { for (long long k = a; k < e; ++k) own[k] = frame_facts(reinterpret_cast<const double*>(frames + k), 1e300, 1e300); });
        facts = own.data();
    }
    plan_tracks(nUtterances, frameStart, FrameSource{frames, nullptr, nullptr}, facts, meta.data(), eligible, budgetMB, plan);
    // what klatt_verify_shared does on the device, here on the host: every frame recognised by its hash against the frame that stood for it
    if (collisionAt) *collisionAt = -1;
    for (long long k = 0; k < nF && !plan.rep.empty(); ++k) {
        const uint32_t r = plan.rep[(size_t)k];
        if (r == 0xFFFFFFFFu || (long long)r == k) continue;
        if (!shape_values_equal(reinterpret_cast<const double*>(frames + k), reinterpret_cast<const double*>(frames + r))) {
            if (collisionAt) *collisionAt = k;
            set_error("planTracks: frames %lld and %u carry one 128-bit shape hash and different values", k, r);
            return -2;
        }
    }
    for (long long k = 0; k < nF; ++k) {
        if (trackOff) trackOff[k] = plan.ref[k].off;
        if (trackMask) trackMask[k] = plan.ref[k].mask;
    }
    // references of utterances that ended up untracked mean nothing: clear them
    for (long long u = 0; u < nUtterances; ++u) {
        if (tracked) tracked[u] = plan.tracked[u];
        if (!plan.tracked[u])
            for (long long k = frameStart[u]; k < frameStart[u + 1]; ++k) { if (trackOff) trackOff[k] = 0; if (trackMask) trackMask[k] = 0; }
    }
    if (nEntries) *nEntries = plan.entries;
    return (long long)plan.jobs.size();
}

long long speechPlayer_planTracks(long long nUtterances, const long long* frameStart, const speechPlayer_frame_t* frames,
                                  const unsigned int* fadeDuration, const unsigned char* isNull, const unsigned char* eligible,
                                  long long budgetMB, unsigned long long* trackOff, unsigned int* trackMask, unsigned char* tracked,
                                  unsigned long long* nEntries)
{
    try {
        return plan_tracks_view(nUtterances, frameStart, frames, fadeDuration, isNull, eligible, budgetMB, nullptr, trackOff, trackMask, tracked, nEntries, nullptr);
    } catch (const std::exception& e) { set_error("planTracks: %s", e.what()); return -1; }
}

long long speechPlayer_planTracksFacts(long long nUtterances, const long long* frameStart, const speechPlayer_frame_t* frames,
                                       const unsigned int* fadeDuration, const unsigned char* isNull, const unsigned char* eligible,
                                       long long budgetMB, const void* facts24, unsigned long long* trackOff, unsigned int* trackMask,
                                       unsigned char* tracked, unsigned long long* nEntries, long long* collisionAt)
{
    try {
        return plan_tracks_view(nUtterances, frameStart, frames, fadeDuration, isNull, eligible, budgetMB, facts24, trackOff, trackMask, tracked, nEntries, collisionAt);
    } catch (const std::exception& e) { set_error("planTracksFacts: %s", e.what()); return -1; }
}

// the engine's worker threads for the other translation unit (frame_producer.cpp): fn(ctx, a, e) over [0, n) in ranges
void speechPlayer_internal_parallel(long long n, long long grain, void (*fn)(void* ctx, long long a, long long e), void* ctx)
{
    if (n <= 0 || !fn) return;
    parallel_ranges(n, grain, [&](long long a, long long e) { fn(ctx, a, e); });
}

long long speechPlayer_frameFacts(const speechPlayer_frame_t* frames, long long nFrames, int sampleRate, int onDevice, void* facts24)
{
    begin_call();
    if (nFrames < 0 || (nFrames > 0 && (!frames || !facts24)) || sampleRate <= 0) { set_error("frameFacts: bad arguments"); return -1; }
    const double maxBw = 690.0 * sampleRate / M_PI, maxF = 9900.0 * sampleRate / (2.0 * M_PI);
    FrameFacts* const out = static_cast<FrameFacts*>(facts24);
    if (!onDevice) {
        parallel_ranges(nFrames, 1 << 13, [&](long long a, long long e) { for (long long k = a; k < e; ++k) out[k] = frame_facts(reinterpret_cast<const double*>(frames + k), maxF, maxBw); });
        return nFrames;
    }
    if (nFrames == 0) return 0;
    DeviceBuffer<double> dF;
    DeviceBuffer<FrameFacts> dO;
    struct Release { DeviceBuffer<double>& a; DeviceBuffer<FrameFacts>& b; ~Release() { a.release(); b.release(); } } release{dF, dO};
    if (dF.reserve((size_t)nFrames * kNumParams) || dO.reserve((size_t)nFrames)) return -1;
    HIP_TRY(hipMemcpy(dF.ptr, frames, (size_t)nFrames * kNumParams * sizeof(double), hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)std::min<long long>((nFrames + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(klatt_frame_facts, dim3(grid), dim3(256), 0, nullptr, dF.ptr, dO.ptr, nFrames, maxF, maxBw);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dO.ptr, (size_t)nFrames * sizeof(FrameFacts), hipMemcpyDeviceToHost));
    return nFrames;
}

// Host-only view of what speechPlayer_batch_setUtterances hands klatt_seeds for the utterances it sends to the direct stages
// (tests; touches no device): per frame the frames its fade starts from and ends on (0xFFFFFFFF: none -- all zero) and the flags
// (bit 0: the start's preFormantGain is gated off, bit 1: the end's).  Returns the number of frames; -1 on bad arguments.
long long speechPlayer_planDirect(long long nUtterances, const long long* frameStart, const unsigned char* isNull,
                                  unsigned int* from, unsigned int* to, unsigned int* flags)
{
    begin_call();
    if (nUtterances < 0 || !frameStart || frameStart[0] != 0) { set_error("planDirect: bad arguments"); return -1; }
    for (long long u = 0; u < nUtterances; ++u)
        if (frameStart[u + 1] < frameStart[u]) { set_error("planDirect: frameStart not monotone at %lld", u); return -1; }
    const long long nF = frameStart[nUtterances];
    if (nF >= 0xFFFFFFFFll) { set_error("planDirect: too many frames"); return -1; }
    std::vector<FrameMeta> meta((size_t)nF);
    for (long long k = 0; k < nF; ++k) { meta[k].minSamples = 0; meta[k].fadeSamples = 1; meta[k].userIndex = -1; meta[k].flags = (isNull && isNull[k]) ? FRAME_NULL : 0u; }
    std::vector<DirectJob> jobs;
    jobs.reserve((size_t)nF);
    for (long long u = 0; u < nUtterances; ++u) walk_fade_ends(frameStart[u], frameStart[u + 1], meta.data(), jobs);
    for (long long k = 0; k < nF; ++k) {
        if (from) from[k] = jobs[(size_t)k].from;
        if (to) to[k] = jobs[(size_t)k].to;
        if (flags) flags[k] = jobs[(size_t)k].flags;
    }
    return nF;
}

}  // extern "C"
