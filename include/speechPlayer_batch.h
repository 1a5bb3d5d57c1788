/*
 * speechPlayer_batch.h -- additive batch entry points of the MI355X Klatt engine.
 *
 * The reference has no batch interface: one handle is one stream
 * (reference src/speechPlayer.cpp:19-23) and a caller loops
 * speechPlayer_queueFrame / speechPlayer_synthesize per stream
 * (test_speakIpa.py:24-27, nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:62-81,222-233).
 * A batch replaces N such loops: N frame streams in, N int16 PCM streams out, one
 * utterance per wavefront lane on the GPU.  Every utterance behaves exactly like a
 * fresh handle that had all its frames queued (no purge) and was drained.
 *
 * Plain C: host pointers and sizes only.  All functions return 0 on success and a
 * negative value on failure unless stated; speechPlayer_lastError() describes it.
 */
#ifndef NVSP_AMD_SPEECHPLAYER_BATCH_H
#define NVSP_AMD_SPEECHPLAYER_BATCH_H

#include "speechPlayer.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef void* speechPlayer_batch_t;

/* Arithmetic modes (speechPlayer_batch_setOption(b, "mode", ...)). */
#define SPEECHPLAYER_MODE_EXACT 0 /* f64, separate rounding of every operation, libm-grade coefficients */
#define SPEECHPLAYER_MODE_FAST 1  /* f64 state, the resonators' multiply-adds fused (within the tolerance, not bit-exact by construction) */

/* Bind a batch engine to HIP device `device` (-1: the current device). NULL on failure. */
speechPlayer_batch_t speechPlayer_batch_create(int sampleRate, int device);
void speechPlayer_batch_destroy(speechPlayer_batch_t batch);

/* Options: "mode" (see above); "sort" (1: pack wavefronts by utterance length, default 1);
 * "layout" (-1: chosen per batch, default; 2: lane-pipelined workgroups for the quiet, nasal-free utterances
 * whatever the batch size; 1: stage-parallel workgroups, four wavefronts per 64 utterances; 0: one wavefront
 * per 64 utterances);
 * "tracks" (1, default: noisy utterances whose parameters are all finite take what they need on fade samples -- resonator
 * coefficients, interpolated gains -- from tracks: evaluated densely by a kernel of its own before the synthesis kernel, one
 * track per distinct fade of the batch, instead of exp/cos, interpolation and a frame state machine inside the sample
 * recurrence; 0: never) and "track_budget_mb" (device memory the tracks of a batch may take, default 4096, which is also the most: the flat stages address the tracks with 32-bit byte offsets; a batch whose
 * tracks do not fit runs without them);
 * "direct" (what runs the noisy, finite utterances that got no tracks -- a batch whose fades share nothing would need 12 bytes of
 * track per output sample: 1, default: the direct stages (every fade sample's coefficients computed in place from per-frame
 * seeds, eight wavefronts per 64 utterances) unless -- in MODE_EXACT -- the utterances are time-aligned copies of few sentences,
 * which the stages with the frame state machine run faster; 2: the direct stages always; 0: never).
 * "quiet_last" (1, default: a launch queues the quiet groups' kernels behind the noisy groups', whose few long workgroups then start first;
 * 0: in front; read by every launch).
 * "tracks", "track_budget_mb" and "direct" are read by speechPlayer_batch_setUtterances: set them before it.  No option changes the
 * PCM of MODE_EXACT; MODE_FAST stays within its tolerance whichever kernel runs (the direct stages advance the coefficients of a
 * fade by recurrences, re-seeded exactly at every fade's first sample: relative error <= 4 F 2^-53 after F fade samples). */
int speechPlayer_batch_setOption(speechPlayer_batch_t batch, const char* name, int value);

/*
 * Describe the batch and make it resident in HBM (host arrays are copied).
 *   frameStart[nUtterances+1]  utterance u owns frames frameStart[u] .. frameStart[u+1]-1
 *   frames[nFrames]            what each speechPlayer_queueFrame call would have been given
 *   minFrameDuration/fadeDuration[nFrames]  in samples, as in speechPlayer_queueFrame
 *   userIndex[nFrames]         may be NULL (all -1)
 *   isNull[nFrames]            may be NULL (none); nonzero = the call passed framePtr==NULL
 *   noiseSeed[nUtterances]     may be NULL (seed = utterance number); selects the utterance's
 *                              noise stream (the reference's rand() is process-global,
 *                              src/speechWaveGenerator.cpp:40; the engine defines one stream per
 *                              utterance instead -- see DESIGN.md "Noise")
 */
int speechPlayer_batch_setUtterances(speechPlayer_batch_t batch, long long nUtterances, const long long* frameStart,
	const speechPlayer_frame_t* frames, const unsigned int* minFrameDuration, const unsigned int* fadeDuration,
	const int* userIndex, const unsigned char* isNull, const unsigned int* noiseSeed);

/*
 * The same batch in COMPACT form: frame lists that utterances SHARE, and frames as 32-byte records the device expands.
 * The reference produces frames lazily, one utterance at a time (ipa.py:336-353), and hands each to speechPlayer_queueFrame as a
 * 376-byte struct (src/frame.cpp:90-101).  A batch of N utterances seldom holds N different frame streams (BASELINE's configurations
 * are 512 streams instanced), and a frame a producer emits is one of a few hundred parameter vectors plus a pitch pair and two
 * durations.  Both entry points describe `nLists` frame lists once and say per utterance which list it speaks (listOf[u]; NULL:
 * utterance u speaks list u, nUtterances == nLists); utterances of one list read the same frames in HBM and differ in their noise
 * seed alone.  Everything else is as in speechPlayer_batch_setUtterances (each utterance a fresh handle with the list's frames queued).
 *   speechPlayer_batch_setUtterancesShared   the lists as full frames (listStart[nLists+1] into frames / durations / ...)
 *   speechPlayer_batch_setRecords            the lists as records: record k stands for the frame whose parameters 1..45 are
 *                                            shapes[records[k].shape] and whose voicePitch / endVoicePitch are the record's own
 *                                            (shape SPEECHPLAYER_RECORD_SILENCE: framePtr == NULL).  32 bytes per frame cross the
 *                                            link, the 376-byte frames are built in HBM (klatt_expand_frames), and the planner
 *                                            recognises equal frames by their shape number -- exactly, nothing is hashed.
 */
#define SPEECHPLAYER_RECORD_SILENCE 0xFFFFFFFFu
typedef struct {
	double voicePitch, endVoicePitch;             /* parameters 0 and 46 of the frame */
	unsigned int shape;                           /* row of the call's shape table, or SPEECHPLAYER_RECORD_SILENCE */
	unsigned int minFrameDuration, fadeDuration;  /* samples, as in speechPlayer_queueFrame */
	int userIndex;                                /* -1: none */
} speechPlayer_frameRecord_t;                     /* 32 bytes */
int speechPlayer_batch_setUtterancesShared(speechPlayer_batch_t batch, long long nLists, const long long* listStart,
	const speechPlayer_frame_t* frames, const unsigned int* minFrameDuration, const unsigned int* fadeDuration,
	const int* userIndex, const unsigned char* isNull, long long nUtterances, const unsigned int* listOf, const unsigned int* noiseSeed);
int speechPlayer_batch_setRecords(speechPlayer_batch_t batch, long long nShapes, const speechPlayer_frame_t* shapes,
	long long nLists, const long long* listStart, const speechPlayer_frameRecord_t* records,
	long long nUtterances, const unsigned int* listOf, const unsigned int* noiseSeed);
/* The frames of utterance u as they are resident in HBM (downloaded; after any of the set calls): what each speechPlayer_queueFrame
 * call of that utterance would have been given.  Returns the utterance's number of frames; fills the arrays (each may be NULL)
 * when it is <= capacity.  fadeDuration comes back as the engine uses it (>= 1: reference src/speechPlayer.cpp:36). */
long long speechPlayer_batch_frames(speechPlayer_batch_t batch, long long utterance, speechPlayer_frame_t* frames,
	unsigned int* minFrameDuration, unsigned int* fadeDuration, int* userIndex, unsigned char* isNull, long long capacity);

/* Number of samples utterance u produces: sum over its frames of max(M, F+1)+1. */
long long speechPlayer_batch_utteranceSamples(speechPlayer_batch_t batch, long long utterance);
long long speechPlayer_batch_totalSamples(speechPlayer_batch_t batch);
long long speechPlayer_batch_totalFrames(speechPlayer_batch_t batch);
int speechPlayer_batch_sampleRate(speechPlayer_batch_t batch);

/* Launch the synthesis kernel on the batch's stream (asynchronous), and wait for it. */
int speechPlayer_batch_synthesize(speechPlayer_batch_t batch);
int speechPlayer_batch_wait(speechPlayer_batch_t batch);

/* Copy utterance u's PCM to the host (after wait). Returns samples copied (<= capacity). */
long long speechPlayer_batch_read(speechPlayer_batch_t batch, long long utterance, sample* sampleBuf, long long capacity);
/* The same as float samples in [-1, 1] (value / 32767, the scaling of the reference's audio sink,
 * lavPlayer.py:17); the conversion runs on the device. Returns samples copied. */
long long speechPlayer_batch_readFloat(speechPlayer_batch_t batch, long long utterance, float* sampleBuf, long long capacity);
/* Copy every utterance's PCM, concatenated in utterance order; outStart[nUtterances+1] receives
 * the offsets. Returns total samples. */
long long speechPlayer_batch_readAll(speechPlayer_batch_t batch, sample* sampleBuf, long long capacity, long long* outStart);
/* The two large transfers of a batch -- frames in (speechPlayer_batch_setUtterances), PCM out (speechPlayer_batch_readAll) -- run
 * as ONE DMA at the link's rate when the host side is page-locked memory: from speechPlayer_hostAlloc (freed with
 * speechPlayer_hostFree), or memory the caller registered itself (hipHostRegister).  Pageable buffers work as before, through
 * bounce buffers.  The PCM is put into dense utterance order ON THE DEVICE first (the pool pads utterances to 32 samples), so the
 * bytes that cross the link are the bytes of the caller's buffer.  (Additive: the reference has one stream per handle and copies
 * through speechPlayer_synthesize, src/speechPlayer.cpp:41-45.) */
void* speechPlayer_hostAlloc(long long bytes);
void speechPlayer_hostFree(void* p);
/* speechPlayer_batch_readAll without waiting: compaction and copy are queued behind the synthesis and run beside whatever is launched
 * next; sampleBuf must be page-locked (else -1).  Returns the samples that will have arrived when speechPlayer_batch_readWait returns. */
long long speechPlayer_batch_readAllAsync(speechPlayer_batch_t batch, sample* sampleBuf, long long capacity, long long* outStart);
int speechPlayer_batch_readWait(speechPlayer_batch_t batch);
/* Digest of the PCM, computed on the device (for checks of batches whose PCM is too large to copy): perUtterance[u]
 * (may be NULL) = sum over utterance u's samples of mix64(position, value), *whole (may be NULL) = a digest of those in
 * utterance order.  Equal PCM <=> equal digests (up to 2^-64); the kernel is an HBM-bound read of the pool. */
int speechPlayer_batch_digest(speechPlayer_batch_t batch, unsigned long long* perUtterance, unsigned long long* whole);
/* speechPlayer_getLastIndex for utterance u after the run. */
int speechPlayer_batch_getLastIndex(speechPlayer_batch_t batch, long long utterance);

/* Zero-copy access for GPU consumers: device pointer of the PCM pool and the sample offset of
 * utterance u in it (offsets are padded to 64-sample boundaries). */
const sample* speechPlayer_batch_devicePcm(speechPlayer_batch_t batch);
long long speechPlayer_batch_deviceOffset(speechPlayer_batch_t batch, long long utterance);

/* Measurement: run `launches` synthesis launches back to back on the batch's stream and report
 * each launch's duration in milliseconds from HIP events recorded on that stream. */
int speechPlayer_batch_time(speechPlayer_batch_t batch, int launches, float* msPerLaunch);

/* Kernel resource facts for reports: fills vgprs, ldsBytes, wavefronts launched, workgroups per CU. */
int speechPlayer_batch_kernelInfo(speechPlayer_batch_t batch, int* info, int nInfo);
/* info[12..15] (nInfo >= 16): utterances that take their coefficients from tracks, distinct tracks of the batch, their size in
 * MB, 1 if the reported kernel is the tracked (flat-stage) instantiation.
 * info[16..19] (nInfo >= 20): utterances on the direct stages, 1 if the reported kernel is the direct one, the size in MB of the
 * per-frame seeds every launch writes (1376 + 128 bytes per frame of those utterances), 0. */

/* Host-only view of the track planning of speechPlayer_batch_setUtterances (tests, tools; touches no device):
 * the plan for these utterances under a budget of budgetMB.  eligible[u] != 0: utterance u may be tracked (NULL: all; the
 * engine itself tracks the noisy utterances whose parameters are all finite).  Per frame: first entry and mask of the entry
 * kinds that move (bits 0..13: N0, NP, c6..c1, p1..p6; 14..23: pairs of gains) of its fade's track; per utterance: tracked or not (nothing is, once a tenth of the
 * eligible utterances did not fit).  Returns the number of distinct tracks, *nEntries their 16-byte entries; -1 on bad
 * arguments.  A fade's end points follow reference src/frame.cpp:55-72; equal fades share a track. */
long long speechPlayer_planTracks(long long nUtterances, const long long* frameStart, const speechPlayer_frame_t* frames,
	const unsigned int* fadeDuration, const unsigned char* isNull, const unsigned char* eligible, long long budgetMB,
	unsigned long long* trackOff, unsigned int* trackMask, unsigned char* tracked, unsigned long long* nEntries);

/* The same with the per-frame facts GIVEN (facts24: nFrames x 24 bytes as speechPlayer_frameFacts writes them; NULL: computed) and with
 * the check the engine makes of every frame the planner recognised by its 128-bit hash: its 45 shape values compared with those of the
 * first frame that carried the hash (on the device in speechPlayer_batch_setUtterances -- klatt_verify_shared --, here on the host).
 * Returns -2 and the frame in *collisionAt when two frames share a hash and differ (a test forges such facts; the engine then plans
 * the batch again without tracks and leaves a message in speechPlayer_lastError with the call succeeding). */
long long speechPlayer_planTracksFacts(long long nUtterances, const long long* frameStart, const speechPlayer_frame_t* frames,
	const unsigned int* fadeDuration, const unsigned char* isNull, const unsigned char* eligible, long long budgetMB, const void* facts24,
	unsigned long long* trackOff, unsigned int* trackMask, unsigned char* tracked, unsigned long long* nEntries, long long* collisionAt);
/* Host-only view of the fade end points speechPlayer_batch_setUtterances derives for the utterances it sends to the direct stages
 * (tests; touches no device; follows reference src/frame.cpp:55-72): per frame the frames its fade starts from and ends on
 * (0xFFFFFFFF: none -- all values zero) and flags (bit 0: the start's preFormantGain is gated off -- silence --, bit 1: the end's).
 * Returns the number of frames; -1 on bad arguments. */
long long speechPlayer_planDirect(long long nUtterances, const long long* frameStart, const unsigned char* isNull,
                                  unsigned int* from, unsigned int* to, unsigned int* flags);
/* What speechPlayer_batch_setUtterances learns of every frame before it plans a batch (tests, tools): per frame 24 bytes --
 * {u64 h0, u64 h1: a 128-bit hash of the 45 values a track depends on (every parameter but the two pitches), u32 flags, u32 0};
 * flags: 1 a noise gain is set or the parallel bank's coefficients may not be finite, 2 a parameter is NaN or infinite, 4 the nasal
 * pair is coupled in (or could not be skipped safely), 8 a frequency or bandwidth outside the direct stages' range.  onDevice = 0: the
 * host's evaluation (touches no device); 1: the device's (klatt_frame_facts: what frames arriving from page-locked memory get) -- the
 * same function (csrc/klatt_plan.h), the same bytes.  Returns nFrames, -1 on error. */
long long speechPlayer_frameFacts(const speechPlayer_frame_t* frames, long long nFrames, int sampleRate, int onDevice, void* facts24);

/*
 * One batch over the GPUs of a node.  Utterances are independent (the reference's only cross-handle coupling
 * is rand(), src/speechWaveGenerator.cpp:40, replaced by per-utterance noise streams), so the batch is cut into contiguous
 * shards of near-equal total SAMPLE count, one per device; one host thread per device uploads its shard, the devices
 * synthesise side by side, nothing is exchanged between them.  Results are addressed by the batch's own utterance numbers.
 *   devices[nDevices]  HIP device of each shard (NULL: 0 .. nDevices-1; a device may be listed more than once)
 */
typedef void* speechPlayer_node_t;
speechPlayer_node_t speechPlayer_node_create(int sampleRate, int nDevices, const int* devices);
void speechPlayer_node_destroy(speechPlayer_node_t node);
int speechPlayer_node_devices(speechPlayer_node_t node);
int speechPlayer_node_setOption(speechPlayer_node_t node, const char* name, int value);
/* Arguments as speechPlayer_batch_setUtterances; noiseSeed NULL = the utterance's number in the WHOLE batch. */
int speechPlayer_node_setUtterances(speechPlayer_node_t node, long long nUtterances, const long long* frameStart,
	const speechPlayer_frame_t* frames, const unsigned int* minFrameDuration, const unsigned int* fadeDuration,
	const int* userIndex, const unsigned char* isNull, const unsigned int* noiseSeed);
/* The node's batch in compact form (speechPlayer_batch_setRecords, speechPlayer_batch_setIpa / _setIpaVoices): lists, records and the shape
 * table go to every shard, the deal decides which utterances each shard speaks.  sampleRate: the node's (as given to speechPlayer_node_create);
 * voiceOf NULL: voiceName for every text (NULL / "": none). */
int speechPlayer_node_setRecords(speechPlayer_node_t node, long long nShapes, const speechPlayer_frame_t* shapes, long long nLists,
	const long long* listStart, const speechPlayer_frameRecord_t* records, long long nUtterances, const unsigned int* listOf,
	const unsigned int* noiseSeed);
int speechPlayer_node_setIpa(speechPlayer_node_t node, int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed,
	const double* basePitch, double inflection, const char* clauseTypes, const int* voiceOf, const char* voiceName,
	double trailingSilenceMs, const unsigned int* noiseSeed);
int speechPlayer_node_synthesize(speechPlayer_node_t node);   /* asynchronous on every device */
int speechPlayer_node_wait(speechPlayer_node_t node);
long long speechPlayer_node_totalSamples(speechPlayer_node_t node);
long long speechPlayer_node_read(speechPlayer_node_t node, long long utterance, sample* sampleBuf, long long capacity);
int speechPlayer_node_getLastIndex(speechPlayer_node_t node, long long utterance);
/* Option "deal" (besides the batch options, which go to every shard): 0 (default) contiguous shards of near-equal total sample count;
 * 1 the SORTED deal of SURVEY 8(e) -- utterances sorted by length, blocks of 64 (one wavefront) dealt round-robin -- so that every
 * device sees the same length distribution whatever the order of the batch (the shards' frames are then gathered on the host).
 * Shard `shard`: its first utterance (-1 under the sorted deal: its utterances are not a range), utterance count, sample count and
 * device (each pointer may be NULL); speechPlayer_node_shardUtterances lists the shard's utterances in the shard's own order
 * (returns their number; fills when it is <= capacity). */
int speechPlayer_node_shardInfo(speechPlayer_node_t node, int shard, long long* firstUtterance, long long* nUtterances, long long* samples, int* device);
long long speechPlayer_node_shardUtterances(speechPlayer_node_t node, int shard, long long* utterances, long long capacity);
/* The shard's own batch object, for everything else (digest, float output, device pointers); utterance numbers are
 * relative to the shard's first utterance there. */
speechPlayer_batch_t speechPlayer_node_part(speechPlayer_node_t node, int shard);
/* `launches` passes over the whole batch, all devices at once; wall-clock milliseconds per pass. */
int speechPlayer_node_time(speechPlayer_node_t node, int launches, float* msPerLaunch);

/*
 * Many LIVE streams on one GPU: advance nHandles handles created by
 * speechPlayer_initialize by up to sampleCount samples each in ONE kernel launch, one handle per
 * wavefront lane.  Exactly equivalent to calling speechPlayer_synthesize(handles[i], sampleCount,
 * sampleBufs[i]) for every i (queued frames, purge requests, index marks and saved state are per handle);
 * produced[i] receives each call's return value.  Handles must be distinct and share sample rate.
 * The reference's consumer loop is one thread per stream pulling 8192 samples
 * (nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:62-81); this is that loop for N streams.
 * Host cost per call: one control block (44 bytes per handle) and the frames queued since the last call travel, nothing else --
 * a handle's saved state and its queued frames live in a per-device arena (speechPlayer_queueFrame writes a frame once, into a
 * pinned log; a scatter kernel places the log's entries in the handles' rings).  Listing the handles in ascending order of
 * creation saves a sort.
 */
int speechPlayer_synthesizeMany(speechPlayer_handle_t* handles, int nHandles, unsigned int sampleCount, sample** sampleBufs, int* produced);
/* The same with the PCM left in HBM: handle i's samples start at *devicePcm + i * *rowStride (device memory, valid until the next
 * live call on that device).  For consumers on the GPU and for measuring the engine without the PCIe copy of the PCM. */
int speechPlayer_synthesizeManyDevice(speechPlayer_handle_t* handles, int nHandles, unsigned int sampleCount, const sample** devicePcm,
	long long* rowStride, int* produced);
/* Kernel time in milliseconds of the last live call on HIP device `device` (HIP events on its stream). */
float speechPlayer_lastLiveKernelMs(int device);
/* Kernel launches that call took: 1, unless a handle had more frames queued than the 256 its device-side ring holds and the
 * ring's frames ended before sampleCount samples -- such a call proceeds in pieces, with the same result. */
int speechPlayer_lastLiveLaunches(int device);
/* Process-wide options.  "live_layout": the kernel that advances live handles -- 1 (default): the stage-parallel kernel, four
 * wavefronts per 64 handles; 0: the lane kernel, one wavefront per 64 handles.  Same saved state, same PCM.
 * "live_cus": pulls of more than live_cus x 64 handles take the two-workgroups-per-CU instantiation of the stream kernel (0, default:
 * the device's CU count -- 16 384 handles on MI355X; a small value lets a test reach that kernel with a few hundred handles).
 * Memory: every live handle owns a slot of ~100 KB of HBM in a per-device arena (a ring of 256 queued frames with their durations
 * and a 240-double state block); the arena doubles when the slots run out (old and new coexist during the move: ~2.4 GB transient
 * at 16 384 slots), and speechPlayer_initialize fails with SPEECHPLAYER_ERR_HIP when the device cannot hold it.  It does not shrink while
 * a handle lives; "live_trim" = 1 releases a device's arena (and the pull buffers) when the LAST handle on that device is terminated --
 * and at once on devices where none lives; 0 (default) keeps it for the next handles.
 * "live_replicate" (default 1): a pull of fewer than 32 handles fills the empty lanes of their wavefront with replicas of them (a sparse
 * wavefront runs up to 1.7 times slower), and ONE handle pulled alone is advanced in all 64 lanes by a kernel instantiation of its own that
 * computes its fades side by side across the lanes; 0: one lane per handle.  Same PCM, marks and counts either way.
 * "live_mode" (default 0): the arithmetic mode of handles created FROM NOW ON -- 0 MODE_EXACT (the reference's rounding, sample for sample),
 * 1 MODE_FAST (the filters' multiply-adds fused; within north_star's tolerance, held to <= 1 LSB and <= 5 one-LSB differences per million
 * samples against the oracle like the batches' MODE_FAST; one stream 1.27 -> 1.18 ms per 8192-sample pull).  Handles pulled together must share it.
 * "live_alone" (default 1536; 1: only a handle pulled alone): a pull of up to this many handles gives EVERY handle a wavefront of its own
 * (one workgroup per handle, 256 side by side on MI355X, further ones in rounds): handles that share a wavefront pay for one another --
 * unrelated handles 11.2 ms per 8192-sample pull however few they are -- while 2 .. 256 handles alone in their wavefronts take 1.4-1.7 ms
 * and 1024 take 6.2, 1536 9.3 (the two policies meet near 1850 handles).  Handles that speak IN STEP (same frames from the same sample) are the exception: beyond 256 of them sharing
 * wavefronts is faster (2.5 ms) -- set 1 for those.  Needs "live_replicate" 1 and "live_layout" 1.
 * "plan_hash_bits" (tests): how many bits of a frame's 128-bit shape hash the track planner looks at (default 128). */
int speechPlayer_setGlobalOption(const char* name, int value);
/* Choose a handle's noise stream (default 0); see DESIGN.md "Noise". */
int speechPlayer_setNoiseSeed(speechPlayer_handle_t playerHandle, unsigned int seed);

/*
 * The frame producer: IPA text -> the frame stream a caller would queue.  Host code, no GPU needed.
 * Native counterpart of the reference's ipa.generateFramesAndTiming (ipa.py:336-353, with :39-334 behind it) and of the
 * NVDA driver's voice presets (nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:86-125); same values for the same input.
 *   clauseType   '.', ',', '?', '!' or 0 (none: the statement contour), as the reference's clauseType argument
 *   voiceName    NULL / "" for none, else one of speechPlayer_voiceName(0 .. speechPlayer_voiceCount()-1); the preset is
 *                applied to every non-silence frame: absolute values first, then multipliers (applyVoiceToFrame)
 */
/* One utterance.  Returns its number of frames n; fills the arrays (each may be NULL) when n <= capacity.  isNull[k] != 0
 * marks silence (the reference yields None); durations are in MILLISECONDS as the reference yields them.  -1: unknown voice;
 * -2: a clauseType the intonation table does not hold (the reference raises KeyError, ipa.py:281). */
long long speechPlayer_ipa_frames(const char* ipaUtf8, double speed, double basePitch, double inflection, int clauseType,
	const char* voiceName, speechPlayer_frame_t* frames, unsigned char* isNull, double* durationMs, double* fadeMs, long long capacity);
/* Many utterances, packed as speechPlayer_batch_setUtterances takes them: durations converted to samples the way the
 * reference wrapper does (speechPlayer.py:53), each utterance followed by silence of trailingSilenceMs (fade 0) as
 * test_speakIpa.py:27 queues it (negative: none).  basePitch[nTexts] may be NULL (100 Hz), clauseTypes[nTexts] may be NULL
 * (none).  Returns the total number of frames; writes frameStart[nTexts+1] when given, and the frame arrays when all four
 * are given and frameCapacity suffices (call once with NULL arrays to size them).  Distinct (text, clause, pitch)
 * combinations are built once per call and instanced.  -1: bad arguments or unknown voice; -2: unknown clause type. */
long long speechPlayer_ipa_pack(int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed, const double* basePitch,
	double inflection, const char* clauseTypes, const char* voiceName, double trailingSilenceMs,
	long long* frameStart, speechPlayer_frame_t* frames, unsigned int* minFrameDuration, unsigned int* fadeDuration,
	unsigned char* isNull, long long frameCapacity);
/* Text in, batch resident in HBM: the producer's compact form (distinct (text, clause, base pitch, voice) combinations built once, as
 * lists of 32-byte records over a table of (voice, phoneme) shapes) handed to speechPlayer_batch_setRecords. */
int speechPlayer_batch_setIpa(speechPlayer_batch_t batch, long long nTexts, const char* const* ipaUtf8, double speed,
	const double* basePitch, double inflection, const char* clauseTypes, const char* voiceName, double trailingSilenceMs,
	const unsigned int* noiseSeed);
/* The same with a voice PER TEXT: voiceOf[i] = index of text i's voice (0 .. speechPlayer_voiceCount()-1; -1 none); NULL: none.
 * BASELINE configs[4] -- 256 voice-parameter variants x 16 384 utterances -- is this call with 256 defined voices. */
int speechPlayer_batch_setIpaVoices(speechPlayer_batch_t batch, long long nTexts, const char* const* ipaUtf8, double speed,
	const double* basePitch, double inflection, const char* clauseTypes, const int* voiceOf, double trailingSilenceMs,
	const unsigned int* noiseSeed);
/* The compact form itself, for callers that keep a batch's description (and for the tests): an object that owns the arrays
 * speechPlayer_batch_setRecords takes.  NULL on bad arguments. */
typedef void* speechPlayer_records_t;
typedef struct {
	long long nShapes; const speechPlayer_frame_t* shapes;
	long long nLists; const long long* listStart;
	long long nRecords; const speechPlayer_frameRecord_t* records;
	long long nUtterances; const unsigned int* listOf;
} speechPlayer_recordsView_t;
speechPlayer_records_t speechPlayer_ipa_records(int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed, const double* basePitch,
	double inflection, const char* clauseTypes, const int* voiceOf, const char* voiceName, double trailingSilenceMs);
int speechPlayer_records_view(speechPlayer_records_t records, speechPlayer_recordsView_t* view);
void speechPlayer_records_free(speechPlayer_records_t records);
/*
 * Optional text front-end: what the NVDA driver does before the frame producer (reference
 * nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:189-234), with eSpeak NG loaded at run time (dlopen of libespeak-ng.so.1, or of
 * $SPEECHPLAYER_ESPEAK_LIB) -- the library links against nothing of it.  PARITY UNPINNED: eSpeak NG is absent from the reference tree
 * and from this image; only the clause splitting, the replacements and the error path are tested.  Without the library every
 * function that needs it returns -3 (speechPlayer_text_available: 0) and speechPlayer_lastError() says what to install;
 * IPA input (speechPlayer_batch_setIpa) never needs it.
 *   speechPlayer_text_clauses   split text where white space follows one of . ? ! , : ; (:84, :189); per clause its byte range
 *                               [begin, end) in textUtf8, its type ('.', '!', '?', ',' or 0) and the pause after it in ms (:195-205);
 *                               returns the number of clauses (fills up to `capacity` of them; any array may be NULL)
 *   speechPlayer_text_fixups    the four replacements of :214-217 and the strip of :218; returns the bytes needed with the NUL
 *   speechPlayer_text_toIpa     one clause through espeak_TextToPhonemes (UTF-8 in, mode word 0x36100 + 0x82 as :210) + the fix-ups
 *   speechPlayer_batch_setText  one utterance per text: its clauses' frame streams one after the other (clause type per clause),
 *                               then silence of the last clause's pause / speed with a fade of max(10, 10 / speed) ms (:234);
 *                               espeakVoice NULL = "en"; basePitch[nTexts] may be NULL (100 Hz); -1 bad arguments, -3 no eSpeak
 */
int speechPlayer_text_available(void);
long long speechPlayer_text_clauses(const char* textUtf8, long long* begin, long long* end, char* clauseType, double* endPauseMs, long long capacity);
long long speechPlayer_text_fixups(const char* ipaUtf8, char* out, long long capacity);
long long speechPlayer_text_toIpa(const char* textUtf8, const char* espeakVoice, char* out, long long capacity);
int speechPlayer_batch_setText(speechPlayer_batch_t batch, long long nTexts, const char* const* textUtf8, const char* espeakVoice, double speed,
	const double* basePitch, double inflection, const char* voiceName, const unsigned int* noiseSeed);
/* The phoneme table the producer is driven by (the reference's data.py, as numbers): entry `index` of
 * speechPlayer_ipa_phonemeCount() -- its IPA symbol (UTF-8, NUL-terminated, symbolCapacity bytes), its 47 parameter values in
 * speechPlayer_frame_t order, which of them the entry sets (bit k of *fieldMask), and its class bits (1 _isVowel, 2 _isVoiced,
 * 4 _isNasal, 8 _isStop, 16 _isLiquid, 32 _isSemivowel, 64 _isAfricate, 128 _copyAdjacent).  Any output may be NULL.  0, or -1. */
int speechPlayer_ipa_phonemeCount(void);
int speechPlayer_ipa_phoneme(int index, char* symbolUtf8, int symbolCapacity, double* values, unsigned long long* fieldMask, unsigned int* classBits);
/* The voice presets of the NVDA driver (reference __init__.py:86-116), by index and by name. */
int speechPlayer_voiceCount(void);          /* presets, then the voices defined with speechPlayer_voiceDefine */
int speechPlayer_voicePresetCount(void);    /* the driver's presets alone: indices 0 .. this - 1 */
const char* speechPlayer_voiceName(int index);
/* Index of a voice by name (-1: none such), and a voice of the caller's own in the presets' form (reference __init__.py:86-116: per
 * parameter an absolute value, a multiplier, or both -- absolute first): entry e sets parameter param[e] (0..46) to absValue[e] unless
 * that is NaN (or absValue NULL), then multiplies it by multiplier[e] unless that is NaN (or multiplier NULL).  Defining a name again
 * replaces its entries; built-in names cannot be redefined.  Returns the voice's index, -1 on bad arguments. */
int speechPlayer_voiceIndex(const char* voiceName);
int speechPlayer_voiceDefine(const char* voiceName, int nEntries, const int* param, const double* absValue, const double* multiplier);
/* reference __init__.py:118-125.  0, or -1 for an unknown voice. */
int speechPlayer_applyVoiceToFrame(speechPlayer_frame_t* frame, const char* voiceName);

const char* speechPlayer_lastError(void);
/* The reference has no error convention (src/speechPlayer.cpp:25-53 return nothing but counts): its
 * speechPlayer_synthesize returns 0 both when the queue has drained and -- here -- when the GPU call failed.
 * This tells the two apart: 0 after a call that succeeded, one of the codes below after one that failed
 * (per calling thread, describes the most recent speechPlayer_* call that can fail). */
#define SPEECHPLAYER_OK 0
#define SPEECHPLAYER_ERR_ARGUMENT 1   /* invalid handle, NULL pointer, inconsistent arrays, limit exceeded */
#define SPEECHPLAYER_ERR_NO_DEVICE 2  /* no HIP device: the engine has no CPU path */
#define SPEECHPLAYER_ERR_HIP 3        /* a HIP runtime call failed (allocation, copy, launch) */
#define SPEECHPLAYER_ERR_TEXT_FRONTEND 4   /* the optional text front-end: eSpeak NG is not installed, or one of its calls failed */
int speechPlayer_lastErrorCode(void);

#ifdef __cplusplus
}
#endif

#endif
