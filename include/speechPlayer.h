/*
 * speechPlayer.h -- the drop-in C-ABI of the MI355X Klatt engine.
 *
 * These five entry points and the frame layout are what the reference exports
 * (reference: src/speechPlayer.h:25-31, src/speechPlayer.def:1-6, src/frame.h:22-43,
 * src/sample.h:18-22) and what its ctypes wrapper binds (speechPlayer.py:48-65).
 * Signatures, units (durations in SAMPLES) and semantics are the reference's; the
 * implementation behind them is the HIP engine in nvspeechplayer_amd/csrc.
 *
 * Handles are small integers cast to void*, so that a prototype-less ctypes caller
 * (the reference wrapper sets no restype, speechPlayer.py:48) survives on LP64.
 */
#ifndef NVSP_AMD_SPEECHPLAYER_H
#define NVSP_AMD_SPEECHPLAYER_H

#ifdef __cplusplus
extern "C" {
#else
#include <stdbool.h>
#endif

typedef double speechPlayer_frameParam_t;

/* 47 doubles, 376 bytes; field order is the ABI (reference src/frame.h:24-42). */
typedef struct {
	speechPlayer_frameParam_t voicePitch;
	speechPlayer_frameParam_t vibratoPitchOffset;
	speechPlayer_frameParam_t vibratoSpeed;
	speechPlayer_frameParam_t voiceTurbulenceAmplitude;
	speechPlayer_frameParam_t glottalOpenQuotient;
	speechPlayer_frameParam_t voiceAmplitude;
	speechPlayer_frameParam_t aspirationAmplitude;
	speechPlayer_frameParam_t cf1, cf2, cf3, cf4, cf5, cf6, cfN0, cfNP;
	speechPlayer_frameParam_t cb1, cb2, cb3, cb4, cb5, cb6, cbN0, cbNP;
	speechPlayer_frameParam_t caNP;
	speechPlayer_frameParam_t fricationAmplitude;
	speechPlayer_frameParam_t pf1, pf2, pf3, pf4, pf5, pf6;
	speechPlayer_frameParam_t pb1, pb2, pb3, pb4, pb5, pb6;
	speechPlayer_frameParam_t pa1, pa2, pa3, pa4, pa5, pa6;
	speechPlayer_frameParam_t parallelBypass;
	speechPlayer_frameParam_t preFormantGain;
	speechPlayer_frameParam_t outputGain;
	speechPlayer_frameParam_t endVoicePitch;
} speechPlayer_frame_t;

#define SPEECHPLAYER_FRAME_NUMPARAMS 47

typedef short sampleVal;
typedef struct {
	sampleVal value;
} sample;

typedef void* speechPlayer_handle_t;

/* replaces reference src/speechPlayer.cpp:25-32 */
speechPlayer_handle_t speechPlayer_initialize(int sampleRate);
/* replaces reference src/speechPlayer.cpp:34-37 (+ src/frame.cpp:90-115): framePtr==NULL queues
 * silence; the frame is copied; fadeDuration is clamped to >= 1; purgeQueue drops pending frames
 * and cuts over from the current interpolated frame. */
void speechPlayer_queueFrame(speechPlayer_handle_t playerHandle, speechPlayer_frame_t* framePtr, unsigned int minFrameDuration, unsigned int fadeDuration, int userIndex, bool purgeQueue);
/* replaces reference src/speechPlayer.cpp:39-41 (+ src/speechWaveGenerator.cpp:197-214): returns the
 * number of samples written; a short count means the queue drained. State persists across calls. */
int speechPlayer_synthesize(speechPlayer_handle_t playerHandle, unsigned int sampleCount, sample* sampleBuf);
/* replaces reference src/speechPlayer.cpp:43-46 (+ src/frame.cpp:117-119) */
int speechPlayer_getLastIndex(speechPlayer_handle_t playerHandle);
/* replaces reference src/speechPlayer.cpp:48-53 */
void speechPlayer_terminate(speechPlayer_handle_t playerHandle);

#ifdef __cplusplus
}
#endif

#endif
