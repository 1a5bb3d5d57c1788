/* A plain-C consumer of the five reference entry points (include/speechPlayer.h), linked against the
 * engine library the way a C/C++ host of the reference's speechPlayer.dll would be.
 *   c_client <frame.bin> <out.pcm> <sampleRate> <minSamples> <fadeSamples> <pull>
 * frame.bin holds one speechPlayer_frame_t (47 doubles).  Queues it (index 7), a NULL frame (fade only), pulls
 * in chunks of `pull` samples until the player drains, writes the PCM, prints "<samples> <lastIndex>". */
#include <stdio.h>
#include <stdlib.h>
#include "speechPlayer.h"

int main(int argc, char** argv)
{
    speechPlayer_frame_t frame;
    FILE* f;
    speechPlayer_handle_t h;
    sample* buf;
    long total = 0;
    int got, rate, pull;
    unsigned minSamples, fadeSamples;
    if (argc != 7) return 2;
    f = fopen(argv[1], "rb");
    if (!f || fread(&frame, sizeof frame, 1, f) != 1) return 3;
    fclose(f);
    rate = atoi(argv[3]); minSamples = (unsigned)atoi(argv[4]); fadeSamples = (unsigned)atoi(argv[5]); pull = atoi(argv[6]);
    h = speechPlayer_initialize(rate);
    if (!h) return 4;
    speechPlayer_queueFrame(h, &frame, minSamples, fadeSamples, 7, false);
    speechPlayer_queueFrame(h, NULL, fadeSamples, fadeSamples, -1, false);
    buf = (sample*)malloc(sizeof(sample) * (size_t)pull);
    f = fopen(argv[2], "wb");
    if (!buf || !f) return 5;
    do {
        got = speechPlayer_synthesize(h, (unsigned)pull, buf);
        if (got < 0 || got > pull) return 6;
        fwrite(buf, sizeof(sample), (size_t)got, f);
        total += got;
    } while (got == pull);
    fclose(f);
    printf("%ld %d\n", total, speechPlayer_getLastIndex(h));
    speechPlayer_terminate(h);
    free(buf);
    return 0;
}
