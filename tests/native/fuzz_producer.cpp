// Sanitizer run of the frame producer (nvspeechplayer_amd/csrc/frame_producer.cpp) on the CPU: random symbol soup -- tie bars, stress
// and length marks in any position, unknown symbols, invalid UTF-8, every clause type and voice -- through speechPlayer_ipa_frames and
// speechPlayer_ipa_pack under AddressSanitizer + UBSan (tests/test_ipa_producer.py builds and runs it).  The engine entry points the
// producer calls are stubbed: this links no GPU code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <cmath>
#include "speechPlayer_batch.h"
extern "C" int speechPlayer_batch_sampleRate(speechPlayer_batch_t) { return 22050; }
extern "C" void speechPlayer_internal_setError(int, const char*) {}
// (the producer hands a batch over as records and runs its per-text / per-list loops on the engine's worker threads: here, in two ranges on this thread)
extern "C" int speechPlayer_batch_setRecords(speechPlayer_batch_t, long long nShapes, const speechPlayer_frame_t*, long long nLists, const long long* listStart,
                                             const speechPlayer_frameRecord_t* rec, long long nUtt, const unsigned int* listOf, const unsigned int*)
{
    for (long long u = 0; listOf && u < nUtt; ++u) if ((long long)listOf[u] >= nLists) return -1;
    for (long long k = 0; k < listStart[nLists]; ++k) if (rec[k].shape != SPEECHPLAYER_RECORD_SILENCE && (long long)rec[k].shape >= nShapes) return -1;
    return 0;
}
extern "C" int speechPlayer_node_setRecords(speechPlayer_node_t, long long, const speechPlayer_frame_t*, long long, const long long*, const speechPlayer_frameRecord_t*,
                                            long long, const unsigned int*, const unsigned int*) { return 0; }
extern "C" void speechPlayer_internal_parallel(long long n, long long, void (*fn)(void*, long long, long long), void* ctx)
{
    if (n > 0) { fn(ctx, 0, n / 2); fn(ctx, n / 2, n); }
}
int main() {
    std::mt19937 rng(1);
    const char* alphabet[] = {"a","h","t","\xcd\xa1","\xca\x83","\xcb\x88","\xcb\x8c","\xcb\x90"," ","p","s","z","m","n","l","j","w","\xc9\x91","\xc3\xa6","i","u","#","\xff","\xc9","d","\xca\x92","k","b","\xc9\xb9","\xc5\x8b"};
    const int na = sizeof alphabet / sizeof *alphabet;
    long long total = 0;
    for (int iter = 0; iter < 40000; ++iter) {
        std::string s;
        int len = rng() % 24;
        for (int i = 0; i < len; ++i) s += alphabet[rng() % na];
        const char clause = ".,?!\0"[rng() % 5];
        const char* voices[] = {nullptr, "Adam", "Benjamin", "Caleb", "David", "x"};
        const char* v = voices[rng() % 6];
        long long n = speechPlayer_ipa_frames(s.c_str(), 0.5 + (rng() % 20) / 10.0, 60 + rng() % 200, (rng() % 10) / 10.0, clause, v, nullptr, nullptr, nullptr, nullptr, 0);
        if (n > 0) {
            std::vector<speechPlayer_frame_t> fr(n); std::vector<unsigned char> nu(n); std::vector<double> d(n), f(n);
            long long m = speechPlayer_ipa_frames(s.c_str(), 1.0, 100, 0.5, clause, v, fr.data(), nu.data(), d.data(), f.data(), n);
            if (m != n && !(v && !strcmp(v, "x"))) { printf("mismatch\n"); return 1; }
            total += n;
        }
    }
    // the text front-end's own string handling (no eSpeak here): clause splitting and the replacements on random bytes
    for (int iter = 0; iter < 20000; ++iter) {
        const char* bits[] = {"a", " ", ".", ",", "?", "!", ":", ";", "\t", "\n", "\xc2\xa0", "\xe2\x80\x89", "\xe3\x80\x80", "\xc9\x99", "\xcd\xa1", "l", "\xc9\xaa", "e", "\xca\x8a", "\xff", "\xe2", "\xc2"};
        std::string s;
        int len = rng() % 40;
        for (int i = 0; i < len; ++i) s += bits[rng() % (sizeof bits / sizeof *bits)];
        long long b[64], e[64]; char ty[64]; double pa[64];
        long long nc = speechPlayer_text_clauses(s.c_str(), b, e, ty, pa, 64);
        for (long long i = 0; i < nc && i < 64; ++i)
            if (b[i] < 0 || e[i] > (long long)s.size() || b[i] >= e[i]) { printf("bad clause range\n"); return 1; }
        std::vector<char> out(3 * s.size() + 8);
        long long need = speechPlayer_text_fixups(s.c_str(), out.data(), (long long)out.size());
        if (need < 1 || need > (long long)out.size()) { printf("bad fixups size\n"); return 1; }
    }
    // batch packer with duplicates and empties
    std::vector<const char*> texts;
    for (int i = 0; i < 3000; ++i) texts.push_back(alphabet[i % na]);
    std::vector<long long> start(texts.size() + 1);
    long long tot = speechPlayer_ipa_pack(16000, (long long)texts.size(), texts.data(), 1.0, nullptr, 0.5, nullptr, "Adam", 150.0, start.data(), nullptr, nullptr, nullptr, nullptr, 0);
    std::vector<speechPlayer_frame_t> fr(tot); std::vector<unsigned> mi(tot), fa(tot); std::vector<unsigned char> nu(tot);
    long long tot2 = speechPlayer_ipa_pack(16000, (long long)texts.size(), texts.data(), 1.0, nullptr, 0.5, nullptr, "Adam", 150.0, start.data(), fr.data(), mi.data(), fa.data(), nu.data(), tot);
    // the compact form of the same batch, a voice per text (a defined one among them), and the batch entry points over the stubs
    const int pr[2] = {7, 0}; const double ab[2] = {std::nan(""), 90.0}, mu[2] = {1.1, std::nan("")};
    const int mine = speechPlayer_voiceDefine("fuzz", 2, pr, ab, mu);
    if (mine != speechPlayer_voiceCount() - 1 || speechPlayer_voiceIndex("fuzz") != mine || speechPlayer_voiceDefine("Adam", 0, nullptr, nullptr, nullptr) != -1) { printf("voiceDefine\n"); return 1; }
    std::vector<int> voiceOf(texts.size());
    for (size_t i = 0; i < texts.size(); ++i) voiceOf[i] = (int)(i % (size_t)(mine + 2)) - 1;
    speechPlayer_records_t ro = speechPlayer_ipa_records(16000, (long long)texts.size(), texts.data(), 1.0, nullptr, 0.5, nullptr, voiceOf.data(), nullptr, 150.0);
    speechPlayer_recordsView_t view;
    if (!ro || speechPlayer_records_view(ro, &view) || view.nUtterances != (long long)texts.size() || view.listStart[view.nLists] != view.nRecords) { printf("records\n"); return 1; }
    for (long long k = 0; k < view.nRecords; ++k)
        if (view.records[k].shape != SPEECHPLAYER_RECORD_SILENCE && (long long)view.records[k].shape >= view.nShapes) { printf("record shape\n"); return 1; }
    speechPlayer_records_free(ro);
    voiceOf[5] = mine + 1;
    if (speechPlayer_ipa_records(16000, (long long)texts.size(), texts.data(), 1.0, nullptr, 0.5, nullptr, voiceOf.data(), nullptr, 150.0)) { printf("bad voice accepted\n"); return 1; }
    voiceOf[5] = -1;
    if (speechPlayer_batch_setIpa(nullptr, (long long)texts.size(), texts.data(), 1.0, nullptr, 0.5, nullptr, "Benjamin", 150.0, nullptr) != 0 ||
        speechPlayer_batch_setIpaVoices(nullptr, (long long)texts.size(), texts.data(), 1.0, nullptr, 0.5, nullptr, voiceOf.data(), -1.0, nullptr) != 0) { printf("setIpa\n"); return 1; }
    printf("ok %lld frames fuzzed, pack %lld %lld\n", total, tot, tot2);
    return tot == tot2 ? 0 : 1;
}
