// frame_producer.cpp -- IPA text -> frame streams, on the host, for whole batches.
//
// The producer that feeds the synthesis hot path: the native counterpart of the reference's
// ipa.generateFramesAndTiming (reference ipa.py:336-353) together with the NVDA driver's voice presets
// (reference nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:86-125).  For the same text and settings it yields
// the same (frame | silence, duration ms, fade ms) triples, value for value (tests/test_ipa_producer.py compares
// it with streams captured from the reference).
//
// It is written for batches, not as a generator: an utterance is three flat passes over an array of fixed-size
// segment records (no per-phoneme dictionaries), all table-driven --
//   lex       code points -> segments (table row, class bits, prosodic bits), with the inserted pre-stop gaps and
//             post-stop aspirations                                              (behaviour of ipa.py:39-119)
//   colour    /h/-like segments borrow the fields they lack from a neighbour      (ipa.py:121-133)
//   time      duration and fade from the class bits                               (ipa.py:135-184)
//   contour   the clause is cut into spans (pre-head, head runs, nucleus, tail); a span is a linear pitch glide over
//             the voiced time it covers                                           (ipa.py:186-334)
// -- and a batch call builds each distinct (text, clause, base pitch) once and instances it: BASELINE's 65 536-utterance
// configuration has 512 distinct streams.  The constant tables (phoneme rows, intonation rows, voice presets) are
// numbers dumped from the reference's tables into frame_tables.inc by tests/golden/make_golden.py.
//
// Plain host C++: no HIP here.  Everything is exported through the C-ABI of include/speechPlayer_batch.h.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>

#include "../../include/speechPlayer_batch.h"

namespace {

constexpr int kParams = SPEECHPLAYER_FRAME_NUMPARAMS;
constexpr int P_VOICEPITCH = 0, P_PREGAIN = 44, P_OUTGAIN = 45, P_ENDPITCH = 46;

// ---- tables ---------------------------------------------------------------------------------
enum : uint8_t { C_VOWEL = 1, C_VOICED = 2, C_NASAL = 4, C_STOP = 8, C_LIQUID = 16, C_SEMIVOWEL = 32, C_AFFRICATE = 64, C_COPY_ADJACENT = 128 };

struct PhonemeRow {
    uint32_t symbol[3];     // code points of the table key
    uint8_t nSymbol;
    uint8_t cls;            // C_* bits
    uint64_t mask;          // bit k: the entry sets parameter k
    double value[kParams];
};
struct IntonationRow {
    char clause;
    int preHeadStart, preHeadEnd, headExtendFrom, headStart, headEnd, headStressEndDelta, headUnstressedRunStartDelta,
        headUnstressedRunEndDelta, nucleus0Start, nucleus0End, nucleusStart, nucleusEnd, tailStart, tailEnd;
    int nHeadSteps;
    int headSteps[12];
};
struct VoiceEntry { int param; int hasAbs; double abs; int hasMul; double mul; };
struct VoiceRow { const char* name; int nEntries; VoiceEntry entries[16]; };

#include "frame_tables.inc"

constexpr int kNumPhonemes = (int)(sizeof kPhonemeRows / sizeof kPhonemeRows[0]);
constexpr int kNumVoices = (int)(sizeof kVoiceRows / sizeof kVoiceRows[0]);

constexpr uint32_t CP_PRIMARY = 0x2C8, CP_SECONDARY = 0x2CC, CP_LONG = 0x2D0, CP_TIE = 0x361;

// symbol -> row, for keys of one, two and three code points
struct SymbolIndex {
    std::unordered_map<uint64_t, int> byKey;
    static uint64_t key(const uint32_t* cp, int n)
    {
        uint64_t k = (uint64_t)n;
        for (int i = 0; i < n; ++i) k = k * 0x200000ull + cp[i];     // code points are below 2^21
        return k;
    }
    SymbolIndex()
    {
        for (int r = 0; r < kNumPhonemes; ++r) byKey.emplace(key(kPhonemeRows[r].symbol, kPhonemeRows[r].nSymbol), r);
    }
    int find(const uint32_t* cp, int n) const
    {
        auto it = byKey.find(key(cp, n));
        return it == byKey.end() ? -1 : it->second;
    }
};
const SymbolIndex& symbols()
{
    static const SymbolIndex idx;
    return idx;
}
int row_of_h()
{
    static const int r = [] { const uint32_t h = 'h'; return symbols().find(&h, 1); }();
    return r;
}

// ---- an utterance as flat records --------------------------------------------------------------
enum : uint16_t { S_TIED_TO = 1, S_TIED_FROM = 2, S_LONG = 4, S_WORD_START = 8, S_SYLLABLE_START = 16, S_GAP = 32, S_PUFF = 64 };

struct Segment {
    int row;             // table row; -1 for a pre-stop gap (silence)
    int comp;            // the segment's values: the row itself, or the row with the fields it took over from a neighbour (Composites)
    uint8_t cls;
    uint8_t stress;      // 0, 1 primary, 2 secondary; meaningful on syllable heads
    uint16_t pros;       // S_* bits
    double duration, fade;
};
struct Utterance { std::vector<Segment> seg; };

// A COMPOSITE is what a segment's parameter values are before pitch and voice: a table row, or a copy-adjacent row completed with
// the fields of a neighbour's composite (reference ipa.py:121-133).  Composites are identified exactly -- id below kNumPhonemes: the
// row; above: (base composite, donor composite), a pure function of the two -- so a frame's 45 non-pitch values are known by a small
// integer, and everything behind the producer (the device's frame expansion, the track planner) recognises equal frames by that
// integer instead of by comparing or hashing 376 bytes.
struct Composite { uint64_t mask; double value[kParams]; };
class Composites {
public:
    int count()
    {
        std::lock_guard<std::mutex> g(mu_);
        return kNumPhonemes + (int)extra_.size();
    }
    void get(int id, Composite& out)
    {
        if (id < kNumPhonemes) { out.mask = kPhonemeRows[id].mask; memcpy(out.value, kPhonemeRows[id].value, sizeof out.value); return; }
        std::lock_guard<std::mutex> g(mu_);
        out = extra_[(size_t)(id - kNumPhonemes)];
    }
    // `base` with every field it lacks taken from `donor`; `base` itself when the donor adds nothing
    int combine(int base, int donor)
    {
        Composite b, d;
        get(base, b); get(donor, d);
        const uint64_t take = d.mask & ~b.mask;
        if (!take) return base;
        const uint64_t key = (uint64_t)(uint32_t)base << 32 | (uint32_t)donor;
        std::lock_guard<std::mutex> g(mu_);
        auto it = byKey_.find(key);
        if (it != byKey_.end()) return it->second;
        for (int k = 0; k < kParams; ++k) if (take >> k & 1) b.value[k] = d.value[k];
        b.mask |= take;
        extra_.push_back(b);
        const int id = kNumPhonemes + (int)extra_.size() - 1;
        byKey_.emplace(key, id);
        return id;
    }
private:
    std::mutex mu_;
    std::vector<Composite> extra_;
    std::unordered_map<uint64_t, int> byKey_;
};

void decode_utf8(const char* text, std::vector<uint32_t>& out)
{
    out.clear();
    const unsigned char* p = reinterpret_cast<const unsigned char*>(text ? text : "");
    while (*p) {
        uint32_t c = *p;
        int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
        if (extra < 0) { out.push_back(0xFFFD); ++p; continue; }       // stray byte: an unknown symbol
        if (extra) c &= 0x3F >> extra;
        ++p;
        int k = 0;
        for (; k < extra && (*p & 0xC0) == 0x80; ++k, ++p) c = (c << 6) | (*p & 0x3F);
        out.push_back(k == extra ? c : 0xFFFD);
    }
}

Segment make_segment(int row)
{
    Segment s;
    s.row = row; s.comp = row; s.cls = row < 0 ? 0 : kPhonemeRows[row].cls; s.stress = 0; s.pros = 0;
    s.duration = s.fade = 0.0;
    return s;
}

// Code points -> segments.  A stress mark waits for the next symbol the table knows; a tie bar joins two symbols into
// one row when the table has the pair, else it marks its neighbours as tied; a length mark prefers the lengthened row.
// Between a voiceless stop and a voiced continuant goes a short aspiration (/h/); before a stop or affricate that
// does not carry the stress mark itself goes a gap.
void lex(const std::vector<uint32_t>& cp, Utterance& u)
{
    u.seg.clear();
    const SymbolIndex& sym = symbols();
    const size_t n = cp.size();
    int waitingStress = 0;
    bool wordBoundary = true;
    int last = -1, head = -1;              // indices into u.seg: the previous symbol's segment, the current syllable's head
    for (size_t pos = 0; pos < n;) {
        const uint32_t c = cp[pos];
        if (c == CP_PRIMARY || c == CP_SECONDARY) { waitingStress = (c == CP_PRIMARY) ? 1 : 2; ++pos; continue; }
        const uint32_t next = pos + 1 < n ? cp[pos + 1] : 0;
        const bool longMark = next == CP_LONG, tieAfter = next == CP_TIE, tieBefore = pos > 0 && cp[pos - 1] == CP_TIE;
        int row = -1;
        size_t step = 1;
        if (tieAfter) {
            row = pos + 2 < n ? sym.find(&cp[pos], 3) : -1;
            step = row >= 0 ? 3 : 2;
        } else if (longMark) {
            row = sym.find(&cp[pos], 2);
            step = 2;
        }
        if (row < 0) row = sym.find(&c, 1);
        pos += step;
        if (c == ' ') { wordBoundary = true; continue; }
        if (row < 0) continue;                                          // a symbol the table does not know
        Segment s = make_segment(row);
        if (tieBefore) s.pros |= S_TIED_FROM; else if (tieAfter) s.pros |= S_TIED_TO;
        if (longMark) s.pros |= S_LONG;
        const int stress = waitingStress;
        waitingStress = 0;
        const bool vowel = s.cls & C_VOWEL;
        bool headIsNew = false;            // the syllable head is the segment being built (not yet in u.seg)
        if (last >= 0 && !(u.seg[last].cls & C_VOWEL) && vowel) {
            u.seg[last].pros |= S_SYLLABLE_START;                       // consonant + vowel: the consonant opens the syllable
            head = last;
        } else if (stress == 1 && last >= 0 && (u.seg[last].cls & C_VOWEL)) {
            s.pros |= S_SYLLABLE_START;                                 // a stressed symbol after a vowel opens one itself
            headIsNew = true;
        }
        if (last >= 0 && (u.seg[last].cls & C_STOP) && !(u.seg[last].cls & C_VOICED) && (s.cls & C_VOICED) &&
            !(s.cls & (C_STOP | C_AFFRICATE))) {
            Segment puff = make_segment(row_of_h());
            puff.pros |= S_PUFF;
            u.seg.push_back(puff);
        }
        if (wordBoundary) {
            wordBoundary = false;
            s.pros |= S_WORD_START | S_SYLLABLE_START;
            headIsNew = true;
        }
        if (stress) {
            if (headIsNew) s.stress = (uint8_t)stress; else if (head >= 0) u.seg[head].stress = (uint8_t)stress;
        } else if (s.cls & (C_STOP | C_AFFRICATE)) {
            Segment gap = make_segment(-1);
            gap.pros |= S_GAP;
            u.seg.push_back(gap);
        }
        u.seg.push_back(s);
        last = (int)u.seg.size() - 1;
        if (headIsNew) head = last;
    }
}

// A segment of the copy-adjacent class takes every field it lacks from the segment after it -- or, when that is a gap or
// the end, from the one before -- as that neighbour stands at this point of a left-to-right pass.
void colour(Utterance& u, Composites& comps)
{
    const int n = (int)u.seg.size();
    for (int i = 0; i < n; ++i) {
        Segment& s = u.seg[i];
        if (!(s.cls & C_COPY_ADJACENT)) continue;
        const int from = (i + 1 < n && !(u.seg[i + 1].pros & S_GAP)) ? i + 1 : i - 1;
        if (from < 0 || u.seg[from].row < 0) continue;                  // (a gap has no fields to give)
        s.comp = comps.combine(s.comp, u.seg[from].comp);
    }
}

// Duration and fade in milliseconds, by class.  The tempo of a syllable depends on its head's stress.
void time_segments(Utterance& u, double baseSpeed)
{
    const int n = (int)u.seg.size();
    int syllableStress = 0;
    double speed = baseSpeed;
    for (int i = 0; i < n; ++i) {
        Segment& s = u.seg[i];
        const Segment* prev = i > 0 ? &u.seg[i - 1] : nullptr;
        const Segment* next = i + 1 < n ? &u.seg[i + 1] : nullptr;
        const bool opens = s.pros & S_SYLLABLE_START;
        if (opens) {
            syllableStress = s.stress;
            speed = syllableStress == 1 ? baseSpeed / 1.4 : syllableStress ? baseSpeed / 1.1 : baseSpeed;
        }
        double dur = 60.0 / speed, fade = 10.0 / speed;
        const uint8_t c = s.cls;
        if (s.pros & S_GAP) dur = 41.0 / speed;
        else if (s.pros & S_PUFF) dur = 20.0 / speed;
        else if (c & C_STOP) { dur = std::fmin(6.0 / speed, 6.0); fade = 0.001; }
        else if (c & C_AFFRICATE) { dur = 24.0 / speed; fade = 0.001; }
        else if (!(c & C_VOICED)) dur = 45.0 / speed;
        else if (c & C_VOWEL) {
            if (prev && (prev->cls & (C_LIQUID | C_SEMIVOWEL))) fade = 25.0 / speed;
            if (s.pros & S_TIED_TO) dur = 40.0 / speed;
            else if (s.pros & S_TIED_FROM) { dur = 20.0 / speed; fade = 20.0 / speed; }
            else if (!syllableStress && !opens && next && !(next->pros & S_WORD_START) && (next->cls & (C_LIQUID | C_NASAL)))
                dur = (next->cls & C_LIQUID) ? 30.0 / speed : 40.0 / speed;
        } else {
            dur = 30.0 / speed;
            if (c & (C_LIQUID | C_SEMIVOWEL)) fade = 20.0 / speed;
        }
        if (s.pros & S_LONG) dur *= 1.05;
        s.duration = dur; s.fade = fade;
    }
}

// A span [a, b) of segments whose pitch glides from `from` to `to` (percent of the inflection range; 50 = base pitch),
// linearly over the voiced time inside the span.
struct Span { int a, b; double from, to; };

const IntonationRow& intonation_for(int clause)
{
    for (const IntonationRow& r : kIntonationRows)
        if (r.clause == clause) return r;
    return kIntonationRows[0];        // no clause type: the statement contour
}

// Cut the clause into spans.  The pre-head runs up to the first primary-stressed syllable; the nucleus is the last
// primary-stressed syllable and the tail what follows it; in between, the head steps down: each stressed syllable
// starts at the next step of the table and falls by a fixed amount, the unstressed run behind it continues below.
void contour_spans(const Utterance& u, int clause, std::vector<Span>& spans)
{
    spans.clear();
    const IntonationRow& t = intonation_for(clause);
    const int n = (int)u.seg.size();
    auto primary_head = [&](int i) { return (u.seg[i].pros & S_SYLLABLE_START) && u.seg[i].stress == 1; };
    int preHeadEnd = n;
    for (int i = 0; i < n; ++i) if (primary_head(i)) { preHeadEnd = i; break; }
    if (preHeadEnd > 0) spans.push_back({0, preHeadEnd, (double)t.preHeadStart, (double)t.preHeadEnd});
    int nucleusStart = n, nucleusEnd = n, tailStart = n;
    for (int i = n - 1; i >= preHeadEnd; --i) {
        if (!(u.seg[i].pros & S_SYLLABLE_START)) continue;
        if (u.seg[i].stress == 1) { nucleusStart = i; break; }
        nucleusEnd = tailStart = i;
    }
    const bool tail = n > tailStart;
    if (tail) spans.push_back({tailStart, n, (double)t.tailStart, (double)t.tailEnd});
    if (nucleusEnd > nucleusStart)
        spans.push_back(tail ? Span{nucleusStart, nucleusEnd, (double)t.nucleusStart, (double)t.nucleusEnd}
                             : Span{nucleusStart, nucleusEnd, (double)t.nucleus0Start, (double)t.nucleus0End});
    if (preHeadEnd >= nucleusStart) return;
    // the head: syllable heads between the first primary stress and the nucleus, the nucleus closing the last run
    int step = 0, stressedFrom = -1, runFrom = -1;
    double stressEnd = 0.0;
    for (int i = preHeadEnd; i <= nucleusStart; ++i) {
        if (!(u.seg[i].pros & S_SYLLABLE_START)) continue;
        if (stressedFrom >= 0) {
            const int percent = t.headSteps[step];
            ++step;
            if (step == t.nHeadSteps) step = t.headExtendFrom;                  // past the table: cycle through its tail
            const double start = t.headEnd + (((t.headStart - t.headEnd) / 100.0) * percent);
            stressEnd = start + t.headStressEndDelta;
            spans.push_back({stressedFrom, i, start, stressEnd});
            stressedFrom = -1;
        }
        if (u.seg[i].stress == 1) {
            if (runFrom >= 0) {
                spans.push_back({runFrom, i, stressEnd + t.headUnstressedRunStartDelta, stressEnd + t.headUnstressedRunEndDelta});
                runFrom = -1;
            }
            stressedFrom = i;
        } else if (runFrom < 0) {
            runFrom = i;
        }
    }
}

// the pitch pair of every segment a span covers (a later span overwrites an earlier one, as in the reference's loop)
struct Pitches {
    std::vector<double> pitch, endPitch;
    std::vector<unsigned char> has;
    void reset(size_t n) { pitch.assign(n, 0.0); endPitch.assign(n, 0.0); has.assign(n, 0); }
};
void apply_spans(const Utterance& u, const std::vector<Span>& spans, double basePitch, double inflection, Pitches& out)
{
    out.reset(u.seg.size());
    for (const Span& sp : spans) {
        const double p0 = basePitch * std::pow(2.0, ((sp.from - 50) / 50.0) * inflection);
        const double p1 = basePitch * std::pow(2.0, ((sp.to - 50) / 50.0) * inflection);
        double voiced = 0.0;
        for (int i = sp.a; i < sp.b; ++i) if (u.seg[i].cls & C_VOICED) voiced += u.seg[i].duration;
        const double delta = p1 - p0;
        double done = 0.0, cur = p0;
        for (int i = sp.a; i < sp.b; ++i) {
            const Segment& s = u.seg[i];
            out.pitch[i] = cur;
            if (s.cls & C_VOICED) {
                done += s.duration;
                cur = p0 + (delta * (done / voiced));
            }
            out.endPitch[i] = cur;
            out.has[i] = 1;
        }
    }
}

// ---- voices: the driver's presets (frame_tables.inc) and the ones a caller defines -------------------------------
struct VoiceDef { std::string name; std::vector<VoiceEntry> entries; };
std::mutex g_voiceMutex;
std::vector<std::unique_ptr<VoiceDef>> g_definedVoices;       // indices kNumVoices .. ; never shrinks: names handed out stay valid

// index of a voice by name, -1 if there is none.  (The reference's table has one key with a trailing blank, "Caleb ": the name
// without it is accepted too.)
int voice_index(const char* name)
{
    if (!name || !*name) return -1;
    for (int i = 0; i < kNumVoices; ++i)
        if (!strcmp(kVoiceRows[i].name, name)) return i;
    const size_t len = strlen(name);
    for (int i = 0; i < kNumVoices; ++i) {
        const char* v = kVoiceRows[i].name;
        if (strlen(v) == len + 1 && !strncmp(v, name, len) && v[len] == ' ') return i;
    }
    std::lock_guard<std::mutex> g(g_voiceMutex);
    for (size_t i = 0; i < g_definedVoices.size(); ++i)
        if (g_definedVoices[i]->name == name) return kNumVoices + (int)i;
    return -1;
}
int voice_count()
{
    std::lock_guard<std::mutex> g(g_voiceMutex);
    return kNumVoices + (int)g_definedVoices.size();
}
// the entries of voice `index` (a copy: a producer call works on what the voice was when the call started); false if out of range
bool voice_entries(int index, std::vector<VoiceEntry>& out)
{
    out.clear();
    if (index < 0) return false;
    if (index < kNumVoices) { out.assign(kVoiceRows[index].entries, kVoiceRows[index].entries + kVoiceRows[index].nEntries); return true; }
    std::lock_guard<std::mutex> g(g_voiceMutex);
    const size_t i = (size_t)(index - kNumVoices);
    if (i >= g_definedVoices.size()) return false;
    out = g_definedVoices[i]->entries;
    return true;
}

// absolute value first, then the multiplier, parameter by parameter in the entries' order (reference __init__.py:118-125)
void apply_voice(double* frame, const std::vector<VoiceEntry>& v)
{
    for (const VoiceEntry& x : v) {
        if (x.hasAbs) frame[x.param] = x.abs;
        if (x.hasMul) frame[x.param] = frame[x.param] * x.mul;
    }
}
// the same for ONE parameter whose value is `value` before the voice (an entry touches nothing but its own parameter)
double apply_voice_param(const std::vector<VoiceEntry>& v, int param, double value)
{
    for (const VoiceEntry& x : v) {
        if (x.param != param) continue;
        if (x.hasAbs) value = x.abs;
        if (x.hasMul) value = value * x.mul;
    }
    return value;
}

// milliseconds -> samples as the reference wrapper converts them (reference speechPlayer.py:53): int(ms * (sr / 1000.0))
unsigned int ms_to_samples(double ms, int sampleRate)
{
    const double v = ms * (sampleRate / 1000.0);
    return v <= 0.0 ? 0u : (v >= 4294967295.0 ? 4294967295u : (unsigned int)(long long)v);
}

}  // namespace

extern "C" {
// klatt_engine.hip: fn(ctx, a, e) over [0, n) cut into ranges, on the engine's worker threads (the caller's thread takes part)
void speechPlayer_internal_parallel(long long n, long long grain, void (*fn)(void* ctx, long long a, long long e), void* ctx);
}

namespace {

template <class F>
void parallel_for(long long n, long long grain, F body)
{
    speechPlayer_internal_parallel(n, grain, [](void* ctx, long long a, long long e) { (*static_cast<F*>(ctx))(a, e); }, &body);
}

// ---- a batch in compact form ----------------------------------------------------------------------------------------
// What the reference's generator yields frame by frame (ipa.py:336-353: a 47-field Frame, a duration, a fade), a batch call yields as
//   shapes   [nVoices x nComposites] frames: every (voice, composite) pair's 45 non-pitch values, voice applied (a few hundred rows)
//   records  32 bytes per frame: the two pitches (voice applied), the shape's row, the durations in samples
//   lists    one per DISTINCT (text, clause, base pitch, voice); listOf[u]: the list utterance u speaks
// -- include/speechPlayer_batch.h, speechPlayer_batch_setRecords: the device builds the 376-byte frames from it (klatt_expand_frames),
// and utterances that speak the same list share its frames in HBM.
struct PackArgs {
    int sampleRate = 22050;
    long long nTexts = 0;
    const char* const* texts = nullptr;
    double speed = 1.0;
    const double* basePitch = nullptr;       // [nTexts] or NULL (100 Hz)
    double inflection = 0.5;
    const char* clauseTypes = nullptr;       // [nTexts] or NULL (none)
    const int* voiceOf = nullptr;            // [nTexts] voice index per text (-1: none), or NULL: voiceAll
    int voiceAll = -1;
    double tailMs = -1.0;                    // silence after every utterance (negative: none)
};
struct Compact {
    long long nComposites = 0;
    int nVoices = 0;
    std::vector<speechPlayer_frame_t> shapes;
    std::vector<long long> listStart;
    std::vector<speechPlayer_frameRecord_t> records;
    std::vector<uint32_t> listOf;
    std::vector<double> durationMs, fadeMs;          // per record (keepMs)
    long long frames_of(long long u) const { const uint32_t l = listOf[(size_t)u]; return listStart[l + 1] - listStart[l]; }
};

struct Timed {                                        // a text, lexed, coloured and timed: everything that does not depend on pitch or voice
    Utterance u;
    std::vector<unsigned int> minSamples, fadeSamples;
};

struct ListKey {
    int text, clause, slot;
    uint64_t pitchBits;
    bool operator==(const ListKey& o) const { return text == o.text && clause == o.clause && slot == o.slot && pitchBits == o.pitchBits; }
};
struct ListKeyHash {
    size_t operator()(const ListKey& k) const
    {
        uint64_t h = k.pitchBits * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h += (uint64_t)(uint32_t)k.text * 0xC2B2AE3D27D4EB4Full + (uint64_t)(uint32_t)k.clause * 0x165667B19E3779F9ull + (uint64_t)(uint32_t)k.slot;
        h *= 0xFF51AFD7ED558CCDull;
        return (size_t)(h ^ (h >> 32));
    }
};

// 0, or -1 (bad voice index).  sizeOnly: listOf and listStart alone (no shapes, no records).
int build_compact(const PackArgs& a, bool sizeOnly, bool keepMs, Compact& out)
{
    const long long n = a.nTexts;
    const bool tail = a.tailMs >= 0.0;
    // ---- every utterance's list: distinct texts (by pointer first: a batch repeats its strings), distinct (text, clause, pitch, voice) ----
    std::vector<const char*> textPtr;                 // text id -> the string
    std::unordered_map<const char*, int> textByPtr;
    std::unordered_map<std::string, int> textByContent;
    std::vector<int> voiceOfSlot;                     // slot -> voice index (-1: none)
    std::unordered_map<int, int> slotOfVoice;
    std::vector<ListKey> lists;
    std::unordered_map<ListKey, uint32_t, ListKeyHash> listByKey;
    out.listOf.resize((size_t)n);
    const int nVoicesKnown = voice_count();
    const char* lastPtr = nullptr; int lastText = -1;
    for (long long i = 0; i < n; ++i) {
        const char* t = a.texts[i] ? a.texts[i] : "";
        int text;
        if (t == lastPtr) text = lastText;
        else {
            auto ip = textByPtr.find(t);
            if (ip != textByPtr.end()) text = ip->second;
            else {
                auto ic = textByContent.find(t);
                if (ic == textByContent.end()) { ic = textByContent.emplace(t, (int)textPtr.size()).first; textPtr.push_back(t); }
                text = ic->second;
                textByPtr.emplace(t, text);
            }
            lastPtr = t; lastText = text;
        }
        const int voice = a.voiceOf ? a.voiceOf[i] : a.voiceAll;
        if (voice < -1 || voice >= nVoicesKnown) return -1;
        auto is = slotOfVoice.find(voice);
        if (is == slotOfVoice.end()) { is = slotOfVoice.emplace(voice, (int)voiceOfSlot.size()).first; voiceOfSlot.push_back(voice); }
        const double pitch = a.basePitch ? a.basePitch[i] : 100.0;
        ListKey key{text, a.clauseTypes ? (int)(unsigned char)a.clauseTypes[i] : 0, is->second, 0};
        memcpy(&key.pitchBits, &pitch, 8);
        auto il = listByKey.find(key);
        if (il == listByKey.end()) { il = listByKey.emplace(key, (uint32_t)lists.size()).first; lists.push_back(key); }
        out.listOf[(size_t)i] = il->second;
    }
    // ---- every distinct text: segments, colours, durations (in parallel: nothing here depends on another text but the composite ids) ----
    Composites comps;
    std::vector<Timed> timed(textPtr.size());
    parallel_for((long long)textPtr.size(), 64, [&](long long ta, long long te) {
        std::vector<uint32_t> cp;
        for (long long t = ta; t < te; ++t) {
            Timed& x = timed[(size_t)t];
            decode_utf8(textPtr[(size_t)t], cp);
            lex(cp, x.u);
            colour(x.u, comps);
            time_segments(x.u, a.speed);
            const size_t ns = x.u.seg.size();
            x.minSamples.resize(ns); x.fadeSamples.resize(ns);
            for (size_t k = 0; k < ns; ++k) {
                x.minSamples[k] = ms_to_samples(x.u.seg[k].duration, a.sampleRate);
                x.fadeSamples[k] = ms_to_samples(x.u.seg[k].fade, a.sampleRate);
            }
        }
    });
    const size_t nLists = lists.size();
    out.listStart.assign(nLists + 1, 0);
    for (size_t l = 0; l < nLists; ++l) out.listStart[l + 1] = out.listStart[l] + (long long)timed[(size_t)lists[l].text].u.seg.size() + (tail ? 1 : 0);
    if (sizeOnly) return 0;
    // ---- the shape table: one row per (voice, composite) ----
    const int C = comps.count(), V = (int)voiceOfSlot.size();
    out.nComposites = C; out.nVoices = V;
    std::vector<std::vector<VoiceEntry>> voices((size_t)V);
    for (int s = 0; s < V; ++s)
        if (voiceOfSlot[(size_t)s] >= 0 && !voice_entries(voiceOfSlot[(size_t)s], voices[(size_t)s])) return -1;
    out.shapes.resize((size_t)C * (size_t)std::max(V, 0));
    for (int s = 0; s < V; ++s)
        for (int c = 0; c < C; ++c) {
            double* f = reinterpret_cast<double*>(&out.shapes[(size_t)s * C + c]);
            Composite co;
            comps.get(c, co);
            for (int k = 0; k < kParams; ++k) f[k] = 0.0;
            f[P_PREGAIN] = 1.0;                 // reference ipa.py:349-351
            f[P_OUTGAIN] = 2.0;
            for (int k = 0; k < kParams; ++k) if (co.mask >> k & 1) f[k] = co.value[k];
            apply_voice(f, voices[(size_t)s]);  // (parameters 0 and 46 of a row serve the segments no span covers)
        }
    // ---- the records, list by list ----
    const size_t nRec = (size_t)out.listStart[nLists];
    out.records.resize(nRec);
    if (keepMs) { out.durationMs.assign(nRec, 0.0); out.fadeMs.assign(nRec, 0.0); }
    const unsigned int tailSamples = tail ? ms_to_samples(a.tailMs, a.sampleRate) : 0u;
    parallel_for((long long)nLists, 16, [&](long long la, long long le) {
        std::vector<Span> spans;
        Pitches pt;
        for (long long l = la; l < le; ++l) {
            const ListKey& key = lists[(size_t)l];
            const Timed& x = timed[(size_t)key.text];
            const std::vector<VoiceEntry>& voice = voices[(size_t)key.slot];
            double basePitch;
            memcpy(&basePitch, &key.pitchBits, 8);
            const size_t ns = x.u.seg.size();
            if (ns) {
                contour_spans(x.u, key.clause, spans);
                apply_spans(x.u, spans, basePitch, a.inflection, pt);
            }
            speechPlayer_frameRecord_t* r = out.records.data() + out.listStart[(size_t)l];
            for (size_t k = 0; k < ns; ++k) {
                const Segment& s = x.u.seg[k];
                r[k].minFrameDuration = x.minSamples[k]; r[k].fadeDuration = x.fadeSamples[k]; r[k].userIndex = -1;
                if (keepMs) { out.durationMs[(size_t)out.listStart[(size_t)l] + k] = s.duration; out.fadeMs[(size_t)out.listStart[(size_t)l] + k] = s.fade; }
                if (s.pros & S_GAP) { r[k].shape = SPEECHPLAYER_RECORD_SILENCE; r[k].voicePitch = 0.0; r[k].endVoicePitch = 0.0; continue; }
                const uint32_t shape = (uint32_t)((size_t)key.slot * C + s.comp);
                r[k].shape = shape;
                const double* row = reinterpret_cast<const double*>(out.shapes.data() + shape);
                r[k].voicePitch = pt.has[k] ? apply_voice_param(voice, P_VOICEPITCH, pt.pitch[k]) : row[P_VOICEPITCH];
                r[k].endVoicePitch = pt.has[k] ? apply_voice_param(voice, P_ENDPITCH, pt.endPitch[k]) : row[P_ENDPITCH];
            }
            if (tail) {                      // silence after the utterance, as reference test_speakIpa.py:27 queues it
                speechPlayer_frameRecord_t& t = r[ns];
                t.shape = SPEECHPLAYER_RECORD_SILENCE; t.voicePitch = 0.0; t.endVoicePitch = 0.0;
                t.minFrameDuration = tailSamples; t.fadeDuration = 0; t.userIndex = -1;
                if (keepMs) { out.durationMs[(size_t)out.listStart[(size_t)l] + ns] = a.tailMs; out.fadeMs[(size_t)out.listStart[(size_t)l] + ns] = 0.0; }
            }
        }
    });
    return 0;
}

// record -> the 47-value frame it stands for (what klatt_expand_frames does on the device); silence: zeros
void expand_record(const Compact& c, const speechPlayer_frameRecord_t& r, speechPlayer_frame_t* out, unsigned char* isNull)
{
    if (r.shape == SPEECHPLAYER_RECORD_SILENCE) { memset(out, 0, sizeof *out); *isNull = 1; return; }
    *out = c.shapes[r.shape];
    out->voicePitch = r.voicePitch;
    out->endVoicePitch = r.endVoicePitch;
    *isNull = 0;
}

// the clause types the intonation table knows (reference ipa.py:207-276; the reference raises KeyError for any other), and 0: none
bool clause_known(int clause)
{
    if (clause == 0) return true;
    for (const IntonationRow& r : kIntonationRows)
        if (r.clause == clause) return true;
    return false;
}

void set_producer_error(const char* msg);

// nothing a host allocation throws crosses the C ABI
template <class R, class F>
R producer_call(const char* what, R fail, F body)
{
    try {
        return body();
    } catch (const std::exception& e) {
        set_producer_error((std::string(what) + ": " + e.what()).c_str());
        return fail;
    }
}

}  // namespace

extern "C" {

void speechPlayer_internal_setError(int code, const char* message);      // klatt_engine.hip

int speechPlayer_voiceCount(void) { return voice_count(); }

int speechPlayer_voicePresetCount(void) { return kNumVoices; }

const char* speechPlayer_voiceName(int i)
{
    if (i < 0) return nullptr;
    if (i < kNumVoices) return kVoiceRows[i].name;
    std::lock_guard<std::mutex> g(g_voiceMutex);
    const size_t k = (size_t)(i - kNumVoices);
    return k < g_definedVoices.size() ? g_definedVoices[k]->name.c_str() : nullptr;
}

int speechPlayer_voiceIndex(const char* voiceName) { return voice_index(voiceName); }

int speechPlayer_voiceDefine(const char* voiceName, int nEntries, const int* param, const double* absValue, const double* multiplier)
{
    return producer_call<int>("speechPlayer_voiceDefine", -1, [&]() -> int {
        if (!voiceName || !*voiceName || nEntries < 0 || (nEntries > 0 && !param)) { set_producer_error("speechPlayer_voiceDefine: bad arguments"); return -1; }
        std::vector<VoiceEntry> entries;
        for (int e = 0; e < nEntries; ++e) {
            if (param[e] < 0 || param[e] >= kParams) { set_producer_error("speechPlayer_voiceDefine: parameter index out of range"); return -1; }
            VoiceEntry x;
            x.param = param[e];
            x.hasAbs = absValue && !std::isnan(absValue[e]); x.abs = x.hasAbs ? absValue[e] : 0.0;
            x.hasMul = multiplier && !std::isnan(multiplier[e]); x.mul = x.hasMul ? multiplier[e] : 1.0;
            entries.push_back(x);
        }
        for (int i = 0; i < kNumVoices; ++i)
            if (!strcmp(kVoiceRows[i].name, voiceName)) { set_producer_error("speechPlayer_voiceDefine: the name of a built-in preset"); return -1; }
        std::lock_guard<std::mutex> g(g_voiceMutex);
        for (size_t i = 0; i < g_definedVoices.size(); ++i)
            if (g_definedVoices[i]->name == voiceName) { g_definedVoices[i]->entries.swap(entries); return kNumVoices + (int)i; }
        g_definedVoices.emplace_back(new VoiceDef{voiceName, std::move(entries)});
        return kNumVoices + (int)g_definedVoices.size() - 1;
    });
}

int speechPlayer_applyVoiceToFrame(speechPlayer_frame_t* frame, const char* voiceName)
{
    std::vector<VoiceEntry> v;
    if (!frame || !voice_entries(voice_index(voiceName), v)) return -1;
    apply_voice(reinterpret_cast<double*>(frame), v);
    return 0;
}

int speechPlayer_ipa_phonemeCount(void) { return kNumPhonemes; }

int speechPlayer_ipa_phoneme(int index, char* symbolUtf8, int symbolCapacity, double* values, unsigned long long* fieldMask, unsigned int* classBits)
{
    if (index < 0 || index >= kNumPhonemes) return -1;
    const PhonemeRow& r = kPhonemeRows[index];
    if (symbolUtf8) {
        std::string u;
        for (int i = 0; i < r.nSymbol; ++i) {
            const uint32_t c = r.symbol[i];
            if (c < 0x80) u.push_back((char)c);
            else if (c < 0x800) { u.push_back((char)(0xC0 | (c >> 6))); u.push_back((char)(0x80 | (c & 0x3F))); }
            else if (c < 0x10000) { u.push_back((char)(0xE0 | (c >> 12))); u.push_back((char)(0x80 | ((c >> 6) & 0x3F))); u.push_back((char)(0x80 | (c & 0x3F))); }
            else { u.push_back((char)(0xF0 | (c >> 18))); u.push_back((char)(0x80 | ((c >> 12) & 0x3F))); u.push_back((char)(0x80 | ((c >> 6) & 0x3F))); u.push_back((char)(0x80 | (c & 0x3F))); }
        }
        if ((int)u.size() + 1 > symbolCapacity) return -1;
        memcpy(symbolUtf8, u.c_str(), u.size() + 1);
    }
    if (values) memcpy(values, r.value, sizeof r.value);
    if (fieldMask) *fieldMask = r.mask;
    if (classBits) *classBits = r.cls;
    return 0;
}

long long speechPlayer_ipa_frames(const char* ipaUtf8, double speed, double basePitch, double inflection, int clauseType,
                                  const char* voiceName, speechPlayer_frame_t* frames, unsigned char* isNull,
                                  double* durationMs, double* fadeMs, long long capacity)
{
    return producer_call<long long>("speechPlayer_ipa_frames", -1, [&]() -> long long {
        const int voice = voice_index(voiceName);
        if (voiceName && *voiceName && voice < 0) return -1;
        if (!clause_known(clauseType)) return -2;
        PackArgs a;
        const char clause = (char)clauseType;
        a.nTexts = 1; a.texts = &ipaUtf8; a.speed = speed; a.basePitch = &basePitch; a.inflection = inflection;
        a.clauseTypes = &clause; a.voiceAll = voice; a.tailMs = -1.0;
        Compact c;
        if (build_compact(a, false, true, c)) return -1;
        const long long n = (long long)c.records.size();
        if (n <= capacity) {
            for (long long k = 0; k < n; ++k) {
                speechPlayer_frame_t f; unsigned char nul;
                expand_record(c, c.records[(size_t)k], &f, &nul);
                if (frames) frames[k] = f;
                if (isNull) isNull[k] = nul;
                if (durationMs) durationMs[k] = c.durationMs[(size_t)k];
                if (fadeMs) fadeMs[k] = c.fadeMs[(size_t)k];
            }
        }
        return n;
    });
}

static int check_pack_args(long long nTexts, const char* const* ipaUtf8, const char* clauseTypes, const int* voiceOf, const char* voiceName, int* voiceAll)
{
    if (nTexts < 0 || (nTexts > 0 && !ipaUtf8)) return -1;
    *voiceAll = voice_index(voiceName);
    if (!voiceOf && voiceName && *voiceName && *voiceAll < 0) return -1;
    for (long long i = 0; clauseTypes && i < nTexts; ++i)
        if (!clause_known((int)(unsigned char)clauseTypes[i])) return -2;
    return 0;
}

long long speechPlayer_ipa_pack(int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed, const double* basePitch,
                                double inflection, const char* clauseTypes, const char* voiceName, double trailingSilenceMs,
                                long long* frameStart, speechPlayer_frame_t* frames, unsigned int* minFrameDuration,
                                unsigned int* fadeDuration, unsigned char* isNull, long long frameCapacity)
{
    return producer_call<long long>("speechPlayer_ipa_pack", -1, [&]() -> long long {
        PackArgs a;
        if (int rc = check_pack_args(nTexts, ipaUtf8, clauseTypes, nullptr, voiceName, &a.voiceAll)) return rc;
        a.sampleRate = sampleRate; a.nTexts = nTexts; a.texts = ipaUtf8; a.speed = speed; a.basePitch = basePitch; a.inflection = inflection;
        a.clauseTypes = clauseTypes; a.tailMs = trailingSilenceMs;
        const bool store = frames && minFrameDuration && fadeDuration && isNull;
        Compact c;
        if (build_compact(a, !store, false, c)) return -1;
        long long pos = 0;
        std::vector<long long> start((size_t)nTexts + 1);
        for (long long i = 0; i < nTexts; ++i) { start[(size_t)i] = pos; pos += c.frames_of(i); }
        start[(size_t)nTexts] = pos;
        if (frameStart) memcpy(frameStart, start.data(), sizeof(long long) * ((size_t)nTexts + 1));
        if (store && pos <= frameCapacity) {
            parallel_for(nTexts, 256, [&](long long ua, long long ue) {
                for (long long i = ua; i < ue; ++i) {
                    const speechPlayer_frameRecord_t* r = c.records.data() + c.listStart[c.listOf[(size_t)i]];
                    const long long n = c.frames_of(i), at = start[(size_t)i];
                    for (long long k = 0; k < n; ++k) {
                        expand_record(c, r[k], &frames[at + k], &isNull[at + k]);
                        minFrameDuration[at + k] = r[k].minFrameDuration;
                        fadeDuration[at + k] = r[k].fadeDuration;
                    }
                }
            });
        }
        return pos;
    });
}

// ---- the compact form handed out (tests, callers that keep a batch's description) ---------------------------------------
struct speechPlayer_recordsObject { Compact c; };

speechPlayer_records_t speechPlayer_ipa_records(int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed, const double* basePitch,
                                                double inflection, const char* clauseTypes, const int* voiceOf, const char* voiceName,
                                                double trailingSilenceMs)
{
    return producer_call<speechPlayer_records_t>("speechPlayer_ipa_records", nullptr, [&]() -> speechPlayer_records_t {
        PackArgs a;
        if (check_pack_args(nTexts, ipaUtf8, clauseTypes, voiceOf, voiceName, &a.voiceAll)) { set_producer_error("speechPlayer_ipa_records: bad arguments, unknown voice or clause type"); return nullptr; }
        a.sampleRate = sampleRate; a.nTexts = nTexts; a.texts = ipaUtf8; a.speed = speed; a.basePitch = basePitch; a.inflection = inflection;
        a.clauseTypes = clauseTypes; a.voiceOf = voiceOf; a.tailMs = trailingSilenceMs;
        std::unique_ptr<speechPlayer_recordsObject> o(new speechPlayer_recordsObject);
        if (build_compact(a, false, false, o->c)) { set_producer_error("speechPlayer_ipa_records: voice index out of range"); return nullptr; }
        return o.release();
    });
}

int speechPlayer_records_view(speechPlayer_records_t records, speechPlayer_recordsView_t* view)
{
    if (!records || !view) return -1;
    const Compact& c = static_cast<speechPlayer_recordsObject*>(records)->c;
    view->nShapes = (long long)c.shapes.size(); view->shapes = c.shapes.data();
    view->nLists = (long long)c.listStart.size() - 1; view->listStart = c.listStart.data();
    view->nRecords = (long long)c.records.size(); view->records = c.records.data();
    view->nUtterances = (long long)c.listOf.size(); view->listOf = c.listOf.data();
    return 0;
}

void speechPlayer_records_free(speechPlayer_records_t records) { delete static_cast<speechPlayer_recordsObject*>(records); }

static int set_ipa(const char* what, speechPlayer_batch_t batch, long long nTexts, const char* const* ipaUtf8, double speed,
                   const double* basePitch, double inflection, const char* clauseTypes, const int* voiceOf, const char* voiceName,
                   double trailingSilenceMs, const unsigned int* noiseSeed)
{
    return producer_call<int>(what, -1, [&]() -> int {
        const int rate = speechPlayer_batch_sampleRate(batch);
        PackArgs a;
        if (rate <= 0 || check_pack_args(nTexts, ipaUtf8, clauseTypes, voiceOf, voiceName, &a.voiceAll)) {
            set_producer_error((std::string(what) + ": bad batch, text array, voice or clause type").c_str());
            return -1;
        }
        a.sampleRate = rate; a.nTexts = nTexts; a.texts = ipaUtf8; a.speed = speed; a.basePitch = basePitch; a.inflection = inflection;
        a.clauseTypes = clauseTypes; a.voiceOf = voiceOf; a.tailMs = trailingSilenceMs;
        Compact c;
        if (build_compact(a, false, false, c)) { set_producer_error((std::string(what) + ": voice index out of range").c_str()); return -1; }
        return speechPlayer_batch_setRecords(batch, (long long)c.shapes.size(), c.shapes.data(), (long long)c.listStart.size() - 1, c.listStart.data(),
                                             c.records.data(), nTexts, c.listOf.data(), noiseSeed);
    });
}

int speechPlayer_batch_setIpa(speechPlayer_batch_t batch, long long nTexts, const char* const* ipaUtf8, double speed,
                              const double* basePitch, double inflection, const char* clauseTypes, const char* voiceName,
                              double trailingSilenceMs, const unsigned int* noiseSeed)
{
    return set_ipa("speechPlayer_batch_setIpa", batch, nTexts, ipaUtf8, speed, basePitch, inflection, clauseTypes, nullptr, voiceName, trailingSilenceMs, noiseSeed);
}

int speechPlayer_node_setIpa(speechPlayer_node_t node, int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed,
                             const double* basePitch, double inflection, const char* clauseTypes, const int* voiceOf, const char* voiceName,
                             double trailingSilenceMs, const unsigned int* noiseSeed)
{
    return producer_call<int>("speechPlayer_node_setIpa", -1, [&]() -> int {
        PackArgs a;
        if (sampleRate <= 0 || check_pack_args(nTexts, ipaUtf8, clauseTypes, voiceOf, voiceName, &a.voiceAll)) {
            set_producer_error("speechPlayer_node_setIpa: bad sample rate, text array, voice or clause type");
            return -1;
        }
        a.sampleRate = sampleRate; a.nTexts = nTexts; a.texts = ipaUtf8; a.speed = speed; a.basePitch = basePitch; a.inflection = inflection;
        a.clauseTypes = clauseTypes; a.voiceOf = voiceOf; a.tailMs = trailingSilenceMs;
        Compact c;
        if (build_compact(a, false, false, c)) { set_producer_error("speechPlayer_node_setIpa: voice index out of range"); return -1; }
        return speechPlayer_node_setRecords(node, (long long)c.shapes.size(), c.shapes.data(), (long long)c.listStart.size() - 1, c.listStart.data(),
                                            c.records.data(), nTexts, c.listOf.data(), noiseSeed);
    });
}

int speechPlayer_batch_setIpaVoices(speechPlayer_batch_t batch, long long nTexts, const char* const* ipaUtf8, double speed,
                                    const double* basePitch, double inflection, const char* clauseTypes, const int* voiceOf,
                                    double trailingSilenceMs, const unsigned int* noiseSeed)
{
    return set_ipa("speechPlayer_batch_setIpaVoices", batch, nTexts, ipaUtf8, speed, basePitch, inflection, clauseTypes, voiceOf, nullptr, trailingSilenceMs, noiseSeed);
}

}  // extern "C"

// ==========================================================================================
// Optional text front-end: text -> IPA through eSpeak NG, when that library is installed
// ==========================================================================================
// What the NVDA driver does before it reaches the frame producer (reference nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py):
// split the text into clauses at white space that follows one of . ? ! , : ; (:84, :189), take the clause type and the pause
// after the last clause from each clause's last character (:195-205), send the clause through espeak_TextToPhonemes with the
// mode word 0x36100 + 0x82 (:210: IPA in UTF-8, U+0361 as the tie), apply four replacements to what comes back (:214-217) and
// hand the IPA to ipa.generateFramesAndTiming (:222); after the last clause, silence of that pause (:234).
// PARITY UNPINNED: eSpeak NG is a third-party library that neither the reference tree nor this image contains; nothing here
// has been run against it.  The tests cover the splitting (against the reference's regular expression), the replacements and
// the error when the library is absent -- not the phonemes.
namespace {

void set_text_error(const char* msg);

bool is_space_at(const unsigned char* p, int* len)
{
    // Python's \s on str: ASCII white space and the Unicode spaces
    const unsigned char c = p[0];
    if (c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f)) { *len = 1; return true; }
    if (c == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) { *len = 2; return true; }
    if (c == 0xE1 && p[1] == 0x9A && p[2] == 0x80) { *len = 3; return true; }
    if (c == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) { *len = 3; return true; }
    if (c == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) { *len = 3; return true; }
    if (c == 0xE3 && p[1] == 0x80 && p[2] == 0x80) { *len = 3; return true; }
    return false;
}

struct Clause { size_t begin, end; char type; double endPauseMs; };

// reference __init__.py:84 (re_textPause = (?<=[.?!,:;])\s), :189-205
void split_clauses(const char* text, std::vector<Clause>& out)
{
    out.clear();
    const unsigned char* t = reinterpret_cast<const unsigned char*>(text);
    const size_t n = strlen(text);
    size_t start = 0;
    auto close = [&](size_t from, size_t to) {
        int l;
        while (from < to && is_space_at(t + from, &l)) from += l;      // chunk.strip()
        for (;;) {
            size_t k = to;
            while (k > from && (t[k - 1] & 0xC0) == 0x80) --k;         // the start of the last code point
            if (k > from && is_space_at(t + k - 1, &l) && k - 1 + l == to) to = k - 1; else break;
        }
        if (from >= to) return;
        Clause c; c.begin = from; c.end = to;
        const char last = (char)t[to - 1];
        if (last == '.' || last == '!' || last == '?') { c.type = last; c.endPauseMs = 150.0; }
        else if (last == ',') { c.type = ','; c.endPauseMs = 120.0; }
        else { c.type = 0; c.endPauseMs = 100.0; }
        out.push_back(c);
    };
    size_t i = 0;
    while (i < n) {
        int l;
        if (i > 0 && is_space_at(t + i, &l) && strchr(".?!,:;", (char)t[i - 1]) && t[i - 1] < 0x80) {
            close(start, i);
            start = i + l;
            i += l;
        } else ++i;
    }
    close(start, n);
}

// reference __init__.py:214-218
std::string ipa_fixups(const std::string& in)
{
    static const char* const pairs[4][2] = {
        {"\xC9\x99\xCD\xA1l", "\xCA\x8A\xCD\xA1l"},                     // ə͡l -> ʊ͡l
        {"a\xCD\xA1\xC9\xAA", "\xC9\x91\xCD\xA1\xC9\xAA"},           // a͡ɪ -> ɑ͡ɪ
        {"e\xCD\xA1\xC9\xAA", "e\xCD\xA1i"},                             // e͡ɪ -> e͡i
        {"\xC9\x99\xCD\xA1\xCA\x8A", "o\xCD\xA1u"},                    // ə͡ʊ -> o͡u
    };
    std::string s = in;
    for (const auto& p : pairs) {
        const std::string from = p[0], to = p[1];
        for (size_t at = 0; (at = s.find(from, at)) != std::string::npos; at += to.size()) s.replace(at, from.size(), to);
    }
    size_t a = 0, b = s.size();
    int l;
    while (a < b && is_space_at(reinterpret_cast<const unsigned char*>(s.c_str()) + a, &l)) a += l;
    while (b > a && (s[b - 1] == ' ' || (s[b - 1] >= 9 && s[b - 1] <= 13))) --b;
    return s.substr(a, b - a);
}

// eSpeak NG's C interface (speak_lib.h, as published): only what TextToPhonemes needs
struct Espeak {
    void* lib = nullptr;
    int (*Initialize)(int output, int buflength, const char* path, int options) = nullptr;
    int (*SetVoiceByName)(const char* name) = nullptr;
    const char* (*TextToPhonemes)(const void** textptr, int textmode, int phonememode) = nullptr;
    std::string voice;
    std::string why;            // why it is not available
    bool tried = false;
};
std::mutex g_espeakMutex;       // eSpeak keeps global state: one caller at a time
Espeak g_espeak;

bool espeak_ready()             // g_espeakMutex held
{
    Espeak& e = g_espeak;
    if (e.tried) return e.lib != nullptr;
    e.tried = true;
    const char* named = getenv("SPEECHPLAYER_ESPEAK_LIB");
    const char* names[] = {named, "libespeak-ng.so.1", "libespeak-ng.so", "libespeak.so.1"};
    std::string tried;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        e.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (e.lib) break;
        tried += tried.empty() ? "" : ", "; tried += nm;
    }
    if (!e.lib) {
        e.why = "the text front-end needs eSpeak NG, and none of " + tried + " could be loaded (install libespeak-ng1, or point SPEECHPLAYER_ESPEAK_LIB at the library); "
                "IPA input (speechPlayer_batch_setIpa) needs no such library";
        return false;
    }
    e.Initialize = reinterpret_cast<int (*)(int, int, const char*, int)>(dlsym(e.lib, "espeak_Initialize"));
    e.SetVoiceByName = reinterpret_cast<int (*)(const char*)>(dlsym(e.lib, "espeak_SetVoiceByName"));
    e.TextToPhonemes = reinterpret_cast<const char* (*)(const void**, int, int)>(dlsym(e.lib, "espeak_TextToPhonemes"));
    if (!e.Initialize || !e.SetVoiceByName || !e.TextToPhonemes) {
        e.why = "the eSpeak library that was loaded lacks espeak_Initialize / espeak_SetVoiceByName / espeak_TextToPhonemes";
        dlclose(e.lib); e.lib = nullptr;
        return false;
    }
    // AUDIO_OUTPUT_RETRIEVAL (1): no audio device is opened; espeakINITIALIZE_DONT_EXIT (0x8000): report, do not exit()
    if (e.Initialize(1, 0, nullptr, 0x8000) < 0) {
        e.why = "espeak_Initialize failed (is espeak-ng-data installed?)";
        dlclose(e.lib); e.lib = nullptr;
        return false;
    }
    return true;
}

// one clause of text -> IPA (reference __init__.py:206-218); false with the reason in g_espeak.why
bool clause_to_ipa(const char* text, size_t len, const char* voice, std::string& ipa)
{
    Espeak& e = g_espeak;
    if (!espeak_ready()) return false;
    const std::string v = (voice && *voice) ? voice : "en";
    if (v != e.voice) {
        if (e.SetVoiceByName(v.c_str()) != 0) { e.why = "espeak_SetVoiceByName(" + v + ") failed"; set_text_error(e.why.c_str()); return false; }
        e.voice = v;
    }
    const std::string chunk(text, len);
    const void* ptr = chunk.c_str();
    std::string got;
    while (ptr) {
        const char* ph = e.TextToPhonemes(&ptr, 1 /* espeakCHARS_UTF8 */, 0x36100 + 0x82);
        if (!ph) continue;
        got += ph;
    }
    ipa = ipa_fixups(got);
    return true;
}

}  // namespace

extern "C" {

void speechPlayer_internal_setError(int code, const char* message);      // klatt_engine.hip

}

namespace {
void set_text_error(const char* msg) { speechPlayer_internal_setError(SPEECHPLAYER_ERR_TEXT_FRONTEND, msg); }
void set_arg_error(const char* msg) { speechPlayer_internal_setError(SPEECHPLAYER_ERR_ARGUMENT, msg); }
void set_producer_error(const char* msg) { speechPlayer_internal_setError(SPEECHPLAYER_ERR_ARGUMENT, msg); }
// Every text entry point runs through this: the error code of an earlier call is cleared, and nothing a host allocation throws
// (std::bad_alloc, std::length_error) crosses the C ABI -- the caller gets `fail` and a message instead.
template <class R, class F>
R text_call(const char* what, R fail, F body)
{
    speechPlayer_internal_setError(0, "");
    try {
        return body();
    } catch (const std::exception& e) {
        set_text_error((std::string(what) + ": " + e.what()).c_str());
        return fail;
    }
}
}

extern "C" {

int speechPlayer_text_available(void)
{
    std::lock_guard<std::mutex> g(g_espeakMutex);
    if (espeak_ready()) return 1;
    set_text_error(g_espeak.why.c_str());
    return 0;
}

long long speechPlayer_text_fixups(const char* ipaUtf8, char* out, long long capacity)
{
    return text_call<long long>("speechPlayer_text_fixups", -1, [&]() -> long long {
        if (!ipaUtf8) { set_arg_error("speechPlayer_text_fixups: NULL text"); return -1; }
        const std::string s = ipa_fixups(ipaUtf8);
        if (out && capacity > (long long)s.size()) memcpy(out, s.c_str(), s.size() + 1);
        return (long long)s.size() + 1;
    });
}

long long speechPlayer_text_clauses(const char* textUtf8, long long* begin, long long* end, char* clauseType, double* endPauseMs, long long capacity)
{
    return text_call<long long>("speechPlayer_text_clauses", -1, [&]() -> long long {
        if (!textUtf8) { set_arg_error("speechPlayer_text_clauses: NULL text"); return -1; }
        std::vector<Clause> cl;
        split_clauses(textUtf8, cl);
        for (size_t i = 0; i < cl.size() && (long long)i < capacity; ++i) {
            if (begin) begin[i] = (long long)cl[i].begin;
            if (end) end[i] = (long long)cl[i].end;
            if (clauseType) clauseType[i] = cl[i].type;
            if (endPauseMs) endPauseMs[i] = cl[i].endPauseMs;
        }
        return (long long)cl.size();
    });
}

long long speechPlayer_text_toIpa(const char* textUtf8, const char* espeakVoice, char* out, long long capacity)
{
    return text_call<long long>("speechPlayer_text_toIpa", -1, [&]() -> long long {
        if (!textUtf8) { set_arg_error("speechPlayer_text_toIpa: NULL text"); return -1; }
        std::lock_guard<std::mutex> g(g_espeakMutex);
        std::string ipa;
        if (!clause_to_ipa(textUtf8, strlen(textUtf8), espeakVoice, ipa)) { set_text_error(g_espeak.why.c_str()); return -3; }
        if (out && capacity > (long long)ipa.size()) memcpy(out, ipa.c_str(), ipa.size() + 1);
        return (long long)ipa.size() + 1;
    });
}

int speechPlayer_batch_setText(speechPlayer_batch_t batch, long long nTexts, const char* const* textUtf8, const char* espeakVoice, double speed,
                               const double* basePitch, double inflection, const char* voiceName, const unsigned int* noiseSeed)
{
    return text_call<int>("speechPlayer_batch_setText", -1, [&]() -> int {
        const int rate = speechPlayer_batch_sampleRate(batch);
        if (rate <= 0 || nTexts < 0 || (nTexts > 0 && !textUtf8) || !(speed > 0.0)) { set_arg_error("speechPlayer_batch_setText: bad batch, text array or speed"); return -1; }
        const int voice = voice_index(voiceName);
        if (voiceName && *voiceName && voice < 0) { set_arg_error("speechPlayer_batch_setText: unknown voice preset"); return -1; }
        // every clause of every text becomes one ITEM of a single producer call (its IPA, its clause type, its text's base pitch)
        std::vector<Clause> cl;
        std::unordered_map<std::string, std::string> ipaOf;       // clause text -> IPA: a batch repeats its sentences
        std::vector<const std::string*> itemIpa;
        std::string itemClause;
        std::vector<double> itemPitch;
        std::vector<size_t> firstItem((size_t)nTexts + 1, 0);
        std::vector<double> endPauseOf((size_t)nTexts, 20.0);     // reference __init__.py:182: not divided by the rate unless a clause sets it (:204)
        {
            std::lock_guard<std::mutex> g(g_espeakMutex);
            for (long long i = 0; i < nTexts; ++i) {
                firstItem[(size_t)i] = itemIpa.size();
                if (!textUtf8[i]) { set_arg_error("speechPlayer_batch_setText: NULL text"); return -1; }
                split_clauses(textUtf8[i], cl);
                for (const Clause& c : cl) {
                    endPauseOf[(size_t)i] = c.endPauseMs / speed;
                    const std::string key(textUtf8[i] + c.begin, c.end - c.begin);
                    auto it = ipaOf.find(key);
                    if (it == ipaOf.end()) {
                        std::string ipa;
                        if (!clause_to_ipa(key.c_str(), key.size(), espeakVoice, ipa)) { set_text_error(g_espeak.why.c_str()); return -3; }
                        it = ipaOf.emplace(key, ipa).first;
                    }
                    if (it->second.empty()) continue;             // :219
                    itemIpa.push_back(&it->second);               // (nodes of an unordered_map stay where they are)
                    itemClause.push_back(c.type);
                    itemPitch.push_back(basePitch ? basePitch[i] : 100.0);
                }
            }
            firstItem[(size_t)nTexts] = itemIpa.size();
        }
        std::vector<const char*> itemPtr(itemIpa.size());
        for (size_t k = 0; k < itemIpa.size(); ++k) itemPtr[k] = itemIpa[k]->c_str();
        PackArgs a;
        a.sampleRate = rate; a.nTexts = (long long)itemPtr.size(); a.texts = itemPtr.data(); a.speed = speed; a.basePitch = itemPitch.data();
        a.inflection = inflection; a.clauseTypes = itemClause.c_str(); a.voiceAll = voice; a.tailMs = -1.0;
        Compact c;
        if (build_compact(a, false, false, c)) { set_arg_error("speechPlayer_batch_setText: voice index out of range"); return -1; }
        // an utterance: its clauses' records one after the other, then silence after the last clause (:234):
        // queueFrame(None, endPause / rate, max(10, 10 / rate))
        std::vector<long long> start((size_t)nTexts + 1, 0);
        std::vector<speechPlayer_frameRecord_t> recs;
        for (long long i = 0; i < nTexts; ++i) {
            start[(size_t)i] = (long long)recs.size();
            for (size_t k = firstItem[(size_t)i]; k < firstItem[(size_t)i + 1]; ++k) {
                const uint32_t l = c.listOf[k];
                recs.insert(recs.end(), c.records.begin() + c.listStart[l], c.records.begin() + c.listStart[l + 1]);
            }
            speechPlayer_frameRecord_t t;
            t.shape = SPEECHPLAYER_RECORD_SILENCE; t.voicePitch = 0.0; t.endVoicePitch = 0.0; t.userIndex = -1;
            t.minFrameDuration = ms_to_samples(endPauseOf[(size_t)i], rate);
            t.fadeDuration = ms_to_samples(std::max(10.0, 10.0 / speed), rate);
            recs.push_back(t);
        }
        start[(size_t)nTexts] = (long long)recs.size();
        return speechPlayer_batch_setRecords(batch, (long long)c.shapes.size(), c.shapes.data(), nTexts, start.data(), recs.data(), nTexts, nullptr, noiseSeed);
    });
}

}  // extern "C"
