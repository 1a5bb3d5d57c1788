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
enum : uint16_t { S_TIED_TO = 1, S_TIED_FROM = 2, S_LONG = 4, S_WORD_START = 8, S_SYLLABLE_START = 16, S_GAP = 32, S_PUFF = 64, S_HAS_PITCH = 128 };

struct Segment {
    int row;             // table row; -1 for a pre-stop gap (silence)
    uint8_t cls;
    uint8_t stress;      // 0, 1 primary, 2 secondary; meaningful on syllable heads
    uint16_t pros;       // S_* bits
    int borrowed;        // index into Utterance::extra of the fields taken over from a neighbour, or -1
    double duration, fade, pitch, endPitch;
};
struct Borrowed { uint64_t mask; double value[kParams]; };

struct Utterance {
    std::vector<Segment> seg;
    std::vector<Borrowed> extra;
    uint64_t mask_of(const Segment& s) const { return (s.row < 0 ? 0 : kPhonemeRows[s.row].mask) | (s.borrowed < 0 ? 0 : extra[s.borrowed].mask); }
    double field(const Segment& s, int k) const
    {
        if (s.row >= 0 && (kPhonemeRows[s.row].mask >> k & 1)) return kPhonemeRows[s.row].value[k];
        return extra[s.borrowed].value[k];
    }
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
    s.row = row; s.cls = row < 0 ? 0 : kPhonemeRows[row].cls; s.stress = 0; s.pros = 0; s.borrowed = -1;
    s.duration = s.fade = s.pitch = s.endPitch = 0.0;
    return s;
}

// Code points -> segments.  A stress mark waits for the next symbol the table knows; a tie bar joins two symbols into
// one row when the table has the pair, else it marks its neighbours as tied; a length mark prefers the lengthened row.
// Between a voiceless stop and a voiced continuant goes a short aspiration (/h/); before a stop or affricate that
// does not carry the stress mark itself goes a gap.
void lex(const std::vector<uint32_t>& cp, Utterance& u)
{
    u.seg.clear(); u.extra.clear();
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
void colour(Utterance& u)
{
    const int n = (int)u.seg.size();
    for (int i = 0; i < n; ++i) {
        Segment& s = u.seg[i];
        if (!(s.cls & C_COPY_ADJACENT)) continue;
        int from = (i + 1 < n && !(u.seg[i + 1].pros & S_GAP)) ? i + 1 : i - 1;
        if (from < 0) continue;
        const Segment& d = u.seg[from];
        const uint64_t take = u.mask_of(d) & ~u.mask_of(s);
        if (!take) continue;
        Borrowed b;
        b.mask = take;
        for (int k = 0; k < kParams; ++k) b.value[k] = (take >> k & 1) ? u.field(d, k) : 0.0;
        if (s.borrowed >= 0) {      // (cannot happen in one pass; kept so that a second pass would merge)
            Borrowed& old = u.extra[s.borrowed];
            for (int k = 0; k < kParams; ++k) if (take >> k & 1) old.value[k] = b.value[k];
            old.mask |= take;
        } else {
            u.extra.push_back(b);
            s.borrowed = (int)u.extra.size() - 1;
        }
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

void apply_spans(Utterance& u, const std::vector<Span>& spans, double basePitch, double inflection)
{
    for (const Span& sp : spans) {
        const double p0 = basePitch * std::pow(2.0, ((sp.from - 50) / 50.0) * inflection);
        const double p1 = basePitch * std::pow(2.0, ((sp.to - 50) / 50.0) * inflection);
        double voiced = 0.0;
        for (int i = sp.a; i < sp.b; ++i) if (u.seg[i].cls & C_VOICED) voiced += u.seg[i].duration;
        const double delta = p1 - p0;
        double done = 0.0, cur = p0;
        for (int i = sp.a; i < sp.b; ++i) {
            Segment& s = u.seg[i];
            s.pitch = cur;
            if (s.cls & C_VOICED) {
                done += s.duration;
                cur = p0 + (delta * (done / voiced));
            }
            s.endPitch = cur;
            s.pros |= S_HAS_PITCH;
        }
    }
}

const VoiceRow* find_voice(const char* name)
{
    if (!name || !*name) return nullptr;
    for (const VoiceRow& v : kVoiceRows)
        if (!strcmp(v.name, name)) return &v;
    // the reference's table has one key with a trailing blank ("Caleb "): accept the name without it too
    const size_t len = strlen(name);
    for (const VoiceRow& v : kVoiceRows)
        if (strlen(v.name) == len + 1 && !strncmp(v.name, name, len) && v.name[len] == ' ') return &v;
    return nullptr;
}

// absolute value first, then the multiplier, parameter by parameter in frame order (entries are stored in that order)
void apply_voice(double* frame, const VoiceRow& v)
{
    for (int e = 0; e < v.nEntries; ++e) {
        const VoiceEntry& x = v.entries[e];
        if (x.hasAbs) frame[x.param] = x.abs;
        if (x.hasMul) frame[x.param] = frame[x.param] * x.mul;
    }
}

// one finished stream: what the reference's generator yields, as arrays
struct Stream {
    std::vector<double> frames;          // [n][47]; zeros for silence
    std::vector<unsigned char> isNull;
    std::vector<double> durationMs, fadeMs;
    size_t size() const { return isNull.size(); }
};

void emit(const Utterance& u, const VoiceRow* voice, Stream& out)
{
    const size_t n = u.seg.size();
    out.frames.assign(n * kParams, 0.0);
    out.isNull.assign(n, 0);
    out.durationMs.resize(n); out.fadeMs.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const Segment& s = u.seg[i];
        out.durationMs[i] = s.duration; out.fadeMs[i] = s.fade;
        if (s.pros & S_GAP) { out.isNull[i] = 1; continue; }
        double* f = &out.frames[i * kParams];
        f[P_PREGAIN] = 1.0;                 // reference ipa.py:349-351
        f[P_OUTGAIN] = 2.0;
        const uint64_t m = u.mask_of(s);
        for (int k = 0; k < kParams; ++k) if (m >> k & 1) f[k] = u.field(s, k);
        if (s.pros & S_HAS_PITCH) { f[P_VOICEPITCH] = s.pitch; f[P_ENDPITCH] = s.endPitch; }
        if (voice) apply_voice(f, *voice);
    }
}

// Builder with the per-call memo: segments + timing per text, finished streams per (text, clause, base pitch).
class Producer {
public:
    Producer(double speed, double inflection, const VoiceRow* voice) : speed_(speed), inflection_(inflection), voice_(voice) {}

    const Stream& stream(const char* text, int clause, double basePitch)
    {
        std::string key(text ? text : "");
        key.push_back('\0'); key.push_back((char)clause);
        uint64_t bits; memcpy(&bits, &basePitch, 8);
        key.append(reinterpret_cast<const char*>(&bits), 8);
        auto it = done_.find(key);
        if (it != done_.end()) return it->second;
        const Timed& t = timed(text);
        Utterance u = t.u;                         // pitches are per (clause, base pitch): work on a copy
        std::vector<Span>& spans = spansScratch_;
        if (!u.seg.empty()) {
            contour_spans(u, clause, spans);
            apply_spans(u, spans, basePitch, inflection_);
        }
        Stream& s = done_[key];
        emit(u, voice_, s);
        return s;
    }

private:
    struct Timed { Utterance u; };
    const Timed& timed(const char* text)
    {
        std::string key(text ? text : "");
        auto it = timed_.find(key);
        if (it != timed_.end()) return it->second;
        Timed& t = timed_[key];
        decode_utf8(text, cpScratch_);
        lex(cpScratch_, t.u);
        colour(t.u);
        time_segments(t.u, speed_);
        return t;
    }
    double speed_, inflection_;
    const VoiceRow* voice_;
    std::unordered_map<std::string, Timed> timed_;
    std::unordered_map<std::string, Stream> done_;
    std::vector<uint32_t> cpScratch_;
    std::vector<Span> spansScratch_;
};

// milliseconds -> samples as the reference wrapper converts them (reference speechPlayer.py:53): int(ms * (sr / 1000.0))
unsigned int ms_to_samples(double ms, int sampleRate)
{
    const double v = ms * (sampleRate / 1000.0);
    return v <= 0.0 ? 0u : (v >= 4294967295.0 ? 4294967295u : (unsigned int)(long long)v);
}

}  // namespace

extern "C" {

int speechPlayer_voiceCount(void) { return kNumVoices; }

const char* speechPlayer_voiceName(int i) { return (i >= 0 && i < kNumVoices) ? kVoiceRows[i].name : nullptr; }

int speechPlayer_applyVoiceToFrame(speechPlayer_frame_t* frame, const char* voiceName)
{
    const VoiceRow* v = find_voice(voiceName);
    if (!frame || !v) return -1;
    apply_voice(reinterpret_cast<double*>(frame), *v);
    return 0;
}

// the clause types the intonation table knows (reference ipa.py:207-276; the reference raises KeyError for any other), and 0: none
static bool clause_known(int clause)
{
    if (clause == 0) return true;
    for (const IntonationRow& r : kIntonationRows)
        if (r.clause == clause) return true;
    return false;
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
    const VoiceRow* voice = find_voice(voiceName);
    if (voiceName && *voiceName && !voice) return -1;
    if (!clause_known(clauseType)) return -2;
    Producer p(speed, inflection, voice);
    const Stream& s = p.stream(ipaUtf8, clauseType, basePitch);
    const long long n = (long long)s.size();
    if (n <= capacity) {
        if (n && frames) memcpy(frames, s.frames.data(), (size_t)n * sizeof(speechPlayer_frame_t));
        if (n && isNull) memcpy(isNull, s.isNull.data(), (size_t)n);
        if (n && durationMs) memcpy(durationMs, s.durationMs.data(), (size_t)n * sizeof(double));
        if (n && fadeMs) memcpy(fadeMs, s.fadeMs.data(), (size_t)n * sizeof(double));
    }
    return n;
}

// one pass over the texts with a producer (and its memo) the caller keeps: sizes when the arrays are absent, else fills them
static long long pack_with(Producer& p, int sampleRate, long long nTexts, const char* const* ipaUtf8, const double* basePitch,
                           const char* clauseTypes, double trailingSilenceMs, long long* frameStart, speechPlayer_frame_t* frames,
                           unsigned int* minFrameDuration, unsigned int* fadeDuration, unsigned char* isNull, long long frameCapacity);

long long speechPlayer_ipa_pack(int sampleRate, long long nTexts, const char* const* ipaUtf8, double speed, const double* basePitch,
                                double inflection, const char* clauseTypes, const char* voiceName, double trailingSilenceMs,
                                long long* frameStart, speechPlayer_frame_t* frames, unsigned int* minFrameDuration,
                                unsigned int* fadeDuration, unsigned char* isNull, long long frameCapacity)
{
    if (nTexts < 0 || (nTexts > 0 && !ipaUtf8)) return -1;
    const VoiceRow* voice = find_voice(voiceName);
    if (voiceName && *voiceName && !voice) return -1;
    for (long long i = 0; clauseTypes && i < nTexts; ++i)
        if (!clause_known((int)(unsigned char)clauseTypes[i])) return -2;
    Producer p(speed, inflection, voice);
    return pack_with(p, sampleRate, nTexts, ipaUtf8, basePitch, clauseTypes, trailingSilenceMs, frameStart, frames, minFrameDuration, fadeDuration, isNull, frameCapacity);
}

static long long pack_with(Producer& p, int sampleRate, long long nTexts, const char* const* ipaUtf8, const double* basePitch,
                           const char* clauseTypes, double trailingSilenceMs, long long* frameStart, speechPlayer_frame_t* frames,
                           unsigned int* minFrameDuration, unsigned int* fadeDuration, unsigned char* isNull, long long frameCapacity)
{
    const bool tail = trailingSilenceMs >= 0.0;
    const bool store = frames && minFrameDuration && fadeDuration && isNull;
    long long pos = 0;
    for (long long i = 0; i < nTexts; ++i) {
        const Stream& s = p.stream(ipaUtf8[i], clauseTypes ? (int)(unsigned char)clauseTypes[i] : 0, basePitch ? basePitch[i] : 100.0);
        const long long n = (long long)s.size();
        if (frameStart) frameStart[i] = pos;
        if (store && pos + n + (tail ? 1 : 0) <= frameCapacity) {
            if (n) {
                memcpy(frames + pos, s.frames.data(), (size_t)n * sizeof(speechPlayer_frame_t));
                memcpy(isNull + pos, s.isNull.data(), (size_t)n);
            }
            for (long long k = 0; k < n; ++k) {
                minFrameDuration[pos + k] = ms_to_samples(s.durationMs[k], sampleRate);
                fadeDuration[pos + k] = ms_to_samples(s.fadeMs[k], sampleRate);
            }
            if (tail) {                      // silence after the utterance, as reference test_speakIpa.py:27 queues it
                memset(frames + pos + n, 0, sizeof(speechPlayer_frame_t));
                isNull[pos + n] = 1;
                minFrameDuration[pos + n] = ms_to_samples(trailingSilenceMs, sampleRate);
                fadeDuration[pos + n] = 0;
            }
        }
        pos += n + (tail ? 1 : 0);
    }
    if (frameStart) frameStart[nTexts] = pos;
    return pos;
}

int speechPlayer_batch_setIpa(speechPlayer_batch_t batch, long long nTexts, const char* const* ipaUtf8, double speed,
                              const double* basePitch, double inflection, const char* clauseTypes, const char* voiceName,
                              double trailingSilenceMs, const unsigned int* noiseSeed)
{
    int rate = speechPlayer_batch_sampleRate(batch);
    if (rate <= 0 || nTexts < 0 || (nTexts > 0 && !ipaUtf8)) return -1;
    const VoiceRow* voice = find_voice(voiceName);
    if (voiceName && *voiceName && !voice) return -1;
    for (long long i = 0; clauseTypes && i < nTexts; ++i)
        if (!clause_known((int)(unsigned char)clauseTypes[i])) return -1;
    // ONE producer for the sizing pass and the filling pass: the second finds every stream in the first's memo
    Producer p(speed, inflection, voice);
    std::vector<long long> start((size_t)nTexts + 1, 0);
    const long long total = pack_with(p, rate, nTexts, ipaUtf8, basePitch, clauseTypes, trailingSilenceMs, start.data(), nullptr, nullptr, nullptr, nullptr, 0);
    if (total < 0) return -1;
    std::vector<speechPlayer_frame_t> frames((size_t)total);
    std::vector<unsigned int> mins((size_t)total), fades((size_t)total);
    std::vector<unsigned char> nul((size_t)total);
    if (pack_with(p, rate, nTexts, ipaUtf8, basePitch, clauseTypes, trailingSilenceMs, start.data(), frames.data(), mins.data(), fades.data(), nul.data(), total) != total)
        return -1;
    return speechPlayer_batch_setUtterances(batch, nTexts, start.data(), frames.data(), mins.data(), fades.data(), nullptr, nul.data(), noiseSeed);
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
        const VoiceRow* voice = find_voice(voiceName);
        if (voiceName && *voiceName && !voice) { set_arg_error("speechPlayer_batch_setText: unknown voice preset"); return -1; }
        Producer p(speed, inflection, voice);
        std::vector<long long> start((size_t)nTexts + 1, 0);
        std::vector<speechPlayer_frame_t> frames;
        std::vector<unsigned int> mins, fades;
        std::vector<unsigned char> nul;
        std::vector<Clause> cl;
        std::unordered_map<std::string, std::string> ipaOf;       // clause text -> IPA: a batch repeats its sentences
        {
            std::lock_guard<std::mutex> g(g_espeakMutex);
            for (long long i = 0; i < nTexts; ++i) {
                start[(size_t)i] = (long long)nul.size();
                if (!textUtf8[i]) { set_arg_error("speechPlayer_batch_setText: NULL text"); return -1; }
                split_clauses(textUtf8[i], cl);
                double endPause = 20.0;                           // reference __init__.py:182: not divided by the rate unless a clause sets it (:204)
                for (const Clause& c : cl) {
                    endPause = c.endPauseMs / speed;
                    const std::string key(textUtf8[i] + c.begin, c.end - c.begin);
                    auto it = ipaOf.find(key);
                    if (it == ipaOf.end()) {
                        std::string ipa;
                        if (!clause_to_ipa(key.c_str(), key.size(), espeakVoice, ipa)) { set_text_error(g_espeak.why.c_str()); return -3; }
                        it = ipaOf.emplace(key, ipa).first;
                    }
                    if (it->second.empty()) continue;             // :219
                    const Stream& s = p.stream(it->second.c_str(), (int)(unsigned char)c.type, basePitch ? basePitch[i] : 100.0);
                    const size_t n = s.size(), at = nul.size();
                    frames.resize(at + n); mins.resize(at + n); fades.resize(at + n); nul.resize(at + n);
                    if (n) {
                        memcpy(&frames[at], s.frames.data(), n * sizeof(speechPlayer_frame_t));
                        memcpy(&nul[at], s.isNull.data(), n);
                    }
                    for (size_t k = 0; k < n; ++k) {
                        mins[at + k] = ms_to_samples(s.durationMs[k], rate);
                        fades[at + k] = ms_to_samples(s.fadeMs[k], rate);
                    }
                }
                // silence after the last clause (:234): queueFrame(None, endPause / rate, max(10, 10 / rate))
                speechPlayer_frame_t zero;
                memset(&zero, 0, sizeof zero);
                frames.push_back(zero); nul.push_back(1);
                mins.push_back(ms_to_samples(endPause, rate));
                fades.push_back(ms_to_samples(std::max(10.0, 10.0 / speed), rate));
            }
            start[(size_t)nTexts] = (long long)nul.size();
        }
        return speechPlayer_batch_setUtterances(batch, nTexts, start.data(), frames.data(), mins.data(), fades.data(), nullptr, nul.data(), noiseSeed);
    });
}

}  // extern "C"
