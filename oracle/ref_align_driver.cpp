// oracle/ref_align_driver.cpp -- TEST INFRASTRUCTURE.
// Known-answer generator for Sealer's consensus step.  It is
// linked by oracle/Makefile against the REFERENCE's own sources from where they lie
// (/root/reference/subprojects/sealer/Align/alignGlobal.cc, Common/Sequence.cpp,
// Common/Options.cpp: standard library only, nothing copied, no stand-ins) and prints
// vectors as JSON.  Built only in the authoring container (the GPU box has no
// /root/reference); tests/golden/make_nw_kat.py commits its output as
// tests/golden/nw_kat.json.
//
// What it pins:
//   pairs  -- alignGlobal(a, b): consensus string + number of matches
//             (Align/alignGlobal.cc:55-198)
//   multi  -- align(seqs): alignPair for two, alignMulti's left fold for more, as
//             Konnector/konnector.h:393 calls it (Align/alignGlobal.h:44-83)
//   revcomp, amb_or, amb_and -- Common/Sequence.cpp:21-57, Sequence.h:97-106,
//             tables Sequence.cpp:141-204
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "Align/alignGlobal.h"
#include "Common/Sequence.h"

static uint64_t rng_state = 0x2545F4914F6CDD1DULL;
static uint64_t rng()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

// alphabets: what a connecting sequence can hold.  A path through the de Bruijn graph is
// upper-case ACGT; a folded consensus (alignMulti's running `alignment`) also holds
// IUPAC codes and lower case, so the pair vectors cover those too.
static const char* kAcgt = "ACGT";
static const char* kIupac = "ACGTACGTACGTMRWSYKVHDBN";
static const char* kMixed = "ACGTACGTACGTacgtMRWSYKVHDBNmrwsykvhdbn";

static std::string random_seq(size_t len, const char* alphabet, unsigned low_complexity)
{
    size_t na = 0;
    while (alphabet[na])
        na++;
    std::string s(len, 'A');
    for (size_t i = 0; i < len; i++) {
        if (low_complexity && i >= low_complexity && rng() % 4 != 0)
            s[i] = s[i - low_complexity]; // tandem repeats: many co-optimal alignments
        else
            s[i] = alphabet[rng() % na];
    }
    return s;
}

// a noisy copy: substitutions, insertions, deletions (runs included) at `pct` percent
static std::string mutate(const std::string& s, unsigned pct, const char* alphabet)
{
    size_t na = 0;
    while (alphabet[na])
        na++;
    std::string o;
    for (size_t i = 0; i < s.size(); i++) {
        unsigned r = rng() % 100;
        if (r < pct) {
            switch (rng() % 3) {
            case 0:
                o += alphabet[rng() % na];
                break;
            case 1: {
                unsigned n = 1 + rng() % 4;
                for (unsigned j = 0; j < n; j++)
                    o += alphabet[rng() % na];
                o += s[i];
                break;
            }
            default:
                i += rng() % 4; // deletes 1..4
                break;
            }
        } else
            o += s[i];
    }
    if (o.empty())
        o = "A";
    return o;
}

static void print_pair(const std::string& a, const std::string& b, bool& first)
{
    NWAlignment aln;
    unsigned m = alignGlobal(a, b, aln);
    printf("%s{\"a\":\"%s\",\"b\":\"%s\",\"consensus\":\"%s\",\"matches\":%u}", first ? "" : ",\n", a.c_str(),
           b.c_str(), aln.match_align.c_str(), m);
    first = false;
}

int main()
{
    printf("{\"pairs\":[\n");
    bool first = true;
    // hand cases: ends, single characters, pure gaps, equal strings
    const char* hand[][2] = {
        { "A", "A" }, { "A", "C" }, { "A", "ACGT" }, { "ACGT", "T" }, { "ACGTACGT", "ACGTACGT" },
        { "AAAAAAAA", "AAAA" }, { "ACGTTTTTTTTACGT", "ACGTACGT" }, { "ACGTACGT", "ACGTTTTTTTTACGT" },
        { "GATTACA", "GCATGCT" }, { "ACGTNACGT", "ACGTRACGT" }, { "acgtacgt", "ACGTACGT" },
        { "ACMGT", "ACAGT" }, { "ACMGT", "ACGGT" }, { "NNNN", "ACGT" }, { "aCgT", "AcGt" },
        { "ACGTACGTACGTACGTACGT", "TGCATGCATGCATGCA" },
    };
    for (auto& h : hand)
        print_pair(h[0], h[1], first);
    for (int rep = 0; rep < 160; rep++) {
        const char* alpha = rep % 4 == 0 ? kMixed : rep % 4 == 1 ? kIupac : kAcgt;
        size_t len = 1 + rng() % (rep < 120 ? 120 : 420);
        unsigned lc = rep % 5 == 4 ? 1 + rng() % 6 : 0;
        std::string a = random_seq(len, alpha, lc);
        std::string b = rep % 7 == 6 ? random_seq(1 + rng() % 150, alpha, lc) : mutate(a, 2 + rng() % 30, alpha);
        print_pair(a, b, first);
    }
    printf("\n],\n\"multi\":[\n");
    first = true;
    for (int rep = 0; rep < 60; rep++) {
        // 2..10 connecting sequences: noisy copies of one path (upper-case ACGT, as DBG paths are)
        unsigned n = 2 + rng() % 9;
        std::string base = random_seq(30 + rng() % 300, kAcgt, rep % 4 == 3 ? 1 + rng() % 5 : 0);
        std::vector<Sequence> seqs;
        for (unsigned i = 0; i < n; i++)
            seqs.push_back(i == 0 && rep % 2 ? base : mutate(base, 1 + rng() % 15, kAcgt));
        NWAlignment aln;
        std::pair<unsigned, unsigned> r = align(seqs, aln);
        printf("%s{\"seqs\":[", first ? "" : ",\n");
        first = false;
        for (unsigned i = 0; i < n; i++)
            printf("%s\"%s\"", i ? "," : "", seqs[i].c_str());
        printf("],\"consensus\":\"%s\",\"matches\":%u,\"size\":%u}", aln.match_align.c_str(), r.first, r.second);
    }
    printf("\n],\n\"revcomp\":[\n");
    first = true;
    for (int rep = 0; rep < 24; rep++) {
        std::string s = random_seq(1 + rng() % 80, rep % 2 ? kMixed : kAcgt, 0);
        printf("%s{\"seq\":\"%s\",\"rc\":\"%s\"}", first ? "" : ",\n", s.c_str(), reverseComplement(s).c_str());
        first = false;
    }
    // every pair of the 15 IUPAC letters + N, both cases
    const std::string codes = "ACGTMRWSYKVHDBNacgtmrwsykvhdbn";
    printf("\n],\n\"codes\":\"%s\",\n\"amb_or\":[", codes.c_str());
    for (size_t i = 0; i < codes.size(); i++) {
        std::string row;
        for (size_t j = 0; j < codes.size(); j++)
            row += ambiguityOr(codes[i], codes[j]);
        printf("%s\"%s\"", i ? "," : "", row.c_str());
    }
    printf("],\n\"amb_and\":[");
    for (size_t i = 0; i < codes.size(); i++) {
        std::string row;
        for (size_t j = 0; j < codes.size(); j++)
            row += ambiguityAnd(codes[i], codes[j]);
        printf("%s\"%s\"", i ? "," : "", row.c_str());
    }
    printf("]}\n");
    return 0;
}
