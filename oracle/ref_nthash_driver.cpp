// oracle/ref_nthash_driver.cpp -- TEST INFRASTRUCTURE.
// Known-answer generator that includes the REFERENCE's own ntHash header from
// where it lies (/root/reference/subprojects/ntedit/lib/nthash.hpp; passed with
// -I by oracle/Makefile, never copied) and prints vectors as JSON.  Built only
// in the authoring container (the GPU box has no /root/reference); its output
// is committed as tests/golden/nthash_kat.json by tests/golden/make_nthash_kat.py.
// the header relies on its includer for <vector>/<string>, exactly as
// ntedit.cpp:4-24 includes them first
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lib/nthash.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rng()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main()
{
    const unsigned ks[] = { 20, 24, 28, 32, 7, 64 };
    const char alphabet[] = "ACGTACGTACGTACGTacgtNRYKMSWBDHVn";
    printf("[\n");
    bool first = true;
    for (unsigned k : ks) {
        for (int rep = 0; rep < 6; rep++) {
            size_t len = k + 40 + rng() % 30;
            std::string s(len, 'A');
            // rep 0..2: pure upper-case ACGT; 3: with lower case; 4,5: with IUPAC / N
            for (auto& c : s) {
                unsigned r = rng() % (rep < 3 ? 16 : rep == 3 ? 20 : 32);
                c = alphabet[r];
            }
            const unsigned m = 4;
            std::vector<uint64_t> hv(m);
            uint64_t fh = 0, rh = 0;
            if (!first)
                printf(",\n");
            first = false;
            printf("{\"k\":%u,\"seq\":\"%s\",\n \"fwd\":[", k, s.c_str());
            // from-scratch forward / reverse / canonical+extensions at each position
            for (size_t p = 0; p + k <= len; p++)
                printf("%s%llu", p ? "," : "", (unsigned long long)NTF64(s.c_str() + p, k));
            printf("],\n \"rev\":[");
            for (size_t p = 0; p + k <= len; p++)
                printf("%s%llu", p ? "," : "", (unsigned long long)NTR64(s.c_str() + p, k));
            printf("],\n \"rolled\":[");
            // rolling canonical multi-hash (NTMC64 init then roll), 4 hashes each
            NTMC64(s.c_str(), k, m, fh, rh, hv.data());
            for (size_t p = 0; p + k <= len; p++) {
                if (p)
                    NTMC64((unsigned char)s[p - 1], (unsigned char)s[p + k - 1], k, m, fh, rh,
                           hv.data());
                printf("%s[%llu,%llu,%llu,%llu]", p ? "," : "", (unsigned long long)hv[0],
                       (unsigned long long)hv[1], (unsigned long long)hv[2],
                       (unsigned long long)hv[3]);
            }
            printf("],\n \"changelast\":[");
            // change the last base of the first k-mer to each of A,C,G,T,a,N
            const char repl[] = "ACGTaNY";
            for (size_t r = 0; r < strlen(repl); r++) {
                uint64_t f2 = NTF64(s.c_str(), k), r2 = NTR64(s.c_str(), k);
                NTMC64_changelast((unsigned char)s[k - 1], (unsigned char)repl[r], k, m, f2, r2,
                                  hv.data());
                printf("%s[\"%c\",%llu,%llu,%llu,%llu,%llu,%llu]", r ? "," : "", repl[r],
                       (unsigned long long)f2, (unsigned long long)r2, (unsigned long long)hv[0],
                       (unsigned long long)hv[1], (unsigned long long)hv[2],
                       (unsigned long long)hv[3]);
            }
            printf("]}");
        }
    }
    printf("\n]\n");
    return 0;
}
