// CPU check of csrc/lev2_stream.inc (the streaming closed form behind the Levenshtein <= 2 scan)
// against the textbook edit distance: g++ -O2 -std=c++17 tools/lev2_stream_check.cpp -o check && ./check [seed] [cases]
//   * every split of a read into rounds of 1..8 cycles must give the same verdict;
//   * the final verdict and distance equal min(levenshtein, 3) for equal-length reads over {A,C,G,T,N};
//   * a pair within distance 2 is alive after every prefix (the scan prunes on that);
//   * lev2_codes4 on four raw bytes equals four lev2_code calls.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#define WD_HD
#include "../well_duplicates_amd/csrc/lev2_stream.inc"

static int textbook(const std::vector<uint8_t> &a, const std::vector<uint8_t> &b)
{
    const size_t n = a.size(), m = b.size();
    std::vector<int> prev(m + 1), cur(m + 1);
    for (size_t j = 0; j <= m; j++)
        prev[j] = (int)j;
    for (size_t i = 1; i <= n; i++) {
        cur[0] = (int)i;
        for (size_t j = 1; j <= m; j++)
            cur[j] = std::min({prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1] ? 1 : 0)});
        std::swap(prev, cur);
    }
    return prev[m];
}

// raw BCL byte of a symbol 0..4 (4 = no-call): base in the low two bits, some quality above
static uint8_t raw_byte(int sym, std::mt19937 &rng) { return sym == 4 ? 0 : (uint8_t)(sym | ((1 + rng() % 40) << 2)); }

int main(int argc, char **argv)
{
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const long cases = argc > 2 ? atol(argv[2]) : 200000;
    std::mt19937 rng(seed);
    long bad = 0, within = 0;
    for (long c = 0; c < cases && bad < 10; c++) {
        const int L = 1 + (int)(rng() % 40);
        const int alpha = (rng() % 4 == 0) ? 2 : 5;              // small alphabets make shifts and repeats likely
        std::vector<uint8_t> a((size_t)L), b;
        for (auto &v : a)
            v = (uint8_t)(rng() % alpha == 4 ? 4 : rng() % std::min(alpha, 4));
        b = a;
        switch (rng() % 4) {
        case 0:
            for (int q = (int)(rng() % 4); q > 0; q--)
                b[rng() % L] = (uint8_t)(rng() % 5);
            break;
        case 1:
            if (L >= 2) {
                b.erase(b.begin() + (long)(rng() % L));
                b.insert(b.begin() + (long)(rng() % L), (uint8_t)(rng() % 5));
                if (rng() & 1)
                    b[rng() % L] = (uint8_t)(rng() % 5);
            }
            break;
        case 2:
            for (auto &v : b)
                v = (uint8_t)(rng() % alpha == 4 ? 4 : rng() % std::min(alpha, 4));
            break;
        default:                                                // two shifted stretches: distance 4 that looks like 2 locally
            if (L >= 6) {
                b.erase(b.begin() + 1);
                b.insert(b.begin() + L / 2, (uint8_t)(rng() % 5));
                b.erase(b.begin() + L / 2 + 1);
                b.push_back((uint8_t)(rng() % 5));
            }
        }
        const int want = std::min(textbook(a, b), 3);
        within += want <= 2;
        std::vector<uint8_t> ra((size_t)L), rb((size_t)L);
        for (int i = 0; i < L; i++) {
            ra[(size_t)i] = raw_byte(a[(size_t)i], rng);
            rb[(size_t)i] = raw_byte(b[(size_t)i], rng);
        }
        for (int split = 0; split < 4; split++) {
            uint32_t st = lev2_init();
            int j = 0;
            bool alive_all = true;
            while (j < L) {
                int n = split == 0 ? 1 : split == 1 ? 8 : 1 + (int)(rng() % 8);
                n = std::min(n, L - j);
                uint32_t wa = 0, wb = 0;
                if (n == 4 && split == 2) {                      // the interleaved layout's dword path
                    uint32_t da = 0, db = 0;
                    for (int q = 0; q < 4; q++) {
                        da |= (uint32_t)ra[(size_t)(j + q)] << (8 * q);
                        db |= (uint32_t)rb[(size_t)(j + q)] << (8 * q);
                    }
                    wa = lev2_codes4(da);
                    wb = lev2_codes4(db);
                } else {
                    for (int q = 0; q < n; q++) {
                        wa |= lev2_code(ra[(size_t)(j + q)]) << (4 * q);
                        wb |= lev2_code(rb[(size_t)(j + q)]) << (4 * q);
                    }
                }
                // (garbage above the round's fields must not matter)
                if (n < 8 && (rng() & 1)) {
                    wa |= (uint32_t)rng() << (4 * n);
                    wb |= (uint32_t)rng() << (4 * n);
                }
                st = lev2_step(st, wa, wb, n);
                j += n;
                alive_all = alive_all && lev2_alive(st);
            }
            const int got = lev2_alive(st) ? lev2_dist(st) : 3;
            if (got != want || (want <= 2 && !alive_all)) {
                bad++;
                fprintf(stderr, "MISMATCH case %ld split %d: L %d want %d got %d alive_all %d\n", c, split, L, want, got, (int)alive_all);
            }
        }
    }
    // lev2_codes4 against lev2_code on every byte value in every position
    for (uint32_t v = 0; v < 256; v++)
        for (int q = 0; q < 4; q++) {
            const uint32_t others = 0x9C004D01u & ~(0xFFu << (8 * q));
            const uint32_t w = others | (v << (8 * q));
            uint32_t want = 0;
            for (int r = 0; r < 4; r++)
                want |= lev2_code((w >> (8 * r)) & 0xFFu) << (4 * r);
            if (lev2_codes4(w) != want) {
                bad++;
                fprintf(stderr, "MISMATCH codes4 %08x\n", w);
            }
        }
    printf("%ld cases (%ld within distance 2), %ld disagreements\n", cases, within, bad);
    return bad ? 1 : 0;
}
