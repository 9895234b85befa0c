// Sanitizer fuzz of csrc/fast_inflate.inc against zlib (host only):
//   g++ -O1 -g -fsanitize=address,undefined -std=c++17 -o /tmp/fuzz_inflate tools/fuzz_inflate.cpp -lz && /tmp/fuzz_inflate 60000
// Streams come from zlib at random levels/strategies/window sizes, two thirds of them damaged
// (bit flips, truncation); buffers are exactly as large as the decoder's contract allows.
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
namespace {
#include "../well_duplicates_amd/csrc/fast_inflate.inc"
}
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    long ok = 0, declined = 0, mism = 0;
    for (int it = 0; it < iters; it++) {
        size_t n = rnd() % 20000;
        std::vector<uint8_t> raw(n);
        int kind = rnd() % 4;
        for (size_t i = 0; i < n; i++)
            raw[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? (uint8_t)(rnd() % 4 + 0x70) : kind == 2 ? (uint8_t)((i / 37) & 0xFF) : (uint8_t)((rnd() % 7 == 0) ? rnd() : 0x41);
        z_stream zs; memset(&zs, 0, sizeof(zs));
        int level = rnd() % 10, strat = (int[]){Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED}[rnd() % 5];
        deflateInit2(&zs, level, Z_DEFLATED, 31, 1 + rnd() % 9, strat);
        std::vector<uint8_t> comp(deflateBound(&zs, n) + 64);
        zs.next_in = raw.data(); zs.avail_in = n; zs.next_out = comp.data(); zs.avail_out = comp.size();
        deflate(&zs, Z_FINISH);
        size_t clen = comp.size() - zs.avail_out;
        deflateEnd(&zs);
        bool corrupt = rnd() % 3 != 0;
        if (corrupt) {
            int flips = 1 + rnd() % 3;
            for (int f = 0; f < flips; f++) comp[rnd() % clen] ^= 1u << (rnd() % 8);
            if (rnd() % 5 == 0) clen = rnd() % (clen + 1);
        }
        // exact-size heap buffers so that the sanitizer sees any access beyond the contract
        uint8_t *in = (uint8_t *)malloc(clen + 16); if (clen) memcpy(in, comp.data(), clen); memset(in + clen, 0, 16);
        size_t cap = n + 274 + (rnd() % 3 == 0 ? 0 : rnd() % 100);
        uint8_t *out = (uint8_t *)malloc(cap + 320);
        size_t prod = 0;
        bool r = fast_gunzip(in, clen, out, cap, &prod);
        if (r) {
            ok++;
            if (!corrupt && (prod != n || memcmp(out, raw.data(), n))) { mism++; printf("MISMATCH it=%d\n", it); }
            if (corrupt) {      // accepted: must equal what zlib makes of it
                std::vector<uint8_t> ref(cap + 320);
                z_stream is; memset(&is, 0, sizeof(is)); inflateInit2(&is, 31);
                is.next_in = in; is.avail_in = clen; is.next_out = ref.data(); is.avail_out = ref.size();
                int zr = inflate(&is, Z_FINISH);
                size_t zn = ref.size() - is.avail_out; inflateEnd(&is);
                if (zr != Z_STREAM_END || zn != prod || memcmp(ref.data(), out, prod)) { mism++; printf("ACCEPTED-BAD it=%d\n", it); }
            }
        } else {
            declined++;
            if (!corrupt) { mism++; printf("DECLINED-GOOD it=%d n=%zu level=%d strat=%d\n", it, n, level, strat); }
        }
        free(in); free(out);
    }
    printf("ok=%ld declined=%ld problems=%ld\n", ok, declined, mism);
    return mism != 0;
}
