// inflate_emu.cpp - runs well_duplicates_amd/csrc/gpu_inflate.inc on the CPU: 64 threads play the
// 64 lanes of the wave, every wave operation is a barrier.  Test infrastructure (not shipped, not
// linked into libwelldup.so): it lets the GPU decoder's control flow - sync passes, window caps,
// match resolution, every error exit - be run under a sanitizer and compared with zlib before a
// kernel is launched on a shared GPU.
//
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -pthread [-DEMU_WAVES=1|4] [-DEMU_SPAN=512|256] [-DEMU_OUTDIV=2|4]
//       tools/inflate_emu.cpp -lz -o inflate_emu
//   ./inflate_emu file.gz [...]           decode, compare with zlib, print the statistics
//   ./inflate_emu --fuzz SEED COUNT file.gz   corrupt the file COUNT times; the decoder must end with
//                                          a status and, when it says OK, agree with zlib
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

// ---- the workgroup, emulated: NW waves of 64 lanes, one thread each --------------------------
#ifndef EMU_WAVES
#define EMU_WAVES 4
#endif
#define WD_WAVE_OPS
#define WV_DEV inline
static pthread_barrier_t g_wbar[EMU_WAVES], g_gbar;
static uint32_t g_x[EMU_WAVES][64];
static thread_local int t_lane, t_wave;
inline void bar() { pthread_barrier_wait(&g_wbar[t_wave]); }
inline int wv_lane() { return t_lane; }
inline int wv_wave() { return t_wave; }
inline void wg_barrier() { pthread_barrier_wait(&g_gbar); }
inline void wv_atomic_or(uint32_t *p, uint32_t v) { __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline void wv_atomic_and(uint32_t *p, uint32_t v) { __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline unsigned long long wv_ballot(bool p)
{
    g_x[t_wave][t_lane] = p;
    bar();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++)
        m |= (unsigned long long)(g_x[t_wave][i] & 1) << i;
    bar();
    return m;
}
inline uint32_t wv_shfl(uint32_t v, int src)
{
    g_x[t_wave][t_lane] = v;
    bar();
    const uint32_t r = g_x[t_wave][src & 63];
    bar();
    return r;
}
inline uint32_t wv_shfl_up(uint32_t v, int d)
{
    g_x[t_wave][t_lane] = v;
    bar();
    const uint32_t r = t_lane >= d ? g_x[t_wave][t_lane - d] : v;
    bar();
    return r;
}
inline uint32_t wv_uniform(uint32_t v) { return wv_shfl(v, 0); }
inline uint32_t wv_readlane(uint32_t v, uint32_t src) { return wv_shfl(v, (int)src); }
inline void wv_sync() { bar(); }
#define WV_GLOBAL                        /* (the GPU build: address_space(1)) */
inline uint32_t wv_hist_byte(const uint8_t *p) { return *p; }
inline void wv_stores_done() {}
inline unsigned long long wv_clock() { return 0; }
inline unsigned long long wv_realtime() { return 0; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline uint32_t __brev(uint32_t v)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++, v >>= 1)
        r = (r << 1) | (v & 1);
    return r;
}

#include "../well_duplicates_amd/csrc/gpu_inflate.inc"

struct Job {
    const uint32_t *file;
    uint32_t file_bytes, stream_off;
    uint8_t *obase;
    uint32_t out_cap;
    InfResult res;
};
#ifndef EMU_SPAN
#define EMU_SPAN 512
#endif
#ifndef EMU_OUTDIV
#define EMU_OUTDIV 2
#endif
static InfLdsT<EMU_WAVES, EMU_SPAN, EMU_OUTDIV> g_lds;
static Job g_job;

static void *lane_main(void *arg)
{
    t_lane = (int)(intptr_t)arg & 63;
    t_wave = (int)(intptr_t)arg >> 6;
    inf_member(g_lds, g_job.file, g_job.file_bytes, g_job.stream_off, g_job.obase, g_job.out_cap, &g_job.res);
    return nullptr;
}

static void run_wave()
{
    pthread_t th[EMU_WAVES * 64];
    for (int i = 0; i < EMU_WAVES * 64; i++)
        pthread_create(&th[i], nullptr, lane_main, (void *)(intptr_t)i);
    for (int i = 0; i < EMU_WAVES * 64; i++)
        pthread_join(th[i], nullptr);
}

// zlib's view of the first member: rc (Z_STREAM_END = fine), output
static int zlib_member(const std::vector<uint8_t> &gz, std::vector<uint8_t> &out, size_t cap)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, 16 + MAX_WBITS);
    out.assign(cap + 1, 0);
    zs.next_in = const_cast<Bytef *>(gz.data());
    zs.avail_in = (uInt)gz.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    inflateEnd(&zs);
    return rc;
}

// returns 0 when the decoder and zlib agree (or the decoder declined), 1 on a disagreement
static int check(const std::vector<uint8_t> &gz, size_t cap, bool verbose)
{
    uint32_t off = 0;
    std::vector<uint8_t> ref;
    const int zrc = zlib_member(gz, ref, cap);
    if (!inf_gzip_header(gz.data(), gz.size(), &off)) {
        if (verbose)
            printf("header declined (zlib rc %d)\n", zrc);
        return 0;
    }
    std::vector<uint32_t> file((gz.size() + 3) / 4 + 1, 0xDEADBEEF);       // garbage after the end must not matter
    memcpy(file.data(), gz.data(), gz.size());
    std::vector<uint8_t> out(cap + 64, 0xEE);
    g_job = Job{file.data(), (uint32_t)gz.size(), off, out.data(), (uint32_t)cap, {}};
    run_wave();
    const InfResult &r = g_job.res;
    if (verbose)
        printf("status %u produced %u end_byte %u (file %zu) head %08x | blocks %u windows %u passes %u (%.2f/window) "
               "rounds %u | zlib rc %d out %zu\n", r.status, r.produced, r.end_byte, gz.size(), r.head, r.blocks,
               r.windows, r.passes, r.windows ? (double)r.passes / r.windows : 0.0, r.rounds, zrc, ref.size());
    for (size_t i = cap; i < out.size(); i++)
        if (out[i] != 0xEE) {
            printf("MISMATCH: wrote past the output's end at %zu\n", i);
            return 1;
        }
    if (r.status != INF_OK)
        return 0;                                              // declined: the host decoder takes over
    // OK means: a complete deflate stream was decoded.  zlib must agree up to the trailer check
    // (the CRC and length are compared by the caller, not by the kernel).
    if (r.produced > cap) {
        printf("MISMATCH: produced %u > cap %zu\n", r.produced, cap);
        return 1;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, -MAX_WBITS);                             // raw deflate: no trailer involved
    std::vector<uint8_t> raw(cap + 1);
    zs.next_in = const_cast<Bytef *>(gz.data() + off);
    zs.avail_in = (uInt)(gz.size() - off);
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    const int rrc = inflate(&zs, Z_FINISH);
    const size_t rn = zs.total_out, rin = zs.total_in;
    inflateEnd(&zs);
    if (rrc != Z_STREAM_END || rn != r.produced || off + rin != r.end_byte) {
        printf("MISMATCH: zlib raw rc %d out %zu in-end %zu, decoder out %u end %u\n", rrc, rn, off + rin, r.produced,
               r.end_byte);
        return 1;
    }
    uint8_t head[4];
    memcpy(head, &r.head, 4);
    for (size_t i = 0; i < rn; i++) {
        const uint8_t got = i < 4 ? head[i] : out[i];
        if (got != raw[i]) {
            printf("MISMATCH at output byte %zu: %02x vs zlib %02x\n", i, got, raw[i]);
            return 1;
        }
    }
    return 0;
}

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> d;
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
        exit(2);
    }
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0)
        d.insert(d.end(), buf, buf + k);
    fclose(f);
    return d;
}

int main(int argc, char **argv)
{
    for (auto &b : g_wbar)
        pthread_barrier_init(&b, nullptr, 64);
    pthread_barrier_init(&g_gbar, nullptr, EMU_WAVES * 64);
    int bad = 0;
    if (argc >= 5 && !strcmp(argv[1], "--fuzz")) {
        uint64_t seed = strtoull(argv[2], nullptr, 0);
        const int count = atoi(argv[3]);
        const std::vector<uint8_t> gz = slurp(argv[4]);
        std::vector<uint8_t> ref;
        zlib_member(gz, ref, 1u << 26);
        const size_t cap = ref.size();
        auto rnd = [&]() {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            return (uint32_t)(seed >> 33);
        };
        for (int it = 0; it < count; it++) {
            std::vector<uint8_t> m = gz;
            const int kind = (int)(rnd() % 4);
            if (kind == 0) {                                   // flip a few bits
                for (int k = 0, n = 1 + (int)(rnd() % 4); k < n; k++)
                    m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
            } else if (kind == 1) {                            // truncate
                m.resize(1 + rnd() % m.size());
            } else if (kind == 2) {                            // overwrite a run with noise
                const size_t at = rnd() % m.size(), n = 1 + rnd() % 64;
                for (size_t k = at; k < m.size() && k < at + n; k++)
                    m[k] = (uint8_t)rnd();
            } else {                                           // damage near the start (block headers)
                m[10 + rnd() % 64 % (m.size() - 10)] ^= (uint8_t)(1u << (rnd() % 8));
            }
            // room: sometimes exactly enough, sometimes too little
            const size_t c = (rnd() % 4 == 0) ? cap / 2 + 1 : cap;
            bad += check(m, c, false);
        }
        printf("fuzz: %d cases, %d disagreements\n", count, bad);
        return bad != 0;
    }
    for (int i = 1; i < argc; i++) {
        const std::vector<uint8_t> gz = slurp(argv[i]);
        std::vector<uint8_t> ref;
        zlib_member(gz, ref, 1u << 26);
        printf("%s: ", argv[i]);
        bad += check(gz, ref.size(), true);
    }
    return bad != 0;
}
