// The host pipeline's inner loops alone, single-threaded, on a pre-generated stream: the block twist, the Fisher-Yates target scan
// (masked rejection, red_blue.py:80) and the word copy -- what each costs per step of N walkers without any other thread around.
// build: /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -ffp-contract=off -pthread -I include tools/ubench/mt_scan_bench.cpp -o tools/ubench/bin/mt_scan_bench
#include "../../emcee_amd/csrc/emx_mtpipe.cpp"
using namespace emx;
int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 65536;
    const int reps = 200;
    WordStream ws;
    ws.nblk = 4096;
    ws.own.resize(ws.nblk * BLK + RING_MIRROR + 16);
    ws.ring = reinterpret_cast<uint32_t*>(((uintptr_t)ws.own.data() + 63) & ~(uintptr_t)63);
    uint32_t seed[BLK];
    uint32_t x = 12345u;
    for (int i = 0; i < BLK; ++i) { x = x * 1664525u + 1013904223u; seed[i] = x; }
    const TwistFn twist = pick_twist(false);
    alignas(64) uint32_t pkey[2][BLK + 16];
    memcpy(pkey[0], seed, BLK * 4);
    memcpy(ws.ring, seed, BLK * 4);
    uint64_t t0 = now_ns();
    for (int r = 0; r < 8; ++r)
        for (uint64_t b = 1; b < ws.nblk; ++b) twist(pkey[(b - 1) & 1], pkey[b & 1], &ws.ring[b * BLK]);
    uint64_t t1 = now_ns();
    printf("twist + copy out: %.1f ns/block, %.4f ns/word -> %.1f us per step of %lld walkers (6.4 N words)\n", (t1 - t0) / (8.0 * (ws.nblk - 1)),
           (t1 - t0) / (8.0 * (ws.nblk - 1) * BLK), (t1 - t0) / (8.0 * (ws.nblk - 1) * BLK) * 6.4 * N * 1e-3, (long long)N);
    ws.produced.store(ws.nblk);
    std::atomic<bool> stop{false};
    std::vector<uint32_t> jr((size_t)N + 16);
    printf("pick_compaction() = %d\n", pick_compaction());
    for (int vec = 3; vec >= 0; --vec) {
        if (vec) g_compaction = vec - 1;
        uint64_t words = 0, best = ~0ull;
        for (int r = 0; r < reps; ++r) {
            Reader rd;
            rd.ws = &ws;
            rd.stop = &stop;
            rd.seek((uint64_t)(r % 7) * 1000 + 1);
            const uint64_t a0 = rd.pos();
            const uint64_t s0 = now_ns();
            rd.shuffle_targets(jr.data(), N, vec != 0);
            best = std::min<uint64_t>(best, now_ns() - s0);
            words = rd.pos() - a0;
        }
        printf("shuffle scan (%s): %.1f us best of %d, %llu words (%.3f ns/word)\n", vec == 1 ? "avx512, vpcompressd" : vec == 2 ? "avx512, table permutes" : vec == 3 ? "avx512, no compaction" : "scalar", best * 1e-3, reps, (unsigned long long)words, (double)best / words);
    }
    std::vector<uint32_t> dst((size_t)5 * N);
    uint64_t best = ~0ull;
    for (int r = 0; r < reps; ++r) {
        const uint64_t s0 = now_ns();
        memcpy(dst.data(), ws.ring + 1000, (size_t)5 * N * 4);
        best = std::min<uint64_t>(best, now_ns() - s0);
    }
    printf("copy of 5 N words: %.1f us (%.1f GB/s)\n", best * 1e-3, 5.0 * N * 4 / best);
    return 0;
}
// ---- pieces of the scan loop, one at a time (which instruction group costs what on this core) ----
__attribute__((target("avx512f,avx512vl,avx512bw,avx2,popcnt"))) static void pieces(const uint32_t* p, size_t n) {
    const __m512i vmask = _mm512_set1_epi32(65535);
    for (int piece = 0; piece < 6; ++piece) {
        uint64_t best = ~0ull;
        unsigned long long sink = 0;
        for (int rep = 0; rep < 50; ++rep) {
            const uint64_t t0 = now_ns();
            __m512i acc = _mm512_setzero_si512();
            int64_t i = 60000;
            unsigned cnt = 0;
            for (size_t u = 0; u + 64 <= n; u += 64) {
                __m512i v[4];
                for (int k = 0; k < 4; ++k) v[k] = _mm512_loadu_si512(p + u + 16 * k);
                if (piece >= 1)
                    for (int k = 0; k < 4; ++k) v[k] = _mm512_and_si512(temper_v(v[k]), vmask);
                if (piece < 2) {
                    for (int k = 0; k < 4; ++k) acc = _mm512_xor_si512(acc, v[k]);
                    continue;
                }
                const __m512i hi = _mm512_set1_epi32((int)(uint32_t)i), lo = _mm512_set1_epi32((int)(uint32_t)(i - 128));
                __mmask16 a[4], r[4];
                for (int k = 0; k < 4; ++k) {
                    a[k] = _mm512_cmple_epu32_mask(v[k], lo);
                    if (piece >= 3) r[k] = _mm512_cmpgt_epu32_mask(v[k], hi);
                }
                if (piece == 2) {
                    cnt += (unsigned)a[0] ^ (unsigned)a[1] ^ (unsigned)a[2] ^ (unsigned)a[3];
                    continue;
                }
                if (piece >= 4) {
                    const int c = __builtin_popcount((unsigned)a[0]) + __builtin_popcount((unsigned)a[1]) + __builtin_popcount((unsigned)a[2]) + __builtin_popcount((unsigned)a[3]);
                    if (piece >= 5) i -= (c & 1);          // (the loop carries i, slowly)
                    cnt += (unsigned)c;
                }
                cnt += (unsigned)((a[0] | r[0]) & (a[1] | r[1]) & (a[2] | r[2]) & (a[3] | r[3]));
            }
            sink += cnt + (unsigned)_mm512_reduce_add_epi32(acc);
            best = std::min<uint64_t>(best, now_ns() - t0);
        }
        static const char* names[] = {"loads", "+ temper, mask", "+ 4 compares (accept)", "+ 4 compares (reject), mask logic", "+ popcounts", "+ thresholds from a changing i"};
        printf("pieces: %-40s %.3f ns/word  (%llu)\n", names[piece], (double)best / (double)(n / 64 * 64), sink & 1);
    }
}
struct PiecesRun {
    PiecesRun() {
        std::vector<uint32_t> w(90880);
        uint32_t x = 7u;
        for (auto& v : w) { x = x * 1664525u + 1013904223u; v = x; }
        pieces(w.data(), w.size());
    }
} g_pieces_run;
