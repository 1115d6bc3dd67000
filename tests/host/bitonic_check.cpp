// Host emulation of the per-tile depth sort (litegs_amd/csrc/lg_tilesort_body.h compiled with LG_TILESORT_HOST): the three regimes of
// tilesort.hip run sequentially on the CPU and are compared with std::stable_sort for every list length 2..700, a sweep of lengths up
// to 20 000 (all three regimes, non powers of two, chunk edges) and duplicate-heavy depths (ties must keep ascending id order).
// Built and run by tests/test_tilesort_host.py:  g++ -O2 -std=c++17 -I litegs_amd/csrc tests/host/bitonic_check.cpp
#define LG_TILESORT_HOST
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "lg_tilesort_body.h"

static int check(int n, int n_splats, int distinct_depths, std::mt19937& rng, bool radix = false, bool any_order = false)
{
    std::vector<float> depth(n_splats);
    std::uniform_real_distribution<float> ud(0.01f, 40.0f);
    std::vector<float> palette(distinct_depths > 0 ? distinct_depths : 0);
    for (auto& p : palette) p = ud(rng);
    for (auto& d : depth) d = distinct_depths > 0 ? palette[rng() % distinct_depths] : ud(rng);
    if (n_splats > 3) { depth[1] = -0.5f; depth[2] = -0.0f; depth[3] = 0.0f; }       // negative / signed-zero keys
    // a tile list: n distinct ids, ascending (the emission order)
    std::vector<int> ids(n_splats);
    for (int i = 0; i < n_splats; i++) ids[i] = i;
    std::shuffle(ids.begin(), ids.end(), rng);
    std::vector<int> v(ids.begin(), ids.begin() + n);
    std::sort(v.begin(), v.end());
    std::vector<int> want = v;
    auto bits = [&](int id) { uint32_t u; std::memcpy(&u, &depth[id], 4); return u; };
    std::stable_sort(want.begin(), want.end(), [&](int a, int b) { return ts_depth_key(bits(a)) < ts_depth_key(bits(b)); });
    if (any_order) std::shuffle(v.begin(), v.end(), rng);          // the tile scatter's arrival order: the result must not depend on it
    std::vector<uint64_t> sk(TS_CHUNK);
    std::vector<uint32_t> dk(n);
    const bool small = n <= TS_SMALL;
    if (radix) {
        std::vector<uint32_t> exch(TS_RADIX_MAX);
        std::vector<int> cnt(256);
        if (any_order) ts_radix_sort_tile<false, true>(v.data(), n, exch.data(), cnt.data(), bits);
        else ts_radix_sort_tile<false>(v.data(), n, exch.data(), cnt.data(), bits);
    } else {
        ts_sort_tile(v.data(), n, sk.data(), dk.data(), bits, small ? 64 : 256, small);
    }
    for (int i = 0; i < n; i++)
        if (v[i] != want[i]) { std::printf("MISMATCH n=%d at %d: got %d want %d\n", n, i, v[i], want[i]); return 1; }
    return 0;
}

int main()
{
    std::mt19937 rng(12345);
    int bad = 0, cases = 0;
    for (int n = 2; n <= 700 && !bad; n++) { bad |= check(n, n + 17, 0, rng); cases++; }
    const int sweep[] = {1023, 1024, 1025, 2047, 2048, 2049, 2050, 3000, 4095, 4096, 4097, 5000, 6143, 6144, 6145, 8191, 8192, 8193, 10000, 16384, 16385, 20000};
    for (int n : sweep) { if (bad) break; bad |= check(n, n + 100, 0, rng); cases++; }
    for (int n : {2, 3, 64, 65, 511, 512, 513, 700, 2048, 2049, 5000, 9000}) {       // few distinct depths: ties everywhere
        if (bad) break;
        bad |= check(n, n + 5, 3, rng); cases++;
        bad |= check(n, n + 5, 1, rng); cases++;
    }
    // regime R (wave-private LSD radix sort, lists up to TS_RADIX_MAX): every length, few distinct depths (ties), one depth (all passes skipped)
    for (int n = 2; n <= TS_RADIX_MAX && !bad; n++) { bad |= check(n, n + 9, 0, rng, true); cases++; }
    for (int n : {2, 63, 64, 65, 128, 500, 1000, 1023, 1024}) {
        if (bad) break;
        bad |= check(n, n + 5, 3, rng, true); cases++;
        bad |= check(n, n + 5, 1, rng, true); cases++;
        bad |= check(n, n + 5, 200, rng, true); cases++;
    }
    // regime R on lists that arrive in ARBITRARY order (tile scatter through atomic cursors): equal depths must still end in ascending id order
    for (int n = 2; n <= TS_RADIX_MAX && !bad; n += (n < 130 ? 1 : 7)) { bad |= check(n, n + 9, 0, rng, true, true); cases++; }
    for (int n : {2, 3, 5, 63, 64, 65, 128, 129, 500, 1000, 1023, 1024}) {
        if (bad) break;
        bad |= check(n, n + 5, 3, rng, true, true); cases++;           // long runs of equal depth
        bad |= check(n, n + 5, 1, rng, true, true); cases++;           // one depth: every pass skipped, the repair does all the work
        bad |= check(n, n + 5, 200, rng, true, true); cases++;         // short runs
        bad |= check(n, n + 5, n / 2 + 1, rng, true, true); cases++;   // pairs (duplicated Gaussians)
    }
    // regimes M / L sort by the 64-bit key (depth, id): any arrival order
    for (int n : {1025, 2048, 2049, 5000, 9000}) { if (bad) break; bad |= check(n, n + 5, 3, rng, false, true); bad |= check(n, n + 5, 0, rng, false, true); cases += 2; }
    // the flip/step pair generators enumerate disjoint pairs covering [0, P)
    for (int lp = 1; lp <= 12 && !bad; lp++) {
        const int P = 1 << lp;
        for (int lk = 1; lk <= lp; lk++) {
            std::vector<int> seen(P, 0);
            for (int t = 0; t < P / 2; t++) { int i, p; lg_bitonic_flip_pair(t, lk, i, p); if (i >= p || p >= P) bad = 1; else { seen[i]++; seen[p]++; } }
            for (int x : seen) if (x != 1) bad = 1;
            for (int lj = lk - 2; lj >= 0; lj--) {
                std::fill(seen.begin(), seen.end(), 0);
                for (int t = 0; t < P / 2; t++) { int i, p; lg_bitonic_step_pair(t, lj, i, p); if (p != i + (1 << lj) || p >= P) bad = 1; else { seen[i]++; seen[p]++; } }
                for (int x : seen) if (x != 1) bad = 1;
            }
        }
    }
    std::printf("%s: %d list cases\n", bad ? "FAIL" : "OK", cases);
    return bad;
}
