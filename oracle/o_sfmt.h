/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_sfmt.h: the reference's random generator, SFMT19937 (Saito & Matsumoto), restated from its
 * published recursion with the reference's specific seeding (file:line under /root/reference):
 *   src/libcore/random.cpp:68-99      parameters (MEXP 19937, POS1 122, SL1 18, SL2 1, SR1 11, SR2 1, masks, parity)
 *   src/libcore/random.cpp:130-205    128-bit shifts, do_recursion
 *   src/libcore/random.cpp:288-297    gen_rand64
 *   src/libcore/random.cpp:300-390    period_certification, gen_rand_all
 *   src/libcore/random.cpp:397-407    init_gen_rand: 64-bit LCG seeding (6364136223846793005)
 *   src/libcore/random.cpp:409-468    init_by_array
 *   src/libcore/random.cpp:528-548    seed(Random*) = init_by_array over 312 nextULong()
 *   src/libcore/random.cpp:632-641    nextFloat: low 32 bits >> 9 | 0x3f800000, minus 1
 *   include/mitsuba/core/random.h:113 default seed 5489
 * Pinned by the golden vector of src/tests/test_random.cpp:433-473 (tests/golden/sfmt_seed4321.json).
 */
#pragma once
#include <cstdint>
#include <cstring>

namespace orc {

class SFMT {
public:
    enum { N = 19937 / 128 + 1, N32 = N * 4, N64 = N * 2 };

    explicit SFMT(uint64_t seed = 5489ULL) { initGenRand(seed); }

    void seed(uint64_t s) { initGenRand(s); }

    /* random.cpp:528-548 */
    void seedFrom(SFMT &parent) {
        uint64_t buf[N64];
        for (int i = 0; i < N64; ++i) buf[i] = parent.nextULong();
        initByArray(reinterpret_cast<const uint32_t *>(buf), N64 * 2);
    }

    uint64_t nextULong() {
        if (idx >= N32) { genRandAll(); idx = 0; }
        uint64_t r;
        memcpy(&r, &st[idx], 8);
        idx += 2;
        return r;
    }

    float nextFloat() {
        uint32_t u = (uint32_t) ((nextULong() & 0xFFFFFFFFULL) >> 9) | 0x3f800000u;
        float f; memcpy(&f, &u, 4);
        return f - 1.0f;
    }

private:
    uint32_t st[N32];
    int idx;

    static uint32_t func1(uint32_t x) { return (x ^ (x >> 27)) * (uint32_t) 1664525UL; }
    static uint32_t func2(uint32_t x) { return (x ^ (x >> 27)) * (uint32_t) 1566083941UL; }

    void periodCertification() {
        static const uint32_t parity[4] = { 0x00000001U, 0x00000000U, 0x00000000U, 0x13c9e684U };
        uint32_t inner = 0;
        for (int i = 0; i < 4; i++) inner ^= st[i] & parity[i];
        for (int i = 16; i > 0; i >>= 1) inner ^= inner >> i;
        inner &= 1;
        if (inner == 1) return;
        for (int i = 0; i < 4; i++) {
            uint32_t work = 1;
            for (int j = 0; j < 32; j++) {
                if ((work & parity[i]) != 0) { st[i] ^= work; return; }
                work = work << 1;
            }
        }
    }

    void initGenRand(uint64_t seed) {
        uint64_t s64[N64];
        s64[0] = seed;
        for (int i = 1; i < N64; ++i)
            s64[i] = 6364136223846793005ULL * (s64[i - 1] ^ (s64[i - 1] >> 62)) + (uint64_t) i;
        memcpy(st, s64, sizeof(st));
        idx = N32;
        periodCertification();
    }

    void initByArray(const uint32_t *init_key, int key_length) {
        int i, j, count;
        uint32_t r;
        const int size = N * 4;
        const int lag = size >= 623 ? 11 : (size >= 68 ? 7 : (size >= 39 ? 5 : 3));
        const int mid = (size - lag) / 2;
        memset(st, 0x8b, sizeof(st));
        count = (key_length + 1 > N32) ? key_length + 1 : N32;
        r = func1(st[0] ^ st[mid] ^ st[N32 - 1]);
        st[mid] += r;
        r += (uint32_t) key_length;
        st[mid + lag] += r;
        st[0] = r;
        count--;
        for (i = 1, j = 0; (j < count) && (j < key_length); j++) {
            r = func1(st[i] ^ st[(i + mid) % N32] ^ st[(i + N32 - 1) % N32]);
            st[(i + mid) % N32] += r;
            r += init_key[j] + (uint32_t) i;
            st[(i + mid + lag) % N32] += r;
            st[i] = r;
            i = (i + 1) % N32;
        }
        for (; j < count; j++) {
            r = func1(st[i] ^ st[(i + mid) % N32] ^ st[(i + N32 - 1) % N32]);
            st[(i + mid) % N32] += r;
            r += (uint32_t) i;
            st[(i + mid + lag) % N32] += r;
            st[i] = r;
            i = (i + 1) % N32;
        }
        for (j = 0; j < N32; j++) {
            r = func2(st[i] + st[(i + mid) % N32] + st[(i + N32 - 1) % N32]);
            st[(i + mid) % N32] ^= r;
            r -= (uint32_t) i;
            st[(i + mid + lag) % N32] ^= r;
            st[i] = r;
            i = (i + 1) % N32;
        }
        idx = N32;
        periodCertification();
    }

    /* 128-bit helpers on 4 x u32 little-endian words */
    static void rshift128(uint32_t out[4], const uint32_t in[4], int shift) {
        uint64_t th = ((uint64_t) in[3] << 32) | in[2], tl = ((uint64_t) in[1] << 32) | in[0];
        uint64_t oh = th >> (shift * 8), ol = tl >> (shift * 8);
        ol |= th << (64 - shift * 8);
        out[1] = (uint32_t) (ol >> 32); out[0] = (uint32_t) ol; out[3] = (uint32_t) (oh >> 32); out[2] = (uint32_t) oh;
    }
    static void lshift128(uint32_t out[4], const uint32_t in[4], int shift) {
        uint64_t th = ((uint64_t) in[3] << 32) | in[2], tl = ((uint64_t) in[1] << 32) | in[0];
        uint64_t oh = th << (shift * 8), ol = tl << (shift * 8);
        oh |= tl >> (64 - shift * 8);
        out[1] = (uint32_t) (ol >> 32); out[0] = (uint32_t) ol; out[3] = (uint32_t) (oh >> 32); out[2] = (uint32_t) oh;
    }
    static void doRecursion(uint32_t r[4], const uint32_t a[4], const uint32_t b[4], const uint32_t c[4], const uint32_t d[4]) {
        static const uint32_t MSK[4] = { 0xdfffffefU, 0xddfecb7fU, 0xbffaffffU, 0xbffffff6U };
        uint32_t x[4], y[4];
        lshift128(x, a, 1);   /* SL2 */
        rshift128(y, c, 1);   /* SR2 */
        for (int i = 0; i < 4; ++i)
            r[i] = a[i] ^ x[i] ^ ((b[i] >> 11) & MSK[i]) ^ y[i] ^ (d[i] << 18);   /* SR1 = 11, SL1 = 18 */
    }
    void genRandAll() {
        const int POS1 = 122;
        uint32_t *r1 = &st[(N - 2) * 4], *r2 = &st[(N - 1) * 4];
        int i;
        for (i = 0; i < N - POS1; ++i) {
            uint32_t tmp[4];
            doRecursion(tmp, &st[i * 4], &st[(i + POS1) * 4], r1, r2);
            memcpy(&st[i * 4], tmp, 16);
            r1 = r2; r2 = &st[i * 4];
        }
        for (; i < N; ++i) {
            uint32_t tmp[4];
            doRecursion(tmp, &st[i * 4], &st[(i + POS1 - N) * 4], r1, r2);
            memcpy(&st[i * 4], tmp, 16);
            r1 = r2; r2 = &st[i * 4];
        }
    }
};

} // namespace orc
