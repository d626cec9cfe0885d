// chain_emul.c -- CPU emulation (64 "lanes" as arrays) of sq_chain_spec, the segment-speculative exact evaluation of the
// rmsnorm sum-of-squares chain  l <- fma(x_k, x_k, l), k ascending  (one of the 4 strided lanes of simd::square_sum,
// reference src/platforms/arch/x86_simd.cpp:942-960).  Build-host check of the ALGORITHM (bit equality with the plain
// sequential chain on friendly and adversarial data) before it is trusted on the GPU, where tests/test_gpu_ops.py checks
// the kernel itself.  Not part of the product.
//   gcc -O2 -mfma -o /tmp/chain_emul tools/chain_emul.c -lm && /tmp/chain_emul
//
// Idea.  The terms are non-negative, so l only grows.  While l stays inside one binade [A, 2A), A = 2^E, every step rounds
// l + x^2 to a multiple of u = ulp(A), and the increment t = fl(l + x^2) - l does not depend on l (exact ties aside): it is
// t = fma(x, x, A) - A.  Lane L owns B consecutive elements.  (1) approximate prefix of sum x^2 -> the binade E_L each lane
// expects to start in; (2) T_L = sum of its increments against 2^E_L (multiples of u: exact in any order), tie flags;
// (3) exact exclusive prefix S_L of the T's in fp64 (all T are multiples of 2^(Emin-23) and the total is < 2^(Emax+2): exact
// when Emax - Emin <= 26); (4) rounds: with an exact (base lane, base value), lane t's start is base + (S_t - S_base); a
// lane is consistent if that start is in its expected binade, start + T stays below 2^(E+1), and it has no tie.  The first
// inconsistent lane f has an exact start (everything before it is consistent): it runs its B steps for real, and becomes
// the new base.  A binade is crossed ~log2(n) times, so a handful of rounds replace n dependent steps.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NL 64
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float chain_seq(const float* p, int n) { float l = 0.f; for (int k = 0; k < n; ++k) l = fmaf(p[k], p[k], l); return l; }

static int g_rounds, g_fallback;

// head: elements run in order before the speculative part (in lanes' worth)
static float chain_spec(const float* p, int n, int head_lanes) {
    const int B = (n + NL - 1) / NL;
    // ---- per lane: approximate sum of squares, all-zero flag
    float s[NL]; int allzero[NL];
    for (int L = 0; L < NL; ++L) {
        float a = 0.f; int z = 1; uint32_t orb = 0;
        for (int j = 0; j < B; ++j) { int k = L * B + j; float x = k < n ? p[k] : 0.f; a = fmaf(x, x, a); orb |= f2u(x); }
        z = (orb & 0x7fffffffu) == 0;                       // every element +-0 (the kernel: one OR over the bit patterns)
        s[L] = a; allzero[L] = z;
    }
    // inclusive scan (any order: approximate), exclusive prefix P
    float P[NL]; { float run = 0.f; for (int L = 0; L < NL; ++L) { P[L] = run; run += s[L]; } }
    // ---- head: lanes [0, head_lanes) in order
    int base = head_lanes < NL ? head_lanes : NL;
    float base_val = 0.f;
    for (int k = 0; k < base * B && k < n; ++k) base_val = fmaf(p[k], p[k], base_val);
    if (base >= NL) return base_val;
    // ---- speculative increments
    float T[NL], top[NL]; uint32_t Eb[NL]; int valid[NL], tie[NL];
    uint32_t emin = 0xffffffffu, emax = 0;
    for (int L = 0; L < NL; ++L) {
        const uint32_t eb = f2u(P[L]) & 0x7f800000u;
        valid[L] = (eb >= (27u << 23)) && (eb <= (250u << 23));       // P finite, normal, room for u/2 and 2A
        Eb[L] = eb; T[L] = 0.f; tie[L] = 0; top[L] = 0.f;
        if (!valid[L] || L < base) { valid[L] = valid[L] && L >= base; continue; }
        if (!allzero[L]) { if (eb < emin) emin = eb; if (eb > emax) emax = eb; }
        // round 4: T = chain(A) - A (inside the binade every step adds RN_u(x^2) whatever multiple of u it started from; a chain that leaves the binade
        // gives T >= A, which fails okTop); ties from the same chain run from A + u: without a tie the two end exactly u apart
        const float A = u2f(eb), u1 = u2f(eb - (23u << 23));
        top[L] = A + A;
        float ca = A, cb = A + u1;
        for (int j = 0; j < B; ++j) {
            int k = L * B + j; float x = k < n ? p[k] : 0.f;
            ca = fmaf(x, x, ca); cb = fmaf(x, x, cb);
        }
        T[L] = ca - A;
        tie[L] = (cb - ca) != u1;
    }
    // fp64 exactness of the prefix: all T multiples of 2^(emin-23), every partial sum < 2^(emax+2)
    if (emin != 0xffffffffu && ((emax - emin) >> 23) > 26) { ++g_fallback; return chain_seq(p, n); }
    for (int L = 0; L < NL; ++L) if (!(T[L] < INFINITY)) T[L] = 0.f, valid[L] = 0;      // (NaN / inf garbage never enters the prefix)
    double S[NL + 1]; S[0] = 0.0; for (int L = 0; L < NL; ++L) S[L + 1] = S[L] + (double)T[L];
    // ---- rounds
    double bv = (double)base_val, Sb = S[base];
    for (;;) {
        ++g_rounds;
        int f = NL;
        float st_f = 0.f;
        for (int L = base; L < NL; ++L) {
            const double d = bv + (S[L] - Sb);
            const float st = (float)d;
            const int exact = (double)st == d;
            const int okE = (f2u(st) & 0x7f800000u) == Eb[L];
            const int okTop = (st + T[L]) < top[L];
            (void)exact;                                    // (round 4: no exactness test -- the first inconsistent lane's start is exact by induction)
            const int ok = allzero[L] || (valid[L] && !tie[L] && okE && okTop);
            if (!ok) { f = L; st_f = st; (void)st_f; break; }
        }
        if (f == NL) { const double d = bv + (S[NL] - Sb); return (float)d; }
        // lane f: exact start (every lane in [base, f) is consistent), B real steps
        const double d = bv + (S[f] - Sb);
        float l = (float)d;     // exact: it IS the chain value there
        for (int j = 0; j < B; ++j) { int k = f * B + j; float x = k < n ? p[k] : 0.f; l = fmaf(x, x, l); }
        if (f == NL - 1) return l;
        base = f + 1; bv = (double)l; Sb = S[base];
    }
}

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static double urand(void) { return (rnd() >> 11) * (1.0 / 9007199254740992.0); }
static double nrand(void) { double u = urand() + 1e-300, v = urand(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

int main(void) {
    const int sizes[] = {16, 64, 192, 256, 1024, 2752, 4096, 1028};
    long bad = 0, total = 0, rounds = 0;
    for (int rep = 0; rep < 40000; ++rep) {
        const int n = sizes[rep % 8];
        float* x = malloc(sizeof(float) * n);
        const int kind = (rep / 8) % 14;
        const double mag = exp2((double)((int)(rnd() % 40) - 20));
        for (int i = 0; i < n; ++i) {
            switch (kind) {
            case 0: x[i] = (float)(nrand() * mag); break;
            case 1: x[i] = (float)(nrand() * exp2((double)((int)(rnd() % 80) - 40))); break;
            case 2: x[i] = (float)((i + 1) * 0.37); break;
            case 3: x[i] = (float)(1000.0 / (i + 1)); break;
            case 4: x[i] = (float)exp2((double)((int)(rnd() % 17) - 14)); break;
            case 5: x[i] = 1.0f; break;
            case 6: x[i] = (float)((int)(rnd() % 11) - 5); break;
            case 7: x[i] = urand() < 0.7 ? 0.f : (float)nrand(); break;
            case 8: x[i] = 0.f; break;
            case 9: x[i] = (float)(nrand() * 1e-22); break;
            case 10: x[i] = (float)(nrand() * 1e18); break;
            case 11: x[i] = i == n / 2 ? 3e19f : (float)nrand(); break;
            case 12: x[i] = i == 5 ? 4096.f : (float)(nrand() * 1e-3); break;
            default: x[i] = i == 0 ? 1.f : 0x1p-12f; break;
            }
        }
        if (kind == 0 && (rep & 64)) for (int i = 0; i < n; ++i) if (urand() < 0.3) x[i] = (float)(exp2((double)((int)(rnd() % 17) - 13)) * (1.0 + 0.5 * (rnd() % 3)));
        for (int head = 0; head <= 4; head += 2) {
            g_rounds = 0;
            const float a = chain_seq(x, n), b = chain_spec(x, n, head);
            ++total; rounds += g_rounds;
            if (f2u(a) != f2u(b) && !(a != a && b != b)) { if (bad < 20) printf("MISMATCH kind %d n %d head %d: seq %a spec %a\n", kind, n, head, a, b); ++bad; }
        }
        free(x);
    }
    printf("%ld cases, %ld mismatches, %.2f rounds per case, %d sequential fall-backs\n", total, bad, (double)rounds / total, g_fallback);
    // rounds on the friendly case that matters: n = 1024 normal data
    for (int head = 0; head <= 8; head += 2) {
        long r = 0; for (int rep = 0; rep < 200; ++rep) { float x[1024]; for (int i = 0; i < 1024; ++i) x[i] = (float)nrand(); g_rounds = 0; chain_spec(x, 1024, head); r += g_rounds; }
        printf("n=1024 normal data, head %d lanes: %.2f rounds\n", head, r / 200.0);
    }
    return bad != 0;
}
