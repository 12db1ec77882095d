// Symmetric eigensolver for the CMA-ES model update, hand-written for gfx950: parallel two-sided block
// Jacobi.  The matrix never leaves the GPU and nothing here calls a vendor solver.
//
// Reference code replaced (paths relative to the reference checkout):
//   stochopy/optimize/cmaes/_cmaes.py:303-305   C = triu(C) + triu(C,1).T;  D, B = np.linalg.eigh(C);
//                                               idx = argsort(D); D = D[idx]; B = B[:, idx]
// (numpy.linalg.eigh is LAPACK dsyevd, a third-party call of the reference: SURVEY.md section 8c.)
//
// Method.  M = V^T C V is kept explicitly (V starts as I).  The index range is cut into nb blocks of 16; a
// sweep is the nb-1 rounds of a round-robin tournament, each round pairing every block with one partner, and
// a round is ONE kernel (eigh_round_kernel) whose workgroups all depend on the previous launch only:
//   * pair workgroups (one per pair (I,J) of the round): the 32x32 pivot matrix [M_II M_IJ; M_JI M_JJ] of the
//     CURRENT matrix is formed from the previous round's tiles and rotations (three small MFMA products), then
//     one cyclic Jacobi sweep in LDS (31 rounds of 16 disjoint plane rotations; 256 threads = one 2x2 block of
//     the two-sided update each) yields the orthogonal 32x32 U of the pair.
//   * tile workgroups apply the PREVIOUS round's rotations to everything: tile (P,Q) of M becomes
//     U_P^T (X U_Q) and tile (R,Q) of V becomes X U_Q, as fp64 MFMA (v_mfma_f64_16x16x4_f64) products, read
//     from one buffer pair and written to the other.
//   So the similarity updates (the flops) of round r-1 run beside the pivot sweeps (the latency) of round r,
//   and a round costs one kernel boundary.
// A sweep is the last one when the off-diagonal mass it leaves behind is below tol * ||C||_F.  That mass is MEASURED:
// the tile workgroups of the launch that applies a sweep's last rotations add up the off-diagonal squares of what
// they write, and the next launch (round 1 of the following sweep) ends the run if the sum is below the threshold --
// one extra round instead of a verifying sweep (round 3; before, a sweep was only accepted from the mass met DURING
// it, eigh_last_sweep(), which is kept as a second rule: 7 -> 6 sweeps per decomposition inside a CMA-ES run at
// n=512).  The decision is taken on the device and later launches of the run are no-ops, so the host never waits.
// Finalisation: column norms of V (removes the drift of |v_j| over hundreds of rounds), eigenvalues
// M_jj / |v_j|^2, ascending order (ties: lower position first), and the CANONICAL SIGN: the component of
// largest magnitude of every eigenvector (lowest index on ties) is positive -- the rule
// oracle/engine.py::eigh_canonical applies to LAPACK's vectors, so both sides of a parity test see the same basis.
//
// MFMA operand layout (cdna_hip_programming.md section 3): A (16x4): lane l holds A[l & 15][l >> 4];
// B (4x16): lane l holds B[l >> 4][l & 15]; C/D: 4 doubles per lane, col = l & 15, row = (l >> 4) + 4 * reg.
#include "sx_device.hpp"
#include "sx_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

using namespace sx;

#ifdef SX_EIGH_TRACE
// debug build only (tools/trace_eigh.py): shader-clock stamps of the pair workgroups of the LAST launch that ran them
__device__ unsigned long long sx_eigh_trace_buf[64 * 16];
#define SX_ETP(k)                                                                             \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 64) sx_eigh_trace_buf[blockIdx.x * 16 + (k)] = clock64(); \
    } while (0)
extern "C" int sx_eigh_trace_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sx_eigh_trace_buf), sizeof(unsigned long long) * 64 * 16);
}
#define SX_ETQ(cond, k)                                                                       \
    do {                                                                                      \
        if ((cond) && blockIdx.x < 64) sx_eigh_trace_buf[blockIdx.x * 16 + (k)] = clock64(); \
    } while (0)
// the resident kernel (tools/trace_eigh_flow.py): wall-clock stamps (100 MHz, the same clock on every CU) of workgroups
// 0..63 in launch numbers [kFtRound0, kFtRound0 + 8): [workgroup][round][slot]
__device__ unsigned long long sx_eigh_ftrace_buf[64 * 8 * 16];
__device__ int sx_eigh_ftrace_round0 = 40;
__shared__ int g_ft_round;
#define SX_FT_BEGIN(k)                                                                        \
    do {                                                                                      \
        if (threadIdx.x == 0) g_ft_round = (k) - sx_eigh_ftrace_round0;                       \
    } while (0)
#define SX_FTP(slot)                                                                          \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 64 && g_ft_round >= 0 && g_ft_round < 8)         \
            sx_eigh_ftrace_buf[(blockIdx.x * 8 + g_ft_round) * 16 + (slot)] = wall_clock64(); \
    } while (0)
extern "C" int sx_eigh_ftrace_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sx_eigh_ftrace_buf), sizeof(unsigned long long) * 64 * 8 * 16);
}
extern "C" int sx_eigh_ftrace_set_round(int k) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sx_eigh_ftrace_round0), &k, sizeof(int));
}
#else
#define SX_ETP(k) do {} while (0)
#define SX_ETQ(cond, k) do {} while (0)
#define SX_FT_BEGIN(k) do {} while (0)
#define SX_FTP(slot) do {} while (0)
#endif

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kEighMaxSweeps = 60;
constexpr int kBS = 16;        // block size of the outer method
constexpr int kM2 = 2 * kBS;   // pivot matrices are 32x32
constexpr int kUU = kM2 * kM2; // doubles per pair rotation

struct EighInfo {  // first bytes of the workspace
    int32_t done_seq;   // launch number at which the run ended (0: running); launches with a larger number do nothing
    int32_t sweeps;     // sweeps carried out
    int32_t parity;     // which (M, V) buffer pair holds the result
    int32_t converged;  // 1: the stopping rule was met within max_sweeps
    double norm2;       // ||C||_F^2 (upper triangle mirrored)
    double thr2;        // tol^2 * norm2
    double acc[kEighMaxSweeps];   // squared off-diagonal mass met during each sweep
    double offm[kEighMaxSweeps];  // squared off-diagonal mass of M AFTER each sweep, measured exactly by the tile
                                  // workgroups that apply the sweep's last rotations
    int32_t refine;     // 1: the run ended with the first-order refinement step still to be applied (see below)
    int32_t fault;      // resident kernel only: a bounded wait ran out (kFlowWait*): the run record is not to be trusted
    unsigned long long kmax2[kEighMaxSweeps];  // bits of max (M_ij / (M_jj - M_ii))^2 over the significant elements
                                               // of M after each sweep (same launch as offm)
    double deep_off[4];                        // the same two measurements of M after each deep refinement step (below)
    unsigned long long deep_k[4];
    int32_t deep_steps;                        // deep steps carried out (diagnosis)
    int32_t pad2_;
};

// The last sweep of a run meets a matrix whose off-diagonal part E is tiny against every gap of the diagonal D: its
// rotations are the first-order solution of (D + E) -> diagonal, K_ij = E_ij / (d_j - d_i), which needs no sequential
// sweep at all.  With `refine` on (the CMA-ES loops), a run whose matrix after sweep s has
//     off(M) <= kRefineOff |C|_F,   max |K_ij| <= kRefineCap   and   max |K_ij| * off(M) / |C|_F <= kRefineProd
// ends there, and V <- V (I + K + K^2 / 2) (orthogonal to O(K^3); two n^3 products on the matrix cores, ~35 us at
// n = 512), d_i <- d_i - sum_j K_ij E_ij (the second-order term of the eigenvalues) stand in for sweep s + 1
// (~320 us).  What the step leaves behind is ~0.5 max|K| off(M) (measured: tools/eigh_refine_check.py,
// profiles/r3_eigh_refine.txt) -- hence the product rule, which keeps the residual at the 1e-12 |C|_F level and the
// eigenvectors within ~1e-9 of LAPACK's, where two solvers differ anyway inside a cluster (eps / gap).
// Elements below tol |C|_F / npad are left alone (together they stay below the tolerance) -- a multiple eigenvalue,
// where d_j - d_i is rounding noise, thus never blocks the step, while a significant element over a tiny gap
// (max |K| large) does, and the run goes on sweeping as before.
constexpr double kRefineOff = 1.0e-7, kRefineCap = 1.0e-3, kRefineProd = 1.0e-12;
// Round 6: the deep refinement step -- the SECOND-to-last sweep's replacement.  After sweep s of a C4-like run off(M) is ~5e-7 |C|_F
// and max |K| ~1e-2 ... 4e-2: too much for the first-order step above (what it leaves is ~K off), but one EXACT similarity with the
// first-order rotation squares it away:  T = exp(K) (as the square of a Newton-Schulz-corrected second-order exp(K / 2), see
// eigh_deep_gemm_kernel),  M' = T^T M T,  V' = V T  -- seven n^3 products on the matrix cores (~90 us at n = 512) where a sweep is 31 launches
// (~365 us).  M' is then measured like a sweep's result: the first-order step finishes (typically off ~3e-9, K ~1e-4), or a second deep
// step runs first; if neither gets there the run is reported as not converged (it does not happen on the test matrices: the entry
// rule keeps K off ~1e-7).  Entry: off(M) <= kDeepOff |C|_F and max |K| <= kDeepCap, measured by the tile workgroups as before;
// only with the refinement allowed (bit 1 of the `refine` argument) and npad >= 256 (the no-op launches of an unused deep step
// cost ~20 us, a sweep of a smaller matrix less than ten times that).  info->refine: 1 = the first-order step is due, 2 / 3 = the
// first / second deep step is due.
#ifndef SX_DEEP_OFF
#define SX_DEEP_OFF 2.0e-6
#endif
#ifndef SX_DEEP_CAP
#define SX_DEEP_CAP 5.0e-2
#endif
#ifndef SX_DEEP_STEPS
#define SX_DEEP_STEPS 2
#endif
#ifndef SX_DEEP_ATAN
#define SX_DEEP_ATAN 0
#endif
constexpr double kDeepOff = SX_DEEP_OFF, kDeepCap = SX_DEEP_CAP;
constexpr int kDeepSteps = SX_DEEP_STEPS;   // deep steps a run may take (<= 4: EighInfo::deep_off)
constexpr double kDeepNs2 = 5.0e-2;         // max |K| above which exp(K / 2) gets a second Newton-Schulz correction
constexpr int kEighFailsOffset = 2040;  // int32 inside the 2048-byte info block, behind EighInfo: runs that fell short

// Stopping rule, evaluated from the off-diagonal mass a_s = sqrt(acc[s] / |C|_F^2) met DURING the sweeps so far:
// the sweep s that just ended is the last one when a_s <= tol (nothing left), or when the iteration is in its
// asymptotic regime (a_s <= 1e-10) and the mass it leaves behind, extrapolated as a_s * (a_s / a_{s-1}) -- exact
// for linear convergence (multiple eigenvalues), pessimistic for quadratic -- is below tol.
__device__ __forceinline__ bool eigh_last_sweep(const double *acc, int s, double norm2, double tol) {
    const double a2 = acc[s], t2 = tol * tol * norm2;
    if (a2 <= t2) return true;
    if (a2 > 1.0e-20 * norm2) return false;
    return s == 0 || a2 * a2 <= t2 * acc[s - 1];
}

// round-robin tournament of m (even) players: pair k of round r (0 <= r < m-1, 0 <= k < m/2)
__host__ __device__ __forceinline__ void rr_pair(int k, int r, int m, int &a, int &b) {
    const int q = m - 1;
    if (k == 0) {
        a = q;
        b = r;
        return;
    }
    int x = r + k;
    if (x >= q) x -= q;
    int y = r - k;
    if (y < 0) y += q;
    a = x;
    b = y;
}
// inverse: the pair of player x in round r and whether x is its first (0) or second (1) member
__device__ __forceinline__ void rr_find(int x, int r, int m, int &k, int &pos) {
    const int q = m - 1;
    if (x == q) {
        k = 0, pos = 0;
        return;
    }
    if (x == r) {
        k = 0, pos = 1;
        return;
    }
    int k1 = x - r;
    if (k1 < 0) k1 += q;
    if (k1 < m / 2) {
        k = k1, pos = 0;
    } else {
        k = q - k1, pos = 1;
    }
}

// Jacobi rotation J = [[c, s], [-s, c]] on the (p,q) plane that (nearly) annihilates a_pq, |angle| <= pi/4.
// With d = a_qq - a_pp, h = 2 a_pq, r = hypot(d, h):  cos 2phi = |d| / r,  sin 2phi = sign(d) h / r,
// c = sqrt((1 + cos 2phi) / 2),  s = sin 2phi / (2 c).  The ANGLE is worked out in single precision (two
// v_rsq_f32, no division, no fp64 square root: this sits on the critical path of every inner round), then
// (c, s) is renormalised in fp64 so that c^2 + s^2 = 1 to rounding: the transform is orthogonal to fp64
// accuracy, and an angle that is off by 1e-7 relative only leaves 1e-7 of a_pq behind (every update below
// applies the rotation that was actually chosen, nothing assumes an exact zero), which the quadratic
// convergence of the sweeps absorbs.
__device__ __forceinline__ void rotation(double app, double aqq, double apq, double &c, double &s) {
    double d = aqq - app, h = 2.0 * apq;
    const double mx = fmax(fabs(d), fabs(h));
    if (h == 0.0 || !(mx < __builtin_inf())) {  // nothing to annihilate (or non-finite input: leave it alone)
        c = 1.0, s = 0.0;
        return;
    }
    int ex;
    (void)frexp(mx, &ex);
    const float df = (float)ldexp(d, -ex), hf = (float)ldexp(h, -ex);  // max(|d|, |h|) in [0.5, 1)
    const float ir = __builtin_amdgcn_rsqf(fmaf(df, df, hf * hf));
    const float x2 = fmaf(0.5f, fabsf(df) * ir, 0.5f);  // c^2 in [0.5, 1]
    const float ic = __builtin_amdgcn_rsqf(x2);
    const double cd = (double)(x2 * ic);
    const double sd = (double)((0.5f * ((df < 0.0f ? -hf : hf) * ir)) * ic);
    const double e = fma(-cd, cd, fma(-sd, sd, 1.0));  // 1 - (c^2 + s^2) ~ 1e-7
    const double k = fma(e, fma(0.375, e, 0.5), 1.0);  // (1 - e)^(-1/2) to e^3
    c = cd * k, s = sd * k;
}

// the LDS arrays one Jacobi sweep works on (the callers own the storage; everything is double-buffered)
struct JacobiView {
    double *S0, *S1;  // [M2][M2 + 1] each: the matrix
    double *W0, *W1;  // [M2][M2 + 1] each: accumulated rotations (rows fixed, columns move with the matrix)
    double *c, *s;    // [2][M2 / 2] rotation of every pair of the current / next inner round
};

// The sweep is systolic: position pair k is ALWAYS (2k, 2k+1), and after every round rows and columns move by
// a fixed permutation, so every thread reads and writes at addresses that never change (no index arithmetic
// inside the loop) and after the last round everything is back in place.
//   MODE 0 (all pairs; M2-1 rounds): the round-robin tournament -- position 0 stays, the other M2-1 rotate by one
//          place: 1 -> 2 -> 4 -> ... -> M2-2 -> M2-1 -> M2-3 -> ... -> 3 -> 1.
//   MODE 1 (only pairs (even, odd) position = (block I, block J) of the outer method; M2/2 rounds): even
//          positions stay, odd positions move on by one pair.
template <int MODE>
__host__ __device__ constexpr int sys_perm(int x, int m) {
    if (MODE == 1) return (x & 1) ? (x + 2 >= m ? 1 : x + 2) : x;
    return x == 0 ? 0 : (x == 1 ? 2 : ((x & 1) ? x - 2 : (x == m - 2 ? m - 1 : x + 2)));
}
template <int MODE>
__host__ __device__ constexpr int sys_perm_inv(int y, int m) {
    if (MODE == 1) return (y & 1) ? (y == 1 ? m - 1 : y - 2) : y;
    return y == 0 ? 0 : (y == 2 ? 1 : ((y & 1) ? (y == m - 1 ? m - 2 : y + 2) : y - 2));
}
template <int M2>
constexpr int jacobi_threads() {  // NP*NP updating threads, plus one wave that only prepares the next rotations
    return (M2 / 2) * (M2 / 2) + 64 <= 1024 ? (M2 / 2) * (M2 / 2) + 64 : (M2 / 2) * (M2 / 2);
}

// one cyclic sweep on the M2 x M2 matrix in S0 (cur = 0) / S1; W <- W J for every rotation.
// One barrier per inner round: while the NP*NP updating threads apply the rotations of round r to their 2x2
// blocks (writing to the permuted places of the other buffer), NP rotation lanes -- a wave of their own when
// the workgroup has room for one -- work out the pivot of their pair of round r+1 from the OLD matrix and the
// rotations of round r, and from it the next rotation.
template <int M2, int MODE>
__device__ void jacobi_sweep(const JacobiView &L, int &cur, const int tid) {
    constexpr int NP = M2 / 2, LD = M2 + 1, NREG = NP * NP, ROUNDS = MODE == 1 ? NP : M2 - 1;
    constexpr bool OWN_WAVE = jacobi_threads<M2>() > NREG;
    const bool reg = tid < NREG;
    const int kp = reg ? tid / NP : 0, kq = reg ? tid % NP : 0;
    // updating threads: source block (2kp, 2kp+1) x (2kq, 2kq+1), destination at the permuted places
    const int o00 = (2 * kp) * LD + 2 * kq, o10 = o00 + LD;
    const int dr0 = sys_perm<MODE>(2 * kp, M2) * LD, dr1 = sys_perm<MODE>(2 * kp + 1, M2) * LD;
    const int dc0 = sys_perm<MODE>(2 * kq, M2), dc1 = sys_perm<MODE>(2 * kq + 1, M2);
    // Rotation lanes.  With a wave of their own (OWN_WAVE) FOUR lanes serve pair k of the next round -- its old
    // positions (i, j) = perm^-1(2k, 2k+1) -- and work out one pivot element each (lane 0: (i,i), 1: (j,j), 2 and 3:
    // (i,j)) with the very operations the updating threads apply, so the predicted pivot IS the next matrix's; a DPP
    // quad broadcast collects the three values and every lane forms the rotation (lane 0 of the quad stores it).
    // Without room for an extra wave (M2 = 64) lane k of wave 0 does the three elements one after the other.
    const int dl = OWN_WAVE ? tid - NREG : tid;
    const bool duty = dl >= 0 && dl < (OWN_WAVE ? 4 * NP : NP);
    const int k2 = duty ? (OWN_WAVE ? dl >> 2 : dl) : 0, part = OWN_WAVE ? (dl & 3) : 0;
    const int i = sys_perm_inv<MODE>(2 * k2, M2), j = sys_perm_inv<MODE>(2 * k2 + 1, M2);
    // element (ea, eb) this lane predicts
    const int ea = part == 1 ? j : i, eb = part == 0 ? i : j;
    const int ka = ea >> 1, kb = eb >> 1;
    const bool pa = ea & 1, pb = eb & 1;
    const int qab = (2 * ka) * LD + 2 * kb;
    auto predicted = [&](const double *S, const double *rc, const double *rs, int ka_, bool pa_, int kb_, bool pb_, int q_) {
        const double ca = rc[ka_], sa = rs[ka_], cb = rc[kb_], sb = rs[kb_];
        const double b00 = S[q_], b01 = S[q_ + 1], b10 = S[q_ + LD], b11 = S[q_ + LD + 1];
        const double r0 = pa_ ? fma(sa, b00, ca * b10) : fma(ca, b00, -(sa * b10));
        const double r1 = pa_ ? fma(sa, b01, ca * b11) : fma(ca, b01, -(sa * b11));
        return pb_ ? fma(sb, r0, cb * r1) : fma(cb, r0, -(sb * r1));
    };
    if (duty) {
        const double *S = cur ? L.S1 : L.S0;
        const int o = (2 * k2) * LD + 2 * k2;
        double c, s;
        rotation(S[o], S[o + LD + 1], S[o + 1], c, s);
        if (part == 0) L.c[k2] = c, L.s[k2] = s;
    }
    __syncthreads();
    for (int r = 0; r < ROUNDS; ++r) {
        const double *S = cur ? L.S1 : L.S0;
        double *Sn = cur ? L.S0 : L.S1;
        const double *W = cur ? L.W1 : L.W0;
        double *Wn = cur ? L.W0 : L.W1;
        const double *rc = L.c + (r & 1) * NP, *rs = L.s + (r & 1) * NP;
        SX_ETQ(r == 5 && tid == 0, 6);
        SX_ETQ(r == 5 && duty && dl == 0, 10);
        if (duty && r + 1 < ROUNDS) {
            double nii, njj, nij;
            if (OWN_WAVE) {
                const double v = predicted(S, rc, rs, ka, pa, kb, pb, qab);
                nii = dpp_f64<0x00>(v);  // quad_perm:[0,0,0,0]
                njj = dpp_f64<0x55>(v);  // quad_perm:[1,1,1,1]
                nij = dpp_f64<0xAA>(v);  // quad_perm:[2,2,2,2]
            } else {
                const int ki = i >> 1, kj = j >> 1;
                const bool pi = i & 1, pj = j & 1;
                nii = predicted(S, rc, rs, ki, pi, ki, pi, (2 * ki) * LD + 2 * ki);
                njj = predicted(S, rc, rs, kj, pj, kj, pj, (2 * kj) * LD + 2 * kj);
                nij = predicted(S, rc, rs, ki, pi, kj, pj, (2 * ki) * LD + 2 * kj);
            }
            SX_ETQ(r == 5 && dl == 0 && nij != 12345.0, 11);
            double c, s;
            rotation(nii, njj, nij, c, s);
            SX_ETQ(r == 5 && dl == 0 && c != 12345.0, 12);
            if (part == 0) L.c[((r + 1) & 1) * NP + k2] = c, L.s[((r + 1) & 1) * NP + k2] = s;
        }
        if (reg) {
            const double c1 = rc[kp], s1 = rs[kp], c2 = rc[kq], s2 = rs[kq];
            const double b00 = S[o00], b01 = S[o00 + 1], b10 = S[o10], b11 = S[o10 + 1];
            const double w00 = W[o00], w01 = W[o00 + 1], w10 = W[o10], w11 = W[o10 + 1];
            // rows: J^T B;  columns: (J^T B) J
            const double r00 = fma(c1, b00, -(s1 * b10)), r01 = fma(c1, b01, -(s1 * b11));
            const double r10 = fma(s1, b00, c1 * b10), r11 = fma(s1, b01, c1 * b11);
            Sn[dr0 + dc0] = fma(c2, r00, -(s2 * r01));
            Sn[dr0 + dc1] = fma(s2, r00, c2 * r01);
            Sn[dr1 + dc0] = fma(c2, r10, -(s2 * r11));
            Sn[dr1 + dc1] = fma(s2, r10, c2 * r11);
            // W <- W J: rows 2kp, 2kp+1 (rows of W stay), columns (2kq, 2kq+1) move like the matrix's
            Wn[(2 * kp) * LD + dc0] = fma(c2, w00, -(s2 * w01));
            Wn[(2 * kp) * LD + dc1] = fma(s2, w00, c2 * w01);
            Wn[(2 * kp + 1) * LD + dc0] = fma(c2, w10, -(s2 * w11));
            Wn[(2 * kp + 1) * LD + dc1] = fma(s2, w10, c2 * w11);
        }
        SX_ETQ(r == 5 && tid == 0, 7);
        SX_ETQ(r == 5 && duty && dl == 0, 13);
        __syncthreads();
        SX_ETQ(r == 5 && tid == 0, 8);
        SX_ETQ(r == 5 && duty && dl == 0, 14);
        cur ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------
// The pivot sweep of the round kernel (32 x 32), round 3.  Same systolic schedule and the same arithmetic as
// jacobi_sweep<32, MODE> above, with the work cut differently (tools/trace_eigh.py, round 2: an inner round took
// 950 cycles, 850 of them the LDS traffic of the 256 updating threads -- matrix AND accumulated rotations, both
// read and written through LDS every round):
//   * waves 0-3 (256 threads) update the MATRIX only, one 2x2 block each, through LDS as before;
//   * wave 4 works out the next round's rotations from the old matrix (four lanes per pair), as before, with a
//     shorter chain: the pivot arrives scaled by a power of two to entries below 1 (rotations do not depend on the
//     scale), so the single-precision angle needs no frexp / ldexp;
//   * wave 5 keeps the accumulated rotations W = J_1 J_2 ... in REGISTERS: a column rotation never mixes rows, so
//     two adjacent lanes own one row of W (16 columns each); the pair of position k is always columns (2k, 2k+1) and
//     after every round the columns move by the systolic permutation -- register moves with compile-time indices
//     and ONE column swapped between the two lanes (DPP).  (c, s) of the round come as 8 reads; W reaches LDS once,
//     after the last round.
// (c, s) pairs are interleaved in LDS, one 16-byte read each.  One barrier per inner round.
// ---------------------------------------------------------------------------------------------------
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// rotation for a pivot whose entries are below 2 in magnitude (see above); entries whose squares vanish in single
// precision (< 3e-19 of the scale, i.e. of |C|_F) are left alone: far below any tolerance.  Branch-free: the identity
// is selected at the end (an rsq of 0 / inf only produces values that are thrown away).
__device__ __forceinline__ void rotation_scaled(double app, double aqq, double apq, double &c, double &s) {
    const float df = (float)(aqq - app), hf = 2.0f * (float)apq;
    const float r2 = fmaf(df, df, hf * hf);
    const bool live = r2 >= 1.0e-37f && r2 < 1.0e30f;  // else: nothing to annihilate, or non-finite input
    const float ir = __builtin_amdgcn_rsqf(r2);
    const float x2 = fmaf(0.5f, fabsf(df) * ir, 0.5f);  // c^2 in [0.5, 1]
    const float ic = __builtin_amdgcn_rsqf(x2);
    const double cd = (double)(x2 * ic);
    const double sd = (double)((0.5f * ((df < 0.0f ? -hf : hf) * ir)) * ic);
    const double e = fma(-cd, cd, fma(-sd, sd, 1.0));  // 1 - (c^2 + s^2) ~ 1e-7
    const double k = fma(e, fma(0.375, e, 0.5), 1.0);  // (1 - e)^(-1/2) to e^3
    c = live ? cd * k : 1.0, s = live ? sd * k : 0.0;
}

struct SweepView {
    double *S0, *S1;  // [kM2][kM2 + 1] each: the matrix, double-buffered (the sweep starts in S0)
    double *cs;       // [2][kBS][2], 16-byte aligned: (c, s) of every pair of the current / next inner round
    double *Wout;     // [kM2][kM2 + 1]: the accumulated rotations, written once at the end
};

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double flip_sign(double v, unsigned mask_hi) {
    return __hiloint2double(__double2hiint(v) ^ (int)mask_hi, __double2loint(v));
}

// 384 threads: tid < 256 matrix, 256..319 rotations, 320..383 accumulated rotations.  Returns 0 / 1: the final matrix is
// in S0 / S1; W is in Wout (visible after the caller's next barrier).
// Issue priorities: the rotation wave's chain (LDS read -> pivot -> angle -> LDS write) IS the inner round, so it goes
// first; the wave of the accumulated rotations works one round BEHIND (it applies round r-1 while round r is worked
// out, and fetches round r's (c, s) at the END of its turn, when the LDS pipe is idle), so nothing ever waits for it.
template <int MODE>
__device__ int pivot_sweep(const SweepView &L, const int tid) {
    constexpr int M2 = kM2, NP = kBS, LD = M2 + 1, NREG = NP * NP, ROUNDS = MODE == 1 ? NP : M2 - 1;
    const bool reg = tid < NREG;
    const int kp = reg ? tid / NP : 0, kq = reg ? tid % NP : 0;
    const int o00 = (2 * kp) * LD + 2 * kq, o10 = o00 + LD;
    const int dr0 = sys_perm<MODE>(2 * kp, M2) * LD, dr1 = sys_perm<MODE>(2 * kp + 1, M2) * LD;
    const int dc0 = sys_perm<MODE>(2 * kq, M2), dc1 = sys_perm<MODE>(2 * kq + 1, M2);
    // rotation lanes: four per pair of the NEXT round (old positions (i, j) = perm^-1(2k, 2k+1)); lane 0: (i,i), 1: (j,j),
    // 2 and 3: (i,j).  Element (ea, eb) of the next matrix = (row combination of block row ka) then (column
    // combination of block column kb): x = alpha b0. + beta b1., v = gamma x0 + delta x1 with
    // (alpha, beta) = (c, -s) for the even member of pair ka, (s, c) for the odd one -- picked by ADDRESS and a sign mask.
    const int dl = tid - NREG;
    const bool duty = dl >= 0 && dl < 4 * NP;
    const int k2 = duty ? dl >> 2 : 0, part = dl & 3;
    const int i = sys_perm_inv<MODE>(2 * k2, M2), j = sys_perm_inv<MODE>(2 * k2 + 1, M2);
    const int ea = part == 1 ? j : i, eb = part == 0 ? i : j;
    const int ka = ea >> 1, kb = eb >> 1;
    const int pa = ea & 1, pb = eb & 1;
    const int qab = (2 * ka) * LD + 2 * kb;
    const int ia = 2 * ka + pa, ja = 2 * ka + 1 - pa, ib = 2 * kb + pb, jb = 2 * kb + 1 - pb;
    const unsigned ma = pa ? 0u : 0x80000000u, mb = pb ? 0u : 0x80000000u;
    // accumulated rotations: lanes (2 * row + h) of the last wave own row `row` of W, positions 16h .. 16h+15 (a pair
    // (2k, 2k+1) never straddles the halves; the systolic move passes ONE column per round between the two lanes of a
    // row: a DPP swap)
    const int wl = tid - (NREG + 64);
    const bool wduty = wl >= 0;
    const int wrow = wl >> 1, wh = wl & 1;
    constexpr int HW = M2 / 2, HP = NP / 2;  // columns / pairs per lane
    double w[HW], cc[HP], ss[HP];
#pragma unroll
    for (int col = 0; col < HW; ++col) w[col] = (HW * wh + col) == wrow ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < HP; ++k) cc[k] = 1.0, ss[k] = 0.0;
    auto apply_w = [&]() {  // W <- W J on columns (2k, 2k+1), then the columns move like the matrix's
        double t[HW];
        static_for<HP>([&](auto kc_) {
            constexpr int k = decltype(kc_)::value;
            const double w0 = w[2 * k], w1 = w[2 * k + 1];
            t[2 * k] = fma(cc[k], w0, -(ss[k] * w1));
            t[2 * k + 1] = fma(ss[k], w0, cc[k] * w1);
        });
        if (MODE == 1) {  // even positions stay, odd ones move on by one pair: 15 -> 17, 31 -> 1
            const double recv = dpp_f64<kDppXor1>(t[HW - 1]);
#pragma unroll
            for (int x = 0; x < HW; x += 2) w[x] = t[x];
            w[1] = recv;
#pragma unroll
            for (int x = 3; x < HW; x += 2) w[x] = t[x - 2];
        } else {  // 0 stays; 1 -> 2 -> 4 -> ... -> 30 -> 31 -> 29 -> ... -> 3 -> 1: 14 -> 16 and 17 -> 15 change lanes
            const double recv = dpp_f64<kDppXor1>(wh ? t[1] : t[HW - 2]);
            w[0] = wh ? recv : t[0];
            w[2] = wh ? t[0] : t[1];
#pragma unroll
            for (int x = 4; x < HW; x += 2) w[x] = t[x - 2];
#pragma unroll
            for (int x = 1; x < HW - 1; x += 2) w[x] = t[x + 2];
            w[HW - 1] = wh ? t[HW - 2] : recv;
        }
    };
    if (duty) {
        const int o = (2 * k2) * LD + 2 * k2;
        double c, s;
        rotation_scaled(L.S0[o], L.S0[o + LD + 1], L.S0[o + 1], c, s);
        if (part == 0) *(v2d *)(L.cs + 2 * k2) = (v2d){c, s};
        __builtin_amdgcn_s_setprio(3);
    } else if (reg) {
        __builtin_amdgcn_s_setprio(2);
    }  // (the wave of the accumulated rotations stays at the pair workgroup's base priority 1)
    __syncthreads();
    int cur = 0;
    for (int r = 0; r < ROUNDS; ++r) {
        const double *S = cur ? L.S1 : L.S0;
        double *Sn = cur ? L.S0 : L.S1;
        const double *cs = L.cs + (r & 1) * 2 * NP;
        SX_ETQ(r == 5 && tid == 0, 6);
        SX_ETQ(r == 5 && duty && dl == 0, 10);
        if (duty && r + 1 < ROUNDS) {
            const double al = cs[ia], be = flip_sign(cs[ja], ma), ga = cs[ib], de = flip_sign(cs[jb], mb);
            const double b00 = S[qab], b01 = S[qab + 1], b10 = S[qab + LD], b11 = S[qab + LD + 1];
            const double x0 = fma(al, b00, be * b10), x1 = fma(al, b01, be * b11);
            const double v = fma(ga, x0, de * x1);
            const double nii = dpp_f64<0x00>(v);  // quad_perm:[0,0,0,0]
            const double njj = dpp_f64<0x55>(v);  // quad_perm:[1,1,1,1]
            const double nij = dpp_f64<0xAA>(v);  // quad_perm:[2,2,2,2]
            SX_ETQ(r == 5 && dl == 0 && nij != 12345.0, 11);
            double c, s;
            rotation_scaled(nii, njj, nij, c, s);
            SX_ETQ(r == 5 && dl == 0 && c != 12345.0, 12);
            if (part == 0) *(v2d *)(L.cs + ((r + 1) & 1) * 2 * NP + 2 * k2) = (v2d){c, s};
        }
        if (reg) {
            const v2d r1 = *(const v2d *)(cs + 2 * kp), r2 = *(const v2d *)(cs + 2 * kq);
            const double c1 = r1[0], s1 = r1[1], c2 = r2[0], s2 = r2[1];
            const double b00 = S[o00], b01 = S[o00 + 1], b10 = S[o10], b11 = S[o10 + 1];
            // rows: J^T B;  columns: (J^T B) J
            const double r00 = fma(c1, b00, -(s1 * b10)), r01 = fma(c1, b01, -(s1 * b11));
            const double r10 = fma(s1, b00, c1 * b10), r11 = fma(s1, b01, c1 * b11);
            Sn[dr0 + dc0] = fma(c2, r00, -(s2 * r01));
            Sn[dr0 + dc1] = fma(s2, r00, c2 * r01);
            Sn[dr1 + dc0] = fma(c2, r10, -(s2 * r11));
            Sn[dr1 + dc1] = fma(s2, r10, c2 * r11);
        }
        if (wduty) {
            if (r > 0) apply_w();  // the rotations of round r - 1, fetched at the end of the previous turn
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                const v2d t = *(const v2d *)(cs + 2 * (HP * wh + k));
                cc[k] = t[0], ss[k] = t[1];
            }
        }
        SX_ETQ(r == 5 && tid == 0, 7);
        SX_ETQ(r == 5 && duty && dl == 0, 13);
        __syncthreads();
        SX_ETQ(r == 5 && tid == 0, 8);
        SX_ETQ(r == 5 && duty && dl == 0, 14);
        cur ^= 1;
    }
    __builtin_amdgcn_s_setprio(1);
    if (wduty) {  // the last round's rotations; after them every column is back in its place
        apply_w();
#pragma unroll
        for (int col = 0; col < HW; ++col) L.Wout[wrow * LD + HW * wh + col] = w[col];
    }
    return cur;
}

// sum over the workgroup (all threads get the result); NT threads, fixed order
template <int NT>
__device__ double block_sum(double v, double *red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    constexpr int NW = (NT + 63) / 64;
    static_assert(NW <= 17, "red[] holds 17 partial sums");
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}

// ---------------------------------------------------------------------------------------------------
// n <= 64: the whole decomposition in one workgroup (M2 = 16 / 32 / 64 by size), sweeps until the rule holds
// ---------------------------------------------------------------------------------------------------
template <int M2>
__global__ __launch_bounds__(jacobi_threads<M2>()) void eigh_small_kernel(const double *__restrict__ C, int n,
                                                                         double *__restrict__ Mout,
                                                                         double *__restrict__ Vout, EighInfo *info,
                                                                         int max_sweeps, double tol) {
    constexpr int NP = M2 / 2, NT = jacobi_threads<M2>(), LD = M2 + 1;
    __shared__ double sS[2][M2 * LD], sW[2][M2 * LD], sc[2 * NP], ss[2 * NP], sred[17];
    const JacobiView L{sS[0], sS[1], sW[0], sW[1], sc, ss};
    const int tid = threadIdx.x;
    double n2 = 0.0;
    for (int e = tid; e < M2 * M2; e += NT) {
        const int i = e / M2, j = e % M2;
        double v = 0.0;
        if (i < n && j < n) v = i <= j ? C[(int64_t)i * n + j] : C[(int64_t)j * n + i];
        sS[0][i * LD + j] = v;
        sW[0][i * LD + j] = i == j ? 1.0 : 0.0;
        n2 += v * v;
    }
    n2 = block_sum<NT>(n2, sred, tid);
    const double thr2 = tol * tol * n2;
    int cur = 0, sw = 0, conv = 0;
    for (;;) {  // the off-diagonal mass is measured exactly before every sweep: stop as soon as it is below tol
        double off2 = 0.0;
        for (int e = tid; e < M2 * M2; e += NT) {
            const int i = e / M2, j = e % M2;
            const double v = sS[cur][i * LD + j];
            if (i != j) off2 += v * v;
        }
        off2 = block_sum<NT>(off2, sred, tid);
        if (tid == 0 && sw < kEighMaxSweeps) info->acc[sw] = off2;
        if (tid == 0 && sw > 0) info->offm[sw - 1] = off2;
        if (off2 <= thr2) {
            conv = 1;
            break;
        }
        if (sw >= max_sweeps) break;
        jacobi_sweep<M2, 0>(L, cur, tid);
        ++sw;
    }
    for (int e = tid; e < M2 * M2; e += NT) {
        const int i = e / M2, j = e % M2;
        Mout[e] = sS[cur][i * LD + j];
        Vout[e] = sW[cur][i * LD + j];
    }
    if (tid == 0) {
        info->done_seq = 1, info->sweeps = sw, info->parity = 0, info->converged = conv;
        info->norm2 = n2, info->thr2 = thr2;
    }
}

// ---------------------------------------------------------------------------------------------------
// n > 64
// ---------------------------------------------------------------------------------------------------
// M0 = C (upper triangle mirrored) padded with zeros to npad; V0 = I, or the caller's starting basis padded with
// the identity; both rotation buffers = I; ||C||_F^2 into info->norm2 (info zeroed by the host beforehand)
// A grid-stride loop over a bounded grid: every workgroup ends with ONE atomic on info->norm2, and a thousand of
// them on one address were most of this kernel's 15 us at n = 512 (~12 ns each, serialised).
// skip != NULL and *skip != 0 (the CMA-ES loop's done flag): the run is closed at once and every later launch of it
// is a no-op -- generations enqueued ahead of the host's look at the state cost launches, not decompositions.
constexpr unsigned kPrepareMaxBlocks = 128;
__global__ __launch_bounds__(256) void eigh_prepare_kernel(const double *__restrict__ C, int n, int npad,
                                                           const double *__restrict__ Vstart, double *__restrict__ M0,
                                                           double *__restrict__ V0, double *__restrict__ U,
                                                           int64_t ucount, EighInfo *info, const int *__restrict__ skip) {
    __shared__ double red[4];
    if (skip != nullptr && *skip != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) info->done_seq = 1;  // (sweeps = 0, converged = 0: nothing was decomposed)
        return;
    }
    const unsigned np2 = (unsigned)npad * (unsigned)npad, un = (unsigned)npad;
    const unsigned tot = np2 > (unsigned)ucount ? np2 : (unsigned)ucount;
    double acc = 0.0;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < tot; e += gridDim.x * 256u) {
        if (e < np2) {
            const unsigned i = e / un, j = e - i * un;
            const bool in = i < (unsigned)n && j < (unsigned)n;
            double v = 0.0;
            if (in) v = i <= j ? C[(int64_t)i * n + j] : C[(int64_t)j * n + i];
            M0[e] = v;
            V0[e] = (in && Vstart) ? Vstart[(int64_t)i * n + j] : (i == j ? 1.0 : 0.0);
            acc += v * v;
        }
        if (e < (unsigned)ucount) {
            const unsigned w = e % (unsigned)kUU;
            U[e] = (w / kM2 == w % kM2) ? 1.0 : 0.0;
        }
    }
    const double s = block_sum<256>(acc, red, threadIdx.x);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(&info->norm2, s);
}

// position i (0..31) of the pair (a, b) of 16-blocks -> matrix index
__device__ __forceinline__ int pair_index(int i, int a, int b) { return (i < kBS ? a * kBS + i : b * kBS + i - kBS); }

constexpr int LDX = kM2 + 2;   // X tile rows: the A-operand column-slab reads hit 32 distinct banks
constexpr int LDU = kM2 + 16;  // U read transposed as an A operand: two k rows land 16 banks apart
constexpr int LDY = kBS;       // 32x16 intermediate

// acc (16x16, rows i0.., cols j0..) = A[i0.., 0:32] * B[0:32, j0..];  A row-major stride lda, B row-major stride ldb
__device__ __forceinline__ v4d mma_ab(const double *A, int lda, int i0, const double *B, int ldb, int j0, int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    double av[kM2 / 4], bv[kM2 / 4];  // all sixteen LDS operand reads in flight before the first MFMA
#pragma unroll
    for (int k = 0; k < kM2 / 4; ++k) av[k] = A[(i0 + lr) * lda + 4 * k + lk], bv[k] = B[(4 * k + lk) * ldb + j0 + lr];
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kM2 / 4; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], bv[k], acc, 0, 0, 0);
    return acc;
}
// acc (16x16) = A[0:32, i0..]^T * B[0:32, j0..]
__device__ __forceinline__ v4d mma_atb(const double *A, int lda, int i0, const double *B, int ldb, int j0, int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    double av[kM2 / 4], bv[kM2 / 4];
#pragma unroll
    for (int k = 0; k < kM2 / 4; ++k) av[k] = A[(4 * k + lk) * lda + i0 + lr], bv[k] = B[(4 * k + lk) * ldb + j0 + lr];
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kM2 / 4; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], bv[k], acc, 0, 0, 0);
    return acc;
}

// Out = alpha * op(A) B + diag * I on npad x npad matrices (npad a multiple of 32): the four products of a warm
// start.  One workgroup per 32x32 tile, four waves = its four 16x16 quadrants, K in chunks of 32 through LDS with
// the next chunk's global loads in flight during the MFMAs.
template <bool TA>
__device__ __forceinline__ void eigh_gemm_body(const double *__restrict__ A, const double *__restrict__ B,
                                               double *__restrict__ Out, int npad, double alpha, double diag,
                                               const double *__restrict__ addend, double *As, double *Bs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * kM2, j0 = blockIdx.x * kM2;
    const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
    const int lr = lane & 15, lk = lane >> 4;
    // kAhead chunks of both operands live in registers: one workgroup (four waves) per CU has nothing else to hide the
    // ~1 us of a global load behind (round 3: one chunk ahead = 13.5 us per n = 512 product, of which 3.6 us MFMA)
    constexpr int kAhead = 4;
    double ra[kAhead][4], rb[kAhead][4];
    auto fetch = [&](int k0, double (&a4)[4], double (&b4)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, r = e / kM2, c = e % kM2;
            // As[i][k]: TA reads A[k0 + r][i0 + c] (r = k, c = i: coalesced along i), else A[i0 + r][k0 + c]
            a4[u] = TA ? A[(int64_t)(k0 + r) * npad + i0 + c] : A[(int64_t)(i0 + r) * npad + k0 + c];
            b4[u] = B[(int64_t)(k0 + r) * npad + j0 + c];
        }
    };
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    const int nchunk = npad / kM2;
    // two LDS images: chunk c + 1 is staged while chunk c is multiplied, ONE barrier per chunk (a wave reaches the barrier of
    // chunk c + 1 only with chunk c's fragments in its registers, so image c & 1 is free again when chunk c + 2 is staged)
    int img = 0;
    auto stage_and_multiply = [&](double (&a4)[4], double (&b4)[4], auto &&refill) {
        double *Ai = As + img * (kM2 * LDX), *Bi = Bs + img * (kM2 * LDU);
        img ^= 1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, r = e / kM2, c = e % kM2;
            if (TA)
                Ai[c * LDX + r] = a4[u];
            else
                Ai[r * LDX + c] = a4[u];
            Bi[r * LDU + c] = b4[u];
        }
        __syncthreads();
        refill();
        double av[kM2 / 4], bv[kM2 / 4];  // all sixteen fragment reads in flight before the first MFMA
#pragma unroll
        for (int k = 0; k < kM2 / 4; ++k) av[k] = Ai[(wi + lr) * LDX + 4 * k + lk], bv[k] = Bi[(4 * k + lk) * LDU + wj + lr];
#pragma unroll
        for (int k = 0; k < kM2 / 4; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[k], bv[k], acc, 0, 0, 0);
    };
    if (nchunk % kAhead == 0) {
        // whole groups (npad a multiple of 128: n = 512, 1024 ...): not a single guard between the loads and their use, so
        // the compiler's s_waitcnt vmcnt(N) leaves the three younger chunks in flight (with uniform guards in the way it
        // falls back to vmcnt(0) at the joins and the depth collapses: the lesson of the CMA-ES GEMM kernels)
#pragma unroll
        for (int s = 0; s < kAhead; ++s) fetch(s * kM2, ra[s], rb[s]);
        for (int c0 = 0; c0 < nchunk - kAhead; c0 += kAhead) {
#pragma unroll
            for (int s = 0; s < kAhead; ++s)
                stage_and_multiply(ra[s], rb[s], [&] { fetch((c0 + s + kAhead) * kM2, ra[s], rb[s]); });
        }
#pragma unroll
        for (int s = 0; s < kAhead; ++s) stage_and_multiply(ra[s], rb[s], [] {});
    } else {
#pragma unroll
        for (int s = 0; s < kAhead; ++s)
            if (s < nchunk) fetch(s * kM2, ra[s], rb[s]);
        for (int c0 = 0; c0 < nchunk; c0 += kAhead) {
#pragma unroll
            for (int s = 0; s < kAhead; ++s) {  // (static register indices: slot s holds chunk c0 + s)
                const int ck = c0 + s;
                if (ck < nchunk)
                    stage_and_multiply(ra[s], rb[s], [&] { if (ck + kAhead < nchunk) fetch((ck + kAhead) * kM2, ra[s], rb[s]); });
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + lk + 4 * r, gj = j0 + wj + lr;
        double v = alpha * acc[r] + (gi == gj ? diag : 0.0);
        if (addend != nullptr) v += addend[(int64_t)gi * npad + gj];
        Out[(int64_t)gi * npad + gj] = v;
    }
}

template <bool TA>
__global__ __launch_bounds__(256) void eigh_gemm_kernel(const double *__restrict__ A, const double *__restrict__ B,
                                                        double *__restrict__ Out, int npad, double alpha, double diag) {
    __shared__ double As[2 * kM2 * LDX], Bs[2 * kM2 * LDU];
    eigh_gemm_body<TA>(A, B, Out, npad, alpha, diag, nullptr, As, Bs);
}

// ---- the refinement step (see kRefineOff): three launches that do nothing unless the run ended with info->refine ----
// With p = info->parity:  M[p], V[p] hold the result of the sweeps; M[p ^ 1], V[p ^ 1] are free.
//   1. K  -> M[p ^ 1]:  K_ij = M_ij / (d_j - d_i) for the significant elements (upper triangle, mirrored with the sign)
//   2. T = I + K + K K / 2 -> V[p ^ 1]
//   3. V[p] T -> M[p ^ 1]   (the eigenvectors; eigh_colstats / eigh_write read them from there, the eigenvalues stay d)
__global__ __launch_bounds__(256) void eigh_refine_k_kernel(const double *__restrict__ M0, const double *__restrict__ M1,
                                                            double *K0, double *K1, int npad,
                                                            const EighInfo *info, double tol, int want, double cap, double scale) {
    if (info->refine != want) return;
    const double *M = info->parity ? M1 : M0;
    double *K = info->parity ? K0 : K1;
    const double tolel2 = tol * tol * info->norm2 / ((double)npad * (double)npad);
    const unsigned e = blockIdx.x * 256u + threadIdx.x, un = (unsigned)npad;
    if (e >= un * un) return;
    const unsigned i = e / un, j = e - i * un;
    double k = 0.0;
    if (i != j) {
        const unsigned a = i < j ? i : j, b = i < j ? j : i;  // the upper-triangle element decides for both
        const double m = M[(int64_t)a * npad + b];
        const double g = M[(int64_t)b * npad + b] - M[(int64_t)a * npad + a];
        if (m * m > tolel2) k = m / g;
        if (!(fabs(k) <= 2.0 * cap)) k = 0.0;  // (cannot happen after the rule held; never divide by noise)
        // (deep steps: the 2 x 2 problem's own angle instead of its first-order value -- the same to second order, and what
        //  an isolated pair needs exactly when K is not small)
        if (SX_DEEP_ATAN && want >= 2) k = 0.5 * atan(2.0 * k);
        if (i > j) k = -k;
    }
    K[e] = k * scale;  // (the deep step works with K / 2: its rotation is the square of exp(K / 2))
}
// d_i <- d_i - sum_j K_ij M_ij  (= d_i + sum_j M_ij^2 / (d_i - d_j)), one wavefront per row; runs after step 1
__global__ __launch_bounds__(256) void eigh_refine_diag_kernel(double *M0, double *M1, int npad, const EighInfo *info) {
    if (info->refine != 1) return;
    double *M = info->parity ? M1 : M0;
    const double *K = info->parity ? M0 : M1;
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= npad) return;
    double acc = 0.0;
    for (int j = lane; j < npad; j += 64)
        if (j != i) acc = fma(K[(int64_t)i * npad + j], M[(int64_t)i * npad + j], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    // (the diagonal is only read by other rows' K -- already formed -- and by eigh_colstats afterwards)
    if (lane == 0) M[(int64_t)i * npad + i] -= acc;
}
// step 2 (which = 0) and step 3 (which = 1)
__global__ __launch_bounds__(256) void eigh_refine_gemm_kernel(double *__restrict__ M0, double *__restrict__ M1,
                                                               double *__restrict__ V0, double *__restrict__ V1, int npad,
                                                               const EighInfo *info, int which) {
    __shared__ double As[2 * kM2 * LDX], Bs[2 * kM2 * LDU];
    if (info->refine != 1) return;
    const int p = info->parity;
    double *Kb = p ? M0 : M1, *Tb = p ? V0 : V1;
    const double *Vb = p ? V1 : V0;
    if (which == 0)
        eigh_gemm_body<false>(Kb, Kb, Tb, npad, 0.5, 1.0, Kb, As, Bs);
    else
        eigh_gemm_body<false>(Vb, Tb, Kb, npad, 1.0, 0.0, nullptr, As, Bs);
}

// ---- the deep refinement step (see kDeepOff); every kernel does nothing unless info->refine == want (2: first, 3: second step) ----
// With p = info->parity, q = p ^ 1:  H = K / 2 -> M[q] (eigh_refine_k_kernel, cap kDeepCap), then `which` =
//   0: T2 = I + H + H H / 2 -> V[q]      1: E = I - T2^T T2 -> W1      2: Th = T2 + T2 E / 2 -> W2      3: T = Th Th -> W3
//   4: M[p] T -> W1                      5: T^T (M[p] T) -> M[q]       6: V[p] T -> V[q]
// (the rotation exp(K) as the SQUARE of a Newton-Schulz-corrected exp(K / 2): T2^T T2 = I + H^4 / 4 exactly, one correction step
// leaves ~(3/8) (H^4 / 4)^2 -- 4e-15 at max |K| = 5e-2 where the plain form T2(K) + one step left 1e-11 in V^T V -- and the
// square of an orthogonal matrix is orthogonal), then eigh_deep_measure_kernel on M[q] and eigh_deep_finish_kernel.
__global__ __launch_bounds__(256) void eigh_deep_gemm_kernel(double *__restrict__ M0, double *__restrict__ M1,
                                                             double *__restrict__ V0, double *__restrict__ V1,
                                                             double *__restrict__ W1, double *__restrict__ W2,
                                                             double *__restrict__ W3, int npad, const EighInfo *info, int want,
                                                             int which) {
    __shared__ double As[2 * kM2 * LDX], Bs[2 * kM2 * LDU];
    if (info->refine != want) return;
    const int p = info->parity;
    double *Mp = p ? M1 : M0, *Mq = p ? M0 : M1, *Vp = p ? V1 : V0, *Vq = p ? V0 : V1;
    // a second correction (7: E2 = I - Th^T Th -> W1, 8: Th + Th E2 / 2 -> V[q], between 2 and 3) when max |K| of the matrix
    // this step starts from is above kDeepNs2: T2^T T2 - I = H^4 / 4 is 2.5e-5 at max |K| = 0.2, one step leaves 2e-10, two 1e-20
    const double k2 = __longlong_as_double((long long)(want == 2 ? info->kmax2[info->sweeps - 1] : info->deep_k[want - 3]));
    const bool twice = k2 > kDeepNs2 * kDeepNs2;
    if ((which == 7 || which == 8) && !twice) return;
    switch (which) {
        case 0: eigh_gemm_body<false>(Mq, Mq, Vq, npad, 0.5, 1.0, Mq, As, Bs); break;
        case 1: eigh_gemm_body<true>(Vq, Vq, W1, npad, -1.0, 1.0, nullptr, As, Bs); break;
        case 2: eigh_gemm_body<false>(Vq, W1, W2, npad, 0.5, 0.0, Vq, As, Bs); break;
        case 7: eigh_gemm_body<true>(W2, W2, W1, npad, -1.0, 1.0, nullptr, As, Bs); break;
        case 8: eigh_gemm_body<false>(W2, W1, Vq, npad, 0.5, 0.0, W2, As, Bs); break;
        case 3: eigh_gemm_body<false>(twice ? Vq : W2, twice ? Vq : W2, W3, npad, 1.0, 0.0, nullptr, As, Bs); break;
        case 4: eigh_gemm_body<false>(Mp, W3, W1, npad, 1.0, 0.0, nullptr, As, Bs); break;
        case 5: eigh_gemm_body<true>(W3, W1, Mq, npad, 1.0, 0.0, nullptr, As, Bs); break;
        default: eigh_gemm_body<false>(Vp, W3, Vq, npad, 1.0, 0.0, nullptr, As, Bs); break;
    }
}
// off(M')^2 and max K^2 of M' = M[q] (the measurements the tile workgroups take of a sweep's result), into deep_off / deep_k[slot]
__global__ __launch_bounds__(256) void eigh_deep_measure_kernel(const double *__restrict__ M0, const double *__restrict__ M1, int npad,
                                                                EighInfo *info, double tol, int want, int slot) {
    __shared__ double s_off[4], s_k[4];
    if (info->refine != want) return;
    const double *M = info->parity ? M0 : M1;
    const double tolel2 = tol * tol * info->norm2 / ((double)npad * (double)npad);
    double off = 0.0, km = 0.0;
    const unsigned un = (unsigned)npad, tot = un * un;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < tot; e += gridDim.x * 256u) {
        const unsigned i = e / un, j = e - i * un;
        if (i == j) continue;
        const double m = M[e], a2 = m * m;
        off += a2;
        if (a2 > tolel2) {
            const double g = M[(int64_t)j * npad + j] - M[(int64_t)i * npad + i];
            km = fmax(km, a2 / (g * g));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o, kWave), km = fmax(km, __shfl_xor(km, o, kWave));
    if ((threadIdx.x & 63) == 0) s_off[threadIdx.x >> 6] = off, s_k[threadIdx.x >> 6] = km;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (s_off[0] + s_off[1]) + (s_off[2] + s_off[3]);
        const double k = fmax(fmax(s_k[0], s_k[1]), fmax(s_k[2], s_k[3]));
        if (t != 0.0) atomicAdd(&info->deep_off[slot], t);
        if (k > 0.0) atomicMax(&info->deep_k[slot], (unsigned long long)__double_as_longlong(k));
    }
}
__global__ void eigh_deep_finish_kernel(EighInfo *info, double tol, int want, int slot, int *fails) {
    if (threadIdx.x != 0 || info->refine != want) return;
    info->parity ^= 1;  // M', V' are the matrices from here on
    info->deep_steps = slot + 1;
    const double left = info->deep_off[slot], kmax2 = __longlong_as_double((long long)info->deep_k[slot]), nrm2 = info->norm2;
    if (left <= tol * tol * nrm2) {
        info->refine = 0;  // (nothing left above the tolerance)
    } else if (left <= kRefineOff * kRefineOff * nrm2 && kmax2 <= kRefineCap * kRefineCap &&
               kmax2 * left <= kRefineProd * kRefineProd * nrm2) {
        info->refine = 1;  // the first-order step finishes the run
    } else if (want < 1 + kDeepSteps) {
        info->refine = want + 1;  // once more
    } else {
        info->refine = 0, info->converged = 0;
        atomicAdd(fails, 1);
    }
}

struct RoundLds {
    union {
        struct {  // tile workgroups
            double X[kM2 * LDX];
            double UP[kM2 * LDU];
            double UQ[kM2 * LDU];
            double Y[kM2 * LDU];
        } t;
        struct {  // pair workgroups, stage 1: three source tiles, two rotations, the intermediates
            double X[3][kM2 * LDX];
            double UA[kM2 * LDU];
            double UB[kM2 * LDU];
            double Y[3][kM2 * LDY];
        } p;
        struct {  // pair workgroups, stage 2 (the sweep): second pivot buffer; the accumulated rotations of the sweep
            double S1[kM2 * (kM2 + 1)];
            double W0[kM2 * (kM2 + 1)];  // (written once, from registers, at its end)
        } j;
    };
    double S0[kM2 * (kM2 + 1)];  // the pivot matrix (written in stage 1, so outside the union)
    alignas(16) double cs[4 * kBS];  // (c, s) of every pair of the current / next inner round, interleaved
    double red[17];
    double scale;                // power of two that brings |C|_F (hence every pivot entry) below 1
    double tolel2;               // square of the size below which an off-diagonal element does not count (tol |C|_F / npad)
    int flag;
    int ok;                      // (the resident kernel: a wait came back without its condition)
};

// (namespace scope: the resident kernel's pair round is a function of its own, see flow_pair_round)
__shared__ RoundLds g_round_lds;

constexpr int kRoundThreads = jacobi_threads<kM2>() + 64;  // 256 updating threads + the rotation wave of the pivot
                                                            // sweep + a sixth wave: the pivot products split six ways
constexpr int kTileRegs = (kUU + kRoundThreads - 1) / kRoundThreads;  // elements of a 32x32 tile per thread

// Global accesses of the round's work.  FLOW = false: one launch per round, the kernel boundary makes everything visible:
// plain loads and stores.  FLOW = true: all rounds inside ONE resident launch (eigh_flow_kernel): whatever one workgroup writes
// and another reads in the same launch goes as agent-scope accesses -- write-through stores, loads that bypass this CU's vector
// L1 (cdna_hip_programming.md section 6, Guideline 16, form R1) -- ordered by counters, never by placement.
template <bool FLOW>
__device__ __forceinline__ double gld(const double *p) {
    if constexpr (FLOW)  // (a GLOBAL access: through a generic pointer it would be a flat_load, counted as an LDS access too)
        return __hip_atomic_load((__attribute__((address_space(1))) double *)const_cast<double *>(p), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}
template <bool FLOW>
__device__ __forceinline__ void gst(double *p, double v) {
    if constexpr (FLOW)
        __hip_atomic_store((__attribute__((address_space(1))) double *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// 16-byte agent-scope accesses (buffer instructions with the sc1 bit: what Guideline 16 prescribes for payloads; an 8-byte
// write-through store costs 2.7x a 16-byte one per byte on this fabric).  Offsets are bytes from the buffer's base.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int kAuxSc1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t flow_rsrc(const void *base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes > 0x7fffffffll ? 0x7fffffffll : bytes), 0x00020000);
}
__device__ __forceinline__ f64x2 ld16(__amdgpu_buffer_rsrc_t r, int off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxSc1);
    f64x2 d;
    __builtin_memcpy(&d, &v, 16);
    return d;
}
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, int off, f64x2 d) {
    u32x4 v;
    __builtin_memcpy(&v, &d, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, kAuxSc1);
}

// What launch number k (0-based) of a run means -- the same for the launch-per-round form and the resident one: it reads
// the buffer pair k & 1, applies the rotations of round (k - 1) % rps (identities for k = 0 under any pairing), works out
// those of round k % rps, and its launch number is k + 1.  Rotations live in THREE buffers (k % 3): inside the resident
// kernel a fast pair workgroup may publish round k + 2 while a slow one still reads round k.
struct RoundPar {
    const double *Min, *Vin;
    double *Mout, *Vout;
    const double *Uprev;
    double *Ucur;
    int sweep, rprev, rcur, parity_out, seq;
};

// Thread 0 of every workgroup: what this launch has to do, from what earlier launches left in the run record.  Every
// workgroup derives the same answer (the sums it reads are final).  0: a round as usual; 1: apply the last rotations of
// the sweep that just ended and start no new ones; 2: nothing -- the run has ended.  `writer` (workgroup 0) records it.
template <bool FLOW>
__device__ int round_state(EighInfo *info, int ended, double nrm2, const RoundPar &R, double tol, int flush, int refine,
                           bool writer) {
    if (ended != 0 && ended < R.seq) return 2;  // the run ended in an EARLIER launch: nothing to do
    if (flush || R.sweep == 0 || R.rcur > 1) return 0;
    const int sweep = R.sweep;
    const double thr2 = tol * tol * nrm2;
    if (R.rcur == 1) {
        const double left = gld<FLOW>(&info->offm[sweep - 1]);
        if (left <= thr2) {
            // The previous launch applied the last rotations of sweep `sweep - 1` and measured what that sweep left
            // behind: nothing above the tolerance.  The matrix this launch would read IS the result; the rotations of
            // round 0 that were worked out beside the measurement are dropped.  (Costs one round, not a whole
            // verifying sweep.)
            if (writer) {
                info->sweeps = sweep, info->parity = R.parity_out ^ 1, info->converged = 1;
                info->thr2 = thr2;
                info->done_seq = R.seq;
            }
            return 2;
        }
        if (refine) {
            unsigned long long kb;
            if constexpr (FLOW)
                kb = __hip_atomic_load(&info->kmax2[sweep - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                kb = info->kmax2[sweep - 1];
            const double kmax2 = __longlong_as_double((long long)kb);
            if (left <= kRefineOff * kRefineOff * nrm2 && kmax2 <= kRefineCap * kRefineCap &&
                kmax2 * left <= kRefineProd * kRefineProd * nrm2) {
                // what is left is first order against every gap: the run ends here and the refinement step finishes it
                if (writer) {
                    info->sweeps = sweep, info->parity = R.parity_out ^ 1, info->converged = 1;
                    info->thr2 = thr2;
                    info->refine = 1;
                    info->done_seq = R.seq;
                }
                return 2;
            }
            if ((refine & 2) && left <= kDeepOff * kDeepOff * nrm2 && kmax2 <= kDeepCap * kDeepCap) {
                // one sweep earlier still: the deep refinement step(s), then the first-order one
                if (writer) {
                    info->sweeps = sweep, info->parity = R.parity_out ^ 1, info->converged = 1;
                    info->thr2 = thr2;
                    info->refine = 2;
                    info->done_seq = R.seq;
                }
                return 2;
            }
        }
        return 0;
    }
    // rcur == 0: second rule, from the mass met DURING the last two sweeps (eigh_last_sweep)
    const double met1 = gld<FLOW>(&info->acc[sweep - 1]);
    const double met0 = sweep > 1 ? gld<FLOW>(&info->acc[sweep - 2]) : 0.0;
    const bool last = met1 <= thr2 || (met1 <= 1.0e-20 * nrm2 && (sweep == 1 || met1 * met1 <= thr2 * met0));
    if (last) {  // apply the last rotations of the sweep that just ended, start no new ones
        if (writer) {
            info->sweeps = sweep, info->parity = R.parity_out, info->converged = 1;
            info->thr2 = thr2;
            info->done_seq = R.seq;
        }
        return 1;
    }
    return 0;
}

__device__ __forceinline__ void round_scales(RoundLds &L, double nrm2, double tol, int npad) {
    L.tolel2 = tol * tol * nrm2 / ((double)npad * (double)npad);
    int ex = 0;
    const double nrm = sqrt(nrm2);
    if (nrm > 0.0 && nrm < __builtin_inf()) (void)frexp(nrm, &ex);
    L.scale = ldexp(1.0, -ex);
}

// ---- tile work: tile (P, Q) of M becomes U_P^T (X U_Q), tile (R = P, Q) of V becomes X U_Q, with U the rotations of
// round rprev; three steps so that a caller can keep several tiles' loads in flight ----
struct TileIdx {
    int P, Q, ap, bp, aq, bq;
    bool is_m;
};
__device__ __forceinline__ TileIdx tile_index(bool is_m, int P, int Q, int rprev, int nb) {
    TileIdx t{P, Q, 0, 0, 0, 0, is_m};
    rr_pair(Q, rprev, nb, t.aq, t.bq);
    if (is_m) rr_pair(P, rprev, nb, t.ap, t.bp);
    return t;
}
template <bool FLOW>
__device__ __forceinline__ void tile_fetch(const TileIdx &t, const double *__restrict__ src, const double *__restrict__ Uprev,
                                           int npad, int tid, bool want_uq, double (&x)[kTileRegs], double (&uq)[kTileRegs],
                                           double (&up)[kTileRegs]) {
#pragma unroll
    for (int u = 0; u < kTileRegs; ++u) {
        const int e = tid + kRoundThreads * u;
        if (e < kUU) {
            const int i = e / kM2, j = e % kM2;
            const int gi = t.is_m ? pair_index(i, t.ap, t.bp) : t.P * kM2 + i;
            x[u] = gld<FLOW>(src + (int64_t)gi * npad + pair_index(j, t.aq, t.bq));
            if (want_uq) uq[u] = gld<FLOW>(Uprev + (int64_t)t.Q * kUU + e);
            if (t.is_m) up[u] = gld<FLOW>(Uprev + (int64_t)t.P * kUU + e);
        }
    }
}
// the same with 16-byte loads (the resident kernel's workers): thread t holds elements (i, 2 j2), (i, 2 j2 + 1) for
// e2 = t + kRoundThreads u = 16 i + j2 -- two columns of one 16-block, adjacent in memory
constexpr int kTileRegs2 = (kUU / 2 + kRoundThreads - 1) / kRoundThreads;
__device__ __forceinline__ void tile_fetch16(const TileIdx &t, __amdgpu_buffer_rsrc_t src, __amdgpu_buffer_rsrc_t ures, int npad,
                                             int tid, bool want_uq, f64x2 (&x)[kTileRegs2], f64x2 (&uq)[kTileRegs2],
                                             f64x2 (&up)[kTileRegs2]) {
#pragma unroll
    for (int u = 0; u < kTileRegs2; ++u) {
        const int e2 = tid + kRoundThreads * u;
        if (e2 < kUU / 2) {
            const int i = e2 / (kM2 / 2), j = 2 * (e2 % (kM2 / 2));
            const int gi = t.is_m ? pair_index(i, t.ap, t.bp) : t.P * kM2 + i;
            x[u] = ld16(src, (gi * npad + pair_index(j, t.aq, t.bq)) * 8);
            if (want_uq) uq[u] = ld16(ures, (t.Q * kUU + 2 * e2) * 8);
            if (t.is_m) up[u] = ld16(ures, (t.P * kUU + 2 * e2) * 8);
        }
    }
}
__device__ __forceinline__ void tile_stage16(RoundLds &L, bool is_m, int tid, bool want_uq, const f64x2 (&x)[kTileRegs2],
                                             const f64x2 (&uq)[kTileRegs2], const f64x2 (&up)[kTileRegs2]) {
#pragma unroll
    for (int u = 0; u < kTileRegs2; ++u) {
        const int e2 = tid + kRoundThreads * u;
        if (e2 < kUU / 2) {
            const int i = e2 / (kM2 / 2), j = 2 * (e2 % (kM2 / 2));
            *(f64x2 *)&L.t.X[i * LDX + j] = x[u];  // (LDX, LDU even: 16-byte aligned)
            if (want_uq) *(f64x2 *)&L.t.UQ[i * LDU + j] = uq[u];
            if (is_m) *(f64x2 *)&L.t.UP[i * LDU + j] = up[u];
        }
    }
}
__device__ __forceinline__ void tile_stage(RoundLds &L, bool is_m, int tid, bool want_uq, const double (&x)[kTileRegs],
                                           const double (&uq)[kTileRegs], const double (&up)[kTileRegs]) {
#pragma unroll
    for (int u = 0; u < kTileRegs; ++u) {
        const int e = tid + kRoundThreads * u;
        if (e < kUU) {
            const int i = e / kM2, j = e % kM2;
            L.t.X[i * LDX + j] = x[u];
            if (want_uq) L.t.UQ[i * LDU + j] = uq[u];
            if (is_m) L.t.UP[i * LDU + j] = up[u];
        }
    }
}
// (after the staging and a barrier; ends with every global store and atomic of the tile ISSUED, not necessarily complete)
template <bool FLOW>
__device__ __forceinline__ void tile_compute(RoundLds &L, const TileIdx &t, const double *__restrict__ Min,
                                             double *__restrict__ dst, int npad, EighInfo *info, const RoundPar &R,
                                             int refine, int tid) {
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const bool is_m = t.is_m;
    const int i0 = (wave >> 1) * 16, j0 = (wave & 1) * 16;  // waves 0..3: one 16x16 quadrant each
    // the launch that measures what a sweep left also measures max |M_ij / (d_j - d_i)| (refinement rule); the
    // diagonal is taken from the matrix BEFORE this round's rotations -- in this phase it moves by second-order amounts
    const bool measure_k = refine && is_m && R.rcur == 0 && R.sweep > 0 && wave < 4;
    double dgi[4] = {0.0, 0.0, 0.0, 0.0}, dgj = 0.0;
    if (measure_k) {
        const int gj = pair_index(j0 + lr, t.aq, t.bq);
        dgj = gld<FLOW>(Min + (int64_t)gj * npad + gj);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gi = pair_index(i0 + lk + 4 * r, t.ap, t.bp);
            dgi[r] = gld<FLOW>(Min + (int64_t)gi * npad + gi);
        }
    }
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    if (wave < 4) acc = mma_ab(L.t.X, LDX, i0, L.t.UQ, LDU, j0, lane);
    if (is_m) {
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) L.t.Y[(i0 + lk + 4 * r) * LDU + j0 + lr] = acc[r];
        }
        __syncthreads();
        if (wave < 4) acc = mma_atb(L.t.UP, LDU, i0, L.t.Y, LDU, j0, lane);
    }
    double m2 = 0.0, k2 = 0.0;
    if (wave < 4) {
        const int gj = pair_index(j0 + lr, t.aq, t.bq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + lk + 4 * r;
            const int gi = is_m ? pair_index(i, t.ap, t.bp) : t.P * kM2 + i;
            gst<FLOW>(dst + (int64_t)gi * npad + gj, acc[r]);
            if (gi != gj) {
                m2 = fma(acc[r], acc[r], m2);
                if (measure_k) {
                    const double a2 = acc[r] * acc[r], g = dgj - dgi[r];
                    if (a2 > L.tolel2) k2 = fmax(k2, a2 / (g * g));  // (a zero gap under a significant element: inf)
                }
            }
        }
    }
    // this launch applies the LAST rotations of sweep `sweep - 1`: what it writes is the matrix after that sweep,
    // and its off-diagonal mass decides (in the next launch) whether the run is over
    if (is_m && R.rcur == 0 && R.sweep > 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m2 += __shfl_xor(m2, off, kWave);
        if (refine) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) k2 = fmax(k2, __shfl_xor(k2, off, kWave));
        }
        if (lane == 0 && wave < 4) L.red[wave] = m2, L.red[4 + wave] = k2;
        __syncthreads();
        if (tid == 0) {
            const double tt = (L.red[0] + L.red[1]) + (L.red[2] + L.red[3]);
            if (tt != 0.0) atomicAdd(&info->offm[R.sweep - 1], tt);
            const double km = fmax(fmax(L.red[4], L.red[5]), fmax(L.red[6], L.red[7]));
            // (non-negative doubles order like their bit patterns; NaN -- 0/0 cannot occur, a2 > 0 -- would read as huge)
            if (refine && km > 0.0) atomicMax(&info->kmax2[R.sweep - 1], (unsigned long long)__double_as_longlong(km));
        }
    }
}

// ---- pair work: the rotation U of pair (I, J) of round rcur from the current pivot ----
struct PairIdx {
    int I, J, PI, posI, PJ, posJ, aI, bI, aJ, bJ;
};
__device__ __forceinline__ PairIdx pair_lookup(int k, const RoundPar &R, int nb) {
    PairIdx q;
    rr_pair(k, R.rcur, nb, q.I, q.J);
    rr_find(q.I, R.rprev, nb, q.PI, q.posI);
    rr_find(q.J, R.rprev, nb, q.PJ, q.posJ);
    rr_pair(q.PI, R.rprev, nb, q.aI, q.bI);
    rr_pair(q.PJ, R.rprev, nb, q.aJ, q.bJ);
    return q;
}
// source tiles (pairs of rprev) 0: (PI,PI)  1: (PI,PJ)  2: (PJ,PJ); 256 threads x 4 elements of each
template <bool FLOW>
__device__ __forceinline__ void pair_fetch_tiles(const PairIdx &q, const double *__restrict__ Min, int npad, int tid,
                                                 double (&x0)[4], double (&x1)[4], double (&x2)[4]) {
    if (tid < 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, i = e / kM2, j = e % kM2;
            const int giI = pair_index(i, q.aI, q.bI), giJ = pair_index(i, q.aJ, q.bJ);
            const int gjI = pair_index(j, q.aI, q.bI), gjJ = pair_index(j, q.aJ, q.bJ);
            x0[u] = gld<FLOW>(Min + (int64_t)giI * npad + gjI);
            x1[u] = gld<FLOW>(Min + (int64_t)giI * npad + gjJ);
            x2[u] = gld<FLOW>(Min + (int64_t)giJ * npad + gjJ);
        }
    }
}
template <bool FLOW>
__device__ __forceinline__ void pair_fetch_u(const PairIdx &q, const double *__restrict__ Uprev, int tid, double (&ua)[4],
                                             double (&ub)[4]) {
    if (tid < 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u;
            ua[u] = gld<FLOW>(Uprev + (int64_t)q.PI * kUU + e);
            ub[u] = gld<FLOW>(Uprev + (int64_t)q.PJ * kUU + e);
        }
    }
}
// (ends with the rotation's stores and the sweep's atomic ISSUED)
// FLOW: u_staged = the needed columns of both rotations are already in L.p.UA / L.p.UB (pair_poll_granules); Go = this
// pair's granule slot (the rotation goes there FIRST: tagged halves, see pair_poll_granules); false: a wait had run out.
template <bool FLOW>
__device__ __forceinline__ bool pair_compute(RoundLds &L, const PairIdx &q, const double (&x0)[4], const double (&x1)[4],
                                             const double (&x2)[4], const double (&ua)[4], const double (&ub)[4],
                                             double *__restrict__ Uo, EighInfo *info, const RoundPar &R, int tid,
                                             bool u_staged = false, unsigned long long *Go = nullptr) {
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int posI = q.posI, posJ = q.posJ;
    // (a pair workgroup shares its CU with a tile workgroup of the same launch; the round waits for the pair)
    __builtin_amdgcn_s_setprio(1);
    SX_ETP(0);
    if constexpr (FLOW) SX_FTP(5);
    constexpr int LD = kM2 + 1;
    if (tid < 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + 256 * u, i = e / kM2, j = e % kM2;
            L.p.X[0][i * LDX + j] = x0[u];
            L.p.X[1][i * LDX + j] = x1[u];
            L.p.X[2][i * LDX + j] = x2[u];
            if (!u_staged) {
                L.p.UA[i * LDU + j] = ua[u];
                L.p.UB[i * LDU + j] = ub[u];
            }
        }
    }
    __syncthreads();
    SX_ETP(1);
    if constexpr (FLOW) {
        SX_FTP(6);
        if (L.ok == 0) return false;  // (a wave's wait for the predecessors' rotations had run out)
    }
    // The current pivot: T_II = A_I^T X0 A_I, T_IJ = A_I^T X1 A_J, T_JJ = A_J^T X2 A_J with A_X the 16 columns of the
    // previous rotation that belong to block X.  Wave w < 3 forms sub-block w.  The pivot is stored INTERLEAVED:
    // member i of block I at position 2i, member j of block J at position 2j+1 (what the systolic sweep expects).
    // full != 0 (first round of a sweep): every pair of the 32 indices is rotated; otherwise only the pairs
    // (I-member, J-member) -- over the rounds of a sweep every off-diagonal element is then targeted exactly once,
    // and the mass met (all off-diagonal entries in the full round, the I x J block otherwise) adds up to off(M)^2.
    const bool full = R.rcur == 0;
    {   // first products, six ways: wave w forms half h = w & 1 of Y_b = X_b * A_right, b = w >> 1
        const int b = wave >> 1, h = wave & 1;
        const double *UR = b == 0 ? L.p.UA : L.p.UB;
        const int cR = 16 * (b == 0 ? posI : posJ);
        const v4d a = mma_ab(L.p.X[b], LDX, 16 * h, UR, LDU, cR, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) L.p.Y[b][(16 * h + lk + 4 * r) * LDY + lr] = a[r];
    }
    SX_ETP(2);
    __syncthreads();
    SX_ETP(9);
    double m2 = 0.0;  // off-diagonal mass of this wave's sub-block (added to the sweep's total at the very end)
    if (wave < 3) {  // second products: T_b = A_left^T Y_b (16x16), one sub-block per wave
        const double *UL = wave == 2 ? L.p.UB : L.p.UA;
        const int cL = 16 * (wave == 2 ? posJ : posI);
        const v4d tt = mma_atb(UL, LDU, cL, L.p.Y[wave], LDY, 0, lane);
        const double sc = L.scale;
        const int ro = wave == 2 ? 1 : 0, co = wave == 0 ? 0 : 1;  // odd positions: block J
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lk + 4 * r, j = lr;
            const int pi = 2 * i + ro, pj = 2 * j + co;
            const double ts = tt[r] * sc;  // (the sweep works on the scaled pivot: rotations are scale-free)
            if (wave == 1) {
                L.S0[pi * LD + pj] = ts;
                L.S0[pj * LD + pi] = ts;
                m2 = fma(2.0 * tt[r], tt[r], m2);
            } else if (i <= j) {  // diagonal sub-blocks: upper triangle mirrored
                L.S0[pi * LD + pj] = ts;
                L.S0[pj * LD + pi] = ts;
                if (full && i < j) m2 = fma(2.0 * tt[r], tt[r], m2);
            }
        }
    }
    SX_ETP(15);
    __syncthreads();  // stage 1 is over: its LDS is reused for the sweep
    SX_ETP(3);
    if constexpr (FLOW) SX_FTP(7);
    const SweepView view{L.S0, L.j.S1, L.cs, L.j.W0};
    if (full)
        (void)pivot_sweep<0>(view, tid);
    else
        (void)pivot_sweep<1>(view, tid);
    __syncthreads();  // the accumulated rotations have left the registers of the last wave
    SX_ETP(4);
    if constexpr (FLOW) SX_FTP(8);
    // U in the order the tiles use (block I first, then block J): gathered index g <-> position 2g or 2(g-16)+1
    const double *Wf = L.j.W0;
    if constexpr (FLOW) {
        // what the two successors wait for goes first: every double as ONE 16-byte store {low half, tag, high half, tag} --
        // two 8-byte granules side by side (pair_poll_granules); then the plain copy the tile workers read, two doubles a store
        const unsigned tag = (unsigned)R.seq;
        const __amdgpu_buffer_rsrc_t ur = flow_rsrc(Uo, (int64_t)kUU * 8);
        if (Go != nullptr) {
            const __amdgpu_buffer_rsrc_t gr = flow_rsrc(Go, (int64_t)kUU * 16);
            for (int e = tid; e < kUU; e += kRoundThreads) {
                const int i = e / kM2, j = e % kM2;
                const int pi = i < kBS ? 2 * i : 2 * (i - kBS) + 1, pj = j < kBS ? 2 * j : 2 * (j - kBS) + 1;
                const unsigned long long b = (unsigned long long)__double_as_longlong(Wf[pi * LD + pj]);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){(unsigned)b, tag, (unsigned)(b >> 32), tag}, gr, e * 16, 0, kAuxSc1);
            }
        }
        for (int e2 = tid; e2 < kUU / 2; e2 += kRoundThreads) {
            const int i = e2 / (kM2 / 2), j = 2 * (e2 % (kM2 / 2));
            const int pi = i < kBS ? 2 * i : 2 * (i - kBS) + 1;
            const int pj0 = j < kBS ? 2 * j : 2 * (j - kBS) + 1, pj1 = pj0 + 2;
            st16(ur, e2 * 16, (f64x2){Wf[pi * LD + pj0], Wf[pi * LD + pj1]});
        }
    } else {
        for (int e = tid; e < kUU; e += kRoundThreads) {
            const int i = e / kM2, j = e % kM2;
            const int pi = i < kBS ? 2 * i : 2 * (i - kBS) + 1, pj = j < kBS ? 2 * j : 2 * (j - kBS) + 1;
            Uo[e] = Wf[pi * LD + pj];
        }
    }
    // (the global atomic waits for nothing here; before a barrier it would stall the whole pivot phase)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m2 += __shfl_xor(m2, off, kWave);
    if (wave < 3 && lane == 0 && m2 != 0.0) atomicAdd(&info->acc[R.sweep], m2);
    SX_ETP(5);
    if constexpr (FLOW) SX_FTP(9);
    return true;
}

// One launch per round.  blockIdx < np: pair workgroups (rotation U_cur of the round rcur from the current pivot);
// then np*np tiles of M and nr*np tiles of V, which apply the rotations U_prev of the round rprev.
// flush != 0: no pair workgroups' sweeps (the last rotations are applied and the run is closed by the host).
__global__ __launch_bounds__(kRoundThreads) void eigh_round_kernel(const double *__restrict__ Min, const double *__restrict__ Vin,
                                                         double *__restrict__ Mout, double *__restrict__ Vout, int npad,
                                                         int nb, const double *__restrict__ Uprev,
                                                         double *__restrict__ Ucur, EighInfo *info, int sweep, int rprev,
                                                         int rcur, int parity_out, double tol, int flush, int seq,
                                                         int refine) {
    RoundLds &L = g_round_lds;
    const int tid = threadIdx.x;
    const int np = nb / 2;
    const RoundPar R{Min, Vin, Mout, Vout, Uprev, Ucur, sweep, rprev, rcur, parity_out, seq};
    // ---- run state: every workgroup derives the same decision from what earlier launches left ----
    // The record is a miss in every cache after the kernel boundary (~1 us).  Nothing below waits for it before the
    // pair workgroups' own loads are in flight: those do not depend on it (a launch that turns out to be a no-op has
    // read a few valid tiles for nothing).
    int ended = 0;
    double nrm2 = 0.0;
    if (tid == 0) ended = info->done_seq, nrm2 = info->norm2;
    const bool is_pair = (int)blockIdx.x < np && !flush;
    PairIdx q{};
    double x0[4], x1[4], x2[4], ua[4], ub[4];
    if (is_pair) {
        q = pair_lookup((int)blockIdx.x, R, nb);
        pair_fetch_tiles<false>(q, Min, npad, tid, x0, x1, x2);
        pair_fetch_u<false>(q, Uprev, tid, ua, ub);
    }
    if (tid == 0) {
        L.flag = round_state<false>(info, ended, nrm2, R, tol, flush, refine, blockIdx.x == 0);
        round_scales(L, nrm2, tol, npad);
    }
    __syncthreads();
    const int state = L.flag;
    if (state == 2) return;
    if ((int)blockIdx.x >= np) {
        // =========================== tile workgroups ===========================
        const int tile = (int)blockIdx.x - np;
        const bool is_m = tile < np * np;
        const int P = is_m ? tile / np : (tile - np * np) / np;  // M: pair of rprev; V: chunk of 32 rows
        const int Q = is_m ? tile % np : (tile - np * np) % np;
        const TileIdx t = tile_index(is_m, P, Q, rprev, nb);
        double x[kTileRegs], uq[kTileRegs], up[kTileRegs];
        tile_fetch<false>(t, is_m ? Min : Vin, Uprev, npad, tid, true, x, uq, up);
        tile_stage(L, is_m, tid, true, x, uq, up);
        __syncthreads();
        tile_compute<false>(L, t, Min, is_m ? Mout : Vout, npad, info, R, refine, tid);
        return;
    }
    // =========================== pair workgroups ===========================
    if (state == 1 || flush) return;
    (void)pair_compute<false>(L, q, x0, x1, x2, ua, ub, Ucur + (int64_t)blockIdx.x * kUU, info, R, tid);
}

// ---------------------------------------------------------------------------------------------------
// All rounds [k0, k1) of a run inside ONE launch (round 6).  The launch-per-round form pays a kernel boundary per round
// -- 2.5 us plus the skew of 16 pair workgroups -- for a dependency that is much narrower than "everything": the pivot
// of pair (I, J) of round k needs the rotations of exactly TWO pair workgroups of round k - 1 and tiles that were
// finished a whole round earlier.  Here every workgroup is resident for the whole run:
//   * np pair workgroups (blockIdx < np): round after round, pair k of round r waits for the rotations of its two
//     predecessors (a per-slot word uflag[slot] = launch number + 1), forms the pivot, sweeps, publishes its rotation;
//   * `workers` tile workgroups: worker w owns the tile columns c = w, w + workers, ... (c = P * np + Q: tile (P, Q) of
//     M and tile (P, Q) of V, which share U_Q); per launch number k it waits for ALL rotations of round k - 1
//     (ucnt[k - 1] == np) and ALL tiles of launch k - 1 (tcnt[k - 1], eight shards), applies, counts itself in.
// Same arithmetic on the same operands as eigh_round_kernel: the two forms return identical bits.
// Why the buffers cannot be overwritten too early: tiles of launch k go to pair (k + 1) & 1, last read by the tiles of
// launch k - 1 (all done: tcnt) and by the pairs of round k - 1 (all done: ucnt); rotations of round k go to slot k % 3,
// last read by the pairs of round k - 2 -- all done before any tile of launch k - 1 started, which every pair of round k
// has waited for -- and by the tiles of launch k - 2.
// Visibility: Guideline 16, form R1 -- write-through stores, every storing wave drains (s_waitcnt vmcnt(0)), a barrier, ONE
// lane bumps the counter; readers poll relaxed and then load past their L1.  Every wait is bounded (tmo wall-clock ticks):
// a wait that runs out sets info->fault, and every other wait ends as soon as it sees that word.
// ---------------------------------------------------------------------------------------------------
struct FlowSync {
    uint32_t *uflag;  // [pairs]      launch number + 1 of the newest rotation in the slot (kept for diagnosis: nothing polls it)
    uint32_t *ucnt;   // [rounds]     pair workgroups that have published launch k's rotations
    uint32_t *tcnt;   // [rounds][8]  tile columns finished in launch k, by worker & 7
    unsigned long long *gran;  // [3][pairs][2 * kUU]  the rotations once more, as tagged halves (pair_poll_granules)
    int64_t gstride;           // words per buffer
};
constexpr int kFlowWaitUflag = 1, kFlowWaitUcnt = 2, kFlowWaitTcnt = 3, kFlowWaitGran = 4;

__device__ __forceinline__ uint32_t flow_ld(const uint32_t *p) {
    return __hip_atomic_load((__attribute__((address_space(1))) uint32_t *)const_cast<uint32_t *>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
// every lane of ONE wave: lanes with `mine` poll their word until all of them read >= want (sum: until the words add up to
// want).  false: gave up (the fault word is set).
__device__ __forceinline__ bool flow_wait(const uint32_t *p, bool mine, uint32_t want, bool sum, EighInfo *info,
                                          long long tmo, int code) {
    const long long t0 = wall_clock64();
    for (unsigned it = 0;; ++it) {
        uint32_t v = mine ? flow_ld(p) : (sum ? 0u : want);
        bool ok;
        if (sum) {
            v += __shfl_xor(v, 1, kWave);
            v += __shfl_xor(v, 2, kWave);
            v += __shfl_xor(v, 4, kWave);
            ok = (uint32_t)__builtin_amdgcn_readfirstlane((int)v) >= want;
        } else {
            ok = __all(v >= want);
        }
        if (ok) return true;
        if ((it & 15) == 15) {
            if (flow_ld((const uint32_t *)&info->fault) != 0u) return false;
            if (wall_clock64() - t0 > tmo) {
                __hip_atomic_store((__attribute__((address_space(1))) uint32_t *)&info->fault, (uint32_t)code, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// the whole workgroup: wave 0 waits, everybody learns the outcome (one barrier)
__device__ __forceinline__ bool flow_wait_block(RoundLds &L, const uint32_t *p, bool mine, uint32_t want, bool sum,
                                                EighInfo *info, long long tmo, int code, int tid) {
    if (tid < kWave) {
        const bool ok = flow_wait(p, mine, want, sum, info, tmo, code);
        if (tid == 0) L.ok = ok ? 1 : 0;
    }
    __syncthreads();
    return L.ok != 0;
}
// The rotation of a pair travels to its two successors as "granules" (Guideline 16, form R2): every double as two aligned
// 8-byte words {launch number + 1 : 32 | half of the value : 32}, each ONE agent-scope store.  A reader that finds the
// expected tag in a word has its data: arrival is detected on the data itself -- no drain, no flag, no second trip.  The
// buffers are zeroed at the start of a run and tags are never 0; a slot is reused every third round with another tag.
// Threads 0..255, two elements (four words) per predecessor: rows t / 16 and t / 16 + 16, column t % 16 of the sixteen
// columns that belong to the block this pair inherits.  Every wave polls for itself and then writes what it has read
// into L.p.UA / L.p.UB.  A wave whose wait runs out clears L.ok (checked behind the staging barrier).
struct GranPtr {
    __amdgpu_buffer_rsrc_t r;
    int off[4];
};
__device__ __forceinline__ GranPtr pair_gran_ptrs(const PairIdx &q, const unsigned long long *Gprev, int64_t gstride, int tid) {
    const int c = tid & 15, r0 = (tid & 255) >> 4;
    const int ea0 = r0 * kM2 + 16 * q.posI + c, ea1 = ea0 + 16 * kM2;
    const int eb0 = r0 * kM2 + 16 * q.posJ + c, eb1 = eb0 + 16 * kM2;
    return GranPtr{flow_rsrc(Gprev, gstride * 8),
                   {(q.PI * kUU + ea0) * 16, (q.PI * kUU + ea1) * 16, (q.PJ * kUU + eb0) * 16, (q.PJ * kUU + eb1) * 16}};
}
__device__ __forceinline__ void pair_gran_load(const GranPtr &g, unsigned long long (&v)[8]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(g.r, g.off[u], 0, kAuxSc1);
        v[2 * u] = ((unsigned long long)w.y << 32) | w.x;
        v[2 * u + 1] = ((unsigned long long)w.w << 32) | w.z;
    }
}
// v: a first pass that is already on its way (pair_gran_load); re-read until every word carries the tag
__device__ __forceinline__ void pair_poll_granules(RoundLds &L, const PairIdx &q, const GranPtr &g, unsigned long long (&v)[8],
                                                   uint32_t tag, EighInfo *info, long long tmo, int tid) {
    if (tid >= 256) return;
    const int c = tid & 15, r0 = tid >> 4;
    const long long t0 = wall_clock64();
    for (unsigned it = 0;; ++it) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < 8; ++u) ok = ok && (uint32_t)(v[u] >> 32) == tag;
        if (__all(ok)) break;
        if ((it & 15) == 15) {
            bool give_up = flow_ld((const uint32_t *)&info->fault) != 0u;
            if (!give_up && wall_clock64() - t0 > tmo) {
                __hip_atomic_store((__attribute__((address_space(1))) uint32_t *)&info->fault, (uint32_t)kFlowWaitGran,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                give_up = true;
            }
            if (give_up) {
                L.ok = 0;
                break;
            }
        }
        if (it != 0) __builtin_amdgcn_s_sleep(1);
        pair_gran_load(g, v);
    }
    auto join = [](unsigned long long lo, unsigned long long hi) {
        return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
    };
    L.p.UA[r0 * LDU + 16 * q.posI + c] = join(v[0], v[1]);
    L.p.UA[(r0 + 16) * LDU + 16 * q.posI + c] = join(v[2], v[3]);
    L.p.UB[r0 * LDU + 16 * q.posJ + c] = join(v[4], v[5]);
    L.p.UB[(r0 + 16) * LDU + 16 * q.posJ + c] = join(v[6], v[7]);
}

// every storing wave has drained; then ONE lane publishes
#define SX_FLOW_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

__device__ __forceinline__ RoundPar flow_round_par(double *M0, double *M1, double *V0, double *V1, double *U, int64_t us, int nb,
                                                   int k) {
    const int cur = k & 1, rps = nb - 1;
    return RoundPar{cur ? M1 : M0, cur ? V1 : V0, cur ? M0 : M1, cur ? V0 : V1, U + (int64_t)((k + 2) % 3) * us,
                    U + (int64_t)(k % 3) * us, k / rps, k == 0 ? 0 : (k - 1) % rps, k % rps, cur ^ 1, k + 1};
}

// (The pair rounds were also tried as a function of their own, to keep them at eigh_round_kernel's 120 VGPRs so that two
// workgroups fit a CU: the 42 callee-saved registers such a function saves and restores through scratch memory cost 2.1 us
// per round -- tools/trace_eigh_flow.py.  Inlined the kernel takes ~160 VGPRs and ONE workgroup per CU is all that is counted
// on: the grid is kept within the number of CUs.)
__global__ __launch_bounds__(kRoundThreads) void eigh_flow_kernel(double *M0, double *M1, double *V0, double *V1, double *U,
                                                                  int64_t ustride, int npad, int nb, EighInfo *info,
                                                                  FlowSync S, int k0, int k1, double tol, int refine,
                                                                  int workers, long long tmo, int use_gran) {
    RoundLds &L = g_round_lds;
    const int np = nb / 2, ncol = np * np;
    const bool is_pair = (int)blockIdx.x < np;
    const int widx = (int)blockIdx.x - np;
    if (threadIdx.x == 0) {
        const int ended = (int)flow_ld((const uint32_t *)&info->done_seq);
        const double nrm2 = gld<true>(&info->norm2);
        L.flag = ended;
        L.red[16] = nrm2;
        L.ok = 1;
        round_scales(L, nrm2, tol, npad);
    }
    __syncthreads();
    // Launch number k + 1 does nothing once a launch with a SMALLER number has ended the run (eigh_round_kernel's rule, also
    // how a skipped decomposition -- done_seq = 1 from eigh_prepare_kernel -- is passed over).  A workgroup that starts late
    // may read what THIS launch has meanwhile recorded (k' + 1 at round k'): it then still works through the rounds up to
    // k', as everybody else did, and takes the same decision there.
    const int ended0 = L.flag;
    const double nrm2 = L.red[16];
    __syncthreads();
    if (is_pair) {
        // ---- pair workgroups: a software pipeline over the rounds.  What the two successors wait for -- the rotation as
        // tagged halves -- leaves first; then the NEXT round's loads are issued (its tiles, a look at their counter, a first pass
        // over the predecessors' words), and only then the stores are drained and counted: one trip to memory covers both.
        if (k0 >= k1 || (ended0 != 0 && ended0 <= k0)) return;
        RoundPar R = flow_round_par(M0, M1, V0, V1, U, ustride, nb, k0);
        PairIdx q = pair_lookup((int)blockIdx.x, R, nb);
        double x0[4], x1[4], x2[4], ua[4], ub[4];
        {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            if (R.sweep > 0 && R.rcur <= 1) {  // (a launch may start at a round that can end the run: everything it reads is final)
                if (tid == 0) L.flag = round_state<true>(info, 0, nrm2, R, tol, 0, refine, blockIdx.x == 0);
                __syncthreads();
                if (L.flag != 0) return;
            }
            pair_fetch_tiles<true>(q, R.Min, npad, tid, x0, x1, x2);
            pair_fetch_u<true>(q, R.Uprev, tid, ua, ub);
        }
        for (int k = k0;; ++k) {
            // (the thread index made opaque in every round: otherwise every thread-invariant address of the pivot sweep is
            // hoisted out of the round loop and kept alive across it)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63;
            SX_FT_BEGIN(k);
            unsigned long long *Go = (use_gran & 1) ? S.gran + (int64_t)(k % 3) * S.gstride + (int64_t)blockIdx.x * (2 * kUU) : nullptr;
            if (!pair_compute<true>(L, q, x0, x1, x2, ua, ub, R.Ucur + (int64_t)blockIdx.x * kUU, info, R, tid, (use_gran & 1) && k > k0, Go))
                return;
            const int kn = k + 1;
            const bool more = kn < k1 && !(ended0 != 0 && ended0 <= kn);
            GranPtr g = pair_gran_ptrs(q, S.gran, S.gstride, tid);
            unsigned long long gv[8];
            uint32_t tcv = 0;
            bool decide = false;
            if (more) {
                R = flow_round_par(M0, M1, V0, V1, U, ustride, nb, kn);
                q = pair_lookup((int)blockIdx.x, R, nb);
                decide = R.sweep > 0 && R.rcur <= 1;
                g = pair_gran_ptrs(q, S.gran + (int64_t)(k % 3) * S.gstride, S.gstride, tid);
            }
            const bool early = (use_gran & 2) == 0;  // (bit 1 of the switch: the next round's loads only behind the drain)
            if (more && early) {
                if (!decide) {
                    // The tiles of launch k were finished well before this pair's predecessors were (they only needed the
                    // rotations of round k - 1), so they are loaded at once, beside the look at their counter -- and loaded again
                    // in the rare case that the counter says they were not complete (the loads bypass the vector L1)
                    pair_fetch_tiles<true>(q, R.Min, npad, tid, x0, x1, x2);
                    if (tid < 8) tcv = flow_ld(S.tcnt + (int64_t)k * 8 + lane);
                }
                if ((use_gran & 1) && tid < 256) pair_gran_load(g, gv);
            }
            SX_FTP(2);
            SX_FLOW_DRAIN();
            __syncthreads();
            SX_FTP(10);
            if (more && !early) {
                if (!decide) {
                    pair_fetch_tiles<true>(q, R.Min, npad, tid, x0, x1, x2);
                    if (tid < 8) tcv = flow_ld(S.tcnt + (int64_t)k * 8 + lane);
                }
                if ((use_gran & 1) && tid < 256) pair_gran_load(g, gv);
            }
            if (tid == 0) {
                L.ok = 1;
                __hip_atomic_store((__attribute__((address_space(1))) uint32_t *)(S.uflag + blockIdx.x), (uint32_t)(k + 1),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add((__attribute__((address_space(1))) uint32_t *)(S.ucnt + k), 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            SX_FTP(11);
            if (!more) return;
            if (decide) {
                // a round that may end the run (2 of a sweep's rounds): first the sums the rules read -- the tiles of launch k,
                // and for the rule of round 0 the mass met by ALL pairs of the sweep that just ended
                if (!flow_wait_block(L, S.tcnt + (int64_t)k * 8 + lane, lane < 8, (uint32_t)ncol, true, info, tmo, kFlowWaitTcnt, tid))
                    return;
                if (R.rcur == 0 && !flow_wait_block(L, S.ucnt + k, lane == 0, (uint32_t)np, false, info, tmo, kFlowWaitUcnt, tid))
                    return;
                if (tid == 0) L.flag = round_state<true>(info, 0, nrm2, R, tol, 0, refine, blockIdx.x == 0);
                __syncthreads();
                if (L.flag != 0) return;
                pair_fetch_tiles<true>(q, R.Min, npad, tid, x0, x1, x2);
            } else {
                if (tid < kWave) {
                    tcv += __shfl_xor(tcv, 1, kWave);
                    tcv += __shfl_xor(tcv, 2, kWave);
                    tcv += __shfl_xor(tcv, 4, kWave);
                    if (tid == 0) L.flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)tcv) >= (uint32_t)ncol ? 0 : 1;
                }
                __syncthreads();
                if (L.flag != 0) {  // (rare) the tiles were not complete when they were loaded: wait, load again
                    if (!flow_wait_block(L, S.tcnt + (int64_t)k * 8 + lane, lane < 8, (uint32_t)ncol, true, info, tmo, kFlowWaitTcnt, tid))
                        return;
                    pair_fetch_tiles<true>(q, R.Min, npad, tid, x0, x1, x2);
                }
            }
            // the rotations of the two pairs of round k that held blocks I and J
            if (use_gran & 1) {  // (as tagged halves, into L.p.UA / L.p.UB)
                pair_poll_granules(L, q, g, gv, (uint32_t)(k + 1), info, tmo, tid);
            } else {         // (the plain copy, behind the slots' words)
                const uint32_t *pf = S.uflag + (lane == 1 ? q.PJ : q.PI);
                if (!flow_wait_block(L, pf, lane < 2, (uint32_t)(k + 1), false, info, tmo, kFlowWaitUflag, tid)) return;
                pair_fetch_u<true>(q, R.Uprev, tid, ua, ub);
            }
            SX_FTP(3);
        }
    }
    for (int k = k0; k < k1; ++k) {
        if (ended0 != 0 && ended0 <= k) return;
        // (the thread index made opaque in every round: otherwise every thread-invariant address is hoisted out of the
        // round loop and kept alive across it)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const RoundPar R = flow_round_par(M0, M1, V0, V1, U, ustride, nb, k);
        const bool inside = k > k0;
        SX_FT_BEGIN(k);
        __syncthreads();
        SX_FTP(0);
        // the tiles of launch k - 1 (what the tiles of launch k read, and the sums the stopping rules read) and ALL rotations
        // of round k - 1, in one wait: lanes 0..7 add up the tile counter's shards, lane 8 looks at the rotations' counter
        if (inside) {
            if (tid < kWave) {
                const uint32_t *p = lane < 8 ? S.tcnt + (int64_t)(k - 1) * 8 + lane : S.ucnt + (k - 1);
                const long long t0 = wall_clock64();
                bool ok = false;
                for (unsigned it = 0;; ++it) {
                    const uint32_t v = lane < 9 ? flow_ld(p) : 0u;
                    uint32_t sum = lane < 8 ? v : 0u;
                    sum += __shfl_xor(sum, 1, kWave);
                    sum += __shfl_xor(sum, 2, kWave);
                    sum += __shfl_xor(sum, 4, kWave);
                    const uint32_t tdone = (uint32_t)__builtin_amdgcn_readfirstlane((int)sum);
                    const uint32_t udone = (uint32_t)__builtin_amdgcn_readlane((int)v, 8);
                    if (tdone >= (uint32_t)ncol && udone >= (uint32_t)np) {
                        ok = true;
                        break;
                    }
                    if ((it & 15) == 15) {
                        if (flow_ld((const uint32_t *)&info->fault) != 0u) break;
                        if (wall_clock64() - t0 > tmo) {
                            __hip_atomic_store((__attribute__((address_space(1))) uint32_t *)&info->fault,
                                               (uint32_t)(tdone >= (uint32_t)ncol ? kFlowWaitUcnt : kFlowWaitTcnt), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (tid == 0) L.ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (L.ok == 0) return;
        }
        SX_FTP(2);
        int state = 0;
        if (R.sweep > 0 && R.rcur <= 1) {
            if (tid == 0) L.flag = round_state<true>(info, 0, nrm2, R, tol, 0, refine, false);
            __syncthreads();
            state = L.flag;
            if (state == 2) return;
        }
        uint32_t mine = 0;
        const int64_t mbytes = (int64_t)npad * npad * 8;
        const __amdgpu_buffer_rsrc_t rm = flow_rsrc(R.Min, mbytes), rv = flow_rsrc(R.Vin, mbytes), ru = flow_rsrc(R.Uprev, ustride * 8);
        // two tile columns at a time, all their loads in flight at once (one workgroup per CU has nothing else to hide a
        // trip to memory behind)
        for (int c = widx; c < ncol; c += 2 * workers) {
            const int c1 = c + workers;
            const bool two = c1 < ncol;
            const int P = c / np, Q = c % np, P1 = two ? c1 / np : P, Q1 = two ? c1 % np : Q;
            const TileIdx tm = tile_index(true, P, Q, R.rprev, nb), tv = tile_index(false, P, Q, R.rprev, nb);
            const TileIdx sm = tile_index(true, P1, Q1, R.rprev, nb), sv = tile_index(false, P1, Q1, R.rprev, nb);
            f64x2 xm[kTileRegs2], uq[kTileRegs2], up[kTileRegs2], xv[kTileRegs2];
            f64x2 ym[kTileRegs2], vq[kTileRegs2], vp[kTileRegs2], yv[kTileRegs2];
            tile_fetch16(tm, rm, ru, npad, tid, true, xm, uq, up);
            tile_fetch16(tv, rv, ru, npad, tid, false, xv, uq, up);
            if (two) {
                tile_fetch16(sm, rm, ru, npad, tid, true, ym, vq, vp);
                tile_fetch16(sv, rv, ru, npad, tid, false, yv, vq, vp);
            }
            tile_stage16(L, true, tid, true, xm, uq, up);
            __syncthreads();
            SX_FTP(4);
            tile_compute<true>(L, tm, R.Min, R.Mout, npad, info, R, refine, tid);
            __syncthreads();
            SX_FTP(5);
            tile_stage16(L, false, tid, false, xv, uq, up);
            __syncthreads();
            tile_compute<true>(L, tv, R.Min, R.Vout, npad, info, R, refine, tid);
            __syncthreads();
            SX_FTP(6);
            ++mine;
            if (two) {
                tile_stage16(L, true, tid, true, ym, vq, vp);
                __syncthreads();
                tile_compute<true>(L, sm, R.Min, R.Mout, npad, info, R, refine, tid);
                __syncthreads();
                tile_stage16(L, false, tid, false, yv, vq, vp);
                __syncthreads();
                tile_compute<true>(L, sv, R.Min, R.Vout, npad, info, R, refine, tid);
                __syncthreads();
                ++mine;
            }
        }
        SX_FLOW_DRAIN();
        __syncthreads();
        SX_FTP(7);
        if (tid == 0 && mine != 0)
            __hip_atomic_fetch_add((__attribute__((address_space(1))) uint32_t *)(S.tcnt + (int64_t)k * 8 + (widx & 7)), mine,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SX_FTP(8);
        if (state == 1) return;
    }
}

// the run ends without the rule having fired: record where the result lives
// `fails` (a word of the info block that sx_eigh does not clear) counts the runs of this workspace that ended short of
// their tolerance: a loop that reads the record only every few decompositions still learns that one of them fell short.
__global__ void eigh_close_kernel(EighInfo *info, int sweeps, int parity, double tol, int refine, int *fails) {
    if (threadIdx.x != 0) return;
    if (info->fault) {  // the resident kernel gave up on a wait: whatever the record says, the result is not a decomposition
        info->sweeps = sweeps, info->parity = parity, info->converged = 0, info->refine = 0;
        atomicAdd(fails, 1);
        info->done_seq = 1;
        return;
    }
    if (info->done_seq) return;
    info->sweeps = sweeps, info->parity = parity, info->thr2 = tol * tol * info->norm2;
    int conv = (sweeps > 0 && (info->offm[sweeps - 1] <= info->thr2 ||
                               eigh_last_sweep(info->acc, sweeps - 1, info->norm2, tol))) ? 1 : 0;
    if (!conv && refine && sweeps > 0) {  // the allowance ran out where the refinement step takes over: let it
        const double left = info->offm[sweeps - 1], kmax2 = __longlong_as_double((long long)info->kmax2[sweeps - 1]);
        if (left <= kRefineOff * kRefineOff * info->norm2 && kmax2 <= kRefineCap * kRefineCap &&
            kmax2 * left <= kRefineProd * kRefineProd * info->norm2)
            info->refine = 1, conv = 1;
        else if ((refine & 2) && left <= kDeepOff * kDeepOff * info->norm2 && kmax2 <= kDeepCap * kDeepCap)
            info->refine = 2, conv = 1;
    }
    info->converged = conv;
    if (!conv) atomicAdd(fails, 1);
    info->done_seq = 1;
}

// per column j of V: |v_j|^2 and the sign of its largest-magnitude component (lowest row on ties);
// lam[j] = M_jj / |v_j|^2, scl[j] = sign / |v_j|.  One workgroup per 16 columns, 16 row strips (rows of 128 bytes).
__global__ __launch_bounds__(1024) void eigh_colstats_kernel(const double *__restrict__ M0, const double *__restrict__ M1,
                                                             const double *__restrict__ V0, const double *__restrict__ V1,
                                                             int n, int npad, const EighInfo *info,
                                                             double *__restrict__ lam, double *__restrict__ scl) {
    constexpr int RS = 64;  // row strips (round 3: 16 strips of 32 serial loads each took 15 us at n = 512)
    __shared__ double s_n2[RS][16], s_mx[RS][16], s_sg[RS][16];
    __shared__ int s_ix[RS][16];
    const double *M = info->parity ? M1 : M0, *V = info->parity ? V1 : V0;
    if (info->refine == 1) V = info->parity ? M0 : M1;  // the refinement step left the eigenvectors in the free M buffer
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + tx;
    double n2 = 0.0, mx = -1.0, sg = 1.0;
    int ix = 0;
    if (j < n) {
        for (int i0 = ty; i0 < n; i0 += 8 * RS) {  // eight loads in flight
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + u * RS < n ? V[(int64_t)(i0 + u * RS) * npad + j] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                n2 += v[u] * v[u];
                const double a = fabs(v[u]);
                if (i0 + u * RS < n && a > mx) mx = a, ix = i0 + u * RS, sg = v[u] < 0.0 ? -1.0 : 1.0;
            }
        }
    }
    s_n2[ty][tx] = n2, s_mx[ty][tx] = mx, s_sg[ty][tx] = sg, s_ix[ty][tx] = ix;
    __syncthreads();
    if (ty == 0) {
        if (j < n) {
            double tot = 0.0, bm = -1.0, bs = 1.0;
            int bi = 0;
#pragma unroll 8
            for (int k = 0; k < RS; ++k) {
                tot += s_n2[k][tx];
                const double m = s_mx[k][tx];
                if (m > bm || (m == bm && s_ix[k][tx] < bi)) bm = m, bi = s_ix[k][tx], bs = s_sg[k][tx];
            }
            lam[j] = M[(int64_t)j * npad + j] / tot;
            scl[j] = bs / sqrt(tot);
        } else if (j < npad) {
            lam[j] = __builtin_inf();
            scl[j] = 0.0;
        }
    }
}

// ascending rank of every eigenvalue (ties: lower position first); inv[rank] = position, w[rank] = value.
// A wavefront ranks four values: lane l looks at values l, l + 64, ... (through LDS, 4096 at a time) and the rank is
// the population count of the wave's votes (round 3: one workgroup with n serial comparisons per thread took 16 us).
constexpr int kEigRankPerWave = 4;
__global__ __launch_bounds__(256) void eigh_rank_kernel(const double *__restrict__ lam, int n, int *__restrict__ inv,
                                                        double *__restrict__ w) {
    constexpr int CH = 4096;
    __shared__ double keys[CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = ((int)blockIdx.x * 4 + wave) * kEigRankPerWave;
    double lj[kEigRankPerWave];
    int rank[kEigRankPerWave];
#pragma unroll
    for (int u = 0; u < kEigRankPerWave; ++u) lj[u] = j0 + u < n ? lam[j0 + u] : 0.0, rank[u] = 0;
    for (int c0 = 0; c0 < n; c0 += CH) {
        const int len = n - c0 < CH ? n - c0 : CH;
        __syncthreads();
        for (int e = threadIdx.x; e < len; e += 256) keys[e] = lam[c0 + e];
        __syncthreads();
        for (int k0 = 0; k0 < len; k0 += 64) {  // uniform trip count: every lane adds every vote
            const int k = k0 + lane;
            const bool in = k < len;
            const double lk = in ? keys[k] : 0.0;
#pragma unroll
            for (int u = 0; u < kEigRankPerWave; ++u)
                rank[u] += (int)__popcll(__ballot(in && (lk < lj[u] || (lk == lj[u] && c0 + k < j0 + u))));
        }
    }
#pragma unroll
    for (int u = 0; u < kEigRankPerWave; ++u) {
        if (lane == u && j0 + u < n) {
            inv[rank[u]] = j0 + u;
            w[rank[u]] = lj[u];
        }
    }
}

// B[i][r] = V[i][inv[r]] * scl[inv[r]]   (eigenvectors in columns, row-major like numpy's)
__global__ __launch_bounds__(256) void eigh_write_kernel(const double *__restrict__ M0, const double *__restrict__ M1,
                                                         const double *__restrict__ V0, const double *__restrict__ V1,
                                                         int n, int npad, const EighInfo *info,
                                                         const int *__restrict__ inv, const double *__restrict__ scl,
                                                         double *__restrict__ B) {
    const double *V = info->parity ? V1 : V0;
    if (info->refine == 1) V = info->parity ? M0 : M1;
    const int i = blockIdx.x;
    for (int r = threadIdx.x; r < n; r += 256) {
        const int j = inv[r];
        B[(int64_t)i * n + r] = V[(int64_t)i * npad + j] * scl[j];
    }
}

// n <= 32: one workgroup does the whole decomposition; above, the block method (round 3: also for 33 <= n <= 64 -- four blocks,
// three rounds per sweep, warm start and refinement step included: 507 us per decomposition at n = 64 in the one-workgroup
// form, whose 1 024 updating threads move S AND W through LDS every inner round)
constexpr int kSmallPathMax = 32;
inline int eigh_npad(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : (n <= 64 ? 64 : ((n + kM2 - 1) / kM2) * kM2)); }

struct EighWs {
    EighInfo *info;
    double *M[2], *V[2], *U[3], *W[3], *lam, *scl;  // (W: the deep refinement step's work matrices)
    int *inv;
    int64_t ustride;  // doubles per rotation buffer
    int64_t ucount;   // doubles in all three rotation buffers
    uint32_t *sync;   // the resident kernel's words: uflag[pairs16] | ucnt[rounds] | tcnt[rounds][8] | (64-byte aligned)
                      // gran[3][pairs][2 * kUU] 8-byte words
    int64_t sync_words, pairs16, rounds, gran_off, gstride;  // (sync_words, gran_off in 32-bit words)
    int64_t bytes;
};

inline EighWs eigh_layout(void *ws, int n) {
    const int64_t np = eigh_npad(n);
    char *p = (char *)ws;
    EighWs w;
    int64_t off = 2048;  // EighInfo
    static_assert(sizeof(EighInfo) <= kEighFailsOffset, "info block");
    w.info = (EighInfo *)p;
    for (int k = 0; k < 2; ++k) w.M[k] = (double *)(p + off), off += np * np * 8;
    for (int k = 0; k < 2; ++k) w.V[k] = (double *)(p + off), off += np * np * 8;
    const int64_t pairs = np / kM2 + 1;
    w.ustride = pairs * kUU;
    w.ucount = 3 * pairs * kUU;
    for (int k = 0; k < 3; ++k) w.U[k] = (double *)(p + off), off += pairs * kUU * 8;
    for (int k = 0; k < 3; ++k) w.W[k] = (double *)(p + off), off += np * np * 8;
    w.lam = (double *)(p + off), off += np * 8;
    w.scl = (double *)(p + off), off += np * 8;
    w.inv = (int *)(p + off), off += np * 8;
    w.pairs16 = (pairs + 15) / 16 * 16;
    w.rounds = (int64_t)kEighMaxSweeps * (np / kBS - 1) + 2;  // launch numbers of a run
    w.gran_off = (w.pairs16 + 9 * w.rounds + 15) / 16 * 16;
    w.gstride = pairs * 2 * kUU;
    w.sync_words = w.gran_off + 2 * 3 * w.gstride;
    w.sync = (uint32_t *)(p + off), off += (w.sync_words * 4 + 63) / 64 * 64;
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" int64_t sx_eigh_workspace_bytes(int n) {
    if (n < 1) return -1;
    return eigh_layout(nullptr, n).bytes;
}

namespace sx {
// The refinement step: the CMA-ES loops take it unless told otherwise, sx_eigh itself only when asked
// (sx_eigh_set_refine, or the environment variable SX_EIGH_REFINE = 0 / 1 read once).
static int g_refine_mode = -1;  // -1: SX_EIGH_REFINE or the defaults; 0: never; 1: always
static int refine_env() {
    static const int v = [] {
        const char *e = getenv("SX_EIGH_REFINE");
        return e == nullptr ? -1 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : -1));
    }();
    return v;
}
// One resident launch for all rounds of a run (eigh_flow_kernel) instead of one launch per round: OFF unless asked for
// (sx_eigh_set_flow, or the environment variable SX_EIGH_FLOW = 1 read once) -- measured on MI355X it does not beat the
// launch per round (profiles/r6_eigh_flow.txt: a rotation handed from CU to CU under the tile workers' traffic takes 3-4.5 us,
// what a kernel boundary costs).  The resident form needs every workgroup of its grid on the chip at once: processes that
// SHARE a GPU (the tests' ranks on one device) must not use it.
static int g_flow_mode = -1;
static int flow_env() {
    static const int v = [] {
        const char *e = getenv("SX_EIGH_FLOW");
        return e == nullptr ? -1 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : -1));
    }();
    return v;
}
static int flow_workers_env() {
    static const int v = [] {
        const char *e = getenv("SX_EIGH_FLOW_WORKERS");
        return e == nullptr ? 0 : atoi(e);
    }();
    return v;
}
static int flow_gran_env() {
    static const int v = [] {
        const char *e = getenv("SX_EIGH_FLOW_GRAN");
        return e == nullptr ? 0 : atoi(e);
    }();
    return v;
}
static long long flow_timeout_ticks() {
    static const long long v = [] {
        const char *e = getenv("SX_EIGH_FLOW_TIMEOUT_MS");
        const double ms = e == nullptr ? 2000.0 : atof(e);
        int khz = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
            khz = 100000;  // wall_clock64 counts at 100 MHz on gfx950
        return (long long)(ms * (double)khz);
    }();
    return v;
}
int eigh_flow_on() {
    const int m = g_flow_mode >= 0 ? g_flow_mode : flow_env();
    return m == 1 ? 1 : 0;
}
int eigh_refine_default() {
    const int m = g_refine_mode >= 0 ? g_refine_mode : refine_env();
    return m == 1 ? 1 : 0;
}
int eigh_refine_in_loops() {
    const int m = g_refine_mode >= 0 ? g_refine_mode : refine_env();
    return m == 0 ? 0 : 1;
}
// sx_eigh with a device-side skip flag (the CMA-ES loops' done word): see eigh_prepare_kernel.
// phases (bits): 1 = begin (run record, M = V0^T C V0 or C, V = V0 or I), 2 = the rounds [r0, r1) of the tournament (round r
// belongs to sweep r / (nb - 1); which buffers it reads, whose rotations it applies and its launch number all follow from
// r alone, so a caller may enqueue the rounds in pieces and look at the run record in between), 4 = finish (the last
// rotations, the record's closing, the refinement step, eigenvalues / order / signs into w and B).
int eigh_enqueue_phased(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes, double tol,
                        const int *skip, int refine, void *stream, int phases, int r0, int r1) {
    SX_REQUIRE(C && w && B && ws && n >= 1 && n <= 32768, "sx_eigh: bad arguments");
    const EighWs L = eigh_layout(ws, n);
    SX_REQUIRE(ws_bytes >= L.bytes, "sx_eigh: workspace too small (sx_eigh_workspace_bytes)");
    if (!(tol > 0.0)) tol = 1.0e-14;
    hipStream_t st = (hipStream_t)stream;
    const int npad = eigh_npad(n);
    // bit 0: the first-order refinement step allowed; bit 1: the deep one too (SX_EIGH_DEEP=0 switches it off)
    static const bool deep_on = [] { const char *e = getenv("SX_EIGH_DEEP"); return !(e != nullptr && e[0] == '0'); }();
    refine = refine ? (1 | ((deep_on && npad >= 256) ? 2 : 0)) : 0;
    const int nb = npad / kBS, np = nb / 2, rps = nb - 1;
    SX_REQUIRE(r0 >= 0 && r1 >= r0 && r1 <= kEighMaxSweeps * (n <= kSmallPathMax ? 1 : rps), "sx_eigh: bad round range");
    if (phases & 1) {
        SX_HIP(hipMemsetAsync(L.info, 0, sizeof(EighInfo), st));
        if (n > kSmallPathMax) SX_HIP(hipMemsetAsync(L.sync, 0, (size_t)L.sync_words * 4, st));  // (Guideline 16: every polled word, every call)
    }
    if (n <= kSmallPathMax) {
        if (phases & 1) {
            const int max_sweeps = std::max(1, std::min(kEighMaxSweeps, r1));  // (one workgroup: r1 counts sweeps here)
            if (npad == 16)
                hipLaunchKernelGGL((eigh_small_kernel<16>), dim3(1), dim3(jacobi_threads<16>()), 0, st, C, n, L.M[0], L.V[0], L.info, max_sweeps, tol);
            else if (npad == 32)
                hipLaunchKernelGGL((eigh_small_kernel<32>), dim3(1), dim3(jacobi_threads<32>()), 0, st, C, n, L.M[0], L.V[0], L.info, max_sweeps, tol);
            else
                hipLaunchKernelGGL((eigh_small_kernel<64>), dim3(1), dim3(jacobi_threads<64>()), 0, st, C, n, L.M[0], L.V[0], L.info, max_sweeps, tol);
            SX_LAUNCH_CHECK();
        }
    } else {
        if (phases & 1) {
            const int64_t tot = std::max<int64_t>((int64_t)npad * npad, L.ucount);
            const unsigned pgrid = (unsigned)std::min<int64_t>((tot + 255) / 256, kPrepareMaxBlocks);
            if (V0 == nullptr) {
                hipLaunchKernelGGL(eigh_prepare_kernel, dim3(pgrid), dim3(256), 0, st, C, n, npad,
                                   (const double *)nullptr, L.M[0], L.V[0], L.U[0], L.ucount, L.info, skip);
            } else {
                // Warm start from a nearly orthonormal basis (the previous generation's eigenvectors):
                //   V <- V0 (3 I - V0^T V0) / 2   one Newton-Schulz step: orthonormal to rounding, so that a basis handed
                //                                 from decomposition to decomposition cannot drift
                //   M <- V^T (C V)
                // Four n^3 products on the matrix cores (~1 % of a cold decomposition); the sweeps then start from a
                // nearly diagonal M and the stopping rule ends them after the few that are needed.
                const dim3 gg((unsigned)(npad / kM2), (unsigned)(npad / kM2));
                hipLaunchKernelGGL(eigh_prepare_kernel, dim3(pgrid), dim3(256), 0, st, C, n, npad, V0,
                                   L.M[1], L.V[1], L.U[0], L.ucount, L.info, skip);                 // M1 = C, V1 = V0
                hipLaunchKernelGGL((eigh_gemm_kernel<true>), gg, dim3(256), 0, st, L.V[1], L.V[1], L.M[0], npad, -0.5, 1.5);
                hipLaunchKernelGGL((eigh_gemm_kernel<false>), gg, dim3(256), 0, st, L.V[1], L.M[0], L.V[0], npad, 1.0, 0.0);
                hipLaunchKernelGGL((eigh_gemm_kernel<false>), gg, dim3(256), 0, st, L.M[1], L.V[0], L.V[1], npad, 1.0, 0.0);
                hipLaunchKernelGGL((eigh_gemm_kernel<true>), gg, dim3(256), 0, st, L.V[0], L.V[1], L.M[0], npad, 1.0, 0.0);
            }
            SX_LAUNCH_CHECK();
        }
        const unsigned grid = (unsigned)(np + np * np + (npad / kM2) * np);
        if (phases & 2) {
            // launch k (0-based) reads the buffer pair k & 1, applies the rotations of round (k - 1) % rps (identities for
            // k = 0 under any pairing) and works out those of round k % rps; its launch number is k + 1 (RoundPar)
            // ONE workgroup per CU (193 VGPRs): pair workgroups + workers within the CU count, or nothing is guaranteed resident
            static const int cus = [] {
                int dev = 0, v = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
                return v;
            }();
            int workers = flow_workers_env() > 0 ? flow_workers_env() : 128;
            workers = std::max(1, std::min(std::min(workers, np * np), cus - np));
            const bool flow = eigh_flow_on() && r1 > r0 && np + workers <= cus && 2 * np <= cus && r1 <= (int)L.rounds;
            if (flow) {
                const FlowSync S{L.sync, L.sync + L.pairs16, L.sync + L.pairs16 + L.rounds,
                                 (unsigned long long *)(L.sync + L.gran_off), L.gstride};
                hipLaunchKernelGGL(eigh_flow_kernel, dim3((unsigned)(np + workers)), dim3(kRoundThreads), 0, st, L.M[0], L.M[1],
                                   L.V[0], L.V[1], L.U[0], L.ustride, npad, nb, L.info, S, r0, r1, tol, refine, workers,
                                   flow_timeout_ticks(), flow_gran_env());
            } else {
                for (int k = r0; k < r1; ++k) {
                    const int cur = k & 1, rprev = k == 0 ? 0 : (k - 1) % rps;
                    hipLaunchKernelGGL(eigh_round_kernel, dim3(grid), dim3(kRoundThreads), 0, st, L.M[cur], L.V[cur], L.M[cur ^ 1],
                                       L.V[cur ^ 1], npad, nb, L.U[(k + 2) % 3], L.U[k % 3], L.info, k / rps, rprev, k % rps, cur ^ 1,
                                       tol, 0, k + 1, refine);
                }
            }
            SX_LAUNCH_CHECK();
        }
        if (phases & 4) {
            // apply the last rotations, then close (r1 rounds have been enqueued; a run that ended earlier ignores both)
            const int cur = r1 & 1, rprev = r1 == 0 ? 0 : (r1 - 1) % rps, sweeps = r1 / rps;
            hipLaunchKernelGGL(eigh_round_kernel, dim3(grid), dim3(kRoundThreads), 0, st, L.M[cur], L.V[cur], L.M[cur ^ 1],
                               L.V[cur ^ 1], npad, nb, L.U[(r1 + 2) % 3], L.U[r1 % 3], L.info, sweeps, rprev, 0, cur ^ 1, tol, 1, r1 + 1, refine);
            // (refine: the flush launch measures max |K| of what it writes, like every launch that ends a sweep -- the closing
            //  kernel's refinement rules read it; round 6: it was 0 here and those rules saw max |K| = 0)
            SX_LAUNCH_CHECK();
            hipLaunchKernelGGL(eigh_close_kernel, dim3(1), dim3(64), 0, st, L.info, sweeps, cur ^ 1, tol, refine,
                               (int *)((char *)L.info + kEighFailsOffset));
            SX_LAUNCH_CHECK();
            if (refine) {
                const dim3 gg((unsigned)(npad / kM2), (unsigned)(npad / kM2));
                const dim3 ge((unsigned)(((int64_t)npad * npad + 255) / 256));
                if (refine & 2) {  // the deep step, twice (each set does nothing unless the record asks for it)
                    int *fails = (int *)((char *)L.info + kEighFailsOffset);
                    for (int step = 0; step < kDeepSteps; ++step) {
                        const int want = 2 + step;
                        hipLaunchKernelGGL(eigh_refine_k_kernel, ge, dim3(256), 0, st, L.M[0], L.M[1], L.M[0], L.M[1], npad, L.info, tol,
                                           want, kDeepCap, 0.5);
                        static const int order[9] = {0, 1, 2, 7, 8, 3, 4, 5, 6};
                        for (int o = 0; o < 9; ++o) {
                            if (kDeepCap <= kDeepNs2 && (order[o] == 7 || order[o] == 8)) continue;  // (never taken: not launched)
                            hipLaunchKernelGGL(eigh_deep_gemm_kernel, gg, dim3(256), 0, st, L.M[0], L.M[1], L.V[0], L.V[1], L.W[0], L.W[1],
                                               L.W[2], npad, L.info, want, order[o]);
                        }
                        hipLaunchKernelGGL(eigh_deep_measure_kernel, dim3(128), dim3(256), 0, st, L.M[0], L.M[1], npad, L.info, tol, want,
                                           step);
                        hipLaunchKernelGGL(eigh_deep_finish_kernel, dim3(1), dim3(64), 0, st, L.info, tol, want, step, fails);
                    }
                    SX_LAUNCH_CHECK();
                }
                hipLaunchKernelGGL(eigh_refine_k_kernel, ge, dim3(256), 0, st, L.M[0],
                                   L.M[1], L.M[0], L.M[1], npad, L.info, tol, 1, kRefineCap, 1.0);
                hipLaunchKernelGGL(eigh_refine_diag_kernel, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, st, L.M[0], L.M[1], npad, L.info);
                hipLaunchKernelGGL(eigh_refine_gemm_kernel, gg, dim3(256), 0, st, L.M[0], L.M[1], L.V[0], L.V[1], npad, L.info, 0);
                hipLaunchKernelGGL(eigh_refine_gemm_kernel, gg, dim3(256), 0, st, L.M[0], L.M[1], L.V[0], L.V[1], npad, L.info, 1);
                SX_LAUNCH_CHECK();
            }
        }
    }
    if (phases & 4) {
        hipLaunchKernelGGL(eigh_colstats_kernel, dim3((unsigned)((npad + 15) / 16)), dim3(1024), 0, st, L.M[0], L.M[1], L.V[0],
                           L.V[1], n, npad, L.info, L.lam, L.scl);
        hipLaunchKernelGGL(eigh_rank_kernel, dim3((unsigned)((n + 4 * kEigRankPerWave - 1) / (4 * kEigRankPerWave))), dim3(256), 0, st,
                           L.lam, n, L.inv, w);
        hipLaunchKernelGGL(eigh_write_kernel, dim3((unsigned)n), dim3(256), 0, st, L.M[0], L.M[1], L.V[0], L.V[1], n, npad, L.info,
                           L.inv, L.scl, B);
        SX_LAUNCH_CHECK();
    }
    return 0;
}
int eigh_rounds_per_sweep(int n) { return n <= kSmallPathMax ? 0 : eigh_npad(n) / kBS - 1; }
int eigh_enqueue(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes,
                 int max_sweeps, double tol, const int *skip, int refine, void *stream) {
    if (max_sweeps <= 0) max_sweeps = 24;
    if (max_sweeps > kEighMaxSweeps) max_sweeps = kEighMaxSweeps;
    const int rps = std::max(1, eigh_rounds_per_sweep(n));
    return eigh_enqueue_phased(C, n, V0, w, B, ws, ws_bytes, tol, skip, refine, stream, 7, 0, max_sweeps * rps);
}
}  // namespace sx

extern "C" int sx_eigh(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes,
                       int max_sweeps, double tol, void *stream) {
    return sx::eigh_enqueue(C, n, V0, w, B, ws, ws_bytes, max_sweeps, tol, nullptr, sx::eigh_refine_default(), stream);
}

// the same with the refinement step decided by the CALLER for this call (refine > 0: allowed, 0: not, < 0: the library's mode):
// nothing process-wide is touched, so concurrent callers (other host threads, the generation loops) are not affected
extern "C" int sx_eigh_refined(const double *C, int n, const double *V0, double *w, double *B, void *ws, int64_t ws_bytes,
                               int max_sweeps, double tol, int refine, void *stream) {
    return sx::eigh_enqueue(C, n, V0, w, B, ws, ws_bytes, max_sweeps, tol, nullptr,
                            refine < 0 ? sx::eigh_refine_default() : (refine ? 1 : 0), stream);
}

extern "C" int sx_eigh_set_refine(int mode) {
    const int prev = sx::g_refine_mode;
    sx::g_refine_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}

// 0: one launch per round (the default); 1: one resident launch for all rounds of a run; -1: SX_EIGH_FLOW or the default.
// Returns the previous mode (-2: changes nothing and returns the mode in effect, 0 / 1).  Processes that share one GPU must use 0 (see eigh_flow_on).
extern "C" int sx_eigh_set_flow(int mode) {
    if (mode == -2) return sx::eigh_flow_on();  // query: the mode in effect (0 / 1), nothing changes
    const int prev = sx::g_flow_mode;
    sx::g_flow_mode = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}

// sweeps carried out / converged flag of the last sx_eigh on this workspace (synchronises the stream)
extern "C" int sx_eigh_info(const void *ws, int *sweeps, int *converged, double *off_rel, void *stream) {
    SX_REQUIRE(ws, "sx_eigh_info: bad arguments");
    EighInfo h;
    SX_HIP(hipMemcpyAsync(&h, ws, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream));
    SX_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (sweeps) *sweeps = h.sweeps;
    if (converged) *converged = h.converged;
    if (off_rel) {  // what the last sweep left behind (no sweep: the off-diagonal mass of the input)
        const double m = h.sweeps > 0 ? h.offm[h.sweeps - 1] : h.acc[0];
        *off_rel = h.norm2 > 0.0 ? sqrt(m / h.norm2) : 0.0;
    }
    return 0;
}
