// Batched on-device simulator front-end: the two per-step host loops that feed the hot path
// (reference: utils/new_simulator.py:279-321, 745-818 and
// dataloader/statetransformer_Guidance.py:88-124, 185-239, guidance 'Project_G').
//   magat_sim_gso         positions -> GSO  S = W / lambda_max(W),  W = (euclidean distance < R), zero diagonal,
//                         optional D^-1/2 W D^-1/2; edge structure bit-exact, lambda_max by Lanczos + Sturm
//                         bisection in float64 (the reference calls numpy.linalg.eigvalsh per instance on the host)
//   magat_sim_fov_states  obstacle map + agent / goal coordinates -> (B,N,3,FOV+2,FOV+2) {0,1} state tensor,
//                         bit-exact (integer work; the goal projection reproduces arctan2 / round-half-even)
// One workgroup per planning instance; everything an instance needs sits in LDS.
#include <cstdint>

#include "magat_common.h"

namespace {

constexpr int SIM_THREADS = 256;

__device__ __forceinline__ double block_sum(double v, double* red, int t, int nt) {
  // wave reduction through DPP-free shuffles (doubles), then one LDS pass over the waves
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (nt + 63) / 64; ++w) s += red[w];
  return s;
}

// ---- lambda_max for graphs of at most 128 agents by ONE wave (round 6): the workgroup form below spends its time in barriers - two
// block sums and two hand-overs per Lanczos step, a Sturm location every eight steps with two more per pass: 311 us for one
// instance of 100 agents, 0.48 ms of a 3.06 ms closed-loop step at 512 x 100.  A wave owns the instance (rows lane and lane + 64):
// no barrier at all, the two sums of a step are DPP row reductions, and the tridiagonal matrix is located ONCE: the walk runs its
// min(N, 160) steps (or to an invariant subspace) - without reorthogonalisation the extreme Ritz value converges first and stays
// put, copies of it appear beside it, never above it - and one Sturm multisection (64 probes per pass) follows.
__device__ __forceinline__ double dpp_mov_f64(double v, int ctrl_sel) {
  unsigned lo = (unsigned)__builtin_bit_cast(unsigned long long, v), hi = (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32);
  switch (ctrl_sel) {      // (the control word must be a compile-time constant)
    case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, true); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xf, 0xf, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xf, 0xf, true); break;
  }
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double lane_bcast_f64(double v, int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
// wave-uniform sum in a fixed order: the rotations inside each row of 16 lanes, then the four rows
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_mov_f64(v, 8);
  v += dpp_mov_f64(v, 4);
  v += dpp_mov_f64(v, 2);
  v += dpp_mov_f64(v, 1);
  return (lane_bcast_f64(v, 0) + lane_bcast_f64(v, 16)) + (lane_bcast_f64(v, 32) + lane_bcast_f64(v, 48));
}
#define SIM_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// rows: bit rows of W [N][words] (LDS); inv: the D^-1/2 scaling (ones without it); u: N + 1 doubles of LDS scratch; nbr: neighbour
// lists, gso_nbr_words(N) words a row; alpha / b2: the tridiagonal matrix (diagonal, squared off-diagonal; LDS).  Called by wave 0
// only (lane = 0..63), N <= 128.
//   matrix-vector product: the bit rows are unpacked ONCE into byte lists padded with the index N (u[N] = 0) to the wave's largest
//   degree, so that a step reads one word of four neighbours and four independent doubles per row - the walk over the set bits
//   (ctz, clear, two dependent reads, an add) was ~120 cycles an edge and as long as the busiest lane's row.
//   Sturm count: the determinant recurrence p_r = (alpha_r - x) p_{r-1} - beta_r^2 p_{r-2} (sign changes = eigenvalues below x)
//   instead of the pivot recurrence: no fp64 division on the chain; rescaled every eight rows.
// The reference's edge test is float64: sqrt(dx^2 + dy^2) < R on integer cell offsets.  The correctly rounded square root is
// monotone, so the test is "squared distance < q" for one integer q per radius: the smallest q whose rounded root is not below R,
// found by stepping from floor(R^2) with the very same float64 expression - exact, and no square root per pair.
__device__ inline long long sim_dist2_bound(double R) {
  if (!(R > 0.0)) return 0;                                  // sqrt(.) >= 0: nothing is closer than R
  if (R >= 3.0e9) return 0x7fffffffffffffffLL;               // int32 coordinates: squared distances stay below 2^65 / 4
  long long q = (long long)floor(R * R);
  while (q > 0 && !(sqrt((double)(q - 1)) < R)) --q;
  while (sqrt((double)q) < R) ++q;
  return q;
}

__host__ __device__ inline int gso_nbr_words(int N) { return ((N + 3) >> 2) | 1; }      // odd: the lanes' rows fall in different banks

__device__ double gso_lambda_wave(const unsigned* rows, int words, const double* inv, double* u, unsigned* nbr, double* alpha,
                                  double* b2, int N, int lane) {
  const int i0 = lane, i1 = lane + 64;
  const bool ok0 = i0 < N, ok1 = i1 < N;
  const int rs = gso_nbr_words(N);
  unsigned char* nb = reinterpret_cast<unsigned char*>(nbr);
  int d0 = 0, d1 = 0;
  if (ok0) {
    for (int w = 0; w < rs; ++w) nbr[i0 * rs + w] = 0x01010101u * (unsigned)N;
    for (int w = 0; w < words; ++w)
      for (unsigned m = rows[i0 * words + w]; m; m &= m - 1) nb[i0 * rs * 4 + d0++] = (unsigned char)(32 * w + __builtin_ctz(m));
  }
  if (ok1) {
    for (int w = 0; w < rs; ++w) nbr[i1 * rs + w] = 0x01010101u * (unsigned)N;
    for (int w = 0; w < words; ++w)
      for (unsigned m = rows[i1 * words + w]; m; m &= m - 1) nb[i1 * rs * 4 + d1++] = (unsigned char)(32 * w + __builtin_ctz(m));
  }
  int dmax = d0 > d1 ? d0 : d1;
  for (int o = 32; o; o >>= 1) { const int other = __shfl_xor(dmax, o, 64); dmax = other > dmax ? other : dmax; }
  const int ngrp = (dmax + 3) >> 2;
  const unsigned* n0 = nbr + (ok0 ? i0 : 0) * rs;
  const unsigned* n1 = nbr + (ok1 ? i1 : 0) * rs;
  const double inv0 = ok0 ? inv[i0] : 0.0, inv1 = ok1 ? inv[i1] : 0.0;
  const double s0 = 1.0 / sqrt((double)N);
  double v0 = ok0 ? s0 : 0.0, v1 = ok1 ? s0 : 0.0, p0 = 0.0, p1 = 0.0, bk = 0.0;
  if (lane == 0) { u[N] = 0.0; b2[0] = 0.0; }
  const int kmax = N < 160 ? N : 160;
  // Sturm count of T_steps at x: eigenvalues below x = sign changes of the leading minors of T - x I
  auto below_count = [&](double x, int steps) -> int {
    int below = 0;
    double pm = 1.0, pmm = 0.0;
    for (int r0 = 0; r0 < steps; r0 += 8) {
      const int r1 = r0 + 8 < steps ? r0 + 8 : steps;
      for (int r = r0; r < r1; ++r) {
        double p = fma(alpha[r] - x, pm, -(b2[r] * pmm));
        if (p == 0.0) p = pm * 0x1p-200;               // a zero minor takes the sign of the one before it (the pivot form's +tiny)
        below += (int)((__double2hiint(p) ^ __double2hiint(pm)) < 0);
        pmm = pm;
        pm = p;
      }
      int ex;
      (void)frexp(pm, &ex);
      pm = ldexp(pm, -ex);
      pmm = ldexp(pmm, -ex);
    }
    return below;
  };
  // multisection inside [lo, hi] (lambda_max of T_steps known to lie there): lane l probes lo + (hi - lo) (l + 1) / 65
  auto locate = [&](double lo, double hi, int steps) -> double {
    for (int pass = 0; pass < 14 && hi - lo > 4e-16 * fmax(fabs(lo), fabs(hi)); ++pass) {
      const double x = lo + (hi - lo) * (double)(lane + 1) / 65.0;
      const unsigned long long ge = __ballot(below_count(x, steps) < steps);      // lambda_max >= x_lane
      const int bt = ge ? 63 - __builtin_clzll(ge) : -1;
      const double nlo = bt >= 0 ? lo + (hi - lo) * (double)(bt + 1) / 65.0 : lo;
      const double nhi = bt + 1 < 64 ? lo + (hi - lo) * (double)(bt + 2) / 65.0 : hi;
      lo = nlo;
      hi = nhi;
    }
    return 0.5 * (lo + hi);
  };
  double glo = 0.0, ghi = 0.0;      // Gershgorin bounds of T_steps, extended row by row (wave-uniform)
  double bprev = 0.0;               // |beta_k|
  double lam = 0.0;
  bool have = false;
  int stable = 0;
  for (int k = 0; k < kmax; ++k) {
    if (ok0) u[i0] = inv0 * v0;
    if (ok1) u[i1] = inv1 * v1;
    SIM_WAVE_SYNC();
    double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
    for (int g = 0; g < ngrp; ++g) {
      const unsigned q0 = n0[g], q1 = n1[g];
      a0 += u[q0 & 255u];
      c0 += u[(q0 >> 8) & 255u];
      a1 += u[q1 & 255u];
      c1 += u[(q1 >> 8) & 255u];
      a0 += u[(q0 >> 16) & 255u];
      c0 += u[q0 >> 24];
      a1 += u[(q1 >> 16) & 255u];
      c1 += u[q1 >> 24];
    }
    SIM_WAVE_SYNC();
    double w0 = inv0 * (a0 + c0) - bk * p0, w1 = inv1 * (a1 + c1) - bk * p1;
    const double ak = wave_sum_f64(v0 * w0 + v1 * w1);
    w0 -= ak * v0;
    w1 -= ak * v1;
    const double bsq = wave_sum_f64(w0 * w0 + w1 * w1);
    // 1 / sqrt by the hardware estimate and two Newton steps (the divide and the square root were a fifth of the step)
    double rb = __builtin_amdgcn_rsq(bsq);
    rb = fma(0.5 * rb, fma(-(bsq * rb), rb, 1.0), rb);
    rb = fma(0.5 * rb, fma(-(bsq * rb), rb, 1.0), rb);
    const double bn = bsq > 0.0 ? bsq * rb : 0.0;
    if (lane == 0) { alpha[k] = ak; b2[k + 1] = bsq; }
    const int steps = k + 1;
    const bool breakdown = !(bn > 1e-13 * (fabs(ak) + bk + 1.0));      // invariant subspace reached: T_steps is exact
    const bool last = breakdown || steps == kmax;
    // bounds of T_steps: row k with its off-diagonals beta_k and (unless this is the last row) beta_{k+1}; a row's bound with the
    // extra beta_{k+1} is also a bound without it
    {
      const double off = bprev + (last ? 0.0 : bn);
      glo = k ? fmin(glo, ak - off) : ak - off;
      ghi = k ? fmax(ghi, ak + off) : ak + off;
    }
    if (last || (steps & 7) == 0) {
      SIM_WAVE_SYNC();      // alpha / b2 are written
      if (!have) {
        lam = locate(glo, ghi, steps);
        have = true;
      } else {
        // lambda_max(T) never decreases as rows are added: one pass with the lanes at lam + tol 2^l tells whether it moved at all,
        // and by how much at most; the multisection then runs inside that bracket only
        // (stop at the first check that finds it still within 1e-12: over random / path / clustered graphs of 30 .. 128 agents that
        //  leaves <= 1e-12 of lambda_max - the reference's float64 eigvals is compared at 1e-9 - for 28 instead of 35 steps on
        //  average against "within 1e-13, twice")
        const double tol = 1e-12 * fabs(lam);
        const double x = lam + ldexp(tol, lane);
        const unsigned long long ge = __ballot(x < ghi && below_count(x, steps) < steps);
        if (!ge) {
          if (++stable >= 1 && !last) break;
        } else {
          const int bt = 63 - __builtin_clzll(ge);
          lam = locate(lam + ldexp(tol, bt), fmin(lam + ldexp(tol, bt + 1), ghi), steps);
          stable = 0;
        }
      }
      if (last) break;
    }
    p0 = v0; p1 = v1;
    v0 = w0 * rb; v1 = w1 * rb;
    bk = bn;
    bprev = bn;
  }
  return lam;
}

// rows of W as bit masks in LDS while they fit (N <= ~1000: 128 KB); larger instances evaluate the distance test on
// the fly (N <= 2048)
template <bool MASK>
__global__ __launch_bounds__(SIM_THREADS) void gso_kernel(const int* __restrict__ pos, double Rscalar,
                                                          const double* __restrict__ radii, int symmetric_norm,
                                                          int normalize, void* __restrict__ S, int s_is_f64,
                                                          double* __restrict__ lambda_out, int N) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const double R = radii ? radii[b] : Rscalar;       // per-instance radius (grown at step 0 by sim_radius_kernel) or one for all
  const int words = (N + 31) / 32;
  int* px = reinterpret_cast<int*>(smem_raw);
  int* py = px + N;
  double* v = reinterpret_cast<double*>(py + N + ((2 * N) & 1));      // 8-byte aligned
  double* y = v + N;
  double* inv = y + N;
  double* red = inv + N;                                             // [16] wave partials, then [160] alpha, [161] beta
  unsigned* rows = reinterpret_cast<unsigned*>(red + 16 + 160 + 162);   // [N][words]   (MASK only)
  __shared__ int any_edge;
  if (t == 0) any_edge = 0;
  for (int n = t; n < N; n += nt) {
    px[n] = pos[((long long)b * N + n) * 2 + 0];
    py[n] = pos[((long long)b * N + n) * 2 + 1];
  }
  __syncthreads();
  const long long d2_bound = sim_dist2_bound(R);     // squareform(pdist(.)) < R in float64, as an integer test
  auto edge = [&](int i, int j) -> bool {
    if (i == j) return false;
    const long long dx = px[i] - px[j], dy = py[i] - py[j];
    return dx * dx + dy * dy < d2_bound;
  };
  // degrees (+ bit rows: one (row, word) item per thread - a thread per ROW left 156 of 256 threads idle at 100 agents)
  if (MASK) {
    for (int idx = t; idx < N * words; idx += nt) {
      const int i = idx / words, w = idx - i * words;
      unsigned m = 0;
      for (int q = 0; q < 32; ++q) {
        const int j = 32 * w + q;
        if (j < N && edge(i, j)) m |= 1u << q;
      }
      rows[idx] = m;
    }
    __syncthreads();
  }
  for (int i = t; i < N; i += nt) {
    int deg = 0;
    if (MASK) {
      for (int w = 0; w < words; ++w) deg += __popc(rows[i * words + w]);
    } else {
      for (int j = 0; j < N; ++j) deg += edge(i, j) ? 1 : 0;
    }
    if (deg) any_edge = 1;
    double s = 1.0;
    if (symmetric_norm) {                       // deg -> 1/sqrt(deg), isolated nodes -> 0 (new_simulator.py:787-793)
      s = deg > 0 ? sqrt(1.0 / (double)deg) : 0.0;
    }
    inv[i] = s;
    v[i] = 1.0;
  }
  __syncthreads();
  const bool has_edges = any_edge != 0;
  double lam = 0.0;
  if (has_edges && normalize && MASK && N <= 128) {
    // small graphs: the whole eigenvalue problem inside wave 0 (gso_lambda_wave), the other waves wait at the barrier
    double* alpha = red + 16;
    double* b2 = alpha + 160;
    double* u = reinterpret_cast<double*>(rows + ((N * words + 1) & ~1));      // [N + 1]
    unsigned* nbr = reinterpret_cast<unsigned*>(u + N + 1);                    // [N][gso_nbr_words(N)]
    if (t < 64) lam = gso_lambda_wave(rows, words, inv, u, nbr, alpha, b2, N, t);
    if (t == 0) red[0] = lam;
    __syncthreads();
    lam = red[0];
  } else if (has_edges && normalize) {
    // Lanczos on the symmetric matrix W (no reorthogonalisation: only the extreme Ritz value is wanted, and that one
    // converges first and stays put).  The all-ones start has a component along the Perron vector of every connected
    // component, so the largest Ritz value tends to lambda_max(W) = max over components.  Every LCHK steps the largest
    // eigenvalue of the tridiagonal T_k is located by Sturm bisection; stop when it no longer moves.
    constexpr int LMAX = 160, LCHK = 8;
    double* alpha = red + 16;            // [LMAX]
    double* beta = alpha + LMAX;         // [LMAX + 1]   beta[k] couples v_{k-1} and v_k
    // v = ones / sqrt(N), v_prev = 0   (y holds v_prev, inv stays the D^-1/2 scaling)
    double* vprev = y;
    double* wv = vprev + 0;              // w is accumulated into registers and written over vprev's slot after use
    const double s0 = 1.0 / sqrt((double)N);
    for (int i = t; i < N; i += nt) { v[i] = s0; vprev[i] = 0.0; }
    if (t == 0) beta[0] = 0.0;
    __syncthreads();
    double prev = -1.0;
    int steps = 0, stable = 0;
    for (int k = 0; k < LMAX; ++k) {
      // w_i = (W v)_i - beta_k * vprev_i ; alpha = v . w
      double wloc[(2048 + SIM_THREADS - 1) / SIM_THREADS];
      double dot = 0.0;
      const double bk = beta[k];
      int q = 0;
      for (int i = t; i < N; i += nt, ++q) {
        double acc = 0.0;
        if (MASK) {
          for (int w = 0; w < words; ++w) {
            unsigned m = rows[i * words + w];
            while (m) {
              const int j = 32 * w + __builtin_ctz(m);
              m &= m - 1;
              acc += inv[j] * v[j];
            }
          }
        } else {
          for (int j = 0; j < N; ++j)
            if (edge(i, j)) acc += inv[j] * v[j];
        }
        const double wi = inv[i] * acc - bk * vprev[i];
        wloc[q] = wi;
        dot += v[i] * wi;
      }
      const double ak = block_sum(dot, red, t, nt);
      double nrm = 0.0;
      q = 0;
      for (int i = t; i < N; i += nt, ++q) {
        wloc[q] -= ak * v[i];
        nrm += wloc[q] * wloc[q];
      }
      const double bn = sqrt(block_sum(nrm, red, t, nt));
      if (t == 0) { alpha[k] = ak; beta[k + 1] = bn; }
      steps = k + 1;
      const bool breakdown = !(bn > 1e-13 * (fabs(ak) + bk + 1.0));      // invariant subspace reached: T_k is exact
      q = 0;
      for (int i = t; i < N; i += nt, ++q) {
        vprev[i] = v[i];
      }
      __syncthreads();
      q = 0;
      if (!breakdown)
        for (int i = t; i < N; i += nt, ++q) v[i] = wloc[q] / bn;
      __syncthreads();
      if (breakdown || (steps % LCHK) == 0 || steps == LMAX || steps >= N) {
        // largest eigenvalue of T_steps (alpha[0..steps), beta[1..steps)) by multisection on the Sturm count: every
        // thread probes its own abscissa inside [lo, hi], the interval shrinks (nt + 1)-fold per pass
        double lo = alpha[0], hi = alpha[0];
        for (int r = 0; r < steps; ++r) {
          const double off = (r > 0 ? fabs(beta[r]) : 0.0) + (r + 1 < steps ? fabs(beta[r + 1]) : 0.0);
          lo = fmin(lo, alpha[r] - off);
          hi = fmax(hi, alpha[r] + off);
        }
        int* best = reinterpret_cast<int*>(red + 15);       // red[0..nt/64) are the reduction slots; [15] is free
        for (int pass = 0; pass < 12 && hi - lo > 4e-16 * fmax(fabs(lo), fabs(hi)); ++pass) {
          if (t == 0) *best = -1;
          __syncthreads();
          const double x = lo + (hi - lo) * (double)(t + 1) / (double)(nt + 1);
          int below = 0;                                     // eigenvalues of T below x = negative pivots of T - x I
          double d = 1.0;
          for (int r = 0; r < steps; ++r) {
            d = (alpha[r] - x) - (r > 0 ? beta[r] * beta[r] / d : 0.0);
            if (d == 0.0) d = 1e-300;
            if (d < 0.0) ++below;
          }
          if (below < steps) atomicMax(best, t);             // lambda_max >= x_t
          __syncthreads();
          const int bt = *best;
          const double nlo = bt >= 0 ? lo + (hi - lo) * (double)(bt + 1) / (double)(nt + 1) : lo;
          const double nhi = bt + 1 < nt ? lo + (hi - lo) * (double)(bt + 2) / (double)(nt + 1) : hi;
          lo = nlo;
          hi = nhi;
          __syncthreads();
        }
        lam = 0.5 * (lo + hi);
        if (breakdown || steps >= N) break;
        if (fabs(lam - prev) <= 1e-13 * fabs(lam)) {
          if (++stable >= 2) break;
        } else {
          stable = 0;
        }
        prev = lam;
      }
    }
    (void)wv;
  }
  if (lambda_out && t == 0) lambda_out[b] = lam;
  const double div = (has_edges && normalize) ? lam : 1.0;
  const long long base = (long long)b * N * N;
  const double plain = 1.0 / div;        // every edge of the plain form holds this one quotient
  for (unsigned idx = t; idx < (unsigned)N * (unsigned)N; idx += nt) {      // N <= 2048
    const int i = (int)(idx / (unsigned)N), j = (int)(idx - (unsigned)i * (unsigned)N);
    bool e;
    if (MASK) e = (rows[i * words + (j >> 5)] >> (j & 31)) & 1u;
    else e = edge(i, j);
    const double val = e ? (symmetric_norm ? (inv[i] * 1.0 * inv[j]) / div : plain) : 0.0;
    if (s_is_f64) static_cast<double*>(S)[base + idx] = val;
    else static_cast<float*>(S)[base + idx] = (float)val;
  }
}

size_t gso_lds_bytes(int N, bool mask) {
  size_t b = (size_t)2 * N * sizeof(int) + 8 + (size_t)3 * N * sizeof(double) + (16 + 160 + 162) * sizeof(double);
  if (mask) b += (size_t)N * ((N + 31) / 32) * sizeof(unsigned);
  if (mask && N <= 128)      // the one-wave eigenvalue path: u [N + 1] doubles and the neighbour lists
    b += sizeof(unsigned) + (size_t)(N + 1) * sizeof(double) + (size_t)N * gso_nbr_words(N) * sizeof(unsigned);
  return b;
}

// ---- FOV state tensors
__global__ __launch_bounds__(SIM_THREADS) void fov_states_kernel(const uint8_t* __restrict__ map, long long map_stride,
                                                                 int H, int Wm, const int* __restrict__ pos,
                                                                 const int* __restrict__ goal, float* __restrict__ x,
                                                                 int fov, int N) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const int Wt = fov + 2, half = fov / 2, dist = Wt / 2;
  unsigned* occ = reinterpret_cast<unsigned*>(smem_raw);             // agent occupancy bitmap [H*Wm bits]
  const int occ_words = (H * Wm + 31) / 32;
  int* gmark = reinterpret_cast<int*>(occ + occ_words);              // per agent: goal marker pixel (row * Wt + col)
  const uint8_t* mp = map + (long long)b * map_stride;
  for (int w = t; w < occ_words; w += nt) occ[w] = 0u;
  __syncthreads();
  for (int n = t; n < N; n += nt) {
    const int cx = pos[((long long)b * N + n) * 2], cy = pos[((long long)b * N + n) * 2 + 1];
    const int gx = goal[((long long)b * N + n) * 2], gy = goal[((long long)b * N + n) * 2 + 1];
    if (cx >= 0 && cx < H && cy >= 0 && cy < Wm) atomicOr(&occ[(cx * Wm + cy) >> 5], 1u << ((cx * Wm + cy) & 31));
    int row, col;
    const int dx = gx - cx, dy = gy - cy;
    const bool in_map = gx >= 0 && gx < H && gy >= 0 && gy < Wm;
    if (in_map && dx >= -half && dx <= half && dy >= -half && dy <= half) {     // goal inside the FOV window
      row = dx + half + 1;
      col = dy + half + 1;
    } else {
      // projectedgoal (statetransformer_Guidance.py:103-124): the arctan2 test "pi/4 <= |angle| <= 3pi/4" is
      // |dy| >= |dx| for integer offsets (on the exact diagonals both branches give the same pixel); np.round is
      // round-half-to-even = rint in the default rounding mode
      const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
      const int sx = (dx > 0) - (dx < 0), sy = (dy > 0) - (dy < 0);
      if (ady >= adx) {
        col = dist * (sy + 1);
        row = (int)((double)dist + rint((double)dist * (double)dx / (double)ady));
      } else {
        row = dist * (sx + 1);
        col = (int)((double)dist + rint((double)dist * (double)dy / (double)adx));
      }
    }
    gmark[n] = row * Wt + col;
  }
  __syncthreads();
  const int per_agent = 3 * Wt * Wt;
  float* xb = x + (long long)b * N * per_agent;
  // a small batch spreads an instance's tensor over gridDim.y workgroups (each builds the same bitmap; one instance of 100
  // agents in ONE workgroup took 59 us)
  const int total = N * per_agent, chunk = (total + (int)gridDim.y - 1) / (int)gridDim.y;
  const int begin = (int)blockIdx.y * chunk, end = begin + chunk < total ? begin + chunk : total;
  for (int idx = begin + t; idx < end; idx += nt) {
    const int n = idx / per_agent, r = idx - n * per_agent;
    const int ch = r / (Wt * Wt), pix = r - ch * Wt * Wt;
    const int a = pix / Wt, c = pix - a * Wt;
    float val = 0.f;
    if (ch == 1) {
      val = gmark[n] == pix ? 1.f : 0.f;
    } else if (a >= 1 && a <= fov && c >= 1 && c <= fov) {
      const int gx = pos[((long long)b * N + n) * 2] - half + (a - 1);
      const int gy = pos[((long long)b * N + n) * 2 + 1] - half + (c - 1);
      const bool inside = gx >= 0 && gx < H && gy >= 0 && gy < Wm;
      if (ch == 0) val = inside ? (mp[gx * Wm + gy] ? 1.f : 0.f) : 1.f;          // outside the map = obstacle
      else val = inside && ((occ[(gx * Wm + gy) >> 5] >> ((gx * Wm + gy) & 31)) & 1u) ? 1.f : 0.f;
    }
    xb[idx] = val;
  }
}


// ---- step-0 communication radius (multiRobotSimNew.computeAdjacencyMatrix, step == 0 branch, new_simulator.py:759-768):
// r = R0 / 1.1;  do { r = r * 1.1;  W = (distance < r) } while (!isConnected(W)).  The reference tests connectivity through
// the Laplacian's spectrum (graphTools.isConnected: exactly one eigenvalue below 1e-9); here it is a reachability sweep
// from agent 0 over the same float64 distance test - the same predicate, evaluated exactly.  One workgroup per instance.
__global__ __launch_bounds__(SIM_THREADS) void sim_radius_kernel(const int* __restrict__ pos, double R0,
                                                                 double* __restrict__ radii_out, int* __restrict__ steps_out,
                                                                 int N, int max_steps) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  int* px = reinterpret_cast<int*>(smem_raw);
  int* py = px + N;
  int* seen = py + N;                  // 0 / 1 per agent
  __shared__ int changed, count;
  for (int n = t; n < N; n += nt) {
    px[n] = pos[((long long)b * N + n) * 2 + 0];
    py[n] = pos[((long long)b * N + n) * 2 + 1];
  }
  double r = R0 / 1.1;
  int steps = 0;
  bool connected = false;
  while (!connected && steps < max_steps) {
    r = r * 1.1;
    ++steps;
    const long long d2_bound = sim_dist2_bound(r);
    __syncthreads();
    for (int n = t; n < N; n += nt) seen[n] = n == 0 ? 1 : 0;
    while (true) {
      __syncthreads();
      if (t == 0) changed = 0;
      __syncthreads();
      for (int i = t; i < N; i += nt) {
        if (seen[i]) continue;
        bool hit = false;
        for (int j = 0; j < N && !hit; ++j) {
          if (!seen[j] || j == i) continue;
          const long long dx = px[i] - px[j], dy = py[i] - py[j];
          hit = dx * dx + dy * dy < d2_bound;
        }
        if (hit) { seen[i] = 1; changed = 1; }
      }
      __syncthreads();
      if (!changed) break;
    }
    if (t == 0) count = 0;
    __syncthreads();
    int c = 0;
    for (int n = t; n < N; n += nt) c += seen[n];
    if (c) atomicAdd(&count, c);
    __syncthreads();
    connected = count == N;
  }
  if (t == 0) {
    radii_out[b] = r;
    if (steps_out) steps_out[b] = connected ? steps : -steps;     // negative: still disconnected after max_steps
  }
}

// ---- action decode + collision shielding + position update (new_simulator.py:334-454, 471-520).
// One workgroup per instance; a single int32 cell grid in LDS is reused for the three lookups the reference does with
// Python dicts: occupant of a cell (swap test), claimants of a target cell (atomicMin of a priority key), and the
// "forced to stay" cells of the backward cascade.
struct SimBook {
  uint8_t* reach;          // [B][N] sticky reach-goal flags (null: no bookkeeping)
  int* first_move;         // [B][N]
  int* end_step;           // [B][N]
  int step, maxstep;
  int* done_out;           // [B] all agents had reached their goals BEFORE this call
  int* flowtime_out;       // [B] written when the episode is over
  int* makespan_out;       // [B]
};

__global__ __launch_bounds__(1024) void sim_move_kernel(const float* __restrict__ logits, const int* __restrict__ actions_in,
                                                        const uint8_t* __restrict__ map, long long map_stride, int H, int Wm,
                                                        int* __restrict__ pos, const int* __restrict__ goal,
                                                        int* __restrict__ actions_out, signed char* __restrict__ moves_out,
                                                        uint8_t* __restrict__ reached_out, int* __restrict__ flags_out,
                                                        int N, int policy, const double* __restrict__ uniforms,
                                                        SimBook bk) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const int cells = H * Wm;
  // episode bookkeeping of multiRobotSimNew.move (new_simulator.py:471-549), when the caller keeps its state on the device
  __shared__ int n_reached;
  bool active = true, finalize = false;
  if (bk.reach) {
    if (t == 0) n_reached = 0;
    __syncthreads();
    int c = 0;
    for (int n = t; n < N; n += nt) c += bk.reach[(long long)b * N + n] ? 1 : 0;
    if (c) atomicAdd(&n_reached, c);
    __syncthreads();
    const bool all_reached = n_reached == N;                 // evaluated BEFORE this step's move, like the reference
    active = !all_reached && bk.step < bk.maxstep;
    finalize = all_reached || bk.step >= bk.maxstep;
    if (t == 0 && bk.done_out) bk.done_out[b] = all_reached ? 1 : 0;
    __syncthreads();
  }
  if (!active) {
    for (int n = t; n < N; n += nt) {
      const long long a = (long long)b * N + n;
      if (actions_out) actions_out[a] = 4;
      if (moves_out) { moves_out[2 * a] = 0; moves_out[2 * a + 1] = 0; }
      if (reached_out && goal) reached_out[a] = (pos[2 * a] == goal[2 * a] && pos[2 * a + 1] == goal[2 * a + 1]) ? 1 : 0;
    }
    if (flags_out && t == 0) flags_out[b] = 0;
  }
  if (active) {
  unsigned* grid = reinterpret_cast<unsigned*>(smem_raw);            // [cells]
  int* px = reinterpret_cast<int*>(grid + cells);                    // [N]
  int* py = px + N;
  int* mc = py + N;                                                  // move code 0..4 (4 = stop)
  int* aux = mc + N;                                                 // per-agent scratch: swap flag, then forced flag
  __shared__ int changed, flags;
  const uint8_t* mp = map + (long long)b * map_stride;
  const int DX[5] = {-1, 0, 1, 0, 0}, DY[5] = {0, -1, 0, 1, 0};     // up, left, down, right, stop (:56-65)
  if (t == 0) flags = 0;
  for (int c = t; c < cells; c += nt) grid[c] = 0u;
  __syncthreads();
  for (int n = t; n < N; n += nt) {
    const long long a = (long long)b * N + n;
    const int x = pos[2 * a], y = pos[2 * a + 1];
    int key;
    if (logits) {
      const float* l = logits + a * 5;
      key = 0;                          // convectToActionKey_softmax (:863-869): argmax, first maximum wins
      float best = l[0];
      for (int q = 1; q < 5; ++q)
        if (l[q] > best) { best = l[q]; key = q; }
      if (policy != 0 && uniforms) {
        // convectToActionKey_{sum,exp}_multinorm (:871-883): one draw from the categorical distribution with weights
        // x / sum(x) (sum) or exp(x) (exp), float32.  torch.multinomial's generator cannot be replayed on the device, so the draw
        // is defined by a caller-supplied uniform u in [0, 1): inverse CDF over the float32 weights in index order,
        // accumulated in float64 - the first k with w_0 + .. + w_k > u * sum.  (tests: the reference run with
        // torch.multinomial patched to this rule and the same u gives the same keys.)
        double w[5], tot = 0.0;
        bool bad = false;
        const float lsum = (((l[0] + l[1]) + l[2]) + l[3]) + l[4];   // sum_multinorm draws from normalize(x) = x / sum(x) (:857-861)
        for (int q = 0; q < 5; ++q) {
          const float wf = policy == 2 ? (float)exp((double)l[q]) : l[q] / lsum;
          bad |= !(wf >= 0.f) || wf == __builtin_inff();
          w[q] = (double)wf;
          tot += w[q];
        }
        if (bad || !(tot > 0.0)) {
          atomicOr(&flags, 32);         // invalid distribution (torch.multinomial raises): the greedy key stands
        } else {
          const double thr = uniforms[a] * tot;
          double c = 0.0;
          int pick = -1, lastpos = 0;
          for (int q = 0; q < 5; ++q) {
            c += w[q];
            if (w[q] > 0.0) lastpos = q;
            if (pick < 0 && c > thr) pick = q;
          }
          key = pick < 0 ? lastpos : pick;
        }
      }
    } else {
      key = actions_in[a];
      if (key < 0 || key > 4) key = 4;
    }
    if (actions_out) actions_out[a] = key;
    // first step at which the agent proposes a move (:489-490; the reference's own "== 0" test cannot tell "never" from
    // "at step 0" - reproduced as is)
    if (bk.first_move && key != 4 && bk.first_move[a] == 0) bk.first_move[a] = bk.step;
    if (x < 0 || y < 0 || x >= H || y >= Wm) {                       // a position outside the map: flagged, the agent is left alone
      atomicOr(&flags, 16);
      px[n] = -1; py[n] = 0; mc[n] = 4; aux[n] = 0;                   // px < 0: takes no part in the cell claims
      continue;
    }
    const int nx = x + DX[key], ny = y + DY[key];
    if (nx < 0 || ny < 0 || nx >= H || ny >= Wm) {                   // out of the arena -> stop (:354-357)
      key = 4;
      atomicOr(&flags, 1);
    }
    px[n] = x; py[n] = y; mc[n] = key; aux[n] = 0;
    grid[x * Wm + y] = (unsigned)(n + 1);
  }
  __syncthreads();
  // face-to-face swap: the agent in my target cell moves into my cell -> both stop (:361-375)
  for (int n = t; n < N; n += nt) {
    const int k = mc[n];
    if (k != 4) {
      const int j = (int)grid[(px[n] + DX[k]) * Wm + py[n] + DY[k]] - 1;
      if (j >= 0) {
        const int kj = mc[j];
        if (kj != 4 && DX[kj] == -DX[k] && DY[kj] == -DY[k]) aux[n] = 1;
      }
    }
  }
  __syncthreads();
  for (int n = t; n < N; n += nt) {
    int k = mc[n];
    int forced = 0;
    if (aux[n]) { k = 4; atomicOr(&flags, 2); }
    if (k != 4 && mp[(px[n] + DX[k]) * Wm + py[n] + DY[k]] != 0) {   // into an obstacle -> stop, cell stays taken (:392-403)
      k = 4;
      forced = 1;
      atomicOr(&flags, 4);
    }
    mc[n] = k;
    aux[n] = forced;
  }
  __syncthreads();
  for (int c = t; c < cells; c += nt) grid[c] = 0xffffffffu;
  __syncthreads();
  // claimants of every target cell: a stationary claimant wins, otherwise the lowest agent index (the reference draws
  // random.choice here, :416 - the one documented deviation)
  for (int n = t; n < N; n += nt) {
    const int k = mc[n];
    if (px[n] < 0) continue;
    atomicMin(&grid[(px[n] + DX[k]) * Wm + py[n] + DY[k]], (unsigned)((k == 4 ? 0 : 1) << 16 | n));
  }
  __syncthreads();
  for (int n = t; n < N; n += nt) {
    const int k = mc[n];
    if (k != 4 && (int)(grid[(px[n] + DX[k]) * Wm + py[n] + DY[k]] & 0xffffu) != n) {
      mc[n] = 4;
      aux[n] = 1;
      atomicOr(&flags, 8);
    }
  }
  __syncthreads();
  for (int c = t; c < cells; c += nt) grid[c] = 0u;
  __syncthreads();
  for (int n = t; n < N; n += nt)
    if (aux[n]) grid[px[n] * Wm + py[n]] = 1u;                        // cells whose occupant was forced to stay
  // backward cascade (:424-446): whoever moves into such a cell stays as well, and its own cell joins the set
  while (true) {
    __syncthreads();
    if (t == 0) changed = 0;
    __syncthreads();
    for (int n = t; n < N; n += nt) {
      const int k = mc[n];
      if (k != 4 && grid[(px[n] + DX[k]) * Wm + py[n] + DY[k]] != 0u) {
        mc[n] = 4;
        grid[px[n] * Wm + py[n]] = 1u;
        changed = 1;
      }
    }
    __syncthreads();
    if (!changed) break;
  }
  for (int n = t; n < N; n += nt) {
    const long long a = (long long)b * N + n;
    const int k = mc[n];
    const bool inside = pos[2 * a] >= 0 && pos[2 * a + 1] >= 0 && pos[2 * a] < H && pos[2 * a + 1] < Wm;
    const int nx = inside ? px[n] + DX[k] : pos[2 * a], ny = inside ? py[n] + DY[k] : pos[2 * a + 1];
    pos[2 * a] = nx;
    pos[2 * a + 1] = ny;
    if (moves_out) { moves_out[2 * a] = (signed char)(inside ? DX[k] : 0); moves_out[2 * a + 1] = (signed char)(inside ? DY[k] : 0); }
    const bool at_goal = goal && nx == goal[2 * a] && ny == goal[2 * a + 1];
    if (reached_out && goal) reached_out[a] = at_goal ? 1 : 0;
    if (bk.reach && at_goal) {                                      // sticky reach flag, first arrival step (:521-526)
      bk.reach[a] = 1;
      if (bk.end_step && bk.end_step[a] == 0) bk.end_step[a] = bk.step;
    }
  }
  if (flags_out && t == 0) flags_out[b] = flags;
  }
  if (finalize && bk.end_step && bk.first_move) {
    // episode over (everybody arrived before this call, or the step budget is spent): :528-547
    __shared__ int flow, emax, fmin;
    __syncthreads();
    if (t == 0) { flow = 0; emax = -2147483647; fmin = 2147483647; }
    __syncthreads();
    for (int n = t; n < N; n += nt) {
      const long long a = (long long)b * N + n;
      int e = bk.end_step[a];
      if (e == 0) { e = bk.step - 1; bk.end_step[a] = e; }
      const int f = bk.first_move[a];
      atomicAdd(&flow, e - f + 1);
      atomicMax(&emax, e);
      atomicMin(&fmin, f);
    }
    __syncthreads();
    if (t == 0) {
      if (bk.flowtime_out) bk.flowtime_out[b] = flow;
      if (bk.makespan_out) bk.makespan_out[b] = emax - fmin + 1;
    }
  }
}

}  // namespace

namespace {
int sim_gso_launch(const int32_t* pos, double comm_radius, const double* radii, int symmetric_norm, int normalize, void* S,
                   int s_is_f64, double* lambda_out, int B, int N, void* stream) {
  if (!pos || !S) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || (!radii && !(comm_radius > 0.0))) return MAGAT_ERR_BAD_SHAPE;
  if (N > 2048) return MAGAT_ERR_UNSUPPORTED;
  const bool mask = gso_lds_bytes(N, true) <= 160 * 1024;          // bit rows of W in LDS (N <= ~1000), else on the fly
  const size_t lds = gso_lds_bytes(N, mask);
  if (lds > 160 * 1024) return MAGAT_ERR_UNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (lds > 64 * 1024 &&
      magat_ensure_dyn_lds(mask ? reinterpret_cast<const void*>(&gso_kernel<true>)
                                : reinterpret_cast<const void*>(&gso_kernel<false>),
                           mask ? MAGAT_LDS_SIM_GSO_T : MAGAT_LDS_SIM_GSO_F, lds) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  if (mask) {
    hipLaunchKernelGGL(gso_kernel<true>, dim3(B), dim3(SIM_THREADS), lds, st, pos, comm_radius, radii, symmetric_norm,
                       normalize, S, s_is_f64, lambda_out, N);
  } else {
    hipLaunchKernelGGL(gso_kernel<false>, dim3(B), dim3(SIM_THREADS), lds, st, pos, comm_radius, radii, symmetric_norm,
                       normalize, S, s_is_f64, lambda_out, N);
  }
  return magat_check_launch();
}
}  // namespace

extern "C" int magat_sim_gso(const int32_t* pos, double comm_radius, int symmetric_norm, int normalize, void* S,
                             int s_is_f64, double* lambda_out, int B, int N, void* stream) {
  return sim_gso_launch(pos, comm_radius, nullptr, symmetric_norm, normalize, S, s_is_f64, lambda_out, B, N, stream);
}

extern "C" int magat_sim_gso_radii(const int32_t* pos, const double* radii, int symmetric_norm, int normalize, void* S,
                                   int s_is_f64, double* lambda_out, int B, int N, void* stream) {
  if (!radii) return MAGAT_ERR_NULL;
  return sim_gso_launch(pos, 0.0, radii, symmetric_norm, normalize, S, s_is_f64, lambda_out, B, N, stream);
}

extern "C" int magat_sim_connect_radius(const int32_t* pos, double comm_radius, double* radii_out, int32_t* steps_out,
                                        int B, int N, int max_steps, void* stream) {
  if (!pos || !radii_out) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || !(comm_radius > 0.0) || max_steps <= 0) return MAGAT_ERR_BAD_SHAPE;
  const size_t lds = (size_t)3 * N * sizeof(int);
  if (lds > 64 * 1024) return MAGAT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sim_radius_kernel, dim3(B), dim3(SIM_THREADS), lds, static_cast<hipStream_t>(stream), pos, comm_radius,
                     radii_out, steps_out, N, max_steps);
  return magat_check_launch();
}

extern "C" int magat_sim_fov_states(const uint8_t* map, int map_batched, int H, int W, const int32_t* pos,
                                    const int32_t* goal, float* x, int FOV, int B, int N, void* stream) {
  if (!map || !pos || !goal || !x) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || H <= 0 || W <= 0 || FOV <= 0 || !(FOV & 1)) return MAGAT_ERR_BAD_SHAPE;
  const size_t lds = (size_t)((H * W + 31) / 32) * sizeof(unsigned) + (size_t)N * sizeof(int);
  if (lds > 64 * 1024) return MAGAT_ERR_UNSUPPORTED;
  int split = 1;                                           // workgroups per instance: ~512 in flight, >= 8 agents' worth each
  while (split < 32 && (long long)B * split < 512 && N / (2 * split) >= 4) split *= 2;
  hipLaunchKernelGGL(fov_states_kernel, dim3(B, split), dim3(SIM_THREADS), lds, static_cast<hipStream_t>(stream), map,
                     map_batched ? (long long)H * W : 0LL, H, W, pos, goal, x, FOV, N);
  return magat_check_launch();
}

namespace {
int sim_move_launch(const float* logits, const int32_t* actions_in, const uint8_t* map, int map_batched, int H, int W,
                    int32_t* pos, const int32_t* goal, int32_t* actions_out, int8_t* moves_out, uint8_t* reached_out,
                    int32_t* flags_out, int B, int N, int policy, const double* uniforms, const SimBook& bk, void* stream) {
  if ((!logits && !actions_in) || !map || !pos) return MAGAT_ERR_NULL;
  if (B <= 0 || N <= 0 || N > 65535 || H <= 0 || W <= 0) return MAGAT_ERR_BAD_SHAPE;
  if (policy < 0 || policy > 2) return MAGAT_ERR_UNSUPPORTED;
  if (policy != 0 && (!logits || !uniforms)) return MAGAT_ERR_NULL;
  const size_t lds = (size_t)H * W * sizeof(unsigned) + (size_t)4 * N * sizeof(int);
  if (lds > 160 * 1024) return MAGAT_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      magat_ensure_dyn_lds(reinterpret_cast<const void*>(&sim_move_kernel), MAGAT_LDS_SIM_MOVE, lds) != MAGAT_OK)
    return MAGAT_ERR_LAUNCH;
  int threads = 64;
  while (threads < N && threads < 1024) threads *= 2;
  hipLaunchKernelGGL(sim_move_kernel, dim3(B), dim3(threads), lds, static_cast<hipStream_t>(stream), logits, actions_in,
                     map, map_batched ? (long long)H * W : 0LL, H, W, pos, goal, actions_out,
                     reinterpret_cast<signed char*>(moves_out), reached_out, flags_out, N, policy, uniforms, bk);
  return magat_check_launch();
}
}  // namespace

extern "C" int magat_sim_move(const float* logits, const int32_t* actions_in, const uint8_t* map, int map_batched, int H,
                              int W, int32_t* pos, const int32_t* goal, int32_t* actions_out, int8_t* moves_out,
                              uint8_t* reached_out, int32_t* flags_out, int B, int N, void* stream) {
  return sim_move_launch(logits, actions_in, map, map_batched, H, W, pos, goal, actions_out, moves_out, reached_out,
                         flags_out, B, N, 0, nullptr, SimBook{}, stream);
}

extern "C" int magat_sim_step(const magat_sim_step_desc* d, void* stream) {
  if (!d) return MAGAT_ERR_NULL;
  if (!d->goal || !d->reach_goal || !d->first_move || !d->end_step) return MAGAT_ERR_NULL;
  SimBook bk{};
  bk.reach = d->reach_goal;
  bk.first_move = d->first_move;
  bk.end_step = d->end_step;
  bk.step = d->currentstep;
  bk.maxstep = d->maxstep;
  bk.done_out = d->done_out;
  bk.flowtime_out = d->flowtime_out;
  bk.makespan_out = d->makespan_out;
  return sim_move_launch(d->logits, d->actions_in, d->map, d->map_batched, d->H, d->W, d->pos, d->goal, d->actions_out,
                         d->moves_out, nullptr, d->flags_out, d->B, d->N, d->policy, d->uniforms, bk, stream);
}
