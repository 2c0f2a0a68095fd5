// letkf_trio.hip -- small ensembles (k <= 20: BASELINE configs[0], the k = 20 workloads): THREE grid points per wavefront.
//
// The register kernel of letkf_wave.hip gives one wave to one point; its row-split Jacobi keeps one column PAIR per slot of two
// lanes, so at k = 20 ten slots -- 20 of 64 lanes -- work through the eigensolve (43 % of the wave time on C2-k20, 35 % on C1)
// and every lane computes the rotation of its pair anew.  Here a wave takes three points at a time (below: which three):
//   front (per point)   the Gram on the matrix cores as in letkf_wave.hip (one 16 x 16 tile of the first 16 members; members
//                       16 .. 19 by broadcast FMAs, the departure columns by plain FMAs), written as the symmetric matrix
//                       A = Ys^T Ys + (k-1)/rho I -- with r = Ys^T d and r_det beside it -- into the point's PARK in LDS
//   eigensolve (once)   the same one-sided Jacobi, odd-even transposition ordering and rotations as jacobi_split, on a line of
//                       slots that holds the three points' column pairs side by side, an unused slot between two points: a DPP
//                       read from a disabled lane returns 0, which is the "no partner" case of a line's end, so the three
//                       problems never meet; every point keeps its own convergence vote and stops rotating when it is done
//                       (its columns go on swapping places: same permutation for all).  30 of 32 slots busy at k = 20.
//   back (per point)    spectra, status, adaptive inflation, the apply phase on the matrix cores straight from the park
//                       (U = V^T B, C = D U, Out = V C), RTPS / RTPP, beta, q clamp, analysis members
// The three points are the same level of three NEIGHBOURING RUNS (a scheduling unit is three runs), walked in step: each point's
// eigenvectors stay in its park until the next point of ITS run is set up there, and warm-start it (G0 = A Q on the matrix
// cores, as letkf_wave.hip's warm_start_product_mfma) -- no workspace in memory, the sweep count of the one-point kernel.
// Serves mode 0 (lists) with nv = 11, no k x k / w-bar outputs, when the streaming pass owns the trivial points
// (letkf_trivial.hip); everything else stays with letkf_wave.hip.  (scale/letkf/letkf_tools.f90:313-527, common_letkf.f90:52-258)
#define LETKF_WAVE_UNIT3
#include "letkf_wave.hip"

namespace letkf {

namespace {

// the wave's LDS slice (doubles).  P points per wave: 3 (the code is written for any P whose segments fit the line of 32 slots; with more
// than three the staging batch shrinks to 192 observations so that the parks fit the 20 KB of a wave -- see launch_trio_kernel).
template <int KR, int P>
struct TrioLds {
  static constexpr int PSZ = (KR + 2) * KR;              // per point: A, later V, [KR columns][KR rows] | r [KR] | r_det [KR]
  static constexpr int kSC = P > 3 ? 192 : 256;          // observations per staging batch
  static constexpr int park = 0;
  static constexpr int lam = park + P * PSZ;             // [P][32] eigenvalue at every line position
  static constexpr int st = lam + P * 32;                // [P][8]  per-point scalars (ST_*)
  static constexpr int work = st + P * 8;                // front: staging [kSC][4]; back: B [BR][16] | spectra [64][2] | Out [32][16]
  static constexpr int wsz = 4 * kSC > 4 * ((KR + 3) / 4) * 16 + 128 + 512 ? 4 * kSC : 4 * ((KR + 3) / 4) * 16 + 128 + 512;
  static constexpr int total = work + wsz;
  static_assert(PSZ % 2 == 0 && work % 2 == 0, "16-byte accesses");
  static_assert(total <= 2560, "two workgroups of four waves per CU");
};
enum { ST_INFL = 0, ST_P1, ST_P2, ST_P3, ST_N, ST_SWEEPS, ST_CONV, ST_BETA };

// profiling twin (make PROF=1): wave time per phase from s_memtime -- 0 set-up + staging, 1 Gram steps, 2 park / warm product,
// 3 eigensolve, 4 spectra + state loads, 5 apply products, 6 analysis members -- summed over all waves into PointArgs::prof
struct TrioProf {
#ifdef LETKF_WAVE_PROF
  unsigned long long t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last = __builtin_amdgcn_s_memtime(), t0 = last;
  __device__ __forceinline__ void mark(const int i) {
    const unsigned long long n = __builtin_amdgcn_s_memtime();
    t[i] += n - last;
    last = n;
  }
#else
  __device__ __forceinline__ void mark(const int) {}
#endif
};

// ---------------------------------------------------------------------------------------------
// The eigensolve of up to three parked matrices at once (see the head of the file; the iteration itself is jacobi_split's,
// letkf_jacobi_dev.h, copy-free form).  valid: bit p = point p is parked.
template <int KR, int P>
__device__ __forceinline__ void jacobi_trio(double* slice, const int k, const unsigned valid, const int max_sweep) {
  using L = TrioLds<KR, P>;
  constexpr int H = KR / 2;
  const int lane = threadIdx.x & 63;
  const int slot = lane & 31, par = lane >> 5;
  const int ncol = (k + 1) & ~1;
  const int S = ncol >> 1, stride = S + 1;
  int seg = 0;
#pragma unroll
  for (int p = 1; p <= P; ++p) seg += slot >= p * stride ? 1 : 0;
  const int sin = slot - seg * stride;
  const bool act = seg < P && sin < S && ((valid >> seg) & 1u);
  // the last slot of the last point is slot 31 when P S + P - 1 = 32 (three points at k = 19, 20): lane 31's right neighbour in the wave-wide shift
  // is lane 32 -- slot 0 of the other row half --, and the other way round; both fetches are zeroed by hand there
  const bool wrap = __builtin_amdgcn_readfirstlane((P - 1) * stride + S) == 32;
  double* base = slice + L::park + (seg < P ? seg : 0) * L::PSZ;
  unsigned long long segm[P];
#pragma unroll
  for (int p = 0; p < P; ++p) segm[p] = __builtin_amdgcn_ballot_w64(act && seg == p);
  int quiet[P], quiet2[P], pdone[P];
  bool sdone[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    quiet[p] = quiet2[p] = pdone[p] = 0;
    sdone[p] = segm[p] == 0ull;
  }
  int pairs = 0;
  if (act) {
    const int colA = 2 * sin, colB = colA + 1;
    double xa[H], xb[H], xf[H];
#pragma unroll
    for (int rr = 0; rr < H; ++rr) {
      xa[rr] = base[colA * KR + 2 * rr + par];     // (an odd k: one zero column somewhere on the line -- column k of a cold start)
      xb[rr] = base[colB * KR + 2 * rr + par];
    }
    const bool hasL = sin > 0, hasR = sin + 1 < S;
    const unsigned long long hasRm = __builtin_amdgcn_ballot_w64(hasR);
    double alA = 0.0, alB = 0.0, isA = 1.0, isB = 1.0, scA = 1.0, scB = 1.0;
    bool live = true;                            // this lane's point is still iterating
    bool alldone = false;
    for (int sweep = 0; sweep < max_sweep && !alldone; ++sweep) {
      double a0 = 0.0, b0 = 0.0;
#pragma unroll
      for (int rr = 0; rr < H; ++rr) {
        xa[rr] *= isA;
        xb[rr] *= isB;
        a0 = fma(xa[rr], xa[rr], a0);
        b0 = fma(xb[rr], xb[rr], b0);
      }
      alA = slot_sum(a0);
      alB = slot_sum(b0);
      isA = isB = scA = scB = 1.0;
      for (int t = 0; t < ncol && !alldone; t += 2) {
        unsigned long long notconv = 0, notconv2 = 0;
        // ---------------- even step: the slot's own two columns
        {
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            if (rr & 1) p1 = fma(xa[rr], xb[rr], p1);
            else p0 = fma(xa[rr], xb[rr], p0);
          }
          const double ga = slot_sum(p0 + p1) * (isA * isB);
          const double a = alA, b = alB;
          const double g2 = ga * ga, ab = a * b;
          notconv |= __builtin_amdgcn_ballot_w64(g2 > kStopTol2W * ab);
          const bool rot = live && g2 > kRotTol2W * ab;
          const double h = 0.5 * (b - a);
          const double x = fma(h, h, g2);
          const double den = fma(x, fast_rsqrt1(x), fabs(h));
          double tt = ga * fast_rcp1(copysign(den, h));
          tt = rot ? tt : 0.0;
          notconv2 |= __builtin_amdgcn_ballot_w64(g2 > kEarlyTol2W * ab) | __builtin_amdgcn_ballot_w64(fabs(tt) > kEarlyTW);
          const double w = fma(tt, tt, 1.0);
          const double c = fast_rsqrt(w);
          const double tg = tt * ga, wc = w * c;
          const double cA = tt * (isA * scB), cB = -tt * (isB * scA);
          const double nisA = isB * c, nscA = scB * wc, nisB = isA * c, nscB = scA * wc;
          alA = b + tg;
          alB = a - tg;
          isA = nisA;
          scA = nscA;
          isB = nisB;
          scB = nscB;
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            asm("v_fma_f64 %0, %1, %2, %3" : "=&v"(xf[rr]) : "v"(cA), "v"(xa[rr]), "v"(xb[rr]));
            xa[rr] = fma(cB, xb[rr], xa[rr]);
          }
        }
        // ---------------- odd step: own B (in xa[]) with the right slot's A; own A (in xf[]) with the left slot's B
        {
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int rr = 0; rr < H; ++rr) xb[rr] = dpp_shift0<0x130>(xf[rr]);   // A of the right slot
          if (wrap) {                                                            // (wave-uniform)
            if (lane == 31) {
#pragma unroll
              for (int rr = 0; rr < H; ++rr) xb[rr] = 0.0;
            }
          }
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            if (rr & 1) p1 = fma(xa[rr], xb[rr], p1);
            else p0 = fma(xa[rr], xb[rr], p0);
          }
          const double alAr = dpp_shift0<0x130>(alA), isAr = dpp_shift0<0x130>(isA), scAr = dpp_shift0<0x130>(scA);
          const double ga = slot_sum(p0 + p1) * (isB * isAr);
          const double a = alB, b = alAr;
          const double g2 = ga * ga, ab = a * b;
          notconv |= __builtin_amdgcn_ballot_w64(g2 > kStopTol2W * ab) & hasRm;
          const bool rot = live && hasR && g2 > kRotTol2W * ab;
          const double h = 0.5 * (b - a);
          const double x = fma(h, h, g2);
          const double den = fma(x, fast_rsqrt1(x), fabs(h));
          double tt = ga * fast_rcp1(copysign(den, h));
          tt = rot ? tt : 0.0;
          notconv2 |= (__builtin_amdgcn_ballot_w64(g2 > kEarlyTol2W * ab) | __builtin_amdgcn_ballot_w64(fabs(tt) > kEarlyTW)) & hasRm;
          const double w = fma(tt, tt, 1.0);
          const double c = fast_rsqrt(w);
          const double tg = tt * ga, wc = w * c;
          const double q1 = dpp_shift0<0x138>(-tt * scB), q2 = dpp_shift0<0x138>(isB * c),
                       q3 = dpp_shift0<0x138>(scB * wc), q4 = dpp_shift0<0x138>(a - tg);
          const double coefR = hasR ? tt * (isB * scAr) : 1.0;
          const double coefL = hasL ? q1 * isA : 1.0;
          if (hasR) {
            alB = b + tg;
            isB = isAr * c;
            scB = scAr * wc;
          }
          if (hasL) {
            alA = q4;
            isA = q2;
            scA = q3;
          }
#pragma unroll
          for (int rr = 0; rr < H; ++rr) {
            xb[rr] = fma(coefR, xa[rr], xb[rr]);
            xa[rr] = dpp_shift0<0x138>(xa[rr]);
          }
          if (wrap) {
            if (lane == 32) {
#pragma unroll
              for (int rr = 0; rr < H; ++rr) xa[rr] = 0.0;
            }
          }
#pragma unroll
          for (int rr = 0; rr < H; ++rr) xa[rr] = fma(coefL, xf[rr], xa[rr]);
        }
        ++pairs;
        // every point has its own vote: S quiet step pairs in a row are one full cycle of the ordering
        alldone = true;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (!sdone[p]) {
            quiet[p] = (notconv & segm[p]) ? 0 : quiet[p] + 1;
            quiet2[p] = (notconv2 & segm[p]) ? 0 : quiet2[p] + 1;
            if (quiet[p] >= S || quiet2[p] >= S) {
              sdone[p] = true;
              pdone[p] = pairs;
              if (seg == p) live = false;
            }
          }
          alldone = alldone && sdone[p];
        }
      }
    }
    // columns lambda_j v_j -> lambda_j, v_j; V back into the park (line position = column number from here on)
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int rr = 0; rr < H; ++rr) {
      xa[rr] *= isA;
      xb[rr] *= isB;
      sa = fma(xa[rr], xa[rr], sa);
      sb = fma(xb[rr], xb[rr], sb);
    }
    sa = slot_sum(sa);
    sb = slot_sum(sb);
    const double lamA = sqrt(sa), lamB = sqrt(sb);
    const double ilA = sa > 0.0 ? 1.0 / lamA : 0.0, ilB = sb > 0.0 ? 1.0 / lamB : 0.0;
#pragma unroll
    for (int rr = 0; rr < H; ++rr) {
      base[colA * KR + 2 * rr + par] = xa[rr] * ilA;
      base[colB * KR + 2 * rr + par] = xb[rr] * ilB;
    }
    if (par == 0) {
      double* lv = slice + L::lam + seg * 32;
      lv[colA] = lamA;
      lv[colB] = lamB;
    }
    // per point: sweeps, converged (the counters live in the ACTIVE lanes: the first lane of the point's segment writes them)
    if (sin == 0 && par == 0) {
      int mypd = 0;
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (seg == p) mypd = sdone[p] ? pdone[p] : 0;
      const int pr = mypd > 0 ? mypd : pairs;
      slice[L::st + seg * 8 + ST_SWEEPS] = (double)((pr + S - 1) / S);
      slice[L::st + seg * 8 + ST_CONV] = mypd > 0 ? 1.0 : 0.0;
    }
  }
  wave_lds_sync();
}

// ---------------------------------------------------------------------------------------------
// Rolling prefetch (r4): a point's front starts with three dependent memory round trips -- list offsets, the list, the departures
// (7 .. 12 % of the wave time, PROF twin).  The offsets of the point after next and the first 256 list entries of the next point are
// requested before the current point's Gram runs; across the eigensolve and the backs of a level they stay in flight for the first
// point of the next level.  Nothing is waited for where it is requested: the values are touched one slot later.
struct TrioHead {
  long pt;                    // < 0: no point
  long o0, o1;                // obs_off[pt], obs_off[pt + 1] as loaded
  double beta;
};
struct TrioList {
  int iob[4];
  double rlv[4], rdv[4];
};
__device__ __forceinline__ TrioHead trio_head(const PointArgs& A, const long pt) {
  TrioHead h;
  h.pt = pt;
  h.o0 = h.o1 = 0;
  h.beta = 1.0;
  if (pt >= 0) {
    h.o0 = A.obs_off[pt];
    h.o1 = A.obs_off[pt + 1];
    if (A.beta) h.beta = A.beta[pt];
  }
  return h;
}
template <int kSC>
__device__ __forceinline__ TrioList trio_list(const PointArgs& A, const TrioHead& h) {
  TrioList l;
  const int lane = threadIdx.x & 63;
  const int n = (int)(h.o1 - h.o0);
  const int ns = (h.pt >= 0 && h.beta != 0.0) ? (n < kSC ? n : kSC) : 0;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int i = ps * 64 + lane;
    l.iob[ps] = 0;
    l.rdv[ps] = 1.0;
    l.rlv[ps] = 0.0;
    if (i < ns) {
      const long e = h.o0 + i;
      l.iob[ps] = A.obs_idx[e];
      l.rlv[ps] = A.rloc_l[e];
      l.rdv[ps] = A.rdiag_l[e];
    }
  }
  return l;
}

// ---------------------------------------------------------------------------------------------
// front: Gram of one point into its park.  Returns 2: parked; 1 / 0: a point the streaming pass has done -- no observations (it
// leaves no eigenvectors behind for the next point of its run) / beta = 0 (it does not touch the run's).  warm: the park holds
// the eigenvectors of the previous point of this run.
template <int KR, int P>
__device__ __forceinline__ int trio_front(const PointArgs& A, const TrioHead& hd, const TrioList& pl, const int sub, double* slice, const int k, const bool warm, TrioProf& pf) {
  using L = TrioLds<KR, P>;
  constexpr int RS = KR - 16;                  // members of the narrow second block (letkf_wave.hip STRIP): 4 or 0
  constexpr int NBLK = RS > 0 ? 2 : 1;
  constexpr int RSA = RS > 0 ? RS : 1;
  const int lane = threadIdx.x & 63;
  const double km1 = (double)(k - 1);
  const long pt = hd.pt;
  const long o0 = hd.o0;
  const int n = __builtin_amdgcn_readfirstlane((int)(hd.o1 - hd.o0));
  const double beta = uniform(hd.beta);
  if (beta == 0.0) return 0;
  if (n == 0) return 1;
  const double* g0 = A.gues + pt * A.sp;
  bool qskip = false;
  if (A.q_update_top > 0.0) qskip = g0[k * A.sm + A.iv_p * A.sv] < A.q_update_top;
  int v0 = 0;
  while (v0 < A.nv && (!((A.var_mask >> v0) & 1u) || (qskip && v0 >= A.iv_q_first && v0 <= A.iv_q_last))) ++v0;
  const double infl_old = (v0 < A.nv) ? A.infl[pt + A.infl_sv * (long)v0] : 1.0;

  v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
  [[maybe_unused]] double accS[RSA][NBLK];
  double accD[NBLK], accDD[NBLK], accP = 0.0, p3 = 0.0;
#pragma unroll
  for (int I = 0; I < NBLK; ++I) {
    accD[I] = accDD[I] = 0.0;
#pragma unroll
    for (int a = 0; a < RSA; ++a) accS[a][I] = 0.0;
  }
  int q = lane >> 4, c16 = lane & 15;
  asm volatile("" : "+v"(q), "+v"(c16));
  constexpr int kSC = L::kSC;
  double* stg = slice + L::work;
  bool rowok[NBLK];
  long mo[NBLK];
#pragma unroll
  for (int I = 0; I < NBLK; ++I) {
    const int m = 16 * I + c16;
    rowok[I] = m < k;
    mo[I] = rowok[I] ? m : 0;
  }
  struct Step {
    double f[NBLK];
    double sw, dsw, ddsw;
  };
  const double* ybase = A.ensval;
  auto run_steps = [&](const int nsp) {
    const int nch = nsp >> 2;
    auto fetch = [&](const int c, Step& t) {
      const bool ok = c < nch;
      const int i = 4 * (ok ? c : 0) + q;
      const double2 a2 = *reinterpret_cast<const double2*>(&stg[4 * i]);
      const double2 b2 = *reinterpret_cast<const double2*>(&stg[4 * i + 2]);
      const long rb = __double_as_longlong(a2.x);
      t.sw = ok ? a2.y : 0.0;
      t.dsw = ok ? b2.x : 0.0;
      t.ddsw = ok ? b2.y : 0.0;
#pragma unroll
      for (int I = 0; I < NBLK; ++I) t.f[I] = ybase[rb + mo[I]];
    };
    auto mma = [&](const Step& t) {
      double y[NBLK];
#pragma unroll
      for (int I = 0; I < NBLK; ++I) {
        double v = t.f[I] * t.sw;
        if (I > 0 || k < 16) v = rowok[I] ? v : 0.0;          // (wave-uniform: the block reaches past the members)
        y[I] = v;
        accD[I] = fma(v, t.dsw, accD[I]);
        accDD[I] = fma(v, t.ddsw, accDD[I]);
      }
      accP = fma(t.dsw, t.dsw, accP);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y[0], y[0], acc, 0, 0, 0);
      if constexpr (RS > 0) {
        auto strip_row = [&](auto a_) {
          constexpr int a = decltype(a_)::value;
          if constexpr (a < RS) {
#pragma unroll
            for (int I = 0; I < NBLK; ++I) {
              if (a == 0 && I == 0) fmac_row_bcast<a, true>(accS[a][I], y[NBLK - 1], y[I]);
              else fmac_row_bcast<a, false>(accS[a][I], y[NBLK - 1], y[I]);
            }
          }
        };
        strip_row(std::integral_constant<int, 0>{});
        strip_row(std::integral_constant<int, 1>{});
        strip_row(std::integral_constant<int, 2>{});
        strip_row(std::integral_constant<int, 3>{});
      }
    };
#ifndef TRIO_GRAM_DEPTH
#define TRIO_GRAM_DEPTH 5
#endif
    constexpr int PD = TRIO_GRAM_DEPTH;                        // steps in flight (letkf_wave.hip run_steps: why it is written so)
    Step ts[PD];
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      fetch(u, ts[u]);
      asm volatile("" ::: "memory");
    }
    auto pin = [&](Step& t) {
#pragma unroll
      for (int I = 0; I < NBLK; ++I) asm volatile("" : "+v"(t.f[I])::"memory");
    };
    for (int c = 0; c < nch; c += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        pin(ts[u]);
        if (u == 0 || c + u < nch) mma(ts[u]);
        fetch(c + PD + u, ts[u]);
      }
    }
  };
  for (int s0 = 0; s0 < n; s0 += kSC) {
    const int ns = min(kSC, n - s0);
    const int nsp = (ns + 3) & ~3;
    wave_lds_sync();
    constexpr int NPASS = kSC / 64;
    int iob[NPASS];
    double rdv[NPASS], rlv[NPASS], dv[NPASS], ddv[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int i = ps * 64 + lane;
      iob[ps] = 0;
      rdv[ps] = 1.0;
      rlv[ps] = 0.0;
      dv[ps] = 0.0;
      ddv[ps] = 0.0;
      if (s0 == 0) {                                          // (the first batch was requested a slot ago: trio_list<kSC>)
        iob[ps] = pl.iob[ps];
        rlv[ps] = pl.rlv[ps];
        rdv[ps] = pl.rdv[ps];
      } else if (i < ns) {
        const long e = o0 + s0 + i;
        iob[ps] = A.obs_idx[e];
        rlv[ps] = A.rloc_l[e];
        rdv[ps] = A.rdiag_l[e];
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int i = ps * 64 + lane;
      if (i < ns) {
        dv[ps] = A.dep[iob[ps]];
        if (A.det_run) ddv[ps] = A.ensval[(long)iob[ps] * A.kld + k];
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int i = ps * 64 + lane;
      if (i < nsp) {
        double sw = 0.0;
        long rb = 0;
        if (i < ns) {
          sw = fast_rsqrt(rdv[ps]);
          rb = (long)iob[ps] * A.kld;
          p3 += rlv[ps];
        }
        *reinterpret_cast<double2*>(&stg[4 * i]) = double2{__longlong_as_double(rb), sw};
        *reinterpret_cast<double2*>(&stg[4 * i + 2]) = double2{dv[ps] * sw, ddv[ps] * sw};
      }
    }
    wave_lds_sync();
    pf.mark(0);
    run_steps(nsp);
    pf.mark(1);
  }
  // the sums over the four observation residues q (the tile is summed by the matrix instruction itself)
#pragma unroll
  for (int I = 0; I < NBLK; ++I) {
    accD[I] += wshfl_xor(accD[I], 16);
    accD[I] += wshfl_xor(accD[I], 32);
    accDD[I] += wshfl_xor(accDD[I], 16);
    accDD[I] += wshfl_xor(accDD[I], 32);
    if constexpr (RS > 0) {
#pragma unroll
      for (int a = 0; a < RS; ++a) {
        accS[a][I] += wshfl_xor(accS[a][I], 16);
        accS[a][I] += wshfl_xor(accS[a][I], 32);
      }
    }
  }
  accP += wshfl_xor(accP, 16);
  accP += wshfl_xor(accP, 32);
  // A (symmetric), column-major -- element (row, col) at [col * KR + row] -- into the staging area (free now); r, r_det into the park
  double* base = slice + L::park + sub * L::PSZ;
  double* ta = stg;
  wave_lds_sync();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) ta[c16 * KR + q + 4 * reg] = acc[reg];        // tile: rows q + 4 reg, column c16
  if (q == 0) {
    if constexpr (RS > 0) {
#pragma unroll
      for (int a = 0; a < RS; ++a) {
        ta[(16 + a) * KR + c16] = accS[a][0];                                      // (row c16, column 16 + a) and its mirror
        ta[c16 * KR + 16 + a] = accS[a][0];
        if (c16 < RS) ta[(16 + a) * KR + 16 + c16] = accS[a][1];                   // (row 16 + c16, column 16 + a)
      }
    }
    base[KR * KR + c16] = accD[0];                                                 // r = Ys^T sqrt(w) d, r_det
    base[(KR + 1) * KR + c16] = accDD[0];
    if constexpr (RS > 0) {
      if (c16 < RS) {
        base[KR * KR + 16 + c16] = accD[1];
        base[(KR + 1) * KR + 16 + c16] = accDD[1];
      }
    }
  }
  wave_lds_sync();
  // diagonal: trace for the adaptive inflation, then the shift (common_letkf.f90:140-143)
  const double shift = km1 / infl_old;
  double diag = 0.0;
  if (lane < k) {
    diag = ta[lane * KR + lane];
    ta[lane * KR + lane] = diag + shift;
  }
  wave_lds_sync();
  if (warm) {
    // G0 = A Q, Q = the eigenvectors the previous point of this run left in the park: tiles (I, J) on the matrix cores, contraction
    // over the members in steps of 4.  (Rows / columns past KR of a tile read whatever follows in LDS -- finite -- and are dropped.)
    constexpr int NT = KR > 16 ? 2 : 1, KS = (KR + 3) / 4;
    v4d d[NT][NT];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = 0; J < NT; ++J) d[I][J] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      double av[NT], bv[NT];
#pragma unroll
      for (int I = 0; I < NT; ++I) {
        av[I] = ta[(4 * s_ + q) * KR + 16 * I + c16];       // A[16 I + c][4 s + q] (A is symmetric)
        bv[I] = base[(16 * I + c16) * KR + 4 * s_ + q];     // Q[4 s + q][16 J + c]
      }
#pragma unroll
      for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) d[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], d[I][J], 0, 0, 0);
    }
    wave_lds_sync();                                         // (all of Q has been read)
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = 16 * I + q + 4 * reg, col = 16 * J + c16;
          if (row < KR && col < KR) base[col * KR + row] = d[I][J][reg];
        }
  } else {
    for (int i = lane; i < KR * KR; i += 64) base[i] = ta[i];
  }
  double parm2 = 0.0, parm3 = 0.0;
  if (A.infl_adaptive) {
    parm3 = wave_sum(p3);
    parm2 = wave_sum(diag) / km1;
  }
  if (lane == 0) {
    double* st = slice + L::st + sub * 8;
    st[ST_INFL] = infl_old;
    st[ST_P1] = accP;
    st[ST_P2] = parm2;
    st[ST_P3] = parm3;
    st[ST_N] = (double)n;
    st[ST_BETA] = beta;
  }
  wave_lds_sync();
  pf.mark(2);
  return 2;
}

// ---------------------------------------------------------------------------------------------
// back: spectra, status, inflation, apply phase on the matrix cores (letkf_wave.hip MAPPLY), analysis members.
template <int KR, int P>
__device__ __forceinline__ int trio_back(const PointArgs& A, const long pt, const int sub, double* slice, const int k, TrioProf& pf) {
  using L = TrioLds<KR, P>;
  constexpr int NV = 11, NB = NV + 2;
  constexpr int KS = (KR + 3) / 4, BR = 4 * KS;
  const int lane = threadIdx.x & 63;
  const double km1 = (double)(k - 1);
  const int ncol = (k + 1) & ~1;
  double* vh = slice + L::park + sub * L::PSZ;     // V [column][KR], r at column KR, r_det at column KR + 1
  const double* st_ = slice + L::st + sub * 8;
  const double infl_old = uniform(st_[ST_INFL]), beta = uniform(st_[ST_BETA]);
  const int n = __builtin_amdgcn_readfirstlane((int)st_[ST_N]);
  const int sweeps = __builtin_amdgcn_readfirstlane((int)st_[ST_SWEEPS]);
  const bool jconv = uniform(st_[ST_CONV]) != 0.0;
  const double lam = lane < ncol ? slice[L::lam + sub * 32 + lane] : 0.0;
  const bool colvalid = lam > 0.0;
  int st = 0;
  {
    const double lmx = wave_max(colvalid ? lam : 0.0);
    const double lmn = wave_min(colvalid ? lam : 1e300);
    if (!jconv && A.max_sweep >= 60) st = 1;
    else if (!(lmx > 0.0)) st = 2;
    else if (lmn < lmx * 1.4901161193847656e-08) st = 3;
  }
  const double sc1 = colvalid ? sqrt(km1 / lam) : 0.0;      // T spectrum
  const double sc2 = colvalid ? 1.0 / lam : 0.0;            // Pa spectrum
  double infl_new = infl_old;
  if (A.infl_adaptive) {                                     // common_letkf.f90:233-254
    const double parm1 = uniform(st_[ST_P1]), parm2 = uniform(st_[ST_P2]), parm3 = uniform(st_[ST_P3]);
    const double parm4 = (parm1 - parm3) / parm2 - infl_old;
    const double tq = (infl_old * parm2 + parm3) / parm2;
    const double sigma_o = 2.0 / parm3 * (tq * tq);
    const double gain = 0.04 * 0.04 / (sigma_o + 0.04 * 0.04);
    infl_new = infl_old + gain * parm4;
  }
  const double* g0 = A.gues + pt * A.sp;
  double* a0 = A.anal + pt * A.sp;
  long moff = (long)lane * A.sm;
  asm volatile("" : "+v"(moff));
  bool qskip = false;
  if (A.q_update_top > 0.0) qskip = g0[k * A.sm + A.iv_p * A.sv] < A.q_update_top;
  double xv[NV];
  double xm_l = 0.0, xd_l = 0.0;
  {
    const double* gp = g0 + moff;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = (lane < k) ? *gp : 0.0;
      gp += A.sv;
    }
    if (lane < NV) {
      xm_l = g0[k * A.sm + lane * A.sv];
      xd_l = A.det_run ? g0[(k + 1) * A.sm + lane * A.sv] : 0.0;
    }
  }
  pf.mark(4);
  double* bm = slice + L::work;                    // [BR][16]
  double* scl = bm + BR * 16;                      // [64][2]
  double* ob = scl + 128;                          // [32][16]
  static_assert(BR * 16 + 128 + 512 <= 1024, "the back's buffers share the staging area");
  int q = lane >> 4, c = lane & 15;
  asm volatile("" : "+v"(q), "+v"(c));
  wave_lds_sync();
  if (lane < BR) {
    double brow[16];
    brow[0] = (lane < k) ? vh[KR * KR + lane] : 0.0;
    brow[1] = (lane < k) ? vh[(KR + 1) * KR + lane] : 0.0;
#pragma unroll
    for (int b = 2; b < 16; ++b) brow[b] = (b - 2 < NV) ? xv[b - 2 < NV ? b - 2 : 0] : 0.0;
    double* row = bm + lane * 16;
#pragma unroll
    for (int b = 0; b < 16; b += 2) *reinterpret_cast<double2*>(&row[b]) = double2{brow[b], brow[b + 1]};
  }
  *reinterpret_cast<double2*>(&scl[2 * lane]) = double2{sc2, sc1};
  wave_lds_sync();
  v4d accO[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) accO[t] = v4d{0.0, 0.0, 0.0, 0.0};
  double va = 0.0, vg = 0.0;
  {
    v4d accU[2];
    accU[0] = v4d{0.0, 0.0, 0.0, 0.0};
    accU[1] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
      const double bq = bm[(4 * s_ + q) * 16 + c];
      vg = fma(bq, bq, vg);
#pragma unroll
      for (int il = 0; il < 2; ++il) {
        if (16 * il < KR) {
          // (columns past KR of the second tile: zero, not whatever follows the park -- a neighbour's matrix; finite or not, it
          // must not reach this point's U, whose rows past k meet zero spectra but NaN * 0 is NaN)
          double a = vh[(16 * il + c) * KR + 4 * s_ + q];
          if (il > 0) a = (16 * il + c < KR) ? a : 0.0;
          accU[il] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, accU[il], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int il = 0; il < 2; ++il) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int s_ = 4 * il + reg;               // contraction step of Out: eigen-columns j = 4 s_ + q
        if (4 * s_ < KR) {
          const int j = 4 * s_ + q;
          const double2 sc = *reinterpret_cast<const double2*>(&scl[2 * j]);
          const double u = accU[il][reg];
          va = fma(u * u, sc.x, va);
          const double cv = u * (c < 2 ? sc.x : sc.y);
          const double* a = vh + j * KR + 4 * c;
          const double2 lo = *reinterpret_cast<const double2*>(a);
          const double2 hi = *reinterpret_cast<const double2*>(a + 2);
          accO[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.x, cv, accO[0], 0, 0, 0);
          accO[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.y, cv, accO[1], 0, 0, 0);
          accO[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.x, cv, accO[2], 0, 0, 0);
          accO[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.y, cv, accO[3], 0, 0, 0);
        }
      }
    }
  }
  // RTPS factor per variable (letkf_tools.f90:1982-1999) in the lanes of column c = 2 + v
  va += wshfl_xor(va, 16);
  va += wshfl_xor(va, 32);
  vg += wshfl_xor(vg, 16);
  vg += wshfl_xor(vg, 32);
  double cf[NV];
  {
    const int v = c - 2;
    const bool isv = c >= 2 && c < NB;
    double cfv = 1.0;
    if (A.relax_alpha != 0.0) {
      cfv = 1.0 - A.relax_alpha;
    } else if (A.relax_alpha_spread != 0.0) {
      const double parm = (A.relax_to_inflated_prior && isv) ? A.infl[pt + A.infl_sv * (long)v] : 1.0;   // :387-391
      if (vg > 0.0 && va > 0.0) cfv = A.relax_alpha_spread * sqrt(vg * parm / (va * km1)) - A.relax_alpha_spread + 1.0;
    }
    if (A.rtps_out && q == 0 && isv && ((A.var_mask >> v) & 1u)) {   // work3da (letkf_tools.f90:460-462)
      const bool skipv = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
      A.rtps_out[pt + A.infl_sv * (long)v] = (A.relax_alpha == 0.0 && A.relax_alpha_spread != 0.0 && !skipv) ? cfv : 1.0;
    }
#pragma unroll
    for (int vv = 0; vv < NV; ++vv) cf[vv] = readlane_d(cfv, 2 + vv);
  }
  // Out tiles -> lane m holds row m: register `reg` of tile I, lane (q, c) is Out[16 reg + 4 q + I][c]
  wave_lds_sync();
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int reg = 0; reg < 2; ++reg) ob[(16 * reg + 4 * q + I) * 16 + c] = accO[I][reg];
  wave_lds_sync();
  double out[NB];
  {
    const double* row = ob + (lane < 32 ? lane : 31) * 16;
#pragma unroll
    for (int b = 0; b < NB; b += 2) {
      const double2 o2 = *reinterpret_cast<const double2*>(&row[b]);
      out[b] = o2.x;
      if (b + 1 < NB) out[b + 1] = o2.y;
    }
  }
  pf.mark(5);
  // ------------------------------------------------------------ analysis members (letkf_tools.f90:472-513)
  {
    double* ap = a0 + moff;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
      const double x = xv[v];
      const double xm = readlane_d(xm_l, v), xdt = readlane_d(xd_l, v);
      const double sdot = wave_sum(x * out[0]);
      const double sdotd = A.det_run ? wave_sum(x * out[1]) : 0.0;
      double val;
      if (skip) {
        val = xm + x;
      } else {
        double cdv = 0.0;
        if (A.relax_alpha != 0.0) {              // RTPP diagonal term alpha*sqrt(parm), parm read before the update
          const double parm = A.relax_to_inflated_prior ? A.infl[pt + A.infl_sv * (long)v] : 1.0;
          cdv = A.relax_alpha * sqrt(parm);
        }
        const double pert = cf[v] * out[2 + v] + cdv * x;
        val = xm + beta * (pert + sdot) + (1.0 - beta) * x;
        if (A.q_sprd_max > 0.0 && v == A.iv_q_first) {      // :500-513
          const double q_mean = wave_sum(lane < k ? val : 0.0) / (double)k;
          const double dq = (lane < k) ? val - q_mean : 0.0;
          const double q_sprd = sqrt(wave_sum(dq * dq) / km1) / q_mean;
          if (q_sprd > A.q_sprd_max) val = q_mean + dq * A.q_sprd_max / q_sprd;
        }
      }
      const bool inclass = (A.var_mask >> v) & 1u;
      if (lane < k && inclass) *ap = val;
      ap += A.sv;
      if (A.det_run && lane == 0 && inclass) a0[(k + 1) * A.sm + v * A.sv] = skip ? xdt : xdt + sdotd * beta;     // :489-497
    }
    if (A.infl_adaptive) {                       // :396-398, after every parm read above
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const bool skip = qskip && v >= A.iv_q_first && v <= A.iv_q_last;
        if (!skip && lane == 0 && ((A.var_mask >> v) & 1u)) A.infl[pt + A.infl_sv * (long)v] = infl_new;
      }
    }
  }
  if (lane == 0) {
    if (A.status) A.status[pt] = st;
    if (A.nsweep) A.nsweep[pt] = sweeps;
    if (A.nobs_out) A.nobs_out[pt] = n;
  }
  wave_lds_sync();
  pf.mark(6);
  return __builtin_amdgcn_readfirstlane(st);
}

template <int KR, int P>
__global__ void __launch_bounds__(256, 2) letkf_trio_kernel(const PointArgs A) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using L = TrioLds<KR, P>;
  constexpr int PPW = 4;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int k = A.k;
  double* slice = smem + (size_t)wv * L::total;
  // finite contents wherever a padded operand of the apply phase reads past what this wave has written
  for (int i = lane; i < L::total; i += 64) slice[i] = 0.0;
  wave_lds_sync();
  const int run_len = A.run_len;
  const long S = A.warm_stride, nA = A.npts / S;
  bool first_draw = true;
  TrioProf pf;
  for (;;) {
    // a unit = plan.ub consecutive runs (P, or a multiple: launch_trio), taken P at a time and walked in step
    const int slot0 = first_draw ? (int)(blockIdx.x >> 3) * PPW + wv : -1;
    const int code = sched_next(A.plan, A.sched, (int)(blockIdx.x & 7), slot0);
    first_draw = false;
    if (code < 0) break;
    int ir0 = 0, ir1 = run_len;
    if (code & 4) {                              // (quarters of a run: only handed out when a unit is one run)
      ir0 = (code & 3) * run_len >> 2;
      ir1 = ((code & 3) + 1) * run_len >> 2;
    }
    const long rid0 = (long)(code >> 3) * A.plan.ub;
    const long left = A.plan.nruns - rid0;
    const int nr = (int)(left < A.plan.ub ? left : A.plan.ub);
    for (int g = 0; g < nr; g += P) {
      const int nsub = nr - g < P ? nr - g : P;
      long rbs[P], ras[P];
#pragma unroll
      for (int sub = 0; sub < P; ++sub) {
        const long rid = rid0 + g + (sub < nsub ? sub : 0);
        const long rchunk = rid / S;
        rbs[sub] = rid - rchunk * S;
        ras[sub] = rchunk * run_len;
      }
      unsigned warm = 0;                         // bit p: park p holds the eigenvectors of the previous point of run p
      // the points of the group in front order: (level, run) = (ir0, 0), (ir0, 1), ..., (ir0 + 1, 0), ...
      auto point_at = [&](const int ir, const int sub) -> long {
        if (ir >= ir1) return -1;
        long ra = ras[0], rb = rbs[0];           // (a chain of scalar selects: indexing the arrays by `sub` would put them in scratch)
#pragma unroll
        for (int p = 1; p < P; ++p) {
          ra = sub == p ? ras[p] : ra;
          rb = sub == p ? rbs[p] : rb;
        }
        return ra + ir < nA ? (ra + ir) * S + rb : -1;
      };
      int ir_a = ir0, sub_a = 0;                 // the slot whose offsets are requested next
      auto next_head = [&]() -> TrioHead {
        const TrioHead h = trio_head(A, point_at(ir_a, sub_a));
        if (++sub_a == nsub) {
          sub_a = 0;
          ++ir_a;
        }
        return h;
      };
      TrioHead h0 = next_head(), h1 = next_head();
      TrioList l0 = trio_list<L::kSC>(A, h0);
      for (int ir = ir0; ir < ir1; ++ir) {
        unsigned valid = 0;
        // (one copy of the front and of the back in the code: the loops over the three points are not unrolled)
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
          const TrioList l1 = trio_list<L::kSC>(A, h1);  // the next point's list, the offsets of the one after it
          const TrioHead h2 = next_head();
          asm volatile("" ::: "memory");
          if (h0.pt >= 0) {
            const int r = trio_front<KR, P>(A, h0, l0, sub, slice, k, (warm >> sub) & 1u, pf);
            if (r == 2) valid |= 1u << sub;
            else if (r == 1) warm &= ~(1u << sub);
          }
          h0 = h1;
          l0 = l1;
          h1 = h2;
        }
        valid = __builtin_amdgcn_readfirstlane(valid);
        if (valid == 0) continue;
        pf.mark(0);
        jacobi_trio<KR, P>(slice, k, valid, A.max_sweep);
        pf.mark(3);
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
          if ((valid >> sub) & 1u) {
            const int st = trio_back<KR, P>(A, point_at(ir, sub), sub, slice, k, pf);
            if (st == 0) warm |= 1u << sub;
            else warm &= ~(1u << sub);
          }
        }
      }
    }
  }
#ifdef LETKF_WAVE_PROF
  if (A.prof && lane == 0) {
    pf.mark(9);
    for (int i = 0; i < 10; ++i) atomicAdd(&A.prof[i], pf.t[i]);
    atomicMax(&A.prof[10], ~pf.t0);
    atomicMax(&A.prof[11], (unsigned long long)__builtin_amdgcn_s_memtime());
  }
#endif
}

template <int KR, int P>
hipError_t launch_trio(const PointArgs& a, int num_cu, hipStream_t st) {
  using L = TrioLds<KR, P>;
  const size_t lds = (size_t)4 * L::total * sizeof(double);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&letkf_trio_kernel<KR, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  static int occ = 0;
  if (occ == 0) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, letkf_trio_kernel<KR, P>, 256, lds);
    occ = (e == hipSuccess && nb > 0) ? nb : 2;
    (void)hipGetLastError();
  }
  if (!a.sched) return hipErrorInvalidValue;
  const long res = (long)occ * num_cu;
#ifndef TRIO_MIN_RUN
#define TRIO_MIN_RUN 4
#endif
#ifndef TRIO_UNITS_PER_WAVE
#define TRIO_UNITS_PER_WAVE 2
#endif
  long rl_ = a.npts / (P * TRIO_UNITS_PER_WAVE * res * 4);
  if (rl_ < TRIO_MIN_RUN) rl_ = TRIO_MIN_RUN;
  if (rl_ > a.run_len) rl_ = a.run_len;
  const long S = a.warm_stride > 1 ? a.warm_stride : 1, rl = rl_ > 1 ? rl_ : 1;
  const long nruns = S * ((a.npts / S + rl - 1) / rl);
  const long nunits = (nruns + P - 1) / P;
  long grid_ = (nunits + 3) / 4;               // a wave-slot per unit, at most what is resident together
  if (grid_ > res) grid_ = res;
  const int grid = (int)(grid_ < 1 ? 1 : grid_);
  PointArgs b = a;
  // a small domain (BASELINE configs[0]: 1600 columns of 30 levels) has fewer units of three whole runs than the GPU has waves:
  // shorter runs then (the first point of a run starts cold), down to 4 points, until there are two units per wave
  {
    const long slots = res * 4;
    long want = a.npts / (P * TRIO_UNITS_PER_WAVE * slots);
    if (want < TRIO_MIN_RUN) want = TRIO_MIN_RUN;
    if (b.run_len > want) b.run_len = (int)want;
  }
  sched_make_plan(b.plan, a.npts, a.warm_stride, b.run_len, grid, 4, 256, P);   // units of P runs
  bool draws = false;
  for (int x = 0; x < 8; ++x) draws = draws || b.plan.whole[x] + 4 * b.plan.f[x] > b.plan.nstat[x];
  if (draws) {
    hipError_t e = hipMemsetAsync(a.sched, 0, 512, st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((letkf_trio_kernel<KR, P>), dim3(grid), dim3(256), lds, st, b);
  return hipGetLastError();
}

}  // namespace

// the calls this kernel serves (launch() of letkf_api.hip asks before it takes the register kernel's route)
bool trio_kernel_supports(const PointArgs& a) {
  return a.mode == 0 && a.nv == 11 && a.k >= 2 && a.k <= 20 && a.skip_trivial && !a.trans_out && !a.pa_out && !a.transm_out && !a.transmd_out &&
         a.gues && a.anal;
}

hipError_t launch_trio_kernel(const PointArgs& a, int num_cu, hipStream_t st) {
  // (P = 5 for k <= 10 -- five segments of <= 5 slots on the line of 32, staging batches of 192 -- was measured, A/B in one call:
  // C2's grid at MEMBER = 10 44.0 -> 43.5 ms, at MEMBER = 3 37.1 -> 38.7 ms: the group waits for the slowest of five, the parks
  // crowd the staging area.  Three everywhere.)
  if (a.k <= 16) return launch_trio<16, 3>(a, num_cu, st);
  return launch_trio<20, 3>(a, num_cu, st);
}
int trio_points_per_wave(int) { return 3; }

}  // namespace letkf
