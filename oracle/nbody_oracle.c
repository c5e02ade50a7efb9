/*
 * nbody_oracle.c -- CPU restatement of the reference's all-pairs force + kick/drift hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker (or as the timed CPU baseline), never as the thing shipped.  The product path
 * (libnbody_hip.so) never links or calls into this file.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (seq, omp_2, kick, drift,
 * generator) or to fp64 re-association noise (omp_1, whose result depends on the OpenMP thread
 * partition) against the reference's own object code built by oracle/build_ref.sh into
 * oracle/_ref/libnbody_ref.so (tests/test_oracle_vs_ref.py, runs where /root/reference exists),
 * and against the committed golden vectors under tests/golden/ that were produced by that
 * reference build (tests/golden/make_golden.py).
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference/).
 * Plain C11, fp64 IEEE arithmetic, compiled with -ffp-contract=off so that no multiply-add is
 * fused (the reference is built by g++ -O3 for baseline x86-64, which has no FMA contraction).
 *
 * Body<D> memory layout (nbody-sim-new/body.h:8-11, vector.h:9-12):
 *   double position[D]; double velocity[D]; double mass;     => (2*D+1) doubles per body
 * Vector<D> = double[D].
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* nbody-sim-new/utils.h:21 */
#define ORACLE_G 4.471e-21
/* nbody-sim-new/methods.cpp:24 (hard skip, no softening) */
#define ORACLE_R2_SKIP 1e-10
/* nbody-sim-new/utils.h:25-26 */
#define ORACLE_ACCURACY_PCT_THRESHOLD 0.01
#define ORACLE_ACCURACY_FORCE_THRESHOLD 1e-20

#define BODY_STRIDE(D) (2 * (D) + 1)
#define POS(b, i, D) ((b) + (size_t)(i) * BODY_STRIDE(D))
#define VEL(b, i, D) ((b) + (size_t)(i) * BODY_STRIDE(D) + (D))
#define MASS(b, i, D) ((b)[(size_t)(i) * BODY_STRIDE(D) + 2 * (D)])

double oracle_G(void) { return ORACLE_G; }

/*
 * One pair, exactly the statement sequence of nbody-sim-new/methods.cpp:21-33 with the Vector<D>
 * operators of vector.h expanded:
 *   diff = pj - pi                         vector.h:32-36
 *   dist_sq = 0.0 + d0*d0 + d1*d1 (+ d2*d2) vector.h:81-85 (left-to-right from 0.0)
 *   if (dist_sq < 1e-10) continue          methods.cpp:24
 *   dist = sqrt(dist_sq); dist_cb = dist_sq*dist        methods.cpp:26-27
 *   force_mag = ((G*mi)*mj)/dist_cb        methods.cpp:30
 *   force = (diff / mag) * force_mag       vector.h:93-97 (normalized(): mag recomputed as
 *                                          sqrt(magnitude_squared()); zero vector if mag<1e-10),
 *                                          vector.h:46-50 (component-wise divide), :39-43 (scale)
 * Returns 0 if the pair is skipped, 1 otherwise (f[] filled).
 */
static inline int pair_force(const double* pi, const double* pj, double mi, double mj, int D, double* f) {
    double diff[3];
    double dist_sq = 0.0;
    for (int k = 0; k < D; ++k) diff[k] = pj[k] - pi[k];
    for (int k = 0; k < D; ++k) dist_sq += diff[k] * diff[k];
    if (dist_sq < ORACLE_R2_SKIP) return 0;
    double dist = sqrt(dist_sq);
    double dist_cb = dist_sq * dist;
    double force_mag = ORACLE_G * mi * mj / dist_cb;
    double mag = sqrt(dist_sq);
    if (mag < 1e-10) {
        for (int k = 0; k < D; ++k) f[k] = 0.0 * force_mag;
        return 1;
    }
    for (int k = 0; k < D; ++k) f[k] = (diff[k] / mag) * force_mag;
    return 1;
}

/* brute_force_seq_n_body<D>: nbody-sim-new/methods.cpp:7-42.  THE PARITY ORACLE.
 * i<j symmetric sweep; forces[j] += force, forces[i] -= force (methods.cpp:36-37). */
int oracle_brute_force_seq(const double* bodies, size_t n, int D, double* forces) {
    if (D != 2 && D != 3) return -1;
    for (size_t i = 0; i < n * (size_t)D; ++i) forces[i] = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double* pi = POS(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        for (size_t j = i + 1; j < n; ++j) {
            double f[3];
            if (!pair_force(pi, POS(bodies, j, D), mi, MASS(bodies, j, D), D, f)) continue;
            for (int k = 0; k < D; ++k) forces[j * D + k] += f[k];
            for (int k = 0; k < D; ++k) forces[i * D + k] -= f[k];
        }
    }
    return 0;
}

/* brute_force_omp_n_body_2<D>: nbody-sim-new/methods.cpp:98-136.  All-to-all, i==j skipped
 * (:113), forces[i] -= force (:131), j ascending.  Each row is independent of the thread count. */
int oracle_brute_force_omp_2(const double* bodies, size_t n, int D, double* forces) {
    if (D != 2 && D != 3) return -1;
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < D; ++k) forces[i * D + k] = 0.0;
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        const double* pi = POS(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            double f[3];
            if (!pair_force(pi, POS(bodies, j, D), mi, MASS(bodies, j, D), D, f)) continue;
            for (int k = 0; k < D; ++k) forces[i * D + k] -= f[k];
        }
    }
    return 0;
}

/* Selected rows of brute_force_omp_n_body_2 (methods.cpp:110-133): out[r] = forces[rows[r]].
 * Used for sampled-target parity at N where a full CPU evaluation takes too long. */
int oracle_force_rows_omp_2(const double* bodies, size_t n, int D, const int64_t* rows, size_t nrows, double* out) {
    if (D != 2 && D != 3) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t r = 0; r < nrows; ++r) {
        const size_t i = (size_t)rows[r];
        const double* pi = POS(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        double acc[3] = {0.0, 0.0, 0.0};
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            double f[3];
            if (!pair_force(pi, POS(bodies, j, D), mi, MASS(bodies, j, D), D, f)) continue;
            for (int k = 0; k < D; ++k) acc[k] -= f[k];
        }
        for (int k = 0; k < D; ++k) out[r * D + k] = acc[k];
    }
    return 0;
}

/* brute_force_omp_n_body_1<D>: nbody-sim-new/methods.cpp:45-95.  Symmetric i<j inside an
 * `omp for` (default static schedule) with one n-long local array per thread (:54), combined
 * serially in thread order (:88-92).  The result depends on the thread count through the
 * partition of i, exactly as in the reference. */
int oracle_brute_force_omp_1(const double* bodies, size_t n, int D, double* forces) {
    if (D != 2 && D != 3) return -1;
    int num_threads = 1;
#ifdef _OPENMP
    num_threads = omp_get_max_threads();
#endif
    double* local = (double*)calloc((size_t)num_threads * n * (size_t)D, sizeof(double));
    if (!local && n) return -2;
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        double* loc = local + (size_t)t * n * (size_t)D;
#pragma omp for
        for (size_t i = 0; i < n; ++i) {
            const double* pi = POS(bodies, i, D);
            const double mi = MASS(bodies, i, D);
            for (size_t j = i + 1; j < n; ++j) {
                double f[3];
                if (!pair_force(pi, POS(bodies, j, D), mi, MASS(bodies, j, D), D, f)) continue;
                for (int k = 0; k < D; ++k) loc[j * D + k] += f[k];
                for (int k = 0; k < D; ++k) loc[i * D + k] -= f[k];
            }
        }
    }
    for (size_t i = 0; i < n * (size_t)D; ++i) forces[i] = 0.0;
    for (int t = 0; t < num_threads; ++t) {
        const double* loc = local + (size_t)t * n * (size_t)D;
        for (size_t i = 0; i < n * (size_t)D; ++i) forces[i] += loc[i];
    }
    free(local);
    return 0;
}

/* update_body_velocities<D>: nbody-sim-new/methods.cpp:425-438.
 * velocity += forces[i] / mass * dt  ==  v[k] += (F[k]/m)*dt  (Vector/scalar, then *scalar, then +=). */
int oracle_update_body_velocities(double* bodies, const double* forces, size_t n, int D, double dt) {
    if (D != 2 && D != 3) return -1;
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        double* v = VEL(bodies, i, D);
        const double m = MASS(bodies, i, D);
        for (int k = 0; k < D; ++k) v[k] += (forces[i * D + k] / m) * dt;
    }
    return 0;
}

/* update_body_positions<D>: nbody-sim-new/methods.cpp:440-450.  x[k] += v[k]*dt. */
int oracle_update_body_positions(double* bodies, size_t n, int D, double dt) {
    if (D != 2 && D != 3) return -1;
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
        double* x = POS(bodies, i, D);
        const double* v = VEL(bodies, i, D);
        for (int k = 0; k < D; ++k) x[k] += v[k] * dt;
    }
    return 0;
}

/* The time loop the reference never wrote: forces -> kick -> drift, nsteps times,
 * composed from the three reference leaves above.  variant 0 = seq forces, 2 = omp_2 forces. */
int oracle_leapfrog(double* bodies, size_t n, int D, double dt, int nsteps, int variant) {
    double* forces = (double*)malloc(n * (size_t)D * sizeof(double));
    if (!forces && n) return -2;
    int rc = 0;
    for (int s = 0; s < nsteps && rc == 0; ++s) {
        rc = (variant == 0) ? oracle_brute_force_seq(bodies, n, D, forces) : oracle_brute_force_omp_2(bodies, n, D, forces);
        if (rc) break;
        oracle_update_body_velocities(bodies, forces, n, D, dt);
        oracle_update_body_positions(bodies, n, D, dt);
    }
    free(forces);
    return rc;
}

/* ---- std::mt19937 + libstdc++ std::uniform_real_distribution<double>, restated ------------- */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t* g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* libstdc++ generate_canonical<double,53>(mt19937): two 32-bit draws, low word first. */
static double canonical53(mt19937_t* g) {
    double sum = (double)mt_next(g);
    sum += (double)mt_next(g) * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

static double uniform_real(mt19937_t* g, double a, double b) { return canonical53(g) * (b - a) + a; }

/* generate_random_bodies<D>: nbody-sim-new/utils.h:107-135, with the unseeded std::random_device
 * replaced by an explicit seed.  Ranges: position U[1,1e7), velocity U[-10,10), mass U[1,1e8)
 * (:113-115); draw order per body p0,v0,p1,v1,(p2,v2,) mass (:125-130). */
int oracle_generate_random_bodies(uint32_t seed, size_t n, int D, double* bodies) {
    if (D != 2 && D != 3) return -1;
    mt19937_t g;
    mt_seed(&g, seed);
    for (size_t i = 0; i < n; ++i) {
        double* x = POS(bodies, i, D);
        double* v = VEL(bodies, i, D);
        for (int d = 0; d < D; ++d) {
            x[d] = uniform_real(&g, 1.0, 10000000.0);
            v[d] = uniform_real(&g, -10.0, 10.0);
        }
        MASS(bodies, i, D) = uniform_real(&g, 1.0, 100000000.0);
    }
    return 0;
}

/* compute_accuracy_omp<D>: nbody-sim-new/utils.h:170-219.  Percentage of bodies whose every
 * component is within 1 % relative of the reference; |ref|<1e-20 components compared absolutely
 * against 1e-9. */
double oracle_compute_accuracy(const double* forces, const double* ref, size_t n, int D) {
    size_t accurate = 0;
    for (size_t i = 0; i < n; ++i) {
        int ok = 1;
        for (int d = 0; d < D; ++d) {
            double r = ref[i * D + d], f = forces[i * D + d];
            if (fabs(r) < ORACLE_ACCURACY_FORCE_THRESHOLD) {
                if (fabs(f) > 1e-9) { ok = 0; break; }
                continue;
            }
            double rel = fabs((f - r) / r);
            if (rel > ORACLE_ACCURACY_PCT_THRESHOLD) { ok = 0; break; }
        }
        accurate += (size_t)ok;
    }
    return n ? 100.0 * (double)(int)accurate / (double)n : 0.0;
}

/* ---- test-protocol helpers (not reference functions) ---------------------------------------- */

/* Round positions and masses to the nearest fp32 and widen back: the oracle then consumes exactly
 * the values the fp32 device path sees (SURVEY F10 / 8d accuracy protocol). Velocities untouched. */
void oracle_round_inputs_to_f32(double* bodies, size_t n, int D) {
    for (size_t i = 0; i < n; ++i) {
        double* x = POS(bodies, i, D);
        for (int d = 0; d < D; ++d) x[d] = (double)(float)x[d];
        MASS(bodies, i, D) = (double)(float)MASS(bodies, i, D);
    }
}

/* Total energy under the potential that matches the reference law:
 * F_i = -G m_i sum_j m_j (p_j-p_i)/r^4  = -grad_i U,  U = sum_{i<j} G m_i m_j / (2 r^2), pairs with
 * r^2 < 1e-10 excluded; kinetic = sum m v^2 / 2.  out[0]=kinetic, out[1]=potential. */
int oracle_energy(const double* bodies, size_t n, int D, double* out) {
    double ke = 0.0, pe = 0.0;
#pragma omp parallel for reduction(+ : ke, pe) schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) {
        const double* xi = POS(bodies, i, D);
        const double* vi = VEL(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        double v2 = 0.0;
        for (int k = 0; k < D; ++k) v2 += vi[k] * vi[k];
        ke += 0.5 * mi * v2;
        double row = 0.0;
        for (size_t j = i + 1; j < n; ++j) {
            const double* xj = POS(bodies, j, D);
            double r2 = 0.0;
            for (int k = 0; k < D; ++k) { double d = xj[k] - xi[k]; r2 += d * d; }
            if (r2 < ORACLE_R2_SKIP) continue;
            row += MASS(bodies, j, D) / r2;
        }
        pe += 0.5 * ORACLE_G * mi * row;
    }
    out[0] = ke;
    out[1] = pe;
    return 0;
}

/* S_i = sum_j |f_ij| (sum of pair-force magnitudes on body i): the scale against which a rounded
 * evaluation of the cancelling sum F_i = -sum_j f_ij can be judged; kappa_i = S_i/|F_i| is the
 * condition number of that sum.  rows == NULL: all bodies. */
int oracle_force_magnitude_sums(const double* bodies, size_t n, int D, const int64_t* rows, size_t nrows, double* out) {
    if (D != 2 && D != 3) return -1;
    const size_t cnt = rows ? nrows : n;
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t r = 0; r < cnt; ++r) {
        const size_t i = rows ? (size_t)rows[r] : r;
        const double* pi = POS(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        double s = 0.0;
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            double f[3];
            if (!pair_force(pi, POS(bodies, j, D), mi, MASS(bodies, j, D), D, f)) continue;
            double m2 = 0.0;
            for (int k = 0; k < D; ++k) m2 += f[k] * f[k];
            s += sqrt(m2);
        }
        out[r] = s;
    }
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- leaf-pair direct sums of the reference's tree codes --------------------------------
 *
 * The near-field ("P2P") step of the reference's tree methods is a batch of small all-pairs sums: every body
 * of a target leaf against every body of each source leaf on that leaf's list.  Three per-pair laws occur:
 *
 *   law 0  BRUTE     nbody-sim-new/methods.cpp:21-37 (the brute-force law above; repulsive: forces[i] -= f)
 *   law 1  TREE_LEAF nbody-sim-new/octree.cpp:105-125 (single-body leaf of the Barnes-Hut octree) and
 *                    nbody-sim-new/bvh.cpp:150-176 (BVH leaf loop): skip when every |d_k| <= 1e-9 ("same
 *                    position"), skip when dist_sq < 1e-9, else force += (diff/|diff|) * ((G*mi)*mj)/(dist_sq*dist)
 *                    -- attractive (diff = other - body, added).
 *   law 2  FMM_P2P   nbody-sim-new/fmm_parlay.cpp:992-1021: skip when every |d_k| <= 1e-14; dist_sq < 1e-10 is
 *                    NOT skipped but smoothed: dist_sq += (1e-5)^2 before the magnitude; the direction stays
 *                    diff.normalized() of the unsmoothed diff (zero vector below 1e-10, vector.h:93-97); attractive.
 *
 * Parity status: law 0 pinned as above; law 1 pinned against the reference's own octree object code evaluated with
 * theta = 0 (every leaf visited: oracle/ref_driver.cpp ref_octree_direct_forces, tests/test_oracle_vs_ref.py) to fp64
 * re-association noise (the tree visits sources in its own order); law 2 PARITY UNPINNED -- restated from the cited
 * lines only: FMM_Parlay<D>'s constructor leaves its tree pointing into a destroyed local vector
 * (fmm_parlay.cpp:16-22 with fmm.cpp:389-395), so executing the reference's p2p_phase is undefined behaviour and is
 * not done.  The batching (CSR leaf lists) is this build's own; the reference walks pointers.
 *
 * leaf l owns the bodies leaf_bodies[leaf_offsets[l] .. leaf_offsets[l+1]) (indices into `bodies`; a body belongs to
 * at most one leaf); target leaf t sums over the source leaves list_sources[list_offsets[t] .. list_offsets[t+1]) in
 * list order, bodies in leaf order.  forces[n*D] (zero for bodies in no leaf).
 */
static inline void leaf_pair_term(const double* pi, const double* pj, double mi, double mj, int D, int law, double* acc) {
    double diff[3], dist_sq = 0.0;
    for (int k = 0; k < D; ++k) diff[k] = pj[k] - pi[k];
    if (law == 0) {
        double f[3];
        if (!pair_force(pi, pj, mi, mj, D, f)) return;
        for (int k = 0; k < D; ++k) acc[k] -= f[k];
        return;
    }
    const double same_tol = (law == 1) ? 1e-9 : 1e-14;
    int same = 1;
    for (int k = 0; k < D; ++k) if (fabs(diff[k]) > same_tol) { same = 0; break; }
    if (same) return;
    for (int k = 0; k < D; ++k) dist_sq += diff[k] * diff[k];
    const double mag = sqrt(dist_sq);                 /* diff.normalized(): of the unsmoothed diff */
    if (law == 1) {
        if (dist_sq < 1e-9) return;                   /* octree.cpp:119, bvh.cpp:167 */
    } else if (dist_sq < 1e-10) {
        const double epsilon = 1e-5;                  /* fmm_parlay.cpp:1010-1013 */
        dist_sq += epsilon * epsilon;
    }
    const double dist = sqrt(dist_sq);
    const double force_mag = ORACLE_G * mi * mj / (dist_sq * dist);
    if (mag < 1e-10) return;                          /* normalized() returns the zero vector */
    for (int k = 0; k < D; ++k) acc[k] += (diff[k] / mag) * force_mag;
}

int oracle_leaf_pair_forces(const double* bodies, size_t n, int D, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies,
                            size_t n_leaves, const uint32_t* list_offsets, const uint32_t* list_sources, int law, double* forces) {
    if ((D != 2 && D != 3) || law < 0 || law > 2) return -1;
    for (size_t i = 0; i < n * (size_t)D; ++i) forces[i] = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t t = 0; t < n_leaves; ++t) {
        for (uint32_t a = leaf_offsets[t]; a < leaf_offsets[t + 1]; ++a) {
            const size_t i = leaf_bodies[a];
            double acc[3] = {0.0, 0.0, 0.0};
            for (uint32_t e = list_offsets[t]; e < list_offsets[t + 1]; ++e) {
                const uint32_t s = list_sources[e];
                for (uint32_t b = leaf_offsets[s]; b < leaf_offsets[s + 1]; ++b) {
                    const size_t j = leaf_bodies[b];
                    if (law == 0 && i == j) continue;  /* methods.cpp:113 */
                    leaf_pair_term(POS(bodies, i, D), POS(bodies, j, D), MASS(bodies, i, D), MASS(bodies, j, D), D, law, acc);
                }
            }
            for (int k = 0; k < D; ++k) forces[i * D + k] = acc[k];
        }
    }
    return 0;
}

/* sum_j |f_ij| over the same pairs (condition numbers for the stated fp32 tolerance) */
int oracle_leaf_pair_magnitude_sums(const double* bodies, size_t n, int D, const uint32_t* leaf_offsets, const uint32_t* leaf_bodies,
                                    size_t n_leaves, const uint32_t* list_offsets, const uint32_t* list_sources, int law, double* sums) {
    if ((D != 2 && D != 3) || law < 0 || law > 2) return -1;
    for (size_t i = 0; i < n; ++i) sums[i] = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t t = 0; t < n_leaves; ++t) {
        for (uint32_t a = leaf_offsets[t]; a < leaf_offsets[t + 1]; ++a) {
            const size_t i = leaf_bodies[a];
            double s_acc = 0.0;
            for (uint32_t e = list_offsets[t]; e < list_offsets[t + 1]; ++e) {
                const uint32_t s = list_sources[e];
                for (uint32_t b = leaf_offsets[s]; b < leaf_offsets[s + 1]; ++b) {
                    const size_t j = leaf_bodies[b];
                    if (law == 0 && i == j) continue;
                    double acc[3] = {0.0, 0.0, 0.0};
                    leaf_pair_term(POS(bodies, i, D), POS(bodies, j, D), MASS(bodies, i, D), MASS(bodies, j, D), D, law, acc);
                    double m2 = 0.0;
                    for (int k = 0; k < D; ++k) m2 += acc[k] * acc[k];
                    s_acc += sqrt(m2);
                }
            }
            sums[i] = s_acc;
        }
    }
    return 0;
}

/* ---- EXTENSION checker: Plummer-softened pair law (nbx_ctx_set_softening) -----------------------------------
 * NOT in the reference (its brute force is unsoftened; utils.h:24 defines an unused SOFTENING constant):
 * nothing to pin against.  The law is the reference's with r^2 -> r^2 + eps^2 and without the r^2 < 1e-10 skip:
 *   F_i = -(G m_i) sum_{j != i} m_j (p_j - p_i) / (r^2 + eps^2)^2 ,  U = sum_{i<j} G m_i m_j / (2 (r^2 + eps^2)).
 * For eps -> 0 and no pair below 1e-5 it tends to oracle_brute_force_omp_2 (tests/test_leaf_pairs_cpu.py checks that).
 * rows == NULL: all bodies.  sums (optional) receives sum_j |f_ij|. */
int oracle_force_rows_softened(const double* bodies, size_t n, int D, double eps, const int64_t* rows, size_t nrows,
                               double* out, double* sums) {
    if (D != 2 && D != 3) return -1;
    const size_t cnt = rows ? nrows : n;
    const double e2 = eps * eps;
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t r = 0; r < cnt; ++r) {
        const size_t i = rows ? (size_t)rows[r] : r;
        const double* pi = POS(bodies, i, D);
        const double gmi = ORACLE_G * MASS(bodies, i, D);
        double acc[3] = {0.0, 0.0, 0.0}, s = 0.0;
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            const double* pj = POS(bodies, j, D);
            double d[3] = {0.0, 0.0, 0.0}, r2 = 0.0;
            for (int k = 0; k < D; ++k) { d[k] = pj[k] - pi[k]; r2 += d[k] * d[k]; }
            const double q = r2 + e2;
            const double wgt = gmi * MASS(bodies, j, D) / (q * q);
            for (int k = 0; k < D; ++k) acc[k] -= wgt * d[k];
            s += wgt * sqrt(r2);
        }
        for (int k = 0; k < D; ++k) out[r * D + k] = acc[k];
        if (sums) sums[r] = s;
    }
    return 0;
}

int oracle_energy_softened(const double* bodies, size_t n, int D, double eps, double* out) {
    double ke = 0.0, pe = 0.0;
    const double e2 = eps * eps;
#pragma omp parallel for reduction(+ : ke, pe) schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) {
        const double* xi = POS(bodies, i, D);
        const double* vi = VEL(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        double v2 = 0.0;
        for (int k = 0; k < D; ++k) v2 += vi[k] * vi[k];
        ke += 0.5 * mi * v2;
        double row = 0.0;
        for (size_t j = i + 1; j < n; ++j) {
            const double* xj = POS(bodies, j, D);
            double r2 = 0.0;
            for (int k = 0; k < D; ++k) { double d = xj[k] - xi[k]; r2 += d * d; }
            row += MASS(bodies, j, D) / (r2 + e2);
        }
        pe += 0.5 * ORACLE_G * mi * row;
    }
    out[0] = ke;
    out[1] = pe;
    return 0;
}

/* ---- EXTENSION checker: softened Newtonian law (nbx_ctx_set_law(NBX_FORCE_LAW_NEWTON)) -----------------------
 * NOT in the reference's brute force (parity unpinned by construction):
 *   F_i = +(G m_i) sum_{j != i} m_j (p_j - p_i) / (r^2 + eps^2)^(3/2) ,  U = -sum_{i<j} G m_i m_j / sqrt(r^2 + eps^2). */
int oracle_force_rows_newton(const double* bodies, size_t n, int D, double eps, const int64_t* rows, size_t nrows,
                             double* out, double* sums) {
    if (D != 2 && D != 3) return -1;
    const size_t cnt = rows ? nrows : n;
    const double e2 = eps * eps;
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t r = 0; r < cnt; ++r) {
        const size_t i = rows ? (size_t)rows[r] : r;
        const double* pi = POS(bodies, i, D);
        const double gmi = ORACLE_G * MASS(bodies, i, D);
        double acc[3] = {0.0, 0.0, 0.0}, s = 0.0;
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            const double* pj = POS(bodies, j, D);
            double d[3] = {0.0, 0.0, 0.0}, r2 = 0.0;
            for (int k = 0; k < D; ++k) { d[k] = pj[k] - pi[k]; r2 += d[k] * d[k]; }
            const double q = r2 + e2;
            const double wgt = gmi * MASS(bodies, j, D) / (q * sqrt(q));
            for (int k = 0; k < D; ++k) acc[k] += wgt * d[k];
            s += wgt * sqrt(r2);
        }
        for (int k = 0; k < D; ++k) out[r * D + k] = acc[k];
        if (sums) sums[r] = s;
    }
    return 0;
}

int oracle_energy_newton(const double* bodies, size_t n, int D, double eps, double* out) {
    double ke = 0.0, pe = 0.0;
    const double e2 = eps * eps;
#pragma omp parallel for reduction(+ : ke, pe) schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) {
        const double* xi = POS(bodies, i, D);
        const double* vi = VEL(bodies, i, D);
        const double mi = MASS(bodies, i, D);
        double v2 = 0.0;
        for (int k = 0; k < D; ++k) v2 += vi[k] * vi[k];
        ke += 0.5 * mi * v2;
        double row = 0.0;
        for (size_t j = i + 1; j < n; ++j) {
            const double* xj = POS(bodies, j, D);
            double r2 = 0.0;
            for (int k = 0; k < D; ++k) { double d = xj[k] - xi[k]; r2 += d * d; }
            row += MASS(bodies, j, D) / sqrt(r2 + e2);
        }
        pe -= ORACLE_G * mi * row;
    }
    out[0] = ke;
    out[1] = pe;
    return 0;
}
