/* selftest.c -- sanitizer run of the oracle (CPU build only: GPU ASan is not available on this pool).
 * Built with -fsanitize=address,undefined by `make -C oracle asan-check`; exercises every entry point on small
 * and degenerate inputs (n = 0, 1, ragged thread partitions) so out-of-bounds or UB in the checker shows up. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_brute_force_seq(const double*, size_t, int, double*);
int oracle_brute_force_omp_1(const double*, size_t, int, double*);
int oracle_brute_force_omp_2(const double*, size_t, int, double*);
int oracle_force_rows_omp_2(const double*, size_t, int, const int64_t*, size_t, double*);
int oracle_force_magnitude_sums(const double*, size_t, int, const int64_t*, size_t, double*);
int oracle_update_body_velocities(double*, const double*, size_t, int, double);
int oracle_update_body_positions(double*, size_t, int, double);
int oracle_leapfrog(double*, size_t, int, double, int, int);
int oracle_generate_random_bodies(uint32_t, size_t, int, double*);
double oracle_compute_accuracy(const double*, const double*, size_t, int);
void oracle_round_inputs_to_f32(double*, size_t, int);
int oracle_energy(const double*, size_t, int, double*);
int oracle_leaf_pair_forces(const double*, size_t, int, const uint32_t*, const uint32_t*, size_t, const uint32_t*, const uint32_t*, int, double*);
int oracle_leaf_pair_magnitude_sums(const double*, size_t, int, const uint32_t*, const uint32_t*, size_t, const uint32_t*, const uint32_t*, int, double*);
int oracle_force_rows_softened(const double*, size_t, int, double, const int64_t*, size_t, double*, double*);
int oracle_force_rows_newton(const double*, size_t, int, double, const int64_t*, size_t, double*, double*);
int oracle_energy_softened(const double*, size_t, int, double, double*);
int oracle_energy_newton(const double*, size_t, int, double, double*);

int main(void) {
    const size_t sizes[] = {0, 1, 2, 3, 17, 257};
    for (int D = 2; D <= 3; ++D)
        for (size_t s = 0; s < sizeof sizes / sizeof sizes[0]; ++s) {
            const size_t n = sizes[s], w = 2 * (size_t)D + 1;
            double* b = malloc((n * w + 1) * sizeof(double));
            double* f = malloc((n * D + 1) * sizeof(double));
            double* g = malloc((n * D + 1) * sizeof(double));
            int64_t* rows = malloc((n + 1) * sizeof(int64_t));
            if (!b || !f || !g || !rows) return 2;
            oracle_generate_random_bodies(7u + (uint32_t)n, n, D, b);
            oracle_round_inputs_to_f32(b, n, D);
            if (oracle_brute_force_seq(b, n, D, f) || oracle_brute_force_omp_1(b, n, D, g) || oracle_brute_force_omp_2(b, n, D, g)) return 3;
            for (size_t i = 0; i < n; ++i) rows[i] = (int64_t)(n - 1 - i);
            if (oracle_force_rows_omp_2(b, n, D, rows, n, g) || oracle_force_magnitude_sums(b, n, D, rows, n, g)) return 4;
            if (oracle_force_magnitude_sums(b, n, D, NULL, 0, g)) return 5;
            (void)oracle_compute_accuracy(f, f, n, D);
            oracle_update_body_velocities(b, f, n, D, 0.5);
            oracle_update_body_positions(b, n, D, 0.5);
            if (oracle_leapfrog(b, n, D, 1.0, 2, 0) || oracle_leapfrog(b, n, D, 1.0, 2, 2)) return 6;
            double e[2];
            oracle_energy(b, n, D, e);
            if (n > 1 && !(isfinite(e[0]) && isfinite(e[1]))) return 7;
            /* extension checkers: softened / Newtonian rows (all bodies and a row subset) and energies */
            double* sums = malloc((n + 1) * sizeof(double));
            if (!sums) return 2;
            if (oracle_force_rows_softened(b, n, D, 0.5, NULL, 0, g, sums) || oracle_force_rows_newton(b, n, D, 0.5, rows, n, g, NULL)) return 8;
            oracle_energy_softened(b, n, D, 0.5, e);
            oracle_energy_newton(b, n, D, 0.5, e);
            /* leaf-pair sums: three leaves (one empty, ragged sizes), lists with a repeated and an empty entry, every law */
            uint32_t* lb = malloc((n + 1) * sizeof(uint32_t));
            if (!lb) return 2;
            for (size_t i = 0; i < n; ++i) lb[i] = (uint32_t)(n - 1 - i);
            const uint32_t lo[4] = {0, (uint32_t)(n / 3), (uint32_t)(n / 3), (uint32_t)n};
            const uint32_t so[4] = {0, 3, 3, 5};
            const uint32_t ss[5] = {0, 2, 0, 2, 1};
            for (int law = 0; law < 3; ++law)
                if (oracle_leaf_pair_forces(b, n, D, lo, lb, 3, so, ss, law, g) || oracle_leaf_pair_magnitude_sums(b, n, D, lo, lb, 3, so, ss, law, sums)) return 9;
            free(b); free(f); free(g); free(rows); free(sums); free(lb);
        }
    puts("oracle selftest ok");
    return 0;
}
