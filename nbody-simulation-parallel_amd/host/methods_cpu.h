// methods_cpu.h -- the harness's CPU rows (next-row 8f-1): brute-force solvers and integrator
// helpers with the reference's names and signatures (nbody-sim-new/methods.h:29-37, :85-91),
// restated from scratch in fp64.  They exist so the harness can print the reference's own
// BruteForce_Sequential / BruteForce_OpenMP1 / BruteForce_OpenMP2 rows and the `-a 1` accuracy
// column next to BruteForce_HIP.  They are separate, explicitly named methods: the HIP entry points
// in methods_hip.h never fall back to them.
#ifndef NBODY_AMD_METHODS_CPU_H
#define NBODY_AMD_METHODS_CPU_H

#include <vector>

#if __has_include("nbody_types.h")
#include "nbody_types.h"  // this repository's layout-locked Vector<D> / Body<D>
#else
#include "body.h"  // inside the reference tree: the reference's own types (same memory)
#include "vector.h"
#endif

template <int D> std::vector<Vector<D>> brute_force_seq_n_body(const std::vector<Body<D>>& bodies);
template <int D> std::vector<Vector<D>> brute_force_omp_n_body_1(const std::vector<Body<D>>& bodies);
template <int D> std::vector<Vector<D>> brute_force_omp_n_body_2(const std::vector<Body<D>>& bodies);
template <int D> void update_body_velocities(std::vector<Body<D>>& bodies, const std::vector<Vector<D>>& forces, double dt);
template <int D> void update_body_positions(std::vector<Body<D>>& bodies, double dt);

#endif
