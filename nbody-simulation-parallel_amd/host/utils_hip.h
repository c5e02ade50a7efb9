// utils_hip.h -- harness utilities with the reference's contracts (nbody-sim-new/utils.h): constants,
// body generator, timing wrapper, accuracy metric, validation print, run id.  From scratch.
#ifndef NBODY_AMD_UTILS_HIP_H
#define NBODY_AMD_UTILS_HIP_H

#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <exception>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#if __has_include("nbody_types.h")
#include "nbody_types.h"  // this repository's layout-locked Vector<D> / Body<D>
#else
#include "body.h"  // inside the reference tree: the reference's own types (same memory)
#include "vector.h"
#endif

// utils.h:21-27
constexpr double G = 4.471e-21;
constexpr double ACCURACY_PCT_THRESHOLD = 0.01;
constexpr double ACCURACY_FORCE_THRESHOLD = 1e-20;

inline void ensure_results_directory() { std::filesystem::create_directories("results"); }

// MMDDYYYY_HHMMSS in local time (utils.h:67-83)
inline std::string get_run_id() {
    const std::time_t now = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
    std::tm tm_now{};
    localtime_r(&now, &tm_now);
    char buf[32];
    std::strftime(buf, sizeof buf, "%m%d%Y_%H%M%S", &tm_now);
    return buf;
}

// Wall time of one solver call in microseconds; -1 if it threw (utils.h:87-104).  A failing method
// is logged and skipped, it never aborts the sweep.
template <typename Func>
long long safely_execute(std::ofstream& log_file, const std::string& method_name, Func&& func) {
    try {
        const auto t0 = std::chrono::high_resolution_clock::now();
        auto result = func();
        (void)result;
        const auto t1 = std::chrono::high_resolution_clock::now();
        return std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
    } catch (const std::exception& e) {
        for (std::ostream* o : {static_cast<std::ostream*>(&log_file), static_cast<std::ostream*>(&std::cerr)})
            *o << "Error executing " << method_name << ": " << e.what() << std::endl;
    } catch (...) {
        for (std::ostream* o : {static_cast<std::ostream*>(&log_file), static_cast<std::ostream*>(&std::cerr)})
            *o << "Unknown error executing " << method_name << std::endl;
    }
    return -1;
}

// Uniform bodies with the reference's ranges and draw order (utils.h:107-135): per body
// p0,v0,p1,v1,(p2,v2,) mass; position U[1,1e7), velocity U[-10,10), mass U[1,1e8).
// seed < 0: std::random_device like the reference; otherwise a reproducible std::mt19937(seed).
template <int D>
std::vector<Body<D>> generate_random_bodies(int n, long long seed = -1) {
    std::mt19937 gen(seed < 0 ? std::random_device{}() : static_cast<std::uint32_t>(seed));
    std::uniform_real_distribution<double> pos(1, 10000000.0), vel(-10.0, 10.0), mass(1, 100000000.0);
    std::vector<Body<D>> bodies;
    bodies.reserve(static_cast<std::size_t>(n));
    for (int i = 0; i < n; ++i) {
        Vector<D> p, v;
        for (int d = 0; d < D; ++d) {
            p[d] = pos(gen);
            v[d] = vel(gen);
        }
        const double m = mass(gen);
        bodies.emplace_back(p, v, m);
    }
    return bodies;
}

// Plummer sphere (not in the reference; BASELINE config 5): total mass M split equally, scale
// radius a, radii cut at 10 a, isotropic velocities drawn by von Neumann rejection from the
// Plummer distribution function and scaled to virial equilibrium for the NEWTONIAN law -- under
// the reference's repulsive 1/r^3 law the sphere simply expands; energy is still conserved.
template <int D>
std::vector<Body<D>> generate_plummer_bodies(int n, long long seed, double a = 1.0e5, double M = 1.0e12, double Gn = G) {
    std::mt19937_64 gen(static_cast<std::uint64_t>(seed < 0 ? std::random_device{}() : seed));
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> Nrm(0.0, 1.0);
    auto direction = [&](double len) {
        Vector<D> v;
        double s = 0.0;
        do {
            s = 0.0;
            for (int d = 0; d < D; ++d) { v[d] = Nrm(gen); s += v[d] * v[d]; }
        } while (s == 0.0);
        return v * (len / std::sqrt(s));
    };
    std::vector<Body<D>> bodies;
    bodies.reserve(static_cast<std::size_t>(n));
    const double centre = 5.0e6;  // middle of the reference's box
    for (int i = 0; i < n; ++i) {
        double r;
        do {
            const double u = U(gen);
            r = a / std::sqrt(std::pow(u > 0 ? u : 1e-300, -2.0 / 3.0) - 1.0);
        } while (!(r < 10.0 * a));
        double q, g;
        do { q = U(gen); g = 0.1 * U(gen); } while (g > q * q * std::pow(1.0 - q * q, 3.5));
        const double vesc = std::sqrt(2.0 * Gn * M / std::sqrt(r * r + a * a));
        Vector<D> p = direction(r);
        for (int d = 0; d < D; ++d) p[d] += centre;
        bodies.emplace_back(p, direction(q * vesc), M / n);
    }
    return bodies;
}

// % of bodies whose every force component is within 1 % of the reference; components with
// |ref| < 1e-20 are held to |f| <= 1e-9 instead (utils.h:170-219).
template <int D>
double compute_accuracy(const std::vector<Vector<D>>& forces, const std::vector<Vector<D>>& reference) {
    if (forces.size() != reference.size()) {
        std::cerr << "Error: Force vector sizes do not match for accuracy calculation." << std::endl;
        return 0.0;
    }
    const std::size_t n = forces.size();
    long long good = 0;
#pragma omp parallel for reduction(+ : good)
    for (std::size_t i = 0; i < n; ++i) {
        bool ok = true;
        for (int d = 0; d < D && ok; ++d) {
            const double r = reference[i][d], f = forces[i][d];
            if (std::abs(r) < ACCURACY_FORCE_THRESHOLD) ok = !(std::abs(f) > 1e-9);
            else ok = !(std::abs((f - r) / r) > ACCURACY_PCT_THRESHOLD);
        }
        good += ok ? 1 : 0;
    }
    return n ? 100.0 * static_cast<double>(good) / static_cast<double>(n) : 0.0;
}

// Forces of three evenly spaced bodies, the reference's eyeball check (utils.h:138-152).  The
// reference divides by n/3 (undefined for n < 3); here n < 3 prints every body.
template <int D>
void print_validation_forces(const std::vector<Vector<D>>& forces, int n, std::ostream& out) {
    const int stride = n >= 3 ? n / 3 : 1;
    for (int i = 0; i < n; ++i) {
        if ((i + 1) % stride != 0) continue;
        out << "Body #" << i + 1 << " force: (";
        for (int d = 0; d < D; ++d) out << forces[static_cast<std::size_t>(i)][d] << (d < D - 1 ? ", " : "");
        out << ")" << std::endl;
    }
}

#endif  // NBODY_AMD_UTILS_HIP_H
