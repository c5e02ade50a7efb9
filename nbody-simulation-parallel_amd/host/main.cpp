// main.cpp -- benchmark harness with the reference's command line, result files and CSV schema
// (nbody-sim-new/main.cpp:18-875 run_benchmark, :877-951 main), rebuilt around a method table
// instead of one copied block per method, with the MI355X row BruteForce_HIP added.
//
//   ./nbody_sim -N 100000 -d 3 -a 1            reference rows + BruteForce_HIP, accuracy column
//   ./nbody_sim -N 1048576 -m g                HIP only (never gated by the 1e6-body CPU limit)
//   ./nbody_sim -N 65536 -m g --steps 100 --dt 5   device-resident kick/drift loop
//
// Output: results/run_<MMDDYYYY_HHMMSS>_N_<n>_<D>D.{csv,out}; CSV "Method,Bodies,Dimension,Time(s)[,Accuracy(%)]"
// with times fixed to 6 decimals (scientific below 1e-6), exactly the reference's format
// (main.cpp:41-43, 59-63, 159-170), so run_simulations.sh and the analysis notebook read it unchanged.
// The tree methods (b, h, f) are outside this build's scope: accepted on the command line, reported
// as "not built", skipped.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "methods_cpu.h"
#include "leaf_pairs_hip.h"
#include "methods_hip.h"
#include "utils_hip.h"

namespace {

struct Options {
    int dimension = 3;
    int num_bodies = 1000;
    bool accuracy = false;
    std::string methods;            // empty = everything that applies
    bool override_bf_limit = false; // "-m a" alone lifts the 1e6-body gate (main.cpp:905-907)
    long long seed = -1;            // -1: std::random_device like the reference
    int steps = 0;                  // > 0: also run the leapfrog loop on the device
    double dt = 1.0;
    std::string init = "uniform";   // uniform | plummer
    std::string dump;               // prefix: write bodies and each method's forces as raw doubles
    std::string load;               // --load <file>: raw Body<D> doubles (as --dump writes them) instead of generated bodies
    int step_offset = 0;            // --step-offset k: the loaded state is step k of a run (labels of the energy log)
    double e0 = 0.0;                // --e0 E: that run's initial energy (|dE/E0| keeps referring to it)
    bool have_e0 = false;
    double G = ::G;                 // --G: coupling constant of the HIP stepping loop (default: the reference's)
    int energy_every = 0;           // --energy-every k: log E and |dE/E0| every k steps of the loop
    std::string integrator = "kd";  // --integrator kd|kdk: kick-drift as the reference helpers are ordered, or kick-drift-kick (extension)
    std::string law = "reference";  // --law reference|newton: pair law of the stepping loop (newton: extension, needs --softening)
    double softening = 0.0;         // --softening eps: Plummer-softened law in the stepping loop (extension; 0 = reference law)
    std::vector<int> devices;       // --gpus / --devices: shard the HIP rows over these GPUs (one process)
    double refine = -1.0;           // --refine tol: per-body relative tolerance of the HIP rows (mixed mode); 0 = plain fp32; < 0: library default (1e-5)
};

int g_exit_code = 0;   // 3: a self-check of the run failed (sharded row against the 1-GPU row)

template <typename T>
void dump_raw(const std::string& path, const std::vector<T>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(T)));
}

// The state a previous run dumped (fp64 positions, velocities, masses: Body<D> memory, body.h:8-11) back in: a run of k steps,
// dumped, loaded and run k steps more equals one run of 2k steps bit for bit -- the device keeps its integrator state in fp64
// and derives everything else from it (tests/test_gpu_harness.py).
template <int D>
std::vector<Body<D>> load_bodies(const std::string& path, int n) {
    std::vector<Body<D>> bodies(static_cast<std::size_t>(n));
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.is_open()) throw std::runtime_error("--load: cannot open " + path);
    const std::streamsize want = static_cast<std::streamsize>(bodies.size() * sizeof(Body<D>));
    if (f.tellg() != want) throw std::runtime_error("--load: " + path + " does not hold exactly N bodies of this dimension");
    f.seekg(0);
    f.read(reinterpret_cast<char*>(bodies.data()), want);
    if (!f) throw std::runtime_error("--load: short read from " + path);
    return bodies;
}

// tee to log file and stdout, as every message of the reference harness is
struct Tee {
    std::ofstream& log;
    template <typename T>
    Tee& operator<<(const T& v) { log << v; std::cout << v; return *this; }
    Tee& operator<<(std::ostream& (*manip)(std::ostream&)) { manip(log); manip(std::cout); return *this; }
};

void write_time(std::ostream& csv, double seconds) {
    if (seconds < 1e-6) csv << "," << std::scientific << std::setprecision(6) << seconds;
    else csv << "," << std::fixed << std::setprecision(6) << seconds;
}

template <int D>
void run_benchmark(const std::vector<Body<D>>& bodies, const std::string& run_id, const Options& opt) {
    using Forces = std::vector<Vector<D>>;
    const int n = static_cast<int>(bodies.size());
    const std::string& m = opt.methods;
    const bool want_a = m.empty() || m.find('a') != std::string::npos;
    const bool run_cpu_bf = want_a && (n <= 1000000 || opt.override_bf_limit);  // main.cpp:24
    const bool run_hip = want_a || m.find('g') != std::string::npos;            // never size-gated
    const bool asked_tree = m.empty() || m.find_first_of("bhf") != std::string::npos;

    ensure_results_directory();
    const std::string base = "results/run_" + run_id + "_N_" + std::to_string(n) + "_" + std::to_string(D) + "D";
    std::ofstream csv(base + ".csv"), log(base + ".out"), hipcsv;
    if (!csv.is_open() || !log.is_open()) {
        std::cerr << "Failed to open output files: " << base << ".{csv,out}" << std::endl;
        return;
    }
    Tee out{log};
    if (!opt.dump.empty()) dump_raw(opt.dump + "_bodies.f64", bodies);
    csv << "Method,Bodies,Dimension,Time(s)" << (opt.accuracy ? ",Accuracy(%)" : "") << std::endl;

    out << "Running N-body simulation benchmark with:" << std::endl
        << "  Dimension: " << D << "D" << std::endl
        << "  Bodies: " << n << std::endl
        << "  Run ID: " << run_id << std::endl;
    if (opt.accuracy) out << "  Accuracy calculation: ON" << std::endl;
    out << "  Methods: " << (run_cpu_bf ? "Brute Force " : "") << (run_hip ? "Brute Force (HIP) " : "") << std::endl;
    if (asked_tree && !m.empty()) out << "  (Barnes-Hut / BVH / FMM are not built in this tier: skipped)" << std::endl;
    out << std::endl;

    // accuracy reference: sequential below 1e5 bodies, OpenMP all-to-all above (main.cpp:102-124)
    Forces reference;
    if (opt.accuracy) {
        const bool seq = n < 100000;
        out << "Using brute force " << (seq ? "sequential" : "OpenMP") << " as reference for accuracy calculation..." << std::endl;
        reference = seq ? brute_force_seq_n_body<D>(bodies) : brute_force_omp_n_body_2<D>(bodies);
    } else {
        out << "Accuracy calculation disabled." << std::endl;
    }

    struct Method {
        const char* label;       // CSV name
        const char* banner;      // heading printed before the run
        bool threaded;           // print "Using <k> threads..."
        bool enabled;
        std::function<Forces()> solve;
    };
    // GPU-count dimension of the sweep: the one-GPU row keeps the plain label; a row sharded over
    // G ranks is BruteForce_HIP_x<G>, so the reference's notebook groups the counts as separate methods.
    const int hip_ranks = opt.devices.size() > 1 ? static_cast<int>(opt.devices.size()) : 1;
    const std::string hip_label = hip_ranks > 1 ? "BruteForce_HIP_x" + std::to_string(hip_ranks) : std::string("BruteForce_HIP");
    int distinct_devices = 1;
    if (hip_ranks > 1) {
        std::vector<int> u = opt.devices;
        std::sort(u.begin(), u.end());
        distinct_devices = static_cast<int>(std::unique(u.begin(), u.end()) - u.begin());
    }
    const std::vector<Method> table = {
        {"BruteForce_Sequential", "Brute force O(n²) sequential approach:", false, run_cpu_bf,
         [&] { return brute_force_seq_n_body<D>(bodies); }},
        {"BruteForce_OpenMP1", "Brute force OpenMP parallel approach (memory-intensive):", true, run_cpu_bf,
         [&] { return brute_force_omp_n_body_1<D>(bodies); }},
        {"BruteForce_OpenMP2", "Brute force OpenMP parallel approach (memory-efficient):", true, run_cpu_bf,
         [&] { return brute_force_omp_n_body_2<D>(bodies); }},
        {hip_label.c_str(), "Brute force HIP (MI355X, fp32 tiled all-pairs) approach:", false, run_hip,
         [&] { return brute_force_hip_n_body<D>(bodies); }},
    };

    for (const Method& method : table) {
        if (!method.enabled) continue;
        out << method.banner << std::endl;
        if (method.threaded) out << "Using " << omp_get_max_threads() << " threads..." << std::endl;
        Forces forces;
        // the CPU rows before this one may have kept the device idle for minutes: its clocks and the runtime's copy queues
        // are woken outside the timed call, like the start-up in main() (measured: 96 ms instead of 3 ms at N = 100,000
        // after 25 s of CPU rows)
        if (!method.threaded && method.label == hip_label && run_cpu_bf) warm_up_hip();
        const long long us = safely_execute(log, method.label, [&] { forces = method.solve(); return 0; });
        if (us >= 0) {
            const double seconds = static_cast<double>(us) / 1e6;
            double accuracy = -1.0;
            if (opt.accuracy) accuracy = compute_accuracy<D>(forces, reference);
            csv << method.label << "," << n << "," << D;
            write_time(csv, seconds);
            if (opt.accuracy) csv << "," << std::fixed << std::setprecision(2) << accuracy;
            csv << std::endl;
            out << "Time taken: " << seconds << " s" << std::endl;
            if (opt.accuracy) out << "Accuracy: " << std::to_string(accuracy) << "%" << std::endl;
            if (method.label == hip_label && opt.accuracy) {
                // the same metric without the force array leaving the device (nbx_ctx_accuracy); must agree with the host's
                try {
                    const double dev = brute_force_hip_accuracy<D>(bodies, reference);
                    out << "Accuracy (device-side metric): " << std::to_string(dev) << "%" << (dev == accuracy ? "" : "  [differs from the host metric]") << std::endl;
                } catch (const std::exception& e) {
                    out << "Accuracy (device-side metric): unavailable (" << e.what() << ")" << std::endl;
                }
            }
            if (method.label == hip_label) {
                const double kernel_s = last_hip_run_info().kernel_ms * 1e-3;
                const double pairs = static_cast<double>(n) * static_cast<double>(n);
                out << "Kernel time: " << kernel_s << " s  (" << pairs / kernel_s << " pair-interactions/s, "
                    << 100.0 * pairs * 20.0 / kernel_s / 157.3e12 << " % of MI355X fp32 peak at 20 flop/pair)" << std::endl;
                {
                    const HipRunInfo& info = last_hip_run_info();
                    if (info.refine_tolerance > 0.0)
                        out << "Precision: mixed mode, per-body relative tolerance " << std::scientific << std::setprecision(1) << info.refine_tolerance
                            << std::fixed << std::setprecision(6) << ": " << info.refine_selected << " of " << n << " bodies listed by the selection rule, "
                            << info.refine_refined << " re-evaluated in fp64 (kernel time above includes them)" << std::endl;
                    else
                        out << "Precision: plain fp32 pair terms and sums (--refine 0, or the mixed mode does not apply to this kernel variant)" << std::endl;
                }
                if (distinct_devices < hip_ranks)
                    out << "  (" << hip_ranks << " ranks share " << distinct_devices << " device(s): the per-rank kernel time above is no per-GPU figure)" << std::endl;
                if (hip_ranks > 1) {
                    // the sharded row against the same evaluation on ONE GPU, on up to 1,024 evenly spaced bodies: the two differ
                    // by the association of fp32 partial sums only (each within the per-body tolerance of the fp64 result in the
                    // default mixed mode, so <= 2e-5 of each other; with --refine 0 an ill-conditioned row can reach ~1e-4); a stale
                    // or misplaced chunk shows as O(1), a chunk that is one step old as 1e-5..1e-3.  The bound follows the precision that
                    // ran: 10 x the per-body tolerance in mixed mode (1e-4 at the default 1e-5), 1e-3 in plain fp32; it is printed.
                    try {
                        const Forces single = brute_force_hip_single_gpu<D>(bodies, opt.devices[0]);
                        const int rows = std::min(n, 1024);
                        double worst = 0.0;
                        for (int r = 0; r < rows; ++r) {
                            const std::size_t i = static_cast<std::size_t>((static_cast<long long>(r) * (n - 1)) / std::max(rows - 1, 1));
                            double d2 = 0.0, f2 = 0.0;
                            for (int k = 0; k < D; ++k) {
                                d2 += (forces[i][k] - single[i][k]) * (forces[i][k] - single[i][k]);
                                f2 += single[i][k] * single[i][k];
                            }
                            if (f2 > 0.0) worst = std::max(worst, std::sqrt(d2 / f2));
                            else if (d2 > 0.0) worst = 1.0;
                        }
                        const double tol_in_force = hip_refine_tolerance();   // the precision both rows ran in
                        const double bound = tol_in_force > 0.0 ? std::min(1.0e-3, 10.0 * tol_in_force) : 1.0e-3;
                        const bool ok = worst <= bound;
                        out << "Sharded-vs-single-GPU check (" << rows << " sampled rows of " << hip_label << " against the 1-GPU row): max |dF|/|F| = "
                            << std::scientific << std::setprecision(3) << worst << " (bound " << std::setprecision(1) << bound << ": "
                            << (tol_in_force > 0.0 ? "mixed mode" : "plain fp32") << ")" << std::fixed << std::setprecision(6) << (ok ? "  ok" : "  MISMATCH") << std::endl;
                        if (!ok) g_exit_code = 3;
                    } catch (const std::exception& e) {
                        out << "Sharded-vs-single-GPU check: unavailable (" << e.what() << ")" << std::endl;
                    }
                    // the facts bench.py --gpus N carries in its JSON line (exchange_check, per_rank), from one more evaluation
                    // outside the timed row: both RCCL paths describe themselves the same way on first contact with a node
                    try {
                        const HipNodeReport rep = describe_hip_node<D>(bodies);
                        out << "exchange_check: transport " << rep.transport << ", mismatching_values " << rep.mismatching_values << " of "
                            << rep.checked_values_per_rank << " checked per rank" << (rep.mismatching_values ? "  EXCHANGE FAILED" : "  ok") << std::endl;
                        bool hidden = true;
                        for (const auto& k : rep.ranks) {
                            out << "per_rank: rank " << k.rank << " device " << k.device << " targets " << k.targets << std::fixed << std::setprecision(3)
                                << " local_ms " << k.local_ms << " remote_ms " << k.remote_ms << " exchange_ms " << k.exchange_ms
                                << " exchange_hidden " << (k.exchange_hidden ? "yes" : "no") << std::setprecision(6) << std::endl;
                            hidden = hidden && k.exchange_hidden;
                        }
                        out << "exchange_hidden_behind_local_pass: " << (hidden ? "yes" : "no") << ";  mixed mode: " << rep.refine_selected
                            << " listed, " << rep.refine_refined << " re-evaluated over all ranks" << std::endl;
                        if (rep.mismatching_values) g_exit_code = 3;
                    } catch (const std::exception& e) {
                        out << "exchange_check: unavailable (" << e.what() << ")" << std::endl;
                    }
                }
                hipcsv.open(base + "_hip.csv");
                hipcsv << "Method,Bodies,Dimension,Time(s),KernelTime(s),PairInteractionsPerSec,GPUs,DistinctDevices" << std::endl
                       << method.label << "," << n << "," << D << "," << std::fixed << std::setprecision(6) << seconds << ","
                       << kernel_s << "," << std::scientific << pairs / kernel_s << "," << hip_ranks << "," << distinct_devices << std::endl;
            }
            print_validation_forces<D>(forces, n, log);
            print_validation_forces<D>(forces, n, std::cout);
            if (!opt.dump.empty()) dump_raw(opt.dump + "_" + method.label + ".f64", forces);
        }
        out << std::endl;
    }

    // Near-field (leaf-pair direct sums) of the tree codes on the device: `-m p`.  The tree methods themselves are out of
    // scope; this row times the step they would hand to the GPU (FMM_Parlay<D>::p2p_phase, fmm_parlay.cpp:916-1022) on a
    // fixed-depth subdivision with ~64 bodies per leaf and 3^D neighbour lists.
    if (m.find('p') != std::string::npos) {
        int depth = 1;
        while (depth < 10 && static_cast<double>(n) / std::pow(2.0, depth * D) > 64.0) ++depth;
        out << "Near-field direct sums on HIP (FMM P2P law, uniform leaves of depth " << depth << "):" << std::endl;
        Forces forces;
        const LeafLists lists = build_uniform_leaves<D>(bodies, depth);   // host-side tree stand-in, not timed
        if (!opt.dump.empty()) {
            dump_raw(opt.dump + "_leaf_offsets.u32", lists.leaf_offsets);
            dump_raw(opt.dump + "_leaf_bodies.u32", lists.leaf_bodies);
            dump_raw(opt.dump + "_list_offsets.u32", lists.list_offsets);
            dump_raw(opt.dump + "_list_sources.u32", lists.list_sources);
        }
        const long long us = safely_execute(log, "NearField_HIP", [&] {
            forces = leaf_pair_direct_forces_hip<D>(bodies, lists, LeafLaw::FmmP2P);
            return 0;
        });
        if (us >= 0) {
            const double seconds = static_cast<double>(us) / 1e6;
            csv << "NearField_HIP," << n << "," << D;
            write_time(csv, seconds);
            if (opt.accuracy) csv << ",";
            csv << std::endl;
            double pairs = 0.0;
            for (std::size_t t = 0; t < lists.leaves(); ++t)
                for (std::uint32_t e = lists.list_offsets[t]; e < lists.list_offsets[t + 1]; ++e)
                    pairs += static_cast<double>(lists.leaf_offsets[t + 1] - lists.leaf_offsets[t]) *
                             static_cast<double>(lists.leaf_offsets[lists.list_sources[e] + 1] - lists.leaf_offsets[lists.list_sources[e]]);
            const double kernel_ms = last_leaf_pair_kernel_ms();
            auto rate = [&](double ms) {
                std::ostringstream o;
                o << ms << " ms = " << pairs / (ms * 1e-3) << " pair terms/s, " << 100.0 * pairs * 20.0 / (ms * 1e-3) / 157.3e12
                  << " % of MI355X fp32 peak at 20 flop/pair";
                return o.str();
            };
            out << "Time taken: " << seconds << " s  (one-shot call: validation, layout, copies, kernels; " << lists.leaves() << " leaves, " << pairs
                << " pair terms, kernel " << rate(kernel_ms) << "; one launch)" << std::endl;
            // The same sums the way a tree code asks for them: the structure made resident once (nbx_leaf_plan_*), the bodies on the
            // device, every evaluation = gather + pair kernel (bvh.cpp:143-176 / fmm_parlay.cpp:916-1022 evaluate a standing tree)
            try {
                const auto t0 = std::chrono::steady_clock::now();
                LeafPairSimulationHip<D> sim(bodies, lists);
                const auto t1 = std::chrono::steady_clock::now();
                const Forces planned = sim.forces(LeafLaw::FmmP2P, ::G);
                bool same = planned.size() == forces.size();
                for (std::size_t i = 0; same && i < planned.size(); ++i)
                    for (int k = 0; k < D; ++k) same = same && planned[i][k] == forces[i][k];
                double best = 1e30;
                for (int r = 0; r < 10; ++r) {
                    const auto a = std::chrono::steady_clock::now();
                    sim.evaluate(LeafLaw::FmmP2P, ::G);
                    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
                }
                const float single = sim.single_launch_ms(LeafLaw::FmmP2P, ::G);
                const float warm = sim.back_to_back_ms(LeafLaw::FmmP2P, 300);
                out << "Resident plan: created in " << std::chrono::duration<double, std::milli>(t1 - t0).count()
                    << " ms; an evaluation of the unchanged structure from resident bodies takes " << best << " ms wall (best of 10); forces "
                    << (same ? "equal the one-shot call's bit for bit" : "DIFFER from the one-shot call's") << std::endl
                    << "  pair kernel, single launch: " << rate(single) << std::endl
                    << "  pair kernel, mean of launches 151-300 back to back (clocks up): " << rate(warm) << std::endl;
                if (!same) g_exit_code = 3;
                csv << "NearField_HIP_plan," << n << "," << D;
                write_time(csv, best * 1e-3);
                if (opt.accuracy) csv << ",";
                csv << std::endl;
                if (opt.steps > 0) {
                    // a tree code whose far field is zero: k steps of {leaf sums; kick; drift} with the structure standing
                    std::vector<Body<D>> state = bodies;
                    const auto s0 = std::chrono::steady_clock::now();
                    sim.step(LeafLaw::FmmP2P, opt.G, opt.dt, opt.steps);
                    sim.synchronize();
                    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s0).count();
                    sim.download(state);
                    const double with_download = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s0).count();
                    out << "Near-field stepping on HIP: " << opt.steps << " steps of {leaf sums, kick, drift} in " << ms << " ms (" << ms / opt.steps
                        << " ms/step; " << with_download << " ms with the final state copied back), dt = " << opt.dt << ", G = " << opt.G << std::endl;
                    csv << "NearField_HIP_" << opt.steps << "steps," << n << "," << D;
                    write_time(csv, ms * 1e-3);
                    if (opt.accuracy) csv << ",";
                    csv << std::endl;
                    if (!opt.dump.empty()) dump_raw(opt.dump + "_NearField_steps.f64", state);
                }
            } catch (const std::exception& e) {
                out << "Resident plan: unavailable (" << e.what() << ")" << std::endl;
                g_exit_code = 3;
            }
            print_validation_forces<D>(forces, n, log);
            print_validation_forces<D>(forces, n, std::cout);
            if (!opt.dump.empty()) dump_raw(opt.dump + "_NearField_HIP.f64", forces);
        }
        out << std::endl;
    }

    if (opt.steps > 0 && run_hip) {
        out << "Leapfrog (kick-drift) on HIP: " << opt.steps << " steps, dt = " << opt.dt << ", G = " << opt.G;
        if (opt.softening > 0.0) out << ", softening = " << opt.softening;
        if (opt.law != "reference") out << ", law = " << opt.law;
        if (opt.integrator != "kd") out << ", integrator = " << opt.integrator;
        out << std::endl;
        std::vector<Body<D>> state = bodies;
        double kernel_s = 0.0;
        const long long us = safely_execute(log, "Leapfrog_HIP", [&] {
            HipSimulation<D> sim(bodies, opt.G, opt.softening, opt.law == "newton");
            double ke = 0.0, pe = 0.0, e0 = 0.0;
            const int chunk = opt.energy_every > 0 ? opt.energy_every : opt.steps;
            if (opt.energy_every > 0) {
                sim.energy(ke, pe);
                e0 = opt.have_e0 ? opt.e0 : ke + pe;   // a continued run keeps measuring against the energy it started with
                out << "step " << opt.step_offset << "  E = " << std::setprecision(12) << ke + pe;
                if (opt.have_e0) out << "  |dE/E0| = " << std::setprecision(3) << std::abs((ke + pe - e0) / e0) << std::setprecision(12);
                out << "  (kinetic " << ke << ", potential " << pe << ")" << std::endl;
            }
            for (int done = 0; done < opt.steps;) {
                const int k = std::min(chunk, opt.steps - done);
                if (opt.integrator == "kdk") sim.step_kdk(opt.dt, k); else sim.step(opt.dt, k);
                done += k;
                if (opt.energy_every > 0) {
                    sim.energy(ke, pe);
                    out << "step " << opt.step_offset + done << "  E = " << std::setprecision(12) << ke + pe << "  |dE/E0| = " << std::setprecision(3)
                        << std::abs((ke + pe - e0) / e0) << std::setprecision(6) << "  (kinetic " << ke << ", potential " << pe
                        << ", 2K/|U| " << 2.0 * ke / std::abs(pe) << ")" << std::endl;
                }
            }
            sim.download(state);
            kernel_s = sim.force_kernel_seconds();
            return 0;
        });
        if (us >= 0) {
            const double seconds = static_cast<double>(us) / 1e6;
            csv << "Leapfrog_HIP_" << opt.steps << "steps," << n << "," << D;
            write_time(csv, seconds);
            if (opt.accuracy) csv << ",";
            csv << std::endl;
            out << "Time taken: " << seconds << " s (" << seconds / opt.steps << " s/step; force kernels " << kernel_s << " s per rank)" << std::endl;
            if (!opt.dump.empty()) dump_raw(opt.dump + "_Leapfrog_HIP.f64", state);
            out << "Body #1 position: (";
            for (int d = 0; d < D; ++d) out << state[0].position[d] << (d < D - 1 ? ", " : "");
            out << ")" << std::endl;
        }
        out << std::endl;
    }
}

void usage(const char* argv0) {
    std::cout << "Usage: " << argv0 << " [options]" << std::endl
              << "Options:" << std::endl
              << "  -d, --dim <2|3>     Set simulation dimension (default: 3)" << std::endl
              << "  -N, --bodies <num>  Set number of bodies (default: 1000)" << std::endl
              << "  -a, --accuracy <0|1> Enable accuracy calculation (default: 0 - OFF)" << std::endl
              << "  -m, --methods <str> Specify which methods to run (default: all)" << std::endl
              << "                      a=bruteforce (CPU rows + HIP), g=HIP brute force only," << std::endl
              << "                      p=near-field (leaf-pair) direct sums of the tree codes on HIP," << std::endl
              << "                      b=barnes-hut, h=hilbert bvh, f=fmm (not built in this tier)" << std::endl
              << "      --seed <int>    Reproducible bodies (default: random_device, like the reference)" << std::endl
              << "      --init <uniform|plummer>  Initial condition (default: uniform)" << std::endl
              << "      --steps <k>     Also run k kick-drift steps on the device" << std::endl
              << "      --dt <t>        Time step for --steps (default: 1)" << std::endl
              << "      --G <value>     Coupling constant of the stepping loop (default: the reference's 4.471e-21)" << std::endl
              << "      --integrator <kd|kdk> kd = update_body_velocities then update_body_positions per step (default, first order);" << std::endl
              << "                      kdk = the same helpers as a synchronised kick-drift-kick leapfrog (extension, second order)" << std::endl
              << "      --law <reference|newton> Pair law of the stepping loop: the reference's r^-4 form (default) or the attractive" << std::endl
              << "                      softened Newtonian law (extension; needs --softening; Plummer velocities then use --G)" << std::endl
              << "      --softening <eps> Plummer softening of the stepping loop's pair law (extension; default 0 = the reference's law)" << std::endl
              << "      --energy-every <k> Log total energy and |dE/E0| every k steps (potential matching the selected law)" << std::endl
              << "      --refine <tol>  Per-body relative tolerance of the HIP rows: fp32 for all bodies + fp64 re-evaluation of those whose" << std::endl
              << "                      fp32 sum cannot be trusted to <tol> (default 1e-5: every body within 1e-5 of the sequential reference);" << std::endl
              << "                      0 = plain fp32" << std::endl
              << "      --gpus <g>      Shard the HIP rows over GPUs 0..g-1 of this node (one process, RCCL all-gather per step)" << std::endl
              << "      --devices <list> Same with an explicit device list, e.g. 0,0,0 = three virtual ranks on GPU 0" << std::endl
              << "      --device-count  Print the number of HIP devices and exit" << std::endl
              << "      --dump <prefix> Write bodies and every method's forces as raw doubles (<prefix>_<Method>.f64)" << std::endl
              << "      --load <file>   Read the N bodies from raw doubles (a <prefix>_Leapfrog_HIP.f64 or _bodies.f64 of --dump) instead of generating them" << std::endl
              << "      --step-offset <k>, --e0 <E>  Continue a run: the loaded state is step k, |dE/E0| refers to E" << std::endl
              << "  -h, --help          Display this help message" << std::endl;
}

}  // namespace

int main(int argc, char* argv[]) {
    // device memory shared between the ranks of --gpus G (RCCL) needs dmabuf IPC on this pool's hosts; kept if already set
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    Options opt;
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        const bool has_value = i + 1 < argc;
        if ((arg == "-d" || arg == "--dim") && has_value) {
            opt.dimension = std::stoi(argv[++i]);
            if (opt.dimension != 2 && opt.dimension != 3) {
                std::cerr << "Error: Dimension must be either 2 or 3" << std::endl;
                return 1;
            }
        } else if ((arg == "-N" || arg == "--bodies") && has_value) {
            opt.num_bodies = std::stoi(argv[++i]);
            if (opt.num_bodies <= 0) {
                std::cerr << "Error: Number of bodies must be positive" << std::endl;
                return 1;
            }
        } else if ((arg == "-a" || arg == "--accuracy") && has_value) {
            opt.accuracy = std::stoi(argv[++i]) == 1;
        } else if ((arg == "-m" || arg == "--methods") && has_value) {
            opt.methods = argv[++i];
            opt.override_bf_limit = opt.methods == "a";
            for (char c : opt.methods)
                if (std::string("abhfgp").find(c) == std::string::npos) {
                    std::cerr << "Error: Invalid method '" << c << "'" << std::endl
                              << "Valid methods: a=bruteforce, g=hip bruteforce, p=hip near-field sums, b=barnes-hut, h=bvh, f=fmm" << std::endl;
                    return 1;
                }
        } else if (arg == "--seed" && has_value) {
            opt.seed = std::stoll(argv[++i]);
        } else if (arg == "--steps" && has_value) {
            opt.steps = std::stoi(argv[++i]);
        } else if (arg == "--dt" && has_value) {
            opt.dt = std::stod(argv[++i]);
        } else if (arg == "--G" && has_value) {
            opt.G = std::stod(argv[++i]);
        } else if (arg == "--energy-every" && has_value) {
            opt.energy_every = std::stoi(argv[++i]);
        } else if (arg == "--softening" && has_value) {
            opt.softening = std::stod(argv[++i]);
        } else if (arg == "--integrator" && has_value) {
            opt.integrator = argv[++i];
            if (opt.integrator != "kd" && opt.integrator != "kdk") {
                std::cerr << "Error: --integrator must be kd or kdk" << std::endl;
                return 1;
            }
        } else if (arg == "--law" && has_value) {
            opt.law = argv[++i];
            if (opt.law != "reference" && opt.law != "newton") {
                std::cerr << "Error: --law must be reference or newton" << std::endl;
                return 1;
            }
        } else if (arg == "--refine" && has_value) {
            opt.refine = std::stod(argv[++i]);
            if (!(opt.refine == 0.0 || (opt.refine >= 1.0e-7 && opt.refine <= 1.0e-2))) {
                std::cerr << "Error: --refine must be 0 (plain fp32) or in [1e-7, 1e-2]" << std::endl;
                return 1;
            }
        } else if (arg == "--gpus" && has_value) {
            const int g = std::stoi(argv[++i]);
            if (g < 1 || g > 64) {
                std::cerr << "Error: --gpus must be in [1,64]" << std::endl;
                return 1;
            }
            opt.devices.clear();
            for (int d = 0; d < g; ++d) opt.devices.push_back(d);
        } else if (arg == "--devices" && has_value) {   // e.g. --devices 0,0,0 = three virtual ranks on GPU 0
            opt.devices.clear();
            std::string list = argv[++i];
            std::size_t pos = 0;
            while (pos <= list.size()) {
                const std::size_t comma = list.find(',', pos);
                opt.devices.push_back(std::stoi(list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos)));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
        } else if (arg == "--dump" && has_value) {
            opt.dump = argv[++i];
        } else if (arg == "--load" && has_value) {
            opt.load = argv[++i];
        } else if (arg == "--step-offset" && has_value) {
            opt.step_offset = std::stoi(argv[++i]);
        } else if (arg == "--e0" && has_value) {
            opt.e0 = std::stod(argv[++i]);
            opt.have_e0 = true;
        } else if (arg == "--init" && has_value) {
            opt.init = argv[++i];
            if (opt.init != "uniform" && opt.init != "plummer") {
                std::cerr << "Error: --init must be uniform or plummer" << std::endl;
                return 1;
            }
        } else if (arg == "--device-count") {   // for scripts: how many HIP devices this node offers
            std::cout << hip_device_count() << std::endl;
            return 0;
        } else if (arg == "-h" || arg == "--help") {
            usage(argv[0]);
            return 0;
        }
    }

    set_hip_devices(opt.devices);
    if (opt.refine >= 0.0) set_hip_refine(opt.refine);
    if (opt.methods.empty() || opt.methods.find_first_of("agp") != std::string::npos)
        warm_up_hip();  // device start-up stays out of the timed rows; a missing GPU surfaces in the HIP row itself
    const std::string run_id = get_run_id();
    try {
        if (opt.dimension == 2) {
            auto bodies = !opt.load.empty() ? load_bodies<2>(opt.load, opt.num_bodies)
                          : opt.init == "plummer" ? generate_plummer_bodies<2>(opt.num_bodies, opt.seed, 1.0e5, 1.0e12, opt.law == "newton" ? opt.G : ::G)
                                                  : generate_random_bodies<2>(opt.num_bodies, opt.seed);
            run_benchmark<2>(bodies, run_id, opt);
        } else {
            auto bodies = !opt.load.empty() ? load_bodies<3>(opt.load, opt.num_bodies)
                          : opt.init == "plummer" ? generate_plummer_bodies<3>(opt.num_bodies, opt.seed, 1.0e5, 1.0e12, opt.law == "newton" ? opt.G : ::G)
                                                  : generate_random_bodies<3>(opt.num_bodies, opt.seed);
            run_benchmark<3>(bodies, run_id, opt);
        }
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    } catch (...) {
        std::cerr << "Unknown error occurred" << std::endl;
        return 1;
    }
    release_hip_caches();
    return g_exit_code;
}
