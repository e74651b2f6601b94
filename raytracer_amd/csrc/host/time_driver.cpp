// Timing harness with the interface and output format of tests/raytracer/time_driver.cpp:24-66:
//   time_driver [scene.json] [camera_id] [trials] [timing.txt]
// The output path is an addition (the reference always writes tests/raytracer/timing.txt).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "raytracer.h"

static double mean_of(const std::vector<double>& v) {
    double s = 0.0;
    for (double x : v) s += x;
    return v.empty() ? 0.0 : s / static_cast<double>(v.size());
}

static double sample_std(const std::vector<double>& v, double mu) {
    if (v.size() < 2) return 0.0;
    double ss = 0.0;
    for (double x : v) ss += (x - mu) * (x - mu);
    return std::sqrt(ss / static_cast<double>(v.size() - 1));
}

int main(int argc, char** argv) {
    const std::string scene = (argc > 1) ? argv[1] : "s2899221/ASCII/50_rand_1_cam.json";
    const std::string camera = (argc > 2) ? argv[2] : "1";
    const int trials = (argc > 3) ? std::max(1, std::atoi(argv[3])) : 10;
    const std::string out_path = (argc > 4) ? argv[4] : "tests/raytracer/timing.txt";
    try {
        std::vector<double> plain, accel;
        for (int i = 0; i < trials; ++i) plain.push_back(time_trace_all_one_pass(scene, camera));
        for (int i = 0; i < trials; ++i) accel.push_back(time_trace_all_accelerated_one_pass(scene, camera));
        const double mu_p = mean_of(plain), sd_p = sample_std(plain, mu_p), mu_a = mean_of(accel), sd_a = sample_std(accel, mu_a);
        std::ofstream out(out_path);
        if (!out) { std::cerr << "Failed to open output file " << out_path << "\n"; return 1; }
        out << "unaccelerated_times\n";
        for (double t : plain) out << t << "\n";
        out << "accelerated_times\n";
        for (double t : accel) out << t << "\n";
        out << "mean_unaccelerated\n" << mu_p << "\nstd_unaccelerated\n" << sd_p << "\n";
        out << "mean_accelerated\n" << mu_a << "\nstd_accelerated\n" << sd_a << "\n";
        out.close();
        std::cout << "Wrote " << out_path << "\n";
        std::cout << "Mean (unaccelerated): " << mu_p << ", std: " << sd_p << "\n";
        std::cout << "Mean (accelerated):   " << mu_a << ", std: " << sd_a << "\n";
    } catch (const std::exception& e) {
        std::cerr << "time_driver: " << e.what() << "\n";
        return 2;
    }
    return 0;
}
