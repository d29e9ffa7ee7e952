#include "FileHandler.h"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <sys/stat.h>

static bool make_dirs(const std::string &path)
{
    std::string cur;
    for (size_t i = 0; i <= path.size(); i++) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) {
                struct stat st;
                if (stat(cur.c_str(), &st) != 0 && mkdir(cur.c_str(), 0777) != 0) return false;
            }
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
    return true;
}

bool FileHandler::SaveTrajecInformation(const std::vector<MatrixXd> &A, const std::vector<MatrixXd> &B,
                                        const std::vector<MatrixXd> &states, const std::vector<MatrixXd> &controls,
                                        const std::string &root)
{
    if (A.empty() || B.empty()) return false;
    if (!make_dirs(root)) { std::cerr << "Failed to create directory: " << root << std::endl; return false; }
    const int horizon = (int)A.size(), dof = A[0].rows() / 2, num_ctrl = B[0].cols();
    std::ofstream out;
    out.open(root + "/A_matrices.csv");
    for (int i = 0; i < horizon - 1; i++) {
        for (int j = 0; j < 2 * dof; j++) for (int k = 0; k < 2 * dof; k++) out << A[i](j, k) << ",";
        out << std::endl;
    }
    out.close();
    out.open(root + "/B_matrices.csv");
    for (int i = 0; i < horizon - 1; i++) {
        for (int j = 0; j < 2 * dof; j++) for (int k = 0; k < num_ctrl; k++) out << B[i](j, k) << ",";
        out << std::endl;
    }
    out.close();
    out.open(root + "/states.csv");
    for (int i = 0; i < horizon - 1; i++) {
        for (int j = 0; j < 2 * dof; j++) out << states[i](j) << ",";
        out << std::endl;
    }
    out.close();
    out.open(root + "/controls.csv");
    for (int i = 0; i < horizon - 1; i++) {
        for (int j = 0; j < num_ctrl; j++) out << controls[i](j) << ",";
        out << std::endl;
    }
    out.close();
    return true;
}

bool FileHandler::SaveKeypointsToFile(const std::string &root, const std::vector<std::vector<int>> &keypoints)
{
    if (keypoints.empty()) return false;
    if (!make_dirs(root)) { std::cerr << "Failed to create directory: " << root << std::endl; return false; }
    std::ofstream out(root + "/keypoints.csv");
    const int dof = (int)keypoints[0].size();
    for (int i = 0; i < dof; i++) {
        for (size_t j = 0; j < keypoints.size(); j++)
            for (size_t k = 0; k < keypoints[j].size(); k++)
                if (keypoints[j][k] == i) { out << j << ","; break; }
        out << std::endl;
    }
    return true;
}

bool FileHandler::SaveTaskToFile(const std::string &filename, const std::vector<double> &start, const std::vector<double> &targets)
{
    const size_t slash = filename.rfind('/');
    if (slash != std::string::npos && !make_dirs(filename.substr(0, slash))) return false;
    std::ofstream out(filename);
    if (!out) return false;
    for (double v : start) out << v << ",";
    for (double v : targets) out << v << ",";
    out << std::endl;
    return true;
}

bool FileHandler::LoadTaskFromFile(const std::string &filename, int n_start, int n_targets, std::vector<double> &start,
                                   std::vector<double> &targets)
{
    std::ifstream fin(filename);
    if (!fin) { std::cerr << "File " << filename << " does not exist\n"; return false; }
    std::string temp;
    bool any = false;
    while (fin >> temp) {                        // one row; whitespace-free, comma separated (:509-519)
        std::string rest;
        std::getline(fin, rest);
        std::vector<std::string> row;
        std::stringstream s(temp);
        std::string word;
        while (std::getline(s, word, ',')) row.push_back(word);
        if ((int)row.size() != n_start + n_targets) {
            std::cerr << "CSV file has " << row.size() << "elements, num dofs is: " << n_start
                      << "and resids target size is: " << n_targets << "\n";
            return false;
        }
        start.assign(n_start, 0.0); targets.assign(n_targets, 0.0);
        for (int i = 0; i < n_start; i++) start[i] = std::stod(row[i]);
        for (int i = 0; i < n_targets; i++) targets[i] = std::stod(row[n_start + i]);
        any = true;
    }
    return any;
}

int FileHandler::IntAccumulate(const std::vector<double> &v)
{
    int acc = 0;                                  // std::accumulate(first, last, 0): the init type is int
    for (double x : v) acc = (int)(acc + x);
    return acc;
}

bool FileHandler::SaveSummary(const std::string &filename, const std::vector<SummaryRow> &rows)
{
    std::ofstream out(filename);
    if (!out) return false;
    out << "Cost reduction" << "," << "Optimisation time (ms)" << "," << "Number iterations" << ",";
    out << "Average num dofs" << "," << "Average percent derivs" << "," << "Average time derivs (ms)" << ",";
    out << "Average time BP (ms)" << "," << "Average time FP (ms)" << std::endl;
    for (const SummaryRow &r : rows) {
        out << r.cost_reduction << "," << r.optimisation_time_ms << "," << r.num_iterations << ",";
        out << r.avg_num_dofs << "," << r.avg_percent_derivs << "," << IntAccumulate(r.time_derivs_ms) << ",";
        out << IntAccumulate(r.time_bp_ms) << "," << IntAccumulate(r.time_fp_ms) << std::endl;
    }
    return true;
}
